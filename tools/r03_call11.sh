# round 3, GPU call 11: final top-k dispatch (timing + split model), then the whole -m gpu suite
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
for z in 0 1 2 3 6; do
  if [ $z -eq 0 ]; then unset HEAT_CF_TOPK_SPLITS; else export HEAT_CF_TOPK_SPLITS=$z; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_f$z -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_f$z.txt 2>&1
  echo "splits=$z"; find $o/prof_topk_f$z -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-160 | grep fused
done
unset HEAT_CF_TOPK_SPLITS
timeout -k 10 880 python -m pytest tests -m gpu -x -q -rP > $o/pytest_gpu_5.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_gpu_5.txt
