set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "topk or evaluate0" > gpurun_out/b5_pytest_topk.txt 2>&1 || { tail -40 gpurun_out/b5_pytest_topk.txt; echo TOPK_FAIL; }
python tools/eval_bench.py > gpurun_out/b5_eval_bench.txt 2>&1 || true
for V in auto 2,2 1,4; do
  if [ $V = auto ]; then unset HEAT_CF_VARIANT; else export HEAT_CF_VARIANT=$V; fi
  python tools/shard_bench.py --streams 512,1024,1162 >> gpurun_out/b5_shard_bench.txt 2>&1
done
unset HEAT_CF_VARIANT
HEAT_BENCH_FORCE_SYNC=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/b5_bench_forcesync.json 2> gpurun_out/b5_bench_forcesync.err || true
python bench.py --shape yelp18 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/b5_bench_yelp18.json 2> gpurun_out/b5_bench_yelp18.err || true
python bench.py --shape synthetic_hbm --steps 2 --warmup 1 --interactions 4000000 > gpurun_out/b5_bench_hbm.json 2> gpurun_out/b5_bench_hbm.err || true
echo done
