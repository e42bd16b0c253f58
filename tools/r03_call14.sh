set -u
cd "$(dirname "$0")/.." 2>/dev/null || true
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "accl or aggregator or topk or evaluate0 or item_sync or side_stream" > $o/pytest_gpu_7.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_7.txt
timeout -k 10 100 python tools/quick_bench.py --shape amazonbooks --agg --epochs 5 2>&1 | grep kernel=
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_g -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_g.txt 2>&1; find $o/prof_topk_g -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-160 | grep fused
