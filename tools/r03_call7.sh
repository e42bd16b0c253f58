# round 3, GPU call 7: exchange with deferred host issue + fused pass (+ NT), top-k v2 with the asynchronous threshold exchange
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 200 python tools/shard_bench.py --world 8 --sync --epochs 20 --windows 1 > $o/shard_bench_sync2.txt 2>&1; grep exchange $o/shard_bench_sync2.txt
HEAT_CF_SYNC_NT=1 timeout -k 10 200 python tools/shard_bench.py --world 8 --sync --epochs 20 --windows 1 > $o/shard_bench_sync2_nt.txt 2>&1; echo NT; grep exchange $o/shard_bench_sync2_nt.txt
for ab in 0 1; do
  HEAT_CF_TOPK_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_c$ab -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/topk_c$ab.txt 2>&1
  echo "ablate=$ab"; find $o/prof_topk_c$ab -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-160 | tail -1
done
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "topk or item_sync or side_stream or eight_user or two_ranks or accl_hogwild" > $o/pytest_gpu_3.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_3.txt
