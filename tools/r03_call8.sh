# round 3, GPU call 8: top-k 128-user kernel with two workgroups per CU; exchange every 2 epochs (Recall)
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
for w in 1 2; do for ab in 0 1; do
  HEAT_CF_TOPK_WGS=$w HEAT_CF_TOPK_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_w${w}a$ab -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/topk_w${w}a$ab.txt 2>&1
  echo "wgs=$w ablate=$ab"; find $o/prof_topk_w${w}a$ab -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-160 | tail -1
done; done
HEAT_CF_TOPK_WGS=2 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "topk" > $o/pytest_topk_w2.txt 2>&1; echo "pytest topk wgs2 rc=$?"; tail -2 $o/pytest_topk_w2.txt
for ee in 1 2; do
  timeout -k 10 250 python tests/tools/sim_shards.py --world 8 --windows 1 --overlap 1 --streams 0 --seeds 2022,7,99 --oracle-runs 0 --exchange-every $ee > $o/sim_shards_every$ee.txt 2>&1; echo "every $ee rc=$?"
  grep -h "SHARDED\|SINGLE\|Recall" $o/sim_shards_every$ee.txt
done
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "accl_hogwild or item_sync or side_stream or eight_user or two_ranks" > $o/pytest_gpu_4.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_4.txt
