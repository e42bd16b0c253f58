# round 3, GPU call 4: the 128-user top-k kernel — ids, then time against the 64-user kernel
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "topk or evaluate0" > $o/pytest_topk.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $o/pytest_topk.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_v2.txt 2>&1; cat $o/topk_v2.txt
HEAT_CF_TOPK_KERNEL=v1 timeout -k 10 200 python tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_v1.txt 2>&1; cat $o/topk_v1.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk2 -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_v2_prof.txt 2>&1
find $o/prof_topk2 -name "*kernel_stats.csv" -exec head -6 {} \;
# user-run aligned stream boundaries for the shards of an 8-GPU job: does the Recall bound on the slice length move?
for cap in 0 128; do
  HEAT_CF_ALIGN_CAP=$cap timeout -k 10 250 python tests/tools/sim_shards.py --world 8 --windows 1 --overlap 1 --streams 1162,2048,3017 --seeds 2022,7 --oracle-runs 0 > $o/sim_shards_align$cap.txt 2>&1; echo "align $cap rc=$?"
  grep -h "SHARDED\|SINGLE\|Recall" $o/sim_shards_align$cap.txt
  HEAT_CF_ALIGN_CAP=$cap timeout -k 10 100 python tools/shard_bench.py --world 8 --streams 1162,2048,3017 > $o/shard_bench_align$cap.txt 2>&1; cat $o/shard_bench_align$cap.txt
done
