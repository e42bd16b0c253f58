# round 3, GPU call 5: pre-sweep wait in the single-wave training kernels + wave-local queue selection in the 128-user top-k
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 200 python tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_v2b.txt 2>&1; cat $o/topk_v2b.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $o/bench_n1_b.json 2> $o/bench_b.err; python -c "
import json; d=json.load(open('$o/bench_n1_b.json')); print('headline', d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'])"
timeout -k 10 100 python tools/shard_bench.py --world 8 --streams 1162,2048 > $o/shard_bench_b.txt 2>&1; cat $o/shard_bench_b.txt
for s in gowalla_pr1 yelp18; do timeout -k 10 200 python bench.py --shape $s --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/bench_${s}_b.json 2>> $o/bench_b.err; python -c "
import json; d=json.load(open('$o/bench_${s}_b.json')); print('$s', d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['config']['kernel'])"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk3 -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_v2b_prof.txt 2>&1
find $o/prof_topk3 -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-200
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $o/pytest_gpu_2.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest_gpu_2.txt
