# round 3, GPU call 2: ACCL vs stream count (to set beside the CPU stream-order model), tile sampler at matched worker count,
# bench line + the self-launched 2-rank gloo rehearsal
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 300 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --agg --streams 8,80,160,438 --oracle-threads "" --seeds 2022 > $o/accl_streams.txt 2>&1; echo "accl rc=$?"
timeout -k 10 300 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --agg --streams 0 --oracle-threads 8 --seeds 7,99 > $o/accl_seeds.txt 2>&1; echo "accl2 rc=$?"
timeout -k 10 300 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --tile --streams 64,0 --oracle-threads 64,8 --seeds 2022,7,99 > $o/tile_matched_workers.txt 2>&1; echo "tile rc=$?"
timeout -k 10 300 python bench.py > $o/bench_n1_a.json 2> $o/bench_a.err; echo "bench rc=$?"
HEAT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $o/bench_n2_gloo_selflaunch.json 2> $o/bench_n2.err; echo "bench n2 rc=$?"
grep -h "GPU seed\|ORACLE\|Recall" $o/accl_streams.txt $o/accl_seeds.txt $o/tile_matched_workers.txt
cat $o/bench_n2_gloo_selflaunch.json | head -c 1500; tail -5 $o/bench_n2.err
python -c "
import json; d=json.load(open('$o/bench_n1_a.json')); print(d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['scaling'], d['accl']['value']/1e6, d['tile_sampler'])"
