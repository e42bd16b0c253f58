set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
HEAT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 > $o/bench_n2_gloo_b.json 2> $o/bench_n2_b.err; echo "bench n2 rc=$?"; head -c 600 $o/bench_n2_gloo_b.json; echo; tail -3 $o/bench_n2_b.err | cut -c1-200
HEAT_BENCH_FORCE_SYNC=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $o/bench_forcesync_b.json 2>> $o/bench_n2_b.err; echo "forcesync rc=$?"; python -c "
import json; d=json.load(open('$o/bench_forcesync_b.json')); print(d['value']/1e6, d['ms_per_step'], d['config']['item_sync'], d.get('item_sync_blocking',{}).get('ms_per_step'))"
timeout -k 10 880 python -m pytest tests -m gpu -x -q -rP > $o/pytest_gpu_8.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_gpu_8.txt
