set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python tests/tools/sim_shards.py --shape amazonbooks --epochs 5 --world 8 --streams 0,1024,3017 --windows 1,2 --overlap 0,1 --seeds 2022 > gpurun_out/b4_sim_shards.txt 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/b4_bench_n1.json 2> gpurun_out/b4_bench_n1.err
HEAT_BENCH_FORCE_SYNC=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/b4_bench_forcesync.json 2> gpurun_out/b4_bench_forcesync.err
python bench.py --shape yelp18 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/b4_bench_yelp18.json 2> gpurun_out/b4_bench_yelp18.err
echo done
