mkdir -p gpurun_out/x && cd "$(dirname "$0")/.."
timeout -k 10 200 python tools/quick_bench.py --shape yelp18 --epochs 5 --update 44,76 --streams 0,256 > gpurun_out/x/y.log 2>&1; grep coherence gpurun_out/x/y.log | cut -c1-210
timeout -k 10 200 python tools/quick_bench.py --shape gowalla --epochs 5 --update 44,76 > gpurun_out/x/g.log 2>&1; grep coherence gpurun_out/x/g.log | cut -c1-210
