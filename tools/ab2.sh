# A/B: default build (Philox x4 batching) vs permlane-swap reductions; correctness first (gpu tests), then timing.
mkdir -p gpurun_out/ab2
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -4
HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_pl.so timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "serial_walk or collision or randomized or aggregator or sampler" 2>&1 | tail -4
for r in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],3), round(d['value']/1e6,1), round(d['roofline']['frac'],3))"
HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_pl.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('permlane', round(d['ms_per_step'],3), round(d['value']/1e6,1), round(d['roofline']['frac'],3))"
done
