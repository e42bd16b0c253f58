# A/B of experimental builds of the AmazonBooks kernel variant (development aid; run on the GPU box)
mkdir -p gpurun_out/ab
for round in 1 2; do
for v in base occ4 occ5 gpf2 gpf8 nosb; do
  HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_$v.so timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v round $round', round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e6, 1), 'M/s', round(d['roofline']['frac'], 3))" >> gpurun_out/ab/ab.txt
done
done
cat gpurun_out/ab/ab.txt
