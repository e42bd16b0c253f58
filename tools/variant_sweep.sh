# Kernel sizing sweep (BASELINE.json configs[2]: Yelp18 d=128 negs=64; also the synthetic-HBM shape): register groups per
# wave x waves per workgroup, both update policies.  Run on the GPU box; output under gpurun_out/sweep/.
mkdir -p gpurun_out/sweep
out=gpurun_out/sweep/variant_sweep.txt
: > $out
run() { # shape variant mode steps
  HEAT_CF_VARIANT=$2 timeout -k 10 200 python bench.py --shape $1 --steps $4 --warmup 1 --no-cpu-baseline --update-mode $3 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read().strip()
if not t: print('$1 variant=$2 mode=$3: no variant / failed'); raise SystemExit
d = json.loads(t); print('$1 variant=$2 mode=$3', d['config']['kernel'], round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M/s frac', round(d['roofline']['frac'], 3))" >> $out
}
for v in 32,1 16,2 8,4 16,4; do for m in 4 3; do run yelp18 $v $m 4; done; done
for v in 25,4 13,8 16,8; do run synthetic_hbm $v 4 1; done
cat $out
