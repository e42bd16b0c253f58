# Round-end evidence run (GPU box): gpu tests, the bench line, rocprofv3 kernel stats and the two PMC passes for the same
# command, other shapes.  Outputs under gpurun_out/final/ ; copy what is judged into profiles/.
mkdir -p gpurun_out/final && cd "$(dirname "$0")/.." && export TMPDIR=/tmp
o=gpurun_out/final
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $o/pytest_gpu.txt; tail -3 $o/pytest_gpu.txt
timeout -k 10 200 python bench.py > $o/bench_n1.json 2> $o/bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --update-mode 3 --no-cpu-baseline > $o/bench_n1_atomic_wg.json 2>> $o/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_under_rocprof.json 2> $o/prof.err; echo "prof rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/pmc_fetch.json 2> $o/pmc_fetch.err; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $o/pmc_write.json 2> $o/pmc_write.err; echo "write rc=$?"
for s in gowalla yelp18; do timeout -k 10 200 python bench.py --shape $s --steps 5 --warmup 1 --no-cpu-baseline > $o/bench_$s.json 2>> $o/bench.err; done
timeout -k 10 300 python bench.py --shape synthetic_hbm --steps 2 --warmup 1 > $o/bench_synthetic_hbm.json 2>> $o/bench.err
for f in $o/bench_n1.json $o/bench_gowalla.json $o/bench_yelp18.json $o/bench_synthetic_hbm.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],2),'ms frac', round(d['roofline']['frac'],3), d['config']['kernel'])"; done
find $o -name "*kernel_stats.csv" -exec head -3 {} \;
