# Round-end evidence run (GPU box): the bench line, rocprofv3 kernel stats and the two PMC passes for the same command,
# the HBM-resident shape likewise, other shapes.  Outputs under gpurun_out/final/ ; copy what is judged into profiles/.
set -u
cd "$(dirname "$0")/.." && rm -rf gpurun_out/final && mkdir -p gpurun_out/final && export TMPDIR=/tmp   # a fresh directory: collect_profiles.sh must never find an earlier run's files
o=gpurun_out/final
timeout -k 10 300 python bench.py > $o/bench_n1.json 2> $o/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $o/bench_under_rocprof.json 2> $o/prof.err; echo "prof rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/pmc_fetch.json 2> $o/pmc_fetch.err; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/pmc_write.json 2> $o/pmc_write.err; echo "write rc=$?"
H="--shape synthetic_hbm --steps 2 --warmup 1 --interactions 4000000"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_hbm -- python3 bench.py $H > $o/bench_hbm_under_rocprof.json 2> $o/prof_hbm.err; echo "prof hbm rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch_hbm -- python3 bench.py $H > $o/pmc_fetch_hbm.json 2> $o/pmc_fetch_hbm.err; echo "fetch hbm rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write_hbm -- python3 bench.py $H > $o/pmc_write_hbm.json 2> $o/pmc_write_hbm.err; echo "write hbm rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_yelp18 -- python3 bench.py --shape yelp18 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $o/bench_yelp18.json 2> $o/prof_yelp18.err; echo "prof yelp rc=$?"
Y="--shape yelp18 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch_yelp18 -- python3 bench.py $Y > $o/pmc_fetch_yelp18.json 2> $o/pmc_fetch_yelp18.err; echo "fetch yelp rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write_yelp18 -- python3 bench.py $Y > $o/pmc_write_yelp18.json 2> $o/pmc_write_yelp18.err; echo "write yelp rc=$?"
for s in gowalla gowalla_pr1; do timeout -k 10 200 python bench.py --shape $s --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $o/bench_$s.json 2>> $o/bench.err; done
HEAT_BENCH_FORCE_SYNC=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $o/bench_forcesync.json 2>> $o/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_under_rocprof.txt 2> $o/prof_topk.err; echo "topk rc=$?"
# behaviour aggregation (ACCL) kernel alone: rocprofv3 stats + the two PMC passes
A="--shape amazonbooks --agg --epochs 3"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_accl -- python3 tools/quick_bench.py $A > $o/accl_under_rocprof.txt 2> $o/prof_accl.err; echo "prof accl rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch_accl -- python3 tools/quick_bench.py $A > $o/pmc_fetch_accl.txt 2> $o/pmc_fetch_accl.err; echo "fetch accl rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/pmc_write_accl -- python3 tools/quick_bench.py $A > $o/pmc_write_accl.txt 2> $o/pmc_write_accl.err; echo "write accl rc=$?"
timeout -k 10 200 python tools/shard_bench.py --world 8 --sync --epochs 20 --windows 1 > $o/shard_bench_exchange_modes.txt 2>&1; grep exchange $o/shard_bench_exchange_modes.txt
for f in $o/bench_n1.json $o/bench_gowalla.json $o/bench_gowalla_pr1.json $o/bench_yelp18.json $o/bench_hbm_under_rocprof.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],2),'ms frac', round(d['roofline']['frac'],3), d['config']['kernel'])"; done
find $o -name "*kernel_stats.csv" -exec head -3 {} \;
