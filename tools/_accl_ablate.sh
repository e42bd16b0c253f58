set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/accl_ablate.txt
: > $o
echo "## product library" >> $o
timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --agg --epochs 3 --streams 0,768,1024 >> $o 2>&1 || exit 1
for v in NO_W0_ATOMIC NO_W0_REFRESH NO_HIS; do
  echo "## $v" >> $o
  HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_$v.so timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --agg --epochs 3 --streams 0,768,1024 >> $o 2>&1 || exit 1
done
grep -v "^shape" $o | cut -c1-150
mkdir -p gpurun_out/sq && bash tools/sq_counters.sh amazonbooks --agg > gpurun_out/sq/accl_summary.txt 2>&1; cat gpurun_out/sq/accl_summary.txt
