set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/nodebug.txt
: > $o
for lib in "$PWD/heat_amd/lib/libheat_cf.so" "$PWD/heat_amd/lib/exp/libheat_cf_NODBG.so"; do
  echo "## lib=$lib" >> $o
  HEAT_CF_LIB=$lib timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --agg --epochs 3 --streams 0 >> $o 2>&1 || exit 1
  HEAT_CF_LIB=$lib timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --epochs 5 --streams 0 >> $o 2>&1 || exit 1
done
grep "^##\|coherence" $o | cut -c1-150
