set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/wide_ring_ablate.txt
: > $o
for lib in libheat_cf.so exp/libheat_cf_nomult.so exp/libheat_cf_hash.so exp/libheat_cf_both.so; do
  echo "## lib=$lib" >> $o
  HEAT_CF_LIB=$PWD/heat_amd/lib/$lib timeout -k 10 200 python tools/quick_bench.py --shape yelp18 --epochs 4 --streams 0 >> $o 2>&1 || exit 1
done
grep "^##\|coherence" $o | cut -c1-150
