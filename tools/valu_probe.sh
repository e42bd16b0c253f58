# How much of the AmazonBooks epoch is VALU-side work?  Development builds with the duplicate scan removed and/or the
# Philox rounds cut to 1 (results are WRONG in these builds; timing only).
mkdir -p gpurun_out/valu
for r in 1 2; do for v in base nodup cheaprng both; do
  HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_$v.so timeout -k 10 100 python tools/quick_bench.py --shape amazonbooks --epochs 5 --update 4 2>/dev/null | grep coherence | cut -c1-150 | sed "s/^/$v /" >> gpurun_out/valu/valu.txt
done; done
cat gpurun_out/valu/valu.txt
