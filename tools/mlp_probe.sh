# Is the AmazonBooks kernel limited by memory-level parallelism?  More resident streams (beyond the parity-validated cap:
# performance exploration only) with the 3- and 4-waves/SIMD builds.
mkdir -p gpurun_out/mlp
for v in base occ4; do for st in 0 3072 4096 6144; do
  HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_$v.so timeout -k 10 100 python tools/quick_bench.py --shape amazonbooks --epochs 5 --streams $st --update 4 2>/dev/null | grep coherence | sed "s/^/$v /" >> gpurun_out/mlp/mlp.txt
done; done
cat gpurun_out/mlp/mlp.txt
