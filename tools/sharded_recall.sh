# Evidence run (GPU box): Recall@20 of the reference-shaped CLI at AmazonBooks shape, one rank vs 2 and 4 ranks sharing
# this GPU through gloo (user-sharded, item table all-reduced once per epoch, `sum` rule).
mkdir -p gpurun_out/shard
CFG=heat_amd/cf/benchmarks/AmazonBooks/MF_CCL/configs/config0.yaml
for N in 1 2 4; do
  if [ $N -eq 1 ]; then B=nccl; else B=gloo; fi
  HEAT_CF_DIST_BACKEND=$B timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 \
      --master-port $((29600 + N)) -m heat_amd.cf.main --config $CFG --synthetic amazonbooks --distributed \
      > gpurun_out/shard/n$N.txt 2>&1 || { echo "N=$N failed"; tail -5 gpurun_out/shard/n$N.txt; exit 1; }
  echo "== ranks=$N backend=$B"; grep -h "^epoch:\|Metrics" gpurun_out/shard/n$N.txt
done
