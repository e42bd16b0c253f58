# Evidence run (GPU box): 4 ranks sharing this GPU through gloo, AmazonBooks shape — final-epoch loss and Recall@20 as a
# function of the item-sync window (yaml extension key sync_interactions; 0 = once per epoch at this shape).
mkdir -p gpurun_out/window
CFG=heat_amd/cf/benchmarks/AmazonBooks/MF_CCL/configs/config0.yaml
for W in 0 148796 37199; do
python3 - <<PY
import yaml
c = yaml.safe_load(open("$CFG"))
c["model_config"]["sync_interactions"] = $W
open("gpurun_out/window/w$W.yaml", "w").write(yaml.safe_dump(c))
PY
  HEAT_CF_DIST_BACKEND=gloo timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 \
      --master-port $((29700 + W % 97)) -m heat_amd.cf.main --config gpurun_out/window/w$W.yaml --synthetic amazonbooks --distributed \
      > gpurun_out/window/w$W.txt 2>&1 || { echo "W=$W failed"; tail -5 gpurun_out/window/w$W.txt; exit 1; }
  echo "== ranks=4 sync_interactions=$W"; grep -h "^epoch:\|Metrics" gpurun_out/window/w$W.txt
done
