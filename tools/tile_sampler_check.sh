# Evidence run (GPU box): the random-tile negative sampler (neg_sampler: 1, tile_size 512, refresh_interval 8192;
# random_tile_negative_sampler.cpp:22-45) against the uniform sampler at AmazonBooks shape — Recall@20 and epoch time.
mkdir -p gpurun_out/tile
CFG=heat_amd/cf/benchmarks/AmazonBooks/MF_CCL/configs/config0.yaml
python3 - <<PY
import yaml
c = yaml.safe_load(open("$CFG"))
c["model_config"]["neg_sampler"] = 1
open("gpurun_out/tile/tile.yaml", "w").write(yaml.safe_dump(c))
PY
for S in uniform tile; do
  if [ $S = uniform ]; then C=$CFG; else C=gpurun_out/tile/tile.yaml; fi
  timeout -k 10 280 python -m heat_amd.cf.main --config $C --synthetic amazonbooks > gpurun_out/tile/$S.txt 2>&1 || { echo "$S failed"; tail -5 gpurun_out/tile/$S.txt; exit 1; }
  echo "== neg_sampler=$S"; grep -h "^epoch:\|Recall" gpurun_out/tile/$S.txt
done
