mkdir -p gpurun_out
export TMPDIR=/tmp
for A in MFMA LOADS SELECT; do
  HEAT_CF_LIB=$PWD/heat_amd/lib/exp/libheat_cf_no$A.so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b10_topk_no$A -- python3 tools/eval_bench.py amazonbooks 20 fused > gpurun_out/b10_topk_no$A.txt 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/b10_topk_no$A/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "topk_fused" in row["Name"]: print("no$A", row["AverageNs"])
PY
done
HEAT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/b10_bench_gloo2.json 2> gpurun_out/b10_bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 1500 gpurun_out/b10_bench_gloo2.json
echo done
