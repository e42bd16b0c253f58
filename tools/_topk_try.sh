set -u
cd /root/repo; o=gpurun_out/r03; mkdir -p $o; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "topk or evaluate or recall_ndcg_parity_amazonbooks" > $o/pytest_topk.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_topk.txt
rm -rf gpurun_out/topk_stats_final; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/topk_stats_final -- python3 tools/eval_bench.py amazonbooks 20,50 fused > $o/topk_bench_final.txt 2>&1; grep "amazonbooks:" $o/topk_bench_final.txt
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/topk_stats_final/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "topk_fused" in row["Kernel_Name"]: print(row["Kernel_Name"][:50], (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6, "ms")
PY
cp gpurun_out/topk_stats_final/*/*kernel_stats.csv $o/topk_kernel_stats_final.csv
