mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "topk or evaluate0" > gpurun_out/b21_pytest_topk.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/b21_pytest_topk.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b21_topk_stats -- python3 tools/eval_bench.py amazonbooks 20,50 fused > gpurun_out/b21_topk_stats.txt 2>&1; grep "top-" gpurun_out/b21_topk_stats.txt
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/b21_topk_stats/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "topk_fused" in row["Name"]: print(row["Name"][40:70], row["AverageNs"])
PY
python tools/eval_scale.py > gpurun_out/b21_eval_scale.txt 2>&1; tail -4 gpurun_out/b21_eval_scale.txt
