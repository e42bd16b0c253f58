mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 -p no:cacheprovider > gpurun_out/b9_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/b9_pytest_gpu.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b9_topk_stats -- python3 tools/eval_bench.py amazonbooks 20,50 fused > gpurun_out/b9_topk_stats.txt 2>&1; echo "rc=$?"; grep "top-" gpurun_out/b9_topk_stats.txt
echo done
