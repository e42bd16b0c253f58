mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 -p no:cacheprovider > gpurun_out/b7_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/b7_pytest_gpu.txt
python tools/eval_bench.py amazonbooks 20,50 fused > gpurun_out/b7_eval_bench.txt 2>&1; tail -3 gpurun_out/b7_eval_bench.txt
timeout -k 10 200 python bench.py --shape synthetic_hbm --steps 2 --warmup 1 --interactions 4000000 > gpurun_out/b7_hbm.json 2> gpurun_out/b7_hbm.err; echo "hbm rc=$?"
echo done
