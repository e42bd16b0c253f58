mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 1000 python -m pytest tests -m gpu -q -rP --timeout 600 -p no:cacheprovider > gpurun_out/b22_pytest_gpu_$i.txt 2>&1; echo "run $i pytest rc=$?"; tail -2 gpurun_out/b22_pytest_gpu_$i.txt | cut -c1-200
done
