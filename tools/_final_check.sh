set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -rP > $o/pytest_gpu_9.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_gpu_9.txt
