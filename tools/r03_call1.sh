# round 3, GPU call 1: regression run of the whole -m gpu suite on the changed tree, then the serial aggregation arbiter
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 800 python -m pytest tests -m gpu -x -q -rP > $o/pytest_gpu_1.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_1.txt
timeout -k 10 200 python tools/serial_arbiter.py 400 11 68,344,362,2,5,8,11 > $o/serial_arbiter.txt 2>&1; echo "arbiter rc=$?"; cat $o/serial_arbiter.txt
