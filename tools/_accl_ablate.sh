set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/accl_gather_first.txt
timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --agg --epochs 3 --streams 0,256,768 > $o 2>&1 || exit 1
grep -v "^shape" $o | cut -c1-150
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "agg or accl or aggregat or serial or randomized" > gpurun_out/r03/pytest_accl.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03/pytest_accl.txt
