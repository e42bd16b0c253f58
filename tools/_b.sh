mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "aggregator or serial_walk or randomized" > gpurun_out/b17_pytest.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/b17_pytest.txt | cut -c1-220
