mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_distributed_cpu.py -x -q -k "aggregator or distributed_main or two_ranks" > gpurun_out/b19_pytest.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/b19_pytest.txt | cut -c1-220
