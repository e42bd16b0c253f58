mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -rP -k "synthetic_hbm or tile" > gpurun_out/b20_pytest.txt 2>&1; echo "rc=$?"; tail -12 gpurun_out/b20_pytest.txt | cut -c1-220
