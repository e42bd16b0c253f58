mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -rP -k "tile" > gpurun_out/b14_pytest_tile.txt 2>&1; echo "pytest rc=$?"; grep -v "^$" gpurun_out/b14_pytest_tile.txt | tail -30 | cut -c1-250
