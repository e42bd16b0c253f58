set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "randomized_serial or serial_walk or collision" > gpurun_out/b3_pytest.txt 2>&1 || { tail -40 gpurun_out/b3_pytest.txt; exit 1; }
P="python tests/tools/recall_parity.py --shape yelp18 --epochs 8 --clip 0.1"
for V in 8,4 4,8 2,16; do
HEAT_CF_VARIANT=$V $P --clusters 64 --update 44,46 --streams 170,256,400 --seeds 1,2 --oracle-threads "" > gpurun_out/b3_c64_v$V.txt 2>&1
done
HEAT_CF_VARIANT=4,8 $P --zipf 0.6 --update 44,46 --streams 170,256,400 --seeds 1 --oracle-threads "" > gpurun_out/b3_z06_v48.txt 2>&1
echo done
