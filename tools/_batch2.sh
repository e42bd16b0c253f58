set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "randomized_serial or hiprand or serial_walk or collision or sampler or windows" > gpurun_out/b2_pytest.txt 2>&1 || { tail -40 gpurun_out/b2_pytest.txt; exit 1; }
P="python tests/tools/recall_parity.py --shape yelp18 --epochs 8 --clip 0.1"
$P --clusters 64 --update 44,46,3,4 --streams 128,170,256 --seeds 1,2 --oracle-threads "" > gpurun_out/b2_c64.txt 2>&1
HEAT_CF_VARIANT=4,8 $P --clusters 64 --update 44,3 --streams 170,256 --seeds 1 --oracle-threads "" > gpurun_out/b2_c64_v48.txt 2>&1
HEAT_CF_VARIANT=2,16 $P --clusters 64 --update 44,3 --streams 170,256 --seeds 1 --oracle-threads "" > gpurun_out/b2_c64_v216.txt 2>&1
$P --zipf 0.6 --update 44,46 --streams 128,170,256 --seeds 1 --oracle-threads "" > gpurun_out/b2_z06.txt 2>&1
echo done
