set -e
mkdir -p gpurun_out
P="python tests/tools/recall_parity.py --shape yelp18 --epochs 8 --clip 0.1 --oracle-threads 8"
$P --clusters 64 --update 4,3 --streams 85,170,327,650 --seeds 1,2 > gpurun_out/b1_c64.txt 2>&1
$P --zipf 1.0 --update 4,3 --streams 85,170,327,650 --seeds 1 > gpurun_out/b1_z10.txt 2>&1
$P --zipf 0.6 --update 4,3 --streams 85,170,327,650 --seeds 1 > gpurun_out/b1_z06.txt 2>&1
echo done
