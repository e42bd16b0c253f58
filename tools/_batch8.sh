mkdir -p gpurun_out
export TMPDIR=/tmp
P="python tests/tools/recall_parity.py --epochs 8 --clip 0.1 --clusters 64"
$P --shape gowalla --update 0 --streams 96,128,170,200 --seeds 1,2 --oracle-threads 8 --oracle-seeds 1 > gpurun_out/b8_gowalla.txt 2>&1
$P --shape yelp18 --interactions 810128 --update 0 --streams 128,170,256 --seeds 1,2 --oracle-threads 8 > gpurun_out/b8_yelp_T810k.txt 2>&1
$P --shape yelp18 --interactions 2000000 --update 0 --streams 256,400 --seeds 1,2 --oracle-threads 8 --oracle-seeds 1 > gpurun_out/b8_yelp_T2M.txt 2>&1
echo done
