mkdir -p gpurun_out
python tests/tools/recall_parity.py --shape yelp18 --epochs 8 --clip 0.1 --clusters 64 --update 0 --streams 0 --seeds 1,2,3,4 --oracle-threads "" > gpurun_out/b16_yelp_gpf4.txt 2>&1
python tests/tools/recall_parity.py --shape gowalla --epochs 8 --clip 0.1 --clusters 64 --update 0 --streams 0 --seeds 1,2,3,4 --oracle-threads "" > gpurun_out/b16_gowalla_gpf4.txt 2>&1
grep "^GPU\|Recall\|^  kernel" gpurun_out/b16_yelp_gpf4.txt gpurun_out/b16_gowalla_gpf4.txt | cut -c1-230
