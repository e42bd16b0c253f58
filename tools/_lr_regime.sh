set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/lr_regime.txt
: > $o
for lr in 0.03 0.1; do
  echo "##### amazonbooks lr=$lr clip=1.0" >> $o
  timeout -k 10 400 python tests/tools/recall_parity.py --shape amazonbooks --lr $lr --clip 1.0 --streams 3017,1024,512,256 --seeds 2022,7 --oracle-threads 8 >> $o 2>&1 || exit 1
done
for lr in 0.03; do
  echo "##### yelp18 clusters=64 lr=$lr clip=0.1" >> $o
  timeout -k 10 500 python tests/tools/recall_parity.py --shape yelp18 --clusters 64 --epochs 8 --lr $lr --clip 0.1 --streams 220,110,64 --seeds 1,2 --oracle-threads 8 >> $o 2>&1 || exit 1
done
tail -5 $o
