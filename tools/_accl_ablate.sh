set -u
cd /root/repo; mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03/pipeline.txt
: > $o
for np in 1 ""; do
  echo "## HEAT_CF_NO_PIPELINE=$np" >> $o
  env ${np:+HEAT_CF_NO_PIPELINE=1} timeout -k 10 200 python tools/quick_bench.py --shape amazonbooks --epochs 5 --streams 0,1024,3017 >> $o 2>&1 || exit 1
  env ${np:+HEAT_CF_NO_PIPELINE=1} timeout -k 10 200 python tools/quick_bench.py --shape gowalla_pr1 --epochs 5 --streams 0 >> $o 2>&1 || exit 1
  env ${np:+HEAT_CF_NO_PIPELINE=1} timeout -k 10 200 python tools/shard_bench.py --world 8 --epochs 20 >> $o 2>&1 || exit 1
done
grep "^##\|coherence\|epoch-shard" $o | cut -c1-170
