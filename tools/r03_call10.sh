# round 3, GPU call 10: top-k 128-user kernel, static-sweep selection, split count
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
HEAT_CF_TOPK_WGS=2 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "topk" > $o/pytest_topk_w2b.txt 2>&1; rc=$?; echo "pytest topk wgs2 rc=$rc"; tail -3 $o/pytest_topk_w2b.txt
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "topk" > $o/pytest_topk_w1b.txt 2>&1; rc1=$?; echo "pytest topk wgs1 rc=$rc1"; tail -3 $o/pytest_topk_w1b.txt
for z in 0 1 2 3 6; do for w in 2; do
  if [ $z -eq 0 ]; then unset HEAT_CF_TOPK_SPLITS; else export HEAT_CF_TOPK_SPLITS=$z; fi
  HEAT_CF_TOPK_WGS=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_z${z}w$w -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/topk_z${z}w$w.txt 2>&1
  echo "splits=$z wgs=$w"; find $o/prof_topk_z${z}w$w -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-160 | tail -1
done; done
unset HEAT_CF_TOPK_SPLITS
for cfg in "1 0" "1 1" "2 1"; do set -- $cfg
  HEAT_CF_TOPK_WGS=$1 HEAT_CF_TOPK_ABLATE=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_d$1$2 -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/topk_d$1$2.txt 2>&1
  echo "wgs=$1 ablate=$2"; find $o/prof_topk_d$1$2 -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-160 | tail -1
done
for shape in yelp18 gowalla_pr1; do for k in 20 50; do
  echo "== $shape k=$k"
  timeout -k 10 100 python tools/eval_bench.py $shape $k fused 2>&1 | grep fused | sed 's/^/v2 1wg  /'
  HEAT_CF_TOPK_WGS=2 timeout -k 10 100 python tools/eval_bench.py $shape $k fused 2>&1 | grep fused | sed 's/^/v2 2wg  /'
done; done
HEAT_CF_TOPK_KERNEL=v2 timeout -k 10 200 python tools/eval_scale.py 2>&1 | tail -4
