# round 3, GPU call 9: which top-k kernel wins where (64-user v1, 128-user v2 with one / two workgroups per CU)
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
for shape in yelp18 gowalla_pr1; do for k in 20 50; do
  echo "== $shape k=$k"
  HEAT_CF_TOPK_KERNEL=v1 timeout -k 10 100 python tools/eval_bench.py $shape $k fused 2>&1 | grep fused | sed 's/^/v1      /'
  timeout -k 10 100 python tools/eval_bench.py $shape $k fused 2>&1 | grep fused | sed 's/^/v2 1wg  /'
  HEAT_CF_TOPK_WGS=2 timeout -k 10 100 python tools/eval_bench.py $shape $k fused 2>&1 | grep fused | sed 's/^/v2 2wg  /'
done; done > $o/topk_matrix.txt 2>&1; cat $o/topk_matrix.txt
for kern in v1 v2; do
  HEAT_CF_TOPK_KERNEL=$kern timeout -k 10 200 python tools/eval_scale.py > $o/topk_scale_$kern.txt 2>&1; echo "== scale $kern"; tail -4 $o/topk_scale_$kern.txt
done
