# round 3, GPU call 15: the Yelp18-yaml bias against the 8-thread oracle as a function of the stream count, on a
# SECOND graph the constants of make_plan were not fitted on (32 clusters instead of 64, another graph seed), six seeds; and once at clip 1.0
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 800 python tests/tools/recall_parity.py --shape yelp18 --clusters 32 --graph-seed 7 --epochs 8 --clip 0.1 --streams 110,160,220 --oracle-threads 8 --seeds 1,2,3,4,5,6 > $o/yelp18_second_graph.txt 2>&1; echo "rc=$?"
grep -h "GPU seed\|ORACLE\|Recall" $o/yelp18_second_graph.txt | cut -c1-200
timeout -k 10 300 python tests/tools/recall_parity.py --shape yelp18 --clusters 32 --graph-seed 7 --epochs 8 --clip 1.0 --streams 110,220 --oracle-threads 8 --seeds 1,2 > $o/yelp18_second_graph_clip1.txt 2>&1; echo "rc=$?"
grep -h "GPU seed\|ORACLE\|Recall" $o/yelp18_second_graph_clip1.txt | cut -c1-200
