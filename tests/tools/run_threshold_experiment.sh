mkdir -p gpurun_out/r1k
for spec in "amazonbooks 0.5 0" "amazonbooks 0.25 0" "gowalla 1.0 0" "amazonbooks 0.5 64" "gowalla 1.0 64"; do
  set -- $spec
  echo "=== shape=$1 scale=$2 clusters=$3" >> gpurun_out/r1k/thr.txt
  timeout -k 10 200 python tests/tools/recall_parity.py --shape $1 --scale $2 --clusters $3 --epochs 5 --streams 0 --coherence 2 --update 28,31 --oracle-threads 8 >> gpurun_out/r1k/thr.txt 2>&1
done
cat gpurun_out/r1k/thr.txt
