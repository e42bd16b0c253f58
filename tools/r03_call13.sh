set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 200 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --agg --streams 512,640,768 --oracle-threads "" --seeds 2022,7,99 > $o/accl_640.txt 2>&1; grep -h "kernel \|GPU seed\|Recall" $o/accl_640.txt
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "gowalla_pr1 or config_s_regime or eight_user" -rP > $o/pytest_gpu_6.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_6.txt; grep -n "means:" $o/pytest_gpu_6.txt | cut -c1-200
