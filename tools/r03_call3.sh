# round 3, GPU call 3
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
timeout -k 10 500 python -m pytest tests -m gpu -x -q -rP -k "accl_hogwild or serial_aggregation or randomized_serial or item_sync_kernels or side_stream" > $o/pytest_gpu_new.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu_new.txt
timeout -k 10 200 python tools/shard_bench.py --world 8 --sync --epochs 20 --windows 1,2 > $o/shard_bench_sync.txt 2>&1; echo "shard rc=$?"; cat $o/shard_bench_sync.txt
timeout -k 10 200 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --tile --streams 8,16 --oracle-threads 8,16 --seeds 2022,7,99 > $o/tile_matched_8_16.txt 2>&1; echo "tile rc=$?"
timeout -k 10 100 python tests/tools/recall_parity.py --shape amazonbooks --epochs 5 --agg --streams 0 --oracle-threads "" --seeds 1,2,3,4,5,6 > $o/accl_more_seeds.txt 2>&1; echo "accl rc=$?"
timeout -k 10 500 python tests/tools/recall_parity.py --shape synthetic_hbm --users 75000 --items 200000 --interactions 1500000 --clusters 64 --epochs 3 --seeds 1,2 --oracle-threads 8 > $o/config_s_regime.txt 2>&1; echo "S rc=$?"
grep -h "GPU seed\|ORACLE\|Recall" $o/tile_matched_8_16.txt $o/accl_more_seeds.txt $o/config_s_regime.txt
