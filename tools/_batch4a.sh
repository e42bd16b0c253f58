set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "randomized_serial or serial_walk or collision" > gpurun_out/b4_pytest.txt 2>&1 || { tail -40 gpurun_out/b4_pytest.txt; exit 1; }
P="python tests/tools/recall_parity.py --shape yelp18 --epochs 8 --clip 0.1"
$P --clusters 64 --update 0 --streams 0,200,225 --seeds 1,2,3 --oracle-threads 8 > gpurun_out/b4_c64_default.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/b4_pmc_sq -- python3 tools/quick_bench.py --shape yelp18 --epochs 2 > gpurun_out/b4_pmc_sq.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/b4_pmc_sq2 -- python3 tools/quick_bench.py --shape yelp18 --epochs 2 > gpurun_out/b4_pmc_sq2.txt 2>&1 || true
echo done
