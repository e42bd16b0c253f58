mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/b11_pmc_a1 -- python3 tools/quick_bench.py --shape amazonbooks --epochs 3 > gpurun_out/b11_pmc_a1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/b11_pmc_a2 -- python3 tools/quick_bench.py --shape amazonbooks --epochs 3 > gpurun_out/b11_pmc_a2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/b11_pmc_a3 -- python3 tools/quick_bench.py --shape amazonbooks --epochs 3 > gpurun_out/b11_pmc_a3.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q -rP --timeout 600 -p no:cacheprovider > gpurun_out/b11_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/b11_pytest_gpu.txt
echo done
