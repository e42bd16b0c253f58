mkdir -p gpurun_out
export TMPDIR=/tmp
HEAT_BENCH_FORCE_SYNC=1 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/b6_bench_forcesync.json 2> gpurun_out/b6_bench_forcesync.err; echo "forcesync rc=$?" | tee -a gpurun_out/b6_bench_forcesync.err
for V in auto 13,8 16,8; do
  if [ $V = auto ]; then unset HEAT_CF_VARIANT; else export HEAT_CF_VARIANT=$V; fi
  timeout -k 10 200 python bench.py --shape synthetic_hbm --steps 2 --warmup 1 --interactions 4000000 > gpurun_out/b6_hbm_$V.json 2> gpurun_out/b6_hbm_$V.err; echo "hbm $V rc=$?"
done
export HEAT_CF_VARIANT=13,8
timeout -k 10 200 python bench.py --shape synthetic_hbm --steps 2 --warmup 1 --interactions 4000000 --num-streams 512 > gpurun_out/b6_hbm_13,8_s512.json 2> gpurun_out/b6_hbm_13,8_s512.err; echo "rc=$?"
unset HEAT_CF_VARIANT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b6_topk_stats -- python3 tools/eval_bench.py amazonbooks 20 fused > gpurun_out/b6_topk_stats.txt 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/b6_topk_pmc -- python3 tools/eval_bench.py amazonbooks 20 fused > gpurun_out/b6_topk_pmc.txt 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/b6_topk_pmc2 -- python3 tools/eval_bench.py amazonbooks 20 fused > gpurun_out/b6_topk_pmc2.txt 2>&1; echo "rc=$?"
echo done
