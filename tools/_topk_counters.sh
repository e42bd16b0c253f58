set -u
cd /root/repo; o=gpurun_out/topk_sq; mkdir -p $o; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/p1 -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/p1.txt 2> $o/p1.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $o/p2 -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/p2.txt 2> $o/p2.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT --output-format csv -d $o/p3 -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/p3.txt 2> $o/p3.err
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$o/%s/*/*counter_collection.csv" % p):
        for row in csv.DictReader(open(f)):
            if "topk_fused" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print("  %-28s launches=%d mean per launch=%.4g" % (k, len(acc[k]), sum(acc[k]) / len(acc[k])))
PY
