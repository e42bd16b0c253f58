# SQ counters of the training kernel for one shape (GPU box): two rocprofv3 --pmc passes (counters only, no trace domains),
# summarised per launch.   usage: bash tools/sq_counters.sh yelp18 [quick_bench options, e.g. --agg] > gpurun_out/sq/summary.txt
shape=${1:-yelp18}; shift; o=gpurun_out/sq; mkdir -p $o; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/p1 -- python3 tools/quick_bench.py --shape $shape --epochs 2 "$@" > $o/p1.txt 2> $o/p1.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $o/p2 -- python3 tools/quick_bench.py --shape $shape --epochs 2 "$@" > $o/p2.txt 2> $o/p2.err
grep coherence $o/p1.txt | cut -c1-160
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$o/%s/*/*counter_collection.csv" % p):
        for row in csv.DictReader(open(f)):
            if "ccl_train_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print("  %-24s launches=%d mean per launch=%.4g" % (k, len(acc[k]), sum(acc[k]) / len(acc[k])))
PY
