# round 3, GPU call 6: where the 128-user top-k kernel spends its time (ablations), timeline of the exchange modes
set -u
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03 && export TMPDIR=/tmp
o=gpurun_out/r03
for ab in 0 1 2; do
  HEAT_CF_TOPK_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_topk_ab$ab -- python3 tools/eval_bench.py amazonbooks 20 fused > $o/topk_ab$ab.txt 2>&1
  echo "ablate=$ab"; find $o/prof_topk_ab$ab -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-160 | tail -1
done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $o/prof_shard_sync -- python3 tools/shard_bench.py --world 8 --sync --epochs 6 --windows 1 > $o/shard_sync_trace.txt 2>&1; cat $o/shard_sync_trace.txt | grep exchange
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03/prof_shard_sync/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
out = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    out.append((s - t0, e - s, r.get("Queue_Id", ""), r.get("Stream_Id", ""), name))
# print the last 120 kernels (steady state of the last mode = pipelined) and a window in the middle (overlap mode)
for part in (out[len(out)//2 - 40: len(out)//2 + 40], out[-90:]):
    print("-----")
    for s, d, q, st, n in part:
        print(f"{s/1e3:12.1f} us  +{d/1e3:8.1f} us  q={q} s={st} {n}")
PY
