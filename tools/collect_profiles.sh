# Copies the summaries of tools/final_profile.sh (gpurun_out/final/, scratch) into profiles/ under this round's names and
# derives the per-launch memory-side traffic from the two PMC passes.   usage: bash tools/collect_profiles.sh r02
set -e
r=${1:-r03}; o=gpurun_out/final; p=profiles
newest() { ls -t $1 | head -1; }
cp $o/bench_n1.json $p/${r}_bench_n1.json
cp $o/bench_under_rocprof.json $p/${r}_bench_under_rocprof.json
cp $(newest "$o/prof/*/*_kernel_stats.csv") $p/${r}_bench_kernel_stats.csv
cp $(newest "$o/pmc_fetch/*/*_counter_collection.csv") $p/${r}_pmc_fetch_counter_collection.csv
cp $(newest "$o/pmc_write/*/*_counter_collection.csv") $p/${r}_pmc_write_counter_collection.csv
cp $o/bench_hbm_under_rocprof.json $p/${r}_bench_hbm.json
cp $(newest "$o/prof_hbm/*/*_kernel_stats.csv") $p/${r}_bench_hbm_kernel_stats.csv
cp $(newest "$o/pmc_fetch_hbm/*/*_counter_collection.csv") $p/${r}_pmc_fetch_hbm_counter_collection.csv
cp $(newest "$o/pmc_write_hbm/*/*_counter_collection.csv") $p/${r}_pmc_write_hbm_counter_collection.csv
cp $o/bench_yelp18.json $p/${r}_bench_yelp18.json
cp $(newest "$o/prof_yelp18/*/*_kernel_stats.csv") $p/${r}_bench_yelp18_kernel_stats.csv
cp $(newest "$o/pmc_fetch_yelp18/*/*_counter_collection.csv") $p/${r}_pmc_fetch_yelp18_counter_collection.csv
cp $(newest "$o/pmc_write_yelp18/*/*_counter_collection.csv") $p/${r}_pmc_write_yelp18_counter_collection.csv
cp $o/bench_gowalla.json $p/${r}_bench_gowalla.json
cp $o/bench_gowalla_pr1.json $p/${r}_bench_gowalla_pr1.json
cp $o/bench_forcesync.json $p/${r}_bench_forcesync.json
cp $(newest "$o/prof_topk/*/*_kernel_stats.csv") $p/${r}_topk_kernel_stats.csv
cp $(newest "$o/prof_accl/*/*_kernel_stats.csv") $p/${r}_accl_kernel_stats.csv
cp $(newest "$o/pmc_fetch_accl/*/*_counter_collection.csv") $p/${r}_pmc_fetch_accl_counter_collection.csv
cp $(newest "$o/pmc_write_accl/*/*_counter_collection.csv") $p/${r}_pmc_write_accl_counter_collection.csv
cp $o/accl_under_rocprof.txt $p/${r}_accl_quick_bench.txt
cp $o/shard_bench_exchange_modes.txt $p/${r}_shard_bench_exchange_modes.txt
python - <<PY
import json, subprocess, sys
for tag, bench in (("", "$o/bench_n1.json"), ("_hbm", "$o/bench_hbm_under_rocprof.json"), ("_yelp18", "$o/bench_yelp18.json")):
    d = json.load(open(bench))
    rf = d.get("roofline_hbm_resident", d["roofline"]) if tag == "_hbm" else d["roofline"]
    kernel = rf.get("kernel", d["config"]["kernel"]) if tag == "_hbm" else d["config"]["kernel"]
    inter = round(rf["algorithmic_gb_per_launch"] * 1e9 / rf["bytes_per_interaction"])
    subprocess.check_call([sys.executable, "tools/pmc_traffic.py", "$p/${r}_pmc_fetch%s_counter_collection.csv" % tag,
                           "$p/${r}_pmc_write%s_counter_collection.csv" % tag, "$p/${r}_pmc_traffic%s.json" % tag, kernel,
                           str(inter), str(rf["bytes_per_interaction"]), d["config"]["workload"]], stdout=subprocess.DEVNULL)
    t = json.load(open("$p/${r}_pmc_traffic%s.json" % tag))
    print(tag or "headline", kernel, "traffic/algorithmic =", round(t["traffic_over_algorithmic"], 4))
PY
python - <<PY
# ACCL: traffic of the aggregation kernel from its own two PMC passes; algorithmic bytes from quick_bench's header line
import json, re, subprocess, sys
txt = open("$o/accl_under_rocprof.txt").read()
m = re.search(r"n=(\d+) d=(\d+) N=(\d+) B/sample=(\d+)", txt)
n, B = int(m.group(1)), int(m.group(4))
kern = re.search(r"kernel=(\S+)", txt).group(1)
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", "$p/${r}_pmc_fetch_accl_counter_collection.csv",
                       "$p/${r}_pmc_write_accl_counter_collection.csv", "$p/${r}_pmc_traffic_accl.json", kern, str(n), str(B),
                       "AmazonBooks shape with behaviour aggregation (histories of up to 100 items)"], stdout=subprocess.DEVNULL)
t = json.load(open("$p/${r}_pmc_traffic_accl.json"))
print("accl", kern, "traffic/algorithmic =", round(t["traffic_over_algorithmic"], 4))
PY

