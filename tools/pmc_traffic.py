"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
per-launch HBM-side traffic of the training kernel.  Corrections per that guide: both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read; WRITE_SIZE is exact for 16-B/lane
stores and float atomics.  The guide also says the counters sit on the L2's memory-side requests and appear to count
Infinity-Cache hits: for a working set below 256 MiB this is fabric traffic, not DRAM traffic.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> \
        <kernel name as bench.py prints it> <interactions per launch> <bytes per interaction> "<workload>"
"""
import csv
import json
import sys


def mean_counter(path, counter, kernel_substr):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel_substr} in {path}")
    return sum(vals) / len(vals), len(vals)


if __name__ == "__main__":
    fetch_csv, write_csv, out, kernel = sys.argv[1:5]
    interactions, bytes_per = int(sys.argv[5]), int(sys.argv[6])
    workload = sys.argv[7] if len(sys.argv) > 7 else ""
    fetch_kib, n1 = mean_counter(fetch_csv, "FETCH_SIZE", "ccl_train_kernel")
    write_kib, n2 = mean_counter(write_csv, "WRITE_SIZE", "ccl_train_kernel")
    res = {
        "kernel": kernel.split("/")[0], "workload": workload, "launches_averaged": [n1, n2],
        "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_bytes_per_launch": fetch_kib * 1024 * 2,     # gfx950: x2 for 16 B/lane reads
        "write_bytes_per_launch": write_kib * 1024,
    }
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["algorithmic_bytes_per_launch"] = bytes_per * interactions
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
