"""Regenerates tests/golden/metrics_golden.json by IMPORTING the reference's pure-numpy
cf_cpu/cf/metrics.py (metrics.py:5-159) and running it on small seeded inputs.

Run in the build container only:  python tests/golden/gen_metrics_golden.py
The fixture holds inputs and expected outputs only.
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_CF = "/root/reference/cf_cpu/cf"


def make_case(rng, num_users, num_items, n_train, n_test, ties=False, skip_test_users=()):
    sim = rng.standard_normal((num_users, num_items)).astype(np.float32)
    if ties:
        sim = np.round(sim * 2) / 2  # many exact ties -> exercises argpartition/argsort tie order
    train, test = {}, {}
    for u in range(num_users):
        items = rng.permutation(num_items)
        train[u] = [int(x) for x in items[:n_train + (u % 3)]]
        if u not in skip_test_users:
            test[u] = [int(x) for x in items[n_train + 3:n_train + 3 + n_test + (u % 4)]]
    return sim, train, test


def main():
    if not os.path.isdir(REF_CF):
        sys.exit("reference checkout not present; fixture cannot be regenerated here")
    sys.path.insert(0, REF_CF)
    import metrics as ref_metrics  # the reference module (numpy only)

    rng = np.random.default_rng(2022)
    metric_sets = [
        ["Recall(k=20)"],
        ["Recall(k=20)", "Recall(k=50)", "NDCG(k=20)", "NDCG(k=50)", "HitRate(k=20)", "HitRate(k=50)"],
        ["Recall(k=5)", "NormalizedRecall(k=5)", "Precision(k=5)", "F1(k=5)", "DCG(k=5)", "NDCG(k=5)", "MRR(k=5)",
         "HitRate(k=5)", "MAP(k=5)"],
    ]
    cases = []
    specs = [dict(num_users=3, num_items=60, n_train=5, n_test=3),
             dict(num_users=12, num_items=80, n_train=9, n_test=6),
             dict(num_users=7, num_items=64, n_train=4, n_test=25),   # more true items than k
             dict(num_users=9, num_items=70, n_train=6, n_test=4, skip_test_users=(2, 5))]
    for si, spec in enumerate(specs):
        sim, train, test = make_case(rng, **spec)
        for ms in metric_sets:
            if max(int(m.split("k=")[-1].strip(")")) for m in ms) >= spec["num_items"]:
                continue
            train_data = types.SimpleNamespace(user_items_dic=train)
            test_data = types.SimpleNamespace(user_items_dic=test)
            with contextlib.redirect_stdout(io.StringIO()):
                res = ref_metrics.evaluate_metrics(train_data, test_data, sim.copy(), ms)
            cases.append(dict(sim_bits=sim.view(np.uint32).tolist(), shape=list(sim.shape),
                              train={str(k): v for k, v in train.items()},
                              test={str(k): v for k, v in test.items()},
                              metrics=ms, expected=res))
    out = os.path.join(ROOT, "tests", "golden", "metrics_golden.json")
    with open(out, "w") as f:
        json.dump(dict(generator="tests/golden/gen_metrics_golden.py", numpy=np.__version__, cases=cases), f)
    print("wrote", out, os.path.getsize(out), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
