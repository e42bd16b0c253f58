import os, sys, types, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import metrics, synthetic
g, d, N = synthetic.make_named("amazonbooks")
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist() for u in range(g.num_users) if ep[u + 1] > ep[u]})
ms = ["Recall(k=20)", "NDCG(k=20)"]
mode = sys.argv[1]
for seed in (2022, 7, 99, 2022, 2022, 7):
    uw0, iw0 = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=seed)
    uw, iw = uw0.copy(), iw0.copy()
    kw = dict(num_streams=0) if mode == "kw" else {}
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=seed, flags=abi.FLAG_LAZY_SYNC, **kw)
    if mode == "loop":
        losses = []
        for _ in range(5):
            losses.append(eng.train_one_epoch())
    else:
        losses = [eng.train_one_epoch() for _ in range(5)]
    eng.sync_to_host()
    name = eng.kernel_name
    eng.close()
    ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
    top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
    ev.close()
    r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
    print(mode, seed, name, round(r[ms[0]], 5), round(r[ms[1]], 5), [round(x, 4) for x in losses], flush=True)
