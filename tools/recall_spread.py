"""Development aid (GPU box): run-to-run spread of Recall@20 / NDCG@20 of ONE configuration (AmazonBooks shape, 5 epochs)."""
import os, sys, types, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import metrics, synthetic
g, d, N = synthetic.make_named("amazonbooks")
ep = g.test_indptr.astype(np.int64)
test = types.SimpleNamespace(user_items_dic={u: g.test_items[ep[u]:ep[u + 1]].tolist() for u in range(g.num_users) if ep[u + 1] > ep[u]})
ms = ["Recall(k=20)", "NDCG(k=20)"]
for streams in [int(x) for x in sys.argv[1].split(",")]:
    res = []
    for rep in range(int(sys.argv[2])):
        uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=2022)
        eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=2022, flags=abi.FLAG_LAZY_SYNC, num_streams=streams)
        losses = [eng.train_one_epoch() for _ in range(5)]
        eng.sync_to_host()
        name = eng.kernel_name
        eng.close()
        ev = abi.Engine(g.clicks[:1].copy(), uw, iw, num_negs=N)
        top = ev.topk(20, mask_indptr=g.train_indptr, mask_items=g.train_items)
        ev.close()
        r = metrics.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
        res.append((r[ms[0]], r[ms[1]], losses[-1]))
    a = np.array(res)
    print(name, "Recall", np.round(a[:, 0], 5).tolist(), "NDCG", np.round(a[:, 1], 5).tolist(), "loss", np.round(a[:, 2], 4).tolist(), flush=True)
    print("   mean", a.mean(0).round(5).tolist(), "std", a.std(0).round(5).tolist(), "min", a.min(0).round(5).tolist(), "max", a.max(0).round(5).tolist(), flush=True)
