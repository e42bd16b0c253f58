"""Development aid (GPU box): ACCL epoch time at AmazonBooks shape per kernel variant (HEAT_CF_VARIANT) and stream count."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import abi
from heat_amd.cf import synthetic
g, d, N = synthetic.make_named("amazonbooks", with_test=False)
his, masks = synthetic.make_history(g, 100, seed=2022)
w0 = (np.random.default_rng(2022).standard_normal((d, d)) * 0.01).astype(np.float32)
for streams in [int(x) for x in sys.argv[1].split(",")]:
    uw, iw = synthetic.init_embeddings(g.num_users, g.num_items, d, seed=2022)
    eng = abi.Engine(g.clicks, uw, iw, num_negs=N, seed=2022, his=his, masks=masks, w0=w0.copy(), use_aggregator=True,
                     flags=abi.FLAG_LAZY_SYNC, num_streams=streams)
    losses = [eng.train_one_epoch()]
    eng.kernel_time(reset=True)
    for _ in range(3): losses.append(eng.train_one_epoch())
    ms, n = eng.kernel_time()
    print(os.environ.get("HEAT_CF_VARIANT", "auto"), streams, eng.kernel_name, round(ms / n, 2), "ms/epoch", round(g.clicks.shape[0] / (ms / n) / 1e3, 1), "M/s", [round(x, 4) for x in losses], flush=True)
    eng.close()
