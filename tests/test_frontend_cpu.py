"""CPU tests of the host-side mirror of the reference's Python frontend: metrics against golden vectors generated
by importing the reference's metrics.py (tests/golden/gen_metrics_golden.py), the LightGCN parser against a
hand-made fixture, the cf_c module surface and the C-ABI symbol table (no compute calls: there is no GPU here)."""
import contextlib
import io
import json
import os
import re
import types

import numpy as np
import pytest

from heat_amd.cf import metrics as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def metric_cases(golden_dir):
    with open(os.path.join(golden_dir, "metrics_golden.json")) as f:
        return json.load(f)["cases"]


def test_metrics_match_reference_golden(metric_cases):
    assert len(metric_cases) >= 10
    for case in metric_cases:
        sim = np.asarray(case["sim_bits"], dtype=np.uint32).view(np.float32).reshape(case["shape"]).copy()
        train = types.SimpleNamespace(user_items_dic={int(k): v for k, v in case["train"].items()})
        test = types.SimpleNamespace(user_items_dic={int(k): v for k, v in case["test"].items()})
        got = M.evaluate_metrics(train, test, sim, case["metrics"], quiet=True)
        for name, want in case["expected"].items():
            assert got[name] == pytest.approx(want, rel=1e-12, abs=1e-15), name


def test_evaluate_topk_equals_dense_path(metric_cases):
    case = metric_cases[1]
    sim = np.asarray(case["sim_bits"], dtype=np.uint32).view(np.float32).reshape(case["shape"]).copy()
    train = {int(k): v for k, v in case["train"].items()}
    test = types.SimpleNamespace(user_items_dic={int(k): v for k, v in case["test"].items()})
    masked = sim.copy()
    for u, items in train.items():
        masked[u, items] = -np.inf
    top = np.argsort(-masked, axis=1, kind="stable")[:, :50]
    got = M.evaluate_topk(test, top, case["metrics"], quiet=True)
    for name, want in case["expected"].items():
        assert got[name] == pytest.approx(want, rel=1e-12)


def test_vectorised_scorer_equals_the_per_user_loop_bit_for_bit():
    """metrics._score_vectorised (what evaluate_metrics / evaluate_topk run) against metrics._score_per_user (the
    reference's loop, metrics.py:31-34) on random lists: every metric, cut-offs above and below the list lengths, users
    with one item, duplicated test items, hits forced into the head of the list."""
    rng = np.random.default_rng(5)
    U, I, k = 400, 900, 50
    top = np.stack([rng.permutation(I)[:k] for _ in range(U)]).astype(np.uint32)
    true = [rng.choice(I, int(rng.integers(1, 70)), replace=False).tolist() for _ in range(U)]
    for u in range(0, U, 3):                                   # some early hits
        keep = [x for x in top[u] if x not in true[u][:3]]
        top[u] = np.array(true[u][:3] + keep, dtype=np.uint32)[:k]
    true[7] = true[7] + true[7][:2]                            # duplicated test items count in len(true), as in the reference
    names = ["Recall(k=20)", "NDCG(k=20)", "NormalizedRecall(k=10)", "Precision(k=50)", "F1(k=20)", "DCG(k=50)", "MRR(k=20)",
             "HitRate(k=5)", "MAP(k=50)", "NDCG(k=50)", "Recall(k=1)"]
    callers = [M._parse(m) for m in names]
    slow = M._score_per_user(top, true, callers)
    fast = M._score_vectorised(top, true, callers)
    assert np.array_equal(slow, fast)
    assert fast[:, 0].max() > 0 and fast[:, 8].max() > 0


def test_unknown_metric_raises():
    with pytest.raises(NotImplementedError):
        M.evaluate_topk(types.SimpleNamespace(user_items_dic={0: [1]}), np.zeros((1, 5), int), ["Bogus(k=3)"])


def test_cf_c_module_surface():
    """Names, keyword arguments and attributes of pybind/init_modules.cpp:13-152."""
    from heat_amd import cf_c
    m = cf_c.modules
    for path in ["CFConfig", "datasets.Dataset", "datasets.ClickDataset", "models.Model", "models.MatrixFactorization",
                 "behavior_aggregators.AggregatorWeights", "train.Engine", "test.test_out"]:
        obj = m
        for part in path.split("."):
            obj = getattr(obj, part)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = m.CFConfig(emb_dim=64, num_negs=16, num_users=4, num_items=10, train_size=12, neg_sampler=0, tile_size=512,
                         refresh_interval=8192, num_subepoches=2, l2=1e-7, clip_val=1.0, milestones=[10], l_r=0.01)
    assert cfg.emb_dim == 64
    cfg.emb_dim = 64
    assert issubclass(m.datasets.ClickDataset, m.datasets.Dataset)
    assert issubclass(m.models.MatrixFactorization, m.models.Model)
    ds = m.datasets.ClickDataset(click_dataset=np.zeros((12, 2), np.uint64), historical_items=np.zeros((4, 5), np.uint64),
                                 masks=np.ones((4, 1), np.uint64))
    assert ds.data_rows == 12 and ds.max_his == 0
    ds.max_his = 5
    ds.data_rows = 12
    with pytest.raises(ValueError):   # the reference would silently bind a converted temporary
        m.datasets.ClickDataset(click_dataset=np.zeros((12, 2), np.int32), historical_items=np.zeros((4, 5), np.uint64),
                                masks=np.ones((4, 1), np.uint64))
    with pytest.raises(ValueError):
        m.datasets.ClickDataset(click_dataset=np.zeros((12, 4), np.uint64)[:, ::2], historical_items=np.zeros((4, 5), np.uint64),
                                masks=np.ones((4, 1), np.uint64))
    model = m.models.MatrixFactorization(cf_config=cfg, user_weights=np.zeros((4, 64), np.float32),
                                         item_weights=np.zeros((10, 64), np.float32))
    with pytest.raises(ValueError):
        m.models.MatrixFactorization(cf_config=cfg, user_weights=np.zeros((4, 32), np.float32),
                                     item_weights=np.zeros((10, 64), np.float32))
    agg = m.behavior_aggregators.AggregatorWeights(aggregator_weights0=np.zeros((64, 64), np.float32))
    assert agg.emb_dim == 64
    assert hasattr(m.train.Engine, "train_one_epoch") and hasattr(m.train.Engine, "evaluate0")
    # no GPU in this container: constructing the engine must fail loudly, never fall back to a CPU path
    with pytest.raises(RuntimeError, match="no usable HIP device|no CPU fallback"):
        m.train.Engine(dataset=ds, aggregator_weights=agg, model=model, cf_config=cfg)


def test_abi_exports_every_declared_symbol():
    """Every function include/heat_cf.h declares is exported by lib/libheat_cf.so and typed in heat_amd.abi."""
    from heat_amd import abi
    hdr = open(os.path.join(ROOT, "include", "heat_cf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(heat_cf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    lib = abi.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.heat_cf_abi_version() == 1


def test_abi_struct_layout_matches_header(tmp_path):
    """ctypes mirror of heat_cf_config / heat_cf_device_view vs the layout gcc gives the header's structs."""
    import ctypes as C
    import subprocess
    from heat_amd import abi
    fields = [f[0] for f in abi.Config._fields_]
    vfields = [f[0] for f in abi.DeviceView._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/heat_cf.h"', 'int main(void){',
             'printf("%zu\\n", sizeof(heat_cf_config));']
    lines += [f'printf("%zu\\n", offsetof(heat_cf_config, {f}));' for f in fields]
    lines += ['printf("%zu\\n", sizeof(heat_cf_device_view));']
    lines += [f'printf("%zu\\n", offsetof(heat_cf_device_view, {f}));' for f in vfields]
    lines += ['return 0;}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.Config)] + [getattr(abi.Config, f).offset for f in fields]
    want += [C.sizeof(abi.DeviceView)] + [getattr(abi.DeviceView, f).offset for f in vfields]
    assert out == want


def test_lightgcn_parser(golden_dir):
    """cf/datasets.py:31-79 semantics on a hand-made file: file order, history padding with the last item, masks."""
    from heat_amd.cf.cf_config import CFConfig
    from heat_amd.cf.datasets import ClickDataset
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = CFConfig(emb_dim=64, num_negs=4, max_his=4, milestones=[10])
        tr = ClickDataset(os.path.join(golden_dir, "tiny_lightgcn", "train.txt"), config=cfg, seed=1, cache=False)
        te = ClickDataset(os.path.join(golden_dir, "tiny_lightgcn", "test.txt"), config=cfg, seed=1, cache=False)
    assert tr.click_dataset.dtype == np.uint64
    assert tr.click_dataset.tolist() == [[0, 3], [0, 4], [0, 7], [1, 0], [1, 1], [1, 2], [1, 3], [1, 5], [1, 6], [2, 9],
                                         [3, 2], [3, 8]]
    assert (cfg.num_users, cfg.num_items, cfg.train_size) == (4, 10, 12)
    assert tr.masks[:, 0].tolist() == [3, 4, 1, 2]
    assert tr.his_items[0].tolist() == [3, 4, 7, 7] and tr.his_items[2].tolist() == [9, 9, 9, 9]
    assert sorted(tr.his_items[1].tolist()) == sorted(set(tr.his_items[1].tolist())) and set(tr.his_items[1]) <= {0, 1, 2, 3, 5, 6}
    assert tr.c_instance.max_his == 4 and tr.c_instance.data_rows == 12
    assert te.c_instance is None and te.user_items_dic == {0: [1, 2], 1: [9], 3: [0, 5]}
    indptr, items = tr.train_csr()
    assert indptr.tolist() == [0, 3, 9, 10, 12] and items.tolist() == [3, 4, 7, 0, 1, 2, 3, 5, 6, 9, 2, 8]


def test_click_dataset_three_inputs_one_state(tmp_path):
    """ClickDataset from a LightGCN file (with a user listed twice: first line's position, last line's items,
    datasets.py:56), from {user: [items]} and from a CSR builds the same interaction list, histories (seeded sample for the
    long lists, padding for the short ones), lengths, dict, lazily materialised pair list and evaluation CSR."""
    from heat_amd.cf.cf_config import CFConfig
    from heat_amd.cf.datasets import ClickDataset
    rng = np.random.default_rng(3)
    U, I, H = 40, 300, 6
    lists = [rng.choice(I, int(rng.integers(1, 15)), replace=False).tolist() for _ in range(U)]
    path = tmp_path / "train.txt"
    lines = [f"{u} " + " ".join(map(str, lists[u])) for u in range(U)]
    lines.insert(5, "17 1 2 3")                      # user 17 appears early with other items: its later line wins
    path.write_text("\n".join(lines) + "\n")
    order = [u for u in range(U) if u != 17]
    order.insert(5, 17)                              # dict order = first occurrence
    as_dict = {u: lists[u] for u in order}
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    items = np.array([i for x in lists for i in x], dtype=np.uint32)
    with contextlib.redirect_stdout(io.StringIO()):
        cfgs = [CFConfig(emb_dim=64, num_negs=4, max_his=H, milestones=[10]) for _ in range(3)]
        a = ClickDataset(str(path), config=cfgs[0], seed=9, cache=False)
        b = ClickDataset(config=cfgs[1], seed=9, user_items=as_dict, is_train=True)
        c = ClickDataset(config=cfgs[2], seed=9, csr=(indptr, items), is_train=True)
    assert a.user_items_dic == b.user_items_dic and list(a.user_items_dic) == order
    assert np.array_equal(a.click_dataset, b.click_dataset) and np.array_equal(a.his_items, b.his_items)
    assert np.array_equal(a.masks, b.masks) and a.user_item_ids == b.user_item_ids == list(map(tuple, a.click_dataset.tolist()))
    assert a.click_dataset[:len(lists[0])].tolist() == [[0, i] for i in lists[0]]
    # the CSR lists the users in id order: same per-user state, interaction list in id order
    assert c.user_items_dic == {u: lists[u] for u in range(U)} and np.array_equal(c.masks, a.masks)
    short = [u for u in range(U) if len(lists[u]) < H]
    assert np.array_equal(c.his_items[short], a.his_items[short])
    for u in range(U):
        if len(lists[u]) >= H:
            assert set(c.his_items[u].tolist()) <= set(lists[u]) and len(set(c.his_items[u].tolist())) == H
    assert (cfgs[0].num_users, cfgs[0].train_size) == (cfgs[2].num_users, cfgs[2].train_size) == (U, sum(map(len, lists)))
    for ds in (a, b, c):
        ip, it = ds.train_csr()
        assert np.array_equal(ip, indptr) and np.array_equal(it, items)


def test_click_dataset_from_an_empty_dict():
    """An empty split (e.g. a test file without lines) builds an empty dataset instead of raising (round-2 regression)."""
    from heat_amd.cf.cf_config import CFConfig
    from heat_amd.cf.datasets import ClickDataset
    with contextlib.redirect_stdout(io.StringIO()):
        ds = ClickDataset(config=CFConfig(emb_dim=64, num_negs=4, max_his=6, milestones=[10]), user_items={}, is_train=False)
        one = ClickDataset(config=CFConfig(emb_dim=64, num_negs=4, max_his=6, milestones=[10]), user_items={0: [1, 2], 3: []},
                           is_train=False)
    assert ds.user_item_ids == [] and ds.user_items_dic == {} and ds.his_items.shape[0] == 0
    assert one.user_item_ids == [(0, 1), (0, 2)] and one.masks[:, 0].tolist() == [2, 0, 0, 0]


def test_yaml_configs_carry_the_reference_keys():
    from heat_amd.cf import utils
    base = os.path.join(ROOT, "heat_amd", "cf", "benchmarks")
    want = {"AmazonBooks/MF_CCL/configs/config0.yaml": (64, 16, 1.0, 5, 2), "Yelp18/MF_CCL/configs/config0.yaml": (128, 64, 0.1, 8, 10),
            "Gowalla/MF_CCL/configs/config0.yaml": (128, 64, 0.1, 8, 10), "Gowalla/MF_CCL/configs/config_pr1.yaml": (64, 16, 1.0, 5, 2)}
    for rel, (d, n, clip, ep, ev) in want.items():
        c = utils.load_config(os.path.join(base, rel))["model_config"]
        assert (c["embedding_dim"], c["num_negs"], c["clip_val"], c["epochs"], c["eval_interval"]) == (d, n, clip, ep, ev)
        for k in ["max_his", "neg_sampler", "tile_size", "refresh_interval", "embedding_regularizer", "milestones",
                  "learning_rate", "seed"]:
            assert k in c


def test_synthetic_graph_properties():
    from heat_amd.cf import synthetic
    g = synthetic.make_graph(500, 800, 9000, seed=3)
    assert g.clicks.shape == (9000, 2) and g.clicks.dtype == np.uint64
    u = g.clicks[:, 0].astype(np.int64)
    assert (np.diff(u) >= 0).all()                                  # LightGCN order: grouped by user
    key = g.clicks[:, 0] * np.uint64(800) + g.clicks[:, 1]
    assert np.unique(key).size == 9000                              # no duplicate (user,item)
    assert (np.bincount(u, minlength=500) >= 1).all()
    g2 = synthetic.make_graph(500, 800, 9000, seed=3)
    assert np.array_equal(g.clicks, g2.clicks) and np.array_equal(g.test_items, g2.test_items)   # seeded
    # test items are disjoint from the user's train items
    tp, ep = g.train_indptr.astype(np.int64), g.test_indptr.astype(np.int64)
    for usr in range(0, 500, 37):
        assert not set(g.train_items[tp[usr]:tp[usr + 1]].tolist()) & set(g.test_items[ep[usr]:ep[usr + 1]].tolist())


def test_native_lightgcn_ingest_matches_the_reference_loop(tmp_path):
    """heat_cf_parse_lightgcn vs a restatement of the reference's per-line loop (cf/datasets.py:46-78): same pairs in
    file order, empty users, trailing separators, CRLF; malformed input raises."""
    from heat_amd import abi
    rng = np.random.default_rng(0)
    lines, want = [], []
    for u in range(3000):
        n = int(rng.integers(0, 40)) if u % 97 else 0          # some users without items
        items = rng.integers(0, 50000, size=n).tolist()
        tail = " " if u % 5 == 0 else ""
        eol = "\r\n" if u % 7 == 0 else "\n"
        lines.append(" ".join([str(u)] + [str(i) for i in items]) + tail + eol)
        want.extend([u, i] for i in items)
    path = tmp_path / "train.txt"
    path.write_text("".join(lines))
    clicks, line_user, line_start = abi.parse_lightgcn(str(path))
    assert clicks.dtype == np.uint64 and clicks.tolist() == want
    assert line_user.tolist() == list(range(3000)) and line_start[-1] == len(want)
    assert line_start[97] == line_start[98]                    # user 97 has no items
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    c, lu, ls = abi.parse_lightgcn(str(empty))
    assert c.shape == (0, 2) and lu.size == 0 and ls.tolist() == [0]
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1 2\n1 x 3\n")
    with pytest.raises(ValueError):
        abi.parse_lightgcn(str(bad))
    with pytest.raises(ValueError):
        abi.parse_lightgcn(str(tmp_path / "missing.txt"))


def test_launch_plan_host_logic():
    """heat_cf_plan (no GPU): kernel variant table, asynchrony cap and AUTO update policy."""
    from heat_amd import abi
    A = dict(emb_dim=64, num_negs=16, num_users=52643, num_items=91599, train_size=2380730)
    p = abi.plan(resident_workgroups=256 * 12, **A)
    assert (p["lanes_per_row"], p["groups_per_wave"], p["waves_per_workgroup"], p["negative_capacity"]) == (16, 4, 1, 16)
    assert p["cap_items"] == int(0.56 * 91599 / 17) == 3017 and p["cap_users"] == 2380730 // 256
    # a user shard of an 8-GPU job (297 591 interactions): a stream still walks >= 256 interactions
    assert abi.plan(resident_workgroups=256 * 12, **dict(A, num_users=6580, train_size=297591))["streams"] == 297591 // 256 == 1162
    assert p["streams"] == 3017 and p["update_mode"] == "ATOMIC_POS" and p["update_bits"] == 0xC and p["coherence"] == "device"
    # the plan names the bound that set the stream count
    assert p["binding"] == "in-flight touches per item row"
    assert abi.plan(resident_workgroups=256 * 12, **dict(A, num_users=6580, train_size=297591))["binding"] == "interactions per stream"
    assert abi.plan(resident_workgroups=1024, **A)["binding"] == "resident workgroups"
    assert abi.plan(num_streams=77, **A)["binding"] == "num_streams"
    # random-tile sampler (its sampling() call): the tile's weight deltas live in LDS where tile_size x emb_dim x 4 B <= 128 KB
    T_ = dict(A, neg_sampler=1, tile_size=512, refresh_interval=8192)
    LDS = abi.FLAG_SAMPLING_CALL | abi.FLAG_TILE_LDS
    assert abi.plan(flags=LDS, **T_)["tile_in_lds"] is True
    assert abi.plan(flags=abi.FLAG_SAMPLING_CALL, **T_)["tile_in_lds"] is False            # opt-in since round 3
    assert abi.plan(flags=abi.FLAG_TILE_LDS, **T_)["tile_in_lds"] is False                 # ignore_pos_sampling never uses the tile
    assert abi.plan(flags=LDS | abi.FLAG_TILE_GLOBAL, **T_)["tile_in_lds"] is False
    assert abi.plan(flags=LDS, **dict(T_, tile_size=1024))["tile_in_lds"] is False   # 256 KB
    assert abi.plan(flags=LDS, emb_dim=128, num_negs=64, num_users=31668, num_items=38048,
                    train_size=1237259, neg_sampler=1, tile_size=512)["tile_in_lds"] is False           # multi-wave variant
    # fewer resident workgroups than the cap: the chip is the limit
    assert abi.plan(resident_workgroups=1024, **A)["streams"] == 1024
    # forcing many streams pushes the expected negative-row collisions past 0.56 -> every item row goes atomic
    assert abi.plan(num_streams=8192, **A)["update_mode"] == "ATOMIC_WG"
    assert abi.plan(num_streams=8192, update_mode=abi.UPDATE_ATOMIC_POS, **A)["update_bits"] == 0xC
    assert abi.plan(flags=abi.FLAG_SERIAL, **A)["streams"] == 1
    # plain (non-coherent) row traffic cannot use atomics
    assert abi.plan(coherence=abi.COHERENCE_PLAIN, **A)["update_mode"] == "OVERWRITE"
    with pytest.raises(ValueError):
        abi.plan(coherence=abi.COHERENCE_PLAIN, update_mode=abi.UPDATE_ATOMIC_WG, **A)
    # the late re-read applies to plain negative-row stores: bit 4 together with a negative-row atomic is refused
    assert abi.plan(update_mode=16 + 0x10, **A)["update_bits"] == 0x10
    with pytest.raises(ValueError):
        abi.plan(update_mode=16 + 0x1E, **A)
    # variant table: Yelp18 (d128, N64) -> 32 lanes/row, 4 groups x 8 waves; synthetic-HBM (d256, N100) -> 13 groups x 8 waves (4 % capacity slack)
    y = abi.plan(emb_dim=128, num_negs=64, num_users=31668, num_items=38048, train_size=1237259)
    # 65 rows per interaction: 0.45 in-flight touches per item row with the late re-read write-back (263 streams) and at
    # least 5600 interactions of the epoch per stream (220 streams)
    assert y["cap_items"] == int(0.45 * 38048 / 65) == 263 and y["cap_users"] == 1237259 // 5600 == 220
    assert (y["lanes_per_row"], y["groups_per_wave"], y["waves_per_workgroup"], y["streams"]) == (32, 4, 8, 220)
    assert abi.plan(emb_dim=128, num_negs=64, num_users=29858, num_items=40981, train_size=810128)["streams"] == 144
    assert y["update_mode"] == "REREAD_POS" and y["update_bits"] == 0x1C
    # the plan says how its bounds relate to the step size they were measured at (profiles/r03_lr_regime.txt): the yaml's
    # l_r 0.01 everywhere, up to 0.1 with <= 17 rows per interaction; wide interactions above 0.01 walk proportionally longer
    # slices (an extrapolation, named as such), anything further is reported as outside the measured range
    Y = dict(emb_dim=128, num_negs=64, num_users=31668, num_items=38048, train_size=1237259)
    assert y["regime"].startswith("measured") and p["regime"].startswith("measured")
    assert abi.plan(l_r=0.1, **A)["regime"].startswith("measured") and abi.plan(l_r=0.1, **A)["streams"] == 3017
    y3 = abi.plan(l_r=0.03, **Y)
    assert y3["regime"].startswith("extrapolated") and y3["cap_users"] == 1237259 // 16800 == 73 and y3["streams"] == 73
    assert abi.plan(l_r=0.5, **A)["regime"].startswith("outside") and abi.plan(l_r=0.1, **Y)["regime"].startswith("outside")
    # ... without device-coherent row traffic there is neither a fresh value to re-read nor an atomic: the reference's
    # literal overwrite, which stays at a worker count the reference itself could have
    yp = abi.plan(emb_dim=128, num_negs=64, num_users=31668, num_items=38048, train_size=1237259, coherence=abi.COHERENCE_PLAIN)
    assert yp["cap_items"] == int(0.15 * 38048 / 65) and yp["update_mode"] == "OVERWRITE" and yp["streams"] == 64
    assert abi.plan(update_mode=abi.UPDATE_OVERWRITE, **A)["streams"] == 64
    # few streams on a large table: collisions are rare, positives-atomic is enough
    assert abi.plan(emb_dim=128, num_negs=64, num_users=31668, num_items=38048, train_size=1237259, num_streams=64)["update_mode"] == "ATOMIC_POS"
    s_ = abi.plan(emb_dim=256, num_negs=100, num_users=10_000_000, num_items=1_000_000, train_size=200_000_000,
                  resident_workgroups=256)
    assert (s_["lanes_per_row"], s_["groups_per_wave"], s_["waves_per_workgroup"], s_["streams"]) == (64, 13, 8, 256)
    assert s_["update_mode"] == "ATOMIC_POS"          # 256 x 101 / 1 M = 0.03 in-flight touches per item row
    # masked shapes: emb_dim 20 -> 8 lanes/row (5 used); 5 negatives -> 1 group of 8 rows
    m = abi.plan(emb_dim=20, num_negs=5, num_users=100, num_items=1000, train_size=10)
    assert (m["lanes_per_row"], m["negative_capacity"]) == (8, 8)
    # the aggregator sizes itself by the single-wave table and spreads that capacity over up to 4 waves where compiled:
    # d128 / 64 negatives <32,32>x1 -> <32,8>x4; AmazonBooks config <16,4>x1 -> <16,1>x4; its streams are not rounded to CUs
    ag = abi.plan(emb_dim=128, num_negs=64, num_users=31668, num_items=38048, train_size=10, use_aggregator=True)
    assert (ag["lanes_per_row"], ag["groups_per_wave"], ag["waves_per_workgroup"]) == (32, 8, 4)
    ab = abi.plan(emb_dim=64, num_negs=16, num_users=52643, num_items=91599, train_size=2380730, use_aggregator=True)
    assert (ab["lanes_per_row"], ab["groups_per_wave"], ab["waves_per_workgroup"]) == (16, 1, 4)
    assert ab["streams"] % 256 != 0
    # outside the compiled family
    for bad in (dict(emb_dim=6), dict(emb_dim=260), dict(num_negs=500)):
        cfg = dict(emb_dim=64, num_negs=16, num_users=10, num_items=10, train_size=1)
        cfg.update(bad)
        with pytest.raises(abi.HeatError):
            abi.plan(**cfg)
    with pytest.raises(ValueError):
        abi.plan(emb_dim=64, num_negs=0, num_users=10, num_items=10, train_size=1)
    with pytest.raises(ValueError):
        abi.plan(emb_dim=64, num_negs=4, num_users=10, num_items=10, train_size=1, milestones=())


def test_ingest_cache_roundtrip(tmp_path):
    """The binary side-car written next to a LightGCN file reproduces the parse and is invalidated when the file changes."""
    import shutil
    from heat_amd.cf.cf_config import CFConfig
    from heat_amd.cf.datasets import ClickDataset
    src = os.path.join(ROOT, "tests", "golden", "tiny_lightgcn", "train.txt")
    path = tmp_path / "train.txt"
    shutil.copy(src, path)
    with contextlib.redirect_stdout(io.StringIO()):
        a = ClickDataset(str(path), config=CFConfig(emb_dim=64, num_negs=4, max_his=4, milestones=[10]), seed=1)
        assert os.path.exists(str(path) + ".heatcf.npz")
        b = ClickDataset(str(path), config=CFConfig(emb_dim=64, num_negs=4, max_his=4, milestones=[10]), seed=1)
        assert np.array_equal(a.click_dataset, b.click_dataset) and a.user_items_dic == b.user_items_dic
        path.write_text(path.read_text() + "4 1 2\n")            # the source changed: the cache must not be used
        c = ClickDataset(str(path), config=CFConfig(emb_dim=64, num_negs=4, max_his=4, milestones=[10]), seed=1)
    assert c.click_dataset.shape[0] == a.click_dataset.shape[0] + 2 and c.user_items_dic[4] == [1, 2]


def test_metrics_cache_follows_the_test_lists():
    """evaluate_topk keeps the flattened test lists on the test-data object between evaluations (cf/main.py scores the same
    test set every eval_interval epochs); a changed list must be noticed, and the cached path must give the same numbers."""
    import types
    rng = np.random.default_rng(3)
    U, I, k = 300, 500, 20
    dic = {u: rng.choice(I, size=int(rng.integers(1, 30)), replace=False).tolist() for u in range(U)}
    test = types.SimpleNamespace(user_items_dic=dic)
    top = np.argsort(rng.random((U, I)), axis=1)[:, :k].astype(np.uint32)
    ms = ["Recall(k=20)", "NDCG(k=20)", "HitRate(k=10)"]
    first = M.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
    assert "_heat_metrics_cache" in test.__dict__
    assert M.evaluate_topk(test, top, ms, quiet=True, by_user_id=True) == first          # served from the cache
    fresh = types.SimpleNamespace(user_items_dic={u: list(v) for u, v in dic.items()})
    assert M.evaluate_topk(fresh, top, ms, quiet=True, by_user_id=True) == first
    dic[7] = dic[7] + [int(top[7, 0])] if int(top[7, 0]) not in dic[7] else dic[7][:-1]   # one list changes length
    changed = M.evaluate_topk(test, top, ms, quiet=True, by_user_id=True)
    want = M.evaluate_topk(types.SimpleNamespace(user_items_dic={u: list(v) for u, v in dic.items()}), top, ms, quiet=True, by_user_id=True)
    assert changed == want and changed != first
