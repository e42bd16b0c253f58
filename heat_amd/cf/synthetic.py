"""Seeded synthetic (user,item) graphs of the reference datasets' shapes.

The real LightGCN datasets are not available offline, so every measurement and parity run uses graphs made
here: per-user degree from a clipped power law matched to the mean degree, item popularity Zipf(1.0) over a
random permutation of item ids, no duplicate (user,item) pairs, stored in LightGCN order (grouped by user,
cf/datasets.py:63-67), with an 80/20 per-user train/test split for Recall/NDCG.
"""
from dataclasses import dataclass

import numpy as np

# name -> (num_users, num_items, train interactions, emb_dim, num_negs)   (README.md:77-79; paper §5.1; BASELINE.json)
SHAPES = {
    "amazonbooks": (52643, 91599, 2380730, 64, 16),
    "gowalla": (29858, 40981, 810128, 128, 64),      # this fork's Gowalla/MF_CCL/configs/config0.yaml:9-11 (d=128, 64 negatives)
    "gowalla_pr1": (29858, 40981, 810128, 64, 16),   # BASELINE.json configs[0] ("PR1": d=64, 16 negatives; config_pr1.yaml)
    "yelp18": (31668, 38048, 1237259, 128, 64),
    "synthetic_hbm": (10_000_000, 1_000_000, 200_000_000, 256, 100),
}


@dataclass
class Graph:
    num_users: int
    num_items: int
    clicks: np.ndarray        # [train_size, 2] uint64, grouped by user (LightGCN order)
    train_indptr: np.ndarray  # [num_users+1] uint64 CSR of train items per user (same order as clicks)
    test_indptr: np.ndarray   # [num_users+1] uint64
    test_items: np.ndarray    # [n_test] uint32

    @property
    def train_items(self):
        return self.clicks[:, 1].astype(np.uint32)


def make_graph(num_users, num_items, train_size, seed=2022, test_frac=0.2, zipf_s=1.0, deg_alpha=1.8, min_deg=2,
               with_test=True, n_clusters=0, in_cluster=0.8):
    """Returns a Graph with exactly `train_size` train interactions (requires train_size >= num_users).

    n_clusters = 0 is the SURVEY §8d generator (popularity only: the best recommender is the popularity ranking).
    n_clusters > 0 adds latent structure: users and items belong to clusters and a user draws `in_cluster` of its
    items from its own cluster (skewed inside the cluster), the rest from the global Zipf; Recall/NDCG then depend on
    the learned user/item geometry, which makes parity checks more discriminating."""
    if train_size < num_users:
        raise ValueError("train_size must be >= num_users (every user keeps at least one train item)")
    rng = np.random.default_rng(seed)
    total = int(round(train_size / (1.0 - test_frac))) if with_test else train_size
    if total > 0.25 * num_users * num_items:
        raise ValueError("graph too dense: (user,item) pairs are unique, ask for <= 25% of num_users*num_items")
    # degrees: clipped Pareto rescaled to the target mean
    raw = rng.pareto(deg_alpha, size=num_users) + 1.0
    cap = max(min_deg + 1, min(num_items // 4, int(40 * total / num_users)))
    deg = raw * (total / raw.sum())
    for _ in range(8):
        deg = np.clip(deg, min_deg, cap)
        deg *= total / deg.sum()
    deg = np.clip(np.round(deg), min_deg, cap).astype(np.int64)
    # item popularity: Zipf(s) over a random permutation of ids
    ranks = np.arange(1, num_items + 1, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-zipf_s))
    cdf /= cdf[-1]
    perm = rng.permutation(num_items)

    def draw(n):
        return perm[np.searchsorted(cdf, rng.random(n), side="right").clip(0, num_items - 1)]

    if n_clusters > 0:
        user_cl = rng.integers(0, n_clusters, size=num_users)
        item_cl = rng.integers(0, n_clusters, size=num_items)
        order_c = np.argsort(item_cl, kind="stable")
        cl_start = np.searchsorted(item_cl[order_c], np.arange(n_clusters))
        cl_size = np.maximum(1, np.bincount(item_cl, minlength=n_clusters))

    def draw_for(users_arr):
        out = draw(users_arr.size).astype(np.int64)
        if n_clusters > 0:
            inc = rng.random(users_arr.size) < in_cluster
            c = user_cl[users_arr[inc]]
            r = np.floor(cl_size[c] * rng.random(c.size) ** 2).astype(np.int64).clip(0, cl_size[c] - 1)
            out[inc] = order_c[(cl_start[c] + r).clip(0, num_items - 1)]
        return out

    # oversample, dedup (user,item), keep up to deg[u] per user
    over = 1.35 if n_clusters == 0 else 1.8
    users = np.repeat(np.arange(num_users, dtype=np.int64), np.ceil(deg * over).astype(np.int64) + 2)
    items = draw_for(users)
    key = np.unique(users * num_items + items)
    users, items = key // num_items, key % num_items
    # shuffle within user so that the kept subset / split is not item-id ordered
    order = np.lexsort((rng.random(users.size), users))
    users, items = users[order], items[order]
    start = np.searchsorted(users, np.arange(num_users))
    rank_in_user = np.arange(users.size) - start[users]
    keep = rank_in_user < deg[users]
    users, items, rank_in_user = users[keep], items[keep], rank_in_user[keep]
    cnt = np.bincount(users, minlength=num_users)
    # users that ended up empty get one uniformly random item
    empty = np.flatnonzero(cnt == 0)
    if empty.size:
        users = np.concatenate([users, empty])
        items = np.concatenate([items, rng.integers(0, num_items, size=empty.size)])
        rank_in_user = np.concatenate([rank_in_user, np.zeros(empty.size, dtype=np.int64)])
        order = np.argsort(users, kind="stable")
        users, items, rank_in_user = users[order], items[order], rank_in_user[order]
        cnt = np.bincount(users, minlength=num_users)
    # per-user split: the first ceil((1-test_frac)*cnt) entries are train (at least one)
    n_train_u = np.maximum(1, np.ceil((1.0 - test_frac) * cnt).astype(np.int64)) if with_test else cnt
    is_train = rank_in_user < n_train_u[users]
    # hit train_size exactly: move random rows between the splits / drop surplus
    n_tr = int(is_train.sum())
    if n_tr > train_size:
        # drop surplus train rows, never a user's rank-0 row
        cand = np.flatnonzero(is_train & (rank_in_user > 0))
        drop = rng.choice(cand, size=n_tr - train_size, replace=False)
        mask = np.ones(users.size, dtype=bool)
        mask[drop] = False
        users, items, is_train = users[mask], items[mask], is_train[mask]
    elif n_tr < train_size:
        need = train_size - n_tr
        cand = np.flatnonzero(~is_train)
        if cand.size >= need:
            is_train[rng.choice(cand, size=need, replace=False)] = True
        else:
            is_train[cand] = True
            need -= cand.size
            # still short: add fresh unique pairs
            have = set((users * num_items + items).tolist()) if users.size < 5_000_000 else None
            add_u, add_i = [], []
            while need > 0:
                u = rng.integers(0, num_users, size=need * 2)
                i = draw(need * 2)
                for uu, ii in zip(u.tolist(), i.tolist()):
                    k = uu * num_items + ii
                    if have is not None and k in have:
                        continue
                    if have is not None:
                        have.add(k)
                    add_u.append(uu)
                    add_i.append(ii)
                    need -= 1
                    if need == 0:
                        break
            users = np.concatenate([users, np.array(add_u, dtype=np.int64)])
            items = np.concatenate([items, np.array(add_i, dtype=np.int64)])
            is_train = np.concatenate([is_train, np.ones(len(add_u), dtype=bool)])
            order = np.argsort(users, kind="stable")
            users, items, is_train = users[order], items[order], is_train[order]
    tr_u, tr_i = users[is_train], items[is_train]
    te_u, te_i = users[~is_train], items[~is_train]
    clicks = np.stack([tr_u, tr_i], axis=1).astype(np.uint64)
    train_indptr = np.concatenate([[0], np.cumsum(np.bincount(tr_u, minlength=num_users))]).astype(np.uint64)
    test_indptr = np.concatenate([[0], np.cumsum(np.bincount(te_u, minlength=num_users))]).astype(np.uint64)
    assert clicks.shape[0] == train_size
    return Graph(num_users, num_items, np.ascontiguousarray(clicks), train_indptr, test_indptr, te_i.astype(np.uint32))


def make_named(name, seed=2022, scale=1.0, **kw):  # kw: with_test, n_clusters, in_cluster, ...
    """Graph of a named shape; `scale` < 1 shrinks users/items/interactions together (parity-test sizes)."""
    U, I, T, d, N = SHAPES[name]
    U, I, T = max(8, int(U * scale)), max(64, int(I * scale)), max(8, int(T * scale))
    return make_graph(U, I, max(T, U), seed=seed, **kw), d, N


def init_embeddings(num_users, num_items, emb_dim, seed=2022, std=1e-2):
    """N(0, std^2) fp32 tables (cf/models.py:13-16 uses nn.init.normal_(std=1e-2)); numpy Philox stream so the
    same tables are produced on every box."""
    rng = np.random.Generator(np.random.Philox(seed))
    uw = (rng.standard_normal((num_users, emb_dim), dtype=np.float32) * np.float32(std))
    iw = (rng.standard_normal((num_items, emb_dim), dtype=np.float32) * np.float32(std))
    return np.ascontiguousarray(uw), np.ascontiguousarray(iw)


def make_clicks_torch(num_users, num_items, n, device, seed=2022, zipf_s=1.0, per_user=0):
    """Interaction list for shapes too large for the numpy generator (synthetic-HBM config: 10 M users x 1 M items):
    built on the GPU with torch, returned as an int64 [n,2] tensor (same bit pattern as the u64 pairs of the C ABI),
    grouped by user, item popularity Zipf(s) over a random permutation.  (user,item) pairs may repeat; there is no
    test split — this generator feeds bandwidth measurements, not Recall.
    per_user > 0: a SAMPLE of a longer list — n / per_user users drawn over the whole id range, each with per_user
    interactions (the config's 200 M / 10 M = 20), so that the run length per user row is the full list's."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if per_user > 0:
        picked = torch.randint(0, num_users, (max(1, n // per_user),), device=device, generator=gen, dtype=torch.int64)
        users = torch.sort(picked)[0].repeat_interleave(per_user)
        if users.numel() < n:
            users = torch.cat([users, users[-1:].expand(n - users.numel())])
        users = users[:n].contiguous()
    else:
        users = torch.randint(0, num_users, (n,), device=device, generator=gen, dtype=torch.int64)
        users, _ = torch.sort(users)
    ranks = torch.arange(1, num_items + 1, device=device, dtype=torch.float64)
    cdf = torch.cumsum(ranks.pow(-zipf_s), 0)
    cdf /= cdf[-1].clone()
    perm = torch.randperm(num_items, device=device, generator=gen)
    r = torch.rand((n,), device=device, generator=gen, dtype=torch.float64)
    items = perm[torch.searchsorted(cdf, r).clamp_(0, num_items - 1)]
    del r, cdf, ranks
    return torch.stack([users, items], dim=1).contiguous()


def make_history(graph, max_his=100, seed=2022):
    """historical_items [num_users, max_his] u64 and masks [num_users, 1] u64 as cf/datasets.py:58-72 builds them:
    a user with >= max_his train items gets a random sample of max_his of them, otherwise its items padded with the
    last one; masks = min(len, max_his).  Vectorised (numpy Generator instead of random.sample)."""
    rng = np.random.default_rng(seed)
    U = graph.num_users
    tp = graph.train_indptr.astype(np.int64)
    items = graph.train_items.astype(np.uint64)
    his = np.zeros((U, max_his), dtype=np.uint64)
    masks = np.zeros((U, 1), dtype=np.uint64)
    for u in range(U):
        row = items[tp[u]:tp[u + 1]]
        n = row.size
        if n >= max_his:
            his[u] = rng.choice(row, size=max_his, replace=False)
            masks[u, 0] = max_his
        elif n > 0:
            his[u, :n] = row
            his[u, n:] = row[-1]
            masks[u, 0] = n
    return his, masks
