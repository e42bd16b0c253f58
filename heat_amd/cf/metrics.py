"""Host-side mirror of the reference's cf/metrics.py:5-159 (same metric names, same definitions, same
`evaluate_metrics(train_data, test_data, sim_matrix, metrics)` entry), pinned to the reference by
tests/golden/metrics_golden.json.  `evaluate_topk` is the same evaluation from a top-k id matrix (GPU path)."""
import re

import numpy as np


def _parse(metric):
    m = re.fullmatch(r"\s*([A-Za-z0-9]+)\(k=(\d+)\)\s*", metric)
    if not m or m.group(1) not in _METRICS:
        raise NotImplementedError('metrics={} not implemented.'.format(metric))
    return _METRICS[m.group(1)], int(m.group(2))


def _recall(top, true, k):
    return len(set(true) & set(top[:k])) / (len(true) + 1e-12)                  # metrics.py:44-48


def _normalized_recall(top, true, k):
    return len(set(true) & set(top[:k])) / min(k, len(true) + 1e-12)            # :56-60


def _precision(top, true, k):
    return len(set(true) & set(top[:k])) / (k + 1e-12)                          # :68-72


def _f1(top, true, k):
    p, r = _precision(top, true, k), _recall(top, true, k)
    return 2 * p * r / (p + r + 1e-12)                                          # :80-84


def _dcg(top, true, k):
    true = set(true)
    dcg = 0
    for i, item in enumerate(top[:k]):
        if item in true:
            dcg += 1 / np.log(2 + i)                                            # :99 natural log
    return dcg


def _ndcg(top, true, k):
    idcg = _dcg(true[:k], true, k)                                              # :111 ideal = first k true items
    return _dcg(top, true, k) / (idcg + 1e-12)


def _mrr(top, true, k):
    true = set(true)
    mrr = 0
    for i, item in enumerate(top[:k]):
        if item in true:
            mrr += 1 / (i + 1.0)                                                # :125-127 (sum, not first hit)
    return mrr


def _hit_rate(top, true, k):
    return 1 if len(set(true) & set(top[:k])) > 0 else 0


def _map(top, true, k):
    true = set(true)
    pos, precision = 0, 0
    for i, item in enumerate(top[:k]):
        if item in true:
            pos += 1
            precision += pos / (i + 1.0)
    return precision / (pos + 1e-12)


_METRICS = {"Recall": _recall, "NormalizedRecall": _normalized_recall, "Precision": _precision, "F1": _f1,
            "DCG": _dcg, "NDCG": _ndcg, "MRR": _mrr, "HitRate": _hit_rate, "MAP": _map}


def _score_per_user(top_k_items, true_items, callers):
    """The reference's scoring loop, one Python call per (user, metric): metrics.py:31-34.  Kept as the definition the
    vectorised scorer below is checked against (tests/test_frontend_cpu.py); ~2 s per evaluation at AmazonBooks size."""
    return np.array([[fn([int(x) for x in top], true, k) for fn, k in callers] for top, true in zip(top_k_items, true_items)])


def _flatten_true(true_items, n, cache):
    """(lengths, concatenated ids) of the first n test lists.  The lists are Python lists of a dict that does not change between
    the evaluations of a run (cf/main.py evaluates the same test set every eval_interval epochs), and walking 500 k ints
    through the interpreter is most of an evaluation's host time — so the arrays are kept in `cache` (a dict the caller
    hangs on its test-data object) under a fingerprint of the lists: count, total length, first and last id of every 64th."""
    probe = true_items[:n:64]
    key = (n, tuple((len(t), t[0] if len(t) else -1, t[-1] if len(t) else -1) for t in probe))
    if cache is not None and cache.get("key") == key and cache.get("total") == sum(len(t) for t in true_items[:n]):
        return cache["n_true"], cache["flat_true"]
    n_true = np.fromiter((len(t) for t in true_items[:n]), dtype=np.int64, count=n)
    flat_true = np.fromiter((int(x) for t in true_items[:n] for x in t), dtype=np.int64, count=int(n_true.sum()))
    if cache is not None:
        cache.update(key=key, total=int(n_true.sum()), n_true=n_true, flat_true=flat_true)
    return n_true, flat_true


def _score_vectorised(top_k_items, true_items, callers, cache=None):
    """Same per-user values, bit for bit, from a [users, k] hit matrix: membership by one sorted-key lookup, every sum over
    the rank positions accumulated position by position in the loop order of the functions above (adding 0.0 for a miss
    is exact), so the averages equal the per-user loop's.  Needs distinct ids per top-k row (a top-k list has them; _score
    checks and falls back to the loop otherwise)."""
    n = min(len(top_k_items), len(true_items))
    kmax = max(k for _, k in callers)
    top = np.asarray(top_k_items)[:n, :kmax].astype(np.int64)
    n_true, flat_true = _flatten_true(true_items, n, cache)
    span = int(max(top.max(initial=0), flat_true.max(initial=0))) + 1
    row_of_true = np.repeat(np.arange(n, dtype=np.int64), n_true)
    keys = np.unique(row_of_true * span + flat_true)
    query = np.arange(n, dtype=np.int64)[:, None] * span + top
    pos = np.searchsorted(keys, query)
    hits = (keys[np.minimum(pos, max(keys.size - 1, 0))] == query) if keys.size else np.zeros_like(query, dtype=bool)
    hits &= top >= 0
    ranks = np.arange(kmax)
    log_w = 1 / np.log(2 + ranks)                                                # _dcg's weights
    ideal = np.concatenate([[0.0], np.zeros(kmax)])
    for i in range(kmax):
        ideal[i + 1] = ideal[i] + log_w[i]                                       # _dcg(true[:k], true, k) by prefix length
    out = np.zeros((n, len(callers)))
    for c, (fn, k) in enumerate(callers):
        h = hits[:, :k]
        inter = h.sum(axis=1)
        if fn is _recall:
            out[:, c] = inter / (n_true + 1e-12)
        elif fn is _normalized_recall:
            out[:, c] = inter / np.minimum(k, n_true + 1e-12)
        elif fn is _precision:
            out[:, c] = inter / (k + 1e-12)
        elif fn is _f1:
            p, r = inter / (k + 1e-12), inter / (n_true + 1e-12)
            out[:, c] = 2 * p * r / (p + r + 1e-12)
        elif fn is _hit_rate:
            out[:, c] = inter > 0
        elif fn in (_dcg, _ndcg, _mrr):
            w = 1 / (ranks[:k] + 1.0) if fn is _mrr else log_w[:k]
            acc = np.zeros(n)
            for i in range(k):
                acc += np.where(h[:, i], w[i], 0.0)
            out[:, c] = acc / (ideal[np.minimum(n_true, k)] + 1e-12) if fn is _ndcg else acc
        elif fn is _map:
            found = np.zeros(n)
            precision = np.zeros(n)
            for i in range(k):
                found += h[:, i]
                precision += np.where(h[:, i], found / (i + 1.0), 0.0)
            out[:, c] = precision / (found + 1e-12)
        else:
            raise NotImplementedError(fn.__name__)
    return out


def _score(top_k_items, test_items_dic, test_user_ids, metrics, quiet=False, cache=None):
    callers = [_parse(m) for m in metrics]
    true_items = [test_items_dic[u] for u in test_user_ids]
    # metrics.py:31-32 zips ROW i of the top-k matrix with the i-th test user (not with row `user id`)
    rows = np.asarray(top_k_items)
    srt = np.sort(rows, axis=1) if rows.ndim == 2 and rows.size else rows
    if rows.ndim != 2 or rows.shape[0] == 0 or (srt[:, 1:] == srt[:, :-1]).any():    # repeated ids in a row: set semantics
        results = _score_per_user(top_k_items, true_items, callers)
    else:
        results = _score_vectorised(rows, true_items, callers, cache)
    average_result = np.average(results, axis=0).tolist()
    if not quiet:
        print('[Metrics] ' + ' - '.join('{}: {:.6f}'.format(k, v) for k, v in zip(metrics, average_result)))
    return dict(zip(metrics, average_result))


def evaluate_metrics(train_data, test_data, sim_matrix, metrics, quiet=False):
    """metrics.py:5-36: mask train items with -inf (in place), argpartition top-k, sort, score per user, average."""
    if not quiet:
        print(f'Evaluating metrics {metrics} ...')
    train_items_dic = train_data.user_items_dic
    test_items_dic = test_data.user_items_dic
    test_user_ids = list(test_items_dic.keys())
    max_top_k = max(_parse(m)[1] for m in metrics)
    for u in test_user_ids:
        sim_matrix[u, train_items_dic[u]] = -np.inf                                            # :24
    item_indices = np.argpartition(-sim_matrix, max_top_k)[:, 0:max_top_k]                      # :26
    part = sim_matrix[np.arange(item_indices.shape[0])[:, None], item_indices]
    sorted_ids = np.argsort(-part, axis=1)                                                      # :28
    top_k_items = item_indices[np.arange(sorted_ids.shape[0])[:, None], sorted_ids]
    return _score(top_k_items, test_items_dic, test_user_ids, metrics, quiet, _cache_of(test_data))


def _cache_of(test_data):
    """a dict on the test-data object for _flatten_true (None for objects that cannot carry one)"""
    try:
        return test_data.__dict__.setdefault("_heat_metrics_cache", {})
    except AttributeError:
        return None


def evaluate_topk(test_data, top_k_items, metrics, quiet=False, by_user_id=False):
    """Same scoring from a [num_users, k] id matrix (e.g. Engine.topk with the train CSR as mask).
    by_user_id=False reproduces the reference's row pairing (row i <-> i-th test user); True pairs row `u` with user u."""
    test_items_dic = test_data.user_items_dic
    test_user_ids = list(test_items_dic.keys())
    if max(_parse(m)[1] for m in metrics) > top_k_items.shape[1]:
        raise ValueError("top_k_items holds fewer columns than the largest k requested")
    rows = top_k_items[np.asarray(test_user_ids, dtype=np.int64)] if by_user_id else top_k_items
    return _score(rows, test_items_dic, test_user_ids, metrics, quiet, _cache_of(test_data))
