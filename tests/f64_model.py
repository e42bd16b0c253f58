"""float64 analytic model of ONE interaction of the reference's fused step
(models/matrix_factorization.cpp:15-181 + optimizers/sgd.cpp:14-26 +
behavior_aggregators/behavior_aggregators.cpp:51-153), used to bound the fp32 error of both the
oracle and the HIP path.  Test infrastructure only."""
import numpy as np


def step_f64(user_w, item_w, user_g, item_g, user, pos, negs, *, clip, lr, agg=None):
    """Mutates the float64 tables in place, exactly in the reference's order.  `agg` (optional) is a dict
    with his (list of item ids), w0 (dxd float64, mutated), state dict(iteration, accu), gamma, agg_lr."""
    d = user_w.shape[1]
    u = user_w[user].copy()
    p = item_w[pos].copy()
    if agg is not None:
        his = agg["his"]
        means = item_w[his].sum(axis=0) * (1.0 / len(his))
        f = means @ agg["w0"]
        u = agg["gamma"] * u + (1 - agg["gamma"]) * f
        agg["state"]["iteration"] += 1
    uu, pp, up = u @ u, p @ p, u @ p
    eps = float(np.float32(1e-8))
    un, pn = np.sqrt(max(uu, eps)), np.sqrt(max(pp, eps))
    upu = (uu * p - up * u) / (un ** 3 * pn)
    upp = -(pp * u - up * p) / (un * pn ** 3)
    N = len(negs)
    nrows = np.stack([item_w[n].copy() for n in negs])
    und = nrows @ u
    nnd = (nrows * nrows).sum(axis=1)
    nn = np.sqrt(np.where(nnd < eps, eps, nnd))
    score = (und / (un * nn) - up / (un * pn)) * float(np.float32(1.0 / 0.07))
    m = score.max()
    es = np.exp(score - m)
    Z = es.sum() + np.exp(-m)
    loss = m + np.log(Z)
    lg = es / Z * float(np.float32(1.0 / 0.07))
    gu = user_g[user].copy()
    gp = item_g[pos].copy()
    for k in range(N):
        nid = negs[k]
        n = nrows[k]
        gn = item_g[nid].copy()
        unu = (uu * n - und[k] * u) / (un ** 3 * nn[k])
        unn = (nnd[k] * u - und[k] * n) / (un * nn[k] ** 3)
        gu += lg[k] * (unu - upu)
        gp += lg[k] * upp
        gn += lg[k] * unn
        gn = np.clip(gn, -clip, clip)
        n = n - lr * gn
        nrows[k] = n
        item_w[nid] = n
        item_g[nid] = gn
    if agg is not None:
        st = agg["state"]
        fgrad = gu * (1 - agg["gamma"])
        st["accu"] += np.outer(means, fgrad)
        if st["iteration"] > 0 and st["iteration"] % 32 == 0:
            agg["w0"] -= agg["agg_lr"] * (st["accu"] / 32)
            st["accu"][:] = 0
        gu = gu * agg["gamma"]
    gu = np.clip(gu, -clip, clip)
    u = u - lr * gu
    gp = np.clip(gp, -clip, clip)
    p = p - lr * gp
    user_w[user] = u
    user_g[user] = gu
    item_w[pos] = p
    item_g[pos] = gp
    return loss


def loss_only_f64(u, p, nrows):
    """log(1 + sum_k exp((cos(u,n_k) - cos(u,p))/0.07)) in float64 — for finite-difference gradient checks."""
    eps = 1e-8
    un = np.sqrt(max(u @ u, eps))
    pn = np.sqrt(max(p @ p, eps))
    nn = np.sqrt(np.maximum((nrows * nrows).sum(axis=1), eps))
    score = ((nrows @ u) / (un * nn) - (u @ p) / (un * pn)) / 0.07
    return np.log1p(np.exp(score).sum())
