"""Near-tie analysis for structure mismatches.

The reference sums float32 gradients sequentially per candidate (node.cpp:336-352): its split scores carry a relative
error of order eps32*sqrt(n) and it can therefore pick a candidate that is NOT the true maximum when two candidates'
true scores are closer than that (the reference even disagrees with ITSELF between OMP_NUM_THREADS=3 and 8 on such
inputs, because the thread count changes the rounding of its L2 standardisation, math_ops.cpp:255-300).  The product
evaluates scores from exact integer sums.  `explain_first_mismatch` locates the first split where two ensembles differ,
re-evaluates both candidates in float64 on exactly the rows of that node, and reports the relative gap.
"""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)


def _conds(e, policy, tree, leaf, d):
    row = tree if policy == "oblivious" else leaf
    return (bool(e["is_numerics"][row, d]), int(e["feature_indices"][row, d]), e["feature_values"][row, d],
            bytes(e["categorical_values"][row, d]))


def _test(X, Xc, cond):
    is_num, f, v, cat = cond
    if is_num:
        return X[:, f] > v
    return Xc[:, f] == np.array(cat, dtype="S128")


def first_mismatch(e_a, e_b, policy):
    """(tree, leaf, depth) of the first differing condition in tree order, or None."""
    T = min(len(e_a["tree_indices"]), len(e_b["tree_indices"]))
    for t in range(T):
        la = int(e_a["tree_indices"][t])
        if la != int(e_b["tree_indices"][t]):
            return (t - 1 if t else 0, None, None)
        n_a = (int(e_a["tree_indices"][t + 1]) if t + 1 < len(e_a["tree_indices"]) else len(e_a["values"])) - la
        n_b = (int(e_b["tree_indices"][t + 1]) if t + 1 < len(e_b["tree_indices"]) else len(e_b["values"])) - la
        best = None
        for leaf in range(la, la + min(n_a, n_b)):
            row = t if policy == "oblivious" else leaf
            da, db = int(e_a["depths"][row]), int(e_b["depths"][row])
            for d in range(min(da, db)):
                if _conds(e_a, policy, t, leaf, d) != _conds(e_b, policy, t, leaf, d):
                    if best is None or d < best[2]:
                        best = (t, leaf, d)
                    break
            else:
                if da != db and (best is None or min(da, db) < best[2]):
                    best = (t, leaf, min(da, db))
            if policy == "oblivious":
                break
            if best is not None:
                break   # greedy leaves are in depth-first order: once one differs, later leaf indices no longer correspond
        if best is not None:
            return best
        if n_a != n_b:
            return (t, None, None)
    return None


def explain_first_mismatch(case, X, Xc, G_t, e_ref, e_prod):
    """G_t: the gradients the mismatching tree was fitted on.  Returns dict(gap_rel, tol, n_rows, ...)."""
    policy = case["policy"].lower()
    mm = first_mismatch(e_ref, e_prod, policy)
    if mm is None:
        return None
    t, leaf, d = mm
    if leaf is None:
        return dict(tree=t, explained=False, why="leaf layout differs before any condition does")
    g = np.asarray(G_t, np.float64).reshape(len(G_t), -1)
    if case["score"].lower() == "l2":
        bg = (g - g.mean(axis=0)) / (g.std(axis=0, ddof=1) + 1e-8)
    else:
        bg = g
    # node set: rows that satisfy the common path prefix (greedy: this leaf's prefix; oblivious: every node of level d)
    n = len(g)
    if policy == "greedy":
        rows = np.ones(n, bool)
        for k in range(d):
            rows &= (_test(X, Xc, _conds(e_ref, policy, t, leaf, k)) == bool(e_ref["inequality_directions"][leaf, k]))
        node_ids = np.where(rows, 0, -1)
    else:
        node_ids = np.zeros(n, np.int64)
        for k in range(d):
            node_ids = node_ids * 2 + _test(X, Xc, _conds(e_ref, policy, t, leaf, k)).astype(np.int64)

    def score(cond):
        right = _test(X, Xc, cond)
        tot = 0.0
        for nid in np.unique(node_ids[node_ids >= 0]):
            sel = node_ids == nid
            x = 0.0
            for side in (sel & right, sel & ~right):
                c = int(side.sum())
                if c:
                    x += float((bg[side].sum(axis=0) ** 2).sum()) / c
            tot += np.sqrt(x) if case["score"].lower() == "cosine" else x
        return tot

    d_ref = int(e_ref["depths"][t if policy == "oblivious" else leaf])
    d_prod = int(e_prod["depths"][t if policy == "oblivious" else leaf])
    if d >= d_ref or d >= d_prod:
        # One side made this node a leaf, the other split it (greedy: split iff best gain >= 0, fitter.cpp:357).  That happens
        # when the best candidate's gain is zero in exact arithmetic -- typically a node whose rows all fall on one side of
        # every candidate, so that the "split" score equals the parent score -- and the reference's float32 sums land a hair
        # below or above zero.  Explained iff the gain of the split that one side took is within the rounding tolerance.
        if policy != "greedy":
            return dict(tree=t, leaf=leaf, depth=d, explained=False, why="one side stopped splitting here")
        taker = e_ref if d < d_ref else e_prod
        sel = node_ids >= 0
        n_rows = int(sel.sum())
        S = bg[sel].sum(axis=0)
        parent = float((S ** 2).sum()) / max(n_rows, 1)
        if case["score"].lower() == "cosine":
            parent = float(np.sqrt(parent))
        s_split = score(_conds(taker, policy, t, leaf, d))
        gain = s_split - (0.0 if d == 0 else parent)
        tol = (4.0 * EPS32 * np.sqrt(max(n_rows, 1)) + 4.0 * EPS32) * max(abs(parent), abs(s_split), 1e-300)
        return dict(tree=t, leaf=leaf, depth=d, n_rows=n_rows, gain=gain, tol=tol, gap_rel=abs(gain) / max(abs(parent), 1e-300),
                    explained=bool(abs(gain) <= tol), product_is_true_max=True, why="zero-gain split decided by rounding")
    s_ref, s_prod = score(_conds(e_ref, policy, t, leaf, d)), score(_conds(e_prod, policy, t, leaf, d))
    n_rows = int((node_ids >= 0).sum())
    gap = abs(s_ref - s_prod) / max(abs(s_ref), abs(s_prod), 1e-300)
    tol = 4.0 * EPS32 * np.sqrt(max(n_rows, 1)) + 4.0 * EPS32
    return dict(tree=t, leaf=leaf, depth=d, n_rows=n_rows, score_ref=s_ref, score_prod=s_prod, gap_rel=gap, tol=tol,
                explained=bool(gap <= tol), product_is_true_max=bool(s_prod >= s_ref))
