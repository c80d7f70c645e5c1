"""Independent float64 re-evaluation of the split CHOICE at sizes where the brute-force oracle cannot follow.

NumPy only, histogram-based (sufficient statistics), written from the reference's definitions:
  * thresholds: sorted column at the cumulative ranks of `computeQuantiles` (split_candidate_generator.cpp:216-249):
    bin j holds N // (B+1) rows, the first N % (B+1) bins one more; threshold i = column[cum_i - 1], i in [0, B).
  * class code of (row, feature) = #{k : x > t_k} (node.cpp:336: rows with `x > t` go right), so candidate b sends a row to the
    right iff code > b.
  * L2 score of a node and candidate (node.cpp:321-376): |S_L|^2 / n_L + |S_R|^2 / n_R (an empty side contributes 0);
    Cosine (node.cpp:187-251, math_ops.h:552-574): sqrt of the same expression on the RAW gradients.
  * oblivious (fitter.cpp:411-459): sum over the level's nodes, candidates already on the path are -inf, lowest index among
    maxima wins (feature-major, bin-minor); greedy (fitter.cpp:318-357): score * w[feature] - parent score, same arg-max.

Nothing here touches the product or the oracle: it is the third opinion the full-size tests compare the product's stored
(feature, threshold) against.
"""
import numpy as np


def quantile_ranks(n, n_bins):
    counts = np.full(n_bins + 1, n // (n_bins + 1), np.int64)
    counts[: n % (n_bins + 1)] += 1
    return np.maximum(np.cumsum(counts)[:n_bins] - 1, 0)


def quantile_thresholds(X, n_bins):
    """[F, n_bins] float32: exact data values at the reference's ranks."""
    n, F = X.shape
    ranks = quantile_ranks(n, n_bins)
    out = np.empty((F, n_bins), np.float32)
    for f in range(F):
        out[f] = np.sort(X[:, f])[ranks]
    # a signed zero at a rank is stored as +0.0 by the product (the reference keeps whichever of -0.0 / +0.0 its sort leaves there;
    # no comparison x > t can tell them apart: DESIGN.md section 9) -- canonicalise, the checks compare threshold BITS
    out[out == 0] = np.float32(0.0)
    return out


def uniform_thresholds(X, n_bins):
    """[F, n_bins] float32: min + b * step with ONE rounding (the reference's contracted expression, split_candidate_generator.cpp:59-76, Q5);
    step = (max - min) / float(n_bins) in float32."""
    mn, mx = X.min(axis=0).astype(np.float32), X.max(axis=0).astype(np.float32)
    step = ((mx - mn) / np.float32(n_bins)).astype(np.float32)
    b = np.arange(n_bins, dtype=np.float64)
    return (b[None, :] * step.astype(np.float64)[:, None] + mn.astype(np.float64)[:, None]).astype(np.float32)


def class_codes(X, thr):
    """[F, N] int16: #{k : x > t_k} (thresholds ascending, duplicates kept)."""
    n, F = X.shape
    codes = np.empty((F, n), np.int16)
    for f in range(F):
        codes[f] = np.searchsorted(thr[f], X[:, f], side="left")
    return codes


def standardise(G):
    g = np.asarray(G, np.float64)
    return (g - g.mean(axis=0)) / (g.std(axis=0, ddof=1) + 1e-8)


def candidate_scores(codes, node_ids, n_nodes, bg, n_bins, cosine):
    """[n_nodes, F, n_bins] float64 scores of every (node, feature, bin); rows with node_ids < 0 are ignored."""
    F = codes.shape[0]
    D = bg.shape[1]
    C = n_bins + 1
    sel = node_ids >= 0
    nid = node_ids[sel].astype(np.int64)
    g = bg[sel]
    out = np.empty((n_nodes, F, n_bins), np.float64)
    for f in range(F):
        idx = nid * C + codes[f][sel]
        cnt = np.bincount(idx, minlength=n_nodes * C).reshape(n_nodes, C).astype(np.float64)
        S = np.empty((n_nodes, C, D), np.float64)
        for d in range(D):
            S[:, :, d] = np.bincount(idx, weights=g[:, d], minlength=n_nodes * C).reshape(n_nodes, C)
        # right of candidate b = classes b+1 .. B  (suffix sums), left = node total - right
        Sr = np.cumsum(S[:, ::-1, :], axis=1)[:, ::-1, :][:, 1:, :]
        nr = np.cumsum(cnt[:, ::-1], axis=1)[:, ::-1][:, 1:]
        Sl = S.sum(axis=1, keepdims=True) - Sr
        nl = cnt.sum(axis=1, keepdims=True) - nr
        with np.errstate(divide="ignore", invalid="ignore"):
            x = np.where(nl > 0, (Sl ** 2).sum(axis=2) / nl, 0.0) + np.where(nr > 0, (Sr ** 2).sum(axis=2) / nr, 0.0)
        out[:, f, :] = np.sqrt(x) if cosine else x
    return out


def check_oblivious_tree(X, G, e, n_bins, score, report=None, rel_tol=1e-6, gen="Quantile"):
    """Every level's stored (feature, threshold) of the FIRST tree must be the float64 arg-max (lowest index on ties) or within
    rel_tol of the maximum.  Returns the per-level records."""
    cosine = score.lower() == "cosine"
    bg = np.asarray(G, np.float64) if cosine else standardise(G)
    thr = quantile_thresholds(X, n_bins) if gen.lower() == "quantile" else uniform_thresholds(X, n_bins)
    codes = class_codes(X, thr)
    depth = int(np.asarray(e["depths"])[0])
    fi = np.asarray(e["feature_indices"])[0]
    fv = np.asarray(e["feature_values"])[0]
    n = X.shape[0]
    node = np.zeros(n, np.int64)
    used = []
    out = []
    for lvl in range(depth):
        sc = candidate_scores(codes, node, 1 << lvl, bg, n_bins, cosine).sum(axis=0)       # [F, B]
        for (uf, ub) in used:                                                             # same (feature, value) on the path
            sc[uf, thr[uf] == thr[uf, ub]] = -np.inf
        flat = sc.reshape(-1)
        best = int(np.argmax(flat))                                                        # first maximum = lowest index
        bf, bb = divmod(best, n_bins)
        f_st = int(fi[lvl])
        t_bits = fv[lvl:lvl + 1].view(np.uint32)[0]
        hits = np.nonzero(thr[f_st].view(np.uint32) == t_bits)[0]
        rec = dict(level=lvl, want=(bf, bb), stored_feature=f_st, threshold_is_rank_exact=bool(len(hits)))
        assert len(hits), "level %d: stored threshold %r of feature %d is not a rank-exact data value" % (lvl, fv[lvl], f_st)
        sb = int(hits[0])
        rec.update(stored=(f_st, sb), score_best=float(flat[best]), score_stored=float(sc[f_st, sb]))
        rec["gap_rel"] = float((flat[best] - sc[f_st, sb]) / max(abs(flat[best]), 1e-300))
        rec["exact"] = bool((f_st, sb) == (bf, bb))
        assert rec["exact"] or rec["gap_rel"] <= rel_tol, rec
        out.append(rec)
        used.append((f_st, sb))
        node = node * 2 + (X[:, f_st] > fv[lvl])
    if report is not None:
        report.extend(out)
    return out


def check_greedy_nodes(X, G, e, n_bins, score, leaves, levels, rel_tol=1e-6, gen="Quantile"):
    """For the given leaves of a greedy tree: the condition stored at depth `lvl` of the leaf's path must be the float64
    arg-max of `score - parent` on the rows that satisfy the path prefix (feature weights 1)."""
    cosine = score.lower() == "cosine"
    bg = np.asarray(G, np.float64) if cosine else standardise(G)
    thr = quantile_thresholds(X, n_bins) if gen.lower() == "quantile" else uniform_thresholds(X, n_bins)
    fi, fv = np.asarray(e["feature_indices"]), np.asarray(e["feature_values"])
    dirs, dep = np.asarray(e["inequality_directions"]), np.asarray(e["depths"])
    out = []
    codes_cache = {}
    for leaf, lvl in zip(leaves, levels):
        assert lvl < int(dep[leaf])
        rows = np.ones(X.shape[0], bool)
        used = []
        for k in range(lvl):
            rows &= (X[:, fi[leaf, k]] > fv[leaf, k]) == bool(dirs[leaf, k])
            used.append((int(fi[leaf, k]), fv[leaf, k]))
        idx = np.nonzero(rows)[0]
        Xn = X[idx]
        key = (leaf, lvl)
        codes_cache[key] = class_codes(Xn, thr)
        sc = candidate_scores(codes_cache[key], np.zeros(len(idx), np.int64), 1, bg[idx], n_bins, cosine)[0]
        for (uf, uv) in used:
            sc[uf, thr[uf] == uv] = -np.inf
        flat = sc.reshape(-1)
        best = int(np.argmax(flat))
        bf, bb = divmod(best, n_bins)
        f_st = int(fi[leaf, lvl])
        hits = np.nonzero(thr[f_st].view(np.uint32) == fv[leaf, lvl:lvl + 1].view(np.uint32)[0])[0]
        assert len(hits), "leaf %d level %d: stored threshold is not a rank-exact data value" % (leaf, lvl)
        sb = int(hits[0])
        rec = dict(leaf=leaf, level=lvl, n_rows=len(idx), want=(bf, bb), stored=(f_st, sb), score_best=float(flat[best]),
                   score_stored=float(sc[f_st, sb]))
        rec["gap_rel"] = float((flat[best] - sc[f_st, sb]) / max(abs(flat[best]), 1e-300))
        rec["exact"] = bool((f_st, sb) == (bf, bb))
        S = bg[idx].sum(axis=0)
        parent = float((S ** 2).sum()) / max(len(idx), 1)
        parent = float(np.sqrt(parent)) if cosine else parent
        rec["gain"] = float(flat[best] - (0.0 if lvl == 0 else parent))
        assert rec["exact"] or rec["gap_rel"] <= rel_tol, rec
        out.append(rec)
    return out
