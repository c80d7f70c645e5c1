"""Randomised split-CHOICE sweep at sizes the brute-force oracle cannot reach (GPU box): every stored (feature, threshold) of the first tree
must be the float64 arg-max of tests/fullsize.py (NumPy: sort -> thresholds -> searchsorted codes -> bincount histograms -> suffix sums
-> L2 / Cosine score), for oblivious trees level by level, for greedy trees at the root and the deepest split of three leaves.
    python scripts/fullsize_sweep.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import fullsize
import gbrl_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(seed0)
exact = close = bad = 0
worst = 0.0
t0 = time.time()
for i in range(n_cases):
    N = int(rng.integers(20000, 400000))
    F = int(rng.choice([3, 8, 16, 17, 33, 48]))
    D = int(rng.choice([1, 2, 3, 5, 8, 12]))
    B = int(rng.choice([15, 64, 100, 255, 256]))
    policy = str(rng.choice(["greedy", "oblivious"]))
    score = str(rng.choice(["L2", "Cosine"]))
    depth = int(rng.choice([3, 4, 6])) if policy == "oblivious" else int(rng.choice([3, 4, 5]))
    gen = str(rng.choice(["Quantile", "Quantile", "Uniform"]))
    case = dict(name="fs%d" % i, seed=seed0 + i, N=N, F=F, D=D, depth=depth, n_bins=B, score=score, gen=gen, policy=policy, trees=1)
    X = rng.standard_normal((N, F)).astype(np.float32)
    for f in range(F):
        kind = rng.integers(0, 8)
        if kind == 0: X[:, f] = np.round(X[:, f] * 2) / 2
        elif kind == 1: X[:, f] = np.exp(X[:, f])
    k = min(F, 4)
    sig = float(rng.choice([0.0, 0.3, 1.0]))            # 0: pure noise -- all candidates nearly tie
    G = (sig * np.tanh(X[:, :k] @ rng.standard_normal((k, D)).astype(np.float32)) + rng.standard_normal((N, D)).astype(np.float32) * 0.5).astype(np.float32)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, None, G, None)
    e = {kk: np.asarray(v) for kk, v in m.get_ensemble_data().items()}
    try:
        if policy == "oblivious":
            recs = fullsize.check_oblivious_tree(X, G, e, B, score, rel_tol=1e-4, gen=gen)
        else:
            L = len(e["values"])
            leaves = [0, 0, L // 2, L - 1]
            levels = [0] + [max(0, int(e["depths"][l]) - 1) for l in leaves[1:]]
            if int(e["depths"][0]) == 0:
                recs = []
            else:
                recs = fullsize.check_greedy_nodes(X, G, e, B, score, leaves, levels, rel_tol=1e-4, gen=gen)
        g = max([r["gap_rel"] for r in recs], default=0.0)
        worst = max(worst, g)
        if all(r["exact"] for r in recs): exact += 1
        else: close += 1
        tag = "exact" if all(r["exact"] for r in recs) else "gap %.1e" % g
    except AssertionError as ex:
        bad += 1
        tag = "BAD " + str(ex)[:300]
    print("case %d N=%d F=%d D=%d B=%d %s/%s/%s depth %d signal %.1f: %s" % (i, N, F, D, B, policy, score, gen, depth, sig, tag), flush=True)
print("fullsize sweep: %d cases, exact %d, within 1e-4 of the float64 maximum %d (worst gap %.1e), bad %d, %.0f s" % (n_cases, exact, close, worst, bad, time.time() - t0))
sys.exit(1 if bad else 0)
