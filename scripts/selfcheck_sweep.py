"""Large-batch self-consistency sweep (GPU box): at sizes the brute-force oracle cannot reach, the default path (transpose with the fused
first radix digit, radix selection) must grow the same trees as (a) the separate first counting pass and (b) the 32-pass bisection, and
every stored threshold must be a rank-exact data value (tests/fullsize.py ranks).
    python scripts/selfcheck_sweep.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import fullsize
import gbrl_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(seed0)
bad = 0
t0 = time.time()
for i in range(n_cases):
    N = int(rng.choice([65536, 70000, 100004, 131072, 200000, 262144 + 4]))
    if os.environ.get("SELFCHECK_RANDOM_N") == "1" and rng.random() < 0.7:
        N = int(rng.integers(16400, 420000))          # any size: odd, n % 4 != 0, partial strips, ragged chunks
    F = int(rng.choice([4, 8, 12, 20, 36, 64]))
    D = int(rng.choice([1, 2, 4, 8]))
    B = int(rng.choice([16, 64, 255, 256]))
    case = dict(name="self%d" % i, seed=seed0 + i, N=N, F=F, D=D, depth=int(rng.choice([2, 4, 6])), n_bins=B, score=str(rng.choice(["L2", "Cosine"])),
                gen="Quantile", policy=str(rng.choice(["greedy", "oblivious"])), trees=2)
    if case["policy"] == "greedy" and case["depth"] >= 6: case["depth"] = 5
    X = rng.standard_normal((N, F)).astype(np.float32)
    for f in range(F):
        kind = rng.integers(0, 8)
        if kind == 0: X[:, f] = np.round(X[:, f] * 2) / 2
        elif kind == 1: X[:, f] = (rng.random(N) < 0.9) * 1.0
        elif kind == 2: X[:, f] = np.exp(2 * X[:, f])
        elif kind == 3: X[:, f] = 0.25
    G = (np.tanh(X[:, :min(F, 3)] @ rng.standard_normal((min(F, 3), D)).astype(np.float32)) + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    only = os.environ.get("SELFCHECK_ONLY")
    if only and str(i) not in only.split(","):
        continue
    outs = []
    for env in ({}, {"GBRL_HIP_TRANSPOSE_COUNT": "0"}, {"GBRL_HIP_FORCE_BISECTION": "1"}):
        for k in ("GBRL_HIP_TRANSPOSE_COUNT", "GBRL_HIP_FORCE_BISECTION"):
            os.environ[k] = env.get(k, "1" if k == "GBRL_HIP_TRANSPOSE_COUNT" else "0")
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, None, G, None)
        outs.append({k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS})
    ok = all(np.array_equal(outs[j][k], outs[2][k]) for j in (0, 1) for k in K.ENSEMBLE_KEYS)
    if not ok:
        for j, what in ((0, "fused transpose"), (1, "separate first pass")):
            for k in K.ENSEMBLE_KEYS:
                if not np.array_equal(outs[j][k], outs[2][k]):
                    a, b = outs[j][k], outs[2][k]
                    print("   ", what, "vs bisection differs in", k, a.shape, b.shape, (np.argwhere(a != b)[:5].tolist() if a.shape == b.shape else ""))
    thr = fullsize.quantile_thresholds(X, B)
    e = outs[0]
    fi, fv, dep = e["feature_indices"], e["feature_values"], e["depths"]
    rows = fi.shape[0]
    for r in range(rows):
        for d in range(int(dep[r]) if case["policy"] == "greedy" else int(dep[min(r, len(dep) - 1)])):
            bits = fv[r, d:d + 1].view(np.uint32)[0] & 0x7fffffff if fv[r, d] == 0 else fv[r, d:d + 1].view(np.uint32)[0]
            tb = thr[int(fi[r, d])].view(np.uint32)
            if not ((tb == bits).any() or (fv[r, d] == 0 and (thr[int(fi[r, d])] == 0).any())):
                print("    threshold not rank-exact: row", r, "level", d, "feature", int(fi[r, d]), "value", fv[r, d])
                ok = False
    bad += not ok
    print("case %d N=%d F=%d D=%d B=%d %s/%s depth %d: %s" % (i, N, F, D, B, case["policy"], case["score"], case["depth"], "ok" if ok else "MISMATCH"), flush=True)
print("selfcheck: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
