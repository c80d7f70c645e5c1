"""Randomised sweep of the register-tile predict kernels (predict_reg.hip) against the general kernel, bit for bit (GPU box; not part of
the test suite; prints a summary and exits non-zero on the first mismatch).
    python scripts/predict_reg_sweep.py [n_cases] [first_seed]
Every case: a random oblivious ensemble (1-70 trees grown by step() on a small batch: depth 1-8 (7 and 8: the shapes the register-tile
kernels decline, swept at large n through the older kernels' persistent / resident modes too), 1-16 outputs, 1-300 numeric features,
0-9 categorical columns, both generators, optional feature weights / bias / two optimisers), a random batch (1 .. 70 000 rows with NaN,
infinities, signed zeros and unseen categories), random tree ranges -- predicted through the default dispatch with the row threshold
lowered to 1 (fp32 register tiles where they apply, packed codes otherwise), through the forced grouped shape, and by the general kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import gbrl_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
HOOKS = ("GBRL_HIP_PREDICT_GENERIC", "GBRL_HIP_PREDICT_NO_REG", "GBRL_HIP_PREDICT_NO_PC", "GBRL_HIP_PREDICT_REG_ONLY", "GBRL_HIP_PREDICT_REG_MIN_ROWS", "GBRL_HIP_PREDICT_REG_GROUPED")
def env(d):
    for k in HOOKS: os.environ.pop(k, None)
    os.environ.update(d)
rng = np.random.default_rng(seed0)
t0 = time.time()
taken = {"reg": 0, "declined": 0}
for i in range(n_cases):
    Fc = int(rng.choice([0, 0, 0, 1, 3, 9]))
    F = int(rng.choice([1, 4, 8, 17, 40, 64, 128, 130, 200, 300])) if Fc == 0 or rng.random() < 0.8 else 0
    D = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 12, 16]))
    case = dict(name="prs%d" % i, seed=seed0 + i, N=int(rng.choice([200, 800, 2000])), F=F, Fc=Fc, D=D, depth=int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8])),
                n_bins=int(rng.choice([7, 32, 100, 256])), score=str(rng.choice(["L2", "Cosine"])), gen=str(rng.choice(["Quantile", "Uniform"])),
                policy="oblivious", trees=int(rng.choice([1, 2, 5, 16, 33, 70])), n_tokens=int(rng.choice([3, 8, 32])))
    if case["N"] < case["n_bins"] + 1: case["n_bins"] = 32
    if rng.random() < 0.3 and F > 0: case["discrete_cols"] = [0]
    if D > 1 and rng.random() < 0.5:
        case["opts"] = [dict(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D - 1), dict(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 1, stop_idx=D)]
    if rng.random() < 0.3: case["bias"] = [float(v) for v in rng.standard_normal(D).astype(np.float32)]
    env({})
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    T = m.get_num_trees()
    n = int(rng.choice([1, 63, 64, 65, 500, 4097, 20000, 70000]))
    idx = rng.integers(0, case["N"], size=n)
    Xn = None
    if X is not None:
        Xn = np.ascontiguousarray(X[idx] + rng.standard_normal((n, F)).astype(np.float32) * np.float32(0.05))
        pick = rng.integers(0, 50, size=Xn.shape)
        Xn[pick == 0] = np.nan; Xn[pick == 1] = np.inf; Xn[pick == 2] = -np.inf; Xn[pick == 3] = 0.0; Xn[pick == 4] = -0.0
    Cn = None
    if Xc is not None:
        Cn = np.ascontiguousarray(Xc[idx]); Cn[rng.integers(0, 9, size=Cn.shape) == 0] = b"never-seen"
    ranges = [(0, 0)] + [tuple(sorted(rng.integers(0, T + 1, size=2).tolist())) for _ in range(3)]
    ranges = [(a, b) for a, b in ranges if (a, b) == (0, 0) or b > a]
    env({"GBRL_HIP_PREDICT_GENERIC": "1"})
    want = [np.asarray(m.predict(Xn, Cn, a, b)) for a, b in ranges]
    fp32_rows = Fc == 0 and 0 < F <= 128 and F % 4 == 0
    covered = (case["depth"] <= 6 and (D <= 8 or fp32_rows) and (F + 1) // 2 + 0 <= 160) or (case["depth"] in (7, 8) and D <= 8 and fp32_rows)
    for mode in ({"GBRL_HIP_PREDICT_REG_MIN_ROWS": "1"}, {"GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": str(int(rng.choice([1, 3, 8])))}):
        env(dict(mode, **({"GBRL_HIP_PREDICT_REG_ONLY": "1"} if covered and Fc == 0 else {})))
        try:
            got = [np.asarray(m.predict(Xn, Cn, a, b)) for a, b in ranges]
        except RuntimeError as e:
            print("case", i, case, "n", n, "->", e); sys.exit(2)
        for r, a, b in zip(ranges, got, want):
            if a.shape != b.shape or a.tobytes() != b.tobytes():
                # NaN payloads never reach the outputs; bytes must match
                print("MISMATCH case", i, case, "n", n, "range", r, "mode", mode, "max abs", float(np.nanmax(np.abs(a - b)))); sys.exit(1)
    taken["reg" if covered else "declined"] += 1
print("predict_reg_sweep: %d cases from seed %d, %d through the register-tile kernels, %d declined shapes (older kernels), 0 mismatches, %.0f s" % (n_cases, seed0, taken["reg"], taken["declined"], time.time() - t0))
