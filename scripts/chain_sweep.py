"""Randomised sweep of predict()'s latency paths on a GPU box (not part of the test suite; prints a summary).
    python scripts/chain_sweep.py [n_cases] [first_seed]
Every case: a random ensemble (policy, depth, numeric / categorical columns, outputs, one or two optimisers with outputs left
uncovered, 130..700 trees grown on random minibatches, some of them shallow), predicted for random batch sizes and tree ranges:
(1) kern::predict_chain forced (GBRL_HIP_PREDICT_CHAIN=1) must equal the one-thread-per-row general kernel bit for bit;
(2) the dispatcher's own choice must equal it bit for bit wherever it promises the chain (<= 1024 rows from 128 trees; the tree-range
    split takes larger batches up to 2048 trees) and stay within 1e-5 of it everywhere."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import gbrl_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = checks = exact_default = 0
t0 = time.time()
for ci in range(n_cases):
    rng = np.random.default_rng(seed0 + ci)
    policy = str(rng.choice(["oblivious", "greedy"]))
    Fc = int(rng.choice([0, 0, 1, 3]))
    F = int(rng.choice([1, 3, 8, 17, 64, 65]))
    D = int(rng.choice([1, 2, 3, 8, 11, 16]))
    depth = int(rng.choice([1, 2, 3, 4, 5, 6] + ([7, 8] if policy == "oblivious" else [])))
    T = int(rng.integers(130, 700))
    N = 3000
    X = rng.standard_normal((N, F), dtype=np.float32)
    Xc = K.TOKENS[rng.integers(0, 5, size=(N, Fc))] if Fc else None
    m = gbrl_amd.GBRL(input_dim=F + Fc, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=int(rng.choice([16, 64, 256])), par_th=10,
                      cv_beta=0.9, split_score_func=str(rng.choice(["cosine", "L2"])), generator_type=str(rng.choice(["Quantile", "Uniform"])),
                      use_control_variates=False, batch_size=5000, grow_policy=policy, verbose=0, device="cpu", learner_name="cs")
    m.set_bias(rng.standard_normal(D).astype(np.float32)); m.set_feature_weights(np.ones(F + Fc, np.float32))
    if D >= 3 and rng.random() < 0.6:
        a = int(rng.integers(1, D - 1)); b = int(rng.integers(a, D)) if rng.random() < 0.5 else D
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.05, start_idx=0, stop_idx=a)
        if b > a: m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=a, stop_idx=b)
    else:
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.05, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F + Fc, dtype=np.int32), np.array([True] * F + [False] * Fc))
    for t in range(T):
        nb = int(rng.choice([40, 256, 700]))
        rows = rng.integers(0, N, size=nb)
        G = rng.standard_normal((nb, D), dtype=np.float32) + np.float32(0.5) * X[rows][:, :1]
        if Fc: G[:, 0] += (Xc[rows][:, 0] == K.TOKENS[1]).astype(np.float32) * 2
        if rng.random() < 0.05: G[:] = G[:1]           # constant gradients: a tree that stops early
        m.step(np.ascontiguousarray(X[rows]), None if Xc is None else np.ascontiguousarray(Xc[rows]), G)
    scale = 1.0
    for n in [int(v) for v in rng.choice([1, 3, 19, 20, 63, 64, 65, 500, 1024, 1025, 3000], size=4, replace=False)]:
        xa, xc = np.ascontiguousarray(X[:n]), (None if Xc is None else np.ascontiguousarray(Xc[:n]))
        for _ in range(3):
            a = int(rng.integers(0, T - 1)); b = int(rng.integers(a + 1, T + 1))
            if rng.random() < 0.4: a, b = 0, 0
            trees = (b if b else T) - a
            os.environ["GBRL_HIP_PREDICT_CHAIN"] = "0"; os.environ["GBRL_HIP_PREDICT_GENERIC"] = "1"
            want = np.asarray(m.predict(xa, xc, a, b)).reshape(n, -1)
            del os.environ["GBRL_HIP_PREDICT_GENERIC"]
            os.environ["GBRL_HIP_PREDICT_CHAIN"] = "1"; os.environ["GBRL_HIP_PREDICT_NOSPLIT"] = "1"
            got = np.asarray(m.predict(xa, xc, a, b)).reshape(n, -1)
            del os.environ["GBRL_HIP_PREDICT_CHAIN"]; del os.environ["GBRL_HIP_PREDICT_NOSPLIT"]
            dflt = np.asarray(m.predict(xa, xc, a, b)).reshape(n, -1)
            checks += 1
            promised = (n <= 1024 and trees >= 128) or (n <= 8192 and trees > 2048)
            ok = np.array_equal(got, want) and (np.array_equal(dflt, want) if promised else True)
            err = float(np.max(np.abs(dflt.astype(np.float64) - want) / np.maximum(np.abs(want), scale))) if n else 0.0
            exact_default += int(np.array_equal(dflt, want))
            if not ok or err > 1e-5:
                bad += 1
                print("MISMATCH seed %d %s Fc=%d F=%d D=%d depth=%d T=%d n=%d range=(%d,%d) forced_equal=%s default_equal=%s err=%.2e" %
                      (seed0 + ci, policy, Fc, F, D, depth, T, n, a, b, np.array_equal(got, want), np.array_equal(dflt, want), err), flush=True)
    print("case %d seed %d %s Fc=%d F=%d D=%d depth=%d T=%d: ok so far (%d checks, %d bad) %.0f s" % (ci, seed0 + ci, policy, Fc, F, D, depth, T, checks, bad, time.time() - t0), flush=True)
print("SUMMARY cases %d checks %d bad %d default-bitwise-chain %d" % (n_cases, checks, bad, exact_default))
