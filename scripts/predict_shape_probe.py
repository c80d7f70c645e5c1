"""Timing probe (GPU box): predict() across shapes (rows, features, outputs, depth, trees, policy).
    python scripts/predict_shape_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gbrl_amd

def run(N, F, D, depth, trees, policy, fit_rows=8192):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.randn((N, F), device="cuda", generator=g)
    fit_rows = min(fit_rows, N)
    Xs = X[:fit_rows].contiguous()
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=64, par_th=10, cv_beta=0.9,
                      split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer("SGD", "Const", 0.1, 0, D)
    ti = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    for t in range(trees):
        G = (torch.randn((fit_rows, D), device="cuda", generator=g) * 0.5 + torch.tanh(Xs[:, t % F:t % F + 1])).contiguous()
        m.step(ti(Xs), None, ti(G))
    for _ in range(3): m.predict(ti(X), None, 0, 0)
    torch.cuda.synchronize(); t0 = time.time(); reps = 10
    for _ in range(reps): m.predict(ti(X), None, 0, 0)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    print("N=%8d F=%3d D=%2d depth=%d trees=%4d %-9s predict %8.3f ms  %.3g rows/s  %.3g row-trees/s" % (N, F, D, depth, trees, policy, dt * 1e3, N / dt, N * trees / dt), flush=True)

base = dict(N=1 << 20, F=128, D=8, depth=6, trees=100, policy="oblivious")
for v in [{}, dict(F=16), dict(F=400), dict(depth=4), dict(depth=8), dict(depth=10), dict(D=1), dict(D=3), dict(N=1 << 14), dict(N=1 << 10, trees=500),
          dict(policy="greedy"), dict(policy="greedy", depth=8), dict(policy="greedy", depth=4, D=3), dict(trees=1), dict(trees=1000, N=1 << 18)]:
    run(**dict(base, **v))
