"""Timing probe (GPU box): step() for output dimensions beyond the templated histogram kernels (D > 16 uses the run-time-D path).
    python scripts/wide_output_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gbrl_amd

def run(N, F, D, depth, policy):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.randn((N, F), device="cuda", generator=g)
    G = torch.randn((N, D), device="cuda", generator=g) + torch.tanh(X[:, :1])
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func="Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer("SGD", "Const", 0.1, 0, D)
    ti = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    for _ in range(3): m.step(ti(X), None, ti(G))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): m.step(ti(X), None, ti(G))
    torch.cuda.synchronize()
    print("N=%7d F=%3d D=%3d depth=%d %-9s step %.3f ms" % (N, F, D, depth, policy, (time.time() - t0) * 100))

for D in (8, 16, 17, 24, 40, 64):
    run(1 << 17, 64, D, 4, "greedy")
run(1 << 20, 128, 18, 6, "greedy")
