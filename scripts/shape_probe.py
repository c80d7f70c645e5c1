"""Timing probe (GPU box): step() across shapes away from the benchmark's (rows, features, outputs, depth, bins, generator, policy).
    python scripts/shape_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gbrl_amd

def run(N, F, D, depth, B, gen, policy, score="L2"):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.randn((N, F), device="cuda", generator=g)
    G = (torch.randn((N, D), device="cuda", generator=g) * 0.5 + torch.tanh(X[:, :1])).contiguous()
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=B, par_th=10, cv_beta=0.9,
                      split_score_func=score, generator_type=gen, use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer("SGD", "Const", 0.1, 0, D)
    ti = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    for _ in range(3): m.step(ti(X), None, ti(G))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): m.step(ti(X), None, ti(G))
    torch.cuda.synchronize(); dt = (time.time() - t0) * 100
    m.set_profiling(2); m.step(ti(X), None, ti(G)); ph = dict(m.last_phase_times())
    top = sorted(ph.items(), key=lambda kv: -kv[1])[:4]
    print("N=%8d F=%3d D=%2d depth=%d B=%4d %-8s %-9s %-6s step %7.3f ms   %s" % (N, F, D, depth, B, gen, policy, score, dt,
          " ".join("%s=%.2f" % kv for kv in top)), flush=True)

base = dict(N=1 << 20, F=128, D=8, depth=6, B=256, gen="Quantile", policy="oblivious")
variants = [{}, dict(depth=4), dict(depth=8), dict(N=1 << 18), dict(N=1 << 16), dict(N=1 << 22, F=32), dict(B=64), dict(B=32), dict(B=512, D=4),
            dict(gen="Uniform"), dict(policy="greedy"), dict(policy="greedy", depth=8), dict(policy="greedy", score="Cosine"), dict(D=1), dict(D=2), dict(D=16),
            dict(F=8), dict(F=200), dict(F=1000, N=1 << 17)]
for v in variants:
    run(**dict(base, **v))
