"""Predict kernel sweep (diagnostic): first- vs second-generation oblivious kernel, rows-per-block / trees-per-group settings.
    python scripts/predict_sweep.py [n_trees] [F] [D] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 6
policy = sys.argv[5] if len(sys.argv) > 5 else "oblivious"
N = 1 << 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g)
G = torch.randn((N, D), device=dev, generator=g)
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                  split_score_func="L2" if policy == "oblivious" else "Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                  grow_policy=policy, verbose=0, device="cuda", learner_name="probe")
m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
t0 = time.time()
for i in range(T):
    o = (i * 4096) % (N - 4096)
    m.step(tup(X[o:o + 4096]), None, tup(G[o:o + 4096]))
torch.cuda.synchronize()
print("grew %d trees in %.1f s" % (T, time.time() - t0), flush=True)
m.set_profiling(1)

def run(stop, env):
    for k in ("GBRL_HIP_PREDICT_OBL1", "GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_TT", "GBRL_HIP_PREDICT_NB", "GBRL_HIP_PREDICT_STAGGER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = torch.from_dlpack(m.predict(tup(X), None, 0, stop)); torch.cuda.synchronize()
    ks = []
    for _ in range(3):
        q = m.predict(tup(X), None, 0, stop); torch.cuda.synchronize(); ks.append(m.last_phase_times().get("predict", 0.0)); del q
    return p, min(ks)

for stop in [15, 128, T]:
    if stop > T: continue
    ref, t1 = run(stop, {"GBRL_HIP_PREDICT_OBL1": "1"})
    print("trees %5d  gen1            kernel_ms %8.3f  row-trees/s %.3e  rows/s %.3e" % (stop, t1, N * stop / (t1 * 1e-3), N / (t1 * 1e-3)), flush=True)
    for rg in (1, 2, 3, 4):
        for nb in (2, 1):
            for tt in (4, 8, 12, 16):
                try:
                    p, t2 = run(stop, {"GBRL_HIP_PREDICT_RG": str(rg), "GBRL_HIP_PREDICT_TT": str(tt), "GBRL_HIP_PREDICT_NB": str(nb)})
                except Exception as e:
                    print("rg %d tt %d failed: %r" % (rg, tt, e)); continue
                same = bool(torch.equal(p, ref))
                print("trees %5d  gen2 rg %d nb %d tt %2d  kernel_ms %8.3f  row-trees/s %.3e  rows/s %.3e  bitwise==gen1 %s" % (stop, rg, nb, tt, t2, N * stop / (t2 * 1e-3), N / (t2 * 1e-3), same), flush=True)
    p, t2 = run(stop, {})
    print("trees %5d  gen2 default    kernel_ms %8.3f  row-trees/s %.3e  bitwise==gen1 %s" % (stop, t2, N * stop / (t2 * 1e-3), bool(torch.equal(p, ref))), flush=True)
