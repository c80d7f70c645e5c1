"""BASELINE configs[2] (greedy / Cosine, two SGD optimisers: policy [0,7) lr 0.1 + value [7,8) lr 0.01) step and predict timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
N, F, D, depth = 1 << 20, 128, 8, 6
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
X = torch.randn((N, F), device=dev, generator=g)
W = torch.randn((8, D), device=dev, generator=g)
G = (torch.tanh(X[:, :8] @ W) + 0.5 * torch.randn((N, D), device=dev, generator=g)).contiguous()
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D - 1, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                  split_score_func="Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                  grow_policy="greedy", verbose=0, device="cuda", learner_name="cfg3")
m.set_feature_weights(np.ones(F, np.float32))
m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D - 1)
m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 1, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
for _ in range(3): m.step(tup(X), None, tup(G))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): m.step(tup(X), None, tup(G))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
m.set_profiling(2); m.step(tup(X), None, tup(G)); ph = m.last_phase_times()
e = m.get_ensemble_data()
print("greedy/Cosine step: %.3f ms (%.1f trees/s), leaves of last tree: %d" % (dt * 1e3, 1 / dt, len(e["values"]) - int(e["tree_indices"][-1])))
print({k: round(v, 3) for k, v in ph.items()})
m.set_profiling(1)
p = m.predict(tup(X), None, 0, 0); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3): p = m.predict(tup(X), None, 0, 0); del p
torch.cuda.synchronize(); dp = (time.perf_counter() - t1) / 3
T = m.get_num_trees()
print("greedy predict: %d trees, %.3f ms per call, %.3e rows/s, %.3e row-trees/s, kernel %.3f ms" % (T, dp * 1e3, N / dp, N * T / dp, m.last_phase_times().get("predict", 0)))
