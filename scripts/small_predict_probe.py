import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, gbrl_amd
dev = torch.device("cuda:0")
N, F, D = 4096, 16, 1
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); y = torch.randn((N, D), device=dev, generator=g)
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=4, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9, split_score_func="L2",
                  generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy="greedy", verbose=0, device="cuda", learner_name="p")
m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
G = torch.randn((N, D), device=dev, generator=g)
for _ in range(30): m.step(tup(X), None, tup(G))
torch.cuda.synchronize()
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("predict only            %.1f us" % t(lambda: m.predict(tup(X), None, 0, 0)))
print("predict + from_dlpack   %.1f us" % t(lambda: torch.from_dlpack(m.predict(tup(X), None, 0, 0))))
print("predict + grad          %.1f us" % t(lambda: (torch.from_dlpack(m.predict(tup(X), None, 0, 0)).reshape(N, D) - y).contiguous()))
m.set_profiling(2); m.predict(tup(X), None, 0, 0); print(dict(m.last_phase_times()))
