"""A few RL-sized step() calls for a kernel timeline (rocprofv3 --kernel-trace):  python3 scripts/small_step_trace.py [N] [F] [D] [depth] [policy]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
F = int(sys.argv[2]) if len(sys.argv) > 2 else 24
D = int(sys.argv[3]) if len(sys.argv) > 3 else 6
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
policy = sys.argv[5] if len(sys.argv) > 5 else "greedy"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); G = torch.randn((N, D), device=dev, generator=g)
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                  split_score_func="Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                  grow_policy=policy, verbose=0, device="cuda", learner_name="small")
m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
import time
for _ in range(20): m.step(tup(X), None, tup(G))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): m.step(tup(X), None, tup(G))
torch.cuda.synchronize(); print("step ms", (time.perf_counter() - t0) * 10)
