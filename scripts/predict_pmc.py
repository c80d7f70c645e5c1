"""One large-ensemble predict for rocprofv3 --pmc passes:  python3 scripts/predict_pmc.py [n_trees] [F] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N, D, depth = 1 << 20, 8, 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g)
G = torch.randn((N, D), device=dev, generator=g)
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                  split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                  grow_policy="oblivious", verbose=0, device="cuda", learner_name="probe")
m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
for i in range(T):
    o = (i * 4096) % (N - 4096)
    m.step(tup(X[o:o + 4096]), None, tup(G[o:o + 4096]))
for _ in range(calls):
    p = m.predict(tup(X), None, 0, 0); torch.cuda.synchronize(); del p
