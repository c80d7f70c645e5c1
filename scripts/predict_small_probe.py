"""predict() over the 15-tree ensemble of the bench (HBM-bound regime): kernel ms for a few plans.  python scripts/predict_small_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
N, F, D, depth, T = 1 << 20, 128, 8, 6, 15
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); G = torch.randn((N, D), device=dev, generator=g)
for policy in ("oblivious", "greedy"):
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy=policy, verbose=0, device="cuda", learner_name="p")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    for i in range(T): m.step(tup(X[i * 4096:(i + 1) * 4096]), None, tup(G[i * 4096:(i + 1) * 4096]))
    m.set_profiling(1)
    for env in ({}, {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_NB": "2", "GBRL_HIP_PREDICT_TT": "8"}, {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_NB": "1", "GBRL_HIP_PREDICT_TT": "4"},
                {"GBRL_HIP_PREDICT_RG": "2", "GBRL_HIP_PREDICT_NB": "1", "GBRL_HIP_PREDICT_TT": "8"}, {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_NB": "1", "GBRL_HIP_PREDICT_TT": "16"}):
        for k in ("GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_NB", "GBRL_HIP_PREDICT_TT"): os.environ.pop(k, None)
        os.environ.update(env)
        ks = []
        for _ in range(5):
            p = m.predict(tup(X), None, 0, 0); torch.cuda.synchronize(); ks.append(m.last_phase_times().get("predict", 0.0)); del p
        print("%-9s %-70s kernel_ms %.4f  (%.2f TB/s of the 0.57 GB)" % (policy, env or "default", min(ks), 0.5705 / min(ks)), flush=True)
