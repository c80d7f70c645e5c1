"""Predict kernel time vs number of trees (diagnostic):  python scripts/predict_probe.py [n_trees]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
N, F, D, depth = 1 << 20, 128, 8, 6
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
small = (X[:4096].contiguous(), G[:4096].contiguous())
for _ in range(T): m.step(tup(small[0]), None, tup(small[1]))
m.set_profiling(1)
for stop in [1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096]:
    if stop > T: break
    m.predict(tup(X), None, 0, stop); torch.cuda.synchronize()
    ks = []
    for _ in range(3):
        p = m.predict(tup(X), None, 0, stop); torch.cuda.synchronize(); ks.append(m.last_phase_times().get("predict", 0.0)); del p
    print("trees %5d  kernel_ms %8.3f  row-trees/s %.3e  rows/s %.3e" % (stop, min(ks), N * stop / (min(ks) * 1e-3), N / (min(ks) * 1e-3)), flush=True)
