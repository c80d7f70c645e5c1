"""BASELINE configs[0]'s loop (predict, gradient = prediction - target, step) for a kernel timeline:
    rocprofv3 --kernel-trace -d DIR -o t -- python3 scripts/cfg1_loop_trace.py [trees]     then scripts/step_timeline.py DIR out.txt k_small_prep"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, F, depth, B = 4096, 16, 4, 256
dev = torch.device("cuda:0")
rng = np.random.default_rng(21)
X = rng.standard_normal((N, F)).astype(np.float32)
x0 = np.clip(X[:, 0], -2, 2)
y = (x0 - x0 ** 3 / 6.0 + 0.1 * rng.standard_normal(N)).astype(np.float32).reshape(N, 1)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
def loop(reps):
    for rep in range(reps):
        m = gbrl_amd.GBRL(input_dim=F, output_dim=1, policy_dim=1, max_depth=depth, min_data_in_leaf=0, n_bins=B, par_th=10, cv_beta=0.9, split_score_func="L2",
                          generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy="greedy", verbose=0, device="cuda", learner_name="cfg1")
        m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=1)
        m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool)); m.set_bias(np.array([float(y.mean())], np.float32))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(trees):
            pred = torch.from_dlpack(m.predict(tup(Xd), None, 0, 0)).reshape(N, 1)
            g = (pred - yd).contiguous()
            m.step(tup(Xd), None, tup(g))
        torch.cuda.synchronize()
        print("iteration ms", (time.perf_counter() - t0) * 1e3 / trees, flush=True)
loop(6)
