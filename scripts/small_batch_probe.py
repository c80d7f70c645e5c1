"""step()/predict() latency at RL-sized batches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
dev = torch.device("cuda:0")
for N, F, D, depth, policy in ((512, 16, 4, 4, "greedy"), (4096, 16, 4, 4, "greedy"), (4096, 128, 8, 6, "oblivious"), (16384, 64, 8, 4, "greedy")):
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((N, F), device=dev, generator=g)
    G = torch.randn((N, D), device=dev, generator=g)
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func="Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda", learner_name="small")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    for _ in range(5): m.step(tup(X), None, tup(G))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.step(tup(X), None, tup(G))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    m.set_profiling(2); m.step(tup(X), None, tup(G)); ph = m.last_phase_times(); m.set_profiling(0)
    t1 = time.perf_counter()
    for _ in range(50): p = m.predict(tup(X), None, 0, 0); del p
    torch.cuda.synchronize(); dp = (time.perf_counter() - t1) / 50
    print("N=%5d F=%3d D=%d depth=%d %-9s step %.3f ms  predict(%d trees) %.3f ms  gpu phases sum %.3f ms %s" % (
        N, F, D, depth, policy, dt * 1e3, m.get_num_trees(), dp * 1e3, sum(ph.values()), {k: round(v, 3) for k, v in ph.items()}))
