"""Large-N sanity: N = 2^24 rows x 24 features (and 2^22 x 128): step + predict, leaf populations must add up to N."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
dev = torch.device("cuda:0")
for N, F, D, policy in ((1 << 24, 24, 2, "oblivious"), (1 << 22, 128, 8, "greedy")):
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((N, F), device=dev, generator=g)
    G = (torch.sign(X[:, :D]) + 0.3 * torch.randn((N, D), device=dev, generator=g)).contiguous()
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=5, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda", learner_name="big")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=1.0, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    m.step(tup(X), None, tup(G)); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.step(tup(X), None, tup(G)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e = m.get_ensemble_data()
    ew = np.asarray(e["edge_weights"]).astype(np.float64); dep = np.asarray(e["depths"])
    L0 = int(e["tree_indices"][1])
    if policy == "oblivious":
        share = np.prod(ew[:L0, :int(dep[0])], axis=1)
    else:
        share = np.array([np.prod(ew[l, :int(dep[l])]) for l in range(L0)])
    p = torch.from_dlpack(m.predict(tup(X), None, 0, 0)); torch.cuda.synchronize()
    print("N=%d F=%d %s: step %.1f ms, leaves %d, sum of leaf shares %.6f, pred finite %s, mean |pred| %.4f" % (
        N, F, policy, dt * 1e3, L0, share.sum(), bool(torch.isfinite(p).all()), float(p.abs().mean())))
    del X, G, p, m
    torch.cuda.empty_cache()
