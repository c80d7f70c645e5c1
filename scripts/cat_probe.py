"""step() time with many categorical columns, device vs host candidate generation."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gbrl_amd
rng = np.random.default_rng(0)
N, F, Fc, D = 4096, 16, 50, 4
X = rng.standard_normal((N, F)).astype(np.float32)
toks = np.array([("obj%02d" % i).encode() for i in range(12)], dtype="S128")
Xc = toks[rng.integers(0, 12, size=(N, Fc))]
G = rng.standard_normal((N, D)).astype(np.float32)
for host in ("0", "1"):
    os.environ["GBRL_HIP_HOST_CATEGORICAL"] = host
    m = gbrl_amd.GBRL(input_dim=F + Fc, output_dim=D, policy_dim=D, max_depth=4, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func="Cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy="greedy", verbose=0, device="cpu", learner_name="catprobe")
    m.set_feature_weights(np.ones(F + Fc, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F + Fc, dtype=np.int32), np.array([True] * F + [False] * Fc))
    for _ in range(2): m.step(X, Xc, G)
    t0 = time.perf_counter()
    for _ in range(5): m.step(X, Xc, G)
    dt = (time.perf_counter() - t0) / 5
    t1 = time.perf_counter()
    for _ in range(5): m.predict(X, Xc, 0, 0)
    dp = (time.perf_counter() - t1) / 5
    print("host_categorical=%s  step %.2f ms   predict %.2f ms" % (host, dt * 1e3, dp * 1e3))
