"""Timing probe (GPU box): ensemble_shap of a config-2 shaped model (oblivious, F=128, D=8, depth 6) on the device and on the host.
    python scripts/shap_probe.py [rows=131072] [trees=15]
Host pointers in and out, like the reference's binding: the device figure includes the PCIe copies of obs and of the
[rows][F][D] result (4 KiB per row)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import gbrl_amd
import cases as K

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 15
F, D, depth = 128, 8, 6
rng = np.random.default_rng(0)
X = rng.standard_normal((rows, F)).astype(np.float32)
W = rng.standard_normal((8, D)).astype(np.float32)
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                  split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=5000,
                  grow_policy="oblivious", verbose=0, device="cpu")
m.set_feature_weights(np.ones(F, np.float32))
m.set_optimizer("SGD", "Const", 0.1, 0, D)
for t in range(trees):
    G = (np.tanh(X[:, :8] @ W) + 0.5 * rng.standard_normal((rows, D))).astype(np.float32)
    m.step(X, None, G)
base, norm, offset = K.poly_vectors(depth)
res = {}
for host in ("0", "1"):
    os.environ["GBRL_HIP_SHAP_HOST"] = host
    n = rows if host == "0" else min(rows, 16384)
    m.ensemble_shap(X[:256], None, norm, base, offset)
    t0 = time.time()
    res[host] = m.ensemble_shap(X[:n], None, norm, base, offset)
    dt = time.time() - t0
    print("%s: %d rows x %d trees in %.3f s = %.3g row-trees/s" % ("host (%d threads)" % os.cpu_count() if host == "1" else "device", n, trees, dt, n * trees / dt))
n = res["1"].shape[0]
print("device == host on the common rows:", bool(np.array_equal(res["0"][:n], res["1"])))
