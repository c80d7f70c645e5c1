"""Why does a fit() case differ from the reference?  Re-runs one fit_sweep case (first_seed, index) and reports (a) the first differing tree, (b) whether the REFERENCE agrees
with itself at 1 / 3 / 8 OpenMP threads (child processes), (c) whether the product's result depends on its kernel switches.
    python scripts/fit_diff_probe.py first_seed index"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
if os.environ.get("GBRL_PROBE_R01") == "1":   # the round-1 library built side by side under _r01/ (regression hunting)
    sys.path.insert(0, os.path.join(ROOT, "_r01"))
import numpy as np
import cases as K
seed0, idx = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else None
rng = np.random.default_rng(seed0)
for i in range(idx + 1):
    D = int(rng.choice([1, 1, 2, 3])); bs = int(rng.choice([1200, 2400]))
    case = dict(name="fit%d" % i, seed=seed0 + i, N=bs * int(rng.choice([1, 2, 3])), F=int(rng.choice([3, 6, 10])), Fc=int(rng.choice([0, 0, 1, 2])),
                D=D, depth=int(rng.choice([2, 3, 4])), n_bins=int(rng.choice([16, 32, 64])), score=str(rng.choice(["L2", "Cosine"])),
                gen=str(rng.choice(["Quantile", "Uniform"])), policy=str(rng.choice(["greedy", "oblivious"])), loop="rmse", y_cat_weight=1.0,
                batch_size=bs, fit_iterations=int(rng.choice([3, 5, 8])),
                opts=[dict(algo="SGD", scheduler="Const", init_lr=float(rng.choice([0.2, 0.5])), start_idx=0, stop_idx=D)])
X, Xc, G, y = K.make_inputs(case)
KEYS = ("tree_indices", "depths", "feature_indices", "feature_values", "is_numerics", "inequality_directions")
def digest(e):
    import hashlib
    h = hashlib.sha256()
    for k in KEYS: h.update(np.ascontiguousarray(np.asarray(e[k])).tobytes())
    return h.hexdigest()[:12]
if which == "ref":
    import oracle
    m = oracle.load_ref().GBRL(**K.ctor_kwargs(case)); K.drive_fit(m, case, X, y, Xc); e = m.get_ensemble_data()
    print(digest(e), np.asarray(e["feature_indices"]).reshape(-1)[:24].tolist(), np.asarray(e["feature_values"]).reshape(-1)[:8].tolist()); sys.exit(0)
if which == "prod":
    import gbrl_amd
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case)); K.drive_fit(m, case, X, y, Xc); e = m.get_ensemble_data()
    print(digest(e), np.asarray(e["feature_indices"]).reshape(-1)[:24].tolist(), np.asarray(e["feature_values"]).reshape(-1)[:8].tolist()); sys.exit(0)
print(case)
for th in ("1", "3", "8"):
    o = subprocess.run([sys.executable, __file__, str(seed0), str(idx), "ref"], env=dict(os.environ, OMP_NUM_THREADS=th), capture_output=True, text=True)
    print("reference OMP=%s:" % th, o.stdout.strip().splitlines()[-1][:200])
for env in ({}, {"GBRL_PROBE_R01": "1"}, {"GBRL_HIP_HIST_PIPE": "0"}, {"GBRL_HIP_PREDICT_OBL1": "1"}, {"GBRL_HIP_PREDICT_GENERIC": "1"}, {"GBRL_HIP_HIST_GENERIC": "1"}):
    o = subprocess.run([sys.executable, __file__, str(seed0), str(idx), "prod"], env=dict(os.environ, **env), capture_output=True, text=True)
    print("product %s:" % (env or "default"), (o.stdout.strip().splitlines() or [o.stderr[-300:]])[-1][:200])
