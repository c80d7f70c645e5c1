"""RL-sized predict(): wall time of one call for small batches against growing ensembles, numpy in / numpy out (the rollout path of the
reference's learners) and device in / device out; the reference's own CPU build (oracle/_ref) beside it when LAT_REF=1 (slow: the
reference grows the same ensemble on the host).  python scripts/predict_latency.py [depth] [policy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
policy = sys.argv[2] if len(sys.argv) > 2 else "oblivious"
F, D, NB = 64, 8, 4096
rng = np.random.default_rng(0)
Xb = rng.standard_normal((16384, F)).astype(np.float32)
W = rng.standard_normal((F, D)).astype(np.float32)


def make(cls, device):
    m = cls(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
            split_score_func="cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy=policy, verbose=0,
            device=device, learner_name="lat")
    m.set_bias(np.zeros(D, np.float32)); m.set_feature_weights(np.ones(F, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=0, stop_idx=D)
    return m


ref = None
if os.environ.get("LAT_REF"):
    import oracle as _o
    ref = _o.load_ref()
m = make(gbrl_amd.GBRL, "cuda")
r = make(ref.GBRL, "cpu") if ref is not None else None
sizes = tuple(int(v) for v in os.environ.get("LAT_TREES", "100,1000,5000,20000").split(","))
batches = tuple(int(v) for v in os.environ.get("LAT_ROWS", "1,16,64,256,1024,4096,16384").split(","))
modes = [("default", {})] + [(kv, dict([kv.split("=")])) for kv in os.environ.get("LAT_MODES", "").split(";") if kv]
grown = 0
print("depth %d %s F=%d D=%d" % (depth, policy, F, D))
print("%6s %6s | %12s %12s %12s | %12s" % ("trees", "rows", "hip numpy ms", "hip device ms", "kernel ms", "ref cpu ms"))
for T in sizes:
    while grown < T:
        G = np.tanh(Xb[:NB] @ W * (0.3 + 0.01 * (grown % 50))).astype(np.float32) + 0.1 * rng.standard_normal((NB, D)).astype(np.float32)
        m.step(Xb[:NB], None, G)
        if r is not None and T <= 5000: r.step(Xb[:NB], None, G)
        grown += 1
    for n in batches:
        x = np.ascontiguousarray(Xb[:n])
        xd = torch.from_numpy(x).cuda()
        tup = (xd.data_ptr(), tuple(xd.shape), "torch.float32", "cuda")
        for name, env in modes:
            os.environ.update(env)
            for _ in range(3): m.predict(x, None, 0, 0)
            reps = 20 if T <= 5000 else 8
            t0 = time.perf_counter()
            for _ in range(reps): m.predict(x, None, 0, 0)
            t_np = (time.perf_counter() - t0) / reps
            m.set_profiling(1)
            ks = []
            t0 = time.perf_counter()
            for _ in range(reps):
                p = m.predict(tup, None, 0, 0); ks.append(m.last_phase_times().get("predict", 0.0)); del p
            torch.cuda.synchronize()
            t_dev = (time.perf_counter() - t0) / reps
            m.set_profiling(0)
            for k in env: os.environ.pop(k, None)
            t_ref = float("nan")
            if r is not None and T <= 5000 and name == "default":
                r.predict(x, None, 0, 0)
                t0 = time.perf_counter()
                for _ in range(5): r.predict(x, None, 0, 0)
                t_ref = (time.perf_counter() - t0) / 5
            print("%6d %6d | %12.3f %12.3f %12.3f | %12.3f  %s" % (T, n, t_np * 1e3, t_dev * 1e3, min(ks), t_ref * 1e3, name), flush=True)
