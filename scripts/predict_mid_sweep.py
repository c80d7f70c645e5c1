"""k_predict_obl2 launch plans for mid-sized ensembles (49 .. 1024 trees, 2^20 x 128, D = 8, depth 6) through the per-call hooks
GBRL_HIP_PREDICT_RG / _TT / _NB (the persistent small-ensemble mode stops at 48 trees, so no latched hook is involved).
python3 scripts/predict_mid_sweep.py [trees ...]"""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd, bench

dev = torch.device("cuda:0")
N, F, D = 1 << 20, 128, 8
g = torch.Generator(device=dev); g.manual_seed(3)
X = torch.randn((N, F), device=dev, generator=g)
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
m = bench.make_model(gbrl_amd, np, "cfg2", F, 0, D, 6, 256, "pmid")
sizes = [int(a) for a in sys.argv[1:]] or [49, 64, 96, 128, 192, 256, 512, 1024]
HOOKS = ("GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_TT", "GBRL_HIP_PREDICT_NB")

def timed(env, reps):
    for k in HOOKS: os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(5): p = m.predict(tup(X), None, 0, 0); del p
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): p = m.predict(tup(X), None, 0, 0); del p
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

Xs = X[: 1 << 14].contiguous()
for T in sizes:
    while m.get_num_trees() < T:
        G = torch.randn((1 << 14, D), device=dev, generator=g)
        m.step(tup(Xs), None, tup(G))
    reps = max(10, min(100, int(20000 / T)))
    res = [("default", timed({}, reps))]
    for rg, nb, tt in itertools.product((1, 2, 3, 4), (1, 2), (8, 12, 16)):
        res.append(("rg%d nb%d tt%d" % (rg, nb, tt), timed({"GBRL_HIP_PREDICT_RG": str(rg), "GBRL_HIP_PREDICT_NB": str(nb), "GBRL_HIP_PREDICT_TT": str(tt)}, reps)))
    d = dict(res)
    res.sort(key=lambda r: r[1])
    print("T=%d: default %.4f |" % (T, d["default"]), "  ".join("%s %.4f" % r for r in res[:5]), flush=True)
