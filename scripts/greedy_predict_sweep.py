"""cfg3's predict (greedy depth-6 trees, 2^20 x 128 rows, D = 8) under the launch-plan hooks of k_predict_obl2<GREEDY> (measurement):
    python3 scripts/greedy_predict_sweep.py [trees]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gbrl_amd, bench
dev = torch.device("cuda:0")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, F, D = 1 << 20, 128, 8
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); G = torch.randn((N, D), device=dev, generator=g)
m = bench.make_model(gbrl_amd, np, "cfg3", F, 0, D, 6, 256, "sweep")
for i in range(T):
    sl = slice(i * 4096, (i + 1) * 4096)
    m.step(tup(X[sl].contiguous()), None, tup(G[sl].contiguous()))
base = None
HOOKS = ("GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_TT", "GBRL_HIP_PREDICT_NB", "GBRL_HIP_PREDICT_NO_PERSIST")
for env in ({}, {"GBRL_HIP_PREDICT_TT": "4"}, {"GBRL_HIP_PREDICT_NO_PERSIST": "1"}, {"GBRL_HIP_PREDICT_TT": "4", "GBRL_HIP_PREDICT_NO_PERSIST": "1"},
            {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_NB": "2"}, {"GBRL_HIP_PREDICT_RG": "2"}, {"GBRL_HIP_PREDICT_RG": "2", "GBRL_HIP_PREDICT_TT": "4"},
            {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_TT": "4", "GBRL_HIP_PREDICT_NB": "2"}, {}):
    for h in HOOKS: os.environ.pop(h, None)
    os.environ.update(env)
    m.set_profiling(1)
    p = torch.from_dlpack(m.predict(tup(X), None, 0, 0)).clone()
    if base is None: base = p
    same = bool(torch.equal(p, base))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        q = m.predict(tup(X), None, 0, 0); del q
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 200
    print("%-70s %.1f us per call, kernel %.1f us, bits %s" % (env or "default", dt * 1e6, m.last_phase_times().get("predict", 0.0) * 1e3, "same" if same else "DIFFERENT"), flush=True)
