"""cfg3's predict (greedy depth-6 trees, 2^20 x 128 rows, D = 8): the streaming kernel against the block-cooperative one (measurement;
SWEEP_PLANS=1: also the launch-plan hooks of k_predict_obl2<GREEDY>):
    python3 scripts/greedy_predict_sweep.py [trees]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, gbrl_amd, bench
dev = torch.device("cuda:0")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, F, D = int(os.environ.get("SWEEP_ROWS", 1 << 20)), 128, 8
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); G = torch.randn((N, D), device=dev, generator=g)
m = bench.make_model(gbrl_amd, np, "cfg3", F, 0, D, 6, 256, "sweep")
for i in range(T):
    sl = slice(i * 4096, (i + 1) * 4096)
    m.step(tup(X[sl].contiguous()), None, tup(G[sl].contiguous()))
HOOKS = ("GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_TT", "GBRL_HIP_PREDICT_NB", "GBRL_HIP_PREDICT_NO_PERSIST", "GBRL_HIP_PREDICT_NO_GRD_STREAM", "GBRL_HIP_PREDICT_GRD_STREAM_WAVES")
ENVS = [{}, {"GBRL_HIP_PREDICT_NO_GRD_STREAM": "1"}]
if os.environ.get("SWEEP_PLANS") == "1":      # the launch-plan hooks of the cooperative kernel
    ENVS += [dict(e, GBRL_HIP_PREDICT_NO_GRD_STREAM="1") for e in ({"GBRL_HIP_PREDICT_TT": "4"}, {"GBRL_HIP_PREDICT_NO_PERSIST": "1"},
             {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_NB": "2"}, {"GBRL_HIP_PREDICT_RG": "2"}, {"GBRL_HIP_PREDICT_RG": "2", "GBRL_HIP_PREDICT_TT": "4"})]
base = None
for env in ENVS:
    for h in HOOKS: os.environ.pop(h, None)
    os.environ.update(env)
    m.set_profiling(1)
    p = torch.from_dlpack(m.predict(tup(X), None, 0, 0)).clone()
    if base is None: base = p
    same = bool(torch.equal(p, base))
    for _ in range(100):      # back to back, like bench.py's time_predict (spaced calls run ~7 % slower: clocks)
        q = m.predict(tup(X), None, 0, 0); del q
    best = m.last_phase_times().get("predict", 1e9)
    print("%7d rows %3d trees  %-60s kernel %.1f us, bits %s" % (N, T, env or "default (k_predict_grd_stream where it fits)", best * 1e3, "same" if same else "DIFFERENT"), flush=True)
