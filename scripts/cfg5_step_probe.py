"""step() on 4096-row minibatches of BASELINE configs[4] (192 numeric + 64 categorical S128 columns, uniform candidates, oblivious depth 6):
wall time per step, per-phase device times, and the same with the categorical columns dropped.  python3 scripts/cfg5_step_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
import bench

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, F, Fc, D, mini = 1 << 16, 192, 64, 8, 4096
gen = torch.Generator(device=dev); gen.manual_seed(55)
X = torch.randn((N, F), device=dev, generator=gen)
tok = torch.randint(0, 32, (N, Fc), device=dev, generator=gen, dtype=torch.int64)
cells = torch.zeros((N, Fc, 128), device=dev, dtype=torch.uint8)
cells[:, :, 0] = ord("c"); cells[:, :, 1] = (ord("0") + tok // 10).to(torch.uint8); cells[:, :, 2] = (ord("0") + tok % 10).to(torch.uint8)
G = torch.randn((N, D), device=dev, generator=gen)
G = (G + ((tok[:, :D] % 8) == 3).to(torch.float32) * 2.0).contiguous()
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
ctup = lambda t: (t.data_ptr(), (t.shape[0], t.shape[1]), "S128", "cuda")
for fc in (Fc, 0):
    m = bench.make_model(gbrl_amd, np, "cfg5", F, fc, D, 6, 256, "probe_cfg5_%d" % fc)
    m.set_profiling(0)
    n_mb = N // mini
    def one(i):
        o = (i % n_mb) * mini
        m.step(tup(X[o:o + mini]), ctup(cells[o:o + mini]) if fc else None, tup(G[o:o + mini]))
    for i in range(20): one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): one(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    m.set_profiling(2); acc = {}
    for i in range(10):
        one(i)
        for k, v in m.last_phase_times().items(): acc[k] = acc.get(k, 0.0) + v / 10
    m.set_profiling(0)
    print("Fc=%2d: %.3f ms per step; device phases (sum %.3f ms): %s" % (fc, dt * 1e3, sum(v for k, v in acc.items() if not k.startswith("exchange")), {k: round(v, 3) for k, v in sorted(acc.items())}))
