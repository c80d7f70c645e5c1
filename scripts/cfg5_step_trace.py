"""RL-sized steps of BASELINE configs[4]'s growth (4096-row minibatches, 192 numeric + 64 categorical columns, uniform candidates,
oblivious depth 6, D = 8) for a kernel timeline:  rocprofv3 --kernel-trace -- python3 scripts/cfg5_step_trace.py [steps]
then scripts/step_timeline.py <dir> out.txt k_cat_distinct_insert"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
N, F, Fc, D, mini = 1 << 16, 192, 64, 8, 4096
gen = torch.Generator(device=dev); gen.manual_seed(55)
X = torch.randn((N, F), device=dev, generator=gen)
tok = torch.randint(0, 32, (N, Fc), device=dev, generator=gen, dtype=torch.int64)
cells = torch.zeros((N, Fc, 128), device=dev, dtype=torch.uint8)
cells[:, :, 0] = ord("c"); cells[:, :, 1] = (ord("0") + tok // 10).to(torch.uint8); cells[:, :, 2] = (ord("0") + tok % 10).to(torch.uint8)
G = (torch.randn((N, D), device=dev, generator=gen) + ((tok[:, :D] % 8) == 3).float() * 2.0).contiguous()
m = bench.make_model(gbrl_amd, np, "cfg5", F, Fc, D, 6, 256, "cfg5_trace")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
ctup = lambda t: (t.data_ptr(), (t.shape[0], t.shape[1]), "S128", "cuda")
n_mb = N // mini
def run(k):
    for i in range(k):
        o = (i % n_mb) * mini
        m.step(tup(X[o:o + mini]), ctup(cells[o:o + mini]), tup(G[o:o + mini]))
    torch.cuda.synchronize()
run(30)
t0 = time.perf_counter(); run(steps); print("step ms", (time.perf_counter() - t0) * 1e3 / steps)
