"""Where do the microseconds of a predict() call go (T trees, 2^20 x 128 and 1024 x 128 rows, D 8, depth 6, device inputs)?
wall time per call with profiling off / on, and the kernel's own time.
    python scripts/predict_overhead_probe.py [trees]         (GBRL_HIP_PREDICT_SYNC=1: hipStreamSynchronize instead of the published flag)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gbrl_amd
import bench as B

dev = torch.device("cuda:0")
N, F, D, depth = 1 << 20, 128, 8, 6
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g)
G = torch.randn((N, D), device=dev, generator=g)
m = B.make_model(gbrl_amd, np, "cfg2", F, 0, D, depth, 256, "probe")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 28
for _ in range(trees):
    m.step(tup(X), None, tup(G))
torch.cuda.synchronize()
print("GBRL_HIP_PREDICT_SYNC =", os.environ.get("GBRL_HIP_PREDICT_SYNC", "0"))
for rows in (N, 1024):
    xr = tup(X[:rows])
    for prof in (0, 1):
        m.set_profiling(prof)
        for reps in (5, 50):
            m.predict(xr, None, 0, 0); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                p = m.predict(xr, None, 0, 0); del p
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            print("  rows %7d profiling %d reps %2d: %.2f us per call, kernel %.2f us" % (rows, prof, reps, dt * 1e6, m.last_phase_times().get("predict", 0.0) * 1e3))
