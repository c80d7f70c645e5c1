"""What the near-tie replay costs at the headline size (round 6): one tree of the full-size fixtures (tests/golden/full_cfg2 / full_cfg3 inputs)
and the bench's own synthetic inputs, grown with GBRL_HIP_NEARTIE_MAX_ROWS = 65536 (default), 0 (no limit) and with the replay off; per run the
step time (device-resident inputs, third of three identical steps) and the number of replayed levels.
    python scripts/neartie_fullsize_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
import cases as K
import gbrl_amd

dev = torch.device("cuda:0")
def tup(t): return (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")

def run(case, X, G, env, steps=3):
    for k in ("GBRL_HIP_NEARTIE_MAX_ROWS", "GBRL_HIP_NO_NEARTIE_REPLAY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case, device="cuda"))
    F = case["F"]
    m.set_feature_weights(np.ones(F, np.float32))
    for o in K.optimizers(case):
        m.set_optimizer(**o)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    m.set_profiling(2)
    ts, reps = [], []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.step(tup(X), None, tup(G))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        reps.append(dict(m.last_phase_times()).get("near_replays", 0.0))
    e = m.get_ensemble_data()
    return ts, reps, (np.asarray(e["feature_indices"])[:1].tolist(), len(np.asarray(e["values"])))

NAMES = sys.argv[1].split(",") if len(sys.argv) > 1 else ("full_cfg2", "full_cfg3", "full_cfg2_s1", "bench_cfg2", "bench_cfg3")
LABELS = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name in NAMES:
    if name.startswith("full"):
        case = K.FULLSIZE_BY_NAME[name]
        Xh, _, Gh, _ = K.make_inputs(case)
        X, G = torch.from_numpy(Xh).to(dev), torch.from_numpy(Gh).to(dev)
    else:
        case = dict(K.FULLSIZE_BY_NAME["full_cfg2" if name.endswith("2") else "full_cfg3"], name=name)
        gen = torch.Generator(device=dev); gen.manual_seed(1234)
        wgen = torch.Generator(device=dev); wgen.manual_seed(99)
        W = torch.randn((8, 8), device=dev, dtype=torch.float32, generator=wgen)
        X = torch.randn((1 << 20, 128), device=dev, dtype=torch.float32, generator=gen)
        G = (torch.tanh(X[:, :8] @ W) + 0.5 * torch.randn((1 << 20, 8), device=dev, dtype=torch.float32, generator=gen)).contiguous()
    for label, env in (("off (GBRL_HIP_NO_NEARTIE_REPLAY=1)", {"GBRL_HIP_NO_NEARTIE_REPLAY": "1"}), ("default (no replay above 65536 rows)", {}), ("limit65536 (nodes <= 65536 rows)", {"GBRL_HIP_NEARTIE_MAX_ROWS": "65536"}), ("limit262144", {"GBRL_HIP_NEARTIE_MAX_ROWS": "262144"}), ("nolimit (GBRL_HIP_NEARTIE_MAX_ROWS=0)", {"GBRL_HIP_NEARTIE_MAX_ROWS": "0"})):
        if LABELS is not None and label.split()[0] not in LABELS:
            continue
        ts, reps, tree = run(case, X, G, env)
        print("%-14s %-40s ms per step %s   cumulative replayed levels %s   first tree %s" % (name, label, ["%.2f" % t for t in ts], reps, tree), flush=True)
    del X, G
    torch.cuda.empty_cache()
