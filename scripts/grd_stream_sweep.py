"""Randomised check of k_predict_grd_stream (predict_grd_stream.hip): greedy ensembles of random shape -- 32 / 64 / 96 / 128 features, 1..8 outputs,
depth 1..6, 1..24 trees, L2 / Cosine, 2000..70000 rows (the row threshold lifted: GBRL_HIP_PREDICT_GRD_STREAM_MIN_ROWS=1), random tree ranges,
NaN / inf cells -- against the block-cooperative kernel and the general kernel, bit for bit.
    python3 scripts/grd_stream_sweep.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import gbrl_amd

HOOKS = ("GBRL_HIP_PREDICT_NO_GRD_STREAM", "GBRL_HIP_PREDICT_GENERIC")


def run(n_cases, seed0):
    os.environ["GBRL_HIP_PREDICT_GRD_STREAM_MIN_ROWS"] = "1"
    t0 = time.time()
    bad = ranges = 0
    for i in range(n_cases):
        rng = np.random.default_rng(seed0 + i)
        case = dict(name="gsw%d" % i, seed=seed0 + i, N=int(rng.choice([700, 2500, 6000])), F=int(rng.choice([32, 64, 96, 128])), Fc=0, D=int(rng.integers(1, 9)),
                    depth=int(rng.integers(1, 7)), n_bins=int(rng.choice([16, 64, 256])), score=str(rng.choice(["L2", "Cosine"])), gen=str(rng.choice(["Quantile", "Uniform"])),
                    policy="greedy", trees=int(rng.integers(1, 25)), min_data_in_leaf=int(rng.choice([0, 0, 20])))
        X, Xc, G, y = K.make_inputs(case)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, Xc, G, y)
        T = m.get_num_trees()
        n = int(rng.integers(2000, 70000))
        Xp = rng.standard_normal((n, case["F"])).astype(np.float32)
        for v in (np.nan, np.inf, -np.inf):
            Xp[rng.integers(0, n, 30), rng.integers(0, case["F"], 30)] = v
        for rep in range(3):
            a, b = (0, 0) if rep == 0 else sorted(int(v) for v in rng.integers(0, T + 1, 2))
            if rep and a == b: continue
            outs = []
            for env in ({}, {"GBRL_HIP_PREDICT_NO_GRD_STREAM": "1"}, {"GBRL_HIP_PREDICT_GENERIC": "1"}):
                for h in HOOKS: os.environ.pop(h, None)
                os.environ.update(env)
                outs.append(np.asarray(m.predict(Xp, None, a, b)).tobytes())
            for h in HOOKS: os.environ.pop(h, None)
            ranges += 1
            if not (outs[0] == outs[1] == outs[2]):
                bad += 1
                print("MISMATCH", case, "rows", n, "range", (a, b), "stream==coop", outs[0] == outs[1], "coop==generic", outs[1] == outs[2], flush=True)
    print("grd_stream sweep: %d ensembles from seed %d, %d (batch, tree range) checks, %d mismatches  (%.0f s)" % (n_cases, seed0, ranges, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 61000) else 0)
