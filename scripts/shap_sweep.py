"""Randomised SHAP sweep on a GPU box: models grown by the product on random data are saved, loaded by the REFERENCE's own build
(oracle/_ref) and explained by both; the product's device kernel and host evaluation must agree BITWISE with each other and with the
reference to 1e-5 of the array's scale (the float32 recursion is rounding-sensitive from depth 5 on; the product makes the roundings of
the shipped reference build explicit).
    python scripts/shap_sweep.py [n_cases] [first_seed]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import gbrl_amd, oracle

ref = oracle.load_ref()
assert ref is not None
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(seed0)
ok = bad = 0
t0 = time.time()
with tempfile.TemporaryDirectory() as d:
    for i in range(n_cases):
        Fc = int(rng.choice([0, 0, 1, 3]))
        case = dict(name="shap%d" % i, seed=seed0 + i, N=int(rng.choice([400, 1500, 5000])), F=int(rng.choice([0 if Fc else 2, 2, 5, 12])), Fc=Fc,
                    D=int(rng.choice([1, 2, 5, 9])), depth=int(rng.choice([2, 3, 4, 5, 6])), n_bins=int(rng.choice([8, 32, 64])),
                    score=str(rng.choice(["L2", "Cosine"])), gen=str(rng.choice(["Quantile", "Uniform"])), policy=str(rng.choice(["greedy", "oblivious"])),
                    trees=int(rng.choice([1, 3, 6])), min_data_in_leaf=int(rng.choice([0, 10])))
        if case["policy"] == "greedy" and case["depth"] >= 6: case["depth"] = 5
        if case["F"] + case["Fc"] == 0: case["F"] = 2
        X, Xc, G, y = K.make_inputs(case)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, Xc, G, y)
        p = os.path.join(d, "m.gbrl_model")
        m.save(p)
        r = ref.GBRL.load(p)
        base, norm, off = K.poly_vectors(case["depth"])
        n = 64
        xs = None if X is None else np.ascontiguousarray(X[:n]); xcs = None if Xc is None else np.ascontiguousarray(Xc[:n])
        want = np.array(r.ensemble_shap(xs, xcs, norm, base, off))
        os.environ["GBRL_HIP_SHAP_HOST"] = "0"; dev = m.ensemble_shap(xs, xcs, norm, base, off)
        os.environ["GBRL_HIP_SHAP_HOST"] = "1"; host = m.ensemble_shap(xs, xcs, norm, base, off)
        scale = max(float(np.abs(want).max()), 1e-30)
        err = float(np.abs(dev - want).max()) / scale
        text_ok = True
        q = os.path.join(d, "r.gbrl_model")       # the file written here, re-saved by the reference, must be the same bytes
        r.save(q)
        mask = lambda b: b[:6] + b"\0\0" + b[8:20] + b"\0\0\0\0" + b[24:]     # uninitialised header padding in the reference
        text_ok = mask(open(p, "rb").read()) == mask(open(q, "rb").read())
        if case["policy"] == "oblivious":      # the exported header of the same model file must be byte-identical (the product LOADS it too)
            fmt, typ = str(rng.choice(["float", "fxp8", "fxp16"])), str(rng.choice(["full", "compact"]))
            m2 = gbrl_amd.GBRL.load(p)
            a, b = os.path.join(d, "a.h"), os.path.join(d, "b.h")
            m2.export(a, "net", fmt, typ, "P_"); r.export(b, "net", fmt, typ, "P_")
            text_ok = text_ok and open(a, "rb").read() == open(b, "rb").read()
        if np.array_equal(dev, host) and err <= 1e-5 and text_ok:
            ok += 1
        else:
            bad += 1
            print("BAD", case, "device==host:", bool(np.array_equal(dev, host)), "err vs reference %.3g" % err, "file / export identical:", text_ok, flush=True)
print("shap cases %d: ok %d, bad %d  (%.1f s)" % (n_cases, ok, bad, time.time() - t0))
sys.exit(1 if bad else 0)
