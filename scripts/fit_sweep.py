"""Randomised fit() sweep on a GPU box: GBRL.fit of the product against the REFERENCE's own CPU fit (oracle/_ref) on random supervised
problems.  Tree structure must be bit-identical; a mismatch is reported with the first differing tree (the reference's bias is a
thread-count dependent float32 mean, so values are compared at 1e-4).  A structure mismatch is classified like parity_sweep's (round 6,
ADVICE r05): the first differing tree t was fitted on batch t % n_batches with gradients proportional to predict(trees < t) - y (the common
prefix of both ensembles), both candidates are re-scored in float64 on the node's rows (tests/neartie.py) and the case counts as an explained
near-tie when their gap is inside the reference's float32 summation noise.  fit() takes its thresholds from the WHOLE data set once, so the
level loop's near-tie replay applies to its trees like to any step of that batch size.
    python scripts/fit_sweep.py [n_cases] [first_seed]"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")   # the reference's fit() bias is a thread-count dependent float32 mean: pin it (256 threads on the GPU box disagree with 1 / 3 / 8)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import neartie
import gbrl_amd, oracle

ref = oracle.load_ref()
assert ref is not None, "oracle/_ref is needed"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
rng = np.random.default_rng(seed0)
exact = diff = explained = 0
t0 = time.time()
for i in range(n_cases):
    D = int(rng.choice([1, 1, 2, 3]))
    bs = int(rng.choice([1200, 2400]))
    case = dict(name="fit%d" % i, seed=seed0 + i, N=bs * int(rng.choice([1, 2, 3])), F=int(rng.choice([3, 6, 10])), Fc=int(rng.choice([0, 0, 1, 2])),
                D=D, depth=int(rng.choice([2, 3, 4])), n_bins=int(rng.choice([16, 32, 64])), score=str(rng.choice(["L2", "Cosine"])),
                gen=str(rng.choice(["Quantile", "Uniform"])), policy=str(rng.choice(["greedy", "oblivious"])), loop="rmse", y_cat_weight=1.0,
                batch_size=bs, fit_iterations=int(rng.choice([3, 5, 8])),
                opts=[dict(algo="SGD", scheduler="Const", init_lr=float(rng.choice([0.2, 0.5])), start_idx=0, stop_idx=D)])
    X, Xc, G, y = K.make_inputs(case)
    out = []
    for mod in (gbrl_amd.GBRL, ref.GBRL):
        m = mod(**K.ctor_kwargs(case))
        loss, pred = K.drive_fit(m, case, X, y, Xc)
        out.append((m.get_ensemble_data(), loss, pred))
    (e, l1, p1), (r, l2, p2) = out
    same = all(np.array_equal(np.asarray(e[k]), np.asarray(r[k])) for k in ("tree_indices", "depths", "feature_indices", "is_numerics", "inequality_directions", "categorical_values"))
    same = same and np.array_equal(np.asarray(e["feature_values"]).view(np.uint32) | 0, np.asarray(r["feature_values"]).view(np.uint32) | 0)
    close = same and np.allclose(np.asarray(e["values"]), np.asarray(r["values"]), rtol=1e-4, atol=1e-5) and abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2))
    if close:
        exact += 1
    else:
        diff += 1
        nt = min(len(np.asarray(e["depths"])), len(np.asarray(r["depths"])))
        bad_keys = [k for k in ("tree_indices", "depths", "feature_indices", "feature_values", "is_numerics", "inequality_directions", "categorical_values")
                    if not np.array_equal(np.asarray(e[k]), np.asarray(r[k]))]
        vdiff = float(np.max(np.abs(np.asarray(e["values"]) - np.asarray(r["values"])))) if np.asarray(e["values"]).shape == np.asarray(r["values"]).shape else -1.0
        why = None
        if not same:
            try:
                en = {k: np.asarray(v) for k, v in e.items()}; rn = {k: np.asarray(v) for k, v in r.items()}
                mm = neartie.first_mismatch(rn, en, case["policy"])
                t = mm[0]
                n_b = (case["N"] + bs - 1) // bs
                b = t % n_b
                rows = slice(b * bs, min(case["N"], (b + 1) * bs))
                Xb = X[rows]; Xcb = None if Xc is None else Xc[rows]
                mp = gbrl_amd.GBRL(**K.ctor_kwargs(case))
                K.drive_fit(mp, case, X, y, Xc)
                pt = np.asarray(mp.predict(Xb, Xcb, 0, t)).reshape(rows.stop - rows.start, -1) if t > 0 else np.tile(np.asarray(mp.get_bias(), np.float32), (rows.stop - rows.start, 1))
                G_t = pt - np.asarray(y[rows], np.float32).reshape(pt.shape)
                # the ensembles restricted to tree t, renumbered, on the batch's rows
                def tree_only(a):
                    ti = a["tree_indices"]; lo = int(ti[t]); hi = int(ti[t + 1]) if t + 1 < len(ti) else len(a["values"])
                    srow = slice(t, t + 1) if case["policy"] == "oblivious" else slice(lo, hi)
                    out = {k: a[k][srow] for k in ("depths", "feature_indices", "feature_values", "is_numerics", "categorical_values")}
                    out["inequality_directions"] = a["inequality_directions"][lo:hi]; out["values"] = a["values"][lo:hi]; out["tree_indices"] = np.array([0], np.int32)
                    return out
                why = neartie.explain_first_mismatch(case, Xb, Xcb, G_t, tree_only(rn), tree_only(en))
                if why is not None:
                    why["tree"] = t; why["batch"] = b
            except Exception as ex:
                why = {"explained": False, "why": "analysis failed: %r" % (ex,)}
        if why and why.get("explained"):
            explained += 1
        print("DIFF", case, "structure equal:", same, "differing keys:", bad_keys, "max |value diff|", vdiff, "losses", l1, l2, "near-tie analysis:", why, flush=True)
print("fit cases %d: structure + values agree %d, differ %d of which explained near-ties %d, unexplained %d  (%.1f s)" % (n_cases, exact, diff, explained, diff - explained, time.time() - t0))
