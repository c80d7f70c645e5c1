"""Generate tests/golden/explain_*.npz: SHAP values, exported C headers and print_tree / print_ensemble_metadata text
of the REFERENCE's own CPU build (oracle/_ref) for models it grew itself.  Authoring container only.

    OMP_NUM_THREADS=8 python tests/golden/make_explain_golden.py [case ...]

Each fixture is self-contained: the reference's saved .gbrl_model bytes (the product LOADS it, so these checks need no
GPU), the SHAP inputs (first EXPLAIN_ROWS rows of the case's inputs + the polynomial vectors) and the expected outputs.
The poly vectors are checked against the reference's own gbrl.common.utils.get_poly_vectors.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as K  # noqa: E402
import oracle  # noqa: E402

CHILD = r"""
import sys
sys.path.insert(0, %r)
import oracle
ref = oracle.load_ref()
m = ref.GBRL.load(sys.argv[1])      # prints the reference's load banner (gbrl.cpp:1183-1247) ...
sys.stdout.flush()
import os
os.write(1, b"\n@@@ end of load banner @@@\n")   # ... which is cut off below: only the method's own text is pinned
what = sys.argv[2]
if what == "meta": m.print_ensemble_metadata()
else: m.print_tree(int(what))
"""


NATIVE_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, oracle
nat = oracle.load_ref(native=True)
assert nat is not None
m = nat.GBRL.load(sys.argv[1])
a = np.load(sys.argv[2])
g = lambda k: a[k] if k in a.files else None
np.save(sys.argv[3], np.array(m.ensemble_shap(g("xs"), g("xcs"), a["norm"], a["base"], a["offset"])))
"""


def stdout_of(model_path, what):
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT, model_path, str(what)], capture_output=True, check=True)
    return out.stdout.split(b"\n@@@ end of load banner @@@\n", 1)[1]


def main():
    ref = oracle.load_ref()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"
    assert os.environ.get("OMP_NUM_THREADS") == "8", "fixtures are defined at OMP_NUM_THREADS=8"
    # pin the restated poly vectors to the reference's Python layer.  `import gbrl` needs the package's compiled module, so
    # the two pure-Python files are loaded by path with a stub parent package.
    import importlib.util
    import types
    for modname in ("gbrl", "gbrl.common"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    for modname in ("config", "utils"):
        spec = importlib.util.spec_from_file_location("gbrl.common." + modname, "/root/reference/gbrl/common/%s.py" % modname)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["gbrl.common." + modname] = mod
        spec.loader.exec_module(mod)
    get_poly_vectors = sys.modules["gbrl.common.utils"].get_poly_vectors
    for name in (sys.argv[1:] or list(K.EXPLAIN_CASES)):
        case = K.BY_NAME[name]
        X, Xc, G, y = K.make_inputs(case)
        m = ref.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, Xc, G, y)
        base, norm, offset = K.poly_vectors(case["depth"])
        rb, rn, ro = get_poly_vectors(case["depth"], np.float32)
        assert np.array_equal(rb, base) and np.array_equal(rn, norm) and np.array_equal(ro, offset), "poly vectors differ"
        n = K.EXPLAIN_ROWS
        xs = None if X is None else np.ascontiguousarray(X[:n])
        xcs = None if Xc is None else np.ascontiguousarray(Xc[:n])
        T = m.get_num_trees()
        out = dict(case_json=np.array(json.dumps(case)), inputs_sha256=np.array(K.inputs_digest(X, Xc, G, y)),
                   base_poly=base, norm_values=norm, offset=offset, n_trees=np.int32(T),
                   )
        for t in sorted({0, T // 2, T - 1}):
            out["shap_tree_%d" % t] = np.array(m.tree_shap(t, xs, xcs, norm, base, offset))
        out["shap_ensemble"] = np.array(m.ensemble_shap(xs, xcs, norm, base, offset))
        out["shap_one_row"] = np.array(m.tree_shap(0, None if xs is None else xs[0], None if xcs is None else xcs[0], norm, base, offset))
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "m.gbrl_model")
            assert m.save(p) == 0
            out["model_file"] = np.frombuffer(open(p, "rb").read(), np.uint8)
            # Exports come from the model LOADED back by the reference: the product loads the same file, so both sides hold the same
            # capacity fields (the comment block prints max_trees / max_leaves / alloc_data_size; on a live reference model these
            # also depend on whether get_ensemble_data() was called before, which shrinks them as a side effect, types.cpp:381-382).
            lm = ref.GBRL.load(p)
            for k, (mname, fmt, typ, prefix) in enumerate(K.EXPLAIN_CASES[name]):
                h = os.path.join(d, "m%d.h" % k)
                assert lm.export(h, mname, fmt, typ, prefix) == 0
                out["export_%d" % k] = np.frombuffer(open(h, "rb").read(), np.uint8)
            # for the record: how far the reference's -march=native build is from the shipped x86-64-v3 build on the same model
            # (separate process: both builds register the same pybind11 module name)
            np.savez(os.path.join(d, "in.npz"), **{k: v for k, v in dict(xs=xs, xcs=xcs, norm=norm, base=base, offset=offset).items() if v is not None})
            r = subprocess.run([sys.executable, "-c", NATIVE_CHILD % ROOT, p, os.path.join(d, "in.npz"), os.path.join(d, "nat.npy")],
                               capture_output=True)
            if r.returncode == 0:
                other = np.load(os.path.join(d, "nat.npy"))
                out["shap_native_build_rel_diff"] = np.float64(np.abs(other - out["shap_ensemble"]).max() / np.abs(out["shap_ensemble"]).max())
            out["alloc_data_size"] = np.int64(lm.get_ensemble_data()["alloc_data_size"])
            out["print_meta"] = np.frombuffer(stdout_of(p, "meta"), np.uint8)
            out["print_tree_0"] = np.frombuffer(stdout_of(p, 0), np.uint8)
            out["print_tree_last"] = np.frombuffer(stdout_of(p, -1), np.uint8)
        path = os.path.join(HERE, "explain_" + name + ".npz")
        np.savez_compressed(path, **out)
        print("%-22s trees=%d  shap|max|=%.4g  exports=%d  %.0f KiB  native-build SHAP differs by %.2g of scale" % (
            name, T, np.abs(out["shap_ensemble"]).max(), len(K.EXPLAIN_CASES[name]), os.path.getsize(path) / 1024,
            float(out.get("shap_native_build_rel_diff", np.nan))))


if __name__ == "__main__":
    main()
