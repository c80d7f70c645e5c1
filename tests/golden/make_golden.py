"""Generate tests/golden/*.npz from the REFERENCE's own CPU path (oracle/_ref, built by oracle/Makefile from
/root/reference).  Runs in the authoring container only; the reference never travels.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py obl_l2_q   # selected cases

Each fixture holds: the case dict (json), sha256 of the synthesised inputs, every hot-path array of
get_ensemble_data() (binding.cpp:330-390), get_metadata(), the final predict() output, the split candidates are
NOT exposed by the reference binding and therefore not stored.  For MODEL_FILE_CASES the bytes written by
GBRL.save() are stored too.  Both the -march=x86-64-v3 build (shipped) and a -march=native build are run and
must agree byte for byte (printed), which pins that the portable flags did not change the oracle's rounding.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as K  # noqa: E402
import oracle  # noqa: E402


def run(mod, case):
    X, Xc, G, y = K.make_inputs(case)
    m = mod.GBRL(**K.ctor_kwargs(case))
    fit_loss = None
    if "fit_iterations" in case:
        fit_loss, pred = K.drive_fit(m, case, X, y, Xc)
    else:
        pred = K.drive(m, case, X, Xc, G, y)
    e = m.get_ensemble_data()
    out = {k: np.array(e[k]) for k in K.ENSEMBLE_KEYS}
    out["pred"] = np.array(pred, np.float32)
    if fit_loss is not None:
        out["fit_loss"] = np.float32(fit_loss)
        out["bias"] = np.array(m.get_bias(), np.float32)
    out["n_trees"] = np.int32(m.get_num_trees())
    out["iteration"] = np.int32(m.get_iteration())
    out["inputs_sha256"] = np.array(K.inputs_digest(X, Xc, G, y))
    out["case_json"] = np.array(json.dumps(case))
    out["meta_json"] = np.array(json.dumps({k: (v if not isinstance(v, (np.generic,)) else v.item())
                                             for k, v in m.get_metadata().items()}))
    if case["name"] in K.MODEL_FILE_CASES:
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "m.gbrl_model")
            assert m.save(p) == 0
            out["model_file"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    return out


STRUCTURE_KEYS = ("tree_indices", "depths", "feature_indices", "feature_values", "is_numerics", "inequality_directions",
                  "categorical_values", "edge_weights")


def digest(a):
    import hashlib
    h = hashlib.sha256()
    # fit(): the bias is a thread-count dependent float32 mean (math_ops.cpp:255-300), so leaf values and predictions differ in
    # their last bits between thread counts by construction; what must be stable there is the tree STRUCTURE
    keys = STRUCTURE_KEYS if "fit_loss" in a else K.ENSEMBLE_KEYS + ("pred",)
    for k in keys:
        h.update(np.ascontiguousarray(a[k]).tobytes())
    return h.hexdigest()


def main():
    ref = oracle.load_ref()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"
    if len(sys.argv) == 3 and sys.argv[1] == "--digest":     # child mode: print the digest under this OMP_NUM_THREADS
        print(digest(run(ref, K.BY_NAME[sys.argv[2]])))
        return
    assert os.environ.get("OMP_NUM_THREADS") == "8", "fixtures are defined at OMP_NUM_THREADS=8"
    nat = oracle.load_ref(native=True)
    names = sys.argv[1:] or [c["name"] for c in K.CASES + K.FIT_CASES]
    for name in names:
        case = K.BY_NAME[name]
        a = run(ref, case)
        msg = ""
        if nat is not None:
            b = run(nat, case)
            same = all(np.array_equal(a[k], b[k]) for k in K.ENSEMBLE_KEYS + ("pred",))
            msg = "native==v3" if same else "NATIVE BUILD DIFFERS"
        # Is the reference's answer well defined?  Its float32 column statistics depend on the OpenMP thread count
        # (math_ops.cpp:255-300); a fixture whose trees change with it sits on a near-tie and cannot pin parity.
        import subprocess
        stable = True
        for th in ("1", "3"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--digest", name], capture_output=True, text=True,
                                 env=dict(os.environ, OMP_NUM_THREADS=th))
            stable &= out.stdout.strip().splitlines()[-1] == digest(a)
        a["ref_stable_across_threads"] = np.bool_(stable)
        if not stable and not case.get("fragile"):
            print(f"{name}: REJECTED -- the reference disagrees with itself across OMP_NUM_THREADS in (1,3,8); pick another seed")
            continue
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **a)
        sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
        print(f"{name:28s} trees={int(a['n_trees'])} leaves={a['values'].shape[0]} {sz/1024:.0f} KiB {msg} "
              f"thread-stable={stable}")


if __name__ == "__main__":
    main()
