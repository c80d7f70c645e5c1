"""Generate tests/golden/full_*.npz: ONE tree of the REFERENCE's own CPU path (oracle/_ref, built by oracle/Makefile from
/root/reference) at the headline size of BASELINE.json configs[1] / configs[2] -- 2^20 rows x 128 features, D = 8, depth 6.

    OMP_NUM_THREADS=8 python tests/golden/make_fullsize_golden.py full_cfg2          # ~20-40 min on 8 vCPU
    OMP_NUM_THREADS=8 python tests/golden/make_fullsize_golden.py full_cfg3          # capacity-patched build
    OMP_NUM_THREADS=4 python tests/golden/make_fullsize_golden.py full_cfg2 --tag t4 # a second thread count (stored beside it)

Runs in the authoring container only; the reference never travels.  The inputs (512 MiB) cannot be committed, so they are
synthesised by tests/golden/cases.py::make_inputs -- integer PCG64 draws of numpy.random.default_rng(seed) and exactly rounded
float32 arithmetic only -- and re-checked on the GPU box through their SHA-256.  Stored per fixture (a few KiB): the case, the
SHA-256, the thread count, the wall time and the tree the reference grew (structure keys + leaf values).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as K  # noqa: E402
import oracle  # noqa: E402


def main():
    args = sys.argv[1:]
    tag = ""
    if "--tag" in args:
        i = args.index("--tag")
        tag = "_" + args[i + 1]
        del args[i:i + 2]
    name = args[0]
    case = K.FULLSIZE_BY_NAME[name]
    threads = int(os.environ["OMP_NUM_THREADS"])
    ref = oracle.load_ref_capacity() if "ref_patch" in case else oracle.load_ref()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref ref-capacity"
    t0 = time.time()
    X, Xc, G, y = K.make_inputs(case)
    sha = K.inputs_digest(X, Xc, G, y)
    print(f"{name}: inputs {X.shape} {G.shape} sha256 {sha} ({time.time() - t0:.0f} s)", flush=True)
    m = ref.GBRL(**K.ctor_kwargs(case))
    F = case["F"]
    m.set_feature_weights(np.ones(F, np.float32))
    for o in K.optimizers(case):
        m.set_optimizer(**o)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    t0 = time.time()
    m.step(X, None, G)
    wall = time.time() - t0
    e = m.get_ensemble_data()
    out = {k: np.array(e[k]) for k in K.ENSEMBLE_KEYS if k != "categorical_values"}
    out["inputs_sha256"] = np.array(sha)
    out["case_json"] = np.array(json.dumps(case))
    out["omp_threads"] = np.int32(threads)
    out["wall_s"] = np.float32(wall)
    out["cpu"] = np.array(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"))
    path = os.path.join(HERE, name + tag + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}{tag}: one reference tree in {wall:.1f} s on {threads} threads; depths {out['depths'][:4]} "
          f"leaves {out['values'].shape[0]}; {os.path.getsize(path)} B", flush=True)
    print("feature_indices", out["feature_indices"][:2].tolist())
    print("feature_values", out["feature_values"][:2].tolist())


if __name__ == "__main__":
    main()
