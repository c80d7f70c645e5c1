"""In-tree build of the HIP library and its Python binding (no cmake: one hipcc object per source, compiled in parallel, one link, one g++ for the binding).

    python gbrl_amd/build.py [--force]   (run as a script: importing the package needs the built extension)

Outputs (git-ignored, but they travel with the gpurun snapshot):
    gbrl_amd/libgbrl_hip.so                      hipcc --offload-arch=gfx950   (kernels + engine + C ABI)
    gbrl_amd/gbrl_cpp.<abi>.so                   g++ + pybind11, links only against libgbrl_hip.so (C ABI)
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgbrl_hip.so")
EXT = os.path.join(HERE, "gbrl_cpp" + sysconfig.get_config_var("EXT_SUFFIX"))
ARCH = os.environ.get("PYTORCH_ROCM_ARCH", "gfx950").split(";")[0]

LIB_SRCS = ["kernels.hip", "small_grow.hip", "small_prep.hip", "neartie.hip", "seqsum.hip", "predict.hip", "predict_obl2.hip", "predict_grd_stream.hip", "predict_reg.hip", "predict_chain.hip", "categorical.hip", "quantile.hip", "radix_select.hip", "engine.hip", "engine_step.hip", "engine_candidates.hip", "engine_grow.hip", "engine_predict.hip", "engine_explain.hip", "shap.hip", "c_api.cpp", "model.cpp", "rccl_dyn.cpp", "explain.cpp", "hooks.cpp"]
LIB_DEPS = LIB_SRCS + ["kernels.h", "kernels_common.h", "score_common.h", "neartie_core.h", "small_prep.h", "predict_reg_asm.h", "engine.h", "model.h", "explain.h", "cat_hash.h", "rccl_dyn.h", "hooks.h", "engine_step_detail.h", os.path.join("..", "..", "include", "gbrl_hip.h")]
EXT_SRCS = ["binding.cpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        sys.stderr.write(out.stdout + out.stderr)
        raise RuntimeError("build failed: " + cmd[0])
    if out.stderr.strip():
        sys.stderr.write(out.stderr)


def _compile_objects(hipcc, force):
    """One object per translation unit under gbrl_amd/build/ (git-ignored), compiled in parallel and only when stale."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [d for d in LIB_DEPS if d not in LIB_SRCS]
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in headers)
    jobs, objs = [], []
    for src in LIB_SRCS:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-inline-asm", "-c", path, "-o", obj])
    workers = max(1, min(len(jobs), int(os.environ.get("GBRL_BUILD_JOBS", "6"))))
    if jobs:
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(_run, jobs))
    return objs, bool(jobs)


def build(force: bool = False) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if force or _stale(LIB, LIB_DEPS):
        objs, _ = _compile_objects(hipcc, force)
        _run([hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", LIB])
    if force or _stale(EXT, EXT_SRCS + [os.path.join("..", "..", "include", "gbrl_hip.h")]) or \
            os.path.getmtime(EXT) < os.path.getmtime(LIB):
        import pybind11
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
              f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
              *[os.path.join(CSRC, s) for s in EXT_SRCS], f"-L{HERE}", "-lgbrl_hip", "-Wl,-rpath,$ORIGIN", "-o", EXT])
    _write_build_info()


def _write_build_info():
    """gbrl_amd/build_info.json: which sources the shipped .so was built from (the GPU box has no .git; bench.py prints this next to
    its numbers so that a figure can be tied to a commit)."""
    import hashlib
    import json
    import time
    h = hashlib.sha256()
    for name in sorted(LIB_DEPS + EXT_SRCS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    root = os.path.dirname(HERE)
    def git(*a):
        try:
            return subprocess.run(["git", "-C", root] + list(a), capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception:
            return ""
    info = {"commit": git("rev-parse", "--short", "HEAD") or "unknown", "dirty": bool(git("status", "--porcelain", "gbrl_amd", "include")),
            "src_sha256": h.hexdigest()[:16], "built": time.strftime("%Y-%m-%d %H:%M:%S")}
    with open(os.path.join(HERE, "build_info.json"), "w") as f:
        json.dump(info, f)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built:", LIB, EXT)
