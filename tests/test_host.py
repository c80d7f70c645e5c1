"""CPU-side checks of the product: the C ABI exports what include/gbrl_hip.h declares, the .gbrl_model format is byte
compatible with files written by the reference, argument checking mirrors the reference binding, and the product fails
loudly (never falls back) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import cases as K
import gbrl_amd
from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gbrl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(gbrl_hip_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n not in ("gbrl_hip_model",)))


def test_c_abi_exports_every_declared_symbol():
    lib = ctypes.CDLL(gbrl_amd.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    lib.gbrl_hip_abi_version.restype = ctypes.c_int
    assert lib.gbrl_hip_abi_version() == 1
    assert lib.gbrl_hip_device_count() >= 0


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/gbrl_hip.h is the drop-in boundary: it must be valid C99 (no C++, no torch types) and a C program must link
    against libgbrl_hip.so and call an entry point without a GPU."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "gbrl_hip.h"\n#include <stdio.h>\nint main(void) { printf("%d %d\\n", gbrl_hip_abi_version(), '
                   'gbrl_hip_rccl_available() >= 0); return gbrl_hip_abi_version() > 0 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    libdir = os.path.dirname(gbrl_amd.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                    "-lgbrl_hip", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True)
    assert int(out.stdout.split()[0]) >= 1


def test_metadata_struct_is_80_bytes_like_the_reference():
    class Meta(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("n_leaves", "n_trees", "max_trees", "max_leaves", "max_trees_batch",
                                                  "max_leaves_batch", "input_dim", "output_dim", "policy_dim", "max_depth",
                                                  "min_data_in_leaf", "n_bins", "par_th")] + \
                   [("cv_beta", ctypes.c_float), ("verbose", ctypes.c_int32), ("batch_size", ctypes.c_int32),
                    ("use_cv", ctypes.c_uint8), ("split_score_func", ctypes.c_uint8), ("generator_type", ctypes.c_uint8),
                    ("grow_policy", ctypes.c_uint8), ("n_num_features", ctypes.c_int32), ("n_cat_features", ctypes.c_int32),
                    ("iteration", ctypes.c_int32)]
    assert ctypes.sizeof(Meta) == 80 and Meta.iteration.offset == 76 and Meta.use_cv.offset == 64


def _mask_header_padding(b):
    b = bytearray(b)
    b[6:8] = b"\0\0"      # serializationHeader padding after the three u16 (uninitialised in the reference)
    b[20:24] = b"\0\0\0\0"  # ... and after reserved2
    return bytes(b)


@pytest.mark.parametrize("name", K.MODEL_FILE_CASES)
def test_model_file_roundtrip_is_byte_compatible(name, tmp_path):
    case, g, _ = load_golden(name)
    ref_bytes = g["model_file"].tobytes()
    p = tmp_path / "ref.gbrl_model"
    p.write_bytes(ref_bytes)
    m = gbrl_amd.GBRL.load(str(p))            # a file written by the reference loads ...
    e = m.get_ensemble_data()
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(np.asarray(e[k]), g[k]), k
    assert m.get_num_trees() == int(g["n_trees"]) and m.get_iteration() == int(g["iteration"])
    assert m.get_learner_name() == case["name"]
    assert len(m.get_optimizers()) == len(K.optimizers(case))
    q = tmp_path / "ours.gbrl_model"
    assert m.save(str(q)) == 0                # ... and what we write back is the same file
    assert _mask_header_padding(q.read_bytes()) == _mask_header_padding(ref_bytes)
    m2 = gbrl_amd.GBRL(m)                     # copy constructor keeps everything
    assert np.array_equal(np.asarray(m2.get_ensemble_data()["values"]), g["values"])


def test_model_file_of_a_fresh_model_is_the_file_the_reference_writes(tmp_path):
    """A model constructed HERE (not loaded) and saved before any tree: byte for byte the file the reference writes for the same
    constructor and optimizer calls -- header, the 80-byte metadata, the parallel_predict flag (true unless Adam, gbrl.h:511: the
    reference's small-batch tree-parallel predict depends on it after a load), use_cv, learner name, optimizer records."""
    import oracle
    kw = dict(input_dim=4, output_dim=2, policy_dim=2, max_depth=3, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9, split_score_func="L2",
              generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy="oblivious", verbose=0, device="cpu",
              learner_name="fresh")
    m = gbrl_amd.GBRL(**kw)
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=1)
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=1, stop_idx=2)
    p = tmp_path / "fresh.gbrl_model"
    assert m.save(str(p)) == 0
    ours = p.read_bytes()
    assert ours[24 + 80] == 1 and ours[24 + 81] == 0          # parallel_predict, use_cv
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    r = ref.GBRL(**kw)
    r.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=1)
    r.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=1, stop_idx=2)
    q = tmp_path / "fresh_ref.gbrl_model"
    assert r.save(str(q)) == 0
    assert _mask_header_padding(ours) == _mask_header_padding(q.read_bytes())


def test_model_file_written_here_loads_in_the_reference(tmp_path):
    import oracle
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    case, g, _ = load_golden("grd_cos_q_ac")
    p = tmp_path / "a.gbrl_model"
    p.write_bytes(g["model_file"].tobytes())
    q = tmp_path / "b.gbrl_model"
    gbrl_amd.GBRL.load(str(p)).save(str(q))
    r = ref.GBRL.load(str(q))
    er = r.get_ensemble_data()
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(np.asarray(er[k]), g[k]), k


def _model(**kw):
    base = dict(input_dim=4, output_dim=2, policy_dim=2, max_depth=3, split_score_func="L2", generator_type="Quantile",
                grow_policy="oblivious", device="cpu")
    base.update(kw)
    return gbrl_amd.GBRL(**base)


def test_constructor_and_setters_mirror_reference_errors():
    m = _model()
    md = m.get_metadata()
    assert md["grow_policy"] == "Oblivious" and md["split_score_func"] == "L2" and md["n_bins"] == 256
    with pytest.raises(RuntimeError):
        _model(split_score_func="l3")
    with pytest.raises(RuntimeError):
        _model(use_control_variates=True)
    with pytest.raises(RuntimeError):
        m.set_bias(np.zeros(3, np.float32))
    m.set_bias(np.array([1.0, -2.0], np.float32))
    assert np.array_equal(m.get_bias(), [1.0, -2.0])
    m.set_feature_weights(np.arange(4, dtype=np.float32))
    assert np.array_equal(m.get_feature_weights(), np.arange(4))
    m.set_feature_mapping(np.array([0, 1, 2, 3], np.int32), np.array([True, False, True, False]))
    e = m.get_ensemble_data()
    assert e["reverse_num_feature_mapping"].tolist() == [0, 2, -1, -1]
    assert e["reverse_cat_feature_mapping"].tolist() == [1, 3, -1, -1]
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=1)
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=1, stop_idx=2)
    with pytest.raises(RuntimeError):          # limit = output_dim (gbrl.cpp:457)
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=2)
    with pytest.raises(RuntimeError):
        _model().set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=1, stop_idx=1)
    with pytest.raises(RuntimeError):          # Adam / Linear are CPU-only in the reference (gbrl.cpp:477-506)
        _model().set_optimizer(algo="Adam", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=2)
    assert np.allclose(m.get_scheduler_lrs(), [0.1, 0.01])
    assert [o["stop_idx"] for o in m.get_optimizers()] == [1, 2]


def test_step_and_predict_shape_checks():
    m = _model()
    X = np.zeros((300, 4), np.float32)
    with pytest.raises(RuntimeError, match="Gradient output dim"):
        m.step(X, None, np.zeros((300, 3), np.float32))
    with pytest.raises(RuntimeError, match="Number of observations"):
        m.step(X, None, np.zeros((200, 2), np.float32))
    with pytest.raises(RuntimeError, match="Total number of features"):
        m.step(np.zeros((300, 3), np.float32), None, np.zeros((300, 2), np.float32))
    with pytest.raises(RuntimeError, match="Expected array of format"):
        m.step(X, np.zeros((300, 1)), np.zeros((300, 2), np.float32))   # categorical must be S128
    with pytest.raises(RuntimeError, match="without observations"):
        m.predict(None, None)
    with pytest.raises(RuntimeError, match="stop_tree_idx is out of bounds"):
        m.predict(X, None, 0, 5)
    with pytest.raises(RuntimeError, match="Invalid tree index"):      # inspection of an empty model (tests/test_explain.py has the rest)
        m.print_tree()


def test_no_gpu_means_loud_failure_not_fallback():
    if gbrl_amd.cuda_available():
        pytest.skip("a HIP device is present")
    m = _model()
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=2)
    with pytest.raises(RuntimeError, match="no HIP device"):
        m.step(np.zeros((300, 4), np.float32), None, np.zeros((300, 2), np.float32))
    with pytest.raises(RuntimeError, match="no HIP device"):
        m.predict(np.zeros((300, 4), np.float32), None)


def test_product_does_not_touch_the_oracle():
    """The shipped package must not import / link / load anything under oracle/."""
    for root, _, files in os.walk(os.path.join(ROOT, "gbrl_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "oracle.h" not in txt, f
    out = os.popen("ldd " + gbrl_amd.LIB_PATH).read()
    assert "oracle" not in out


def test_python_signatures_match_the_reference_class():
    """Every public method of the reference's gbrl_cpp.GBRL exists here with the same argument names, order, defaults and annotated
    types (pybind11 writes them into the docstrings); only additions are allowed."""
    import oracle
    ref_mod = oracle.load_ref()
    if ref_mod is None:
        pytest.skip("oracle/_ref not built")

    def signatures(cls):
        out = {}
        for n in dir(cls):
            if n.startswith("__") and n != "__init__":
                continue
            doc = getattr(cls, n).__doc__ or ""
            lines = [ln for ln in doc.split("\n") if re.match(r"\s*(\d+\.\s*)?%s\(" % re.escape(n), ln)]
            out[n] = [re.sub(r"gbrl_cpp(_ref)?\.GBRL", "GBRL", re.sub(r"^\s*\d+\.\s*", "", ln)).strip() for ln in lines]
        return out

    ref, mine = signatures(ref_mod.GBRL), signatures(gbrl_amd.GBRL)
    assert not set(ref) - set(mine), sorted(set(ref) - set(mine))
    for name, sig in ref.items():
        if name == "get_feature_mapping":       # same call, the return annotation here is more specific
            assert [s.split(" -> ")[0] for s in mine[name]] == [s.split(" -> ")[0] for s in sig]
            continue
        assert mine[name] == sig, (name, sig, mine[name])


def test_corrupt_model_files_are_refused_not_loaded(tmp_path):
    """Model::load must not hand the predict kernels out-of-range feature indices, leaf offsets or depths, and a
    corrupt header must not drive a huge allocation: every such file fails with the reference's load error."""
    import struct
    case, g, _ = load_golden("obl_l2_q")
    ref = bytearray(g["model_file"].tobytes())
    meta_off = 24                                 # serializationHeader
    def field(name_idx):                          # int32 fields of ensembleMetaData, types.h:218-242
        return meta_off + 4 * name_idx
    def write(path, data):
        open(path, "wb").write(bytes(data))
        return str(path)
    ok = write(tmp_path / "ok.gbrl_model", ref)
    assert gbrl_amd.GBRL.load(ok).get_num_trees() == int(g["n_trees"])
    bad = bytearray(ref); struct.pack_into("<i", bad, field(0), 1 << 30)          # n_leaves: payloads larger than the file
    with pytest.raises(RuntimeError):
        gbrl_amd.GBRL.load(write(tmp_path / "b1.gbrl_model", bad))
    bad = bytearray(ref); struct.pack_into("<i", bad, field(9), 31)               # max_depth beyond the supported range
    with pytest.raises(RuntimeError):
        gbrl_amd.GBRL.load(write(tmp_path / "b2.gbrl_model", bad))
    # a feature index beyond n_num_features: locate the feature_indices record (all values < input_dim in the good file)
    m = gbrl_amd.GBRL.load(ok)
    fi = np.asarray(m.get_ensemble_data()["feature_indices"], np.int32)
    pat = fi.tobytes()
    pos = bytes(ref).find(pat)
    assert pos > 0
    bad = bytearray(ref); struct.pack_into("<i", bad, pos, 10 ** 6)
    with pytest.raises(RuntimeError):
        gbrl_amd.GBRL.load(write(tmp_path / "b3.gbrl_model", bad))
    ti = np.asarray(m.get_ensemble_data()["tree_indices"], np.int32)
    pos = bytes(ref).find(ti.tobytes())
    assert pos > 0
    bad = bytearray(ref); struct.pack_into("<i", bad, pos + 4, -5)                # second tree starts at a negative leaf
    with pytest.raises(RuntimeError):
        gbrl_amd.GBRL.load(write(tmp_path / "b4.gbrl_model", bad))


def test_candidate_order_replay_equals_the_libstdcxx_container(tmp_path):
    """hash_order_replay.h (the order the engine gives a step's categorical candidates, Q8) against the real std::unordered_map on
    500 random insertion sequences: uniform, heavily colliding and string-key hashes (tests/cxx/replay_check.cpp)."""
    import subprocess
    exe = tmp_path / "replay_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "gbrl_amd", "csrc"), os.path.join(ROOT, "tests", "cxx", "replay_check.cpp"),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr


def test_register_tile_asm_header_is_current_and_the_compiler_keeps_out_of_the_bank(tmp_path):
    """predict_reg.hip keeps a row tile in a bank of VGPRs that the compiler must never allocate (amdgpu_num_vgpr caps it below the
    bank; the bank is only named in clobber lists).  (1) The generated assembly text is the generator's current output.  (2) Audit of
    the gfx950 ISA (hipcc cross-compiles here): outside the inline-assembly statements no instruction of the register-tile kernels
    touches a register at or above the bank's base, nothing spills (no scratch), and every kernel fits two waves per SIMD."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "scripts", "gen_predict_reg_asm.py")
    out = subprocess.run([sys.executable, gen, "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    hdr = open(os.path.join(ROOT, "gbrl_amd", "csrc", "predict_reg_asm.h")).read()
    base = {"k_predict_reg": int(re.search(r"#define PR_TILE_BASE (\d+)", hdr).group(1)), "k_predict_pc": int(re.search(r"#define PC_TILE_BASE (\d+)", hdr).group(1))}
    asm = tmp_path / "predict_reg.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "gbrl_amd", "csrc", "predict_reg.hip"), "-o", str(asm)], check=True, capture_output=True)
    cur, in_asm, top, seen = None, False, {}, {}
    for line in asm.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            top[cur] = -1
        if "ASMSTART" in line:
            in_asm = True
            continue
        if "ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy): (\d+)", line)
        if m and cur is not None:
            seen.setdefault(cur, {})[m.group(1)] = int(m.group(2))
        s = line.strip()
        if in_asm or cur is None or not s or s[0] in ";.":
            continue
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", s):
            top[cur] = max(top[cur], int(m.group(1)) if m.group(1) else int(m.group(3)))
    kernels = [k for k in top if "k_predict_reg" in k or "k_predict_pc" in k]
    assert len(kernels) == 24, kernels   # 12 register-tile + 8 packed-code instantiations, + 4 deep (8-level) ones in round 5
    for k in kernels:
        limit = base["k_predict_pc"] if "k_predict_pc" in k else base["k_predict_reg"]
        assert 0 <= top[k] < limit, (k, top[k], limit)
        assert seen[k]["ScratchSize"] == 0 and seen[k]["NumVgprs"] <= 256 and seen[k]["Occupancy"] >= 2, (k, seen[k])


def test_environment_hooks_are_listed_once_and_documented():
    """every GBRL_HIP_* hook lives in ONE table (gbrl_amd/csrc/hooks.h), the library's only getenv is hooks.cpp's, and
    INTEGRATION.md section 5 documents every entry of the table (and nothing that is not in it)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "gbrl_amd", "csrc")
    names = re.findall(r"^\s+X\(([A-Z0-9_]+)\)", open(os.path.join(csrc, "hooks.h")).read(), re.M)
    assert len(names) >= 40 and len(set(names)) == len(names)
    for path in glob.glob(os.path.join(csrc, "*")):
        if os.path.basename(path) in ("hooks.cpp", "hooks.h") or not path.endswith((".hip", ".cpp", ".h")):
            continue
        text = open(path).read()
        assert "getenv(" not in text, os.path.basename(path) + " reads the environment itself"
        for used in re.findall(r"hooks::(?:raw|on|num)\((?:gbrl::)?hooks::([A-Z0-9_]+)", text):
            assert used in names, used
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in names:
        assert "GBRL_HIP_" + n in doc, "GBRL_HIP_%s is not documented in INTEGRATION.md section 5" % n
    documented = set(re.findall(r"`GBRL_HIP_([A-Z0-9_]+)", doc))
    constants = {"RCCL_KEEP_WORLD1"}          # (C macros of include/gbrl_hip.h that INTEGRATION.md mentions, not environment hooks)
    assert documented - constants <= set(names), sorted(documented - constants - set(names))
