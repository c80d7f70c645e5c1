"""`encode_categorical` / `predict_encoded` (include/gbrl_hip.h, round 6): an extension over the reference, whose predict
compares the 128-byte cells of every row inside every call (predictor.cpp:231-265 oblivious, 188-229 greedy).  A serving loop encodes a batch's
cells once -- int32 dictionary ids, 0 = a category no condition of the model mentions -- and predicts from the ids.  Checked here:
  * predict_encoded gives predict's bits: oblivious and greedy ensembles, numeric + categorical and categorical-only rows, host ids (NumPy) and
    device ids (DLPack capsule of a "cuda" model, handed back as a 4-tuple), tree ranges, cells the model never saw;
  * the ids belong to a dictionary: after a step that makes the model mention a NEW category the old token is refused (a loud error, not a
    silently stale prediction); a clone and a saved / loaded copy accept them; another model does not;
  * the C ABI entry points directly (ctypes), host buffers.
"""
import ctypes
import os

import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu


def _grown(case, device="cpu"):
    import gbrl_amd
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case, device=device))
    to_np = np.asarray
    if device == "cuda":
        import torch
        to_np = lambda c: torch.from_dlpack(c).cpu().numpy()
    K.drive(m, case, X, Xc, G, y, to_numpy=to_np)
    return m, X, Xc, G


def _case(name, **kw):
    base = dict(name=name, seed=4321, N=3000, F=6, Fc=3, D=3, depth=4, n_bins=32, score="L2", gen="Quantile", policy="oblivious", trees=6, n_tokens=12)
    base.update(kw)
    return base


@pytest.mark.parametrize("kw", [dict(), dict(policy="greedy", score="Cosine"), dict(F=0, Fc=4, policy="greedy"), dict(F=40, Fc=9, D=8, depth=6, trees=12, gen="Uniform"),
                                dict(F=3, Fc=1, D=1, trees=3)])
def test_predict_from_encoded_ids_is_predict_bit_for_bit(kw):
    case = _case("enc", **kw)
    m, X, Xc, G = _grown(case)
    rng = np.random.default_rng(9)
    # a fresh batch: tokens the model may never have split on, and one it cannot know
    n = 5000
    Xn = None if case["F"] == 0 else rng.standard_normal((n, case["F"])).astype(np.float32)
    tok = np.array(["c%02d" % t for t in range(case["n_tokens"])] + ["never-seen"], dtype="S128")
    Xcn = tok[rng.integers(0, len(tok), (n, case["Fc"]))]
    want = np.asarray(m.predict(Xn, Xcn, 0, 0))
    ids, token = m.encode_categorical(Xcn)
    ids = np.asarray(ids)
    assert ids.dtype == np.int32 and ids.shape == (n, case["Fc"]) and ids.min() >= 0
    assert (ids[Xcn == b"never-seen"] == 0).all()
    got = np.asarray(m.predict_encoded(Xn, ids, token, 0, 0))
    assert got.tobytes() == want.tobytes()
    T = m.get_num_trees()
    if T >= 3:
        a, b = np.asarray(m.predict(Xn, Xcn, 1, T - 1)), np.asarray(m.predict_encoded(Xn, ids, token, 1, T - 1))
        assert a.tobytes() == b.tobytes()
    # a clone and a saved / loaded copy share the dictionary: the same ids and token serve them
    import gbrl_amd
    c = gbrl_amd.GBRL(m)
    assert np.asarray(c.predict_encoded(Xn, ids, token, 0, 0)).tobytes() == want.tobytes()
    # ... another model does not (its dictionary differs unless it mentions exactly the same categories in the same order)
    other, _, _, _ = _grown(_case("enc_other", seed=99, **kw))
    _, other_token = other.encode_categorical(Xcn[:4])
    if other_token != token:
        with pytest.raises(RuntimeError, match="another category dictionary"):
            other.predict_encoded(Xn, ids, token, 0, 0)


def test_device_ids_and_a_stale_token():
    import torch
    case = _case("enc_dev", F=8, Fc=2, D=2, depth=3, trees=2, n_tokens=20)
    m, X, Xc, G = _grown(case, device="cuda")
    ids_cap, token = m.encode_categorical(Xc)
    ids = torch.from_dlpack(ids_cap)
    assert ids.dtype == torch.int32 and ids.is_cuda and tuple(ids.shape) == Xc.shape
    tup = (ids.data_ptr(), tuple(ids.shape), "torch.int32", "cuda")
    want = torch.from_dlpack(m.predict(X, Xc, 0, 0)).cpu().numpy()
    got = torch.from_dlpack(m.predict_encoded(X, tup, token, 0, 0)).cpu().numpy()
    assert got.tobytes() == want.tobytes()
    # grow until the dictionary grows (a tree that splits on a category no earlier tree mentioned): the token must then be refused
    rng = np.random.default_rng(1)
    refused = False
    for _ in range(40):
        m.step(X, Xc, np.ascontiguousarray(rng.standard_normal(G.shape).astype(np.float32)))
        _, t2 = m.encode_categorical(Xc[:8])
        if t2 != token:
            with pytest.raises(RuntimeError, match="another category dictionary"):
                m.predict_encoded(X, tup, token, 0, 0)
            refused = True
            break
    assert refused, "the dictionary never grew in 40 steps on 20 tokens x 2 columns"
    ids2_cap, token2 = m.encode_categorical(Xc)
    ids2 = torch.from_dlpack(ids2_cap)
    got = torch.from_dlpack(m.predict_encoded(X, (ids2.data_ptr(), tuple(ids2.shape), "torch.int32", "cuda"), token2, 0, 0)).cpu().numpy()
    assert got.tobytes() == torch.from_dlpack(m.predict(X, Xc, 0, 0)).cpu().numpy().tobytes()
    with pytest.raises(RuntimeError, match="torch.int32"):
        m.predict_encoded(X, (ids2.data_ptr(), tuple(ids2.shape), "torch.float32", "cuda"), token2, 0, 0)


def test_c_abi_entry_points_with_host_buffers(tmp_path):
    """gbrl_hip_encode_categorical / gbrl_hip_predict_encoded through ctypes on a model loaded from a file the binding saved."""
    import gbrl_amd
    case = _case("enc_c", F=5, Fc=2, D=2, depth=3, trees=4)
    m, X, Xc, G = _grown(case)
    path = str(tmp_path / "enc_c")
    m.save(path)
    want = np.asarray(m.predict(X, Xc, 0, 0))
    lib = ctypes.CDLL(os.path.join(os.path.dirname(gbrl_amd.__file__), "libgbrl_hip.so"))
    lib.gbrl_hip_load.restype = ctypes.c_void_p
    lib.gbrl_hip_load.argtypes = [ctypes.c_char_p]
    lib.gbrl_hip_last_error.restype = ctypes.c_char_p
    fname = path if os.path.exists(path) else path + ".gbrl_model"
    h = lib.gbrl_hip_load(fname.encode())
    assert h, lib.gbrl_hip_last_error()
    n, Fc, F, D = X.shape[0], Xc.shape[1], X.shape[1], case["D"]
    ids = np.empty((n, Fc), np.int32)
    token = ctypes.c_uint64(0)
    vp = ctypes.c_void_p
    lib.gbrl_hip_encode_categorical.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    rc = lib.gbrl_hip_encode_categorical(h, Xc.ctypes.data, 0, n, Fc, ids.ctypes.data, 0, ctypes.byref(token))
    assert rc == 0, lib.gbrl_hip_last_error()
    out = np.empty((n, D), np.float32)
    lib.gbrl_hip_predict_encoded.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_uint64] + [ctypes.c_int] * 5 + [vp, ctypes.c_int]
    rc = lib.gbrl_hip_predict_encoded(h, X.ctypes.data, 0, ids.ctypes.data, 0, token, n, F, Fc, 0, 0, out.ctypes.data, 0)
    assert rc == 0, lib.gbrl_hip_last_error()
    assert out.tobytes() == want.tobytes()
    rc = lib.gbrl_hip_predict_encoded(h, X.ctypes.data, 0, ids.ctypes.data, 0, ctypes.c_uint64(token.value ^ 1), n, F, Fc, 0, 0, out.ctypes.data, 0)
    assert rc == -1 and b"another category dictionary" in lib.gbrl_hip_last_error()
    lib.gbrl_hip_destroy.argtypes = [vp]
    lib.gbrl_hip_destroy(h)
