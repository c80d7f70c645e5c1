"""Near-tie replay (round 5, gbrl_amd/csrc/neartie.hip).

Where the exact arg-max of a node has a runner-up within 2^-20 relative (or a greedy gain that close to zero), the product re-scores the
candidates in the window in the reference's float32 operation sequence and takes the reference's comparison of them, so that the tree has
the reference's structure even where the reference's own rounding decided.

* the replay kernel against the restated sequence (oracle/replay_sequence.cpp, pinned to the reference's own functions by
  tests/test_oracle.py) and, where oracle/_ref is present, against those functions themselves -- bit for bit;
* the near-tie specimen of the golden fixtures becomes exact, and is the explained near-tie of rounds 1-4 with the hook
  GBRL_HIP_NO_NEARTIE_REPLAY=1;
* the one-launch growth kernel (which replays a near-tie at one node of a greedy level itself and hands every other flagged tree to the level
  loop) and the level loop give the same bytes;
* random cases against the reference build / the restatement: fewer structural differences with the replay than without, none unexplained.
"""
import ctypes
import os

import numpy as np
import pytest

import cases as K
from helpers import load_golden

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_NO_NEARTIE_REPLAY", "GBRL_HIP_NO_SMALL_GROW", "GBRL_HIP_NO_SMALL_PREP", "GBRL_HIP_NEARTIE_REL")


def _lib():
    import gbrl_amd
    lib = ctypes.CDLL(os.path.join(os.path.dirname(gbrl_amd.__file__), "libgbrl_hip.so"))
    lib.gbrl_hip_replay_scores.restype = ctypes.c_int
    lib.gbrl_hip_replay_scores.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.gbrl_hip_last_error.restype = ctypes.c_char_p
    return lib


def _device_scores(lib, g, in_node, right, meanden, cosine, min_data):
    out = np.zeros(2, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    in_node = np.ascontiguousarray(in_node, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    md = None if meanden is None else np.ascontiguousarray(meanden, np.float32)
    rc = lib.gbrl_hip_replay_scores(g.ctypes.data, in_node.ctypes.data, right.ctypes.data, g.shape[0], g.shape[1],
                                    None if md is None else md.ctypes.data, 1 if cosine else 0, min_data, out.ctypes.data)
    assert rc == 0, lib.gbrl_hip_last_error()
    return out


@pytest.mark.parametrize("cosine", [True, False])
def test_replay_kernel_is_the_reference_sequence_bit_for_bit(cosine):
    import oracle
    lib = _lib()
    probe = oracle.ref_score_probe()
    rng = np.random.default_rng(11 if cosine else 12)
    shapes = [(1, 1), (2, 3), (5, 4), (64, 7), (65, 8), (300, 1), (300, 33), (1000, 9), (4096, 16), (4097, 5), (20000, 2), (65536, 3), (700, 130), (50, 520)]
    for rep in range(60):
        N, D = shapes[rep % len(shapes)]
        sc = np.exp(rng.standard_normal() * 2)
        g = (rng.standard_normal((N, D)) * sc + (0.4 * sc if rep % 3 else 0)).astype(np.float32)
        x = rng.standard_normal((N, 1)).astype(np.float32)
        in_node = (rng.integers(0, 4, N) > 0) if rep % 4 else np.ones(N, bool)
        if rep % 9 == 8:
            in_node[:] = False
            in_node[rng.integers(0, N, 2)] = True        # one- and two-row nodes
        thr = float(np.float32(rng.standard_normal() * (2.5 if rep % 5 == 4 else 0.6)))   # (now and then every row on one side)
        right = x[:, 0] > thr
        rows = np.flatnonzero(in_node).astype(np.int32)
        md = 0 if rep % 6 else int(rng.integers(1, 4))
        meanden = None
        gs = g
        if not cosine:   # L2 scores the standardised gradients: (g - mean) / (std + 1e-8f), float32 operation by operation
            mean = g.mean(axis=0).astype(np.float32)
            den = (g.std(axis=0).astype(np.float32) + np.float32(1e-8)).astype(np.float32)
            meanden = np.concatenate([mean, den])
            gs = ((g - mean[None, :]) / den[None, :]).astype(np.float32)
        got = _device_scores(lib, g, in_node, right, meanden, cosine, md)
        want = oracle.replay_scores(x, gs, rows, 0, thr, md, cosine)
        assert got[0].tobytes() == want[0].tobytes(), ("split score", rep, N, D, len(rows), got, want)
        assert got[1].tobytes() == want[1].tobytes(), ("parent score", rep, N, D, len(rows), got, want)
        if probe is not None:
            ref = oracle.ref_scores(x, gs, rows, 0, thr, md, cosine)
            assert got[0].tobytes() == ref[0].tobytes() and got[1].tobytes() == ref[1].tobytes(), ("reference functions", rep, got, ref)


def _grow(case, inputs, monkeypatch, env, counters=False):
    import gbrl_amd
    for k in HOOKS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    X, Xc, G, y = inputs
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    m.set_profiling(2)
    pred = K.drive(m, case, X, Xc, G, y)
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    if not counters:
        return e, np.asarray(pred)
    m.step(X, Xc, np.ascontiguousarray(G if G is not None else np.zeros((len(X), case["D"]), np.float32)))
    return e, np.asarray(pred), dict(m.last_phase_times())


def test_near_tie_specimen_is_exact_with_the_replay_and_explained_without(monkeypatch):
    import neartie
    case, g, inputs = load_golden("grd_cos_q_ac_d6_neartie")
    e, _ = _grow(case, inputs, monkeypatch, {})
    assert neartie.first_mismatch(g, e, case["policy"]) is None, "the replay should reproduce the reference's choice at the 547-row node"
    e_loop, _ = _grow(case, inputs, monkeypatch, {"GBRL_HIP_NO_SMALL_GROW": "1", "GBRL_HIP_NO_SMALL_PREP": "1"})
    assert neartie.first_mismatch(g, e_loop, case["policy"]) is None
    e_off, _ = _grow(case, inputs, monkeypatch, {"GBRL_HIP_NO_NEARTIE_REPLAY": "1"})
    mm = neartie.first_mismatch(g, e_off, case["policy"])
    assert mm is not None, "without the replay the product keeps the float64 maximum here (rounds 1-4)"
    info = neartie.explain_first_mismatch(case, inputs[0], inputs[1], inputs[2], g, e_off)
    assert info["explained"] and info["product_is_true_max"], info


def _sweep_case(seed):
    rng = np.random.default_rng(seed)
    policy = ["greedy", "oblivious"][int(rng.integers(0, 2))] if seed % 3 == 0 else "greedy"
    return dict(name="nt%d" % seed, seed=seed, N=int(rng.choice([300, 1000, 3000, 6000, 9000])), F=int(rng.choice([1, 3, 8, 17])), Fc=int(rng.choice([0, 0, 1])),
                D=int(rng.choice([1, 2, 3, 8, 11, 33])), depth=int(rng.choice([3, 4, 5])), n_bins=int(rng.choice([7, 64, 256])), score="Cosine",
                gen=["Quantile", "Uniform"][int(rng.integers(0, 2))], policy=policy, trees=int(rng.choice([1, 2, 4])), min_data_in_leaf=int(rng.choice([0, 0, 5])))


def test_replay_matches_the_reference_where_the_exact_argmax_does_not(monkeypatch):
    """Cosine cases against the reference build (the restatement where oracle/_ref is absent).  The replay may only ever turn a structural
    difference into agreement: every case that agrees without it agrees with it, both growth paths give the same bytes, and over the
    set there are fewer differences with the replay than without."""
    import neartie
    import oracle
    ref_mod = oracle.load_ref()
    diffs_on = diffs_off = replays = in_kernel = 0
    for seed in range(31000, 31060):
        case = _sweep_case(seed)
        inputs = K.make_inputs(case)
        r = ref_mod.GBRL(**K.ctor_kwargs(case)) if ref_mod is not None else oracle.OracleGBRL(**K.ctor_kwargs(case))
        K.drive(r, case, *inputs)
        g = {k: np.asarray(v) for k, v in r.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
        e_on, p_on, ph = _grow(case, inputs, monkeypatch, {}, counters=True)
        e_off, _ = _grow(case, inputs, monkeypatch, {"GBRL_HIP_NO_NEARTIE_REPLAY": "1"})
        e_loop, p_loop = _grow(case, inputs, monkeypatch, {"GBRL_HIP_NO_SMALL_GROW": "1"})
        for k in e_on:
            assert e_on[k].shape == e_loop[k].shape and e_on[k].tobytes() == e_loop[k].tobytes(), (case, k, "one-launch growth vs level loop")
        assert p_on.tobytes() == p_loop.tobytes(), case
        replays += ph.get("near_replays", 0)
        in_kernel += ph.get("near_in_kernel", 0)
        m_on, m_off = neartie.first_mismatch(g, e_on, case["policy"]), neartie.first_mismatch(g, e_off, case["policy"])
        diffs_on += m_on is not None
        diffs_off += m_off is not None
        if m_off is None:
            assert m_on is None, ("the replay broke a case that agreed without it", case)
        if m_on is not None:
            info = neartie.explain_first_mismatch(case, inputs[0], inputs[1], inputs[2], g, e_on)
            assert info["explained"], (case, info)
    print("near-tie sweep: %d cases, differences without the replay %d, with it %d, levels replayed by the level loop %d, inside the one-launch kernel %d" % (60, diffs_off, diffs_on, replays, in_kernel))
    assert diffs_on <= diffs_off
    assert replays > 0, "no level of 60 cases was replayed by the level loop: the detection never fired"
    assert in_kernel > 0, "the one-launch growth kernel never replayed a near-tie itself"


@pytest.mark.parametrize("policy,score,D,depth,mini", [("oblivious", "L2", 8, 7, 4096), ("greedy", "L2", 1, 4, 4096), ("greedy", "Cosine", 4, 5, 4096),
                                                        ("greedy", "L2", 2, 5, 8192), ("oblivious", "Cosine", 3, 6, 6000)])   # (4097+ rows: the int64 histogram variant)
def test_one_launch_kernel_replays_what_the_level_loop_replays(policy, score, D, depth, mini, monkeypatch):
    """A boosting-like loop on 4096-row minibatches with structured gradients (where near-ties are frequent: one tree in 30-50): the
    one-launch kernel -- which replays a flagged level itself, greedy and oblivious -- must grow the bytes the level loop grows, and must
    have met flagged levels (otherwise this test shows nothing)."""
    import gbrl_amd
    rng = np.random.default_rng(5)
    N, F, trees = 1 << 15, 24, 120
    X = rng.standard_normal((N, F)).astype(np.float32)
    W = rng.standard_normal((6, D)).astype(np.float32)
    out = {}
    for name, env in (("kernel", {}), ("loop", {"GBRL_HIP_NO_SMALL_GROW": "1"})):
        for k in HOOKS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        case = dict(name="ik", seed=1, N=mini, F=F, Fc=0, D=D, depth=depth, n_bins=256, score=score, gen="Quantile", policy=policy, trees=1, min_data_in_leaf=0)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        m.set_feature_weights(np.ones(F, np.float32))
        for o in K.optimizers(case):
            m.set_optimizer(**o)
        m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
        m.set_profiling(2)
        g_rng = np.random.default_rng(9)
        for i in range(trees):
            o = (i * mini) % (N - mini + 1)
            xs = np.ascontiguousarray(X[o:o + mini])
            g = (np.tanh(xs[:, :6] @ W) + 0.5 * g_rng.standard_normal((mini, D))).astype(np.float32)
            m.step(xs, None, np.ascontiguousarray(g))
        out[name] = ({k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}, dict(m.last_phase_times()))
    (ek, pk), (el, pl) = out["kernel"], out["loop"]
    for k in ek:
        assert ek[k].shape == el[k].shape and ek[k].tobytes() == el[k].tobytes(), (policy, score, k)
    print("in-kernel replays %d, trees handed to the level loop %d; level loop alone replayed %d levels" % (pk.get("near_in_kernel", 0), pk.get("near_bailouts", 0), pl.get("near_replays", 0)))
    assert pl.get("near_replays", 0) > 0, "no level of this loop was flagged: choose other inputs"
    assert pk.get("near_in_kernel", 0) > 0, "the one-launch kernel never replayed a level itself"


BIGN = ["bign9", "bign30", "bign156", "bign162", "bign356"]


@pytest.mark.parametrize("name", BIGN)
def test_near_ties_above_65536_rows_follow_the_reference_on_request(name, monkeypatch):
    """Round 6: batches above the LDS bitmaps' 65 536 rows.  Five specimens of scripts/bign_sweep.py (143 357 .. 327 945 rows; fixtures written by
    the REFERENCE build, tests/golden/make_fullsize_golden.py) in which the reference's float32 summation noise preferred a candidate 6e-7 ..
    9e-6 below the exact maximum.  Default: such batches keep the exact arg-max (rounds 1-5) -- the tree differs from the reference's and the
    first difference is an explained near-tie.  GBRL_HIP_NEARTIE_MAX_ROWS=0 (every node replayed; the row bitmap lives in global memory):
    structure bit-identical to the reference's tree.  GBRL_HIP_NEARTIE_MAX_ROWS=65536: identical exactly where the deciding node is that small."""
    import json
    import os
    import gbrl_amd
    import neartie
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    case = json.loads(str(fx["case_json"]))
    X, Xc, G, y = K.make_inputs(case)
    assert K.inputs_digest(X, Xc, G, y) == str(fx["inputs_sha256"])
    ref = {k: fx[k] for k in fx.files}
    ref["categorical_values"] = np.zeros(ref["feature_indices"].shape, "S128")
    keys = ("tree_indices", "depths", "feature_indices", "inequality_directions")

    def grow(limit):
        monkeypatch.delenv("GBRL_HIP_NO_NEARTIE_REPLAY", raising=False)
        if limit is None:
            monkeypatch.delenv("GBRL_HIP_NEARTIE_MAX_ROWS", raising=False)
        else:
            monkeypatch.setenv("GBRL_HIP_NEARTIE_MAX_ROWS", str(limit))
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, None, G, None)
        return {k: np.asarray(v) for k, v in m.get_ensemble_data().items()}

    def same(e):
        return all(np.array_equal(e[k], ref[k]) for k in keys) and np.array_equal(e["feature_values"].view(np.uint32), ref["feature_values"].view(np.uint32))

    e_all = grow(0)
    assert same(e_all), "with every node replayed the tree must be the reference's"
    # the chains through seqsum.hip (default for these shapes when D % 4 == 0) and through the one-lane-per-chain core: the same bytes
    monkeypatch.setenv("GBRL_HIP_NEARTIE_SERIAL", "1")
    e_serial = grow(0)
    monkeypatch.delenv("GBRL_HIP_NEARTIE_SERIAL")
    for k in e_all:
        assert np.asarray(e_all[k]).tobytes() == np.asarray(e_serial[k]).tobytes(), k
    scale = float(np.abs(G).mean())
    assert float(np.max(np.abs(e_all["values"] - ref["values"]) / np.maximum(np.abs(ref["values"]), scale))) <= 1e-5
    e_def = grow(None)
    assert not same(e_def)                                 # (the specimens were selected because the exact arg-max differs)
    why = neartie.explain_first_mismatch(case, X, None, G, ref, e_def)
    assert why and why["explained"] and why["product_is_true_max"], why
    small_node = {"bign30": True, "bign162": True, "bign356": True, "bign9": False, "bign156": False}[name]   # deciding node <= 65 536 rows?
    assert same(grow(65536)) == small_node


@pytest.mark.parametrize("policy,score,D,Fc", [("greedy", "Cosine", 4, 1), ("oblivious", "L2", 8, 1), ("greedy", "L2", 12, 0), ("oblivious", "Cosine", 4, 1), ("greedy", "Cosine", 6, 0)])
def test_big_batch_replay_parallel_chains_equal_the_one_lane_core(policy, score, D, Fc, monkeypatch):
    """Batches above 65 536 rows with every near-tie replayed, through seqsum.hip's parallel chains and through the one-lane core
    (GBRL_HIP_NEARTIE_SERIAL=1): the same bytes.  The inputs are built to BE near-ties: a step signal on column 0, column 1 a copy of column 0
    with 1e-4 of noise (its thresholds cut the rows a few places away from column 0's), and -- Fc = 1 -- a categorical column that says on which
    side of a nearby value column 0 lies (the side test of a category is `code == class`).  Widths that are and are not multiples of 4 (D = 6:
    the parallel evaluation declines and both runs take the core).  The replay counter must say that levels were replayed."""
    import gbrl_amd
    rng = np.random.default_rng(200 + D)
    N, F = 150000 + 1000 * D, 4
    X = rng.standard_normal((N, F)).astype(np.float32)
    X[:, 1] = (X[:, 0] + 1e-4 * rng.standard_normal(N)).astype(np.float32)
    step = np.where(X[:, 0] > 0.1, 1.0, -1.0)
    G = (step[:, None] * rng.uniform(0.5, 1.5, D)[None, :] + 2.0 * rng.standard_normal((N, D)) + 0.2).astype(np.float32)
    Xc = None
    if Fc:
        Xc = np.where(X[:, 0] > 0.1003, K.TOKENS[1], K.TOKENS[0]).reshape(N, 1).astype("S128")
    case = dict(name="bigrep", seed=0, N=N, F=F, Fc=Fc, D=D, depth=3, n_bins=256, score=score, gen="Quantile", policy=policy, trees=1)
    out = []
    for serial in ("0", "1"):
        monkeypatch.delenv("GBRL_HIP_NO_NEARTIE_REPLAY", raising=False)
        monkeypatch.setenv("GBRL_HIP_NEARTIE_MAX_ROWS", "0")
        monkeypatch.setenv("GBRL_HIP_NEARTIE_SERIAL", serial)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        m.set_profiling(2)
        K.drive(m, case, X, Xc, G, None)
        m.step(X, Xc, np.ascontiguousarray(G))
        out.append(({k: np.asarray(v) for k, v in m.get_ensemble_data().items()}, dict(m.last_phase_times()).get("near_replays", 0)))
    (a, ra), (b, rb) = out
    assert ra == rb and ra >= 1, (ra, rb)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
