"""seqsum.hip: sequential float32 sums evaluated in parallel -- block summaries that depend on the running sum's parity only, the serial loop
where the running sum crosses a power of two -- must equal the plain loop `s = float32(s + x)` BIT FOR BIT (round 6; the near-tie replay's
chains over up to 2^20 rows, node.cpp:336-352).  Random chains with drift, without, heavy-tailed, tie-heavy (multiples of 1/8), sign changes,
cancellations, zeros, huge and tiny elements, inf / nan, lengths around the 256-element block size, non-zero starts."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lib():
    import gbrl_amd
    lib = ctypes.CDLL(os.path.join(os.path.dirname(gbrl_amd.__file__), "libgbrl_hip.so"))
    lib.gbrl_hip_seq_sums.restype = ctypes.c_int
    lib.gbrl_hip_seq_sums.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.gbrl_hip_last_error.restype = ctypes.c_char_p
    return lib


def _plain(x, start):
    s = np.float32(start)
    with np.errstate(all="ignore"):
        for v in x:
            s = np.float32(s + v)
    return s


def _device(lib, chains, starts):
    x = np.ascontiguousarray(np.concatenate(chains) if chains else np.zeros(0, np.float32), np.float32)
    lens = np.array([len(c) for c in chains], np.uint32)
    st = np.ascontiguousarray(starts, np.float32)
    out = np.zeros(len(chains), np.float32)
    slow = np.zeros(1, np.uint32)
    rc = lib.gbrl_hip_seq_sums(x.ctypes.data, lens.ctypes.data, st.ctypes.data, len(chains), out.ctypes.data, slow.ctypes.data)
    assert rc == 0, lib.gbrl_hip_last_error()
    return out, int(slow[0])


def _chain(rng, kind, n):
    if kind == 0: return rng.standard_normal(n).astype(np.float32)
    if kind == 1: return (rng.standard_normal(n) + 0.3).astype(np.float32)
    if kind == 2: return (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 3)).astype(np.float32)
    if kind == 3: return (np.round(rng.standard_normal(n) * 8) / 8 + 0.5).astype(np.float32)           # many exact ties
    if kind == 4: return (-np.abs(rng.standard_normal(n))).astype(np.float32)
    if kind == 5: return (rng.integers(-3, 4, n) * 0.25).astype(np.float32)                           # returns to zero again and again
    if kind == 6:
        x = (rng.standard_normal(n) + 1.0).astype(np.float32)
        x[rng.integers(0, n, max(1, n // 200))] = np.float32(1e30)                                   # elements far above the running sum
        x[rng.integers(0, n, max(1, n // 200))] = np.float32(-1e30)
        return x
    if kind == 7:
        x = (rng.standard_normal(n) * 1e-3 + 1.0).astype(np.float32)
        x[rng.integers(0, n, max(1, n // 100))] = np.float32(1e-38)                                  # subnormal-range dust
        x[rng.integers(0, n, max(1, n // 100))] = np.float32(0.0)
        return x
    if kind == 8:
        x = (rng.standard_normal(n) + 0.5).astype(np.float32)
        if n > 3: x[n // 2] = np.float32(np.inf) if rng.integers(0, 2) else np.float32(np.nan)
        return x
    return (np.float32(2.0) ** rng.integers(-30, 30, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)   # pure powers of two


def test_parallel_sequential_sums_equal_the_plain_loop_bit_for_bit():
    lib = _lib()
    rng = np.random.default_rng(5)
    lengths = [0, 1, 2, 3, 63, 64, 255, 256, 257, 511, 512, 513, 1000, 4096, 4097, 20000]
    chains, starts = [], []
    for t in range(400):
        n = lengths[t % len(lengths)] if t < 160 else int(rng.integers(1, 6000))
        chains.append(_chain(rng, t % 10, n) if n else np.zeros(0, np.float32))
        starts.append(np.float32(0.0) if t % 3 else np.float32(rng.standard_normal() * 100))
    got, slow = _device(lib, chains, starts)
    bad = []
    for i, (c, s0) in enumerate(zip(chains, starts)):
        want = _plain(c, s0)
        if np.float32(got[i]).tobytes() != want.tobytes() and not (np.isnan(got[i]) and np.isnan(want)):
            bad.append((i, i % 10, len(c), float(want), float(got[i])))
    assert not bad, bad[:10]
    n_blocks = sum((len(c) + 255) // 256 for c in chains)
    print("400 chains, %d blocks of 256 elements, %d took the serial fallback" % (n_blocks, slow))


def test_long_chains_mostly_take_the_summaries():
    """2^20-element chains of the replay's kinds (a drifting column sum, a zero-mean one, a dot chain of rounded products): exact, and all but a
    few per cent of the blocks are applied in O(1)."""
    lib = _lib()
    rng = np.random.default_rng(6)
    n = 1 << 20
    chains = [(rng.standard_normal(n) * 0.7 + 0.3).astype(np.float32), rng.standard_normal(n).astype(np.float32),
              ((rng.standard_normal(n) + 0.3) * np.float32(0.31)).astype(np.float32)]
    got, slow = _device(lib, chains, [0.0, 0.0, 0.0])
    for i, c in enumerate(chains):
        want = np.float32(0)
        # (the plain loop in NumPy scalars is slow: cumulative float32 sum IS the sequential loop)
        want = np.cumsum(c, dtype=np.float32)[-1]
        assert np.float32(got[i]).tobytes() == np.float32(want).tobytes(), (i, float(want), float(got[i]))
    n_blocks = 3 * (n // 256)
    print("3 chains of 2^20 elements: %d of %d blocks took the serial fallback" % (slow, n_blocks))
    assert slow <= n_blocks // 4
