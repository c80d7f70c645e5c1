"""Prototype: exact evaluation of a sequential float32 sum through block summaries (functions of the running sum's parity)."""
import numpy as np

def seq_sum(x, s0=np.float32(0)):
    s = np.float32(s0)
    for v in x:
        s = np.float32(s + v)
    return s

def decomp(x):
    """x (float32, finite) -> (sign, mantissa int, exponent e) with |x| = m * 2^(e-23), m in [2^23, 2^24) (normal) or e = -126 (subnormal)"""
    b = np.asarray(x, np.float32).view(np.uint32).astype(np.int64)
    sign = np.where(b >> 31, -1, 1)
    ex = (b >> 23) & 0xff
    man = b & 0x7fffff
    m = np.where(ex == 0, man, man | (1 << 23))
    e = np.where(ex == 0, -126, ex - 127)
    return sign, m, e

def block_summary(x, e):
    """Under the hypothesis that every partial sum has exponent e (ulp u = 2^(e-23)): per element f = floor(x/u), h in {0,1,2 (tie)};
    returns for parity p in (0,1): (delta, min prefix, max prefix) of A (in units of u) relative to A0, or None if an element is too big."""
    sign, m, ex = decomp(x)
    k = e - ex                                   # right shift of the mantissa
    res = []
    f = np.zeros(len(x), np.int64); h = np.zeros(len(x), np.int64)
    for i in range(len(x)):
        ki = int(k[i]); mi = int(m[i]) * int(sign[i])
        if ki <= 0:
            if -ki > 30: return None
            fi = mi << (-ki); hi = 0
        else:
            if ki > 60: fi = -1 if mi < 0 else 0; rem = (mi != 0); half = 2   # tiny: fraction in (0,1) never reaches a half from below... handled below
            fi = mi >> min(ki, 62)                # floor division by 2^ki (python ints: arithmetic shift = floor)
            if ki > 62:
                hi = 0 if mi >= 0 else 1          # x in (-u/2.., 0): floor = -1, frac = 1 - tiny > 1/2 -> +1 -> net 0
                if mi == 0: fi, hi = 0, 0
            else:
                rem = mi - (fi << ki)             # in [0, 2^ki)
                halfv = 1 << (ki - 1)
                hi = 0 if rem < halfv else (1 if rem > halfv else 2)
        f[i] = fi; h[i] = hi
    out = []
    for p in (0, 1):
        A = p; lo = hi_ = 0
        for i in range(len(x)):
            t = A + int(f[i])
            if h[i] == 1: t += 1
            elif h[i] == 2: t += (t & 1)
            A = t
            lo = min(lo, A - p); hi_ = max(hi_, A - p)
        out.append((A - p, lo, hi_))
    return out

def fast_sum(x, block=64):
    """sequential float32 sum of x through block summaries; falls back to the serial loop where the binade hypothesis fails."""
    s = np.float32(0)
    n_fast = n_slow = 0
    for b0 in range(0, len(x), block):
        xb = x[b0:b0 + block]
        ok = False
        if s != 0 and np.isfinite(s):
            sg, m, e = decomp(np.array([s], np.float32))
            sg, m, e = int(sg[0]), int(m[0]), int(e[0])
            if m >= (1 << 23):                      # normal
                summ = block_summary(xb, e)
                if summ is not None:
                    A0 = sg * m
                    d, lo, hi = summ[A0 & 1]
                    if sg > 0: ok = (A0 + lo >= (1 << 23)) and (A0 + hi < (1 << 24))
                    else: ok = (A0 + hi <= -(1 << 23)) and (A0 + lo > -(1 << 24))
                    if ok:
                        A = A0 + d
                        s = np.float32(np.ldexp(np.float64(A), e - 23))
                        n_fast += 1
        if not ok:
            s = seq_sum(xb, s)
            n_slow += 1
    return s, n_fast, n_slow

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    bad = 0
    for t in range(300):
        n = int(rng.integers(1, 3000))
        kind = t % 6
        if kind == 0: x = rng.standard_normal(n).astype(np.float32)
        elif kind == 1: x = (rng.standard_normal(n) + 0.3).astype(np.float32)
        elif kind == 2: x = (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 3)).astype(np.float32)
        elif kind == 3: x = (np.round(rng.standard_normal(n) * 8) / 8 + 0.5).astype(np.float32)      # many ties
        elif kind == 4: x = np.abs(rng.standard_normal(n)).astype(np.float32) * np.float32(-1 if t % 12 == 4 else 1)
        else: x = (rng.integers(-3, 4, n) * 0.25).astype(np.float32)
        want = seq_sum(x)
        got, nf, ns = fast_sum(x)
        if want.tobytes() != got.tobytes():
            bad += 1
            print("MISMATCH", t, kind, n, want, got)
        if t < 12: print(t, kind, n, want, got, "fast blocks", nf, "slow", ns)
    print("bad", bad)
