"""World-size-2 CPU (gloo) tests of the row-sharded protocol.

There is no GPU here, so the HIP kernels cannot run; what IS exercised is (a) the exact hook code the engine calls at its
exchange points (gbrl_amd/dist.py::TorchCollective, with host pointers instead of device pointers), through the same
ctypes callback signatures as include/gbrl_hip.h's gbrl_hip_collective, and (b) the protocol itself, emulated in NumPy
with the engine's arithmetic (integer fixed-point histograms, integer quantile counts, integer leaf sums): every rank
must reach the SAME decision as a single process holding all rows, bit for bit, for any split of the rows."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hist(codes, q, n_bins):
    """int64 histogram [F][n_bins][D+1] of fixed-point gradients q (int64) keyed by class codes."""
    n, F = codes.shape
    D = q.shape[1]
    h = np.zeros((F, n_bins, D + 1), np.int64)
    for f in range(F):
        for d in range(D):
            np.add.at(h[f, :, d], codes[:, f], q[:, d])
        np.add.at(h[f, :, D], codes[:, f], 1)
    return h


def _best_split(h):
    """argmax over (feature, threshold) of |S_L|^2/n_L + |S_R|^2/n_R from suffix sums; lowest index wins ties."""
    F, NB, W = h.shape
    suf = np.cumsum(h[:, ::-1, :], axis=1)[:, ::-1, :]
    tot = suf[:, 0, :]
    best, arg = -np.inf, -1
    for f in range(F):
        for k in range(NB - 1):
            R = suf[f, k + 1]
            L = tot[f] - R
            x = 0.0
            if L[-1] > 0:
                x += float((L[:-1].astype(np.float64) ** 2).sum()) / float(L[-1])
            if R[-1] > 0:
                x += float((R[:-1].astype(np.float64) ** 2).sum()) / float(R[-1])
            x = np.float32(x)
            if x > best:
                best, arg = x, f * (NB - 1) + k
    return arg, best


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gbrl_amd.dist import TorchCollective
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = TorchCollective(device=None)
    assert coll.world_size == world and coll.rank == rank and coll.struct.world_size == world
    # ---- (a) the four hooks on host buffers -------------------------------------------------------------------------
    a = np.arange(10, dtype=np.int64) * (rank + 1)
    assert coll.allreduce_sum_i64(a) == 0 and np.array_equal(a, np.arange(10) * 3)
    b = np.full(5, 0.5 + rank, np.float64)
    assert coll.allreduce_sum_f64(b) == 0 and np.allclose(b, 2.0)
    c = np.array([rank, -rank, 7.0], np.float32)
    d = c.copy()
    assert coll.allreduce_max_f32(c) == 0 and np.array_equal(c, [1, 0, 7])
    assert coll.allreduce_min_f32(d) == 0 and np.array_equal(d, [0, -1, 7])
    # ---- (b) the sharded step protocol --------------------------------------------------------------------------------
    rng = np.random.default_rng(5)            # every rank generates the FULL data set and takes its contiguous block
    N, F, D, NB = 3001, 6, 3, 17
    X = rng.standard_normal((N, F)).astype(np.float32)
    G = (np.tanh(X[:, :D]) + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    lo, hi = (0, 1234) if rank == 0 else (1234, N)   # deliberately uneven shards
    Xs, Gs = X[lo:hi], G[lo:hi]
    # 1. gradient statistics: global sum -> mean; centred squares -> std (fp64 sums exchanged)
    #    (round 6: the first message also carries the rank's row count -- integers below 2^53 add exactly in float64 -- so the global count needs no
    #    exchange of its own: engine_step.hip::exchange_stats)
    msg = np.concatenate([Gs.astype(np.float64).sum(axis=0), [float(hi - lo)]])
    coll.allreduce_sum_f64(msg)
    s1, n_from_message = msg[:D], int(msg[D])
    assert n_from_message == N
    mean = (s1 / n_from_message).astype(np.float32)
    s2 = ((Gs - mean).astype(np.float64) ** 2).sum(axis=0)
    coll.allreduce_sum_f64(s2)
    den = (np.sqrt((s2.astype(np.float32)) * np.float32(1.0 / (N - 1.0))) + np.float32(1e-8)).astype(np.float32)
    mx = np.abs((Gs - mean) / den).max(axis=0).astype(np.float32)
    coll.allreduce_max_f32(mx)
    scale = 2.0 ** 12
    q = np.rint(((Gs - mean) / den) * scale).astype(np.int64)
    # 2. quantile thresholds by bisection on counts (only integer counts cross ranks): one feature, a few ranks
    ranks = np.cumsum(np.full(NB - 1, N // NB))
    keys = Xs.view(np.uint32).astype(np.int64)
    keys = np.where(keys & 0x80000000, 0xFFFFFFFF - keys, keys | 0x80000000)      # order-preserving keys
    thr_keys = np.zeros((F, NB - 1), np.int64)
    for bit in range(31, -1, -1):
        trial = thr_keys | (1 << bit)
        cnt = np.stack([(keys[:, f][:, None] < trial[f][None, :]).sum(axis=0) for f in range(F)]).astype(np.int64)
        coll.allreduce_sum_i64(cnt)
        thr_keys = np.where(cnt < ranks[None, :], trial, thr_keys)
    # 3. codes, per-level integer histograms, all-reduce, split decision
    codes = np.stack([(keys[:, f][:, None] > thr_keys[f][None, :]).sum(axis=1) for f in range(F)], axis=1)
    h = _hist(codes, q, NB)
    coll.allreduce_sum_i64(h.reshape(-1))
    arg, best = _best_split(h)
    # 4. leaf sums of raw gradients (fixed point) for the two children
    f_star, k_star = divmod(arg, NB - 1)
    right = codes[:, f_star] > k_star
    leaf = np.zeros((2, D + 1), np.int64)
    for side in (0, 1):
        sel = right == bool(side)
        leaf[side, :D] = np.rint(Gs[sel].astype(np.float64) * 2.0 ** 30).astype(np.int64).sum(axis=0)
        leaf[side, D] = sel.sum()
    coll.allreduce_sum_i64(leaf.reshape(-1))
    np.savez(os.path.join(result_dir, f"r{rank}.npz"), thr=thr_keys, arg=arg, best=best, leaf=leaf, hist=h, calls=coll.calls)
    if rank == 0:   # single-process answer on all rows, same arithmetic
        keys_all = X.view(np.uint32).astype(np.int64)
        keys_all = np.where(keys_all & 0x80000000, 0xFFFFFFFF - keys_all, keys_all | 0x80000000)
        want_thr = np.stack([np.sort(keys_all[:, f])[ranks - 1] for f in range(F)])
        m = (G.astype(np.float64).sum(axis=0) / N).astype(np.float32)
        dd = (np.sqrt((((G - m).astype(np.float64) ** 2).sum(axis=0)).astype(np.float32) * np.float32(1.0 / (N - 1.0))) + np.float32(1e-8)).astype(np.float32)
        qa = np.rint(((G - m) / dd) * scale).astype(np.int64)
        ca = np.stack([(keys_all[:, f][:, None] > want_thr[f][None, :]).sum(axis=1) for f in range(F)], axis=1)
        ha = _hist(ca, qa, NB)
        np.savez(os.path.join(result_dir, "single.npz"), thr=want_thr, hist=ha, arg=_best_split(ha)[0])
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_sharded_protocol(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1, single = (np.load(tmp_path / n) for n in ("r0.npz", "r1.npz", "single.npz"))
    for k in ("thr", "arg", "best", "leaf", "hist"):
        assert np.array_equal(r0[k], r1[k]), k                    # every rank holds the same state
    assert np.array_equal(r0["thr"], single["thr"])               # distributed exact quantiles == global sort
    assert np.array_equal(r0["hist"], single["hist"])             # integer histograms: shard-count invariant
    assert int(r0["arg"]) == int(single["arg"])                   # same split
    assert int(r0["calls"]) > 30


def test_collective_struct_matches_the_c_header():
    """ctypes mirror of gbrl_hip_collective: pointer, two ints, four function pointers."""
    import ctypes as C
    from gbrl_amd.dist import _Coll
    assert [n for n, _ in _Coll._fields_] == ["ctx", "world_size", "rank", "allreduce_sum_i64", "allreduce_sum_f64",
                                              "allreduce_max_f32", "allreduce_min_f32"]
    assert C.sizeof(_Coll) == 8 + 4 + 4 + 4 * 8
