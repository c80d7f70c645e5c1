#!/usr/bin/env python3
"""bench.py -- trees-fit/sec (+ predict rows/sec) of the MI355X-native GBRL hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Timed workload (BASELINE.json configs[1], the configuration the metric is quoted on): oblivious tree, L2 split score, quantile
candidates, batch = 2^20 rows PER GPU, 128 features, depth 6, output_dim 8, n_bins 256, one SGD optimiser; synthetic
inputs X ~ N(0,1), G = tanh(X[:, :8] W) + 0.5 N(0,1) generated on the device before the timed region (inputs resident in
HBM).  One "step" = one GBRL.step() = one tree fitted on the whole batch.  With N > 1 the rows are sharded over the ranks
(weak scaling: 2^20 rows per GPU, the SAME tree is grown on every rank from all-reduced integer histograms), and `value`
counts 2^20-row batches: value = steps * n_gpus / seconds.  `python bench.py --gpus N` without a launcher starts its own N ranks.

Printed JSON (rank 0, one line): the driver contract + "roofline" for the dominant kernel (split-score histogram build,
HBM-bound, algorithmic bytes from SURVEY.md 8(d)) + "cpu_baseline" (the reference's own CPU path if oracle/_ref is
loadable, else this repo's restatement) on a bounded sample + predict throughput with its own roofline ("predict",
"predict_large_ensemble") + two further single-GPU legs measured after the timed region: "cfg3" (BASELINE configs[2]: greedy /
Cosine / policy + value optimisers, full size) and "predict_cfg5" (BASELINE configs[4]: 192 numeric + 64 categorical columns,
uniform candidates, predict over 10 000 trees).  --workload cfg3 makes configs[2] the timed workload instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def hist_algorithmic_bytes(n_rows, n_feat, out_dim, depth, n_bins):
    """per level read N*(F*1 B code + D*4 B grads + 4 B row id) + write 2^l * F*B*(D+1)*4 B."""
    per_level_read = n_rows * (n_feat * 1 + out_dim * 4 + 4)
    hist = n_feat * n_bins * (out_dim + 1) * 4
    return depth * per_level_read + ((1 << depth) - 1) * hist


def cpu_baseline(n_feat, out_dim, depth, n_bins, full_rows, sample_rows, budget_s=20.0):
    """Reference CPU path on a bounded sample of the same workload (rows only are reduced).  The reference's brute-force
    scan is linear in the row count (depth * N * candidates * (1 + D)), so trees/s at the full batch is reported as the
    sample's trees/s * sample_rows / full_rows (optimistic for the CPU: larger samples fall out of its caches and run
    slower per row).  sample_rows <= 0: the sample is grown 4x at a time from 4096 rows for as long as the next size is
    predicted (at 6x the last time) to keep the whole leg within `budget_s` seconds; the largest sample is reported."""
    import numpy as np
    import oracle
    if sample_rows <= 0:
        spent, rows, best = 0.0, 4096, None
        while True:
            best = cpu_baseline(n_feat, out_dim, depth, n_bins, full_rows, rows)
            spent += best["sample_seconds"]
            if rows * 4 > full_rows or spent + 6.0 * best["sample_seconds"] > budget_s:
                return best
            rows *= 4
    rng = np.random.default_rng(0)
    X = rng.standard_normal((sample_rows, n_feat)).astype(np.float32)
    W = rng.standard_normal((8, out_dim)).astype(np.float32)
    G = (np.tanh(X[:, :8] @ W) + 0.5 * rng.standard_normal((sample_rows, out_dim))).astype(np.float32)
    kind, mod = "reference", oracle.load_ref()
    kw = dict(input_dim=n_feat, output_dim=out_dim, policy_dim=out_dim, max_depth=depth, min_data_in_leaf=0, n_bins=n_bins,
              par_th=10, cv_beta=0.9, split_score_func="L2", generator_type="Quantile", use_control_variates=False,
              batch_size=5000, grow_policy="oblivious", verbose=0, device="cpu", learner_name="cpu_baseline")
    if mod is not None:
        m = mod.GBRL(**kw)
    else:
        kind = "port"
        m = oracle.OracleGBRL(**kw)
    m.set_feature_weights(np.ones(n_feat, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=out_dim)
    m.set_feature_mapping(np.arange(n_feat, dtype=np.int32), np.ones(n_feat, dtype=bool))
    t0 = time.perf_counter()
    m.step(X, None, G)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    m.predict(X, None, 0, 0)
    dtp = time.perf_counter() - t1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"value": (1.0 / dt) * sample_rows / full_rows, "unit": "trees/s at batch=2^20 (extrapolated x%d linearly in rows from the sample)" % max(1, full_rows // sample_rows),
            "cores": cores, "kind": kind,
            "sample": "1 tree on %d of %d rows (same F=%d, D=%d, depth=%d, n_bins=%d): %.2f s; predict 1 tree %.3f s" % (
                sample_rows, full_rows, n_feat, out_dim, depth, n_bins, dt, dtp),
            "sample_seconds": dt}


def committed_full_tree():
    """The builder's own full-size measurement of the reference (ONE tree at 2^20 x 128 on the GPU box's host cores, `bench.py --cpu-full-tree`),
    read from the committed profile and labelled as such: the row extrapolation of the sample is 30-45 % optimistic for the CPU."""
    for name in ("r06_cpu_full_tree.json", "r05_cpu_full_tree.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            j = json.load(open(path))
            ft = j.get("cpu_baseline", {}).get("full_tree", j.get("full_tree", j))
            return {"full_tree_measured_s": ft["seconds_per_tree"], "full_tree_measured_cores": ft.get("cores"),
                    "full_tree_measured_by": "builder, not by this run: profiles/%s (one reference tree at the full 2^20 x 128 batch, no extrapolation)" % name}
        except Exception:
            continue
    return {}


def cpu_full_tree(gbrl_amd):
    """--cpu-full-tree: ONE tree of the reference's CPU path at the full batch of BASELINE configs[1] (2^20 x 128, D = 8, depth 6, all host
    cores: minutes), on the inputs of the committed fixture tests/golden/full_cfg2.npz (tests/golden/cases.py::make_inputs -- integer PCG64
    draws + exactly rounded float32 arithmetic, SHA-256 checked), and the SAME tree grown by the product on the GPU: the two structures are
    compared level by level, and both with the fixture (the reference on 8 threads in the build container)."""
    import numpy as np
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as K
    case = K.FULLSIZE_BY_NAME["full_cfg2"]
    X, Xc, G, y = K.make_inputs(case)
    sha = K.inputs_digest(X, Xc, G, y)
    F, D = case["F"], case["D"]

    def drive_one(m):
        m.set_feature_weights(np.ones(F, np.float32))
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
        m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
        t0 = time.perf_counter()
        m.step(X, None, G)
        dt = time.perf_counter() - t0
        e = m.get_ensemble_data()
        return dt, {k: np.asarray(e[k]) for k in ("depths", "feature_indices", "feature_values", "values")}

    def levels_equal(a, b):
        da, db = int(a["depths"][0]), int(b["depths"][0])
        k = 0
        while k < min(da, db) and int(a["feature_indices"][0][k]) == int(b["feature_indices"][0][k]) and \
                a["feature_values"][0][k:k + 1].view(np.uint32)[0] == b["feature_values"][0][k:k + 1].view(np.uint32)[0]:
            k += 1
        first = None if (k == da == db) else {"level": k, "a": [int(a["feature_indices"][0][k]), float(a["feature_values"][0][k])] if k < da else None,
                                               "b": [int(b["feature_indices"][0][k]), float(b["feature_values"][0][k])] if k < db else None}
        return k, da, db, first

    mod = oracle.load_ref()
    if mod is None:
        return {"error": "oracle/_ref is not built here"}
    dt_ref, e_ref = drive_one(mod.GBRL(**K.ctor_kwargs(case)))
    dt_gpu, e_gpu = drive_one(gbrl_amd.GBRL(**K.ctor_kwargs(case)))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    k, da, db, first = levels_equal(e_ref, e_gpu)
    out = {"what": "ONE tree of the reference's CPU path at the full batch (2^20 x 128, D = 8, depth 6), no extrapolation, and the product's tree on the same inputs",
           "inputs_sha256": sha, "seconds_per_tree": dt_ref, "trees_per_s": 1.0 / dt_ref, "rows": case["N"], "cores": cores, "kind": "reference",
           "omp_threads": os.environ.get("OMP_NUM_THREADS", "unset (all cores)"),
           "product_seconds_host_inputs": dt_gpu,
           "structure_equal": bool(first is None), "levels_identical": k, "depth_reference": da, "depth_product": db, "first_difference": first,
           "reference_tree": {"feature_indices": e_ref["feature_indices"][0].tolist(), "feature_values": [float(v) for v in e_ref["feature_values"][0]]},
           "product_tree": {"feature_indices": e_gpu["feature_indices"][0].tolist(), "feature_values": [float(v) for v in e_gpu["feature_values"][0]]}}
    if first is None:
        scale = float(np.abs(G).mean())
        out["leaf_values_max_rel_err"] = float(np.max(np.abs(e_ref["values"] - e_gpu["values"]) / np.maximum(np.abs(e_ref["values"]), scale)))
    fpath = os.path.join(ROOT, "tests", "golden", "full_cfg2.npz")
    if os.path.exists(fpath):
        fx = np.load(fpath)
        if str(fx["inputs_sha256"]) == sha:
            fxe = {kk: fx[kk] for kk in ("depths", "feature_indices", "feature_values", "values")}
            out["fixture_threads"] = int(fx["omp_threads"])
            out["reference_here_equals_fixture"] = bool(levels_equal(e_ref, fxe)[3] is None)
            out["product_equals_fixture"] = bool(levels_equal(e_gpu, fxe)[3] is None)
            out["product_levels_identical_to_fixture"] = levels_equal(e_gpu, fxe)[0]
    return out


def launch_ranks(n):
    """Start `n` copies of this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the
    contract torch.distributed.run uses), wait for all of them and return the worst exit code.  Children inherit stdout, so
    rank 0's JSON line is the launcher's output."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # Poll all ranks: the first one to fail takes the others down (they may be blocked inside an RCCL collective waiting for it),
    # and the whole launch has a wall-clock limit -- the launcher must exit non-zero, not hang (torch.distributed.run does the same).
    limit = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3600"))
    t0 = time.monotonic()
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0:
                    rc = abs(r) or 1
            if time.monotonic() - t0 > limit:
                rc = 124
            if live and rc == 0:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
    return rc


# ---- predict roofline -------------------------------------------------------------------------------------------------
# Small ensembles are HBM-bound: N*(F + D)*4 bytes + model per call.  Large ensembles are bound by instruction
# issue.  Two bounds are reported:
#   "issue" (rounds 2-3, kept so that records stay comparable): a traversal that reads features from an LDS tile needs, per (64 rows,
#       depth-6 tree, 8 outputs), 6 feature reads + 6 compares + 6 index updates and, for the 4 output pairs, 4 offset extractions +
#       4 value reads + 4 packed FMAs: 30 VALU instructions (4 cycles each on a SIMD-32 for the compare / carry / packed forms, 4 SIMDs
#       per CU) and 10 LDS reads -> 30 cycles per CU -> 256 CUs * 2.4e9 / 30 * 64 = 1.31e12 row-trees/s.
#   "issue_reg" (round 4): the register-tile kernel (predict_reg.hip) keeps the row in VGPRs -- 6 VGPR-relative compares + 6 add-with-
#       carry + 1 address + 4 packed FMAs = 17 VALU, 4 LDS value reads, 6 scalar M0 writes: 17 cycles per CU by the same accounting ->
#       2.31e12 row-trees/s.  This is the bound of the formulation that ships.
PREDICT_ISSUE_BOUND = 256 * 2.4e9 / 30.0 * 64.0
PREDICT_ISSUE_BOUND_REG = 256 * 2.4e9 / 17.0 * 64.0


def predict_roofline(n_rows, n_feat, out_dim, trees, depth, seconds):
    model_bytes = trees * ((1 << depth) * out_dim * 4 + depth * 9)
    alg = n_rows * (n_feat + out_dim) * 4 + model_bytes
    rt = n_rows * trees / seconds
    return {"hbm": {"bound": "hbm", "achieved": alg / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / seconds / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_call": alg},
            "issue": {"bound": "valu-issue", "achieved": rt, "peak": PREDICT_ISSUE_BOUND, "unit": "row-trees/s", "frac": rt / PREDICT_ISSUE_BOUND,
                      "note": "LDS-tile formulation (rounds 2-3): 30 VALU (4 clk) + 10 LDS (2 clk) instructions per (64 rows, depth-6 tree, 8 outputs); see bench.py / DESIGN.md section 5"},
            "issue_reg": {"bound": "valu-issue", "achieved": rt, "peak": PREDICT_ISSUE_BOUND_REG, "unit": "row-trees/s", "frac": rt / PREDICT_ISSUE_BOUND_REG,
                          "note": "register-tile formulation (round 4): 17 VALU + 4 LDS reads + 6 scalar M0 writes per (64 rows, depth-6 tree, 8 outputs)"}}


class _QuietStdout:
    """The reference's load() prints a banner with printf: keep it out of this script's one-line stdout."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)
        os.close(null)
    def __exit__(self, *a):
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_predict_baseline(np, model_path, n_feat, rows, budget_s=8.0):
    """The reference's own predict_cpu (predictor.cpp:122-265) on this host: the ensemble the bench grew, written by the product as a
    .gbrl_model file and loaded by oracle/_ref (the file format is the reference's), `rows` synthetic rows; repeated while the leg
    stays within `budget_s`."""
    import oracle
    mod = oracle.load_ref()
    if mod is None:
        return {"error": "oracle/_ref not built"}
    with _QuietStdout():
        ref = mod.GBRL.load(model_path)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((rows, n_feat)).astype(np.float32)
    ref.predict(X, None, 0, 0)
    t0 = time.perf_counter()
    reps = 0
    while True:
        ref.predict(X, None, 0, 0)
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    trees = int(ref.get_num_trees())
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"rows_per_s": rows / dt, "row_trees_per_s": rows * trees / dt, "trees": trees, "rows": rows, "seconds_per_call": dt, "calls": reps, "cores": cores,
            "kind": "reference", "what": "Predictor::predict_cpu of oracle/_ref on the model file the product saved"}


def leg_collective(torch, np, gbrl_amd, workload, X, G, F, D, depth, B, steps, plain_ms):
    """The row-sharded code path on ONE GPU over a world-size-1 native RCCL communicator (gbrl_amd.dist.install_rccl_single): every exchange of a
    sharded step -- row count, gradient statistics, the selection's digit counts, the per-level histogram reduce-scatter and winner
    all-reduce, the leaf sums -- is enqueued on the model's stream and executed by RCCL, but no byte crosses xGMI.  What it costs over the
    plain one-GPU step is the fixed price of the multi-GPU path; the 1/2/4/8 curve needs a multi-GPU node."""
    import ctypes
    from gbrl_amd.dist import install_rccl_single
    # RCCL prints a version banner on C stdout when its first communicator comes up: this process's stdout carries ONE JSON line, so file
    # descriptor 1 points at stderr while the leg runs (and C stdio is flushed before it is restored)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        m = make_model(gbrl_amd, np, workload, F, 0, D, depth, B, "bench_collective")
        install_rccl_single(m)
        tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
        xo, go = tup(X), tup(G)
        for _ in range(3):
            m.step(xo, None, go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.step(xo, None, go)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        m.set_profiling(2)
        m.step(xo, None, go)
        ph = dict(m.last_phase_times())
        del m
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)
    return {"phases_ms": {k: v for k, v in sorted(ph.items()) if not k.startswith("exchange_")},"world_size": 1, "transport": "rccl (native: collectives enqueued on the model's stream)", "steps": steps, "ms_per_step": ms,
            "plain_ms_per_step": plain_ms, "overhead_vs_plain": ms / plain_ms if plain_ms > 0 else None,
            "exchanges": ph.get("exchange_calls"), "payload_mb": ph.get("exchange_payload_mb"),
            "note": "one GPU, no xGMI traffic: the fixed cost of the row-sharded path; multi-GPU scaling is measured by the driver's --gpus N runs"}


def leg_predict_deep(torch, np, gbrl_amd, dev, X, D, B, depth=8, trees=1000, mini=4096):
    """predict() over 1000 oblivious trees of max_depth 8 on the bench's 2^20 x 128 rows (the reference has no depth limit,
    predictor.cpp:231-265; since round 5 the register-tile kernels take depth 7-8 too -- `VariantDeep`, one record buffer and two value sets per group
    of 8 trees -- and this leg exercises them).  The trees are grown on
    4096-row minibatches of the same matrix (the one-launch growth: a second per 1000 trees)."""
    N, F = X.shape
    m = make_model(gbrl_amd, np, "cfg2", F, 0, D, depth, B, "bench_deep")
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    W = torch.randn((8, D), device=dev, generator=gen)
    m.set_profiling(0)
    t0 = time.perf_counter()
    for i in range(trees):
        o = (i * mini) % (N - mini + 1)
        xs = X[o:o + mini]
        g = (torch.tanh(xs[:, :8] @ W) + 0.5 * torch.randn((mini, D), device=dev, generator=gen)).contiguous()
        m.step(tup(xs), None, tup(g))
    torch.cuda.synchronize()
    grow_s = time.perf_counter() - t0
    m.set_profiling(1)
    xo = tup(X)
    dt = time_predict(torch, m, xo, None, 5)
    T = m.get_num_trees()
    depths = np.asarray(m.get_ensemble_data()["depths"])
    return {"workload": "oblivious, max_depth %d, %d trees grown on %d-row minibatches; predict on %d x %d rows" % (depth, T, mini, N, F),
            "trees": T, "mean_depth": float(depths.mean()), "ms_per_call": dt * 1e3, "kernel_ms": m.last_phase_times().get("predict", 0.0),
            "rows_per_s": N / dt, "row_trees_per_s": N * T / dt, "grown": "%d steps in %.2f s (%.2f ms/step)" % (T, grow_s, grow_s * 1e3 / max(1, T)),
            "roofline_hbm_frac": (N * (F * 4 + D * 4)) / dt / 1e9 / HBM_PEAK_GBS}


def leg_cfg1(torch, np, gbrl_amd, dev, trees=30, N=4096, F=16, depth=4, B=256):
    """BASELINE configs[0], the reference's own CPU-runnable case (tests/test_gbt_single.py:46-61): single-output MultiRMSE loop --
    predict, gradient = prediction - target, step -- 4096 rows x 16 features, greedy / L2 / quantile, depth 4, 30 trees.  The product
    on the GPU (device tensors in, DLPack out) and the REFERENCE's CPU path (oracle/_ref) on this box's host cores at full size, same
    inputs; the final predictions are compared (1e-5 of the target's scale: the two grow the same trees unless a float32 near-tie
    intervenes, tests/neartie.py)."""
    import oracle
    rng = np.random.default_rng(21)
    X = rng.standard_normal((N, F)).astype(np.float32)
    x0 = np.clip(X[:, 0], -2, 2)
    y = (x0 - x0 ** 3 / 6.0 + 0.1 * rng.standard_normal(N)).astype(np.float32).reshape(N, 1)
    kw = dict(input_dim=F, output_dim=1, policy_dim=1, max_depth=depth, min_data_in_leaf=0, n_bins=B, par_th=10, cv_beta=0.9, split_score_func="L2",
              generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy="greedy", verbose=0, learner_name="bench_cfg1")

    def setup(m):
        m.set_feature_weights(np.ones(F, np.float32))
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=1)
        m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
        m.set_bias(np.array([float(y.mean())], np.float32))
    out = {"workload": "BASELINE configs[0]: single-output MultiRMSE loop, %d rows x %d features, greedy / L2 / quantile, depth %d, %d trees" % (N, F, depth, trees)}
    # product
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    best = None        # the loop as a user runs it: no synchronisation inside (step() returns with the tree in the model)
    best_split = None  # the same loop with a device synchronisation between its two halves, to time them separately
    for rep in range(4):   # (the first pass pays allocations and code loading)
        split = rep % 2 == 1
        m = gbrl_amd.GBRL(device="cuda", **kw)
        setup(m)
        torch.cuda.synchronize()
        t_pred = t_step = 0.0
        t0 = time.perf_counter()
        for _ in range(trees):
            ta = time.perf_counter()
            pred = torch.from_dlpack(m.predict(tup(Xd), None, 0, 0)).reshape(N, 1)
            g = (pred - yd).contiguous()
            if split:
                torch.cuda.synchronize()
            tb = time.perf_counter()
            m.step(tup(Xd), None, tup(g))
            tc = time.perf_counter()
            t_pred += tb - ta
            t_step += tc - tb
        torch.cuda.synchronize()
        loop = time.perf_counter() - t0
        if split and (best_split is None or loop < best_split[0]):
            best_split = (loop, t_pred, t_step)
        if not split and rep > 0 and (best is None or loop < best):
            best = loop
        final = torch.from_dlpack(m.predict(tup(Xd), None, 0, 0)).cpu().numpy().reshape(N, 1)
    # near-tie replay counters of the last model (cumulative since it was created; an extra, untimed step reads them)
    m.set_profiling(2)
    m.step(tup(Xd), None, tup(g))
    ph = dict(m.last_phase_times())
    out["near_tie_replay"] = {"levels_replayed_inside_the_growth_kernel": int(ph.get("near_in_kernel", 0)), "levels_replayed_by_the_level_loop": int(ph.get("near_replays", 0)),
                              "trees_handed_to_the_level_loop": int(ph.get("near_bailouts", 0)), "trees": trees + 1,
                              "what": "nodes whose runner-up is within 2^-20 of the exact best gain are re-scored in the reference's float32 sequence (DESIGN 3a); GBRL_HIP_NO_NEARTIE_REPLAY=1 switches it off"}
    out["product"] = {"ms_per_iteration": best / trees * 1e3, "trees_per_s": trees / best,
                      "ms_per_step": best_split[2] / trees * 1e3, "ms_per_predict_and_gradient": best_split[1] / trees * 1e3,
                      "ms_per_iteration_with_a_synchronisation_between_the_halves": best_split[0] / trees * 1e3,
                      "rmse": float(np.sqrt(np.mean((final - y) ** 2)))}
    # the reference's CPU path at full size on this host
    mod = oracle.load_ref()
    if mod is None:
        out["cpu_reference"] = {"error": "oracle/_ref not built"}
        return out
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cbest = None
    for rep in range(2):
        r = mod.GBRL(device="cpu", **kw)
        setup(r)
        t_step = 0.0
        t0 = time.perf_counter()
        for _ in range(trees):
            pred = np.asarray(r.predict(X, None, 0, 0)).reshape(N, 1)
            g = np.ascontiguousarray((pred - y).astype(np.float32))
            tb = time.perf_counter()
            r.step(X, None, g)
            t_step += time.perf_counter() - tb
        loop = time.perf_counter() - t0
        if cbest is None or loop < cbest[0]:
            cbest = (loop, t_step)
        rfinal = np.asarray(r.predict(X, None, 0, 0)).reshape(N, 1)
    out["cpu_reference"] = {"ms_per_iteration": cbest[0] / trees * 1e3, "ms_per_step": cbest[1] / trees * 1e3, "trees_per_s": trees / cbest[0], "cores": cores,
                            "kind": "reference", "rmse": float(np.sqrt(np.mean((rfinal - y) ** 2))), "note": "Fitter::step_cpu + Predictor::predict_cpu of oracle/_ref, full size, no extrapolation"}
    out["speedup_per_iteration"] = cbest[0] / best
    out["max_abs_prediction_difference"] = float(np.max(np.abs(final - rfinal)))
    return out


def make_model(gbrl_amd, np, kind, F, Fc, D, depth, B, name):
    """kind: cfg2 = oblivious / L2 / quantile, one optimiser; cfg3 = greedy / Cosine / quantile, policy + value optimisers;
    cfg5 = oblivious / L2 / uniform, numeric + categorical columns."""
    kw = dict(input_dim=F + Fc, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=B, par_th=10, cv_beta=0.9,
              use_control_variates=False, batch_size=5000, verbose=0, device="cuda", learner_name=name)
    if kind == "cfg3":
        kw.update(split_score_func="Cosine", generator_type="Quantile", grow_policy="greedy")
    elif kind == "cfg5":
        kw.update(split_score_func="L2", generator_type="Uniform", grow_policy="oblivious")
    else:
        kw.update(split_score_func="L2", generator_type="Quantile", grow_policy="oblivious")
    m = gbrl_amd.GBRL(**kw)
    m.set_feature_weights(np.ones(F + Fc, np.float32))
    if kind == "cfg3":
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D - 1)
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 1, stop_idx=D)
    else:
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F + Fc, dtype=np.int32), np.array([True] * F + [False] * Fc, dtype=bool))
    return m


def time_predict(torch, m, xo, co, reps):
    p = m.predict(xo, co, 0, 0)
    del p
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        p = m.predict(xo, co, 0, 0)
        del p
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def leg_cfg3(torch, np, gbrl_amd, dev, X, G, D, depth, B, steps=6, warmup=2):
    """BASELINE configs[2] at full size on one GPU: greedy / Cosine / quantile, policy + value optimisers (shared actor-critic)."""
    N, F = X.shape
    m = make_model(gbrl_amd, np, "cfg3", F, 0, D, depth, B, "bench_cfg3")
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    xo, go = tup(X), tup(G)
    m.set_profiling(0)
    for _ in range(warmup):
        m.step(xo, None, go)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.step(xo, None, go)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.set_profiling(2)
    acc = {}
    for _ in range(2):
        m.step(xo, None, go)
        for k, v in m.last_phase_times().items():
            acc[k] = acc.get(k, 0.0) + v / 2
    m.set_profiling(1)
    dtp = time_predict(torch, m, xo, None, 5)
    T = m.get_num_trees()
    e_leaves = int(np.asarray(m.get_ensemble_data()["values"]).shape[0])
    return {"workload": "BASELINE configs[2]: greedy tree, Cosine score, quantile candidates, policy [0,%d) lr 0.1 + value [%d,%d) lr 0.01" % (D - 1, D - 1, D),
            "rows": N, "n_features": F, "output_dim": D, "max_depth": depth, "steps": steps, "ms_per_step": dt * 1e3, "trees_per_s": 1.0 / dt,
            "leaves_per_tree": e_leaves / float(T), "phases_ms_per_step": {k: v for k, v in sorted(acc.items())},
            "predict": {"trees": T, "ms_per_call": dtp * 1e3, "rows_per_s": N / dtp, "row_trees_per_s": N * T / dtp,
                        "kernel_ms": m.last_phase_times().get("predict", 0.0), "roofline": predict_roofline(N, F, D, T, depth, dtp),
                        "kernel": "k_predict_grd_stream (round 6: the ensemble in LDS, one barrier-free row-tile pipeline per wave; HBM-bound -- the issue bounds "
                                  "above describe the oblivious kernels)"}}


def leg_fullsize_parity(torch, np, gbrl_amd, dev):
    """north_star's acceptance sentence, measured live: the product's tree at 2^20 x 128 against the tree the REFERENCE's own CPU path grew on
    the same inputs (committed fixtures tests/golden/full_cfg2.npz / full_cfg3.npz, written in the build container by oracle/_ref: 835 s and
    1230 s per tree on 8 vCPU; the 512 MiB of inputs are regenerated from the seed -- tests/golden/cases.py, exactly rounded arithmetic -- and
    checked through their SHA-256).  Two modes: the default (exact float64 arg-max above 65 536 rows) and GBRL_HIP_NEARTIE_MAX_ROWS=0 (every
    near-tie re-scored in the reference's float32 sequence), each with its step time (device-resident inputs)."""
    import json as _json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as K
    out = {}
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    saved = os.environ.get("GBRL_HIP_NEARTIE_MAX_ROWS")
    try:
        for name in ("full_cfg2", "full_cfg3"):
            path = os.path.join(ROOT, "tests", "golden", name + ".npz")
            if not os.path.exists(path):
                out[name] = {"error": "fixture missing"}
                continue
            fx = np.load(path)
            case = _json.loads(str(fx["case_json"]))
            Xh, _, Gh, _ = K.make_inputs(case)
            if K.inputs_digest(Xh, None, Gh, None) != str(fx["inputs_sha256"]):
                out[name] = {"error": "regenerated inputs differ from the fixture's"}
                continue
            X, G = torch.from_numpy(Xh).to(dev), torch.from_numpy(Gh).to(dev)
            rec = {"workload": "BASELINE configs[%d] at 2^20 x 128, depth 6, D = 8; reference tree from %d OpenMP threads (%.0f s)" % (1 if name == "full_cfg2" else 2, int(fx["omp_threads"]), float(fx["wall_s"]))}
            for mode, env in (("default", None), ("near_tie_replay_everywhere", "0")):
                if env is None:
                    os.environ.pop("GBRL_HIP_NEARTIE_MAX_ROWS", None)
                else:
                    os.environ["GBRL_HIP_NEARTIE_MAX_ROWS"] = env
                ms = []
                for rep in range(2):     # (first: allocations)
                    m = gbrl_amd.GBRL(**K.ctor_kwargs(case, device="cuda"))
                    m.set_feature_weights(np.ones(case["F"], np.float32))
                    for o in K.optimizers(case):
                        m.set_optimizer(**o)
                    m.set_feature_mapping(np.arange(case["F"], dtype=np.int32), np.ones(case["F"], dtype=bool))
                    if rep:
                        m.step(tup(X), None, tup(G)); torch.cuda.synchronize()      # warm
                    t0 = time.perf_counter()
                    m.step(tup(X), None, tup(G))
                    torch.cuda.synchronize()
                    ms.append((time.perf_counter() - t0) * 1e3)
                    if not rep:
                        e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items()}
                if case["policy"] == "oblivious":
                    d_ref, d_got = int(fx["depths"][0]), int(e["depths"][0])
                    same = sum(1 for l in range(d_ref) if l < d_got and e["feature_indices"][0][l] == fx["feature_indices"][0][l]
                               and bits(e["feature_values"][0][l:l + 1])[0] == bits(fx["feature_values"][0][l:l + 1])[0])
                    r = {"levels_identical": same, "levels": d_ref}
                    identical = same == d_ref == d_got
                else:
                    def nodes(a):
                        fi, fv, dr, dp = np.asarray(a["feature_indices"]), bits(np.asarray(a["feature_values"])).reshape(np.asarray(a["feature_values"]).shape), np.asarray(a["inequality_directions"]), np.asarray(a["depths"])
                        tab = {}
                        for leaf in range(len(dp)):
                            pre = ()
                            for k in range(int(dp[leaf])):
                                sp = (int(fi[leaf, k]), int(fv[leaf, k])); tab.setdefault(pre, sp); pre = pre + ((sp[0], sp[1], int(dr[leaf, k])),)
                        return tab
                    nr, ng = nodes({k: fx[k] for k in fx.files}), nodes(e)
                    same = sum(1 for pth, sp in nr.items() if ng.get(pth) == sp)
                    r = {"internal_nodes_identical": same, "internal_nodes": len(nr)}
                    identical = same == len(nr) == len(ng)
                if identical:
                    scale = float(np.abs(Gh).mean())
                    r["leaf_values_max_rel_err"] = float(np.max(np.abs(e["values"] - fx["values"]) / np.maximum(np.abs(fx["values"]), scale)))
                r["structure_identical"] = bool(identical)
                r["ms_per_step"] = ms[-1]
                rec[mode] = r
            out[name] = rec
            del X, G
            torch.cuda.empty_cache()
    finally:
        if saved is None:
            os.environ.pop("GBRL_HIP_NEARTIE_MAX_ROWS", None)
        else:
            os.environ["GBRL_HIP_NEARTIE_MAX_ROWS"] = saved
    return out


def leg_cfg5(torch, np, gbrl_amd, dev, D, depth, B, trees, N=1 << 20, F=192, Fc=64, mini=4096):
    """BASELINE configs[4]: 192 numeric + 64 categorical columns (32 ASCII tokens, S128 cells), uniform candidates, an ensemble of
    `trees` oblivious depth-6 trees grown on 4096-row minibatches, predict on 2^20 rows.  The categorical input of predict is the
    RAW cell matrix (2^20 x 64 x 128 bytes = 8 GiB, resident in HBM): every call hashes and dictionary-encodes it on the device
    before the traversal -- nothing is pre-encoded."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(55)
    X = torch.randn((N, F), device=dev, dtype=torch.float32, generator=gen)
    tok = torch.randint(0, 32, (N, Fc), device=dev, generator=gen, dtype=torch.int64)
    cells = torch.zeros((N, Fc, 128), device=dev, dtype=torch.uint8)      # "cXX" zero padded to 128 bytes, like numpy S128
    cells[:, :, 0] = ord("c")
    cells[:, :, 1] = (ord("0") + tok // 10).to(torch.uint8)
    cells[:, :, 2] = (ord("0") + tok % 10).to(torch.uint8)
    wgen = torch.Generator(device=dev)
    wgen.manual_seed(56)
    W = torch.randn((8, D), device=dev, dtype=torch.float32, generator=wgen)
    G = torch.tanh(X[:, :8] @ W) + 0.5 * torch.randn((N, D), device=dev, dtype=torch.float32, generator=gen)
    # categorical columns carry signal too: output j is shifted for four tokens of column j and two tokens of column j + D
    catsig = ((tok[:, :D] % 8) == 3).to(torch.float32) * 2.0 + ((tok[:, D:2 * D] % 16) == 5).to(torch.float32) * 3.0
    G = (G + catsig).contiguous()
    # Every fifth minibatch carries ONLY the categorical signal (plus noise): against the dense numeric signal a single token equality
    # never wins a level (round 2: 78 categorical conditions in 60 000, so the categorical traversal was hardly exercised); with these
    # batches about a fifth of the ensemble's conditions compare dictionary ids (reported as categorical_conditions / conditions).
    Gc = (catsig + 0.5 * torch.randn((N, D), device=dev, dtype=torch.float32, generator=gen)).contiguous()
    del catsig
    del tok
    m = make_model(gbrl_amd, np, "cfg5", F, Fc, D, depth, B, "bench_cfg5")
    tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    ctup = lambda t: (t.data_ptr(), (t.shape[0], t.shape[1]), "S128", "cuda")
    m.set_profiling(0)
    t0 = time.perf_counter()
    n_mb = N // mini
    for i in range(trees):
        o = (i % n_mb) * mini
        scale = 1.0 / (1.0 + 0.01 * i)                                    # later trees see smaller residual-like gradients
        gi = ((Gc if i % 5 == 4 else G)[o:o + mini] * scale).contiguous()
        m.step(tup(X[o:o + mini]), ctup(cells[o:o + mini]), tup(gi))
    torch.cuda.synchronize()
    grow_s = time.perf_counter() - t0
    e = m.get_ensemble_data()
    n_cat_conditions = int((np.asarray(e["is_numerics"]) == 0).sum())
    m.set_profiling(2)
    xo, co = tup(X), ctup(cells)
    dtp = time_predict(torch, m, xo, co, 2)
    ph = m.last_phase_times()
    T = m.get_num_trees()
    out = {"workload": "BASELINE configs[4]: %d numeric + %d categorical columns, uniform candidates, oblivious depth %d, predict over %d trees" % (F, Fc, depth, T),
           "rows": N, "trees": T, "ms_per_call": dtp * 1e3, "rows_per_s": N / dtp, "row_trees_per_s": N * T / dtp,
           "kernel_ms": ph.get("predict", 0.0), "encode_ms": ph.get("inputs", 0.0),
           "categorical_input": "raw S128 cells (%.1f GiB in HBM), hashed and dictionary-encoded on the device inside every call" % (N * Fc * 128 / 2.0**30),
           "categorical_conditions": n_cat_conditions, "conditions": int(np.asarray(e["depths"]).sum()),
           "grown": "%d steps on %d-row minibatches in %.1f s (%.2f ms/step)" % (trees, mini, grow_s, grow_s * 1e3 / trees),
           "roofline": predict_roofline(N, F + Fc, D, T, depth, dtp)}
    out["categorical_fraction"] = n_cat_conditions / max(1, out["conditions"])
    # The same batch with its cells encoded ONCE (SURVEY 8d allows it when stated; the product's extension encode_categorical /
    # predict_encoded, include/gbrl_hip.h): what a serving loop pays per call when it keeps the int32 ids of its batch.  Same bits.
    try:
        t1 = time.perf_counter()
        ids_cap, token = m.encode_categorical(co)
        ids = torch.from_dlpack(ids_cap)
        torch.cuda.synchronize()
        enc_once = time.perf_counter() - t1
        io = (ids.data_ptr(), tuple(ids.shape), "torch.int32", "cuda")
        ref = torch.from_dlpack(m.predict(xo, co, 0, 0))
        same = bool(torch.equal(ref, torch.from_dlpack(m.predict_encoded(xo, io, token, 0, 0))))
        del ref
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            p = m.predict_encoded(xo, io, token, 0, 0)
            del p
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t1) / 2
        out["pre_encoded"] = {"ms_per_call": dte * 1e3, "rows_per_s": N / dte, "row_trees_per_s": N * T / dte, "encode_once_ms": enc_once * 1e3,
                              "ids_bytes": int(ids.numel()) * 4, "bitwise_equal_to_the_raw_call": same,
                              "what": "categorical cells of the batch encoded once into int32 dictionary ids (encode_categorical), every call predicts from the ids "
                                      "(predict_encoded); an extension -- the reference and the numbers above take the raw cells in every call"}
        del ids, ids_cap
    except Exception as ex:      # (the leg's headline numbers above do not depend on the extension)
        out["pre_encoded"] = {"error": str(ex)[:200]}
    del X, cells, G, Gc
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU (default 2^20 = the metric's batch)")
    ap.add_argument("--features", type=int, default=128)
    ap.add_argument("--out-dim", type=int, default=8)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="0: grow the CPU sample while the leg stays within ~20 s on this host")
    ap.add_argument("--cpu-full-tree", action="store_true",
                    help="opt-in (minutes of CPU time): ALSO time ONE tree of the reference's CPU path at the full batch (2^20 x 128 on this host's cores, "
                         "SURVEY 8d) and report it as cpu_baseline.full_tree -- no extrapolation")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the row-sharded code path (collective hooks through RCCL) on ONE GPU")
    ap.add_argument("--exchange", choices=["rccl", "hooks"], default="rccl",
                    help="multi-GPU exchange: the model's own RCCL communicator (default) or torch.distributed hooks")
    ap.add_argument("--predict-trees", type=int, default=0, help="0: predict over the ensemble grown by the bench")
    ap.add_argument("--workload", choices=["cfg2", "cfg3"], default="cfg2",
                    help="timed workload: cfg2 = BASELINE configs[1] (oblivious / L2 / quantile, the metric's configuration); "
                         "cfg3 = BASELINE configs[2] (greedy / Cosine / policy [0,D-1) lr 0.1 + value [D-1,D) lr 0.01)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the cfg3 and predict_cfg5 legs that follow the timed region (single GPU)")
    ap.add_argument("--cfg5-trees", type=int, default=10000, help="ensemble size of the predict_cfg5 leg (BASELINE configs[4]: 10 000)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="test hook (one process): the rows of K ranks (their seeds, concatenated in rank order) as ONE unsharded batch -- the "
                         "single-process twin of `--gpus K` for tree-equality checks")
    ap.add_argument("--dump-ensemble", default=None, help="test hook: rank 0 writes the grown ensemble's structure arrays to this .npz")
    ap.add_argument("--large-ensemble", type=int, default=1000,
                    help="also time predict() over an ensemble of this many trees (the row-trees/s rate is flat from ~100 trees on; "
                         "BASELINE configs[4] has 10000).  The extra trees are grown after the timed region with the same "
                         "full-size step(), so every k_hist_build launch of the run has the benchmark's shape; 0 disables")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process becomes the launcher (it never touches the GPU) and starts one
    # rank per GPU as child processes; rank 0's JSON line passes through on stdout.  Under torch.distributed.run (the driver's
    # N>1 command) WORLD_SIZE is already set and this is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("BENCH_LAUNCH_ONLY") == "1":   # test hook (CPU): show what the launcher handed to this rank, run nothing
        print(json.dumps({"rank": rank, "local_rank": local_rank, "n_gpus": world, "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}), flush=True)
        fail = os.environ.get("BENCH_LAUNCH_FAIL_RANK")   # test hook: this rank dies, the others "hang in a collective"
        if fail is not None:
            if int(fail) == rank:
                raise SystemExit(3)
            time.sleep(600)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import gbrl_amd

    if not torch.cuda.is_available() or not gbrl_amd.cuda_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # BENCH_SHARE_DEVICE=1 (test hook): every rank uses cuda:0 and the ranks talk through gloo (reductions staged through the
    # host by gbrl_amd.dist) -- lets the multi-rank code path of this script run on a single-GPU box.  Numbers are meaningless.
    share = os.environ.get("BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.force_collective:
        os.environ["GBRL_HIP_FORCE_COLLECTIVE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    N, F, D, depth, B = args.rows, args.features, args.out_dim, args.depth, args.bins
    wgen = torch.Generator(device=dev)
    wgen.manual_seed(99)
    W = torch.randn((8, D), device=dev, dtype=torch.float32, generator=wgen)

    def shard(r):   # the rows of rank r: its own generator, so that K ranks and one process emulating K ranks see the same global batch
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + r)
        Xr = torch.randn((N, F), device=dev, dtype=torch.float32, generator=gen)
        Gr = torch.tanh(Xr[:, :8] @ W) + 0.5 * torch.randn((N, D), device=dev, dtype=torch.float32, generator=gen)
        return Xr, Gr
    if args.emulate_ranks > 1 and world == 1:
        parts = [shard(r) for r in range(args.emulate_ranks)]
        X = torch.cat([p[0] for p in parts]).contiguous()
        G = torch.cat([p[1] for p in parts]).contiguous()
        del parts
        N = X.shape[0]
    else:
        X, G = shard(rank)
        G = G.contiguous()

    m = make_model(gbrl_amd, np, args.workload, F, 0, D, depth, B, "bench")
    # level 1: a HIP event pair on one k_hist_build launch in seven (every tree level in turn), attached to the dispatch itself (hipExtLaunchKernelGGL start/stop
    # events on the engine's own stream: the kernel's begin/end timestamps, no extra packet in the stream), resolved after each
    # call -- no sync inside the step.  The full phase table needs ~60 hipEventRecord calls per step, each a few-microsecond
    # stream bubble, so it is taken in a separate diagnostic pass after the timed region.
    m.set_profiling(int(os.environ.get("BENCH_TIMED_PROFILING", "1")))   # (measurement hook: 0 = no events at all, 2 = every phase inside the timed region)
    coll, exchange = None, None
    if world > 1 or args.force_collective:
        from gbrl_amd.dist import install_rccl, install_torch_collective
        if args.exchange == "rccl" and not share:
            try:   # the model's own RCCL communicator: all-reduces enqueued on its stream, no host synchronisation
                install_rccl(m, dev)
                exchange = "rccl (native, stream-ordered)"
            except Exception as e:
                print("bench: native RCCL exchange unavailable (%r); falling back to torch.distributed hooks" % (e,), file=sys.stderr, flush=True)
        if exchange is None:
            coll = install_torch_collective(m, dev)
            exchange = "torch.distributed hooks (host-synchronous)"

    def tup(t):
        return (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")

    xo, go = tup(X), tup(G)
    for _ in range(args.warmup):
        m.step(xo, None, go)
    phase_acc = {}

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.step(xo, None, go)
        for k, v in m.last_phase_times().items():
            phase_acc[k] = phase_acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if args.dump_ensemble and rank == 0:   # test hook: the trees of the warm-up + timed steps (every rank grows the same ones)
        e = m.get_ensemble_data()
        np.savez(args.dump_ensemble, **{k: np.asarray(e[k]) for k in ("tree_indices", "depths", "feature_indices", "feature_values",
                                                                    "inequality_directions", "edge_weights", "values")})

    # diagnostic pass (untimed): every phase bracketed with events
    m.set_profiling(2)
    diag_steps = max(1, min(3, args.steps))
    diag_acc = {}
    for _ in range(diag_steps):
        m.step(xo, None, go)
        for k, v in m.last_phase_times().items():
            diag_acc[k] = diag_acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    m.set_profiling(1)

    # predict over the ensemble (rows stay sharded; no exchange)
    n_trees = m.get_num_trees()
    torch.cuda.synchronize()
    reps = 20   # (5 calls read 6-9 % high: the first calls after the step loop run slower, scripts/predict_overhead_probe.py)
    m.predict(xo, None, 0, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        p = m.predict(xo, None, 0, 0)
        del p
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t1) / reps
    pk = m.last_phase_times().get("predict", 0.0)
    model_files = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import tempfile
        tmpdir = tempfile.mkdtemp(prefix="gbrl_bench_")
        model_files["small"] = os.path.join(tmpdir, "small.gbrl_model")
        m.save(model_files["small"])

    # predict over a large ensemble: extra trees grown with the same full-size step(), outside every timed region above
    large = None
    if args.large_ensemble > n_trees and world == 1:   # single-GPU leg (growing 10^4 trees through the collective path is slow)
        m.set_profiling(0)
        t2 = time.perf_counter()
        for _ in range(args.large_ensemble - n_trees):
            m.step(xo, None, go)
        torch.cuda.synchronize()
        grow_s = time.perf_counter() - t2
        m.set_profiling(1)
        p = m.predict(xo, None, 0, 0)
        del p
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(2):
            p = m.predict(xo, None, 0, 0)
            del p
        torch.cuda.synchronize()
        dtl = (time.perf_counter() - t3) / 2
        T2 = m.get_num_trees()
        large = {"trees": T2, "ms_per_call": dtl * 1e3, "rows_per_s": world * N / dtl, "row_trees_per_s": world * N * T2 / dtl,
                 "kernel_ms": m.last_phase_times().get("predict", 0.0), "roofline": predict_roofline(N, F, D, T2, depth, dtl),
                 "grown": "%d extra full-size steps in %.1f s (%.2f ms/step, no profiling events)" % (T2 - n_trees, grow_s, grow_s * 1e3 / max(1, T2 - n_trees))}
        # the same ensemble asked for an RL-sized batch (an agent's minibatch): the latency path (exact chain in two launches up to 8192 rows,
        # the reference's thread slices below 2 * par_th rows -- DESIGN.md section 5, profiles/r02_predict_latency.txt)
        small = {}
        for rows in (16, 1024, 4096):
            xs_ = X[:rows]
            xr = (xs_.data_ptr(), tuple(xs_.shape), str(xs_.dtype), "cuda")
            dts = time_predict(torch, m, xr, None, 20)
            small[str(rows)] = {"ms_per_call": dts * 1e3, "kernel_ms": m.last_phase_times().get("predict", 0.0), "row_trees_per_s": rows * T2 / dts}
        large["small_batches"] = small
        if model_files:
            model_files["large"] = os.path.join(os.path.dirname(model_files["small"]), "large.gbrl_model")
            m.save(model_files["large"])

    extra = {}
    if world == 1 and not args.no_extra_legs and not args.force_collective:
        try:   # (a leg is reporting only; never let it hide the timed measurement)
            extra["collective"] = leg_collective(torch, np, gbrl_amd, args.workload, X, G, F, D, depth, B, args.steps, dt / args.steps * 1e3)
        except Exception as e:
            extra["collective"] = {"error": repr(e)}
        del m
        torch.cuda.empty_cache()
        if args.workload != "cfg3":
            try:
                extra["cfg3"] = leg_cfg3(torch, np, gbrl_amd, dev, X, G, D, depth, B)
            except Exception as e:   # a leg is reporting only; never let it hide the timed measurement
                extra["cfg3"] = {"error": repr(e)}
        try:
            extra["predict_depth8"] = leg_predict_deep(torch, np, gbrl_amd, dev, X, D, B)
        except Exception as e:
            extra["predict_depth8"] = {"error": repr(e)}
        del X, G
        torch.cuda.empty_cache()
        try:
            extra["predict_cfg5"] = leg_cfg5(torch, np, gbrl_amd, dev, D, depth, B, args.cfg5_trees)
        except Exception as e:
            extra["predict_cfg5"] = {"error": repr(e)}
        try:
            extra["cfg1"] = leg_cfg1(torch, np, gbrl_amd, dev)
        except Exception as e:
            extra["cfg1"] = {"error": repr(e)}
        try:
            extra["full_size_parity"] = leg_fullsize_parity(torch, np, gbrl_amd, dev)
        except Exception as e:
            extra["full_size_parity"] = {"error": repr(e)}

    if rank == 0:
        steps = args.steps
        ms_per_step = dt / steps * 1e3
        # live, inside the timed region: at profiling level 1 the engine attaches the event pair to one k_hist_build launch in seven (every
        # tree level in turn; a pair on every launch cost 1.8 % of the step), so the per-tree time is depth x the mean sampled launch
        n_samp = phase_acc.get("hist_build_sampled_launches", 0.0)
        if n_samp > 0:
            build_ms = depth * phase_acc.get("hist_build", 0.0) / n_samp
        else:   # (fewer timed launches than one sampling period, or every launch timed: BENCH_TIMED_PROFILING=2)
            build_ms = (phase_acc.get("hist_build", 0.0) / steps) if phase_acc.get("hist_build", 0.0) > 0 else diag_acc.get("hist_build", 0.0) / diag_steps
        hist_ms = build_ms + diag_acc.get("hist_reduce", 0.0) / diag_steps
        alg = hist_algorithmic_bytes(N, F, D, depth, B)
        # dominant kernel = k_hist_build: `depth` launches per tree; per-launch figures are the per-tree ones / depth
        achieved = alg / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0
        traffic = None
        traffic_src = None
        traffic_levels = None
        tpath = os.path.join(ROOT, "profiles", "hist_traffic.json")
        if os.path.exists(tpath):   # HBM bytes per launch from the committed rocprofv3 PMC passes (scripts/pmc_summary.py)
            try:
                tj = json.load(open(tpath))
                traffic = tj["bytes_per_launch"]
                traffic_levels = tj.get("levels")
                traffic_src = "profiles/hist_traffic.json = the builder's rocprofv3 PMC passes (%s; %s, taken %s)" % (tj.get("method", "2*FETCH_SIZE + WRITE_SIZE"), tj.get("commit", "?"), tj.get("taken", "?"))
            except Exception:
                traffic = None
        try:
            build_info = json.load(open(os.path.join(ROOT, "gbrl_amd", "build_info.json")))
        except Exception:
            build_info = None
        out = {
            "build": build_info,
            "metric": "trees-fit/sec + predict rows/sec at batch=2^20, feat=128, depth=6, out=8",
            "value": steps * world * (N / float(1 << 20)) / dt,
            "unit": "trees/s (2^20-row batches fitted per second; one tree per batch per step)",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 fixed-point sums / f32 scores",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: oblivious tree, L2 score, quantile candidates" if args.workload == "cfg2" else
                                    "BASELINE configs[2]: greedy tree, Cosine score, quantile candidates, policy + value optimisers"), "rows_per_gpu": N,
                       "n_features": F, "output_dim": D, "max_depth": depth, "n_bins": B, "sharding": "rows x%d" % world + (" (collective path forced)" if args.force_collective else ""), "exchange": exchange},
            "predict": {"rows_per_s": world * N / dtp, "trees": n_trees, "ms_per_call": dtp * 1e3, "kernel_ms": pk,
                        "row_trees_per_s": world * N * n_trees / dtp, "roofline": predict_roofline(N, F, D, n_trees, depth, dtp)},
            "predict_large_ensemble": large,
            "cfg3": extra.get("cfg3"), "predict_cfg5": extra.get("predict_cfg5"), "predict_depth8": extra.get("predict_depth8"), "cfg1": extra.get("cfg1"),
            "full_size_parity": extra.get("full_size_parity"),
            "phases_ms_per_step": {k: v / diag_steps for k, v in sorted(diag_acc.items())},
            "collective": (extra.get("collective") if coll is None else
                           {"calls_per_step": coll.calls / float(args.warmup + steps + diag_steps), "bytes_per_step": coll.bytes / float(args.warmup + steps + diag_steps)}),
            "phases_note": "diagnostic pass of %d extra steps after the timed region (events around every phase)" % diag_steps,
            "roofline": {"bound": "hbm", "kernel": "k_hist_build (split-score histogram build), %d launches per tree" % depth,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_levels": traffic_levels,   # per tree level: HBM-side bytes, FETCH_SIZE factor, duration under --pmc (profiles/r06_hist_levels_traffic.txt)
                         "traffic_measured_by": ("builder, not by this run: committed PMC summary profiles/hist_traffic.json" if traffic is not None else None),
                         "algorithmic_bytes_per_launch": alg / depth,
                         "avg_launch_us": build_ms * 1e3 / depth, "launches_per_tree": depth,
                         "launches_timed": (("%d of the %d launches of the timed region (one in seven carries the HIP event pair, every tree level in turn)"
                                             % (int(n_samp), steps * depth)) if n_samp > 0 else "every launch"),
                         "algorithmic_bytes_per_tree": alg, "hist_build_ms_per_tree": build_ms,
                         "hist_build_plus_reduce_ms_per_tree": hist_ms,
                         "frac_including_reduce": (alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hist_ms > 0 else 0.0,
                         "note": "LDS-atomic-issue bound in practice: 9 ds_add_u32 per (row, feature), 8 at the root; see DESIGN.md section 5"},
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(F, D, depth, B, 1 << 20, args.cpu_sample_rows)
                out["cpu_baseline"].update(committed_full_tree())
            except Exception as e:  # the baseline is reporting only; never let it hide the measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
            if args.cpu_full_tree:
                try:
                    out["cpu_baseline"]["full_tree"] = cpu_full_tree(gbrl_amd)
                except Exception as e:
                    out["cpu_baseline"]["full_tree"] = {"error": repr(e)}
            try:   # the reference's predict_cpu on the ensembles the bench grew (rows/s on this host's cores)
                import numpy as _np
                if "small" in model_files:
                    out["cpu_baseline"]["predict"] = cpu_predict_baseline(_np, model_files["small"], F, 1 << 18)
                if "large" in model_files:
                    out["cpu_baseline"]["predict_large_ensemble"] = cpu_predict_baseline(_np, model_files["large"], F, 1 << 16)
            except Exception as e:
                out["cpu_baseline"]["predict"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
