#!/usr/bin/env python3
"""bench.py -- trees-fit/sec (+ predict rows/sec) of the MI355X-native GBRL hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): oblivious tree, L2 split score, quantile
candidates, batch = 2^20 rows PER GPU, 128 features, depth 6, output_dim 8, n_bins 256, one SGD optimiser; synthetic
inputs X ~ N(0,1), G = tanh(X[:, :8] W) + 0.5 N(0,1) generated on the device before the timed region (inputs resident in
HBM).  One "step" = one GBRL.step() = one tree fitted on the whole batch.  With N > 1 the rows are sharded over the ranks
(weak scaling: 2^20 rows per GPU, the SAME tree is grown on every rank from all-reduced integer histograms), and `value`
counts 2^20-row batches: value = steps * n_gpus / seconds.

Printed JSON (rank 0, one line): the driver contract + "roofline" for the dominant kernel (split-score histogram build,
HBM-bound, algorithmic bytes from SURVEY.md 8(d)) + "cpu_baseline" (the reference's own CPU path if oracle/_ref is
loadable, else this repo's restatement) on a bounded sample + predict throughput.
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
    return {"value": (1.0 / dt) * sample_rows / full_rows, "unit": "trees/s at batch=2^20 (extrapolated linearly in rows)",
            "cores": cores, "kind": kind,
            "sample": "1 tree on %d of %d rows (same F=%d, D=%d, depth=%d, n_bins=%d): %.2f s; predict 1 tree %.3f s" % (
                sample_rows, full_rows, n_feat, out_dim, depth, n_bins, dt, dtp),
            "sample_seconds": dt}


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
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


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
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the row-sharded code path (collective hooks through RCCL) on ONE GPU")
    ap.add_argument("--exchange", choices=["rccl", "hooks"], default="rccl",
                    help="multi-GPU exchange: the model's own RCCL communicator (default) or torch.distributed hooks")
    ap.add_argument("--predict-trees", type=int, default=0, help="0: predict over the ensemble grown by the bench")
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
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    X = torch.randn((N, F), device=dev, dtype=torch.float32, generator=gen)
    wgen = torch.Generator(device=dev)
    wgen.manual_seed(99)
    W = torch.randn((8, D), device=dev, dtype=torch.float32, generator=wgen)
    G = (torch.tanh(X[:, :8] @ W) + 0.5 * torch.randn((N, D), device=dev, dtype=torch.float32, generator=gen)).contiguous()

    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=B, par_th=10,
                      cv_beta=0.9, split_score_func="L2", generator_type="Quantile", use_control_variates=False,
                      batch_size=5000, grow_policy="oblivious", verbose=0, device="cuda", learner_name="bench")
    m.set_feature_weights(np.ones(F, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    # level 1: one HIP event pair per k_hist_build launch, attached to the dispatch itself (hipExtLaunchKernelGGL start/stop
    # events on the engine's own stream: the kernel's begin/end timestamps, no extra packet in the stream), resolved after each
    # call -- no sync inside the step.  The full phase table needs ~60 hipEventRecord calls per step, each a few-microsecond
    # stream bubble, so it is taken in a separate diagnostic pass after the timed region.
    m.set_profiling(1)
    coll, exchange = None, None
    if world > 1 or args.force_collective:
        from gbrl_amd.dist import install_rccl, install_torch_collective
        if args.exchange == "rccl" and not share:
            try:   # the model's own RCCL communicator: all-reduces enqueued on its stream, no host synchronisation
                install_rccl(m, dev)
                exchange = "rccl (native, stream-ordered)"
            except Exception as e:
                print("bench: native RCCL exchange unavailable (%r); falling back to torch.distributed hooks" % (e,), flush=True)
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
    reps = 5
    m.predict(xo, None, 0, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        p = m.predict(xo, None, 0, 0)
        del p
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t1) / reps
    pk = m.last_phase_times().get("predict", 0.0)

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
                 "kernel_ms": m.last_phase_times().get("predict", 0.0),
                 "grown": "%d extra full-size steps in %.1f s (%.2f ms/step, no profiling events)" % (T2 - n_trees, grow_s, grow_s * 1e3 / max(1, T2 - n_trees))}

    if rank == 0:
        steps = args.steps
        ms_per_step = dt / steps * 1e3
        build_ms = phase_acc.get("hist_build", 0.0) / steps            # live, inside the timed region
        hist_ms = build_ms + diag_acc.get("hist_reduce", 0.0) / diag_steps
        alg = hist_algorithmic_bytes(N, F, D, depth, B)
        # dominant kernel = k_hist_build: `depth` launches per tree; per-launch figures are the per-tree ones / depth
        achieved = alg / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hist_traffic.json")
        if os.path.exists(tpath):   # HBM bytes per launch from the committed rocprofv3 PMC passes (scripts/pmc_summary.py)
            try:
                traffic = json.load(open(tpath))["bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "trees-fit/sec + predict rows/sec at batch=2^20, feat=128, depth=6, out=8",
            "value": steps * world * (N / float(1 << 20)) / dt,
            "unit": "trees/s (2^20-row batches fitted per second; one tree per batch per step)",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 fixed-point sums / f32 scores",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: oblivious tree, L2 score, quantile candidates", "rows_per_gpu": N,
                       "n_features": F, "output_dim": D, "max_depth": depth, "n_bins": B, "sharding": "rows x%d" % world + (" (collective path forced)" if args.force_collective else ""), "exchange": exchange},
            "predict": {"rows_per_s": world * N / dtp, "trees": n_trees, "ms_per_call": dtp * 1e3, "kernel_ms": pk,
                        "row_trees_per_s": world * N * n_trees / dtp},
            "predict_large_ensemble": large,
            "phases_ms_per_step": {k: v / diag_steps for k, v in sorted(diag_acc.items())},
            "collective": ({"calls_per_step": coll.calls / float(args.warmup + steps + diag_steps), "bytes_per_step": coll.bytes / float(args.warmup + steps + diag_steps)} if coll is not None else None),
            "phases_note": "diagnostic pass of %d extra steps after the timed region (events around every phase)" % diag_steps,
            "roofline": {"bound": "hbm", "kernel": "k_hist_build (split-score histogram build), %d launches per tree" % depth,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg / depth,
                         "avg_launch_us": build_ms * 1e3 / depth, "launches_per_tree": depth,
                         "algorithmic_bytes_per_tree": alg, "hist_build_ms_per_tree": build_ms,
                         "hist_build_plus_reduce_ms_per_tree": hist_ms,
                         "frac_including_reduce": (alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hist_ms > 0 else 0.0,
                         "note": "LDS-atomic-issue bound in practice: 9 ds_add_u32 per (row, feature); see DESIGN.md section 5"},
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(F, D, depth, B, 1 << 20, args.cpu_sample_rows)
            except Exception as e:  # the baseline is reporting only; never let it hide the measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
