"""Randomised ROW-SHARDED sweep on one GPU: `world` processes (gloo group, reductions staged through the host, like
tests/test_gpu_sharded_world2.py) grow random cases on uneven row shards; every rank must produce the tree -- and its shard of the
predictions -- that ONE process grows from all the rows, bit for bit.  (Row-sharded runs do not replay near-ties, DESIGN section 3a: the
one-process side runs with GBRL_HIP_NO_NEARTIE_REPLAY=1, i.e. both sides take the exact arg-max.)
    python scripts/sharded_sweep.py [n_cases] [first_seed] [world]      (world 1: the sharded path with the native RCCL exchange)"""
import json, os, socket, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K


def make_cases(n_cases, seed0):
    rng = np.random.default_rng(seed0)
    out = []
    for i in range(n_cases):
        Fc = int(rng.choice([0, 0, 1, 3, 6]))
        c = dict(name="sh%d" % i, seed=seed0 + i, N=int(rng.choice([600, 2500, 9000, 40000])), F=int(rng.choice([0 if Fc else 1, 3, 17, 40, 130])),
                 Fc=Fc, D=int(rng.choice([1, 3, 8, 11, 18])), depth=int(rng.choice([1, 3, 4, 5])), n_bins=int(rng.choice([7, 32, 256, 400])),
                 score=str(rng.choice(["L2", "Cosine"])), gen=str(rng.choice(["Quantile", "Uniform"])), policy=str(rng.choice(["greedy", "oblivious"])),
                 trees=int(rng.choice([1, 2, 3])), min_data_in_leaf=int(rng.choice([0, 0, 5])), n_tokens=int(rng.choice([4, 8, 20])))
        if c["N"] < c["n_bins"] + 1: c["n_bins"] = 32
        if c["n_bins"] >= 300: c["D"] = min(c["D"], 11)
        if c["F"] + c["Fc"] == 0: c["F"] = 2
        # how the level histograms travel (GBRL_HIP_HIST_ALLREDUCE_MAX_KB, the same on every rank): whole-level all-reduce everywhere
        # (fixture-sized levels are all below the default 10 MB), feature reduce-scatter + winner exchange everywhere, or a switch inside the tree
        node_kb = (c["F"] + c["Fc"]) * (c["n_bins"] + 1) * (c["D"] + 1) * 8 / 1024.0
        c["hist_kb"] = [None, "0", str(int(2.5 * node_kb) + 1)][i % 3]
        out.append(c)
    return out


def grow(case, lo, hi, install=None):
    import gbrl_amd
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    keep = install(m) if install else None
    sl = slice(lo, hi)
    pred = np.asarray(K.drive(m, case, None if X is None else np.ascontiguousarray(X[sl]), None if Xc is None else np.ascontiguousarray(Xc[sl]),
                              np.ascontiguousarray(G[sl]), None))
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    del keep
    return e, pred


def cuts_of(N, world):
    return [0] + [int(N * (r + 1) / world * 0.8) for r in range(world - 1)] + [N]


def worker(rank, world, port, cases_json, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
    if world == 1: os.environ["GBRL_HIP_FORCE_COLLECTIVE"] = "1"      # the sharded code path with the NATIVE exchange (own RCCL communicator)
    import torch, torch.distributed as dist
    from gbrl_amd.dist import install_torch_collective, install_rccl
    dev = torch.device("cuda:0")
    if world == 1:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        install = lambda m: install_rccl(m, dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        install = lambda m: install_torch_collective(m, dev)
    for case in json.load(open(cases_json)):
        cuts = cuts_of(case["N"], world)
        os.environ.pop("GBRL_HIP_HIST_ALLREDUCE_MAX_KB", None)
        if case.get("hist_kb") is not None: os.environ["GBRL_HIP_HIST_ALLREDUCE_MAX_KB"] = case["hist_kb"]      # (read at every call)
        try:
            e, pred = grow(case, cuts[rank], cuts[rank + 1], install)
            np.savez(os.path.join(outdir, "%s_r%d.npz" % (case["name"], rank)), pred=pred, **e)
        except RuntimeError as ex:
            open(os.path.join(outdir, "%s_r%d.err" % (case["name"], rank)), "w").write(str(ex))
        dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]); return
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cases = make_cases(n_cases, seed0)
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        json.dump(cases, open(os.path.join(d, "cases.json"), "w"))
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), port, os.path.join(d, "cases.json"), d],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        single = {}
        for c in cases:      # the single-process reference trees, grown here while the workers run
            os.environ["GBRL_HIP_NO_NEARTIE_REPLAY"] = "1"      # (read per call; the sharded side never replays)
            try: single[c["name"]] = grow(c, 0, c["N"])
            except RuntimeError as ex: single[c["name"]] = str(ex)
        logs = [p.communicate(timeout=3000)[0].decode(errors="replace") for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
        same = diff = unsupported = 0
        for c in cases:
            ref = single[c["name"]]
            errs = [os.path.exists(os.path.join(d, "%s_r%d.err" % (c["name"], r))) for r in range(world)]
            if isinstance(ref, str) or any(errs):
                msg = ref if isinstance(ref, str) else open(os.path.join(d, "%s_r%d.err" % (c["name"], errs.index(True)))).read()
                print("UNSUPPORTED", c, msg[:160], flush=True); unsupported += 1; continue
            e, pred = ref
            cuts = cuts_of(c["N"], world)
            ok = True
            for r in range(world):
                z = np.load(os.path.join(d, "%s_r%d.npz" % (c["name"], r)))
                ok &= all(np.array_equal(e[k], z[k]) for k in K.ENSEMBLE_KEYS) and np.array_equal(pred[cuts[r]:cuts[r + 1]], z["pred"])
            if ok: same += 1
            else: diff += 1; print("DIFF", c, flush=True)
    print("sharded world %d: %d cases, identical %d, different %d, unsupported %d  (%.1f s)" % (world, n_cases, same, diff, unsupported, time.time() - t0))
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    main()
