"""Worker of tests/test_gpu_sharded_world2.py: one rank of a world_size-N row-sharded run on ONE GPU (gloo process group,
reductions staged through the host by gbrl_amd.dist).  argv: rank world port case_name out_npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def main():
    rank, world, port, name, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist
    import gbrl_amd
    from gbrl_amd.dist import install_torch_collective
    import cases as K
    from helpers import load_golden
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case, g, (X, Xc, G, y) = load_golden(name)
    N = case["N"]
    # deliberately uneven contiguous shards
    cuts = [0] + [int(N * (r + 1) / world * 0.8) for r in range(world - 1)] + [N]
    lo, hi = cuts[rank], cuts[rank + 1]
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    coll = install_torch_collective(m, torch.device("cuda:0"))
    Gs = None if G is None else G[lo:hi]
    ys = None if y is None else y[lo:hi]
    Xs = None if X is None else np.ascontiguousarray(X[lo:hi])
    Xcs = None if Xc is None else np.ascontiguousarray(Xc[lo:hi])
    pred = np.asarray(K.drive(m, case, Xs, Xcs, Gs, ys))
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    np.savez(out, lo=lo, hi=hi, pred=pred, calls=coll.calls, **e)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
