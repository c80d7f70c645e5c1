"""RL-sized step() timings with the one-launch growth (default) and with the level loop (GBRL_HIP_NO_SMALL_GROW=1), same process, same data:
    python3 scripts/small_grow_ab.py [reps]
Shapes: BASELINE configs[0] (4096 x 16, greedy / L2 / quantile, depth 4, 1 output) and configs[4]'s minibatch (4096 rows, 192 numeric + 64
categorical columns, uniform candidates, oblivious depth 6, 8 outputs), plus a few RL shapes in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
ctup = lambda t: (t.data_ptr(), (t.shape[0], t.shape[1]), "S128", "cuda")


def timed(fn, n):
    fn(20)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / n)
    return best


def numeric(N, F, D, depth, policy, score, gen_type):
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((N, F), device=dev, generator=g); G = torch.randn((N, D), device=dev, generator=g)
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func=score, generator_type=gen_type, use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda", learner_name="small")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    def run(k):
        for _ in range(k): m.step(tup(X), None, tup(G))
    return run


def cfg5():
    N, F, Fc, D, mini = 1 << 16, 192, 64, 8, 4096
    gen = torch.Generator(device=dev); gen.manual_seed(55)
    X = torch.randn((N, F), device=dev, generator=gen)
    tok = torch.randint(0, 32, (N, Fc), device=dev, generator=gen, dtype=torch.int64)
    cells = torch.zeros((N, Fc, 128), device=dev, dtype=torch.uint8)
    cells[:, :, 0] = ord("c"); cells[:, :, 1] = (ord("0") + tok // 10).to(torch.uint8); cells[:, :, 2] = (ord("0") + tok % 10).to(torch.uint8)
    G = (torch.randn((N, D), device=dev, generator=gen) + ((tok[:, :D] % 8) == 3).float() * 2.0).contiguous()
    m = bench.make_model(gbrl_amd, np, "cfg5", F, Fc, D, 6, 256, "cfg5_ab")
    n_mb = N // mini
    state = {"i": 0}
    def run(k):
        for _ in range(k):
            o = (state["i"] % n_mb) * mini; state["i"] += 1
            m.step(tup(X[o:o + mini]), ctup(cells[o:o + mini]), tup(G[o:o + mini]))
    return run


shapes = [("configs[0] 4096x16 D1 greedy/L2/quantile depth 4", lambda: numeric(4096, 16, 1, 4, "greedy", "L2", "Quantile")),
          ("2048x24 D6 greedy/cosine/quantile depth 4", lambda: numeric(2048, 24, 6, 4, "greedy", "Cosine", "Quantile")),
          ("4096x192 D8 oblivious/L2/uniform depth 6", lambda: numeric(4096, 192, 8, 6, "oblivious", "L2", "Uniform")),
          ("8192x64 D8 oblivious/L2/quantile depth 6", lambda: numeric(8192, 64, 8, 6, "oblivious", "L2", "Quantile")),
          ("512x8 D2 greedy/cosine/quantile depth 4", lambda: numeric(512, 8, 2, 4, "greedy", "Cosine", "Quantile")),
          ("configs[4] minibatch 4096x(192+64) D8 oblivious/L2/uniform depth 6", cfg5)]
for name, mk in shapes:
    out = []
    for hook in ("0", "1"):
        os.environ["GBRL_HIP_NO_SMALL_GROW"] = hook
        out.append(timed(mk(), reps))
    print("%-72s one launch %.4f ms   level loop %.4f ms" % (name, out[0], out[1]), flush=True)
os.environ.pop("GBRL_HIP_NO_SMALL_GROW", None)
