"""How often the near-tie replay fires on RL-sized steps and what it costs:  python3 scripts/neartie_rate.py [N] [F] [D] [depth] [policy] [score] [steps]
Fresh random gradients every step (a boosting loop never sees the same gradients twice); the same loop with GBRL_HIP_NO_NEARTIE_REPLAY=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
policy = sys.argv[5] if len(sys.argv) > 5 else "greedy"
score = sys.argv[6] if len(sys.argv) > 6 else "L2"
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 400
dev = torch.device("cuda:0")
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
def run(env):
    for k in ("GBRL_HIP_NO_NEARTIE_REPLAY", "GBRL_HIP_NEARTIE_REL"): os.environ.pop(k, None)
    os.environ.update(env)
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((N, F), device=dev, generator=g)
    Gs = [torch.randn((N, D), device=dev, generator=g) for _ in range(32)]
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9,
                      split_score_func=score, generator_type="Quantile", use_control_variates=False, batch_size=5000,
                      grow_policy=policy, verbose=0, device="cuda", learner_name="small")
    m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    for i in range(20): m.step(tup(X), None, tup(Gs[i % 32]))
    ms = 1e9
    for _ in range(5):     # (best of five passes over the same gradients: the replays are part of every pass)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): m.step(tup(X), None, tup(Gs[i % 32]))
        torch.cuda.synchronize(); ms = min(ms, (time.perf_counter() - t0) * 1e3 / steps)
    m.set_profiling(2); m.step(tup(X), None, tup(Gs[0])); ph = dict(m.last_phase_times())
    return ms, ph.get("near_replays", 0), ph.get("near_bailouts", 0), ph.get("near_in_kernel", 0)
for env in ({"GBRL_HIP_NO_NEARTIE_REPLAY": "1"}, {}, {"GBRL_HIP_NEARTIE_REL": "7.6e-6"}):
    ms, rp, bo, ik = run(env)
    print("%-40s %dx%d D=%d depth %d %s %s: %.4f ms/step, levels replayed in the one-launch kernel %d, by the level loop %d, trees handed to the level loop %d of %d" % (env or "default (2^-20)", N, F, D, depth, policy, score, ms, ik, rp, bo, 5 * steps + 21))
