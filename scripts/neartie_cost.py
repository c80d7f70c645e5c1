"""What a replayed level costs on an RL-sized step: the window forced wide (GBRL_HIP_NEARTIE_REL, default 1e-3) so that most trees are flagged:
    python3 scripts/neartie_cost.py [window] [outputs] [score]      (4096 x 16, greedy depth 4, fresh gradients per step)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["GBRL_HIP_NEARTIE_REL"] = sys.argv[1] if len(sys.argv) > 1 else "1e-3"
import numpy as np, torch, gbrl_amd
N, F, D, depth = 4096, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 8, 4
score = sys.argv[3] if len(sys.argv) > 3 else "Cosine"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((N, F), device=dev, generator=g); Gs = [torch.randn((N, D), device=dev, generator=g) for _ in range(16)]
m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=depth, min_data_in_leaf=0, n_bins=256, par_th=10, cv_beta=0.9, split_score_func=score, generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy="greedy", verbose=0, device="cuda", learner_name="small")
m.set_feature_weights(np.ones(F, np.float32)); m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
tup = lambda t: (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
for i in range(20): m.step(tup(X), None, tup(Gs[i % 16]))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200): m.step(tup(X), None, tup(Gs[i % 16]))
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 5
m.set_profiling(2); m.step(tup(X), None, tup(Gs[0])); ph = dict(m.last_phase_times())
print("rel %s D=%d %s: %.4f ms/step, in-kernel %d, level-loop replays %d bailouts %d of 221; last step phases: %s" % (os.environ["GBRL_HIP_NEARTIE_REL"], D, score, ms, ph.get("near_in_kernel", 0), ph.get("near_replays", 0), ph.get("near_bailouts", 0), {k: round(v, 4) for k, v in ph.items() if not k.startswith("near_")}))
