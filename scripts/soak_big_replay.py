"""Big-batch replay soak (round 6): ONE model per configuration takes 40 steps on batches of changing size above 65 536 rows with
GBRL_HIP_NEARTIE_MAX_ROWS=0 (every near-tie replayed: per-node bitmaps, the parallel order, seqsum chains) -- buffer growth and reuse across
levels and steps, four policy / score / width combinations; must not fault and must keep finite predictions.
    python scripts/soak_big_replay.py"""
import os, sys, time
os.environ["GBRL_HIP_NEARTIE_MAX_ROWS"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import gbrl_amd

rng = np.random.default_rng(7)
for policy, score, D in (("greedy", "Cosine", 8), ("oblivious", "L2", 4), ("greedy", "L2", 4), ("oblivious", "Cosine", 12)):
    F = 6
    m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=5, min_data_in_leaf=0, n_bins=64, par_th=10, cv_beta=0.9, split_score_func=score,
                      generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy=policy, verbose=0, device="cpu", learner_name="soak")
    m.set_feature_weights(np.ones(F, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
    m.set_profiling(2)
    t0 = time.time()
    reps = 0
    for it in range(40):
        N = int(rng.choice([70001, 131072, 200000, 66000, 400003, 90000]))
        X = rng.standard_normal((N, F)).astype(np.float32)
        G = (np.tanh(X[:, :1]) * 0.05 + rng.standard_normal((N, D))).astype(np.float32)
        m.step(X, None, G)
        reps = dict(m.last_phase_times()).get("near_replays", 0)
    p = np.asarray(m.predict(X, None, 0, 0))
    assert np.isfinite(p).all() and m.get_num_trees() == 40
    print(policy, score, "D", D, ": 40 steps on 66 000 .. 400 003 rows, replayed levels", reps, "%.1f s" % (time.time() - t0), flush=True)
