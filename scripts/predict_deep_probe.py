"""bench.py's predict_depth8 leg alone (1000 oblivious depth-8 trees, 2^20 x 128 rows), plus the same ensemble shape at depth 6 for comparison:
    python3 scripts/predict_deep_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gbrl_amd, bench
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1234)
X = torch.randn((1 << 20, 128), device=dev, generator=gen)
for depth in (8, 7, 6):
    print(json.dumps(bench.leg_predict_deep(torch, np, gbrl_amd, dev, X, 8, 256, depth=depth)), flush=True)
