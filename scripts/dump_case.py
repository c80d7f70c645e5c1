"""Run golden cases through the product on the GPU and dump the ensembles (debug aid; output under gpurun_out/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases as K
import gbrl_amd
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name in sys.argv[1:]:
    case = K.BY_NAME[name]
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = K.drive(m, case, X, Xc, G, y)
    e = m.get_ensemble_data()
    np.savez(os.path.join(ROOT, "gpurun_out", "prod_" + name + ".npz"), pred=np.asarray(pred), **{k: np.asarray(e[k]) for k in K.ENSEMBLE_KEYS})
    print("dumped", name)
