"""Product vs product: trees grown with the templated / wide histogram kernels against the run-time-D kernel (GBRL_HIP_HIST_GENERIC=1),
one child process per setting.  Used by tests/test_gpu_edges.py.
    python scripts/hist_variants_probe.py"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')): sys.path.insert(0, p)
import numpy as np
if len(sys.argv) > 1:
    import cases as K, gbrl_amd
    from test_gpu_edges import _case
    D=int(sys.argv[1]); nb=int(sys.argv[2]); pol=sys.argv[3]
    case=_case("h", D=D, F=19, N=4100, depth=4, n_bins=nb, policy=pol, trees=2, score="Cosine" if D%2 else "L2")
    X,Xc,G,y=K.make_inputs(case)
    m=gbrl_amd.GBRL(**K.ctor_kwargs(case)); K.drive(m,case,X,Xc,G,y)
    e=m.get_ensemble_data()
    np.savez(sys.argv[4], **{k:np.asarray(e[k]) for k in K.ENSEMBLE_KEYS})
else:
    for D,nb,pol in [(11,256,"greedy"),(17,256,"greedy"),(18,256,"oblivious"),(24,256,"greedy"),(31,64,"oblivious"),(38,256,"oblivious"),(40,64,"greedy"),(45,256,"oblivious"),(63,256,"greedy"),(10,1500,"oblivious"),(3,1000,"greedy"),(6,2000,"oblivious")]:
        outs=[]
        for gen in ("0","1"):
            f=os.path.join(__import__("tempfile").gettempdir(), "hist_variant_%s.npz" % gen)
            subprocess.run([sys.executable,__file__,str(D),str(nb),pol,f],env=dict(os.environ,GBRL_HIP_HIST_GENERIC=gen),check=True,capture_output=True)
            outs.append(np.load(f))
        same=all(np.array_equal(outs[0][k],outs[1][k]) for k in outs[0].files)
        print(D,nb,pol,"wide==generic:",same, [k for k in outs[0].files if not np.array_equal(outs[0][k],outs[1][k])])
