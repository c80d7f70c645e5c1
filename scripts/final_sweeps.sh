#!/bin/bash
# Run on the GPU box: the randomised sweeps on the current build (the record kept as profiles/rNN_final_sweeps.txt).
#   bash scripts/final_sweeps.sh > gpurun_out/final_sweeps.txt 2>&1
R="$GRAFT_REPO_ROOT"; cd "$R"
export GBRL_HIP_ROOT_COUNTS=2 GBRL_HIP_CAT_CHECK=1
python3 -c "import json;d=json.load(open('gbrl_amd/build_info.json'));print('# build of commit %s%s (sources %s); GBRL_HIP_ROOT_COUNTS=2, GBRL_HIP_CAT_CHECK=1; RL-sized cases take k_small_prep + k_small_grow'%(d['commit'],'+dirty' if d['dirty'] else '',d['src_sha256']))"
run() { echo "== $*"; timeout 1500 python3 "$@" 2>&1 | tail -${TAILN:-1}; }
run scripts/selfcheck_sweep.py 120 9100
SELFCHECK_RANDOM_N=1 run scripts/selfcheck_sweep.py 80 9300
run scripts/parity_sweep.py 400 9500
run scripts/parity_sweep.py 200 9700 cat
run scripts/parity_sweep.py 200 9700 wide
run scripts/parity_sweep.py 200 9700 ref
run scripts/parity_sweep.py 200 9700 dev
run scripts/parity_sweep.py 200 9700 weights
echo "== the same seeds through the level loop (GBRL_HIP_NO_SMALL_GROW=1 GBRL_HIP_NO_SMALL_PREP=1)"
GBRL_HIP_NO_SMALL_GROW=1 GBRL_HIP_NO_SMALL_PREP=1 run scripts/parity_sweep.py 400 9500
run scripts/predict_reg_sweep.py 1200
run scripts/fit_sweep.py 60 9900
run scripts/fit_sweep.py 100 13600
run scripts/fit_sweep.py 40 7000
run scripts/sharded_sweep.py 120
TAILN=1 run scripts/chain_sweep.py
run scripts/parity_sweep.py 300 23000 refcat
run scripts/parity_sweep.py 200 24000 refweights
run scripts/parity_sweep.py 500 20000
echo "== batches above 65 536 rows against the reference build: default (exact arg-max) and with every near-tie replayed"
run scripts/bign_sweep.py 200 35000
GBRL_HIP_NEARTIE_MAX_ROWS=0 run scripts/bign_sweep.py 200 35000
