cd "$GRAFT_REPO_ROOT"
export GBRL_HIP_ROOT_COUNTS=2 GBRL_HIP_CAT_CHECK=1
run() { echo "== $*"; timeout 1500 python3 "$@" 2>&1 | grep -v "amdgpu.ids" | tail -${TAILN:-8}; }
run scripts/parity_sweep.py 400 9500
run scripts/parity_sweep.py 200 9700 cat
run scripts/parity_sweep.py 200 9700 ref
run scripts/parity_sweep.py 200 9700 weights
run scripts/parity_sweep.py 200 9700 dev
echo "== level loop"
GBRL_HIP_NO_SMALL_GROW=1 GBRL_HIP_NO_SMALL_PREP=1 run scripts/parity_sweep.py 400 9500
run scripts/parity_sweep.py 200 9700 wide
