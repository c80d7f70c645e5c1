#!/bin/bash
# Run on the GPU box: how often the near-tie replay fires on RL-sized steps and what it costs (the record kept as profiles/rNN_neartie_cost.txt).
R="$GRAFT_REPO_ROOT"; cd "$R"
python3 -c "import json;d=json.load(open('gbrl_amd/build_info.json'));print('# build of commit %s%s (sources %s): near-tie replay on RL-sized steps, one MI355X; best of five passes of fresh random gradients per step'%(d['commit'],'+dirty' if d['dirty'] else '',d['src_sha256']))"
run() { timeout 600 python3 "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-3}; }
echo "== frequency and cost at the default window (2^-20), with the replay off, and at eps32 * sqrt(4096) = 7.6e-6"
run scripts/neartie_rate.py 4096 16 1 4 greedy L2 600
run scripts/neartie_rate.py 4096 16 8 4 greedy Cosine 600
run scripts/neartie_rate.py 512 24 4 5 greedy Cosine 600
run scripts/neartie_rate.py 4096 192 8 6 oblivious L2 300
run scripts/neartie_rate.py 4096 128 8 8 oblivious L2 200
echo "== cost of a replayed level: the window forced to 1e-3 (most trees flagged, 16 candidate classes each)"
TAILN=1 run scripts/neartie_cost.py 1e-3 1 L2
TAILN=1 run scripts/neartie_cost.py 1e-3 8 Cosine
