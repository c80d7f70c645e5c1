#!/bin/bash
# Run on the GPU box: dispatch timelines of RL-sized steps (rocprofv3 --kernel-trace + scripts/step_timeline.py).
#   bash scripts/small_grow_timeline.sh <tag>     -> gpurun_out/<tag>_cfg1_timeline.txt, <tag>_cfg5_timeline.txt, <tag>_4096x192_timeline.txt
set -u
R="$GRAFT_REPO_ROOT"; T="${1:-r05}"; W=/tmp/gbrl_sg_tl; rm -rf "$W"; mkdir -p "$W" "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$W/a" -o t -- python3 "$R/scripts/small_step_trace.py" 4096 16 1 4 greedy > "$W/a.txt" 2>&1
python3 "$R/scripts/step_timeline.py" "$W/a" "$R/gpurun_out/${T}_cfg1_timeline.txt" k_small_prep 40 | tail -30; tail -1 "$W/a.txt"
rocprofv3 --kernel-trace -d "$W/b" -o t -- python3 "$R/scripts/cfg5_step_trace.py" 100 > "$W/b.txt" 2>&1
python3 "$R/scripts/step_timeline.py" "$W/b" "$R/gpurun_out/${T}_cfg5_timeline.txt" k_cat_distinct_insert 40 | tail -40; tail -1 "$W/b.txt"
rocprofv3 --kernel-trace -d "$W/c" -o t -- python3 "$R/scripts/small_step_trace.py" 4096 192 8 6 oblivious > "$W/c.txt" 2>&1
python3 "$R/scripts/step_timeline.py" "$W/c" "$R/gpurun_out/${T}_4096x192_timeline.txt" k_small_prep 40 | tail -30; tail -1 "$W/c.txt"
