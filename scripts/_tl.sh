R="$GRAFT_REPO_ROOT"; W=/tmp/gbrl_c1; rm -rf "$W"; mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$W/a" -o t -- python3 "$R/scripts/cfg1_loop_trace.py" 30 > "$W/a.txt" 2>&1
python3 "$R/scripts/step_timeline.py" "$W/a" "$R/gpurun_out/r05_cfg1_loop_timeline.txt" k_small_prep 60 | tail -30; tail -3 "$W/a.txt"
