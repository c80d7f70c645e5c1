export GBRL_HIP_SMALL_GROW_PROF=1
python3 scripts/small_step_trace.py 4096 16 1 4 greedy 2>&1 | tail -3
python3 scripts/small_step_trace.py 4096 192 8 6 oblivious 2>&1 | tail -3
python3 scripts/cfg5_step_trace.py 50 2>&1 | tail -3
