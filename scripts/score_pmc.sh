set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/evidence"; W=/tmp/gbrl_score_pmc; rm -rf "$W"; mkdir -p "$O" "$W"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 --large-ensemble 32"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$W/a" -o a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$W/b" -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$W/c" -o c -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES -d "$W/d" -o d -- $B > /dev/null 2>&1
python3 "$R/scripts/pmc_levels.py" k_score 6 "$O/score_levels_pmc.txt" "k_score per tree level" "$W/a" "$W/b" "$W/c" "$W/d" > /dev/null
cat "$O/score_levels_pmc.txt"
