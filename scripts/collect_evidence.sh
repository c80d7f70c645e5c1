#!/bin/bash
# Run on the GPU box (through gpurun): the kernel-trace summary and the PMC passes the judged numbers come from.  Outputs under
# gpurun_out/evidence/; scripts/pmc_summary.py, scripts/pmc_levels.py and scripts/rocprof_summary.py turn them into profiles/*.txt.
#   PMC counters are collected in their own runs (--pmc only, never combined with trace domains other than the kernel trace).
set -u
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/evidence"
rm -rf "$O"; mkdir -p "$O"
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 --large-ensemble 32"
rocprofv3 --kernel-trace --stats -d "$O/trace" -o t -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace_bench.json" 2> "$O/trace_bench.err"
rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$O/pmc_lds1" -o a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$O/pmc_lds2" -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$O/pmc_lds3" -o c -- $B > /dev/null 2>&1
ls -R "$O" | head -40
