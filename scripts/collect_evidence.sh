#!/bin/bash
# Run on the GPU box (through gpurun): the kernel-trace summary and the PMC passes the judged numbers come from.  The rocprofv3
# databases are summarised ON the box (scripts/rocprof_summary.py, pmc_summary.py, pmc_levels.py) and deleted: only the text / json
# summaries under gpurun_out/evidence/ travel back (gpurun merges at most 64 MiB); copy them into profiles/.
#   PMC counters are collected in their own runs (--pmc only, never combined with trace domains other than the kernel trace).
set -u
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/evidence"
W=/tmp/gbrl_evidence
rm -rf "$O" "$W"; mkdir -p "$O" "$W"
cd /tmp && export TMPDIR=/tmp
STAMP="$(python3 -c "import json;d=json.load(open('$R/gbrl_amd/build_info.json'));print('build of commit %s%s (sources %s)'%(d['commit'],'+dirty' if d['dirty'] else '',d['src_sha256']))")"
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 --large-ensemble 32"
rocprofv3 --kernel-trace --stats -d "$W/trace" -o t -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace_bench.json" 2> "$O/trace_bench.err"
python3 "$R/scripts/rocprof_summary.py" "$W/trace" "$O/kernel_stats.txt" "round 6, $STAMP: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline"
rm -rf "$W/trace"
# HBM-side traffic: per kernel (2 FETCH + WRITE, calibrated below) and, for k_hist_build, per tree level with the request-size counters
rocprofv3 --pmc FETCH_SIZE -d "$W/pmc_fetch" -o f -- $B > /dev/null 2> "$O/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$W/pmc_write" -o w -- $B > /dev/null 2> "$O/pmc_write.err"
python3 "$R/scripts/pmc_summary.py" "$W/pmc_fetch" "$W/pmc_write" "$O/pmc_traffic.txt" "$O/hist_traffic_2fw.json" "$STAMP" > /dev/null
bash "$R/scripts/hist_traffic_levels.sh" > /dev/null 2>&1     # -> $O/hist_levels_traffic.txt, $O/hist_traffic.json (per-level list)
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$W/pmc_lds1" -o a -- $B > /dev/null 2> "$O/pmc_lds1.err"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$W/pmc_lds2" -o b -- $B > /dev/null 2> "$O/pmc_lds2.err"
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$W/pmc_lds3" -o c -- $B > /dev/null 2> "$O/pmc_lds3.err"
python3 "$R/scripts/pmc_levels.py" k_hist_build 6 "$O/hist_levels_pmc.txt" "round 6, $STAMP: k_hist_build per tree level, rocprofv3 --pmc (three counter sets, separate runs) -- $B" "$W/pmc_lds1" "$W/pmc_lds2" "$W/pmc_lds3" > /dev/null
python3 "$R/bench.py" > "$O/bench_line.json" 2> "$O/bench_line.err"
rm -rf "$W"
ls -la "$O"
