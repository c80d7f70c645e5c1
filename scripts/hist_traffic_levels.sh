#!/bin/bash
# Run on the GPU box (through gpurun): k_hist_build's HBM-side traffic PER TREE LEVEL, with the raw request-size counters beside
# FETCH_SIZE / WRITE_SIZE so that the gfx950 "x2" correction is calibrated on THIS kernel.
#   level 0 streams contiguous rows: its bytes are known (N (2F + 4D) read, 2^0 F 257 (D+1) 4 written) -> the factor for this access width;
#   levels 1-5 gather 32-byte code records / 32-byte gradient rows through the row list: the request-size split
#   (TCC_EA0_RDREQ_32B / _64B / _128B) gives their bytes without any factor.
# One --pmc pass per counter set (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2), never combined with a trace domain.
#   bash scripts/hist_traffic_levels.sh            -> gpurun_out/evidence/hist_levels_traffic.txt + hist_traffic.json
set -u
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/evidence"
W=/tmp/gbrl_traffic
rm -rf "$W"; mkdir -p "$O" "$W"
cd /tmp && export TMPDIR=/tmp
STAMP="$(python3 -c "import json;d=json.load(open('$R/gbrl_amd/build_info.json'));print('build of commit %s%s (sources %s)'%(d['commit'],'+dirty' if d['dirty'] else '',d['src_sha256']))")"
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1 --large-ensemble 32"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $set -d "$W/p$i" -o p -- $B > /dev/null 2> "$O/traffic_p$i.err" || echo "pass $i ($set) failed: see traffic_p$i.err" >> "$O/hist_levels_traffic.err"
done
python3 "$R/scripts/hist_traffic_table.py" "$O/hist_levels_traffic.txt" "$O/hist_traffic.json" "$STAMP" "$W"/p*
rm -rf "$W"
