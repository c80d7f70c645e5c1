#!/bin/bash
# Run on the GPU box (through gpurun): PMC counters of the round-4 predict kernels (register tiles), the same two sets of 8 SQ
# counters as profiles/r02_pmc_predict.txt so that the instruction mix per (64 rows, tree) can be compared.  Counters are collected
# in their own runs (--pmc only).  Only the text summary travels back (gpurun_out/evidence/r04_pmc_predict.txt -> profiles/).
set -u
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/evidence"
W=/tmp/gbrl_pmc_predict
rm -rf "$W"; mkdir -p "$O" "$W"
cd /tmp && export TMPDIR=/tmp
STAMP="$(python3 -c "import json;d=json.load(open('$R/gbrl_amd/build_info.json'));print('build of commit %s%s (sources %s)'%(d['commit'],'+dirty' if d['dirty'] else '',d['src_sha256']))")"
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY"
SET2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
OUT="$O/r04_pmc_predict.txt"
echo "# round 4, $STAMP: rocprofv3 --pmc (two passes of 8 SQ counters, separate runs) --kernel-include-regex 'k_predict_|k_pack_codes' -- python3 scripts/predict_pmc.py <trees> <features> 2" > "$OUT"
echo "# per-call sums over all CUs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (compare profiles/r02_pmc_predict.txt: k_predict_obl2, the same 1024-tree shape)" >> "$OUT"
for shape in "1024 128" "28 128" "1024 192"; do
  set -- $shape
  rocprofv3 --pmc $SET1 --kernel-include-regex 'k_predict_|k_pack_codes' -d "$W/a" -o a -- python3 "$R/scripts/predict_pmc.py" $1 $2 2 > /dev/null 2> "$O/pmc_predict_a.err"
  rocprofv3 --pmc $SET2 --kernel-include-regex 'k_predict_|k_pack_codes' -d "$W/b" -o b -- python3 "$R/scripts/predict_pmc.py" $1 $2 2 > /dev/null 2> "$O/pmc_predict_b.err"
  for k in k_predict_reg k_predict_pc k_pack_codes; do
    T="$(python3 "$R/scripts/pmc_table.py" $k "$W/a" "$W/b")"
    if [ -n "$T" ]; then
      echo "" >> "$OUT"; echo "## $1 trees x 2^20 rows x $2 features, depth 6, 8 outputs: $k" >> "$OUT"; echo "$T" >> "$OUT"
    fi
  done
  rm -rf "$W/a" "$W/b"
done
cat "$OUT"
