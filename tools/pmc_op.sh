#!/bin/bash
# HBM traffic / MFMA-busy / LDS-conflict counters of ONE op of the recorded programs, one rocprofv3 --pmc pass per counter group:
#   bash tools/pmc_op.sh gpurun_out/pmc_fwd25 fwd:25 conv_up8      (run on the GPU box from the repo root; summary on stdout)
set -e
OUT=$1; OP=$2; KSUB=$3
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && CTSEG_SIDE_STREAM=0 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/$tag" -- \
     python3 "$ROOT/tools/bench_layers.py" --only $OP --loop 20 > "$ROOT/$OUT/$tag.log" 2>&1)
done
python3 tools/pmc_op.py "$OUT" "$KSUB" "$OP"
