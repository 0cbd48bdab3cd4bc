#!/bin/bash
# HBM-side traffic and MFMA-busy counters of the bottleneck Conv3d 256->256 launch (forward op 22 of the training plan):
# one rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), summarised by
# tools/pmc_summarise.py into profiles/r01_pmc_bottleneck.json.  Run on the GPU box from the repo root:
#   bash tools/pmc_bottleneck.sh gpurun_out/pmc
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && CTSEG_SIDE_STREAM=0 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOT/$OUT/$tag" -- \
     python3 "$ROOT/tools/bench_layers.py" --only fwd:22 --loop 20 > "$ROOT/$OUT/$tag.log" 2>&1)
done
python3 tools/pmc_summarise.py "$OUT"
