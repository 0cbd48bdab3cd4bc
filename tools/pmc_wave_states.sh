#!/bin/bash
# Where the waves of ONE kernel spend their cycles (SQ wave-state counters, MI355X_MICROARCH.md "rocprofv3 PMC slots":
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES): tools/pmc_wave_states.sh <outdir> <kernel substring> -- <python script + args>
# (run on the GPU box from the repo root; separate --pmc passes with --kernel-trace only)
set -e
OUT=$(pwd)/$1; KSUB=$2; shift 3
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/p$i.log" 2>&1) || echo "pass $i failed"
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, json
out, ksub = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    by = {}
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, lst in by.items():
        lst.sort()
        last = lst[-10:]
        vals[name] = sum(v for _, v in last) / len(last)
dur = []
for f in glob.glob(os.path.join(out, "p*", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if ksub in r["Kernel_Name"]]
    dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-10:]]
vals["kernel_us"] = sum(dur) / max(len(dur), 1)
wc = vals.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
        if k in vals: vals[k + "/WAVE_CYCLES"] = vals[k] / wc
print(json.dumps(vals, indent=1))
PY
