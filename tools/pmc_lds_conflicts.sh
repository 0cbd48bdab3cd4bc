#!/bin/bash
# LDS bank-conflict share per kernel over whole training steps (single stream): one rocprofv3 --pmc pass with SQ_LDS_BANK_CONFLICT and
# SQ_LDS_IDX_ACTIVE; prints, per kernel name, conflict cycles / active cycles of the last step's launches.
# usage: bash tools/pmc_lds_conflicts.sh [outdir]
set -e
OUT=${1:-gpurun_out/pmc_lds}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && CTSEG_SIDE_STREAM=0 timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$OUT/run" -- \
   python3 "$ROOT/bench.py" --no-cpu-baseline --fp32-steps 0 --steps 3 --warmup 2 > "$ROOT/$OUT/run.log" 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "run", "**", "*counter_collection.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = defaultdict(dict)
for r in rows:
    disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
ids = sorted(disp)
adam = [i for i in ids if "adam_kernel" in disp[i]["name"]]
step = [i for i in ids if adam[-2] < i <= adam[-1]]
agg = defaultdict(lambda: [0.0, 0.0, 0])
for i in step:
    d = disp[i]
    a = agg[d["name"][:110]]
    a[0] += d.get("SQ_LDS_BANK_CONFLICT", 0.0); a[1] += d.get("SQ_LDS_IDX_ACTIVE", 0.0); a[2] += 1
for name, (c, act, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if act > 0:
        print("%6.1f %% conflict  %12.0f conflict cycles  %12.0f active  x%d  %s" % (100 * c / act, c, act, n, name))
PY
