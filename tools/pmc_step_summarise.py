"""Per-dispatch HBM-side bytes of the last training step in the tools/pmc_step.sh passes, in launch order
(FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 B)."""
import csv
import glob
import os
import sys


def load(out, name):
    f = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main(out):
    fe, wr = load(out, "FETCH_SIZE"), load(out, "WRITE_SIZE")
    # last step = dispatches after the second-to-last adam kernel
    def last_step(rows):
        idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
        return rows[idx[-2] + 1: idx[-1] + 1]
    fe, wr = last_step(fe), last_step(wr)
    assert len(fe) == len(wr), (len(fe), len(wr))
    tot_f = tot_w = 0.0
    lines = []
    for a, b in zip(fe, wr):
        f = float(a["Counter_Value"]) * 1024 * 2 / 1e6
        w = float(b["Counter_Value"]) * 1024 / 1e6
        tot_f += f
        tot_w += w
        dur = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
        if f + w > 20:
            lines.append("%8.1f MB read %8.1f MB written %8.1f us  %s" % (f, w, dur, a["Kernel_Name"][:100]))
    print("\n".join(lines))
    print("total per step: %.2f GB read, %.2f GB written" % (tot_f / 1e3, tot_w / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
