"""Summarise rocprofv3 --pmc output (counter_collection csv): per kernel name, mean
counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are reported in KiB by this stack; on
gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x
(MI355X_MICROARCH.md section HBM) - the corrected figure is printed next to the raw one."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
traffic_json = sys.argv[2] if len(sys.argv) > 2 else None     # optional: write profiles/pmc_traffic.json
# PMC_LAST=k: mean over the LAST k dispatches of every kernel (the timed steps of a `--warmup W --steps k` run: the first
# frames of a sequence search wider windows); default: all dispatches
LAST = int(os.environ.get("PMC_LAST", "0"))
SHAPE = [int(t) for t in os.environ.get("PMC_SHAPE", "1024,320,240,100").split(",")]    # batch, width, height, features
GIT = os.environ.get("PMC_GIT", "unknown")
traffic = defaultdict(dict)
for cdir in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(cdir):
        continue
    files = glob.glob(os.path.join(cdir, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?").split("(")[0].replace("sl2::", "").replace("void ", "")
                name = name.split("<")[0].strip()
                acc[name][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print("==", os.path.basename(cdir), "files:", len(files))
    for name, cs in sorted(acc.items()):
        parts = []
        for c, vals in sorted(cs.items()):
            if LAST > 0:
                vals = vals[-LAST:]
            mean = sum(vals) / len(vals)
            if c == "FETCH_SIZE":
                parts.append("%s=%.0f KiB (x2 corrected %.1f MiB)" % (c, mean, mean * 2 / 1024.0))
                traffic[name]["fetch_raw_bytes"] = int(mean * 1024)
                traffic[name]["fetch_corrected_bytes"] = int(mean * 2048)
            elif c == "WRITE_SIZE":
                parts.append("%s=%.0f KiB (%.1f MiB)" % (c, mean, mean / 1024.0))
                traffic[name]["write_bytes"] = int(mean * 1024)
            elif c == "SQ_INSTS_VALU":
                parts.append("%s=%.4g" % (c, mean))
                traffic[name]["valu_insts"] = mean
            elif c == "SQ_VALU_MFMA_BUSY_CYCLES":
                parts.append("%s=%.4g" % (c, mean))
                traffic[name]["mfma_busy_cycles"] = mean
            elif c == "SQ_INSTS_SALU":
                parts.append("%s=%.4g" % (c, mean))
                traffic[name]["salu_insts"] = mean
            else:
                parts.append("%s=%.4g" % (c, mean))
        print("%-28s n=%-4d %s" % (name[:28], len(next(iter(cs.values()))), "  ".join(parts)))

if traffic_json:
    for name, d in traffic.items():
        d["hbm_bytes"] = d.get("fetch_corrected_bytes", 0) + d.get("write_bytes", 0)
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU ... (separate passes, --kernel-trace only) on "
                  "`python bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile`; mean per dispatch over the last %d "
                  "dispatches of every kernel" % LAST if LAST else "all dispatches",
        "meta": {"shape": SHAPE, "git": GIT},
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); calibrated on the "
                      "kernel that streams the covariance P once (k_build_AS): 1024 x 320 x 320 x 8 B = 800 MiB",
        "kernels": dict(sorted(traffic.items()))}, open(traffic_json, "w"), indent=1)
