"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last steps of a run, each kernel's duration and the idle gap before it
(the dependent chain of a one-sequence step).  Usage: python scripts/trace_gaps.py <kernel_trace.csv> [first_kernel_substring] [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_small_front"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
starts = starts[-nsteps - 1:]
shapes = defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    step = rows[a:b]
    key = tuple(name(r) for r in step)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in step]
    gap = [0.0] + [(int(step[i]["Start_Timestamp"]) - int(step[i - 1]["End_Timestamp"])) / 1e3 for i in range(1, len(step))]
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    shapes[key].append((dur, gap, span))
for key, v in shapes.items():
    n = len(v)
    print("%d steps of %d kernels, first-start to last-end %.1f us (mean)" % (n, len(key), sum(x[2] for x in v) / n))
    for i, k in enumerate(key):
        print("   %-22s %6.2f us   gap before %6.2f us" % (k, sum(x[0][i] for x in v) / n, sum(x[1][i] for x in v) / n))
