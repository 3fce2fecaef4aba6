#!/usr/bin/env python3
"""Timeline of the LAST call of an entry point in a rocprofv3 --kernel-trace: start offset, duration and the idle gap before every kernel
(all streams), from the last launch of the kernel whose name contains <first-kernel> (default msm_prep: one bucket product of tools/msm_only.py).
    python3 tools/kernel_timeline.py <trace dir> [first-kernel substring]"""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
FIRST = sys.argv[2] if len(sys.argv) > 2 else "msm_prep"
starts = [i for i, e in enumerate(ev) if FIRST in e[2]]
first = starts[-1]
t0 = ev[first][0]
end_prev = t0
print("%9s %9s %8s  q  kernel" % ("start us", "dur us", "gap us"))
for s, e, name, q in ev[first:]:
    name = re.sub(r"^void ", "", name.split("(")[0])
    if "rocprim" in name:
        m = re.search(r"rocprim::(?:detail::)?(\w+)", name)
        name = "rocprim::" + (m.group(1) if m else "kernel")
    print("%9.1f %9.1f %8.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end_prev) / 1e3, q, name[:70]))
    end_prev = max(end_prev, e)
print("total %.3f ms" % ((end_prev - t0) / 1e6))
