#!/usr/bin/env python3
"""Timeline of the LAST bucket product in a rocprofv3 --kernel-trace of tools/msm_only.py: start offset, duration and the idle gap before
every kernel (both streams).   python3 tools/msm_timeline.py <trace dir>"""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
starts = [i for i, e in enumerate(ev) if "msm_prep" in e[2]]
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
