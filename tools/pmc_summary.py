#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel dispatch (largest dispatch of each kernel)."""
import collections
import csv
import glob
import sys


def main(dirs):
    for d in dirs:
        for path in glob.glob(d + "/*/*_counter_collection.csv"):
            rows = list(csv.DictReader(open(path)))
            agg = collections.defaultdict(dict)
            for r in rows:
                name = r["Kernel_Name"].split("(")[1].split(")")[-1] if False else r["Kernel_Name"]
                short = name.replace("(anonymous namespace)::", "").split("(")[0]
                if short.startswith("__amd"):
                    continue
                k = (short, int(r["Grid_Size"]), r["Dispatch_Id"])
                agg[k][r["Counter_Name"]] = float(r["Counter_Value"])
                agg[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                agg[k]["vgpr"] = r["VGPR_Count"]; agg[k]["scratch"] = r["Scratch_Size"]
            best = {}
            for (short, grid, disp), v in agg.items():
                if short not in best or grid > best[short][0]:
                    best[short] = (grid, v)
            for short, (grid, v) in best.items():
                print(d, short, "grid", grid, {a: (("%.5g" % b) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main(sys.argv[1:])
