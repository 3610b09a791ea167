"""Aggregate rocprofv3 --pmc counter_collection.csv passes (one directory per pass) per kernel -> CSV + stdout.
usage: python tools/pmc_summary.py <dir with pass subdirs> <out.csv> [name filter ...]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main(root, out, filters):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kn = short(r["Kernel_Name"])
            if filters and not any(x in kn for x in filters):
                continue
            agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[kn]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            agg[kn]["VGPR"].append(float(r["VGPR_Count"]))
            agg[kn]["LDS_bytes"].append(float(r["LDS_Block_Size"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "samples", "mean_per_launch"])
        for kn in sorted(agg):
            print(kn)
            for c, vals in sorted(agg[kn].items()):
                m = sum(vals) / len(vals)
                w.writerow([kn, c, len(vals), f"{m:.6g}"])
                print(f"   {c:28s} n={len(vals):3d} mean={m:.5g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
