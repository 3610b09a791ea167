"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small CSV for profiles/.
usage: python tools/rocpd_summary.py <results.db> <out.csv>"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:120]


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([short(name), calls, round(tot, 1), round(avg, 2), round(pct, 3)])
    for name, calls, tot, avg, pct in rows[:12]:
        print(f"{short(name)[:70]:70s} calls={calls:5d} total_ms={tot/1e3:10.2f} avg_us={avg:10.1f} {pct:6.2f}%")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
