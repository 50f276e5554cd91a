#!/usr/bin/env python3
"""Per-kernel medians (and the gap to the next kernel) from a rocprofv3 --kernel-trace CSV: the last `--last` dispatches only (steady state).
    python tools/trace_summary.py <dir-or-csv> [--last 1400] [--filter bd::]"""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--last", type=int, default=1400)
    ap.add_argument("--filter", default="bd::")
    ap.add_argument("--width", type=int, default=110)
    a = ap.parse_args()
    path = a.path
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-a.last:]
    per, gaps = collections.defaultdict(list), collections.defaultdict(list)
    for x, y in zip(last[:-1], last[1:]):
        nm = x["Kernel_Name"][:a.width]
        if a.filter and a.filter not in nm:
            continue
        per[nm].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
        gaps[nm].append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
    span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
    print(f"# {len(last)} dispatches over {span:.1f} us; sum of listed kernel time {sum(sum(v) for v in per.values()):.1f} us")
    for nm, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        d, g = sorted(ds), sorted(gaps[nm])
        print(f"  {len(ds):5d} x median {d[len(d) // 2]:7.2f} us  avg {sum(d) / len(d):7.2f}  (+ median gap after {g[len(g) // 2]:5.2f})  {nm}")


if __name__ == "__main__":
    main()
