#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 outputs: kernel-trace durations + any number of --pmc counter_collection CSVs.

usage: pmc_table.py <kernel_trace.csv> <counter_collection.csv>... [--match substring]
Counters are averaged per dispatch (summed over dimensions/instances inside one dispatch first)."""
import collections
import csv
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    args = [a for a in args if a != match]


def short(name):
    m = re.match(r"(?:void )?(?:bd::)?(\w+)<([^>]*)>", name)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else name[:60]


dur = collections.defaultdict(list)
for r in csv.DictReader(open(args[0])):
    if match and match not in r["Kernel_Name"]:
        continue
    dur[(short(r["Kernel_Name"]), "*")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(dict))
for path in args[1:]:
    for r in csv.DictReader(open(path)):
        if match and match not in r["Kernel_Name"]:
            continue
        key = (short(r["Kernel_Name"]), "*")
        d = ctr[key][r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for key in sorted(dur):
    v = sorted(dur[key])
    print(f"{key[0]}  grid={key[1]}  n={len(v)}  median {v[len(v)//2]:.1f} us  min {v[0]:.1f} us")
    for cname in sorted(ctr.get(key, {})):
        vals = list(ctr[key][cname].values())
        print(f"    {cname:32s} {sum(vals)/len(vals):16.0f}")
