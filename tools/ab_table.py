#!/usr/bin/env python3
"""Tabulate `bd_harness stream_ab` / `dec600` JSON lines (cold = weights rotated through > 2x the Infinity Cache, i.e. from HBM).
usage: ab_table.py <file.jsonl>..."""
import json
import sys

for path in sys.argv[1:]:
    print(f"# {path}")
    for line in open(path):
        if not line.startswith("{") or "cold_us" not in line:
            continue
        r = json.loads(line)
        shape = f"T={r.get('B', r.get('T'))} N={r['N']} K={r['K']}"
        print(f"{r['tag']:<28s} {shape:<24s} variant={r.get('used', '-')!s:<4} cold {r['cold_us']:7.2f} us {r['cold_gbps']:5.0f} GB/s   "
              f"warm {r['warm_us']:7.2f} us {r['warm_gbps']:5.0f} GB/s   bad={r.get('bad')}")
