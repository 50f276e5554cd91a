#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel durations (kernel-trace) and PMC counter sums."""
import sqlite3
import sys
from collections import defaultdict


def table(cur, prefix):
    r = cur.execute("select name from sqlite_master where type='table' and name like ?", (prefix + '%',)).fetchall()
    return r[0][0] if r else None


def main(path, kfilter=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    kd, ks, pe, pi = (table(cur, t) for t in ('rocpd_kernel_dispatch', 'rocpd_info_kernel_symbol', 'rocpd_pmc_event', 'rocpd_info_pmc'))
    kcols = [r[1] for r in cur.execute(f'pragma table_info("{kd}")')]
    names = {r[0]: r[1] for r in cur.execute(f'select id, kernel_name from "{ks}"')}
    rows = cur.execute(f'select id, kernel_id, start, end, event_id from "{kd}"').fetchall() if 'event_id' in kcols else \
        cur.execute(f'select id, kernel_id, start, end, id from "{kd}"').fetchall()
    per = defaultdict(list)
    ev2k = {}
    for did, kid, st, en, ev in rows:
        nm = names.get(kid, str(kid))
        per[nm].append((en - st) / 1000.0)
        ev2k[ev] = nm
    print(f"# {path}")
    for nm, ds in sorted(per.items(), key=lambda x: -sum(x[1])):
        if kfilter and kfilter not in nm:
            continue
        ds2 = sorted(ds)
        print(f"kernel {nm[:110]}\n   calls={len(ds)} total_us={sum(ds):.1f} avg_us={sum(ds)/len(ds):.2f} min_us={ds2[0]:.2f} med_us={ds2[len(ds)//2]:.2f}")
    if pe and pi and cur.execute(f'select count(*) from "{pe}"').fetchone()[0]:
        pnames = {r[0]: r[1] for r in cur.execute(f'select id, name from "{pi}"')}
        ecols = [r[1] for r in cur.execute(f'pragma table_info("{pe}")')]
        sums = defaultdict(lambda: defaultdict(float))
        cnts = defaultdict(set)
        for ev, pid, val in cur.execute(f'select event_id, pmc_id, value from "{pe}"'):
            nm = ev2k.get(ev, '?')
            sums[nm][pnames.get(pid, pid)] += val
            cnts[nm].add(ev)
        for nm, d in sums.items():
            if kfilter and kfilter not in nm:
                continue
            n = len(cnts[nm])
            print(f"pmc {nm[:110]} (dispatches={n}; per-dispatch averages)")
            for c, v in sorted(d.items()):
                print(f"   {c:32s} {v / n:16.1f}")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
