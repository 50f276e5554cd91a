#!/usr/bin/env python3
"""Summarise the basic blocks of every kernel in a gfx950 .s file (instruction mix per block)."""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else None
parts = re.split(r'\n([.A-Za-z_][\w$.]*):\s*(?:;[^\n]*)?\n', s)
cur = None
for i in range(1, len(parts), 2):
    name, txt = parts[i], parts[i + 1]
    if not name.startswith('.LBB'):
        cur = name
        if 'v_mfma' not in txt and 's_endpgm' not in txt and not name.startswith('_Z'):
            continue
        print('==', name)
    if want and cur and want not in cur:
        continue
    lines = [l.strip() for l in txt.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    if not lines:
        continue
    cnt = lambda pat: sum(1 for l in lines if re.match(pat, l))
    print(f"  {name:12s} n={len(lines):4d} mfma={cnt('v_mfma'):3d} ds_rd={cnt('ds_read'):3d} dma={cnt('global_load_lds'):2d} "
          f"gload={cnt('global_load_d'):2d} gstore={cnt('global_store'):3d} valu={cnt('v_(?!mfma)'):4d} salu={cnt('s_(?!waitcnt|barrier|nop)'):3d} "
          f"wait={cnt('s_waitcnt'):2d} bar={cnt('s_barrier'):1d} nop={cnt('s_nop'):2d}")
