#!/usr/bin/env python3
"""Issue-slot audit of an MFMA loop in hipcc's -save-temps assembly: for every kernel whose mangled name contains PATTERN,
print the histogram of instructions issued between consecutive v_mfma in the loop body, and (with -v) the gaps above a limit.
    python tools/isa_gaps.py file.s PATTERN [-v LIMIT]
A one-wave-per-SIMD stream hides at most ~5 single-issue instructions per v_mfma_f32_32x32x16 (MI355X_MICROARCH.md)."""
import collections
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip("\n"))
            if ".end_amdhsa_kernel" in line:
                yield name, body
                name = None


def is_instr(l):
    t = l.strip()
    return bool(t) and not t.startswith(";") and not t.startswith(".") and not t.endswith(":")


def main():
    path, pat = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[sys.argv.index("-v") + 1]) if "-v" in sys.argv else None
    for name, body in kernels(path):
        if pat not in name:
            continue
        idx = [i for i, l in enumerate(body) if "v_mfma" in l]
        if not idx:
            continue
        loop = body[idx[0]:idx[-1] + 1]
        gaps, cur, buf = [], 0, []
        big = []
        for l in loop[1:]:
            if "v_mfma" in l:
                gaps.append(cur)
                if limit is not None and cur > limit:
                    big.append((cur, buf))
                cur, buf = 0, []
            elif is_instr(l):
                cur += 1
                buf.append(l)
        h = collections.Counter(gaps)
        scratch = sum("scratch_" in l for l in body)
        print(f"{name[:110]}\n  mfma {len(idx)}  fillers {sum(gaps)} ({sum(gaps) / max(len(gaps), 1):.2f}/mfma)  scratch ops in kernel {scratch}")
        print("  gap histogram:", " ".join(f"{k}:{h[k]}" for k in sorted(h)))
        for n, b in big:
            print(f"  ---- gap of {n}")
            for l in b:
                print("   ", l.strip())


if __name__ == "__main__":
    main()
