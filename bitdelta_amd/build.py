"""Build libbitdelta_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m bitdelta_amd.build`.

The rebuild gate is a CONTENT hash of every source the library is compiled from (csrc/*.hip, csrc/*.h, include/*.h) plus the
compiler command, stored next to the .so -- not mtimes, and not a hand-kept dependency list (round 1's list missed the header of
the headline kernel, so editing it silently kept a stale library).
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bd_api.hip")
OUT = os.path.join(HERE, "lib", "libbitdelta_hip.so")
STAMP = OUT + ".srchash"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed"] + os.environ.get("HIPCC_EXTRA", "").split()


def sources():
    """Every file the library is built from, in a stable order."""
    files = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))
    return files


def source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sources():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=True):
    """Returns (path, compiled): `compiled` says whether hipcc actually ran."""
    if not force and not needs_build():
        if verbose:
            print(f"bitdelta_amd.build: {OUT} is up to date (source hash matches)", file=sys.stderr)
        return OUT, False
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    stamp = source_hash()                # of what the compiler is about to read (an edit during the build must not be stamped as built)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(stamp + "\n")
    if verbose:
        print(f"bitdelta_amd.build: compiled {OUT}", file=sys.stderr)
    return OUT, True


if __name__ == "__main__":
    path, compiled = build(force="--force" in sys.argv)
    print(path, "(compiled)" if compiled else "(up to date)")
