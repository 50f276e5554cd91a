"""Build libbitdelta_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m bitdelta_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bd_api.hip")
OUT = os.path.join(HERE, "lib", "libbitdelta_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("bd_api.hip", "bd_common.h", "bd_bits.h", "bd_gemm_mfma.h", "bd_gemm_pp.h", "bd_gemm_pf.h", "bd_gemm_generic.h", "bd_gemv.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "bitdelta_hip.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
