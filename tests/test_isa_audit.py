"""ISA audit of the four-wave persistent kernels (bd_gemm_w4.h) -- runs WITHOUT a GPU: hipcc cross-compiles the two shipped
instantiations to assembly and the properties their performance hangs on are asserted on the text.  Each was a measured regression
when it broke during development:
  * one wave per SIMD: 256 AGPRs (all 16 accumulators) + <= 256 VGPRs, no scratch (a spill's reload carries `s_waitcnt vmcnt(0)`,
    which serialises the output stores and drains the LDS-DMA ring);
  * inside the MFMA loop the ONLY vmcnt wait is the hand-counted one (a VMEM load the compiler's waitcnt pass believes pending at the
    loop entry puts `vmcnt(0)` in front of the first reuse of its register: 2970 instead of 2355 cycles per k-tile);
  * 64 MFMAs per k-tile, and the fillers between consecutive MFMAs stay within what a one-wave-per-SIMD stream can hide on average;
  * M0 is written only by the LDS-DMA statements (they do not save / restore it).
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    import isa_gaps
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "w4.s"
    src = os.path.join(ROOT, "tests", "native", "w4_isa_probe.hip")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only", "-S", "-o", str(out), src])
    text = open(out).read()
    ks = {name: body for name, body in isa_gaps.kernels(str(out)) if "delta_gemm_w4_kernel" in name}
    assert len(ks) == 2
    return text, ks


def _meta(text, name, key):
    m = re.search(re.escape(name) + r"\n(?:.*\n)*?\s*\." + key + r":\s*(\d+)", text)
    return int(m.group(1)) if m else None


def test_register_file_and_scratch(kernels):
    text, ks = kernels
    for name, body in ks.items():
        blob = "\n".join(body)
        assert "scratch_" not in blob, name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", blob), name
        accum = re.search(r"\.amdhsa_accum_offset (\d+)", blob)
        nxt = re.search(r"\.amdhsa_next_free_vgpr (\d+)", blob)
        assert accum and nxt and int(accum.group(1)) <= 256 and int(nxt.group(1)) - int(accum.group(1)) == 256, (name, accum, nxt)


def test_mfma_loop_waits_and_fillers(kernels):
    import isa_gaps
    _, ks = kernels
    for name, body in ks.items():
        idx = [i for i, l in enumerate(body) if "v_mfma_f32_32x32x16_bf16" in l]
        assert len(idx) == 64, (name, len(idx))                      # one k-tile = 16 regions x 4 MFMAs, not unrolled further
        loop = body[idx[0]:idx[-1] + 1]
        fused = "ELb1E" in name
        waits = [l.strip() for l in loop if "vmcnt" in l]
        assert waits == [f"s_waitcnt vmcnt({13 if fused else 10})"], (name, waits)
        gaps, cur = [], 0
        for l in loop[1:]:
            if "v_mfma" in l:
                gaps.append(cur)
                cur = 0
            elif isa_gaps.is_instr(l):
                cur += 1
        assert sum(gaps) / len(gaps) <= 3.0 and max(gaps) <= 12, (name, sum(gaps) / len(gaps), max(gaps))
        # accumulators live in AGPRs, operands in VGPRs
        assert all(re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], v\[", body[i]) for i in idx), name


def test_m0_only_written_by_the_dma_statements(kernels):
    _, ks = kernels
    for name, body in ks.items():
        writers = [l.strip() for l in body if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert writers and all(w.startswith("s_add_u32 m0,") for w in writers), (name, [w for w in writers if not w.startswith("s_add_u32 m0,")][:5])
