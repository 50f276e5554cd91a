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
        # one k-tile = 16 regions x 4 MFMAs; the shipped form (OPT & 8192) unrolls the k loop by four and keeps one rolled copy for nk % 4
        assert len(idx) == 5 * 64, (name, len(idx))
        loop = body[idx[0]:idx[-1] + 1]
        fused = "ELb1E" in name
        waits = [l.strip() for l in loop if "vmcnt" in l]
        assert waits == [f"s_waitcnt vmcnt({13 if fused else 10})"] * 5, (name, waits)
        gaps, cur = [], 0
        for l in loop[1:]:
            if "v_mfma" in l:
                gaps.append(cur)
                cur = 0
            elif isa_gaps.is_instr(l):
                cur += 1
        # between two k-tile copies the layout holds the loader's once-per-tile slow path (next tile's coordinates: ~300 scalar
        # instructions the hot path branches over): at most one such block per copy boundary; every other gap is an MFMA shadow
        cold = [g for g in gaps if g > 100]
        hot = [g for g in gaps if g <= 100]
        assert len(cold) <= 5, (name, cold)
        assert sum(hot) / len(hot) <= 3.0 and max(hot) <= 12, (name, sum(hot) / len(hot), max(hot))
        # accumulators live in AGPRs, operands in VGPRs
        assert all(re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], v\[", body[i]) for i in idx), name


def test_m0_only_written_by_the_dma_statements(kernels):
    _, ks = kernels
    for name, body in ks.items():
        writers = [l.strip() for l in body if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert writers and all(w.startswith("s_add_u32 m0,") for w in writers), (name, [w for w in writers if not w.startswith("s_add_u32 m0,")][:5])


# ---------------------------------------------------------------------------------------------------- prefill attention kernel
@pytest.fixture(scope="module")
def attn_kernels(tmp_path_factory):
    import isa_gaps
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "attn.s"
    src = os.path.join(ROOT, "tests", "native", "attn_isa_probe.hip")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), src])
    ks = {name: body for name, body in isa_gaps.kernels(str(out)) if "prefill_attn_kernel" in name}
    assert len(ks) == 2
    return ks


def test_prefill_attention_isa(attn_kernels):
    """The compile-time properties the attention kernel's correctness and speed hang on, each a bug found on the device:
    * two workgroups per CU: <= 256 VGPRs, no scratch;
    * every inline-assembly v_cvt_pk_bf16_f32 is preceded by its s_nop (its operands come straight from v_exp_f32; without the wait
      state the P fragments were wrong);
    * no vmcnt wait between the top of the tile loop and the last QK^T MFMA (query-fragment loads left pending at loop entry made every
      tile wait for the NEXT tile's K / V loads there);
    * per tile: 32 MFMAs, 33 exponentials, 16 + 32 LDS fragment reads, the lane ^ 32 exchange on the VALU (no ds_bpermute)."""
    for name, body in attn_kernels.items():
        blob = "\n".join(body)
        bf16 = "ILi1E" in name
        assert "scratch_" not in blob, name
        nxt = re.search(r"\.amdhsa_next_free_vgpr (\d+)", blob)
        assert nxt and int(nxt.group(1)) <= 256, (name, nxt)
        mf = "v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x16_f16"
        idx = [i for i, l in enumerate(body) if mf in l]
        assert len(idx) == 32, (name, len(idx))
        # the tile loop: backward branch target that precedes the first MFMA
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        back = [(labels[m.group(1)], i) for i, l in enumerate(body)
                for m in [re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)] if m and m.group(1) in labels and labels[m.group(1)] < i]
        loops = [(lo, hi) for lo, hi in back if lo < idx[0] and hi > idx[-1]]
        assert loops, name
        lo, hi = max(loops)                                           # innermost loop around all 32 MFMAs
        qk_section = body[lo:idx[15] + 1]
        assert not any(re.search(r"s_waitcnt.*vmcnt", l) for l in qk_section), [l for l in qk_section if "vmcnt" in l]
        loop = body[lo:hi]
        assert sum("v_exp_f32" in l for l in loop) == 33, name
        assert sum("ds_read_b64_tr_b16" in l for l in loop) == 32 and sum(re.search(r"\bds_read_b128\b", l) is not None for l in loop) == 16, name
        assert "ds_bpermute" not in blob and "v_permlane32_swap" in blob, name
        if bf16:
            code = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
            for i, l in enumerate(code):
                if l.startswith("v_cvt_pk_bf16_f32"):
                    assert code[i - 1].startswith("s_nop"), (name, code[i - 2:i + 1])


# ---------------------------------------------------------------------------------------------------------------------------------
# Decode-step kernels: waits and prologue work the source never asked for, each found in the ISA in round 5 and each worth 2 - 3 % of the
# decode step (profiles/r05_decode_step.txt, r05_reference_notebook_shapes.txt).  Asserted on the text so they cannot come back silently.
@pytest.fixture(scope="module")
def decode_kernels(tmp_path_factory):
    import isa_gaps
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "decode.s"
    src = os.path.join(ROOT, "tests", "native", "decode_isa_probe.hip")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only", "-S", "-o", str(out), src])
    return {name: body for name, body in isa_gaps.kernels(str(out))}


def _one(ks, key):
    hits = [(n, b) for n, b in ks.items() if key in n]
    assert len(hits) == 1, (key, [n for n, _ in hits])
    return hits[0][1]


def test_decode_attention_has_no_accidental_waits(decode_kernels):
    body = _one(decode_kernels, "decode_attn_kernel")
    first_barrier = next(i for i, l in enumerate(body) if "s_barrier" in l)
    head = body[:first_barrier]
    # RoPE inputs first, then the K / V ring (2 rows x (K, V) + the validity byte each), and no wait for the ring before the barrier that
    # publishes the rotated query (the validity byte's compare used to wait vmcnt(0) behind every row: one HBM round trip per ring slot)
    loads = [i for i, l in enumerate(head) if re.search(r"\bglobal_load_", l)]
    ring = [i for i in loads if "global_load_dwordx4" in head[i]]
    assert len(ring) == 4 and sum("global_load_ubyte" in head[i] for i in loads) == 2
    assert sum("global_load_ushort" in head[i] for i in loads if i < ring[0]) == 7, "the seven RoPE inputs are issued ahead of the ring"
    assert not any("vmcnt(0)" in l for l in head[ring[0]:]), "no full drain between the ring's issue and the first barrier"
    # the split merge: all 12 partial loads of an element in flight together, then counted waits
    sc1 = [i for i, l in enumerate(body) if "global_load_dword " in l and "sc1" in l]
    assert len(sc1) == 12
    between = body[sc1[0]:sc1[-1] + 1]
    assert not any("s_waitcnt vmcnt" in l for l in between), "a wait between the merge loads serialises them (8 dependent L2 round trips before)"


def test_resident_row_linear_reaches_its_first_weight_load_early(decode_kernels):
    body = _one(decode_kernels, "gemv_stream_kernel")
    loads = [i for i, l in enumerate(body) if "buffer_load_dwordx4" in l]
    first_barrier = next(i for i, l in enumerate(body) if "s_barrier" in l)
    # 16 activation-row chunks, then the first weight stage: it sat behind ~870 instructions (32 run-time integer divisions per thread in the
    # rows -> LDS copy) before round 5; incremental addressing brought it to ~590 and the barrier from ~1380 to ~910
    assert loads[16] < 700, loads[16]
    assert first_barrier < 1050, first_barrier
    head = body[:loads[16]]
    assert sum(1 for l in head if "v_rcp_iflag_f32" in l) <= 4, "integer divisions in front of the first weight load"
    assert not any("scratch_" in l for l in body)


def test_rows_kernel_stream_is_branch_free(decode_kernels):
    for key, nmfma in (("delta_rows_kernelILi0ELi2ELi4ELi2ELi0ELi4E", 128), ("delta_rows_kernelILi0ELi1ELi4ELi2ELi0ELi2E", 32)):
        body = _one(decode_kernels, key)
        idx = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x32" in l]
        assert len(idx) == nmfma, (key, len(idx))                      # 4 stages x 4 steps x (masks x tiles) MFMAs, one rolled round
        loop = body[idx[0]:idx[-1] + 1]
        # the sign / activation loads of a stage are selected (v_cndmask on an opaque offset), never wrapped in a divergent branch: hipcc sank them
        # into both arms of one, with s_waitcnt vmcnt(0) between, when the offset was a plain conditional expression (-15 ... -25 %)
        assert not any("s_cbranch" in l for l in loop), key
        assert not any("vmcnt(0)" in l for l in loop), key
        assert sum("buffer_load_dword" in l for l in loop) >= 4 * 5 - 5, key
        assert not any("scratch_" in l for l in body), key
