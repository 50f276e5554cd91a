"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/bitdelta_hip.h declares,
the Python mirror keeps the reference's names / signatures / error behaviour, the product never touches oracle/,
and nothing computes without a GPU (no CPU fallback)."""
import ast
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name=None):
    """functions declared in include/<name> (default: the stable header + the test-hook header, round 6 split)"""
    out = []
    for n in ([name] if name else ["bitdelta_hip.h", "bitdelta_hip_test.h"]):
        txt = open(os.path.join(ROOT, "include", n)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out += re.findall(r"\b(bd_[a-z0-9_]+)\s*\(", txt)
    return out


def test_tuning_hooks_live_in_the_test_header_only():
    stable, hooks = header_functions("bitdelta_hip.h"), header_functions("bitdelta_hip_test.h")
    assert not [n for n in stable if n.startswith("bd_set_") or n.startswith("bd_last_")], "A/B hooks belong in include/bitdelta_hip_test.h"
    assert hooks and all(n.startswith("bd_set_") or n.startswith("bd_last_") for n in hooks)
    assert not set(stable) & set(hooks)


def test_library_exports_every_declared_symbol():
    from bitdelta_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/*.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in bitdelta_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert L.bd_version() >= 1
    assert L.bd_error_string(0) == b"ok" and b"divisible" in L.bd_error_string(-1)
    # argument validation happens before any device work, so these are safe without a GPU
    assert L.bd_pack(None, 1, 33, 4, 0, 4, 1, None, 32, None) == -1
    assert L.bd_pack(None, 1, 24, 4, 0, 4, 1, None, 12, None) == -2
    assert L.bd_pack(None, 0, 64, 4, 0, 4, 1, None, 32, None) == 0           # empty batch is a no-op
    assert L.bd_gemm_workspace_bytes(6, 1, 4096, 4096) > 0 and L.bd_gemm_workspace_bytes(1, 2048, 4096, 4096) == 0
    assert L.bd_binarize_workspace_bytes(4096, 4096) == 64 * 16 * 4
    # decode path: <= 16 rows per launch, up to 4 chunks; mid-size M: split-k slabs (ticket area + B*KS*M*N fp32, KS = 8 here)
    assert L.bd_gemm_workspace_bytes(64, 1, 4096, 4096) > 0 and L.bd_gemm_workspace_bytes(65, 1, 4096, 4096) == 0
    assert L.bd_gemm_workspace_bytes(1, 64, 4096, 4096) == 65536 + 8 * 64 * 4096 * 4
    assert L.bd_gemm_workspace_bytes(1, 1024, 4096, 4096) == 0
    # decode attention: ticket area + one (acc[128], max, sum) partial per (tenant, query head, key-range split), sized for the LARGEST split count
    # the run-time rule can pick (16), a function of the geometry only; caches shorter than 256 positions run unsplit and need none
    assert L.bd_srv_decode_attention_workspace_bytes(6, 32, 8, 128, 576) == 16384 + 6 * 32 * 16 * 130 * 4
    assert L.bd_srv_decode_attention_workspace_bytes(1, 32, 32, 128, 4096) == 16384 + 32 * 16 * 130 * 4
    assert L.bd_srv_decode_attention_workspace_bytes(6, 32, 8, 128, 128) == 0


def test_python_surface_matches_reference_names_and_signatures():
    import bitdelta_amd.binary_gemm_kernel as k
    import bitdelta_amd.diff as d
    import bitdelta_amd.serving as s

    def params(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values()
                if p.kind is not inspect.Parameter.KEYWORD_ONLY]
    E = inspect.Parameter.empty
    # reference: bitdelta/binary_gemm_kernel.py:6, :34, :153, :297
    assert params(k.pack) == [("x", E), ("n_bits", 32)]
    assert params(k.unpack) == [("x", E), ("n_bits", 32)]
    assert params(k.binary_matmul) == [("a", E), ("b", E), ("n_bits", 32), ("activation", "")]
    assert params(k.binary_bmm) == [("a", E), ("b", E), ("n_bits", 32), ("activation", "")]
    # reference: bitdelta/diff.py:9, :41, :66, :81, :108
    assert params(d.BinaryDiff.__init__)[1:] == [("base", E), ("finetune", E)]
    assert params(d.compress_diff) == [("base_model", E), ("finetuned_model", E), ("finetuned_compressed_model", E)]
    assert params(d.save_diff) == [("finetuned_compressed_model", E), ("save_dir", E)]
    assert params(d.load_diff) == [("model", E), ("diff_dir", E)]
    assert params(d.save_full_model) == [("base_model_name", E), ("finetuned_model_name", E), ("diff_dir", E),
                                         ("save_dir", E), ("device", E)]
    # reference: demo/demo_backend.py:62, :82, :107, :156, :170
    assert params(s.DiffCompressModule.__init__)[1:] == [("module", E), ("mask_list", E), ("coeff_list", E)]
    assert params(s.DataParallelModule.__init__)[1:] == [("module", E), ("weight_list", E)]
    assert params(s.register_diff_compress) == [("model", E), ("checkpoint_list", E)]
    for name in ("pack", "unpack", "binary_bmm"):                # `from bitdelta.diff import ...` re-exports these
        assert hasattr(d, name)


def test_reference_asserts_fire_before_any_device_work():
    import bitdelta_amd as bd
    a = torch.zeros(2, 4, 64, dtype=torch.float16)
    b = torch.zeros(2, 2, 8, dtype=torch.int32)
    with pytest.raises(AssertionError, match="3D"):
        bd.binary_bmm(a[0], b)
    with pytest.raises(AssertionError, match="Incompatible dimensions"):
        bd.binary_bmm(a, b[:, :1])
    with pytest.raises(AssertionError, match="batch"):
        bd.binary_bmm(a, b[:1])
    with pytest.raises(AssertionError, match="contiguous"):
        bd.binary_bmm(a.transpose(1, 2).contiguous().transpose(1, 2), b)
    with pytest.raises(AssertionError, match="divisible"):
        bd.pack(torch.zeros(33, 4, dtype=torch.bool))
    with pytest.raises(AssertionError, match="Incompatible"):
        bd.binary_matmul(a[0], b[0, :1])


def test_no_cpu_fallback():
    """Valid CPU inputs must raise, not compute: the product path exists on the GPU only."""
    import bitdelta_amd as bd
    from bitdelta_amd._lib import BitDeltaHipError
    with pytest.raises(BitDeltaHipError):
        bd.pack(torch.zeros(32, 4, dtype=torch.bool))
    with pytest.raises(BitDeltaHipError):
        bd.unpack(torch.zeros(1, 4, dtype=torch.int32))
    with pytest.raises(BitDeltaHipError):
        bd.binary_bmm(torch.zeros(1, 4, 64, dtype=torch.float16), torch.zeros(1, 2, 8, dtype=torch.int32))
    with pytest.raises(BitDeltaHipError):
        bd.BinaryDiff(torch.zeros(8, 32, dtype=torch.bfloat16), torch.zeros(8, 32, dtype=torch.bfloat16))


def test_round6_entries_validate_their_arguments_before_any_device_work():
    """bd_srv_rope_kv_append / bd_binary_linear_residual_norm return their error codes from the host-side checks (no launch, no GPU needed):
    geometry, alignment, the cache bound pos0 + S <= Lc, decode-size rows, rows wider than the norm kernels take"""
    import ctypes
    from bitdelta_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(1 << 16)
    base = (ctypes.addressof(buf) + 255) & ~255                # an aligned, non-null host address: never dereferenced by the checks
    BAD_SHAPE, BAD_DTYPE, NULL = (L.bd_srv_rope(None, None, None, -1, 1, 128, 0, 1, 0, 0, None),
                                  L.bd_srv_rope(None, None, None, 1, 1, 128, 0, 1, 0, 7, None), L.bd_srv_rope(None, None, None, 1, 1, 128, 0, 1, 0, 0, None))
    assert len({BAD_SHAPE, BAD_DTYPE, NULL, 0}) == 4
    rk = L.bd_srv_rope_kv_append
    W = (8 + 2 * 2) * 128
    assert rk(base, base, base, base, base, 0, 64, 8, 2, 128, W, 64, 0, 1, None) == 0                  # empty request
    assert rk(None, base, base, base, base, 2, 64, 8, 2, 128, W, 64, 0, 1, None) == NULL
    assert rk(base, base, base, base, base, 2, 64, 8, 2, 64, W, 64, 0, 1, None) == BAD_SHAPE           # head_dim != 128
    assert rk(base, base, base, base, base, 2, 64, 8, 2, 128, W - 128, 64, 0, 1, None) == BAD_SHAPE    # rows narrower than q|k|v
    assert rk(base, base, base, base, base, 2, 64, 8, 2, 128, W, 64, 1, 1, None) == BAD_SHAPE          # pos0 + S > Lc
    assert rk(base + 2, base, base, base, base, 2, 64, 8, 2, 128, W, 64, 0, 1, None) == BAD_SHAPE      # unaligned
    assert rk(base, base, base, base, base, 2, 64, 8, 2, 128, W, 64, 0, 2, None) == BAD_DTYPE          # fp32
    rn = L.bd_binary_linear_residual_norm
    def call(B=2, M=64, N=4096, K=512, dtype=1, nw=base, h=base, sYm=None, sYb=None):
        sYm = N if sYm is None else sYm
        sYb = M * sYm if sYb is None else sYb
        return rn(base, base, base, base, base, B, M, N, K, M * K, K, K, (K // 32) * N, 1, 1, sYb, sYm, dtype, nw, N, 1e-5, h, M * N, N, None, 0, None)
    assert call(B=0) == 0
    assert call(dtype=2) == BAD_DTYPE
    assert call(nw=None) == NULL and call(h=None) == NULL
    assert call(M=1) == BAD_SHAPE and call(M=16) == BAD_SHAPE                                          # decode rows: the hand-off entry's business
    assert call(N=8200) == BAD_SHAPE and call(N=16384) == BAD_SHAPE                                    # wider than the norm kernels' rows
    assert call(sYm=4100) == BAD_SHAPE and call(sYb=64 * 4096 + 8) == BAD_SHAPE                        # ragged / gapped residual rows
    sb, se = L.bd_srv_step_begin, L.bd_srv_step_end
    assert sb(base, 0, 4096, base, base, 4096, base, 64, base, 0, 32000, 4096, None) == 0               # no tenants
    assert sb(None, 0, 4096, base, base, 4096, base, 64, base, 2, 32000, 4096, None) == NULL
    assert sb(base, 0, 4096, base, base, 4096, base, 64, base, 2, 32000, 4100, None) == BAD_SHAPE       # H % 8
    assert sb(base, 0, 4000, base, base, 4096, base, 64, base, 2, 32000, 4096, None) == BAD_SHAPE       # table rows shorter than H
    assert se(base, 32000, 32000, base, base, 16, 16, base, 1, base, base, base, base, 0, 1, None) == 0
    assert se(base, 32000, 32000, base, base, 16, 16, base, 1, base, base, base, None, 2, 1, None) == NULL      # no ticket word
    assert se(base, 32000, 32004, base, base, 16, 16, base, 1, base, base, base, base, 2, 1, None) == BAD_SHAPE   # V % 8
    assert se(base, 32000, 32000, base, base, 8, 16, base, 1, base, base, base, base, 2, 1, None) == BAD_SHAPE    # out rows shorter than out_cap
    assert se(base, 32000, 32000, base, base, 16, 16, base, 1, base, base, base, base, 2, 2, None) == BAD_DTYPE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bitdelta_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m.split(".")[0] == "oracle" for m in mods), path
            if f.endswith((".h", ".hip", ".cpp", ".py")):
                assert "bd_oracle" not in open(path, errors="ignore").read(), path


def test_build_gate_is_a_content_hash_of_every_source(tmp_path, monkeypatch):
    """Editing ANY source the library is compiled from -- including the header of the headline fused kernel, which round 1's
    hand-kept dependency list missed -- must make the build stale; restoring the bytes must make it fresh again (content, not mtime)."""
    import shutil
    from bitdelta_amd import build as b
    names = {os.path.basename(f) for f in b.sources()}
    for must in ("bd_api.hip", "bd_gemm_fx.h", "bd_gemm_pf.h", "bd_gemv_stream.h", "bd_serving.h", "bitdelta_hip.h"):
        assert must in names, must
    # work on a copy of the package's source tree so the real stamp / sources are never touched
    root = tmp_path / "pkg"
    shutil.copytree(os.path.join(b.HERE, "csrc"), root / "bitdelta_amd" / "csrc")
    shutil.copytree(os.path.join(os.path.dirname(b.HERE), "include"), root / "include")
    os.makedirs(root / "bitdelta_amd" / "lib")
    monkeypatch.setattr(b, "HERE", str(root / "bitdelta_amd"))
    monkeypatch.setattr(b, "OUT", str(root / "bitdelta_amd" / "lib" / "libbitdelta_hip.so"))
    monkeypatch.setattr(b, "STAMP", str(root / "bitdelta_amd" / "lib" / "libbitdelta_hip.so.srchash"))
    assert b.needs_build()                                   # no library yet
    open(b.OUT, "wb").write(b"\0")
    open(b.STAMP, "w").write(b.source_hash() + "\n")
    assert not b.needs_build()
    hdr = root / "bitdelta_amd" / "csrc" / "bd_gemm_fx.h"
    orig = hdr.read_bytes()
    hdr.write_bytes(orig + b"\n// touched\n")
    assert b.needs_build()                                   # a header edit is seen
    hdr.write_bytes(orig)
    os.utime(hdr, None)                                      # newer mtime, same bytes
    assert not b.needs_build()
    api = root / "include" / "bitdelta_hip.h"
    api.write_bytes(api.read_bytes() + b"\n")
    assert b.needs_build()


def test_variant_table_has_an_environment_override():
    """SURVEY.md section 5 build note: the shape -> variant table can be overridden from the environment (BD_GEMM_VARIANT), not only through
    the bd_set_gemm_variant hook -- checked on the source (the library reads it once per thread) and on the header that documents it."""
    api = open(os.path.join(ROOT, "bitdelta_amd", "csrc", "bd_api.hip")).read()
    assert 'env_int("BD_GEMM_VARIANT", -1)' in api and 'env_int("BD_TAIL_SPLIT", 1)' in api
    assert "BD_GEMM_VARIANT" in open(os.path.join(ROOT, "include", "bitdelta_hip_test.h")).read()
