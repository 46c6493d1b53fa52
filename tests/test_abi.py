"""The C-ABI library loads without a GPU and exports every symbol include/llada_hip.h declares;
argument validation (which runs before any HIP call) reports errors the way the header says."""
import ctypes as C
import os
import re

from llada_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "llada_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(llada_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 15
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.llada_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "llada_hip.h")).read()
    assert f"#define LLADA_ABI_VERSION {_lib.ABI_VERSION}\n" in header


def test_struct_layouts_match_header():
    # llada_config: 5 x int32 + float; llada_remask_args: field order of the header
    assert C.sizeof(_lib.Config) == 24
    assert [f[0] for f in _lib.RemaskArgs._fields_][:6] == ["logits", "uncond_logits", "cfg_scale", "row_pos", "n_rows", "vocab"]
    assert [f[0] for f in _lib.RemaskArgs._fields_][-4:] == ["x0", "confidence", "workspace", "workspace_bytes"]
    assert C.sizeof(_lib.RemaskArgs) == 160 and _lib.RemaskArgs.workspace.offset == 144
    assert C.sizeof(_lib.Weights) == 8 + 9 * 8


def test_argument_errors_without_gpu(lib):
    assert lib.llada_gemm_bf16(None, None, None, 1, 8, 64, 64, 8, 0, None, 0, None, None, 0, None, 0, None) == _lib.ERR_INVALID
    assert b"null" in lib.llada_last_error()
    one = C.c_void_p(16)
    assert lib.llada_gemm_bf16(one, one, one, 4, 8, 100, 104, 8, 0, None, 0, None, None, 0, None, 0, None) == _lib.ERR_UNSUPPORTED
    args = _lib.RemaskArgs()
    assert lib.llada_remask_scores(C.byref(args), None) == _lib.ERR_INVALID
    assert lib.llada_forward_workspace_bytes(None, 1, 1, 1) == 0
    cfg = _lib.Config(4096, 32, 32, 12288, 126464, 1e-5)
    need = lib.llada_forward_workspace_bytes(C.byref(cfg), 8, 384, 256)
    M = 8 * 384
    assert need >= M * 2 * (3 * 4096 + 3 * 4096 + 3 * 12288)


def test_check_maps_status_to_reference_exceptions(lib):
    import pytest
    lib.llada_gemm_bf16(None, None, None, 1, 8, 64, 64, 8, 0, None, 0, None, None, 0, None, 0, None)
    with pytest.raises(AssertionError):
        _lib.check(_lib.ERR_INVALID)
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.ERR_UNSUPPORTED)
    with pytest.raises(_lib.LLaDAHipError):
        _lib.check(_lib.ERR_HIP)
