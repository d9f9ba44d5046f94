"""The C-ABI boundary: the header, the ctypes mirror and the built library must agree (no compute, CPU only)."""
import ctypes
import os
import re
import subprocess

import pytest

from caduceus_amd import _lib
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "caduceus_hip.h")


def _header_text():
    txt = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def _header_functions():
    return set(re.findall(r"\b(cad_[A-Za-z0-9_]+)\s*\(", _header_text()))


def _header_structs():
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(cad_[a-z0-9_]+);", _header_text(), flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        fields = [re.sub(r"\[.*\]", "", f) for f in fields]  # array members
        out[name] = fields
    return out


def test_every_declared_symbol_is_bound_and_exported():
    funcs = _header_functions()
    assert funcs == set(_lib.SYMBOLS), funcs ^ set(_lib.SYMBOLS)
    if not os.path.exists(_lib.LIB_PATH):
        from caduceus_amd import _build
        _build.build_hip()
    lib = ctypes.CDLL(_lib.LIB_PATH)  # loads without a GPU (no compute calls)
    for name in funcs:
        assert hasattr(lib, name), f"{name} not exported by libcaduceus_hip.so"
    lib.cad_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cad_version()
    lib.cad_is_device_build.restype = ctypes.c_int
    assert lib.cad_is_device_build() == 1
    lib.cad_status_string.restype = ctypes.c_char_p
    assert b"bad argument" in lib.cad_status_string(1)


def test_ctypes_structs_mirror_header():
    hs = _header_structs()
    mirror = {"cad_embed_args": _lib.EmbedArgs, "cad_embed_bwd_args": _lib.EmbedBwdArgs,
              "cad_add_norm_args": _lib.AddNormArgs, "cad_add_norm_bwd_args": _lib.AddNormBwdArgs,
              "cad_conv1d_args": _lib.Conv1dArgs, "cad_conv1d_bwd_args": _lib.Conv1dBwdArgs,
              "cad_scan_args": _lib.ScanArgs, "cad_scan_bwd_args": _lib.ScanBwdArgs,
              "cad_lm_head_args": _lib.LmHeadArgs, "cad_lm_head_bwd_args": _lib.LmHeadBwdArgs, "cad_mlm_args": _lib.MlmArgs,
              "cad_proj_args": _lib.ProjArgs, "cad_quant_fp8_args": _lib.QuantFp8Args,
              "cad_proj_fp8_args": _lib.ProjFp8Args, "cad_proj_tm_args": _lib.ProjTmArgs, "cad_reduce_job": _lib.ReduceJob,
              "cad_gemm_stream_args": _lib.GemmStreamArgs, "cad_fold_args": _lib.FoldArgs, "cad_fold_f32_job": _lib.FoldF32Job,
              "cad_gemm_f32_args": _lib.GemmF32Args}
    assert set(hs) == set(mirror)
    for name, cls in mirror.items():
        assert [f[0] for f in cls._fields_] == hs[name], name


def test_struct_sizes_match_compiler(tmp_path):
    """sizeof() of every argument struct as seen by a C compiler equals the ctypes mirror (padding/ordering)."""
    names = sorted(_header_structs())
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "caduceus_hip.h"\nint main(){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    mirror = {"cad_embed_args": _lib.EmbedArgs, "cad_embed_bwd_args": _lib.EmbedBwdArgs,
              "cad_add_norm_args": _lib.AddNormArgs, "cad_add_norm_bwd_args": _lib.AddNormBwdArgs,
              "cad_conv1d_args": _lib.Conv1dArgs, "cad_conv1d_bwd_args": _lib.Conv1dBwdArgs,
              "cad_scan_args": _lib.ScanArgs, "cad_scan_bwd_args": _lib.ScanBwdArgs,
              "cad_lm_head_args": _lib.LmHeadArgs, "cad_lm_head_bwd_args": _lib.LmHeadBwdArgs, "cad_mlm_args": _lib.MlmArgs,
              "cad_proj_args": _lib.ProjArgs, "cad_quant_fp8_args": _lib.QuantFp8Args,
              "cad_proj_fp8_args": _lib.ProjFp8Args, "cad_proj_tm_args": _lib.ProjTmArgs, "cad_reduce_job": _lib.ReduceJob,
              "cad_gemm_stream_args": _lib.GemmStreamArgs, "cad_fold_args": _lib.FoldArgs, "cad_fold_f32_job": _lib.FoldF32Job,
              "cad_gemm_f32_args": _lib.GemmF32Args}
    for n, cls in mirror.items():
        assert int(sizes[n]) == ctypes.sizeof(cls), n


def test_no_cpu_fallback_without_gpu():
    """Product path: with the real (device) library and CPU tensors every op must raise, never compute on the host."""
    import torch
    from caduceus_amd import ops
    _lib.use_library_for_testing(None)
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.selective_scan(torch.zeros(2, 1, 8), torch.zeros(2, 1, 8), -torch.ones(2, 4), torch.zeros(4, 1, 8),
                           torch.zeros(4, 1, 8), torch.ones(2), torch.zeros(2, 1, 8), torch.zeros(2), 1, 0, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.embed(torch.zeros(1, 4, dtype=torch.long), torch.zeros(16, 8), None, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    _lib.use_library_for_testing(None)
    with pytest.raises(RuntimeError, match="not found"):
        _lib.get_lib()
    monkeypatch.undo()
    _lib.use_library_for_testing(None)


def test_timing_builds_are_fenced(monkeypatch):
    """A library whose scan kernels were cut down for timing experiments (-DSC_WHATIF / -DSC_TIMING: wrong results by construction, e.g.
    build()'s libcaduceus_hip_floor.so) names itself in cad_version() and is refused by the loader unless the caller opts in; the
    product library carries neither marker."""
    from caduceus_amd import _build
    _build.build_hip()
    floor = _build.build_floor()
    monkeypatch.delenv("CADUCEUS_AMD_ALLOW_TIMING_BUILD", raising=False)
    with pytest.raises(RuntimeError, match="timing build"):
        _lib._bind(floor)
    monkeypatch.setenv("CADUCEUS_AMD_ALLOW_TIMING_BUILD", "1")
    ver = _lib._bind(floor).cad_version().decode()
    assert "TIMING-BUILD" in ver and "variant[SC_WHATIF=14434]" in ver, ver
    monkeypatch.delenv("CADUCEUS_AMD_ALLOW_TIMING_BUILD")
    ver = _lib._bind(_build.LIB).cad_version().decode()
    assert "TIMING-BUILD" not in ver and "variant" not in ver, ver
