"""ctypes binding of the C-ABI in include/caduceus_hip.h (libcaduceus_hip.so).

The product path has NO fallback: if the HIP library has not been built, or a tensor is not on the GPU, the ops raise.
`use_library_for_testing()` exists only so that the test-suite can inject the host-emulator build of the *same kernel
sources* (tests/emu) on a machine without a GPU; nothing in this package calls it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# CADUCEUS_AMD_LIB selects another gfx950 build of the same sources (tuning variants); never a different backend
LIB_PATH = os.environ.get("CADUCEUS_AMD_LIB") or os.path.join(HERE, "libcaduceus_hip.so")

CAD_F32, CAD_BF16 = 0, 1
PROF_KINDS = ("scan_fwd", "scan_bwd", "conv_fwd", "conv_bwd", "add_norm_fwd", "add_norm_bwd", "embed", "lm_head", "proj")

_p = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_f = C.c_float


class EmbedArgs(C.Structure):
    _fields_ = [("ids", _p), ("comp", _p), ("weight", _p), ("out", _p), ("B", _i64), ("L", _i64), ("D", _i), ("V", _i),
                ("n_strands", _i), ("w_dtype", _i), ("out_dtype", _i)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [("ids", _p), ("comp", _p), ("dout", _p), ("dweight", _p), ("B", _i64), ("L", _i64), ("D", _i),
                ("V", _i), ("n_strands", _i), ("dout_dtype", _i)]


class AddNormArgs(C.Structure):
    _fields_ = [("x", _p), ("residual_in", _p), ("weight", _p), ("bias", _p), ("y", _p), ("residual_out", _p),
                ("rstd", _p), ("mean", _p), ("rows_per_strand", _i64), ("n_strands", _i), ("D", _i), ("eps", _f),
                ("is_rms", _i), ("swap_flip", _i), ("x_dtype", _i), ("y_dtype", _i), ("y_fp8", _p), ("y_scale", _p)]


class AddNormBwdArgs(C.Structure):
    _fields_ = [("dy", _p), ("dres_out", _p), ("sum_saved", _p), ("rstd", _p), ("mean", _p), ("weight", _p),
                ("dx", _p), ("dres_in", _p), ("dweight", _p), ("dbias", _p), ("rows_per_strand", _i64),
                ("n_strands", _i), ("D", _i), ("is_rms", _i), ("swap_flip", _i), ("x_dtype", _i), ("y_dtype", _i)]


class Conv1dArgs(C.Structure):
    _fields_ = [("x", _p), ("w", _p), ("bias", _p), ("out", _p), ("SB", _i64), ("L", _i64), ("split", _i64),
                ("E", _i), ("K", _i), ("rev_lo", _i), ("rev_hi", _i), ("dtype", _i)]


class ReduceJob(C.Structure):
    _fields_ = [("src", _p), ("dst", _p)]


class ProjTmArgs(C.Structure):
    _fields_ = [("W", _p), ("X", _p), ("X2", _p), ("out", _p), ("T", _i64), ("M", _i), ("K", _i), ("ldw", _i64), ("ldx", _i64),
                ("ldo", _i64)]


class GemmStreamArgs(C.Structure):
    _fields_ = [("A", _p), ("B", _p), ("out", _p), ("R", _i64), ("C", _i64), ("K", _i64), ("lda", _i64), ("ldb", _i64), ("ldo", _i64),
                ("nslices", _i), ("mode", _i), ("col_fastest", _i)]


class FoldF32Job(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("n", _i64), ("stride", _i64), ("stride2", _i64), ("nparts", _i), ("nparts2", _i)]


class GemmF32Args(C.Structure):
    _fields_ = [("A", _p), ("B", _p), ("D", _p), ("addend", _p), ("M", _i64), ("N", _i64), ("K", _i64),
                ("a_rs", _i64), ("a_cs", _i64), ("b_rs", _i64), ("b_cs", _i64), ("d_rs", _i64), ("d_cs", _i64),
                ("batch", _i64), ("a_bs", _i64), ("b_bs", _i64), ("d_bs", _i64)]


class Conv1dBwdArgs(C.Structure):
    _fields_ = [("x", _p), ("w", _p), ("bias", _p), ("dout", _p), ("dx", _p), ("dw", _p), ("dbias", _p),
                ("SB", _i64), ("L", _i64), ("split", _i64), ("E", _i), ("K", _i), ("rev_lo", _i), ("rev_hi", _i),
                ("dtype", _i), ("accumulate", _i)]


class ScanArgs(C.Structure):
    _fields_ = [("u", _p), ("delta", _p), ("A", _p), ("Bm", _p), ("Cm", _p), ("D", _p), ("z", _p),
                ("delta_bias", _p), ("out", _p), ("chunk_state", _p), ("SB", _i64), ("L", _i64), ("split", _i64),
                ("E", _i), ("N", _i), ("rev_lo", _i), ("rev_hi", _i), ("dtype", _i), ("h0", _p), ("hT", _p),
                ("sum_dt", _p), ("delta_is_dt", _i), ("map_only", _i)]


class ScanBwdArgs(C.Structure):
    _fields_ = [("u", _p), ("delta", _p), ("A", _p), ("Bm", _p), ("Cm", _p), ("D", _p), ("z", _p),
                ("delta_bias", _p), ("dout", _p), ("out", _p), ("chunk_state", _p), ("du", _p), ("ddelta", _p), ("dz", _p),
                ("dA", _p), ("dB", _p), ("dC", _p), ("dD", _p), ("ddelta_bias", _p), ("SB", _i64), ("L", _i64),
                ("split", _i64), ("E", _i), ("N", _i), ("rev_lo", _i), ("rev_hi", _i), ("dtype", _i),
                ("n_partials", _i), ("dhT", _p), ("dh0", _p), ("out2", _p), ("gate_fix_list", _p), ("gate_fix_count", _p),
                ("gate_fix_dz", _p), ("delta_is_dt", _i), ("carry_only", _i), ("fold_counters", _p)]


class FoldArgs(C.Structure):
    _fields_ = [("dB_slots", _p), ("dC_slots", _p), ("dB", _p), ("dC", _p), ("counters", _p), ("abort_from", _p), ("SB", _i64),
                ("L", _i64), ("split", _i64), ("N", _i), ("n_partials", _i), ("rev_lo", _i), ("rev_hi", _i), ("dtype", _i)]


class MlmArgs(C.Structure):
    _fields_ = [("bases", _p), ("rc_flags", _p), ("lengths", _p), ("input_ids", _p), ("labels", _p), ("B", _i64),
                ("L", _i64), ("ld_bases", _i64), ("seed", C.c_uint64), ("offset", C.c_uint64), ("thr_mask", C.c_uint32),
                ("pad_id", _i), ("mask_id", _i), ("unk_id", _i), ("n_id", _i), ("vocab", _i), ("base_ids", _i * 4), ("row_ids", _p)]


class ProjArgs(C.Structure):
    _fields_ = [("W", _p), ("X", _p), ("out", _p), ("T", _i64), ("M", _i), ("K", _i), ("ldw", _i64), ("ldx", _i64),
                ("ldo", _i64), ("acc", _p), ("ldacc", _i64), ("bias", _p), ("act", _i), ("wg_y", _p), ("ld_wg_y", _i64),
                ("wg_partials", _p)]


class QuantFp8Args(C.Structure):
    _fields_ = [("x", _p), ("q", _p), ("scale", _p), ("T", _i64), ("K", _i), ("ldx", _i64), ("ldq", _i64), ("dtype", _i)]


class ProjFp8Args(C.Structure):
    _fields_ = [("Wq", _p), ("Xq", _p), ("sw", _p), ("sx", _p), ("out", _p), ("T", _i64), ("M", _i), ("K", _i),
                ("ldw", _i64), ("ldx", _i64), ("ldo", _i64)]


class LmHeadArgs(C.Structure):
    _fields_ = [("hidden", _p), ("weight", _p), ("comp", _p), ("labels", _p), ("logits", _p), ("loss_sum", _p),
                ("count", _p), ("rows", _i64), ("D", _i), ("V", _i), ("n_strands", _i), ("ignore_index", _i64),
                ("dtype", _i), ("block_partials", _p)]


class LmHeadBwdArgs(C.Structure):
    _fields_ = [("hidden", _p), ("weight", _p), ("comp", _p), ("labels", _p), ("logits", _p), ("dlogits", _p),
                ("loss_scale", _p), ("dhidden", _p), ("dw_partials", _p), ("rows", _i64), ("D", _i), ("V", _i),
                ("n_strands", _i), ("ignore_index", _i64), ("dtype", _i), ("ld", _i64)]


# every exported symbol of include/caduceus_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "cad_version": (C.c_char_p, []),
    "cad_status_string": (C.c_char_p, [_i]),
    "cad_is_device_build": (_i, []),
    "cad_embed_fwd": (_i, [C.POINTER(EmbedArgs), _p]),
    "cad_embed_bwd": (_i, [C.POINTER(EmbedBwdArgs), _p]),
    "cad_add_norm_fwd": (_i, [C.POINTER(AddNormArgs), _p]),
    "cad_add_norm_bwd": (_i, [C.POINTER(AddNormBwdArgs), _p]),
    "cad_conv1d_fwd": (_i, [C.POINTER(Conv1dArgs), _p]),
    "cad_conv1d_bwd": (_i, [C.POINTER(Conv1dBwdArgs), _p]),
    "cad_conv1d_fwd_multi": (_i, [C.POINTER(Conv1dArgs), _i, _p]),
    "cad_conv1d_bwd_multi": (_i, [C.POINTER(Conv1dBwdArgs), _i, _p]),
    "cad_scan_fwd": (_i, [C.POINTER(ScanArgs), _p]),
    "cad_scan_chunk_len": (_i64, []),
    "cad_scan_state_floats": (_i64, [_i, _i64, _i64, _i]),
    "cad_scan_bwd": (_i, [C.POINTER(ScanBwdArgs), _p]),
    "cad_scan_fwd_multi": (_i, [C.POINTER(ScanArgs), _i, _p]),
    "cad_scan_bwd_multi": (_i, [C.POINTER(ScanBwdArgs), _i, _p]),
    "cad_reduce_partials": (_i, [_p, _i, _i64, _p, _i, _p]),
    "cad_reduce_partials_multi": (_i, [C.POINTER(ReduceJob), _i, _i, _i64, _i, _p]),
    "cad_scan_bwd_partials": (_i, [_i]),
    "cad_fold_partials_stream": (_i, [C.POINTER(FoldArgs), _i, _i, _p]),
    "cad_fold_stream_supported": (_i, [_i, _i, _i64, _i]),
    "cad_scan_bwd_chunk_len": (_i64, []),
    "cad_scan_bwd_fold_counter_ints": (_i64, [_i64, _i64]),
    "cad_stream_probe": (_i, [_p, _p, _p, _p, _i64]),
    "cad_scan_bwd_gate_fix": (_i, [C.POINTER(ScanBwdArgs), _i, _p]),
    "cad_scan_gate_fix_entries": (_i64, [_i, _i64, _i64]),
    "cad_proj_wxT": (_i, [C.POINTER(ProjArgs), _p]),
    "cad_proj_supported": (_i, [_i]),
    "cad_proj_wx": (_i, [C.POINTER(ProjArgs), _p]),
    "cad_proj_wx_supported": (_i, [_i, _i64]),
    "cad_proj_wx_thin_supported": (_i, [_i, _i, _i64]),
    "cad_proj_wx_wgrad": (_i, [C.POINTER(ProjArgs), _p]),
    "cad_proj_wx_wgrad_supported": (_i, [_i, _i, _i64]),
    "cad_proj_wx_wgrad_partials": (_i, [_i64]),
    "cad_proj_wgrad_only_supported": (_i, [_i, _i, _i64]),
    "cad_proj_xTw": (_i, [C.POINTER(ProjTmArgs), _p]),
    "cad_proj_xTw_supported": (_i, [_i, _i, _i64]),
    "cad_gemm_stream": (_i, [C.POINTER(GemmStreamArgs), _p]),
    "cad_gemm_stream_supported": (_i, [_i64, _i64, _i64, _i]),
    "cad_fold_f32_multi": (_i, [C.POINTER(FoldF32Job), _i, _p]),
    "cad_gemm_f32": (_i, [C.POINTER(GemmF32Args), _p]),
    "cad_quant_rows_fp8": (_i, [C.POINTER(QuantFp8Args), _p]),
    "cad_proj_wxT_fp8": (_i, [C.POINTER(ProjFp8Args), _p]),
    "cad_proj_fp8_supported": (_i, [_i]),
    "cad_lm_head_fwd": (_i, [C.POINTER(LmHeadArgs), _p]),
    "cad_lm_head_partials": (_i64, [_i64]),
    "cad_lm_head_bwd": (_i, [C.POINTER(LmHeadBwdArgs), _p]),
    "cad_lm_head_bwd_supported": (_i, [_i, _i]),
    "cad_lm_head_bwd_partials": (_i64, [_i64]),
    "cad_tokenize_mlm": (_i, [C.POINTER(MlmArgs), _p]),
    "cad_mlm_threshold": (C.c_uint32, [C.c_double]),
    "cad_hg38_interval": (_i, [_i64, _i64, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "cad_fasta_open": (_i, [C.c_char_p, C.POINTER(_p)]),
    "cad_fasta_close": (_i, [_p]),
    "cad_fasta_num_seqs": (_i64, [_p]),
    "cad_fasta_seq_name": (C.c_char_p, [_p, _i64]),
    "cad_fasta_seq_len": (_i64, [_p, _i64]),
    "cad_fasta_find": (_i64, [_p, C.c_char_p]),
    "cad_fasta_fetch": (_i, [_p, _i64, _i64, _i64, _p]),
    "cad_prof_enable": (_i, [_i]),
    "cad_prof_enable_kinds": (_i, [C.c_uint]),
    "cad_prof_reset": (_i, []),
    "cad_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_i64)]),
}

_lock = threading.Lock()
_lib = None
_is_device = None


def _bind(path: str):
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    # a timing build (-DSC_WHATIF / -DSC_TIMING: kernels cut down for measurements, WRONG results by construction) says so in
    # cad_version() (csrc/api.hip) and is never loaded as the product: only the measurement tools opt in
    ver = lib.cad_version().decode()
    if "TIMING-BUILD" in ver and os.environ.get("CADUCEUS_AMD_ALLOW_TIMING_BUILD") != "1":
        raise RuntimeError(f"{path} is a timing build ({ver}): its kernels produce wrong results by construction. It is loaded only "
                           "with CADUCEUS_AMD_ALLOW_TIMING_BUILD=1 (bench.py's floor worker, tools/ab_layer.sh).")
    return lib


def get_lib():
    """The kernel library; raises loudly if it has not been built (no CPU fallback exists)."""
    global _lib, _is_device
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} not found: the HIP kernels are not built. Run `python -m caduceus_amd._build` "
                        "(needs hipcc; cross-compiles gfx950 without a GPU). caduceus_amd has no CPU fallback.")
                _lib = _bind(LIB_PATH)
                _is_device = bool(_lib.cad_is_device_build())
    return _lib


def is_device_build() -> bool:
    get_lib()
    return bool(_is_device)


def use_library_for_testing(path: str | None):
    """TEST HOOK: load an alternative build of the same C-ABI (the host emulator in tests/emu), or reset with None."""
    global _lib, _is_device
    with _lock:
        if path is None:
            _lib, _is_device = None, None
        else:
            _lib = _bind(path)
            _is_device = bool(_lib.cad_is_device_build())


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed: {get_lib().cad_status_string(status).decode()} (status {status})")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return CAD_F32
    if dt == torch.bfloat16:
        return CAD_BF16
    raise TypeError(f"caduceus_amd kernels support float32 and bfloat16 activations, got {dt}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_and_check(*tensors, contiguous=True):
    """Validates device placement against the loaded library flavour and returns the launch stream handle.
    contiguous=False: the caller passes explicit row strides to the kernel (token-major column slices)."""
    dev_build = is_device_build()
    ref = None
    for t in tensors:
        if t is None:
            continue
        if dev_build and not t.is_cuda:
            raise RuntimeError("caduceus_amd: tensors must live on the GPU (no CPU fallback); got a CPU tensor")
        if not dev_build and t.is_cuda:
            raise RuntimeError("caduceus_amd: host-emulator library loaded but a GPU tensor was passed")
        if ref is None:
            ref = t
        elif t.device != ref.device:
            raise RuntimeError("caduceus_amd: all tensors of one op must be on the same device")
        if contiguous and not t.is_contiguous():
            raise RuntimeError("caduceus_amd: kernel arguments must be contiguous")
    if dev_build:
        return C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)
    return None


def version() -> str:
    """cad_version() of the loaded library, e.g. "caduceus_amd 0.1.0 (hip gfx950) src 3f2a9c0d1b7e" (hash of the kernel sources)."""
    return get_lib().cad_version().decode()


def prof_enable(on: bool, kinds=None):
    """Kernel timer on / off.  kinds: names from PROF_KINDS to time only those (every timed launch costs two event records on its
    stream: ~10 us of idle queue per launch, 5 % of a training step with all kinds on)."""
    if on and kinds is not None:
        mask = 0
        for k in kinds:
            mask |= 1 << PROF_KINDS.index(k)
        check(get_lib().cad_prof_enable_kinds(mask), "cad_prof_enable_kinds")
    else:
        check(get_lib().cad_prof_enable(int(on)), "cad_prof_enable")


def prof_reset():
    check(get_lib().cad_prof_reset(), "cad_prof_reset")


def prof_read():
    """{kind: (total_ms, launches)} for every timed kernel kind (synchronises the recorded events)."""
    out = {}
    for k, name in enumerate(PROF_KINDS):
        ms, n = C.c_double(0), _i64(0)
        check(get_lib().cad_prof_read(k, C.byref(ms), C.byref(n)), "cad_prof_read")
        out[name] = (ms.value, n.value)
    return out
