"""Hand-scheduled forward/backward of one weight-tied BiMamba mixer over both RCPS strands (the production path:
`bidirectional=True, bidirectional_strategy="add", bidirectional_weight_tie=True, bias=False`).

engine.py composes the same computation from per-op autograd Functions (kept for every other configuration and as the
readable specification).  Scheduling the backward by hand removes what generic autograd cannot know
(profiles/r01_step_and_scan_v3_summary.txt: ~40 ms of copies / adds / duplicate GEMMs per 382 ms step):
  * with a tied out_proj, d(y_f) == d(y_r): ONE GEMM, written channel-major directly (no transposing copies);
  * the scan backward writes dz of parameter set f straight into the dxz buffer and set r's dz is added in place;
  * weight gradients (reductions over all T tokens with tiny outputs) run as strided-batch GEMMs over 64 K-chunks plus an
    fp32 sum, 3-5x faster than the un-split library GEMM;
  * the conv forward / backward of both parameter sets run as one launch each (x read once, dx = dx_f + dx_r written once); dB/dC partial sums are reduced straight into the rows of
    the x_proj gradient operand; du is folded into the x_proj backward GEMM (addmm).
All kernels are the C-ABI entry points of include/caduceus_hip.h.  Projections: in_proj, x_proj, dt_proj (+ bias + softplus), d(y),
d(dt_lr) + dW_dt (one pass, cad_proj_wx_wgrad) and the x_proj input gradient run on the library's own MFMA kernels (csrc/gemm.hip,
gemm_fp8.hip); out_proj forward, d(x2d) and the three remaining weight gradients are hipBLASLt through torch.
Scan launches with fewer workgroups than the GPU has CUs are L-split (ops.lsplit_factor).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import ops


def _conv_fwd2(x, params, split, dirs):
    """Both parameter sets' causal conv + SiLU of the same x in one launch (x is read once).  params: [(wf, bf)] * 2."""
    E, SB, Lq = x.shape
    n = len(params)
    args = (L.Conv1dArgs * n)()
    outs = []
    for i, (wf, bf) in enumerate(params):
        out = torch.empty_like(x)
        stream = L.stream_and_check(x, wf, bf, out)
        args[i] = L.Conv1dArgs(L.ptr(x), L.ptr(wf), L.ptr(bf), L.ptr(out), SB, Lq, split, E, wf.shape[1], dirs[i][0],
                               dirs[i][1], L.dtype_code(x.dtype))
        outs.append(out)
    L.check(L.get_lib().cad_conv1d_fwd_multi(args, n, stream), "cad_conv1d_fwd_multi")
    return outs


def _zeros_f32(shapes, device):
    """fp32 zero tensors of the given shapes carved out of ONE allocation (one fill kernel instead of one per tensor:
    the accumulated per-channel gradients of a layer are ten tiny tensors)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + (n + 3) // 4 * 4)  # 16-byte aligned views
    flat = torch.zeros((offs[-1],), dtype=torch.float32, device=device)
    return [flat[o:o + n].view(s) for o, n, s in zip(offs, sizes, shapes)]


def _conv_bwd2(x, params, douts, dx, split, dirs, bufs=None):
    """dx = sum over both parameter sets of their input gradients (written once), dw / dbias per set.
    bufs: optional pre-zeroed fp32 (dw, db) per set."""
    E, SB, Lq = x.shape
    n = len(params)
    args = (L.Conv1dBwdArgs * n)()
    res = []
    for i, (wf, bf) in enumerate(params):
        if bufs is not None:
            dw, db = bufs[i]
        else:
            dw = torch.zeros_like(wf)
            db = None if bf is None else torch.zeros_like(bf)
        stream = L.stream_and_check(x, wf, bf, douts[i], dx, dw, db)
        args[i] = L.Conv1dBwdArgs(L.ptr(x), L.ptr(wf), L.ptr(bf), L.ptr(douts[i]), L.ptr(dx), L.ptr(dw), L.ptr(db), SB, Lq,
                                  split, E, wf.shape[1], dirs[i][0], dirs[i][1], L.dtype_code(x.dtype), 0)
        res.append((dw, db))
    L.check(L.get_lib().cad_conv1d_bwd_multi(args, n, stream), "cad_conv1d_bwd_multi")
    return res


def _kchunks(T: int) -> int:
    """Number of K-chunks for the weight-gradient GEMMs (reduction over all T tokens, tiny outputs).  hipBLASLt does not
    split K by itself for these shapes (0.6 ms at 0.65 TB/s); as a strided-batch GEMM over 64 chunks plus an fp32 sum of
    the partial products the same gradient takes 0.1-0.2 ms (tools/gemm_bench.py), with the same rounding error."""
    n = 64
    while n > 1 and (T % n != 0 or T // n < 1024):
        n //= 2
    return n


def _wgrad_cm_cm(a_cm: torch.Tensor, b_cm: torch.Tensor) -> torch.Tensor:
    """a (M, T) @ b (N, T)^T with both operands channel-major (T contiguous) -> (M, N) fp32."""
    M, T = a_cm.shape
    if a_cm.dtype == torch.float32:  # the own fp32 matrix-core kernel cuts K itself (ops.mm_f32)
        return ops.mm_f32(a_cm, b_cm.t())
    n = _kchunks(T)
    if b_cm.shape[0] <= 16 and n >= 64 and T % 16 == 0:
        n = 16  # thin products (dW_dt): fewer, deeper chunks (tools/wgrad_sweep.py: 48 vs 55 us at T = 262144)
    if n == 1:
        return ops.mm(a_cm, b_cm.t()).float()
    Kc = T // n
    # (sum with fp32 accumulation straight from the bf16 partial products: no separate up-cast launch)
    return torch.sum(ops.bmm(a_cm.view(M, n, Kc).permute(1, 0, 2), b_cm.view(-1, n, Kc).permute(1, 2, 0)), dim=0,
                     dtype=torch.float32)


def _own_wgrad_chunked_ok(X: torch.Tensor, M: int) -> bool:
    """Reductions over ALL tokens with K = d_inner > 512 rows (d_model 512: configs[4]): the own weight-gradient kernel takes 512 rows of
    X per launch, so X is cut into row blocks -- the rows of dW are independent."""
    K, T = X.shape
    return K > 512 and K % 512 == 0 and ops.proj_wgrad_only_supported(X, M, 512, T)


def _own_wgrad_chunked(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """dW (M, K) fp32 = Y (M, T) @ X (K, T)^T with fp32 accumulation over all tokens (cad_proj_wx_wgrad's weight-gradient stage), K in
    blocks of 512 rows.  Replaces the K-split library bmm whose bf16-rounded partial products cost 3-4 % relative error on
    dW_x of a d_model 512 layer (found by tests/test_configs.py::test_config4_one_layer_d512_L262144_every_gradient_vs_oracle)."""
    K = X.shape[0]
    return torch.cat([ops.proj_wgrad_only(X[h:h + 512], Y) for h in range(0, K, 512)], dim=1)


def _wgrad_cm_tm(a_cm: torch.Tensor, b_tm: torch.Tensor) -> torch.Tensor:
    """a (M, T) channel-major @ b (T, N) token-major -> (M, N) fp32."""
    M, T = a_cm.shape
    if _OWN_GEMM and (b_tm.shape[1] <= 256 or _OWN_GEMM_D512):  # (one column tile: at d_model 512 the library's K-split was 12 % faster, profiles/r04_gemm_stream.txt)
        own = ops.wgrad_cm_tm(a_cm, b_tm)
        if own is not None:
            return own
    if a_cm.dtype == torch.float32:
        return ops.mm_f32(a_cm, b_tm)
    n = _kchunks(T)
    if n == 1:
        return ops.mm(a_cm, b_tm).float()
    Kc = T // n
    return torch.sum(ops.bmm(a_cm.view(M, n, Kc).permute(1, 0, 2), b_tm.view(n, Kc, -1)), dim=0, dtype=torch.float32)


# The two scans of a BiMamba layer share the gate z and the upstream gradient: set 0's backward kernel evaluates the gate
# gradient of both (cad_scan_bwd_args.out2).  CADUCEUS_AMD_SHARED_GATE=0 keeps one dz per set + an add (A/B switch).
_SHARED_GATE = os.environ.get("CADUCEUS_AMD_SHARED_GATE", "1") != "0"
# delta_bias + softplus in the epilogue of the dt_proj kernel (HBM-bound, idle VALU) instead of the scan prologues
_FUSED_SOFTPLUS = os.environ.get("CADUCEUS_AMD_FUSED_SOFTPLUS", "1") != "0"


# d(dt_lr) and dW_dt from one pass over d(delta) (cad_proj_wx_wgrad); CADUCEUS_AMD_FUSED_WGRAD=0 keeps the two-kernel path (A/B switch)
_FUSED_WGRAD = os.environ.get("CADUCEUS_AMD_FUSED_WGRAD", "1") != "0"
# dW_x = d(dbc) . xc^T on the weight-gradient stage of the same kernel (W == NULL) instead of the GEMM library + a partial sum
_OWN_DWX = os.environ.get("CADUCEUS_AMD_OWN_DWX", "1") != "0"
# out_proj forward on the own token-major-output MFMA kernel (cad_proj_xTw); CADUCEUS_AMD_OWN_OUT_PROJ=0: hipBLASLt on [y_f ; y_r]
_OWN_OUT_PROJ = os.environ.get("CADUCEUS_AMD_OWN_OUT_PROJ", "1") != "0"
# d(x2d), dW_in and dW_out -- the products whose two operands both stream -- on the own tiled MFMA kernel (cad_gemm_stream: fp32
# accumulation over ALL tokens for the weight gradients); CADUCEUS_AMD_OWN_GEMM=0: torch.mm / K-split bmm (hipBLASLt)
_OWN_GEMM = os.environ.get("CADUCEUS_AMD_OWN_GEMM", "1") != "0"
# ... also at d_model 512 (configs[4]: two 256-column tiles, the strided operand is walked twice).  The library products are 0.5 ms per
# layer faster there (31.75 vs 32.28 ms per layer, 535.2 vs 538.2 ms per step: profiles/r05_gemm_stream_d512.txt), i.e. 0.6 % of a step
# is the price of a configs[4] step without a library GEMM and with fp32-accumulated weight gradients; CADUCEUS_AMD_OWN_GEMM_D512=0
# takes torch.mm / the K-split bmm for these three products at d_model > 256
_OWN_GEMM_D512 = os.environ.get("CADUCEUS_AMD_OWN_GEMM_D512", "1") != "0"
# in_proj and d(y) at d_model 512 on the same tiled kernel (cad_gemm_stream, column tiles fastest) instead of the W-stationary cad_proj_wxT,
# which streams the token operand once per 128-row block of W (16 times at M = 2048); CADUCEUS_AMD_STREAM_PROJ_D512=0: cad_proj_wxT
_STREAM_PROJ_D512 = os.environ.get("CADUCEUS_AMD_STREAM_PROJ_D512", "1") != "0"
# A/B switch: out_proj and x_proj at d_model 512 as plain torch products (hipBLASLt) instead of cad_gemm_stream / the two K halves of
# cad_proj_wx -- the library is ~1.3 % of the configs[4] step faster on these two (profiles/r05_step_trace_c4.txt); default: own kernels
_LIB_OUT_X_D512 = os.environ.get("CADUCEUS_AMD_LIB_OUT_X_PROJ_D512", "0") == "1"
# BASELINE configs[4]: in_proj on the fp8 (OCP e4m3) matrix cores (csrc/gemm_fp8.hip); set CADUCEUS_AMD_FP8_PROJ=1 or call
# set_fp8_in_proj(True).  Forward only: the backward keeps the bf16 activations it saves today.
_FP8_IN_PROJ = os.environ.get("CADUCEUS_AMD_FP8_PROJ", "0") == "1"
# the dB / dC partial slots folded by cad_fold_partials_stream on a second stream WHILE the scan backward runs (the scan is bound by VALU
# issue, the fold by memory latency: 0.39 ms per layer of fold kernel leave the critical path); CADUCEUS_AMD_STREAM_FOLD=0: the fold
# kernel behind the scan (cad_reduce_partials_multi) -- same summation order, bit-identical gradients
_STREAM_FOLD = os.environ.get("CADUCEUS_AMD_STREAM_FOLD", "1") != "0"
_STREAM_FOLD_MIN_CHUNKS = 16  # rows shorter than this many 512-position chunks keep the fold kernel behind the scan (see backward)
# the fp32 partial tiles of a layer's weight gradients (dW_in, dW_out: K slices of cad_gemm_stream; dW_x, dW_dt: per-workgroup slots of
# cad_proj_wx_wgrad) summed by ONE own launch at the end of the backward (cad_fold_f32_multi) instead of four torch reductions + an add;
# CADUCEUS_AMD_GLUE_FOLD=0: torch.sum per tensor
_GLUE_FOLD = os.environ.get("CADUCEUS_AMD_GLUE_FOLD", "1") != "0"


# test hook (tests/test_configs.py): a list here receives, per backward call, the operands of the x_proj weight gradient of both
# parameter sets as the kernels saw them -- {"ddbc": [d(dt_lr ; B ; C) (R + 2N, T)] * 2, "xc": [conv output (E, T)] * 2} (clones)
CAPTURE_XPROJ_OPERANDS = None


def set_fp8_in_proj(on: bool) -> None:
    global _FP8_IN_PROJ
    _FP8_IN_PROJ = bool(on)
    ops.FP8_ACTIVATIONS = _FP8_IN_PROJ


ops.FP8_ACTIVATIONS = _FP8_IN_PROJ


def prepare_step_cache(pairs, act: torch.dtype) -> None:
    """Compute-dtype copies of every layer's projection weights and A = -exp(A_log), for ALL layers at once with
    multi-tensor (foreach) launches -- instead of ten tiny cast / exp / neg kernels per layer per step
    (profiles/r02_step_trace.txt: ~1000 sub-5-us launches per step).  pairs: [(mamba_fwd, mamba_rev)] of the layers that run
    BiMambaMixerFn.  The copies are attached to mamba_fwd together with the parameter versions they were made from;
    BiMambaMixerFn.forward uses them only while those versions are current."""
    if not pairs:
        return
    src, dst, alog, owners = [], [], [], []
    # the compute-dtype copies of one kind of weight (in / out / x / dt) of ALL layers are slices of one stacked buffer, so that the
    # backward's transposed operands (W_out^T, W_x^T, W_dt^T) come from ONE batched transpose per kind and step instead of five
    # strided copy launches per layer (a multi-tensor copy from transposed views falls back to one launch per tensor)
    kinds = ("in", "out", "x", "dt")
    plist = [{"in": [mf.in_proj.weight], "out": [mf.out_proj.weight], "x": [mf.x_proj.weight, mr.x_proj.weight],
              "dt": [mf.dt_proj.weight, mr.dt_proj.weight]} for mf, mr in pairs]
    stacked = {}
    for kd in kinds:
        shapes = {tuple(p.shape) for pl in plist for p in pl[kd]}
        n = sum(len(pl[kd]) for pl in plist)
        if len(shapes) == 1:
            stacked[kd] = torch.empty((n,) + next(iter(shapes)), dtype=act, device=plist[0][kd][0].device)
    cursor = {kd: 0 for kd in kinds}

    def alloc(kd, p):
        if kd in stacked:
            v = stacked[kd][cursor[kd]]
            cursor[kd] += 1
            return v
        return torch.empty(p.shape, dtype=act, device=p.device)

    for (mf, mr), pl in zip(pairs, plist):
        ps = [pl["in"][0], pl["out"][0], pl["x"][0], pl["dt"][0], pl["x"][1], pl["dt"][1]]
        out = [alloc(kd, p) for kd, p in zip(("in", "out", "x", "dt", "x", "dt"), ps)]
        src += [p.detach() for p in ps]
        dst += out
        alog += [mf.A_log.detach().float(), mr.A_log.detach().float()]
        owners.append((mf, ps + [mf.A_log, mr.A_log], out))
    torch._foreach_copy_(dst, src)
    # (W_in^T feeds cad_gemm_stream: d(x2d) -- d_model <= 256, or 512 with the own tiled GEMM on -- and the streamed in_proj forward at
    # d_model 512, which must not depend on the d(x2d) switches: ADVICE r5)
    need_in_T = "in" in stacked and ((_OWN_GEMM and (stacked["in"].shape[2] <= 256 or _OWN_GEMM_D512)) or
                                     (_STREAM_PROJ_D512 and stacked["in"].shape[2] > 256))
    trans = {kd: stacked[kd].transpose(1, 2).contiguous() for kd in ("out", "x", "dt") + (("in",) if need_in_T else ()) if kd in stacked}
    cursor = {kd: 0 for kd in kinds}

    def transposed(kd, w):
        if kd in trans:
            v = trans[kd][cursor[kd]]
            cursor[kd] += 1
            return v
        return w.t().contiguous()

    for mf, ps, out in owners:
        # [W_out^T, W_x_f^T, W_dt_f^T, W_x_r^T, W_dt_r^T, W_in^T], in the order the stacked buffers were filled
        out.append([transposed(kd, out[j]) for kd, j in (("out", 1), ("x", 2), ("dt", 3), ("x", 4), ("dt", 5))] +
                   [transposed("in", out[0]) if need_in_T else None])
    negA = torch._foreach_exp(alog)
    torch._foreach_neg_(negA)
    for i, (mf, ps, out) in enumerate(owners):
        tr = out.pop()  # [W_out^T, W_x_f^T, W_dt_f^T, W_x_r^T, W_dt_r^T, W_in^T]
        mf._cad_step_cache = {"versions": [(id(p), p._version) for p in ps], "w": out, "A": (negA[2 * i], negA[2 * i + 1]),
                              "wT": {"out": tr[0], "x": (tr[1], tr[3]), "dt": (tr[2], tr[4]), "in": tr[5]}}
        if _STREAM_PROJ_D512 and not _LIB_OUT_X_D512 and out[1].shape[0] > 256:
            # d_model 512: out_proj streams [y_f ; y_r] against [W_out, W_out] (K = 2 E) -- built once per step, not per layer call
            mf._cad_step_cache["w_out2"] = torch.cat([out[1], out[1]], 1)
        if _FP8_IN_PROJ and act == torch.bfloat16 and ops.fp8_proj_supported(out[0], out[0].shape[1]):
            mf._cad_step_cache["w_in_fp8"] = ops.quant_weight_fp8(ps[0])  # from the fp32 master weight, once per step


def _cached(mf, params):
    c = getattr(mf, "_cad_step_cache", None)
    if c is None or c["versions"] != [(id(p), p._version) for p in params]:
        return None
    return c


class BiMambaMixerFn(torch.autograd.Function):
    """out (T, D) = out_proj(scan_f + scan_r) for normed input x2d (T, D), T = S*B*L rows in t-frame order.

    Tensor arguments: x2d, W_in (2E, D), W_out (D, E), then per parameter set (f, r):
    conv_w (E,1,K), conv_b (E), W_x (R+2N, E), W_dt (E, R), dt_bias (E), A_log (E, N), D (E)."""

    @staticmethod
    def forward(ctx, x2d, SB, Lq, split, cache, fp8_act, W_in, W_out, *ps):
        lib = L.get_lib()
        act = x2d.dtype
        T, Dm = x2d.shape
        E = W_in.shape[0] // 2
        if cache is not None and cache["w"][0].dtype != act:
            cache = None
        w_in = cache["w"][0] if cache else W_in.to(act)
        w_out = cache["w"][1] if cache else W_out.to(act)
        if _FP8_IN_PROJ and act == torch.bfloat16 and ops.fp8_proj_supported(x2d, Dm):
            # fp8 matrix cores: per-token e4m3 activations x per-row e4m3 weights, fp32 accumulation, bf16 channel-major output
            wq, sw = cache["w_in_fp8"] if (cache and "w_in_fp8" in cache) else ops.quant_weight_fp8(W_in)
            # e4m3 activations + per-token scales: written by the add + norm kernel that produced x2d (fp8_act), else quantised here
            xq, sx = (fp8_act[0].view(T, Dm), fp8_act[1]) if fp8_act is not None else ops.quant_rows_fp8(x2d)
            xz = ops.proj_wxT_fp8(wq, sw, xq, sx).view(2 * E, SB, Lq)
        else:
            xz = None
            if _STREAM_PROJ_D512 and Dm > 256 and act == torch.bfloat16:
                w_inT = ((cache or {}).get("wT") or {}).get("in")
                if w_inT is None:  # no step cache (eval, or parameters changed since prepare_step_cache): transpose here
                    w_inT = w_in.t().contiguous()
                # d_model 512: both operands streamed through the tiled kernel (A = tokens, B = W_in^T from the step cache), channel-major
                # result (None if the shape is not served)
                xz = ops.gemm_out_t(x2d, w_inT)
            if xz is None and ops.proj_supported(x2d, Dm):  # bf16: the W-stationary MFMA kernel (csrc/gemm.hip) writes channel-major directly
                xz = ops.proj_wxT(w_in, x2d)
            if xz is None:
                xz = ops.mm(w_in, x2d.t())  # fp32: cad_gemm_f32; a bf16 shape no own kernel serves: the library
            xz = xz.view(2 * E, SB, Lq)
        x, z = xz[:E], xz[E:]
        sets, saved = [], []
        dirs = ((0, 1), (1, 0))
        cparams = []
        for i in range(2):
            conv_w, conv_b = ps[7 * i], ps[7 * i + 1]
            cparams.append((conv_w.float().reshape(E, -1).contiguous(),
                            None if conv_b is None else conv_b.float().contiguous()))
        xcs = _conv_fwd2(x, cparams, split, dirs)
        fused_sp = []
        for i in range(2):
            conv_w, conv_b, W_x, W_dt, dt_bias, A_log, Dp = ps[7 * i:7 * i + 7]
            N, R = A_log.shape[1], W_dt.shape[1]
            wf, bf = cparams[i]
            xc = xcs[i]
            w_x, w_dt = (cache["w"][2 + 2 * i], cache["w"][3 + 2 * i]) if cache else (W_x.to(act), W_dt.to(act))
            if ops.proj_wx_supported(xc, E, T, M=R + 2 * N):  # thin-M / deep-K MFMA kernel: xc read once, W_x in LDS
                dbc = ops.proj_wx(w_x, xc.view(E, T)).view(R + 2 * N, SB, Lq)
            elif not _LIB_OUT_X_D512 and E % 128 == 0 and ops.proj_wx_supported(xc, E // 2, T, M=R + 2 * N):
                # d_inner 1024 (configs[4]): 64 rows x 1024 of W_x do not fit LDS next to the X ring -- two K halves.  The first half is
                # STORED in bf16 and widened again as the addend of the second, so dt_lr / B / C see two roundings (first half, then the
                # sum), not one fp32 accumulation over K; xc is still read once
                dbc = ops.proj_wx(w_x[:, :E // 2], xc.view(E, T)[:E // 2])
                ops.proj_wx(w_x[:, E // 2:], xc.view(E, T)[E // 2:], out=dbc, acc=dbc)
                dbc = dbc.view(R + 2 * N, SB, Lq)
            else:
                dbc = ops.mm(w_x, xc.view(E, T)).view(R + 2 * N, SB, Lq)
            if ops.proj_wx_supported(xc, R, T):  # thin-K MFMA kernel (transposing LDS reads), csrc/gemm.hip
                # ... with delta_bias + softplus in its epilogue (fp32): the scans take dt as it is (delta_is_dt)
                delta = ops.proj_wx(w_dt, dbc[:R].view(R, T),
                                    softplus_bias=dt_bias.float().contiguous() if _FUSED_SOFTPLUS else None).view(E, SB, Lq)
                fused_sp.append(_FUSED_SOFTPLUS)
            else:
                delta = ops.mm(w_dt, dbc[:R].view(R, T)).view(E, SB, Lq)
                fused_sp.append(False)
            A = cache["A"][i] if cache else -torch.exp(A_log.float())
            sets.append((xc, delta, A, dbc, Dp.float().contiguous(), dt_bias.float().contiguous(), wf, bf, w_x, w_dt))
        # both parameter sets in one scan launch; k > 1: every row cut into k segments along L (ops.lsplit_factor) when the
        # launch would otherwise leave CUs idle (Caduceus-Ph at batch 1)
        k = ops.lsplit_factor(E, SB, Lq, 2)
        args = (L.ScanArgs * 2)()
        outs, states = [], []
        ycat = torch.empty((2 * E, SB, Lq), dtype=act, device=x2d.device)  # [y_f ; y_r]: one out_proj GEMM with K = 2E
        for i, (xc, delta, A, dbc, Df, bfz, *_rest) in enumerate(sets):
            N, R = A.shape[1], dbc.shape[0] - 2 * A.shape[1]
            out = ycat[i * E:(i + 1) * E]
            state = torch.empty((lib.cad_scan_state_floats(E, SB * k, Lq // k, N),), dtype=torch.float32, device=xc.device)
            Bm, Cm = dbc[R:R + N], dbc[R + N:]
            stream = L.stream_and_check(xc, delta, A, Bm, Cm, Df, z, bfz, out, state)
            args[i] = L.ScanArgs(L.ptr(xc), L.ptr(delta), L.ptr(A), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z), L.ptr(bfz),
                                 L.ptr(out), L.ptr(state), SB * k, Lq // k, split * k, E, N, dirs[i][0], dirs[i][1],
                                 L.dtype_code(act))
            args[i].delta_is_dt = int(fused_sp[i])
            outs.append(out)
            states.append(state)
        _keep, seg_P = ops.scan_fwd_launch(lib, args, 2, stream, k, [st[2] for st in sets], dirs, split)
        y_f, y_r = outs
        if _OWN_OUT_PROJ and ops.proj_xTw_supported(ycat, Dm, E, T):
            # W_out (y_f + y_r): both panels through one set of resident W_out fragments, token-major output (cad_proj_xTw)
            out2d = ops.proj_xTw(w_out, y_f.view(E, T), y_r.view(E, T))
        else:
            # d_model 512 (configs[4]): W_out is too deep for resident fragments -- both operands streamed (cad_gemm_stream), [y_f ; y_r]
            # against [W_out, W_out] with K = 2 E; plain products for anything the kernel does not serve
            out2d = None
            if _STREAM_PROJ_D512 and not _LIB_OUT_X_D512 and act == torch.bfloat16:
                w_out2 = (cache or {}).get("w_out2")
                out2d = ops.proj_xTw_stream(w_out2 if w_out2 is not None else torch.cat([w_out, w_out], 1), ycat.view(2 * E, T))
            if out2d is None:
                out2d = ops.mm(ycat.view(2 * E, T).t(), torch.cat([w_out, w_out], 1).t())  # W_out (y_f + y_r), tied out_proj
        wT = cache.get("wT") if cache else None
        keep = [x2d, xz, w_in, w_out, ycat]
        for i in range(2):
            xc, delta, A, dbc, Df, bfz, wf, bf, w_x, w_dt = sets[i]
            keep += [xc, delta, A, dbc, Df, bfz, wf, bf, w_x, w_dt, states[i], ps[7 * i + 5]]
        ctx.wT = wT  # (not saved tensors: plain per-step copies owned by the cache)
        ctx.save_for_backward(*keep, *seg_P)
        ctx.meta = (SB, Lq, split, [tuple(None if p is None else (p.dtype, p.shape) for p in ps[7 * i:7 * i + 7])
                                    for i in range(2)], W_in.dtype, W_out.dtype, tuple(fused_sp), k)
        return out2d

    @staticmethod
    def backward(ctx, dout2d):
        lib = L.get_lib()
        x2d, xz, w_in, w_out, ycat, *rest = ctx.saved_tensors
        SB, Lq, split, pmeta, win_dt, wout_dt, fused_sp, k = ctx.meta
        seg_P, rest = (rest[24:], rest[:24]) if k > 1 else ([], rest)
        act = x2d.dtype
        T, Dm = x2d.shape
        E = xz.shape[0] // 2
        x, z = xz[:E], xz[E:]
        dirs = ((0, 1), (1, 0))
        dout2d = dout2d.contiguous()
        # tied out_proj: the gradient w.r.t. y_f and y_r is the same tensor, produced channel-major
        wT = ctx.wT
        dy = None
        if _STREAM_PROJ_D512 and Dm > 256 and act == torch.bfloat16:
            dy = ops.gemm_out_t(dout2d, w_out)  # (T, D) @ W_out (D, E) -> (E, T): the weight as it lies is the row-major B operand
        if dy is None and ops.proj_supported(dout2d, Dm):
            dy = ops.proj_wxT(wT["out"] if wT else w_out.t().contiguous(), dout2d)
        if dy is None:
            dy = ops.mm(w_out.t(), dout2d.t())
        dy = dy.view(E, SB, Lq)
        y_f, y_r = ycat[:E], ycat[E:]
        glue = []  # (src, dst, n, nparts, stride, nparts2, stride2) jobs of the one fp32 fold launch at the end (_GLUE_FOLD)
        part_out = None
        if _GLUE_FOLD and _OWN_GEMM and (Dm <= 256 or _OWN_GEMM_D512):
            part_out = ops.wgrad_cm_tm(ycat.view(2 * E, T), dout2d, return_partials=True)  # (slices, 2E, D)
        if part_out is not None:
            # both halves multiply the same tied weight: the fold adds them (second level) as it sums the slices
            dW_out_ED = torch.empty((E, Dm), dtype=torch.float32, device=x2d.device)
            glue.append((part_out, dW_out_ED, E * Dm, part_out.shape[0], 2 * E * Dm, 2, E * Dm))
            dW_out = dW_out_ED.t()
        else:
            dW_cat = _wgrad_cm_tm(ycat.view(2 * E, T), dout2d)  # (2E, D): both halves multiply the same tied weight
            dW_out = (dW_cat[:E] + dW_cat[E:]).t()
        dxz = torch.empty_like(xz)       # [dx ; dz]: the gate z is shared, set 0's kernel writes the gradient of both gates
        sets = [rest[12 * i:12 * i + 12] for i in range(2)]
        args = (L.ScanBwdArgs * 2)()
        work = []
        zshapes = []
        for i in range(2):  # per set: dA, dD, ddelta_bias (scan), dw, db (conv): accumulated by the kernels -> zeroed
            _, _, A_, _, Df_, bfz_, wf_, bf_ = sets[i][:8]
            zshapes += [A_.shape, Df_.shape, bfz_.shape, wf_.shape, (bf_.shape if bf_ is not None else (0,))]
        zshapes += [(1,), (1,)]  # worklist counters of the exact z == 0 gate gradient (int32 views of zero bits)
        N0 = sets[0][2].shape[1]
        npart0 = lib.cad_scan_bwd_partials(E)
        nch = max(1, (Lq // k) // int(lib.cad_scan_bwd_chunk_len()))
        # (the fold follows the scan chunk by chunk with one workgroup per CU: it pays for long rows -- many chunks per (row, slice) item,
        # few items per workgroup; short rows in large batches (configs[1]: 2 chunks, 64 items per workgroup) keep the streaming fold
        # kernel behind the scan: 4.18 vs 4.83 ms per layer, profiles/r06_ab_stream_fold.txt)
        stream_fold = (_STREAM_FOLD and sets[1][2].shape[1] == N0 and ops.fold_stream_supported(N0, npart0, Lq // k, act)
                       and nch >= _STREAM_FOLD_MIN_CHUNKS and npart0 * SB * k * 2 <= 4 * ops._cu_count()
                       and ops.fold_side_available(x2d.device))
        if stream_fold:  # arrival counters (set, row, chunk) and give-up records (set, row, slice) of the concurrent fold: zero bits
            nci = (int(lib.cad_scan_bwd_fold_counter_ints(SB * k, Lq // k)) + 3) // 4 * 4  # chunk arrivals + started count + CU marks
            zshapes += [(2, nci), (2, SB * k, npart0)]
        zbuf = _zeros_f32(zshapes, x2d.device)
        fix_cnt = [zbuf[10].view(torch.int32), zbuf[11].view(torch.int32)]
        n_fix = lib.cad_scan_gate_fix_entries(E, SB, Lq)
        fix_list = [torch.empty((n_fix,), dtype=torch.int64, device=x2d.device) for _ in range(2)]
        wg_dt = wg_x = None  # fp32 partial slots of the own weight-gradient kernels, both sets
        for i in range(2):
            xc, delta, A, dbc, Df, bfz, wf, bf, w_x, w_dt, state, A_log = sets[i]
            N = A.shape[1]
            R = dbc.shape[0] - 2 * N
            du, ddelta = torch.empty_like(xc), torch.empty_like(xc)
            dz = dxz[E:] if i == 0 else (None if _SHARED_GATE else torch.empty_like(z))
            dz_r = dz if i == 1 else None
            dA, dD, dbias = zbuf[5 * i:5 * i + 3]
            npart = lib.cad_scan_bwd_partials(E)
            dBC = torch.empty((2, npart, N, SB, Lq), dtype=act, device=xc.device)
            Bm, Cm = dbc[R:R + N], dbc[R + N:]
            stream = L.stream_and_check(xc, delta, A, Bm, Cm, Df, z, bfz, dy, state, du, ddelta, dz, dA, dBC, dD, dbias)
            args[i] = L.ScanBwdArgs(L.ptr(xc), L.ptr(delta), L.ptr(A), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z),
                                    L.ptr(bfz), L.ptr(dy), L.ptr(y_f if i == 0 else y_r), L.ptr(state), L.ptr(du),
                                    L.ptr(ddelta), L.ptr(dz), L.ptr(dA),
                                    L.ptr(dBC[0]), L.ptr(dBC[1]), L.ptr(dD), L.ptr(dbias), SB * k, Lq // k, split * k, E, N,
                                    dirs[i][0], dirs[i][1], L.dtype_code(act), npart, None, None,
                                    L.ptr(y_r) if (i == 0 and _SHARED_GATE) else None, L.ptr(fix_list[i]),
                                    L.ptr(fix_cnt[i]), L.ptr(dxz[E:]) if (_SHARED_GATE or i == 0) else L.ptr(dz))
            args[i].delta_is_dt = int(fused_sp[i])
            if stream_fold:
                args[i].fold_counters = L.ptr(zbuf[12][i])
            work.append((du, ddelta, dA, dD, dbias, dBC, npart))
        # the dB / dC partial slots of BOTH sets are folded straight into the rows of each set's x_proj gradient operand ...
        ddbcs = [torch.empty_like(sets[i][3]) for i in range(2)]
        launch_scan = lambda: ops.scan_bwd_launch(lib, args, 2, stream, k, seg_P, dirs, split)
        if stream_fold:
            # ... chunk by chunk on a second stream while the scan still runs (cad_fold_partials_stream)
            fargs = (L.FoldArgs * 2)()
            for i in range(2):
                N_, R_ = sets[i][2].shape[1], sets[i][3].shape[0] - 2 * sets[i][2].shape[1]
                dBC_ = work[i][5]
                fargs[i] = L.FoldArgs(L.ptr(dBC_[0]), L.ptr(dBC_[1]), L.ptr(ddbcs[i][R_:R_ + N_]), L.ptr(ddbcs[i][R_ + N_:]),
                                      L.ptr(zbuf[12][i]), L.ptr(zbuf[13][i]), SB * k, Lq // k, split * k, N_, work[i][6],
                                      dirs[i][0], dirs[i][1], L.dtype_code(act))
            _keep = ops.fold_behind_scan(lib, fargs, 2, x2d.device, launch_scan, give_ups=zbuf[13])
        else:
            _keep = launch_scan()
        L.check(lib.cad_scan_bwd_gate_fix(args, 2, stream), "cad_scan_bwd_gate_fix")  # no-op unless some z == 0 exactly
        grads, dxcs, part = [], [], []
        if not stream_fold:
            # ... by one launch behind the scan (four folds: cad_reduce_partials_multi)
            jobs = (L.ReduceJob * 4)()
            for i in range(2):
                N_, R_ = sets[i][2].shape[1], sets[i][3].shape[0] - 2 * sets[i][2].shape[1]
                dBC_ = work[i][5]
                jobs[2 * i] = L.ReduceJob(L.ptr(dBC_[0]), L.ptr(ddbcs[i][R_:R_ + N_]))
                jobs[2 * i + 1] = L.ReduceJob(L.ptr(dBC_[1]), L.ptr(ddbcs[i][R_ + N_:]))
            assert work[0][6] == work[1][6] and sets[0][2].shape[1] == sets[1][2].shape[1], "one fold launch: both sets share depth and d_state"
            L.check(lib.cad_reduce_partials_multi(jobs, 4, work[0][6], sets[0][2].shape[1] * SB * Lq, L.dtype_code(act), stream),
                    "cad_reduce_partials_multi")
        for i in range(2):
            xc, delta, A, dbc, Df, bfz, wf, bf, w_x, w_dt, state, A_log = sets[i]
            du, ddelta, dA, dD, dbias, dBC, npart = work[i]
            N = A.shape[1]
            R = dbc.shape[0] - 2 * N
            # gradient of [dt_lr ; B ; C] assembled in place: rows [R:] by the partial-slot reduction, rows [:R] by a GEMM
            ddbc = ddbcs[i]
            if _FUSED_WGRAD and ops.proj_wx_wgrad_supported(ddelta, R, E, T):
                # d(dt_lr) = W_dt^T d(delta) and dW_dt = d(delta) dt_lr^T from ONE pass over d(delta) (cad_proj_wx_wgrad); the
                # partial slots of both parameter sets are folded by one sum after the loop
                if wg_dt is None:
                    wg_dt = ops.wgrad_partials(T, E, R, xc.device, nsets=2)
                ops.proj_wx_wgrad(wT["dt"][i] if wT else w_dt.t().contiguous(), ddelta.view(E, T),
                                  dbc[:R].view(R, T), out=ddbc[:R].view(R, T), part=wg_dt[i])
                dW_dt = None
            else:
                if ops.proj_wx_supported(ddelta, E, T, M=R):
                    ops.proj_wx(wT["dt"][i] if wT else w_dt.t().contiguous(), ddelta.view(E, T), out=ddbc[:R].view(R, T))
                else:
                    ops.mm(w_dt.t(), ddelta.view(E, T), out=ddbc[:R].view(R, T))
                if _OWN_DWX and _own_wgrad_chunked_ok(ddelta.view(E, T), R):
                    dW_dt = _own_wgrad_chunked(ddelta.view(E, T), dbc[:R].view(R, T)).t()
                else:
                    dW_dt = _wgrad_cm_cm(ddelta.view(E, T), dbc[:R].view(R, T))
            if _OWN_DWX and ops.proj_wgrad_only_supported(xc, R + 2 * N, E, T):
                if wg_x is None:
                    wg_x = ops.wgrad_partials(T, E, R + 2 * N, xc.device, nsets=2)
                ops.proj_wgrad_only(xc.view(E, T), ddbc.view(R + 2 * N, T), part=wg_x[i])
                dW_x = None
            elif _OWN_DWX and _own_wgrad_chunked_ok(xc.view(E, T), R + 2 * N):
                dW_x = _own_wgrad_chunked(xc.view(E, T), ddbc.view(R + 2 * N, T))
            else:
                dW_x = _wgrad_cm_cm(ddbc.view(R + 2 * N, T), xc.view(E, T))
            # d(xc) = du + W_x^T . d(dbc), in place (no copy of the 268 MB addend)
            if ops.proj_wx_supported(du, R + 2 * N, T):
                ops.proj_wx(wT["x"][i] if wT else w_x.t().contiguous(), ddbc.view(R + 2 * N, T), out=du.view(E, T),
                            acc=du.view(E, T))
            else:
                if du.dtype == torch.float32:
                    ops.mm_f32(w_x.t(), ddbc.view(R + 2 * N, T), out=du.view(E, T), addend=du.view(E, T))
                else:
                    du.view(E, T).addmm_(w_x.t(), ddbc.view(R + 2 * N, T))
            dxcs.append(du)
            part.append((dW_x, dW_dt, dbias, dA * A, dD))  # A = -exp(A_log)  =>  dA/dA_log = A
        if CAPTURE_XPROJ_OPERANDS is not None:
            CAPTURE_XPROJ_OPERANDS.append({"ddbc": [d.reshape(d.shape[0], T).clone() for d in ddbcs],
                                           "xc": [sets[i][0].reshape(E, T).clone() for i in range(2)]})
        conv_g = _conv_bwd2(x, [(sets[i][6], sets[i][7]) for i in range(2)], dxcs, dxz[:E], split, dirs,
                            bufs=[(zbuf[5 * i + 3], zbuf[5 * i + 4] if sets[i][7] is not None else None) for i in range(2)])
        # one fold per weight for BOTH sets' partial slots (fixed order): (2, P, K, M) -> (2, K, M) / (2, M, K)
        # (the slots hold (K, M); summed as they lie -- a reduction over a permuted view runs at a quarter of the rate -- and the
        # small (2, K, M) result is transposed)
        sum_dt = sum_x_km = None
        if _GLUE_FOLD:
            for wg, name in ((wg_dt, "dt"), (wg_x, "x")):
                if wg is not None:
                    _, P_, K_, M_ = wg.shape
                    res = torch.empty((2, K_, M_), dtype=torch.float32, device=x2d.device)
                    for i in range(2):
                        glue.append((wg[i], res[i], K_ * M_, P_, K_ * M_, 1, 0))
                    if name == "dt":
                        sum_dt = res
                    else:
                        sum_x_km = res
        else:
            sum_dt = None if wg_dt is None else wg_dt.sum(dim=1)
            sum_x_km = None if wg_x is None else wg_x.sum(dim=1)
        if dz_r is not None:
            dxz[E:].add_(dz_r)
        # d(x2d) and dW_in first (the fold launch below reads dW_in's partial tiles)
        # (d_model 256: one 256-row tile, dxz read once; d_model 512: profiles/r05_gemm_stream_d512.txt)
        dx2d = None
        if _OWN_GEMM and (Dm <= 256 or _OWN_GEMM_D512):
            w_inT = wT["in"] if (wT and wT.get("in") is not None) else w_in.t().contiguous()
            dx2d = ops.proj_xTw_stream(w_inT, dxz.view(2 * E, T))
        if dx2d is None:
            dx2d = ops.mm(dxz.view(2 * E, T).t(), w_in)
        part_in = None
        if _GLUE_FOLD and _OWN_GEMM and (Dm <= 256 or _OWN_GEMM_D512):
            part_in = ops.wgrad_cm_tm(dxz.view(2 * E, T), x2d, return_partials=True)
        if part_in is not None:
            dW_in = torch.empty((2 * E, Dm), dtype=torch.float32, device=x2d.device)
            glue.append((part_in, dW_in, 2 * E * Dm, part_in.shape[0], 2 * E * Dm, 1, 0))
        else:
            dW_in = None
        if glue:
            ops.fold_f32(glue)
        sum_x = None if sum_x_km is None else sum_x_km.transpose(1, 2).contiguous()
        for i in range(2):
            meta = pmeta[i]
            (dwc, dbc_conv), (dW_x, dW_dt, dbias, dA_log, dD) = conv_g[i], part[i]
            dW_x = sum_x[i] if dW_x is None else dW_x
            dW_dt = sum_dt[i] if dW_dt is None else dW_dt
            grads += [dwc.reshape(meta[0][1]).to(meta[0][0]), None if dbc_conv is None else dbc_conv.to(meta[1][0]),
                      dW_x.to(meta[2][0]), dW_dt.to(meta[3][0]), dbias.to(meta[4][0]), dA_log.to(meta[5][0]),
                      dD.to(meta[6][0])]
        if dW_in is None:
            dW_in = _wgrad_cm_tm(dxz.view(2 * E, T), x2d)
        return (dx2d, None, None, None, None, None, dW_in.to(win_dt), dW_out.to(wout_dt), *grads)


def can_use(mamba_fwd, mamba_rev, strategy) -> bool:
    """The hand-scheduled path covers the released-model configuration; everything else goes through engine.py."""
    if mamba_rev is None or (strategy or "add") != "add":
        return False
    tied = (mamba_rev.in_proj.weight is mamba_fwd.in_proj.weight and mamba_rev.out_proj.weight is mamba_fwd.out_proj.weight)
    no_bias = mamba_fwd.in_proj.bias is None and mamba_fwd.out_proj.bias is None
    same = mamba_fwd.d_state == mamba_rev.d_state and mamba_fwd.dt_rank == mamba_rev.dt_rank
    return bool(tied and no_bias and same)


def bimamba_mixer(hn: torch.Tensor, mamba_fwd, mamba_rev, split: int) -> torch.Tensor:
    S, B, Lq, Dm = hn.shape
    ps = []
    for m in (mamba_fwd, mamba_rev):
        ps += [m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight, m.dt_proj.bias, m.A_log, m.D]
    cache = _cached(mamba_fwd, [mamba_fwd.in_proj.weight, mamba_fwd.out_proj.weight, mamba_fwd.x_proj.weight,
                                mamba_fwd.dt_proj.weight, mamba_rev.x_proj.weight, mamba_rev.dt_proj.weight,
                                mamba_fwd.A_log, mamba_rev.A_log])
    fp8_act = ops.fp8_operand_of(hn) if _FP8_IN_PROJ else None  # (ops.add_norm(want_fp8=True) attaches it; stale copies are refused)
    out = BiMambaMixerFn.apply(hn.reshape(S * B * Lq, Dm), S * B, Lq, split, cache, fp8_act, mamba_fwd.in_proj.weight,
                               mamba_fwd.out_proj.weight, *ps)
    return out.view(S, B, Lq, Dm)
