"""ctypes binding + autograd wrapper of oracle/cad_oracle.c (OpenMP CPU restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cad_oracle.c")
LIB = os.path.join(HERE, "libcad_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.cad_oracle_num_threads.restype = C.c_int
    return _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def num_threads() -> int:
    return int(lib().cad_oracle_num_threads())


class _ScanC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, bias):
        ts = [t.detach().float().contiguous() for t in (u, delta, A, Bm, Cm, D, z, bias)]
        u_, d_, A_, B_, C_, D_, z_, b_ = ts
        nb, E, L = u_.shape
        N = A_.shape[1]
        out = torch.empty_like(u_)
        lib().cad_oracle_scan_fwd(_p(u_), _p(d_), _p(A_), _p(B_), _p(C_), _p(D_), _p(z_), _p(b_), _p(out),
                                  C.c_int64(nb), C.c_int64(E), C.c_int64(L), C.c_int64(N))
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, dout):
        u_, d_, A_, B_, C_, D_, z_, b_ = ctx.saved_tensors
        nb, E, L = u_.shape
        N = A_.shape[1]
        dout = dout.float().contiguous()
        du, dd, dz = torch.empty_like(u_), torch.empty_like(u_), torch.empty_like(u_)
        dA, dB, dC = torch.empty_like(A_), torch.empty_like(B_), torch.empty_like(C_)
        dD, db = torch.empty_like(D_), torch.empty_like(b_)
        lib().cad_oracle_scan_bwd(_p(u_), _p(d_), _p(A_), _p(B_), _p(C_), _p(D_), _p(z_), _p(b_), _p(dout), _p(du),
                                  _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(db), C.c_int64(nb), C.c_int64(E),
                                  C.c_int64(L), C.c_int64(N))
        return du, dd, dA, dB, dC, dD, dz, db


def selective_scan_c(u, delta, A, Bm, Cm, D, z, bias):
    """Same contract as oracle_model.selective_scan, backed by the C restatement (all host cores)."""
    return _ScanC.apply(u, delta, A, Bm, Cm, D, z, bias)


def conv_fwd_c(x, w, bias):
    x_, w_, b_ = x.float().contiguous(), w.float().contiguous(), bias.float().contiguous()
    nb, E, L = x_.shape
    out = torch.empty_like(x_)
    lib().cad_oracle_conv_fwd(_p(x_), _p(w_), _p(b_), _p(out), C.c_int64(nb), C.c_int64(E), C.c_int64(L),
                              C.c_int64(w_.shape[1]))
    return out
