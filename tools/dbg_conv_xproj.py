"""Diagnostic: fused conv/x_proj backward against the three-kernel path on the device -- where do dx mismatches sit?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import mixer, ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
for case in [(32, 2, 1000, 48, 4, 1), (64, 3, 2048, 48, 4, 2), (32, 1, 496, 24, 3, 0), (64, 2, 8, 64, 4, 1), (32, 2, 4096, 48, 4, 1), (512, 2, 16384, 48, 4, 1)]:
    for seed in (11, 12):
        E, SB, L, M, K, split = case
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(E, SB, L, generator=g).to(dev).to(bf)
        params = [(torch.randn(E, K, generator=g).to(dev), torch.randn(E, generator=g).to(dev)) for _ in range(2)]
        dus = [torch.randn(E, SB, L, generator=g).to(dev).to(bf) for _ in range(2)]
        ddbcs = [torch.randn(M, SB, L, generator=g).to(dev).to(bf) for _ in range(2)]
        wxTs = [(0.2 * torch.randn(M, E, generator=g)).to(dev).to(bf).t().contiguous() for _ in range(2)]
        dirs = ((0, 1), (1, 0))
        T = SB * L
        dx_ref = torch.empty_like(x)
        dxc = [ops.proj_wx(wxTs[i], ddbcs[i].view(M, T), acc=dus[i].view(E, T)).view(E, SB, L) for i in range(2)]
        ref = mixer._conv_bwd2(x, params, dxc, dx_ref, split, dirs)
        dx = torch.empty_like(x)
        bufs = [(torch.zeros_like(params[i][0]), torch.zeros_like(params[i][1])) for i in range(2)]
        slots = mixer._conv_xproj_bwd2(x, params, dus, ddbcs, wxTs, dx, split, dirs, bufs)
        torch.cuda.synchronize()
        bad = (dx != dx_ref)
        n = int(bad.sum())
        msg = f"case {case} seed {seed}: mismatches {n} / {dx.numel()}"
        if n:
            idx = bad.nonzero()[:8].tolist()
            d = (dx.float() - dx_ref.float()).abs()
            msg += f" max|d| {float(d.max()):.4g} rel-to-max {float(d.max() / dx_ref.float().abs().max()):.3g} first {idx}"
            # single-set variants: which stage differs?
        print(msg, flush=True)
