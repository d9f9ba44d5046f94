"""Times every dense projection of one BiMamba layer (forward and backward shapes, as issued by caduceus_amd/mixer.py)
through torch.mm (hipBLASLt) at the BASELINE configs[2] size and prints achieved GB/s of its algorithmic traffic."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
T, E, D, R, N = 262144, 512, 256, 16, 16
dt = torch.bfloat16
r = lambda *s: torch.randn(*s, device=dev, dtype=dt)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


x2d, w_in, w_out = r(T, D), r(2 * E, D), r(D, E)
xc, w_x, w_dt = r(E, T), r(R + 2 * N, E), r(E, R)
dbc, y, dout, ddelta, dxz, dzr = r(R + 2 * N, T), r(E, T), r(T, D), r(E, T), r(2 * E, T), r(E, T)
du = r(E, T)
cases = {
    "fwd in_proj   W(1024x256) @ x^T(256xT)": (lambda: torch.mm(w_in, x2d.t()), (T * D + 2 * E * T) * 2),
    "fwd x_proj    W(48x512) @ xc(512xT)": (lambda: torch.mm(w_x, xc), (E * T + 48 * T) * 2),
    "fwd dt_proj   W(512x16) @ dt(16xT)": (lambda: torch.mm(w_dt, dbc[:R]), (R * T + E * T) * 2),
    "fwd out_proj  y^T(Tx512) @ W^T(512x256)": (lambda: torch.mm(y.t(), w_out.t()), (E * T + T * D) * 2),
    "bwd dy        W^T(512x256) @ dout^T(256xT)": (lambda: torch.mm(w_out.t(), dout.t()), (T * D + E * T) * 2),
    "bwd dW_out    dout^T(256xT) @ y^T(Tx512)": (lambda: torch.mm(dout.t(), y.t()), (T * D + E * T) * 2),
    "bwd ddt_lr    W^T(16x512) @ ddelta(512xT)": (lambda: torch.mm(w_dt.t(), ddelta), (E * T + R * T) * 2),
    "bwd dW_dt     ddelta(512xT) @ dt^T(Tx16)": (lambda: torch.mm(ddelta, dbc[:R].t()), (E * T + R * T) * 2),
    "bwd dW_x      ddbc(48xT) @ xc^T(Tx512)": (lambda: torch.mm(dbc, xc.t()), (48 * T + E * T) * 2),
    "bwd dxc       du + W^T(512x48) @ ddbc(48xT)": (lambda: torch.addmm(du, w_x.t(), dbc), (48 * T + 2 * E * T) * 2),
    "bwd dx2d      dxz^T(Tx1024) @ W(1024x256)": (lambda: torch.mm(dxz.t(), w_in), (2 * E * T + T * D) * 2),
    "bwd dx2d(+)   dzr^T(Tx512) @ W(512x256)": (lambda: torch.addmm(x2d, dzr.t(), w_in[E:]), (E * T + 2 * T * D) * 2),
    "bwd dW_in     dxz(1024xT) @ x(Tx256)": (lambda: torch.mm(dxz, x2d), (2 * E * T + T * D) * 2),
    "bwd dW_in(+)  dzr(512xT) @ x(Tx256)": (lambda: torch.mm(dzr, x2d), (E * T + T * D) * 2),
}
tot = 0.0
for k, (fn, nbytes) in cases.items():
    ms = timeit(fn)
    tot += ms
    print(f"{ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {k}")
print(f"sum {tot:.3f} ms (x_proj/dt_proj/ddt_lr/dW_dt/dW_x/dxc occur twice per layer)")

print("--- weight-gradient GEMMs (K = T) as strided-batch GEMMs over K-chunks + fp32 sum of the partials")
for nch in (16, 64, 256):
    Kc = T // nch
    t1 = timeit(lambda: torch.bmm(dxz.view(2 * E, nch, Kc).permute(1, 0, 2), x2d.view(nch, Kc, D)).float().sum(0))
    t2 = timeit(lambda: torch.bmm(dout.view(nch, Kc, D).transpose(1, 2), y.view(E, nch, Kc).permute(1, 2, 0)).float().sum(0))
    t3 = timeit(lambda: torch.bmm(dbc.view(48, nch, Kc).permute(1, 0, 2), xc.view(E, nch, Kc).permute(1, 2, 0)).float().sum(0))
    t4 = timeit(lambda: torch.bmm(ddelta.view(E, nch, Kc).permute(1, 0, 2), dbc[:R].view(R, nch, Kc).permute(1, 2, 0)).float().sum(0))
    print(f"chunks={nch:4d}: dW_in {t1:.3f} ms  dW_out {t2:.3f} ms  dW_x {t3:.3f} ms  dW_dt {t4:.3f} ms")
ref = torch.mm(dxz.float(), x2d.float())
got = torch.bmm(dxz.view(2 * E, 64, T // 64).permute(1, 0, 2), x2d.view(64, T // 64, D)).float().sum(0)
print("dW_in chunked vs fp32 reference: rel err", float((got - ref).norm() / ref.norm()),
      " plain bf16 mm rel err", float((torch.mm(dxz, x2d).float() - ref).norm() / ref.norm()))
