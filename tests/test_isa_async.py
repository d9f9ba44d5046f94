"""The B/C tile prefetch of the scan kernels is issued through inline asm (scan_common.h: sc_async_load / sc_async_wait),
so the compiler does not know the destination registers are in flight.  This test compiles the kernels to gfx950 assembly
(no GPU needed) and checks, for EVERY instantiation, that no instruction reads, writes, copies or spills those registers
between the load and its hand-placed s_waitcnt."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def _mentioned(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", line):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def _walk_to_wait(lines, labels, start, pending, where):
    """Follows the executed path from line index `start` until the hand-placed `s_waitcnt vmcnt(0)`: conditional branches
    are not taken (the prefetch and its wait sit under the same wave-uniform conditions, laid out in source order), except
    `s_cbranch_execnz` (the compiler moved the guarded block out of line: taken = the block runs) and `s_branch`.
    Every instruction on the way is checked against the in-flight registers."""
    i, steps, in_asm = start, 0, True
    while steps < 20000:
        steps += 1
        line = lines[i].strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
        elif line.startswith(";;#ASMEND"):
            in_asm = False
        elif not line or line.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", line) or line.endswith(":"):
            pass
        elif in_asm and line.startswith(("s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(6)")):
            # vmcnt(6): the counted form of the pair-step in which SC_NDMA = 6 LDS-DMA operations were issued BEHIND the
            # tile loads (scan_common.h: sc_async_wait_keep) -- the tile loads are complete, only the DMA stays in flight
            return steps
        elif in_asm and line.startswith("global_load_dword"):
            assert not (pending & _mentioned(line.split(None, 2)[2])), f"{where}: `{line}` uses an in-flight register as address"
            pending = pending | _regs(line.split()[1].rstrip(","))  # the second vector of the same prefetch
        else:
            assert not line.startswith("s_endpgm"), f"{where}: prefetch still in flight at s_endpgm"
            hit = pending & _mentioned(line)
            assert not hit, f"{where}: `{line}` (line {i + 1}) touches in-flight prefetch registers v{sorted(hit)}"
            m = re.match(r"(s_branch|s_cbranch_execnz)\s+(\.LBB\d+_\d+)", line)
            if m:
                i = labels[m.group(2)]
                continue
            # inverted layout of a guarded block: `s_cbranch_<cc> BLOCK ; s_branch SKIP` -- the fall-through is only the jump
            # around the block, the conditional target is the block the prefetch's wait lives in
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", line)
            if m and re.match(r"s_branch\s+\.LBB\d+_\d+", lines[i + 1].strip()):
                i = labels[m.group(1)]
                continue
        i += 1
    raise AssertionError(f"{where}: no s_waitcnt vmcnt(0) found on the path after the prefetch")


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["scan_fwd.hip", "scan_bwd.hip"])
def test_async_prefetch_registers_untouched(tmp_path, src):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                           "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "caduceus_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    # split into kernels (labels are local to a function)
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    n_loads = n_waits = 0
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(lines)
        body = lines[a:b]
        kernel = body[0].split(":")[0]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        in_asm = False
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and t.startswith("s_waitcnt vmcnt(0)"):
                n_waits += 1
        for i, l in enumerate(body):
            t = l.strip()
            if t.startswith("global_load_dword") and body[i - 1].strip().startswith(";;#ASMSTART"):
                n_loads += 1
                _walk_to_wait(body, labels, i + 1, _regs(t.split()[1].rstrip(",")), f"{kernel} line {i + 1}")
    assert n_loads >= 8 and n_waits >= 4, (n_loads, n_waits)  # every VEC instantiation contains the pattern


def _kernel_bodies(asm):
    """{mangled kernel name: text of its body} of a gfx950 assembly file."""
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    out = {}
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        body = lines[a:b]  # (a kernel may hold several s_endpgm: its end is the .Lfunc_end label)
        end = next((k for k, l in enumerate(body) if l.startswith(".Lfunc_end")), len(body))
        out[lines[a].rstrip(":")] = "\n".join(body[:end])
    return out


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,kernels", [
    ("gemm.hip", ["proj_wxT_kernel", "proj_wx_kernel"]),
    ("conv1d.hip", ["conv1d_fwd_kernelI6bf16_tLi2ELb1", "conv1d_bwd_kernelI6bf16_tLi2ELb1"]),  # bf16, both sets, vector path
    ("addnorm.hip", ["add_norm_fwd_vec_kernelI6bf16_t", "add_norm_bwd_vec_kernelI6bf16_t"]),
])
def test_streaming_outputs_leave_with_nontemporal_stores(tmp_path, src, kernels):
    """csrc/cad_stream.h: the large write-once outputs of the HBM-bound kernels are stored with the `nt` cache policy (131.4 -> 129.8 ms
    per training step, profiles/r03_ab_nt_stores.txt).  Held on the ISA: every listed (production: bf16, vector path) instantiation must carry
    nontemporal global stores (a plain store slipping back in would cost the gain silently)."""
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                           "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "caduceus_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    bodies = _kernel_bodies(open(out).read())
    for kern in kernels:
        inst = [n for n in bodies if kern in n]
        assert inst, f"{src}: no instantiation of {kern}"
        for n in inst:
            nt = len(re.findall(r"global_store_dword\w*\s.*\bnt\b", bodies[n]))
            assert nt > 0, f"{n}: no nontemporal store"


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm_stream_counted_waits_match_the_issued_operations(tmp_path):
    """cad_gemm_stream waits with `s_waitcnt vmcnt(N)` where N counts what the wave issued BEHIND the awaited chunk: the LDS-DMA
    instructions of the two later chunks (4 per wave and chunk) and -- in the token-major output mode -- the 32 output stores of a
    finished tile (vmcnt retires in issue order).  Held on the ISA: the immediates are 8 and 8 + 32, a wave issues its DMA in groups of
    four, and the store burst of a tile is exactly 32 instructions (a 33rd would let a wait return before its chunk has landed)."""
    out = tmp_path / "gemm.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o",
                           str(out), os.path.join(ROOT, "caduceus_amd", "csrc", "gemm.hip")], stderr=subprocess.DEVNULL)
    bodies = {n: b for n, b in _kernel_bodies(open(out).read()).items() if "gemm_stream_kernel" in n}
    assert len(bodies) == 2
    for name, body in bodies.items():
        waits = {int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)}
        dma = len(re.findall(r"global_load_lds_dwordx4", body))
        assert dma > 0 and dma % 4 == 0, (name, dma)
        assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", body)) == 32, name  # one k step of a 128 x 64 wave tile
        if "ILi1E" in name:  # CAD_GEMM_OUT_T_BF16
            assert waits == {0, 8, 40}, (name, waits)
            assert len(re.findall(r"global_store_dwordx2", body)) == 32 and not re.findall(r"global_store_dword\b", body), name
        else:                # CAD_GEMM_PARTIALS: the fp32 tile leaves once, behind the last chunk
            assert waits == {0, 8}, (name, waits)
            assert len(re.findall(r"global_store_dword\b", body)) == 128, name


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
def test_legacy_multiply_is_compiler_visible(tmp_path):
    """Round 5, found on the device: `v_mul_legacy_f32` written as inline asm read a `v_rcp_f32` result one wait state early (a transcendental
    -> VALU hazard the compiler pads only for instructions it can see; the host emulator cannot show it) -- dz of the first item of ~10 % of the
    lanes came out 0.  The instruction must reach the kernel through the LLVM intrinsic: present in the lean backward instantiation, never
    inside an inline-asm block, and never issued directly behind the transcendental that produces its operand."""
    out = tmp_path / "scan_bwd.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "caduceus_amd", "csrc", "scan_bwd.hip")], stderr=subprocess.DEVNULL)
    bodies = _kernel_bodies(open(out).read())
    lean = [b for n, b in bodies.items() if "scan_bwd_kernelI6bf16_tLb1ELb0ELi8ELb1E" in n]
    assert len(lean) == 1
    lines = [l.strip() for l in lean[0].split("\n")]
    in_asm, n, prev = False, 0, ""
    for t in lines:
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t.startswith("v_mul_legacy_f32"):
            n += 1
            assert not in_asm, "v_mul_legacy_f32 inside an inline-asm block"
            m = re.match(r"(v_rcp_f32|v_exp_f32|v_log_f32)\w*\s+(v\d+)", prev)
            if m:  # the instruction right in front is a transcendental: it must not be the producer of one of the operands
                assert m.group(2) not in re.split(r"[,\s]+", t)[2:], f"`{prev}` directly in front of `{t}`"
        if t and not t.startswith((";", ".")):
            prev = t
    assert n >= 8, n  # one per item of a lane's segment in the gate gradient
