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


# ---- general lint: inline asm and the producer -> consumer wait states the compiler only pads for instructions it can SEE ------------------
# gfx940-class hardware does not interlock (i) a transcendental's result read by the next VALU instruction (1 wait state), (ii) a matrix
# core (MFMA) result read by a VALU instruction (passes + 3 wait states: 11 for the 8-pass 16x16x32 forms), (iii) a DOT result read by a
# non-DOT VALU instruction.  LLVM's hazard recognizer inserts the s_nop -- for MachineInstrs.  An inline-asm block is ONE opaque
# instruction to it: it knows the registers the block reads and writes, not that the block holds a VALU / transcendental / MFMA.  So the
# two unprotected shapes are: a producer of class (i)-(iii) directly in front of an inline-asm VALU instruction that reads its result
# (round 5: v_mul_legacy_f32 behind v_rcp_f32, found on the device only), and a producer of those classes INSIDE an asm block directly in
# front of any reader.  Checked on the gfx950 assembly of every kernel of the library, in layout order inside a basic block.
_TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32",
          "v_exp_f16", "v_log_f16", "v_rcp_f16", "v_rsq_f16", "v_sqrt_f16")
_NEED = {"trans": 1, "mfma": 11, "dot": 3}


def _klass(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_dot"):
        return "dot"
    if op.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0] in _TRANS:
        return "trans"
    return None


def _reg_set(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return {(tok[0], r) for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r"[va](\d+)$", tok)
    return {(tok[0], int(m.group(1)))} if m else set()


def _split_operands(text):
    ops = [o for o in re.split(r",\s*|\s+", text.strip()) if o]
    return ops


def lint_asm_hazards(body):
    """Returns the list of violations in one kernel body (assembly text)."""
    bad = []
    hist = []  # recent instructions of the current basic block: (op, dest regs, in_asm, wait states it provides)
    in_asm = False
    for raw in body.split("\n"):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";"):
            continue
        if t.endswith(":") or t.startswith("."):
            if re.match(r"^\.?L?BB\d+_\d+:", t) or t.endswith(":"):
                hist = []  # a label: the predecessor is unknown, nothing can be concluded (and nothing is assumed)
            continue
        t = t.split(";")[0].strip()
        parts = t.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            hist = []
            continue
        ops = _split_operands(rest)
        dest = _reg_set(ops[0]) if ops and op.startswith("v_") else set()
        srcs = set()
        if op.startswith("v_"):
            for o in ops[1:]:
                srcs |= _reg_set(o)
            if op.startswith(("v_mfma", "v_smfmac", "v_fmac", "v_pk_fma", "v_dot")) and ops:
                srcs |= set()  # (the accumulator is among ops[1:] already)
            if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c")):
                srcs |= dest  # reads its destination
        if op.startswith("v_") and srcs:
            slots = 0
            for pop, pdest, pasm, pws in reversed(hist):
                k = _klass(pop)
                if k and (in_asm or pasm) and (pdest & srcs):
                    same = (k == "mfma" and _klass(op) == "mfma") or (k == "dot" and _klass(op) == "dot")  # back-to-back accumulation is interlocked
                    if not same and slots < _NEED[k]:
                        bad.append(f"`{t}` reads the result of `{pop}` after {slots} wait state(s) (needs {_NEED[k]}); "
                                   f"{'consumer' if in_asm else 'producer'} is inline asm")
                slots += pws
                if slots >= 12:
                    break
        ws = 1
        if op == "s_nop":
            ws = int(ops[0]) + 1 if ops else 1
        hist.append((op, dest, in_asm, ws))
        if len(hist) > 16:
            hist.pop(0)
    return bad


def test_asm_hazard_lint_catches_the_round5_bug():
    """The lint on the failing shape of round 5 and on its fixed forms."""
    bug = "\n".join(["v_rcp_f32_e32 v5, v4", ";;#ASMSTART", "v_mul_legacy_f32 v6, v7, v5", ";;#ASMEND"])
    assert lint_asm_hazards(bug)
    ok1 = "\n".join(["v_rcp_f32_e32 v5, v4", "s_nop 0", ";;#ASMSTART", "v_mul_legacy_f32 v6, v7, v5", ";;#ASMEND"])
    ok2 = "\n".join(["v_rcp_f32_e32 v5, v4", "v_mul_legacy_f32_e32 v6, v7, v5"])  # compiler-visible: padded by the hazard recognizer if needed
    ok3 = "\n".join(["v_rcp_f32_e32 v5, v4", ";;#ASMSTART", "v_mul_legacy_f32 v6, v7, v8", ";;#ASMEND"])
    assert not lint_asm_hazards(ok1) and not lint_asm_hazards(ok2) and not lint_asm_hazards(ok3)
    mf = "\n".join(["v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]", "s_nop 4", ";;#ASMSTART",
                    "v_cvt_pk_bf16_f32 v12, v0, v1", ";;#ASMEND"])
    assert lint_asm_hazards(mf)  # 6 wait states < 11
    asm_prod = "\n".join([";;#ASMSTART", "v_exp_f32 v5, v4", ";;#ASMEND", "v_add_f32_e32 v6, v5, v5"])
    assert lint_asm_hazards(asm_prod)


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
def test_asm_hazard_lint_every_kernel(tmp_path):
    """Every kernel of the library (all csrc/*.hip, every instantiation), gfx950 assembly."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(ROOT, "caduceus_amd", "csrc", "*.hip")))

    def compile_one(src):
        out = tmp_path / (os.path.basename(src) + ".s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-pass-failed", "-S",
                               "--cuda-device-only", "-o", str(out), src], stderr=subprocess.DEVNULL)
        return open(out).read()

    with ThreadPoolExecutor(4) as ex:
        asms = list(ex.map(compile_one, srcs))
    n_kernels = n_asm = 0
    problems = []
    for src, asm in zip(srcs, asms):
        for name, body in _kernel_bodies(asm).items():
            n_kernels += 1
            n_asm += body.count(";;#ASMSTART")
            for v in lint_asm_hazards(body):
                problems.append(f"{os.path.basename(src)} {name[:70]}: {v}")
    assert n_kernels >= 40 and n_asm >= 100, (n_kernels, n_asm)  # the scan kernels alone hold hundreds of asm statements
    assert not problems, "\n".join(problems[:20])


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
def test_fold_kernel_fits_next_to_two_resident_scan_waves(tmp_path):
    """The dB / dC fold runs on the CUs WHILE the scan backward occupies them (cad_fold_partials_stream): per SIMD two scan waves + one fold
    wave must fit the 512-entry register file (allocation granule 8) and the two kernels' LDS the 160 KB of a CU -- held on the compiled
    resource usage of the production instantiations (a scan kernel that grows past 232 VGPRs, or a fold kernel past 48, would silently
    serialise the two kernels on the device)."""
    out = tmp_path / "scan_bwd.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-pass-failed", "-S",
                           "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "caduceus_amd", "csrc", "scan_bwd.hip")],
                          stderr=subprocess.DEVNULL)
    asm = open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        res[name] = (vg, lds)
    scan = [v for k, v in res.items() if "scan_bwd_kernelI6bf16_tLb1ELb0ELi8ELb1E" in k]
    fold = [v for k, v in res.items() if "fold_stream_kernel" in k]
    gate = [v for k, v in res.items() if "fold_gate_kernel" in k]
    assert len(scan) == 1 and len(fold) == 1 and len(gate) == 1, list(res)
    up8 = lambda v: (v + 7) // 8 * 8
    assert 2 * up8(scan[0][0]) + up8(fold[0][0]) <= 512, (scan, fold)
    scan_dyn_lds = 135168  # bytes the launcher asks for at the production shape (4 tiles + 2 packed slab buffers + 6 DMA slots)
    assert scan_dyn_lds + (fold[0][1] + 1279) // 1280 * 1280 <= 160 * 1024, fold
    assert gate[0][0] <= 16 and gate[0][1] == 0, gate
