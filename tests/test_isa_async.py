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


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["scan_fwd.hip", "scan_bwd.hip"])
def test_async_prefetch_registers_untouched(tmp_path, src):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S",
                           "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "caduceus_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    kernel, pending, n_loads, n_waits = None, set(), 0, 0
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        if re.match(r"^_Z\w+:$", line):
            assert not pending, f"{kernel}: prefetch still in flight at the end of the kernel"
            kernel = line[:-1]
        elif line.startswith(";;#ASMSTART"):
            body = []
            i += 1
            while not lines[i].strip().startswith(";;#ASMEND"):
                body.append(lines[i].strip())
                i += 1
            for b in body:
                if b.startswith("global_load_dword"):
                    pending |= _regs(b.split()[1].rstrip(","))
                    n_loads += 1
                elif b.startswith("s_waitcnt vmcnt(0)"):
                    pending = set()
                    n_waits += 1
                else:
                    assert not (pending & _mentioned(b)), f"{kernel}: `{b}` touches an in-flight prefetch register"
        elif line and not line.startswith((";", ".")) and not line.endswith(":"):
            if line.startswith("s_endpgm"):
                pending = set()  # early-exit paths (threads that do not stage never wait)
            else:
                hit = pending & _mentioned(line)
                assert not hit, f"{kernel}: `{line}` touches in-flight prefetch registers v{sorted(hit)}"
        i += 1
    assert n_loads >= 8 and n_waits >= 4, (n_loads, n_waits)  # every VEC instantiation contains the pattern
