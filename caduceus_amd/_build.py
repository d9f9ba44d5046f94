"""Build libcaduceus_hip.so (hipcc, gfx950) in-tree.  `python -m caduceus_amd._build` or __graft_entry__.build()."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcaduceus_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash(defines=()) -> str:
    """Identifies the kernel sources (+ tuning defines) a library was built from: compiled into cad_version(), recorded next to
    every committed counter profile, compared by bench.py before it quotes such a profile."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "caduceus_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    for d in sorted(defines):
        h.update(d.encode())
    return h.hexdigest()[:12]


SCAN_SOURCES = ("scan_fwd.hip", "scan_bwd.hip", "scan_common.h", "scan_prims_gfx950.h", "cad_common.h", "cad_prims_gfx950.h", "cad_types.h")


def scan_source_hash() -> str:
    """Identifies the sources the two scan kernels are compiled from (their translation units include nothing else of csrc/ besides the
    C-ABI header).  A counter profile of the scans stays quotable while THESE files are unchanged, whatever happens to the kernels
    around them (bench.py compares it with the `scan_src` recorded in profiles/r03_scan_pmc.json)."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, n) for n in SCAN_SOURCES]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    # of the C-ABI header only the scan section (cad_scan_args / cad_scan_bwd_args and their entry points): a new field of another
    # kernel's argument struct does not change the scans' code
    hdr = open(os.path.join(HERE, "..", "include", "caduceus_hip.h")).read()
    a, b = hdr.find(" * Selective SSM scan"), hdr.find(" * Dense projections of the mixer")
    assert 0 <= a < b, "include/caduceus_hip.h: scan section markers not found"
    h.update(b"caduceus_hip.h[scan]")
    h.update(hdr[a:b].encode())
    return h.hexdigest()[:12]


def build_hip(force: bool = False, verbose: bool = True, defines=(), out: str = LIB, extra_flags=()) -> str:
    """Cross-compiles every kernel for gfx950 (works without a GPU).  Objects are built in parallel.
    `defines` / `out` build tuning variants (e.g. ("SC_S=8",)) next to the default library for A/B measurements."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "caduceus_hip.h")]
    if not force and not _stale(out, deps):
        return out
    objdir = os.path.join(HERE, "build", os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-Wno-pass-failed"]
    flags += [f"-D{d}" for d in defines] + list(extra_flags) + [f'-DCAD_SRC_HASH="{source_hash(defines)}"']
    if defines:  # cad_version() names the tuning defines of a variant build (and marks timing builds: csrc/api.hip)
        flags.append('-DCAD_VARIANT="' + ",".join(sorted(defines)) + '"')
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        procs.append((src, obj, subprocess.Popen([hipcc, *flags, "-c", src, "-o", obj], stdout=subprocess.PIPE,
                                                 stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{log.decode()}")
        if verbose and log.strip():
            print(log.decode())
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


FLOOR_LIB = os.path.join(HERE, "libcaduceus_hip_floor.so")
FLOOR_DEFINES = ("SC_WHATIF=14434",)  # = SC_WHATIF_ARITH_ONLY (scan_common.h)


def build_floor(force: bool = False) -> str:
    """The scans' "stripped timing build" (VERDICT r3 item 1): the same library with the scan kernels reduced to their input stream and
    arithmetic (no stores, no dB / dC slab or flush, no B / C tiles, no barrier; WRONG results by construction).  bench.py times one
    layer's scan launches on it next to the real ones and reports kernel / floor in `roofline.arithmetic_floor`.  Never loaded by the
    product: only through CADUCEUS_AMD_LIB in bench.py's floor worker."""
    return build_hip(force=force, verbose=False, defines=FLOOR_DEFINES, out=FLOOR_LIB)


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
    print(build_floor(force="--force" in sys.argv))
