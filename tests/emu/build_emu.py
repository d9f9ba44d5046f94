"""Build the HOST-EMULATOR flavour of the kernel library (tests only; see emu_runtime.h)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "caduceus_amd", "csrc")
# CAD_EMU_DEFINES="SC_S=8" builds (and names) a tuning variant of the emulated library
_DEFS = [d for d in os.environ.get("CAD_EMU_DEFINES", "").split() if d]
LIB = os.path.join(HERE, "libcaduceus_emu" + ("_" + "_".join(d.replace("=", "") for d in _DEFS) if _DEFS else "") + ".so")


def build_emu(force: bool = False) -> str:
    """Several pytest-xdist workers may ask for the library at once: one builds (file lock), the link goes to a
    temporary name and is renamed into place, so nobody ever maps a half-written file."""
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", os.path.basename(LIB) + ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_emu_locked(force)


def _build_emu_locked(force: bool) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "emu_runtime.*")) + glob.glob(os.path.join(HERE, "*.h")) + \
        [os.path.join(ROOT, "include", "caduceus_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objdir = os.path.join(HERE, "build", os.path.basename(LIB))
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O2", "-std=c++17", "-fPIC", "-fopenmp", "-DCAD_EMU", "-I", HERE, "-Wno-attributes", "-Wno-unknown-pragmas"]
    flags += [f"-D{d}" for d in _DEFS]
    procs = []
    for s in srcs:
        obj = os.path.join(objdir, os.path.basename(s) + ".o")
        procs.append((s, obj, subprocess.Popen(["g++", *flags, "-x", "c++", "-c", s, "-o", obj],
                                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    rt = os.path.join(objdir, "emu_runtime.o")
    procs.append((os.path.join(HERE, "emu_runtime.cpp"), rt,
                  subprocess.Popen(["g++", *flags, "-c", os.path.join(HERE, "emu_runtime.cpp"), "-o", rt],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ (emu) failed for {s}:\n{out.decode()}")
        objs.append(obj)
    tmp = LIB + f".tmp{os.getpid()}"
    subprocess.check_call(["g++", "-shared", "-fPIC", "-fopenmp", *objs, "-o", tmp])
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
