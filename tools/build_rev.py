"""Builds the kernel library of ANOTHER git revision next to the current one (caduceus_amd/libcaduceus_hip_<name>.so, git-ignored; it travels
to the GPU box with gpurun) for same-box A/B runs of a source change against its predecessor:

    python tools/build_rev.py HEAD~1 base [-DSC_X=1 ...]
    gpurun -- 'bash tools/gpu_ab.sh 3 base default'

The C-ABI (include/caduceus_hip.h) of the revision must be the one the current python side binds."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rev, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    tree = os.path.join(os.environ.get("CAD_EXP_DIR", "/tmp/cad_exp"), "rev_" + name)
    shutil.rmtree(tree, ignore_errors=True)
    os.makedirs(tree)
    ar = subprocess.Popen(["git", "-C", ROOT, "archive", rev, "caduceus_amd/csrc", "include"], stdout=subprocess.PIPE)
    subprocess.check_call(["tar", "-x", "-C", tree], stdin=ar.stdout)
    assert ar.wait() == 0
    sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", rev]).decode().strip()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-Wno-pass-failed",
             f'-DCAD_SRC_HASH="rev-{sha}"', *extra]
    procs = []
    for f in sorted(glob.glob(os.path.join(tree, "caduceus_amd", "csrc", "*.hip"))):
        procs.append((f + ".o", subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", f, "-o", f + ".o"])))
    objs = []
    for o, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed for {o}")
        objs.append(o)
    out = os.path.join(ROOT, "caduceus_amd", f"libcaduceus_hip_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    print("built", out, "from", rev, sha)


if __name__ == "__main__":
    main()
