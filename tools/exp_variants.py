"""Source-level A/B variants WITHOUT touching the product sources (so the library hash and the committed counter profiles stay valid while
an idea is being measured): copies caduceus_amd/csrc + include/ to a scratch tree, applies the named edits, cross-compiles
caduceus_amd/libcaduceus_hip_<name>.so (git-ignored; it travels to the GPU box with gpurun) and leaves the product library alone.

    python tools/exp_variants.py base nt_proj_x_loads ...      # build
    gpurun -- 'bash tools/gpu_ab.sh 2 base nt_proj_x_loads'   # same-box A/B of one production layer (tools/layer_bench.py)

An edit is (file, old text, new text); `old` must occur exactly once (prefix __ALL__: every occurrence).  Adopt a winner by making the same edit in csrc/ (then the usual
evidence: GPU suite, bench, and -- if a scan source changed -- tools/prof_scan.sh + tools/make_scan_pmc_json.py)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.environ.get("CAD_EXP_DIR", "/tmp/cad_exp")

_DMA_NT = ('global_load_lds_dwordx4 %1, off\\n\\ts_mov_b32 m0, %0"', 'global_load_lds_dwordx4 %1, off nt\\n\\ts_mov_b32 m0, %0"')
VARIANTS = {
    "base": [],
    # read-once operands of the projection kernels by LDS-DMA with the nt policy (queued in DESIGN.md section 9 item 10); note that
    # cad_glds16 is shared with the scans' prefetch (measured there: noise)
    "nt_dma_loads": [("cad_common.h",) + _DMA_NT],
    # ... the projection kernels only (the candidate for adoption: measured -0.4 ms per step for the family with nt_dma_loads, while nt on
    # the scans' prefetch alone was noise): a streaming twin of cad_glds16 in cad_stream.h, used by every DMA of gemm.hip / gemm_fp8.hip
    "nt_proj_x_loads": [
        ("cad_stream.h", "template <int FAMILY, typename V>\n__device__ __forceinline__ void cad_store_stream(V* p, V v) {",
         "__device__ __forceinline__ void cad_glds16_stream(const void* gsrc, uint32_t lds_base) {\n"
         "#ifdef CAD_EMU\n    cad_glds16(gsrc, lds_base);\n#else\n    uint32_t keep;\n"
         '    asm volatile("s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, off nt\\n\\ts_mov_b32 m0, %0"\n'
         '                 : "=&s"(keep)\n                 : "v"(gsrc), "s"(lds_base)\n                 : "memory");\n#endif\n}\n'
         "template <int FAMILY, typename V>\n__device__ __forceinline__ void cad_store_stream(V* p, V v) {"),
        ("gemm.hip", "__ALL__cad_glds16(", "cad_glds16_stream("),
        ("gemm_fp8.hip", "__ALL__cad_glds16(", "cad_glds16_stream("),
    ],
    # ordinary stores everywhere (the state before csrc/cad_stream.h)
    "plain_stores": [("cad_stream.h", "#define CAD_NT_MASK 7", "#define CAD_NT_MASK 0")],
}


def build(name):
    tree = os.path.join(SCRATCH, name)
    shutil.rmtree(tree, ignore_errors=True)
    csrc = os.path.join(tree, "caduceus_amd", "csrc")  # same relative position of include/ as in the repo
    shutil.copytree(os.path.join(ROOT, "caduceus_amd", "csrc"), csrc)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tree, "include"))
    for fname, old, new in VARIANTS[name]:
        p = os.path.join(csrc, fname)
        s = open(p).read()
        if old.startswith("__ALL__"):  # every occurrence (at least one)
            old = old[len("__ALL__"):]
            assert s.count(old) >= 1, f"{name}: `{old[:60]}` does not occur in {fname}"
        else:
            assert s.count(old) == 1, f"{name}: `{old[:60]}` occurs {s.count(old)} times in {fname}"
        open(p, "w").write(s.replace(old, new))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-Wno-pass-failed",
             f'-DCAD_SRC_HASH="exp-{name}"']
    procs = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        procs.append((f + ".o", subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", f, "-o", f + ".o"])))
    objs = []
    for o, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"{name}: hipcc failed for {o}")
        objs.append(o)
    out = os.path.join(ROOT, "caduceus_amd", f"libcaduceus_hip_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    print("built", out)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["base"]:
        if n not in VARIANTS:
            raise SystemExit(f"unknown variant {n}; known: {', '.join(VARIANTS)}")
        build(n)
