"""Static instruction mix of the two production scan kernels (hipcc -S, no GPU needed) -> profiles/r06_scan_isa.json, stamped with the hash of
the scan sources (caduceus_amd/_build.scan_source_hash) so that bench.py quotes it only for the kernels it was counted on.
    python tools/make_scan_isa_json.py [out.json]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from caduceus_amd import _build  # noqa: E402
from isa_mix import loop_mix  # noqa: E402

KERNELS = {  # production instantiations: bf16, vector path, d_state 16 (unrolled pair loop); backward: + dt from the dt_proj epilogue
    "scan_fwd": ("scan_fwd.hip", "scan_fwd_kernelI6bf16_tLb1ELb0ELi8E", 1024),
    "scan_bwd": ("scan_bwd.hip", "scan_bwd_kernelI6bf16_tLb1ELb0ELi8ELb1E", 512),
}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_scan_isa.json")
    res = {"source": "tools/make_scan_isa_json.py: hipcc --offload-arch=gfx950 -O3 -S of csrc/scan_*.hip, instructions of the basic blocks inside the "
                     "chunk loop of each production instantiation, counted statically over ALL paths (a wave executes one direction's store "
                     "variant and one gate variant of them: the executed count per chunk is lower; rocprofv3 SQ_INSTS_VALU / SQ_WAVES is the "
                     "executed count)",
           "scan_src": _build.scan_source_hash(), "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        for kind, (src, sub, chunk) in KERNELS.items():
            asm = os.path.join(td, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-pass-failed",
                                   "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "caduceus_amd", "csrc", src)],
                                  stderr=subprocess.DEVNULL)
            name, mix = loop_mix(asm, sub)
            txt = open(asm).read()
            meta = txt[txt.index(".amdhsa_kernel " + name):]
            vgpr = int(meta.split(".amdhsa_next_free_vgpr")[1].split()[0])
            scratch = int(meta.split(".amdhsa_private_segment_fixed_size")[1].split()[0])
            valu = sum(v for k, v in mix.items() if k.startswith("valu"))
            res["kernels"][kind] = {"kernel": name, "positions_per_chunk": chunk, "vgprs": vgpr, "scratch_bytes": scratch,
                                    "chunk_loop_static": mix, "valu_total": valu, "transcendental": mix.get("trans", 0)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {"valu": v["valu_total"], "trans": v["transcendental"], "vgprs": v["vgprs"]} for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()
