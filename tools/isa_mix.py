"""Static instruction mix of one kernel of a gfx950 assembly file, per basic block (label to label) and in total.
usage: python tools/isa_mix.py file.s <kernel-substring> [--blocks]"""
import re
import sys
from collections import Counter

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.endswith("_dpp") or "_dpp" in op:
        return "valu_dpp"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_lane"
    if op.startswith("v_cvt"):
        return "valu_cvt"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu_mov"
    if op.startswith("v_cndmask"):
        return "valu_sel"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load_lds",)):
        return "dma"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def kernel_body(path, sub):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        if sub in lines[a]:
            body = lines[a:b]
            end = next((k for k, l in enumerate(body) if l.startswith(".Lfunc_end")), len(body))
            return lines[a].split(":")[0], body[:end]
    raise SystemExit(f"no kernel matching {sub}")


def loop_mix(path, sub):
    """{class: count} over the basic blocks of the kernel's (single, outermost) loop -- the 512 / 1024-position chunk loop of a scan kernel --
    counted statically over ALL paths (both directions' store variants, gate / no gate, tail variants), plus the kernel name."""
    name, body = kernel_body(path, sub)
    tot, inloop = Counter(), False
    for l in body[1:]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            inloop = "Loop Header" in t or "in Loop" in t
            continue
        if not inloop or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        if c.startswith("valu") and ("row_sh" in t or "row_bcast" in t or "quad_perm" in t or "row_half_mirror" in t or "wave_sh" in t):
            c = "valu_dpp"
        tot[c] += 1
    return name, dict(tot)


def main():
    path, sub = sys.argv[1], sys.argv[2]
    name, body = kernel_body(path, sub)
    blocks, cur, curname = [], Counter(), "entry"
    dpp = 0
    for l in body[1:]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((curname, cur))
            cur, curname = Counter(), m.group(1)
            continue
        mk = re.match(r"^;\s*MARK\s+(\S+)", t)
        if mk:
            blocks.append((curname, cur))
            cur, curname = Counter(), "MARK:" + mk.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        if c.startswith("valu") and ("row_sh" in t or "row_bcast" in t or "quad_perm" in t or "row_ror" in t or "wave_sh" in t):
            c = "valu_dpp"
        cur[c] += 1
    blocks.append((curname, cur))
    tot = Counter()
    for _, c in blocks:
        tot.update(c)
    keys = sorted(tot)
    print(name)
    if "--blocks" in sys.argv:
        for n, c in blocks:
            s = sum(c.values())
            if s >= 20:
                v = sum(x for k, x in c.items() if k.startswith("valu"))
                print(f"{n:24s} n={s:5d} VALU={v:5d} trans={c['trans']:3d} salu={c['salu']:4d} lds={c['lds']:3d} vmem={c['vmem']+c['dma']:3d} "
                      f"pk={c['valu_pk']:4d} dpp={c['valu_dpp']:4d} cvt={c['valu_cvt']:3d} mov={c['valu_mov']:3d} sel={c['valu_sel']:3d} plain={c['valu']:4d} nop={c['nop']:3d} wait={c['wait']:3d} mfma={c['mfma']}")
    print("TOTAL", {k: tot[k] for k in keys}, "VALU(all)=", sum(x for k, x in tot.items() if k.startswith("valu")))


if __name__ == "__main__":
    main()
