"""ISA evidence for NOTES.md section 5 item 14: compile csrc/norm_bwd_repro.hip with and without hipcc's SLP
vectoriser, cut k_norm_bwd out of both listings and tabulate, for every packed-fp32 instruction (v_pk_mul/add/fma_f32),
  * who produced each of its sources and how many instructions earlier (VMEM load behind s_waitcnt, DPP op, v_readlane -> SGPR pair, VALU),
  * who consumes its result first and how many instructions later (DPP op, v_readlane, store, VALU, another v_pk),
  * the s_nop / s_waitcnt instructions the compiler put in between.
CPU only (cross-compiles gfx950).    python tools/isa_pk_hazards.py [out_dir]      (writes the two listings + prints the summary)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pdb2reaction_amd", "csrc")
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa"
os.makedirs(out, exist_ok=True)
KERNEL = "_ZN3umx10k_norm_bwdEPKfS1_S1_S1_Pfl"


def listing(tag, extra):
    s = os.path.join(out, f"repro_{tag}.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, "-I", CSRC, "-S", "--cuda-device-only", "-o", s,
                    os.path.join(CSRC, "norm_bwd_repro.hip")], check=True, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for ln in open(s):
        if ln.startswith(KERNEL + ":"):
            on = True
            continue
        if on:
            t = ln.strip()
            if t.startswith("s_endpgm"):
                lines.append(t)
                break
            if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
                continue
            lines.append(t.split(";")[0].strip())
    with open(os.path.join(out, f"k_norm_bwd_{tag}.s"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return lines


def regs(tok):
    """'v[16:17]' -> {'v16','v17'}; 's[0:1]' -> {'s0','s1'}; 'v5' -> {'v5'}"""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([vs])(\d+)", tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()


def parse(ln):
    op, _, rest = ln.partition(" ")
    toks = [t.strip() for t in rest.split(",")]
    toks = [t.split(" ")[0] for t in toks]            # drop modifiers after the last operand
    ops = [regs(t) for t in toks]
    if op.startswith(("global_store", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_cbranch", "s_branch")):
        return op, set(), set().union(*ops) if ops else set()
    dst = ops[0] if ops else set()
    src = set().union(*ops[1:]) if len(ops) > 1 else set()
    return op, dst, src


def kind(op, ln):
    if "dpp" in op or "quad_perm" in ln or "row_ror" in ln:
        return "DPP"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "READLANE"
    if op.startswith("global_load"):
        return "VMEM-LOAD"
    if op.startswith("global_store"):
        return "STORE"
    if op.startswith("v_pk_"):
        return "V_PK"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def analyse(lines):
    ins = [parse(l) + (l,) for l in lines]
    prod, cons = collections.Counter(), collections.Counter()
    dmin_p, dmin_c = {}, {}
    examples = {}
    for i, (op, dst, src, ln) in enumerate(ins):
        if not op.startswith("v_pk_"):
            continue
        for r in sorted(src):
            for j in range(i - 1, -1, -1):
                if r in ins[j][1]:
                    k = kind(ins[j][0], ins[j][3])
                    nops = sum(1 for q in range(j + 1, i) if ins[q][0] in ("s_nop",))
                    waits = sum(1 for q in range(j + 1, i) if ins[q][0] == "s_waitcnt")
                    key = f"{k} -> v_pk ({'SGPR' if r[0] == 's' else 'VGPR'} source)"
                    prod[key] += 1
                    d = i - j - 1
                    if key not in dmin_p or d < dmin_p[key][0]:
                        dmin_p[key] = (d, nops, waits)
                        examples[key] = (ins[j][3], ln)
                    break
        for r in sorted(dst):
            for j in range(i + 1, len(ins)):
                if r in ins[j][2]:
                    k = kind(ins[j][0], ins[j][3])
                    key = f"v_pk -> {k}"
                    cons[key] += 1
                    d = j - i - 1
                    nops = sum(1 for q in range(i + 1, j) if ins[q][0] == "s_nop")
                    if key not in dmin_c or d < dmin_c[key][0]:
                        dmin_c[key] = (d, nops)
                        examples[key] = (ln, ins[j][3])
                    break
                if r in ins[j][1]:
                    break
    return ins, prod, cons, dmin_p, dmin_c, examples


res = {}
for tag, extra in (("slp", []), ("noslp", ["-fno-slp-vectorize"])):
    lines = listing(tag, extra)
    ins, prod, cons, dmin_p, dmin_c, ex = analyse(lines)
    c = collections.Counter(kind(i[0], i[3]) for i in ins)
    pk = collections.Counter(i[0] for i in ins if i[0].startswith("v_pk_"))
    print(f"== k_norm_bwd, {tag}: {len(ins)} instructions; " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())) + f"; s_nop {sum(1 for i in ins if i[0] == 's_nop')}, s_waitcnt {sum(1 for i in ins if i[0] == 's_waitcnt')}")
    print("   packed-fp32: " + (", ".join(f"{k} {v}" for k, v in sorted(pk.items())) or "none"))
    for key in sorted(prod):
        d, nops, waits = dmin_p[key]
        print(f"   {key:44s} x{prod[key]:3d}   min distance {d} instr ({nops} s_nop, {waits} s_waitcnt between)   e.g. `{ex[key][0]}` ... `{ex[key][1]}`")
    for key in sorted(cons):
        d, nops = dmin_c[key]
        print(f"   {key:44s} x{cons[key]:3d}   min distance {d} instr ({nops} s_nop between)   e.g. `{ex[key][0]}` ... `{ex[key][1]}`")
