#!/usr/bin/env python3
"""Static instruction mix of every product kernel, by vector-ISSUE class (what bench.py's `valu_floor_ms` is built from):
compiles the kernel sources with the product's flags to assembly and counts, per kernel,
    packed  v_pk_*                                   (two fp32 per register pair: ~2.1 ns per wave-instruction per SIMD)
    plain   VOP1/VOP2 fp32/int: v_add/sub/mul_f32, v_mov_b32, v_add_u32, shifts, and/or ... (~1.0 ns)
    other   converts, v_med3, v_perm, v_rndne, v_fma, v_sat_pk, DPP/SDWA forms, every 3-operand VOP3 (~1.7 ns)
(costs: tools/valubench2, line VALU_ISSUE_COSTS_NS).  Straight-line kernels execute exactly these counts per wave -- compare with the
PMC's SQ_INSTS_VALU / SQ_WAVES in profiles/traffic.json.   python3 tools/isa_classes.py > profiles/r05_isa_classes.json"""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "simd_dct_amd", "csrc"), "-S", "--cuda-device-only"]
PLAIN = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32",
         "v_or_b32", "v_xor_b32", "v_not_b32", "v_max_f32", "v_min_f32", "v_add_co_u32", "v_addc_co_u32", "v_cndmask_b32", "v_mul_u32_u24", "v_mul_i32_i24", "v_mov_b64", "v_add_u16")


def classify(op):
    if op.startswith("v_pk_"):
        return "packed"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith(("_dpp", "_sdwa", "_e64")):
        return "other"
    return "plain" if base in PLAIN else "other"


out = {}
for src in ("mdct_kernels.hip", "stages.hip"):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(ROOT, "simd_dct_amd", "csrc", src), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        s = open(asm).read()
    funcs = re.split(r"\n(_Z[\w]+):[^\n]*\n", s)
    names = [funcs[i] for i in range(1, len(funcs), 2)]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    for k, i in enumerate(range(1, len(funcs), 2)):
        rest = funcs[i + 1]
        if ".amdhsa_kernel" not in rest and "s_endpgm" not in rest:
            continue
        body = rest.split(".Lfunc_end")[0]
        ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = {"packed": 0, "plain": 0, "other": 0}
        for op in ins:
            if op.startswith("v_"):
                c[classify(op)] += 1
        m = lambda key: (re.search(r"; %s: (\d+)" % key, rest) or [None, None])[1]
        out[dem[k].strip()] = dict(c, valu=sum(c.values()), salu=sum(1 for o in ins if o.startswith("s_") and not o.startswith(("s_nop", "s_waitcnt", "s_load"))),
                                   s_nop=sum(1 for o in ins if o == "s_nop"), lds=sum(1 for o in ins if o.startswith("ds_")),
                                   vmem=sum(1 for o in ins if o.startswith(("global_", "buffer_", "flat_"))), vgprs=int(m("NumVgprs") or 0), occupancy=int(m("Occupancy") or 0))
json.dump({"what": "static instruction counts per kernel (one wave's straight-line stream), tools/isa_classes.py", "kernels": out}, sys.stdout, indent=1, sort_keys=True)
print()
