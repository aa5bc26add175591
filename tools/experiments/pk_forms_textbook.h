// pk_forms_textbook.h -- NOT PRODUCT CODE.  The statement-per-operation alternates that used to sit in
// simd_dct_amd/csrc/mdct_kernels.hip under MDCT_PK_REORDER 0 / 1 (rounds 2-3): the packed butterflies of dct8_h in
// textbook order, dct8_v and the quantisers as one inline-asm statement per operation.  Same operations, operands and
// modifiers as the product's one-block / plain-vector forms, i.e. the readable specification and the A/B baseline
// (tools/exp_u8_r3.hip carries its own copies for timing).  Kept as text in the order it was cut; not compiled anywhere.
#if 0
  f32x2 s1, s2, d, e, pqp, pqm, r, t, m1, m2, m3, m4, t13, t57;
  MDCT_PKA(s1, a01, a67, MDCT_X);                                  // (p0+p7, p1+p6)
  MDCT_PKA(s2, a23, a45, MDCT_X);                                  // (p2+p5, p3+p4)
  MDCT_PKA(d, a01, a67, MDCT_X " neg_lo:[0,1] neg_hi:[1,0]");      // (p0-p7, p6-p1)
  MDCT_PKA(e, a23, a45, MDCT_X " neg_lo:[0,1] neg_hi:[1,0]");      // (p2-p5, p4-p3)
  MDCT_PKA(pqp, s1, s2, MDCT_X);                                   // (x07p+x34p, x16p+x25p)
  MDCT_PKA(pqm, s1, s2, MDCT_X " " MDCT_NEG_B);                    // (x07p-x34p, x16p-x25p)
  MDCT_PKA(o04, pqp, pqp, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]"); // (pp+qp, pp-qp)
  MDCT_PKM(r, pqm, K.be, MDCT_K_LL);                               // (Cb pm, Cb qm)
  MDCT_PKM(t, pqm, K.be, MDCT_K_HH);                               // (Ce pm, Ce qm)
  MDCT_PKA(o26, r, t, MDCT_X " neg_hi:[1,0]");                     // (Cb pm + Ce qm, Ce pm - Cb qm)
  MDCT_PKM(m1, d, K.af, MDCT_K_LH);                                // (Ca x07m, Cf x61m)
  MDCT_PKM(m2, d, K.cd, MDCT_K_LL);                                // (Cc x07m, Cc x61m)
  MDCT_PKM(m3, d, K.cd, MDCT_K_HH);                                // (Cd x07m, Cd x61m)
  MDCT_PKM(m4, d, K.af, MDCT_K_HL);                                // (Cf x07m, Ca x61m)
  MDCT_PKA(t13, m1, m2, MDCT_X " neg_lo:[0,1]");                   // (Ca x07m - Cc x61m, Cf x61m + Cc x07m)
  MDCT_PKA(t57, m3, m4, MDCT_X);                                   // (Cd x07m + Ca x61m, Cd x61m + Cf x07m)
  if constexpr (K1D == K_TRUE)
  { // sequential association: two more terms added one after the other
    f32x2 g1, g2, g3, g4, h13, h57;
    MDCT_PKM(g1, e, K.da, "op_sel:[0,0] op_sel_hi:[0,1]");         // (Cd x25m, Ca x25m)
    MDCT_PKM(g2, e, K.fd, "op_sel:[1,0] op_sel_hi:[1,1]");         // (Cf x43m, Cd x43m)
    MDCT_PKM(g3, e, K.fc, "op_sel:[0,0] op_sel_hi:[0,1]");         // (Cf x25m, Cc x25m)
    MDCT_PKM(g4, e, K.ca, "op_sel:[1,0] op_sel_hi:[1,1]");         // (Cc x43m, Ca x43m)
    MDCT_PKA(h13, t13, g1, "neg_hi:[0,1]");                        // (t1 + Cd x25m, t3 - Ca x25m)
    MDCT_PKA(o13, h13, g2, "neg_lo:[0,1]");                        // (.. - Cf x43m, .. + Cd x43m)
    MDCT_PKA(h57, t57, g3, "");                                    // (t5 + Cf x25m, t7 + Cc x25m)
    MDCT_PKA(o57, h57, g4, "neg_lo:[0,1]");                        // (.. - Cc x43m, .. + Ca x43m)
  }
  else
  {
    f32x2 n1, n2, n3, n4, u13, u57;
    MDCT_PKM(n1, e, K.cd, MDCT_K_HH);                              // (Cd x25m, Cd x43m)
    MDCT_PKM(n2, e, K.af, MDCT_K_LH);                              // (Ca x25m, Cf x43m)
    MDCT_PKM(n3, e, K.af, MDCT_K_HL);                              // (Cf x25m, Ca x43m)
    MDCT_PKM(n4, e, K.cd, MDCT_K_LL);                              // (Cc x25m, Cc x43m)
    MDCT_PKA(u57, n3, n4, MDCT_X " neg_lo:[0,1]");                 // (Cf x25m - Cc x43m, Ca x43m + Cc x25m)
    if constexpr (K1D == K_AVX)
    {
      MDCT_PKA(u13, n1, n2, MDCT_X " neg_lo:[0,1]");               // (Cd x25m - Cf x43m, Cd x43m + Ca x25m)
      MDCT_PKA(o13, t13, u13, "neg_hi:[0,1]");                     // (t1 + u1, t3 - u3): the k=3 quirk of :2181
    }
    else
    {
      static_assert(K1D == K_SSE, "unknown 1-D kernel");
      MDCT_PKA(u13, n1, n2, MDCT_X " neg_hi:[0,1]");               // (Cd x25m + Cf x43m [k=1 quirk, :550], Cd x43m - Ca x25m)
      MDCT_PKA(o13, t13, u13, "");
    }
    MDCT_PKA(o57, t57, u57, "");                                   // (t5 + u5, t7 + u7)
  }
  MDCT_PKM(o04, o04, K.nm, MDCT_K_LL);
  MDCT_PKM(o26, o26, K.nm, MDCT_K_LL);
  MDCT_PKM(o13, o13, K.nm, MDCT_K_LL);
  MDCT_PKM(o57, o57, K.nm, MDCT_K_LL);
  // products and sums software-pipelined: mul A(i+1) sits between mul B(i) and add(i), so no statement reads its predecessor
  f32x2 x07p, x16p, x25p, x34p, x07m, x61m, x25m, x43m, pp, pm, qp, qm, o0, o1, o2, o3, o4, o5, o6, o7;
  MDCT_PKA(x07p, p[0], p[7], ""); MDCT_PKA(x16p, p[1], p[6], ""); MDCT_PKA(x25p, p[2], p[5], ""); MDCT_PKA(x34p, p[3], p[4], "");
  MDCT_PKA(x07m, p[0], p[7], MDCT_NEG_B); MDCT_PKA(x61m, p[6], p[1], MDCT_NEG_B);
  MDCT_PKA(x25m, p[2], p[5], MDCT_NEG_B); MDCT_PKA(x43m, p[4], p[3], MDCT_NEG_B);
  MDCT_PKA(pp, x07p, x34p, ""); MDCT_PKA(pm, x07p, x34p, MDCT_NEG_B);
  MDCT_PKA(qp, x16p, x25p, ""); MDCT_PKA(qm, x16p, x25p, MDCT_NEG_B);
  f32x2 a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, t1, t3, t5, t7;
  MDCT_PKM(a1, pm, K.be, MDCT_K_LL);                       // Cb pm
  MDCT_PKA(o0, pp, qp, "");
  MDCT_PKM(b1, qm, K.be, MDCT_K_HH);                       // Ce qm
  MDCT_PKA(o4, pp, qp, MDCT_NEG_B);
  MDCT_PKM(a2, pm, K.be, MDCT_K_HH);                       // Ce pm
  MDCT_PKA(o2, a1, b1, "");                                // Cb pm + Ce qm
  MDCT_PKM(b2, qm, K.be, MDCT_K_LL);                       // Cb qm
  MDCT_PKM(a3, x07m, K.af, MDCT_K_LL);                     // Ca x07m
  MDCT_PKA(o6, a2, b2, MDCT_NEG_B);                        // Ce pm - Cb qm
  MDCT_PKM(b3, x61m, K.cd, MDCT_K_LL);                     // Cc x61m
  MDCT_PKM(a4, x07m, K.cd, MDCT_K_LL);                     // Cc x07m
  MDCT_PKA(t1, a3, b3, MDCT_NEG_B);                        // Ca x07m - Cc x61m
  MDCT_PKM(b4, x61m, K.af, MDCT_K_HH);                     // Cf x61m
  MDCT_PKM(a5, x07m, K.cd, MDCT_K_HH);                     // Cd x07m
  MDCT_PKA(t3, a4, b4, "");                                // Cc x07m + Cf x61m
  MDCT_PKM(b5, x61m, K.af, MDCT_K_LL);                     // Ca x61m
  MDCT_PKM(a6, x07m, K.af, MDCT_K_HH);                     // Cf x07m
  MDCT_PKA(t5, a5, b5, "");                                // Cd x07m + Ca x61m
  MDCT_PKM(b6, x61m, K.cd, MDCT_K_HH);                     // Cd x61m
  if constexpr (K1D == K_TRUE)
  { // ((t + c1 x25m) +- c2 x43m), :166-171
    f32x2 c1, c3, c5, c7, d1, d3, d5, d7;
    MDCT_PKM(c1, x25m, K.cd, MDCT_K_HH);                   // Cd x25m
    MDCT_PKA(t7, a6, b6, "");                              // Cf x07m + Cd x61m
    MDCT_PKM(c3, x25m, K.af, MDCT_K_LL);                   // Ca x25m
    MDCT_PKM(c5, x25m, K.af, MDCT_K_HH);                   // Cf x25m
    MDCT_PKA(t1, t1, c1, "");
    MDCT_PKM(c7, x25m, K.cd, MDCT_K_LL);                   // Cc x25m
    MDCT_PKA(t3, t3, c3, MDCT_NEG_B);
    MDCT_PKM(d1, x43m, K.af, MDCT_K_HH);                   // Cf x43m
    MDCT_PKA(t5, t5, c5, "");
    MDCT_PKM(d3, x43m, K.cd, MDCT_K_HH);                   // Cd x43m
    MDCT_PKA(t7, t7, c7, "");
    MDCT_PKM(d5, x43m, K.cd, MDCT_K_LL);                   // Cc x43m
    MDCT_PKA(o1, t1, d1, MDCT_NEG_B);
    MDCT_PKM(d7, x43m, K.af, MDCT_K_LL);                   // Ca x43m
    MDCT_PKA(o3, t3, d3, "");
    MDCT_PKA(o5, t5, d5, MDCT_NEG_B);
    MDCT_PKA(o7, t7, d7, "");
  }
  else
  {
    f32x2 c7, d7, c8, d8, c9, d9, c10, d10, u1, u3, u5, u7;
    MDCT_PKM(c7, x25m, K.af, MDCT_K_HH);                   // Cf x25m
    MDCT_PKA(t7, a6, b6, "");                              // Cf x07m + Cd x61m
    MDCT_PKM(d7, x43m, K.cd, MDCT_K_LL);                   // Cc x43m
    MDCT_PKM(c8, x25m, K.cd, MDCT_K_LL);                   // Cc x25m
    MDCT_PKA(u5, c7, d7, MDCT_NEG_B);                      // Cf x25m - Cc x43m
    MDCT_PKM(d8, x43m, K.af, MDCT_K_LL);                   // Ca x43m
    MDCT_PKM(c9, x25m, K.cd, MDCT_K_HH);                   // Cd x25m
    MDCT_PKA(u7, c8, d8, "");                              // Cc x25m + Ca x43m
    MDCT_PKM(d9, x43m, K.af, MDCT_K_HH);                   // Cf x43m
    MDCT_PKM(c10, x25m, K.af, MDCT_K_LL);                  // Ca x25m
    if constexpr (K1D == K_AVX)
      MDCT_PKA(u1, c9, d9, MDCT_NEG_B);                    // Cd x25m - Cf x43m
    else
      MDCT_PKA(u1, c9, d9, "");                            // Cd x25m + Cf x43m (k=1 quirk, :550)
    MDCT_PKM(d10, x43m, K.cd, MDCT_K_HH);                  // Cd x43m
    MDCT_PKA(o5, t5, u5, "");
    if constexpr (K1D == K_AVX)
      MDCT_PKA(u3, c10, d10, "");                          // Ca x25m + Cd x43m
    else
      MDCT_PKA(u3, d10, c10, MDCT_NEG_B);                  // Cd x43m - Ca x25m
    MDCT_PKA(o7, t7, u7, "");
    MDCT_PKA(o1, t1, u1, "");
    if constexpr (K1D == K_AVX)
      MDCT_PKA(o3, t3, u3, MDCT_NEG_B);                    // the k=3 quirk of :2181
    else
      MDCT_PKA(o3, t3, u3, "");
  }
  MDCT_PKM(p[0], o0, K.nm, MDCT_K_LL); MDCT_PKM(p[4], o4, K.nm, MDCT_K_LL); MDCT_PKM(p[2], o2, K.nm, MDCT_K_LL); MDCT_PKM(p[6], o6, K.nm, MDCT_K_LL);
  MDCT_PKM(p[5], o5, K.nm, MDCT_K_LL); MDCT_PKM(p[7], o7, K.nm, MDCT_K_LL); MDCT_PKM(p[1], o1, K.nm, MDCT_K_LL); MDCT_PKM(p[3], o3, K.nm, MDCT_K_LL);
  f32x2 x07p, x16p, x25p, x34p, x07m, x61m, x25m, x43m, pp, pm, qp, qm, o0, o4, a, b, o2, o6;
  MDCT_PKA(x07p, p[0], p[7], ""); MDCT_PKA(x16p, p[1], p[6], ""); MDCT_PKA(x25p, p[2], p[5], ""); MDCT_PKA(x34p, p[3], p[4], "");
  MDCT_PKA(x07m, p[0], p[7], MDCT_NEG_B); MDCT_PKA(x61m, p[6], p[1], MDCT_NEG_B);
  MDCT_PKA(x25m, p[2], p[5], MDCT_NEG_B); MDCT_PKA(x43m, p[4], p[3], MDCT_NEG_B);
  MDCT_PKA(pp, x07p, x34p, ""); MDCT_PKA(pm, x07p, x34p, MDCT_NEG_B);
  MDCT_PKA(qp, x16p, x25p, ""); MDCT_PKA(qm, x16p, x25p, MDCT_NEG_B);
  MDCT_PKA(o0, pp, qp, ""); MDCT_PKA(o4, pp, qp, MDCT_NEG_B);
  MDCT_PKM(a, pm, K.be, MDCT_K_LL); MDCT_PKM(b, qm, K.be, MDCT_K_HH); MDCT_PKA(o2, a, b, "");          // Cb pm + Ce qm
  MDCT_PKM(a, pm, K.be, MDCT_K_HH); MDCT_PKM(b, qm, K.be, MDCT_K_LL); MDCT_PKA(o6, a, b, MDCT_NEG_B);  // Ce pm - Cb qm
  f32x2 t1, t3, t5, t7, c, dd, o1, o3, o5, o7;
  MDCT_PKM(a, x07m, K.af, MDCT_K_LL); MDCT_PKM(b, x61m, K.cd, MDCT_K_LL); MDCT_PKA(t1, a, b, MDCT_NEG_B);   // Ca x07m - Cc x61m
  MDCT_PKM(a, x07m, K.cd, MDCT_K_LL); MDCT_PKM(b, x61m, K.af, MDCT_K_HH); MDCT_PKA(t3, a, b, "");           // Cc x07m + Cf x61m
  MDCT_PKM(a, x07m, K.cd, MDCT_K_HH); MDCT_PKM(b, x61m, K.af, MDCT_K_LL); MDCT_PKA(t5, a, b, "");           // Cd x07m + Ca x61m
  MDCT_PKM(a, x07m, K.af, MDCT_K_HH); MDCT_PKM(b, x61m, K.cd, MDCT_K_HH); MDCT_PKA(t7, a, b, "");           // Cf x07m + Cd x61m
  if constexpr (K1D == K_TRUE)
  { // ((t + c1 x25m) +- c2 x43m), :166-171
    MDCT_PKM(c, x25m, K.cd, MDCT_K_HH); MDCT_PKA(t1, t1, c, "");         MDCT_PKM(dd, x43m, K.af, MDCT_K_HH); MDCT_PKA(o1, t1, dd, MDCT_NEG_B);
    MDCT_PKM(c, x25m, K.af, MDCT_K_LL); MDCT_PKA(t3, t3, c, MDCT_NEG_B); MDCT_PKM(dd, x43m, K.cd, MDCT_K_HH); MDCT_PKA(o3, t3, dd, "");
    MDCT_PKM(c, x25m, K.af, MDCT_K_HH); MDCT_PKA(t5, t5, c, "");         MDCT_PKM(dd, x43m, K.cd, MDCT_K_LL); MDCT_PKA(o5, t5, dd, MDCT_NEG_B);
    MDCT_PKM(c, x25m, K.cd, MDCT_K_LL); MDCT_PKA(t7, t7, c, "");         MDCT_PKM(dd, x43m, K.af, MDCT_K_LL); MDCT_PKA(o7, t7, dd, "");
  }
  else
  {
    f32x2 u1, u3, u5, u7;
    MDCT_PKM(c, x25m, K.af, MDCT_K_HH); MDCT_PKM(dd, x43m, K.cd, MDCT_K_LL); MDCT_PKA(u5, c, dd, MDCT_NEG_B); // Cf x25m - Cc x43m
    MDCT_PKM(c, x25m, K.cd, MDCT_K_LL); MDCT_PKM(dd, x43m, K.af, MDCT_K_LL); MDCT_PKA(u7, c, dd, "");         // Cc x25m + Ca x43m
    if constexpr (K1D == K_AVX)
    {
      MDCT_PKM(c, x25m, K.cd, MDCT_K_HH); MDCT_PKM(dd, x43m, K.af, MDCT_K_HH); MDCT_PKA(u1, c, dd, MDCT_NEG_B); // Cd x25m - Cf x43m
      MDCT_PKM(c, x25m, K.af, MDCT_K_LL); MDCT_PKM(dd, x43m, K.cd, MDCT_K_HH); MDCT_PKA(u3, c, dd, "");         // Ca x25m + Cd x43m
      MDCT_PKA(o1, t1, u1, ""); MDCT_PKA(o3, t3, u3, MDCT_NEG_B);
    }
    else
    {
      MDCT_PKM(c, x25m, K.cd, MDCT_K_HH); MDCT_PKM(dd, x43m, K.af, MDCT_K_HH); MDCT_PKA(u1, c, dd, "");         // Cd x25m + Cf x43m (quirk)
      MDCT_PKM(c, x43m, K.cd, MDCT_K_HH); MDCT_PKM(dd, x25m, K.af, MDCT_K_LL); MDCT_PKA(u3, c, dd, MDCT_NEG_B); // Cd x43m - Ca x25m
      MDCT_PKA(o1, t1, u1, ""); MDCT_PKA(o3, t3, u3, "");
    }
    MDCT_PKA(o5, t5, u5, ""); MDCT_PKA(o7, t7, u7, "");
  }
  MDCT_PKM(p[0], o0, K.nm, MDCT_K_LL); MDCT_PKM(p[1], o1, K.nm, MDCT_K_LL); MDCT_PKM(p[2], o2, K.nm, MDCT_K_LL); MDCT_PKM(p[3], o3, K.nm, MDCT_K_LL);
  MDCT_PKM(p[4], o4, K.nm, MDCT_K_LL); MDCT_PKM(p[5], o5, K.nm, MDCT_K_LL); MDCT_PKM(p[6], o6, K.nm, MDCT_K_LL); MDCT_PKM(p[7], o7, K.nm, MDCT_K_LL);
      MDCT_PKM(m, col[j][v], qp, MDCT_K_LH);
        MDCT_PKA(t, m, K.nm, MDCT_K_HH); // + (magic, magic)
      MDCT_PKM(a01, a01, K.bias, MDCT_K_LL);
      MDCT_PKM(a23, a23, K.bias, MDCT_K_LL);
      MDCT_PKM(a45, a45, K.bias, MDCT_K_LL);
      MDCT_PKM(a67, a67, K.bias, MDCT_K_LL);
      MDCT_PKM(v, P[j][m], qp, MDCT_K_LH);
        MDCT_PKA(v, v, K.bias, MDCT_K_HH);
          MDCT_PKA(t, v, K.nm, MDCT_K_HH);
        MDCT_PKA(v, v, K.bias, MDCT_K_LL);
        MDCT_PKM(x, v, K.bias, MDCT_K_HH);
        MDCT_PKA(t, x, K.nm, MDCT_K_HH);
        MDCT_PKA(r, t, K.nm, MDCT_K_HH " " MDCT_NEG_B);
        MDCT_PKA(d, x, r, MDCT_NEG_B);
#endif
