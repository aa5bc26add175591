"""The scalar tiers' quantiser issues  trunc(fma(c, 255, pred(0.5)))  where the reference writes  (uint8_t)roundf(c * 255.f)
(simd_dct.cpp:245, :362).  tools/check_roundf_forms.py proves the two equal for all 2^30 + 1 floats c in [0, 1]
(profiles/r04_roundf_forms_exhaustive.log); this is the quick slice of it that runs with every CPU test round: every float within 64 ulps
of a c whose product lands on a half-integer, four million random ones, and the one c that breaks the forms with 0.5."""
import numpy as np


def _roundf(x64):
    fl = np.floor(x64)
    return fl + ((x64 - fl) >= 0.5)


def test_trunc_fma_pred_half_equals_roundf_of_the_float_product():
    h = np.nextafter(np.float32(0.5), np.float32(0))
    rng = np.random.default_rng(4)
    near = []
    for k in range(255):
        c = np.float32((k + 0.5) / 255.0)
        bits = int(np.array([c]).view(np.uint32)[0])
        near.append(np.arange(max(bits - 64, 0), min(bits + 65, 0x3F800001), dtype=np.uint32))
    pats = np.concatenate(near + [rng.integers(0, 0x3F800001, 4_000_000, dtype=np.uint32), np.array([0x3B008080, 0, 0x3F800000], dtype=np.uint32)])
    c = pats.view(np.float32)
    x = c * np.float32(255.0)
    want = _roundf(x.astype(np.float64))
    fused = (c.astype(np.float64) * 255.0 + float(h)).astype(np.float32)  # c * 255 is exact in float64: one rounding, like v_pk_fma_f32
    assert np.array_equal(np.trunc(fused.astype(np.float64)), want)
    naive = (c.astype(np.float64) * 255.0 + 0.5).astype(np.float32)
    bad = np.nonzero(np.trunc(naive.astype(np.float64)) != want)[0]
    assert [hex(int(p)) for p in np.unique(pats[bad])] == ["0x3b008080"]  # why the addend is pred(0.5), not 0.5
