"""SHA-256 of the ORACLE's outputs for the default synthetic planes of the engine-own configurations (BASELINE.json
configs[2], [3], [4]: int16 / float32 transforms the reference does not have -- "parity unpinned" by the reference, pinned
by oracle/dct_oracle.c, which tests/test_oracle.py checks against a double-precision DCT-II by definition).

    python tests/golden/make_engine_hashes.py      (CPU only; ~1 minute on 8 cores)

bench.py compares the device outputs of its timed launches with these hashes, so that the driver-run JSON line carries a
verdict for every configuration without calling the oracle inside bench.py's GPU legs.  Data only: hashes."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as O  # noqa: E402
from simd_dct_amd import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if "--only-u8" in sys.argv:  # add / refresh configs[2]'s 8-bit hashes without recomputing the rest (~10 s)
        path = os.path.join(HERE, "engine_own_sha256.json")
        out = json.load(open(path))
        c3u = {}
        for (w, h, so, tab) in synth.CONFIG3_PLANES:
            src = synth.plane_u8_np(w, h, "photo", seed=synth.SEED + so)
            lut = synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA
            c3u[f"roundtrip_u8__{w}x{h}__seed+{so}__{tab}"] = sha(O.roundtrip_u8(src, w, h, lut=lut, level_shift=True, threads=O.host_threads()))
        out["config3_420_u8"] = c3u
        json.dump(out, open(path, "w"), indent=1)
        print(json.dumps(c3u, indent=1))
        return
    out = {"what": "sha256 of oracle/dct_oracle.c outputs (engine-own arithmetic, DESIGN.md 4.2) for simd_dct_amd.synth planes; inputs: plane_i16 'photo', 8 bit"}
    # configs[2]: fused round trip with the Annex K.1 tables, every plane of the frame
    c3 = {}
    for (w, h, so, tab) in synth.CONFIG3_PLANES:
        src = synth.plane_i16_np(w, h, "photo", seed=synth.SEED + so)
        lut = synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA
        c3[f"roundtrip__{w}x{h}__seed+{so}__{tab}"] = sha(O.i16_par("roundtrip", src, w, h, lut=lut))
        c3[f"fwd__{w}x{h}__seed+{so}__{tab}"] = sha(O.i16_par("fwd", src, w, h, lut=lut))
    out["config3_420"] = c3
    # configs[2] as SURVEY.md 8(d) states it: the same frame as 8-bit planes, u8 in -> u8 out, level shift on (mdct_roundtrip_u8_batch)
    c3u = {}
    for (w, h, so, tab) in synth.CONFIG3_PLANES:
        src = synth.plane_u8_np(w, h, "photo", seed=synth.SEED + so)
        lut = synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA
        c3u[f"roundtrip_u8__{w}x{h}__seed+{so}__{tab}"] = sha(O.roundtrip_u8(src, w, h, lut=lut, level_shift=True, threads=O.host_threads()))
    out["config3_420_u8"] = c3u
    # configs[3]: forward, no table, 4096^2 planes with seeds SEED + 100 + p; the first and the last of the 256
    c4 = {}
    for p in (0, 1, 255):
        src = synth.plane_i16_np(4096, 4096, "photo", seed=synth.SEED + 100 + p)
        c4[f"fwd__4096x4096__seed+{100 + p}"] = sha(O.i16_par("fwd", src, 4096, 4096))
    out["config4_planes"] = c4
    # configs[4]: float32 forward DCT-II of float(int16 photo plane), 8192^2
    src = synth.plane_i16_np(8192, 8192, "photo", seed=synth.SEED).astype(np.float32)
    f = O.f32_par("fwd", src, 8192, 8192)
    out["config5_f32"] = {"fwd__8192x8192__seed+0": sha(f)}
    # the double-precision reference of the same plane, as the worst block-relative deviation of the oracle (config 5's tolerance is
    # stated on the GPU output in tests; recorded here so that the hash is known to belong to an output inside it)
    ref = O.f32_par("f64ref", src, 8192, 8192)
    blk = lambda a: a.reshape(1024, 8, 1024, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    rel = (np.abs(blk(f.astype(np.float64)) - blk(ref)).max(1) / np.abs(blk(ref)).max(1)).max()
    out["config5_f32"]["max_err_over_block_max_vs_double"] = float(rel)
    assert rel < 1e-5
    with open(os.path.join(HERE, "engine_own_sha256.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
