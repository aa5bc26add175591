"""Child process of tests/test_gpu_parity.py::test_shim_autopin_* (TEST INFRASTRUCTURE): MDCT_SHIM_AUTOPIN is read once per process, so the
opt-in is exercised in a process of its own.  Host buffers passed again and again to the three reference functions: from the third
sighting on they are page-locked in place (hipPointerGetAttributes says so), every output stays the oracle's, canaries stay canaries,
mdct_shim_release() lets everything go again.  Prints one JSON line."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MDCT_NO_TORCH_PRELOAD"] = "1"

import oracle as O  # noqa: E402
from simd_dct_amd import _lib, synth  # noqa: E402
from simd_dct_amd.api import QUANTIZE_BASE  # noqa: E402

CANARY = 0xA5


def memory_type(hip, ptr):
    """hipPointerGetAttributes(...).type: 0 unregistered, 1 host (pinned / registered), 2 device"""
    attr = (ctypes.c_int * 16)()  # hipPointerAttribute_t starts with `type`; 64 bytes are enough room
    rc = hip.hipPointerGetAttributes(attr, ctypes.c_void_p(ptr))
    return attr[0] if rc == 0 else 0


def main():
    lib = _lib.load()
    hip_path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)  # the runtime libmdct_hip.so is bound to
    hip = ctypes.CDLL(hip_path)
    assert lib.mdct_init(0) == 0
    W, H = 4096, 1024
    img = np.ascontiguousarray(synth.plane_u8_np(W, H, "photo").reshape(-1))
    rep = {"env": os.environ.get("MDCT_SHIM_AUTOPIN", "")}
    f32p = ctypes.POINTER(ctypes.c_float)
    for which, beh, scale in ((0, "q32_avx", 2000), (1, "stereo_sse", 8), (2, "encq_sse", 8)):
        lut = np.ascontiguousarray((QUANTIZE_BASE * np.float32(scale)).astype(np.float32))
        out = np.full(W * H, CANARY, dtype=np.uint8)
        want = np.full(W * H, CANARY, dtype=np.uint8)
        O.run_behaviour(beh, img, lut, W, H, 0, H, out=want)
        types, ok = [], True
        for call in range(6):
            out[:] = CANARY
            rc = lib.mdct_shim_call(which, img.ctypes.data, out.ctypes.data, lut.ctypes.data_as(f32p), W, H, 0, H)
            ok = ok and rc == 0 and bool(np.array_equal(out, want))
            types.append((memory_type(hip, img.ctypes.data), memory_type(hip, out.ctypes.data)))
        # disjoint row ranges of the same planes (the reference's multi-core hook), shorter reaches than what is registered
        out[:] = CANARY
        for (y0, y1) in ((0, 255), (256, 511), (512, H)):
            ok = ok and lib.mdct_shim_call(which, img.ctypes.data, out.ctypes.data, lut.ctypes.data_as(f32p), W, H, y0, y1) == 0
        ok = ok and bool(np.array_equal(out, want))
        rep[beh] = {"ok": ok, "types": types}
        lib.mdct_shim_release()
        rep[beh]["after_release"] = (memory_type(hip, img.ctypes.data), memory_type(hip, out.ctypes.data))
        # and the buffers work again afterwards (three fresh sightings)
        out[:] = CANARY
        ok2 = lib.mdct_shim_call(which, img.ctypes.data, out.ctypes.data, lut.ctypes.data_as(f32p), W, H, 0, H) == 0 and bool(np.array_equal(out, want))
        rep[beh]["ok_after_release"] = ok2
        lib.mdct_shim_release()
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
