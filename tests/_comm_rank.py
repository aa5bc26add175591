"""One rank of tests/test_comm_multirank.py (a child process; TEST INFRASTRUCTURE).

Runs csrc/comm.hip's multi-rank code on HOST buffers: libmdct_hip.so binds tests/fake_rccl.c through
MDCT_RCCL_LIB.  The rank fills ONLY its own shard of a canary-filled full-size buffer (with the oracle
standing in for the kernel -- the kernels are covered on the GPU; what is under test here is the piece
arithmetic and the collective calls), gathers, and compares EVERY byte with the oracle run over the full
range.  Prints one JSON line."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MDCT_NO_TORCH_PRELOAD"] = "1"

import oracle as O  # noqa: E402
from simd_dct_amd import _lib, synth  # noqa: E402
from simd_dct_amd.api import QUANTIZE_BASE  # noqa: E402

CANARY = 0xEE


def shard(lib, n, world, rank):
    b0, b1 = ctypes.c_size_t(), ctypes.c_size_t()
    lib.mdct_shard_rows(n, world, rank, ctypes.byref(b0), ctypes.byref(b1))
    return b0.value, b1.value


def main():
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    cases = json.loads(sys.argv[4])
    lib = _lib.load()
    fake = ctypes.CDLL(os.environ["MDCT_RCCL_LIB"])
    fake.fake_rccl_stats.argtypes = [ctypes.POINTER(ctypes.c_long)] * 4

    def stats():
        v = [ctypes.c_long() for _ in range(4)]
        fake.fake_rccl_stats(*[ctypes.byref(x) for x in v])
        return [x.value for x in v]  # collectives, allgathers, broadcasts, groups

    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        assert lib.mdct_comm_get_unique_id(ident) == 0, lib.mdct_last_error()
        with open(idfile + ".tmp", "wb") as f:
            f.write(ident.raw)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "rank 0 never published the id"
            time.sleep(0.01)
        with open(idfile, "rb") as f:
            ident = ctypes.create_string_buffer(f.read(), 128)
    comm = ctypes.c_void_p()
    assert lib.mdct_comm_init(ctypes.byref(comm), rank, world, ident) == 0, lib.mdct_last_error()
    assert lib.mdct_comm_rank(comm) == rank and lib.mdct_comm_world(comm) == world
    report = []
    for case in cases:
        before = stats()
        kind = case["kind"]
        if kind == "q32_rows":  # Q32 strips (simd_dct.cpp:2227-2230): block row = 8*W contiguous bytes
            W, n_rows = case["W"], case["n_rows"]
            H = 8 * n_rows
            img = synth.plane_u8_np(W, H, "photo")
            lut = (QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
            b0, b1 = shard(lib, n_rows, world, rank)
            buf = np.full(W * H, CANARY, dtype=np.uint8)
            O.q32_native(img, lut, W, H, b0, b1, out=buf)
            rc = lib.mdct_allgather_rows(comm, buf.ctypes.data, 8 * W, n_rows, None)
            _, want = O.q32_native(img, lut, W, H, 0, n_rows)
            pieces = sum(1 for q in range(world) if shard(lib, n_rows, world, q)[1] > shard(lib, n_rows, world, q)[0])
            equal = n_rows % world == 0
            expect = [1, 1, 0, 0] if equal else [pieces, 0, pieces, 1]
        elif kind == "i16_rows":  # int16 plane strips (config 4's layout): row_bytes = 8 * pitch * 2
            W, n_rows = case["W"], case["n_rows"]
            H = 8 * n_rows
            src = synth.plane_i16_np(W, H, "photo")
            b0, b1 = shard(lib, n_rows, world, rank)
            buf = np.full((H, W), np.int16(-4370), dtype=np.int16)  # 0xEEEE
            if b1 > b0:
                O.i16("fwd", src, W, H, by0=b0, by1=b1, out=buf)
            rc = lib.mdct_allgather_rows(comm, buf.ctypes.data, 8 * W * 2, n_rows, None)
            want = O.i16("fwd", src, W, H)
            pieces = sum(1 for q in range(world) if shard(lib, n_rows, world, q)[1] > shard(lib, n_rows, world, q)[0])
            equal = n_rows % world == 0
            expect = [1, 1, 0, 0] if equal else [pieces, 0, pieces, 1]
        elif kind == "stereo":  # 64 coefficient planes, one strided piece per rank in each (:1061-1099)
            W, H = case["W"], case["H"]
            img = synth.plane_u8_np(W, H, "photo")
            lut = (QUANTIZE_BASE * np.float32(8)).astype(np.float32)
            b0, b1 = shard(lib, H // 16, world, rank)
            buf = np.full(W * H, CANARY, dtype=np.uint8)
            if b1 > b0:  # the reference's inclusive range in its 2y units: stereo block rows b0 .. b1-1
                O.run_behaviour("stereo_sse", img, lut, W, H, 16 * b0, 16 * (b1 - 1), out=buf)
            rc = lib.mdct_allgather_stereo(comm, buf.ctypes.data, W, H, None)
            _, want = O.run_behaviour("stereo_sse", img, lut, W, H, 0, H)
            pieces = sum(1 for q in range(world) if shard(lib, H // 16, world, q)[1] > shard(lib, H // 16, world, q)[0])
            equal = (H // 16) % world == 0
            expect = [64, 64, 0, 1] if equal else [64 * pieces, 0, 64 * pieces, 1]
        else:
            raise ValueError(kind)
        after = stats()
        delta = [a - b for a, b in zip(after, before)]
        bad = int(np.count_nonzero(buf.reshape(-1).view(np.uint8) != np.asarray(want).reshape(-1).view(np.uint8)))
        report.append({"case": case, "rc": rc, "err": lib.mdct_last_error().decode() if rc else "", "mismatching_bytes": bad,
                       "collectives": delta, "expected_collectives": expect})
    assert lib.mdct_comm_destroy(comm) == 0
    print(json.dumps({"rank": rank, "world": world, "report": report}))


if __name__ == "__main__":
    main()
