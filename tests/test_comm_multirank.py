"""csrc/comm.hip with world > 1, on the CPU.

The gather code of the C-ABI (gather_pieces: ONE in-place ncclAllGather for equal shards, one grouped
ncclBroadcast per rank for ragged ones, 64 grouped collectives for the stereo layout) is host code that
only forwards pointers to RCCL, so it runs unchanged on host buffers when libmdct_hip.so binds
tests/fake_rccl.c (MDCT_RCCL_LIB) -- the eight RCCL symbols over POSIX shared memory between processes.
One process per rank, like the real thing.  Reference lines this layout arithmetic mirrors: the row-range
hook simd_dct.cpp:2245-2255 and the stereo cursors :1061-1099."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")


def run_world(fake, world, cases, tmp_path, timeout=300):
    idfile = str(tmp_path / f"id_{world}")
    env = dict(os.environ, MDCT_RCCL_LIB=fake, MDCT_NO_TORCH_PRELOAD="1", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_comm_rank.py"), str(r), str(world), idfile, json.dumps(cases)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f"rank {r} of {world} exited {rc}\n{o}\n{e}"
    return [json.loads(o.strip().splitlines()[-1]) for _, o, _ in outs]


def check(reports, world):
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        for c in r["report"]:
            assert c["rc"] == 0, (r["rank"], c)
            assert c["mismatching_bytes"] == 0, (r["rank"], c)  # every byte, on every rank
            assert c["collectives"] == c["expected_collectives"], (r["rank"], c)  # the branch that was meant to run did


@pytest.mark.parametrize("world", [2, 3, 8])
def test_allgather_rows_equal_and_ragged(fake_rccl, tmp_path, world):
    """n_rows 7, 8, 1023 over world 2, 3, 8: equal shards take the in-place all-gather, ragged ones the grouped
    broadcasts (with empty ranks when n_rows < world); Q32 strips and int16 strips"""
    cases = [{"kind": "q32_rows", "W": 64, "n_rows": n} for n in (7, 8, 1023, 24)]
    cases += [{"kind": "i16_rows", "W": 64, "n_rows": n} for n in (7, 48)]
    check(run_world(fake_rccl, world, cases, tmp_path), world)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_allgather_stereo_64_pieces(fake_rccl, tmp_path, world):
    """the stereo layout's 64 strided pieces per rank: H/16 = 6 (ragged for 8 with empty ranks, equal for 2 and 3),
    24 (equal for all three) and 7 (ragged for all three)"""
    cases = [{"kind": "stereo", "W": 128, "H": 16 * n} for n in (6, 24, 7)]
    check(run_world(fake_rccl, world, cases, tmp_path), world)


def test_a_rank_that_disagrees_fails_loudly_instead_of_hanging(fake_rccl, tmp_path):
    """the stand-in cross-checks (kind, bytes, root) of every collective across ranks: a shard-arithmetic bug that
    made ranks issue different collectives would be an error, not a hang or silent corruption"""
    code = r'''
import ctypes, os, sys, time
lib = ctypes.CDLL(os.environ["MDCT_RCCL_LIB"])
class Id(ctypes.Structure): _fields_ = [("internal", ctypes.c_char * 128)]
rank, idfile = int(sys.argv[1]), sys.argv[2]
ident = Id()
if rank == 0:
    assert lib.ncclGetUniqueId(ctypes.byref(ident)) == 0
    open(idfile + ".tmp", "wb").write(bytes(ident)); os.rename(idfile + ".tmp", idfile)
else:
    while not os.path.exists(idfile): time.sleep(0.01)
    ctypes.memmove(ctypes.byref(ident), open(idfile, "rb").read(), 128)
comm = ctypes.c_void_p()
lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Id, ctypes.c_int]
assert lib.ncclCommInitRank(ctypes.byref(comm), 2, ident, rank) == 0
buf = ctypes.create_string_buffer(64)
lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
rc = lib.ncclAllGather(buf, buf, 16 if rank == 0 else 8, 1, comm, None)   # ranks disagree on the count
sys.exit(0 if rc != 0 else 7)
'''
    idfile = str(tmp_path / "id_bad")
    env = dict(os.environ, MDCT_RCCL_LIB=fake_rccl)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), idfile], env=env) for r in range(2)]
    assert [p.wait(timeout=200) for p in procs] == [0, 0]
