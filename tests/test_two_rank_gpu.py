"""N > 1 with the REAL kernels, in the driver-run GPU suite.

tests/test_sharding_gloo.py covers the partition / gather logic on the CPU with the oracle standing in for the
kernel; here two (and four) fresh processes -- one per rank, as bench.py and `torch.distributed.run` start the
engine -- each call mdct_init and transform their shard_rows range of a q32 plane, an int16 plane and a stereo
buffer with the HIP kernels, exchange the shards over gloo and compare every byte with the oracle's whole-plane
output.  The box has one GPU: the ranks share it (RCCL itself refuses two ranks on one device, so the collective
here is gloo; csrc/comm.hip's RCCL calls are covered by test_comm.py at world 1 and test_comm_multirank.py).
The shard arithmetic is the reference's own hook, startY/endY (simd_dct.cpp:2245-2255)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")


def run_world(world, timeout=600):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
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
        assert rc == 0, f"rank {r} of {world} exited {rc}\n{o}\n{e[-3000:]}"
    return [json.loads(o.strip().splitlines()[-1]) for _, o, _ in outs]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_transform_their_shards_with_the_hip_kernels_and_gather(world):
    reports = run_world(world)
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        rep = r["report"]
        assert rep.pop("native_so") == "libmdct_hip.so"
        assert len(rep) >= 12, rep
        bad = [k for k, v in rep.items() if v is not True]
        assert not bad, (r["rank"], bad)
