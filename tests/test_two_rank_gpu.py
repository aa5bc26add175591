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


@pytest.mark.gpu
def test_cxx_cli_two_ranks_on_the_box(tmp_path):
    """`simd_dct_cli --gpus 2`: one forked process per rank, block rows sharded by the reference's startY/endY hook, RCCL all-gather through the
    C-ABI (csrc/comm.hip).  On a box with two GPUs this is the real thing and the dump must be the oracle's bytes; on the one-GPU test box the
    ranks share the device (MDCT_CLI_SHARE_DEVICES=1) and RCCL decides -- it refuses two ranks on one device, and then this test SKIPS with
    RCCL's reason (the gather code itself runs at world 2 / 3 / 8 in tests/test_comm_multirank.py, over real RCCL at world 1 in tests/test_comm.py)."""
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import __graft_entry__ as G
    import oracle as O
    from simd_dct_amd import synth
    from simd_dct_amd.api import QUANTIZE_BASE

    cli = G.build_cli()
    W, H = 1024, 512
    dump = tmp_path / "g2.bin"
    env = dict(os.environ, MDCT_CLI_GPUS_TIMEOUT="90")
    if torch.cuda.device_count() < 2:
        env["MDCT_CLI_SHARE_DEVICES"] = "1"
    r = subprocess.run([cli, "synthetic:photo", str(W), str(H), "--mode", "enc-quant32", "--quality", "8", "--runs", "3", "--gpus", "2", "--to", str(dump)],
                       capture_output=True, text=True, timeout=240, env=env)
    if r.returncode != 0:
        said = (r.stdout + "\n" + r.stderr).splitlines()
        why = " | ".join(l.strip() for l in said if any(k in l for k in ("rank", "--gpus", "NCCL", "nccl", "invalid", "Duplicate")))[:500]
        if torch.cuda.device_count() < 2:
            pytest.skip(f"two ranks on one device were not accepted (exit {r.returncode}): " + (why or " ".join(said)[-400:]))
        pytest.fail(r.stdout + r.stderr)
    lut = (QUANTIZE_BASE * np.float32(8)).astype(np.float32)
    rc, want = O.run_behaviour("q32_avx", synth.plane_u8_np(W, H, "photo"), lut, W, H, 0, H)
    assert np.array_equal(np.fromfile(dump, dtype=np.uint8), want)
    assert "over 2 GPU(s)" in r.stdout
