"""N>1 path on CPU: world_size-2 gloo.  Each rank transforms its block-row shard (with the
oracle standing in for the kernel -- this tests the partition / gather logic, which is what
differs from N=1) and the all-gathered result must equal the single-range result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_partition():
    from simd_dct_amd.sharding import equal_shards, shard_rows

    for n in (0, 1, 7, 8, 64, 1023, 1024):
        for w in (1, 2, 3, 8):
            cuts = [shard_rows(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for (a0, a1), (b0, b1) in zip(cuts[:-1], cuts[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
            assert equal_shards(n, w) == (len(set(sizes)) == 1)
    with pytest.raises(ValueError):
        shard_rows(8, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle as O
    from simd_dct_amd import synth
    from simd_dct_amd.api import QUANTIZE_BASE
    from simd_dct_amd.sharding import shard_rows

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H = 128, 64
        rows = H // 8
        b0, b1 = shard_rows(rows, world, rank)
        lut = (QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
        # u8 q32: a row shard's output is one contiguous slab at b0*8*W
        img = synth.plane_u8_np(W, H, "photo")
        mine = np.zeros(W * H, dtype=np.uint8)
        O.q32_native(img, lut, W, H, b0, b1, out=mine)
        shard = torch.from_numpy(mine[b0 * 8 * W:b1 * 8 * W].copy())
        parts = [torch.empty_like(shard) for _ in range(world)]
        dist.all_gather(parts, shard)
        gathered = torch.cat(parts).numpy()
        rc, full = O.q32_native(img, lut, W, H, 0, rows)
        ok_u8 = bool(np.array_equal(gathered, full))
        # int16 forward (config 4's shape of work): in-place all_gather into one tensor
        src = synth.plane_i16_np(W, H, "photo")
        out = torch.zeros((H, W), dtype=torch.int16)
        out[b0 * 8:b1 * 8] = torch.from_numpy(O.i16("fwd", src, W, H, by0=b0, by1=b1)[b0 * 8:b1 * 8])
        # byte views: the collective moves bytes, and gloo has no int16
        dist.all_gather_into_tensor(out.view(torch.uint8), out[b0 * 8:b1 * 8].clone().view(torch.uint8))
        ok_i16 = bool(np.array_equal(out.numpy(), O.i16("fwd", src, W, H)))
        q.put((rank, ok_u8, ok_i16))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]
