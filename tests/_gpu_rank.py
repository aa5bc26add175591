"""One rank of tests/test_two_rank_gpu.py (a fresh child process; TEST INFRASTRUCTURE).

Every rank: mdct_init(device), transforms ONLY its shard_rows range of each plane WITH THE HIP KERNELS
(canary-filled full-size device buffers), the shards meet through gloo (one process per rank, as bench.py
and the driver run the engine), and every rank compares every gathered byte with the oracle's single-range
output of the whole plane.  The ranks share GPU 0 when the box has fewer GPUs than ranks.  Prints one JSON line.

The sharding is the reference's own parallel hook: block row y is processed iff startY <= 2y <= endY
(simd_dct.cpp:2245-2255) -- one case goes through exactly that call form of the drop-in API."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    import oracle as O
    import simd_dct_amd as M
    from simd_dct_amd import synth

    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    M.init(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    report = {}
    try:
        lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
        lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)

        def gather_slabs(buf_dev, b0, b1, row_bytes, n_rows):
            """equal shards: in-place all_gather_into_tensor of byte slabs (what mdct_allgather_rows does over RCCL); ragged: padded all_gather"""
            flat = buf_dev.reshape(-1).view(torch.uint8).cpu()
            if n_rows % world == 0:
                mine = flat[b0 * row_bytes:b1 * row_bytes].clone()
                dist.all_gather_into_tensor(flat, mine)
                return flat.numpy()
            per = -(-n_rows // world)
            pad = torch.zeros(per * row_bytes, dtype=torch.uint8)
            pad[:(b1 - b0) * row_bytes] = flat[b0 * row_bytes:b1 * row_bytes]
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            out = flat.clone()
            for r in range(world):
                r0, r1 = M.shard_rows(n_rows, world, r)
                out[r0 * row_bytes:r1 * row_bytes] = parts[r][:(r1 - r0) * row_bytes]
            return out.numpy()

        for (W, H) in ((1024, 512), (576, 120)):  # 64 block rows (equal shards for 2, 4, 8) and 15 (ragged)
            rows = H // 8
            b0, b1 = M.shard_rows(rows, world, rank)
            assert (b0, b1) == M.shard_rows_c(rows, world, rank)
            tag = f"{W}x{H}"
            # ---- q32, native C-ABI (mdct_fwd_quant_u8): a row shard's output is one contiguous slab at b0 * 8 * W
            img = synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + 3)
            img_np = synth.plane_u8_np(W, H, "photo", seed=synth.SEED + 3)
            out = torch.full((W * H,), 0xEE, dtype=torch.uint8, device="cuda")
            M.fwd_quant_u8(img, out, lut, W, H, b0, b1)
            torch.cuda.synchronize()
            mine = out.cpu().numpy()
            untouched = bool((mine[:b0 * 8 * W] == 0xEE).all() and (mine[b1 * 8 * W:] == 0xEE).all())
            got = gather_slabs(out, b0, b1, 8 * W, rows)
            want = O.q32_native(img_np, lut, W, H, 0, rows)[1]
            report[f"q32_native_{tag}"] = bool(np.array_equal(got, want)) and untouched
            # ---- q32 through the drop-in API with the reference's startY/endY (2y units, inclusive), sizeY = 2H call form
            if b1 > b0:
                out2 = torch.full((W * H,), 0xEE, dtype=torch.uint8, device="cuda")
                rc = M.simdDCT_EncodeQuantize32ReorderBuffer(img, out2, lut, W, 2 * H, 16 * b0, 16 * b1 - 16)
                report[f"q32_shim_rc_{tag}"] = rc == 0
                got2 = gather_slabs(out2, b0, b1, 8 * W, rows)
                report[f"q32_shim_{tag}"] = bool(np.array_equal(got2, want))
            # ---- int16 forward (configs[3]'s work) and the fused round trip (the bench workload), in place in the full plane
            src = synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + 4)
            src_np = synth.plane_i16_np(W, H, "photo", seed=synth.SEED + 4)
            for mode, fn in (("fwd", M.fwd_i16), ("roundtrip", M.roundtrip_i16)):
                o16 = torch.full((H, W), -4370, dtype=torch.int16, device="cuda")
                if b1 > b0:
                    fn(src, o16, W, H, by0=b0, by1=b1)
                torch.cuda.synchronize()
                got = gather_slabs(o16, b0, b1, 8 * W * 2, rows).view(np.int16).reshape(H, W)
                report[f"i16_{mode}_{tag}"] = bool(np.array_equal(got, O.i16(mode, src_np, W, H)))
        # ---- stereo layout (two stacked half-height images, 64 coefficient planes): each rank's shard is 64 strided pieces; every output
        # byte is written by exactly one rank, so zero-filled buffers summed over the ranks are the gather
        W, H = 512, 256
        sb0, sb1 = M.shard_rows(H // 16, world, rank)
        img = synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + 5)
        img_np = synth.plane_u8_np(W, H, "photo", seed=synth.SEED + 5)
        so = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
        if sb1 > sb0:
            M.fwd_quant_u8(img, so, lut8, W, H, sb0, sb1, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SSE)
        torch.cuda.synchronize()
        acc = so.cpu().to(torch.int32)
        dist.all_reduce(acc)
        want = O.run_behaviour("stereo_sse", img_np, lut8, W, H, 0, H)[1]
        report["stereo_sse_512x256"] = bool(np.array_equal(acc.numpy().astype(np.uint8), want)) and int(acc.max()) <= 255
        first, stride, piece = M.stereo_shard_piece(W, H, world, rank)
        mine = so.cpu().numpy()
        mask = np.zeros(W * H, dtype=bool)
        for c in range(64):
            mask[first + c * stride:first + c * stride + piece] = True
        report["stereo_pieces_are_the_c_abi_s"] = bool((mine[~mask] == 0).all())
        report["native_so"] = os.path.basename(M.api._lib.load()._name)
    finally:
        dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "device": dev, "report": report}))


if __name__ == "__main__":
    main()
