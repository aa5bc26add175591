"""The multi-GPU leg of the C-ABI (include/mdct.h, csrc/comm.hip).  CPU: the shard arithmetic and the
stereo layout's 64-piece table against the oracle run as fake ranks; argument checks.  GPU: a one-rank
RCCL communicator through the C-ABI (row strips and the 64 grouped stereo collectives)."""
import numpy as np
import pytest

import oracle as O
from simd_dct_amd import api, synth
from simd_dct_amd.sharding import shard_rows


def test_c_shard_rows_equals_python():
    for n in (0, 1, 7, 8, 64, 1023, 1024, 131072):
        for w in (1, 2, 3, 5, 8):
            for r in range(w):
                assert api.shard_rows_c(n, w, r) == shard_rows(n, w, r), (n, w, r)
    assert api.shard_rows_c(8, 2, 2) == (0, 0)  # out of range: empty


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_stereo_shard_pieces_reassemble_the_plane(world):
    """a block-row shard of the stereo layout (simd_dct.cpp:1061-1099) is 64 strided pieces.
    Fake ranks: every rank runs the oracle's stereo tier on its row range into its own buffer; copying
    exactly the pieces mdct_stereo_shard_piece names out of each rank's buffer must give the full result."""
    W, H = 128, 96  # 6 stereo block rows: ragged for world = 8 (some ranks empty) and world = 4
    img = synth.plane_u8_np(W, H, "photo")
    lut = (api.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
    rc, full = O.run_behaviour("stereo_sse", img, lut, W, H, 0, H)
    got = np.full(W * H, 0xEE, dtype=np.uint8)
    covered = np.zeros(W * H, dtype=bool)
    for r in range(world):
        b0, b1 = shard_rows(H // 16, world, r)
        off, stride, nbytes = api.stereo_shard_piece(W, H, world, r)
        assert stride == W * H // 64 and off == b0 * 2 * (W // 8) and nbytes == (b1 - b0) * 2 * (W // 8)
        mine = np.zeros(W * H, dtype=np.uint8)
        if b1 > b0:  # the reference's inclusive range in its 2y units: rows b0 .. b1-1
            O.run_behaviour("stereo_sse", img, lut, W, H, 16 * b0, 16 * (b1 - 1), out=mine)
        for k in range(64):
            got[k * stride + off:k * stride + off + nbytes] = mine[k * stride + off:k * stride + off + nbytes]
            assert not covered[k * stride + off:k * stride + off + nbytes].any()
            covered[k * stride + off:k * stride + off + nbytes] = True
    assert covered.all() and np.array_equal(got, full)


def test_comm_argument_checks_without_device():
    from simd_dct_amd import _lib

    lib = _lib.load()
    a = np.zeros(64, dtype=np.uint8)
    assert lib.mdct_allgather_rows(None, a.ctypes.data, 8, 8, None) == 1
    assert lib.mdct_allgather_stereo(None, a.ctypes.data, 64, 16, None) == 1
    assert lib.mdct_comm_destroy(None) == 0
    assert lib.mdct_comm_rank(None) == -1 and lib.mdct_comm_world(None) == 0
    with pytest.raises(api.MdctError):
        api.stereo_shard_piece(60, 16, 2, 0)
    with pytest.raises(api.MdctError):
        api.stereo_shard_piece(64, 16, 2, 2)


@pytest.mark.gpu
def test_one_rank_rccl_through_the_cabi():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    api.init(0)
    comm = api.Comm(0, 1, api.comm_unique_id())
    try:
        W, H = 512, 256
        lut = (api.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
        img = synth.plane_u8_np(W, H, "photo")
        src = torch.from_numpy(img).cuda()
        # Q32 strips: transform the rank's shard in place, gather, compare with the oracle
        out = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
        b0, b1 = api.shard_rows_c(H // 8, comm.world, comm.rank)
        api.fwd_quant_u8(src, out, lut, W, H, b0, b1)
        comm.allgather_rows(out, 8 * W, H // 8)
        torch.cuda.synchronize()
        rc, want = O.q32_native(img, lut, W, H, 0, H // 8)
        assert np.array_equal(out.cpu().numpy(), want)
        # int16 plane rows (config 4's layout): row_bytes = 8 * pitch * sizeof(int16)
        s16 = synth.plane_i16_np(W, H, "photo")
        d16 = torch.zeros((H, W), dtype=torch.int16, device="cuda")
        api.fwd_i16(torch.from_numpy(s16).cuda(), d16, W, H, by0=b0, by1=b1)
        comm.allgather_rows(d16, 8 * W * 2, H // 8)
        torch.cuda.synchronize()
        assert np.array_equal(d16.cpu().numpy(), O.i16("fwd", s16, W, H))
        # stereo layout: 64 collectives in one RCCL group
        lut8 = (api.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
        st = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
        s0, s1 = api.shard_rows_c(H // 16, comm.world, comm.rank)
        api.fwd_quant_u8(src, st, lut8, W, H, s0, s1, layout=api.LAYOUT_STEREO, profile=api.PROFILE_REF_SSE)
        comm.allgather_stereo(st, W, H)
        torch.cuda.synchronize()
        rc, want = O.run_behaviour("stereo_sse", img, lut8, W, H, 0, H)
        assert np.array_equal(st.cpu().numpy(), want)
    finally:
        comm.close()


@pytest.mark.gpu
def test_one_rank_rccl_ragged_form_and_stereo_pieces(monkeypatch):
    """The form ragged shards take -- every rank broadcasts its piece in place inside one RCCL group -- and the stereo layout's 64 grouped
    pieces, through the REAL RCCL (one rank is all this box has; MDCT_FORCE_RAGGED_GATHER=1 selects the form, which world = 1 alone
    never would): the symbols resolve, the calls are accepted with the arguments csrc/comm.hip passes, the buffer comes back intact.
    (With several ranks the same code runs against tests/fake_rccl.c: tests/test_comm_multirank.py, world 2 / 3 / 8 incl. empty ranks.)"""
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    api.init(0)
    comm = api.Comm(0, 1, api.comm_unique_id())
    try:
        monkeypatch.setenv("MDCT_FORCE_RAGGED_GATHER", "1")
        W, H = 1024, 200  # 25 block rows: no multiple of anything
        s16 = synth.plane_i16_np(W, H, "photo", seed=3)
        d16 = torch.zeros((H, W), dtype=torch.int16, device="cuda")
        api.fwd_i16(torch.from_numpy(s16).cuda(), d16, W, H)
        want = d16.clone()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        for _ in range(3):
            comm.allgather_rows(d16, 8 * W * 2, H // 8, stream=stream)  # grouped ncclBroadcast, root 0, in place
        stream.synchronize()
        assert torch.equal(d16, want) and np.array_equal(d16.cpu().numpy(), O.i16("fwd", s16, W, H))
        Ws, Hs = 512, 208  # 13 stereo block rows
        img = synth.plane_u8_np(Ws, Hs, "photo", seed=4)
        lut8 = (api.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
        st = torch.zeros(Ws * Hs, dtype=torch.uint8, device="cuda")
        api.fwd_quant_u8(torch.from_numpy(img).cuda(), st, lut8, Ws, Hs, 0, Hs // 16, layout=api.LAYOUT_STEREO, profile=api.PROFILE_REF_SSE)
        for forced in ("1", "0"):  # 64 grouped broadcasts, then 64 grouped all-gathers
            monkeypatch.setenv("MDCT_FORCE_RAGGED_GATHER", forced)
            comm.allgather_stereo(st, Ws, Hs, stream=stream)
            stream.synchronize()
            rc, want8 = O.run_behaviour("stereo_sse", img, lut8, Ws, Hs, 0, Hs)
            assert np.array_equal(st.cpu().numpy(), want8), forced
    finally:
        comm.close()


@pytest.mark.gpu
def test_cxx_cli_one_process_per_gpu(tmp_path):
    """tools/simd_dct_cli --gpus N: C++ host code, one forked process per GPU, the reference's own
    startY/endY shard hook, RCCL all-gather through the C-ABI; rank 0's gathered output is the oracle's"""
    import subprocess

    import __graft_entry__ as G

    cli = G.build_cli()
    W, H = 512, 256
    img = synth.plane_u8_np(W, H, "photo")
    for mode, scale in (("enc-quant32", 2000), ("enc-quant-stereo", 8)):
        dump = tmp_path / (mode + ".bin")
        r = subprocess.run([cli, "synthetic:photo", str(W), str(H), "--mode", mode, "--quality", str(scale), "--runs", "3", "--gpus", "1", "--to", str(dump)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "sdr_Success" in r.stdout and "RCCL" in r.stdout
        got = np.fromfile(dump, dtype=np.uint8)
        lut = (api.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)
        if mode == "enc-quant32":
            rc, want = O.q32_native(img, lut, W, H, 0, H // 8)
        else:
            rc, want = O.run_behaviour("stereo_sse", img, lut, W, H, 0, H)
        assert np.array_equal(got, want), mode
