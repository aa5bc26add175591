"""bench_extras.py -- everything bench.py measures BESIDE the headline: the other kernels and BASELINE.json
configurations, the verbose per-kernel roofline blocks, and the optional whole-node all-gather legs.

bench.py prints ONE short JSON line (the driver parses it); what is built here goes, in full, to
bench_extras.json next to the script (and its compact number-only form, `compact_kernels`, into the line).
Nothing in this file contributes to `value`.
"""
import json
import os
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def measure(M, torch, np, synth, srcs, dsts, timer, rank, W, H, NSETS, achieved):
    """every extra measurement, each pre-conditioned and verified in the run; returns the raw dict"""
    extras = {}
    def rate(fn, bytes_per_launch, n=200, warm=None):
        for i in range(warm if warm else (300 if n >= 100 else 80)):  # >= 15 ms of the kernel itself: past the power transient
            fn(i)
        torch.cuda.synchronize()
        timer.start()
        for i in range(n):
            fn(i)
        timer.stop()
        ms = timer.elapsed_ms() / n
        return {"ms": round(ms, 4), "GBps": round(bytes_per_launch / (ms * 1e-3) / 1e9, 1)}

    side = torch.cuda.Stream()
    probe_buf = torch.zeros(16, dtype=torch.int64, device="cuda")

    def clock_under(fn, ms_per_launch, n=400):
        """shader clock (GHz) the chip holds while `fn` runs back to back: mdct_clock_probe on a second stream beside n launches"""
        try:
            for i in range(60):
                fn(i)
            M.clock_probe(probe_buf, max(1000, int(ms_per_launch * 1e5 * (n - 120) * 0.5)), waves=8, stream=side)
            for i in range(60, n):
                fn(i)
            torch.cuda.synchronize()
            pr = probe_buf.cpu().numpy().reshape(8, 2)
            return round(float((pr[:, 0] / (pr[:, 1] * 10.0)).mean()), 3)
        except Exception:
            return None

    nbytes = W * H * 2
    def prepared(make):
        calls = [make(i) for i in range(NSETS)]
        return lambda i: calls[i % NSETS]()

    extras["stream_copy_roofline"] = rate(prepared(lambda i: M.prepare_stream_copy(srcs[i], dsts[i], nbytes)), 2 * nbytes)
    extras["fwd_i16"] = rate(prepared(lambda i: M.prepare_plane_i16("fwd", srcs[i], dsts[i], W, H)), 2 * nbytes)
    extras["inv_i16"] = rate(prepared(lambda i: M.prepare_plane_i16("inv", srcs[i], dsts[i], W, H)), 2 * nbytes)
    # ---- the reference's three products (simd_dct.h:29-31) on the same plane size, every tier the engine reproduces.
    # Plane set 0 is the default synthetic "photo" plane, whose outputs from the REAL reference are committed as SHA-256
    # (tests/golden/ref_vectors.json: config0_sha256, written by tests/golden/make_golden.py where /root/reference
    # exists); the whole 64 MiB output of the timed call is hashed -- no oracle call, no sampled stripe.
    import hashlib

    try:
        ref_sha = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")))["config0_sha256"]
    except Exception:
        ref_sha = {}
    u8s = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + (50 + i if i else 0)).reshape(-1) for i in range(NSETS)]
    u8d = [torch.zeros(W * H, dtype=torch.uint8, device="cuda") for _ in range(NSETS)]  # zeroed: the SSE encq tier leaves half of every block pair untouched (simd_dct.cpp:1662-1676)
    # (kernel name, table scale, layout, profile, block rows of the call, golden key, reference lines)
    products = {
        "fwd_quant_u8_q32": ("mdct::k_q32_tile", 2000, M.LAYOUT_Q32, M.PROFILE_REF_AVX, H // 8, "q32_avx__photo__8192x8192__x2000__full",
                             "simdDCT_EncodeQuantize32ReorderBuffer, AVX2 = AVX-512VL tier, simd_dct.cpp:2064-2262"),
        "fwd_quant_u8_stereo_sse": ("mdct::k_fwd_quant_u8<REF_SSE, STEREO, false, TILED>", 8, M.LAYOUT_STEREO, M.PROFILE_REF_SSE, H // 16, "stereo_sse__photo__8192x8192__x8",
                                    "simdDCT_EncodeQuantizeReorderStereoBuffer, SSE4.1 = SSSE3 = SSE2 tiers, simd_dct.cpp:896-1103"),
        "fwd_quant_u8_stereo_scalar": ("mdct::k_fwd_quant_u8<REF_SCALAR, STEREO, false, TILED>", 8, M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, H // 16, "stereo_scalar__photo__8192x8192__x8",
                                       "simdDCT_EncodeQuantizeReorderStereoBuffer, scalar tier, simd_dct.cpp:177-298"),
        "fwd_quant_u8_encq_sse": ("mdct::k_fwd_quant_u8<REF_SSE, BLOCK_SSE, false, TILED>", 8, M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, H // 8, "encq_sse__photo__8192x8192__x8__full",
                                  "simdDCT_EncodeQuantizeBuffer, SSE4.1 = SSSE3 tiers (half-written block pairs), simd_dct.cpp:1540-1704"),
        "fwd_quant_u8_encq_scalar": ("mdct::k_fwd_quant_u8<REF_SCALAR, BLOCK, false, TILED>", 8, M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, H // 8, "encq_scalar__photo__8192x8192__x8__full",
                                     "simdDCT_EncodeQuantizeBuffer, scalar tier, simd_dct.cpp:300-395"),
    }
    # VALU-heavy kernels after HBM-bound ones: the change of load sends the chip through a ~400-launch power-management
    # transient (29 -> 47 -> 32 us, profiles/r02_b_kernel_stats_bench_with_extras.csv); like the headline metric each is
    # pre-conditioned with untimed launches and measured in steady state
    for name, (kernel, scale, layout, profile, rows, key, refline) in products.items():
        try:
            lut = (M.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)
            for d in u8d:
                d.zero_()
            call = prepared(lambda i: M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, rows, layout=layout, profile=profile))
            r = rate(call, 2 * W * H, n=500, warm=1500)
            r["clock_GHz"] = clock_under(call, r["ms"])
            r["Mpx_s"] = round(W * H / (r["ms"] * 1e-3) / 1e6, 0)
            r["kernel"], r["reference"] = kernel, refline
            torch.cuda.synchronize()
            got = hashlib.sha256(u8d[0].cpu().numpy().tobytes()).hexdigest()  # what the timed launches left in set 0's output
            r["sha256_equals_real_reference"] = (got == ref_sha[key]) if key in ref_sha else f"no committed hash {key}"
            extras[name] = r
        except Exception as e:
            extras[name] = {"error": str(e)[:160]}
    # ---- the other BASELINE.json configurations, each on its own entry point, pre-conditioned like the rest and verified in the
    # run: the device output of the timed launches is hashed and compared with the SHA-256 the CPU checker's output has for the same
    # synthetic planes (tests/golden/engine_own_sha256.json, written by tests/golden/make_engine_hashes.py; no oracle call here)
    try:
        own_sha = json.load(open(os.path.join(ROOT, "tests", "golden", "engine_own_sha256.json")))
    except Exception:
        own_sha = {}

    def sha_of(t):
        return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()

    # configs[2]: Y 7680x4320 + Cb/Cr 3840x2160, per-plane Annex-K tables, fused fwd -> quantise -> dequantise -> inv, ONE call
    try:
        frames = []
        for f in range(NSETS):  # frame 0 = the planes the committed hashes belong to; the others only defeat the Infinity Cache
            pl = []
            for (w, h, so, tab) in synth.CONFIG3_PLANES:
                a = synth.plane_i16_torch(w, h, "photo", seed=synth.SEED + so + 10 * f)
                pl.append((a, torch.zeros_like(a), w, h, synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA))
            frames.append(pl)
        fpx = sum(w * h for (w, h, _, _) in synth.CONFIG3_PLANES)
        calls3 = [M.prepare_roundtrip_i16_planes(f) for f in frames]
        r = rate(lambda i: calls3[i % NSETS](), 4 * fpx, n=500, warm=1500)
        r["clock_GHz"] = clock_under(lambda i: calls3[i % NSETS](), r["ms"])
        torch.cuda.synchronize()
        want = own_sha.get("config3_420", {})
        keys = [f"roundtrip__{w}x{h}__seed+{so}__{tab}" for (w, h, so, tab) in synth.CONFIG3_PLANES]
        r["sha256_equals_cpu_checker"] = all(k in want and sha_of(frames[0][j][1]) == want[k] for j, k in enumerate(keys)) if want else "no committed hashes"
        r["Mpx_s"] = round(fpx / (r["ms"] * 1e-3) / 1e6, 0)
        dev3 = [M.Batch("roundtrip", f) for f in frames]
        pr3 = [b.prepared() for b in dev3]
        r["device_table_form"] = rate(lambda i: pr3[i % NSETS](), 4 * fpx, n=500, warm=500)
        fw3 = [M.prepare_i16_batch("fwd", f) for f in frames]
        r["forward_only_batch"] = rate(lambda i: fw3[i % NSETS](), 4 * fpx, n=500, warm=500)
        # the per-launch fill and drain (~6 us of the 36) is shared when a call carries several frames: the same entry point with
        # all four frames' twelve planes in ONE call, and single-frame calls alternating over two streams (extras, never the headline)
        four = M.prepare_roundtrip_i16_planes([pl for f in frames for pl in f])
        r["four_frames_per_call_ms_per_frame"] = round(rate(lambda i: four(), 16 * fpx, n=200, warm=300)["ms"] / 4, 4)
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        two = [M.prepare_i16_batch("roundtrip", frames[i], stream=s2[i % 2].cuda_stream) for i in range(NSETS)]
        for st in s2:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(400):
            two[i % NSETS]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1000):
            two[i % NSETS]()
        torch.cuda.synchronize()
        r["two_streams_ms_per_frame"] = round((time.perf_counter() - t0), 4)  # 1000 frames: seconds == ms per frame
        extras["config3_420_roundtrip_one_call"] = r
        del frames, calls3, dev3, pr3, fw3, four, two
    except Exception as e:
        extras["config3_420_roundtrip_one_call"] = {"error": str(e)[:200]}
    # configs[2] as SURVEY.md 8(d) states it: the same frame as 8-bit planes, u8 in -> u8 out (2 algorithmic bytes per pixel = 99,532,800 B),
    # forward -> quantise -> dequantise -> inverse in ONE launch of k_u8_batch
    try:
        NF8 = 6  # 99.5 MB per frame: six rotate well past the 256 MB Infinity Cache
        frames8 = []
        for f in range(NF8):
            pl = []
            for (w, h, so, tab) in synth.CONFIG3_PLANES:
                a = synth.plane_u8_torch(w, h, "photo", seed=synth.SEED + so + 10 * f)
                pl.append((a, torch.zeros_like(a), w, h, synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA))
            frames8.append(pl)
        fpx = sum(w * h for (w, h, _, _) in synth.CONFIG3_PLANES)
        dev8 = [M.Batch("roundtrip_u8", f) for f in frames8]
        pr8 = [b.prepared() for b in dev8]
        r = rate(lambda i: pr8[i % NF8](), 2 * fpx, n=600, warm=1500)
        r["clock_GHz"] = clock_under(lambda i: pr8[i % NF8](), r["ms"])
        torch.cuda.synchronize()
        want = own_sha.get("config3_420_u8", {})
        keys = [f"roundtrip_u8__{w}x{h}__seed+{so}__{tab}" for (w, h, so, tab) in synth.CONFIG3_PLANES]
        r["sha256_equals_cpu_checker"] = all(k in want and sha_of(frames8[0][j][1]) == want[k] for j, k in enumerate(keys)) if want else "no committed hashes"
        r["launches_per_call"] = dev8[0].launches
        r["Mpx_s"] = round(fpx / (r["ms"] * 1e-3) / 1e6, 0)
        kept8 = [t[1].clone() for t in frames8[0]]
        for t in frames8[0]:
            t[1].zero_()
        ar8 = [M.prepare_u8_batch(f) for f in frames8]
        r["kernel_argument_form"] = rate(lambda i: ar8[i % NF8](), 2 * fpx, n=600, warm=600)
        torch.cuda.synchronize()
        r["kernel_argument_form_equals_device_table_form"] = all(torch.equal(a, t[1]) for a, t in zip(kept8, frames8[0]))
        four8 = M.Batch("roundtrip_u8", [pl for f in frames8[:4] for pl in f])
        run48 = four8.prepared()
        r["four_frames_per_call_ms_per_frame"] = round(rate(lambda i: run48(), 8 * fpx, n=200, warm=300)["ms"] / 4, 4)
        # what it replaces: the two calls it fuses (3 + 3 bytes per pixel through an int16 plane), Y plane only
        yw, yh = synth.CONFIG3_PLANES[0][0], synth.CONFIG3_PLANES[0][1]
        coef8 = torch.empty((yh, yw), dtype=torch.int16, device="cuda")
        f8 = [M.prepare_u8_i16("fwd", frames8[i][0][0], coef8, yw, yh, lut=synth.JPEG_LUMA) for i in range(NF8)]
        i8 = [M.prepare_u8_i16("inv", coef8, frames8[i][0][1], yw, yh, lut=synth.JPEG_LUMA) for i in range(NF8)]
        y1 = [M.prepare_roundtrip_u8(frames8[i][0][0], frames8[i][0][1], yw, yh, lut=synth.JPEG_LUMA) for i in range(NF8)]
        r["y_plane_two_calls_ms"] = round(rate(lambda i: (f8[i % NF8](), i8[i % NF8]()), 6 * yw * yh, n=300, warm=300)["ms"], 4)
        r["y_plane_fused_ms"] = round(rate(lambda i: y1[i % NF8](), 2 * yw * yh, n=300, warm=300)["ms"], 4)
        # the two halves on the whole frame, one launch each (3 B/px: 8-bit pixels one side, int16 coefficients the other) -- what an encoder
        # (forward: the coefficients go on to the scan stages) and a decoder (inverse) call; forward then inverse == the fused launch, checked
        coefs8 = [[torch.empty((h, w), dtype=torch.int16, device="cuda") for (w, h, _, _) in synth.CONFIG3_PLANES] for _ in range(NF8)]
        back8 = [torch.zeros_like(t[1]) for t in frames8[0]]
        fwd8 = [M.Batch("fwd_u8_i16", [(t[0], c, t[2], t[3], t[4]) for t, c in zip(frames8[i], coefs8[i])]).prepared() for i in range(NF8)]
        inv8 = [M.Batch("inv_i16_u8", [((back8[j] if i == 0 else t[1]), c, t[2], t[3], t[4]) for j, (t, c) in enumerate(zip(frames8[i], coefs8[i]))]).prepared() for i in range(NF8)]
        r["forward_only_batch"] = rate(lambda i: fwd8[i % NF8](), 3 * fpx, n=300, warm=300)
        r["inverse_only_batch"] = rate(lambda i: inv8[i % NF8](), 3 * fpx, n=300, warm=300)
        torch.cuda.synchronize()
        r["forward_then_inverse_equals_fused"] = all(torch.equal(a, b) for a, b in zip(back8, kept8))
        del coefs8, back8, fwd8, inv8
        # the same frame as the REFERENCE's product (q32 layout, AVX2-tier bytes: parity pinned by the reference), one launch against the three
        # calls the reference's caller makes (main.cpp:543); 2 B/px
        qlut = [(M.QUANTIZE_BASE * np.float32(2000 if tab == "luma" else 1200)).astype(np.float32) for (_, _, _, tab) in synth.CONFIG3_PLANES]
        qout = [[torch.empty(t[2] * t[3], dtype=torch.uint8, device="cuda") for t in frames8[i]] for i in range(NF8)]
        q1 = [M.Batch("q32", [(t[0], o, t[2], t[3], l) for t, o, l in zip(frames8[i], qout[i], qlut)]).prepared() for i in range(NF8)]
        q3 = [[M.prepare_fwd_quant_u8(t[0], o, l, t[2], t[3], 0, t[3] // 8) for t, o, l in zip(frames8[i], qout[i], qlut)] for i in range(NF8)]
        for c in q3[0]:
            c()
        torch.cuda.synchronize()
        three = [o.clone() for o in qout[0]]
        for o in qout[0]:
            o.zero_()
        r["reference_q32_product_one_launch"] = rate(lambda i: q1[i % NF8](), 2 * fpx, n=300, warm=300)
        r["reference_q32_product_one_launch"]["clock_GHz"] = clock_under(lambda i: q1[i % NF8](), r["reference_q32_product_one_launch"]["ms"])
        torch.cuda.synchronize()
        r["reference_q32_product_one_launch"]["equals_three_single_plane_calls"] = all(torch.equal(a, b) for a, b in zip(three, qout[0]))
        r["reference_q32_product_three_calls_ms"] = round(rate(lambda i: [c() for c in q3[i % NF8]], 2 * fpx, n=300, warm=300)["ms"], 4)
        del qout, q1, q3, three
        # the same kernel on one 8192x8192 8-bit plane (the bench's plane size; 134,217,728 B)
        one8 = [M.prepare_roundtrip_u8(u8s[i].view(H, W), u8d[i].view(H, W), W, H, lut=synth.JPEG_LUMA) for i in range(NSETS)]
        r["plane_8192_ms"] = round(rate(lambda i: one8[i % NSETS](), 2 * W * H, n=300, warm=300)["ms"], 4)
        r["plane_8192_Mpx_s"] = round(W * H / (r["plane_8192_ms"] * 1e-3) / 1e6, 0)
        extras["config3_420_u8_roundtrip_one_call"] = r
        del frames8, dev8, pr8, ar8, four8, run48, coef8, f8, i8, y1, kept8, one8
    except Exception as e:
        extras["config3_420_u8_roundtrip_one_call"] = {"error": str(e)[:200]}
    # configs[4]: float32 DCT-II on the 8192x8192 plane (8 algorithmic bytes per pixel)
    try:
        fsrc = [srcs[i].to(torch.float32) for i in range(2)]  # plane set 0 = float(int16 photo plane, seed SEED): the committed hash
        fdst = [torch.zeros_like(t) for t in fsrc]
        r = rate(lambda i: M.fwd_f32(fsrc[i % 2], fdst[i % 2], W, H), 8 * W * H, n=300, warm=600)
        torch.cuda.synchronize()
        want = own_sha.get("config5_f32", {}).get("fwd__8192x8192__seed+0")
        r["sha256_equals_cpu_checker"] = (sha_of(fdst[0]) == want) if want and rank == 0 else ("rank-0 planes only" if want else "no committed hash")
        r["Mpx_s"] = round(W * H / (r["ms"] * 1e-3) / 1e6, 0)
        r["cpu_checker_max_err_over_block_max_vs_double"] = own_sha.get("config5_f32", {}).get("max_err_over_block_max_vs_double")
        r["stream_copy_same_bytes"] = rate(lambda i: M.stream_copy(fsrc[i % 2], fdst[i % 2], W * H * 4), 8 * W * H, n=300, warm=300)
        extras["config5_f32_fwd"] = r
        del fsrc, fdst
    except Exception as e:
        extras["config5_f32_fwd"] = {"error": str(e)[:200]}
    # configs[3] on ONE GPU: 256 independent (separately allocated) 4096x4096 int16 planes, forward only -- one call of the
    # plane-batch entry point (device-table form: one launch; kernel-argument form: ~6), the same planes stacked as one
    # tall plane (one launch, needs contiguous memory), and one launch per plane
    try:
        PW = PH = 4096
        NPL = 256
        pin = [synth.plane_i16_torch(PW, PH, "photo", seed=synth.SEED + 100 + p) for p in range(NPL)]
        pout = [torch.zeros_like(t) for t in pin]
        desc = [(a, b, PW, PH, None) for a, b in zip(pin, pout)]
        bpx = NPL * PW * PH
        b4 = M.Batch("fwd", desc)
        run4 = b4.prepared()
        c4 = {"one_call_device_table": rate(lambda i: run4(), 4 * bpx, n=10, warm=5)}
        c4["one_call_device_table"]["launches"] = b4.launches
        torch.cuda.synchronize()
        want = own_sha.get("config4_planes", {})
        if rank == 0:
            c4["sha256_equals_cpu_checker"] = all(f"fwd__4096x4096__seed+{100 + p}" in want and sha_of(pout[p]) == want[f"fwd__4096x4096__seed+{100 + p}"] for p in (0, 1, 255)) if want else "no committed hashes"
        kept = [t.clone() for t in pout[:8]]
        for t in pout:
            t.zero_()
        args4 = M.prepare_i16_batch("fwd", desc)
        c4["one_call_kernel_arguments"] = rate(lambda i: args4(), 4 * bpx, n=10, warm=3)
        torch.cuda.synchronize()
        c4["kernel_argument_form_equals_device_table_form"] = all(torch.equal(a, b) for a, b in zip(kept, pout[:8]))
        per4 = [M.prepare_plane_i16("fwd", a, b, PW, PH) for a, b in zip(pin, pout)]

        def all_planes(i):
            for c in per4:
                c()

        c4["one_launch_per_plane"] = rate(all_planes, 4 * bpx, n=5, warm=2)
        tall_in = torch.cat(pin, dim=0)
        del pin, per4, args4, run4, b4, desc
        tall_out = torch.zeros_like(tall_in)
        st4 = M.prepare_plane_i16("fwd", tall_in, tall_out, PW, NPL * PH)
        c4["stacked_one_launch"] = rate(lambda i: st4(), 4 * bpx, n=10, warm=3)
        torch.cuda.synchronize()
        c4["every_plane_equals_the_stacked_launch"] = all(torch.equal(pout[p], tall_out[p * PH:(p + 1) * PH]) for p in range(NPL))
        c4["Mpx_s_one_call"] = round(bpx / (c4["one_call_device_table"]["ms"] * 1e-3) / 1e6, 0)
        extras["config4_256_planes_one_gpu"] = c4
        del pout, tall_in, tall_out, st4, kept
        torch.cuda.empty_cache()
    except Exception as e:
        extras["config4_256_planes_one_gpu"] = {"error": str(e)[:200]}
    extras["roundtrip_frac_of_measured_copy"] = round(achieved / extras["stream_copy_roofline"]["GBps"], 3)
    # independent planes on two HIP streams: plane k+1's head overlaps plane k's drain
    # (an extra, never `value`: per-kernel durations and throughput differ once launches overlap)
    try:
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        two = [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H, stream=s2[i % 2].cuda_stream) for i in range(NSETS)]
        for st in s2:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(400):
            two[i % NSETS]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n2 = 1000
        for i in range(n2):
            two[i % NSETS]()
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / n2 * 1e3
        extras["roundtrip_two_streams"] = {"ms_per_plane": round(ms2, 4), "GBps": round(2 * nbytes / (ms2 * 1e-3) / 1e9, 1)}
    except Exception as e:
        extras["roundtrip_two_streams"] = {"error": str(e)[:120]}
    # the same kernel without the per-launch drain: 8 planes stacked in memory are one tall
    # plane (blocks are independent), one launch
    try:
        nb = 8
        tall_in = torch.cat(srcs + srcs, dim=0)
        tall_out = torch.empty_like(tall_in)
        call = M.prepare_plane_i16("roundtrip", tall_in, tall_out, W, nb * H)
        extras["roundtrip_8_planes_one_launch"] = rate(lambda i: call(), nb * 2 * nbytes, n=40)
        extras["roundtrip_8_planes_one_launch"]["ms_per_plane"] = round(extras["roundtrip_8_planes_one_launch"]["ms"] / nb, 4)
        del tall_in, tall_out
    except Exception as e:
        extras["roundtrip_8_planes_one_launch"] = {"error": str(e)[:120]}

    # the stages after the transform: quantised coefficients -> zig-zag + run/level records -> baseline Huffman rows
    try:
        q60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
        M.fwd_i16(srcs[0], dsts[0], W, H, lut=q60)
        nblk = (W // 8) * (H // 8)
        lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
        rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
        ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
        hstride = M.huffman_seg_stride(W)
        hseg = torch.empty(((H // 8) * hstride,), dtype=torch.uint8, device="cuda")
        hnb = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
        extras["zigzag_rle_i16"] = rate(lambda i: M.zigzag_rle_i16(dsts[0], W, H, lv, rn, ct), 5 * W * H + nblk, n=100, warm=200)
        extras["fwd_u8_records_fused"] = rate(lambda i: M.fwd_u8_records(u8s[i % NSETS], W, H, lv, rn, ct, lut=q60), 4 * W * H + nblk, n=100, warm=200)
        M.zigzag_rle_i16(dsts[0], W, H, lv, rn, ct)  # back to the dense records the Huffman figure is quoted on
        extras["huffman_rows"] = rate(lambda i: M.huffman_rows(lv, rn, ct, W, H, hseg, hnb), 3 * W * H + nblk, n=100, warm=200)
        extras["huffman_rows"]["pairs_per_block"] = round(float(ct.float().mean()), 1)
        extras["huffman_rows"]["bits_per_px"] = round(float(hnb.sum()) * 8 / (W * H), 3)
        # pixels -> Huffman rows in ONE kernel (records only in LDS), and pixels -> finished scan (stuffed, RSTm) in one launch
        hff = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
        extras["px_to_huffman_rows_fused"] = rate(lambda i: M.fwd_u8_huffman_rows(u8s[i % NSETS], W, H, hseg, hnb, lut=q60, ff_counts=hff), W * H + int(hnb.sum()), n=100, warm=200)
        extras["px_to_huffman_rows_fused"]["table"] = "QUANTIZE_BASE x 60 (the records above)"
        k1 = synth.JPEG_LUMA  # ITU-T T.81 Annex K.1
        work = torch.zeros((H // 8 + 2,), dtype=torch.int64, device="cuda")
        scan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
        off = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
        M.fwd_u8_jpeg_scan(u8s[0], W, H, hseg, work, scan, off, lut=k1)
        torch.cuda.synchronize()
        nscan = int(off[-1].item())
        extras["px_to_jpeg_scan_one_launch"] = rate(lambda i: M.fwd_u8_jpeg_scan(u8s[i % NSETS], W, H, hseg, work, scan, off, lut=k1), W * H + nscan, n=100, warm=200)
        extras["px_to_jpeg_scan_one_launch"].update({"table": "ITU-T T.81 Annex K.1", "scan_bytes": nscan, "bits_per_px": round(nscan * 8 / (W * H), 3)})
        del lv, rn, ct, hseg, hnb, hff, work, scan, off
    except Exception as e:
        extras["huffman_rows"] = {"error": str(e)[:120]}
    return extras


def roofline_blocks(extras, synth, W, H, isa_file="isa_classes.json"):
    """the verbose per-kernel blocks (bench_extras.json): achieved vs the 8 TB/s spec and vs the measured copy, the PMC traffic of
    profiles/traffic.json, the vector-issue floor, and how each output was verified"""
    line = {}
    traffic = {}
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr):
        try:
            traffic = json.load(open(tr))
        except Exception:
            traffic = {}
    copy = extras.get("stream_copy_roofline", {}).get("GBps")

    try:
        isa_path = os.path.join(ROOT, "profiles", isa_file)
        if not os.path.exists(isa_path):  # before the round's own count exists: the newest committed one
            import glob

            isa_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_classes.json")))[-1]
        isa = json.load(open(isa_path))["kernels"]
        costs = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_costs.json")))
    except Exception:
        isa, costs = {}, {}
    pmc = traffic.get("kernels", {})

    def valu(isa_name, waves, avg_ms, clock_ghz, pmc_key):
        """the vector-ISSUE floor of a kernel: its static instruction mix (one wave's straight-line stream, profiles/isa_classes.json) priced
        at the issue cycles measured per class (profiles/valu_issue_costs.json) and at the clock the chip held under THIS kernel in THIS run"""
        k = isa.get(isa_name)
        if not k or not costs or not clock_ghz or not avg_ms:
            return None
        cyc = sum(k[c] * costs["cycles"][c] for c in ("plain", "packed", "other"))
        executed = pmc.get(pmc_key, {}).get("valu_insts_per_wave")
        scaled = bool(executed) and abs(executed - k["valu"]) > 0.02 * k["valu"]
        if scaled:  # a kernel with a rarely taken branch (the SSE encq tier's spill): the static mix, scaled to the count the PMC saw executed
            cyc *= executed / k["valu"]
        floor_ms = waves * cyc / (1024 * clock_ghz * 1e9) * 1e3
        return {"insts_per_wave_static": {c: k[c] for c in ("plain", "packed", "other")}, "valu_insts_per_wave_static": k["valu"],
                "valu_insts_per_wave": executed, "static_mix_scaled_to_executed_count": scaled, "waves_per_launch": waves,
                "issue_cycles_per_wave": round(cyc, 0), "issue_cycles_per_instruction": costs["cycles"], "clock_GHz_under_this_kernel": clock_ghz,
                "valu_floor_ms": round(floor_ms, 4), "frac_of_valu_floor": round(floor_ms / avg_ms, 3), "simds": 1024,
                "how": "valu_floor_ms = waves x sum(class count x measured issue cycles) / (1024 SIMDs x measured clock); counts: tools/isa_classes.py (static; "
                       "valu_insts_per_wave is the PMC's SQ_INSTS_VALU / SQ_WAVES of tools/profile_round.sh), cycles: tools/valubench2, clock: mdct_clock_probe beside the timed kernel"}

    ISA_NAME = {"k_q32_avx": "mdct::k_q32_tile(mdct::U8Args)", "k_stereo_sse": "void mdct::k_fwd_quant_u8<1, 1, false, true>(mdct::U8Args)",
                "k_stereo_scalar": "void mdct::k_fwd_quant_u8<2, 1, false, true>(mdct::U8Args)", "k_encq_sse": "void mdct::k_fwd_quant_u8<1, 3, false, true>(mdct::U8Args)",
                "k_encq_scalar": "void mdct::k_fwd_quant_u8<2, 2, false, true>(mdct::U8Args)", "k_u8_batch_420": "void mdct::k_u8_batch<0, false, false>(mdct::BatchArgs)",
                "k_i16_batch_420": "void mdct::k_i16_batch<2, 1, false, false>(mdct::BatchArgs)", "k_q32_batch_420": "void mdct::k_q32_batch<false>(mdct::BatchArgs)", "k_i16_roundtrip": "void mdct::k_i16_tile<2, false, true, 2>(mdct::I16Args)"}

    def u8_block(q, key=None):
        # the reference's own products on the same plane size: 2 algorithmic bytes per pixel
        if not q or "GBps" not in q:
            return q
        traffic_key = key + "_bytes_per_launch" if key else None
        v = valu(ISA_NAME.get(key), (W // 8) * (H // 8) // 64, q["ms"], q.get("clock_GHz"), key)
        return {"bound": "vector issue at the clock the chip holds under this kernel (valu.frac_of_valu_floor); the bytes alone would take algorithmic_bytes / measured copy rate",
                "valu": v, "traffic_round": traffic.get("traffic_round"), "kernel": q["kernel"], "reference": q["reference"],
                "parity": "pinned: byte-identical to the real reference built from /root/reference with -O2 -ffp-contract=off (SHA-256 of its output for this plane)",
                "achieved": q["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(q["GBps"] / HBM_PEAK_GBPS, 4),
                "frac_of_measured_copy": round(q["GBps"] / copy, 3) if copy else None,
                "algorithmic_bytes_per_launch": 2 * W * H, "avg_launch_ms": q["ms"], "Mpx_s": q.get("Mpx_s"),
                "traffic": traffic.get(traffic_key) if traffic_key else None,
                "traffic_source": ("profiles/traffic.json: " + traffic.get("source", "")) if traffic_key and traffic.get(traffic_key) else None,
                "bit_exact_vs_reference": q.get("sha256_equals_real_reference"),
                "verified_by": "SHA-256 of the whole output plane of the timed call == tests/golden/ref_vectors.json (bytes of the real reference)"}

    def own_block(q, kernel, alg_bytes, what, traffic_key=None, verified_key="sha256_equals_cpu_checker"):
        # (traffic_key: "<kernel key>_bytes_per_launch" of profiles/traffic.json)
        # engine-own kernels (no reference counterpart: "parity unpinned" by the reference, pinned by the CPU checker)
        if not q or "GBps" not in q:
            return q
        return {"bound": "hbm", "kernel": kernel, "what": what, "parity": "unpinned by the reference (no counterpart there); pinned by the CPU checker (oracle/dct_oracle.c)",
                "achieved": q["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(q["GBps"] / HBM_PEAK_GBPS, 4),
                "frac_of_measured_copy": round(q["GBps"] / copy, 3) if copy else None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": q["ms"],
                "traffic": traffic.get(traffic_key) if traffic_key else None,
                "traffic_round": traffic.get("traffic_round"),
                "traffic_source": ("profiles/traffic.json: " + traffic.get("source", "")) if traffic_key and traffic.get(traffic_key) else None,
                "bit_exact_vs_cpu_checker": q.get(verified_key),
                "verified_by": "SHA-256 of the device output of the timed launches == tests/golden/engine_own_sha256.json (output of oracle/dct_oracle.c for the same synthetic planes)"}

    c3 = extras.get("config3_420_roundtrip_one_call", {})
    if "GBps" in c3:
        fpx3 = sum(w * h for (w, h, _, _) in synth.CONFIG3_PLANES)
        line["roofline_config3_420"] = own_block(c3, "mdct::k_i16_batch<MODE_ROUNDTRIP, every plane its table, no saturations>", 4 * fpx3,
                                                 "the configs[2] frame held as int16 planes (4 B/px; rounds 2-4 measured configs[2] in this form; the 8-bit form SURVEY.md 8(d) specifies is roofline_config3_420_u8): "
                                                 "Y 7680x4320 + Cb/Cr 3840x2160, Annex-K tables, fused fwd+inv, ONE call (mdct_roundtrip_i16_planes) = one launch",
                                                 "k_i16_batch_420_bytes_per_launch")
        line["roofline_config3_420"]["Mpx_s"] = c3.get("Mpx_s")
        tiles3 = sum(((w // 8 + 63) // 64) * (h // 8) for (w, h, _, _) in synth.CONFIG3_PLANES)
        line["roofline_config3_420"]["valu"] = valu(ISA_NAME["k_i16_batch_420"], tiles3, c3["ms"], c3.get("clock_GHz"), "k_i16_batch_420")
        line["roofline_config3_420"]["device_table_form_ms"] = c3.get("device_table_form", {}).get("ms")
        line["roofline_config3_420"]["forward_only_batch_ms"] = c3.get("forward_only_batch", {}).get("ms")
        line["roofline_config3_420"]["four_frames_per_call_ms_per_frame"] = c3.get("four_frames_per_call_ms_per_frame")
        line["roofline_config3_420"]["two_streams_ms_per_frame"] = c3.get("two_streams_ms_per_frame")
    c3u = extras.get("config3_420_u8_roundtrip_one_call", {})
    if "GBps" in c3u:
        fpx3 = sum(w * h for (w, h, _, _) in synth.CONFIG3_PLANES)
        blk = own_block(c3u, "mdct::k_u8_batch<tame tables: no saturations, v_sat_pk_u8_i16 output stage>", 2 * fpx3,
                        "BASELINE.json configs[2] as SURVEY.md 8(d) defines it: Y 7680x4320 + Cb/Cr 3840x2160 8-bit planes in, 8-bit planes out, Annex-K tables, "
                        "forward -> quantise -> dequantise -> inverse fused, ONE call (mdct_batch_run of mdct_batch_create_u8) = one launch",
                        "k_u8_batch_420_bytes_per_launch")
        blk["bound"] = "vector issue at the clock the chip holds under this kernel (valu.frac_of_valu_floor); the bytes alone would take 99.5 MB / measured copy rate, DESIGN.md 4.2b"
        tiles3 = sum(((w // 8 + 63) // 64) * (h // 8) for (w, h, _, _) in synth.CONFIG3_PLANES)
        blk["valu"] = valu(ISA_NAME["k_u8_batch_420"], tiles3, c3u["ms"], c3u.get("clock_GHz"), "k_u8_batch_420")
        blk["parity"] = "unpinned by the reference (it has no inverse); pinned by the CPU checker's composition orc_fwd_u8_i16 -> orc_inv_i16_u8 and equal to the two-call path on the device (tests/test_u8_roundtrip.py)"
        for k in ("Mpx_s", "launches_per_call", "kernel_argument_form", "kernel_argument_form_equals_device_table_form", "four_frames_per_call_ms_per_frame", "y_plane_two_calls_ms", "y_plane_fused_ms",
                  "plane_8192_ms", "plane_8192_Mpx_s", "forward_only_batch", "inverse_only_batch", "forward_then_inverse_equals_fused",
                  "reference_q32_product_one_launch", "reference_q32_product_three_calls_ms"):
            blk[k] = c3u.get(k)
        line["roofline_config3_420_u8"] = blk
        q1f = c3u.get("reference_q32_product_one_launch") or {}
        if "GBps" in q1f:  # the same frame as the REFERENCE's product: its own block, parity pinned by the reference
            line["roofline_q32_frame_420"] = {
                "bound": "vector issue at the clock the chip holds under this kernel (valu.frac_of_valu_floor), like roofline_u8",
                "kernel": "mdct::k_q32_batch<fast quantiser>", "what": "Y 7680x4320 + Cb/Cr 3840x2160 8-bit planes -> the reference's q32 product of each (own table, own output buffer), "
                "ONE launch (mdct_batch_run of mdct_batch_create_q32) where the reference's caller makes three calls (main.cpp:543)",
                "reference": "simdDCT_EncodeQuantize32ReorderBuffer_AVX2_Float per plane, simd_dct.cpp:2064-2262",
                "parity": "pinned by the reference: every plane equals the single-plane call on the device in this run (equals_three_single_plane_calls), whose 8192^2 output "
                          "hashes to the real reference's (roofline_u8); against the checker: tests/test_q32_batch.py",
                "achieved": q1f["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(q1f["GBps"] / HBM_PEAK_GBPS, 4),
                "algorithmic_bytes_per_launch": 2 * fpx3, "avg_launch_ms": q1f["ms"], "Mpx_s": round(fpx3 / (q1f["ms"] * 1e-3) / 1e6, 0),
                "three_single_plane_calls_ms": c3u.get("reference_q32_product_three_calls_ms"), "equals_three_single_plane_calls": q1f.get("equals_three_single_plane_calls"),
                "traffic": traffic.get("k_q32_batch_420_bytes_per_launch"), "traffic_round": traffic.get("traffic_round"),
                "valu": valu(ISA_NAME["k_q32_batch_420"], tiles3, q1f["ms"], q1f.get("clock_GHz"), "k_q32_batch_420")}
    c5 = extras.get("config5_f32_fwd", {})
    if "GBps" in c5:
        line["roofline_f32"] = own_block(c5, "mdct::k_f32_tile<MODE_FWD>", 8 * W * H, "BASELINE.json configs[4]: float32 DCT-II, 8192x8192 plane (mdct_fwd_f32)", "k_f32_tile_fwd_bytes_per_launch")
        line["roofline_f32"]["Mpx_s"] = c5.get("Mpx_s")
        line["roofline_f32"]["tolerance"] = {"stated": "1e-5 of the block's max-abs coefficient vs a double-precision DCT-II (SURVEY.md 8c)",
                                             "cpu_checker_output_with_this_hash": c5.get("cpu_checker_max_err_over_block_max_vs_double")}
        if "GBps" in c5.get("stream_copy_same_bytes", {}):
            line["roofline_f32"]["frac_of_measured_copy_same_bytes"] = round(c5["GBps"] / c5["stream_copy_same_bytes"]["GBps"], 3)
    c4 = extras.get("config4_256_planes_one_gpu", {})
    if "GBps" in c4.get("one_call_device_table", {}):
        bpx4 = 256 * 4096 * 4096
        blk = own_block(dict(c4["one_call_device_table"], sha256_equals_cpu_checker=c4.get("sha256_equals_cpu_checker")), "mdct::k_i16_batch<MODE_FWD>", 4 * bpx4,
                        "BASELINE.json configs[3] on ONE GPU: 256 separately allocated 4096x4096 int16 planes, forward only, one call of mdct_batch_run (one launch)",
                        "k_i16_batch_fwd_256_bytes_per_launch")
        blk["verified_by"] += " for planes 0, 1, 255; every plane == the stacked single launch on the device"
        blk["every_plane_equals_the_stacked_launch"] = c4.get("every_plane_equals_the_stacked_launch")
        blk["Mpx_s"] = c4.get("Mpx_s_one_call")
        for k in ("one_call_kernel_arguments", "stacked_one_launch", "one_launch_per_plane"):
            if "GBps" in c4.get(k, {}):
                blk[k] = {"ms": c4[k]["ms"], "GBps": c4[k]["GBps"], "frac": round(c4[k]["GBps"] / HBM_PEAK_GBPS, 4)}
        line["roofline_config4_one_gpu"] = blk
    if "GBps" in extras.get("fwd_quant_u8_q32", {}):
        line["roofline_u8"] = u8_block(extras["fwd_quant_u8_q32"], "k_q32_avx")
    if "GBps" in extras.get("fwd_quant_u8_stereo_sse", {}):
        line["roofline_stereo"] = u8_block(extras["fwd_quant_u8_stereo_sse"], "k_stereo_sse")
        line["roofline_stereo"]["scalar_tier"] = u8_block(extras.get("fwd_quant_u8_stereo_scalar"), "k_stereo_scalar")
    if "GBps" in extras.get("fwd_quant_u8_encq_sse", {}):
        line["roofline_encq"] = u8_block(extras["fwd_quant_u8_encq_sse"], "k_encq_sse")
        line["roofline_encq"]["scalar_tier"] = u8_block(extras.get("fwd_quant_u8_encq_scalar"), "k_encq_scalar")
        # the SSE encq tier writes only half of every block pair (simd_dct.cpp:1662-1676): `frac` above charges the layout's
        # nominal 2 B/px; on the bytes the tier really moves (1 B/px in + 0.5 B/px out + the one spill) it is lower
        moved = W * H + W * H // 2 + 64
        gb = moved / (extras["fwd_quant_u8_encq_sse"]["ms"] * 1e-3) / 1e9
        line["roofline_encq"]["on_bytes_really_moved"] = {"bytes_per_launch": moved, "achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBPS, 4),
                                                          "frac_of_measured_copy": round(gb / copy, 3) if copy else None}
    return line, traffic


def allgather_leg(M, torch, dist, synth, rank, world, ranks_seen):
    """north_star's whole-node run at configs[3]'s own shape (never `value`)"""
    try:
        from simd_dct_amd.sharding import equal_shards, shard_planes

        PW = PH = 4096
        NPL = 256
        if not equal_shards(NPL, world):
            raise RuntimeError(f"{NPL} planes do not split evenly over {world} ranks (all_gather_into_tensor needs equal shards)")
        torch.cuda.empty_cache()
        p0, p1 = shard_planes(NPL, world, rank)
        per = p1 - p0
        chunk = min(8, per)
        nch = per // chunk
        src4 = torch.empty((per * PH, PW), dtype=torch.int16, device="cuda")
        for i in range(per):
            src4[i * PH:(i + 1) * PH] = synth.plane_i16_torch(PW, PH, "photo", seed=synth.SEED + 1000 + p0 + i)
        gbuf = torch.zeros((nch, world, chunk * PH, PW), dtype=torch.int16, device="cuda")  # [chunk][owner rank][planes of the chunk]
        fwd = [M.prepare_plane_i16("fwd", src4[c * chunk * PH:(c + 1) * chunk * PH], gbuf[c, rank], PW, chunk * PH) for c in range(nch)]
        flat = [gbuf[c].view(torch.uint8).reshape(-1) for c in range(nch)]
        mine = [gbuf[c, rank].view(torch.uint8).reshape(-1) for c in range(nch)]

        def timed(body, reps=3):
            body()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                body()
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item()

        def compute_only():
            for c in range(nch):
                fwd[c]()

        def gather_only():
            for c in range(nch):
                dist.all_gather_into_tensor(flat[c], mine[c])

        def pipelined():
            works = []
            for c in range(nch):
                fwd[c]()
                works.append(dist.all_gather_into_tensor(flat[c], mine[c], async_op=True))
            for w in works:
                w.wait()

        tc, tg, tp = timed(compute_only), timed(gather_only), timed(pipelined)
        # every slot every rank now holds must be what its owner computed: compare checksums of all slots
        sums = gbuf.to(torch.int64).sum(dim=(2, 3))  # [chunk][owner]
        own = sums[:, rank].contiguous()
        allown = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(allown, own)
        gathered_ok = all(torch.equal(sums[:, r], allown[r]) for r in range(world))
        batch_px = NPL * PW * PH
        out_bytes = batch_px * 2
        allgather = {"what": f"configs[3]: {NPL} planes of {PW}x{PH} int16, forward only, {per} planes per rank in chunks of {chunk}, "
                             f"in-place all_gather_into_tensor (RCCL) of every chunk to all {world} ranks",
                     "seconds_per_batch": {"compute_only": round(tc, 5), "gather_only": round(tg, 5), "pipelined": round(tp, 5)},
                     "Mpx_s_whole_batch_pipelined": round(batch_px / tp / 1e6, 0), "Mpx_s_compute_only": round(batch_px / tc / 1e6, 0),
                     "busbw_GBps_gather_only": round((world - 1) / world * out_bytes / tg / 1e9, 1),
                     "xgmi_ceiling_GBps_per_gpu": 7 * 153, "busbw_frac_of_xgmi_ceiling": round((world - 1) / world * out_bytes / tg / 1e9 / (7 * 153), 3),
                     "ranks_seen": ranks_seen,
                     "gathered_checksums_match_owners": bool(gathered_ok)}
        del src4, gbuf
    except Exception as e:
        allgather = {"error": str(e)[:200]}
    return allgather


def cabi_gather_leg(M, torch, dist, synth, rank, world):
    """the same gather through the C-ABI's own RCCL leg (mdct_comm_* / mdct_allgather_rows, csrc/comm.hip); result to stderr"""
    import sys

    try:
        ident = [M.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        comm = M.Comm(rank, world, ident[0])
        PW = PH = 4096
        NPL = 64  # a quarter of configs[3]'s batch is enough for a rate
        rows = NPL * PH // 8
        buf = torch.zeros((NPL * PH, PW), dtype=torch.int16, device="cuda")
        src = synth.plane_i16_torch(PW, PH, "photo", seed=synth.SEED + 7)
        b0, b1 = M.shard_rows_c(rows, world, rank)
        srcs4 = src.repeat(((b1 - b0) * 8 + PH - 1) // PH + 1, 1)[: (b1 - b0) * 8]
        M.fwd_i16(srcs4, buf[b0 * 8:b1 * 8], PW, (b1 - b0) * 8)  # this rank's block rows, in place in the full buffer
        comm.allgather_rows(buf, 8 * PW * 2, rows)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            comm.allgather_rows(buf, 8 * PW * 2, rows)
        torch.cuda.synchronize()
        dist.barrier()
        tg = (time.perf_counter() - t0) / reps
        # every plane of the gathered buffer is the same picture's coefficients: compare all shards with this rank's own
        want = buf[b0 * 8:b0 * 8 + PH] if (b0 * 8) % PH == 0 else None
        ok = bool(all(torch.equal(buf[p * PH:(p + 1) * PH], want) for p in range(NPL))) if want is not None else None
        if rank == 0:
            nbytes = NPL * PW * PH * 2
            print("[bench cabi-gather] " + json.dumps({"what": f"{NPL} planes of {PW}x{PH} int16 coefficients, block rows sharded over {world} ranks, mdct_allgather_rows (RCCL via the C-ABI)",
                                                      "seconds": round(tg, 6), "busbw_GBps": round((world - 1) / world * nbytes / tg / 1e9, 1),
                                                      "all_planes_complete_on_rank0": ok}), file=sys.stderr, flush=True)
        comm.close()
    except Exception as e:
        if rank == 0:
            print("[bench cabi-gather] " + json.dumps({"error": str(e)[:200]}), file=sys.stderr, flush=True)


def compact_kernels(extras, blocks):
    """one number-only dict per extra kernel / configuration for bench.py's JSON line: ms per launch, GB/s on the algorithmic bytes,
    frac of the 8 TB/s spec, fvf = fraction of the kernel's vector-issue floor, the shader clock held under it, ok = output verified"""
    def entry(b, ok_keys=("bit_exact_vs_reference", "bit_exact_vs_cpu_checker", "equals_three_single_plane_calls")):
        if not isinstance(b, dict) or "achieved" not in b:
            return None
        e = {"ms": b.get("avg_launch_ms"), "GBps": b["achieved"], "frac": b.get("frac")}
        v = b.get("valu") or {}
        if v.get("frac_of_valu_floor") is not None:
            e["fvf"], e["clock_GHz"] = v["frac_of_valu_floor"], v.get("clock_GHz_under_this_kernel")
        oks = [b[k] for k in ok_keys if k in b and b[k] is not None]
        if oks:
            e["ok"] = all(o is True for o in oks)
        return e

    out = {}
    names = {"q32": blocks.get("roofline_u8"), "stereo_sse": blocks.get("roofline_stereo"), "stereo_scalar": (blocks.get("roofline_stereo") or {}).get("scalar_tier"),
             "encq_sse": blocks.get("roofline_encq"), "encq_scalar": (blocks.get("roofline_encq") or {}).get("scalar_tier"),
             "cfg3_u8_frame": blocks.get("roofline_config3_420_u8"), "cfg3_i16_frame": blocks.get("roofline_config3_420"), "cfg3_q32_frame": blocks.get("roofline_q32_frame_420"),
             "cfg4_256_planes": blocks.get("roofline_config4_one_gpu"), "cfg5_f32": blocks.get("roofline_f32")}
    for k, b in names.items():
        e = entry(b)
        if e:
            out[k] = e
    for k, src in (("copy", "stream_copy_roofline"), ("fwd_i16", "fwd_i16"), ("inv_i16", "inv_i16")):
        q = extras.get(src) or {}
        if "GBps" in q:
            out[k] = {"ms": q["ms"], "GBps": q["GBps"], "frac": round(q["GBps"] / HBM_PEAK_GBPS, 4)}
    return out
