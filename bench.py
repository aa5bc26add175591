#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

    python bench.py --gpus N --steps K --warmup W

Workload (configs[1]): one 8192x8192 int16 plane per GPU per step, forward + inverse 8x8 DCT in ONE fused
kernel (4 algorithmic bytes per pixel: 2 in + 2 out).  Four distinct plane pairs (1 GiB) are rotated so that no
step can be served from the 256 MiB Infinity Cache.  Inputs are resident in HBM before the timed region.
N > 1: one process per GPU (torch.distributed / RCCL), every rank transforms its own planes -- the path shards by
independent planes / block rows with no data-path collective ("weak" scaling).  Started without WORLD_SIZE,
`--gpus N` launches its own `python -m torch.distributed.run` child and relays the child's line.

The LAST stdout line is one short JSON object (`build_line`, < 4 KB: the driver parses it).  `roofline` is the
fused round-trip kernel, timed with HIP events on the launch stream; `kernels` holds one number-only entry per
other kernel / configuration; `cpu_baseline` is bench_cpu.py.  Everything verbose -- sources, how each figure
was verified, the issue-floor model, the all-gather leg -- goes to bench_extras.json next to this file
(bench_extras.py) and to stderr.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W = H = 8192
NSETS = 4
PRECONDITION = int(os.environ.get("MDCT_BENCH_PRECONDITION", "1000"))  # untimed launches (~50 ms) before warmup, see main()
ALG_BYTES_PER_PX = 4  # int16 in + int16 out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
METRIC = "Mpixels/s 8x8 fwd+inv int16 DCT, 8192x8192 plane"
GATHER_TIMEOUT_S = float(os.environ.get("MDCT_BENCH_GATHER_TIMEOUT", "300"))  # watchdog of the optional all-gather legs
EXTRAS_FILE = os.environ.get("MDCT_BENCH_EXTRAS_FILE", os.path.join(ROOT, "bench_extras.json"))
LINE_LIMIT = 4096


def build_line(m):
    """The one JSON line, from the measured figures `m` (a plain dict; no GPU needed: tests/test_bench_line.py).
    Numbers and short names only -- prose belongs in bench_extras.json."""
    px = W * H
    wall, steps, world, kernel_ms = m["wall_s"], m["steps"], m["world"], m["kernel_ms"]
    achieved = px * ALG_BYTES_PER_PX / (kernel_ms * 1e-3) / 1e9
    copy = m.get("copy_GBps")
    line = {
        "metric": METRIC, "value": round(world * px * steps / wall / 1e6, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": m["warmup"],
        "ms_per_step": round(wall / steps * 1e3, 4), "kernel_ms": round(kernel_ms, 4), "value_hip_events": round(world * px / (kernel_ms * 1e-3) / 1e6, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "io_dtype": "int16", "data": "synthetic",
        "config": {"workload": "configs[1]: 8192x8192 int16 plane per GPU, fwd+inv 8x8 DCT fused in one kernel", "plane": [W, H], "rotating_plane_sets": NSETS,
                   "untimed_preconditioning_launches": m.get("precondition", PRECONDITION), "timed_region": "wall: before launch 1 -> polled stop event behind launch K",
                   "parallelism": f"independent planes x{world}" if world > 1 else "single GPU", "device": m.get("device", "?")},
        "roofline": {"bound": "hbm", "kernel": "mdct::k_i16_tile<MODE_ROUNDTRIP>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(achieved / copy, 3) if copy else None,
                     "traffic": m.get("traffic"), "traffic_round": m.get("traffic_round"), "algorithmic_bytes_per_launch": px * ALG_BYTES_PER_PX,
                     "avg_launch_ms": round(kernel_ms, 4), "parity": "unpinned by the reference (no int16/inverse there); CPU checker + bit-exact round trip"},
        "cold": {"cold_first_launch_ms": m.get("cold_first_launch_ms"), "from_idle_20_launch_ms": m.get("from_idle_20_launch_ms"), "steady_ms": round(kernel_ms, 4)},
        "bit_exact_roundtrip_verified": m["verified"],
    }
    if world > 1 or m.get("backend"):
        line["ranks_seen"], line["backend"], line["per_rank_Mpx_s"] = m.get("ranks_seen"), m.get("backend"), m.get("per_rank_Mpx_s")
    if m.get("kernels"):
        line["kernels"] = m["kernels"]
    if m.get("allgather"):
        line["allgather"] = m["allgather"]
    if m.get("cpu_baseline"):
        line["cpu_baseline"] = m["cpu_baseline"]
    if m.get("extras_file"):
        line["extras_file"] = m["extras_file"]
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("allgather", "kernels"):  # never let an optional block cost the driver its line
        if len(text) > LINE_LIMIT and drop in line:
            line[drop] = {"see": "bench_extras.json"}
            text = json.dumps(line, separators=(",", ":"))
    return line, text


def relaunch(args, argv):
    """`--gpus N` without WORLD_SIZE: start torch.distributed.run as a CHILD (never exec; nothing here has touched HIP), one rank per
    GPU, relay the child's JSON line as this process's last stdout line and exit with the child's code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + argv
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: child {' '.join(cmd[2:])}", file=sys.stderr, flush=True)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    last = None
    for ln in child.stdout.splitlines():
        try:
            if isinstance(json.loads(ln), dict):
                last = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)  # anything else a rank printed
    if last:
        print(last, flush=True)
    sys.exit(child.returncode if child.returncode or last else 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (default).  gloo only rehearses the multi-rank control flow on a box with fewer GPUs than ranks.")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args, sys.argv[1:])
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch

    import simd_dct_amd as M
    from simd_dct_amd import synth

    dist = None
    local = local % max(1, torch.cuda.device_count())  # rehearsal: more ranks than GPUs share a device
    torch.cuda.set_device(local)
    emit = lambda text: print(text, flush=True)
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run, also for one rank (exercises RCCL)
        import torch.distributed as dist

        # stdout carries exactly one line, the JSON: RCCL prints a version banner to file descriptor 1 when its first
        # communicator comes up, so fd 1 is pointed at stderr for the rest of the run and the line goes to a private copy
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)

        def emit(text):
            json_out.write(text + "\n")
            json_out.flush()

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    M.init(local)
    t_start = time.perf_counter()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    # ---- inputs resident in HBM before anything is timed
    srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + rank * NSETS + i) for i in range(NSETS)]
    dsts = [torch.empty_like(s) for s in srcs]
    torch.cuda.synchronize()
    log("inputs resident")
    # launches with their arguments marshalled once (the kernel runs ~45 us; re-deriving pointers, table and stream in Python per
    # call costs ~10 us and would starve the queue)
    steps = [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H) for i in range(NSETS)]
    step = lambda i: steps[i % NSETS]()
    timer = M.Timer()
    stream_arg = M.api._stream()  # torch's current stream, resolved once
    lib = M.api._lib.load()

    def clocked(n, first=0):
        """n launches back to back; (wall seconds to the polled stop event behind the n-th launch, HIP-event ms per launch)"""
        M.api._check(lib.mdct_timer_start(timer._t, stream_arg))
        t0 = time.perf_counter()
        for i in range(first, first + n):
            step(i)
        M.api._check(lib.mdct_timer_stop(timer._t, stream_arg))
        timer.wait_spin()  # polling: a blocking wait pays tens of microseconds of interrupt wake-up (profiles/r05_exp_timed_region.log)
        return time.perf_counter() - t0, timer.elapsed_ms() / n

    # ---- what a caller doing one plane at a time sees (no pre-conditioning): the first launch of the process (code-object load
    # included), then 20 launches from an idle chip.  Both also serve as the workload's correctness check below.
    cold_first_ms = clocked(1)[0] * 1e3
    time.sleep(0.25)
    from_idle_ms = clocked(20, first=1)[0] * 1e3 / 20
    torch.cuda.synchronize()
    verified = all(torch.equal(s, d) for s, d in zip(srcs, dsts))  # fused fwd->inv of every plane set is a bit-exact round trip
    if dist is not None:  # every rank's planes, not just rank 0's
        v = torch.tensor([1 if verified else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())
    if not verified:  # a broken kernel must not produce a headline number
        if rank == 0:
            emit(json.dumps({"metric": METRIC, "value": None, "unit": "Mpixels/s", "n_gpus": world, "error": "fused forward+inverse round trip is not bit-exact on at least one rank"}))
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed pre-conditioning.  From idle the chip's power management overshoots for the first ~400 launches (63 -> 47 us per
    # launch, profiles/r01_transient_from_idle.log) and any idle gap longer than ~1 ms restarts that.  The metric is the steady
    # state (the from-idle figures are reported beside it under "cold"), so: align the ranks, 1000 untimed launches back to back,
    # the W warmup steps, and into the timed region through the (sub-millisecond) barrier with no other work between.
    if dist is not None:
        dist.barrier()
    for i in range(PRECONDITION + args.warmup):
        step(i)
    # The timed region: barrier + synchronize, EXACTLY K steps, synchronize + barrier; each rank clocks its own K steps, the job's
    # time is the MAX over ranks.  Nothing but the K launches sits between the two clock reads (`clocked`).
    barrier()
    wall, kernel_ms = clocked(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    m = {"wall_s": wall, "kernel_ms": kernel_ms, "steps": args.steps, "warmup": args.warmup, "world": world, "verified": verified, "device": M.device_info()["name"],
         "cold_first_launch_ms": round(cold_first_ms, 3), "from_idle_20_launch_ms": round(from_idle_ms, 4)}
    if dist is not None:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([wall, kernel_ms, 1.0], dtype=torch.float64, device=dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # every rank's own clock, and a head count of the ranks that really took part
        m["per_rank_Mpx_s"] = [round(W * H * args.steps / float(e[0]) / 1e6, 1) for e in every]
        m["ranks_seen"] = int(round(sum(float(e[2]) for e in every)))
        m["backend"] = "rccl" if args.backend == "nccl" else "gloo (rehearsal)"
        m["wall_s"] = wall = max(float(e[0]) for e in every)
        m["kernel_ms"] = kernel_ms = max(float(e[1]) for e in every)
    log(f"timed region done: {wall / args.steps * 1e3:.4f} ms/step")

    verbose = {}
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        m["traffic"], m["traffic_round"] = traffic.get("k_i16_roundtrip_bytes_per_launch"), traffic.get("traffic_round")
    except Exception:
        pass
    if not args.no_extras:
        import numpy as np

        import bench_extras as X

        achieved = W * H * ALG_BYTES_PER_PX / (kernel_ms * 1e-3) / 1e9
        extras = X.measure(M, torch, np, synth, srcs, dsts, timer, rank, W, H, NSETS, achieved)
        blocks, _ = X.roofline_blocks(extras, synth, W, H)
        m["copy_GBps"] = extras.get("stream_copy_roofline", {}).get("GBps")
        m["kernels"] = X.compact_kernels(extras, blocks)
        verbose = {"extras": extras, "roofline_blocks": blocks}
        log("extras done")

    # The optional whole-node leg runs AFTER the headline is complete and under a watchdog: a collective that hangs on some node must
    # not cost the run its JSON line.  Exactly one line: whoever takes `line_lock` first and finds `line_out` unset prints it.  A
    # watchdog that ends a process which has touched the GPU exits NON-ZERO, so that torchrun and the driver see the hang.
    EXIT_GATHER_HUNG, EXIT_CABI_GATHER_HUNG = 17, 18
    line_lock = threading.Lock()
    line_out = [False]

    def emit_line_once(note=None):
        with line_lock:
            if line_out[0]:
                return
            line_out[0] = True
            if rank == 0:
                if note:
                    m["allgather"] = note
                m["extras_file"] = None
                if verbose:
                    try:
                        line, _ = build_line(m)
                        json.dump(dict(line, **verbose, allgather=m.get("allgather")), open(EXTRAS_FILE, "w"), indent=1)
                        m["extras_file"] = os.path.basename(EXTRAS_FILE)
                    except Exception as e:
                        log(f"could not write {EXTRAS_FILE}: {e}")
                emit(build_line(m)[1])

    def watchdog_fire():
        emit_line_once({"error": f"all-gather leg did not finish within {GATHER_TIMEOUT_S} s; headline unaffected; exit code {EXIT_GATHER_HUNG}"})
        print(f"[bench] rank {rank}: all-gather leg hung for {GATHER_TIMEOUT_S} s, exiting {EXIT_GATHER_HUNG}", file=sys.stderr, flush=True)
        os._exit(EXIT_GATHER_HUNG)

    gather_legs = dist is not None and not args.no_extras and args.backend == "nccl"
    if gather_legs:
        import bench_extras as X

        watchdog = threading.Timer(GATHER_TIMEOUT_S, watchdog_fire)
        watchdog.daemon = True
        watchdog.start()
        del srcs, dsts, steps, step  # 2 x 8 GiB are needed by the leg
        full = X.allgather_leg(M, torch, dist, synth, rank, world, m.get("ranks_seen"))
        watchdog.cancel()
        verbose["allgather_full"] = full
        with line_lock:
            if not line_out[0]:
                m["allgather"] = {k: full[k] for k in ("seconds_per_batch", "busbw_GBps_gather_only", "busbw_frac_of_xgmi_ceiling", "gathered_checksums_match_owners", "error") if k in full}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline ...")
        import bench_cpu

        m["cpu_baseline"] = bench_cpu.cpu_baseline()
    emit_line_once()
    if gather_legs:  # after the line (stdout stays one line whatever happens here); own watchdog: a hang must only not keep the job alive
        def tail_fire():
            print(f"[bench] rank {rank}: C-ABI gather leg hung for {GATHER_TIMEOUT_S} s, exiting {EXIT_CABI_GATHER_HUNG}", file=sys.stderr, flush=True)
            os._exit(EXIT_CABI_GATHER_HUNG)

        tail_dog = threading.Timer(GATHER_TIMEOUT_S, tail_fire)
        tail_dog.daemon = True
        tail_dog.start()
        X.cabi_gather_leg(M, torch, dist, synth, rank, world)
        tail_dog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
