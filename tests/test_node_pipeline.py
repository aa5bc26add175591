"""The whole-node run's control flow (tools/node_pipeline.h: chunked transform on a compute stream, each chunk's all-gather on a
second stream behind an event, so gather k runs under kernel k + 1) with world 1, 2 and 8 on the CPU: one process per rank,
worker-thread streams on host buffers, the product's mdct_allgather_rows over tests/fake_rccl.c, every gathered byte compared
after compute-only, gather-only and three pipelined passes (tests/node_pipeline_driver.cpp).  The same header drives
`tools/simd_dct_cli --gpus N --batch 256x4096x4096` on the GPUs (the reference's hook: simd_dct.cpp:2245-2255)."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    G.build_hip()
    exe = str(tmp_path / ("node_pipeline" + ("_san" if sanitizer else "")))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests", "node_pipeline_driver.cpp"),
           "-L" + os.path.join(ROOT, "simd_dct_amd"), "-lmdct_hip", "-Wl,-rpath," + os.path.join(ROOT, "simd_dct_amd"), "-pthread", "-o", exe]
    if sanitizer:
        cmd[4:4] = ["-fsanitize=" + sanitizer, "-fno-sanitize-recover=all"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and sanitizer and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    return exe


def _run_world(exe, fake, world, planes, elems, chunk, tmp_path, tag):
    idfile = str(tmp_path / f"id_{tag}_{world}")
    env = dict(os.environ, MDCT_RCCL_LIB=fake, TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    procs = [subprocess.Popen([exe, str(r), str(world), idfile, str(planes), str(elems), str(chunk)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=300)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if any("FATAL: ThreadSanitizer" in e and "memory mapping" in e for _, _, e in outs):
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this environment (address-space layout), not a finding")
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and "node pipeline ok" in o, f"rank {r} of {world} exited {rc}\n{o}\n{e[-3000:]}"
        assert "WARNING: ThreadSanitizer" not in e, e[-3000:]
    return outs


@pytest.mark.parametrize("world", [1, 2, 8])
def test_chunked_transform_and_gather_with_overlap(fake_rccl, tmp_path, world):
    """configs[3] in miniature: 32 planes of 4096 int16, shards of 32 / world planes in chunks of 2 (a wanted chunk of 3 is shrunk to a divisor)"""
    exe = _build(tmp_path, None)
    outs = _run_world(exe, fake_rccl, world, 32, 4096, 2, tmp_path, "a")
    assert f"{32 // world // 2} chunks of 2" in outs[0][1]
    outs = _run_world(exe, fake_rccl, world, 48, 1000, 4, tmp_path, "b")  # 48 / 8 = 6 planes per rank: chunk 4 -> 3
    if world == 8:
        assert "2 chunks of 3" in outs[0][1]


def test_pipeline_under_thread_sanitizer(fake_rccl, tmp_path):
    """the event between a chunk's kernel and its gather is what orders the two worker threads: built with -fsanitize=thread"""
    exe = _build(tmp_path, "thread")
    _run_world(exe, fake_rccl, 2, 16, 2048, 2, tmp_path, "tsan")
    _run_world(exe, fake_rccl, 1, 8, 2048, 2, tmp_path, "tsan1")


def test_the_test_notices_a_missing_event_wait(fake_rccl, tmp_path):
    """negative control: with the wait between kernel k and gather k left out, the gathers move canary bytes and the comparison fails"""
    exe = _build(tmp_path, None)
    idfile = str(tmp_path / "id_neg")
    env = dict(os.environ, MDCT_RCCL_LIB=fake_rccl, NODE_PIPELINE_SKIP_EVENT_WAIT="1")
    # two ranks: a gather that runs ahead of its chunk's kernel hands the peer canary bytes, which nothing overwrites later
    procs = [subprocess.Popen([exe, str(r), "2", idfile, "8", "2048", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) + (p.returncode,) for p in procs]
    assert all(rc != 0 for _, _, rc in outs) and any("mismatches" in e for _, e, _ in outs), outs


def test_cli_is_built_on_the_header_the_cpu_test_ran():
    src = open(os.path.join(ROOT, "tools", "simd_dct_cli.cpp")).read()
    assert '#include "node_pipeline.h"' in src and "mdct_node::Pipeline<HipNode>" in src and "mdct_node::make_shape(" in src
    hdr = open(os.path.join(ROOT, "tools", "node_pipeline.h")).read()
    code = "\n".join(l.split("//")[0] for l in hdr.splitlines())
    assert "hip" not in code.lower() and "nccl" not in code.lower()
