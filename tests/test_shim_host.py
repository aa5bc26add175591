"""Host-side sanitizers for the drop-in boundary (the GPU pool cannot run sanitizers).

csrc/shim_host.h is the host-only part of csrc/shim.hip -- the reference's row-range arithmetic (simd_dct.cpp:2243-2261,
:375-387), the tier choice from its CPU-flag globals (:78-85, :100-105, :120-127), the helper-thread CopyPool and the
chunked strip pipeline (one stream per stage, four slots) behind host-pointer calls.  tests/shim_host_driver.cpp runs it on worker-thread
"streams" over exact-size heap buffers: 4 callers x multi-chunk on disjoint ranges of the same planes, every early-error
path (failed copy / launch / stream wait at every chunk, buffers freed the moment the call returns), a thread that exits
with jobs queued -- under ThreadSanitizer and under AddressSanitizer + UBSan."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_shim_host_logic_under_sanitizers(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "shim_host"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "simd_dct_amd", "csrc"),
                        os.path.join(ROOT, "tests", "shim_host_driver.cpp"), "-o", str(exe), "-pthread"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout).lower() and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # the sanitizer runtime must come first in the library list
    env["TSAN_OPTIONS"] = "halt_on_error=1"
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "shim host ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_shim_hip_uses_the_host_header():
    """the product's shim really is built on the code the sanitizers ran: no second copy of the pool or the range arithmetic"""
    src = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "shim.hip")).read()
    assert '#include "shim_host.h"' in src and "mdct_host::StripPipeline<HipDev>" in src and "mdct_host::CopyPool<HipDev> pool_in, pool_out" in src
    assert "struct CopyPool" not in src and "void ref_range(" not in src
    hdr = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "shim_host.h")).read()
    code = "\n".join(l.split("//")[0] for l in hdr.splitlines())  # comments aside, nothing of the HIP runtime: it builds with plain g++
    assert "hip" not in code.lower()
