python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from simd_dct_amd import synth
synth.plane_u8_np(8192, 8192, "photo").tofile("/tmp/plane8192.raw")
PY
echo "== pageable"; tools/simd_dct_cli /tmp/plane8192.raw 8192 8192 --quality 2000 --runs 16 2>&1 | grep -v "^Features" | cut -c1-400
echo "== pinned (--pin)"; tools/simd_dct_cli /tmp/plane8192.raw 8192 8192 --quality 2000 --runs 16 --pin 2>&1 | grep -v "^Features" | cut -c1-400
