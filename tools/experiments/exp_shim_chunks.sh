# host-pointer calls against the number of chunks the pipeline aims for.  Needs a build whose csrc/shim.hip reads the divisor of
# `chunk = clamp(bytes / 8, 1 MiB, 4 MiB)` from MDCT_SHIM_CHUNKS (a one-line change, not in the product); result: profiles/r05_exp_shim_chunks.log
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from simd_dct_amd import synth
synth.plane_u8_np(8192, 8192, "photo").tofile("/tmp/plane8192.raw")
PY
for n in 8 12 16 24 32; do
  for pin in "" "--pin"; do
    echo "== chunks $n $pin"
    MDCT_SHIM_CHUNKS=$n tools/simd_dct_cli /tmp/plane8192.raw 8192 8192 --quality 2000 --runs 16 $pin 2>&1 | grep "^enc-quant" | cut -c1-17,128-150,300-330
  done
done
