# A/B of the bounce-buffer copies: needs a build whose csrc/shim_host.h copy_pieces() switches to a streaming-store loop unless MDCT_EXP_NO_STREAMING_COPY is set
# (not in the product); result: profiles/r05_exp_streaming_copy.log
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from simd_dct_amd import synth
synth.plane_u8_np(8192, 8192, "photo").tofile("/tmp/plane8192.raw")
PY
for rep in 1 2 3; do
  for v in stream plain; do
    if [ $v = plain ]; then export MDCT_EXP_NO_STREAMING_COPY=1; else unset MDCT_EXP_NO_STREAMING_COPY; fi
    echo "== $v"; tools/simd_dct_cli /tmp/plane8192.raw 8192 8192 --quality 2000 --runs 16 2>&1 | grep "^enc-quant" | awk -F'|' '{print $1, $7, $NF}'
  done
done
