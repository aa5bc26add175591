#!/bin/bash
# The reference's UNCHANGED harness relinked against the engine (oracle/_ref/simd_dct_relinked_warm), host pointers, 8192^2 file:
# default / MDCT_SHIM_AUTOPIN=1 / MDCT_SHIM_EVENT_WAIT=block / both, in ONE box session (boxes differ by more than the effects).
cd "$(dirname "$0")/.."
RAW=${TMPDIR:-/tmp}/plane8192.raw
python3 - "$RAW" <<'PY'
import sys
sys.path.insert(0, ".")
from simd_dct_amd import synth
synth.plane_u8_np(8192, 8192, "photo").tofile(sys.argv[1])
PY
run() { # stops the whole script at the first run that fails or reports a GPU fault: nothing else is started on a box in that state
  echo "== $1"
  env $2 ./oracle/_ref/simd_dct_relinked_warm "$RAW" 8192 8192 --quality 2000 --runs ${RUNS:-48} --mode enc-quant32 --mode enc-quant-stereo --mode enc-quant > "$RAW.log" 2>&1
  rc=$?
  tr "\r" "\n" < "$RAW.log" | grep -v "^Features\|^$\|^ *[0-9]*:" 
  if [ $rc -ne 0 ] || grep -q "Memory access fault\|HSA_STATUS_ERROR" "$RAW.log"; then echo "FAILED (rc=$rc): stopping"; rm -f "$RAW" "$RAW.log"; exit 1; fi
}
for round in 1 2; do
  run "default (round $round)" "A=1"
  run "MDCT_SHIM_AUTOPIN=1 (round $round)" "MDCT_SHIM_AUTOPIN=1"
  run "MDCT_SHIM_EVENT_WAIT=block (round $round)" "MDCT_SHIM_EVENT_WAIT=block"
  run "MDCT_SHIM_AUTOPIN=1 MDCT_SHIM_EVENT_WAIT=block (round $round)" "MDCT_SHIM_AUTOPIN=1 MDCT_SHIM_EVENT_WAIT=block"
done
rm -f "$RAW" "$RAW.log"
