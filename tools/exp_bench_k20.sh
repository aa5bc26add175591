# bench.py at the driver's --steps 20 against --steps 2000, for several lengths of untimed pre-conditioning (MDCT_BENCH_PRECONDITION)
show() { python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(sys.argv[1], {k:d[k] for k in ('value','value_hip_events','ms_per_step','kernel_ms')})" "$1"; }
for p in 1000 1000 4000 4000 16000 16000; do
  MDCT_BENCH_PRECONDITION=$p timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | show "precondition=$p steps=20"
done
timeout -k 10 100 python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | show "precondition=1000 steps=2000"
