run() { env "$@" timeout 120 python tools/perf_probe.py --only gruB1 --out gpurun_out/sp_sweep2.jsonl 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l[:300]); continue
    print(r['tune'], '%.3f us/step'%r['us_per_step'], r.get('scan_modes'))"; }
run AESARA_HIP_SCAN_ROWS=8 AESARA_HIP_SCAN_WAVES=4
run AESARA_HIP_SCAN_ROWS=8 AESARA_HIP_SCAN_WAVES=8
run AESARA_HIP_SCAN_ROWS=16 AESARA_HIP_SCAN_WAVES=8
run AESARA_HIP_SCAN_ROWS=16 AESARA_HIP_SCAN_WAVES=4
run AESARA_HIP_SCAN_ROWS=32 AESARA_HIP_SCAN_WAVES=8
run AESARA_HIP_SCAN_ROWS=16 AESARA_HIP_SCAN_WAVES=8 AESARA_HIP_SP_POLLW=4
run AESARA_HIP_SCAN_ROWS=16 AESARA_HIP_SCAN_WAVES=8 AESARA_HIP_SP_POLLW=1
run AESARA_HIP_SCAN_ROWS=32 AESARA_HIP_SCAN_WAVES=8 AESARA_HIP_SP_POLLW=4
run AESARA_HIP_SCAN_ROWS=32 AESARA_HIP_SCAN_WAVES=8 AESARA_HIP_SP_SLEEP=0
