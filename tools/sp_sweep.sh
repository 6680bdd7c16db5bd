run() { env "$@" timeout 120 python tools/perf_probe.py --only gruB1 --out gpurun_out/sp_sweep.jsonl 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l[:200]); continue
    print(r['tune'], '%.3f us/step'%r['us_per_step'], r.get('scan_modes'))"; }
export AESARA_HIP_SCAN_ROWS=8
run A=1
run AESARA_HIP_SP_POLLW=1
run AESARA_HIP_SP_POLLW=2
run AESARA_HIP_SP_SLEEP=0
run AESARA_HIP_SP_SLEEP=4
run AESARA_HIP_SP_REPOLL=1
run AESARA_HIP_SP_REPOLL=1 AESARA_HIP_SP_SLEEP=0
run AESARA_HIP_SP_REPOLL=1 AESARA_HIP_SP_POLLW=1
export AESARA_HIP_SCAN_ROWS=4
run AESARA_HIP_SP_REPOLL=1
run AESARA_HIP_SP_POLLW=1
