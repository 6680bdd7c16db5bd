run() { env "$@" timeout 120 python tools/perf_probe.py --only cfg4b64 --out /dev/null 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l[:300]); continue
    print(r['tune'], '%.3f ms'%r['dev_ms'], '%.2f us/step'%r['us_per_step'])"; }
run AESARA_HIP_SM_XMODE=flag
run AESARA_HIP_SM_XMODE=granule AESARA_HIP_SM_CHUNK=32
