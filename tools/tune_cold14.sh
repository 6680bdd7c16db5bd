# (the s_setprio-by-wavefront experiment, AESARA_HIP_WAVE_PRIO, was removed again after this run: no effect)
#!/bin/bash
# round 5: wavefront priorities inside the one workgroup of a CU (the timeline: wave 0 is done 3.2 us
# before its workgroup's slowest wavefront in the exp-sum kernel, 0.9 us in the pure sum)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); c=r["config"]
        print("%-34s window %.2f us (%.4f)  sustained %.4f  exec %.4f  ceiling %.4f" % (os.environ.get("TAG",""), r["roofline"]["kernel_ms"]*1e3, r["roofline"]["frac"], c["sustained"]["frac"], c["executor_level"]["frac"], c["read_only_ceiling"]["frac"]))'
run() { TAG="$*" env "$@" timeout 300 python bench.py --no-cpu-baseline --no-warm --no-secondary --steps 20 --warmup 5 2>&1 | grep -v amdgpu | TAG="$*" python -c "$fmt"; }
run A=default
run AESARA_HIP_WAVE_PRIO=1
run AESARA_HIP_WAVE_PRIO=2
run A=default
run AESARA_HIP_WAVE_PRIO=1
run AESARA_HIP_WAVE_PRIO=2
