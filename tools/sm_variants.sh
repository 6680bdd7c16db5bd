#!/bin/bash
# A/B list of schedules of the matrix-state Scan kernel: result check + device time (sm_ab.py) and,
# for the names listed in TRACE, the per-phase timeline (sm_trace.py).
# usage: [TRACE="a1 a3"] [DRY=1] tools/sm_variants.sh out_dir
out=${1:-gpurun_out/sm_variants}; mkdir -p $out
run() { name=$1; shift
  [ -n "$DRY" ] && { echo "$name $*"; return; }
  env "$@" timeout 200 python tools/sm_ab.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/$name /" | tee -a $out/ab.txt
  case " $TRACE " in *" $name "*) env "$@" timeout 200 python tools/sm_trace.py > $out/trace_$name.json 2>$out/trace_$name.err;; esac; }
k() { for kv in "$@"; do echo -n "AESARA_HIP_SM_$kv "; done; }
run two $(k XREG=0)
run r3 $(k XREG=1)
run r3s8 $(k XSPLIT=8)
run r3s16 $(k XSPLIT=16)
run r3s16p8 $(k XSPLIT=16 XPRE=8 LOOK=10)
run r3s12 $(k XSPLIT=12 LOOK=8)
run r3s8l12 $(k XSPLIT=8 LOOK=12)
run r3s8l16 $(k XSPLIT=8 LOOK=16 XTAIL=10)
run r3e8 $(k XSPLIT=8 EPRE=8 ELOOK=5)
run r3t12 $(k XSPLIT=8 XTAIL=12)
run r3pin0 $(k XSPLIT=8 PIN=2 FENCE=0)
