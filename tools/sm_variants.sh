#!/bin/bash
# A/B list of schedules of the matrix-state Scan kernel: result check + device time (sm_ab.py) and,
# for the names listed in TRACE, the per-phase timeline (sm_trace.py).
# usage: [TRACE="a1 a3"] tools/sm_variants.sh out_dir [list]       (list: "base" or "sweep")
out=${1:-gpurun_out/sm_variants}; mkdir -p $out
run() { name=$1; shift
  [ -n "$DRY" ] && { echo "$name $*"; return; }
  env "$@" timeout 200 python tools/sm_ab.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/$name /" | tee -a $out/ab.txt
  case " $TRACE " in *" $name "*) env "$@" timeout 200 python tools/sm_trace.py > $out/trace_$name.json 2>$out/trace_$name.err;; esac; }
P="AESARA_HIP_SM_INIT=publish AESARA_HIP_SM_PIN=2"
S="$P AESARA_HIP_SM_NXT=fetch AESARA_HIP_SM_ACKFILL=1"
k() { for kv in "$@"; do echo -n "AESARA_HIP_SM_$kv "; done; }
run base AESARA_HIP_SM_INIT=branch
run pub $P
run s5p4l3 $P $(k ACKFILL=1 NXT=ack XRELOAD=early XPRE=4 LOOK=3 XTAIL=8)
run a1 $S $(k EPRE=6 ELOOK=5 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=8)
run a2 $S $(k EPRE=6 ELOOK=7 XSPLIT=8 XPRE=6 LOOK=8 XTAIL=8)
run a3 $S $(k EPRE=6 ELOOK=6 XSPLIT=4 XPRE=6 LOOK=8 XTAIL=8)
run a4 $S $(k EPRE=8 ELOOK=4 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=8)
run a5 $S $(k EPRE=6 ELOOK=9 XSPLIT=8 XPRE=6 LOOK=10 XTAIL=8)
run a6 $S $(k EPRE=6 ELOOK=5 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=6)
run a7 $P $(k ACKFILL=1 NXT=top EPRE=6 ELOOK=5 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=8)
run a8 $S $(k EPRE=6 ELOOK=5 XSPLIT=12 XPRE=6 LOOK=4 XTAIL=8)
run a9 $S $(k EPRE=4 ELOOK=6 XSPLIT=8 XPRE=4 LOOK=8 XTAIL=8)
run a10 $S $(k EPRE=6 ELOOK=5 XSPLIT=8 XPRE=8 LOOK=6 XTAIL=8)
run a11 $S $(k EPRE=6 ELOOK=5 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=10)
run a12 $S $(k EPRE=6 ELOOK=3 XSPLIT=8 XPRE=6 LOOK=3 XTAIL=8)
run a13 $S $(k EPRE=16 XSPLIT=8 XPRE=6 LOOK=6 XTAIL=8)
