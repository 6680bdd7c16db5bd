cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time
sys.path[:0]=[os.getcwd(), os.getcwd()+"/oracle", os.getcwd()+"/tests"]
import reference_files as rf
t=time.time()
rep=rf.run("device", ["tests/scan/test_basic.py::TestScan::test_grad_mitsot","tests/scan/test_basic.py::TestScan::test_R_op"], workers=1, timeout=900, extra=["--durations=5"])
print(time.time()-t, rep)
PY
