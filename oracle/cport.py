"""ctypes loader for the oracle's C port (oracle/c_port.c).  TEST/BASELINE INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcport.so")


def load():
    if not os.path.exists(_PATH):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(_PATH)
    lib.cport_cfg2_eval.restype = C.c_double
    lib.cport_cfg2_eval.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_double]
    lib.cport_cfg2_eval_omp.restype = C.c_double
    lib.cport_cfg2_eval_omp.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_double, C.c_int]
    lib.cport_cfg1b_add.restype = None
    lib.cport_cfg1b_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    return lib


def cfg2_eval(x, mu, sigma):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return load().cport_cfg2_eval(x.ctypes.data, x.size, float(mu), float(sigma))


def cfg2_eval_omp(x, mu, sigma, threads):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return load().cport_cfg2_eval_omp(x.ctypes.data, x.size, float(mu), float(sigma), int(threads))
