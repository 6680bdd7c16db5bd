"""Import the *reference* Aesara (overlay of /root/reference) in the authoring container.

TEST INFRASTRUCTURE ONLY.  Only tests/, oracle/ scripts and bench.py's cpu_baseline leg may
import this; the product package `aesara_amd` never does.  On the GPU box `/root/reference`
does not exist and `available()` returns False.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
OVERLAY = os.environ.get("AESARA_REF_OVERLAY", "/tmp/aesara_ref_overlay")
REFERENCE = os.environ.get("AESARA_REFERENCE", "/root/reference")

_CXXFLAGS = (
    "-DNPY_PY3K=1 -DPyInt_AsLong=PyLong_AsLong -DPyInt_FromLong=PyLong_FromLong "
    "-DPyInt_Check=PyLong_Check -DPyInt_AS_LONG=PyLong_AsLong -DPyArray_MoveInto=PyArray_CopyInto "
    "-DPyString_FromString=PyUnicode_FromString -DPyString_Check=PyUnicode_Check "
    "-DPyString_AsString=PyUnicode_AsUTF8"
)


def available():
    return os.path.isdir(os.path.join(REFERENCE, "aesara"))


def import_reference():
    """Build the overlay if needed and return the imported reference `aesara` module."""
    if "aesara" in sys.modules:
        return sys.modules["aesara"]
    if not available():
        raise ImportError("reference Aesara not present (expected on the GPU box)")
    subprocess.run([os.path.join(_HERE, "build_ref_overlay.sh")], check=True,
                   stdout=subprocess.DEVNULL)
    stubs = os.path.join(_HERE, "stubs")
    for p in (stubs, OVERLAY):
        if p not in sys.path:
            sys.path.insert(0, p)
    flags = os.environ.get("AESARA_FLAGS", "")
    extra = f"base_compiledir=/tmp/aesara_ref_compiledir,gcc__cxxflags={_CXXFLAGS}"
    os.environ["AESARA_FLAGS"] = (flags + "," if flags else "") + extra
    import warnings

    warnings.filterwarnings("ignore")
    import np2shim  # noqa: F401  (must precede aesara)
    import aesara

    return aesara
