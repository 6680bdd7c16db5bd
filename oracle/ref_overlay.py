"""Import the *reference* Aesara (an overlay of /root/reference) as the checker / CPU baseline.

TEST INFRASTRUCTURE ONLY.  Only tests/, oracle/ scripts and bench.py's ``cpu_baseline`` and
``through_function`` legs may import this; the product package `aesara_amd` never does.

Where the reference front end comes from, in this order:

1. an overlay that is already built (``$AESARA_REF_OVERLAY/.built2``);
2. ``/root/reference`` (authoring container): ``build_ref_overlay.sh`` builds the overlay;
3. ``oracle/_ref/aesara_ref_overlay.tar.gz`` — the packed overlay ``pack_ref_overlay.sh`` writes
   (git-ignored build artefact, never committed; it travels to the GPU box with the snapshot the
   same way the built ``.so`` files do) — unpacked under ``/tmp``.

With none of them ``available()`` is False and everything that needs the front end skips.
"""
import os
import subprocess
import sys
import tarfile

_HERE = os.path.dirname(os.path.abspath(__file__))
OVERLAY = os.environ.get("AESARA_REF_OVERLAY", "/tmp/aesara_ref_overlay")
REFERENCE = os.environ.get("AESARA_REFERENCE", "/root/reference")
ARCHIVE = os.path.join(_HERE, "_ref", "aesara_ref_overlay.tar.gz")
COMPILEDIR = "/tmp/aesara_ref_compiledir" + os.environ.get("AESARA_REF_COMPILEDIR_SUFFIX", "")

_CXXFLAGS = (
    "-DNPY_PY3K=1 -DPyInt_AsLong=PyLong_AsLong -DPyInt_FromLong=PyLong_FromLong "
    "-DPyInt_Check=PyLong_Check -DPyInt_AS_LONG=PyLong_AsLong -DPyArray_MoveInto=PyArray_CopyInto "
    "-DPyString_FromString=PyUnicode_FromString -DPyString_Check=PyUnicode_Check "
    "-DPyString_AsString=PyUnicode_AsUTF8"
)


def _built():
    return os.path.isfile(os.path.join(OVERLAY, ".built2"))


def source():
    """Which of the three sources would be used: 'overlay', 'reference', 'archive' or None."""
    if _built():
        return "overlay"
    if os.path.isdir(os.path.join(REFERENCE, "aesara")):
        return "reference"
    if os.path.isfile(ARCHIVE):
        return "archive"
    return None


def available():
    return source() is not None


def _unpack():
    """Unpack the archive into OVERLAY — once, also when several processes (pytest-xdist workers,
    bench.py's baseline child next to a test) want it at the same moment: under a file lock, into
    a private directory that is renamed into place when it is complete."""
    import fcntl
    import shutil
    parent = os.path.dirname(OVERLAY.rstrip("/")) or "/"
    os.makedirs(parent, exist_ok=True)
    with open(OVERLAY.rstrip("/") + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if _built():
                return
            tmp = OVERLAY.rstrip("/") + ".unpack.%d" % os.getpid()
            shutil.rmtree(tmp, ignore_errors=True)
            os.makedirs(tmp)
            with tarfile.open(ARCHIVE) as tf:
                top = tf.getnames()[0].split("/")[0]
                tf.extractall(tmp)
            shutil.rmtree(OVERLAY, ignore_errors=True)
            os.replace(os.path.join(tmp, top), OVERLAY)
            shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def import_reference():
    """Make the overlay importable and return the imported reference `aesara` module."""
    if "aesara" in sys.modules:
        return sys.modules["aesara"]
    src = source()
    if src is None:
        raise ImportError("reference Aesara not present (no /root/reference, no packed overlay)")
    if src == "reference":
        subprocess.run([os.path.join(_HERE, "build_ref_overlay.sh")], check=True,
                       stdout=subprocess.DEVNULL)
    elif src == "archive":
        _unpack()
    stubs = os.path.join(_HERE, "stubs")
    for p in (stubs, OVERLAY):
        if p not in sys.path:
            sys.path.insert(0, p)
    flags = os.environ.get("AESARA_FLAGS", "")
    extra = f"base_compiledir={COMPILEDIR},gcc__cxxflags={_CXXFLAGS}"
    os.environ["AESARA_FLAGS"] = (flags + "," if flags else "") + extra
    import warnings

    warnings.filterwarnings("ignore")
    import np2shim  # noqa: F401  (must precede aesara)
    import aesara

    return aesara
