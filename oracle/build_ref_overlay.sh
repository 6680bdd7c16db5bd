#!/bin/bash
# TEST INFRASTRUCTURE ONLY (authoring container).  Builds a writable, importable overlay of
# the read-only reference at /root/reference so the reference's own C/py linkers can be run
# to generate golden vectors.  The overlay lives OUTSIDE the repo
# ($AESARA_REF_OVERLAY, default /tmp/aesara_ref_overlay): no reference source is ever copied
# into /root/repo, and nothing here runs on the GPU box.
set -euo pipefail
REF=${AESARA_REFERENCE:-/root/reference}
OVL=${AESARA_REF_OVERLAY:-/tmp/aesara_ref_overlay}
if [ ! -d "$REF/aesara" ]; then echo "no reference at $REF" >&2; exit 3; fi
if [ -f "$OVL/.built2" ]; then echo "$OVL"; exit 0; fi
rm -rf "$OVL"; mkdir -p "$OVL"
cp -r "$REF/aesara" "$OVL/"
# the reference's own test suite (tests/test_gpu_reference_suites.py runs its backend-parameterised
# classes under the HIP mode); its top-level package is called `tests`
cp -r "$REF/tests" "$OVL/"; chmod -R u+w "$OVL"
# 1. hatch-vcs generated version file absent from the archive (aesara/version.py:1-8)
echo '__version__ = "2.9.4+ref"' > "$OVL/aesara/_version.py"
# 2. NumPy-2 C-API: PyArray_DESCR(x)->elsize is gone (tensor/blas.py:575,2474; blas_headers.py:1083)
sed -i -E 's/PyArray_DESCR\(([^)]*\)s?)\)->elsize/PyArray_ITEMSIZE(\1)/g' \
    "$OVL/aesara/tensor/blas.py" "$OVL/aesara/tensor/blas_headers.py"
# 3. regenerate the vendored Cython scan loop with the installed Cython (scan/scan_perform_ext.py:3-7)
(cd "$OVL/aesara/scan" && cython -3 scan_perform.pyx -o c_code/scan_perform.c >/dev/null 2>&1 || true)
touch "$OVL/.built2"
echo "$OVL"
