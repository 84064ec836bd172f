#!/bin/bash
# oracle/_ref: the reference's two Cython twins (src/cython/*.pyx) compiled from the sources where they lie under
# /root/reference, with cython + g++ directly (not the reference's setup.py).  Build container only; outputs only
# into oracle/_ref/ (git-ignored).  These modules contain no pixel arithmetic of their own: they call cv2 / numpy like
# the Python classes, so they run here only with cv2 bound to tests/golden/cv2_oracle_shim.py
# (tests/golden/make_reference_runs.py uses them to check that the Cython twins and the Python classes agree).
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref/src/cython
mkdir -p "$OUT"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
for m in frame_enhancer_cython change_detector_cython; do
    cython -3 --cplus "$REF/src/cython/$m.pyx" -o "$OUT/$m.cpp"
    g++ -O2 -fPIC -shared -w -I"$PYINC" -I"$NPINC" "$OUT/$m.cpp" -o "$OUT/$m$EXT"
    rm -f "$OUT/$m.cpp"
done
ls "$OUT"
