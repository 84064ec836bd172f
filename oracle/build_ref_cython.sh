#!/bin/bash
# The reference's two Cython twins (src/cython/*.pyx) compiled from the sources where they lie under /root/reference,
# with cython + g++ directly (not the reference's setup.py), into the directory given as $2: a scratch directory OUTSIDE
# this repository (tests/golden/make_reference_runs.py passes its temporary directory and deletes it afterwards), so no
# compiled reference code ever sits in the tree that travels to the GPU box.  Build container only.  The modules contain
# no pixel arithmetic of their own: they call cv2 / numpy like the Python classes, so they run here only with cv2 bound
# to tests/golden/cv2_oracle_shim.py; the generator uses them to check that twins and Python classes agree.
set -e
REF=${1:-/root/reference}
DEST=${2:?usage: build_ref_cython.sh <reference dir> <scratch output dir outside the repository>}
HERE=$(cd "$(dirname "$0")" && pwd)
case "$(cd "$(dirname "$DEST")" && pwd)/" in "$(dirname "$HERE")"/*) echo "refusing to build reference code inside the repository" >&2; exit 2;; esac
OUT=$DEST/src/cython
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
