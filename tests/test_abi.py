"""The C-ABI library loads without a GPU, exports every symbol include/cbv.h
declares, and fails loudly (ImportError / CBV_ERR_NODEV) instead of falling
back when no device exists.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(names=("cbv.h", "cbv_chess.h")):
    out = []
    for name in names:
        txt = open(os.path.join(ROOT, "include", name)).read()
        out += re.findall(r"^CBV_API\s+[\w\s\*]+?\b(cbv_\w+)\s*\(", txt, flags=re.M)
    return out


def test_header_declares_the_boundary():
    syms = header_symbols()
    assert len(syms) == len(set(syms)) >= 40
    for must in ("cbv_process_pipeline", "cbv_warp_perspective", "cbv_squares_stats", "cbv_pipeline_run", "cbv_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from chessboard_vision_amd import _native as N
    from chessboard_vision_amd import chess_rules
    lib = N.load()
    chess_rules._L()
    for s in header_symbols():
        assert hasattr(lib, s), "libcbv_hip.so does not export %s" % s
        assert getattr(lib, s).argtypes is not None or s in ("cbv_device_count",), "no prototype bound for %s" % s


def test_struct_layouts_match_the_header():
    from chessboard_vision_amd import _native as N
    assert C.sizeof(N.ColorProfile) == 72 and C.sizeof(N.SqStats) == 72 and C.sizeof(N.Roi) == 16
    assert C.sizeof(N.FrameResult) == 64 and C.sizeof(N.Scene) == 32
    assert C.sizeof(N.EnhanceParams) == 72 + 8 + 8 + 8 + 16 + 36 + 4


def test_header_is_plain_c_and_every_struct_has_the_ctypes_layout(tmp_path):
    """include/cbv.h and include/cbv_chess.h compile as C99 (what a cgo / JNI / ctypes binding needs), and the size of
    every struct and the offset of every field the ctypes mirror names are what gcc lays out."""
    import shutil
    import subprocess
    from chessboard_vision_amd import _native as N
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    pairs = [("cbv_color_profile", N.ColorProfile), ("cbv_enhance_params", N.EnhanceParams), ("cbv_roi", N.Roi),
             ("cbv_square_view", N.SquareView), ("cbv_sq_stats", N.SqStats), ("cbv_scene", N.Scene),
             ("cbv_hough_params", N.HoughParams), ("cbv_hough_result", N.HoughResult), ("cbv_pipeline_config", N.PipelineConfig),
             ("cbv_frame_result", N.FrameResult), ("cbv_noise_result", N.NoiseResult), ("cbv_noise_state", N.NoiseDevState),
             ("cbv_host_image", N.HostImage), ("cbv_piece_result", N.PieceResult), ("cbv_detect_params", N.DetectParams),
             ("cbv_change_params", N.ChangeParams), ("cbv_change_result", N.ChangeResult)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "cbv.h"', '#include "cbv_chess.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), (cname, got[cname], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_no_device_means_import_error_not_fallback():
    from chessboard_vision_amd import _native as N
    lib = N.load()
    if lib.cbv_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.cbv_ctx_create(0, C.byref(h)) == -3 and not h
    assert b"no CPU fallback" in lib.cbv_last_error(None)
    with pytest.raises(ImportError):
        N.context()
    # the reference's selector pattern (frame_enhancer.py:13-21) must see ImportError
    with pytest.raises(ImportError):
        from chessboard_vision_amd.frame_enhancer import ImageEnhancer
        ImageEnhancer()


def test_host_only_perspective_transform_matches_oracle():
    from chessboard_vision_amd.board_detection import get_perspective_transform
    from chessboard_vision_amd import synth as S
    from oracle import cbv_oracle as O
    dst = np.float32([[0, 0], [620, 0], [0, 620], [620, 620]])
    for pts in (S.scaled_corners(1920, 1080), S.scaled_corners(3840, 2160), np.float32([[3, 7], [500, -20], [-40, 610], [700, 650]])):
        assert np.array_equal(get_perspective_transform(pts, dst), O.get_perspective_transform(pts, dst))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "chessboard-vision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "cbv_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
