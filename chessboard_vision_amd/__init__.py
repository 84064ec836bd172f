"""Import shim: the product package lives in ``chessboard-vision_amd/`` (a name
Python cannot import directly); this module makes it importable as
``chessboard_vision_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "chessboard-vision_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
