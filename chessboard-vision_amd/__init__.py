"""chessboard_vision_amd — MI355X-native digitisation path behind the call
sites of hericmr/chessboard-vision (ImageEnhancer, warp_image, split_board,
ChangeDetector, PieceDetector).  Pixel work runs in hand-written HIP kernels
(csrc/, gfx950) reached through a C-ABI (include/cbv.h) via ctypes; there is
no CPU fallback: importing a class that needs the device raises ImportError
when the library or a GPU is missing, which is the convention the
reference's own plugin selector relies on (frame_enhancer.py:13-21)."""

__version__ = "0.1.0"
