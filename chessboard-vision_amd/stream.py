"""Device-resident batched path: enhance -> warp -> 64-square detect over
frames that stay in HBM (the composed chain of SURVEY.md §3 D).

One BoardPipeline = one camera stream on one GPU: an input frame ring, the
per-square temporal state of PieceDetector.detect_all_pieces
(reference squares, cached raw results, 5-frame history) and the result ring.
Streams are independent, so N GPUs run N pipelines with no exchange.
"""
import ctypes as C

import numpy as np

from . import _native as N
from . import synth as S
from .board_detection import get_perspective_transform
from .grid_extractor import GridExtractor, SmartGridExtractor


def bits_to_positions(bits, rois_rc):
    """u64 bitset over roi indices -> {(file, rank)} (a1 = (0,0), row 0 = rank 8)."""
    return {(c, 7 - r) for i, (r, c) in enumerate(rois_rc) if (bits >> i) & 1}


class BoardPipeline:
    def __init__(self, w, h, max_frames, ctx=None):
        self.ctx = ctx or N.context()
        self.w, self.h, self.max_frames = w, h, max_frames
        hdl = C.c_void_p()
        self.ctx.check(self.ctx.lib.cbv_pipeline_create(self.ctx.h, w, h, max_frames, C.byref(hdl)))
        self.h_ = hdl
        self.rois_rc = []
        self.board_size = 0

    def close(self):
        if self.h_:
            self.ctx.lib.cbv_pipeline_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, points, profile=None, grid_lines=None, rot180=False, chunk=0, lanes=0, keep_enhanced=False,
                  clahe_clip_limit=3.0, tile_grid_size=(8, 8), sharpen_kernel=None, display_size=(1280, 720), margin=100,
                  history_size=5, min_presence=0.6, change_threshold=25, z_threshold=2.5, initial_variance=100,
                  use_hough=True, min_radius_ratio=0.20, max_radius_ratio=0.55, hough_param1=100, hough_param2=25,
                  enhance_region=False):
        """`use_hough` and the radii mirror PieceDetector's attributes (piece_detector.py:33-35,222-230);
        pass min_radius / 100 and max_radius / 100 of piece_detector_settings.json as the application does.
        `enhance_region` (only without keep_enhanced): enhance the part of each frame the warp samples first and the rest
        only when normalize's global min / max could depend on it; every output stays identical (include/cbv.h)."""
        cfg = N.PipelineConfig()
        e = cfg.enhance
        e.profile = N.ColorProfile.from_dict(profile)
        e.clahe_clip_limit = clahe_clip_limit
        e.tiles_x, e.tiles_y = tile_grid_size
        e.bilateral_d, e.sigma_color, e.sigma_space = 9, 75.0, 75.0
        k = np.asarray(sharpen_kernel if sharpen_kernel is not None else [[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]],
                       dtype=np.float32).reshape(9)
        for i in range(9):
            e.sharpen_kernel[i] = float(k[i])
        S_ = min(display_size) - margin
        M = get_perspective_transform(np.float32(points), np.float32([[0, 0], [S_, 0], [0, S_], [S_, S_]]))
        for i in range(9):
            cfg.M[i] = float(M.reshape(9)[i])
        cfg.board_size, cfg.rot180 = S_, 1 if rot180 else 0
        if grid_lines is not None:
            ge = SmartGridExtractor()
            ge.grid_lines_x, ge.grid_lines_y = list(grid_lines[0]), list(grid_lines[1])
        else:
            ge = GridExtractor()
        table = ge.roi_table(S_, S_)
        cfg.n_rois = len(table)
        self.rois_rc = []
        for i, (r, c, x0, y0, w, h) in enumerate(table):
            cfg.rois[i].x0, cfg.rois[i].y0, cfg.rois[i].w, cfg.rois[i].h = x0, y0, w, h
            self.rois_rc.append((r, c))
        cfg.history_size, cfg.min_presence, cfg.change_threshold = history_size, min_presence, change_threshold
        cfg.chunk, cfg.lanes, cfg.keep_enhanced = chunk, lanes, 1 if keep_enhanced else 0
        cfg.z_threshold, cfg.initial_variance = z_threshold, initial_variance
        cfg.enhance_region = 1 if enhance_region else 0
        cfg.use_hough = int(use_hough)  # 2 = evaluate HoughCircles on every non-uniform square (inspection)
        cfg.hough = N.HoughParams(1.2, float(hough_param1), float(hough_param2), float(min_radius_ratio), float(max_radius_ratio))
        self.ctx.check(self.ctx.lib.cbv_pipeline_configure(self.h_, cfg))
        self.board_size = S_
        self.matrix = M
        self._cfg = cfg

    def frames_ptr(self):
        return self.ctx.lib.cbv_pipeline_frames_dev(self.h_)

    def upload(self, slot, frame):
        f = N.as_bgr(frame)
        assert f.shape[:2] == (self.h, self.w)
        self.ctx.check(self.ctx.lib.cbv_pipeline_upload(self.h_, slot, N.ptr(f), f.strides[0]))

    def host_ring(self):
        """Pinned host mirror of the frame ring as a numpy array [max_frames, h, w, 3]: the capture side writes
        frames here, `submit` copies them to the GPU asynchronously."""
        ptr = self.ctx.lib.cbv_pipeline_host_ring(self.h_)
        if not ptr:
            raise RuntimeError(self.ctx.lib.cbv_last_error(self.ctx.h).decode())
        fs = (self.w * self.h * 3 + 255) & ~255  # frames are 256-byte aligned in both rings
        buf = (C.c_uint8 * (fs * self.max_frames)).from_address(ptr)
        flat = np.frombuffer(buf, dtype=np.uint8)
        return np.lib.stride_tricks.as_strided(flat, shape=(self.max_frames, self.h, self.w, 3), strides=(fs, self.w * 3, 3, 1))

    def submit(self, slot0, count):
        """Enqueue host ring -> device ring for the slots; run() of those slots waits for the copy."""
        self.ctx.check(self.ctx.lib.cbv_pipeline_submit(self.h_, slot0, count))

    def wait_submitted(self):
        """Block until every submitted copy has left the host ring (runs stay in flight); the ring may be rewritten."""
        self.ctx.check(self.ctx.lib.cbv_pipeline_wait_submitted(self.h_))

    def synth(self, slot0, count, stream_id=0, frame0=0, scene="normal", frames_per_ply=32, points=None):
        """Fill slots with synthetic frames of stream `stream_id`, frame indices
        frame0.. (scripted game, one ply every `frames_per_ply` frames)."""
        pts = points if points is not None else S.scaled_corners(self.w, self.h)
        Hinv = np.ascontiguousarray(get_perspective_transform(pts, S.BOARD_UNIT_QUAD).reshape(9))
        seeds = np.array([S.frame_seed(stream_id, frame0 + i) for i in range(count)], dtype=np.uint64)
        boards = np.concatenate([S.board_array(S.position_for_frame(frame0 + i, frames_per_ply)) for i in range(count)])
        boards = np.ascontiguousarray(boards, dtype=np.uint8)
        sc = N.Scene.from_dict(S.SCENES[scene]) if isinstance(scene, str) else scene
        self.ctx.check(self.ctx.lib.cbv_pipeline_synth(self.h_, slot0, count, N.ptr(seeds), N.ptr(Hinv), N.ptr(boards), sc))

    def set_check_squares(self, slot0, sets, count=None):
        """`squares_to_check` of detect_all_pieces per frame: a list of {(file, rank)} sets for the slots slot0.. —
        those squares are evaluated afresh even when unchanged and cached.  `sets=None` clears `count` slots
        (default: all from slot0)."""
        if sets is None:
            n = self.max_frames - slot0 if count is None else count
            self.ctx.check(self.ctx.lib.cbv_pipeline_set_check_squares(self.h_, slot0, n, None))
            return
        roi_of = {(c, 7 - r): i for i, (r, c) in enumerate(self.rois_rc)}
        masks = np.zeros(len(sets), np.uint64)
        for k, st in enumerate(sets):
            m = 0
            for pos in (st or ()):
                if pos in roi_of:
                    m |= 1 << roi_of[pos]
            masks[k] = m
        self.ctx.check(self.ctx.lib.cbv_pipeline_set_check_squares(self.h_, slot0, len(sets), N.ptr(masks)))

    def update_references(self, slot, reset_noise=False):
        """PieceDetector.update_references with the squares of a processed slot (and NoiseHandler.reset() when
        `reset_noise`): what the session does right after it accepted a move (game_session.py:219-223)."""
        self.ctx.check(self.ctx.lib.cbv_pipeline_update_references(self.h_, slot, 1 if reset_noise else 0))

    def reset_state(self):
        self.ctx.check(self.ctx.lib.cbv_pipeline_reset_state(self.h_))

    def calibrate_changes(self, slot):
        """ChangeDetector.calibrate from an already processed slot: later runs also classify
        every square's change (LEVE / PARCIAL / TOTAL) against that background model."""
        self.ctx.check(self.ctx.lib.cbv_pipeline_calibrate(self.h_, slot))

    def hough(self, slot):
        """HoughCircles outcome of every square of a processed slot (index = roi)."""
        out = (N.HoughResult * N.MAX_SQUARES)()
        self.ctx.check(self.ctx.lib.cbv_pipeline_hough(self.h_, slot, out))
        return out

    def changes_detailed(self, result, slot):
        """The dict ChangeDetector.detect_changes_detailed returns (change_detector.py:105-167) for one frame."""
        st = self.square_stats(slot)
        out = {}
        for i, (r, c) in enumerate(self.rois_rc):
            if not (result.changed >> i) & 1:
                continue
            pct = (st[i].z_count / st[i].n) * 100
            inten = "TOTAL" if (result.total >> i) & 1 else ("PARCIAL" if (result.parcial >> i) & 1 else "LEVE")
            out[(c, 7 - r)] = {"z_score": float(st[i].z_max), "pct_changed": pct, "intensity": inten,
                               "is_circular": bool((result.circular >> i) & 1), "center_ratio": 1.0}
        return out

    def run(self, slot0, count):
        """Asynchronous on the context's stream."""
        self.ctx.check(self.ctx.lib.cbv_pipeline_run(self.h_, slot0, count))

    def results(self, slot0, count):
        out = (N.FrameResult * count)()
        self.ctx.check(self.ctx.lib.cbv_pipeline_results(self.h_, slot0, count, out))
        return out

    def noise_results(self, slot0, count):
        """NoiseHandler.process outputs of the frames, as (NoiseState, data) tuples (game_session.py:165)."""
        from .noise_handler import decode_device_result
        out = (N.NoiseResult * count)()
        self.ctx.check(self.ctx.lib.cbv_pipeline_noise_results(self.h_, slot0, count, out))
        idx2pos = [(c, 7 - r) for (r, c) in self.rois_rc]
        return [decode_device_result(r, idx2pos) for r in out]

    def download(self, which, slot):
        shape = (self.h, self.w, 3) if which in (0, 1) else (self.board_size, self.board_size, 3)
        out = np.empty(shape, np.uint8)
        self.ctx.check(self.ctx.lib.cbv_pipeline_download(self.h_, which, slot, N.ptr(out)))
        return out

    def square_stats(self, slot):
        out = (N.SqStats * len(self.rois_rc))()
        self.ctx.check(self.ctx.lib.cbv_pipeline_square_stats(self.h_, slot, out))
        return out

    def occupied(self, result, stable=True):
        return bits_to_positions(result.stable_occupied if stable else result.raw_occupied, self.rois_rc)
