"""PieceDetector on MI355X — drop-in for piece_detector.py.

The GPU produces, for every square, the preprocessed gray image (BGR2GRAY +
5x5 Gaussian on the square alone) and the integer sums the decision chain
needs; the chain itself (thresholds, temporal smoothing, reference refresh)
runs here with the reference's own arithmetic (float64 means, np.var).

cv2.HoughCircles (piece_detector.py:210-270) runs on the device as well
(k_hough.hip): the transform is restated from the published OpenCV 4.x
algorithm and is bit-identical to this repository's oracle, but no OpenCV
output was available to pin it against ("parity unpinned").
"""
import json
import os

import numpy as np

from . import _native as N
from ._squares import GRAY, REF, PlaneDict, SquareSet, plan_of

SETTINGS_FILE = "piece_detector_settings.json"


class PieceDetectorHIP:
    def __init__(self):
        self.min_radius_ratio = 0.20
        self.max_radius_ratio = 0.55
        self.edge_threshold = 50
        self.circle_threshold = 0.6

        self.history_size = 5
        self.min_presence = 0.6
        self.detection_history = {}

        self.load_settings()

        self._state = SquareSet()        # the board's squares: gray + reference planes
        self._scratch = SquareSet(self._state.ctx)  # single squares passed to detect_piece
        self.reference_squares = PlaneDict(self._state, REF)
        self.cached_results = {}
        self.change_threshold = 25
        self._prm = N.DetectParams()

    def load_settings(self):
        """Radii from piece_detector_settings.json in the cwd (piece_detector.py:52-68)."""
        if os.path.exists(SETTINGS_FILE):
            try:
                with open(SETTINGS_FILE, "r") as f:
                    params = json.load(f)
                if "min_radius" in params:
                    self.min_radius_ratio = params["min_radius"] / 100.0
                if "max_radius" in params:
                    self.max_radius_ratio = params["max_radius"] / 100.0
                print(f"[PieceDetector] Settings loaded from {SETTINGS_FILE}")
            except Exception as e:
                print(f"[PieceDetector] Error loading settings: {e}")

    # -- decision chain on device statistics ----------------------------------
    def _hough_kwargs(self):
        return dict(dp=1.2, param1=getattr(self, "hough_param1", 100), param2=getattr(self, "hough_param2", 25),
                    min_radius_ratio=self.min_radius_ratio, max_radius_ratio=self.max_radius_ratio)

    def _detect_circle_unified(self, gray):
        """(found, center, radius, type) like piece_detector.py:210-270, for one preprocessed gray square."""
        self._scratch.load({0: gray}, 5)          # geometry; the plane is replaced by `gray` as given
        self._scratch.set(GRAY, 0, gray)
        hg = self._scratch.hough(**self._hough_kwargs())[0]
        if not hg.found:
            return False, None, None, None
        return True, (int(hg.cx), int(hg.cy)), int(hg.r), ("tower_top" if hg.kind == 2 else "hough")

    def _decide(self, st, shape, hg=None):
        """piece_detector.py:289-345 for one square given its cbv_sq_stats and cbv_hough_result."""
        h, w = shape
        result = {"has_piece": False, "confidence": 0.0, "center": None, "radius": None, "method": None,
                  "center_border_diff": 0, "is_ellipse": False, "axes": None}
        n, s, ss = int(st.n), int(st.sum), int(st.sumsq)
        # np.std(gray) < 15  <=>  n*sumsq - sum^2 < 225 n^2
        if n * ss - s * s < 225 * n * n:
            return result
        if hg is not None and hg.flags & N.HOUGH_OVERFLOW:
            # more accumulator maxima than the kernel keeps: a truncated list could change has_piece, so it is
            # never passed on as HoughCircles' answer
            raise RuntimeError("HoughCircles candidate list overflowed (more than 512 accumulator maxima in a %dx%d square)" % (w, h))
        if hg is not None and hg.found:
            kind = "tower_top" if hg.kind == 2 else "hough"
            result.update(has_piece=True, center=(int(hg.cx), int(hg.cy)), radius=int(hg.r), method=kind,
                          confidence=0.9 if kind == "hough" else 0.75)
            return result
        center_mean = np.float64(st.center_sum) / st.center_cnt if st.center_cnt else np.float64("nan")
        border_mean = np.float64(st.border_sum) / st.border_cnt if st.border_cnt else np.float64("nan")
        diff = abs(center_mean - border_mean)
        result["center_border_diff"] = diff
        if diff > 40:
            result.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="center_diff",
                          confidence=min(1.0, diff / 80))
            return result
        ring_means = [np.float64(st.ring_sum[k]) / st.ring_cnt[k] for k in range(4) if st.ring_cnt[k] > 0]
        symmetry = 0.0 if len(ring_means) < 2 else min(1.0, np.var(ring_means) / 500)
        if symmetry > self.circle_threshold:
            result.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="symmetry", confidence=symmetry)
        return result

    def _gray_stats(self, gray):
        """cbv_sq_stats (mask sums on the device) of one already preprocessed gray square."""
        self._scratch.load({0: gray}, 5)          # geometry; the plane is replaced by `gray` as given
        self._scratch.set(GRAY, 0, gray)
        return self._scratch.stats()[0]

    def _detect_center_vs_border(self, gray):
        """(diff, center_mean, border_mean), piece_detector.py:177-207: means over the centre disc and the four corners."""
        st = self._gray_stats(gray)
        center_mean = np.float64(st.center_sum) / st.center_cnt if st.center_cnt else np.float64("nan")
        border_mean = np.float64(st.border_sum) / st.border_cnt if st.border_cnt else np.float64("nan")
        return abs(center_mean - border_mean), center_mean, border_mean

    def _analyze_radial_symmetry(self, gray):
        """0..1 score, piece_detector.py:141-175: variance of the mean intensity over four concentric rings / 500."""
        st = self._gray_stats(gray)
        ring_means = [np.float64(st.ring_sum[k]) / st.ring_cnt[k] for k in range(4) if st.ring_cnt[k] > 0]
        if len(ring_means) < 2:
            return 0.0
        return min(1.0, np.var(ring_means) / 500)

    def _preprocess_square(self, square_img):
        self._scratch.load({0: square_img}, 5)
        return self._scratch.get(0, 0)

    def _detect_many(self, imgs):
        """detect_piece for a list of independent square images, one device round trip."""
        if not imgs:
            return []
        out = []
        for i0 in range(0, len(imgs), N.MAX_SQUARES):
            part = {i: im for i, im in enumerate(imgs[i0:i0 + N.MAX_SQUARES])}
            self._scratch.load(part, 5)
            st = self._scratch.stats()
            hg = self._scratch.hough(**self._hough_kwargs())
            out += [self._decide(st[i], self._scratch.shapes[i], hg[i]) for i in range(len(part))]
        return out

    def detect_piece(self, square_img, pos=None):
        return self._detect_many([square_img])[0]

    # -- reference handling ----------------------------------------------------
    def calibrate_reference(self, squares_dict):
        """piece_detector.py:70-80"""
        self.reference_squares.clear()
        self.cached_results.clear()
        self._state.load(squares_dict, 5)
        self._state.set_ref(None)
        self.reference_squares._mark(self._state.keys)
        st = self._state.stats()
        hg = self._state.hough(**self._hough_kwargs())
        for i, pos in enumerate(self._state.keys):
            self.cached_results[pos] = self._decide(st[i], self._state.shapes[i], hg[i])

    def update_references(self, squares_dict):
        """piece_detector.py:447-453"""
        self._load_state(squares_dict)
        lay = self._state._layout
        if lay is not None:
            self._state.set_ref_mask(lay.mask(squares_dict.keys()))  # asynchronous, the set rides in the launch
        else:
            self._state.set_ref(list(squares_dict.keys()))
        self.reference_squares._mark(squares_dict.keys())
        self.cached_results.clear()

    def _fill_hough(self, hp):
        k = self._hough_kwargs()
        hp.dp, hp.param1, hp.param2 = float(k["dp"]), float(k["param1"]), float(k["param2"])
        hp.min_radius_ratio, hp.max_radius_ratio = float(k["min_radius_ratio"]), float(k["max_radius_ratio"])

    def _load_state(self, squares_dict):
        keys = self._state.keys if (self._state.keys and set(squares_dict.keys()) <= set(self._state.keys)) else None
        if self._state.load(squares_dict, 5, keys=keys):
            self.reference_squares.clear()  # geometry changed: device planes were reset

    def _update_history(self, pos, has_piece):
        history = self.detection_history.setdefault(pos, [])
        history.append(has_piece)
        if len(history) > self.history_size:
            history.pop(0)

    def _get_stable_detection(self, pos):
        if pos not in self.detection_history:
            return False
        history = self.detection_history[pos]
        if len(history) < 3:
            return history[-1] if history else False
        return sum(history) / len(history) >= self.min_presence

    def detect_all_pieces(self, squares_dict, use_smoothing=True, use_delta=True, squares_to_check=None):
        """piece_detector.py:348-440.  Returns (results, visual_changes).

        When the squares are views of one image (split_board's dict) the device half is ONE library call
        (cbv_squares_detect_all: upload of the board at call time, preprocess, |gray - reference|, the should_process
        gate, HoughCircles on the squares the reference would run it on, detect_piece) and the loop below only does
        what the reference's class keeps on the host: cache, history, smoothing, which references to refresh."""
        results, visual_changes = {}, set()
        if not squares_dict:
            return results, visual_changes
        plan = plan_of(squares_dict)
        if plan is None or (self._state.keys and plan[1].keys != self._state.keys and set(plan[1].keys) < set(self._state.keys)):
            return self._detect_all_pieces_views(squares_dict, use_smoothing, use_delta, squares_to_check)
        img, lay = plan
        state, refs, cached, hist_all = self._state, self.reference_squares, self.cached_results, self.detection_history
        if state.adopt(lay):
            refs.clear()  # geometry changed: the device planes are void
        prm = self._prm
        prm.change_threshold = float(self.change_threshold)
        prm.circle_threshold = float(self.circle_threshold)
        self._fill_hough(prm.hough)
        prm.has_ref = lay.mask(refs._valid)
        prm.cached = lay.mask(cached)
        prm.check_given = 0 if squares_to_check is None else 1
        prm.check = 0 if squares_to_check is None else lay.mask(squares_to_check)
        prm.use_delta = 1 if use_delta else 0
        rows = state.detect_all(img, lay, prm)
        hs, mp, names = self.history_size, self.min_presence, N.METHOD_NAMES
        refresh, refreshed = 0, []
        for pos, bit, (has_piece, method, changed, should, evaluated, cx, cy, radius, conf, diff) in zip(lay.keys, lay.bits, rows):
            if changed:
                visual_changes.add(pos)
            if evaluated:
                if has_piece:
                    raw = {"has_piece": True, "confidence": conf, "center": (cx, cy), "radius": radius, "method": names[method],
                           "center_border_diff": diff, "is_ellipse": False, "axes": None}
                else:
                    raw = {"has_piece": False, "confidence": conf, "center": None, "radius": None, "method": None,
                           "center_border_diff": diff, "is_ellipse": False, "axes": None}
                cached[pos] = raw.copy()
            else:
                raw = cached[pos].copy()
            raw_has = raw["has_piece"]
            hist = hist_all.get(pos)
            if hist is None:
                hist = hist_all[pos] = []
            hist.append(raw_has)
            if len(hist) > hs:
                hist.pop(0)
            if use_smoothing:
                nh = len(hist)
                stable = hist[-1] if nh < 3 else sum(hist) / nh >= mp
                raw["has_piece"] = stable
                if should and raw_has == stable:
                    refresh |= bit
                    refreshed.append(pos)
            elif should:
                refresh |= bit
                refreshed.append(pos)
            results[pos] = raw
        if refresh:
            state.set_ref_mask(refresh)
            refs._mark(refreshed)
        return results, visual_changes

    def _detect_all_pieces_views(self, squares_dict, use_smoothing=True, use_delta=True, squares_to_check=None):
        """The same for squares that are NOT views of one image (or a subset of the known squares): packed view
        copies and separate statistics / HoughCircles calls."""
        results, visual_changes, refresh = {}, set(), []
        if not squares_dict:
            return results, visual_changes
        self._load_state(squares_dict)
        st = self._state.stats(use_ref=bool(self.reference_squares))
        hg = self._state.hough(**self._hough_kwargs())
        for pos in squares_dict:
            i = self._state.index[pos]
            s = st[i]
            shape = self._state.shapes[i]
            has_changed_visual = pos not in self.reference_squares or (np.float64(s.sad_ref) / s.n) > self.change_threshold
            if has_changed_visual:
                visual_changes.add(pos)
            should_process = squares_to_check is not None and pos in squares_to_check
            if not should_process and (squares_to_check is None or use_delta):
                if pos not in self.cached_results or has_changed_visual:
                    should_process = True
            if should_process or pos not in self.cached_results:
                raw_result = self._decide(s, shape, hg[i])
                self.cached_results[pos] = raw_result.copy()
            else:
                raw_result = self.cached_results[pos].copy()
            raw_has_piece = raw_result["has_piece"]
            self._update_history(pos, raw_has_piece)
            is_stable_update = True
            if use_smoothing:
                stable = self._get_stable_detection(pos)
                raw_result["has_piece"] = stable
                if raw_has_piece != stable:
                    is_stable_update = False
            if should_process and is_stable_update:
                refresh.append(pos)
            results[pos] = raw_result
        if refresh:
            self._state.set_ref(refresh)
            self.reference_squares._mark(refresh)
        return results, visual_changes

    def get_occupied_squares(self, squares_dict, use_smoothing=True):
        results, _ = self.detect_all_pieces(squares_dict, use_smoothing)
        return {pos for pos, info in results.items() if info["has_piece"]}


PieceDetector = PieceDetectorHIP
