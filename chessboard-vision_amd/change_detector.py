"""ChangeDetector on MI355X — drop-in for change_detector.py
(ChangeDetectorPython / ChangeDetectorCython).

The per-pixel Gaussian background model (mean/variance float32 planes) lives
on the GPU; `means` / `variances` are dict-like views that fetch a plane when
read.  Scalar attributes (z_threshold, initial_variance, alpha, blur_kernel)
are plain and read at call time, as calibrate_sensitivity.py:135-139 expects.
"""
from . import _native as N
from ._squares import MEAN, VAR, PlaneDict, SquareSet, plan_of
from .piece_detector import PieceDetector

_INTENSITY = (None, "LEVE", "PARCIAL", "TOTAL")


class ChangeDetectorHIP:
    def __init__(self):
        self.z_threshold = 2.5
        self.initial_variance = 100
        self.alpha = 0.1
        self.blur_kernel = 5
        self._kernel = 5

        self._state = SquareSet()
        self.means = PlaneDict(self._state, MEAN)
        self.variances = PlaneDict(self._state, VAR)
        self.is_calibrated = False
        self.focus_squares = set()

        self.piece_detector = PieceDetector()
        self._prm = N.ChangeParams()

    def _k(self):
        return int(self.blur_kernel) | 1

    def _load(self, squares):
        keys = self._state.keys if (self._state.keys and set(squares.keys()) <= set(self._state.keys)) else None
        if self._state.load(squares, self._k(), keys=keys):
            self.means.clear()
            self.variances.clear()

    def _preprocess(self, img):
        tmp = SquareSet(self._state.ctx)
        tmp.load({0: img}, self._k())
        return tmp.get(0, 0)

    def calibrate(self, squares):
        """change_detector.py:36-47"""
        self.means.clear()
        self.variances.clear()
        self._state.load(squares, self._k())
        self._state.calibrate(self.initial_variance, None)
        self.means._mark(self._state.keys)
        self.variances._mark(self._state.keys)
        self.is_calibrated = True

    def set_focus_squares(self, squares):
        self.focus_squares = set(squares)

    def clear_focus(self):
        self.focus_squares = set()

    def get_focus_count(self):
        return len(self.focus_squares) if self.focus_squares else 64

    def update_all_references(self, squares):
        """change_detector.py:67-92"""
        if not self.is_calibrated:
            self.calibrate(squares)
            return
        self._load(squares)
        todo = [p for p in squares if p in self.means and (not self.focus_squares or p in self.focus_squares)]
        if todo:
            self._state.ema(self.alpha, todo)

    def detect_changes(self, squares):
        """change_detector.py:94-103"""
        detailed = self.detect_changes_detailed(squares)
        return {pos: info["pct_changed"] for pos, info in detailed.items() if info["intensity"] in ["PARCIAL", "TOTAL"]}

    def detect_changes_detailed(self, squares):
        """change_detector.py:105-167.  For a split_board dict (all squares views of one image) the device half is one
        library call: upload of the board at call time, _preprocess, z-scores against the device-resident model, and
        the circular test of the squares that changed."""
        results = {}
        if not self.is_calibrated:
            return results
        plan = plan_of(squares)
        if plan is None or plan[1].keys != self._state.keys or plan[1].shapes != self._state.shapes:
            return self._detect_changes_detailed_views(squares)
        img, lay = plan
        self._state.adopt(lay)
        to_check = self.focus_squares if self.focus_squares else squares.keys()
        means = self.means
        prm = self._prm
        prm.z_threshold = float(self.z_threshold)
        prm.select = lay.mask(pos for pos in to_check if pos in means)
        pd = self.piece_detector
        prm.circle_threshold = float(pd.circle_threshold)
        pd._fill_hough(prm.hough)
        rows = self._state.detect_changes(img, lay, self._k(), prm)
        index = lay.index
        for pos in to_check:  # the reference's dict is in this order
            i = index.get(pos)
            if i is None:
                continue
            in_result, intensity, is_circular, z_max, z_count, n = rows[i]
            if in_result:
                results[pos] = {"z_score": z_max, "pct_changed": (z_count / n) * 100, "intensity": _INTENSITY[intensity],
                                "is_circular": bool(is_circular), "center_ratio": 1.0}
        return results

    def _detect_changes_detailed_views(self, squares):
        """The same for squares that are not views of one image, a subset of the calibrated squares, or new geometry."""
        results = {}
        self._load(squares)
        st = self._state.stats(use_model=True, z_threshold=self.z_threshold)
        to_check = self.focus_squares if self.focus_squares else squares.keys()
        hits = []
        for pos in to_check:
            if pos not in squares or pos not in self.means:
                continue
            s = st[self._state.index[pos]]
            pct_changed = (s.z_count / s.n) * 100
            if pct_changed < 5.0:
                continue
            intensity = "TOTAL" if pct_changed > 75 else ("PARCIAL" if pct_changed > 15 else "LEVE")
            hits.append((pos, float(s.z_max), pct_changed, intensity))
        circ = self.piece_detector._detect_many([squares[p] for p, _, _, _ in hits])
        for (pos, z, pct, intensity), pd in zip(hits, circ):
            results[pos] = {"z_score": z, "pct_changed": pct, "intensity": intensity, "is_circular": pd["has_piece"],
                            "center_ratio": 1.0}
        return results

    def classify_hand_pattern(self, detailed):
        """change_detector.py:169-201"""
        total_squares = len(detailed)
        total_intensity = sum(1 for v in detailed.values() if v["intensity"] == "TOTAL")
        if total_intensity >= 2 or total_squares >= 4 or total_squares > 2:
            return {"is_hand": True, "is_move": False, "move_candidates": set()}
        move_candidates = set(detailed.keys())
        return {"is_hand": False, "is_move": len(move_candidates) == 2, "move_candidates": move_candidates}


ChangeDetector = ChangeDetectorHIP
