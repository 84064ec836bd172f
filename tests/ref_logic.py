"""Oracle-side restatement of the reference's per-square host logic, used only
by tests: PieceDetector.detect_piece without HoughCircles
(piece_detector.py:272-345), detect_all_pieces (piece_detector.py:348-440) and
ChangeDetector (change_detector.py:36-167) on top of the C oracle's pixel
functions."""
import numpy as np

from oracle import cbv_oracle as O


def detect_circle_unified(gray, min_radius_ratio=0.20, max_radius_ratio=0.55, param1=100, param2=25):
    """piece_detector.py:210-270 on the oracle's HoughCircles restatement (numpy float32 arithmetic for the pick,
    as `cx`, `cy` are float32 there)."""
    h, w = gray.shape
    min_dim = min(h, w)
    circles = O.hough_circles(gray, 1.2, min_dim // 3, param1, param2, int(min_dim * min_radius_ratio), int(min_dim * max_radius_ratio))
    best, best_dist = None, float("inf")
    max_offset = min_dim * 0.3
    for (cx, cy, r, _) in circles:
        cx, cy = np.float32(cx), np.float32(cy)
        dist = np.sqrt((cx - (w // 2)) ** 2 + (cy - (h // 2)) ** 2)
        if dist < max_offset and dist < best_dist:
            best_dist, best = dist, (cx, cy, np.float32(r))
    if best is None:
        return False, None, None, None, circles
    r = int(best[2])
    kind = "tower_top" if r < min_dim * 0.20 else "hough"
    return True, (int(best[0]), int(best[1])), r, kind, circles


def detect_piece(square_img, circle_threshold=0.6, hough=None):
    """`hough`: None = without the HoughCircles step; else kwargs for detect_circle_unified."""
    gray = O.square_preprocess(square_img, 5)
    h, w = gray.shape
    st = O.square_stats(gray)
    res = {"has_piece": False, "confidence": 0.0, "center": None, "radius": None, "method": None,
           "center_border_diff": 0, "is_ellipse": False, "axes": None}
    std = np.std(gray)
    if std < 15:
        return res, gray
    if hough is not None:
        found, center, radius, kind, _ = detect_circle_unified(gray, **hough)
        if found:
            res.update(has_piece=True, center=center, radius=radius, method=kind, confidence=0.9 if kind == "hough" else 0.75)
            return res, gray
    cm = np.float64(st.center_sum) / st.center_cnt
    bm = np.float64(st.border_sum) / st.border_cnt
    diff = abs(cm - bm)
    res["center_border_diff"] = diff
    if diff > 40:
        res.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="center_diff", confidence=min(1.0, diff / 80))
        return res, gray
    rm = [np.float64(st.ring_sum[k]) / st.ring_cnt[k] for k in range(4) if st.ring_cnt[k] > 0]
    sym = 0.0 if len(rm) < 2 else min(1.0, np.var(rm) / 500)
    if sym > circle_threshold:
        res.update(has_piece=True, center=(w // 2, h // 2), radius=min(h, w) // 3, method="symmetry", confidence=sym)
    return res, gray


class RefPieceDetector:
    def __init__(self, hough=None):
        self.hough = hough
        self.history_size, self.min_presence, self.change_threshold = 5, 0.6, 25
        self.detection_history, self.reference_squares, self.cached_results = {}, {}, {}

    def _stable(self, pos):
        hst = self.detection_history.get(pos)
        if hst is None:
            return False
        if len(hst) < 3:
            return hst[-1] if hst else False
        return sum(hst) / len(hst) >= self.min_presence

    def update_references(self, squares):
        for pos, img in squares.items():
            self.reference_squares[pos] = O.square_preprocess(img, 5)
        self.cached_results.clear()

    def calibrate_reference(self, squares):
        self.reference_squares.clear()
        self.cached_results.clear()
        for pos, img in squares.items():
            res, gray = detect_piece(img, hough=self.hough)
            self.reference_squares[pos] = gray
            self.cached_results[pos] = res

    def get_occupied_squares(self, squares, use_smoothing=True):
        results, _ = self.detect_all_pieces(squares, use_smoothing)
        return {pos for pos, info in results.items() if info["has_piece"]}

    def detect_all_pieces(self, squares, use_smoothing=True, use_delta=True, squares_to_check=None):
        results, visual = {}, set()
        self.last_processed = set()
        for pos, img in squares.items():
            raw, gray = detect_piece(img, hough=self.hough)
            changed = True
            if pos in self.reference_squares:
                diff = np.abs(gray.astype(np.int16) - self.reference_squares[pos].astype(np.int16))
                changed = np.mean(diff) > self.change_threshold
            if changed:
                visual.add(pos)
            should = squares_to_check is not None and pos in squares_to_check
            if not should and (squares_to_check is None or use_delta):
                if pos not in self.cached_results or changed:
                    should = True
            if should:
                self.last_processed.add(pos)
            if should or pos not in self.cached_results:
                self.cached_results[pos] = raw.copy()
                raw_result = raw
            else:
                raw_result = self.cached_results[pos].copy()
            raw_has = raw_result["has_piece"]
            hst = self.detection_history.setdefault(pos, [])
            hst.append(raw_has)
            if len(hst) > self.history_size:
                hst.pop(0)
            stable_update = True
            if use_smoothing:
                stable = self._stable(pos)
                raw_result = dict(raw_result, has_piece=stable)
                stable_update = raw_has == stable
            if should and stable_update:
                self.reference_squares[pos] = gray.copy()
            results[pos] = raw_result
        return results, visual


class RefChangeDetector:
    def __init__(self, hough=None):
        self.hough = hough
        self.z_threshold, self.initial_variance, self.alpha, self.blur_kernel = 2.5, 100, 0.1, 5
        self.means, self.variances, self.is_calibrated, self.focus_squares = {}, {}, False, set()

    def _pre(self, img):
        return O.square_preprocess(img, self.blur_kernel | 1)

    def calibrate(self, squares):
        self.means, self.variances = {}, {}
        for pos, img in squares.items():
            g = self._pre(img)
            self.means[pos] = g.astype(np.float32)
            self.variances[pos] = np.full(g.shape, self.initial_variance, dtype=np.float32)
        self.is_calibrated = True

    def set_focus_squares(self, squares):
        self.focus_squares = set(squares)

    def clear_focus(self):
        self.focus_squares = set()

    def get_focus_count(self):
        return len(self.focus_squares) if self.focus_squares else 64

    def detect_changes(self, squares):
        return {p: v["pct_changed"] for p, v in self.detect_changes_detailed(squares).items() if v["intensity"] in ("PARCIAL", "TOTAL")}

    def classify_hand_pattern(self, detailed):
        n = len(detailed)
        if sum(1 for v in detailed.values() if v["intensity"] == "TOTAL") >= 2 or n >= 4 or n > 2:
            return {"is_hand": True, "is_move": False, "move_candidates": set()}
        return {"is_hand": False, "is_move": n == 2, "move_candidates": set(detailed.keys())}

    def update_all_references(self, squares):
        if not self.is_calibrated:
            return self.calibrate(squares)
        for pos, img in squares.items():
            if self.focus_squares and pos not in self.focus_squares:
                continue
            O.ema_update(self._pre(img), self.alpha, self.means[pos], self.variances[pos])

    def detect_changes_detailed(self, squares):
        out = {}
        if not self.is_calibrated:
            return out
        for pos in (self.focus_squares if self.focus_squares else squares.keys()):
            if pos not in squares or pos not in self.means:
                continue
            g = self._pre(squares[pos])
            st = O.square_stats(g, mean=self.means[pos], var=self.variances[pos], z_thresh=self.z_threshold)
            pct = (st.z_count / st.n) * 100
            if pct < 5.0:
                continue
            inten = "TOTAL" if pct > 75 else ("PARCIAL" if pct > 15 else "LEVE")
            out[pos] = {"z_score": float(st.z_max), "pct_changed": pct, "intensity": inten,
                        "is_circular": detect_piece(squares[pos], hough=self.hough)[0]["has_piece"], "center_ratio": 1.0}
        return out
