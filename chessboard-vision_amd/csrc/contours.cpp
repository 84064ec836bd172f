// Host part of board_detection.find_chessboard_corners (board_detection.py:4-58): the pixel stages
// (gray, 7x7 Gaussian sigma 1, Canny 30/100, three 5x5 dilations) run on the GPU (k_canny.hip); what follows —
// cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE), contourArea, arcLength, approxPolyDP, the "largest
// four-cornered contour" rule and reorder — is sequential border following and belongs on the host.
// Restated from the published algorithms (Suzuki-Abe border following as OpenCV implements it, Douglas-Peucker
// with OpenCV's start-point search and clean-up pass).  PARITY UNPINNED: no OpenCV output is available.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cbv_internal.h"

namespace {

struct Pt {
    int x, y;
};

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// External borders of the top-level components of a 0/255 image, every border pixel in following order.
void find_external_contours(const uint8_t* bin, int w, int h, std::vector<std::vector<Pt>>& out)
{
    const int pw = w + 2, ph = h + 2;
    std::vector<int8_t> img((size_t)pw * ph, 0); // one zero pixel around the image
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * pw + x + 1] = bin[(size_t)y * w + x] ? 1 : 0;
    int off[8];
    for (int k = 0; k < 8; k++) off[k] = DY[k] * pw + DX[k];
    const int8_t NBD = 2, NBD_RIGHT = (int8_t)(NBD | -128);
    for (int y = 1; y <= h; y++) {
        int lnbd = y * pw; // last border pixel met in this row (the padding pixel to start with)
        int prev = 0;
        for (int x = 1; x <= w; x++) {
            const int idx = y * pw + x;
            const int p = img[idx];
            if (p == 1 && prev == 0 && !(img[lnbd] > 0)) {
                // an outer border that does not lie inside another one: follow it
                std::vector<Pt> c;
                const int i0 = idx;
                int s = 4, s_end = 4, i1 = i0;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + off[s];
                } while (img[i1] == 0 && s != s_end);
                if (s == s_end) {
                    img[i0] = NBD_RIGHT;
                    c.push_back({x - 1, y - 1});
                } else {
                    int i3 = i0;
                    Pt pt = {x - 1, y - 1};
                    for (;;) {
                        s_end = s;
                        int i4;
                        for (;;) {
                            i4 = i3 + off[++s & 7];
                            if (img[i4] != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) img[i3] = NBD_RIGHT; // the pixel to the right was examined and is 0
                        else if (img[i3] == 1) img[i3] = NBD;
                        c.push_back(pt);
                        pt.x += DX[s];
                        pt.y += DY[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                out.push_back(std::move(c));
            }
            const int q = img[idx];
            if (q != 0 && q != 1) lnbd = idx;
            prev = q;
        }
    }
}

double contour_area(const std::vector<Pt>& c)
{
    if (c.empty()) return 0;
    double a = 0;
    Pt prev = c.back();
    for (const Pt& p : c) {
        a += (double)prev.x * p.y - (double)prev.y * p.x;
        prev = p;
    }
    return std::fabs(a * 0.5);
}

double arc_length_closed(const std::vector<Pt>& c)
{
    if (c.size() < 2) return 0;
    double per = 0;
    Pt prev = c.back();
    for (const Pt& p : c) {
        const float dx = (float)p.x - (float)prev.x, dy = (float)p.y - (float)prev.y;
        per += (double)std::sqrt(dx * dx + dy * dy); // float square root, double sum
        prev = p;
    }
    return per;
}

// Douglas-Peucker on a closed integer contour, OpenCV's scheme: three passes of "farthest point from the current
// start" pick the initial split, a slice is kept whole when its farthest point lies within eps of the chord, and a
// last pass drops points that ended up on an almost straight line.
void approx_poly_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst)
{
    dst.clear();
    const int count = (int)src.size();
    if (count == 0) return;
    struct Range {
        int start, end;
    };
    std::vector<Range> stack;
    eps *= eps;
    auto next = [&](int& pos) {
        const Pt p = src[pos];
        if (++pos >= count) pos = 0;
        return p;
    };
    Range slice = {0, 0}, right = {0, 0};
    int pos = 0;
    bool le_eps = false;
    Pt start_pt = {-1000000, -1000000};
    for (int it = 0; it < 3; it++) {
        double max_dist = 0;
        pos = (pos + right.start) % count;
        start_pt = next(pos);
        for (int j = 1; j < count; j++) {
            const Pt pt = next(pos);
            const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right.start = j;
            }
        }
        le_eps = max_dist <= eps;
    }
    if (!le_eps) {
        right.end = slice.start = pos % count;
        slice.end = right.start = (right.start + slice.start) % count;
        stack.push_back(right);
        stack.push_back(slice);
    } else dst.push_back(start_pt);
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        const Pt end_pt = src[slice.end];
        pos = slice.start;
        start_pt = next(pos);
        if (pos != slice.end) {
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            double max_dist = 0;
            while (pos != slice.end) {
                const Pt pt = next(pos);
                const double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right.start = (pos + count - 1) % count;
                }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src[slice.start];
        }
        if (le_eps) dst.push_back(start_pt);
        else {
            right.end = slice.end;
            slice.end = right.start;
            stack.push_back(right);
            stack.push_back(slice);
        }
    }
    // clean-up: remove points on [almost] straight lines
    int cnt = (int)dst.size(), new_count = cnt;
    if (cnt < 3) return;
    auto dnext = [&](int& p) {
        const Pt v = dst[p];
        if (++p >= cnt) p = 0;
        return v;
    };
    pos = cnt - 1;
    start_pt = dnext(pos);
    int wpos = pos;
    Pt pt = dnext(pos);
    for (int i = 0; i < cnt && new_count > 2; i++) {
        const Pt end_pt = dnext(pos);
        const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        const double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        const double inner = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            pt = dnext(pos);
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    dst.resize(new_count);
}

} // namespace

// From the dilated edge image to the four corners.  Returns 1 and pts = the approximated polygon's vertices (x, y) when a
// contour qualifies (area > 100000, approxPolyDP(0.02 * perimeter) has four vertices; the largest wins), else 0.
int board_corners_from_edges(const uint8_t* dilated, int w, int h, int32_t pts[8], int* n_contours)
{
    std::vector<std::vector<Pt>> cs;
    find_external_contours(dilated, w, h, cs);
    if (n_contours) *n_contours = (int)cs.size();
    // cv2.findContours lists contours from the last found to the first; sorted(..., reverse=True) is stable
    int best = -1;
    double best_area = 0;
    std::vector<Pt> best_poly, poly;
    for (int i = (int)cs.size() - 1; i >= 0; i--) {
        const double area = contour_area(cs[i]);
        if (!(area > 100000)) continue;
        approx_poly_closed(cs[i], 0.02 * arc_length_closed(cs[i]), poly);
        if (poly.size() != 4) continue;
        if (best < 0 || area > best_area) {
            best = i;
            best_area = area;
            best_poly = poly;
        }
    }
    if (best < 0) return 0;
    for (int k = 0; k < 4; k++) { // approxPolyDP's own vertex order; the caller applies reorder()
        pts[2 * k] = best_poly[k].x;
        pts[2 * k + 1] = best_poly[k].y;
    }
    return 1;
}

// inspection: the approxPolyDP polygon (eps = eps_frac * perimeter) of the external contour with the largest area
int largest_contour_polygon(const uint8_t* img, int w, int h, double eps_frac, int32_t* pts, int cap, double* area_out, int* contour_len)
{
    std::vector<std::vector<Pt>> cs;
    find_external_contours(img, w, h, cs);
    int best = -1;
    double best_area = -1;
    for (int i = (int)cs.size() - 1; i >= 0; i--) {
        const double a = contour_area(cs[i]);
        if (a > best_area) {
            best_area = a;
            best = i;
        }
    }
    if (best < 0) return 0;
    std::vector<Pt> poly;
    approx_poly_closed(cs[best], eps_frac * arc_length_closed(cs[best]), poly);
    if (area_out) *area_out = best_area;
    if (contour_len) *contour_len = (int)cs[best].size();
    for (int k = 0; k < (int)poly.size() && k < cap; k++) {
        pts[2 * k] = poly[k].x;
        pts[2 * k + 1] = poly[k].y;
    }
    return (int)poly.size();
}
