// Host-side word-box / line geometry of the DiT box processor, as native code: these are O(N^2) pure-Python loops in the
// reference (N ~ 400 word boxes a page, ~100 ms) and would cap a rank at a few pages/s.  Pure functions, no device work.
//
//   mhip_merge_boxes         marie/utils/overlap.py:268-330 (+ find_overlap_horizontal :106-183, merge_bboxes_as_block :186-204)
//   mhip_line_merge          marie/boxes/line_processor.py:47-171
//   mhip_find_line_numbers   marie/boxes/line_processor.py:15-44 (+ find_overlap_vertical, overlap.py:42-103)
//   mhip_lines_from_bboxes   marie/boxes/dit/ulim_dit_box_processor.py:201-288
//
// Arithmetic follows what numpy 2 evaluates for the reference's expressions: float32 for the detector's xyxy boxes
// (np.float32 scalars stay float32 against Python literals), int64 / float64 for the integer line boxes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/marie_hip.h"

#pragma clang fp contract(off)

namespace {

struct BoxF { float x, y, w, h; };
struct BoxI { int64_t x, y, w, h; };

inline float round6(float v) { return rintf(v * 1e6f) / 1e6f; }   // round(np.float32, 6): multiply, rint, divide in fp32

inline int64_t floordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    return ((a % b != 0) && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// ---- line_merge --------------------------------------------------------------------------------------------------------
inline bool same(const BoxI& a, const BoxI& b) { return a.x == b.x && a.y == b.y && a.w == b.w && a.h == b.h; }

// find_overlap_vertical's membership test and clamped 1-D IoU
inline bool v_overlap(const BoxI& a, const BoxI& b, double* iou) {
    if (a.h <= 0 || b.h <= 0 || same(a, b)) return false;
    int64_t a1 = a.y + a.h, b1 = b.y + b.h;
    if (!(a.y < b1 && b.y < a1)) return false;
    int64_t inter = std::min(a1, b1) - std::max(a.y, b.y);
    double v = (double)inter / (double)(a.h + b.h - inter);
    *iou = std::max(std::min(v, 1.0), 0.0);
    return true;
}

std::vector<BoxI> line_merge_pass(std::vector<BoxI> b, double min_iou) {
    // equal-y boxes keep their input order (the reference leaves tie order to numpy's unstable default sort)
    std::stable_sort(b.begin(), b.end(), [](const BoxI& p, const BoxI& q) { return p.y < q.y; });
    const int n = (int)b.size();
    std::vector<int> count(n, 0);
    double iou;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (v_overlap(b[i], b[j], &iou)) ++count[i];
    std::vector<char> visited(n, 0);
    std::vector<BoxI> out;
    for (int i = 0; i < n; ++i) {
        if (visited[i]) continue;
        visited[i] = 1;
        int64_t mx = b[i].x, my = b[i].y, xr = b[i].x + b[i].w, mh = b[i].h;
        for (int j = 0; j < n; ++j) {
            if (!v_overlap(b[i], b[j], &iou)) continue;
            if (visited[j] || iou < min_iou) continue;
            if (count[j] != count[i]) continue;          // "the candidate sees as many overlaps as the anchor"
            visited[j] = 1;
            mx = std::min(mx, b[j].x);
            my = std::min(my, b[j].y);
            xr = std::max(xr, b[j].x + b[j].w);
            mh = std::max(mh, b[j].h);
        }
        out.push_back({mx, my, xr - mx, mh});
    }
    return out;
}

std::vector<BoxI> line_merge(std::vector<BoxI> b) {
    if (b.empty()) return b;
    static const double thr[7] = {0.8, 0.7, 0.6, 0.5, 0.4, 0.37, 0.35};
    int still = 0;
    for (int t = 0; t < 7; ++t) {
        size_t before = b.size();
        b = line_merge_pass(std::move(b), thr[t]);
        if (b.size() == before && ++still > 2) break;
    }
    const int n = (int)b.size();
    std::vector<char> drop(n, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            if (b[j].x > b[i].x && b[j].x + b[j].w < b[i].x + b[i].w && b[j].y > b[i].y && b[j].y + b[j].h < b[i].y + b[i].h)
                drop[j] = 1;
        }
    std::vector<BoxI> kept;
    for (int i = 0; i < n; ++i)
        if (!drop[i]) kept.push_back(b[i]);
    std::stable_sort(kept.begin(), kept.end(), [](const BoxI& p, const BoxI& q) { return p.y < q.y; });
    return kept;
}

int find_line_number(const std::vector<BoxI>& lines, const BoxI& box) {
    int number = -1, hits = 0, only = -1;
    double best = 0.0, iou;
    for (int j = 0; j < (int)lines.size(); ++j) {
        if (!v_overlap(box, lines[j], &iou)) continue;
        ++hits;
        only = j;
        if (iou > best) { best = iou; number = j + 1; }
    }
    if (hits == 1) number = only + 1;
    if (number == -1) {
        int64_t min_y = 100000;
        int64_t cy = box.y + floordiv(box.h, 2);
        for (int j = 0; j < (int)lines.size(); ++j) {
            int64_t dy = std::llabs(cy - (lines[j].y + lines[j].h));
            if (dy < min_y) { number = j + 1; min_y = dy; }
        }
    }
    return number;
}

// ---- lines_from_bboxes: the mask is a union of rectangles, so it is handled as row bands of x intervals -----------------
struct Iv { int l, r; };

void merge_sorted(std::vector<Iv>& v) {      // union of intervals; touching pixels ([.., 3] [4, ..]) are one run
    std::sort(v.begin(), v.end(), [](const Iv& a, const Iv& b) { return a.l < b.l; });
    size_t o = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        if (o && v[i].l <= v[o - 1].r + 1) v[o - 1].r = std::max(v[o - 1].r, v[i].r);
        else v[o++] = v[i];
    }
    v.resize(o);
}

int uf_find(std::vector<int>& p, int a) {
    while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; }
    return a;
}

std::vector<BoxI> line_fragments(const float* xyxy, int n, int H, int W) {
    struct Rect { int xa, xb, ya, yb; };
    std::vector<Rect> rects;
    std::vector<int> cuts;
    for (int i = 0; i < n; ++i) {
        int64_t x1 = (int32_t)xyxy[4 * i], y1 = (int32_t)xyxy[4 * i + 1], x2 = (int32_t)xyxy[4 * i + 2], y2 = (int32_t)xyxy[4 * i + 3];
        int64_t q = floordiv(y2 - y1, 8);
        int64_t h = floordiv(y2 - y1, 2) + q;
        int64_t ya = y1 + floordiv(h, 2) - q, yb = ya + h;
        int64_t xa = std::min(x1, x2), xb = std::max(x1, x2);
        if (yb < ya) std::swap(ya, yb);
        xa = std::max<int64_t>(xa, 0); ya = std::max<int64_t>(ya, 0);
        xb = std::min<int64_t>(xb, W - 1); yb = std::min<int64_t>(yb, H - 1);
        if (xa > xb || ya > yb) continue;
        rects.push_back({(int)xa, (int)xb, (int)ya, (int)yb});
        cuts.push_back((int)ya);
        cuts.push_back((int)yb + 1);
    }
    std::vector<BoxI> frags;
    if (rects.empty()) return frags;
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());

    const int stride = W / std::min(160, W);
    const int k = stride > 1 ? stride : W / 2;
    const int a = k / 2, b = k - 1 - a;              // element window [x - a, x + b]

    struct Node { int l, r, y0, y1; };               // one x run over rows [y0, y1)
    std::vector<Node> nodes;
    std::vector<int> band_first;                     // index of a band's first node; bands tile [cuts[0], cuts.back())
    std::vector<Iv> row;
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        const int y0 = cuts[c], y1 = cuts[c + 1];
        band_first.push_back((int)nodes.size());
        row.clear();
        for (const Rect& r : rects)
            if (r.ya <= y0 && r.yb >= y0) row.push_back({r.xa, r.xb});
        if (row.empty()) continue;
        merge_sorted(row);
        for (Iv& v : row) { v.l = std::max(0, v.l - b); v.r = std::min(W - 1, v.r + a); }   // erode white: grow runs
        merge_sorted(row);
        for (const Iv& v : row) {                                                           // dilate white: shrink runs
            int l = v.l == 0 ? 0 : v.l + a;
            int r = v.r == W - 1 ? W - 1 : v.r - b;
            if (l <= r) nodes.push_back({l, r, y0, y1});
        }
    }
    band_first.push_back((int)nodes.size());
    const int nn = (int)nodes.size();
    std::vector<int> parent(nn);
    std::iota(parent.begin(), parent.end(), 0);
    for (size_t c = 0; c + 2 < band_first.size(); ++c) {          // 4-connectivity: runs of adjacent rows sharing a column
        int i = band_first[c], ie = band_first[c + 1], j = band_first[c + 1], je = band_first[c + 2];
        while (i < ie && j < je) {
            if (nodes[i].l <= nodes[j].r && nodes[j].l <= nodes[i].r) {
                int ra = uf_find(parent, i), rb = uf_find(parent, j);
                if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
            }
            if (nodes[i].r < nodes[j].r) ++i; else ++j;
        }
    }
    // nodes are already in raster order of their first pixel (bands top-down, runs left-right) and every root is the
    // smallest node index of its component, so emitting roots in index order is OpenCV's label order.
    struct Stat { int l, r, t, b; };
    std::vector<Stat> st(nn, Stat{0, 0, 0, 0});
    std::vector<char> seen(nn, 0);
    for (int i = 0; i < nn; ++i) {
        int r = uf_find(parent, i);
        if (!seen[r]) { seen[r] = 1; st[r] = {nodes[i].l, nodes[i].r, nodes[i].y0, nodes[i].y1}; }
        else {
            st[r].l = std::min(st[r].l, nodes[i].l); st[r].r = std::max(st[r].r, nodes[i].r);
            st[r].t = std::min(st[r].t, nodes[i].y0); st[r].b = std::max(st[r].b, nodes[i].y1);
        }
    }
    for (int i = 0; i < nn; ++i) {
        if (parent[i] != i) continue;
        int w = st[i].r - st[i].l + 1, h = st[i].b - st[i].t;
        if (h < 2 || w < 4) continue;
        frags.push_back({st[i].l, st[i].t, w, h});
    }
    return frags;
}

}  // namespace

extern "C" {

int mhip_merge_boxes(const float* xyxy, int n, float* out_xyxy, int* n_out) {
    if (n < 0 || !n_out || (n && (!xyxy || !out_xyxy))) return MHIP_EINVAL;
    std::vector<BoxF> cur(n);
    for (int i = 0; i < n; ++i)
        cur[i] = {xyxy[4 * i], xyxy[4 * i + 1], xyxy[4 * i + 2] - xyxy[4 * i], xyxy[4 * i + 3] - xyxy[4 * i + 1]};
    size_t last = cur.size();
    for (int round = 0; round < 3; ++round) {
        const int m = (int)cur.size();
        std::vector<float> xr(m), cy(m), lo(m), hi(m);
        for (int i = 0; i < m; ++i) {
            xr[i] = cur[i].x + cur[i].w;
            cy[i] = cur[i].y + floorf(cur[i].h * 0.5f);        // y + h // 2
            lo[i] = cy[i] - cur[i].h * 0.5f;
            hi[i] = cy[i] + cur[i].h * 0.5f;
        }
        std::vector<char> visited(m, 0);
        std::vector<BoxF> next;
        for (int i = 0; i < m; ++i) {
            if (visited[i]) continue;
            visited[i] = 1;
            float mx = cur[i].x, my = cur[i].y, mr = xr[i], mb = cur[i].y + cur[i].h;
            for (int j = 0; j < m; ++j) {
                const BoxF &p = cur[i], &q = cur[j];
                if (p.x == q.x && p.y == q.y && p.w == q.w && p.h == q.h) continue;
                if (!(p.x < xr[j] && q.x < xr[i])) continue;
                if (cy[j] < lo[i] || cy[j] > hi[i]) continue;
                visited[j] = 1;
                mx = std::min(mx, q.x); my = std::min(my, q.y);
                mr = std::max(mr, xr[j]); mb = std::max(mb, q.y + q.h);
            }
            next.push_back({mx, my, mr - mx, mb - my});
        }
        if ((int)next.size() == m) break;
        for (BoxF& v : next) v = {round6(v.x), round6(v.y), round6(v.w), round6(v.h)};
        cur.swap(next);
        if (last == cur.size()) break;
        last = cur.size();
    }
    for (size_t i = 0; i < cur.size(); ++i) {
        out_xyxy[4 * i] = cur[i].x; out_xyxy[4 * i + 1] = cur[i].y;
        out_xyxy[4 * i + 2] = cur[i].x + cur[i].w; out_xyxy[4 * i + 3] = cur[i].y + cur[i].h;
    }
    *n_out = (int)cur.size();
    return MHIP_OK;
}

static std::vector<BoxI> to_boxes(const int32_t* xywh, int n) {
    std::vector<BoxI> b(n);
    for (int i = 0; i < n; ++i) b[i] = {xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3]};
    return b;
}

static int put_boxes(const std::vector<BoxI>& b, int32_t* out, int cap, int* n_out) {
    *n_out = (int)b.size();
    if ((int)b.size() > cap) return MHIP_ENOMEM;
    for (size_t i = 0; i < b.size(); ++i) {
        out[4 * i] = (int32_t)b[i].x; out[4 * i + 1] = (int32_t)b[i].y;
        out[4 * i + 2] = (int32_t)b[i].w; out[4 * i + 3] = (int32_t)b[i].h;
    }
    return MHIP_OK;
}

int mhip_line_merge(const int32_t* xywh, int n, int32_t* out_xywh, int* n_out) {
    if (n < 0 || !n_out || (n && (!xywh || !out_xywh))) return MHIP_EINVAL;
    return put_boxes(line_merge(to_boxes(xywh, n)), out_xywh, n, n_out);
}

int mhip_find_line_numbers(const int32_t* lines_xywh, int n_lines, const int32_t* boxes_xywh, int n, int32_t* out) {
    if (n < 0 || n_lines < 0 || (n && (!boxes_xywh || !out)) || (n_lines && !lines_xywh)) return MHIP_EINVAL;
    std::vector<BoxI> lines = to_boxes(lines_xywh, n_lines), boxes = to_boxes(boxes_xywh, n);
    for (int i = 0; i < n; ++i) out[i] = find_line_number(lines, boxes[i]);
    return MHIP_OK;
}

int mhip_lines_from_bboxes(const float* xyxy, int n, int height, int width, int32_t* out_xywh, int cap, int* n_out) {
    if (n < 0 || height <= 0 || width <= 0 || cap < 0 || !n_out || (n && !xyxy) || (cap && !out_xywh)) return MHIP_EINVAL;
    return put_boxes(line_merge(line_fragments(xyxy, n, height, width)), out_xywh, cap, n_out);
}

}  // extern "C"
