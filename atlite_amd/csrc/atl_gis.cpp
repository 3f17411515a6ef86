// Host-side indicator matrix: I[i, j] = area(shape_i ∩ cell_j) / area(cell_j) for polygon
// rings against the cutout's regular grid.  Replaces the per-pair shapely loop of
// compute_indicatormatrix (atlite/gis.py:104-145) for cutout grids, whose cells are the boxes
// centre ± (dx/2, dy/2) (atlite/cutout.py:369-376).  Pure C++ (no GPU): Sutherland-Hodgman
// clipping of each ring to a row strip, then to each column of the strip; signed shoelace
// areas, holes subtract.  Cells that a shape merely touches (zero area) get no entry, as in
// the reference where the LIL assignment of 0.0 stores nothing.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "atl_internal.h"

namespace {

struct Pt {
    double x, y;
};

// keep the part of the ring with (axis coordinate) >= bound (sign=+1) or <= bound (sign=-1)
void clip_halfplane(const std::vector<Pt> &in, std::vector<Pt> &out, int axis, double bound, int sign) {
    out.clear();
    const size_t n = in.size();
    if (n == 0) return;
    auto coord = [axis](const Pt &p) { return axis == 0 ? p.x : p.y; };
    auto inside = [&](const Pt &p) { return sign > 0 ? coord(p) >= bound : coord(p) <= bound; };
    Pt prev = in[n - 1];
    bool prev_in = inside(prev);
    for (size_t i = 0; i < n; ++i) {
        const Pt cur = in[i];
        const bool cur_in = inside(cur);
        if (cur_in != prev_in) {
            const double t = (bound - coord(prev)) / (coord(cur) - coord(prev));
            Pt q;
            if (axis == 0) {
                q.x = bound;
                q.y = prev.y + t * (cur.y - prev.y);
            } else {
                q.y = bound;
                q.x = prev.x + t * (cur.x - prev.x);
            }
            out.push_back(q);
        }
        if (cur_in) out.push_back(cur);
        prev = cur;
        prev_in = cur_in;
    }
}

double shoelace(const std::vector<Pt> &r) {
    const size_t n = r.size();
    if (n < 3) return 0.0;
    double a = 0.0;
    // relative to the first vertex: keeps the cancellation small for far-from-origin rings
    const double ox = r[0].x, oy = r[0].y;
    for (size_t i = 0; i < n; ++i) {
        const Pt &p = r[i], &q = r[(i + 1) % n];
        a += (p.x - ox) * (q.y - oy) - (q.x - ox) * (p.y - oy);
    }
    return 0.5 * a;
}

// keep the part of the ring on the left of the directed line a -> b (a convex, counter-clockwise clip polygon's inside)
void clip_left_of(const std::vector<Pt> &in, std::vector<Pt> &out, const Pt &a, const Pt &b) {
    out.clear();
    const size_t n = in.size();
    if (n == 0) return;
    const double ex = b.x - a.x, ey = b.y - a.y;
    auto side = [&](const Pt &p) { return ex * (p.y - a.y) - ey * (p.x - a.x); };
    Pt prev = in[n - 1];
    double sp = side(prev);
    for (size_t i = 0; i < n; ++i) {
        const Pt cur = in[i];
        const double sc = side(cur);
        if ((sc >= 0) != (sp >= 0)) {
            const double t = sp / (sp - sc);
            out.push_back({prev.x + t * (cur.x - prev.x), prev.y + t * (cur.y - prev.y)});
        }
        if (sc >= 0) out.push_back(cur);
        prev = cur;
        sp = sc;
    }
}

}  // namespace

extern "C" {

int atl_indicator_polygons(int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                           const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole, const double *h_xy,
                           int64_t X, int64_t Y, double x0, double dx, double y0, double dy,
                           int64_t **out_indptr, int32_t **out_indices, double **out_data) {
    ATL_REQUIRE(out_indptr && out_indices && out_data, "atl_indicator_polygons: NULL output");
    *out_indptr = nullptr;
    *out_indices = nullptr;
    *out_data = nullptr;
    ATL_REQUIRE(n_shapes >= 0 && n_rings >= 0 && X > 0 && Y > 0, "atl_indicator_polygons: bad shape");
    ATL_REQUIRE(dx > 0 && dy > 0, "atl_indicator_polygons: grid spacing must be positive (ascending x, y)");
    ATL_REQUIRE(X * Y < (int64_t(1) << 31), "atl_indicator_polygons: grid too large");
    ATL_REQUIRE(n_shapes == 0 || (h_shape_ring_ptr && h_ring_ptr && h_xy),
                "atl_indicator_polygons: NULL input");
    ATL_REQUIRE(n_shapes == 0 || (atl::offsets_ok(h_shape_ring_ptr, n_shapes) && h_shape_ring_ptr[n_shapes] <= n_rings &&
                                  atl::offsets_ok(h_ring_ptr, n_rings)),
                "atl_indicator_polygons: shape / ring offsets must be non-negative and non-decreasing");
    const double cell_area = dx * dy;
    const double xlo = x0 - 0.5 * dx, ylo = y0 - 0.5 * dy;  // lower-left corner of cell (0,0)
    std::vector<int64_t> indptr(size_t(n_shapes) + 1, 0);
    std::vector<int32_t> indices;
    std::vector<double> data;
    std::vector<std::pair<int32_t, double>> acc;
    std::vector<Pt> ring, strip, tmp, cellp;
    for (int64_t s = 0; s < n_shapes; ++s) {
        acc.clear();
        const bool finite = atl::shape_is_finite(s, h_shape_ring_ptr, h_ring_ptr, h_xy);  // else: an empty row
        for (int64_t r = h_shape_ring_ptr[s]; finite && r < h_shape_ring_ptr[s + 1]; ++r) {
            ATL_REQUIRE(r >= 0 && r < n_rings, "atl_indicator_polygons: ring index out of range");
            const int64_t v0 = h_ring_ptr[r], v1 = h_ring_ptr[r + 1];
            ring.clear();
            for (int64_t v = v0; v < v1; ++v) ring.push_back({h_xy[2 * v], h_xy[2 * v + 1]});
            if (ring.size() >= 2 && ring.front().x == ring.back().x && ring.front().y == ring.back().y)
                ring.pop_back();  // closed ring given with repeated first vertex
            if (ring.size() < 3) continue;
            const double sign_ring = shoelace(ring) >= 0 ? 1.0 : -1.0;
            const double sign = (h_ring_is_hole && h_ring_is_hole[r]) ? -1.0 : 1.0;
            double bx0 = ring[0].x, bx1 = bx0, by0 = ring[0].y, by1 = by0;
            for (const Pt &p : ring) {
                bx0 = std::min(bx0, p.x);
                bx1 = std::max(bx1, p.x);
                by0 = std::min(by0, p.y);
                by1 = std::max(by1, p.y);
            }
            const int64_t j0 = atl::clamped_floor((by0 - ylo) / dy, 0, Y);
            const int64_t j1 = atl::clamped_floor((by1 - ylo) / dy, -1, Y - 1);
            for (int64_t j = j0; j <= j1; ++j) {
                const double ya = ylo + j * dy, yb = ylo + (j + 1) * dy;
                clip_halfplane(ring, tmp, 1, ya, +1);
                clip_halfplane(tmp, strip, 1, yb, -1);
                if (strip.size() < 3) continue;
                double sx0 = strip[0].x, sx1 = sx0;
                for (const Pt &p : strip) {
                    sx0 = std::min(sx0, p.x);
                    sx1 = std::max(sx1, p.x);
                }
                const int64_t i0 = atl::clamped_floor((sx0 - xlo) / dx, 0, X);
                const int64_t i1 = atl::clamped_floor((sx1 - xlo) / dx, -1, X - 1);
                for (int64_t i = i0; i <= i1; ++i) {
                    const double xa = xlo + i * dx, xb = xlo + (i + 1) * dx;
                    clip_halfplane(strip, tmp, 0, xa, +1);
                    clip_halfplane(tmp, cellp, 0, xb, -1);
                    const double a = shoelace(cellp) * sign_ring;  // >= 0 up to rounding
                    if (a > 0.0) acc.push_back({int32_t(j * X + i), sign * a});
                }
            }
        }
        std::sort(acc.begin(), acc.end(),
                  [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) {
                      return a.first < b.first;
                  });
        for (size_t k = 0; k < acc.size();) {
            size_t e = k;
            double a = 0.0;
            while (e < acc.size() && acc[e].first == acc[k].first) a += acc[e++].second;
            if (a > 0.0) {
                indices.push_back(acc[k].first);
                data.push_back(std::min(a / cell_area, 1.0));
            }
            k = e;
        }
        indptr[size_t(s) + 1] = int64_t(indices.size());
    }
    const size_t nnz = indices.size();
    int64_t *pi = static_cast<int64_t *>(malloc(indptr.size() * sizeof(int64_t)));
    int32_t *pj = static_cast<int32_t *>(malloc(std::max<size_t>(nnz, 1) * sizeof(int32_t)));
    double *pd = static_cast<double *>(malloc(std::max<size_t>(nnz, 1) * sizeof(double)));
    if (!pi || !pj || !pd) {
        free(pi);
        free(pj);
        free(pd);
        atl::set_error("atl_indicator_polygons: out of host memory");
        return ATL_E_NOMEM;
    }
    memcpy(pi, indptr.data(), indptr.size() * sizeof(int64_t));
    if (nnz) {
        memcpy(pj, indices.data(), nnz * sizeof(int32_t));
        memcpy(pd, data.data(), nnz * sizeof(double));
    }
    *out_indptr = pi;
    *out_indices = pj;
    *out_data = pd;
    return ATL_OK;
}

int atl_host_free(void *p) {
    free(p);
    return ATL_OK;
}

}  // extern "C"
