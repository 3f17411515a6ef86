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

// The same matrix for cells that are arbitrary CONVEX QUADRILATERALS - a cutout's grid cells after their four corners
// went through a map projection into the shapes' coordinate system, which is how the reference intersects shapes given in
// another crs (atlite/gis.py:128-133: `orig = reproject_shapes(orig, orig_crs, dest_crs)` moves the vertices of the cell
// boxes, the areas are taken in the destination plane).  h_quads: n_cells x 4 x (x, y), any winding.  Candidates come from
// a uniform bucket grid over the cells' centroids; a ring is clipped to a bucket's (padded) rectangle once, then to each of
// the bucket's cells.  I[i, j] = area(shape_i ∩ quad_j) / area(quad_j).
int atl_indicator_polygons_quads(int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                                 const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole, const double *h_xy,
                                 int64_t n_cells, const double *h_quads, int64_t **out_indptr, int32_t **out_indices,
                                 double **out_data) {
    ATL_REQUIRE(out_indptr && out_indices && out_data, "atl_indicator_polygons_quads: NULL output");
    *out_indptr = nullptr;
    *out_indices = nullptr;
    *out_data = nullptr;
    ATL_REQUIRE(n_shapes >= 0 && n_rings >= 0 && n_cells > 0 && h_quads, "atl_indicator_polygons_quads: bad argument");
    ATL_REQUIRE(n_cells < (int64_t(1) << 31), "atl_indicator_polygons_quads: too many cells");
    ATL_REQUIRE(n_shapes == 0 || (h_shape_ring_ptr && h_ring_ptr && h_xy), "atl_indicator_polygons_quads: NULL input");
    ATL_REQUIRE(n_shapes == 0 || (atl::offsets_ok(h_shape_ring_ptr, n_shapes) && h_shape_ring_ptr[n_shapes] <= n_rings &&
                                  atl::offsets_ok(h_ring_ptr, n_rings)),
                "atl_indicator_polygons_quads: shape / ring offsets must be non-negative and non-decreasing");
    struct Quad {
        Pt v[4];
        double area, x0, x1, y0, y1;
    };
    std::vector<Quad> quads(static_cast<size_t>(n_cells));
    double gx0 = HUGE_VAL, gx1 = -HUGE_VAL, gy0 = HUGE_VAL, gy1 = -HUGE_VAL, ext_x = 0.0, ext_y = 0.0, sum_x = 0.0, sum_y = 0.0;
    int64_t n_ok = 0;
    for (int64_t c = 0; c < n_cells; ++c) {
        Quad &q = quads[size_t(c)];
        bool fin = true;
        for (int k = 0; k < 4; ++k) {
            q.v[k] = {h_quads[8 * c + 2 * k], h_quads[8 * c + 2 * k + 1]};
            fin = fin && std::isfinite(q.v[k].x) && std::isfinite(q.v[k].y);
        }
        std::vector<Pt> r(q.v, q.v + 4);
        double a = fin ? shoelace(r) : 0.0;
        if (a < 0) {  // clockwise: reverse
            std::swap(q.v[1], q.v[3]);
            a = -a;
        }
        q.area = a;  // 0: a degenerate or non-finite cell takes no part
        if (!(a > 0)) continue;
        q.x0 = q.x1 = q.v[0].x;
        q.y0 = q.y1 = q.v[0].y;
        for (int k = 1; k < 4; ++k) {
            q.x0 = std::min(q.x0, q.v[k].x);
            q.x1 = std::max(q.x1, q.v[k].x);
            q.y0 = std::min(q.y0, q.v[k].y);
            q.y1 = std::max(q.y1, q.v[k].y);
        }
        // convexity: every vertex on the left of every edge (a projected cell is; anything else is refused)
        for (int k = 0; k < 4; ++k) {
            const Pt &p = q.v[k], &n1 = q.v[(k + 1) % 4], &n2 = q.v[(k + 2) % 4];
            ATL_REQUIRE((n1.x - p.x) * (n2.y - p.y) - (n1.y - p.y) * (n2.x - p.x) >= 0,
                        "atl_indicator_polygons_quads: cell %lld is not a convex quadrilateral", (long long)c);
        }
        gx0 = std::min(gx0, q.x0);
        gx1 = std::max(gx1, q.x1);
        gy0 = std::min(gy0, q.y0);
        gy1 = std::max(gy1, q.y1);
        ext_x = std::max(ext_x, q.x1 - q.x0);
        ext_y = std::max(ext_y, q.y1 - q.y0);
        sum_x += q.x1 - q.x0;
        sum_y += q.y1 - q.y0;
        ++n_ok;
    }
    // buckets of ~8 x 8 mean cell extents over the centroids' range
    const double bsx = n_ok ? std::max(8.0 * sum_x / double(n_ok), (gx1 - gx0) / 2048.0) : 1.0;
    const double bsy = n_ok ? std::max(8.0 * sum_y / double(n_ok), (gy1 - gy0) / 2048.0) : 1.0;
    const int64_t nbx = n_ok ? std::max<int64_t>(1, int64_t(std::ceil((gx1 - gx0) / bsx))) : 1;
    const int64_t nby = n_ok ? std::max<int64_t>(1, int64_t(std::ceil((gy1 - gy0) / bsy))) : 1;
    auto bucket_of = [&](double x, double y) {
        const int64_t bx = atl::clamped_floor((x - gx0) / bsx, 0, nbx - 1), by = atl::clamped_floor((y - gy0) / bsy, 0, nby - 1);
        return by * nbx + bx;
    };
    std::vector<int64_t> bstart(size_t(nbx * nby) + 1, 0);
    std::vector<int32_t> bcell(static_cast<size_t>(n_ok));
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int64_t> fill(bstart.begin(), bstart.end() - 1);
        for (int64_t c = 0; c < n_cells; ++c) {
            const Quad &q = quads[size_t(c)];
            if (!(q.area > 0)) continue;
            const int64_t b = bucket_of(0.25 * (q.v[0].x + q.v[1].x + q.v[2].x + q.v[3].x), 0.25 * (q.v[0].y + q.v[1].y + q.v[2].y + q.v[3].y));
            if (pass == 0)
                ++bstart[size_t(b) + 1];
            else
                bcell[size_t(fill[size_t(b)]++)] = int32_t(c);
        }
        if (pass == 0)
            for (size_t b = 1; b < bstart.size(); ++b) bstart[b] += bstart[b - 1];
    }
    std::vector<int64_t> indptr(size_t(n_shapes) + 1, 0);
    std::vector<int32_t> indices;
    std::vector<double> data;
    std::vector<std::pair<int32_t, double>> acc;
    std::vector<Pt> ring, local, t1, t2;
    for (int64_t s = 0; s < n_shapes; ++s) {
        acc.clear();
        const bool finite = n_ok && atl::shape_is_finite(s, h_shape_ring_ptr, h_ring_ptr, h_xy);  // else: an empty row
        for (int64_t r = h_shape_ring_ptr[s]; finite && r < h_shape_ring_ptr[s + 1]; ++r) {
            ATL_REQUIRE(r >= 0 && r < n_rings, "atl_indicator_polygons_quads: ring index out of range");
            ring.clear();
            for (int64_t v = h_ring_ptr[r]; v < h_ring_ptr[r + 1]; ++v) ring.push_back({h_xy[2 * v], h_xy[2 * v + 1]});
            if (ring.size() >= 2 && ring.front().x == ring.back().x && ring.front().y == ring.back().y) ring.pop_back();
            if (ring.size() < 3) continue;
            const double sign_ring = shoelace(ring) >= 0 ? 1.0 : -1.0;
            const double sign = (h_ring_is_hole && h_ring_is_hole[r]) ? -1.0 : 1.0;
            double rx0 = ring[0].x, rx1 = rx0, ry0 = ring[0].y, ry1 = ry0;
            for (const Pt &p : ring) {
                rx0 = std::min(rx0, p.x);
                rx1 = std::max(rx1, p.x);
                ry0 = std::min(ry0, p.y);
                ry1 = std::max(ry1, p.y);
            }
            if (rx1 < gx0 || rx0 > gx1 || ry1 < gy0 || ry0 > gy1) continue;
            // buckets whose cells (centroid inside the bucket, so the cell within the bucket padded by its extent) can touch the ring
            const int64_t bx0 = atl::clamped_floor((rx0 - ext_x - gx0) / bsx, 0, nbx - 1), bx1 = atl::clamped_floor((rx1 + ext_x - gx0) / bsx, 0, nbx - 1);
            const int64_t by0 = atl::clamped_floor((ry0 - ext_y - gy0) / bsy, 0, nby - 1), by1 = atl::clamped_floor((ry1 + ext_y - gy0) / bsy, 0, nby - 1);
            for (int64_t by = by0; by <= by1; ++by) {
                for (int64_t bx = bx0; bx <= bx1; ++bx) {
                    const int64_t b = by * nbx + bx;
                    if (bstart[size_t(b)] == bstart[size_t(b) + 1]) continue;
                    const double xa = gx0 + double(bx) * bsx - ext_x, xb = gx0 + double(bx + 1) * bsx + ext_x;
                    const double ya = gy0 + double(by) * bsy - ext_y, yb = gy0 + double(by + 1) * bsy + ext_y;
                    clip_halfplane(ring, t1, 0, xa, +1);
                    clip_halfplane(t1, t2, 0, xb, -1);
                    clip_halfplane(t2, t1, 1, ya, +1);
                    clip_halfplane(t1, local, 1, yb, -1);
                    if (local.size() < 3) continue;
                    double lx0 = local[0].x, lx1 = lx0, ly0 = local[0].y, ly1 = ly0;
                    for (const Pt &p : local) {
                        lx0 = std::min(lx0, p.x);
                        lx1 = std::max(lx1, p.x);
                        ly0 = std::min(ly0, p.y);
                        ly1 = std::max(ly1, p.y);
                    }
                    for (int64_t k = bstart[size_t(b)]; k < bstart[size_t(b) + 1]; ++k) {
                        const Quad &q = quads[size_t(bcell[size_t(k)])];
                        if (q.x1 < lx0 || q.x0 > lx1 || q.y1 < ly0 || q.y0 > ly1) continue;
                        clip_left_of(local, t1, q.v[0], q.v[1]);
                        clip_left_of(t1, t2, q.v[1], q.v[2]);
                        clip_left_of(t2, t1, q.v[2], q.v[3]);
                        clip_left_of(t1, t2, q.v[3], q.v[0]);
                        const double a = shoelace(t2) * sign_ring;
                        if (a > 0.0) acc.push_back({bcell[size_t(k)], sign * a / q.area});
                    }
                }
            }
        }
        std::sort(acc.begin(), acc.end(),
                  [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
        for (size_t k = 0; k < acc.size();) {
            size_t e = k;
            double a = 0.0;
            while (e < acc.size() && acc[e].first == acc[k].first) a += acc[e++].second;
            if (a > 0.0) {
                indices.push_back(acc[k].first);
                data.push_back(std::min(a, 1.0));
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
        atl::set_error("atl_indicator_polygons_quads: out of host memory");
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
