// Indicator matrix on the device: I[i, j] = area(shape_i n cell_j) / area(cell_j) for polygon rings against the
// cutout's regular grid - the second half of SURVEY 8 f-2 (atl_gis.cpp is the host clipper with the same
// contract; atlite/gis.py:104-145 is the shapely loop both replace).
//
// No clipping: for a ring with vertices P_k the area of (ring n box [xa,xb] x [ya,yb]) is the line integral
//     -sum_edges  integral over the part of the edge with xa <= x <= xb of (clamp(y, ya, yb) - ya) dx
// (for every x the signed crossings of the vertical line with the ring, clamped into the box, add up to the
// length of the ring's slice inside the box).  Every (cell, edge) pair is independent, so a thread owns a cell
// and walks the edges that overlap its grid COLUMN; the host buckets each shape's edges by column once
// (O(edges)), which keeps the work at rows x edges instead of cells x edges for finely digitised borders.
// Coordinates are taken relative to the cell's corner before they are multiplied, the per-edge pieces are exact
// trapezoids, the result is a sum of <= 3 terms per edge: agreement with the host clipper ~1e-13 of a cell.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "atl_internal.h"

using namespace atl;

namespace {

struct Edge {
    double x1, y1, x2, y2;  // directed; sign folded in by swapping the endpoints of negative rings / holes
};

// integral of (clamp(y, 0, h) - 0) dx along the edge restricted to 0 <= x <= w, coordinates relative to the cell
__host__ __device__ __forceinline__ double edge_term(double x1, double y1, double x2, double y2, double w, double h) {
    const double dx = x2 - x1;
    if (dx == 0.0) return 0.0;
    // parameter range of the edge inside the column
    double t0 = (0.0 - x1) / dx, t1 = (w - x1) / dx;
    if (t0 > t1) {
        const double t = t0;
        t0 = t1;
        t1 = t;
    }
    t0 = fmax(t0, 0.0);
    t1 = fmin(t1, 1.0);
    if (!(t1 > t0)) return 0.0;
    const double dy = y2 - y1;
    // split where the edge crosses y = 0 and y = h
    double ta = t0, tb = t1;
    double cuts[4] = {t0, t1, t1, t1};
    int n = 1;
    if (dy != 0.0) {
        double c0 = (0.0 - y1) / dy, c1 = (h - y1) / dy;
        if (c0 > c1) {
            const double t = c0;
            c0 = c1;
            c1 = t;
        }
        if (c0 > ta && c0 < tb) cuts[n++] = c0;
        if (c1 > ta && c1 < tb && c1 > cuts[n - 1]) cuts[n++] = c1;
    }
    cuts[n] = t1;
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const double u0 = cuts[i], u1 = cuts[i + 1];
        const double ya = y1 + u0 * dy, yb = y1 + u1 * dy;
        const double ym = 0.5 * (ya + yb);
        const double width = (u1 - u0) * dx;
        if (ym <= 0.0) continue;
        s += (ym >= h ? h : ym) * width;  // a piece lies entirely below 0, inside [0, h] or above h
    }
    return s;
}

// blockIdx.y = bucket (shape, column); threads over the rows of the shape's bounding box
__global__ __launch_bounds__(256) void k_indicator(const Edge *__restrict__ edges, const int64_t *__restrict__ bucket_ptr,
                                                  const int32_t *__restrict__ bucket_col, const int32_t *__restrict__ bucket_row0,
                                                  const int32_t *__restrict__ bucket_nrows, const int64_t *__restrict__ bucket_out,
                                                  double xlo, double ylo, double dx, double dy, double *__restrict__ out) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= bucket_nrows[b]) return;
    const double xa = xlo + double(bucket_col[b]) * dx, ya = ylo + double(bucket_row0[b] + r) * dy;
    double a = 0.0;
    for (int64_t e = bucket_ptr[b]; e < bucket_ptr[b + 1]; ++e) {
        const Edge g = edges[e];  // the same address for the whole block: one broadcast load
        a -= edge_term(g.x1 - xa, g.y1 - ya, g.x2 - xa, g.y2 - ya, dx, dy);
    }
    out[bucket_out[b] + r] = a;
}

double ring_area(const double *xy, int64_t n) {
    double a = 0.0;
    const double ox = xy[0], oy = xy[1];
    for (int64_t i = 0; i < n; ++i) {
        const int64_t j = (i + 1) % n;
        a += (xy[2 * i] - ox) * (xy[2 * j + 1] - oy) - (xy[2 * j] - ox) * (xy[2 * i + 1] - oy);
    }
    return 0.5 * a;
}

}  // namespace

// ctx == nullptr: the candidate cells are evaluated by a host loop over the same edge_term() (same source, host
// build) - lets the CPU test suite check the bucketing, the integrals and the compaction against the clipper
static int indicator_integral(atl_ctx *ctx, int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                              const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole, const double *h_xy,
                              int64_t X, int64_t Y, double x0, double dx, double y0, double dy,
                              int64_t **out_indptr, int32_t **out_indices, double **out_data) {
    ATL_REQUIRE(out_indptr && out_indices && out_data, "atl_indicator_polygons_device: NULL argument");
    *out_indptr = nullptr;
    *out_indices = nullptr;
    *out_data = nullptr;
    ATL_REQUIRE(n_shapes >= 0 && n_rings >= 0 && X > 0 && Y > 0, "atl_indicator_polygons_device: bad shape");
    ATL_REQUIRE(dx > 0 && dy > 0, "atl_indicator_polygons_device: grid spacing must be positive (ascending x, y)");
    ATL_REQUIRE(X * Y < (int64_t(1) << 31), "atl_indicator_polygons_device: grid too large");
    ATL_REQUIRE(n_shapes == 0 || (h_shape_ring_ptr && h_ring_ptr && h_xy), "atl_indicator_polygons_device: NULL input");
    ATL_REQUIRE(n_shapes == 0 || (offsets_ok(h_shape_ring_ptr, n_shapes) && h_shape_ring_ptr[n_shapes] <= n_rings &&
                                  offsets_ok(h_ring_ptr, n_rings)),
                "atl_indicator_polygons_device: shape / ring offsets must be non-negative and non-decreasing");
    const double xlo = x0 - 0.5 * dx, ylo = y0 - 0.5 * dy;
    // ---- host preparation: oriented edges, bucketed by (shape, grid column) ---------------------------------
    struct Box {
        int64_t i0, i1, j0, j1;  // inclusive cell ranges, empty if i1 < i0
    };
    std::vector<Box> box(size_t(n_shapes), Box{0, -1, 0, -1});
    std::vector<Edge> all;                     // edges of all shapes, shape by shape
    std::vector<int64_t> shape_edge0(size_t(n_shapes) + 1, 0);
    for (int64_t s = 0; s < n_shapes; ++s) {
        double bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
        bool any = false;
        // a NaN / inf vertex anywhere in the shape: no edges, an empty row (as the host clipper)
        const bool finite = shape_is_finite(s, h_shape_ring_ptr, h_ring_ptr, h_xy);
        for (int64_t r = h_shape_ring_ptr[s]; finite && r < h_shape_ring_ptr[s + 1]; ++r) {
            ATL_REQUIRE(r >= 0 && r < n_rings, "atl_indicator_polygons_device: ring index out of range");
            const double *p = h_xy + 2 * h_ring_ptr[r];
            int64_t n = h_ring_ptr[r + 1] - h_ring_ptr[r];
            if (n >= 2 && p[0] == p[2 * (n - 1)] && p[1] == p[2 * (n - 1) + 1]) --n;  // repeated first vertex
            if (n < 3) continue;
            const bool hole = h_ring_is_hole && h_ring_is_hole[r];
            const bool flip = (ring_area(p, n) >= 0.0) == hole;  // outer rings counter-clockwise, holes clockwise
            for (int64_t i = 0; i < n; ++i) {
                const int64_t j = (i + 1) % n;
                Edge e{p[2 * i], p[2 * i + 1], p[2 * j], p[2 * j + 1]};
                if (flip) {
                    std::swap(e.x1, e.x2);
                    std::swap(e.y1, e.y2);
                }
                all.push_back(e);
                const double lo = std::min(e.x1, e.x2), hi = std::max(e.x1, e.x2);
                const double ylo_e = std::min(e.y1, e.y2), yhi_e = std::max(e.y1, e.y2);
                if (!any) {
                    bx0 = lo;
                    bx1 = hi;
                    by0 = ylo_e;
                    by1 = yhi_e;
                    any = true;
                } else {
                    bx0 = std::min(bx0, lo);
                    bx1 = std::max(bx1, hi);
                    by0 = std::min(by0, ylo_e);
                    by1 = std::max(by1, yhi_e);
                }
            }
        }
        shape_edge0[size_t(s) + 1] = int64_t(all.size());
        if (!any || !(bx1 >= bx0) || !(by1 >= by0)) continue;  // no ring with >= 3 vertices, or a non-finite vertex: no entries
        Box &b = box[size_t(s)];
        b.i0 = clamped_floor((bx0 - xlo) / dx, 0, X);  // X / -1: a box beside the grid stays empty (i0 > i1)
        b.i1 = clamped_floor((bx1 - xlo) / dx, -1, X - 1);
        b.j0 = clamped_floor((by0 - ylo) / dy, 0, Y);
        b.j1 = clamped_floor((by1 - ylo) / dy, -1, Y - 1);
    }
    // buckets: one per (shape, column of its box); counting sort of edge references
    std::vector<int64_t> shape_bucket0(size_t(n_shapes) + 1, 0);
    for (int64_t s = 0; s < n_shapes; ++s) {
        const Box &b = box[size_t(s)];
        const int64_t ncol = (b.i1 >= b.i0 && b.j1 >= b.j0) ? b.i1 - b.i0 + 1 : 0;
        shape_bucket0[size_t(s) + 1] = shape_bucket0[size_t(s)] + ncol;
    }
    const int64_t n_buckets = shape_bucket0[size_t(n_shapes)];
    std::vector<int64_t> bucket_ptr(size_t(n_buckets) + 1, 0), bucket_out(size_t(n_buckets), 0);
    std::vector<int32_t> bucket_col(size_t(n_buckets), 0), bucket_row0(size_t(n_buckets), 0), bucket_nrows(size_t(n_buckets), 0);
    auto col_range = [&](const Edge &e, const Box &b, int64_t *c0, int64_t *c1) {
        const double lo = std::min(e.x1, e.x2), hi = std::max(e.x1, e.x2);
        *c0 = clamped_floor((lo - xlo) / dx, b.i0, b.i1 + 1);
        *c1 = clamped_floor((hi - xlo) / dx, b.i0 - 1, b.i1);
    };
    for (int64_t s = 0; s < n_shapes; ++s) {
        const Box &b = box[size_t(s)];
        if (shape_bucket0[size_t(s) + 1] == shape_bucket0[size_t(s)]) continue;
        for (int64_t e = shape_edge0[size_t(s)]; e < shape_edge0[size_t(s) + 1]; ++e) {
            if (all[size_t(e)].x1 == all[size_t(e)].x2) continue;  // vertical edges integrate to nothing
            int64_t c0, c1;
            col_range(all[size_t(e)], b, &c0, &c1);
            for (int64_t c = c0; c <= c1; ++c) bucket_ptr[size_t(shape_bucket0[size_t(s)] + (c - b.i0)) + 1]++;
        }
    }
    for (int64_t k = 0; k < n_buckets; ++k) bucket_ptr[size_t(k) + 1] += bucket_ptr[size_t(k)];
    std::vector<Edge> bedges;
    bedges.resize(size_t(bucket_ptr[size_t(n_buckets)]));
    {
        std::vector<int64_t> fill(bucket_ptr.begin(), bucket_ptr.end() - 1);
        for (int64_t s = 0; s < n_shapes; ++s) {
            const Box &b = box[size_t(s)];
            if (shape_bucket0[size_t(s) + 1] == shape_bucket0[size_t(s)]) continue;
            for (int64_t e = shape_edge0[size_t(s)]; e < shape_edge0[size_t(s) + 1]; ++e) {
                if (all[size_t(e)].x1 == all[size_t(e)].x2) continue;
                int64_t c0, c1;
                col_range(all[size_t(e)], b, &c0, &c1);
                for (int64_t c = c0; c <= c1; ++c) bedges[size_t(fill[size_t(shape_bucket0[size_t(s)] + (c - b.i0))]++)] = all[size_t(e)];
            }
        }
    }
    int64_t n_cand = 0;
    int32_t max_rows = 0;
    for (int64_t s = 0; s < n_shapes; ++s) {
        const Box &b = box[size_t(s)];
        for (int64_t k = shape_bucket0[size_t(s)]; k < shape_bucket0[size_t(s) + 1]; ++k) {
            bucket_col[size_t(k)] = int32_t(b.i0 + (k - shape_bucket0[size_t(s)]));
            bucket_row0[size_t(k)] = int32_t(b.j0);
            bucket_nrows[size_t(k)] = int32_t(b.j1 - b.j0 + 1);
            bucket_out[size_t(k)] = n_cand;
            n_cand += b.j1 - b.j0 + 1;
            max_rows = std::max(max_rows, bucket_nrows[size_t(k)]);
        }
    }
    // ---- device: one thread per candidate cell ------------------------------------------------------------------
    std::vector<double> cand(size_t(n_cand), 0.0);
    if (n_cand > 0 && !ctx) {
        for (int64_t b = 0; b < n_buckets; ++b)
            for (int32_t r = 0; r < bucket_nrows[size_t(b)]; ++r) {
                const double xa = xlo + double(bucket_col[size_t(b)]) * dx, ya = ylo + double(bucket_row0[size_t(b)] + r) * dy;
                double a = 0.0;
                for (int64_t e = bucket_ptr[size_t(b)]; e < bucket_ptr[size_t(b) + 1]; ++e) {
                    const Edge &g = bedges[size_t(e)];
                    a -= edge_term(g.x1 - xa, g.y1 - ya, g.x2 - xa, g.y2 - ya, dx, dy);
                }
                cand[size_t(bucket_out[size_t(b)] + r)] = a;
            }
    } else if (n_cand > 0) {
        ATL_HIP_TRY(hipSetDevice(ctx->device));
        Edge *d_edges = nullptr;
        int64_t *d_ptr = nullptr, *d_out_off = nullptr;
        int32_t *d_col = nullptr, *d_row0 = nullptr, *d_nrows = nullptr;
        double *d_cand = nullptr;
        auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
            hipError_t e = dev_malloc(d, std::max<size_t>(bytes, 8));
            if (e != hipSuccess) return e;
            return (bytes && h2d(ctx, ctx->stream, *d, h, bytes) != ATL_OK) ? hipErrorUnknown : hipSuccess;
        };
        hipError_t e = up((void **)&d_edges, bedges.data(), bedges.size() * sizeof(Edge));
        if (e == hipSuccess) e = up((void **)&d_ptr, bucket_ptr.data(), bucket_ptr.size() * sizeof(int64_t));
        if (e == hipSuccess) e = up((void **)&d_out_off, bucket_out.data(), bucket_out.size() * sizeof(int64_t));
        if (e == hipSuccess) e = up((void **)&d_col, bucket_col.data(), bucket_col.size() * sizeof(int32_t));
        if (e == hipSuccess) e = up((void **)&d_row0, bucket_row0.data(), bucket_row0.size() * sizeof(int32_t));
        if (e == hipSuccess) e = up((void **)&d_nrows, bucket_nrows.data(), bucket_nrows.size() * sizeof(int32_t));
        if (e == hipSuccess) e = dev_malloc((void **)&d_cand, size_t(n_cand) * sizeof(double));
        if (e == hipSuccess) {
            // grid.y is limited to 65535 buckets per launch
            for (int64_t k0 = 0; k0 < n_buckets && e == hipSuccess; k0 += 65535) {
                const unsigned ny = unsigned(std::min<int64_t>(65535, n_buckets - k0));
                const dim3 grid(unsigned((max_rows + 255) / 256), ny);
                hipLaunchKernelGGL(k_indicator, grid, dim3(256), 0, ctx->stream, d_edges, d_ptr + k0, d_col + k0, d_row0 + k0,
                                   d_nrows + k0, d_out_off + k0, xlo, ylo, dx, dy, d_cand);
                e = hipGetLastError();
            }
        }
        if (e == hipSuccess && d2h(ctx, ctx->stream, cand.data(), d_cand, size_t(n_cand) * sizeof(double)) != ATL_OK) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        for (void *p : {(void *)d_edges, (void *)d_ptr, (void *)d_out_off, (void *)d_col, (void *)d_row0, (void *)d_nrows, (void *)d_cand})
            if (p) (void)dev_free(p);
        if (e != hipSuccess) {
            set_error("atl_indicator_polygons_device: %s", hipGetErrorString(e));
            return ATL_E_HIP;
        }
    }
    // ---- host: compact to CSR (rows = shapes, columns ascending) ----------------------------------------------------
    const double cell_area = dx * dy, tol = 1e-12 * cell_area;  // line-integral residue of cells the shape does not reach
    std::vector<int64_t> indptr(size_t(n_shapes) + 1, 0);
    std::vector<int32_t> indices;
    std::vector<double> data;
    for (int64_t s = 0; s < n_shapes; ++s) {
        const Box &b = box[size_t(s)];
        const int64_t k0 = shape_bucket0[size_t(s)], ncol = shape_bucket0[size_t(s) + 1] - k0;
        if (ncol > 0) {
            for (int64_t j = b.j0; j <= b.j1; ++j)
                for (int64_t c = 0; c < ncol; ++c) {
                    const double a = cand[size_t(bucket_out[size_t(k0 + c)] + (j - b.j0))];
                    if (a > tol) {
                        indices.push_back(int32_t(j * X + b.i0 + c));
                        data.push_back(std::min(a / cell_area, 1.0));
                    }
                }
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
        set_error("atl_indicator_polygons_device: out of host memory");
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

extern "C" {

int atl_indicator_polygons_device(atl_ctx *ctx, int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                                  const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole, const double *h_xy,
                                  int64_t X, int64_t Y, double x0, double dx, double y0, double dy,
                                  int64_t **out_indptr, int32_t **out_indices, double **out_data) {
    ATL_REQUIRE(ctx, "atl_indicator_polygons_device: ctx is NULL");
    return indicator_integral(ctx, n_shapes, h_shape_ring_ptr, n_rings, h_ring_ptr, h_ring_is_hole, h_xy, X, Y, x0, dx, y0, dy,
                              out_indptr, out_indices, out_data);
}

int atl_indicator_polygons_integral_host(int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                                         const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole, const double *h_xy,
                                         int64_t X, int64_t Y, double x0, double dx, double y0, double dy,
                                         int64_t **out_indptr, int32_t **out_indices, double **out_data) {
    return indicator_integral(nullptr, n_shapes, h_shape_ring_ptr, n_rings, h_ring_ptr, h_ring_is_hole, h_xy, X, Y, x0, dx, y0, dy,
                              out_indptr, out_indices, out_data);
}

}  // extern "C"
