// The fast pv kernel family behind a tracker: pv(tracking="horizontal" | "tilted_horizontal" | "vertical" | "dual") with
// the Huld panel, the simple trigon model (Hay-Davies, irradiation() and the bofinger panel behind a tracker: atl_kernels_pvkt.hip), one orientation for the grid
// or one per cell, stored solar angles, with and without the night early-out.  Vectorised launches only: odd cell counts / row lengths and unaligned cubes take the
// general kernel (atl_kernels_pvx.hip), whose tracker is a run-time switch.  Same PvConvT
// template as atl_kernels_pv.hip; a translation unit of its own so that the kernel files compile in parallel.
// Reference arithmetic: atlite/pv/orientation.py:104-196 (closed forms: panel_geom in atl_conv_pv.h),
// atlite/pv/irradiation.py:76-145, 214-255; atlite/pv/solar_panel_model.py:22-41.
#include "atl_kernel_templates.h"

namespace atl {
// atl_kernels_pvx.hip: the general kernel, the fallback for launches that cannot be vectorised
int pvx_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
}  // namespace atl

namespace {

#include "atl_conv_pv.h"
#include "atl_pv_make.h"

// f(converter instance) for (tracker, scalar / per-cell orientation, night early-out)
template <class F>
int pvk_dispatch(const atl_pv_params *p, F &&f) {
    const bool pc = p->d_cell_slope != nullptr, skip = p->night_skip != 0;
    auto tracker = [&](auto trk) {
        constexpr int TR = decltype(trk)::value;
        if (skip) return pc ? f(PvConvT<false, true, true, kTailHuld, TR>()) : f(PvConvT<false, false, true, kTailHuld, TR>());
        return pc ? f(PvConvT<false, true, false, kTailHuld, TR>()) : f(PvConvT<false, false, false, kTailHuld, TR>());
    };
    // compile-time trackers: a run-time switch over the four geometries inside the unrolled batch costs the fused
    // kernel 192-288 B of scratch per lane (measured: 127-160 VGPRs and none with the tracker fixed)
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL: return tracker(std::integral_constant<int, ATL_TRACK_HORIZONTAL>());
        case ATL_TRACK_TILTED_HORIZONTAL: return tracker(std::integral_constant<int, ATL_TRACK_TILTED_HORIZONTAL>());
        case ATL_TRACK_VERTICAL: return tracker(std::integral_constant<int, ATL_TRACK_VERTICAL>());
        case ATL_TRACK_DUAL: return tracker(std::integral_constant<int, ATL_TRACK_DUAL>());
        default: break;
    }
    atl::set_error("atl_pv: tracking code %d has no tracker in the fast family", p->tracking);
    return ATL_E_INVALID;
}

}  // namespace

namespace atl {

int pvk_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out) {
    bool vec;
    const int rc = pvk_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert", in->X);
    });
    return rc == kNeedScalar ? pvx_convert(ctx, in, p, T, S, time_agg, d_out) : rc;
}

int pvk_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    const int rc = pvk_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // namespace atl
