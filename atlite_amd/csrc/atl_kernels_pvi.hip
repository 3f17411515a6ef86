// The fast pv kernel family on influx / outflux datasets (PvConvT<..., HEAD = 1 / 2>): cutouts that store the total
// downward radiation and the reflected outflux instead of a direct / diffuse split and an albedo.  The converter's head
// splits the influx with Reindl's clearsky model - "simple", or "enhanced" with the air temperature and the relative
// humidity (atlite/pv/irradiation.py:13-73) - and takes albedo = outflux / influx (:128-139); the family's usual tail
// follows: the Huld panel after the simple or the Hay-Davies ("other", :76-145) trigon model, fixed panel (one orientation
// for the grid or one per cell), stored solar angles, with and without the night early-out: 48 B (simple) / 56 B
// (enhanced) per cell-step.  Until round 4 Hay-Davies and the enhanced model on such datasets took the general kernel
// (PvxConvT: 112 VGPRs + scratch at 1.7 resident waves, 0.41 of the HBM peak).  Vectorised launches only: what cannot
// be vectorised takes the general kernel (atl_kernels_pvx.hip).  A translation unit of its own so that the kernel files
// compile in parallel.
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

// f(converter instance) for (clearsky model, trigon model, scalar / per-cell orientation, early-out)
template <class F>
int pvi_dispatch(const atl_pv_params *p, F &&f) {
    const bool pc = p->d_cell_slope != nullptr, hd = p->trigon_model == ATL_TRIGON_OTHER, skip = p->night_skip != 0;
    auto with = [&](auto head) {
        constexpr int HD = decltype(head)::value;
        if (hd) {
            if (skip) return pc ? f(PvConvT<false, true, true, kTailHuldHayDavies, ATL_TRACK_NONE, HD>()) : f(PvConvT<false, false, true, kTailHuldHayDavies, ATL_TRACK_NONE, HD>());
            return pc ? f(PvConvT<false, true, false, kTailHuldHayDavies, ATL_TRACK_NONE, HD>()) : f(PvConvT<false, false, false, kTailHuldHayDavies, ATL_TRACK_NONE, HD>());
        }
        if (skip) return pc ? f(PvConvT<false, true, true, kTailHuld, ATL_TRACK_NONE, HD>()) : f(PvConvT<false, false, true, kTailHuld, ATL_TRACK_NONE, HD>());
        return pc ? f(PvConvT<false, true, false, kTailHuld, ATL_TRACK_NONE, HD>()) : f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_NONE, HD>());
    };
    return p->clearsky_model == ATL_CLEARSKY_ENHANCED ? with(std::integral_constant<int, 2>()) : with(std::integral_constant<int, 1>());
}

}  // namespace

namespace atl {

int pvi_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out) {
    bool vec;
    const int rc = pvi_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert", in->X);
    });
    return rc == kNeedScalar ? pvx_convert(ctx, in, p, T, S, time_agg, d_out) : rc;
}

int pvi_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    const int rc = pvi_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // namespace atl
