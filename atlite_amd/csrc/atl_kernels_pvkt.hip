// The fast pv kernel family behind a tracker with a tail other than the default: pv(tracking=..., trigon_model="other")
// (Huld panel after Hay-Davies), irradiation(tracking=...) (the plain tilted irradiation, simple trigon model) and
// pv(panel="KANENA", tracking=...) (bofinger panel, simple trigon model) - FUSED (convert + aggregate) kernels only, one
// orientation for the grid, stored solar angles, with and without the night early-out: 24 kernels.  Until round 6 these
// combinations ran in the general kernel (atl_kernels_pvx.hip) at 0.40-0.58 of the HBM peak; per-cell results and launches
// that cannot be vectorised still do (per-cell orientations: atl_kernels_pvkc.hip; Hay-Davies before the other tails: atl_kernels_pvka.hip).
// Reference arithmetic: atlite/pv/orientation.py:104-196 (closed forms: panel_geom in atl_conv_pv.h),
// atlite/pv/irradiation.py:76-145, 214-255; atlite/pv/solar_panel_model.py:22-74; atlite/convert.py:748-767.
#include "atl_kernel_templates.h"

namespace atl {
int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
}  // namespace atl

namespace {

#include "atl_conv_pv.h"
#include "atl_pv_make.h"

// f(converter instance) for (tracker, tail, night early-out)
template <class F>
int pvkt_dispatch(const atl_pv_params *p, F &&f) {
    const bool skip = p->night_skip != 0;
    auto tail = [&](auto trk, auto tl) {
        constexpr int TR = decltype(trk)::value, TL = decltype(tl)::value;
        return skip ? f(PvConvT<false, false, true, TL, TR>()) : f(PvConvT<false, false, false, TL, TR>());
    };
    auto tracker = [&](auto trk) {
        if (p->panel_model == ATL_PANEL_HULD) return tail(trk, std::integral_constant<int, kTailHuldHayDavies>());
        if (p->panel_model == ATL_PANEL_NONE) return tail(trk, std::integral_constant<int, kTailIrradiation>());
        return tail(trk, std::integral_constant<int, kTailBofinger>());
    };
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

// which calls this unit takes (atl_pv_convert_aggregate asks before it falls back to the general kernel)
bool pvkt_takes(const atl_pv_inputs *in, const atl_pv_params *p) {
    if (!(p->tracking >= ATL_TRACK_HORIZONTAL && p->tracking <= ATL_TRACK_DUAL)) return false;
    if (in->d_influx || !in->d_influx_direct || !in->d_influx_diffuse || !in->d_influx_toa || !in->d_albedo || !in->d_temperature ||
        !in->d_solar_altitude || !in->d_solar_azimuth || p->d_cell_slope || p->orientation_per_time)
        return false;
    if (p->panel_model == ATL_PANEL_HULD) return p->trigon_model == ATL_TRIGON_OTHER && p->irradiation == ATL_IRR_TOTAL;
    if (p->panel_model == ATL_PANEL_NONE)  // (which component the tail returns is a run-time switch: PvConst::irr)
        return p->trigon_model == ATL_TRIGON_SIMPLE && p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND;
    if (p->panel_model == ATL_PANEL_BOFINGER) return p->trigon_model == ATL_TRIGON_SIMPLE && p->irradiation == ATL_IRR_TOTAL;
    return false;
}

int pvkt_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                           const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    const int rc = pvkt_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // namespace atl
