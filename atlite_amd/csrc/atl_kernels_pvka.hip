// The fast pv kernel family behind a tracker, Hay-Davies ("other") trigon model before the irradiation and the bofinger
// tails: irradiation(tracking=..., trigon_model="other"), pv(panel="KANENA", tracking=..., trigon_model="other"), one
// orientation for the grid or one per cell - FUSED (convert + aggregate) kernels only, stored solar angles, with and
// without the night early-out: 32 kernels.  With atl_kernels_pvk.hip (Huld, simple), atl_kernels_pvkt.hip (one orientation:
// Hay-Davies + Huld, irradiation, bofinger) and atl_kernels_pvkc.hip (per-cell orientations of those) every tracker x
// trigon model x panel x orientation combination of an aggregated call is a fast-family kernel since round 6; per-cell
// results of the rare combinations and launches that cannot be vectorised stay with the general kernel
// (atl_kernels_pvx.hip: 0.40-0.46 of the HBM peak).  (A run-time tracker switch - one instantiation instead of four - was
// tried first: 168 VGPRs + scratch, 5.9 ms where these take 3.3.)
// Reference arithmetic: atlite/pv/orientation.py:104-196 (closed forms: panel_geom in atl_conv_pv.h),
// atlite/pv/irradiation.py:76-145, 214-255; atlite/pv/solar_panel_model.py:47-74; atlite/convert.py:748-767.
#include "atl_kernel_templates.h"

namespace atl {
int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
}  // namespace atl

namespace {

#include "atl_conv_pv.h"
#include "atl_pv_make.h"

// the stored-angle, direct / diffuse / albedo dataset every tracked fast-family kernel reads
bool tracked_dataset(const atl_pv_inputs *in, const atl_pv_params *p) {
    return p->tracking >= ATL_TRACK_HORIZONTAL && p->tracking <= ATL_TRACK_DUAL && !in->d_influx && in->d_influx_direct &&
           in->d_influx_diffuse && in->d_influx_toa && in->d_albedo && in->d_temperature && in->d_solar_altitude &&
           in->d_solar_azimuth && !p->orientation_per_time;
}

template <class F>
int with_tracker(const atl_pv_params *p, F &&f) {
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL: return f(std::integral_constant<int, ATL_TRACK_HORIZONTAL>());
        case ATL_TRACK_TILTED_HORIZONTAL: return f(std::integral_constant<int, ATL_TRACK_TILTED_HORIZONTAL>());
        case ATL_TRACK_VERTICAL: return f(std::integral_constant<int, ATL_TRACK_VERTICAL>());
        case ATL_TRACK_DUAL: return f(std::integral_constant<int, ATL_TRACK_DUAL>());
        default: break;
    }
    atl::set_error("atl_pv: tracking code %d has no tracker in the fast family", p->tracking);
    return int(ATL_E_INVALID);
}

// f(converter instance) for (tracker, tail, scalar / per-cell orientation, night early-out)
template <class F>
int pvka_dispatch(const atl_pv_params *p, F &&f) {
    const bool skip = p->night_skip != 0, pc = p->d_cell_slope != nullptr;
    return with_tracker(p, [&](auto trk) {
        constexpr int TR = decltype(trk)::value;
        auto tail = [&](auto tl) {
            constexpr int TL = decltype(tl)::value;
            if (pc) return skip ? f(PvConvT<false, true, true, TL, TR>()) : f(PvConvT<false, true, false, TL, TR>());
            return skip ? f(PvConvT<false, false, true, TL, TR>()) : f(PvConvT<false, false, false, TL, TR>());
        };
        return p->panel_model == ATL_PANEL_NONE ? tail(std::integral_constant<int, kTailIrradiationHayDavies>())
                                                : tail(std::integral_constant<int, kTailBofingerHayDavies>());
    });
}

}  // namespace

namespace atl {

// which calls this unit takes (atl_pv_convert_aggregate asks before it falls back to the general kernel)
bool pvka_takes(const atl_pv_inputs *in, const atl_pv_params *p) {
    if (!tracked_dataset(in, p) || p->trigon_model != ATL_TRIGON_OTHER) return false;
    if (p->panel_model == ATL_PANEL_NONE) return p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND;
    return p->panel_model == ATL_PANEL_BOFINGER && p->irradiation == ATL_IRR_TOTAL;
}

int pvka_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                           const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    const int rc = pvka_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // namespace atl
