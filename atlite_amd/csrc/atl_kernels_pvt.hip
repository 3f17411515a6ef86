// The fast pv kernel family's tails other than the Huld panel: the bofinger panel (pv(panel="KANENA")), the solar
// thermal collector (solar_thermal()) and the plain tilted irradiation (irradiation()), each after the simple or the
// Hay-Davies ("other") trigon model, fixed panel or (bofinger, irradiation) one of the four trackers, stored solar angles.  Same PvConvT template as atl_kernels_pv.hip;
// a translation unit of its own so that the kernel files compile in parallel.
// Reference arithmetic: atlite/convert.py:550-574, 748-767; atlite/pv/irradiation.py:76-145, 214-255;
// atlite/pv/solar_panel_model.py:47-74.
#include "atl_kernel_templates.h"

namespace {

#include "atl_conv_pv.h"
#include "atl_pv_make.h"

// f(converter instance) for (panel model, trigon model, tracker, scalar / per-cell orientation)
template <class F>
int pvt_dispatch(const atl_pv_params *p, F &&f) {
    const bool pc = p->d_cell_slope != nullptr, hd = p->trigon_model == ATL_TRIGON_OTHER;
    auto with = [&](auto simple, auto other, auto trk) {
        constexpr int TS = decltype(simple)::value, TO = decltype(other)::value, TR = decltype(trk)::value;
        if constexpr (TR == ATL_TRACK_NONE) {  // the night early-out: fixed panels only here (compile time)
            if (p->night_skip) {
                if (hd) return pc ? f(PvConvT<false, true, true, TO>()) : f(PvConvT<false, false, true, TO>());
                return pc ? f(PvConvT<false, true, true, TS>()) : f(PvConvT<false, false, true, TS>());
            }
        }
        if (hd) return pc ? f(PvConvT<false, true, false, TO, TR>()) : f(PvConvT<false, false, false, TO, TR>());
        return pc ? f(PvConvT<false, true, false, TS, TR>()) : f(PvConvT<false, false, false, TS, TR>());
    };
    auto panel = [&](auto trk) {
        switch (p->panel_model) {
            case ATL_PANEL_BOFINGER:
                return with(std::integral_constant<int, kTailBofinger>(), std::integral_constant<int, kTailBofingerHayDavies>(), trk);
            case ATL_PANEL_NONE:
                return with(std::integral_constant<int, kTailIrradiation>(), std::integral_constant<int, kTailIrradiationHayDavies>(), trk);
            case ATL_PANEL_SOLAR_THERMAL:  // convert_solar_thermal has no tracking argument: pv_needs_general keeps trackers away
                if constexpr (decltype(trk)::value == ATL_TRACK_NONE)
                    return with(std::integral_constant<int, kTailThermal>(), std::integral_constant<int, kTailThermalHayDavies>(), trk);
                break;
            default: break;
        }
        atl::set_error("atl_pv: panel_model code %d / tracking code %d has no tail in the fast family", p->panel_model, p->tracking);
        return int(ATL_E_INVALID);
    };
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL: return panel(std::integral_constant<int, ATL_TRACK_HORIZONTAL>());
        case ATL_TRACK_TILTED_HORIZONTAL: return panel(std::integral_constant<int, ATL_TRACK_TILTED_HORIZONTAL>());
        case ATL_TRACK_VERTICAL: return panel(std::integral_constant<int, ATL_TRACK_VERTICAL>());
        case ATL_TRACK_DUAL: return panel(std::integral_constant<int, ATL_TRACK_DUAL>());
        default: return panel(std::integral_constant<int, ATL_TRACK_NONE>());
    }
}

}  // namespace

namespace atl {

int pvt_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out) {
    bool vec;
    return pvt_dispatch(p, [&](auto c) {
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert", in->X);
    });
}

int pvt_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    return pvt_dispatch(p, [&](auto c) {
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
}

}  // namespace atl
