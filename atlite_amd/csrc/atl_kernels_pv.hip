// Fast pv kernel family (PvConvT: stored or in-kernel solar position, night early-out, Huld / Hay-Davies /
// bofinger / solar thermal / irradiation tails, closed-form trackers) and the pv entry points of the C ABI.
// The general kernel (PvxConvT) is compiled in atl_kernels_pvx.hip and reached through atl::pvx_convert*.
// Reference arithmetic: atlite/convert.py:550-574, 748-767, 840-854; atlite/pv/*.py.
#include "atl_kernel_templates.h"

namespace atl {
// atl_kernels_pvx.hip
int pvx_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
}  // namespace atl

namespace {

#include "atl_conv_pv.h"

// ---- converter construction + validation ---------------------------------------------
template <class PV>
int make_pv(const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PV *c, bool *vec) {
    ATL_REQUIRE(in && p, "atl_pv: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_pv: negative shape");
    if (in->d_influx) {  // the influx / outflux head (pv_influx_fast)
        ATL_REQUIRE(in->d_outflux && in->d_influx_toa, "atl_pv: an influx dataset needs outflux and influx_toa here");
    } else {
        ATL_REQUIRE(in->d_influx_direct && in->d_influx_diffuse && in->d_influx_toa,
                    "atl_pv: need influx_direct, influx_diffuse and influx_toa (irradiation.py:209-213)");
        ATL_REQUIRE(in->d_albedo, "atl_pv: need albedo (irradiation.py:128-139)");
    }
    ATL_REQUIRE(in->d_temperature, "atl_pv: need temperature");
    if (in->d_solar_altitude || in->d_solar_azimuth) {
        ATL_REQUIRE(in->d_solar_altitude && in->d_solar_azimuth,
                    "atl_pv: solar_altitude and solar_azimuth must be given together");
    } else {
        ATL_REQUIRE(in->d_sin_dec && in->d_cos_dec && in->d_hour_angle && in->d_cos_hour_angle && in->d_sin_lat &&
                        in->d_cos_lat,
                    "atl_pv: need either solar_altitude/solar_azimuth or the solar position tables");
        ATL_REQUIRE(in->X > 0 && S % in->X == 0, "atl_pv: X (%lld) must divide the number of cells (%lld)",
                    (long long)in->X, (long long)S);
    }
    ATL_REQUIRE((p->d_cell_slope == nullptr) == (p->d_cell_azimuth == nullptr),
                "atl_pv: d_cell_slope and d_cell_azimuth must be given together");
    c->in = *in;
    c->S = S;
    c->k = pv_const_of(p);
    c->o.ss = sin(p->slope);
    c->o.cs = cos(p->slope);
    c->o.hp = (1.0 + c->o.cs) / 2.0;
    c->o.hm = (1.0 - c->o.cs) / 2.0;
    c->o.saz = p->azimuth;
    c->o.slope = p->slope;
    {
        const double sh = sin(p->slope / 2.0);
        c->o.sh3 = sh * sh * sh;
    }
    if constexpr (pv_is_sp<PV>::value) {
        c->oa.csaz = cos(p->azimuth);
        c->oa.ssaz = sin(p->azimuth);
    }
    c->cell_slope = p->d_cell_slope;
    c->cell_azimuth = p->d_cell_azimuth;
    *vec = vec_ok(S, {in->d_influx_direct, in->d_influx_diffuse, in->d_influx_toa, in->d_albedo,
                      in->d_temperature, in->d_solar_altitude, in->d_solar_azimuth, in->d_influx, in->d_outflux});
    return ATL_OK;
}

// f(converter instance) with the PvConvT instantiation for (stored / computed solar position,
// scalar / per-cell orientation)
// influx / outflux datasets the fast family takes: pv() defaults on a SARAH-shaped cutout (total influx + outflux,
// "simple" clearsky model, stored solar angles, Huld panel, fixed panel, simple trigon model)
bool pv_influx_fast(const atl_pv_inputs *in, const atl_pv_params *p) {
    return in->d_influx && in->d_outflux && !in->d_albedo && !in->d_influx_direct && !in->d_influx_diffuse &&
           in->d_solar_altitude && in->d_solar_azimuth && in->d_temperature && p->clearsky_model == ATL_CLEARSKY_SIMPLE &&
           p->panel_model == ATL_PANEL_HULD && p->tracking == ATL_TRACK_NONE && p->trigon_model == ATL_TRIGON_SIMPLE &&
           p->irradiation == ATL_IRR_TOTAL && !p->orientation_per_time;
}

template <class F>
int pv_dispatch(const atl_pv_inputs *in, const atl_pv_params *p, bool allow_skip, F &&f) {
    const bool sp = !(in->d_solar_altitude || in->d_solar_azimuth), pc = p->d_cell_slope != nullptr;
    if (pv_influx_fast(in, p)) {
        if (p->night_skip && allow_skip)
            return pc ? f(PvConvT<false, true, true, kTailHuld, ATL_TRACK_NONE, 1>()) : f(PvConvT<false, false, true, kTailHuld, ATL_TRACK_NONE, 1>());
        return pc ? f(PvConvT<false, true, false, kTailHuld, ATL_TRACK_NONE, 1>()) : f(PvConvT<false, false, false, kTailHuld, ATL_TRACK_NONE, 1>());
    }
    // pv_needs_general() admits trackers only with stored angles and the Huld panel
    auto tracker = [&](auto trk) {
        constexpr int TR = decltype(trk)::value;
        if (p->trigon_model == ATL_TRIGON_OTHER)
            return pc ? f(PvConvT<false, true, false, kTailHuldHayDavies, TR>()) : f(PvConvT<false, false, false, kTailHuldHayDavies, TR>());
        return pc ? f(PvConvT<false, true, false, kTailHuld, TR>()) : f(PvConvT<false, false, false, kTailHuld, TR>());
    };
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL: return tracker(std::integral_constant<int, ATL_TRACK_HORIZONTAL>());
        case ATL_TRACK_TILTED_HORIZONTAL: return tracker(std::integral_constant<int, ATL_TRACK_TILTED_HORIZONTAL>());
        case ATL_TRACK_VERTICAL: return tracker(std::integral_constant<int, ATL_TRACK_VERTICAL>());
        case ATL_TRACK_DUAL: return tracker(std::integral_constant<int, ATL_TRACK_DUAL>());
        default: break;
    }
    if (p->panel_model == ATL_PANEL_BOFINGER)  // simple trigon model only (pv_needs_general)
        return pc ? f(PvConvT<false, true, false, kTailBofinger>()) : f(PvConvT<false, false, false, kTailBofinger>());
    if (p->trigon_model == ATL_TRIGON_OTHER)  // ... and Hay-Davies only with stored angles + Huld
        return pc ? f(PvConvT<false, true, false, kTailHuldHayDavies>()) : f(PvConvT<false, false, false, kTailHuldHayDavies>());
    if (p->panel_model == ATL_PANEL_SOLAR_THERMAL)
        return pc ? f(PvConvT<false, true, false, kTailThermal>()) : f(PvConvT<false, false, false, kTailThermal>());
    if (p->panel_model == ATL_PANEL_NONE)
        return pc ? f(PvConvT<false, true, false, kTailIrradiation>()) : f(PvConvT<false, false, false, kTailIrradiation>());
    if (p->night_skip && allow_skip) {
        if (sp) return pc ? f(PvConvT<true, true, true>()) : f(PvConvT<true, false, true>());
        return pc ? f(PvConvT<false, true, true>()) : f(PvConvT<false, false, true>());
    }
    if (sp) return pc ? f(PvConvT<true, true>()) : f(PvConvT<true, false>());
    return pc ? f(PvConvT<false, true>()) : f(PvConvT<false, false>());
}

bool pv_needs_general(const atl_pv_inputs *in, const atl_pv_params *p) {
    if (pv_influx_fast(in, p)) return false;
    if (in->d_influx != nullptr || in->d_albedo == nullptr || p->orientation_per_time) return true;
    if (p->tracking != ATL_TRACK_NONE)  // trackers: fast family for pv() with the Huld panel
        return !(p->tracking >= ATL_TRACK_HORIZONTAL && p->tracking <= ATL_TRACK_DUAL &&
                 (p->trigon_model == ATL_TRIGON_SIMPLE || p->trigon_model == ATL_TRIGON_OTHER) &&
                 p->panel_model == ATL_PANEL_HULD && p->irradiation == ATL_IRR_TOTAL &&
                 in->d_solar_altitude != nullptr && in->d_temperature != nullptr);
    if (p->trigon_model != ATL_TRIGON_SIMPLE)  // Hay-Davies: fast family only for pv() itself
        return !(p->trigon_model == ATL_TRIGON_OTHER && p->panel_model == ATL_PANEL_HULD &&
                 p->irradiation == ATL_IRR_TOTAL && in->d_solar_altitude != nullptr && in->d_temperature != nullptr);
    // fixed panel, simple trigon model, direct / diffuse / albedo cubes: the fast kernel family, with the
    // Huld panel, the solar thermal collector or the plain irradiation as its tail
    const bool stored = in->d_solar_altitude != nullptr, all7 = stored && in->d_temperature != nullptr;
    switch (p->panel_model) {
        case ATL_PANEL_HULD: return p->irradiation != ATL_IRR_TOTAL;
        case ATL_PANEL_SOLAR_THERMAL: return !(all7 && p->irradiation == ATL_IRR_TOTAL);
        case ATL_PANEL_NONE:
            return !(all7 && p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND);
        case ATL_PANEL_BOFINGER: return !(all7 && p->irradiation == ATL_IRR_TOTAL);
        default: return true;
    }
}

}  // namespace

extern "C" {

int atl_pv_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                   int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert: ctx/inputs/params is NULL");
    if (pv_needs_general(in, p)) return pvx_convert(ctx, in, p, T, S, time_agg, d_out);
    bool vec;
    return pv_dispatch(in, p, true, [&](auto c) {  // night skip: k_cells_night for the SKIP converters
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
    });
}

int atl_pv_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                             int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert_aggregate: ctx/inputs/params is NULL");
    if (pv_needs_general(in, p)) return pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    bool vec;
    return pv_dispatch(in, p, true, [&](auto c) {
        int rc = make_pv(in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
}

}  // extern "C"
