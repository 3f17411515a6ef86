// Fast pv kernel family (PvConvT: stored or in-kernel solar position, night early-out, the Huld panel after either
// trigon model, fixed panel) and the pv entry points of the C ABI.  The family's other tails (bofinger, solar thermal,
// irradiation) are compiled in atl_kernels_pvt.hip (atl::pvt_convert*), the Huld panel behind the closed-form trackers
// in atl_kernels_pvk.hip (atl::pvk_convert*).
// The general kernel (PvxConvT) is compiled in atl_kernels_pvx.hip and reached through atl::pvx_convert*.
// Reference arithmetic: atlite/convert.py:550-574, 748-767, 840-854; atlite/pv/*.py.
#include "atl_kernel_templates.h"

namespace atl {
// atl_kernels_pvx.hip
int pvx_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
// atl_kernels_pvi.hip
int pvi_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvi_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
// atl_kernels_pvt.hip
int pvt_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvt_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
// atl_kernels_pvkt.hip
bool pvkt_takes(const atl_pv_inputs *in, const atl_pv_params *p);
// atl_kernels_pvka.hip, atl_kernels_pvkc.hip
bool pvka_takes(const atl_pv_inputs *in, const atl_pv_params *p);
int pvka_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                           const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
bool pvkc_takes(const atl_pv_inputs *in, const atl_pv_params *p);
int pvkc_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                           const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
int pvkt_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                           const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
// atl_kernels_pvk.hip
int pvk_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out);
int pvk_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
}  // namespace atl

namespace {

#include "atl_conv_pv.h"
#include "atl_pv_make.h"

// influx datasets the fast family takes (atl_kernels_pvi.hip): total influx + outflux or an albedo variable, either
// clearsky model, stored solar angles, the Huld panel on a fixed mount after either trigon model
bool pv_influx_fast(const atl_pv_inputs *in, const atl_pv_params *p) {
    return in->d_influx && (in->d_outflux || in->d_albedo) && !in->d_influx_direct && !in->d_influx_diffuse &&
           in->d_solar_altitude && in->d_solar_azimuth && in->d_temperature &&
           (p->clearsky_model == ATL_CLEARSKY_SIMPLE || (p->clearsky_model == ATL_CLEARSKY_ENHANCED && in->d_humidity)) &&
           p->panel_model == ATL_PANEL_HULD && p->tracking == ATL_TRACK_NONE &&
           (p->trigon_model == ATL_TRIGON_SIMPLE || p->trigon_model == ATL_TRIGON_OTHER) && p->irradiation == ATL_IRR_TOTAL &&
           !p->orientation_per_time;
}

// f(converter instance) with the PvConvT instantiation for (stored / computed solar position,
// scalar / per-cell orientation)
template <class F>
int pv_dispatch(const atl_pv_inputs *in, const atl_pv_params *p, bool allow_skip, F &&f) {
    const bool sp = !(in->d_solar_altitude || in->d_solar_azimuth), pc = p->d_cell_slope != nullptr;
    if (p->trigon_model == ATL_TRIGON_OTHER) {  // Hay-Davies with stored angles (pv_needs_general); other panels: pv_other_tail
        if (p->night_skip && allow_skip)
            return pc ? f(PvConvT<false, true, true, kTailHuldHayDavies>()) : f(PvConvT<false, false, true, kTailHuldHayDavies>());
        return pc ? f(PvConvT<false, true, false, kTailHuldHayDavies>()) : f(PvConvT<false, false, false, kTailHuldHayDavies>());
    }
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
    if (p->tracking != ATL_TRACK_NONE) {  // trackers: fast family with stored angles, the Huld panel and the simple trigon model (atl_kernels_pvk.hip)
        if (!(p->tracking >= ATL_TRACK_HORIZONTAL && p->tracking <= ATL_TRACK_DUAL && in->d_solar_altitude != nullptr &&
              in->d_temperature != nullptr && p->panel_model == ATL_PANEL_HULD && p->trigon_model == ATL_TRIGON_SIMPLE))
            return true;
    }
    // fixed panel, direct / diffuse / albedo cubes, either trigon model: the fast kernel family, with the Huld panel,
    // the bofinger panel, the solar thermal collector or the plain irradiation as its tail
    if (p->trigon_model != ATL_TRIGON_SIMPLE && p->trigon_model != ATL_TRIGON_OTHER) return true;
    const bool stored = in->d_solar_altitude != nullptr, all7 = stored && in->d_temperature != nullptr;
    if (p->trigon_model == ATL_TRIGON_OTHER && !all7) return true;
    switch (p->panel_model) {
        case ATL_PANEL_HULD: return p->irradiation != ATL_IRR_TOTAL;
        case ATL_PANEL_SOLAR_THERMAL: return !(all7 && p->irradiation == ATL_IRR_TOTAL);
        case ATL_PANEL_NONE:
            return !(all7 && p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND);
        case ATL_PANEL_BOFINGER: return !(all7 && p->irradiation == ATL_IRR_TOTAL);
        default: return true;
    }
}

// ... of which the tails other than the Huld panel are compiled in atl_kernels_pvt.hip,
bool pv_other_tail(const atl_pv_inputs *in, const atl_pv_params *p) {
    return !pv_influx_fast(in, p) && p->panel_model != ATL_PANEL_HULD;
}
// the Huld panel behind a tracker in atl_kernels_pvk.hip
bool pv_tracked(const atl_pv_inputs *in, const atl_pv_params *p) {
    return !pv_influx_fast(in, p) && p->panel_model == ATL_PANEL_HULD && p->tracking != ATL_TRACK_NONE;
}

}  // namespace

extern "C" {

int atl_pv_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                   int time_agg, double *d_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert: ctx/inputs/params is NULL");
    if (pv_needs_general(in, p)) return pvx_convert(ctx, in, p, T, S, time_agg, d_out);
    if (pv_influx_fast(in, p)) return pvi_convert(ctx, in, p, T, S, time_agg, d_out);
    if (pv_other_tail(in, p)) return pvt_convert(ctx, in, p, T, S, time_agg, d_out);
    if (pv_tracked(in, p)) return pvk_convert(ctx, in, p, T, S, time_agg, d_out);
    bool vec;
    auto run = [&](auto c) {  // night skip: k_cells_night for the SKIP converters
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert", in->X);
    };
    int rc = pv_dispatch(in, p, true, run);
    // a launch that cannot be vectorised: the early-out converters have no such instantiation - the converters that
    // read every byte give the same bits - and of those only pv() with its defaults has one: the general kernel takes the rest
    if (rc == kNeedScalar) rc = pv_dispatch(in, p, false, run);
    return rc == kNeedScalar ? pvx_convert(ctx, in, p, T, S, time_agg, d_out) : rc;
}

int atl_pv_day_map(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, const atl_agg *agg,
                   uint8_t *d_map, int64_t ld) {
    ATL_REQUIRE(ctx && in && p && agg && d_map, "atl_pv_day_map: a NULL argument");
    ATL_REQUIRE(agg->ctx == ctx, "atl_pv_day_map: aggregation plan belongs to another context");
    ATL_REQUIRE(in->d_solar_altitude, "atl_pv_day_map: needs the stored solar altitude");
    ATL_REQUIRE(agg->dev.shift_classes == 0, "atl_pv_day_map: not for line-aligned plans");
    ATL_REQUIRE(agg->dev.n_cells == S, "atl_pv_day_map: matrix has %lld columns but the cutout has %lld cells",
                (long long)agg->dev.n_cells, (long long)S);
    ATL_REQUIRE(T >= 0 && ld >= (T + 7) / 8 * 8 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(d_map) & 7u) == 0,
                "atl_pv_day_map: ld %lld must be a multiple of 8 of at least %lld time steps rounded up to 8, the map 8-byte aligned",
                (long long)ld, (long long)T);
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    const PlanDev &plan = agg->dev;
    if (plan.n_segs == 0) return ATL_OK;
    ATL_HIP_TRY(hipMemsetAsync(d_map, 0, size_t(plan.n_segs) * size_t(ld), ctx->stream));
    if (T == 0) return ATL_OK;
    PvConvT<false, false, true> c;  // the keys of every stored-angle early-out converter: altitude against the cut-off
    const int64_t stride = slot_stride_of(ctx, S);
    ATL_REQUIRE(stride >= S, "atl_pv_day_map: slot stride %lld is smaller than the %lld cells of a slot", (long long)stride, (long long)S);
    c.in = *in;
    c.S = stride;
    c.k = pv_const_of(p);
    c.cell_slope = c.cell_azimuth = nullptr;
    const bool vec = vec_ok(T, S, stride, {in->d_solar_altitude});
    const int64_t n_units = ((T + 63) / 64) * int64_t(plan.n_segs);
    const dim3 grid(unsigned((n_units + 3) / 4));
    if (vec)
        hipLaunchKernelGGL((k_day_map<PvConvT<false, false, true>, true>), grid, dim3(256), 0, ctx->stream, c, plan, T, S, n_units, d_map, ld);
    else
        hipLaunchKernelGGL((k_day_map<PvConvT<false, false, true>, false>), grid, dim3(256), 0, ctx->stream, c, plan, T, S, n_units, d_map, ld);
    return check_launch("atl_pv_day_map");
}

int atl_pv_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                             int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    ATL_REQUIRE(ctx && in && p, "atl_pv_convert_aggregate: ctx/inputs/params is NULL");
    if (pvkt_takes(in, p)) return pvkt_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pvka_takes(in, p)) return pvka_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pvkc_takes(in, p)) return pvkc_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pv_needs_general(in, p)) return pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pv_influx_fast(in, p)) return pvi_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pv_other_tail(in, p)) return pvt_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    if (pv_tracked(in, p)) return pvk_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out);
    bool vec;
    auto run = [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    };
    int rc = pv_dispatch(in, p, true, run);
    if (rc == kNeedScalar) rc = pv_dispatch(in, p, false, run);
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // extern "C"
