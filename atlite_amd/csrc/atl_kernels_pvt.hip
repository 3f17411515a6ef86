// The fast pv kernel family's tails other than the Huld panel: the bofinger panel (pv(panel="KANENA")), the solar
// thermal collector (solar_thermal()) and the plain tilted irradiation (irradiation()), each after the simple or the
// Hay-Davies ("other") trigon model, fixed panel, stored solar angles, with and without the night early-out.  (Behind a
// tracker: the fused kernels of atl_kernels_pvkt.hip for the common combinations, else the general kernel.)  Vectorised launches only:
// odd cell counts / row lengths and unaligned cubes take the general kernel too (atl_kernels_pvx.hip).  Same PvConvT
// template as atl_kernels_pv.hip; a translation unit of its own so that the kernel files compile in parallel.
// Reference arithmetic: atlite/convert.py:550-574, 748-767; atlite/pv/irradiation.py:76-145, 214-255;
// atlite/pv/solar_panel_model.py:47-74.
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

// f(converter instance) for (panel model, trigon model, tracker, scalar / per-cell orientation)
template <class F>
int pvt_dispatch(const atl_pv_params *p, F &&f) {
    const bool pc = p->d_cell_slope != nullptr, hd = p->trigon_model == ATL_TRIGON_OTHER;
    auto with = [&](auto simple, auto other) {
        constexpr int TS = decltype(simple)::value, TO = decltype(other)::value;
        if (p->night_skip) {
            if (hd) return pc ? f(PvConvT<false, true, true, TO>()) : f(PvConvT<false, false, true, TO>());
            return pc ? f(PvConvT<false, true, true, TS>()) : f(PvConvT<false, false, true, TS>());
        }
        if (hd) return pc ? f(PvConvT<false, true, false, TO>()) : f(PvConvT<false, false, false, TO>());
        return pc ? f(PvConvT<false, true, false, TS>()) : f(PvConvT<false, false, false, TS>());
    };
    switch (p->panel_model) {
        case ATL_PANEL_BOFINGER:
            return with(std::integral_constant<int, kTailBofinger>(), std::integral_constant<int, kTailBofingerHayDavies>());
        case ATL_PANEL_NONE:
            return with(std::integral_constant<int, kTailIrradiation>(), std::integral_constant<int, kTailIrradiationHayDavies>());
        case ATL_PANEL_SOLAR_THERMAL:
            return with(std::integral_constant<int, kTailThermal>(), std::integral_constant<int, kTailThermalHayDavies>());
        default: break;
    }
    atl::set_error("atl_pv: panel_model code %d has no tail in the fast family", p->panel_model);
    return int(ATL_E_INVALID);
}

}  // namespace

namespace atl {

int pvt_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out) {
    bool vec;
    const int rc = pvt_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert", in->X);
    });
    return rc == kNeedScalar ? pvx_convert(ctx, in, p, T, S, time_agg, d_out) : rc;
}

int pvt_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    const int rc = pvt_dispatch(p, [&](auto c) {
        int rc = make_pv(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
    return rc == kNeedScalar ? pvx_convert_aggregate(ctx, in, p, T, S, agg, time_agg, d_out, ld_out) : rc;
}

}  // namespace atl
