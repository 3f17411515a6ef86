// General pv kernel (PvxConvT): every option of convert_pv / convert_irradiation / convert_solar_thermal that the
// fast family of atl_kernels_pv.hip does not cover - influx-only / outflux datasets, irradiation() and
// solar_thermal() with a tracker or the Hay-Davies model, bofinger with either, in-kernel solar position with any
// non-default option.  Its own translation unit: these are the largest kernels of the library.
// Reference arithmetic: atlite/pv/irradiation.py:13-255, pv/orientation.py:104-196, pv/solar_panel_model.py.
#include "atl_kernel_templates.h"

namespace {

#include "atl_conv_pv.h"

#include "atl_pvx_make.h"

}  // namespace

namespace atl {

int pvx_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, int time_agg,
                double *d_out) {
    bool vec;
    return pvx_dispatch(p, [&](auto c) {
        int rc = make_pvx(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_cells(ctx, c, vec, 0, T, S, time_agg, d_out, "atl_pv_convert");
    });
}

}  // namespace atl
