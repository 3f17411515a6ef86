// General pv kernel (PvxConvT), fused convert + aggregate instantiations; the per-cell ones: atl_kernels_pvx.hip.
// Reference arithmetic: atlite/pv/irradiation.py:13-255, pv/orientation.py:104-196, pv/solar_panel_model.py, aggregate.py:16-35.
#include "atl_kernel_templates.h"

namespace {

#include "atl_conv_pv.h"
#include "atl_pvx_make.h"

}  // namespace

namespace atl {

int pvx_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                          const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out) {
    bool vec;
    return pvx_dispatch(p, [&](auto c) {
        int rc = make_pvx(ctx, in, p, T, S, &c, &vec);
        if (rc) return rc;
        return run_fused(ctx, c, vec, 0, T, S, agg, time_agg, d_out, ld_out, "atl_pv_convert_aggregate");
    });
}

}  // namespace atl
