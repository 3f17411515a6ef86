// The fused (convert + aggregate) kernels of the wind converters and their C-ABI launcher; the per-cell kernels live in
// atl_kernels.hip.  Reference arithmetic: atlite/convert.py:634-662 (np.interp on the power curve), atlite/wind.py:76-112
// (log / power law), atlite/aggregate.py:16-35.
#include "atl_kernel_templates.h"

namespace {

#include "atl_conv_wind.h"
#include "atl_wind_make.h"

}  // namespace

extern "C" {

int atl_wind_convert_aggregate(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p,
                               int64_t T, int64_t S, const atl_agg *agg, int time_agg, double *d_out,
                               int64_t ld_out) {
    ATL_REQUIRE(ctx, "atl_wind_convert_aggregate: ctx is NULL");
    WindConvT<-1> g;
    bool vec, finite;
    size_t lds;
    int rc = make_wind(ctx, in, p, T, S, &g, &vec, &lds, &finite);
    if (rc) return rc;
    return wind_dispatch(g, finite, [&](const auto &c) {
        return run_fused(ctx, c, vec, lds, T, S, agg, time_agg, d_out, ld_out, "atl_wind_convert_aggregate");
    });
}

}  // extern "C"
