// Host side of the wind converters (WindConvT): validation of the C-ABI structs, construction, and the choice of the
// instantiation for (extrapolation law, power-curve table).  Included by atl_kernels.hip (atl_wind_convert: the per-cell kernels)
// and atl_kernels_wind.hip (atl_wind_convert_aggregate: the fused kernels) inside their anonymous namespace, after
// atl_conv_wind.h - two translation units since round 6: together they were the longest compile of the build.
#pragma once

int make_wind(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T, int64_t S,
              WindConvT<-1> *c, bool *vec, size_t *lds_bytes, bool *table_finite) {
    ATL_REQUIRE(in && p, "atl_wind: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_wind: negative shape");
    ATL_REQUIRE(in->d_wnd, "atl_wind: wind speed is NULL");
    ATL_REQUIRE(p->method == ATL_WIND_NONE || p->method == ATL_WIND_LOG || p->method == ATL_WIND_POWER,
                "Interpolation method must be 'logarithmic' or 'power' (got code %d)", p->method);
    ATL_REQUIRE(p->method == ATL_WIND_NONE || in->d_aux,
                "atl_wind: method needs roughness / wnd_shear_exp (wind.py:94-98,106-110)");
    if (p->n_knots == 0) {  // no power curve: the extrapolated wind speed itself (wind.py:76-112)
        ATL_REQUIRE(p->method == ATL_WIND_NONE || (p->to_height > 0 && p->from_height > 0 && std::isfinite(p->to_height) &&
                                                    std::isfinite(p->from_height)),
                    "atl_wind: heights must be positive and finite");
        c->wnd = in->d_wnd;
        c->aux = in->d_aux;
        c->S = slot_stride_of(ctx, S);  // the converter's S is what separates the slots of its cubes
        c->aux_static = in->aux_is_static;
        c->method = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;
        c->to_height = p->to_height;
        c->from_height = p->from_height;
        c->log_ratio = log(p->to_height / p->from_height);
        c->table = nullptr;
        c->n_knots = c->n_pad = c->tab_doubles = c->b0 = 0;
        c->vmin = c->vmax = c->inv_w = 0.0;
        *table_finite = true;
        *lds_bytes = size_t(2 * kLogTabN) * sizeof(double);
        *vec = vec_ok(T, S, slot_stride_of(ctx, S), {in->d_wnd, in->aux_is_static ? nullptr : in->d_aux});
        return ATL_OK;
    }
    ATL_REQUIRE(p->n_knots >= 1 && p->n_knots <= kMaxKnots && p->h_V && p->h_POWn,
                "atl_wind: power curve needs 1..%d knots", kMaxKnots);
    std::vector<double> tbl;
    bool finite = true;
    int n_pad = 0;
    const int n = wind_table_build(p->h_V, p->h_POWn, p->n_knots, tbl, &n_pad, &finite);
    ATL_REQUIRE(n > 0, "wind speed 'V' in the turbine config is expected to be increasing");
    ATL_REQUIRE(n <= kMaxKnots, "atl_wind: power curve needs 1..%d knots", kMaxKnots);
    // grid-aligned knots (the usual case): the bucket table replaces the search table
    std::vector<double> grid;
    double inv_w = 0.0;
    int b0 = 0;
    const int n_grid = (finite && !getenv("ATLITE_HIP_WIND_NO_GRID")) ? wind_grid_build(tbl.data(), n, n_pad, grid, &inv_w, &b0) : 0;
    const std::vector<double> tbl_search = tbl;  // vmin / vmax below
    if (n_grid > 0) tbl = grid;
    // pinned staging buffer: wait until the previous call's copy has left it, then enqueue the H2D
    // stream-ordered after any earlier kernel that still reads the device table
    if (ctx->table_pending) ATL_HIP_TRY(hipEventSynchronize(ctx->ev_table));
    memcpy(ctx->h_table, tbl.data(), tbl.size() * sizeof(double));
    ATL_HIP_TRY(hipMemcpyAsync(ctx->d_table, ctx->h_table, tbl.size() * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    ATL_HIP_TRY(hipEventRecord(ctx->ev_table, ctx->stream));
    ctx->table_pending = true;
    c->wnd = in->d_wnd;
    c->aux = in->d_aux;
    c->S = slot_stride_of(ctx, S);
    c->aux_static = in->aux_is_static;
    // (to / from) ** shear with to == from is 1 whatever the shear exponent holds (pow(1, NaN) = 1, which
    // exp(NaN * log 1) would not give): no extrapolation at all
    c->method = (p->method == ATL_WIND_POWER && p->to_height == p->from_height) ? ATL_WIND_NONE : p->method;
    c->to_height = p->to_height;
    c->from_height = p->from_height;
    c->log_ratio = log(p->to_height / p->from_height);
    c->table = ctx->d_table;
    c->n_knots = n;
    c->n_pad = n_pad;
    c->tab_doubles = 5 * n_pad;
    c->vmin = tbl_search[0];
    c->vmax = tbl_search[size_t(n - 1)];
    c->inv_w = 0.0;  // 0: not grid-aligned
    c->b0 = 0;
    if (n_grid > 0) {
        c->tab_doubles = 4 * n_grid;
        c->inv_w = inv_w;
        c->b0 = b0;
    }
    *table_finite = finite;
    *lds_bytes = size_t(c->tab_doubles + 2 * kLogTabN) * sizeof(double);
    *vec = vec_ok(T, S, slot_stride_of(ctx, S), {in->d_wnd, in->aux_is_static ? nullptr : in->d_aux});
    return ATL_OK;
}

template <int M, int STEPS = 0>
WindConvT<M, STEPS> wind_as(const WindConvT<-1> &g) {
    WindConvT<M, STEPS> c;
    c.wnd = g.wnd;
    c.aux = g.aux;
    c.S = g.S;
    c.aux_static = g.aux_static;
    c.method = g.method;
    c.to_height = g.to_height;
    c.from_height = g.from_height;
    c.log_ratio = g.log_ratio;
    c.table = g.table;
    c.n_knots = g.n_knots;
    c.n_pad = g.n_pad;
    c.tab_doubles = g.tab_doubles;
    c.vmin = g.vmin;
    c.vmax = g.vmax;
    c.inv_w = g.inv_w;
    c.b0 = g.b0;
    return c;
}

// run `f(converter)` with the instantiation matching (method, table finiteness)
template <class F>
int wind_dispatch(const WindConvT<-1> &g, bool finite, F &&f) {
    // the fast log-law path also needs positive, finite heights (their logs are taken once)
    const bool heights_ok = g.to_height > 0 && g.from_height > 0 && std::isfinite(g.to_height) &&
                            std::isfinite(g.from_height);
    if (g.n_knots == 0) {  // no power curve (make_wind has checked the heights)
        switch (g.method) {
            case ATL_WIND_LOG: return f(wind_as<ATL_WIND_LOG, kWindIdentity>(g));
            case ATL_WIND_POWER: return f(wind_as<ATL_WIND_POWER, kWindIdentity>(g));
            default: return f(wind_as<ATL_WIND_NONE, kWindIdentity>(g));
        }
    }
    if (!finite || (g.method == ATL_WIND_LOG && !heights_ok)) return f(g);
    // unrolled knot search for the usual table sizes (make_wind pads to 16 / 32 / 128 knots)
    auto sized = [&](auto method) {
        constexpr int M = decltype(method)::value;
        if (g.inv_w > 0.0) return f(wind_as<M, kWindGrid>(g));  // grid-aligned knots: every shipped turbine but a few
        if constexpr (M == ATL_WIND_LOG) {  // the default law carries the unrolled searches; the others the sized loop
            if (g.n_pad == 16) return f(wind_as<M, 4>(g));
            if (g.n_pad == 32) return f(wind_as<M, 5>(g));
            if (g.n_pad == 128) return f(wind_as<M, 7>(g));
        }
        return f(wind_as<M>(g));
    };
    switch (g.method) {
        case ATL_WIND_LOG: return sized(std::integral_constant<int, ATL_WIND_LOG>());
        case ATL_WIND_POWER: return sized(std::integral_constant<int, ATL_WIND_POWER>());
        default: return sized(std::integral_constant<int, ATL_WIND_NONE>());
    }
}
