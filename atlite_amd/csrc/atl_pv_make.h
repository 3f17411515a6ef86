// Host side of the fast pv family's converters (PvConvT): validation of the C-ABI structs and construction.
// Included by atl_kernels_pv.hip and atl_kernels_pvt.hip inside their anonymous namespace, after atl_conv_pv.h.
#pragma once

// ---- fast family: converter construction + validation (host) -------------------------
template <class PV>
int make_pv(const atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PV *c, bool *vec) {
    ATL_REQUIRE(in && p, "atl_pv: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_pv: negative shape");
    if (in->d_influx) {  // the influx / outflux head (pv_influx_fast)
        ATL_REQUIRE((in->d_outflux || in->d_albedo) && in->d_influx_toa,
                    "atl_pv: an influx dataset needs albedo or outflux (irradiation.py:128-139) and influx_toa here");
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || p->clearsky_model == ATL_CLEARSKY_ENHANCED,
                    "`clearsky model` must be chosen from 'simple' and 'enhanced'");
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || in->d_humidity,
                    "atl_pv: the enhanced clearsky model needs temperature and humidity");
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
    const int64_t ld = slot_stride_of(ctx, S);
    ATL_REQUIRE(ld >= S, "atl_pv: slot stride %lld is smaller than the %lld cells of a slot", (long long)ld, (long long)S);
    c->in = *in;
    c->S = ld;  // the converter's S is what separates the slots of its cubes
    c->k = pv_const_of(p);
    if (in->d_influx && in->d_albedo) {  // "albedo" wins over "outflux" (irradiation.py:129-131): it takes the outflux's stream
        c->k.alb_cube = 1;
        c->in.d_outflux = in->d_albedo;
    }
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
    *vec = vec_ok(T, S, ld, {in->d_influx_direct, in->d_influx_diffuse, in->d_influx_toa, in->d_albedo,
                      in->d_temperature, in->d_solar_altitude, in->d_solar_azimuth, in->d_influx, in->d_influx ? c->in.d_outflux : in->d_outflux,
                      in->d_influx && p->clearsky_model == ATL_CLEARSKY_ENHANCED ? in->d_humidity : nullptr});
    return ATL_OK;
}
