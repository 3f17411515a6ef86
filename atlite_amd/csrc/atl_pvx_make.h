// Host side of the general pv kernel's converters (PvxConvT): the choice of the instantiation, validation of the C-ABI structs and
// construction.  Included by atl_kernels_pvx.hip (pvx_convert: the per-cell kernels) and atl_kernels_pvxa.hip
// (pvx_convert_aggregate: the fused kernels) inside their anonymous namespace, after atl_conv_pv.h - two translation units since
// round 6: the one they were was the longest compile of the build (40 s).
#pragma once

// f(converter instance) with the PvxConvT instantiation for (tracker or none, trigon model)
template <class F>
int pvx_dispatch(const atl_pv_params *p, F &&f) {
    const bool other = p->trigon_model == ATL_TRIGON_OTHER;
    switch (p->tracking) {
        case ATL_TRACK_HORIZONTAL:
        case ATL_TRACK_TILTED_HORIZONTAL:
        case ATL_TRACK_VERTICAL:
        case ATL_TRACK_DUAL:  // one instantiation for the four trackers: the geometry is a wave-uniform run-time switch
            return other ? f(PvxConvT<kTrackAny, ATL_TRIGON_OTHER>()) : f(PvxConvT<kTrackAny, ATL_TRIGON_SIMPLE>());
        default:  // ATL_TRACK_NONE; out-of-range codes are rejected by make_pvx
            if (p->orientation_per_time)
                return other ? f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_OTHER, true>()) : f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_SIMPLE, true>());
            return other ? f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_OTHER>()) : f(PvxConvT<ATL_TRACK_NONE, ATL_TRIGON_SIMPLE>());
    }
}

template <class PVX>
int make_pvx(const atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S, PVX *c, bool *vec) {
    ATL_REQUIRE(in && p, "atl_pv: inputs/params is NULL");
    ATL_REQUIRE(T >= 0 && S >= 0, "atl_pv: negative shape");
    ATL_REQUIRE(p->tracking >= ATL_TRACK_NONE && p->tracking <= ATL_TRACK_DUAL, "atl_pv: bad tracking code %d",
                p->tracking);
    ATL_REQUIRE(p->trigon_model == ATL_TRIGON_SIMPLE || p->trigon_model == ATL_TRIGON_OTHER,
                "atl_pv: bad trigon_model code %d", p->trigon_model);
    ATL_REQUIRE(p->irradiation >= ATL_IRR_TOTAL && p->irradiation <= ATL_IRR_GROUND, "atl_pv: bad irradiation code %d",
                p->irradiation);
    ATL_REQUIRE(p->panel_model >= ATL_PANEL_HULD && p->panel_model <= ATL_PANEL_SOLAR_THERMAL,
                "atl_pv: bad panel_model code %d", p->panel_model);
    ATL_REQUIRE(in->d_influx_toa, "atl_pv: need influx_toa");
    if (in->d_influx) {
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || p->clearsky_model == ATL_CLEARSKY_ENHANCED,
                    "`clearsky model` must be chosen from 'simple' and 'enhanced'");
        ATL_REQUIRE(p->clearsky_model == ATL_CLEARSKY_SIMPLE || (in->d_temperature && in->d_humidity),
                    "atl_pv: the enhanced clearsky model needs temperature and humidity");
    } else {
        ATL_REQUIRE(in->d_influx_direct && in->d_influx_diffuse,
                    "Need either influx or influx_direct and influx_diffuse in the dataset. Check your cutout and "
                    "dataset module.");
    }
    ATL_REQUIRE(in->d_albedo || in->d_outflux,
                "Need either albedo or outflux as a variable in the dataset. Check your cutout and dataset module.");
    ATL_REQUIRE(p->panel_model == ATL_PANEL_NONE || in->d_temperature, "atl_pv: need temperature");
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
    c->o = pvx_opt_of(p, in->d_influx != nullptr, in->d_albedo != nullptr);
    c->slope = p->slope;
    c->azimuth = p->azimuth;
    c->cell_slope = p->d_cell_slope;
    c->cell_azimuth = p->d_cell_azimuth;
    c->ori_per_time = p->orientation_per_time ? 1 : 0;
    ATL_REQUIRE(!p->orientation_per_time || p->d_cell_slope, "atl_pv: orientation_per_time needs the (T,S) slope / azimuth cubes");
    ATL_REQUIRE(!p->orientation_per_time || p->tracking == ATL_TRACK_NONE,
                "atl_pv: an orientation that depends on time cannot be combined with a tracker");
    *vec = vec_ok(T, S, ld, {in->d_influx_direct, in->d_influx_diffuse, in->d_influx, in->d_influx_toa, in->d_albedo,
                      in->d_outflux, in->d_temperature, in->d_humidity, in->d_solar_altitude, in->d_solar_azimuth,
                      p->orientation_per_time ? p->d_cell_slope : nullptr, p->orientation_per_time ? p->d_cell_azimuth : nullptr});
    return ATL_OK;
}
