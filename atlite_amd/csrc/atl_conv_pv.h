// Solar converters: fast pv path (stored or in-kernel solar position, night early-out) and the
// general kernel (tracking, Hay-Davies, Reindl, bofinger, irradiation, solar thermal).
// Reference: atlite/convert.py:550-574, 748-767, 840-854; atlite/pv/*.py.
// Part of libatlite_hip.so (gfx950); included by the pv kernel files (atl_kernels_pv*.hip) and atl_runtime.cpp (host probes) inside their anonymous namespace.
#pragma once

// solar PV, ERA5-shaped inputs with stored solar position
struct PvConst {
    double c_amb, c_irr, r_tmod, inv_r_irr, k1, k2, k3, k4, k5, k6, inv_eff, alt_thr, sin_alt_thr;
    // tails other than the Huld panel (PvConvT<..., TAIL>): solar thermal collector, plain irradiation
    double st_c0, st_c1, st_t_store;
    int irr;  // ATL_IRR_*: which component the irradiation tail returns
    // bofinger panel (solar_panel_model.py:47-74): A, B, C, D, (NOCT - Tamb) / Intc, D * that / ta, Tstd, threshold,
    // inverter efficiency / capacity
    double bA, bB, bC, bD, bfrac, bDf_ta, bTstd, bthr, bscale;
    int tracking;  // ATL_TRACK_*: read by the instantiations whose tracker is a run-time switch (kTrackAny)
    int alb_cube;  // influx head: the dataset has an albedo variable (irradiation.py:129-130) - it rides where the outflux would
};
// TRACK template value of the converters that choose their tracker at run time (a wave-uniform switch over
// panel_geom's closed forms): the rarely used tracker x panel / trigon / orientation combinations share ONE
// instantiation instead of four; pv(tracking="horizontal") - PyPSA-Eur's solar-hsat - keeps its own.
constexpr int kTrackAny = 100;
constexpr bool track_is(int TRACK, int rt, int which) { return TRACK == which || (TRACK == kTrackAny && rt == which); }

// what follows the tilted irradiation in the fast kernel family
// (panel model x transposition model): the Huld panel, the solar thermal collector, the plain irradiation or the bofinger
// panel, each after the simple or the Hay-Davies ("other") trigon model
constexpr int kTailHuld = 0, kTailThermal = 1, kTailIrradiation = 2, kTailHuldHayDavies = 3, kTailBofinger = 4,
              kTailThermalHayDavies = 5, kTailIrradiationHayDavies = 6, kTailBofingerHayDavies = 7;
constexpr bool tail_hay_davies(int t) { return t == kTailHuldHayDavies || t >= kTailThermalHayDavies; }
constexpr int tail_panel(int t) {  // the simple-model tail with the same panel
    return t == kTailHuldHayDavies ? kTailHuld : t == kTailThermalHayDavies ? kTailThermal : t == kTailIrradiationHayDavies ? kTailIrradiation
           : t == kTailBofingerHayDavies ? kTailBofinger : t;
}

// per-cell orientation factors: sin/cos(slope), (1 +- cos(slope))/2, panel azimuth
struct PvOri {
    double ss, cs, hp, hm, saz;
    double slope;  // radians: the trackers' closed forms start from the angle itself
    double sh3;  // sin(slope / 2)^3: Hay-Davies horizon brightening (irradiation.py:101-108)
};
// cos/sin of the panel azimuth: only the in-kernel solar position variant needs them
template <bool SP>
struct PvAz {
    double csaz, ssaz;
};
template <>
struct PvAz<false> {};

// ---- SurfaceOrientation with a tracker (orientation.py:113-188), fast family and general kernel -----------------------------
// What the irradiation model needs of the panel geometry: cos(incidence) (before the clip at 0),
// cos(surface slope) and - Hay-Davies only - sin(surface slope / 2).
struct PanelGeom {
    double cosinc, cs, sh;
};

// The reference's formulas as written (atan / asin / acos through libm).  Out of line: it only runs
// for the degenerate arguments the closed forms below exclude.
ATL_HD __noinline__ PanelGeom panel_geom_literal(int tracking, double sa, double ca, double az, double slope,
                                                     double sazim) {
    const double pi = 3.14159265358979323846;
    double surface_slope, cosinc;
    if (tracking == ATL_TRACK_HORIZONTAL) {
        const double rotation = atan((ca / sa) * sin(az - sazim));
        surface_slope = fabs(rotation);
        const double surface_azimuth = sazim + asin(sin(rotation) / sin(surface_slope));
        cosinc = cos(surface_slope) * sa + sin(surface_slope) * ca * cos(az - surface_azimuth);
    } else {  // tilted_horizontal
        const double tilt = slope;
        double rotation = atan((ca * sin(az - sazim)) / (ca * cos(az - sazim) * sin(tilt) + sa * cos(tilt)));
        surface_slope = acos(cos(rotation) * cos(tilt));
        double ad = az - sazim;
        ad = ad > pi ? ad - 2 * pi : ad;
        ad = ad < -pi ? 2 * pi + ad : ad;
        rotation = (rotation < 0 && ad > 0) ? rotation + pi : rotation;
        rotation = (rotation > 0 && ad < 0) ? rotation - pi : rotation;
        cosinc = cos(rotation) * (sin(tilt) * ca * cos(az - sazim) + cos(tilt) * sa) + sin(rotation) * ca * sin(az - sazim);
    }
    return PanelGeom{cosinc, cos(surface_slope), sin(surface_slope / 2.0)};
}

// The rotating trackers without inverse trigonometry.  With q = tan(rotation):
//   cos(atan q) = 1 / sqrt(1 + q^2),  sin(atan q) = q / sqrt(1 + q^2)
// horizontal:        slope' = |rotation|;  asin(sin r / sin|r|) = asin(+-1) = +-pi/2, so
//                    cos(az - azimuth') = sgn(q) sin(az - azimuth)  and
//                    cosinc = (sa + q ca sin d) / sqrt(1 + q^2);  q = 0 gives 0/0 = NaN in the reference
// tilted_horizontal: cos(slope') = cos(r) cos(tilt) (the acos is only ever fed back into cos / sin(./2));
//                    the +-pi correction of the rotation flips the sign of both cos(r) and sin(r)
// The results agree with the literal sequence to a few ulp (tests: reference-generated vectors at
// rtol 1e-10); arguments for which the closed forms are not valid (q zero / non-finite) take the
// literal routine.
template <int TRACK, bool NEED_SH>
ATL_HD __forceinline__ PanelGeom panel_geom(double sa, double ca, double az, double slope, double sazim, int tracking_rt = ATL_TRACK_NONE) {
    const double pi = 3.14159265358979323846;
    PanelGeom g;
    g.sh = 0.0;
    if constexpr (TRACK == kTrackAny) {  // wave-uniform: every lane of a launch has the same tracker
        switch (tracking_rt) {
            case ATL_TRACK_HORIZONTAL: return panel_geom<ATL_TRACK_HORIZONTAL, NEED_SH>(sa, ca, az, slope, sazim);
            case ATL_TRACK_TILTED_HORIZONTAL: return panel_geom<ATL_TRACK_TILTED_HORIZONTAL, NEED_SH>(sa, ca, az, slope, sazim);
            case ATL_TRACK_VERTICAL: return panel_geom<ATL_TRACK_VERTICAL, NEED_SH>(sa, ca, az, slope, sazim);
            case ATL_TRACK_DUAL: return panel_geom<ATL_TRACK_DUAL, NEED_SH>(sa, ca, az, slope, sazim);
            default: return panel_geom<ATL_TRACK_NONE, NEED_SH>(sa, ca, az, slope, sazim);
        }
    } else if constexpr (TRACK == ATL_TRACK_NONE) {
        g.cs = lean_cos(slope);
        g.cosinc = lean_sin(slope) * ca * lean_cos(sazim - az) + g.cs * sa;
        if constexpr (NEED_SH) g.sh = lean_sin(slope / 2.0);
    } else if constexpr (TRACK == ATL_TRACK_VERTICAL) {
        g.cs = lean_cos(slope);
        g.cosinc = lean_sin(slope) * ca + g.cs * sa;
        if constexpr (NEED_SH) g.sh = lean_sin(slope / 2.0);
    } else if constexpr (TRACK == ATL_TRACK_DUAL) {
        g.cs = lean_cos(slope);  // the slope stays the panel's; the simple model substitutes sin(alt) itself
        g.cosinc = 1.0;
        if constexpr (NEED_SH) g.sh = lean_sin(slope / 2.0);
    } else if constexpr (TRACK == ATL_TRACK_HORIZONTAL) {
        const double sd = lean_sin(az - sazim);
        // the reciprocal path without guarded_div's range check: a quotient that leaves the normal range makes q zero /
        // non-finite / huge, and those take the literal routine below anyway (C2: 2.65 -> 2.59 ms with the early-out)
        const double q = fast_div(ca, sa) * sd;
        const bool ok = q != 0.0 && __builtin_fabs(q) < 0x1.0p200;  // false for NaN / inf too
        double cr;  // 1 / w, w = sqrt(1 + q^2) in [1, 2^200] whenever ok
        [[maybe_unused]] const double w = lean_sqrt_rsqrt(__builtin_fma(q, q, 1.0), &cr);
        g.cs = cr;
        g.cosinc = cr * (sa + q * ca * sd);
        if constexpr (NEED_SH) {
            double rs;
            lean_sqrt_rsqrt(2.0 * w * (w + 1.0), &rs);
            g.sh = __builtin_fabs(q) * rs;
        }
        if (__builtin_expect(!ok, 0)) g = panel_geom_literal(TRACK, sa, ca, az, slope, sazim);
    } else {  // ATL_TRACK_TILTED_HORIZONTAL
        double sd, cd;
        lean_sincos(az - sazim, &sd, &cd);
        const double st = lean_sin(slope), ct = lean_cos(slope);
        const double num = ca * sd;
        const double den = ca * cd * st + sa * ct;
        const double q = guarded_div(num, den);
        const bool ok = den != 0.0 && __builtin_fabs(q) < 0x1.0p200;
        double cr;
        lean_sqrt_rsqrt(__builtin_fma(q, q, 1.0), &cr);
        g.cs = cr * ct;
        double ad = az - sazim;
        ad = ad > pi ? ad - 2 * pi : ad;
        ad = ad < -pi ? 2 * pi + ad : ad;
        const bool flip = (q < 0.0 && ad > 0.0) || (q > 0.0 && ad < 0.0);
        const double c = cr * (den + q * num);
        g.cosinc = flip ? -c : c;
        if constexpr (NEED_SH) g.sh = lean_sqrt(0.5 * (1.0 - g.cs));  // cs <= 1: the argument is never negative
        if (__builtin_expect(!ok, 0)) g = panel_geom_literal(TRACK, sa, ca, az, slope, sazim);
    }
    return g;
}

// irradiation on the tilted surface + Huld panel model, from sin/cos of the solar altitude and
// cos(surface_azimuth - sun_azimuth)   (irradiation.py:214-226, solar_panel_model.py:22-41)
// from the clipped cos(incidence) and the (1 +- cos(surface slope)) / 2 factors to the converter's output
template <int TAIL = kTailHuld>
ATL_HD __forceinline__ double pv_tail_core(double direct, double diffuse, double influx, double toa, double alb,
                                               double tmp, double sa, double cosinc, double hp, double hm,
                                               double sh3, const PvConst &k) {
    struct {
        double hp, hm, sh3;
    } o{hp, hm, sh3};
    const double kk = fast_div(cosinc, sa);
    const double direct_t = kk * direct;
    double G;
    [[maybe_unused]] double diffuse_t = 0.0, ground_t = 0.0;
    constexpr int PANEL = tail_panel(TAIL);
    if constexpr (tail_hay_davies(TAIL)) {  // trigon_model="other", irradiation.py:76-145, 227-245
        const double f = fill0(sqrt(guarded_div(direct, influx)));
        const double A = guarded_div(direct, toa);
        diffuse_t = ((1.0 - A) * o.hp * (1.0 + f * o.sh3) + A * kk) * diffuse;
        diffuse_t = fill0(np_max(diffuse_t, 0.0));
        ground_t = influx * alb * o.hm;
        G = direct_t + diffuse_t + ground_t;  // no fillna here: a NaN component makes the total NaN
    } else {
        diffuse_t = o.hp * diffuse;
        ground_t = alb * influx * o.hm;
        G = fill0(direct_t) + fill0(diffuse_t) + fill0(ground_t);
    }
    if constexpr (PANEL == kTailIrradiation) {  // convert_irradiation, convert.py:748-767
        return k.irr == ATL_IRR_TOTAL ? G : k.irr == ATL_IRR_DIRECT ? direct_t : k.irr == ATL_IRR_DIFFUSE ? diffuse_t : ground_t;
    }
    if constexpr (PANEL == kTailBofinger) {  // SolarPanelModel, bofinger branch: solar_panel_model.py:47-74
        const double eta_ref = k.bA + k.bB * G + k.bC * lean_log(G != 0.0 ? G : __builtin_nan(""));
        const double eta = fill0(guarded_div(eta_ref * (1.0 + k.bD * (k.bfrac * G + (tmp - k.bTstd))),
                                             1.0 + k.bDf_ta * eta_ref * G));
        const double power = G * eta * k.bscale;
        return (G >= k.bthr) ? power : 0.0;
    }
    if constexpr (PANEL == kTailThermal) {  // convert_solar_thermal, convert.py:565-574
        const double eta = k.st_c0 - k.st_c1 * fill0(guarded_div(k.st_t_store - tmp, G != 0.0 ? G : __builtin_nan("")));
        const double output = G * eta;
        return output > 0.0 ? output : 0.0;
    }
    const double T_ = (k.c_amb * tmp + k.c_irr * G) - k.r_tmod;
    const double G_ = G * k.inv_r_irr;
    double eff = 0.0;
    if (G_ > 0.0) {
        const double l = lean_log(G_);
        const double l2 = l * l;
        eff = 1.0 + k.k1 * l + k.k2 * l2 + T_ * (k.k3 + k.k4 * l + k.k5 * l2) + k.k6 * (T_ * T_);
        eff = fill0(eff);
        eff = eff < 0.0 ? 0.0 : eff;
    }
    return G_ * eff * k.inv_eff;
}

// fixed panel: cos(incidence) from the precomputed orientation factors (orientation.py:114-117,188)
template <int TAIL = kTailHuld>
ATL_HD __forceinline__ double pv_tail(double direct, double diffuse, double influx, double toa, double alb,
                                          double tmp, double sa, double ca, double cosd, const PvOri &o,
                                          const PvConst &k) {
    const double cosinc = np_max(o.ss * ca * cosd + o.cs * sa, 0.0);
    return pv_tail_core<TAIL>(direct, diffuse, influx, toa, alb, tmp, sa, cosinc, o.hp, o.hm, o.sh3, k);
}

// Datasets that store the total influx and the reflected outflux instead of a direct / diffuse split and an albedo
// (SARAH-shaped, irradiation.py:202-205, 128-139): the Reindl split with the "simple" clearsky model (:33-42) and
// albedo = outflux / influx.  pvx_cell evaluates the same expressions (both clearsky models) for the general kernel.
// ENH: the "enhanced" clearsky model (:43-64), which also reads the air temperature and the relative humidity
template <bool ENH = false>
ATL_HD __forceinline__ void reindl_split(double infl, double toa, double sa, double tmp, double rh, double *direct,
                                             double *diffuse) {
    const double influx = np_clip(infl, 0.0, toa);
    const double kk = guarded_div(influx, toa);
    const double m1 = (kk > 0.0 && kk <= 0.3) ? 1.0 : 0.0, m2 = (kk > 0.3 && kk < 0.78) ? 1.0 : 0.0,
                 m3 = (kk >= 0.78) ? 1.0 : 0.0;
    double fraction;
    if constexpr (ENH)
        fraction = m1 * fmin(1.0, 1.000 - 0.232 * kk + 0.0239 * sa - 0.000682 * tmp + 0.0195 * rh) +
                   m2 * fmin(0.97, fmax(0.1, 1.329 - 1.716 * kk + 0.267 * sa - 0.00357 * tmp + 0.106 * rh)) +
                   m3 * fmax(0.1, 0.426 * kk - 0.256 * sa + 0.00349 * tmp + 0.0734 * rh);
    else
        fraction = m1 * fmin(1.0, 1.020 - 0.254 * kk + 0.0123 * sa) +
                   m2 * fmin(0.97, fmax(0.1, 1.400 - 1.749 * kk + 0.177 * sa)) +
                   m3 * fmax(0.1, 0.486 * kk - 0.182 * sa);
    *diffuse = influx * fraction;
    *direct = influx - *diffuse;
}
ATL_HD __forceinline__ double albedo_from_outflux(double outf, double influx) {
    const double alb = fill0(guarded_div(outf, influx != 0.0 ? influx : __builtin_nan("")));
    return np_min(alb, 1.0);
}

// fixed panel, stored angles, influx / outflux dataset: the head above, then the family's usual tail
template <int TAIL = kTailHuld, bool ENH = false>
ATL_HD __forceinline__ double pv_cell_influx(double infl, double outf, double toa, double tmp, double rh, double alt,
                                                 double az, const PvOri &o, const PvConst &k) {
    double sa, ca;
    lean_sincos(alt, &sa, &ca);
    double direct, diffuse;
    reindl_split<ENH>(infl, toa, sa, tmp, rh, &direct, &diffuse);
    const double influx = direct + diffuse;
    if ((alt < k.alt_thr) || (influx <= 0.01)) return 0.0;
    const double alb = k.alb_cube ? outf : albedo_from_outflux(outf, influx);
    return pv_tail<TAIL>(direct, diffuse, influx, toa, alb, tmp, sa, ca, lean_cos(o.saz - az), o, k);
}

template <int TAIL = kTailHuld, int TRACK = ATL_TRACK_NONE>
ATL_HD __forceinline__ double pv_cell(double dir, double dif, double toa, double alb, double tmp,
                                          double alt, double az, const PvOri &o, const PvConst &k) {
    // irradiation.py:206-208
    const double direct = np_clip(dir, 0.0, toa);
    const double diffuse = np_clip(dif, 0.0, toa - direct);
    const double influx = direct + diffuse;
    // irradiation.py:251-252 (NaN compares false: a NaN altitude is not capped)
    const bool capped = (alt < k.alt_thr) || (influx <= 0.01);
    if (capped) return 0.0;  // G = 0 -> G_ = 0, eff -> 0 : 0*0*inv
    double sa, ca;
    lean_sincos(alt, &sa, &ca);
    if constexpr (TRACK != ATL_TRACK_NONE) {  // tracker: panel_geom's closed forms
        constexpr bool HD = tail_hay_davies(TAIL);
        const PanelGeom g = panel_geom<TRACK, HD>(sa, ca, az, o.slope, o.saz, k.tracking);
        // simple model: a dual-axis tracker's surface slope is the sun's zenith angle (irradiation.py:216-219);
        // Hay-Davies keeps the orientation's own slope (:227-245)
        const double cs = (!track_is(TRACK, k.tracking, ATL_TRACK_DUAL) || HD) ? g.cs : sa;
        return pv_tail_core<TAIL>(direct, diffuse, influx, toa, alb, tmp, sa, np_max(g.cosinc, 0.0), (1.0 + cs) / 2.0,
                                  (1.0 - cs) / 2.0, HD ? g.sh * g.sh * g.sh : 0.0, k);
    }
    return pv_tail<TAIL>(direct, diffuse, influx, toa, alb, tmp, sa, ca, lean_cos(o.saz - az), o, k);
}

// ---- the lane's two cells at once, common case first (fixed panel, simple trigon model, Huld panel) ------------
// When all seven inputs of a cell are finite and no intermediate overflows, the reference's NaN-aware
// clip / fillna steps reduce to plain min / max / sums: the pair is evaluated that way, both cells in ONE
// branch-free block (their dependent fp64 chains interleave; per-cell early-outs put each cell into an exec
// region of its own) without the select chains of the general routines - ~165 instead of ~230 VALU
// instructions per cell.  The arithmetic itself is pv_cell's, operation for operation, so wherever this
// evaluation is valid it returns pv_cell's bits (night skip on / off stay bit-identical).  A cell whose inputs
// or result are not finite is re-evaluated by pv_cell (never in physical data; the hostile-value tests go there).
#ifndef ATL_PV_PLAIN
#define ATL_PV_PLAIN 1
#endif
struct PvPlain {
    double r;
    bool ok;
};
// from cos(incidence) on: tilted irradiation (simple trigon model, or Hay-Davies: HD) + Huld panel, plain arithmetic
template <bool HD = false>
ATL_HD __forceinline__ PvPlain pv_tail_plain(double direct, double diffuse, double influx, double alb, double tmp,
                                                 double sa, double ca, double cosd, bool plain, bool capped,
                                                 const PvOri &o, const PvConst &k, double toa = 1.0) {
    const double inf = __builtin_inf();
    const double cosinc = __builtin_fmax(o.ss * ca * cosd + o.cs * sa, 0.0);
    const double kk = fast_div(cosinc, sa);
    const double direct_t = kk * direct;
    double diffuse_t;
    if constexpr (HD) {
        // irradiation.py:76-145 for a cell that is not capped (toa >= influx > 0.01) with direct >= 0: every quotient is a
        // well-scaled reciprocal, the root's argument lies in [0, 1]; anything else (checked by the caller's `plain` and
        // the finiteness of the result) goes to the careful routine
        // (direct < 0: the enhanced clearsky model's diffuse fraction can exceed one - the root is NaN -> 0 there, fillna)
        const bool ok = !capped && direct >= 0.0 && influx < 0x1.0p400 && toa > 0x1.0p-400 && toa < 0x1.0p400;
        const double f = lean_sqrt(fast_div(direct, ok ? influx : 1.0));
        const double A = fast_div(direct, ok ? toa : 1.0);
        diffuse_t = __builtin_fmax(((1.0 - A) * o.hp * (1.0 + f * o.sh3) + A * kk) * diffuse, 0.0);
        plain = plain && (capped || ok);
    } else {
        diffuse_t = o.hp * diffuse;
    }
    const double ground_t = HD ? influx * alb * o.hm : alb * influx * o.hm;
    const double G = direct_t + diffuse_t + ground_t;
    const double T_ = (k.c_amb * tmp + k.c_irr * G) - k.r_tmod;
    const double G_ = G * k.inv_r_irr;
    const bool pos = G_ > 0.0;
    const double l = log_core(pos ? G_ : 1.0);
    const double l2 = l * l;
    double eff = 1.0 + k.k1 * l + k.k2 * l2 + T_ * (k.k3 + k.k4 * l + k.k5 * l2) + k.k6 * (T_ * T_);
    eff = pos ? __builtin_fmax(eff, 0.0) : 0.0;
    const double r = G_ * eff * k.inv_eff;
    PvPlain out;
    out.r = capped ? 0.0 : r;
    // subnormal G_ (log_core wants a normal argument) and any overflow on the way show in these two tests
    out.ok = plain && (capped || (__builtin_fabs(r) < inf && !(pos && G_ < 0x1.0p-1022)));
    return out;
}

ATL_HD __forceinline__ PvPlain pv_cell_plain(double dir, double dif, double toa, double alb, double tmp, double alt,
                                                 double az, const PvOri &o, const PvConst &k) {
    const double inf = __builtin_inf();
    const bool plain = __builtin_fabs(dir) < inf && __builtin_fabs(dif) < inf && __builtin_fabs(toa) < inf &&
                       __builtin_fabs(alb) < inf && __builtin_fabs(tmp) < inf && __builtin_fabs(alt) < 0x1.0p30 &&
                       __builtin_fabs(az) < 0x1.0p29 && __builtin_fabs(o.saz) < 0x1.0p29;
    const double direct = __builtin_fmin(__builtin_fmax(dir, 0.0), toa);
    const double diffuse = __builtin_fmin(__builtin_fmax(dif, 0.0), toa - direct);
    const double influx = direct + diffuse;
    const bool capped = (alt < k.alt_thr) || (influx <= 0.01);
    double sa, ca;
    sincos_core(alt, &sa, &ca);
    const double cosd = cos_core(o.saz - az);
    return pv_tail_plain(direct, diffuse, influx, alb, tmp, sa, ca, cosd, plain, capped, o, k);
}

// the same with the solar position from the separable tables (pv_cell_sp's front end, plain arithmetic)
ATL_HD __forceinline__ PvPlain pv_cell_sp_plain(double dir, double dif, double toa, double alb, double tmp, double sd,
                                                    double cd, double sl, double cl, double h, double ch, const PvOri &o,
                                                    const PvAz<true> &a, const PvConst &k) {
    const double inf = __builtin_inf();
    const double sraw = sd * sl + cd * cl * ch;
    const double num = sd * cl - cd * sl * ch;
    bool plain = __builtin_fabs(dir) < inf && __builtin_fabs(dif) < inf && __builtin_fabs(toa) < inf &&
                 __builtin_fabs(alb) < inf && __builtin_fabs(tmp) < inf && __builtin_fabs(sraw) < inf &&
                 __builtin_fabs(num) < inf && __builtin_fabs(h) < inf;
    const double direct = __builtin_fmin(__builtin_fmax(dir, 0.0), toa);
    const double diffuse = __builtin_fmin(__builtin_fmax(dif, 0.0), toa - direct);
    const double influx = direct + diffuse;
    const double s = __builtin_fmin(__builtin_fmax(sraw, -1.0), 1.0);
    const bool capped = (s < k.sin_alt_thr) || (influx <= 0.01);
    // cos(alt) and its reciprocal from ONE reciprocal-square-root iteration (lean_sqrt_rsqrt: both <= 2 ulp): the quotient
    // num / cos(alt) is a product then - 8 VALU instructions less per cell than a root and a division
    const double c2 = (1.0 - s) * (1.0 + s);
    const bool cok = c2 > 0x1.0p-500;  // the sun (all but) exactly in the zenith divides literally (pv_cell_sp)
    double rca;
    const double ca = lean_sqrt_rsqrt(cok ? c2 : 1.0, &rca);
    plain = plain && (capped || cok);
    const double q = num * rca;
    const double caz = __builtin_fmin(__builtin_fmax(q, -1.0), 1.0);
    double saz = lean_sqrt((1.0 - caz) * (1.0 + caz));
    saz = (h <= 0.0) ? saz : -saz;
    return pv_tail_plain(direct, diffuse, influx, alb, tmp, s, ca, a.csaz * caz + a.ssaz * saz, plain, capped, o, k);
}

// the influx / outflux head (Reindl split with the "simple" clearsky model, albedo = outflux / influx) the same way: for
// finite inputs the masks of reindl_simple select ONE of its three candidates (adding exact zeros changes nothing), the
// clips are plain min / max, and the two divisions take the guarded reciprocal path whenever the cell is not capped
// (toa >= influx > 0.01 then).  Anything else - non-finite inputs, a toa outside the reciprocal's range - is handed to
// pv_cell_influx.
template <bool ENH = false, bool HD = false>
ATL_HD __forceinline__ PvPlain pv_cell_influx_plain(double infl, double outf, double toa, double tmp, double rh, double alt,
                                                        double az, const PvOri &o, const PvConst &k) {
    const double inf = __builtin_inf();
    const bool div_ok = toa > 0x1.0p-400 && toa < 0x1.0p400;
    bool plain = __builtin_fabs(infl) < inf && __builtin_fabs(outf) < 0x1.0p400 && div_ok && __builtin_fabs(tmp) < inf &&
                 __builtin_fabs(alt) < 0x1.0p30 && __builtin_fabs(az) < 0x1.0p29 && __builtin_fabs(o.saz) < 0x1.0p29;
    if constexpr (ENH) plain = plain && __builtin_fabs(rh) < inf;
    double sa, ca;
    sincos_core(alt, &sa, &ca);
    const double influx_c = __builtin_fmin(__builtin_fmax(infl, 0.0), toa);
    const double kk = fast_div(influx_c, div_ok ? toa : 1.0);
    double f1, f2, f3;
    if constexpr (ENH) {
        f1 = __builtin_fmin(1.0, 1.000 - 0.232 * kk + 0.0239 * sa - 0.000682 * tmp + 0.0195 * rh);
        f2 = __builtin_fmin(0.97, __builtin_fmax(0.1, 1.329 - 1.716 * kk + 0.267 * sa - 0.00357 * tmp + 0.106 * rh));
        f3 = __builtin_fmax(0.1, 0.426 * kk - 0.256 * sa + 0.00349 * tmp + 0.0734 * rh);
        // a candidate that overflowed would be a NaN / inf times a zero mask in the literal sum: hand those over
        plain = plain && __builtin_fabs(f1) < inf && __builtin_fabs(f2) < inf && __builtin_fabs(f3) < inf;
    } else {
        f1 = __builtin_fmin(1.0, 1.020 - 0.254 * kk + 0.0123 * sa);
        f2 = __builtin_fmin(0.97, __builtin_fmax(0.1, 1.400 - 1.749 * kk + 0.177 * sa));
        f3 = __builtin_fmax(0.1, 0.486 * kk - 0.182 * sa);
    }
    const double fraction = (kk >= 0.78) ? f3 : (kk > 0.3) ? f2 : (kk > 0.0) ? f1 : 0.0;
    const double diffuse = influx_c * fraction;
    const double direct = influx_c - diffuse;
    const double influx = direct + diffuse;
    const bool capped = (alt < k.alt_thr) || (influx <= 0.01);
    // an albedo variable is used as it is; else outflux / influx, at most 1 (a select, not a branch: the pair's two
    // evaluations stay one block of straight-line code)
    const double alb_derived = __builtin_fmin(fast_div(outf, capped ? 1.0 : influx), 1.0);
    const double alb = k.alb_cube ? outf : alb_derived;
    const double cosd = cos_core(o.saz - az);
    return pv_tail_plain<HD>(direct, diffuse, influx, alb, tmp, sa, ca, cosd, plain, capped, o, k, toa);
}

// what the kernels evaluate for one cell: the plain evaluation where it is valid, pv_cell otherwise (the
// kernels do the two cells of a lane side by side, PvConvT::compute; the host probe calls this)
template <int TAIL, int TRACK>
ATL_HD __forceinline__ double pv_cell_auto(double dir, double dif, double toa, double alb, double tmp, double alt,
                                               double az, const PvOri &o, const PvConst &k) {
    if constexpr (TAIL == kTailHuld && TRACK == ATL_TRACK_NONE && ATL_PV_PLAIN != 0) {
        const PvPlain p = pv_cell_plain(dir, dif, toa, alb, tmp, alt, az, o, k);
        if (p.ok) return p.r;
    }
    return pv_cell<TAIL, TRACK>(dir, dif, toa, alb, tmp, alt, az, o, k);
}

// the influx / outflux head: what the kernels evaluate for one cell (PvConvT::compute does the lane's pair side by
// side; the host probe calls this)
template <int TAIL = kTailHuld, bool ENH = false>
ATL_HD __forceinline__ double pv_cell_influx_auto(double infl, double outf, double toa, double tmp, double rh, double alt,
                                                      double az, const PvOri &o, const PvConst &k) {
    if constexpr (ATL_PV_PLAIN != 0 && (TAIL == kTailHuld || TAIL == kTailHuldHayDavies)) {
        const PvPlain p = pv_cell_influx_plain<ENH, TAIL == kTailHuldHayDavies>(infl, outf, toa, tmp, rh, alt, az, o, k);
        if (p.ok) return p.r;
    }
    return pv_cell_influx<TAIL, ENH>(infl, outf, toa, tmp, rh, alt, az, o, k);
}

// same, with the solar position computed from the separable tables instead of read:
// pv/solar_position.py:100-114.  sin(alt) = s directly, cos(alt) = sqrt(1-s^2),
// cos(az) = clip(.../cos(alt)), sin(az) = +-sqrt(1-cos^2 az) by the sign of the hour angle, so
// cos(surface_az - az) needs no inverse trig at all.  The cut alt < thr becomes s < sin(thr).
ATL_HD __forceinline__ double pv_cell_sp(double dir, double dif, double toa, double alb, double tmp,
                                             double sd, double cd, double sl, double cl, double h, double ch,
                                             const PvOri &o, const PvAz<true> &a, const PvConst &k) {
    const double direct = np_clip(dir, 0.0, toa);
    const double diffuse = np_clip(dif, 0.0, toa - direct);
    const double influx = direct + diffuse;
    const double s = np_clip(sd * sl + cd * cl * ch, -1.0, 1.0);  // :103-105
    const bool capped = (s < k.sin_alt_thr) || (influx <= 0.01);
    if (capped) return 0.0;
    const double ca = lean_sqrt((1.0 - s) * (1.0 + s));
    const double num = sd * cl - cd * sl * ch;
    double q = fast_div(num, ca);
    if (!(ca > 0x1.0p-500)) q = num / ca;  // zenith / NaN: IEEE division like the reference
    const double caz = np_clip(q, -1.0, 1.0);  // :109-113
    double saz = lean_sqrt((1.0 - caz) * (1.0 + caz));
    saz = (h <= 0.0) ? saz : -saz;  // :114  az = az if h <= 0 else 2 pi - az
    return pv_tail(direct, diffuse, influx, toa, alb, tmp, s, ca, a.csaz * caz + a.ssaz * saz, o, k);
}

// SP: in-kernel solar position; PC: per-cell orientation (else the scalar orientation is read from
// the kernel arguments = SGPRs and costs no per-lane registers)
// SKIP: night early-out - the fused kernel is k_fused_segred_night: slots in which every cell of the wave's tile
// is below the altitude cut-off are neither read nor converted (the result is exactly +0.0 whatever the other
// cubes hold, pv_cell).  Stored angles: the altitude cube decides (8 B/cell still read); in-kernel solar
// position: the (T, X) hour-angle table decides, no cube byte is read at night.
// TAIL: the Huld panel model (pv), the solar thermal collector or the plain tilted irradiation - the two
// non-panel tails exist for stored solar angles without night skip (everything else of those calls
// goes through the general kernel).
// TRACK: a tracker (pv(tracking=...)) with the Huld panel, either trigon model, scalar or per-cell orientation
// and stored solar angles - the ways trackers are used with pv(); other mixes stay general.
// HEAD: 1 / 2 = influx / outflux dataset (Reindl split with the "simple" / the "enhanced" clearsky model + albedo from
// outflux - or the dataset's albedo variable, a run-time switch: PvConst::alb_cube; stored angles, Huld panel after
// either trigon model, fixed panel): influx rides in Raw::dir, outflux (albedo) in Raw::alb,
// the diffuse slot carries the relative humidity of the enhanced model or is not loaded (48 / 56 B/cell).
template <bool SP, bool PC = false, bool SKIP = false, int TAIL = kTailHuld, int TRACK = ATL_TRACK_NONE, int HEAD = 0>
struct PvConvT {
    static_assert(HEAD >= 0 && HEAD <= 2, "HEAD: 0 direct / diffuse / albedo, 1 influx / outflux (simple clearsky model), 2 (enhanced)");
    static_assert(HEAD == 0 || (!SP && (TAIL == kTailHuld || TAIL == kTailHuldHayDavies) && TRACK == ATL_TRACK_NONE),
                  "influx head: stored angles, Huld panel, no tracker");
    static_assert(TAIL == kTailHuld || !SP, "the tails other than the Huld panel after the simple trigon model are built for stored angles");
    static_assert(TRACK == ATL_TRACK_NONE || !SP, "trackers: stored angles");
    // the MFMA-carrying instantiation (dense matrices) only for pv() with its defaults: rare options x rare matrices
    static constexpr bool kDenseOk = TAIL == kTailHuld && TRACK == ATL_TRACK_NONE && !PC && HEAD == 0;
    // per-cell series in flat order (k_cells_series_flat) where the per-cell setup is empty: one orientation for the grid, stored angles
    static constexpr bool kFlatSeries = !SP && !PC && !SKIP;
    static constexpr bool kShiftOk = !SP;  // line-aligned plans: the stored-angle converters use a slot index for nothing but slot * S
    static constexpr bool kFlatNightSeries = !SP && !PC && SKIP;  // ... and with the early-out (k_cells_series_flat_night)
    static constexpr int kDenseResident = 0;  // the operand image of dense tiles streams from L2 per sweep (resident: 64 VGPRs per
                                              // group; measured with one group: pv R = 16 / 32 3.38 / 3.74 -> 3.50 / 4.11 ms - its short
                                              // chunks leave nothing to reuse and the conversion wants the registers)
    // the members of the family compiled in atl_kernels_pvt.hip / atl_kernels_pvk.hip (tails other than the Huld
    // panel, trackers) exist for vectorised launches only; odd cell counts / unaligned cubes take the general kernel
    // ... and so do the early-out converters: their results are the bits of the converters that read every byte, whose
    // unvectorised instantiations take such launches
    // Unvectorised launches are a corner case since round 3 (a contiguous cube with an odd cell count that ends on a page
    // boundary): only pv() with its defaults - one orientation for the grid, stored or computed solar position - keeps
    // unvectorised twins; per-cell orientation, Hay-Davies and the influx head take the general kernel there too.
    static constexpr bool kVecOnly = TAIL != kTailHuld || TRACK != ATL_TRACK_NONE || SKIP || PC || HEAD != 0;
    // a vertical-axis or dual-axis tracker's geometry does not involve the sun's azimuth: that cube is not read
    // (compile-time trackers: the compiler drops the loads by itself)
    ATL_HD bool reads_azimuth() const {
        return !(TRACK == kTrackAny && (k.tracking == ATL_TRACK_VERTICAL || k.tracking == ATL_TRACK_DUAL));
    }
    atl_pv_inputs in;
    int64_t S;
    PvConst k;
    PvOri o;                     // scalar orientation
    PvAz<SP> oa;
    const double *cell_slope;    // (S) or nullptr
    const double *cell_azimuth;  // (S)
    struct SpCell {
        double sl0, cl0, sl1, cl1;  // sin/cos(lat) of the two cells
        int x0, x1;                 // grid column of the two cells
    };
    struct NoSp {};
    struct OriCell {
        PvOri o0, o1;
        PvAz<SP> a0, a1;
    };
    struct NoOri {};
    struct Cell : std::conditional_t<SP, SpCell, NoSp>, std::conditional_t<PC, OriCell, NoOri> {
        int no_cell;  // this lane owns no cell at all (tile padding); int: a bool member ends up in a scratch byte
    };
    __device__ void block_init(double *) const {}
    ATL_HD static PvOri make_ori(double slope, double azimuth) {
        PvOri r;
        lean_sincos(slope, &r.ss, &r.cs);
        r.hp = (1.0 + r.cs) / 2.0;
        r.hm = (1.0 - r.cs) / 2.0;
        r.saz = azimuth;
        r.slope = slope;
        r.sh3 = 0.0;
        if constexpr (tail_hay_davies(TAIL)) {
            const double sh = lean_sin(slope / 2.0);
            r.sh3 = sh * sh * sh;
        }
        return r;
    }
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.no_cell = !v0 && !v1;
        if constexpr (PC) {
            c.o0 = make_ori(v0 ? cell_slope[c0] : 0.0, v0 ? cell_azimuth[c0] : 0.0);
            c.o1 = make_ori(v1 ? cell_slope[c0 + 1] : 0.0, v1 ? cell_azimuth[c0 + 1] : 0.0);
            if constexpr (SP) {
                lean_sincos(c.o0.saz, &c.a0.ssaz, &c.a0.csaz);
                lean_sincos(c.o1.saz, &c.a1.ssaz, &c.a1.csaz);
            }
        }
        if constexpr (SP) {
            const int64_t a = v0 ? c0 : 0, b = v1 ? c0 + 1 : 0;
            const int64_t y0 = a / in.X, y1 = b / in.X;
            c.x0 = int(a - y0 * in.X);
            c.x1 = int(b - y1 * in.X);
            c.sl0 = in.d_sin_lat[y0];
            c.cl0 = in.d_cos_lat[y0];
            c.sl1 = in.d_sin_lat[y1];
            c.cl1 = in.d_cos_lat[y1];
        }
        return c;
    }
    static constexpr int kGroup = ATL_PV_GROUP;  // 7 x 16 B per lane per slot already; registers are the limit
    struct Raw {
        double2 dir, dif, toa, alb, tmp;
        double2 a, b;    // getter: altitude, azimuth   SP: hour angle, cos(hour angle)
        double sd, cd;   // SP: sin / cos declination of the slot
    };
    using Carry = NoCarry;
    // ---- k_fused_segred_night interface (night early-out) -----------------------------------------------------
    // a cell below the altitude cut-off converts to +0.0 whatever the other cubes hold (pv_cell: capped; a NaN
    // altitude is NOT capped)
    static constexpr bool kNightPipe = SKIP;
    // stored angles: the early-out's votes can come from a day map built once per (plan, altitude cube, cut-off):
    // atl_pv_day_map -> in.d_day_map (pv() with its defaults and Hay-Davies, either orientation kind)
    static constexpr bool kDayMap = SKIP && !SP && (TAIL == kTailHuld || TAIL == kTailHuldHayDavies) && TRACK == ATL_TRACK_NONE && HEAD == 0;
    // register budget of the fused kernels: the night kernel with stored angles and one orientation for the grid
    // fits 4 waves per SIMD
#ifndef ATL_SP_NIGHT_WAVES
#define ATL_SP_NIGHT_WAVES 3
#endif
#ifndef ATL_PV_WAVES
#define ATL_PV_WAVES 3  // the headline kernel (stored angles, one orientation, every byte read): 149 VGPRs
#endif
#ifndef ATL_SP_WAVES
#define ATL_SP_WAVES 4  // in-kernel solar position, every byte read: 115 VGPRs with the batch's values parked in LDS (kStageValues; round 3: 128 + 32 B of scratch, 138 at 3 waves)
#endif
#ifndef ATL_PV_TRKNIGHT_WAVES
#define ATL_PV_TRKNIGHT_WAVES 3  // night early-out behind a tracker: 48-112 B of scratch at 3 waves (C2 horizontal 2.74 ms) beat 2 waves (2.89 ms)
#endif
#ifndef ATL_PV_BOFTRK_WAVES
#define ATL_PV_BOFTRK_WAVES 3  // bofinger panel behind a tracker, the family's largest converters: 3 waves with 16-80 B of
                               // scratch beat 2 waves without (C2: 3.23 vs 3.50 ms horizontal, 3.62 vs 4.00 ms tilted + Hay-Davies + per-cell)
#endif
#ifndef ATL_SP_STAGE
#define ATL_SP_STAGE 1
#endif
    // in-kernel solar position, every byte read, one orientation: the batch's converted values wait in LDS
#ifndef ATL_PV_STAGE
#define ATL_PV_STAGE 0
#endif
    static constexpr bool kStageValues = (ATL_SP_STAGE != 0 && SP && !SKIP && !PC) || (ATL_PV_STAGE != 0 && !SP && !SKIP);
    static constexpr int kCubes = SP ? 5 : HEAD == 1 ? 6 : 7;  // cubes streamed per slot (fused kernel: how short a chunk may get)
    static constexpr int kMinWaves = (kNightPipe && !PC && HEAD == 0 && TAIL == kTailHuld && TRACK == ATL_TRACK_NONE) ? (SP ? ATL_SP_NIGHT_WAVES : 4)
                                     : (tail_panel(TAIL) == kTailBofinger && TRACK != ATL_TRACK_NONE) ? ATL_PV_BOFTRK_WAVES
                                     : (kNightPipe && TRACK != ATL_TRACK_NONE)                        ? ATL_PV_TRKNIGHT_WAVES
                                     : (SP && !kNightPipe && !PC)                                     ? ATL_SP_WAVES
                                     : (!SP && !kNightPipe && !PC && TAIL == kTailHuld && TRACK == ATL_TRACK_NONE && HEAD == 0) ? ATL_PV_WAVES
                                                                                                    : 3;
    // the per-cell early-out kernel behind a tracker: two waves per SIMD, no scratch (the fused kernel's three waves
    // with 48-112 B of scratch were measured against two; the per-cell kernel carries its accumulators on top)
    static constexpr int kMinWavesCells = (kNightPipe && TRACK != ATL_TRACK_NONE) ? 2 : kMinWaves;
    // stored angles: key = the slot's solar altitude.  In-kernel solar position: key = cos(hour angle) of the
    // lane's two grid columns (a (T, X) table), from which sin(altitude) follows with the slot's declination and
    // the cells' latitude - night is known before a single byte of the cubes is read.
    template <bool VEC>
    __device__ __forceinline__ double2 key_load(int64_t slot, int64_t c0, int64_t c1, const Cell &c) const {
        if constexpr (SP) {
            const int64_t hb = slot * in.X;
            return double2{in.d_cos_hour_angle[hb + c.x0], in.d_cos_hour_angle[hb + c.x1]};
        } else {
            return ld2<VEC>(in.d_solar_altitude, slot * S, c0, c1);
        }
    }
    __device__ __forceinline__ bool key_is_zero(double2 key, int64_t slot, const Cell &c) const {
        if constexpr (SP) {  // pv_cell_sp's own test, same operations: s < sin(threshold)
            const double sd = in.d_sin_dec[slot], cd = in.d_cos_dec[slot];
            const double s0 = np_clip(sd * c.sl0 + cd * c.cl0 * key.x, -1.0, 1.0);
            const double s1 = np_clip(sd * c.sl1 + cd * c.cl1 * key.y, -1.0, 1.0);
            return (s0 < k.sin_alt_thr) && (s1 < k.sin_alt_thr);
        } else {
            return (key.x < k.alt_thr) && (key.y < k.alt_thr);
        }
    }
    template <bool VEC>
    __device__ __forceinline__ Raw rest_load(int64_t slot, int64_t c0, int64_t c1, const Cell &c) const {
        const int64_t off = slot * S;
        Raw r;
        if constexpr (HEAD != 0) {
            r.dir = ld2<VEC>(in.d_influx, off, c0, c1);
            r.dif = double2{0.0, 0.0};
            if constexpr (HEAD == 2) r.dif = ld2<VEC>(in.d_humidity, off, c0, c1);
            r.alb = ld2<VEC>(in.d_outflux, off, c0, c1);
        } else {
            r.dir = ld2<VEC>(in.d_influx_direct, off, c0, c1);
            r.dif = ld2<VEC>(in.d_influx_diffuse, off, c0, c1);
            r.alb = ld2<VEC>(in.d_albedo, off, c0, c1);
        }
        r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
        r.tmp = ld2<VEC>(in.d_temperature, off, c0, c1);
        if constexpr (SP) {
            r.sd = in.d_sin_dec[slot];
            r.cd = in.d_cos_dec[slot];
            const int64_t hb = slot * in.X;
            r.a.x = in.d_hour_angle[hb + c.x0];
            r.a.y = in.d_hour_angle[hb + c.x1];
            r.b = double2{0.0, 0.0};
        } else {
            r.sd = r.cd = 0.0;
            r.a = double2{0.0, 0.0};
            r.b = double2{0.0, 0.0};
            if (reads_azimuth()) r.b = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
        }
        return r;
    }
    __device__ __forceinline__ double2 compute_keyed(const Raw &q, double2 key, bool v0, bool v1, const Cell &c,
                                                     const double *lds) const {
        Raw r = q;
        if constexpr (SP) r.b = key; else r.a = key;
        return compute(r, v0, v1, c, lds);
    }
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int i, int64_t c0, int64_t c1, const Cell &c, Carry &carry) const {
        const int64_t off = slot * S;
        Raw r;
        if constexpr (HEAD != 0) {
            r.dir = ld2<VEC>(in.d_influx, off, c0, c1);
            r.dif = double2{0.0, 0.0};
            if constexpr (HEAD == 2) r.dif = ld2<VEC>(in.d_humidity, off, c0, c1);
            r.alb = ld2<VEC>(in.d_outflux, off, c0, c1);
        } else {
            r.dir = ld2<VEC>(in.d_influx_direct, off, c0, c1);
            r.dif = ld2<VEC>(in.d_influx_diffuse, off, c0, c1);
            r.alb = ld2<VEC>(in.d_albedo, off, c0, c1);
        }
        r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
        r.tmp = ld2<VEC>(in.d_temperature, off, c0, c1);
        if constexpr (SP) {
            r.sd = in.d_sin_dec[slot];
            r.cd = in.d_cos_dec[slot];
            const int64_t hb = slot * in.X;
            r.a.x = in.d_hour_angle[hb + c.x0];
            r.a.y = in.d_hour_angle[hb + c.x1];
            r.b.x = in.d_cos_hour_angle[hb + c.x0];
            r.b.y = in.d_cos_hour_angle[hb + c.x1];
        } else {
            r.sd = r.cd = 0.0;
            r.a = ld2<VEC>(in.d_solar_altitude, off, c0, c1);
            r.b = double2{0.0, 0.0};
            if (reads_azimuth()) r.b = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
        }
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c, const double *) const {
        const PvOri &o0 = [&]() -> const PvOri & { if constexpr (PC) return c.o0; else return o; }();
        const PvOri &o1 = [&]() -> const PvOri & { if constexpr (PC) return c.o1; else return o; }();
        double2 r;
        if constexpr (SP) {
            const PvAz<true> &a0 = [&]() -> const PvAz<true> & { if constexpr (PC) return c.a0; else return oa; }();
            const PvAz<true> &a1 = [&]() -> const PvAz<true> & { if constexpr (PC) return c.a1; else return oa; }();
            if constexpr (TAIL == kTailHuld && ATL_PV_PLAIN != 0) {
                // the pair side by side like the stored-angle case below; the dark test reads every loaded value
                const double inf = __builtin_inf();
                const bool tame = __builtin_fabs(q.dir.x) < inf && __builtin_fabs(q.dir.y) < inf && __builtin_fabs(q.dif.x) < inf &&
                                  __builtin_fabs(q.dif.y) < inf && __builtin_fabs(q.toa.x) < inf && __builtin_fabs(q.toa.y) < inf &&
                                  __builtin_fabs(q.alb.x) < inf && __builtin_fabs(q.alb.y) < inf && __builtin_fabs(q.tmp.x) < inf &&
                                  __builtin_fabs(q.tmp.y) < inf && __builtin_fabs(q.a.x) < inf && __builtin_fabs(q.a.y) < inf;
                const bool dark0 = !v0 || (q.sd * c.sl0 + q.cd * c.cl0 * q.b.x) < k.sin_alt_thr - 1e-9;
                const bool dark1 = !v1 || (q.sd * c.sl1 + q.cd * c.cl1 * q.b.y) < k.sin_alt_thr - 1e-9;
                r.x = r.y = 0.0;
                if (!(dark0 && dark1 && tame)) {
                    const PvPlain p0 = pv_cell_sp_plain(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.sd, q.cd, c.sl0, c.cl0, q.a.x, q.b.x, o0, a0, k);
                    const PvPlain p1 = pv_cell_sp_plain(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.sd, q.cd, c.sl1, c.cl1, q.a.y, q.b.y, o1, a1, k);
                    r.x = p0.r;
                    r.y = p1.r;
                    if (__builtin_expect(!(p0.ok && p1.ok), 0)) {
                        r.x = pv_cell_sp(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.sd, q.cd, c.sl0, c.cl0, q.a.x, q.b.x, o0, a0, k);
                        r.y = pv_cell_sp(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.sd, q.cd, c.sl1, c.cl1, q.a.y, q.b.y, o1, a1, k);
                    }
                    r.x = v0 ? r.x : 0.0;
                    r.y = v1 ? r.y : 0.0;
                }
            } else {
                r.x = v0 ? pv_cell_sp(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.sd, q.cd, c.sl0, c.cl0, q.a.x, q.b.x, o0, a0, k) : 0.0;
                r.y = v1 ? pv_cell_sp(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.sd, q.cd, c.sl1, c.cl1, q.a.y, q.b.y, o1, a1, k) : 0.0;
            }
        } else if constexpr (HEAD != 0 && (TAIL == kTailHuld || TAIL == kTailHuldHayDavies) && ATL_PV_PLAIN != 0) {
            // the pair side by side in plain arithmetic, as the direct / diffuse case below (influx rides in dir, outflux
            // in alb, the humidity of the enhanced clearsky model in dif); the dark test reads every loaded value, so no
            // load is sunk behind it
            constexpr bool ENH = HEAD == 2, HDT = TAIL == kTailHuldHayDavies;
            const double inf = __builtin_inf();
            const bool tame = __builtin_fabs(q.dir.x) < inf && __builtin_fabs(q.dir.y) < inf && __builtin_fabs(q.toa.x) < inf &&
                              __builtin_fabs(q.toa.y) < inf && __builtin_fabs(q.alb.x) < inf && __builtin_fabs(q.alb.y) < inf &&
                              __builtin_fabs(q.tmp.x) < inf && __builtin_fabs(q.tmp.y) < inf && __builtin_fabs(q.b.x) < inf &&
                              __builtin_fabs(q.b.y) < inf && (!ENH || (__builtin_fabs(q.dif.x) < inf && __builtin_fabs(q.dif.y) < inf));
            const bool dark0 = !v0 || q.a.x < k.alt_thr, dark1 = !v1 || q.a.y < k.alt_thr;
            r.x = r.y = 0.0;
            if (!(dark0 && dark1 && tame)) {
                const PvPlain p0 = pv_cell_influx_plain<ENH, HDT>(q.dir.x, q.alb.x, q.toa.x, q.tmp.x, q.dif.x, q.a.x, q.b.x, o0, k);
                const PvPlain p1 = pv_cell_influx_plain<ENH, HDT>(q.dir.y, q.alb.y, q.toa.y, q.tmp.y, q.dif.y, q.a.y, q.b.y, o1, k);
                r.x = p0.r;
                r.y = p1.r;
                if (__builtin_expect(!(p0.ok && p1.ok), 0)) {
                    r.x = pv_cell_influx<TAIL, ENH>(q.dir.x, q.alb.x, q.toa.x, q.tmp.x, q.dif.x, q.a.x, q.b.x, o0, k);
                    r.y = pv_cell_influx<TAIL, ENH>(q.dir.y, q.alb.y, q.toa.y, q.tmp.y, q.dif.y, q.a.y, q.b.y, o1, k);
                }
                r.x = v0 ? r.x : 0.0;
                r.y = v1 ? r.y : 0.0;
            }
        } else if constexpr (HEAD != 0) {  // (ATL_PV_PLAIN=0 builds: the careful routine, cell by cell)
            r.x = v0 ? pv_cell_influx<TAIL, HEAD == 2>(q.dir.x, q.alb.x, q.toa.x, q.tmp.x, q.dif.x, q.a.x, q.b.x, o0, k) : 0.0;
            r.y = v1 ? pv_cell_influx<TAIL, HEAD == 2>(q.dir.y, q.alb.y, q.toa.y, q.tmp.y, q.dif.y, q.a.y, q.b.y, o1, k) : 0.0;
        } else if constexpr (TAIL == kTailHuld && TRACK == ATL_TRACK_NONE && ATL_PV_PLAIN != 0) {
            // (in every kernel of the family, so that night skip on / off, fused / per-cell results share their bits)
            // a pair of night cells (or a lane without cells) leaves at once: a wave in the dark skips the math.
            // The test reads ALL seven values of both cells (`tame`): were it a function of the altitude alone,
            // the compiler would sink the other six loads behind the branch - a per-lane night skip that
            // serialises the loads and makes the kernel read less than the 56 B/cell its roofline figure assumes.
            const double inf = __builtin_inf();
            const bool tame = __builtin_fabs(q.dir.x) < inf && __builtin_fabs(q.dir.y) < inf && __builtin_fabs(q.dif.x) < inf &&
                              __builtin_fabs(q.dif.y) < inf && __builtin_fabs(q.toa.x) < inf && __builtin_fabs(q.toa.y) < inf &&
                              __builtin_fabs(q.alb.x) < inf && __builtin_fabs(q.alb.y) < inf && __builtin_fabs(q.tmp.x) < inf &&
                              __builtin_fabs(q.tmp.y) < inf && __builtin_fabs(q.b.x) < inf && __builtin_fabs(q.b.y) < inf;
            const bool dark0 = !v0 || q.a.x < k.alt_thr, dark1 = !v1 || q.a.y < k.alt_thr;
            r.x = r.y = 0.0;
            if (!(dark0 && dark1 && tame)) {
                const PvPlain p0 = pv_cell_plain(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.a.x, q.b.x, o0, k);
                const PvPlain p1 = pv_cell_plain(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.a.y, q.b.y, o1, k);
                r.x = p0.r;
                r.y = p1.r;
                if (__builtin_expect(!(p0.ok && p1.ok), 0)) {
                    r.x = pv_cell<TAIL, TRACK>(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.a.x, q.b.x, o0, k);
                    r.y = pv_cell<TAIL, TRACK>(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.a.y, q.b.y, o1, k);
                }
                r.x = v0 ? r.x : 0.0;
                r.y = v1 ? r.y : 0.0;
            }
        } else {
            r.x = v0 ? pv_cell<TAIL, TRACK>(q.dir.x, q.dif.x, q.toa.x, q.alb.x, q.tmp.x, q.a.x, q.b.x, o0, k) : 0.0;
            r.y = v1 ? pv_cell<TAIL, TRACK>(q.dir.y, q.dif.y, q.toa.y, q.alb.y, q.tmp.y, q.a.y, q.b.y, o1, k) : 0.0;
        }
        return r;
    }
};
using PvConv = PvConvT<false>;
using PvConvSP = PvConvT<true>;
template <class T>
struct pv_is_sp : std::false_type {};
template <bool PC, bool SK, int TL, int TR, int HD>
struct pv_is_sp<PvConvT<true, PC, SK, TL, TR, HD>> : std::true_type {};

// ---------------------------------------------------------------------------------------
// general pv converter: every option of convert_pv / convert_irradiation / convert_solar_thermal
// (tracking modes, Hay-Davies, Reindl split, albedo from outflux, bofinger, irradiation
// quantities).  A transcription of the reference, operation by operation (lean sin/cos/log of
// atl_math.h), except for the rotating trackers' geometry, which is evaluated in closed form
// (panel_geom) - selected only when an option differs from the defaults the fast PvConvT path covers.
// ---------------------------------------------------------------------------------------
struct PvxOpt {
    int tracking, trigon, clearsky, irradiation, panel, has_influx, has_albedo;
    double bA, bB, bC, bD, bNOCT, bTstd, bTamb, bIntc, bta, bthr;
    double c0, c1, t_store;
    double r_irr;  // Huld division kept literal here
};

// host: the constant blocks the converters carry, from the C-ABI parameter struct
inline PvConst pv_const_of(const atl_pv_params *p) {
    PvConst k{p->c_temp_amb, p->c_temp_irrad, p->r_tmod, 1.0 / p->r_irradiance, p->k_1, p->k_2,
              p->k_3,        p->k_4,          p->k_5,    p->k_6,                 p->inverter_efficiency,
              p->altitude_threshold, sin(p->altitude_threshold),
              p->st_c0, p->st_c1, p->st_t_store_K, p->irradiation,
              0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0};
    if (p->panel_model == ATL_PANEL_BOFINGER) {  // the uniform sub-expressions of pvx_cell's bofinger branch, same order
        const double fraction = (p->bof_NOCT - p->bof_Tamb) / p->bof_Intc;
        const double capacity = (p->bof_A + p->bof_B * 1000.0 + p->bof_C * log(1000.0)) * 1e3;
        k.bA = p->bof_A;
        k.bB = p->bof_B;
        k.bC = p->bof_C;
        k.bD = p->bof_D;
        k.bfrac = fraction;
        k.bDf_ta = p->bof_D * fraction / p->bof_ta;
        k.bTstd = p->bof_Tstd;
        k.bthr = p->bof_threshold;
        k.bscale = p->inverter_efficiency / capacity;
    }
    k.tracking = p->tracking;
    return k;
}


inline PvxOpt pvx_opt_of(const atl_pv_params *p, bool has_influx, bool has_albedo) {
    return PvxOpt{p->tracking, p->trigon_model, p->clearsky_model, p->irradiation, p->panel_model,
                  has_influx ? 1 : 0, has_albedo ? 1 : 0,
                  p->bof_A, p->bof_B, p->bof_C, p->bof_D, p->bof_NOCT, p->bof_Tstd, p->bof_Tamb, p->bof_Intc, p->bof_ta,
                  p->bof_threshold, p->st_c0, p->st_c1, p->st_t_store_K, p->r_irradiance};
}

// SolarPanelModel, bofinger branch (solar_panel_model.py:47-74), literally; out of line like pvx_solar_literal
ATL_HD __noinline__ double pvx_bofinger_literal(double G, double tmp, double bA, double bB, double bC, double bD,
                                                    double bNOCT, double bTamb, double bIntc, double bTstd, double bta,
                                                    double bthr, double inv_eff) {
    const double nan = __builtin_nan("");
    const double fraction = (bNOCT - bTamb) / bIntc;
    const double eta_ref = bA + bB * G + bC * lean_log(G != 0.0 ? G : nan);  // <= 1 ulp; zero / negative / NaN go to libm
    const double eta = fill0(eta_ref * (1.0 + bD * (fraction * G + (tmp - bTstd))) / (1.0 + bD * fraction / bta * eta_ref * G));
    const double capacity = (bA + bB * 1000.0 + bC * log(1000.0)) * 1e3;
    const double power = G * eta * (inv_eff / capacity);
    return (G >= bthr) ? power : 0.0;
}

// TRACK / TRIGON are compile-time: they decide the instruction mix and the register footprint;
// everything else is a wave-uniform run-time switch.
template <int TRACK, int TRIGON>
ATL_HD double pvx_cell(double dir, double dif, double infl, double toa, double albv, double outf, double tmp,
                           double rh, double alt, double az, double slope, double sazim, const PvConst &k,
                           const PvxOpt &o_) {
    const double nan = __builtin_nan("");
    const PvxOpt &o = o_;
    double sa, ca;
    lean_sincos(alt, &sa, &ca);
    // ---- direct / diffuse horizontal (irradiation.py:202-208, 13-73) ------------------------
    double direct, diffuse;
    if (o.has_influx) {
        if (o.clearsky == ATL_CLEARSKY_SIMPLE)
            reindl_split<false>(infl, toa, sa, tmp, rh, &direct, &diffuse);
        else
            reindl_split<true>(infl, toa, sa, tmp, rh, &direct, &diffuse);
    } else {
        direct = np_clip(dir, 0.0, toa);
        diffuse = np_clip(dif, 0.0, toa - direct);
    }
    const double influx = direct + diffuse;
    // irradiation.py:251-252 sets the tilted irradiation to 0 here, and every panel model below maps
    // G = 0 to an output of 0 (Huld: 0 * fillna(eff); bofinger: 0 * fillna(eta) resp. the threshold;
    // solar thermal: 0 * eta; none: G) - so night / overcast cells leave before the geometry.  The
    // cells that stay have sin(alt) >= sin(threshold) > 0 and influx_toa >= influx > 0.01.
    if ((alt < k.alt_thr) || (influx <= 0.01)) return 0.0;
    // ---- SurfaceOrientation ----------------------------------------------------------------------
    const PanelGeom geom = panel_geom<TRACK, TRIGON == ATL_TRIGON_OTHER>(sa, ca, az, slope, sazim, o.tracking);
    const double cosinc = np_max(geom.cosinc, 0.0);
    // ---- albedo (irradiation.py:128-139) ---------------------------------------------------------
    double alb = albv;
    if (!o.has_albedo) {
        alb = fill0(guarded_div(outf, influx != 0.0 ? influx : nan));
        alb = np_min(alb, 1.0);
    }
    // ---- tilted irradiation ---------------------------------------------------------------------
    double direct_t, diffuse_t, ground_t, total_t;
    if constexpr (TRIGON == ATL_TRIGON_SIMPLE) {
        const double kk = guarded_div(cosinc, sa);
        const double cs = !track_is(TRACK, o.tracking, ATL_TRACK_DUAL) ? geom.cs : sa;
        direct_t = kk * direct;
        diffuse_t = (1.0 + cs) / 2.0 * diffuse;
        ground_t = alb * influx * ((1.0 - cs) / 2.0);
        total_t = fill0(direct_t) + fill0(diffuse_t) + fill0(ground_t);
    } else {
        const double f = fill0(sqrt(guarded_div(direct, influx)));
        const double A = guarded_div(direct, toa);
        const double R_b = guarded_div(cosinc, sa);
        const double sh = geom.sh;
        diffuse_t = ((1.0 - A) * ((1 + geom.cs) / 2.0) * (1.0 + f * (sh * sh * sh)) + A * R_b) * diffuse;
        diffuse_t = fill0(np_max(diffuse_t, 0.0));
        direct_t = R_b * direct;
        ground_t = influx * alb * (1.0 - geom.cs) / 2.0;
        total_t = direct_t + diffuse_t + ground_t;
    }
    double G = o.irradiation == ATL_IRR_TOTAL    ? total_t
               : o.irradiation == ATL_IRR_DIRECT ? direct_t
               : o.irradiation == ATL_IRR_DIFFUSE ? diffuse_t
                                                  : ground_t;
    // ---- panel ------------------------------------------------------------------------------------
    if (o.panel == ATL_PANEL_NONE) return G;
    if (o.panel == ATL_PANEL_HULD) {
        const double T_ = (k.c_amb * tmp + k.c_irr * G) - k.r_tmod;
        const double G_ = guarded_div(G, o.r_irr);
        const double l = (G_ > 0.0) ? lean_log(G_) : nan;
        double eff = 1.0 + k.k1 * l + k.k2 * (l * l) + T_ * (k.k3 + k.k4 * l + k.k5 * (l * l)) + k.k6 * (T_ * T_);
        eff = fill0(eff);
        eff = eff < 0.0 ? 0.0 : eff;
        return G_ * eff * k.inv_eff;
    }
    if (o.panel == ATL_PANEL_BOFINGER)
        return pvx_bofinger_literal(G, tmp, o.bA, o.bB, o.bC, o.bD, o.bNOCT, o.bTamb, o.bIntc, o.bTstd, o.bta, o.bthr, k.inv_eff);
    // solar thermal (convert.py:565-574)
    const double eta = o.c0 - o.c1 * fill0((o.t_store - tmp) / (G != 0.0 ? G : nan));
    const double output = G * eta;
    return output > 0.0 ? output : 0.0;
}

// OT: the orientation follows the sun (two more cubes, read per slot) - its own instantiation, so that the
// others do not carry the four extra registers per cell pair through a kernel that spills as it is
// The libm-heavy pieces of the general kernel, out of line and with scalar arguments: inlined, their polynomial
// constants get hoisted into VGPRs that stay occupied through the whole kernel.
// pv/solar_position.py:100-114, literally (altitude, azimuth)
ATL_HD __noinline__ double2 pvx_solar_literal(double sd, double cd, double sl, double cl, double h, double ch) {
    const double a = asin(np_clip(sd * sl + cd * cl * ch, -1.0, 1.0));
    double z = acos(np_clip((sd * cl - cd * sl * ch) / cos(a), -1.0, 1.0));
    z = (h <= 0.0) ? z : 2.0 * 3.14159265358979323846 - z;
    return double2{a, z};
}

#ifndef ATL_PVX_WAVES
#define ATL_PVX_WAVES 2  // waves per SIMD the general kernel is compiled for (256 VGPRs)
#endif
template <int TRACK, int TRIGON, bool OT = false>
struct PvxConvT {
    static constexpr int kMinWaves = ATL_PVX_WAVES;
    static constexpr bool kDenseOk = false;  // no MFMA-carrying instantiation of these (rare options x rare matrices)
    atl_pv_inputs in;
    int64_t S;
    PvConst k;
    PvxOpt o;
    double slope, azimuth;       // scalar orientation (radians)
    const double *cell_slope;    // (S), (T,S) with ori_per_time, or nullptr
    const double *cell_azimuth;  // same
    int ori_per_time;            // the orientation follows the sun: two more cubes, read per slot
    struct Cell {
        double sl0, sl1, az0, az1;   // panel slope / azimuth of the two cells
        double slat0, clat0, slat1, clat1;
        int x0, x1;
    };
    __device__ void block_init(double *) const {}
    __device__ Cell cell_setup(int64_t c0, bool v0, bool v1, const double *lds) const {
        Cell c;
        c.sl0 = c.sl1 = slope;
        c.az0 = c.az1 = azimuth;
        if (cell_slope && !OT) {
            c.sl0 = v0 ? cell_slope[c0] : 0.0;
            c.sl1 = v1 ? cell_slope[c0 + 1] : 0.0;
            c.az0 = v0 ? cell_azimuth[c0] : 0.0;
            c.az1 = v1 ? cell_azimuth[c0 + 1] : 0.0;
        }
        c.slat0 = c.clat0 = c.slat1 = c.clat1 = 0.0;
        c.x0 = c.x1 = 0;
        if (!in.d_solar_altitude) {
            const int64_t a = v0 ? c0 : 0, b = v1 ? c0 + 1 : 0;
            const int64_t y0 = a / in.X, y1 = b / in.X;
            c.x0 = int(a - y0 * in.X);
            c.x1 = int(b - y1 * in.X);
            c.slat0 = in.d_sin_lat[y0];
            c.clat0 = in.d_cos_lat[y0];
            c.slat1 = in.d_sin_lat[y1];
            c.clat1 = in.d_cos_lat[y1];
        }
        return c;
    }
    __device__ static void solar(double sd, double cd, double sl, double cl, double h, double ch, double *alt,
                                 double *az) {
        const double2 r = pvx_solar_literal(sd, cd, sl, cl, h, ch);
        *alt = r.x;
        *az = r.y;
    }
    static constexpr int kGroup = 1;
    struct OriRaw {
        double2 osl, oaz;  // panel slope / azimuth of the slot
    };
    struct NoOriRaw {};
    struct Raw : std::conditional_t<OT, OriRaw, NoOriRaw> {
        double2 dir, dif, inf, toa, alb, ouf, tmp, hum, alt, az;
    };
    using Carry = NoCarry;
    template <bool VEC>
    __device__ __forceinline__ Raw load(int64_t slot, int, int64_t c0, int64_t c1, const Cell &c, Carry &) const {
        const int64_t off = slot * S;
        const double2 zero = {0.0, 0.0};
        Raw r;
        r.dir = in.d_influx_direct ? ld2<VEC>(in.d_influx_direct, off, c0, c1) : zero;
        r.dif = in.d_influx_diffuse ? ld2<VEC>(in.d_influx_diffuse, off, c0, c1) : zero;
        r.inf = in.d_influx ? ld2<VEC>(in.d_influx, off, c0, c1) : zero;
        r.toa = ld2<VEC>(in.d_influx_toa, off, c0, c1);
        r.alb = in.d_albedo ? ld2<VEC>(in.d_albedo, off, c0, c1) : zero;
        r.ouf = in.d_outflux ? ld2<VEC>(in.d_outflux, off, c0, c1) : zero;
        r.tmp = in.d_temperature ? ld2<VEC>(in.d_temperature, off, c0, c1) : zero;
        r.hum = in.d_humidity ? ld2<VEC>(in.d_humidity, off, c0, c1) : zero;
        if constexpr (OT) {
            r.osl = ld2<VEC>(cell_slope, off, c0, c1);
            r.oaz = ld2<VEC>(cell_azimuth, off, c0, c1);
        }
        if (in.d_solar_altitude) {
            r.alt = ld2<VEC>(in.d_solar_altitude, off, c0, c1);
            r.az = ld2<VEC>(in.d_solar_azimuth, off, c0, c1);
        } else {
            const double sd = in.d_sin_dec[slot], cd = in.d_cos_dec[slot];
            const int64_t hb = slot * in.X;
            solar(sd, cd, c.slat0, c.clat0, in.d_hour_angle[hb + c.x0], in.d_cos_hour_angle[hb + c.x0], &r.alt.x, &r.az.x);
            solar(sd, cd, c.slat1, c.clat1, in.d_hour_angle[hb + c.x1], in.d_cos_hour_angle[hb + c.x1], &r.alt.y, &r.az.y);
        }
        return r;
    }
    __device__ __forceinline__ double2 compute(const Raw &q, bool v0, bool v1, const Cell &c, const double *) const {
        double2 r;
        double sl0 = c.sl0, sl1 = c.sl1, az0 = c.az0, az1 = c.az1;
        if constexpr (OT) {
            sl0 = q.osl.x;
            sl1 = q.osl.y;
            az0 = q.oaz.x;
            az1 = q.oaz.y;
        }
        r.x = v0 ? pvx_cell<TRACK, TRIGON>(q.dir.x, q.dif.x, q.inf.x, q.toa.x, q.alb.x, q.ouf.x, q.tmp.x, q.hum.x, q.alt.x, q.az.x, sl0, az0, k, o) : 0.0;
        r.y = v1 ? pvx_cell<TRACK, TRIGON>(q.dir.y, q.dif.y, q.inf.y, q.toa.y, q.alb.y, q.ouf.y, q.tmp.y, q.hum.y, q.alt.y, q.az.y, sl1, az1, k, o) : 0.0;
        return r;
    }
};

