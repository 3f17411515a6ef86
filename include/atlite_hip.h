/*
 * atlite_hip.h — C ABI of libatlite_hip.so, the MI355X (gfx950) implementation of
 * atlite's convert_and_aggregate hot path.
 *
 * The reference (PyPSA/atlite) is pure Python and has no FFI for this path; the seam it
 * offers is the Python callback protocol
 *     convert_and_aggregate(cutout, convert_func, matrix=..., ...)   atlite/convert.py:59-276
 *     aggregate_matrix(da, matrix, index)                             atlite/aggregate.py:16-35
 * Every entry point below therefore cites the reference *function* it replaces; the ctypes
 * binding a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no C++/torch types. All array arguments named d_* are DEVICE pointers
 *    (hipMalloc'd by atl_alloc, or any other HIP allocation of the same process, e.g. a
 *    torch tensor's data_ptr()). Arguments named h_* are HOST pointers.
 *  - cubes are fp64, C-contiguous (time, cell) with cell = y*X + x, x fastest
 *    (stack(spatial=["y","x"]), atlite/aggregate.py:22, atlite/convert.py:244).
 *  - every function returns ATL_OK (0) or a negative ATL_E_* code; the message for the
 *    calling thread's last failure is atl_last_error().
 *  - all work is enqueued on the context's HIP stream; results are complete after
 *    atl_sync() or a blocking atl_download().
 *  - the library never frees caller memory; atl_agg / atl_ctx handles are library-owned.
 *  - one atl_ctx is not thread-safe; distinct contexts are.
 */
#ifndef ATLITE_HIP_H
#define ATLITE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define ATL_VERSION 102 /* 0.1.2: atl_set_slot_stride removed, atl_event_record(..., 2), atl_synth_solar.ld_cells */

#define ATL_OK 0
#define ATL_E_INVALID (-1)     /* bad argument (maps to ValueError) */
#define ATL_E_HIP (-2)         /* HIP runtime failure (RuntimeError) */
#define ATL_E_NOMEM (-3)       /* allocation failure (MemoryError) */
#define ATL_E_UNSUPPORTED (-4) /* valid in the reference, not implemented here */

/* time-axis reduction applied after conversion (convert.py:51-56 `_aggregate_time`) */
#define ATL_TIME_NONE 0 /* keep the series            */
#define ATL_TIME_SUM 1  /* nan-skipping sum over time */
#define ATL_TIME_MEAN 2 /* nan-skipping mean over time */
#define ATL_TIME_SUM_COUNT 3 /* nan-skipping sum AND the number of non-NaN steps: d_out holds 2 n values,
                                [sum(n) | count(n)] - what a time shard of a multi-GPU run contributes to a
                                global sum / mean (convert.py:51-56 over the whole axis) */

typedef struct atl_ctx atl_ctx; /* device + stream + scratch */
typedef struct atl_agg atl_agg; /* indicator matrix, preprocessed and resident on device */

/* ---- library / context ------------------------------------------------------------- */
int atl_version(void);
const char *atl_last_error(void);
int atl_device_count(int *count);
/* stream: a hipStream_t to enqueue on, or NULL to let the context create its own. */
int atl_create(int device, void *stream, atl_ctx **out);
int atl_destroy(atl_ctx *ctx);
int atl_sync(atl_ctx *ctx);
int atl_device_name(atl_ctx *ctx, char *buf, size_t buflen);

/* ---- device memory (caller owns what it allocates) --------------------------------- */
int atl_alloc(atl_ctx *ctx, size_t bytes, void **d_ptr);
int atl_free(atl_ctx *ctx, void *d_ptr);
int atl_upload(atl_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* blocking */
int atl_download(atl_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* blocking */
int atl_memset(atl_ctx *ctx, void *d_dst, int byte_value, size_t bytes);
/* Pitched cubes.  The kernels stream whole 128-byte lines and own them by position, so a (T, S) cube is read fastest when
 * every slot starts on a line: S % 16 == 0, or slots PADDED to a multiple of 16 cells.  The slot stride of a call's (T, S)
 * INPUT cubes (conversions, atl_spmm_csr, atl_agg_create's tile choice, atl_nc_read_slab's output) is an ARGUMENT of the
 * call: the *_ld entry points below (ld_cells >= S; 0 = contiguous); the entry points without it take contiguous cubes.
 * (Rounds 3-5 exported atl_set_slot_stride, which made the stride mutable context state between two calls; removed in
 * round 6, ATL_VERSION 102.)  Per-cell static arrays, tables and every result stay contiguous.  atl_copy_2d moves pitched
 * blocks (kind 0 host -> device, 1 device -> host, 2 device -> device; blocking on the compute stream, or enqueued on the
 * copy stream).  The Python layer pads its own device copies of a cutout this way (atlite_amd.labeled.Dataset.device); the
 * reference has no counterpart - numpy arrays are contiguous (atlite/convert.py:198). */
int atl_copy_2d(atl_ctx *ctx, void *dst, size_t dst_pitch_bytes, const void *src, size_t src_pitch_bytes,
                size_t width_bytes, size_t height, int kind, int async_on_copy_stream);

/* ---- host-resident cutouts: overlap H2D with compute -------------------------------------------
 * A context owns a second (copy) stream.  atl_upload_async enqueues on it; events order the
 * two streams, so a caller can double-buffer time slabs of a cutout that lives in host memory
 * (or does not fit in HBM) while the conversion kernels run: see atlite_amd/streaming.py.
 * atl_host_register pins caller memory in place (DMA at PCIe rate, no staging copy).
 * which_stream: 0 = compute stream (all convert calls), 1 = copy stream.
 */
typedef struct atl_event atl_event;
/* Page-locked host memory of the library's own (hipHostMalloc): the destination of result downloads that should run at
 * the link's rate - a (shapes x time) result copied into pageable memory goes through the runtime's bounce buffers at a
 * fraction of it, which was a quarter of a warm Cutout.pv() call. */
int atl_pinned_alloc(size_t bytes, void **h_ptr);
int atl_pinned_free(void *h_ptr);
int atl_host_register(void *h_ptr, size_t bytes);
int atl_host_unregister(void *h_ptr);
int atl_upload_async(atl_ctx *ctx, void *d_dst, const void *h_src, size_t bytes); /* copy stream */
int atl_event_create(atl_ctx *ctx, atl_event **out);
int atl_event_destroy(atl_event *ev);
/* which_stream 0: the compute stream.  1: the copy stream, OBSERVED - reads whose chunks the device inflates
 * (atl_nc_read_slab) are settled first, on the host: streams the device decoder declined are decoded again by the host
 * decoders, and the verdict of a read that failed (a corrupt chunk) is returned HERE, once, with the host decoders' message
 * (it is kept by the context until a call that can return it comes along: closing the file or reusing a staging slot does
 * not lose it).  2: the copy stream as a fence only - the event is ordered behind those reads on the device, nobody waits and
 * no verdict is consumed (block recycling uses it). */
int atl_event_record(atl_ctx *ctx, atl_event *ev, int which_stream);
int atl_stream_wait_event(atl_ctx *ctx, int which_stream, atl_event *ev);
int atl_event_synchronize(atl_event *ev);

/* ---- hipGraph capture of a call sequence -------------------------------------------------
 * A launch-bound inner loop (a rank's 1/8 time shard of a year: fused kernel + k_combine, ~0.4 ms) replays as ONE
 * graph launch: everything the conversion / aggregation entry points enqueue on the context's stream between
 * atl_capture_begin and atl_capture_end is recorded instead of executed; atl_graph_launch replays it with the
 * pointers and sizes of the recorded calls.  Run the sequence once before capturing it (the scratch arena must not
 * grow inside a capture); blocking entry points (atl_upload, atl_download, atl_sync, atl_alloc) invalidate a
 * capture; the kernel brackets of atl_set_profiling are not recorded. */
typedef struct atl_graph atl_graph;
int atl_capture_begin(atl_ctx *ctx);
int atl_capture_end(atl_ctx *ctx, atl_graph **out);
int atl_graph_launch(atl_ctx *ctx, atl_graph *graph);
int atl_graph_destroy(atl_graph *graph);

/* ---- timing (HIP events on the context's stream) ----------------------------------- */
/* Bracket any sequence of calls; atl_timer_stop synchronises and returns elapsed ms. */
int atl_timer_start(atl_ctx *ctx);
int atl_timer_stop(atl_ctx *ctx, float *ms);
/* When enabled, every convert call brackets its DOMINANT kernel (the streaming
 * convert[+segment-reduce] kernel) with events; atl_last_kernel_ms synchronises and
 * returns the duration of the most recent one. */
int atl_set_profiling(atl_ctx *ctx, int enabled);
int atl_last_kernel_ms(atl_ctx *ctx, float *ms);
/* atl_set_profiling(ctx, n) with n > 1 keeps the brackets of the n most recent launches (a ring
 * of event pairs); atl_kernel_times synchronises and returns up to cap of them, oldest first, so a
 * benchmark loop times every launch of its timed region without a host sync per step. */
int atl_kernel_times(atl_ctx *ctx, float *ms, int64_t cap, int64_t *n_out);

/* ---- aggregation plan ----------------------------------------------------------------
 * Replaces: the scipy CSR matrix built in convert_and_aggregate (convert.py:213-251) and
 * consumed by aggregate_matrix (aggregate.py:16-35).  CSR N x S, rows = shapes/buses,
 * columns = cells in cutout.grid order.  Host arrays are copied; duplicates are summed.
 * Rows containing a NaN weight produce an all-NaN output row (what scipy's product gives).
 * row_len = X of the (Y, X) grid (cells per grid row; 0 if unknown): with it the plan groups
 * cells into compact w x h tiles (w*h = 128) instead of runs of 128 stacked cells, which cuts
 * the number of (tile, shape) partial rows for compact shapes.
 */
int atl_agg_create(atl_ctx *ctx, int64_t n_rows, int64_t n_cells, int64_t row_len,
                   const int64_t *h_indptr, const int32_t *h_indices, const double *h_data,
                   atl_agg **out);
/* Line-aligned plan for CONTIGUOUS (T, S) cubes whose slots do not start on 128-byte lines (S % 16 != 0 - any ERA5 cutout
 * with integer-degree bounds that a caller holds as plain C-ordered device arrays; stack(spatial=...), convert.py:244, gives
 * exactly this layout).  The offset of slot t inside its line, (t * S) % 16 cells, repeats every p = 16 / gcd(S, 16) slots:
 * the plan holds p tilings of the grid (row_len as in atl_agg_create), class r's with its tile rows on the line grid of
 * that class's slots, so that every tile row is whole lines in every slot it reads (no line is fetched by two waves, every
 * lane's 16-byte load is aligned), and the conversions walk each class's slots t = r, r + p, ... .  Same results as atl_agg_create's plan up to the order of the
 * partial sums.  Accepted by the *_convert_aggregate calls of the converters that index their cubes by slot * S only
 * (pv with stored solar angles, wind, runoff, the temperature family, atl_spmm_csr) when no slot stride is set, the cubes
 * are 8-byte aligned and the vectorised kernels apply; refused (ATL_E_ARG) otherwise.  At most 65535 / p rows.  Cubes should
 * start on a 128-byte line to profit. */
int atl_agg_create_aligned(atl_ctx *ctx, int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr,
                           const int32_t *h_indices, const double *h_data, atl_agg **out);
int atl_agg_destroy(atl_agg *agg);
/* Host-only self check of the cell-tile geometry (no GPU needed): for a grid of n_cells cells in rows
 * of row_len (0 = one flat row) and tiles tile_w cells wide (16, 32, 64 or 128), enumerate every lane
 * of every tile with the mapping the kernels use and verify that every cell is owned by exactly one
 * lane, that no lane points outside the cube, and that the plan builder's inverse mapping agrees.
 * *n_errors = 0 means consistent. */
int atl_agg_selfcheck(int64_t n_cells, int64_t row_len, int tile_w, int64_t *n_tiles, int64_t *n_owned,
                      int64_t *n_errors);
/* ... of a line-aligned plan's tilings (atl_agg_create_aligned): every alignment class owns every cell exactly once, no lane
 * points outside a slot, every lane's cell pair is 16-byte aligned and every tile row starts a 128-byte line in the slots
 * of its class, and the plan builder's inverse mapping agrees. */
int atl_agg_selfcheck_aligned(int64_t n_cells, int64_t row_len, int tile_w, int64_t *n_classes, int64_t *n_tiles,
                              int64_t *n_owned, int64_t *n_errors);
/* Host-only: builds the plan of a CSR matrix exactly as atl_agg_create does (no device involved) and verifies it
 * against the matrix - every entry at exactly one place of its shape's partial rows (duplicates summed, NaN weights
 * poison the row), partial rows of a shape in ascending tile order, the per-tile coverage masks, the MFMA operand
 * image of dense tiles.  *n_errors = 0 means consistent.  For the CPU test suite and the sanitizer build. */
int atl_agg_check_host(int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr,
                       const int32_t *h_indices, const double *h_data, int64_t *n_partial_rows,
                       int64_t *n_dense_tiles, int64_t *n_errors);
/* ... the plan atl_agg_create_aligned builds for the same matrix (every alignment class's tiling against the stacked matrix). */
int atl_agg_check_host_aligned(int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr,
                               const int32_t *h_indices, const double *h_data, int64_t *n_partial_rows,
                               int64_t *n_dense_tiles, int64_t *n_errors);
int atl_agg_info(const atl_agg *agg, int64_t *n_rows, int64_t *n_cells, int64_t *n_segments,
                 int64_t *n_partial_rows, int32_t *tile_w, int32_t *tile_h);

/* ---- the slot stride as an ARGUMENT (round 5) -----------------------------------------------------------------------
 * Every entry point whose (T, S) cubes may have padded slots has a twin that takes the stride with the call -
 * ld_cells: cells between the slots of the call's (T, S) input cubes (for atl_nc_read_slab: of the OUTPUT block), 0 =
 * contiguous.  Declared here, the originals' documentation applies. */
int atl_agg_create_ld(atl_ctx *ctx, int64_t ld_cells, int64_t n_rows, int64_t n_cells, int64_t row_len, const int64_t *h_indptr,
                      const int32_t *h_indices, const double *h_data, atl_agg **out);
int atl_spmm_csr_ld(atl_ctx *ctx, int64_t ld_cells, const atl_agg *agg, const double *d_dense, int64_t T, int64_t S, int time_agg,
                    double *d_out, int64_t ld_out);

/* ---- generic aggregation: out = M . D^T ----------------------------------------------
 * Replaces aggregate_matrix(da, matrix, index) (aggregate.py:16-35) for an arbitrary
 * already-converted cube D (T x S).  time_agg NONE: d_out is (N x T) row-major with row
 * stride ld_out >= T; SUM/MEAN: d_out is (N).
 */
int atl_spmm_csr(atl_ctx *ctx, const atl_agg *agg, const double *d_dense, int64_t T, int64_t S,
                 int time_agg, double *d_out, int64_t ld_out);

/* ---- solar PV ------------------------------------------------------------------------
 * Replaces convert_pv (convert.py:840-854) = SolarPosition (pv/solar_position.py:54-60,
 * getter branch) -> SurfaceOrientation (pv/orientation.py:104-117,188-196, tracking=None)
 * -> TiltedIrradiation (pv/irradiation.py:196-226,247-255, ERA5 direct/diffuse branch,
 * trigon_model="simple") -> SolarPanelModel/_power_huld (pv/solar_panel_model.py:12-44).
 */
typedef struct {
    const double *d_influx_direct;  /* (T,S) W m**-2 */
    const double *d_influx_diffuse; /* (T,S) */
    const double *d_influx_toa;     /* (T,S) */
    const double *d_albedo;         /* (T,S) */
    const double *d_temperature;    /* (T,S) K */
    const double *d_solar_altitude; /* (T,S) rad, or NULL -> computed in the kernel */
    const double *d_solar_azimuth;  /* (T,S) rad, or NULL                           */
    /* In-kernel SolarPosition (pv/solar_position.py:71-114) for datasets that do not store
     * the solar angles: the algorithm is separable, so the caller passes the (T)- and
     * (T,X)-sized parts (host-computed exactly as the reference writes them) and the kernel
     * does the cube-sized part.  Used iff d_solar_altitude == NULL. */
    const double *d_sin_dec;        /* (T)   sin(declination)                  :97     */
    const double *d_cos_dec;        /* (T)   cos(declination)                          */
    const double *d_hour_angle;     /* (T,X) h, radians in [-pi, pi)            :95     */
    const double *d_cos_hour_angle; /* (T,X) cos(h)                                     */
    const double *d_sin_lat;        /* (Y)   sin(radians(lat))                  :100    */
    const double *d_cos_lat;        /* (Y)   cos(radians(lat))                          */
    int64_t X;                      /* cells per grid row: needed for the tables; optional
                                     * otherwise (0 = unknown) - with it the per-cell kernels'
                                     * night early-out walks 16 x 8 tiles of the grid         */
    /* datasets without a direct/diffuse split or without albedo (irradiation.py:202-205,
     * 128-139): total influx -> Reindl split in the kernel; albedo = outflux / influx */
    const double *d_influx;         /* (T,S) or NULL (then influx_direct/diffuse are used) */
    const double *d_outflux;        /* (T,S), used iff d_albedo == NULL                    */
    const double *d_humidity;       /* (T,S), "enhanced" clearsky model only               */
    /* (Zero-initialise the struct - memset or = {0} - before filling it in: fields are added at its END from one ATL_VERSION to
     *  the next and 0 / NULL always means "not used", so a caller built against an older header keeps working only if the
     *  bytes it does not know about are zero.  d_day_map / day_map_ld came with ATL_VERSION 101.)
     * night early-out with a DAY MAP (atl_pv_day_map; round 5, line-granular since round 6): d_day_map[tile * day_map_ld + t]
     * holds one bit per 128-byte line of the plan's tile (bit j = lanes 8 j .. 8 j + 7 of the tile's wave = 16 consecutive cells),
     * set iff in time step t some cell of that line that carries a weight is above the altitude cut-off - what the early-out
     * kernel otherwise finds out by loading the tile's altitudes and voting; a zero byte = the whole tile is dark, nothing of
     * that (tile, step) is read, and in a step in which the terminator crosses the tile only its lit lines are.  Belongs to
     * ONE (aggregation plan, d_solar_altitude contents, altitude_threshold, T, slot stride); NULL = vote.  day_map_ld: a
     * multiple of 8, >= T rounded up to 8; the map 8-byte aligned.  Read by atl_pv_convert_aggregate with night_skip = 1 and
     * stored angles; ignored everywhere else. */
    const uint8_t *d_day_map;
    int64_t day_map_ld;
} atl_pv_inputs;

#define ATL_TRACK_NONE 0              /* pv/orientation.py:113-117 */
#define ATL_TRACK_HORIZONTAL 1        /* :119-131 */
#define ATL_TRACK_TILTED_HORIZONTAL 2 /* :133-168 */
#define ATL_TRACK_VERTICAL 3          /* :170-173 */
#define ATL_TRACK_DUAL 4              /* :174-175 */
#define ATL_TRIGON_SIMPLE 0           /* pv/irradiation.py:214-226 */
#define ATL_TRIGON_OTHER 1            /* Hay-Davies, :76-145, 227-236 */
#define ATL_CLEARSKY_SIMPLE 0         /* Reindl, :33-42 */
#define ATL_CLEARSKY_ENHANCED 1       /* :43-64 (temperature + humidity) */
#define ATL_IRR_TOTAL 0
#define ATL_IRR_DIRECT 1
#define ATL_IRR_DIFFUSE 2
#define ATL_IRR_GROUND 3
#define ATL_PANEL_HULD 0          /* pv/solar_panel_model.py:12-44 */
#define ATL_PANEL_BOFINGER 1      /* :47-74 */
#define ATL_PANEL_NONE 2          /* convert_irradiation (convert.py:748-767): W m**-2 */
#define ATL_PANEL_SOLAR_THERMAL 3 /* convert_solar_thermal (convert.py:550-574) */

typedef struct {
    /* Huld panel model constants (resources/solarpanel/CSi.yaml keys) */
    double c_temp_amb, c_temp_irrad, r_tmod, r_irradiance;
    double k_1, k_2, k_3, k_4, k_5, k_6;
    double inverter_efficiency;
    /* orientation in RADIANS: scalar (d_cell_slope == NULL) or one value per cell */
    double slope, azimuth;
    const double *d_cell_slope;   /* (S) or NULL */
    const double *d_cell_azimuth; /* (S) or NULL */
    double altitude_threshold;    /* radians; reference default radians(1.0) */
    /* options; all-zero = convert_pv defaults (no tracking, "simple", total, Huld).  Any other
     * combination runs the general kernel (literal transcription, full-precision libm). */
    int tracking, trigon_model, clearsky_model, irradiation, panel_model;
    /* bofinger constants (resources/solarpanel/KANENA.yaml keys) */
    double bof_A, bof_B, bof_C, bof_D, bof_NOCT, bof_Tstd, bof_Tamb, bof_Intc, bof_ta, bof_threshold;
    /* solar thermal: c0, c1, storage temperature in K (t_store + 273.15) */
    double st_c0, st_c1, st_t_store_K;
    /* 1: night early-out - a wave whose cells are all below the altitude cut-off does not read the
     * other six cubes (its result is exactly +0.0 whatever they hold).  Identical output, ~40 % less
     * HBM traffic on a year of data; 0 (default) reads every byte - what bench.py measures. */
    int night_skip;
    /* 1: d_cell_slope / d_cell_azimuth are (T,S) cubes - an orientation callback that follows the sun
     * (orientation(lon, lat, solar_position), pv/orientation.py:104-107, returning time-dependent angles);
     * evaluated by the general kernel.  0 (default): one value per cell. */
    int orientation_per_time;
} atl_pv_params;

/* per-cell output: time_agg NONE -> d_out (T x S); SUM/MEAN -> d_out (S) */
int atl_pv_convert(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                   int64_t S, int time_agg, double *d_out);
/* fused convert + aggregate: NONE -> d_out (N x T, stride ld_out); SUM/MEAN -> (N) */
int atl_pv_convert_aggregate(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p,
                             int64_t T, int64_t S, const atl_agg *agg, int time_agg,
                             double *d_out, int64_t ld_out);

/* The day map of (plan, in->d_solar_altitude, p->altitude_threshold) for T time steps: d_map is n_tiles x ld bytes
 * (n_tiles: atl_agg_info's n_segments; ld >= T / 8 + 2), every byte written.  The altitude cube is read once (8 B per
 * cell-step); cutout data does not change, so callers keep the map with the device copy of the cube
 * (atlite/pv/irradiation.py:251-252 and solar_position.py:54-60: the cut-off the map encodes). */
int atl_pv_day_map(atl_ctx *ctx, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                   const atl_agg *agg, uint8_t *d_map, int64_t ld);

/* ---- wind ----------------------------------------------------------------------------
 * Replaces convert_wind (convert.py:634-662) = extrapolate_wind_speed (wind.py:76-112)
 * followed by np.interp(v_hub, V, POW/P) (convert.py:648-649).
 */
#define ATL_WIND_NONE 0  /* d_wnd already at hub height (wind.py:76-78 fast lane) */
#define ATL_WIND_LOG 1   /* logarithmic law with roughness   (wind.py:91-102) */
#define ATL_WIND_POWER 2 /* power law with wnd_shear_exp     (wind.py:103-112) */

typedef struct {
    const double *d_wnd; /* (T,S) wind speed at from_height */
    const double *d_aux; /* roughness or wnd_shear_exp: (T,S), or (S) if aux_is_static */
    int aux_is_static;
} atl_wind_inputs;

typedef struct {
    int method; /* ATL_WIND_* */
    double to_height, from_height;
    int n_knots;          /* >= 1; 0 = no power curve: the output is the extrapolated wind speed
                           * itself, atlite.wind.extrapolate_wind_speed (wind.py:76-112) */
    const double *h_V;    /* HOST (n_knots) ascending (ties allowed, resource.py:346-355) */
    const double *h_POWn; /* HOST (n_knots) POW / P */
} atl_wind_params;

int atl_wind_convert(atl_ctx *ctx, const atl_wind_inputs *in, const atl_wind_params *p,
                     int64_t T, int64_t S, int time_agg, double *d_out);
int atl_wind_convert_aggregate(atl_ctx *ctx, const atl_wind_inputs *in,
                               const atl_wind_params *p, int64_t T, int64_t S,
                               const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);

/* ---- heat demand ---------------------------------------------------------------------
 * Replaces convert_heat_demand (convert.py:405-418): nan-skipping mean of temperature over
 * calendar-day groups of the (shifted) time axis, then a*(threshold_K - Tmean), clip(min=0),
 * + constant.  The day grouping is passed as offsets: group g = time steps
 * [d_day_ptr[g], d_day_ptr[g+1]) ; D groups; threshold_K already includes +273.15.
 * Output "time" axis is the D days.
 */
typedef struct {
    double threshold_K, a, constant;
    int64_t n_days;
    const int64_t *d_day_ptr; /* DEVICE (n_days+1) */
    int cooling; /* 0: heat demand a*(threshold - T); 1: convert_cooling_demand (convert.py:475-490) a*(T - threshold) */
} atl_heat_params;

int atl_heat_demand_convert(atl_ctx *ctx, const double *d_temperature,
                            const atl_heat_params *p, int64_t T, int64_t S, int time_agg,
                            double *d_out /* (D x S) or (S) */);
int atl_heat_demand_convert_aggregate(atl_ctx *ctx, const double *d_temperature,
                                      const atl_heat_params *p, int64_t T, int64_t S,
                                      const atl_agg *agg, int time_agg,
                                      double *d_out /* (N x D) or (N) */, int64_t ld_out);

/* ---- temperatures and heat-pump COP ----------------------------------------------------------
 * Replaces convert_temperature / convert_soil_temperature / convert_dewpoint_temperature
 * (convert.py:292-335): x = var + offset (offset = -273.15), fillna0 for the soil variant; and
 * convert_coefficient_of_performance (convert.py:338-364) when quadratic != 0:
 * d = sink_T - x; out = c0 + c1*d + c2*d*d.
 */
typedef struct {
    double offset;
    int fillna0;
    int quadratic;
    double sink_T, c0, c1, c2;
} atl_thermo_params;

int atl_thermo_convert(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T, int64_t S,
                       int time_agg, double *d_out);
int atl_thermo_convert_aggregate(atl_ctx *ctx, const double *d_var, const atl_thermo_params *p, int64_t T,
                                 int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);

/* ---- runoff --------------------------------------------------------------------------
 * Replaces convert_runoff (convert.py:1028-1034): runoff * height (height static (S)),
 * or runoff alone when d_height == NULL (weight_with_height=False).
 */
int atl_runoff_convert(atl_ctx *ctx, const double *d_runoff, const double *d_height, int64_t T,
                       int64_t S, int time_agg, double *d_out);
int atl_runoff_convert_aggregate(atl_ctx *ctx, const double *d_runoff, const double *d_height,
                                 int64_t T, int64_t S, const atl_agg *agg, int time_agg,
                                 double *d_out, int64_t ld_out);

/* ... and the twins of the conversions above with the slot stride as an argument (see atl_spmm_csr_ld) */
int atl_pv_convert_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                      int time_agg, double *d_out);
int atl_pv_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T,
                                int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
int atl_pv_day_map_ld(atl_ctx *ctx, int64_t ld_cells, const atl_pv_inputs *in, const atl_pv_params *p, int64_t T, int64_t S,
                      const atl_agg *agg, uint8_t *d_map, int64_t ld);
int atl_wind_convert_ld(atl_ctx *ctx, int64_t ld_cells, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T, int64_t S,
                        int time_agg, double *d_out);
int atl_wind_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const atl_wind_inputs *in, const atl_wind_params *p, int64_t T,
                                  int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
int atl_heat_demand_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_temperature, const atl_heat_params *p, int64_t T,
                               int64_t S, int time_agg, double *d_out);
int atl_heat_demand_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_temperature, const atl_heat_params *p,
                                         int64_t T, int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
int atl_thermo_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_var, const atl_thermo_params *p, int64_t T, int64_t S,
                          int time_agg, double *d_out);
int atl_thermo_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_var, const atl_thermo_params *p, int64_t T,
                                    int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);
int atl_runoff_convert_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_runoff, const double *d_height, int64_t T, int64_t S,
                          int time_agg, double *d_out);
int atl_runoff_convert_aggregate_ld(atl_ctx *ctx, int64_t ld_cells, const double *d_runoff, const double *d_height, int64_t T,
                                    int64_t S, const atl_agg *agg, int time_agg, double *d_out, int64_t ld_out);

/* ---- runoff post-processing on the (rows x time) result, on the device (atlite/convert.py:1046-1082) -------------
 * The result of atl_runoff_convert_aggregate is a few MB that the reference then smooths, thresholds and rescales
 * with xarray / pandas on the host; here it stays in HBM until it is final.  d: rows x T, row stride ld (time
 * contiguous, what the aggregation writes).
 * atl_rolling_mean     result.rolling(time=window, min_periods=min_periods).mean() (:1046-1052): mean of the finite
 *                      values among the last `window` steps, NaN with fewer than min_periods of them; out of place.
 *                      (+-inf counts as missing, as in pandas' rolling; xarray -> bottleneck.move_mean slides an
 *                      uncompensated sum over the whole row - here compensated sums restart every 256 steps.)
 * atl_order_statistic  the two order statistics that bracket pd.Series(values.ravel()).quantile(q) (:1054-1061,
 *                      numpy's "linear" method: virtual index q (n - 1) over the n non-NaN values): h_pair[0] =
 *                      x_(floor), h_pair[1] = x_(floor + 1) (NaN when there is none), *h_n = n, *h_rank = floor,
 *                      *h_n_le = values <= x_(floor) (the last two may be NULL).  Radix select, blocks until done.
 * atl_zero_below       result.where(result >= threshold, 0.0) (:1062), in place (NaN -> 0.0 as there).
 * atl_normalize_rows   row r *= d_ref[r] / (nan-skipping sum of row r over the steps with d_time_mask[t] != 0)
 *                      (:1064-1082: the full years the series and the reported totals share), in place. */
int atl_rolling_mean(atl_ctx *ctx, const double *d_in, int64_t rows, int64_t T, int64_t ld_in, int64_t window,
                     int64_t min_periods, double *d_out, int64_t ld_out);
int atl_order_statistic(atl_ctx *ctx, const double *d_in, int64_t rows, int64_t T, int64_t ld, double q, int64_t *h_n,
                        double *h_pair, int64_t *h_rank, int64_t *h_n_le);
int atl_zero_below(atl_ctx *ctx, double *d, int64_t rows, int64_t T, int64_t ld, double threshold);
int atl_normalize_rows(atl_ctx *ctx, double *d, int64_t rows, int64_t T, int64_t ld, const uint8_t *d_time_mask,
                       const double *d_ref);

/* ---- indicator matrix (host code, no GPU) -----------------------------------------------
 * Replaces compute_indicatormatrix (atlite/gis.py:104-145) for polygon shapes against the
 * cutout grid (cells = boxes centre +- (dx/2, dy/2), atlite/cutout.py:369-376; x, y ascending):
 * I[i,j] = area(shape_i n cell_j) / area(cell_j), cell j = y*X + x.  Shape i owns rings
 * [h_shape_ring_ptr[i], h_shape_ring_ptr[i+1]); ring r owns vertices
 * [h_ring_ptr[r], h_ring_ptr[r+1]) of h_xy (x0,y0,x1,y1,...); rings flagged in h_ring_is_hole
 * (may be NULL) subtract.  The three CSR arrays are malloc'd by the library; release each with
 * atl_host_free.
 */
int atl_indicator_polygons(int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                           const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole,
                           const double *h_xy, int64_t X, int64_t Y, double x0, double dx, double y0,
                           double dy, int64_t **out_indptr, int32_t **out_indices,
                           double **out_data);
int atl_host_free(void *p);
/* The same contract evaluated on the device (one thread per candidate cell of a shape's bounding box, exact
 * line integrals over the ring edges that overlap the cell's grid column; the host only buckets the edges by
 * column and compacts the result): SURVEY 8 f-2, second half.  Entries below 1e-12 of a cell are dropped
 * (the residue of cells the shape does not reach); otherwise equal to atl_indicator_polygons to ~1e-13. */
/* atl_indicator_polygons_device's algorithm with the candidate cells evaluated on the HOST (same source: the edge
 * bucketing, the per-edge integrals and the compaction) - no device; for the CPU test suite. */
int atl_indicator_polygons_integral_host(int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                                         const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole,
                                         const double *h_xy, int64_t X, int64_t Y, double x0, double dx,
                                         double y0, double dy, int64_t **out_indptr,
                                         int32_t **out_indices, double **out_data);
int atl_indicator_polygons_device(atl_ctx *ctx, int64_t n_shapes, const int64_t *h_shape_ring_ptr, int64_t n_rings,
                                  const int64_t *h_ring_ptr, const uint8_t *h_ring_is_hole,
                                  const double *h_xy, int64_t X, int64_t Y, double x0, double dx, double y0,
                                  double dy, int64_t **out_indptr, int32_t **out_indices,
                                  double **out_data);

/* ---- cutout files: NetCDF-4 (HDF5) ingest -------------------------------------------------------
 * Replaces `xr.open_dataset(path, chunks=...)` + dask chunk reads for the inputs of the path
 * (atlite/cutout.py:143,151-153; files are written with zlib + shuffle, atlite/data.py:139,246-248).
 * The container is parsed natively (no libhdf5); chunk payloads are inflated on host threads into
 * pinned staging, DMA'd on the context's copy stream and un-shuffled / widened to fp64 / CF-decoded
 * (`_FillValue`, `missing_value` -> NaN; `scale_factor`, `add_offset`) ON THE DEVICE.  Everything
 * except atl_nc_read_slab / atl_upload_convert_async is host-only code and works without a GPU. */
typedef struct atl_nc atl_nc;

#define ATL_NC_F32 1
#define ATL_NC_F64 2
#define ATL_NC_I8 3
#define ATL_NC_I16 4
#define ATL_NC_I32 5
#define ATL_NC_I64 6
#define ATL_NC_U8 7
#define ATL_NC_U16 8
#define ATL_NC_U32 9
#define ATL_NC_U64 10
#define ATL_NC_OTHER 0 /* strings, compounds, references: listed, not readable */

typedef struct atl_nc_var {
    int32_t ndim;    /* <= 4 reported; read functions take <= 3 */
    int32_t dtype;   /* ATL_NC_* */
    int32_t elem_size;
    int32_t big_endian;
    int64_t shape[4];
    int64_t chunk[4];       /* == shape for contiguous / compact storage */
    int32_t layout;         /* 0 compact, 1 contiguous, 2 chunked; < 0: index type not supported */
    int32_t shuffle;        /* 1 if the shuffle filter is in the pipeline */
    int32_t deflate;        /* deflate level + 1, 0 = not compressed */
    int32_t fletcher32;
    int32_t has_scale;      /* scale_factor / add_offset present */
    int32_t has_fill;       /* _FillValue present */
    int32_t has_missing;    /* missing_value present */
    int32_t reserved_;
    double scale_factor, add_offset, fill_value, missing_value;
    int64_t n_chunks;       /* chunks in the grid */
    int64_t stored_bytes;   /* bytes on disk */
} atl_nc_var;

int atl_nc_open(const char *path, atl_nc **out);
int atl_nc_close(atl_nc *f);
/* '\n'-separated names of all datasets (groups as "grp/var"); *needed = bytes incl. the NUL */
int atl_nc_list(atl_nc *f, char *buf, int64_t buflen, int64_t *needed);
int atl_nc_inquire(atl_nc *f, const char *name, atl_nc_var *info);
/* '\n'-separated dimension names of a variable (DIMENSION_LIST; else matched by length; "" unknown) */
int atl_nc_dims(atl_nc *f, const char *name, char *buf, int64_t buflen, int64_t *needed);
/* attributes; var = NULL or "" for global ones.  *needed / *n = 0 when the attribute is absent */
int atl_nc_att_text(atl_nc *f, const char *var, const char *att, char *buf, int64_t buflen, int64_t *needed);
int atl_nc_att_double(atl_nc *f, const char *var, const char *att, double *out, int64_t max_n, int64_t *n);
/* rows [start0, start0 + count0) along the first dimension, CF-decoded to fp64 on the host
 * (coordinates, static fields, tests); out holds count0 * prod(shape[1:]) doubles */
int atl_nc_read_host(atl_nc *f, const char *name, int64_t start0, int64_t count0, double *out);
/* same rows as an fp64 (count0, prod(shape[1:])) block at d_out, asynchronously; the result is visible in the order of
 * the context's COPY stream (order against the compute stream with atl_event_record(ev, 1) /
 * atl_stream_wait_event(ctx, 0, ev); the library's own copy-stream calls - atl_upload_async, atl_copy_2d(..., 1),
 * atl_upload_convert_async, later reads - are ordered behind a read whose chunks the device inflates on its staging
 * slot's stream: they make the copy stream wait for that slot first).  Two ways through the zlib streams of a chunked, deflated variable
 * (atlite/data.py:246-248 writes cutouts with zlib + shuffle):
 *  - on the DEVICE, one wavefront per chunk stream (k_inflate; round 5), when the rows asked for span at least
 *    $ATLITE_HIP_INFLATE_MIN_CHUNKS chunks (default 1024; $ATLITE_HIP_INFLATE=device: always): the host threads only
 *    pread the COMPRESSED bytes, batch by batch, into a page-locked ring, PCIe carries those INTO A RUNNING KERNEL (round 6:
 *    the launch comes first, every wave waits for its stream's batch - flags in page-locked memory that the CPU sets when a
 *    batch's DMA has completed; the call returns when the last DMA has landed), and the wave that inflated a chunk checks its
 *    Adler-32 and un-shuffles / widens / CF-decodes it into place.  A stream the device decoder declines (or whose bytes did
 *    not arrive within $ATLITE_HIP_INGEST_TIMEOUT_MS, default 20 s) is decoded by the host decoders before anyone can
 *    observe the copy stream (atl_event_record(ev, 1), the slot's next use, atl_nc_close): a corrupt stream is reported
 *    THERE, with the host decoders' message;
 *  - on host threads ($ATLITE_HIP_INFLATE=host, =zlib, or few chunks): returns after the inflate, DMA + decode enqueued.
 * n_threads <= 0: $ATLITE_HIP_IO_THREADS, else min(2 x usable CPUs (cgroup quota aware), 128). */
int atl_nc_read_slab(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0,
                     double *d_out, int n_threads);
/* chunks of this context's atl_nc_read_slab calls so far: inflated on the device / on host threads / declined by the device
 * decoder and decoded again on the host (settles pending reads first) */
/* The same rows of n_vars variables at once (what one conversion reads: a slab of the pv inputs): when every stored chunk
 * of the group is a plain zlib stream, the rows are cut into JOBS of ~2048 streams across all the variables
 * ($ATLITE_HIP_INGEST_JOB), each one k_inflate launch on its own staging slot and stream - the device holds 8192 streams
 * at a time (round 6: 4.7 kB of LDS and 64 VGPRs per stream), a single variable's rows rarely bring that many - and the
 * jobs are pipelined: pread + DMA of job k+1 run while job k is inflated and job k-1 unpacked.  Otherwise exactly n_vars
 * atl_nc_read_slab calls. */
int atl_nc_read_slabs(atl_ctx *ctx, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0,
                      double *const *d_outs, int n_threads);
int atl_nc_read_slabs_ld(atl_ctx *ctx, int64_t ld_cells, atl_nc *f, int n_vars, const char *const *names, int64_t start0,
                         int64_t count0, double *const *d_outs, int n_threads);
/* atl_nc_read_slab with the OUTPUT block's slot stride as an argument (see atl_spmm_csr_ld) */
int atl_nc_read_slab_ld(atl_ctx *ctx, int64_t ld_cells, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out,
                        int n_threads);
int atl_nc_ingest_stats(atl_ctx *ctx, int64_t *device_chunks, int64_t *host_chunks, int64_t *redone);
/* device-inflate reads so far, accumulated: ms5 = {preads of the compressed bytes (wall clock), DMAs first to last, k_inflate
 * including its waits for the DMAs and its fused Adler-32 + unpack, the NUMBER of stream segments decoded side by side (a count
 * in round 5's checksum-pass slot: reads of few, long streams - atlite's own (time = 100, y, x) chunks - are decoded block by
 * block, $ATLITE_HIP_INFLATE_SPLIT; 0 while streams are decoded whole), k_unpack of never-written chunks} - HIP events on the
 * slot's streams; the first three overlap inside ONE fed launch (round 6), so their
 * sum exceeds the wall time - and the bytes in / out of k_inflate */
int atl_nc_ingest_times(atl_ctx *ctx, double *ms5, int64_t *compressed_bytes, int64_t *inflated_bytes);
/* host array of a narrower dtype (what xarray hands over for a float32 cutout) -> fp64 on the
 * device through the same staging + decode kernel; halves the PCIe bytes of atl_upload_async */
int atl_upload_convert_async(atl_ctx *ctx, double *d_dst, const void *h_src, int dtype, int64_t n);
/* the same for a (rows, cols) host block into device rows ld_cells doubles apart (padded slots) */
int atl_upload_convert_2d_async(atl_ctx *ctx, double *d_dst, int64_t ld_cells, const void *h_src, int dtype,
                                int64_t rows, int64_t cols);

/* ---- multi-GPU (RCCL over xGMI) ------------------------------------------------------------------
 * One process (and one atl_ctx) per GPU, the TIME axis sharded: rank r converts and aggregates its
 * own time slab; the small (shapes x time) result is reassembled with one collective.  The
 * reference has no distributed path (its parallelism is dask threads over time chunks,
 * atlite/cutout.py:143).  RCCL is opened lazily - single-GPU use does not need it.
 *   rank 0: atl_comm_unique_id(id) -> ship the 128 bytes to every rank (MPI, sockets, a file ...)
 *   all   : atl_comm_init(ctx, n_ranks, rank, id, &comm)
 * atl_allgather_time: every rank holds a contiguous (N x T_r) block, same T_r everywhere; on return
 * d_out (N x n_ranks*T_r, row stride ld_out) holds the blocks in rank order on every rank.
 * atl_allreduce_sum: in-place sum over ranks (time sums and counts of aggregate_time="sum"/"mean").
 * Both are enqueued on the context's stream.
 */
#define ATL_COMM_ID_BYTES 128
typedef struct atl_comm atl_comm;
int atl_comm_unique_id(void *h_id128);
int atl_comm_init(atl_ctx *ctx, int n_ranks, int rank, const void *h_id128, atl_comm **out);
int atl_comm_destroy(atl_comm *comm);
int atl_allgather_time(atl_comm *comm, const double *d_local, int64_t N, int64_t T_r, double *d_out,
                       int64_t ld_out);
int atl_allreduce_sum(atl_comm *comm, double *d_buf, int64_t n);
/* Ragged all-gather along time: rank r contributes (N x h_lens[r]) (d_local, row stride T_r = h_lens[own rank]);
 * d_out (N x sum h_lens, row stride ld_out) on every rank.  h_lens: n_ranks host values, the same on all ranks. */
int atl_allgather_time_v(atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens, double *d_out,
                         int64_t ld_out);
/* The same collective on the COMMUNICATOR'S OWN stream, ordered after everything enqueued on the context's stream so
 * far: a step's all-gather and placement run behind the NEXT step's kernel (the reference's time chunks are
 * independent, atlite/aggregate.py:21-32, so nothing but the result buffers orders two steps).  *ticket names the
 * collective; atl_comm_wait(comm, ticket) makes the context's stream wait for it (and every earlier one) - call it
 * before a kernel overwrites that collective's d_local or reads its d_out; atl_comm_sync blocks the host until the
 * communicator's stream has drained.  The staging buffers belong to the communicator. */
int atl_allgather_time_v_async(atl_comm *comm, const double *d_local, int64_t N, const int64_t *h_lens, double *d_out,
                               int64_t ld_out, int64_t *ticket);
int atl_comm_wait(atl_comm *comm, int64_t ticket);
int atl_comm_sync(atl_comm *comm);
/* One thread, several devices: the RCCL communicators of ctxs[0..n_ranks) (distinct devices) in ONE ncclGroupStart /
 * ncclGroupEnd bracket, rank r on ctxs[r] - what a single-process host (atlite_amd.multigpu) uses instead of N threads
 * each calling atl_comm_init (concurrent ncclCommInitRank calls on distinct devices of one process can wait for each
 * other for good).  atl_comm_init itself gives up after $ATLITE_HIP_COMM_TIMEOUT_S (120 s) when the other ranks do
 * not join. */
int atl_comm_init_all(atl_ctx *const *ctxs, int n_ranks, atl_comm **out);
/* What the communicator itself says: ranks in it (ncclCommCount), this rank, its device ordinal (ncclCommCuDevice),
 * the transport.  Any out pointer may be NULL. */
#define ATL_COMM_RCCL 0
#define ATL_COMM_LOCAL 1
int atl_comm_info(atl_comm *comm, int *n_ranks, int *rank, int *device, int *transport);
/* In-process transport: the ranks are host threads of one process (one atl_ctx each, on distinct devices or - for
 * tests on a one-GPU box - sharing one).  Every rank pulls its peers' blocks with peer copies on its own stream
 * (point-to-point xGMI reads, no ring), ordered by events; the host rendezvous inside a collective gives up after
 * $ATLITE_HIP_COMM_TIMEOUT_S (120 s) and fails every waiting rank instead of hanging when a peer never arrives.
 *   once : atl_comm_group_create(n_ranks, &group)
 *   all  : atl_comm_init_local(ctx, group, rank, &comm)      (all ranks inside the call at once, like atl_comm_init)
 * The collectives above work on either kind of communicator; atl_allreduce_sum adds in rank order here (the same
 * bits on every rank).  atl_comm_abort wakes the group's waiting ranks with an error (a rank that failed before its
 * collective calls it on the way out).  Destroy the communicators before the group. */
typedef struct atl_comm_group atl_comm_group;
int atl_comm_group_create(int n_ranks, atl_comm_group **out);
int atl_comm_group_destroy(atl_comm_group *group);
int atl_comm_init_local(atl_ctx *ctx, atl_comm_group *group, int rank, atl_comm **out);
int atl_comm_abort(atl_comm *comm);
/* Host instantiation of the placement step of atl_allgather_time_v (the kernel's own index function walked over the
 * kernel's own grid): h_gathered = [rank][N][max h_lens] as the collective delivers it -> h_out (N x sum h_lens, row
 * stride ld_out).  No device needed: the CPU tests pin the placement of ragged ranks with it. */
int atl_gather_place_v_host(const double *h_gathered, int n_ranks, int64_t N, const int64_t *h_lens, double *h_out,
                            int64_t ld_out);

/* ---- diagnostics ------------------------------------------------------------------------
 * Evaluates the kernels' lean fp64 math (atl_math.h) elementwise, for accuracy tests:
 * fn 0 sin, 1 cos, 2 log: d_in (n) -> d_out (n);  3 sincos: d_out (2n) = sin | cos;
 * 4 fast_div: d_in (2n) = a | b -> d_out (n) = a / b;  5 table-driven log (positive normal x);
 * 7 lean_sqrt (0 <= x < 2^500);  8 lean_sqrt_rsqrt (2^-500 < x < 2^500): d_out (2n) = sqrt | 1 / sqrt.
 */
int atl_math_probe(atl_ctx *ctx, int fn, const double *d_in, int64_t n, double *d_out);
/* the same routines (same source, compiled for the host) on host arrays: lets the CPU test suite check
 * the polynomials, argument reductions and special cases without a GPU.  fn as above, plus 6 =
 * guarded_div (a | b -> a / b with IEEE behaviour for zeros / infinities / NaN / denormals). */
int atl_math_probe_host(int fn, const double *h_in, int64_t n, double *h_out);
/* The pv conversion of n independent cells on the HOST through the kernels' own per-cell routines
 * (same source, host build) - for the CPU test suite.  h_in: 12 arrays of n doubles (NULL = variable
 * absent): influx_direct, influx_diffuse, influx, influx_toa, albedo, outflux, temperature, humidity,
 * solar_altitude, solar_azimuth, panel slope [rad], panel azimuth [rad].  family 0: the fast kernel family
 * (tail and tracker chosen from the options like the dispatcher does), 1: the general kernel's routine. */
int atl_pv_probe_host(const atl_pv_params *p, int family, int64_t n, const double *const *h_in, double *h_out);
/* The wind conversion (hub-height extrapolation + power curve) of n independent cell-steps on the HOST
 * through the wind converter's own routines (same source, host build; h_aux = roughness or shear exponent,
 * NULL for ATL_WIND_NONE). */
int atl_wind_probe_host(const atl_wind_params *p, int64_t n, const double *h_wnd, const double *h_aux, double *h_out);
/* np.interp(x, V, F) through the padded-table search the wind kernels use (same source, host build):
 * bit-for-bit numpy at knots (the upper one of repeated knots), outside the range, at +-inf and for NaN;
 * inside an interval one FMA replaces numpy's multiply-add (<= 1 ulp apart). */
int atl_wind_interp_host(const double *h_V, const double *h_F, int n_knots, const double *h_x, int64_t m,
                         double *h_out);
/* zlib-wrapped DEFLATE stream -> exactly dst_n bytes on the host.  which = 0: the library's fast
 * decoder alone (ATL_E_UNSUPPORTED if it declines the stream), 1: zlib alone, 2: the product
 * combination (fast, zlib on any doubt), 3: the device decoder's serial half run on the host, 4: its
 * segment scheme (a stream's DEFLATE blocks decoded side by side: block finder, count pass, chain,
 * decode pass with markers, resolve) run on the host - *ns = the number of segments then.
 * *ns = wall time of the decode otherwise.  Host only. */
int atl_inflate_probe(const void *h_src, size_t src_n, void *h_dst, size_t dst_n, int which, int64_t *ns);

/* ---- synthetic ERA5-shaped inputs (bench/test tooling) ----------
 * Fills device cubes with a stateless splitmix64-hash field so that any (t, cell) slice can
 * be regenerated or downloaded for the oracle.  Not part of the reference's interface.
 */
#define ATL_SYN_UNIFORM 0  /* lo + (hi-lo)*u                          p0=lo p1=hi */
#define ATL_SYN_RAYLEIGH 1 /* p0*sqrt(-ln(1-u))*(2/sqrt(pi))          p0=mean     */
#define ATL_SYN_EXPLOG 2   /* exp(ln(p0)+u*ln(p1)) static per cell    p0,p1       */
#define ATL_SYN_NEGLOG 3   /* -p0*ln(1-u)                             p0=scale    */
int atl_synth_field(atl_ctx *ctx, int kind, uint64_t seed, uint64_t var_id, double p0, double p1,
                    int per_cell_static, int64_t T, int64_t S, double *d_out);
/* solar position + consistent radiation/temperature fields from per-time and per-(time,x)
 * host-precomputed tables (pv/solar_position.py:86-114 structure). */
typedef struct {
    const double *d_sin_dec; /* (T) */
    const double *d_cos_dec; /* (T) */
    const double *d_h;       /* (T,X) hour angle rad */
    const double *d_lat_rad; /* (Y) */
    const double *d_tseason; /* (T) seasonal+diurnal temperature term, K */
    int64_t X, Y;
    uint64_t seed;
    int64_t ld_cells; /* cells between the slots of the OUTPUT cubes (>= S), 0 = contiguous */
} atl_synth_solar;
int atl_synth_pv_inputs(atl_ctx *ctx, const atl_synth_solar *s, int64_t T, int64_t S,
                        double *d_influx_direct, double *d_influx_diffuse, double *d_influx_toa,
                        double *d_albedo, double *d_temperature, double *d_solar_altitude,
                        double *d_solar_azimuth);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ATLITE_HIP_H */
