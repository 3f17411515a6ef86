// Cutout-file ingest: NetCDF-4 (HDF5) chunks -> fp64 (time, cell) blocks in HBM.
// Replaces xr.open_dataset(path, chunks=...) + dask chunk reads for the inputs of the hot path
// (atlite/cutout.py:143,151-153; atlite/data.py:139,246-248).  Part of libatlite_hip.so (gfx950).
//
// Split of the work:
//   host threads : locate chunks (atl_h5.cpp), strip fletcher32, zlib inflate - the only inherently
//                  serial step - straight from the file mapping into pinned staging
//   copy stream  : one DMA per call, then k_unpack: byte un-shuffle, endian swap, widen to fp64,
//                  _FillValue / missing_value -> NaN, scale_factor / add_offset, scatter of the chunk
//                  grid into the (rows, cells) block the conversion kernels read
// so the CPU never touches the inflated bytes again and PCIe carries the narrow on-disk dtype.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include <unistd.h>

#include "atl_h5.h"
#include "atl_internal.h"
#include "atl_inflate_dev.h"

using namespace atl;
using atl::h5::Attribute;
using atl::h5::Dataset;
using atl::h5::TypeClass;

struct atl_nc {
    h5::File file;
};

namespace {

// ---- dtype helpers ---------------------------------------------------------------------------------
int dtype_code(const h5::Datatype &t) {
    if (t.cls == TypeClass::Float) return t.size == 4 ? ATL_NC_F32 : t.size == 8 ? ATL_NC_F64 : ATL_NC_OTHER;
    if (t.cls == TypeClass::Fixed) {
        switch (t.size) {
            case 1: return t.is_signed ? ATL_NC_I8 : ATL_NC_U8;
            case 2: return t.is_signed ? ATL_NC_I16 : ATL_NC_U16;
            case 4: return t.is_signed ? ATL_NC_I32 : ATL_NC_U32;
            case 8: return t.is_signed ? ATL_NC_I64 : ATL_NC_U64;
        }
    }
    return ATL_NC_OTHER;
}

int dtype_size(int code) {
    switch (code) {
        case ATL_NC_I8: case ATL_NC_U8: return 1;
        case ATL_NC_I16: case ATL_NC_U16: return 2;
        case ATL_NC_F32: case ATL_NC_I32: case ATL_NC_U32: return 4;
        case ATL_NC_F64: case ATL_NC_I64: case ATL_NC_U64: return 8;
    }
    return 0;
}

// raw little-endian bits -> double, by dtype code (shared by host and device)
__host__ __device__ inline double bits_to_double(uint64_t b, int code) {
    switch (code) {
        case ATL_NC_F32: {
            uint32_t u = uint32_t(b);
            float f;
            memcpy(&f, &u, 4);
            return double(f);
        }
        case ATL_NC_F64: {
            double d;
            memcpy(&d, &b, 8);
            return d;
        }
        case ATL_NC_I8: return double(int8_t(b));
        case ATL_NC_I16: return double(int16_t(b));
        case ATL_NC_I32: return double(int32_t(b));
        case ATL_NC_I64: return double(int64_t(b));
        case ATL_NC_U8: return double(uint8_t(b));
        case ATL_NC_U16: return double(uint16_t(b));
        case ATL_NC_U32: return double(uint32_t(b));
        default: return double(b);
    }
}

struct Decode {  // CF conventions as xarray applies them: mask first, then scale
    int dtype = 0, esize = 0, big_endian = 0;
    int has_fill = 0, has_missing = 0, has_scale = 0;
    double fill = 0, missing = 0, scale = 1, offset = 0;
};

__host__ __device__ inline double cf_decode(double v, const Decode &d) {
#pragma clang fp contract(off)  // xarray: data *= scale_factor; data += add_offset - two roundings, no FMA
    if ((d.has_fill && v == d.fill) || (d.has_missing && v == d.missing)) return __builtin_nan("");
    if (!d.has_scale) return v;
    const double scaled = v * d.scale;
    return scaled + d.offset;
}

bool attr_double(const h5::File &f, const Attribute *a, double *out, int64_t max_n, int64_t *n) {
    *n = 0;
    if (!a) return false;
    const int code = dtype_code(a->type);
    if (code == ATL_NC_OTHER) return false;
    const int64_t cnt = int64_t(a->nbytes / a->type.size);
    for (int64_t i = 0; i < cnt && i < max_n; ++i) {
        uint64_t b = 0;
        for (uint32_t k = 0; k < a->type.size; ++k) {
            const uint32_t src = a->type.big_endian ? a->type.size - 1 - k : k;
            b |= uint64_t(a->data[i * a->type.size + src]) << (8 * k);
        }
        out[i] = bits_to_double(b, code);
    }
    *n = cnt;
    return cnt > 0;
}

Decode decode_of(const h5::File &f, const Dataset &d) {
    Decode dc;
    dc.dtype = dtype_code(d.type);
    dc.esize = int(d.type.size);
    dc.big_endian = d.type.big_endian;
    double v;
    int64_t n;
    if (attr_double(f, d.attr("_FillValue"), &v, 1, &n)) dc.has_fill = 1, dc.fill = v;
    if (attr_double(f, d.attr("missing_value"), &v, 1, &n)) dc.has_missing = 1, dc.missing = v;
    if (attr_double(f, d.attr("scale_factor"), &v, 1, &n)) dc.has_scale = 1, dc.scale = v;
    if (attr_double(f, d.attr("add_offset"), &v, 1, &n)) dc.has_scale = 1, dc.offset = v;
    return dc;
}

// ---- variable geometry, normalised to 3 dims (rows, d1, d2) ----------------------------------------------
struct Geometry {
    int64_t shape[3] = {1, 1, 1};
    int64_t chunk[3] = {1, 1, 1};
    int64_t grid[3] = {1, 1, 1};
    int64_t chunk_elems = 1, row_elems = 1;
};

int geometry_of(const Dataset &d, Geometry *g, const char *who) {
    const int rank = int(d.shape.size());
    ATL_REQUIRE(rank >= 1 && rank <= 3, "%s: variable '%s' has %d dimensions; 1 to 3 are supported", who,
                d.name.c_str(), rank);
    if (dtype_code(d.type) == ATL_NC_OTHER) {
        set_error("%s: variable '%s' is not numeric", who, d.name.c_str());
        return ATL_E_UNSUPPORTED;
    }
    if (d.layout < 0 || d.layout > 2) {
        set_error("%s: variable '%s' uses a storage layout / chunk index this reader does not support "
                  "(extensible-array or v2-B-tree chunk index, or a pre-1.6 layout); rewrite the file with "
                  "libver bounds <= v18 (NetCDF-4 default)", who, d.name.c_str());
        return ATL_E_UNSUPPORTED;
    }
    for (int i = 0; i < rank; ++i) {
        g->shape[i] = int64_t(d.shape[i]);
        g->chunk[i] = int64_t(d.chunk[i]);
        g->grid[i] = d.layout == 2 ? int64_t(d.grid[i]) : 1;
    }
    g->chunk_elems = g->chunk[0] * g->chunk[1] * g->chunk[2];
    g->row_elems = g->shape[1] * g->shape[2];
    return ATL_OK;
}

// ---- device side ------------------------------------------------------------------------------------
struct UnpackDesc {
    int64_t src_off;   // byte offset of the chunk payload in the raw buffer
    int64_t org[3];    // element origin of the chunk in the variable
    int64_t dim[3];    // chunk extent (elements, as stored)
    int32_t shuffled;  // payload is byte-shuffled (HDF5 shuffle filter)
    int32_t missing;   // chunk was never written -> _FillValue / NaN
};

struct UnpackParams {
    int64_t shape1, shape2;   // trailing dims of the variable
    int64_t ld;               // elements between the rows of the output block (>= shape1 * shape2)
    int64_t r0, r1;           // rows wanted
    Decode dec;
};

// one element of a chunk: raw little- or big-endian bytes (byte-shuffled or not) -> CF-decoded double
__device__ __forceinline__ double unpack_value(const uint8_t *__restrict__ src, int64_t n, int64_t e, int shuffled, const Decode &dec) {
    const int es = dec.esize;
    uint64_t b = 0;
    if (shuffled) {
        // byte k of element e sits at k*n + e: es coalesced byte streams
        for (int k = 0; k < es; ++k) {
            const int dstk = dec.big_endian ? es - 1 - k : k;
            b |= uint64_t(src[int64_t(k) * n + e]) << (8 * dstk);
        }
    } else if (es == 4 && !dec.big_endian) {
        b = reinterpret_cast<const uint32_t *>(src)[e];
    } else if (es == 8 && !dec.big_endian) {
        b = reinterpret_cast<const uint64_t *>(src)[e];
    } else {
        for (int k = 0; k < es; ++k) {
            const int dstk = dec.big_endian ? es - 1 - k : k;
            b |= uint64_t(src[e * es + k]) << (8 * dstk);
        }
    }
    return cf_decode(bits_to_double(b, dec.dtype), dec);
}

__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ raw, const UnpackDesc *__restrict__ desc,
                                                UnpackParams p, double *__restrict__ out) {
    const UnpackDesc d = desc[blockIdx.y];
    const int64_t n = d.dim[0] * d.dim[1] * d.dim[2];
    const int64_t plane = d.dim[1] * d.dim[2];
    const uint8_t *src = raw + d.src_off;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t ct = e / plane, rem = e - ct * plane;
        const int64_t cy = rem / d.dim[2], cx = rem - cy * d.dim[2];
        const int64_t t = d.org[0] + ct, y = d.org[1] + cy, x = d.org[2] + cx;
        if (t < p.r0 || t >= p.r1 || y >= p.shape1 || x >= p.shape2) continue;
        double v;
        if (d.missing) {
            v = p.dec.has_fill ? cf_decode(p.dec.fill, p.dec) : __builtin_nan("");
        } else {
            v = unpack_value(src, n, e, d.shuffled, p.dec);
        }
        out[(t - p.r0) * p.ld + y * p.shape2 + x] = v;
    }
}

// Byte-shuffled little-endian chunks of 2- or 4-byte elements whose rows are a multiple of four elements long - what atlite
// writes (float32, zlib + shuffle: atlite/data.py:246-248; packed int16 from the CDS): a lane takes FOUR consecutive elements,
// i.e. one aligned word from each of the ES byte planes (all in flight together), and stores four doubles.  The element-wise
// loop below waits for one byte load after the other (the element size is a run-time value there): 29 ms per 960 kB chunk in a
// lone wave, against ~1 ms this way.
template <int ES>
__device__ __forceinline__ void wave_unpack4(const uint8_t *__restrict__ src, uint32_t n, const UnpackDesc &d, const UnpackParams &p,
                                             double *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const uint32_t dim2 = uint32_t(d.dim[2]), plane = uint32_t(d.dim[1]) * dim2, groups = n / 4;
    const uint32_t *pl[ES];
#pragma unroll
    for (int k = 0; k < ES; ++k) pl[k] = reinterpret_cast<const uint32_t *>(src + size_t(k) * n);
#pragma unroll 2
    for (uint32_t g = lane; g < groups; g += 64) {
        uint32_t w[ES];
#pragma unroll
        for (int k = 0; k < ES; ++k) w[k] = pl[k][g];
        const uint32_t e = 4 * g, ct = e / plane, rem = e - ct * plane, cy = rem / dim2, cx = rem - cy * dim2;
        const int64_t t = d.org[0] + ct, y = d.org[1] + cy, x0 = d.org[2] + cx;
        if (t < p.r0 || t >= p.r1 || y >= p.shape1) continue;
        double *o = out + (t - p.r0) * p.ld + y * p.shape2 + x0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t b = 0;
#pragma unroll
            for (int k = 0; k < ES; ++k) b |= ((w[k] >> (8 * j)) & 0xFFu) << (8 * k);
            const double v = cf_decode(bits_to_double(uint64_t(b), p.dec.dtype), p.dec);
            if (x0 + j < p.shape2) o[j] = v;
        }
    }
}

// The same by ONE wavefront, row by row of the chunk (no per-element division): what a k_inflate wave does with the chunk it has
// just inflated (round 6) - the chunk's bytes are still in the L2, and a separate k_unpack pass over a year of C2 was 19-27 ms
// at the end of a read that nothing could overlap with.
__device__ __forceinline__ void wave_unpack(const uint8_t *__restrict__ raw, const UnpackDesc &d, const UnpackParams &p, double *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const int64_t n = d.dim[0] * d.dim[1] * d.dim[2];
    const uint8_t *src = raw + d.src_off;
    if (d.shuffled && !p.dec.big_endian && (d.dim[2] & 3) == 0 && n < (int64_t(1) << 30) && (p.dec.esize == 4 || p.dec.esize == 2)) {
        if (p.dec.esize == 4)
            wave_unpack4<4>(src, uint32_t(n), d, p, out);
        else
            wave_unpack4<2>(src, uint32_t(n), d, p, out);
        return;
    }
    const uint32_t rows = uint32_t(d.dim[0] * d.dim[1]), dim1 = uint32_t(d.dim[1]), dim2 = uint32_t(d.dim[2]);
    uint32_t ct = 0, cy = 0;
    for (uint32_t r = 0; r < rows; ++r) {
        const int64_t t = d.org[0] + ct, y = d.org[1] + cy;
        if (t >= p.r0 && t < p.r1 && y < p.shape1) {
            double *orow = out + (t - p.r0) * p.ld + y * p.shape2 + d.org[2];
            const int64_t e0 = int64_t(r) * dim2;
            for (uint32_t cx = lane; cx < dim2; cx += 64)
                if (d.org[2] + cx < p.shape2) orow[cx] = unpack_value(src, n, e0 + cx, d.shuffled, p.dec);
        }
        if (++cy == dim1) {
            cy = 0;
            ++ct;
        }
    }
}

// Adler-32 of n bytes at p (16-byte aligned) by one wavefront: s1 = 1 + sum b_i, s2 = n + sum (n - i) b_i (mod 65521)
// (chunks <= 64 MiB: the weighted sum stays below 2^60)
__device__ __forceinline__ uint32_t wave_adler(const uint8_t *__restrict__ p, uint64_t n) {
    const uint32_t lane = threadIdx.x;
    const uint64_t n16 = n / 16;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = lane; i < n16; i += 64) {
        const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        uint32_t sum = 0, wsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = (wds[q] >> (8 * k)) & 0xFFu;
                sum += b;
                wsum += uint32_t(4 * q + k) * b;
            }
        s1 += sum;
        s2 += (n - i * 16) * sum - wsum;
    }
    for (uint64_t i = n16 * 16 + lane; i < n; i += 64) {
        s1 += p[i];
        s2 += (n - i) * p[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    const uint32_t a = uint32_t((1 + s1) % 65521ull), b = uint32_t((n % 65521ull + s2 % 65521ull) % 65521ull);
    return (b << 16) | a;
}

// ---- DEFLATE on the device: one wave per chunk stream (serial half: atl_inflate_dev.h) ---------------------------
struct InfDesc {
    int64_t src_off, src_n;  // zlib stream inside the compressed buffer (src_off 128-byte aligned: no cache line holds two streams)
    int64_t dst_off, dst_n;  // inflated chunk inside the raw buffer (16-byte aligned)
    uint32_t part, desc;     // the stream's variable (FedPart) and its chunk (UnpackDesc): the wave unpacks what it inflated
    uint32_t batch;          // fed launches: the stream's bytes are on the device once flags[batch * kFlagPitch] != 0; kNoWait: they are
    uint32_t pad;
};
constexpr uint32_t kNoWait = 0xFFFFFFFFu;
constexpr uint32_t kFlagPitch = 32;   // words between the batches' arrival flags (a 128-byte line each: polled by different waves)
constexpr uint32_t kFlagReady = 1u, kFlagAbort = 2u;

struct FedPart {  // one variable of a read, as the device sees it
    UnpackParams p;
    double *out;
};
struct InfResult {
    int32_t status;       // dinf::Status
    uint32_t adler_want;  // the stream's trailer; k_adler compares
};

#ifndef ATL_INFLATE_WAVES
#define ATL_INFLATE_WAVES 8  // resident streams per SIMD the register allocation aims at (LDS allows 8)
#endif
struct WaveMem {
    typedef __attribute__((address_space(3))) uint32_t *u32p;
    typedef __attribute__((address_space(3))) uint16_t *u16p;
    typedef __attribute__((address_space(3))) uint8_t *u8p;
    typedef __attribute__((address_space(4))) const uint32_t *src_t;  // read-only for the kernel's lifetime: scalar loads
    // every lane reads the same LDS word; the value continues in an SGPR
    static __device__ __forceinline__ uint32_t ld32(u32p p) { return __builtin_amdgcn_readfirstlane(*p); }
    static __device__ __forceinline__ uint32_t ld8(u8p p) { return __builtin_amdgcn_readfirstlane(uint32_t(*p)); }
    // every lane stores the same value to the same address.  (Guarding the store with `lane == 0` puts a lane-dependent branch
    // into every loop of the decoder, after which the compiler treats the loops' whole state as divergent - VGPRs and
    // exec-mask branches instead of SGPRs and scalar branches: measured 0.8 MB/s per stream.)
    static __device__ __forceinline__ void st32(u32p p, uint32_t v) { *p = v; }
    static __device__ __forceinline__ void st8(u8p p, uint32_t v) { *p = uint8_t(v); }
    static __device__ __forceinline__ uint32_t ld16(u16p p) { return __builtin_amdgcn_readfirstlane(uint32_t(*p)); }
    static __device__ __forceinline__ void st16(u16p p, uint32_t v) { *p = uint16_t(v); }
    static __device__ __forceinline__ uint32_t src(src_t w, uint32_t i) { return w[__builtin_amdgcn_readfirstlane(i)]; }
    static __device__ __forceinline__ uint32_t ldv32(u32p p) { return *p; }  // per-lane addresses: an LDS gather / scatter
    static __device__ __forceinline__ void stv32(u32p p, uint32_t v) { *p = v; }
    static __device__ __forceinline__ uint32_t ldv8(u8p p) { return *p; }
    static __device__ __forceinline__ void stv8(u8p p, uint32_t v) { *p = uint8_t(v); }
    static __device__ __forceinline__ uint32_t ldv16(u16p p) { return *p; }
    static __device__ __forceinline__ void stv16(u16p p, uint32_t v) { *p = uint16_t(v); }
    static __device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
    static __device__ __forceinline__ int uni(int v) { return int(__builtin_amdgcn_readfirstlane(uint32_t(v))); }
    static __device__ __forceinline__ uint64_t uni(uint64_t v) {
        return uint64_t(__builtin_amdgcn_readfirstlane(uint32_t(v))) | (uint64_t(__builtin_amdgcn_readfirstlane(uint32_t(v >> 32))) << 32);
    }
};

// the wavefront as decode_batch_wide sees it: a per-lane variable is a register, a per-lane step runs once
struct DevWave {
    template <class T>
    struct Var {
        T v;
        __device__ __forceinline__ T &operator()(int) { return v; }
    };
    template <class F>
    static __device__ __forceinline__ void each(F &&f) {
        f(int(threadIdx.x));
    }
    static __device__ __forceinline__ uint32_t readlane(Var<uint32_t> &x, int lane) { return __builtin_amdgcn_readlane(x.v, lane); }
    static __device__ __forceinline__ uint64_t ballot(Var<uint32_t> &x) { return __ballot(x.v != 0); }
    // a wave-uniform 64-bit value AS the lane mask (one operand of a v_cndmask instead of two ands and a 64-bit compare per
    // use); the lanes of it below this one: v_mbcnt_lo / _hi
    static __device__ __forceinline__ bool in(uint64_t mask, int) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
    static __device__ __forceinline__ uint32_t below(uint64_t mask, int) {
        return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
    }
    // The chain of symbol starts (HostWave::chain), four scalar instructions per hop instead of the six the compiler makes of
    // the C loop (shift + or for the mask, compare + branch for the end): the position is kept as s - 64 (mod 2^32) - the lane
    // select of v_readlane and the bit index of s_bitset1 use the low six bits, which are those of s - and the add's carry-out
    // is "s >= 64".  A lone wave pays ~10 cycles per dependent instruction and a window has ~8 hops.
    static __device__ __forceinline__ uint64_t chain(Var<uint32_t> &jump, uint32_t *end) {
        uint64_t sel = 0;
        uint32_t s = 0xFFFFFFC0u, j;
        asm volatile(".Latl_chain_%=:\n\t"
                     "s_bitset1_b64 %0, %1\n\t"
                     "v_readlane_b32 %2, %3, %1\n\t"
                     "s_add_u32 %1, %1, %2\n\t"
                     "s_cbranch_scc0 .Latl_chain_%="
                     : "+s"(sel), "+s"(s), "=&s"(j)
                     : "v"(jump.v)
                     : "scc");
        *end = s + 64u;
        return sel;
    }
    // x <- sum of x over the lanes below; returns the wave's total.  Four DPP row shifts scan the rows of 16, the rows'
    // totals come over as scalars
    static __device__ __forceinline__ uint32_t excl_scan(Var<uint32_t> &x) {
        const uint32_t v0 = x.v;
        uint32_t v = v0;
        v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, false));  // row_shr:1
        v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, false));  // row_shr:2
        v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, false));  // row_shr:4
        v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, false));  // row_shr:8
        const uint32_t r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47),
                       r3 = __builtin_amdgcn_readlane(v, 63);
        const uint32_t row = threadIdx.x >> 4;  // (three selects of scalars: a chain of ?: compiled to exec-mask branches)
        const uint32_t before = (row > 0u ? r0 : 0u) + (row > 1u ? r1 : 0u) + (row > 2u ? r2 : 0u);
        x.v = v + before - v0;
        return r0 + r1 + r2 + r3;
    }
    // one wavefront per workgroup: its LDS operations execute in order, so lanes see each other's LDS writes without a barrier;
    // what is needed is that the COMPILER keeps the order (a __syncthreads would also wait for every load and store in flight -
    // the window prefetch among them)
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// The words under decode_batch_wide's window, one window ahead: lanes 0 .. 11 hold the twelve words from the PREVIOUS
// window's first word on (the next window starts at most 111 bits later, and reaches at most 63 + 48 bits beyond its start:
// eight words in), loaded while the previous window was decoded; the load for the next window is issued before this one's
// words are needed.  One exposed memory latency per deflate block instead of one per window.
struct DevWindow {
    uint32_t cur = 0, cw0 = 0;
    bool primed = false;
#ifdef ATL_INF_PROFILE
    unsigned long long ticks[4] = {0, 0, 0, 0};
    unsigned windows = 0;
#endif
    __device__ __forceinline__ void reset() { primed = false; }
    __device__ __forceinline__ uint32_t load(const dinf::Areas<WaveMem> &A, WaveMem::src_t w, uint32_t n_words, uint64_t bitpos) {
        const uint32_t lane = threadIdx.x, nw0 = uint32_t(bitpos >> 5);
        const uint32_t *gw = (const uint32_t *)w;
#ifdef ATL_INF_PROFILE
        ++windows;
#endif
        const bool mine = lane < 12u && nw0 + lane < n_words;
        if (!primed) {
            cur = mine ? gw[nw0 + lane] : 0u;
            cw0 = nw0;
            primed = true;
        }
        // the words of THIS window into LDS first - the one place that waits for the load issued a window ago - and only then
        // the load for the next window: issued before the store, the compiler's conservative wait (the priming load above
        // may or may not be outstanding) drained the new load as well, one exposed memory latency per window
        // (every lane stores - lanes 12 .. 63 into the four spare words behind the twelve: a store under a lane condition
        //  sits in a branch, and the wait for `cur` inside a branch does not count for the code behind it)
        A.wbuf[lane < 12u ? lane : 12u + (lane & 3u)] = cur;
        const uint32_t rel = uint32_t(bitpos - uint64_t(cw0) * 32u);
        __builtin_amdgcn_wave_barrier();
        cur = mine ? gw[nw0 + lane] : 0u;
        cw0 = nw0;
        return rel;
    }
};

// The parallel half: lane i owns record i of a batch.  Output bytes of the batch are assembled in the LDS staging area
// (sources inside the batch are LDS reads, sources before it are this wave's own earlier output in HBM) and flushed.
#ifdef ATL_INF_PROFILE
#define ATL_PROF(x) x
#else
#define ATL_PROF(x)
#endif
struct WaveSink {
    WaveMem::u32p qrec, qpos;   // the batch's records (decode_batch_wide)
    ATL_PROF(unsigned long long t_p1 = 0; unsigned long long t_coop = 0; unsigned long long t_flush = 0; unsigned long long t_fence = 0;
             unsigned long long t_res0 = 0; unsigned long long t_res_total = 0; unsigned n_batches = 0; unsigned n_syms = 0; unsigned n_coop = 0;
             unsigned n_far = 0; unsigned n_short = 0; unsigned n_lit = 0;)
    WaveMem::u8p stage;         // [kStage + 3], word aligned
    const uint8_t *src8;        // the stream's bytes (stored blocks)
    uint8_t *dst;               // the chunk's output
    __device__ __forceinline__ void tables_ready() { DevWave::sync(); }
    __device__ __forceinline__ void stored(uint64_t byte_pos, uint32_t len, uint64_t out_pos) {
        for (uint32_t j = threadIdx.x; j < len; j += 64) dst[out_pos + j] = src8[byte_pos + j];
        __threadfence_block();
    }
    __device__ __forceinline__ void resolve(int n, uint64_t bstart, uint64_t bend) {
        const uint32_t lane = threadIdx.x;
        const uint32_t bs = uint32_t(bstart), total = uint32_t(bend - bstart);
        ATL_PROF(const unsigned long long c0 = __builtin_readcyclecounter(); ++n_batches; n_syms += unsigned(n);)
        // the staging area starts (bs & 3) bytes in, so that its aligned words are the output's aligned words (the flush)
        WaveMem::u8p st = stage + (bs & 3u);
        const bool active = int(lane) < n;
        const uint32_t rec = active ? qrec[lane] : 0u, pos = active ? qpos[lane] : 0u;  // this lane's record
        const bool lit = (rec & dinf::kLitFlag) != 0;
        const uint32_t len = rec & 0x1FFu, dist = (rec & 0x7FFFFFFFu) >> 9;
        const uint32_t rel = pos - bs;
        bool coop = false;  // matches that wait for the ordered pass: sources inside the batch, or long
        if (active) {
            if (lit) {
                st[rel] = uint8_t(rec);
            } else if (pos - dist + len <= bs && len <= 16) {  // short, its whole source precedes the batch: on its own
                // eight source bytes per round trip to memory (a byte loop waits for every load; bytes past the match are
                // read and dropped: they lie inside the raw buffer, which ends with 256 bytes of slack)
                const uint8_t *sp = dst + (pos - dist);
                for (uint32_t j0 = 0; j0 < len; j0 += 8) {
                    uint8_t b[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) b[j] = sp[j0 + j];
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (j0 + j < len) st[rel + j0 + j] = b[j];
                }
            } else {
                coop = true;
            }
        }
        uint64_t todo = __ballot(coop);
        DevWave::sync();
        ATL_PROF(const unsigned long long c1 = __builtin_readcyclecounter(); t_p1 += c1 - c0; n_coop += unsigned(__popcll(todo));
                 n_lit += unsigned(__popcll(__ballot(active && lit))); n_short += unsigned(__popcll(__ballot(active && !lit && !coop)));)
        // (tried, round 6: short in-batch matches copied by their own lanes, all whose source bytes are written at once - a bitmap of
        //  owed bytes in LDS, 2.9 rounds per batch instead of 9.8 ordered copies; a round's five dependent LDS trips cost what
        //  it saved: profiles/r06_ingest.txt)
        while (todo) {  // in symbol order; every source byte of a match precedes the match
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = __builtin_amdgcn_readlane(rec, i), p = __builtin_amdgcn_readlane(pos, i);
            const uint32_t L = r & 0x1FFu, D = (r & 0x7FFFFFFFu) >> 9, R = p - bs;
            ATL_PROF(if (p - D < bs) ++n_far;)
            for (uint32_t j = lane; j < L; j += 64) {
                const uint32_t k = D >= L ? j : j % D;  // an overlapping match repeats its first D bytes
                const uint32_t sp = p - D + k;
                st[R + j] = sp >= bs ? st[sp - bs] : dst[sp];
            }
            DevWave::sync();
        }
        ATL_PROF(const unsigned long long c2 = __builtin_readcyclecounter(); t_coop += c2 - c1;)
        // flush: the unaligned head and tail byte by byte, the words in between as words
        const uint32_t head = min((4u - (bs & 3u)) & 3u, total), words = (total - head) >> 2, tail0 = head + 4u * words;
        if (lane < head) dst[bs + lane] = st[lane];
        if (lane >= 32 && lane - 32 < total - tail0) dst[bs + tail0 + (lane - 32)] = st[tail0 + (lane - 32)];
        {
            const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(st + head);
            uint32_t *dw = reinterpret_cast<uint32_t *>(dst + bs + head);
            for (uint32_t k = lane; k < words; k += 64) dw[k] = sw[k];
        }
        ATL_PROF(const unsigned long long c3 = __builtin_readcyclecounter(); t_flush += c3 - c2;)
        __threadfence_block();  // later batches read these bytes back
        DevWave::sync();
        ATL_PROF(const unsigned long long c4 = __builtin_readcyclecounter(); t_fence += c4 - c3; t_res_total += c4 - c0;)
    }
};

// FED launches (round 6).  The kernel is launched BEFORE the compressed bytes are on the device: the host preads them batch by
// batch into a small page-locked ring, each batch is one DMA, and when the DMA's event has completed the host sets the batch's
// arrival flag; a wave waits for its stream's flag (s_sleep between polls, backing off) before it starts.  pread, DMA, inflate,
// checksum and unpack of one read overlap INSIDE one launch - launches from different HIP streams do not (rocprofv3: they run one
// or two at a time, and a launch costs a stream's ~100 ms whatever it holds).
// What can reach a running kernel whose waiting waves fill every wave slot was measured (tools/probes/feed_probe.hip,
// profiles/r06_ingest.txt): bulk DMAs do (the copy engines need no compute unit: 64 MiB in 10 ms), 4-byte DMAs do NOT (the
// runtime copies small transfers with a blit kernel, which finds no free slot until the waves time out - a first version that
// set the flags that way stalled for the whole time-out as soon as a read had more streams than the device holds), a CPU store
// to page-locked host memory that the waves poll across PCIe does, at once.  Hence: flags in the slot's page-locked block, set
// by the CPU.  A wave still gives up after a time-out ($ATLITE_HIP_INGEST_TIMEOUT_MS, default 20 s; status kNotRun: the host
// decoders take the stream), so a stalled host can never hang the device.
__device__ __forceinline__ bool wait_for_batch(const uint32_t *__restrict__ flags, uint32_t batch, unsigned long long timeout_ticks) {
    const uint32_t *f = flags + size_t(batch) * kFlagPitch;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t naps = 1;  // s_sleep(127) ~ 3.4 us each; the gap between polls grows by an eighth per poll up to ~0.9 ms: a wave that
                        // waits 100 ms for a late batch crosses PCIe ~140 times, the 8192 of them together < 1 GB/s
    for (;;) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (v == kFlagReady) break;
        if (v == kFlagAbort || __builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) return false;  // (100 MHz ticks)
        for (uint32_t k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
        naps = min(naps + (naps >> 3) + 1u, 256u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the DMA's bytes, not an older image of the staging buffer, from here on
    __builtin_amdgcn_s_dcache_inv();               // ... through the scalar cache too (the bit reader's s_load)
    return true;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ATL_INFLATE_WAVES, ATL_INFLATE_WAVES))) void k_inflate(
    const uint8_t *__restrict__ comp, const InfDesc *__restrict__ desc, uint8_t *__restrict__ raw, InfResult *__restrict__ res,
    const uint32_t *__restrict__ flags, const FedPart *__restrict__ parts, const UnpackDesc *__restrict__ unp, unsigned long long timeout_ticks) {
    using namespace dinf;
    // 4.7 kB per stream: 32 streams share a CU's 160 KiB (8 waves per SIMD - the register budget of that is 64 VGPRs, which
    // amdgpu_waves_per_eu above asks the compiler for)
    __shared__ uint16_t s_lit[kLitCap];
    __shared__ uint16_t s_off[kOffCap];
    constexpr int kTmpWords = (kStage + 4 > 640 + 352 ? kStage + 4 : 640 + 352) / 4 + 1;
    __shared__ uint32_t s_tmp[kTmpWords];  // codes | code lengths while a table is built; the staging area while symbols are decoded
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_q[2 * kQueue + 1 + 16];  // records | positions (+ 1) | window words
    __shared__ uint32_t s_sym[64];                 // base | extra bits of the length and distance symbols
    static_assert(sizeof(s_tmp) >= size_t(kStage) + 4, "staging area");
    static_assert(ATL_STAGE != 1024 || sizeof(s_lit) + sizeof(s_off) + sizeof(s_tmp) + sizeof(s_cnt) + sizeof(s_q) + sizeof(s_sym) <= 5120, "32 streams per CU");
    const InfDesc d = desc[blockIdx.x];
    if (d.batch != kNoWait && !wait_for_batch(flags, d.batch, timeout_ticks)) {
        if (threadIdx.x == 0) {
            res[blockIdx.x].status = kNotRun;
            res[blockIdx.x].adler_want = 0;
        }
        return;
    }
    ATL_PROF(const unsigned long long t_start = __builtin_readcyclecounter();)
    Areas<WaveMem> A;
    A.lit = (WaveMem::u16p)s_lit;
    A.off = (WaveMem::u16p)s_off;
    A.codes = (WaveMem::u16p)s_tmp;
    A.cnt = (WaveMem::u32p)s_cnt;
    A.nxt = (WaveMem::u32p)(s_cnt + 16);
    A.lens = (WaveMem::u8p)(s_tmp + 160);  // behind the 320 codes
    A.qrec = (WaveMem::u32p)s_q;
    A.qpos = (WaveMem::u32p)(s_q + kQueue);
    A.wbuf = (WaveMem::u32p)(s_q + 2 * kQueue + 1);
    A.sym = (WaveMem::u32p)s_sym;
    WaveSink sink;
    sink.qrec = A.qrec;
    sink.qpos = A.qpos;
    sink.stage = (WaveMem::u8p)s_tmp;
    sink.src8 = comp + d.src_off;
    sink.dst = raw + d.dst_off;
    uint32_t want = 0;
    int st = inflate_stream<WaveMem, DevWave, DevWindow, WaveSink>(A, (WaveMem::src_t)(comp + d.src_off), uint32_t((d.src_n + 3) / 4),
                                                               uint64_t(d.src_n), uint64_t(d.dst_n), sink, &want);
    // the chunk's Adler-32 and, if it is the stream's, the chunk's elements into the block the conversion kernels read (un-shuffle,
    // widen, CF-decode, scatter) - by the wave that has just written the bytes: they are in its XCD's L2
    if (st == kOk) {
        __threadfence_block();
        if (wave_adler(raw + d.dst_off, uint64_t(d.dst_n)) != want) st = kAdler;
    }
    if (st == kOk && parts) wave_unpack(raw, unp[d.desc], parts[d.part].p, parts[d.part].out);
    if (threadIdx.x == 0) {
        res[blockIdx.x].status = st;
        res[blockIdx.x].adler_want = want;
    }
    ATL_PROF(if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        const unsigned long long total = __builtin_readcyclecounter() - t_start;
        printf("[k_inflate %u/%u] out %lld B in %lld B: cycles total %llu | resolve %llu (p1 %llu coop %llu flush %llu fence %llu) | batches %u syms %u lit %u short %u coop %u (far %u)\n",
               blockIdx.x, gridDim.x, (long long)d.dst_n, (long long)d.src_n, total, sink.t_res_total, sink.t_p1, sink.t_coop, sink.t_flush, sink.t_fence,
               sink.n_batches, sink.n_syms, sink.n_lit, sink.n_short, sink.n_coop, sink.n_far);
    })
}

// ---- SEGMENTS: a stream's DEFLATE blocks decoded side by side (the scheme: atl_inflate_dev.h) ------------------------------------
// k_find_blocks: block headers; k_segments<true>: where a segment ends and how much it makes; k_segments<false>: its bytes, markers
// for what it would copy from before its start; k_resolve: markers -> bytes, segment after segment, and the chunk's Adler-32.
// The host follows the chains between the passes (read_group_device: launch_split).
struct FindSpan {
    uint32_t stream, first_bit, n_bits, pad;  // bit offsets [first_bit, first_bit + n_bits) of the stream's bytes
};
constexpr uint32_t kSpanBits = 64u * 1024u * 8u;  // compressed bytes one workgroup searches
constexpr uint32_t kSpanSlots = 7;                // headers kept per span (zlib closes a block every 16 383 symbols: one to three per 64 KiB)
constexpr uint32_t kSpanWords = kSpanSlots + 1;   // count | bit offsets
constexpr uint32_t kFindQueue = 1280;

__global__ __launch_bounds__(256) void k_find_blocks(const uint8_t *__restrict__ comp, const InfDesc *__restrict__ desc,
                                                     const FindSpan *__restrict__ spans, uint32_t *__restrict__ out) {
    using namespace dinf;
    __shared__ uint16_t s_w4[4096];          // the weights of four precode lengths (3 bits each) added up
    __shared__ uint32_t s_q[kFindQueue];     // offsets that passed L1
    __shared__ uint32_t s_nq;
    __shared__ uint8_t s_tab[256 * 132];     // header_l2's lookup table, one per lane (132: the lanes' tables in different banks)
    const uint32_t tid = threadIdx.x;
    const FindSpan sp = spans[blockIdx.x];
    const InfDesc d = desc[sp.stream];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(comp + d.src_off);
    const uint32_t n_words = uint32_t((d.src_n + 3) / 4);
    const uint64_t src_bits = uint64_t(d.src_n) * 8;
    for (uint32_t i = tid; i < 4096u; i += 256u)
        s_w4[i] = uint16_t(precode_weight(i & 7u) + precode_weight((i >> 3) & 7u) + precode_weight((i >> 6) & 7u) + precode_weight((i >> 9) & 7u));
    if (tid == 0) s_nq = 0;
    __syncthreads();
    const uint32_t n_win = (sp.n_bits + 63u) / 64u;
    for (uint32_t w0 = 0; w0 < n_win; w0 += 256u) {
        const uint32_t wi = w0 + tid;
        if (wi < n_win) {
            const uint64_t B = uint64_t(sp.first_bit) + 64ull * wi;
            const uint64_t x0 = bits64_at(w, n_words, B), x1 = bits64_at(w, n_words, B + 64), x2 = bits64_at(w, n_words, B + 128);
            auto v = [&](int k) { return (x0 >> k) | (x1 << (64 - k)); };  // the bits k further on, for all 64 offsets at once
            // BFINAL 0, BTYPE 10b, HLIT and HDIST not 30 / 31 (their upper four bits not all set)
            uint64_t m = ~x0 & ~v(1) & v(2) & ~(v(4) & v(5) & v(6) & v(7)) & ~(v(9) & v(10) & v(11) & v(12));
            const uint32_t rem = sp.n_bits - 64u * wi;
            if (rem < 64u) m &= (uint64_t(1) << rem) - 1u;
            while (m) {
                const uint32_t o = uint32_t(__builtin_ctzll(m));
                m &= m - 1;
                const uint64_t X = o ? (x0 >> o) | (x1 << (64u - o)) : x0, Y = o ? (x1 >> o) | (x2 << (64u - o)) : x1;
                const uint32_t hclen = uint32_t((X >> 13) & 15u) + 4u;
                const uint64_t f = ((X >> 17) | (Y << 47)) & ((uint64_t(1) << (3u * hclen)) - 1u);
                const uint32_t sum = uint32_t(s_w4[f & 4095u]) + s_w4[(f >> 12) & 4095u] + s_w4[(f >> 24) & 4095u] + s_w4[(f >> 36) & 4095u] +
                                     s_w4[(f >> 48) & 4095u];
                if (sum == 128u) {  // a complete precode
                    const uint32_t q = atomicAdd(&s_nq, 1u);
                    if (q < kFindQueue) s_q[q] = uint32_t(B) + o;
                }
            }
        }
        __syncthreads();
        const bool last = w0 + 256u >= n_win;
        for (;;) {  // L2 on full sets of lanes
            const uint32_t nq = min(s_nq, kFindQueue);
            if (nq == 0u || (nq < 256u && !last)) break;
            const uint32_t take = min(nq, 256u), base = nq - take;
            bool ok = false;
            uint32_t bit = 0;
            if (tid < take) {
                bit = s_q[base + tid];
                ok = header_l2(w, n_words, uint64_t(bit), src_bits, (WaveMem::u8p)(s_tab + tid * 132u));
            }
            __syncthreads();
            if (tid == 0) s_nq = base;
            if (ok) {
                const uint32_t slot = atomicAdd(out + size_t(blockIdx.x) * kSpanWords, 1u);
                if (slot < kSpanSlots) out[size_t(blockIdx.x) * kSpanWords + 1u + slot] = bit;
            }
            __syncthreads();
        }
    }
}

struct SegTask {
    uint32_t stream, start_bit, slack, cand0, n_cand;  // cands[cand0 .. cand0 + n_cand): the stream's split points, ascending
    uint32_t res_ix;                                    // where its result goes (tasks are launched longest first)
    uint64_t seg0;                                          // where the segment's output starts in the chunk (count pass: 0)
};
struct SegRes {
    dinf::SegOut o;
    int32_t status;
    uint32_t pad;
};
struct DevSplits {
    const uint32_t *c;
    uint32_t n;
    __device__ __forceinline__ bool is_split(uint64_t bit) const {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t v = __builtin_amdgcn_readfirstlane(c[mid]);
            if (uint64_t(v) < bit) lo = mid + 1; else hi = mid;
        }
        return lo < n && uint64_t(__builtin_amdgcn_readfirstlane(c[lo])) == bit;
    }
};
struct CountSink {
    __device__ __forceinline__ void tables_ready() { DevWave::sync(); }
    __device__ __forceinline__ void stored(uint64_t, uint32_t, uint64_t) {}
    __device__ __forceinline__ void resolve(int, uint64_t, uint64_t) {}
};
// WaveSink with the second plane: a byte whose source lies before the segment's start is a marker (atl_inflate_dev.h)
struct MarkSink {
    WaveMem::u32p qrec, qpos;
    WaveMem::u8p stage, stage_hi;  // [kStage + 3] each, word aligned
    const uint8_t *src8;
    uint8_t *dst, *mk;             // the chunk's value and mark planes
    uint32_t seg0;
    __device__ __forceinline__ void tables_ready() { DevWave::sync(); }
    __device__ __forceinline__ void stored(uint64_t byte_pos, uint32_t len, uint64_t out_pos) {
        for (uint32_t j = threadIdx.x; j < len; j += 64) {
            dst[out_pos + j] = src8[byte_pos + j];
            mk[out_pos + j] = 0;
        }
        __threadfence_block();
    }
    // the two planes of the byte at output position s (before the batch)
    __device__ __forceinline__ void fetch(uint32_t s, uint32_t &lo, uint32_t &hi) const {
        const bool before = s < seg0;
        const uint32_t a = before ? seg0 : s;
        lo = dst[a];
        hi = mk[a];
        if (before) {
            lo = dinf::marker_lo(seg0 - s);
            hi = dinf::marker_hi(seg0 - s);
        }
    }
    __device__ __forceinline__ void resolve(int n, uint64_t bstart, uint64_t bend) {
        const uint32_t lane = threadIdx.x;
        const uint32_t bs = uint32_t(bstart), total = uint32_t(bend - bstart);
        WaveMem::u8p st = stage + (bs & 3u), sh = stage_hi + (bs & 3u);
        const bool active = int(lane) < n;
        const uint32_t rec = active ? qrec[lane] : 0u, pos = active ? qpos[lane] : 0u;
        const bool lit = (rec & dinf::kLitFlag) != 0;
        const uint32_t len = rec & 0x1FFu, dist = (rec & 0x7FFFFFFFu) >> 9;
        const uint32_t rel = pos - bs;
        bool coop = false;
        if (active) {
            if (lit) {
                st[rel] = uint8_t(rec);
                sh[rel] = 0;
            } else if (pos - dist + len <= bs && len <= 16) {  // (dist <= pos: the slack of a segment never exceeds its seg0)
                const uint32_t sp0 = pos - dist;
                for (uint32_t j0 = 0; j0 < len; j0 += 8) {
                    uint32_t lo[8], hi[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) fetch(sp0 + j0 + j, lo[j], hi[j]);
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (j0 + j < len) {
                            st[rel + j0 + j] = uint8_t(lo[j]);
                            sh[rel + j0 + j] = uint8_t(hi[j]);
                        }
                }
            } else {
                coop = true;
            }
        }
        uint64_t todo = __ballot(coop);
        DevWave::sync();
        while (todo) {  // in symbol order; every source byte of a match precedes the match
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = __builtin_amdgcn_readlane(rec, i), p = __builtin_amdgcn_readlane(pos, i);
            const uint32_t L = r & 0x1FFu, D = (r & 0x7FFFFFFFu) >> 9, R = p - bs;
            for (uint32_t j = lane; j < L; j += 64) {
                const uint32_t k = D >= L ? j : j % D;
                const uint32_t sp = p - D + k;
                uint32_t lo, hi;
                if (sp >= bs) {
                    lo = st[sp - bs];
                    hi = sh[sp - bs];
                } else {
                    fetch(sp, lo, hi);
                }
                st[R + j] = uint8_t(lo);
                sh[R + j] = uint8_t(hi);
            }
            DevWave::sync();
        }
        const uint32_t head = min((4u - (bs & 3u)) & 3u, total), words = (total - head) >> 2, tail0 = head + 4u * words;
        if (lane < head) {
            dst[bs + lane] = st[lane];
            mk[bs + lane] = sh[lane];
        }
        if (lane >= 32 && lane - 32 < total - tail0) {
            dst[bs + tail0 + (lane - 32)] = st[tail0 + (lane - 32)];
            mk[bs + tail0 + (lane - 32)] = sh[tail0 + (lane - 32)];
        }
        {
            const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(st + head);
            const __attribute__((address_space(3))) uint32_t *sw2 = (const __attribute__((address_space(3))) uint32_t *)(sh + head);
            uint32_t *dw = reinterpret_cast<uint32_t *>(dst + bs + head), *dw2 = reinterpret_cast<uint32_t *>(mk + bs + head);
            for (uint32_t k = lane; k < words; k += 64) {
                dw[k] = sw[k];
                dw2[k] = sw2[k];
            }
        }
        __threadfence_block();
        DevWave::sync();
    }
};

template <bool COUNT>
__global__ __launch_bounds__(64) void k_segments(const uint8_t *__restrict__ comp, const InfDesc *__restrict__ desc,
                                                 const SegTask *__restrict__ tasks, const uint32_t *__restrict__ cands,
                                                 uint8_t *__restrict__ raw, uint8_t *__restrict__ mark, SegRes *__restrict__ res) {
    using namespace dinf;
    __shared__ uint16_t s_lit[kLitCap];
    __shared__ uint16_t s_off[kOffCap];
    constexpr int kTmpWords = (kStage + 4 > 640 + 352 ? kStage + 4 : 640 + 352) / 4 + 1;
    __shared__ uint32_t s_tmp[kTmpWords];
    __shared__ uint32_t s_hi[COUNT ? 1 : kTmpWords];  // the mark plane's staging area
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_q[2 * kQueue + 1 + 16];
    __shared__ uint32_t s_sym[64];
    const SegTask t = tasks[blockIdx.x];
    const InfDesc d = desc[t.stream];
    Areas<WaveMem> A;
    A.lit = (WaveMem::u16p)s_lit;
    A.off = (WaveMem::u16p)s_off;
    A.codes = (WaveMem::u16p)s_tmp;
    A.cnt = (WaveMem::u32p)s_cnt;
    A.nxt = (WaveMem::u32p)(s_cnt + 16);
    A.lens = (WaveMem::u8p)(s_tmp + 160);
    A.qrec = (WaveMem::u32p)s_q;
    A.qpos = (WaveMem::u32p)(s_q + kQueue);
    A.wbuf = (WaveMem::u32p)(s_q + 2 * kQueue + 1);
    A.sym = (WaveMem::u32p)s_sym;
    const DevSplits splits{cands + t.cand0, t.n_cand};
    SegOut o{};
    int st;
    const WaveMem::src_t w = (WaveMem::src_t)(comp + d.src_off);
    const uint32_t n_words = uint32_t((d.src_n + 3) / 4);
    if constexpr (COUNT) {
        CountSink sink;
        st = inflate_segment<WaveMem, DevWave, DevWindow, CountSink, DevSplits>(A, w, n_words, uint64_t(d.src_n), uint64_t(t.start_bit), 0, t.slack,
                                                                               uint64_t(d.dst_n), splits, sink, &o);
    } else {
        MarkSink sink;
        sink.qrec = A.qrec;
        sink.qpos = A.qpos;
        sink.stage = (WaveMem::u8p)s_tmp;
        sink.stage_hi = (WaveMem::u8p)s_hi;
        sink.src8 = comp + d.src_off;
        sink.dst = raw + d.dst_off;
        sink.mk = mark + d.dst_off;
        sink.seg0 = uint32_t(t.seg0);
        st = inflate_segment<WaveMem, DevWave, DevWindow, MarkSink, DevSplits>(A, w, n_words, uint64_t(d.src_n), uint64_t(t.start_bit), t.seg0, t.slack,
                                                                              uint64_t(d.dst_n), splits, sink, &o);
    }
    if (threadIdx.x == 0) {
        res[t.res_ix].o = o;
        res[t.res_ix].status = st;
    }
}

struct ResDesc {
    uint32_t stream, bound0, n_seg, adler_want;  // bounds[bound0 .. bound0 + n_seg]: where the stream's segments begin (and the chunk's end)
};
// ---- ... in ONE decode pass (no count pass): a segment's output goes into regions of a pool - 8 Ki, 16 Ki, 32 Ki ... 16-bit units, taken
// from a bump allocator as the segment grows - as one unit per byte (the byte, or a marker); when the chains are known k_gather copies
// the segments to their places in the chunk, markers resolved on the way.
#ifndef ATL_POOL_STAGE
#define ATL_POOL_STAGE 1536  // (measured, T = 2000 of 16 MB streams: 496 units 0.118 s, 1024 0.093-0.104, 1536 0.090, 2048 0.090-0.091, 4096 0.102: byte planes
                             //  with runs make long matches, a batch ends when the stage is full; beyond 2048 the LDS costs too many waves)
#endif
constexpr int kPoolStage = ATL_POOL_STAGE;        // output bytes staged per batch (16-bit units: twice the LDS of k_inflate's)
constexpr uint32_t kRegionLog = 13;              // first region: 8192 units (bytes of output); region r holds 8192 << r
constexpr uint32_t kMaxRegions = 14;             // 8192 * (2^14 - 1) = 134 MB: the longest segment (beyond: the stream goes to the host decoders)
struct PoolRef {
    uint16_t *pool;          // 16-bit units
    uint32_t *next;          // bump allocator, in units of 8192
    uint32_t cap;            // ... its end
    uint32_t *regs;          // [tasks][kMaxRegions] the regions' bases (units of 8192): k_gather reads them
};
// position inside a segment -> (region, offset): region r begins at 8192 * (2^r - 1)
__device__ __forceinline__ uint32_t region_of(uint32_t pos) { return 31u - uint32_t(__builtin_clz((pos >> kRegionLog) + 1u)); }
__device__ __forceinline__ uint32_t region_off(uint32_t pos, uint32_t r) { return pos - (((1u << r) - 1u) << kRegionLog); }

struct PoolSink {
    WaveMem::u32p qrec, qpos;
    WaveMem::u16p stage;   // [kPoolStage + 4] units, word aligned
    WaveMem::u32p reg;     // [kMaxRegions] bases of this segment's regions
    const uint8_t *src8;
    uint16_t *pool;
    uint32_t *next, *regs_out;
    uint32_t cap, n_reg = 0, alloc_end = 0;
    bool full = false;
    __device__ __forceinline__ void tables_ready() { DevWave::sync(); }
    __device__ __forceinline__ size_t at(uint32_t pos) const {
        const uint32_t r = region_of(pos);
        return (size_t(reg[r]) << kRegionLog) + region_off(pos, r);
    }
    // regions up to output position `end` (wave-uniform); false: the pool is exhausted
    __device__ __forceinline__ bool ensure(uint32_t end) {
        while (alloc_end < end && !full) {
            uint32_t base = 0;
            if (threadIdx.x == 0) base = atomicAdd(next, 1u << n_reg);
            base = __builtin_amdgcn_readfirstlane(base);
            if (n_reg >= kMaxRegions || base > cap || (1u << n_reg) > cap - base) {
                full = true;
                break;
            }
            if (threadIdx.x == 0) {
                reg[n_reg] = base;
                regs_out[n_reg] = base;
            }
            alloc_end += 1u << (kRegionLog + n_reg);
            ++n_reg;
            DevWave::sync();
        }
        return !full;
    }
    __device__ __forceinline__ void stored(uint64_t byte_pos, uint32_t len, uint64_t out_pos) {
        if (!ensure(uint32_t(out_pos) + len)) return;
        for (uint32_t j = threadIdx.x; j < len; j += 64) pool[at(uint32_t(out_pos) + j)] = src8[byte_pos + j];
        __threadfence_block();
    }
    // the unit of output position s (relative to the segment's start; negative: before it - a marker)
    __device__ __forceinline__ uint32_t fetch(int32_t s) const {
        const bool before = s < 0;
        const uint32_t v = pool[at(before ? 0u : uint32_t(s))];
        return before ? dinf::marker16(uint32_t(-s)) : v;
    }
    __device__ __forceinline__ void resolve(int n, uint64_t bstart, uint64_t bend) {
        const uint32_t lane = threadIdx.x;
        const uint32_t bs = uint32_t(bstart), total = uint32_t(bend - bstart);
        if (!ensure(uint32_t(bend))) return;  // (the segment's status says so at its end: k_segments_pool)
        WaveMem::u16p st = stage + (bs & 1u);  // the staging area's aligned words are the pool's
        const bool active = int(lane) < n;
        const uint32_t rec = active ? qrec[lane] : 0u, pos = active ? qpos[lane] : 0u;
        const bool lit = (rec & dinf::kLitFlag) != 0;
        const uint32_t len = rec & 0x1FFu, dist = (rec & 0x7FFFFFFFu) >> 9;
        const uint32_t rel = pos - bs;
        bool coop = false;
        if (active) {
            if (lit) {
                st[rel] = uint16_t(rec & 0xFFu);
            } else if (int32_t(pos) - int32_t(dist) + int32_t(len) <= int32_t(bs) && len <= 16) {
                const int32_t sp0 = int32_t(pos) - int32_t(dist);
                for (uint32_t j0 = 0; j0 < len; j0 += 8) {
                    // eight units per round trip; nothing behind the match is read (it may not be allocated).  Nearly always the
                    // eight lie in one region: one translation (it is a tenth of this kernel's vector instructions otherwise)
                    uint32_t v[8];
                    const uint32_t last = min(7u, len - 1u - j0);
                    const int32_t sa = sp0 + int32_t(j0), sb = sa + int32_t(last);
                    if (sa >= 0 && region_of(uint32_t(sa)) == region_of(uint32_t(sb))) {
                        const uint16_t *q = pool + at(uint32_t(sa));
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) v[j] = q[min(j, last)];
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) v[j] = fetch(sa + int32_t(min(j, last)));
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j)
                        if (j0 + j < len) st[rel + j0 + j] = uint16_t(v[j]);
                }
            } else {
                coop = true;
            }
        }
        uint64_t todo = __ballot(coop);
        DevWave::sync();
        while (todo) {  // in symbol order; every source byte of a match precedes the match
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t r = __builtin_amdgcn_readlane(rec, i), p = __builtin_amdgcn_readlane(pos, i);
            const uint32_t L = r & 0x1FFu, D = (r & 0x7FFFFFFFu) >> 9, R = p - bs;
            for (uint32_t j = lane; j < L; j += 64) {
                const uint32_t k = D >= L ? j : j % D;
                const int32_t sp = int32_t(p) - int32_t(D) + int32_t(k);
                st[R + j] = uint16_t(sp >= int32_t(bs) ? uint32_t(st[uint32_t(sp) - bs]) : fetch(sp));
            }
            DevWave::sync();
        }
        // flush: an odd first unit and an odd last one on their own, the pairs in between as words
        const uint32_t head = min(bs & 1u, total), words = (total - head) >> 1, tail0 = head + 2u * words;
        if (lane == 0 && head) pool[at(bs)] = st[0];
        if (lane == 32 && tail0 < total) pool[at(bs + tail0)] = st[tail0];
        {
            const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(st + head);
            for (uint32_t k = lane; k < words; k += 64) *reinterpret_cast<uint32_t *>(pool + at(bs + head + 2u * k)) = sw[k];
        }
        __threadfence_block();
        DevWave::sync();
    }
};

#ifdef ATL_POOL_WAVES
__attribute__((amdgpu_waves_per_eu(ATL_POOL_WAVES, ATL_POOL_WAVES)))
#endif
__global__ __launch_bounds__(64) void k_segments_pool(const uint8_t *__restrict__ comp, const InfDesc *__restrict__ desc,
                                                      const SegTask *__restrict__ tasks, const uint32_t *__restrict__ cands, PoolRef pr,
                                                      SegRes *__restrict__ res) {
    using namespace dinf;
    __shared__ uint16_t s_lit[kLitCap];
    __shared__ uint16_t s_off[kOffCap];
    constexpr int kTmpWords = (2 * (kPoolStage + 4) > 640 + 352 ? 2 * (kPoolStage + 4) : 640 + 352) / 4 + 1;
    __shared__ uint32_t s_tmp[kTmpWords];  // codes | code lengths while a table is built; the staging area (16-bit units) while symbols are decoded
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_q[2 * kQueue + 1 + 16];
    __shared__ uint32_t s_sym[64];
    __shared__ uint32_t s_reg[kMaxRegions + 2];
    const SegTask t = tasks[blockIdx.x];
    const InfDesc d = desc[t.stream];
    Areas<WaveMem> A;
    A.lit = (WaveMem::u16p)s_lit;
    A.off = (WaveMem::u16p)s_off;
    A.codes = (WaveMem::u16p)s_tmp;
    A.cnt = (WaveMem::u32p)s_cnt;
    A.nxt = (WaveMem::u32p)(s_cnt + 16);
    A.lens = (WaveMem::u8p)(s_tmp + 160);
    A.qrec = (WaveMem::u32p)s_q;
    A.qpos = (WaveMem::u32p)(s_q + kQueue);
    A.wbuf = (WaveMem::u32p)(s_q + 2 * kQueue + 1);
    A.sym = (WaveMem::u32p)s_sym;
    const DevSplits splits{cands + t.cand0, t.n_cand};
    SegOut o{};
    PoolSink sink;
    sink.qrec = A.qrec;
    sink.qpos = A.qpos;
    sink.stage = (WaveMem::u16p)s_tmp;
    sink.reg = (WaveMem::u32p)s_reg;
    sink.src8 = comp + d.src_off;
    sink.pool = pr.pool;
    sink.next = pr.next;
    sink.cap = pr.cap;
    sink.regs_out = pr.regs + size_t(t.res_ix) * kMaxRegions;
    int st = inflate_segment<WaveMem, DevWave, DevWindow, PoolSink, DevSplits, kPoolStage>(A, (WaveMem::src_t)(comp + d.src_off), uint32_t((d.src_n + 3) / 4),
                                                                              uint64_t(d.src_n), uint64_t(t.start_bit), 0, t.slack, uint64_t(d.dst_n), splits,
                                                                              sink, &o);
    if (st == kOk && sink.full) st = kPoolFull;
    if (threadIdx.x == 0) {
        res[t.res_ix].o = o;
        res[t.res_ix].status = st;
    }
}

struct GatherSeg {
    uint32_t task;    // whose regions (PoolRef::regs); kNoTask: the entry behind a stream's last segment (its start = the chunk's end)
    uint32_t start;   // where the segment's bytes go in the chunk
    uint32_t stream;  // (k_gather_rest)
    uint32_t pad;
};
constexpr uint32_t kNoTask = 0xFFFFFFFFu;
constexpr uint32_t kWindow = 32768;  // DEFLATE's: a marker points at most this far before its segment's start

// Bytes [lo, hi) of a segment that starts at s0 of the chunk at `base`: pool units -> bytes, a marker -> the byte `back` before s0
// (final by then).  By all nt threads of the caller; s_reg: the segment's region bases (LDS), *bad: a marker that points before the chunk.
__device__ __forceinline__ void gather_range(uint8_t *__restrict__ base, uint32_t s0, uint32_t lo, uint32_t hi, const uint32_t *s_reg, const PoolRef &pr,
                                             uint32_t tid, uint32_t nt, int *bad) {
    auto unit = [&](uint32_t i) -> uint32_t {
        const uint32_t r = region_of(i);
        return pr.pool[(size_t(s_reg[r]) << kRegionLog) + region_off(i, r)];
    };
    auto byte_of = [&](uint32_t v) -> uint32_t {
        if (!(v & 0x8000u)) return v & 0xFFu;
        const uint32_t back = dinf::marker16_back(v);
        if (back > s0) {
            *bad = 1;
            return 0u;
        }
        return base[s0 - back];
    };
    // four bytes per thread where the chunk's words allow: the head up to a word boundary and the tail byte by byte
    const uint32_t len = hi - lo;
    const uint32_t h = lo + min((4u - ((s0 + lo) & 3u)) & 3u, len), nw = (hi - h) >> 2, t0 = h + 4u * nw;
    if (lo + tid < h) base[s0 + lo + tid] = uint8_t(byte_of(unit(lo + tid)));
    if (tid >= 32 && t0 + (tid - 32) < hi) base[s0 + t0 + (tid - 32)] = uint8_t(byte_of(unit(t0 + (tid - 32))));
    // the four units of a word: loaded together, then the markers' source bytes together (they lie before s0: final), one store
    auto units4 = [&](uint32_t i, uint32_t *v) {
        if ((i & 1u) == 0u) {  // two aligned pairs (a pair never straddles regions: they begin at multiples of 8192)
            auto pair = [&](uint32_t p) -> uint32_t {
                const uint32_t r = region_of(p);
                return *reinterpret_cast<const uint32_t *>(pr.pool + (size_t(s_reg[r]) << kRegionLog) + region_off(p, r));
            };
            const uint32_t a = pair(i), b = pair(i + 2u);
            v[0] = a & 0xFFFFu;
            v[1] = a >> 16;
            v[2] = b & 0xFFFFu;
            v[3] = b >> 16;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) v[j] = unit(i + j);
        }
    };
    auto sources4 = [&](const uint32_t *v, uint32_t *src) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t back = dinf::marker16_back(v[j]);
            const bool is = (v[j] & 0x8000u) != 0u;
            if (is && back > s0) *bad = 1;
            src[j] = (is && back <= s0) ? s0 - back : 0u;
        }
    };
    auto word_of = [&](const uint32_t *v, const uint32_t *b) -> uint32_t {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) w |= (((v[j] & 0x8000u) ? b[j] : v[j]) & 0xFFu) << (8u * j);
        return w;
    };
    uint32_t k = tid;
    constexpr uint32_t Q = 4;  // words per step: their unit loads, then their source loads, in flight together (8: 128 VGPRs, the sequence 52 -> 61 ms)
    for (; k + (Q - 1u) * nt < nw; k += Q * nt) {
        uint32_t v[Q][4], sx[Q][4], bx[Q][4];
#pragma unroll
        for (uint32_t q = 0; q < Q; ++q) units4(h + 4u * (k + q * nt), v[q]);
#pragma unroll
        for (uint32_t q = 0; q < Q; ++q) sources4(v[q], sx[q]);
#pragma unroll
        for (uint32_t q = 0; q < Q; ++q)
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) bx[q][j] = base[sx[q][j]];
#pragma unroll
        for (uint32_t q = 0; q < Q; ++q) *reinterpret_cast<uint32_t *>(base + s0 + h + 4u * (k + q * nt)) = word_of(v[q], bx[q]);
    }
    for (; k < nw; k += nt) {
        const uint32_t ia = h + 4u * k;
        uint32_t va[4], sa[4], ba[4];
        units4(ia, va);
        sources4(va, sa);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) ba[j] = base[sa[j]];
        *reinterpret_cast<uint32_t *>(base + s0 + ia) = word_of(va, ba);
    }
}

// One workgroup per stream, segment after segment: a segment's markers point into the kWindow bytes before its start, which must be
// final - but only a segment's LAST kWindow bytes are anybody's window.  So the sequence places just those (tails = 1: a third of the
// bytes of 100 KB segments, and the step is what a long stream's gather costs); k_gather_rest places what lies before them, all segments at
// once, when the sequence is through.  Then the chunk's Adler-32: here, or (adler_here = 0: chunks too long for one workgroup, 256 MB: 40 ms)
// by k_adler_parts / k_adler_final.  segs[bound0 .. bound0 + n_seg) and one more entry whose start is the chunk's end.
__global__ __launch_bounds__(1024) void k_gather(const InfDesc *__restrict__ desc, const ResDesc *__restrict__ rds, const GatherSeg *__restrict__ segs,
                                                 PoolRef pr, uint8_t *__restrict__ raw, InfResult *__restrict__ res, int adler_here, int tails) {
    const ResDesc rd = rds[blockIdx.x];
    const InfDesc d = desc[rd.stream];
    uint8_t *base = raw + d.dst_off;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    __shared__ int s_bad;
    __shared__ uint32_t s_reg[2][kMaxRegions];  // this segment's regions | the next one's, fetched while this one is placed
    if (tid == 0) s_bad = 0;
    GatherSeg g = segs[rd.bound0], g1 = segs[rd.bound0 + 1];
    if (tid < kMaxRegions) s_reg[0][tid] = pr.regs[size_t(g.task) * kMaxRegions + tid];
    for (uint32_t c = 0; c < rd.n_seg; ++c) {
        const uint32_t s0 = g.start, len = g1.start - s0;
        __syncthreads();  // (the previous segment's bytes are written; this one's regions are in)
        const GatherSeg g2 = segs[rd.bound0 + min(c + 2u, rd.n_seg)];
        uint32_t next_reg = 0;
        if (tid < kMaxRegions && g1.task != kNoTask) next_reg = pr.regs[size_t(g1.task) * kMaxRegions + tid];
        gather_range(base, s0, (tails && len > kWindow) ? len - kWindow : 0u, len, s_reg[c & 1u], pr, tid, nt, &s_bad);
        if (tid < kMaxRegions) s_reg[(c + 1u) & 1u][tid] = next_reg;
        __threadfence_block();
        g = g1;
        g1 = g2;
    }
    __syncthreads();
    if (!adler_here) {
        if (tid == 0) {
            res[rd.stream].status = s_bad ? int32_t(dinf::kBadDistance) : int32_t(dinf::kOk);  // (k_adler_final has the last word)
            res[rd.stream].adler_want = rd.adler_want;
        }
        return;
    }
    // Adler-32 (wave_adler's sums, the whole workgroup)
    const uint64_t n = uint64_t(d.dst_n), n16 = n / 16;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = tid; i < n16; i += nt) {
        const uint4 v = reinterpret_cast<const uint4 *>(base)[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        uint32_t sum = 0, wsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = (wds[q] >> (8 * k)) & 0xFFu;
                sum += b;
                wsum += uint32_t(4 * q + k) * b;
            }
        s1 += sum;
        s2 += (n - i * 16) * sum - wsum;
    }
    for (uint64_t i = n16 * 16 + tid; i < n; i += nt) {
        s1 += base[i];
        s2 += (n - i) * base[i];
    }
    s1 %= 65521ull;  // (chunks up to 1 GiB: a thread's weighted sum stays below 2^60, the workgroup's would not)
    s2 %= 65521ull;
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ unsigned long long r1[16], r2[16];
    if ((tid & 63u) == 0) {
        r1[tid >> 6] = s1;
        r2[tid >> 6] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long t1 = 0, t2 = 0;
        for (uint32_t k = 0; k < nt / 64u; ++k) {
            t1 += r1[k];
            t2 += r2[k];
        }
        const uint32_t a = uint32_t((1 + t1) % 65521ull), b = uint32_t((n % 65521ull + t2 % 65521ull) % 65521ull);
        const uint32_t got = (b << 16) | a;
        res[rd.stream].status = s_bad ? int32_t(dinf::kBadDistance) : got == rd.adler_want ? int32_t(dinf::kOk) : int32_t(dinf::kAdler);
        res[rd.stream].adler_want = rd.adler_want;
    }
}

// one workgroup per stream: markers -> bytes, then the Adler-32 of the chunk
__global__ __launch_bounds__(1024) void k_resolve(const InfDesc *__restrict__ desc, const ResDesc *__restrict__ rds, const uint32_t *__restrict__ bounds,
                                                  uint8_t *__restrict__ raw, const uint8_t *__restrict__ mark, InfResult *__restrict__ res) {
    const ResDesc rd = rds[blockIdx.x];
    const InfDesc d = desc[rd.stream];
    uint8_t *base = raw + d.dst_off;
    const uint8_t *mb = mark + d.dst_off;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (uint32_t c = 1; c < rd.n_seg; ++c) {
        const uint32_t s0 = bounds[rd.bound0 + c], s1 = bounds[rd.bound0 + c + 1];
        // whole words where no byte is a marker are skipped; the four source bytes of a word with markers are loaded together
        // (they lie before s0: final), the word is stored once
        const uint32_t a0 = min((s0 + 3u) & ~3u, s1), a1 = max(s1 & ~3u, a0);
        auto one = [&](uint32_t i) {
            const uint32_t m = mb[i];
            if (m & 0x80u) {
                const uint32_t back = dinf::marker_back(base[i], m);
                if (back > s0) s_bad = 1; else base[i] = base[s0 - back];
            }
        };
        for (uint32_t i = s0 + tid; i < a0; i += nt) one(i);
        for (uint32_t i = a1 + tid; i < s1; i += nt) one(i);
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(mb);
        uint32_t *bw = reinterpret_cast<uint32_t *>(base);
        auto word = [&](uint32_t k, uint32_t m4) {
            const uint32_t v4 = bw[k];
            uint32_t src[4], b[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t m = (m4 >> (8u * j)) & 0xFFu, back = dinf::marker_back((v4 >> (8u * j)) & 0xFFu, m);
                const bool is = (m & 0x80u) != 0u;
                if (is && back > s0) s_bad = 1;
                src[j] = (is && back <= s0) ? s0 - back : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) b[j] = base[src[j]];
            uint32_t nv = v4;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if ((m4 >> (8u * j)) & 0x80u) nv = (nv & ~(0xFFu << (8u * j))) | (b[j] << (8u * j));
            bw[k] = nv;
        };
        uint32_t k = a0 / 4u + tid;
        const uint32_t k_end = a1 / 4u;
        for (; k + nt < k_end; k += 2u * nt) {  // two words per step: eight source loads in flight
            const uint32_t ma = mw[k], mb2 = mw[k + nt];
            if (ma & 0x80808080u) word(k, ma);
            if (mb2 & 0x80808080u) word(k + nt, mb2);
        }
        if (k < k_end) {
            const uint32_t ma = mw[k];
            if (ma & 0x80808080u) word(k, ma);
        }
        __threadfence_block();
        __syncthreads();
    }
    // Adler-32 (wave_adler's sums, the whole workgroup)
    const uint64_t n = uint64_t(d.dst_n), n16 = n / 16;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = tid; i < n16; i += nt) {
        const uint4 v = reinterpret_cast<const uint4 *>(base)[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        uint32_t sum = 0, wsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = (wds[q] >> (8 * k)) & 0xFFu;
                sum += b;
                wsum += uint32_t(4 * q + k) * b;
            }
        s1 += sum;
        s2 += (n - i * 16) * sum - wsum;
    }
    for (uint64_t i = n16 * 16 + tid; i < n; i += nt) {
        s1 += base[i];
        s2 += (n - i) * base[i];
    }
    s1 %= 65521ull;  // (chunks up to 1 GiB: a thread's weighted sum stays below 2^60, the workgroup's would not)
    s2 %= 65521ull;
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ unsigned long long r1[16], r2[16];
    if ((tid & 63u) == 0) {
        r1[tid >> 6] = s1;
        r2[tid >> 6] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long t1 = 0, t2 = 0;
        for (uint32_t k = 0; k < nt / 64u; ++k) {
            t1 += r1[k];
            t2 += r2[k];
        }
        const uint32_t a = uint32_t((1 + t1) % 65521ull), b = uint32_t((n % 65521ull + t2 % 65521ull) % 65521ull);
        const uint32_t got = (b << 16) | a;
        res[rd.stream].status = s_bad ? int32_t(dinf::kBadDistance) : got == rd.adler_want ? int32_t(dinf::kOk) : int32_t(dinf::kAdler);
        res[rd.stream].adler_want = rd.adler_want;
    }
}

// ... what lies before a segment's last kWindow bytes: pieces of at most kRestPiece bytes, a workgroup each, all at once (every window is
// final).  (A workgroup per SEGMENT took 54 ms for a read whose longest segments - blocks of 258-byte matches - make tens of MB.)
struct GatherPiece {
    uint32_t seg, lo, hi, pad;  // bytes [lo, hi) of segment segs[seg]
};
constexpr uint32_t kRestPiece = 256u << 10;
__global__ __launch_bounds__(256) void k_gather_rest(const InfDesc *__restrict__ desc, const GatherSeg *__restrict__ segs, const GatherPiece *__restrict__ pieces,
                                                     PoolRef pr, uint8_t *__restrict__ raw, InfResult *__restrict__ res) {
    const GatherPiece pc = pieces[blockIdx.x];
    const GatherSeg g = segs[pc.seg];
    __shared__ int s_bad;
    __shared__ uint32_t s_reg[kMaxRegions];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    if (tid < kMaxRegions) s_reg[tid] = pr.regs[size_t(g.task) * kMaxRegions + tid];
    __syncthreads();
    gather_range(raw + desc[g.stream].dst_off, g.start, pc.lo, pc.hi, s_reg, pr, tid, 256u, &s_bad);
    __syncthreads();
    if (tid == 0 && s_bad) res[g.stream].status = int32_t(dinf::kBadDistance);
}

// Adler-32 of long chunks by many workgroups: partial sums of 4 MiB pieces (weights n - i: the chunk's own), added up per stream
constexpr uint64_t kAdlerPiece = uint64_t(4) << 20;
__global__ __launch_bounds__(256) void k_adler_parts(const InfDesc *__restrict__ desc, const ResDesc *__restrict__ rds, const uint8_t *__restrict__ raw,
                                                     unsigned long long *__restrict__ acc) {
    const ResDesc rd = rds[blockIdx.y];
    const InfDesc d = desc[rd.stream];
    const uint64_t n = uint64_t(d.dst_n), p0 = uint64_t(blockIdx.x) * kAdlerPiece;
    if (p0 >= n) return;
    const uint64_t p1 = min(n, p0 + kAdlerPiece);
    const uint8_t *base = raw + d.dst_off;
    const uint32_t tid = threadIdx.x;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = p0 / 16 + tid; i < p1 / 16; i += 256) {  // (pieces begin at multiples of 16; the last one's tail below)
        const uint4 v = reinterpret_cast<const uint4 *>(base)[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        uint32_t sum = 0, wsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = (wds[q] >> (8 * k)) & 0xFFu;
                sum += b;
                wsum += uint32_t(4 * q + k) * b;
            }
        s1 += sum;
        s2 += (n - i * 16) * sum - wsum;
    }
    for (uint64_t i = (p1 / 16) * 16 + tid; i < p1; i += 256) {
        s1 += base[i];
        s2 += (n - i) * base[i];
    }
    s1 %= 65521ull;
    s2 %= 65521ull;
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ unsigned long long r1[4], r2[4];
    if ((tid & 63u) == 0) {
        r1[tid >> 6] = s1;
        r2[tid >> 6] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(acc + 2 * size_t(blockIdx.y), (r1[0] + r1[1] + r1[2] + r1[3]) % 65521ull);
        atomicAdd(acc + 2 * size_t(blockIdx.y) + 1, (r2[0] + r2[1] + r2[2] + r2[3]) % 65521ull);
    }
}
__global__ void k_adler_final(const InfDesc *__restrict__ desc, const ResDesc *__restrict__ rds, uint32_t n_rd, const unsigned long long *__restrict__ acc,
                              InfResult *__restrict__ res) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rd) return;
    const ResDesc rd = rds[k];
    const uint64_t n = uint64_t(desc[rd.stream].dst_n);
    const uint32_t a = uint32_t((1 + acc[2 * size_t(k)]) % 65521ull), b = uint32_t((n % 65521ull + acc[2 * size_t(k) + 1] % 65521ull) % 65521ull);
    if (res[rd.stream].status == int32_t(dinf::kOk) && ((b << 16) | a) != rd.adler_want) res[rd.stream].status = int32_t(dinf::kAdler);
}

// ---- per-context staging: kSlots slots, each {pinned host, device raw, descriptor buffers, event, stream} -------------
// a read whose chunks were inflated on the device and whose verdicts (InfResult) have not been looked at yet
struct Part {  // one variable of a read
    const Dataset *ds = nullptr;
    UnpackParams p{};
    double *d_out = nullptr;
    int64_t chunk_bytes = 0, max_elems = 0;
    size_t desc0 = 0, n_desc = 0;  // its UnpackDesc run
};

struct Pending {
    bool active = false;
    bool aborted = false;  // the read call itself failed (and said so): nothing of this job is decoded again
    atl_nc *nc = nullptr;
    std::vector<Part> parts;
    std::vector<uint32_t> part_of;   // stream -> part
    std::vector<size_t> lin;         // stream -> linear chunk index (in its variable)
    std::vector<uint32_t> desc_of;   // stream -> index of its UnpackDesc (global)
    std::vector<InfDesc> inf;
    size_t off_unp = 0;              // device: the UnpackDesc array inside sl.d
    size_t off_res = 0;              // host: the InfResult array inside sl.h
    size_t off_res_dev = 0;          // ... and inside sl.d (fetched when the slot is settled: finish_slot)
    size_t off_flag = 0;             // host: the batches' arrival flags inside sl.h
};

struct Slot {
    uint8_t *h = nullptr;   // page-locked: host path - inflated chunks + descriptors; device path - descriptors + verdicts
    uint8_t *d = nullptr;   // device: the image of h (host path); compressed streams + descriptors + arrival flags (device path)
    size_t bytes = 0, d_bytes = 0;
    uint8_t *d_raw = nullptr;  // device path: the inflated chunks
    size_t raw_bytes = 0;
    uint8_t *d_pool = nullptr;  // segments decoded in one pass: the pool of output regions + its lists (k_segments_pool)
    size_t pool_bytes = 0;
    hipEvent_t ev = nullptr;
    hipEvent_t ev_fork[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;    // device path: the slot's kernel stream (two reads in flight overlap)
    hipStream_t st2 = nullptr;   // ... segments decoded block by block: the decode kernels' stream (beside the finder's, behind the DMAs)
    hipEvent_t ev_st2 = nullptr;
    hipStream_t st_c = nullptr;  // ... and its copy stream: the batches' DMAs run beside the kernel that waits for them
    hipEvent_t ev_t[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // device path: stage boundaries (timing)
    hipEvent_t ev_c[2] = {nullptr, nullptr};                             // ... first / last DMA on st_c
    hipEvent_t ev_meta = nullptr;
    std::vector<hipEvent_t> ev_batch;  // device path: "this batch's DMA has completed" (then the CPU sets its arrival flag)
    bool pending = false;
    bool joined = true;  // device path: the copy stream has been made to wait for this slot's stream (ingest_join)
    Pending job;
};

constexpr unsigned kSlots = 4;     // reads in flight: host-side staging of one overlaps DMA + device work of the others
constexpr unsigned kFedSlots = 2;  // ... of them used by device-path jobs (each may hold GBs of compressed + inflated chunks)
constexpr unsigned kRing = 3;      // page-locked buffers the batches of a device-path job rotate through

struct Ring {
    uint8_t *h = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool busy = false;
};

struct IngestState {
    Slot slot[kSlots];
    unsigned calls = 0, fed_calls = 0;
    Ring ring[kRing];
    unsigned ring_next = 0;
    atl_ctx *ctx = nullptr;
    int64_t n_device_chunks = 0, n_host_chunks = 0, n_redone = 0;
    // device path, accumulated: host gather of the compressed bytes (wall clock) | H2D (first to last DMA) | k_inflate (incl. its
    // waits for the DMAs) | segments decoded side by side (a count; launch_split) | k_unpack of never-written chunks (events)
    double ms[5] = {0, 0, 0, 0, 0};
    int64_t comp_bytes = 0, raw_bytes = 0;
    // the verdict of a device-inflate read that failed while nobody was listening (atl_nc_close, a slot's reuse by an
    // unrelated call): kept until a caller that propagates it has seen it - ingest_finish, i.e. whoever observes the copy stream
    int sticky_rc = ATL_OK;
    std::string sticky_msg;
};

// remember the calling thread's error as the state's verdict (the first one wins); returns rc
int keep_verdict(IngestState *st, int rc) {
    if (rc && !st->sticky_rc) {
        st->sticky_rc = rc;
        st->sticky_msg = atl_last_error();
    }
    return rc;
}

// hand the kept verdict to a caller that returns it to its own caller: reported once
int take_verdict(IngestState *st) {
    const int rc = st->sticky_rc;
    if (rc) {
        set_error("%s", st->sticky_msg.c_str());
        st->sticky_rc = ATL_OK;
        st->sticky_msg.clear();
    }
    return rc;
}

// live states, so that closing a file can settle the reads that still refer to it
std::mutex g_states_m;
std::vector<IngestState *> g_states;

int finish_slot(atl_ctx *ctx, IngestState *st, Slot &sl);
int read_rows_device(atl_ctx *ctx, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0, double *const *d_outs,
                     int n_threads, bool *done);
int read_slab_one(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out, int n_threads, bool try_device);

void ingest_free(void *p) {
    IngestState *s = static_cast<IngestState *>(p);
    {
        std::lock_guard<std::mutex> lk(g_states_m);
        g_states.erase(std::remove(g_states.begin(), g_states.end(), s), g_states.end());
    }
    for (Slot &sl : s->slot) {
        if (sl.ev) {
            if (sl.pending) (void)hipEventSynchronize(sl.ev);
            (void)hipEventDestroy(sl.ev);
        }
        if (sl.ev_st2) (void)hipEventDestroy(sl.ev_st2);
        for (hipStream_t q : {sl.st, sl.st_c, sl.st2})
            if (q) {
                (void)hipStreamSynchronize(q);
                (void)hipStreamDestroy(q);
            }
        for (hipEvent_t e : sl.ev_fork)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : sl.ev_t)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : sl.ev_c)
            if (e) (void)hipEventDestroy(e);
        if (sl.ev_meta) (void)hipEventDestroy(sl.ev_meta);
        for (hipEvent_t e : sl.ev_batch) (void)hipEventDestroy(e);
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d) (void)dev_free(sl.d);
        if (sl.d_raw) (void)dev_free(sl.d_raw);
        if (sl.d_pool) (void)dev_free(sl.d_pool);
    }
    for (Ring &r : s->ring) {
        if (r.ev) (void)hipEventDestroy(r.ev);
        if (r.h) (void)hipHostFree(r.h);
    }
    delete s;
}

IngestState *state_of(atl_ctx *ctx) {
    if (!ctx->ingest) {
        IngestState *st = new IngestState();
        st->ctx = ctx;
        ctx->ingest = st;
        ctx->ingest_free = ingest_free;
        std::lock_guard<std::mutex> lk(g_states_m);
        g_states.push_back(st);
    }
    return static_cast<IngestState *>(ctx->ingest);
}

// A staging slot with at least h_bytes of page-locked memory, d_bytes of device memory and raw_bytes for inflated chunks.
// fed: a device-path job (rotates over the first kFedSlots slots only).
int slot_acquire(atl_ctx *ctx, size_t h_bytes, size_t d_bytes, size_t raw_bytes, bool fed, Slot **out) {
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    IngestState *st = state_of(ctx);
    // the next slot in turn - unless another one is idle and already holds buffers of this size (a read of a whole slab stages
    // a GB: page-locking that again for every slot of the rotation would cost more than the read)
    const unsigned n_rot = fed ? kFedSlots : kSlots;
    unsigned &calls = fed ? st->fed_calls : st->calls;
    unsigned pick = calls % n_rot;
    for (unsigned k = 0; k < n_rot; ++k) {
        Slot &c = st->slot[(calls + k) % n_rot];
        const bool idle = !c.pending || hipEventQuery(c.ev) == hipSuccess;
        if (idle && c.bytes >= h_bytes && c.d_bytes >= d_bytes && c.raw_bytes >= raw_bytes) {
            pick = (calls + k) % n_rot;
            break;
        }
    }
    (void)hipGetLastError();  // (hipEventQuery's "not ready" is not an error)
    ++calls;
    Slot &sl = st->slot[pick];
    if (!sl.ev) ATL_HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    (void)finish_slot(ctx, st, sl);  // the previous user of this slot has left its stream; its verdicts are in
    int rc = take_verdict(st);        // (a failed read - this slot's or one settled unobserved - ends the call that finds it)
    if (rc) return rc;
    if (sl.bytes < h_bytes) {
        if (sl.h) (void)hipHostFree(sl.h);
        sl.h = nullptr;
        sl.bytes = 0;
        const size_t want = align_up(h_bytes + h_bytes / 4, size_t(1) << 20);
        ATL_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sl.h), want, hipHostMallocDefault));
        sl.bytes = want;
    }
    if (sl.d_bytes < d_bytes) {
        if (sl.d) (void)dev_free(sl.d);
        sl.d = nullptr;
        sl.d_bytes = 0;
        const size_t want = align_up(d_bytes + d_bytes / 4, size_t(1) << 20);
        ATL_HIP_TRY(dev_malloc(reinterpret_cast<void **>(&sl.d), want));
        sl.d_bytes = want;
    }
    if (sl.raw_bytes < raw_bytes) {
        if (sl.d_raw) (void)dev_free(sl.d_raw);
        sl.d_raw = nullptr;
        sl.raw_bytes = 0;
        const size_t want = align_up(raw_bytes + raw_bytes / 8, size_t(1) << 20);
        ATL_HIP_TRY(dev_malloc(reinterpret_cast<void **>(&sl.d_raw), want));
        sl.raw_bytes = want;
    }
    *out = &sl;
    return ATL_OK;
}

// Every ring buffer at least `bytes` long.  BEFORE a fed launch: hipHostFree / hipHostMalloc wait for the device, and a device
// that runs a kernel waiting for these very buffers' contents never gets there (measured: the read stalled for the waves' whole
// time-out the first time a job needed a longer ring than its predecessor).
int ring_reserve(atl_ctx *ctx, size_t bytes) {
    IngestState *st = state_of(ctx);
    for (Ring &r : st->ring) {
        if (!r.ev) ATL_HIP_TRY(hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
        if (r.bytes >= bytes) continue;
        if (r.busy) {
            ATL_HIP_TRY(hipEventSynchronize(r.ev));
            r.busy = false;
        }
        if (r.h) (void)hipHostFree(r.h);
        r.h = nullptr;
        r.bytes = 0;
        const size_t want = align_up(bytes, size_t(1) << 20);
        ATL_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&r.h), want, hipHostMallocDefault));
        r.bytes = want;
    }
    return ATL_OK;
}

// the next page-locked ring buffer (ring_reserve has sized it), free of its last DMA
int ring_acquire(atl_ctx *ctx, size_t bytes, Ring **out) {
    IngestState *st = state_of(ctx);
    Ring &r = st->ring[st->ring_next++ % kRing];
    ATL_REQUIRE(r.h && r.bytes >= bytes, "ring_acquire: the ring was not reserved");
    if (r.busy) {
        ATL_HIP_TRY(hipEventSynchronize(r.ev));
        r.busy = false;
    }
    *out = &r;
    return ATL_OK;
}

// stage `payload` bytes (in sl->h, or in the caller's pinned buffer h_payload) + descriptors, copy, decode
int submit(atl_ctx *ctx, Slot *sl, size_t payload, const std::vector<UnpackDesc> &descs, const UnpackParams &p,
           int64_t max_chunk_elems, double *d_out, const void *h_payload = nullptr) {
    hipStream_t cs;
    int rc = copy_stream_of(ctx, &cs);
    if (rc) return rc;
    rc = ingest_join(ctx, cs);  // behind the device-inflate reads in flight (they may write the same block)
    if (rc) return rc;
    const size_t desc_off = align_up(payload, 256);
    memcpy(sl->h + desc_off, descs.data(), descs.size() * sizeof(UnpackDesc));
    if (h_payload) {
        ATL_HIP_TRY(hipMemcpyAsync(sl->d, h_payload, payload, hipMemcpyHostToDevice, cs));
        ATL_HIP_TRY(hipMemcpyAsync(sl->d + desc_off, sl->h + desc_off, descs.size() * sizeof(UnpackDesc),
                                   hipMemcpyHostToDevice, cs));
    } else {
        ATL_HIP_TRY(hipMemcpyAsync(sl->d, sl->h, desc_off + descs.size() * sizeof(UnpackDesc), hipMemcpyHostToDevice, cs));
    }
    const UnpackDesc *d_desc = reinterpret_cast<const UnpackDesc *>(sl->d + desc_off);
    const unsigned bx = unsigned(std::min<int64_t>((max_chunk_elems + 255) / 256, 2048));
    for (size_t c0 = 0; c0 < descs.size(); c0 += 32768) {
        const unsigned by = unsigned(std::min<size_t>(descs.size() - c0, 32768));
        hipLaunchKernelGGL(k_unpack, dim3(std::max(bx, 1u), by), dim3(256), 0, cs, sl->d, d_desc + c0, p, d_out);
    }
    ATL_HIP_TRY(hipGetLastError());
    ATL_HIP_TRY(hipEventRecord(sl->ev, cs));
    sl->pending = true;
    return ATL_OK;
}

// CPUs this process may actually use: the cgroup CPU quota (containers routinely expose all host
// threads but grant a fraction of them), else the hardware thread count
int usable_cpus() {
    static const int cached = [] {
        int hw = int(std::max(1u, std::thread::hardware_concurrency()));
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1) period = 0;
                fclose(g);
            }
        }
        if (quota > 0 && period > 0) hw = int(std::min<long long>(hw, (quota + period - 1) / period));
        return std::max(1, hw);
    }();
    return cached;
}

int pick_threads(int requested, size_t n_items) {
    int n = requested;
    if (n <= 0) {
        // a few more threads than CPUs smooths over stragglers; measured best on a 16-CPU quota: 32
        n = std::min(2 * usable_cpus(), 128);
        if (const char *e = getenv("ATLITE_HIP_IO_THREADS")) n = std::max(1, atoi(e));
    }
    return int(std::max<size_t>(1, std::min<size_t>(size_t(n), n_items)));
}

// Persistent worker pool: a read call fans its chunks out to up to 128 threads several times per
// slab, so the threads are created once per process and parked on a condition variable in between.
class Pool {
   public:
    static Pool &get() {
        static Pool p;
        return p;
    }
    // run body() on the caller + up to (n_threads - 1) workers; returns when all have finished
    void run(int n_threads, const std::function<void()> &body) {
        std::lock_guard<std::mutex> serial(run_m_);
        const int helpers = std::max(0, n_threads - 1);
        {
            std::unique_lock<std::mutex> lk(m_);
            while (int(workers_.size()) < helpers) workers_.emplace_back([this] { loop(); });
            body_ = &body;
            want_ = helpers;
            started_ = finished_ = 0;
            ++gen_;
        }
        cv_.notify_all();
        body();
        std::unique_lock<std::mutex> lk(m_);
        want_ = started_;  // late wakers find nothing to do
        done_.wait(lk, [this] { return finished_ == started_; });
        body_ = nullptr;
    }

   private:
    Pool() = default;
    ~Pool() {
        {
            std::unique_lock<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (!body_ || started_ >= want_) continue;
            ++started_;
            const std::function<void()> *b = body_;
            lk.unlock();
            (*b)();
            lk.lock();
            ++finished_;
            if (finished_ == started_) done_.notify_all();
        }
    }
    std::mutex run_m_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void()> *body_ = nullptr;
    int want_ = 0, started_ = 0, finished_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// run fn(i) for i in [0, n) on up to n_threads threads; first failure wins (message carried over,
// since atl_last_error is thread-local)
template <class F>
int parallel_for(size_t n, int n_threads, F fn) {
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    std::string msg;
    std::atomic<bool> have_msg{false};
    std::function<void()> body = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n || err.load()) return;
            int rc;
            try {  // an exception must not leave a worker thread (std::terminate): a chunk buffer that cannot be allocated
                rc = fn(i);
            } catch (const std::bad_alloc &) {
                set_error("out of host memory while reading chunk %zu", i);
                rc = ATL_E_NOMEM;
            } catch (const std::exception &e) {
                set_error("%s", e.what());
                rc = ATL_E_INVALID;
            }
            if (rc) {
                int expect = 0;
                if (err.compare_exchange_strong(expect, rc)) {
                    msg = atl_last_error();
                    have_msg = true;
                }
            }
        }
    };
    if (n_threads <= 1 || n <= 1) {
        body();
    } else {
        Pool::get().run(n_threads, body);
    }
    if (err.load() && have_msg) set_error("%s", msg.c_str());
    return err.load();
}

struct Selection {  // chunks of a variable overlapping rows [r0, r1)
    std::vector<size_t> lin;          // linear chunk index
    std::vector<UnpackDesc> desc;
};

void select_chunks(const Geometry &g, int64_t r0, int64_t r1, int64_t chunk_bytes, Selection *s) {
    const int64_t g0a = r0 / g.chunk[0], g0b = (r1 - 1) / g.chunk[0];
    int64_t off = 0;
    for (int64_t a = g0a; a <= g0b; ++a)
        for (int64_t b = 0; b < g.grid[1]; ++b)
            for (int64_t c = 0; c < g.grid[2]; ++c) {
                UnpackDesc d{};
                d.src_off = off;
                d.org[0] = a * g.chunk[0];
                d.org[1] = b * g.chunk[1];
                d.org[2] = c * g.chunk[2];
                d.dim[0] = g.chunk[0];
                d.dim[1] = g.chunk[1];
                d.dim[2] = g.chunk[2];
                s->lin.push_back(size_t((a * g.grid[1] + b) * g.grid[2] + c));
                s->desc.push_back(d);
                off += align_up(size_t(chunk_bytes), 16);
            }
}

int lookup(atl_nc *f, const char *name, const Dataset **out, const char *who) {
    ATL_REQUIRE(f && name, "%s: bad argument", who);
    const Dataset *d = f->file.find(name);
    ATL_REQUIRE(d, "%s: no variable '%s' in the file", who, name);
    *out = d;
    return ATL_OK;
}

// ---- device inflate: submit, and the settling of its verdicts ---------------------------------------------------------
// $ATLITE_HIP_INFLATE_SPLIT: 1 / 0 forces / forbids decoding a stream's blocks side by side (launch_split); unset: for few, long streams
int split_env() {
    const char *e = getenv("ATLITE_HIP_INFLATE_SPLIT");
    return (e && *e) ? (strcmp(e, "0") != 0 ? 1 : 0) : -1;
}
constexpr size_t kSplitMaxStreams = 4096;             // from there on whole-stream waves fill the device
constexpr double kSplitMinBytes = double(2u << 20);   // mean inflated bytes per stream from which a stream is "long"

bool device_inflate_wanted(size_t n_streams, double raw_bytes) {
    const char *e = getenv("ATLITE_HIP_INFLATE");
    if (e && strcmp(e, "device") == 0) return n_streams > 0;
    if (e && *e) return false;  // "host", "zlib"
    size_t min_chunks = 1024;   // below that the host threads finish first: a stream is one wavefront at ~10 MB/s, so a read costs
                                // ~0.1 s however few streams it has; 16 host threads inflate ~1000 chunks of 1 MB in that time
    if (const char *m = getenv("ATLITE_HIP_INFLATE_MIN_CHUNKS")) min_chunks = size_t(std::max(1, atoi(m)));
    if (n_streams >= min_chunks) return true;
    // few but LONG streams (atlite's own cutouts: (time = 100, y, x) chunks): their blocks are decoded side by side, 3 - 4 x the
    // host threads' rate from a few hundred MB on (the scheme's passes cost ~10 ms before the first byte)
    return split_env() != 0 && n_streams > 0 && raw_bytes / double(n_streams) >= kSplitMinBytes && raw_bytes >= double(256u << 20);
}

void launch_unpack(hipStream_t st, const uint8_t *raw, const UnpackDesc *d_desc, size_t n_desc, const UnpackParams &p,
                   int64_t max_chunk_elems, double *d_out) {
    const unsigned bx = unsigned(std::min<int64_t>((max_chunk_elems + 255) / 256, 2048));
    for (size_t c0 = 0; c0 < n_desc; c0 += 32768) {
        const unsigned by = unsigned(std::min<size_t>(n_desc - c0, 32768));
        hipLaunchKernelGGL(k_unpack, dim3(std::max(bx, 1u), by), dim3(256), 0, st, raw, d_desc + c0, p, d_out);
    }
}

static int finish_slot_impl(atl_ctx *ctx, IngestState *st, Slot &sl) {
    if (sl.pending) {
        sl.pending = false;
        ATL_HIP_TRY(hipEventSynchronize(sl.ev));
    }
    sl.joined = true;
    if (sl.d_pool) {
        // the segment scheme's pool of a year-sized read is tens of GB: kept for the next read (allocating and freeing 40 GB costs
        // ~30 ms of a 0.35 s call) unless it is more than a quarter of the device's memory
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            total_b = 0;
        }
        if (sl.pool_bytes > total_b / 4) {
            (void)dev_free(sl.d_pool);
            sl.d_pool = nullptr;
            sl.pool_bytes = 0;
        }
    }
    Pending &job = sl.job;
    if (!job.active) return ATL_OK;
    job.active = false;
    {
        float t = 0.f;
        if (hipEventElapsedTime(&t, sl.ev_c[0], sl.ev_c[1]) == hipSuccess) st->ms[1] += double(t);
        if (hipEventElapsedTime(&t, sl.ev_t[1], sl.ev_t[2]) == hipSuccess) st->ms[2] += double(t);
        if (hipEventElapsedTime(&t, sl.ev_t[2], sl.ev_t[3]) == hipSuccess) st->ms[4] += double(t);
    }
    (void)hipGetLastError();
    if (job.aborted) return ATL_OK;  // the read call failed and reported it: nothing to decode again
    for (const InfDesc &q : job.inf) {
        st->comp_bytes += q.src_n;
        st->raw_bytes += q.dst_n;
    }
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    ATL_HIP_TRY(hipMemcpy(sl.h + job.off_res, sl.d + job.off_res_dev, job.inf.size() * sizeof(InfResult), hipMemcpyDeviceToHost));  // (the kernels are done: sl.ev)
    const InfResult *res = reinterpret_cast<const InfResult *>(sl.h + job.off_res);
    std::vector<size_t> bad;
    for (size_t i = 0; i < job.inf.size(); ++i)
        if (res[i].status != dinf::kOk) bad.push_back(i);
    st->n_device_chunks += int64_t(job.inf.size() - bad.size());
    if (bad.empty()) return ATL_OK;
    if (getenv("ATLITE_HIP_INGEST_DEBUG")) {  // which verdicts, of which variables
        int hist[16] = {0};
        for (size_t i : bad) ++hist[res[i].status & 15];
        fprintf(stderr, "[atlite-hip ingest] %zu of %zu streams go back to the host decoders; status histogram:", bad.size(), job.inf.size());
        for (int k = 0; k < 16; ++k)
            if (hist[k]) fprintf(stderr, " %d x status %d", hist[k], k);
        float t_k = -1.f, t_c = -1.f;
        (void)hipEventElapsedTime(&t_k, sl.ev_t[1], sl.ev_t[2]);
        (void)hipEventElapsedTime(&t_c, sl.ev_c[0], sl.ev_c[1]);
        (void)hipGetLastError();
        const uint32_t *hf = reinterpret_cast<const uint32_t *>(sl.h + job.off_flag);
        fprintf(stderr, "; first: stream %zu of '%s' (%lld -> %lld bytes, batch %u, its flag now %u); kernel span %.1f ms, DMA span %.1f ms\n", bad[0],
                job.parts[job.part_of[bad[0]]].ds->name.c_str(), (long long)job.inf[bad[0]].src_n, (long long)job.inf[bad[0]].dst_n,
                job.inf[bad[0]].batch, job.inf[bad[0]].batch == kNoWait ? 0u : hf[size_t(job.inf[bad[0]].batch) * kFlagPitch], t_k, t_c);
    }
    // streams the device decoder did not accept (corrupt, a shape of code it declines, or bytes that never arrived): the host
    // decoders decide, the chunk is decoded again from their output; a stream they reject too is the caller's error, as on the
    // host path
    ATL_HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint8_t> tmp;
    const UnpackDesc *d_unp = reinterpret_cast<const UnpackDesc *>(sl.d + job.off_unp);
    for (size_t i : bad) {
        const Part &pt = job.parts[job.part_of[i]];
        tmp.resize(size_t(pt.chunk_bytes));
        bool shuffled = false;
        const h5::Chunk &c = pt.ds->chunks[job.lin[i]];
        const int e = h5::chunk_inflate(*pt.ds, c, job.nc->file.base(), -1, tmp.data(), uint64_t(pt.chunk_bytes), &shuffled);
        if (e) return e;
        {
            const int rc = h2d(ctx, sl.st, sl.d_raw + job.inf[i].dst_off, tmp.data(), size_t(pt.chunk_bytes));
            if (rc) return rc;
        }
        launch_unpack(sl.st, sl.d_raw, d_unp + job.desc_of[i], 1, pt.p, pt.max_elems, pt.d_out);
        ATL_HIP_TRY(hipGetLastError());
        ++st->n_redone;
    }
    ATL_HIP_TRY(hipStreamSynchronize(sl.st));
    return ATL_OK;
}

// Settle a slot.  A failure is KEPT in the state (keep_verdict) as well as returned: callers that cannot report it - atl_nc_close,
// a context being recycled - drop the return value, and the next caller that observes the copy stream gets it (take_verdict).
int finish_slot(atl_ctx *ctx, IngestState *st, Slot &sl) { return keep_verdict(st, finish_slot_impl(ctx, st, sl)); }

int copy_text(const std::string &s, char *buf, int64_t buflen, int64_t *needed) {
    if (needed) *needed = int64_t(s.size()) + 1;
    if (buf && buflen > 0) {
        const size_t n = std::min<size_t>(s.size(), size_t(buflen - 1));
        memcpy(buf, s.data(), n);
        buf[n] = '\0';
    }
    return ATL_OK;
}

}  // namespace

namespace {

// geometry + decoding + the (rows x cells) block's row stride of one variable of a read
int part_setup(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0, const char *who, const Dataset **d_out,
               Geometry *g, UnpackParams *p) {
    const Dataset *d;
    int rc = lookup(f, name, &d, who);
    if (rc) return rc;
    rc = geometry_of(*d, g, who);
    if (rc) return rc;
    ATL_REQUIRE(start0 >= 0 && count0 >= 0 && start0 + count0 <= g->shape[0], "%s: rows [%lld, %lld) outside '%s' (%lld rows)", who,
                (long long)start0, (long long)(start0 + count0), name, (long long)g->shape[0]);
    *p = UnpackParams{};
    p->shape1 = g->shape[1];
    p->shape2 = g->shape[2];
    p->ld = g->shape[1] * g->shape[2];
    if (ctx->slot_stride > 0 && d->shape.size() == 3) {
        ATL_REQUIRE(ctx->slot_stride >= p->ld, "%s: slot stride %lld is smaller than a row of %lld cells", who,
                    (long long)ctx->slot_stride, (long long)p->ld);
        p->ld = ctx->slot_stride;
    }
    p->r0 = start0;
    p->r1 = start0 + count0;
    p->dec = decode_of(f->file, *d);
    *d_out = d;
    return ATL_OK;
}

// Streams per ... : the DMA batches of a job.  A batch is a run of consecutive streams of ~128 MiB of compressed bytes
// ($ATLITE_HIP_INGEST_BATCH): pread into one ring buffer by the host threads, one DMA, one flag.
struct Batch {
    size_t first = 0, count = 0;   // streams
    size_t off = 0, bytes = 0;     // inside the job's compressed area
};

// Streams whose blocks are worth decoding side by side: fewer streams than the device has wave slots for whole-stream decoding
// to fill it, and long ones.
bool split_wanted(const std::vector<InfDesc> &inf) {
    const int e = split_env();
    if (e >= 0) return e != 0;
    if (inf.empty() || inf.size() >= kSplitMaxStreams) return false;
    double bytes = 0;
    for (const InfDesc &q : inf) bytes += double(q.dst_n);
    return bytes / double(inf.size()) >= kSplitMinBytes;
}

bool fed_mode() {
    const char *e = getenv("ATLITE_HIP_INGEST_FED");
    return !(e && strcmp(e, "0") == 0);
}

// One JOB of the device-inflate path: the rows [start0, start0 + count0) of n_vars variables, every chunk stream of them in ONE
// k_inflate launch on a staging slot's own stream.  The launch is FED (k_inflate's comment): it is enqueued first, then the
// compressed bytes follow batch by batch - pread by the host threads into a page-locked ring, DMA'd on the slot's copy stream,
// each batch's arrival flag behind it - so that pread, DMA, inflate, checksum and unpack of the job overlap.  The rows land in
// blocks that begin at row `base_row` (the whole read's first row).  *done = false (nothing enqueued): not every stored chunk
// is a plain zlib stream - the caller reads these rows variable by variable.
int read_group_device(atl_ctx *ctx, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0, int64_t base_row,
                      double *const *d_outs, int n_threads, bool *done) {
    *done = false;
    if (count0 == 0) return ATL_OK;
    std::vector<Part> parts;
    std::vector<UnpackDesc> descs, missing;
    std::vector<uint32_t> missing_part;
    std::vector<InfDesc> inf;
    std::vector<size_t> lin;
    std::vector<uint32_t> part_of, desc_of;
    size_t raw_off = 0, comp_off = 0;
    bool big_chunks = false;
    for (int v = 0; v < n_vars; ++v) {
        const Dataset *d;
        Geometry g;
        Part pt;
        int rc = part_setup(ctx, f, names[v], start0, count0, "atl_nc_read_slabs", &d, &g, &pt.p);
        if (rc) return rc;
        if (g.row_elems == 0) continue;
        ATL_REQUIRE(d_outs[v], "atl_nc_read_slabs: d_out of '%s' is NULL", names[v]);
        if (d->layout != 2) return ATL_OK;  // compact / contiguous: the host path
        const int64_t chunk_bytes = g.chunk_elems * pt.p.dec.esize;
        // whole-stream waves: chunks up to 64 MiB (wave_adler's sums); decoded block by block (split): up to 1 GiB - an 800 x 800
        // grid in atlite's (100, y, x) chunks is 256 MB a chunk
        if (chunk_bytes > (int64_t(1) << 30)) return ATL_OK;
        if (chunk_bytes > (int64_t(64) << 20)) big_chunks = true;
        Selection sel;
        select_chunks(g, pt.p.r0, pt.p.r1, chunk_bytes, &sel);
        pt.ds = d;
        pt.d_out = d_outs[v] + (start0 - base_row) * pt.p.ld;
        pt.chunk_bytes = chunk_bytes;
        pt.max_elems = g.chunk_elems;
        pt.desc0 = descs.size();
        pt.n_desc = sel.desc.size();
        for (size_t i = 0; i < sel.lin.size(); ++i) {
            const h5::Chunk &c = d->chunks[sel.lin[i]];
            UnpackDesc ds = sel.desc[i];
            ds.src_off = int64_t(raw_off);  // where the chunk's inflated bytes go in the slot's raw buffer
            if (c.size == 0) {
                ds.missing = 1;
                missing.push_back(ds);
                missing_part.push_back(uint32_t(parts.size()));
            } else {
                raw_off += align_up(size_t(chunk_bytes), 16);
                uint64_t pay = 0;
                bool defl = false, shuf = false;
                if (h5::chunk_filters(*d, c, &pay, &defl, &shuf) != ATL_OK || !defl || pay < 6 || c.addr > f->file.size() ||
                    c.size > f->file.size() - c.addr)
                    return ATL_OK;
                ds.shuffled = shuf;
                InfDesc q{};
                q.src_off = int64_t(comp_off);
                q.src_n = int64_t(pay);
                q.dst_off = ds.src_off;
                q.dst_n = chunk_bytes;
                q.part = uint32_t(parts.size());
                q.desc = uint32_t(descs.size());
                q.batch = kNoWait;
                inf.push_back(q);
                comp_off += align_up(size_t(pay) + 8, 128);  // (+ the bit reader's look-ahead words; a 128-byte line of its own)
                lin.push_back(sel.lin[i]);
                part_of.push_back(uint32_t(parts.size()));
                desc_of.push_back(uint32_t(descs.size()));
            }
            descs.push_back(ds);
        }
        parts.push_back(pt);
    }
    if (parts.empty() || inf.empty()) return ATL_OK;
    // few, long streams (atlite's own cutouts: (time = 100, y, x) chunks): their blocks are decoded side by side (launch_split below)
    const bool split = split_wanted(inf);
    if (big_chunks && !split) return ATL_OK;  // (the host path)
    const bool fed = fed_mode() && !split;
    // batches of consecutive streams
    size_t batch_bytes = size_t(128) << 20;
    if (const char *e = getenv("ATLITE_HIP_INGEST_BATCH")) batch_bytes = size_t(std::max(1, atoi(e))) << 20;
    std::vector<Batch> batches;
    for (size_t i = 0; i < inf.size(); ++i) {
        const size_t len = align_up(size_t(inf[i].src_n) + 8, 128);
        if (batches.empty() || batches.back().bytes + len > batch_bytes) {
            Batch nb;
            nb.first = i;
            nb.off = size_t(inf[i].src_off);
            batches.push_back(nb);
        }
        batches.back().count += 1;
        batches.back().bytes += len;
        if (fed) inf[i].batch = uint32_t(batches.size() - 1);
    }
    size_t ring_bytes = 0;
    for (const Batch &b : batches) ring_bytes = std::max(ring_bytes, b.bytes);
    // device: streams | InfDesc[] | UnpackDesc[] (the chunks', then the never-written ones') | FedPart[] | flags | InfResult[]
    // page-locked: the same from InfDesc[] on (off_meta = 0 there)
    const size_t n = inf.size(), nd = descs.size(), nm = missing.size(), np = parts.size(), nb = batches.size();
    const size_t m_inf = 0, m_unp = align_up(m_inf + n * sizeof(InfDesc), 256), m_mis = m_unp + nd * sizeof(UnpackDesc),
                 m_part = align_up(m_mis + nm * sizeof(UnpackDesc), 256), m_flag = align_up(m_part + np * sizeof(FedPart), 256),
                 m_res = align_up(m_flag + nb * kFlagPitch * sizeof(uint32_t), 256), m_end = m_res + n * sizeof(InfResult);
    const size_t off_meta = align_up(comp_off, 256);
    // split jobs: spans of the finder | its slots | the streams' split points | segment tasks | their results | segment bounds | ResDesc[]
    // (behind the rest, same offsets on both sides), and the mark plane behind the inflated chunks
    size_t n_spans = 0;
    if (split)
        for (const InfDesc &q : inf) {
            const uint64_t bits = uint64_t(q.src_n) * 8;
            if (bits > 17 + 128) n_spans += size_t((bits - 17 + kSpanBits - 1) / kSpanBits);
        }
    const size_t t_max = split ? n + n_spans * kSpanSlots : 0;
    const size_t s_span = align_up(m_end, 256), s_out = align_up(s_span + n_spans * sizeof(FindSpan), 256),
                 s_cand = align_up(s_out + n_spans * kSpanWords * sizeof(uint32_t), 256), s_task = align_up(s_cand + t_max * sizeof(uint32_t), 256),
                 s_res = align_up(s_task + t_max * sizeof(SegTask), 256), s_bound = align_up(s_res + t_max * sizeof(SegRes), 256),
                 s_rd = align_up(s_bound + (t_max + n) * sizeof(uint32_t), 256), s_end = split ? s_rd + n * sizeof(ResDesc) : m_end;
    const size_t mark_off = align_up(raw_off + 256, 256);
    Slot *sl = nullptr;
    int rc = slot_acquire(ctx, s_end + 256, off_meta + s_end + 256, raw_off + 256, true, &sl);  // (count + decode: the mark plane is added when needed)
    if (rc) return rc;
    rc = ring_reserve(ctx, ring_bytes);  // (every allocation of the job happens before its kernel is launched)
    if (rc) return rc;
    IngestState *state = state_of(ctx);
    memcpy(sl->h + m_inf, inf.data(), n * sizeof(InfDesc));
    memcpy(sl->h + m_unp, descs.data(), nd * sizeof(UnpackDesc));
    if (nm) memcpy(sl->h + m_mis, missing.data(), nm * sizeof(UnpackDesc));
    {
        FedPart *fp = reinterpret_cast<FedPart *>(sl->h + m_part);
        for (size_t k = 0; k < np; ++k) {
            fp[k].p = parts[k].p;
            fp[k].out = parts[k].d_out;
        }
    }
    memset(sl->h + m_flag, 0, nb * kFlagPitch * sizeof(uint32_t));
    {  // verdicts: "not run" until the device says otherwise (a launch that never happens must not read as success)
        InfResult *r = reinterpret_cast<InfResult *>(sl->h + m_res);
        for (size_t k = 0; k < n; ++k) r[k] = InfResult{int32_t(dinf::kNotRun), 0u};
    }
    // ---- streams and events of the slot ------------------------------------------------------------------------------------
    hipStream_t cs;
    rc = copy_stream_of(ctx, &cs);
    if (rc) return rc;
    if (!sl->st) ATL_HIP_TRY(hipStreamCreateWithFlags(&sl->st, hipStreamNonBlocking));
    if (!sl->st_c) {  // a priority of its own = a hardware queue of its own (k_inflate's comment)
        int least = 0, greatest = 0;
        ATL_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        ATL_HIP_TRY(hipStreamCreateWithPriority(&sl->st_c, hipStreamNonBlocking, greatest));
    }
    if (!sl->ev_fork[0]) ATL_HIP_TRY(hipEventCreateWithFlags(&sl->ev_fork[0], hipEventDisableTiming));
    if (!sl->ev_meta) ATL_HIP_TRY(hipEventCreateWithFlags(&sl->ev_meta, hipEventDisableTiming));
    for (hipEvent_t &e : sl->ev_t)
        if (!e) ATL_HIP_TRY(hipEventCreate(&e));
    for (hipEvent_t &e : sl->ev_c)
        if (!e) ATL_HIP_TRY(hipEventCreate(&e));
    while (sl->ev_batch.size() < nb) {
        hipEvent_t e = nullptr;
        ATL_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sl->ev_batch.push_back(e);
    }
    // behind whatever the copy stream holds (an earlier read into the same block; what the caller ordered the copy stream
    // after: Context.copy_after_compute).  NOT behind the compute stream: a slab pipeline reads the next slab while the
    // previous one is converted
    ATL_HIP_TRY(hipEventRecord(sl->ev_fork[0], cs));
    ATL_HIP_TRY(hipStreamWaitEvent(sl->st, sl->ev_fork[0], 0));
    uint8_t *d_meta = sl->d + off_meta;
    const InfDesc *d_inf = reinterpret_cast<const InfDesc *>(d_meta + m_inf);
    const UnpackDesc *d_unp = reinterpret_cast<const UnpackDesc *>(d_meta + m_unp);
    const UnpackDesc *d_mis = reinterpret_cast<const UnpackDesc *>(d_meta + m_mis);
    const FedPart *d_part = reinterpret_cast<const FedPart *>(d_meta + m_part);
    uint32_t *h_flag = reinterpret_cast<uint32_t *>(sl->h + m_flag);  // page-locked: the CPU stores, the waves poll (k_inflate's comment)
    InfResult *d_res = reinterpret_cast<InfResult *>(d_meta + m_res);
    ATL_HIP_TRY(hipEventRecord(sl->ev_t[0], sl->st));
    ATL_HIP_TRY(hipMemcpyAsync(d_meta, sl->h, m_end, hipMemcpyHostToDevice, sl->st));  // descriptors, cleared flags, "not run"
    ATL_HIP_TRY(hipEventRecord(sl->ev_meta, sl->st));
    ATL_HIP_TRY(hipEventRecord(sl->ev_t[1], sl->st));
    ATL_HIP_TRY(hipStreamWaitEvent(sl->st_c, sl->ev_meta, 0));  // (a flag must not land before the flags are cleared)
    auto launch = [&]() -> int {
        unsigned long long ticks = 2000000000ull;  // 20 s of the 100 MHz clock
        if (const char *e = getenv("ATLITE_HIP_INGEST_TIMEOUT_MS")) ticks = (unsigned long long)(std::max(1e-5, atof(e)) * 100000.0);  // (fractions allowed: tests)
        hipLaunchKernelGGL(k_inflate, dim3(unsigned(n)), dim3(64), 0, sl->st, sl->d, d_inf, sl->d_raw, d_res, h_flag, d_part, d_unp, ticks);
        ATL_HIP_TRY(hipGetLastError());
        ATL_HIP_TRY(hipEventRecord(sl->ev_t[2], sl->st));
        for (size_t k = 0; k < nm;) {  // never-written chunks: fill value / NaN (runs of one variable)
            size_t e = k;
            while (e < nm && missing_part[e] == missing_part[k]) ++e;
            const Part &pt = sl->job.parts[missing_part[k]];  // (the job owns the parts by the time this runs)
            launch_unpack(sl->st, sl->d_raw, d_mis + k, e - k, pt.p, pt.max_elems, pt.d_out);
            k = e;
        }
        ATL_HIP_TRY(hipGetLastError());
        ATL_HIP_TRY(hipEventRecord(sl->ev_t[3], sl->st));
        return ATL_OK;
    };
    // the job is the slot's from here on: whatever happens below, the slot must be settled before it is used again
    Pending &job = sl->job;
    job.nc = f;
    job.parts = std::move(parts);
    job.part_of = std::move(part_of);
    job.lin = std::move(lin);
    job.desc_of = std::move(desc_of);
    job.inf = std::move(inf);
    job.off_unp = off_meta + m_unp;
    job.off_res = m_res;
    job.off_res_dev = off_meta + m_res;
    job.off_flag = m_flag;
    job.aborted = false;
    // split jobs: the finder's spans (64 KiB of a stream each) go up with the descriptors; a batch's spans are searched as soon as
    // its DMA has landed, while the next batch is read and copied
    std::vector<size_t> span0;  // the streams' spans
    if (split) {
        span0.assign(n + 1, 0);
        FindSpan *h_span = reinterpret_cast<FindSpan *>(sl->h + s_span);
        size_t k = 0;
        for (size_t i = 0; i < n; ++i) {
            span0[i] = k;
            const uint64_t bits = uint64_t(job.inf[i].src_n) * 8;
            if (bits > 17 + 128)
                for (uint64_t b = 17; b < bits; b += kSpanBits)
                    h_span[k++] = FindSpan{uint32_t(i), uint32_t(b), uint32_t(std::min<uint64_t>(kSpanBits, bits - b)), 0u};
        }
        span0[n] = k;
        ATL_REQUIRE(k == n_spans, "atl_nc_read_slabs: span count");
        if (n_spans) {
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_span, h_span, n_spans * sizeof(FindSpan), hipMemcpyHostToDevice, sl->st));
            ATL_HIP_TRY(hipMemsetAsync(d_meta + s_out, 0, n_spans * kSpanWords * sizeof(uint32_t), sl->st));
        }
    }
    auto find_batch = [&](size_t b) -> int {  // (on the kernel stream, behind batch b's DMA)
        const size_t k0 = span0[batches[b].first], k1 = span0[batches[b].first + batches[b].count];
        if (k1 == k0) return ATL_OK;
        ATL_HIP_TRY(hipStreamWaitEvent(sl->st, sl->ev_batch[b], 0));
        hipLaunchKernelGGL(k_find_blocks, dim3(unsigned(k1 - k0)), dim3(256), 0, sl->st, sl->d, d_inf, reinterpret_cast<const FindSpan *>(d_meta + s_span) + k0,
                           reinterpret_cast<uint32_t *>(d_meta + s_out) + k0 * kSpanWords);
        ATL_HIP_TRY(hipGetLastError());
        return ATL_OK;
    };
    // The split order, COUNT + DECODE variant ($ATLITE_HIP_SPLIT_PASSES=2, or no room for the one-pass variant's pool - op_* below):
    // a batch's block headers are found as soon as its bytes are on the device (find_batch, behind the DMAs); when everything has
    // arrived the segments are counted, the chains followed (host), the segments decoded and resolved (split_stage over all streams).  The host reads the device's lists between the passes, so the stage blocks the calling
    // thread.  (split_stage takes a range of streams: a stage per batch, overlapping the next batch's DMA, was measured -
    // T = 2000 of 16 MB chunks 0.190 s in 256 MiB batches, 0.158 in 512 MiB, 0.132 s as ONE stage: a launch lasts as long as its
    // longest segment, a lone wave makes ~10 MB/s, and every batch has a long segment or two.)  A stream whose chain does not
    // close keeps its "not run" verdict and goes to the host decoders with the other declined streams (finish_slot).
    size_t so_cand = 0, so_task = 0, so_bound = 0, so_rd = 0;  // what the stages so far have used of the lists
    const bool dbg = getenv("ATLITE_HIP_INGEST_DEBUG") != nullptr;
    auto split_stage = [&](size_t i0, size_t i1) -> int {  // streams [i0, i1): their finder has been enqueued
        hipStream_t q = sl->st;
        const std::vector<InfDesc> &jn = job.inf;
        uint8_t *d_mark = nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
        const auto t_begin = now();
        uint32_t *h_out = reinterpret_cast<uint32_t *>(sl->h + s_out);
        const size_t k0 = span0[i0], k1 = span0[i1];
        if (k1 > k0)
            ATL_HIP_TRY(hipMemcpyAsync(h_out + k0 * kSpanWords, d_meta + s_out + k0 * kSpanWords * sizeof(uint32_t), (k1 - k0) * kSpanWords * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, q));
        ATL_HIP_TRY(hipStreamSynchronize(q));
        const double ms_find = ms_since(t_begin);
        const auto t_count = now();
        // the streams' split points (ascending) and one task per start
        uint32_t *h_cand = reinterpret_cast<uint32_t *>(sl->h + s_cand);
        SegTask *h_task = reinterpret_cast<SegTask *>(sl->h + s_task);
        const size_t ni = i1 - i0, c_base = so_cand, t_base = so_task;
        std::vector<size_t> task0(ni + 1, 0), cand0(ni + 1, 0);
        size_t nc = c_base, ntask = t_base;
        for (size_t i = i0; i < i1; ++i) {
            cand0[i - i0] = nc;
            task0[i - i0] = ntask;
            for (size_t k = span0[i]; k < span0[i + 1]; ++k) {
                const uint32_t cnt = std::min(h_out[k * kSpanWords], kSpanSlots);
                uint32_t *c = h_out + k * kSpanWords + 1;
                std::sort(c, c + cnt);
                for (uint32_t j = 0; j < cnt; ++j) h_cand[nc++] = c[j];
            }
            const uint32_t ncs = uint32_t(nc - cand0[i - i0]), c0 = uint32_t(cand0[i - i0]);
            h_task[ntask] = SegTask{uint32_t(i), 16u, 0u, c0, ncs, uint32_t(ntask - t_base), 0u};
            ++ntask;
            for (uint32_t j = 0; j < ncs; ++j) {
                h_task[ntask] = SegTask{uint32_t(i), h_cand[c0 + j], dinf::kSegSlack, c0, ncs, uint32_t(ntask - t_base), 0u};
                ++ntask;
            }
        }
        cand0[ni] = nc;
        task0[ni] = ntask;
        ATL_REQUIRE(ntask <= t_max && nc <= t_max, "atl_nc_read_slabs: segment lists");
        const size_t n_count = ntask - t_base;
        SegRes *h_res = reinterpret_cast<SegRes *>(sl->h + s_res);
        const SegTask *d_task = reinterpret_cast<const SegTask *>(d_meta + s_task);
        const uint32_t *d_cand = reinterpret_cast<const uint32_t *>(d_meta + s_cand);
        SegRes *d_sres = reinterpret_cast<SegRes *>(d_meta + s_res);
        if (nc > c_base)
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_cand + c_base * sizeof(uint32_t), h_cand + c_base, (nc - c_base) * sizeof(uint32_t), hipMemcpyHostToDevice, q));
        // longest first (a launch is as long as its last wave): the compressed bits up to the next split point stand for the work.
        // The sorted copy is what goes up (in the decode tasks' place: page-locked, free until the chains are known)
        SegTask *h_sorted = reinterpret_cast<SegTask *>(sl->h + s_res) + t_base;  // (SegRes is not smaller than SegTask; results land behind the sync)
        {
            std::vector<std::pair<uint64_t, uint32_t>> order(n_count);
            for (size_t t = 0; t < n_count; ++t) {
                const SegTask &tk = h_task[t_base + t];
                const bool last_of_stream = t + 1 == n_count || h_task[t_base + t + 1].stream != tk.stream;
                const uint64_t end = last_of_stream ? uint64_t(jn[tk.stream].src_n) * 8 : uint64_t(h_task[t_base + t + 1].start_bit);
                order[t] = {end - tk.start_bit, uint32_t(t)};
            }
            std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (size_t t = 0; t < n_count; ++t) h_sorted[t] = h_task[t_base + order[t].second];
        }
        ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_task + t_base * sizeof(SegTask), h_sorted, n_count * sizeof(SegTask), hipMemcpyHostToDevice, q));
        {  // the mark plane behind the inflated chunks (nothing has been written there yet)
            if (sl->raw_bytes < 2 * mark_off) {
                ATL_HIP_TRY(hipStreamSynchronize(q));
                if (sl->d_raw) (void)dev_free(sl->d_raw);
                sl->d_raw = nullptr;
                sl->raw_bytes = 0;
                ATL_HIP_TRY(dev_malloc(reinterpret_cast<void **>(&sl->d_raw), 2 * mark_off));
                sl->raw_bytes = 2 * mark_off;
            }
            d_mark = sl->d_raw + mark_off;
        }
        hipLaunchKernelGGL((k_segments<true>), dim3(unsigned(n_count)), dim3(64), 0, q, sl->d, d_inf, d_task + t_base, d_cand, sl->d_raw, d_mark, d_sres + t_base);
        ATL_HIP_TRY(hipGetLastError());
        ATL_HIP_TRY(hipMemcpyAsync(h_res + t_base, d_sres + t_base, n_count * sizeof(SegRes), hipMemcpyDeviceToHost, q));
        ATL_HIP_TRY(hipStreamSynchronize(q));
        const double ms_count = ms_since(t_count);
        const auto t_chain = now();
        // the chains: from the stream's first block, every segment ends where the next one starts; output positions on the way
        std::vector<SegTask> run;
        std::vector<uint64_t> run_len;
        std::vector<uint32_t> bounds;
        std::vector<ResDesc> rds;
        for (size_t i = i0; i < i1; ++i) {
            const size_t r0 = run.size(), b0 = bounds.size(), li = i - i0;
            uint64_t at = 0;
            bool ok = true;
            uint32_t want = 0;
            for (size_t t = task0[li];;) {
                const SegRes &r = h_res[t];
                if (r.status != dinf::kOk || r.o.out_end == 0 || at + r.o.out_end > uint64_t(jn[i].dst_n) || run.size() - r0 > size_t(task0[li + 1] - task0[li])) {
                    ok = false;
                    break;
                }
                SegTask tk = h_task[t];
                tk.seg0 = at;
                tk.slack = uint32_t(std::min<uint64_t>(dinf::kSegSlack, at));
                tk.res_ix = uint32_t(run.size());
                run.push_back(tk);
                run_len.push_back(r.o.out_end);
                bounds.push_back(uint32_t(at));
                at += r.o.out_end;
                if (r.o.is_final) {
                    want = r.o.adler;
                    break;
                }
                const uint32_t *c = h_cand + cand0[li], *ce = h_cand + cand0[li + 1];
                const uint32_t *it = std::lower_bound(c, ce, uint32_t(r.o.end_bit));
                if (it == ce || *it != uint32_t(r.o.end_bit)) {
                    ok = false;
                    break;
                }
                t = task0[li] + 1 + size_t(it - c);
            }
            if (!ok || at != uint64_t(jn[i].dst_n)) {  // the host decoders take the stream
                run.resize(r0);
                run_len.resize(r0);
                bounds.resize(b0);
                continue;
            }
            bounds.push_back(uint32_t(at));
            rds.push_back(ResDesc{uint32_t(i), uint32_t(so_bound + b0), uint32_t(run.size() - r0), want});
        }
        state->ms[3] += double(run.size());  // (a count: atl_nc_ingest_times)
        const double ms_chain = ms_since(t_chain);
        if (dbg)
            fprintf(stderr, "[atlite-hip ingest] split: streams %zu .. %zu, %zu spans, %zu candidate headers, %zu streams chained into %zu segments; "
                    "find (wait) %.1f ms, count pass %.1f ms, chains (host) %.1f ms\n", i0, i1, k1 - k0, nc - c_base, rds.size(), run.size(), ms_find, ms_count,
                    ms_chain);
        if (!run.empty()) {
            // the decode pass's tasks take the place of the count pass's (which nobody reads again)
            ATL_REQUIRE(run.size() <= n_count && so_bound + bounds.size() <= t_max + n && so_rd + rds.size() <= n, "atl_nc_read_slabs: segment lists");
            {  // longest first, as the count pass (res_ix: results in chain order; a stream dropped above leaves a gap in the numbering)
                std::vector<uint32_t> order(run.size());
                for (size_t t = 0; t < run.size(); ++t) {
                    order[t] = uint32_t(t);
                    run[t].res_ix = uint32_t(t);
                }
                std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return run_len[a] > run_len[b]; });
                for (size_t t = 0; t < run.size(); ++t) h_task[t_base + t] = run[order[t]];
            }
            uint32_t *h_bound = reinterpret_cast<uint32_t *>(sl->h + s_bound);
            ResDesc *h_rd = reinterpret_cast<ResDesc *>(sl->h + s_rd);
            memcpy(h_bound + so_bound, bounds.data(), bounds.size() * sizeof(uint32_t));
            memcpy(h_rd + so_rd, rds.data(), rds.size() * sizeof(ResDesc));
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_task + t_base * sizeof(SegTask), h_task + t_base, run.size() * sizeof(SegTask), hipMemcpyHostToDevice, q));
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_bound + so_bound * sizeof(uint32_t), h_bound + so_bound, bounds.size() * sizeof(uint32_t), hipMemcpyHostToDevice, q));
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_rd + so_rd * sizeof(ResDesc), h_rd + so_rd, rds.size() * sizeof(ResDesc), hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL((k_segments<false>), dim3(unsigned(run.size())), dim3(64), 0, q, sl->d, d_inf, d_task + t_base, d_cand, sl->d_raw, d_mark, d_sres + t_base);
            ATL_HIP_TRY(hipGetLastError());
            const auto t_dec = now();
            if (dbg) {  // the decode pass must end where the count pass did
                ATL_HIP_TRY(hipMemcpyAsync(h_res + t_base, d_sres + t_base, run.size() * sizeof(SegRes), hipMemcpyDeviceToHost, q));
                ATL_HIP_TRY(hipStreamSynchronize(q));
                size_t bad = 0;
                for (size_t t = 0; t < run.size(); ++t) bad += h_res[t_base + t].status != dinf::kOk;
                fprintf(stderr, "[atlite-hip ingest] split: decode pass %.1f ms, %zu of %zu segments with a status\n", ms_since(t_dec), bad, run.size());
            }
            hipLaunchKernelGGL(k_resolve, dim3(unsigned(rds.size())), dim3(1024), 0, q, d_inf, reinterpret_cast<const ResDesc *>(d_meta + s_rd) + so_rd,
                               reinterpret_cast<const uint32_t *>(d_meta + s_bound), sl->d_raw, d_mark, d_res);
            ATL_HIP_TRY(hipGetLastError());
        }
        so_cand = nc;
        so_task = ntask;
        so_bound += bounds.size();
        so_rd += rds.size();
        return ATL_OK;
    };
    auto split_finish = [&]() -> int {
        hipStream_t q = sl->st;
        ATL_HIP_TRY(hipEventRecord(sl->ev_t[2], q));
        for (const Part &pt : job.parts)  // (a stream the host decoders take is unpacked again behind them)
            if (pt.n_desc) launch_unpack(q, sl->d_raw, d_unp + pt.desc0, pt.n_desc, pt.p, pt.max_elems, pt.d_out);
        ATL_HIP_TRY(hipGetLastError());
        ATL_HIP_TRY(hipEventRecord(sl->ev_t[3], q));
        return ATL_OK;
    };
    // ---- ... the one-pass variant as stages on a second stream: op_launch (tasks from the finder's lists, the kernel, its results on
    // the way back) for a range of streams, then the chains of all stages are followed and ONE k_gather places them (op_finish).
    // The pool of output regions is reserved for the whole job up front (op_prepare; no room: the single stage above, count +
    // decode).  ONE stage, when the last batch has landed, is what runs.  Launching the first half's segments when half of the
    // batches are on the device - to decode inside the DMAs - was measured twice (T = 2000; the knob is gone, tools/jobs/r06_early.sh):
    // with the results' copy queued behind the early kernel 0.118-0.121 s against 0.096-0.101 - that copy sits at the head of a
    // copy engine's queue until the kernel ends, and the later batches' DMAs wait behind it (first to last 72 ms instead of 21);
    // with the results fetched at the end 0.107-0.108 s against 0.096-0.102: the DMAs run on, but two launches have two tails, the
    // finder's workgroups (46 kB of LDS) wait for the decode kernel's CUs, and the kernel is slower beside them.
    struct OpStage {
        size_t i0, i1, t_base, n_task, n_headers = 0;
        std::vector<size_t> task0, cand0;
    };
    std::vector<OpStage> op_stages;
    bool op_ready = false;
    PoolRef op_pr{};
    size_t op_gseg = 0, op_rd = 0, op_acc = 0, op_piece = 0, op_piece_cap = 0;
    auto op_prepare = [&]() -> int {
        if (!split) return ATL_OK;
        if (const char *e = getenv("ATLITE_HIP_SPLIT_PASSES"))
            if (atoi(e) == 2) return ATL_OK;
        double out_bytes = 0;
        for (const InfDesc &q : job.inf) out_bytes += double(q.dst_n);
        // 1.75 x the output (a segment's last region is at most half used) + a first region for every task there may be
        size_t cap_regions = size_t((1.75 * out_bytes) / 8192.0) + (n + 4 * n_spans) + 1024;
        if (const char *e = getenv("ATLITE_HIP_SPLIT_POOL_PERCENT"))  // tests: a pool that runs out (its streams go to the host decoders)
            cap_regions = std::max<size_t>(16, cap_regions * size_t(std::max(1, atoi(e))) / 100);
        if (cap_regions >= (size_t(1) << 32)) return ATL_OK;
        const size_t b_pool = cap_regions * 8192 * sizeof(uint16_t), off_regs = align_up(b_pool, 256),
                     off_next = align_up(off_regs + t_max * kMaxRegions * sizeof(uint32_t), 256);
        op_gseg = off_next + 256;
        op_rd = align_up(op_gseg + (t_max + n + 1) * sizeof(GatherSeg), 256);
        op_acc = align_up(op_rd + (n + 1) * sizeof(ResDesc), 256);
        op_piece = align_up(op_acc + 2 * (n + 1) * sizeof(unsigned long long), 256);
        op_piece_cap = t_max + n + size_t(out_bytes / double(kRestPiece)) + 16;
        const size_t need = op_piece + op_piece_cap * sizeof(GatherPiece) + 256;
        if (sl->pool_bytes < need) {
            if (sl->d_pool) (void)dev_free(sl->d_pool);
            sl->d_pool = nullptr;
            sl->pool_bytes = 0;
            if (dev_malloc(reinterpret_cast<void **>(&sl->d_pool), need) != hipSuccess) {
                (void)hipGetLastError();
                sl->d_pool = nullptr;
                return ATL_OK;  // (no room: count + decode at the end)
            }
            sl->pool_bytes = need;
        }
        if (!sl->st2) ATL_HIP_TRY(hipStreamCreateWithFlags(&sl->st2, hipStreamNonBlocking));
        if (!sl->ev_st2) ATL_HIP_TRY(hipEventCreateWithFlags(&sl->ev_st2, hipEventDisableTiming));
        op_pr.pool = reinterpret_cast<uint16_t *>(sl->d_pool);
        op_pr.regs = reinterpret_cast<uint32_t *>(sl->d_pool + off_regs);
        op_pr.next = reinterpret_cast<uint32_t *>(sl->d_pool + off_next);
        op_pr.cap = uint32_t(cap_regions);
        ATL_HIP_TRY(hipStreamWaitEvent(sl->st2, sl->ev_meta, 0));  // (the descriptors)
        ATL_HIP_TRY(hipMemsetAsync(op_pr.next, 0, 256, sl->st2));
        op_ready = true;
        return ATL_OK;
    };
    auto op_launch = [&](size_t i0, size_t i1) -> int {  // streams [i0, i1): their bytes are on the device, their finder enqueued
        if (i1 <= i0) return ATL_OK;
        const std::vector<InfDesc> &jn = job.inf;
        uint32_t *h_out = reinterpret_cast<uint32_t *>(sl->h + s_out);
        const size_t k0 = span0[i0], k1 = span0[i1];
        if (k1 > k0)
            ATL_HIP_TRY(hipMemcpyAsync(h_out + k0 * kSpanWords, d_meta + s_out + k0 * kSpanWords * sizeof(uint32_t), (k1 - k0) * kSpanWords * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, sl->st));
        ATL_HIP_TRY(hipStreamSynchronize(sl->st));  // the finder's lists
        uint32_t *h_cand = reinterpret_cast<uint32_t *>(sl->h + s_cand);
        SegTask *h_task = reinterpret_cast<SegTask *>(sl->h + s_task);
        OpStage stg;
        stg.i0 = i0;
        stg.i1 = i1;
        stg.t_base = so_task;
        const size_t ni = i1 - i0, c_base = so_cand, t_base = so_task;
        stg.task0.assign(ni + 1, 0);
        stg.cand0.assign(ni + 1, 0);
        size_t nc = c_base, ntask = t_base;
        // Not every header has to start a segment: some segments per wave slot fill the device evenly, and every segment costs
        // k_gather a step of its stream's sequence (and the pool a first region).  Every `stride`-th header of a stream is kept, so
        // that the read has ~16 segments per slot (4: long segments, an uneven last round) - never longer than eight blocks.
        size_t n_headers = 0;
        for (size_t k = k0; k < k1; ++k) n_headers += std::min(h_out[k * kSpanWords], kSpanSlots);
        const size_t want_segments = size_t(ctx->n_cu) * 22 * 16;  // (measured 4 / 8 / 16 / all: profiles/r06_ingest.txt)
        const size_t stride = std::max<size_t>(1, std::min<size_t>(8, n_headers / std::max<size_t>(1, want_segments)));
        stg.n_headers = n_headers;
        for (size_t i = i0; i < i1; ++i) {
            stg.cand0[i - i0] = nc;
            stg.task0[i - i0] = ntask;
            size_t seen = 0;
            for (size_t k = span0[i]; k < span0[i + 1]; ++k) {
                const uint32_t cnt = std::min(h_out[k * kSpanWords], kSpanSlots);
                uint32_t *c = h_out + k * kSpanWords + 1;
                std::sort(c, c + cnt);
                for (uint32_t j = 0; j < cnt; ++j)
                    if (++seen % stride == 0) h_cand[nc++] = c[j];
            }
            const uint32_t ncs = uint32_t(nc - stg.cand0[i - i0]), c0 = uint32_t(stg.cand0[i - i0]);
            h_task[ntask] = SegTask{uint32_t(i), 16u, 0u, c0, ncs, uint32_t(ntask), 0u};
            ++ntask;
            for (uint32_t j = 0; j < ncs; ++j) {
                h_task[ntask] = SegTask{uint32_t(i), h_cand[c0 + j], dinf::kSegSlack, c0, ncs, uint32_t(ntask), 0u};
                ++ntask;
            }
        }
        stg.cand0[ni] = nc;
        stg.task0[ni] = ntask;
        ATL_REQUIRE(ntask <= t_max && nc <= t_max, "atl_nc_read_slabs: segment lists");
        const size_t n_count = ntask - t_base;
        stg.n_task = n_count;
        SegTask *h_sorted = reinterpret_cast<SegTask *>(sl->h + s_res) + t_base;  // (the results' page-locked place: they land behind the kernel)
        {  // longest first: the compressed bits up to the next split point stand for the work
            std::vector<std::pair<uint64_t, uint32_t>> order(n_count);
            for (size_t t = 0; t < n_count; ++t) {
                const SegTask &tk = h_task[t_base + t];
                const bool last_of_stream = t + 1 == n_count || h_task[t_base + t + 1].stream != tk.stream;
                const uint64_t end = last_of_stream ? uint64_t(jn[tk.stream].src_n) * 8 : uint64_t(h_task[t_base + t + 1].start_bit);
                order[t] = {end - tk.start_bit, uint32_t(t)};
            }
            std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (size_t t = 0; t < n_count; ++t) h_sorted[t] = h_task[t_base + order[t].second];
        }
        hipStream_t q2 = sl->st2;
        const SegTask *d_task = reinterpret_cast<const SegTask *>(d_meta + s_task);
        SegRes *d_sres = reinterpret_cast<SegRes *>(d_meta + s_res);
        if (nc > c_base)
            ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_cand + c_base * sizeof(uint32_t), h_cand + c_base, (nc - c_base) * sizeof(uint32_t), hipMemcpyHostToDevice, q2));
        ATL_HIP_TRY(hipMemcpyAsync(d_meta + s_task + t_base * sizeof(SegTask), h_sorted, n_count * sizeof(SegTask), hipMemcpyHostToDevice, q2));
        hipLaunchKernelGGL(k_segments_pool, dim3(unsigned(n_count)), dim3(64), 0, q2, sl->d, d_inf, d_task + t_base,
                           reinterpret_cast<const uint32_t *>(d_meta + s_cand), op_pr, d_sres);
        ATL_HIP_TRY(hipGetLastError());
        // (its results are fetched in op_finish: a copy queued behind the kernel would sit at the head of a copy engine's queue
        //  and hold up the batches' DMAs that are issued after it)
        so_cand = nc;
        so_task = ntask;
        op_stages.push_back(std::move(stg));
        return ATL_OK;
    };
    auto op_finish = [&]() -> int {
        hipStream_t q2 = sl->st2;
        for (const OpStage &stg : op_stages)
            ATL_HIP_TRY(hipMemcpyAsync(reinterpret_cast<SegRes *>(sl->h + s_res) + stg.t_base, reinterpret_cast<SegRes *>(d_meta + s_res) + stg.t_base,
                                       stg.n_task * sizeof(SegRes), hipMemcpyDeviceToHost, q2));
        ATL_HIP_TRY(hipStreamSynchronize(q2));  // every stage's results
        const std::vector<InfDesc> &jn = job.inf;
        const SegRes *h_res = reinterpret_cast<const SegRes *>(sl->h + s_res);
        const uint32_t *h_cand = reinterpret_cast<const uint32_t *>(sl->h + s_cand);
        std::vector<GatherSeg> gs;
        std::vector<ResDesc> rds;
        size_t n_seg_total = 0, n_tasks = 0, n_headers = 0;
        for (const OpStage &stg : op_stages) {
            n_tasks += stg.n_task;
            n_headers += stg.n_headers;
            for (size_t i = stg.i0; i < stg.i1; ++i) {
                const size_t li = i - stg.i0, g0 = gs.size();
                uint64_t at = 0;
                bool ok = true;
                uint32_t want = 0;
                for (size_t t = stg.task0[li];;) {
                    const SegRes &r = h_res[t];
                    if (r.status != dinf::kOk || r.o.out_end == 0 || at + r.o.out_end > uint64_t(jn[i].dst_n) ||
                        gs.size() - g0 > size_t(stg.task0[li + 1] - stg.task0[li])) {
                        ok = false;
                        break;
                    }
                    gs.push_back(GatherSeg{uint32_t(t), uint32_t(at), uint32_t(i), 0u});
                    at += r.o.out_end;
                    if (r.o.is_final) {
                        want = r.o.adler;
                        break;
                    }
                    const uint32_t *c = h_cand + stg.cand0[li], *ce = h_cand + stg.cand0[li + 1];
                    const uint32_t *it = std::lower_bound(c, ce, uint32_t(r.o.end_bit));
                    if (it == ce || *it != uint32_t(r.o.end_bit)) {
                        ok = false;
                        break;
                    }
                    t = stg.task0[li] + 1 + size_t(it - c);
                }
                if (!ok || at != uint64_t(jn[i].dst_n)) {  // the host decoders take the stream
                    gs.resize(g0);
                    continue;
                }
                rds.push_back(ResDesc{uint32_t(i), uint32_t(g0), uint32_t(gs.size() - g0), want});
                n_seg_total += gs.size() - g0;
                gs.push_back(GatherSeg{kNoTask, uint32_t(at), uint32_t(i), 0u});
            }
        }
        state->ms[3] += double(n_seg_total);  // (a count: atl_nc_ingest_times)
        if (dbg) {
            uint64_t longest = 0, n_long = 0;  // a launch lasts at least as long as its longest segment (a lone wave: ~10 MB/s)
            for (const OpStage &stg : op_stages)
                for (size_t t = stg.t_base; t < stg.t_base + stg.n_task; ++t)
                    if (h_res[t].status == dinf::kOk) {
                        longest = std::max<uint64_t>(longest, h_res[t].o.out_end);
                        n_long += h_res[t].o.out_end > (256u << 10);
                    }
            fprintf(stderr, "[atlite-hip ingest] split: the longest segment makes %llu bytes, %llu segments make more than 256 KiB\n", (unsigned long long)longest,
                    (unsigned long long)n_long);
            uint32_t used = 0;
            ATL_HIP_TRY(hipMemcpy(&used, op_pr.next, sizeof used, hipMemcpyDeviceToHost));
            fprintf(stderr, "[atlite-hip ingest] split, one pass in %zu stage(s): %zu streams, %zu spans, %zu block headers, %zu tasks, %zu streams chained into "
                    "%zu segments; the pool: %u of %u regions of 16 KiB used\n", op_stages.size(), n, n_spans, n_headers, n_tasks, rds.size(), n_seg_total, used,
                    op_pr.cap);
        }
        if (!rds.empty()) {
            ATL_REQUIRE(gs.size() <= t_max + n + 1 && rds.size() <= n, "atl_nc_read_slabs: segment lists");
            // long streams (chunks beyond 4 MiB): the sequence places the segments' last 32 KiB only, the rest follows at once, in pieces
            uint64_t longest_chunk = 0;
            for (const ResDesc &rd : rds) longest_chunk = std::max<uint64_t>(longest_chunk, uint64_t(jn[rd.stream].dst_n));
            const bool tails = longest_chunk > (uint64_t(4) << 20);
            std::vector<GatherPiece> pcs;
            if (tails)
                for (size_t g = 0; g + 1 < gs.size(); ++g) {
                    if (gs[g].task == kNoTask) continue;
                    const uint32_t len = gs[g + 1].start - gs[g].start;
                    if (len <= kWindow) continue;
                    for (uint32_t lo = 0; lo < len - kWindow; lo += kRestPiece) pcs.push_back(GatherPiece{uint32_t(g), lo, std::min(lo + kRestPiece, len - kWindow), 0u});
                }
            ATL_REQUIRE(pcs.size() <= op_piece_cap, "atl_nc_read_slabs: piece list");
            ATL_HIP_TRY(hipMemcpyAsync(sl->d_pool + op_gseg, gs.data(), gs.size() * sizeof(GatherSeg), hipMemcpyHostToDevice, q2));
            ATL_HIP_TRY(hipMemcpyAsync(sl->d_pool + op_rd, rds.data(), rds.size() * sizeof(ResDesc), hipMemcpyHostToDevice, q2));
            if (!pcs.empty()) ATL_HIP_TRY(hipMemcpyAsync(sl->d_pool + op_piece, pcs.data(), pcs.size() * sizeof(GatherPiece), hipMemcpyHostToDevice, q2));
            ATL_HIP_TRY(hipStreamSynchronize(q2));  // (the lists are on the stack)
            const ResDesc *d_rds = reinterpret_cast<const ResDesc *>(sl->d_pool + op_rd);
            const GatherSeg *d_gs = reinterpret_cast<const GatherSeg *>(sl->d_pool + op_gseg);
            const bool adler_here = !tails;  // (long chunks: the Adler-32 by many workgroups behind the gather)
            hipLaunchKernelGGL(k_gather, dim3(unsigned(rds.size())), dim3(1024), 0, q2, d_inf, d_rds, d_gs, op_pr, sl->d_raw, d_res, adler_here ? 1 : 0, tails ? 1 : 0);
            ATL_HIP_TRY(hipGetLastError());
            if (tails && !pcs.empty()) {
                hipLaunchKernelGGL(k_gather_rest, dim3(unsigned(pcs.size())), dim3(256), 0, q2, d_inf, d_gs,
                                   reinterpret_cast<const GatherPiece *>(sl->d_pool + op_piece), op_pr, sl->d_raw, d_res);
                ATL_HIP_TRY(hipGetLastError());
            }
            if (!adler_here) {
                unsigned long long *d_acc = reinterpret_cast<unsigned long long *>(sl->d_pool + op_acc);
                ATL_HIP_TRY(hipMemsetAsync(d_acc, 0, 2 * rds.size() * sizeof(unsigned long long), q2));
                hipLaunchKernelGGL(k_adler_parts, dim3(unsigned((longest_chunk + kAdlerPiece - 1) / kAdlerPiece), unsigned(rds.size())), dim3(256), 0, q2, d_inf,
                                   d_rds, sl->d_raw, d_acc);
                hipLaunchKernelGGL(k_adler_final, dim3(unsigned((rds.size() + 255) / 256)), dim3(256), 0, q2, d_inf, d_rds, uint32_t(rds.size()), d_acc, d_res);
                ATL_HIP_TRY(hipGetLastError());
            }
        }
        ATL_HIP_TRY(hipEventRecord(sl->ev_st2, q2));
        ATL_HIP_TRY(hipStreamWaitEvent(sl->st, sl->ev_st2, 0));  // the slot's stream goes on behind the gather: unpack, verdicts
        return split_finish();
    };
    if (fed) {
        rc = launch();
        if (rc) return rc;
    }
    if (split) {
        rc = op_prepare();
        if (rc) return rc;
    }
    // ---- the batches ----------------------------------------------------------------------------------------------------------
    const int fd = f->file.fd();
    const uint8_t *base = f->file.base();
    const size_t nt_max = size_t(pick_threads(n_threads, size_t(1) << 20));
    double gather_ms = 0;
    ATL_HIP_TRY(hipEventRecord(sl->ev_c[0], sl->st_c));
    size_t fed_batches = 0, flagged = 0;
    // arrival flags of the batches whose DMA has completed (wait: of every batch enqueued so far)
    auto set_flags = [&](bool wait) -> int {
        while (fed && flagged < fed_batches) {
            const hipError_t q = wait ? hipEventSynchronize(sl->ev_batch[flagged]) : hipEventQuery(sl->ev_batch[flagged]);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                break;
            }
            if (q != hipSuccess) {
                set_error("atl_nc_read_slabs: waiting for the DMA of a batch of chunk streams failed: %s", hipGetErrorString(q));
                return ATL_E_HIP;
            }
            __atomic_store_n(h_flag + flagged * kFlagPitch, kFlagReady, __ATOMIC_RELEASE);
            ++flagged;
        }
        return ATL_OK;
    };
    for (size_t b = 0; b < nb && !rc; ++b) {
        const Batch &bt = batches[b];
        Ring *ring = nullptr;
        rc = ring_acquire(ctx, ring_bytes, &ring);
        if (!rc) rc = set_flags(false);
        if (rc) break;
        const auto t_gather = std::chrono::steady_clock::now();
        // pieces of at most 8 MiB, so that a batch of one or two long streams still keeps every thread reading
        struct Piece {
            size_t stream;
            uint64_t off, len;
        };
        std::vector<Piece> pieces;
        for (size_t k = 0; k < bt.count; ++k) {
            const uint64_t want = uint64_t(job.inf[bt.first + k].src_n);
            for (uint64_t o = 0; o < want || o == 0; o += uint64_t(8) << 20) {
                pieces.push_back(Piece{bt.first + k, o, std::min<uint64_t>(uint64_t(8) << 20, want - o)});
                if (want == 0) break;
            }
        }
        rc = parallel_for(pieces.size(), int(std::max<size_t>(1, std::min(nt_max, pieces.size()))), [&](size_t k) -> int {
            const Piece &pc = pieces[k];
            const size_t i = pc.stream;
            const h5::Chunk &c = job.parts[job.part_of[i]].ds->chunks[job.lin[i]];
            uint8_t *dst = ring->h + (size_t(job.inf[i].src_off) - bt.off);
            const uint64_t want = uint64_t(job.inf[i].src_n), end = pc.off + pc.len;
            uint64_t got = pc.off;
            while (fd >= 0 && got < end) {  // straight into the pinned ring (a mapping's page faults contend across threads)
                const ssize_t r = pread(fd, dst + got, size_t(end - got), off_t(c.addr + got));
                if (r <= 0) break;
                got += uint64_t(r);
            }
            if (got < end) memcpy(dst + got, base + c.addr + got, size_t(end - got));
            if (end == want) memset(dst + want, 0, size_t(align_up(size_t(want) + 8, 128) - want));  // the bit reader's look-ahead words
            return ATL_OK;
        });
        gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gather).count();
        if (rc) break;
        hipError_t e = hipMemcpyAsync(sl->d + bt.off, ring->h, bt.bytes, hipMemcpyHostToDevice, sl->st_c);
        if (e == hipSuccess) e = hipEventRecord(sl->ev_batch[b], sl->st_c);
        if (e == hipSuccess) e = hipEventRecord(ring->ev, sl->st_c);
        if (e != hipSuccess) {
            set_error("atl_nc_read_slabs: DMA of a batch of chunk streams failed: %s", hipGetErrorString(e));
            rc = ATL_E_HIP;
            break;
        }
        ring->busy = true;
        fed_batches = b + 1;
        rc = set_flags(false);
        if (!rc && split) rc = find_batch(b);
    }
    if (!rc) rc = set_flags(true);  // the call returns when its last DMA has landed; the kernel goes on by itself
    state->ms[0] += gather_ms;
    if (rc && fed) {  // waves that wait for the batches that will not come: let them go (status kNotRun), then report
        for (size_t b = flagged; b < nb; ++b) __atomic_store_n(h_flag + b * kFlagPitch, kFlagAbort, __ATOMIC_RELEASE);
        job.aborted = true;
    }
    (void)hipEventRecord(sl->ev_c[1], sl->st_c);
    if (!rc && !fed) {  // the unfed order (A/B, $ATLITE_HIP_INGEST_FED=0): every byte first, then the launch
        ATL_HIP_TRY(hipStreamWaitEvent(sl->st, sl->ev_c[1], 0));
        if (split && op_ready) {
            rc = op_launch(op_stages.empty() ? 0 : op_stages.back().i1, n);
            if (!rc) rc = op_finish();
            if (rc) (void)hipStreamSynchronize(sl->st2);
        } else if (split) {
            rc = split_stage(0, n);
            if (!rc) rc = split_finish();
        } else {
            rc = launch();
        }
    }
    if (fed || !rc) {
        // the slot's event comes behind the kernel AND the last DMA (an aborted job's kernel may end before its flags' DMAs).  The
        // verdicts are fetched when the slot is settled (finish_slot), not queued here: a copy that waits for a kernel sits at the
        // head of a copy engine's queue and holds up every copy issued after it - the next job's or the next read's DMAs
        // (measured on the segment scheme's stages: profiles/r06_ingest.txt)
        (void)hipStreamWaitEvent(sl->st, sl->ev_c[1], 0);
    }
    (void)hipEventRecord(sl->ev, sl->st);
    sl->pending = true;
    sl->joined = false;
    job.active = true;
    if (rc) {
        if (split && sl->st2) (void)hipStreamSynchronize(sl->st2);  // (an early stage's kernel may still be reading the slot)
        job.aborted = true;
        return rc;
    }
    *done = true;
    // (no hipStreamWaitEvent(cs, sl->ev) here: the NEXT read forks from the copy stream, so that join would put the slots'
    //  streams one behind the other.  Whoever observes the copy stream goes through ingest_finish, which waits for every
    //  pending slot on the host first; the library's other copy-stream calls join lazily: ingest_join.)
    return ATL_OK;
}

// cells between the rows of variable `name`'s output block (what part_setup computes)
int64_t row_stride_of(atl_ctx *ctx, atl_nc *f, const char *name) {
    const Dataset *d = f->file.find(name);
    if (!d) return 0;
    int64_t ld = 1;
    for (size_t k = 1; k < d->shape.size(); ++k) ld *= int64_t(d->shape[k]);
    if (ctx->slot_stride > 0 && d->shape.size() == 3) ld = std::max<int64_t>(ld, ctx->slot_stride);
    return ld;
}

// The device-inflate path of a read: normally ONE job (read_group_device); rows whose chunks inflate to more than
// $ATLITE_HIP_INGEST_JOB_GB (default 12 GiB: the staging slots keep their buffers) are cut into jobs of whole chunk rows, one
// after the other on alternating slots.  *done = false (nothing enqueued): the device path does not apply - the caller reads
// variable by variable.
int read_rows_device(atl_ctx *ctx, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0, double *const *d_outs,
                     int n_threads, bool *done) {
    *done = false;
    if (count0 <= 0 || n_vars <= 0) return ATL_OK;
    // streams and inflated bytes of the whole read, and the coarsest time chunking among the variables (jobs are cut on its
    // boundaries: no chunk of that variable is inflated twice)
    int64_t streams = 0, tchunk = 1;
    double raw_bytes = 0;
    for (int v = 0; v < n_vars; ++v) {
        const Dataset *d = f->file.find(names[v]);
        if (!d) return ATL_OK;  // (the per-variable path words the error)
        if (d->layout != 2 || d->chunk.empty() || d->chunk[0] == 0) return ATL_OK;
        const int64_t c0 = int64_t(d->chunk[0]);
        int64_t per_row = 1;
        double chunk_bytes = double(d->type.size);
        for (size_t k = 1; k < d->grid.size(); ++k) per_row *= int64_t(d->grid[k]);
        for (size_t k = 0; k < d->chunk.size(); ++k) chunk_bytes *= double(d->chunk[k]);
        const int64_t rows = (start0 + count0 - 1) / c0 - start0 / c0 + 1;
        streams += rows * per_row;
        raw_bytes += double(rows * per_row) * chunk_bytes;
        tchunk = std::max(tchunk, c0);
    }
    if (!device_inflate_wanted(size_t(streams), raw_bytes)) return ATL_OK;
    double job_bytes = 12.0 * double(size_t(1) << 30);
    if (const char *e = getenv("ATLITE_HIP_INGEST_JOB_GB")) job_bytes = std::max(0.001, atof(e)) * double(size_t(1) << 30);
    const int64_t first_row = start0 / tchunk, last_row = (start0 + count0 - 1) / tchunk, chunk_rows = last_row - first_row + 1;
    const int64_t n_jobs = std::max<int64_t>(1, std::min<int64_t>(chunk_rows, int64_t(raw_bytes / job_bytes) + 1));
    const int64_t rows_per_job = (chunk_rows + n_jobs - 1) / n_jobs;
    for (int64_t j = 0; j < n_jobs; ++j) {
        const int64_t a = std::max(start0, (first_row + j * rows_per_job) * tchunk);
        const int64_t b = std::min(start0 + count0, (first_row + (j + 1) * rows_per_job) * tchunk);
        if (b <= a) continue;
        bool job_done = false;
        int rc = read_group_device(ctx, f, n_vars, names, a, b - a, start0, d_outs, n_threads, &job_done);
        if (rc) return rc;
        if (!job_done) {
            if (!*done) return ATL_OK;  // nothing enqueued yet: the whole read takes the other path
            // these rows hold a chunk that is not a plain zlib stream: variable by variable on the host threads
            for (int v = 0; v < n_vars; ++v) {
                rc = read_slab_one(ctx, f, names[v], a, b - a, d_outs[v] + (a - start0) * row_stride_of(ctx, f, names[v]), n_threads, false);
                if (rc) return rc;
            }
        }
        *done = true;
    }
    return ATL_OK;
}

}  // namespace

namespace atl {
// settle every read of this context whose chunks were inflated on the device (see submit_device): called before the copy
// stream is observed, so that whoever waits on it sees chunks the device decoder declined decoded by the host decoders
int ingest_finish(atl_ctx *ctx) {
    if (!ctx || !ctx->ingest) return ATL_OK;
    IngestState *st = static_cast<IngestState *>(ctx->ingest);
    for (Slot &sl : st->slot)
        if (sl.job.active) (void)finish_slot(ctx, st, sl);  // (every slot is settled, whatever the others' verdicts)
    return take_verdict(st);
}

// Make the copy stream wait for every device-inflate read still in flight (their kernels run on the slots' own streams, forked
// from the copy stream): whatever is enqueued on the copy stream next is ordered behind them, as include/atlite_hip.h promises
// for atl_nc_read_slab.  No host wait, no verdicts looked at.
int ingest_join(atl_ctx *ctx, hipStream_t cs) {
    if (!ctx || !ctx->ingest) return ATL_OK;
    IngestState *st = static_cast<IngestState *>(ctx->ingest);
    for (Slot &sl : st->slot)
        if (sl.job.active && sl.pending && !sl.joined) {
            ATL_HIP_TRY(hipStreamWaitEvent(cs, sl.ev, 0));
            sl.joined = true;
        }
    return ATL_OK;
}
}  // namespace atl

extern "C" {

int atl_nc_ingest_stats(atl_ctx *ctx, int64_t *device_chunks, int64_t *host_chunks, int64_t *redone) {
    ATL_REQUIRE(ctx, "atl_nc_ingest_stats: ctx is NULL");
    int rc = ingest_finish(ctx);
    if (rc) return rc;
    const IngestState *st = static_cast<const IngestState *>(ctx->ingest);
    if (device_chunks) *device_chunks = st ? st->n_device_chunks : 0;
    if (host_chunks) *host_chunks = st ? st->n_host_chunks : 0;
    if (redone) *redone = st ? st->n_redone : 0;
    return ATL_OK;
}

int atl_nc_ingest_times(atl_ctx *ctx, double *ms5, int64_t *compressed_bytes, int64_t *inflated_bytes) {
    ATL_REQUIRE(ctx && ms5, "atl_nc_ingest_times: bad argument");
    int rc = ingest_finish(ctx);
    if (rc) return rc;
    const IngestState *st = static_cast<const IngestState *>(ctx->ingest);
    for (int k = 0; k < 5; ++k) ms5[k] = st ? st->ms[k] : 0.0;
    if (compressed_bytes) *compressed_bytes = st ? st->comp_bytes : 0;
    if (inflated_bytes) *inflated_bytes = st ? st->raw_bytes : 0;
    return ATL_OK;
}

int atl_nc_open(const char *path, atl_nc **out) {
    ATL_REQUIRE(path && out, "atl_nc_open: bad argument");
    *out = nullptr;
    atl_nc *f = new atl_nc();
    const int rc = f->file.open(path);
    if (rc) {
        delete f;
        return rc;
    }
    *out = f;
    return ATL_OK;
}

int atl_nc_close(atl_nc *f) {
    if (f) {  // reads of this file whose verdicts are still out: settle them while the file is there
        std::lock_guard<std::mutex> lk(g_states_m);
        for (IngestState *st : g_states)
            for (Slot &sl : st->slot)
                if (sl.job.active && sl.job.nc == f) (void)finish_slot(st->ctx, st, sl);  // (a failure stays with the state: keep_verdict)
    }
    delete f;
    return ATL_OK;
}

int atl_nc_list(atl_nc *f, char *buf, int64_t buflen, int64_t *needed) {
    ATL_REQUIRE(f, "atl_nc_list: bad argument");
    std::string s;
    for (auto &d : f->file.datasets()) {
        if (!s.empty()) s += '\n';
        s += d.name;
    }
    return copy_text(s, buf, buflen, needed);
}

int atl_nc_inquire(atl_nc *f, const char *name, atl_nc_var *info) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_inquire");
    if (rc) return rc;
    ATL_REQUIRE(info, "atl_nc_inquire: info is NULL");
    memset(info, 0, sizeof *info);
    info->ndim = int32_t(std::min<size_t>(d->shape.size(), 4));
    const Decode dc = decode_of(f->file, *d);
    info->dtype = dc.dtype;
    info->elem_size = int32_t(d->type.size);
    info->big_endian = d->type.big_endian;
    for (int i = 0; i < info->ndim; ++i) {
        info->shape[i] = int64_t(d->shape[i]);
        info->chunk[i] = i < int(d->chunk.size()) ? int64_t(d->chunk[i]) : int64_t(d->shape[i]);
    }
    info->layout = d->layout;
    for (auto &fl : d->filters) {
        if (fl.id == 2) info->shuffle = 1;
        if (fl.id == 1) info->deflate = 1 + (fl.params.empty() ? 0 : int(fl.params[0]));
        if (fl.id == 3) info->fletcher32 = 1;
    }
    info->has_scale = dc.has_scale;
    info->has_fill = dc.has_fill;
    info->has_missing = dc.has_missing;
    info->scale_factor = dc.scale;
    info->add_offset = dc.offset;
    info->fill_value = dc.fill;
    info->missing_value = dc.missing;
    info->n_chunks = d->layout == 2 ? int64_t(d->chunks.size()) : 1;
    if (d->layout == 2) {
        for (auto &c : d->chunks) info->stored_bytes += int64_t(c.size);
    } else {
        info->stored_bytes = int64_t(d->contiguous_size);
    }
    return ATL_OK;
}

int atl_nc_dims(atl_nc *f, const char *name, char *buf, int64_t buflen, int64_t *needed) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_dims");
    if (rc) return rc;
    std::string s;
    for (size_t i = 0; i < d->dims.size(); ++i) {
        if (i) s += '\n';
        s += d->dims[i];
    }
    return copy_text(s, buf, buflen, needed);
}

static int find_attr(atl_nc *f, const char *var, const char *att, const Attribute **out, const char *who) {
    ATL_REQUIRE(f && att, "%s: bad argument", who);
    *out = nullptr;
    if (!var || !*var) {
        for (auto &a : f->file.global_attrs())
            if (a.name == att) *out = &a;
        return ATL_OK;
    }
    const Dataset *d;
    int rc = lookup(f, var, &d, who);
    if (rc) return rc;
    *out = d->attr(att);
    return ATL_OK;
}

int atl_nc_att_text(atl_nc *f, const char *var, const char *att, char *buf, int64_t buflen, int64_t *needed) {
    const Attribute *a;
    int rc = find_attr(f, var, att, &a, "atl_nc_att_text");
    if (rc) return rc;
    if (needed) *needed = 0;
    if (buf && buflen > 0) buf[0] = '\0';
    if (!a) return ATL_OK;
    std::string s;
    if (a->type.cls == TypeClass::String) {
        s.assign(reinterpret_cast<const char *>(a->data), size_t(a->nbytes));
        const size_t z = s.find('\0');
        if (z != std::string::npos) s.resize(z);
    } else if (a->type.cls == TypeClass::VlenStr && a->nbytes >= uint64_t(8 + f->file.off_size())) {
        const uint8_t *p = nullptr;
        uint64_t n = 0;
        uint32_t cnt = 0;
        if (f->file.vlen_payload(a->data, &p, &n, &cnt)) s.assign(reinterpret_cast<const char *>(p), size_t(std::min<uint64_t>(n, cnt)));
    } else {
        ATL_REQUIRE(false, "atl_nc_att_text: attribute '%s' is not a string", att);
    }
    return copy_text(s, buf, buflen, needed);
}

int atl_nc_att_double(atl_nc *f, const char *var, const char *att, double *out, int64_t max_n, int64_t *n) {
    const Attribute *a;
    int rc = find_attr(f, var, att, &a, "atl_nc_att_double");
    if (rc) return rc;
    ATL_REQUIRE(n && (out || max_n == 0), "atl_nc_att_double: bad argument");
    *n = 0;
    if (!a) return ATL_OK;
    ATL_REQUIRE(attr_double(f->file, a, out, max_n, n), "atl_nc_att_double: attribute '%s' is not numeric", att);
    return ATL_OK;
}

int atl_nc_read_host(atl_nc *f, const char *name, int64_t start0, int64_t count0, double *out) {
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_read_host");
    if (rc) return rc;
    Geometry g;
    rc = geometry_of(*d, &g, "atl_nc_read_host");
    if (rc) return rc;
    ATL_REQUIRE(start0 >= 0 && count0 >= 0 && start0 + count0 <= g.shape[0],
                "atl_nc_read_host: rows [%lld, %lld) outside '%s' (%lld rows)", (long long)start0,
                (long long)(start0 + count0), name, (long long)g.shape[0]);
    if (count0 == 0 || g.row_elems == 0) return ATL_OK;
    ATL_REQUIRE(out, "atl_nc_read_host: out is NULL");
    const Decode dc = decode_of(f->file, *d);
    const int es = dc.esize;
    const int64_t r0 = start0, r1 = start0 + count0;
    auto element = [&](const uint8_t *src, int64_t n, int64_t e, bool shuffled) {
        uint64_t b = 0;
        for (int k = 0; k < es; ++k) {
            const int dstk = dc.big_endian ? es - 1 - k : k;
            const uint8_t byte = shuffled ? src[int64_t(k) * n + e] : src[e * es + k];
            b |= uint64_t(byte) << (8 * dstk);
        }
        return cf_decode(bits_to_double(b, dc.dtype), dc);
    };
    if (d->layout != 2) {
        const uint8_t *src = d->layout == 0 ? d->compact : f->file.base() + d->contiguous_addr;
        const uint64_t need = uint64_t(g.shape[0]) * g.row_elems * es;
        if (d->layout == 1 && (d->contiguous_addr == 0 || d->contiguous_addr > f->file.size() ||
                               need > f->file.size() - d->contiguous_addr || d->contiguous_size < need)) {
            // never written (address undefined) -> fill value
            const double v = dc.has_fill ? cf_decode(dc.fill, dc) : __builtin_nan("");
            std::fill(out, out + count0 * g.row_elems, v);
            return ATL_OK;
        }
        ATL_REQUIRE(d->layout == 1 || d->contiguous_size >= need, "atl_nc_read_host: compact data of '%s' is short", name);
        const int64_t n = g.shape[0] * g.row_elems;
        for (int64_t i = 0; i < count0 * g.row_elems; ++i) out[i] = element(src, n, r0 * g.row_elems + i, false);
        return ATL_OK;
    }
    Selection sel;
    const int64_t chunk_bytes = g.chunk_elems * es;
    select_chunks(g, r0, r1, chunk_bytes, &sel);
    const int nt = pick_threads(0, sel.lin.size());
    return parallel_for(sel.lin.size(), nt, [&](size_t i) -> int {
        const h5::Chunk &c = d->chunks[sel.lin[i]];
        const UnpackDesc &ds = sel.desc[i];
        std::vector<uint8_t> tmp;
        bool shuffled = false;
        const bool missing = c.size == 0;
        if (!missing) {
            tmp.resize(size_t(chunk_bytes));
            const int e = h5::chunk_inflate(*d, c, f->file.base(), -1, tmp.data(), uint64_t(chunk_bytes), &shuffled);
            if (e) return e;
        }
        const double fillv = dc.has_fill ? cf_decode(dc.fill, dc) : __builtin_nan("");
        for (int64_t ct = 0; ct < ds.dim[0]; ++ct) {
            const int64_t t = ds.org[0] + ct;
            if (t < r0 || t >= r1) continue;
            for (int64_t cy = 0; cy < ds.dim[1]; ++cy) {
                const int64_t y = ds.org[1] + cy;
                if (y >= g.shape[1]) break;
                for (int64_t cx = 0; cx < ds.dim[2]; ++cx) {
                    const int64_t x = ds.org[2] + cx;
                    if (x >= g.shape[2]) break;
                    const int64_t e = (ct * ds.dim[1] + cy) * ds.dim[2] + cx;
                    out[(t - r0) * g.row_elems + y * g.shape[2] + x] =
                        missing ? fillv : element(tmp.data(), g.chunk_elems, e, shuffled);
                }
            }
        }
        return ATL_OK;
    });
}

int atl_nc_read_slab(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out,
                     int n_threads) {
    return read_slab_one(ctx, f, name, start0, count0, d_out, n_threads, true);
}

}  // extern "C"

namespace {
int read_slab_one(atl_ctx *ctx, atl_nc *f, const char *name, int64_t start0, int64_t count0, double *d_out, int n_threads, bool try_device) {
    ATL_REQUIRE(ctx, "atl_nc_read_slab: ctx is NULL");
    const Dataset *d;
    int rc = lookup(f, name, &d, "atl_nc_read_slab");
    if (rc) return rc;
    Geometry g;
    rc = geometry_of(*d, &g, "atl_nc_read_slab");
    if (rc) return rc;
    ATL_REQUIRE(start0 >= 0 && count0 >= 0 && start0 + count0 <= g.shape[0],
                "atl_nc_read_slab: rows [%lld, %lld) outside '%s' (%lld rows)", (long long)start0,
                (long long)(start0 + count0), name, (long long)g.shape[0]);
    if (count0 == 0 || g.row_elems == 0) return ATL_OK;
    ATL_REQUIRE(d_out, "atl_nc_read_slab: d_out is NULL");
    const Decode dc = decode_of(f->file, *d);
    const int es = dc.esize;
    const int64_t r0 = start0, r1 = start0 + count0;
    UnpackParams p{};
    p.shape1 = g.shape[1];
    p.shape2 = g.shape[2];
    // rows of the output block: contiguous unless the call asks for padded slots (ld_cells) and the
    // variable has rows to pad (a (time, y, x) cube)
    p.ld = g.shape[1] * g.shape[2];
    if (ctx->slot_stride > 0 && d->shape.size() == 3) {
        ATL_REQUIRE(ctx->slot_stride >= p.ld, "atl_nc_read_slab: slot stride %lld is smaller than a row of %lld cells",
                    (long long)ctx->slot_stride, (long long)p.ld);
        p.ld = ctx->slot_stride;
    }
    p.r0 = r0;
    p.r1 = r1;
    p.dec = dc;

    Selection sel;
    int64_t max_elems;
    size_t payload;
    Slot *sl = nullptr;
    if (d->layout != 2) {
        // compact / contiguous: the wanted rows are one pseudo-chunk, copied in parallel slices
        const uint64_t row_bytes = uint64_t(g.row_elems) * es;
        const uint64_t need = uint64_t(g.shape[0]) * row_bytes;
        UnpackDesc ds{};
        ds.org[0] = r0;
        ds.dim[0] = count0;
        ds.dim[1] = g.shape[1];
        ds.dim[2] = g.shape[2];
        const bool written = d->layout == 0 ? d->contiguous_size >= need
                                            : (d->contiguous_addr != 0 && d->contiguous_addr <= f->file.size() &&
                                               need <= f->file.size() - d->contiguous_addr && d->contiguous_size >= need);
        ds.missing = !written;
        sel.desc.push_back(ds);
        payload = written ? size_t(count0 * row_bytes) : 16;
        max_elems = count0 * g.row_elems;
        {
            const size_t need = align_up(payload, 256) + sizeof(UnpackDesc);
            rc = slot_acquire(ctx, need, need, 0, false, &sl);
        }
        if (rc) return rc;
        if (written) {
            const uint8_t *src = (d->layout == 0 ? d->compact : f->file.base() + d->contiguous_addr) + r0 * row_bytes;
            const size_t slice = size_t(4) << 20;
            const size_t ns = (payload + slice - 1) / slice;
            rc = parallel_for(ns, pick_threads(n_threads, ns), [&](size_t i) -> int {
                const size_t a = i * slice, b = std::min(payload, a + slice);
                memcpy(sl->h + a, src + a, b - a);
                return ATL_OK;
            });
            if (rc) return rc;
        }
    } else {
        const int64_t chunk_bytes = g.chunk_elems * es;
        select_chunks(g, r0, r1, chunk_bytes, &sel);
        payload = sel.desc.size() * align_up(size_t(chunk_bytes), 16);
        max_elems = g.chunk_elems;
        // ---- chunks inflated on the device: every stored chunk of the selection is a zlib stream (read_rows_device) ------------
        if (try_device) {
            const char *one[1] = {name};
            double *outs[1] = {d_out};
            bool done = false;
            rc = read_rows_device(ctx, f, 1, one, start0, count0, outs, n_threads, &done);
            if (rc || done) return rc;
        }
        for (UnpackDesc &ds : sel.desc) ds.shuffled = 0;
        {
            const size_t need = align_up(payload, 256) + sel.desc.size() * sizeof(UnpackDesc);
            rc = slot_acquire(ctx, need, need, 0, false, &sl);
        }
        if (rc) return rc;
        state_of(ctx)->n_host_chunks += int64_t(sel.lin.size());
        rc = parallel_for(sel.lin.size(), pick_threads(n_threads, sel.lin.size()), [&](size_t i) -> int {
            const h5::Chunk &c = d->chunks[sel.lin[i]];
            UnpackDesc &ds = sel.desc[i];
            if (c.size == 0) {
                ds.missing = 1;
                return ATL_OK;
            }
            bool shuffled = false;
            const int e = h5::chunk_inflate(*d, c, f->file.base(), f->file.fd(), sl->h + ds.src_off, uint64_t(chunk_bytes), &shuffled);
            ds.shuffled = shuffled;
            return e;
        });
        if (rc) return rc;
    }
    return submit(ctx, sl, payload, sel.desc, p, max_elems, d_out);
}
}  // namespace

extern "C" {

int atl_nc_read_slabs(atl_ctx *ctx, atl_nc *f, int n_vars, const char *const *names, int64_t start0, int64_t count0, double *const *d_outs,
                      int n_threads) {
    ATL_REQUIRE(ctx && f && n_vars >= 0 && (n_vars == 0 || (names && d_outs)), "atl_nc_read_slabs: bad argument");
    if (n_vars == 0) return ATL_OK;
    bool done = false;
    int rc = read_rows_device(ctx, f, n_vars, names, start0, count0, d_outs, n_threads, &done);  // all variables, pipelined jobs
    if (rc || done) return rc;
    for (int v = 0; v < n_vars; ++v) {  // variable by variable (each decides for itself: device or host threads)
        rc = atl_nc_read_slab(ctx, f, names[v], start0, count0, d_outs[v], n_threads);
        if (rc) return rc;
    }
    return ATL_OK;
}

int atl_inflate_probe(const void *h_src, size_t src_n, void *h_dst, size_t dst_n, int which, int64_t *ns) {
    ATL_REQUIRE(h_src && (h_dst || dst_n == 0) && which >= 0 && which <= 4, "atl_inflate_probe: bad argument");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ATL_OK;
    const uint8_t *src = static_cast<const uint8_t *>(h_src);
    uint8_t *dst = static_cast<uint8_t *>(h_dst);
    bool done = false;
    if (which == 3) {  // the device decoder's serial half on the host
        const int st = h5::device_inflate_emulated(src, src_n, dst, dst_n);
        if (st) {
            set_error("atl_inflate_probe: the device decoder's host emulation returned status %d", st);
            rc = ATL_E_INVALID;
        }
        done = true;
    } else if (which == 4) {  // ... and its segment scheme (*ns = the number of segments of the stream's chain instead of a time)
        int n_seg = 0;
        const int st = h5::device_inflate_split_emulated(src, src_n, dst, dst_n, &n_seg);
        if (st) {
            set_error("atl_inflate_probe: the segment decoder's host emulation returned status %d", st);
            rc = ATL_E_INVALID;
        }
        if (ns) *ns = n_seg;
        return rc;
    } else if (which != 1) {
        done = h5::fast_inflate_zlib(src, src_n, dst, dst_n) == 0;
        if (!done && which == 0) {
            set_error("atl_inflate_probe: the fast decoder declined the stream");
            rc = ATL_E_UNSUPPORTED;
        }
    }
    if (!done && which != 0) {
        uLongf out_n = uLongf(dst_n);
        const int z = uncompress(dst, &out_n, src, uLong(src_n));
        if (z != Z_OK || out_n != dst_n) {
            set_error("atl_inflate_probe: zlib rc %d, %llu of %llu bytes", z, (unsigned long long)out_n,
                      (unsigned long long)dst_n);
            rc = ATL_E_INVALID;
        }
    }
    if (ns) *ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int atl_upload_convert_async(atl_ctx *ctx, double *d_dst, const void *h_src, int dtype, int64_t n) {
    return atl_upload_convert_2d_async(ctx, d_dst, 1, h_src, dtype, n, 1);
}

int atl_upload_convert_2d_async(atl_ctx *ctx, double *d_dst, int64_t ld_cells, const void *h_src, int dtype, int64_t rows,
                                int64_t cols) {
    ATL_REQUIRE(ctx && rows >= 0 && cols >= 0 && ld_cells >= cols, "atl_upload_convert_async: bad argument");
    const int64_t n = rows * cols;
    if (n == 0) return ATL_OK;
    ATL_REQUIRE(d_dst && h_src, "atl_upload_convert_async: NULL buffer");
    const int es = dtype_size(dtype);
    ATL_REQUIRE(es > 0, "atl_upload_convert_async: unknown dtype code %d", dtype);
    const size_t payload = size_t(n) * es;
    Slot *sl = nullptr;
    const size_t need_bytes = align_up(payload, 256) + sizeof(UnpackDesc);
    int rc = slot_acquire(ctx, need_bytes, need_bytes, 0, false, &sl);
    if (rc) return rc;
    // page-locked source (atl_host_register / Dataset.pin): DMA straight from it; pageable: gather
    // through the pinned staging on host threads (a pageable hipMemcpyAsync would serialise the stream)
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, h_src) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned) {
        const size_t slice = size_t(4) << 20;
        const size_t ns = (payload + slice - 1) / slice;
        const uint8_t *src = static_cast<const uint8_t *>(h_src);
        rc = parallel_for(ns, pick_threads(0, ns), [&](size_t i) -> int {
            const size_t a = i * slice, b = std::min(payload, a + slice);
            memcpy(sl->h + a, src + a, b - a);
            return ATL_OK;
        });
        if (rc) return rc;
    }
    UnpackParams p{};  // a (rows, 1, cols) variable in one "chunk": element (r, 0, c) goes to d_dst[r * ld + c]
    p.shape1 = 1;
    p.shape2 = cols;
    p.ld = ld_cells;
    p.r0 = 0;
    p.r1 = rows;
    p.dec.dtype = dtype;
    p.dec.esize = es;
    UnpackDesc ds{};
    ds.dim[0] = rows;
    ds.dim[1] = 1;
    ds.dim[2] = cols;
    std::vector<UnpackDesc> descs{ds};
    return submit(ctx, sl, payload, descs, p, n, d_dst, pinned ? h_src : nullptr);
}

}  // extern "C"
