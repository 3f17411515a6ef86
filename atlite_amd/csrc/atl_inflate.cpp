// A fast DEFLATE (RFC 1951) decoder for zlib-wrapped (RFC 1950) chunk payloads.
//
// File-backed runs are bound by inflating the chunks on the host (zlib 1.2.11 in this image:
// ~0.4 GB/s of output per core).  This decoder uses the usual modern recipe - 64-bit bit buffer with
// branch-free refill, wide (11-bit litlen / 8-bit offset) single-lookup tables whose entries carry
// codeword length, extra-bit count and base value, word-at-a-time match copies, a guarded fast loop
// with a bounds-checked tail loop - and is ~2-3x faster.  Safety net: every stream's Adler-32 is
// verified; on ANY failure (format error, checksum mismatch, size mismatch) the caller falls back to
// zlib's own inflate, so the fast path can only ever make things faster, never different.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

#include <cstdint>
#include <cstring>
#include <type_traits>

#include "atl_h5.h"

#include <vector>
#define ATL_HD
#include "atl_inflate_dev.h"

namespace atl { namespace h5 {

namespace {

constexpr int kLitlenBits = 11, kOffsetBits = 8, kPrecodeBits = 7;
constexpr int kLitlenCap = 4096, kOffsetCap = 1024, kPrecodeCap = 128;

// entry: len (6) | extra (4) << 6 | kind (2) << 10 | value << 16
enum : uint32_t { kBase = 0, kLiteral = 1, kEnd = 2, kSub = 3 };
inline uint32_t mk(uint32_t len, uint32_t extra, uint32_t kind, uint32_t value) {
    return len | (extra << 6) | (kind << 10) | (value << 16);
}
inline uint32_t e_len(uint32_t e) { return e & 0x3F; }
inline uint32_t e_extra(uint32_t e) { return (e >> 6) & 0xF; }
inline uint32_t e_kind(uint32_t e) { return (e >> 10) & 0x3; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                               31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kOffBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                               193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kOffExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bit_reverse(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) {
        r = (r << 1) | (code & 1);
        code >>= 1;
    }
    return r;
}

enum TableKind { kLitlenTable, kOffsetTable, kPrecodeTable };

// canonical Huffman decode table; false = invalid / unsupported code (caller falls back to zlib)
bool build_table(const uint8_t *lens, int n, int table_bits, int cap, TableKind what, uint32_t *table) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    if (count[0] == n) return false;  // no codes at all
    int left = 1, used = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;  // over-subscribed
        used += count[l];
    }
    if (left > 0 && !(used == 1 && count[1] == 1)) return false;  // incomplete (zlib allows a single 1-bit code)
    uint32_t next[16];
    next[0] = 0;
    uint32_t code = 0;
    count[0] = 0;  // unused symbols take no code space
    for (int l = 1; l <= 15; ++l) {
        code = (code + uint32_t(count[l - 1])) << 1;
        next[l] = code;
    }
    const uint32_t tsize = 1u << table_bits;
    for (uint32_t i = 0; i < tsize; ++i) table[i] = 0;  // len 0 = invalid
    auto entry_for = [&](int sym, uint32_t len) -> uint32_t {
        if (what == kPrecodeTable) return mk(len, 0, kLiteral, uint32_t(sym));
        if (what == kOffsetTable) return sym < 30 ? mk(len, kOffExtra[sym], kBase, kOffBase[sym]) : 0u;
        if (sym < 256) return mk(len, 0, kLiteral, uint32_t(sym));
        if (sym == 256) return mk(len, 0, kEnd, 0);
        return sym < 286 ? mk(len, kLenExtra[sym - 257], kBase, kLenBase[sym - 257]) : 0u;
    };
    // pass 1: short codes, and the widest long code per primary slot
    uint8_t sub_bits[1u << kLitlenBits];
    bool any_long = false;
    memset(sub_bits, 0, tsize);
    uint32_t codes[320];
    {
        uint32_t nx[16];
        memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = bit_reverse(nx[l]++, l);
            codes[s] = r;
            if (l <= table_bits) {
                const uint32_t e = entry_for(s, uint32_t(l));
                for (uint32_t i = r; i < tsize; i += 1u << l) table[i] = e;
            } else {
                const uint32_t p = r & (tsize - 1);
                if (l - table_bits > sub_bits[p]) sub_bits[p] = uint8_t(l - table_bits);
                any_long = true;
            }
        }
    }
    if (!any_long) return true;
    uint32_t pos = tsize;
    for (uint32_t p = 0; p < tsize; ++p) {
        if (!sub_bits[p]) continue;
        if (pos + (1u << sub_bits[p]) > uint32_t(cap)) return false;
        table[p] = mk(uint32_t(table_bits), sub_bits[p], kSub, pos);
        for (uint32_t i = 0; i < (1u << sub_bits[p]); ++i) table[pos + i] = 0;
        pos += 1u << sub_bits[p];
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= table_bits) continue;
        const uint32_t r = codes[s];
        const uint32_t p = r & (tsize - 1);
        const uint32_t start = e_value(table[p]), sb = sub_bits[p];
        const uint32_t e = entry_for(s, uint32_t(l - table_bits));
        for (uint32_t i = r >> table_bits; i < (1u << sb); i += 1u << (l - table_bits)) table[start + i] = e;
    }
    return true;
}

inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int cnt = 0;
    inline void refill_fast() {  // needs p + 8 <= end
        buf |= load64(p) << cnt;
        p += (63 - cnt) >> 3;
        cnt |= 56;
    }
    inline void refill_safe() {
        while (cnt <= 56 && p < end) {
            buf |= uint64_t(*p++) << cnt;
            cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return uint32_t(buf & ((uint64_t(1) << n) - 1)); }
    inline void drop(int n) {
        buf >>= n;
        cnt -= n;
    }
};

uint32_t adler32_scalar(const uint8_t *p, size_t n, uint32_t adler) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k >= 8) {
            a += p[0]; b += a;
            a += p[1]; b += a;
            a += p[2]; b += a;
            a += p[3]; b += a;
            a += p[4]; b += a;
            a += p[5]; b += a;
            a += p[6]; b += a;
            a += p[7]; b += a;
            p += 8;
            k -= 8;
        }
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= 65521;
        b %= 65521;
    }
    return (b << 16) | a;
}

// 32 bytes per step: s1 through psadbw, s2 through pmaddubsw with the weights 32..1 plus 32 x the
// running s1; sums are reduced modulo 65521 every 173 blocks (5536 bytes), before anything can overflow
__attribute__((target("avx2"))) uint32_t adler32_avx2(const uint8_t *p, size_t n, uint32_t adler) {
    uint64_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14,
                                             13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m256i ones = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
    while (n >= 32) {
        size_t k = n / 32;
        if (k > 173) k = 173;
        n -= k * 32;
        __m256i v_s1 = zero, v_s2 = zero, v_ps = zero;
        for (size_t i = 0; i < k; ++i) {
            const __m256i bytes = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
            p += 32;
            v_ps = _mm256_add_epi32(v_ps, v_s1);
            v_s1 = _mm256_add_epi32(v_s1, _mm256_sad_epu8(bytes, zero));
            v_s2 = _mm256_add_epi32(v_s2, _mm256_madd_epi16(_mm256_maddubs_epi16(bytes, weights), ones));
        }
        alignas(32) uint32_t t1[8], t2[8], tp[8];
        _mm256_store_si256(reinterpret_cast<__m256i *>(t1), v_s1);
        _mm256_store_si256(reinterpret_cast<__m256i *>(t2), v_s2);
        _mm256_store_si256(reinterpret_cast<__m256i *>(tp), v_ps);
        uint64_t h1 = 0, h2 = 0, hp = 0;
        for (int i = 0; i < 8; ++i) {
            h1 += t1[i];
            h2 += t2[i];
            hp += tp[i];
        }
        s2 = (s2 + s1 * 32 * k + 32 * hp + h2) % 65521;
        s1 = (s1 + h1) % 65521;
    }
    return adler32_scalar(p, n, uint32_t((s2 << 16) | s1));
}

uint32_t adler32_of(const uint8_t *p, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? adler32_avx2(p, n, 1) : adler32_scalar(p, n, 1);
}

struct FixedTables {
    uint32_t litlen[kLitlenCap], offset[kOffsetCap];
    bool ok;
    FixedTables() {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) d[i] = 5;
        ok = build_table(l, 288, kLitlenBits, kLitlenCap, kLitlenTable, litlen) &&
             build_table(d, 32, kOffsetBits, kOffsetCap, kOffsetTable, offset);
    }
};

}  // namespace

// 0 = exactly dst_n bytes produced and the Adler-32 matches; anything else = let zlib decide
int fast_inflate_zlib(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n) {
    if (src_n < 6) return 1;
    if ((src[0] & 0x0F) != 8 || (src[0] >> 4) > 7 || ((uint32_t(src[0]) << 8) | src[1]) % 31 != 0 || (src[1] & 0x20))
        return 1;
    static const FixedTables fixed;
    if (!fixed.ok) return 1;
    Bits b;
    b.p = src + 2;
    b.end = src + src_n;
    uint8_t *out = dst, *const out_end = dst + dst_n;
    static thread_local uint32_t dyn_litlen[kLitlenCap], dyn_offset[kOffsetCap];
    bool final_block = false;
    while (!final_block) {
        b.refill_safe();
        if (b.cnt < 3) return 2;
        final_block = b.peek(1);
        b.drop(1);
        const uint32_t type = b.peek(2);
        b.drop(2);
        const uint32_t *lt, *ot;
        if (type == 0) {
            // stored: give whole bytes back, skip to the byte boundary
            b.drop(b.cnt & 7);
            b.p -= b.cnt >> 3;
            b.buf = 0;
            b.cnt = 0;
            if (b.end - b.p < 4) return 2;
            const uint32_t len = uint32_t(b.p[0]) | (uint32_t(b.p[1]) << 8), nlen = uint32_t(b.p[2]) | (uint32_t(b.p[3]) << 8);
            b.p += 4;
            if ((len ^ nlen) != 0xFFFF || uint64_t(b.end - b.p) < len || uint64_t(out_end - out) < len) return 2;
            memcpy(out, b.p, len);
            out += len;
            b.p += len;
            continue;
        } else if (type == 1) {
            lt = fixed.litlen;
            ot = fixed.offset;
        } else if (type == 2) {
            b.refill_safe();
            if (b.cnt < 14) return 2;
            const int hlit = int(b.peek(5)) + 257;
            b.drop(5);
            const int hdist = int(b.peek(5)) + 1;
            b.drop(5);
            const int hclen = int(b.peek(4)) + 4;
            b.drop(4);
            if (hlit > 286 || hdist > 30) return 2;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t plens[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                b.refill_safe();
                if (b.cnt < 3) return 2;
                plens[order[i]] = uint8_t(b.peek(3));
                b.drop(3);
            }
            uint32_t pre[kPrecodeCap];
            if (!build_table(plens, 19, kPrecodeBits, kPrecodeCap, kPrecodeTable, pre)) return 2;
            uint8_t lens[286 + 30 + 138];
            int i = 0;
            const int total = hlit + hdist;
            while (i < total) {
                b.refill_safe();
                const uint32_t e = pre[b.peek(kPrecodeBits)];
                const int l = int(e_len(e));
                if (!l || b.cnt < l) return 2;
                b.drop(l);
                const uint32_t sym = e_value(e);
                if (sym < 16) {
                    lens[i++] = uint8_t(sym);
                } else if (sym == 16) {
                    if (i == 0 || b.cnt < 2) return 2;
                    const int rep = 3 + int(b.peek(2));
                    b.drop(2);
                    memset(lens + i, lens[i - 1], size_t(rep));
                    i += rep;
                } else if (sym == 17) {
                    if (b.cnt < 3) return 2;
                    const int rep = 3 + int(b.peek(3));
                    b.drop(3);
                    memset(lens + i, 0, size_t(rep));
                    i += rep;
                } else {
                    if (b.cnt < 7) return 2;
                    const int rep = 11 + int(b.peek(7));
                    b.drop(7);
                    memset(lens + i, 0, size_t(rep));
                    i += rep;
                }
            }
            if (i != total || lens[256] == 0) return 2;
            if (!build_table(lens, hlit, kLitlenBits, kLitlenCap, kLitlenTable, dyn_litlen)) return 2;
            // a block of literals only may carry an empty offset code
            bool any_off = false;
            for (int k = 0; k < hdist; ++k) any_off |= lens[hlit + k] != 0;
            if (any_off) {
                if (!build_table(lens + hlit, hdist, kOffsetBits, kOffsetCap, kOffsetTable, dyn_offset)) return 2;
            } else {
                for (int k = 0; k < (1 << kOffsetBits); ++k) dyn_offset[k] = 0;
            }
            lt = dyn_litlen;
            ot = dyn_offset;
        } else {
            return 2;
        }

        // ---- symbols ---------------------------------------------------------------------------
        // One symbol (or a run of up to three literals).  FAST: >= 16 input bytes and >= 320 output
        // bytes are left, so the refill is a single unaligned load, no bit-count or space check can
        // fail (a litlen + offset pair takes at most 48 of the >= 56 buffered bits) and match copies
        // may run past their end.  Returns 0 = go on, 1 = end of block, else an error code.
        auto step = [&](auto fast_tag) -> int {
            constexpr bool FAST = decltype(fast_tag)::value;
            if (FAST) {
                b.refill_fast();
            } else {
                b.refill_safe();
            }
            uint32_t e = lt[b.peek(kLitlenBits)];
            if (e_kind(e) == kLiteral) {  // never a sub-table link: those carry kSub
                if (!FAST && (b.cnt < int(e_len(e)) || out >= out_end)) return 3;
                b.drop(int(e_len(e)));
                *out++ = uint8_t(e_value(e));
                if (FAST) {  // up to two more literals from the >= 41 bits still buffered
                    e = lt[b.peek(kLitlenBits)];
                    if (e_kind(e) == kLiteral) {
                        b.drop(int(e_len(e)));
                        *out++ = uint8_t(e_value(e));
                        e = lt[b.peek(kLitlenBits)];
                        if (e_kind(e) == kLiteral) {
                            b.drop(int(e_len(e)));
                            *out++ = uint8_t(e_value(e));
                        }
                    }
                }
                return 0;
            }
            if (e_kind(e) == kSub) {
                b.drop(kLitlenBits);
                e = lt[e_value(e) + b.peek(int(e_extra(e)))];
            }
            int l = int(e_len(e));
            if (!l || (!FAST && b.cnt < l)) return 3;
            b.drop(l);
            if (e_kind(e) == kLiteral) {
                if (!FAST && out >= out_end) return 3;
                *out++ = uint8_t(e_value(e));
                return 0;
            }
            if (e_kind(e) == kEnd) return 1;
            const int lx = int(e_extra(e));
            if (!FAST && b.cnt < lx) return 3;
            const uint32_t length = e_value(e) + b.peek(lx);
            b.drop(lx);
            if (!FAST) b.refill_safe();
            uint32_t o = ot[b.peek(kOffsetBits)];
            if (e_kind(o) == kSub) {
                b.drop(kOffsetBits);
                o = ot[e_value(o) + b.peek(int(e_extra(o)))];
            }
            l = int(e_len(o));
            if (!l || (!FAST && b.cnt < l)) return 3;
            b.drop(l);
            const int ox = int(e_extra(o));
            if (!FAST) {
                b.refill_safe();
                if (b.cnt < ox) return 3;
            }
            const uint32_t dist = e_value(o) + b.peek(ox);
            b.drop(ox);
            if (dist > uint64_t(out - dst) || uint64_t(out_end - out) < length) return 3;
            const uint8_t *from = out - dist;
            if (FAST) {
                uint8_t *const stop = out + length;
                if (dist >= 8) {
                    uint8_t *o8 = out;
                    do {
                        store64(o8, load64(from));
                        o8 += 8;
                        from += 8;
                    } while (o8 < stop);
                } else if (dist == 1) {
                    const uint64_t v = 0x0101010101010101ull * from[0];
                    uint8_t *o8 = out;
                    do {
                        store64(o8, v);
                        o8 += 8;
                    } while (o8 < stop);
                } else {
                    // period 2..7: lay down m >= 8 bytes (a multiple of the period) one by one, then
                    // copy words from m bytes back
                    uint32_t m = dist;
                    while (m < 8) m += dist;
                    const uint32_t head = length < m ? length : m;
                    for (uint32_t k = 0; k < head; ++k) out[k] = from[k];
                    for (uint8_t *o8 = out + m; o8 < stop; o8 += 8) store64(o8, load64(o8 - m));
                }
            } else {
                for (uint32_t k = 0; k < length; ++k) out[k] = from[k];
            }
            out += length;
            return 0;
        };
        for (;;) {
            int r = 0;
            if ((b.end - b.p) >= 16 && (out_end - out) >= 320) {
                do {
                    r = step(std::true_type{});
                } while (r == 0 && (b.end - b.p) >= 16 && (out_end - out) >= 320);
            } else {
                r = step(std::false_type{});
            }
            if (r == 1) break;
            if (r) return r;
        }
    }
    if (out != out_end) return 4;
    // trailer: Adler-32 of the output, big-endian, after the last (partial) byte
    b.drop(b.cnt & 7);
    b.p -= b.cnt >> 3;
    if (b.end - b.p < 4) return 4;
    const uint32_t want = (uint32_t(b.p[0]) << 24) | (uint32_t(b.p[1]) << 16) | (uint32_t(b.p[2]) << 8) | uint32_t(b.p[3]);
    return adler32_of(dst, dst_n) == want ? 0 : 5;
}


// ---- the device decoder's serial half, run on the host (atl_inflate_dev.h) ------------------------------------------
// Same templates the wave executes, over ordinary memory; the batch resolution (parallel on the device) is the obvious
// serial loop here.  Returns the decoder's Status (0 = ok and Adler-32 verified).  CPU tests and tools/fuzz_inflate.py
// hold it against zlib; it is never on the product's path.
namespace {
struct HostSink {
    const uint8_t *src;
    uint8_t *dst;
    const uint32_t *rec, *pos;  // the decoder's queue (Areas::qrec / qpos)
    uint64_t max_batch = 0;
    void tables_ready() {}
    void stored(uint64_t byte_pos, uint32_t len, uint64_t out_pos) { memcpy(dst + out_pos, src + byte_pos, len); }
    void resolve(int n, uint64_t bstart, uint64_t bend) {
        if (bend - bstart > max_batch) max_batch = bend - bstart;
        for (int i = 0; i < n; ++i) {
            if (rec[i] & dinf::kLitFlag) {
                dst[pos[i]] = uint8_t(rec[i]);
            } else {
                const uint32_t len = rec[i] & 0x1FF, dist = (rec[i] & 0x7FFFFFFF) >> 9;
                for (uint32_t j = 0; j < len; ++j) dst[pos[i] + j] = dst[pos[i] - dist + j];
            }
        }
    }
};
}  // namespace

int device_inflate_emulated(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n) {
    using namespace dinf;
    std::vector<uint32_t> words(size_t(src_n / 4 + 3), 0u);
    if (src_n) memcpy(words.data(), src, size_t(src_n));
    std::vector<uint16_t> lit(kLitCap), off(kOffCap), codes(320);
    std::vector<uint32_t> cnt(16), nxt(16);
    std::vector<uint8_t> lens(32 + 320);
    std::vector<uint32_t> qrec(kQueue), qpos(kQueue + 1), wbuf(16), sym(64);
    Areas<HostMem> A{lit.data(), off.data(), codes.data(), cnt.data(), nxt.data(), lens.data(), qrec.data(), qpos.data(), wbuf.data(), sym.data()};
    HostSink sink{};
    sink.src = reinterpret_cast<const uint8_t *>(words.data());
    sink.dst = dst;
    sink.rec = qrec.data();
    sink.pos = qpos.data();
    uint32_t want = 0;
    const int st = inflate_stream<HostMem, HostWave, HostWindow<HostMem>, HostSink>(A, words.data(), uint32_t((src_n + 3) / 4), src_n, dst_n, sink, &want);
    if (st) return st;
    if (sink.max_batch > uint64_t(kStage)) return 100;  // the staging area of the device would have overflowed
    return adler32_of(dst, dst_n) == want ? kOk : kAdler;
}

// ---- the segment scheme (atl_inflate_dev.h, "SEGMENTS") on the host: finder, count pass, chain, decode pass with markers,
// resolve - the same templates and tests the kernels run, the parallel parts as loops.  *n_segments: how many segments the
// stream's chain had (1 = the finder found nothing to split at).  CPU tests hold it against zlib.
namespace {
struct VecSplits {
    const std::vector<uint64_t> *v;
    bool is_split(uint64_t bit) const { return std::binary_search(v->begin(), v->end(), bit); }
};
struct CountSinkH {
    void tables_ready() {}
    void stored(uint64_t, uint32_t, uint64_t) {}
    void resolve(int, uint64_t, uint64_t) {}
};
struct MarkSinkH {
    const uint8_t *src;
    uint8_t *val, *mark;
    const uint32_t *rec, *pos;
    uint64_t seg0 = 0, max_batch = 0;
    void tables_ready() {}
    void stored(uint64_t byte_pos, uint32_t len, uint64_t out_pos) {
        memcpy(val + out_pos, src + byte_pos, len);
        memset(mark + out_pos, 0, len);
    }
    void resolve(int n, uint64_t bstart, uint64_t bend) {
        if (bend - bstart > max_batch) max_batch = bend - bstart;
        for (int i = 0; i < n; ++i) {
            if (rec[i] & dinf::kLitFlag) {
                val[pos[i]] = uint8_t(rec[i]);
                mark[pos[i]] = 0;
                continue;
            }
            const uint32_t len = rec[i] & 0x1FF, dist = (rec[i] & 0x7FFFFFFF) >> 9;
            for (uint32_t j = 0; j < len; ++j) {
                const int64_t sp = int64_t(pos[i]) + j - int64_t(dist);
                if (sp < int64_t(seg0)) {  // before this segment: a marker
                    const uint32_t back = uint32_t(int64_t(seg0) - sp);
                    val[pos[i] + j] = uint8_t(dinf::marker_lo(back));
                    mark[pos[i] + j] = uint8_t(dinf::marker_hi(back));
                } else {
                    val[pos[i] + j] = val[sp];
                    mark[pos[i] + j] = mark[sp];
                }
            }
        }
    }
};
}  // namespace

int device_inflate_split_emulated(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n, int *n_segments) {
    using namespace dinf;
    if (n_segments) *n_segments = 0;
    std::vector<uint32_t> words(size_t(src_n / 4 + 4), 0u);
    if (src_n) memcpy(words.data(), src, size_t(src_n));
    const uint32_t n_words = uint32_t((src_n + 3) / 4);
    if (src_n < 6 || !zlib_header_ok(words[0])) return kBadHeader;
    std::vector<uint16_t> lit(kLitCap), off(kOffCap), codes(320);
    std::vector<uint32_t> cnt(16), nxt(16);
    std::vector<uint8_t> lens(32 + 320);
    std::vector<uint32_t> qrec(kQueue), qpos(kQueue + 1), wbuf(16), sym(64);
    Areas<HostMem> A{lit.data(), off.data(), codes.data(), cnt.data(), nxt.data(), lens.data(), qrec.data(), qpos.data(), wbuf.data(), sym.data()};
    // 1. find
    const uint64_t src_bits = src_n * 8;
    std::vector<uint64_t> cands;
    uint8_t tab[128];
    for (uint64_t p = 17; p + 74 < src_bits; ++p) {
        if (!find_l1(bits64_at(words.data(), n_words, p), bits64_at(words.data(), n_words, p + 64))) continue;
        if (header_l2(words.data(), n_words, p, src_bits, tab)) cands.push_back(p);
    }
    // 2. count, from the stream's first block and from every candidate
    std::vector<uint64_t> starts;
    starts.push_back(16);
    starts.insert(starts.end(), cands.begin(), cands.end());
    const VecSplits splits{&cands};
    std::vector<SegOut> res(starts.size());
    std::vector<int> status(starts.size());
    for (size_t t = 0; t < starts.size(); ++t) {
        CountSinkH sink;
        status[t] = inflate_segment<HostMem, HostWave, HostWindow<HostMem>, CountSinkH, VecSplits>(
            A, words.data(), n_words, src_n, starts[t], 0, t == 0 ? 0u : kSegSlack, dst_n, splits, sink, &res[t]);
    }
    // the chain
    std::vector<size_t> chain;
    std::vector<uint64_t> seg0;
    uint64_t at = 0;
    uint32_t want = 0;
    for (size_t t = 0;;) {
        if (status[t]) return status[t];
        chain.push_back(t);
        seg0.push_back(at);
        at += res[t].out_end;
        if (at > dst_n) return kOutputFull;
        if (res[t].is_final) {
            want = res[t].adler;
            break;
        }
        const auto it = std::lower_bound(starts.begin() + 1, starts.end(), res[t].end_bit);
        if (it == starts.end() || *it != res[t].end_bit) return 101;  // a segment stops only where another one starts
        t = size_t(it - starts.begin());
    }
    if (at != dst_n) return kShort;
    if (n_segments) *n_segments = int(chain.size());
    // 3. decode with markers
    std::vector<uint8_t> mark(size_t(dst_n) + 1, 0);
    for (size_t c = 0; c < chain.size(); ++c) {
        MarkSinkH sink{};
        sink.src = reinterpret_cast<const uint8_t *>(words.data());
        sink.val = dst;
        sink.mark = mark.data();
        sink.rec = qrec.data();
        sink.pos = qpos.data();
        sink.seg0 = seg0[c];
        SegOut o{};
        const uint32_t slack = uint32_t(std::min<uint64_t>(kSegSlack, seg0[c]));
        const int st = inflate_segment<HostMem, HostWave, HostWindow<HostMem>, MarkSinkH, VecSplits>(
            A, words.data(), n_words, src_n, starts[chain[c]], seg0[c], slack, dst_n, splits, sink, &o);
        if (st) return st;
        if (sink.max_batch > uint64_t(kStage)) return 100;
        if (o.out_end != seg0[c] + res[chain[c]].out_end || o.end_bit != res[chain[c]].end_bit) return 102;  // both passes agree
    }
    // 4. resolve, segment after segment
    uint64_t n_markers = 0;
    for (size_t c = 1; c < chain.size(); ++c) {
        const uint64_t end = c + 1 < chain.size() ? seg0[c + 1] : dst_n;
        for (uint64_t i = seg0[c]; i < end; ++i)
            if (mark[i] & 0x80u) {
                dst[i] = dst[seg0[c] - marker_back(dst[i], mark[i])];
                ++n_markers;
            }
    }
    if (getenv("ATLITE_HIP_INGEST_DEBUG"))
        fprintf(stderr, "[atlite-hip inflate] segment scheme (host emulation): %zu segments, %llu of %llu bytes were markers\n", chain.size(),
                (unsigned long long)n_markers, (unsigned long long)dst_n);
    return adler32_of(dst, dst_n) == want ? kOk : kAdler;
}

}}  // namespace atl::h5
