// Read-only HDF5 container parser (see atl_h5.h for scope).  Host only; everything is parsed
// straight out of a private read-only mapping of the file, chunk payloads are inflated from it
// without an intermediate read() copy.
#include "atl_h5.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "atl_internal.h"

namespace atl { namespace h5 {

namespace {
struct Err {
    int code;
    std::string msg;
};
[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Err{code, buf};
}
inline int log2floor(uint64_t x) {
    int r = -1;
    while (x) {
        x >>= 1;
        ++r;
    }
    return r;
}
inline int enc_size(uint64_t x) { return log2floor(x) / 8 + 1; }  // H5VM_limit_enc_size
constexpr int kMaxDepth = 64;
}  // namespace

File::~File() {
    if (map_) munmap(const_cast<uint8_t *>(map_), size_);
    if (fd_ >= 0) ::close(fd_);
}

const uint8_t *File::at(uint64_t off, uint64_t n) const {
    if (off > size_ || n > size_ - off)
        fail(ATL_E_INVALID, "corrupt or truncated HDF5 file: %llu bytes at offset %llu exceed the file size %llu",
             (unsigned long long)n, (unsigned long long)off, (unsigned long long)size_);
    return map_ + off;
}

uint64_t File::rd(const uint8_t *p, int n) const {
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

bool File::undef(uint64_t a) const { return a == (O_ >= 8 ? ~0ull : ((1ull << (8 * O_)) - 1)); }

int File::open(const char *path) {
    int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        set_error("atl_nc_open: cannot open '%s': %s", path, strerror(errno));
        return ATL_E_INVALID;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 48) {
        ::close(fd);
        set_error("atl_nc_open: '%s' is not an HDF5 / NetCDF-4 file (too small)", path);
        return ATL_E_INVALID;
    }
    size_ = uint64_t(st.st_size);
    void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        ::close(fd);
        set_error("atl_nc_open: mmap of '%s' failed: %s", path, strerror(errno));
        map_ = nullptr;
        return ATL_E_NOMEM;
    }
    map_ = static_cast<const uint8_t *>(m);
    fd_ = fd;
    try {
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        uint64_t sb = ~0ull;
        for (uint64_t off = 0; off + 8 <= size_; off = off ? off * 2 : 512) {
            if (memcmp(map_ + off, sig, 8) == 0) {
                sb = off;
                break;
            }
        }
        if (sb == ~0ull) {
            if (memcmp(map_, "CDF", 3) == 0)
                fail(ATL_E_UNSUPPORTED, "'%s' is a classic NetCDF-3 file; only NetCDF-4 (HDF5) cutouts are read", path);
            fail(ATL_E_INVALID, "'%s' is not an HDF5 / NetCDF-4 file (no superblock signature)", path);
        }
        const uint8_t *p = at(sb, 48);
        const int ver = p[8];
        uint64_t root = 0;
        if (ver <= 1) {
            O_ = p[13];
            L_ = p[14];
            if ((O_ != 4 && O_ != 8) || (L_ != 4 && L_ != 8)) fail(ATL_E_INVALID, "unsupported offset/length sizes");
            uint64_t q = sb + 24 + (ver == 1 ? 4 : 0);
            const uint8_t *r = at(q, 4 * O_ + 2 * O_ + 24);
            base_ = addr(r);
            root = addr(r + 4 * O_ + O_);  // root symbol table entry: link name offset, header address
        } else if (ver <= 3) {
            O_ = p[9];
            L_ = p[10];
            if ((O_ != 4 && O_ != 8) || (L_ != 4 && L_ != 8)) fail(ATL_E_INVALID, "unsupported offset/length sizes");
            const uint8_t *r = at(sb + 12, 4 * O_);
            base_ = addr(r);
            root = addr(r + 3 * O_);
        } else {
            fail(ATL_E_UNSUPPORTED, "HDF5 superblock version %d is not supported", ver);
        }
        if (undef(base_)) base_ = 0;
        std::vector<Msg> msgs;
        read_header(root, msgs);
        collect_attrs(msgs, gattrs_);
        walk(root, "", 0);
        resolve_dims();
    } catch (const Err &e) {
        set_error("atl_nc_open('%s'): %s", path, e.msg.c_str());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_error("atl_nc_open('%s'): out of host memory", path);
        return ATL_E_NOMEM;
    }
    return ATL_OK;
}

const Dataset *File::find(const std::string &name) const {
    for (auto &d : dsets_)
        if (d.name == name) return &d;
    return nullptr;
}

// ---- object headers -------------------------------------------------------------------------
void File::read_header(uint64_t a, std::vector<Msg> &out) const {
    const uint8_t *p = at(base_ + a, 16);
    if (memcmp(p, "OHDR", 4) == 0) {
        if (p[4] != 2) fail(ATL_E_UNSUPPORTED, "object header version %d", p[4]);
        const int flags = p[5];
        uint64_t q = 6;
        if (flags & 0x20) q += 16;
        if (flags & 0x10) q += 4;
        const int szb = 1 << (flags & 3);
        const uint64_t chunk0 = rd(at(base_ + a + q, szb), szb);
        q += szb;
        parse_v2_block(at(base_ + a + q, chunk0), chunk0, (flags & 0x04) != 0, out, 0);
    } else {
        if (p[0] != 1) fail(ATL_E_INVALID, "bad object header at offset %llu", (unsigned long long)a);
        int left = int(rd(p + 2, 2));
        const uint64_t hsize = rd(p + 8, 4);
        parse_v1_block(at(base_ + a + 16, hsize), hsize, out, left, 0);
    }
}

void File::parse_v1_block(const uint8_t *p, uint64_t n, std::vector<Msg> &out, int &left, int depth) const {
    if (depth > kMaxDepth) fail(ATL_E_INVALID, "object header continuation chain too deep");
    uint64_t pos = 0;
    while (left > 0 && pos + 8 <= n) {
        const int type = int(rd(p + pos, 2));
        const uint32_t size = uint32_t(rd(p + pos + 2, 2));
        const int flags = p[pos + 4];
        if (pos + 8 + size > n) break;
        --left;
        const uint8_t *data = p + pos + 8;
        if (type == 0x10) {
            const uint64_t off = addr(data), len = rd(data + O_, L_);
            parse_v1_block(at(base_ + off, len), len, out, left, depth + 1);
        } else {
            out.push_back({type, flags, data, size});
        }
        pos += 8 + size;
    }
}

void File::parse_v2_block(const uint8_t *p, uint64_t n, bool track, std::vector<Msg> &out, int depth) const {
    if (depth > kMaxDepth) fail(ATL_E_INVALID, "object header continuation chain too deep");
    const uint64_t hdr = 4 + (track ? 2 : 0);
    uint64_t pos = 0;
    while (pos + hdr <= n) {
        const int type = p[pos];
        const uint32_t size = uint32_t(rd(p + pos + 1, 2));
        const int flags = p[pos + 3];
        if (pos + hdr + size > n) break;
        const uint8_t *data = p + pos + hdr;
        if (type == 0x10) {
            const uint64_t off = addr(data), len = rd(data + O_, L_);
            if (len < 8) fail(ATL_E_INVALID, "bad continuation block");
            const uint8_t *c = at(base_ + off, len);
            if (memcmp(c, "OCHK", 4) != 0) fail(ATL_E_INVALID, "bad continuation block signature");
            parse_v2_block(c + 4, len - 8, track, out, depth + 1);
        } else {
            out.push_back({type, flags, data, size});
        }
        pos += hdr + size;
    }
}

// ---- groups ---------------------------------------------------------------------------------
void File::parse_link(const uint8_t *p, uint64_t n, std::vector<std::pair<std::string, uint64_t>> &out) const {
    if (n < 4 || p[0] != 1) return;
    const int flags = p[1];
    uint64_t q = 2;
    int type = 0;
    if (flags & 0x08) type = p[q++];
    if (flags & 0x04) q += 8;
    if (flags & 0x10) q += 1;
    const int lsz = 1 << (flags & 3);
    if (q + lsz > n) return;
    const uint64_t len = rd(p + q, lsz);
    q += lsz;
    if (q + len > n) return;
    std::string name(reinterpret_cast<const char *>(p + q), size_t(len));
    q += len;
    if (type == 0 && q + O_ <= n) out.emplace_back(name, addr(p + q));  // hard links only
}

void File::walk_group_btree(uint64_t node, const uint8_t *heap, uint64_t heap_n,
                            std::vector<std::pair<std::string, uint64_t>> &out, int depth) const {
    if (depth > kMaxDepth) fail(ATL_E_INVALID, "group B-tree too deep");
    const uint8_t *p = at(base_ + node, 8 + 2 * O_);
    if (memcmp(p, "TREE", 4) != 0 || p[4] != 0) fail(ATL_E_INVALID, "bad group B-tree node");
    const int level = p[5];
    const int n = int(rd(p + 6, 2));
    const uint64_t q = 8 + 2 * O_;
    p = at(base_ + node, q + uint64_t(n) * (L_ + O_) + L_);
    for (int i = 0; i < n; ++i) {
        const uint64_t child = addr(p + q + L_ + uint64_t(i) * (L_ + O_));
        if (level > 0) {
            walk_group_btree(child, heap, heap_n, out, depth + 1);
            continue;
        }
        const uint8_t *s = at(base_ + child, 8);
        if (memcmp(s, "SNOD", 4) != 0) fail(ATL_E_INVALID, "bad symbol table node");
        const int nsym = int(rd(s + 6, 2));
        const uint64_t esz = 2 * O_ + 24;
        s = at(base_ + child, 8 + nsym * esz);
        for (int k = 0; k < nsym; ++k) {
            const uint8_t *e = s + 8 + k * esz;
            const uint64_t noff = rd(e, O_);
            if (noff >= heap_n) fail(ATL_E_INVALID, "symbol name outside the local heap");
            const char *nm = reinterpret_cast<const char *>(heap + noff);
            out.emplace_back(std::string(nm, strnlen(nm, heap_n - noff)), addr(e + O_));
        }
    }
}

void File::list_group(const std::vector<Msg> &msgs, std::vector<std::pair<std::string, uint64_t>> &out) const {
    for (auto &m : msgs) {
        if (m.type == 0x11 && m.size >= uint32_t(2 * O_)) {
            const uint64_t bt = addr(m.p), hp = addr(m.p + O_);
            const uint8_t *h = at(base_ + hp, 8 + 2 * L_ + O_);
            if (memcmp(h, "HEAP", 4) != 0) fail(ATL_E_INVALID, "bad local heap");
            const uint64_t dsz = rd(h + 8, L_), daddr = addr(h + 8 + 2 * L_);
            const uint8_t *hd = at(base_ + daddr, dsz);
            if (!undef(bt)) walk_group_btree(bt, hd, dsz, out, 0);
        } else if (m.type == 0x06) {
            parse_link(m.p, m.size, out);
        } else if (m.type == 0x02 && m.size >= 2) {
            const int flags = m.p[1];
            uint64_t q = 2;
            if (flags & 1) q += 8;
            if (q + 2 * O_ > m.size) continue;
            const uint64_t fh = addr(m.p + q), bt = addr(m.p + q + O_);
            if (undef(fh)) continue;
            std::vector<std::pair<const uint8_t *, uint64_t>> objs;
            fractal_objects(fh, bt, 4, 7, objs);
            for (auto &o : objs) parse_link(o.first, o.second, out);
        }
    }
}

// ---- B-tree v2: collect every record ----------------------------------------------------------
void File::btree2_records(uint64_t hdr, std::vector<const uint8_t *> &recs, int *rec_size) const {
    const uint8_t *p = at(base_ + hdr, 16 + O_ + 2 + L_ + 4);
    if (memcmp(p, "BTHD", 4) != 0) fail(ATL_E_INVALID, "bad v2 B-tree header");
    const uint64_t node_size = rd(p + 6, 4);
    const int rsz = int(rd(p + 10, 2));
    const int depth = int(rd(p + 12, 2));
    const uint64_t root = addr(p + 16);
    const int nroot = int(rd(p + 16 + O_, 2));
    *rec_size = rsz;
    if (undef(root) || nroot == 0) return;
    if (rsz <= 0 || node_size < 16 || depth > 16) fail(ATL_E_INVALID, "bad v2 B-tree header fields");
    std::vector<uint64_t> max_nrec(depth + 1), cum_max(depth + 1);
    std::vector<int> cum_size(depth + 1, 0);
    max_nrec[0] = (node_size - 10) / rsz;
    cum_max[0] = max_nrec[0];
    const int nrec_size = enc_size(max_nrec[0]);
    for (int d = 1; d <= depth; ++d) {
        const uint64_t ptr = O_ + nrec_size + cum_size[d - 1];
        max_nrec[d] = (node_size - 10 - ptr) / (rsz + ptr);
        cum_max[d] = (max_nrec[d] + 1) * cum_max[d - 1] + max_nrec[d];
        cum_size[d] = enc_size(cum_max[d]);
    }
    struct Rec {
        const File *f;
        std::vector<const uint8_t *> &recs;
        int rsz, nrec_size;
        const std::vector<int> &cum_size;
        uint64_t node_size;
        void node(uint64_t a, int nrec, int d) const {
            const uint8_t *n = f->at(f->base_ + a, node_size);
            if (uint64_t(6) + uint64_t(nrec) * rsz > node_size) fail(ATL_E_INVALID, "bad v2 B-tree node");
            if (d == 0) {
                if (memcmp(n, "BTLF", 4) != 0) fail(ATL_E_INVALID, "bad v2 B-tree leaf");
                for (int i = 0; i < nrec; ++i) recs.push_back(n + 6 + i * rsz);
                return;
            }
            if (memcmp(n, "BTIN", 4) != 0) fail(ATL_E_INVALID, "bad v2 B-tree internal node");
            for (int i = 0; i < nrec; ++i) recs.push_back(n + 6 + i * rsz);
            const uint64_t ptr = f->O_ + nrec_size + (d > 1 ? cum_size[d - 1] : 0);
            const uint8_t *c = n + 6 + uint64_t(nrec) * rsz;
            if (uint64_t(6) + uint64_t(nrec) * rsz + (nrec + 1) * ptr > node_size)
                fail(ATL_E_INVALID, "bad v2 B-tree node");
            for (int i = 0; i <= nrec; ++i) {
                const uint8_t *e = c + i * ptr;
                node(f->addr(e), int(f->rd(e + f->O_, nrec_size)), d - 1);
            }
        }
    };
    Rec r{this, recs, rsz, nrec_size, cum_size, node_size};
    r.node(root, nroot, depth);
}

// ---- fractal heap: resolve the heap IDs listed by a v2 B-tree ------------------------------------
void File::fractal_objects(uint64_t heap_addr, uint64_t btree_addr, int rec_id_off, int id_len,
                           std::vector<std::pair<const uint8_t *, uint64_t>> &out) const {
    const uint64_t fixed = 14 + 10 * uint64_t(L_) + 2 * uint64_t(O_);
    const uint8_t *h = at(base_ + heap_addr, fixed + 2 + 2 * L_ + 2 + 2 + O_ + 2);
    if (memcmp(h, "FRHP", 4) != 0) fail(ATL_E_INVALID, "bad fractal heap header");
    const int heap_id_len = int(rd(h + 5, 2));
    const int filt_len = int(rd(h + 7, 2));
    const uint64_t max_managed = rd(h + 10, 4);
    if (filt_len) fail(ATL_E_UNSUPPORTED, "filtered fractal heaps are not supported");
    const uint8_t *t = h + fixed;
    const uint64_t width = rd(t, 2);
    const uint64_t start = rd(t + 2, L_);
    const uint64_t max_direct = rd(t + 2 + L_, L_);
    const int heap_bits = int(rd(t + 2 + 2 * L_, 2));
    const uint64_t root = addr(t + 2 + 2 * L_ + 4);
    const int cur_rows = int(rd(t + 2 + 2 * L_ + 4 + O_, 2));
    if (!width || !start || (start & (start - 1)) || (width & (width - 1)) || max_direct < start)
        fail(ATL_E_INVALID, "bad fractal heap doubling table");
    // row r holds blocks of start << max(r - 1, 0) bytes: the shifts below must stay inside 64 bits
    if (heap_bits < 1 || heap_bits > 64 || cur_rows < 0 || log2floor(start) + std::max(cur_rows - 2, 0) > 62 || width > (1u << 15))
        fail(ATL_E_INVALID, "bad fractal heap doubling table");
    const int offsz = (heap_bits + 7) / 8;
    const int lensz = enc_size(std::min(max_direct, max_managed));
    const int max_drows = log2floor(max_direct) - log2floor(start) + 2;
    if (id_len > heap_id_len) id_len = heap_id_len;

    std::vector<const uint8_t *> recs;
    int rsz = 0;
    if (undef(btree_addr)) return;
    btree2_records(btree_addr, recs, &rsz);
    if (rec_id_off + 1 + offsz + lensz > rsz) fail(ATL_E_INVALID, "heap ID does not fit the B-tree record");

    for (const uint8_t *rec : recs) {
        const uint8_t *id = rec + rec_id_off;
        const int kind = (id[0] >> 4) & 3;
        if (kind == 2) {  // tiny object stored in the ID itself
            const uint64_t len = uint64_t(id[0] & 0x0f) + 1;
            if (1 + len <= uint64_t(id_len)) out.emplace_back(id + 1, len);
            continue;
        }
        if (kind != 0) fail(ATL_E_UNSUPPORTED, "huge fractal-heap objects are not supported");
        const uint64_t off = rd(id + 1, offsz), len = rd(id + 1 + offsz, lensz);
        // descend the doubling table
        uint64_t blk = root, blk_off = 0;
        int nrows = cur_rows;
        int guard = 0;
        bool direct = (nrows == 0);
        while (!direct) {
            if (++guard > kMaxDepth) fail(ATL_E_INVALID, "fractal heap too deep");
            const uint64_t hsz = 5 + O_ + offsz;
            const uint64_t ndirect = uint64_t(std::min(nrows, max_drows)) * width;
            const uint64_t nind = nrows > max_drows ? uint64_t(nrows - max_drows) * width : 0;
            const uint8_t *ib = at(base_ + blk, hsz + (ndirect + nind) * O_);
            if (memcmp(ib, "FHIB", 4) != 0) fail(ATL_E_INVALID, "bad fractal heap indirect block");
            uint64_t rel = off - blk_off, cum = 0;
            bool found = false;
            for (int r = 0; r < nrows; ++r) {
                const uint64_t bs = start << (r > 0 ? r - 1 : 0);
                const uint64_t span = bs * width;
                if (rel < span) {
                    const uint64_t col = rel / bs;
                    const uint64_t child_off = blk_off + cum + col * bs;
                    if (r < max_drows) {
                        blk = addr(ib + hsz + (uint64_t(r) * width + col) * O_);
                        direct = true;
                    } else {
                        blk = addr(ib + hsz + (ndirect + uint64_t(r - max_drows) * width + col) * O_);
                        nrows = log2floor(bs) - (log2floor(start) + log2floor(width)) + 1;
                    }
                    blk_off = child_off;
                    found = true;
                    break;
                }
                rel -= span;
                cum += span;
            }
            if (!found || undef(blk)) fail(ATL_E_INVALID, "fractal heap offset outside the heap");
        }
        if (off < blk_off) fail(ATL_E_INVALID, "bad fractal heap offset");
        const uint8_t *db = at(base_ + blk, 5);
        if (memcmp(db, "FHDB", 4) != 0) fail(ATL_E_INVALID, "bad fractal heap direct block");
        out.emplace_back(at(base_ + blk + (off - blk_off), len), len);
    }
}

// ---- datatypes, dataspaces, attributes -----------------------------------------------------------
void File::parse_datatype(const uint8_t *p, uint64_t n, Datatype &t) const {
    if (n < 8) fail(ATL_E_INVALID, "short datatype message");
    const int cls = p[0] & 0x0f;
    const int b0 = p[1];
    t.size = uint32_t(rd(p + 4, 4));
    switch (cls) {
        case 0:
            t.cls = TypeClass::Fixed;
            t.big_endian = b0 & 1;
            t.is_signed = (b0 & 8) != 0;
            break;
        case 1:
            t.cls = TypeClass::Float;
            t.big_endian = b0 & 1;
            break;
        case 3:
            t.cls = TypeClass::String;
            break;
        case 7:
            t.cls = TypeClass::Reference;
            break;
        case 9: {
            t.cls = (b0 & 0x0f) == 1 ? TypeClass::VlenStr : TypeClass::VlenSeq;
            if (n >= 16) {
                Datatype b;
                parse_datatype(p + 8, n - 8, b);
                t.base_size = b.size;
                t.base_cls = b.cls;
            }
            break;
        }
        default:
            t.cls = TypeClass::Other;
    }
}

void File::parse_dataspace(const uint8_t *p, uint64_t n, std::vector<uint64_t> &dims, bool *null_space,
                           std::vector<uint64_t> *max_dims) const {
    if (n < 4) fail(ATL_E_INVALID, "short dataspace message");
    const int ver = p[0], rank = p[1];
    const int sflags = p[2];  // bit 0: maximum dimensions follow the current ones
    *null_space = false;
    uint64_t q;
    if (ver == 1) {
        q = 8;
    } else if (ver == 2) {
        q = 4;
        *null_space = p[3] == 2;
    } else {
        fail(ATL_E_UNSUPPORTED, "dataspace message version %d", ver);
    }
    if (rank > 32 || q + uint64_t(rank) * L_ > n) fail(ATL_E_INVALID, "bad dataspace message");
    dims.resize(rank);
    // the element count must stay far inside 64 bits: readers multiply the dimensions (2^48 elements = 2 PiB of fp64)
    uint64_t total = 1;
    for (int i = 0; i < rank; ++i) {
        dims[i] = rd(p + q + uint64_t(i) * L_, L_);
        if (dims[i] && total > (1ull << 48) / dims[i]) fail(ATL_E_INVALID, "dataspace with more than 2^48 elements");
        total *= dims[i] ? dims[i] : 1;
    }
    if (max_dims) {
        max_dims->clear();
        if ((sflags & 1) && q + 2ull * rank * L_ <= n)
            for (int i = 0; i < rank; ++i) max_dims->push_back(rd(p + q + uint64_t(rank + i) * L_, L_));
    }
}

bool File::parse_attribute(const uint8_t *p, uint64_t n, Attribute &a) const {
    if (n < 8) return false;
    const int ver = p[0];
    if (ver < 1 || ver > 3) return false;
    const int flags = ver >= 2 ? p[1] : 0;
    if (flags & 3) return false;  // shared datatype / dataspace: not needed for cutouts
    const uint64_t nsz = rd(p + 2, 2), tsz = rd(p + 4, 2), ssz = rd(p + 6, 2);
    uint64_t q = ver == 3 ? 9 : 8;
    auto pad = [&](uint64_t v) { return ver == 1 ? (v + 7) / 8 * 8 : v; };
    if (q + pad(nsz) + pad(tsz) + pad(ssz) > n) return false;
    a.name.assign(reinterpret_cast<const char *>(p + q), size_t(nsz));
    while (!a.name.empty() && a.name.back() == '\0') a.name.pop_back();
    q += pad(nsz);
    parse_datatype(p + q, tsz, a.type);
    q += pad(tsz);
    bool null_space = false;
    parse_dataspace(p + q, ssz, a.dims, &null_space);
    q += pad(ssz);
    uint64_t cnt = null_space ? 0 : 1;
    for (auto d : a.dims) cnt *= d;
    a.nbytes = cnt * a.type.size;
    if (q + a.nbytes > n) return false;
    a.data = p + q;
    return true;
}

void File::collect_attrs(const std::vector<Msg> &msgs, std::vector<Attribute> &out) const {
    for (auto &m : msgs) {
        if (m.type == 0x0C) {
            if (m.flags & 2) continue;  // shared message
            Attribute a;
            if (parse_attribute(m.p, m.size, a)) out.push_back(std::move(a));
        } else if (m.type == 0x15 && m.size >= 2) {
            const int flags = m.p[1];
            uint64_t q = 2;
            if (flags & 1) q += 2;
            if (q + 2 * O_ > m.size) continue;
            const uint64_t fh = addr(m.p + q), bt = addr(m.p + q + O_);
            if (undef(fh)) continue;
            std::vector<std::pair<const uint8_t *, uint64_t>> objs;
            fractal_objects(fh, bt, 0, 8, objs);
            for (auto &o : objs) {
                Attribute a;
                if (parse_attribute(o.first, o.second, a)) out.push_back(std::move(a));
            }
        }
    }
}

bool File::vlen_payload(const uint8_t *desc, const uint8_t **p, uint64_t *n, uint32_t *count) const {
    try {
        *count = uint32_t(rd(desc, 4));
        const uint64_t ga = addr(desc + 4);
        const uint32_t idx = uint32_t(rd(desc + 4 + O_, 4));
        if (undef(ga) || ga == 0) return false;
        const uint8_t *g = at(base_ + ga, 8 + L_);
        if (memcmp(g, "GCOL", 4) != 0) return false;
        const uint64_t csz = rd(g + 8, L_);
        g = at(base_ + ga, csz);
        uint64_t q = 8 + L_;
        while (q + 8 + L_ <= csz) {
            const uint32_t oi = uint32_t(rd(g + q, 2));
            const uint64_t osz = rd(g + q + 8, L_);
            if (oi == 0) break;
            if (q + 8 + L_ + osz > csz) return false;
            if (oi == idx) {
                *p = g + q + 8 + L_;
                *n = osz;
                return true;
            }
            q += 8 + L_ + (osz + 7) / 8 * 8;
        }
    } catch (const Err &) {
    }
    return false;
}

// ---- datasets --------------------------------------------------------------------------------------
void File::walk_chunk_btree(uint64_t node, int rank, Dataset &d, int depth) const {
    if (depth > kMaxDepth) fail(ATL_E_INVALID, "chunk B-tree too deep");
    const uint8_t *p = at(base_ + node, 8 + 2 * O_);
    if (memcmp(p, "TREE", 4) != 0 || p[4] != 1) fail(ATL_E_INVALID, "bad chunk B-tree node");
    const int level = p[5];
    const int n = int(rd(p + 6, 2));
    const uint64_t q = 8 + 2 * O_;
    const uint64_t ks = 8 + 8 * uint64_t(rank + 1);
    p = at(base_ + node, q + uint64_t(n) * (ks + O_) + ks);
    for (int i = 0; i < n; ++i) {
        const uint8_t *key = p + q + uint64_t(i) * (ks + O_);
        const uint64_t child = addr(key + ks);
        if (level > 0) {
            walk_chunk_btree(child, rank, d, depth + 1);
            continue;
        }
        uint64_t lin = 0;
        bool ok = true;
        for (int k = 0; k < rank; ++k) {
            const uint64_t off = rd(key + 8 + 8 * k, 8);
            const uint64_t g = off / d.chunk[k];
            if (off % d.chunk[k] || g >= d.grid[k]) ok = false;
            lin = lin * d.grid[k] + g;
        }
        if (!ok) fail(ATL_E_INVALID, "chunk offset outside the dataset '%s'", d.name.c_str());
        d.chunks[lin] = {base_ + child, rd(key, 4), uint32_t(rd(key + 4, 4))};
    }
}

void File::fixed_array_chunks(uint64_t hdr, Dataset &d, bool filtered) const {
    const uint8_t *h = at(base_ + hdr, 8 + L_ + O_ + 4);
    if (memcmp(h, "FAHD", 4) != 0) fail(ATL_E_INVALID, "bad fixed array header");
    const int esz = h[6];
    const int page_bits = h[7];
    const uint64_t nel = rd(h + 8, L_);
    const uint64_t db = addr(h + 8 + L_);
    if (undef(db)) return;  // no chunk written yet
    if (nel < d.chunks.size()) fail(ATL_E_INVALID, "fixed array smaller than the chunk grid");
    if (page_bits < 1 || page_bits > 40) fail(ATL_E_INVALID, "bad fixed array page size (2^%d elements)", page_bits);
    const uint64_t per_page = 1ull << page_bits;
    const bool paged = nel > per_page;
    const uint64_t npages = paged ? (nel + per_page - 1) / per_page : 0;
    uint64_t q = 6 + O_;
    const uint8_t *b = at(base_ + db, q);
    if (memcmp(b, "FADB", 4) != 0) fail(ATL_E_INVALID, "bad fixed array data block");
    const uint8_t *bitmap = nullptr;
    if (paged) {
        bitmap = at(base_ + db + q, (npages + 7) / 8);
        q += (npages + 7) / 8;
        q += 4;  // checksum of the prefix
    }
    const int szb = filtered ? esz - O_ - 4 : 0;
    if (esz < O_ || (filtered && szb <= 0)) fail(ATL_E_INVALID, "bad fixed array entry size");
    const uint64_t chunk_bytes = [&] {
        uint64_t v = d.type.size;
        for (auto c : d.chunk) v *= c;
        return v;
    }();
    auto entry = [&](const uint8_t *e, uint64_t lin) {
        const uint64_t a = addr(e);
        if (undef(a)) return;
        if (filtered)
            d.chunks[lin] = {base_ + a, rd(e + O_, szb), uint32_t(rd(e + O_ + szb, 4))};
        else
            d.chunks[lin] = {base_ + a, chunk_bytes, 0};
    };
    if (!paged) {
        const uint8_t *e = at(base_ + db + q, nel * esz);
        for (uint64_t i = 0; i < d.chunks.size(); ++i) entry(e + i * esz, i);
        return;
    }
    for (uint64_t pg = 0; pg < npages; ++pg) {
        const uint64_t cnt = std::min(per_page, nel - pg * per_page);
        const bool init = (bitmap[pg / 8] >> (7 - pg % 8)) & 1;  // bit 7 first, as H5VM_bit_get
        if (init) {
            const uint8_t *e = at(base_ + db + q, cnt * esz);
            for (uint64_t i = 0; i < cnt; ++i) {
                const uint64_t lin = pg * per_page + i;
                if (lin < d.chunks.size()) entry(e + i * esz, lin);
            }
        }
        q += cnt * esz + 4;
    }
}

// bytes of the "chunk size" field of a filtered chunk's index entry (H5D_*_COMPUTE_CHUNK_SIZE_LEN)
static int chunk_size_len(uint64_t chunk_bytes) {
    int lg = 0;
    while ((chunk_bytes >> lg) > 1) ++lg;
    return std::min(8, 1 + (lg + 8) / 8);
}

// Extensible array chunk index (libver >= 1.10 with ONE unlimited dimension; H5EA*.c): header -> index block (the first
// elements, the addresses of the first data blocks, the addresses of the super blocks) -> super blocks -> data blocks
// (paged once they hold more than 2^page_bits elements).  Element i belongs to the chunk with linear index i over the
// chunk grid whose unlimited dimension was moved to the front ("swizzled", so that the array only ever grows at its end).
void File::extensible_array_chunks(uint64_t hdr, Dataset &d, bool filtered, int unlim_dim) const {
    const uint8_t *h = at(base_ + hdr, 12 + 6ull * L_ + O_ + 4);
    if (memcmp(h, "EAHD", 4) != 0) fail(ATL_E_INVALID, "bad extensible array header");
    const int esz = h[6], max_bits = h[7], idx_elmts = h[8], min_elmts = h[9], min_ptrs = h[10], page_bits = h[11];
    const uint64_t iblock = addr(h + 12 + 6ull * L_);
    if (undef(iblock)) return;  // no chunk written yet
    auto log2_exact = [](int v) {
        int l = 0;
        while ((1 << l) < v) ++l;
        return (1 << l) == v ? l : -1;
    };
    const int lg_min = log2_exact(min_elmts), lg_ptrs = log2_exact(min_ptrs);
    if (max_bits < 1 || max_bits > 64 || lg_min < 0 || lg_ptrs < 1 || max_bits < lg_min || page_bits < 1 || page_bits > 30)
        fail(ATL_E_INVALID, "bad extensible array parameters");
    const uint64_t chunk_bytes = [&] {
        uint64_t v = d.type.size;
        for (auto c : d.chunk) v *= c;
        return v;
    }();
    const int szb = filtered ? chunk_size_len(chunk_bytes) : 0;
    if (esz != (filtered ? O_ + szb + 4 : O_)) fail(ATL_E_INVALID, "bad extensible array element size");
    const int rank = int(d.grid.size());
    const uint64_t total = d.chunks.size();
    // element index -> row-major index over the grid (undo the swizzle)
    std::vector<int> order;
    order.push_back(unlim_dim);
    for (int i = 0; i < rank; ++i)
        if (i != unlim_dim) order.push_back(i);
    auto place = [&](uint64_t e, const uint8_t *el) {
        if (e >= total) return;  // allocated beyond the grid
        const uint64_t a = addr(el);
        if (undef(a)) return;
        uint64_t lin = e;
        if (unlim_dim != 0) {
            std::vector<uint64_t> c(rank);
            uint64_t r = e;
            for (int k = rank - 1; k >= 0; --k) {
                c[order[k]] = r % d.grid[order[k]];
                r /= d.grid[order[k]];
            }
            lin = 0;
            for (int k = 0; k < rank; ++k) lin = lin * d.grid[k] + c[k];
        }
        if (filtered)
            d.chunks[lin] = {base_ + a, rd(el + O_, szb), uint32_t(rd(el + O_ + szb, 4))};
        else
            d.chunks[lin] = {base_ + a, chunk_bytes, 0};
    };
    const int nsblks = 1 + (max_bits - lg_min);
    const int iblock_sblks = 2 * lg_ptrs;                   // super blocks whose data blocks hang off the index block
    const uint64_t ndblk_addrs = 2ull * (min_ptrs - 1);
    const int nsblk_addrs = std::max(0, nsblks - iblock_sblks);
    const int off_sz = (max_bits + 7) / 8;
    const uint64_t page_nelmts = 1ull << page_bits;
    const uint64_t ib_size = 6 + O_ + uint64_t(idx_elmts) * esz + ndblk_addrs * O_ + uint64_t(nsblk_addrs) * O_ + 4;
    const uint8_t *ib = at(base_ + iblock, ib_size);
    if (memcmp(ib, "EAIB", 4) != 0) fail(ATL_E_INVALID, "bad extensible array index block");
    uint64_t q = 6 + O_;
    for (int i = 0; i < idx_elmts; ++i) place(uint64_t(i), ib + q + uint64_t(i) * esz);
    q += uint64_t(idx_elmts) * esz;
    const uint8_t *dblk_addrs = ib + q;
    const uint8_t *sblk_addrs = dblk_addrs + ndblk_addrs * O_;
    // one data block: `nelmts` elements starting at array index `first`
    auto data_block = [&](uint64_t a, uint64_t nelmts, uint64_t first, const uint8_t *page_init, uint64_t page_bit0) {
        if (undef(a) || first >= total) return;
        const uint64_t prefix = 6 + O_ + off_sz;
        const uint8_t *b = at(base_ + a, prefix);
        if (memcmp(b, "EADB", 4) != 0) fail(ATL_E_INVALID, "bad extensible array data block");
        if (nelmts <= page_nelmts) {
            const uint8_t *e = at(base_ + a + prefix, nelmts * esz);
            for (uint64_t i = 0; i < nelmts; ++i) place(first + i, e + i * esz);
            return;
        }
        if (!page_init) fail(ATL_E_UNSUPPORTED, "paged extensible array data block below the index block");
        const uint64_t npages = nelmts / page_nelmts, page_size = page_nelmts * esz + 4;
        for (uint64_t pg = 0; pg < npages; ++pg) {
            const uint64_t bit = page_bit0 + pg;
            if (!((page_init[bit / 8] >> (7 - bit % 8)) & 1)) continue;  // bit 7 first, as H5VM_bit_get
            const uint8_t *e = at(base_ + a + prefix + 4 + pg * page_size, page_nelmts * esz);
            for (uint64_t i = 0; i < page_nelmts; ++i) place(first + pg * page_nelmts + i, e + i * esz);
        }
    };
    uint64_t start_idx = uint64_t(idx_elmts), start_dblk = 0;
    for (int u = 0; u < nsblks && start_idx < total; ++u) {
        const uint64_t ndblks = 1ull << (u / 2);
        const uint64_t dblk_nelmts = (1ull << ((u + 1) / 2)) * uint64_t(min_elmts);
        if (u < iblock_sblks) {
            for (uint64_t k = 0; k < ndblks; ++k) {
                if (start_dblk + k >= ndblk_addrs) fail(ATL_E_INVALID, "bad extensible array index block");
                data_block(addr(dblk_addrs + (start_dblk + k) * O_), dblk_nelmts, start_idx + k * dblk_nelmts, nullptr, 0);
            }
        } else {
            const uint64_t sa = addr(sblk_addrs + uint64_t(u - iblock_sblks) * O_);
            if (!undef(sa)) {
                const uint64_t npages = dblk_nelmts > page_nelmts ? dblk_nelmts / page_nelmts : 0;
                const uint64_t init_size = npages ? (npages + 7) / 8 : 0;
                if (ndblks > (1ull << 32)) fail(ATL_E_INVALID, "bad extensible array super block");
                const uint64_t sb_size = 6 + O_ + off_sz + ndblks * init_size + ndblks * O_ + 4;
                const uint8_t *sb = at(base_ + sa, sb_size);
                if (memcmp(sb, "EASB", 4) != 0) fail(ATL_E_INVALID, "bad extensible array super block");
                const uint8_t *page_init = sb + 6 + O_ + off_sz;
                const uint8_t *da = page_init + ndblks * init_size;
                for (uint64_t k = 0; k < ndblks; ++k) {
                    const uint64_t first = start_idx + k * dblk_nelmts;
                    if (first >= total) break;
                    // (the page bits of the blocks are packed one after another: bit k * npages + page, H5EA__lookup_elmt)
                    data_block(addr(da + k * O_), dblk_nelmts, first, npages ? page_init : nullptr, k * npages);
                }
            }
        }
        start_idx += ndblks * dblk_nelmts;
        start_dblk += ndblks;
    }
}

// v2 B-tree chunk index (more than one unlimited dimension): one record per chunk - address, [filtered: size, filter mask],
// then the chunk's grid coordinates ("scaled offsets") as 64-bit values
void File::btree2_chunks(uint64_t hdr, Dataset &d, bool filtered) const {
    std::vector<const uint8_t *> recs;
    int rsz = 0;
    btree2_records(hdr, recs, &rsz);
    const int rank = int(d.grid.size());
    const uint64_t chunk_bytes = [&] {
        uint64_t v = d.type.size;
        for (auto c : d.chunk) v *= c;
        return v;
    }();
    const int szb = filtered ? chunk_size_len(chunk_bytes) : 0;
    const int need = O_ + (filtered ? szb + 4 : 0) + 8 * rank;
    if (!recs.empty() && rsz != need) fail(ATL_E_INVALID, "bad v2 B-tree chunk record size");
    for (const uint8_t *r : recs) {
        const uint64_t a = addr(r);
        const uint8_t *sc = r + O_ + (filtered ? szb + 4 : 0);
        uint64_t lin = 0;
        for (int k = 0; k < rank; ++k) {
            const uint64_t g = rd(sc + 8 * k, 8);
            if (g >= d.grid[k]) fail(ATL_E_INVALID, "chunk offset outside the dataset '%s'", d.name.c_str());
            lin = lin * d.grid[k] + g;
        }
        if (undef(a)) continue;
        if (filtered)
            d.chunks[lin] = {base_ + a, rd(r + O_, szb), uint32_t(rd(r + O_ + szb, 4))};
        else
            d.chunks[lin] = {base_ + a, chunk_bytes, 0};
    }
}

void File::parse_dataset(const std::string &name, uint64_t a, const std::vector<Msg> &msgs, Dataset &d) const {
    d.name = name;
    d.header_addr = a;
    const Msg *space = nullptr, *type = nullptr, *layout = nullptr, *pipe = nullptr;
    for (auto &m : msgs) {
        if (m.type == 0x01) space = &m;
        if (m.type == 0x03) type = &m;
        if (m.type == 0x08) layout = &m;
        if (m.type == 0x0B) pipe = &m;
    }
    if (!space || !type || !layout) return;  // a group or a committed type
    d.is_dataset = true;
    collect_attrs(msgs, d.attrs);
    if (type->flags & 2) {
        d.type.cls = TypeClass::Other;  // committed datatype: listed, not readable
    } else {
        parse_datatype(type->p, type->size, d.type);
    }
    bool null_space = false;
    std::vector<uint64_t> max_dims;
    parse_dataspace(space->p, space->size, d.shape, &null_space, &max_dims);
    if (pipe && !(pipe->flags & 2)) {
        const uint8_t *p = pipe->p;
        const uint64_t n = pipe->size;
        if (n < 2) fail(ATL_E_INVALID, "short filter pipeline message");
        const int ver = p[0], nf = p[1];
        uint64_t q = ver == 1 ? 8 : 2;
        for (int i = 0; i < nf; ++i) {
            if (q + 4 > n) fail(ATL_E_INVALID, "bad filter pipeline message");
            Filter f;
            f.id = int(rd(p + q, 2));
            q += 2;
            uint64_t nlen = 0;
            if (ver == 1 || f.id >= 256) {
                nlen = rd(p + q, 2);
                q += 2;
            }
            q += 2;  // flags
            if (q + 2 > n) fail(ATL_E_INVALID, "bad filter pipeline message");
            const int ncd = int(rd(p + q, 2));
            q += 2;
            q += ver == 1 ? (nlen + 7) / 8 * 8 : nlen;
            if (q + 4ull * ncd > n) fail(ATL_E_INVALID, "bad filter pipeline message");
            for (int k = 0; k < ncd; ++k) f.params.push_back(uint32_t(rd(p + q + 4 * k, 4)));
            q += 4ull * ncd;
            if (ver == 1 && (ncd & 1)) q += 4;
            d.filters.push_back(std::move(f));
        }
    }
    const uint8_t *p = layout->p;
    const uint64_t n = layout->size;
    if (n < 2) fail(ATL_E_INVALID, "short layout message");
    const int ver = p[0];
    if (ver != 3 && ver != 4) {
        d.layout = -1;
        return;  // pre-1.6 layouts: listed, read reports unsupported
    }
    const int cls = p[1];
    const int rank = int(d.shape.size());
    d.layout = cls;
    d.chunk = d.shape;
    if (cls == 0) {
        const uint64_t sz = rd(p + 2, 2);
        if (4 + sz > n) fail(ATL_E_INVALID, "bad compact layout");
        d.compact = p + 4;
        d.contiguous_size = sz;
    } else if (cls == 1) {
        d.contiguous_addr = addr(p + 2);
        d.contiguous_size = rd(p + 2 + O_, L_);
        if (!undef(d.contiguous_addr)) d.contiguous_addr += base_;
    } else if (cls == 2) {
        uint64_t q;
        int dimty;
        std::vector<uint64_t> cd;
        uint64_t bt = ~0ull;
        int idx = 0;
        int lflags = 0;
        if (ver == 3) {
            dimty = p[2];
            bt = addr(p + 3);
            q = 3 + O_;
            if (q + 4ull * dimty > n) fail(ATL_E_INVALID, "bad chunked layout");
            for (int i = 0; i < dimty; ++i) cd.push_back(rd(p + q + 4 * i, 4));
        } else {
            lflags = p[2];
            dimty = p[3];
            const int enc = p[4];
            q = 5;
            if (enc < 1 || enc > 8 || q + uint64_t(enc) * dimty + 1 > n) fail(ATL_E_INVALID, "bad chunked layout");
            for (int i = 0; i < dimty; ++i) cd.push_back(rd(p + q + uint64_t(enc) * i, enc));
            q += uint64_t(enc) * dimty;
            idx = p[q++];
        }
        if (dimty != rank + 1) fail(ATL_E_INVALID, "chunk rank does not match the dataspace of '%s'", name.c_str());
        d.chunk.assign(cd.begin(), cd.begin() + rank);
        d.grid.resize(rank);
        uint64_t total = 1;
        for (int i = 0; i < rank; ++i) {
            if (!d.chunk[i]) fail(ATL_E_INVALID, "zero chunk dimension");
            d.grid[i] = (d.shape[i] + d.chunk[i] - 1) / d.chunk[i];
            total *= d.grid[i];
        }
        if (total > (1ull << 22)) fail(ATL_E_UNSUPPORTED, "dataset '%s' has too many chunks", name.c_str());
        d.chunks.assign(size_t(total), Chunk{});
        // HDF5 itself limits a chunk to 4 GiB - 1; checked factor by factor, the product of three 32-bit dimensions
        // read from the file would not fit 64 bits (and every reader allocates a buffer of this size per chunk)
        uint64_t chunk_bytes = d.type.size;
        if (!chunk_bytes) fail(ATL_E_INVALID, "dataset '%s' has a zero-size element type", name.c_str());
        for (auto c : d.chunk) {
            if (c > 0xffffffffull / chunk_bytes) fail(ATL_E_INVALID, "dataset '%s' has chunks of 4 GiB or more", name.c_str());
            chunk_bytes *= c;
        }
        if (ver == 3) {
            if (!undef(bt)) walk_chunk_btree(bt, rank, d, 0);
        } else if (idx == 1) {  // single chunk
            uint64_t fsize = chunk_bytes;
            uint32_t mask = 0;
            if (lflags & 2) {
                fsize = rd(p + q, L_);
                mask = uint32_t(rd(p + q + L_, 4));
                q += L_ + 4;
            }
            const uint64_t a0 = addr(p + q);
            if (!undef(a0) && total == 1) d.chunks[0] = {base_ + a0, fsize, mask};
        } else if (idx == 2) {  // implicit: consecutive, unfiltered
            const uint64_t a0 = addr(p + q);
            if (!undef(a0))
                for (uint64_t i = 0; i < total; ++i) d.chunks[i] = {base_ + a0 + i * chunk_bytes, chunk_bytes, 0};
        } else if (idx == 3) {  // fixed array
            const uint64_t fa = addr(p + q + 1);
            if (!undef(fa)) fixed_array_chunks(fa, d, !d.filters.empty());
        } else if (idx == 4) {  // extensible array: one unlimited dimension (H5Dearray.c)
            if (q + 5 + O_ > n) fail(ATL_E_INVALID, "bad chunked layout");
            int unlim = -1;
            for (int i = 0; i < rank && i < int(max_dims.size()); ++i)
                if (max_dims[i] == ~0ull || (L_ < 8 && max_dims[i] == (1ull << (8 * L_)) - 1)) {
                    if (unlim < 0) unlim = i;
                }
            if (unlim < 0) fail(ATL_E_INVALID, "extensible array index on '%s', which has no unlimited dimension", name.c_str());
            const uint64_t ea = addr(p + q + 5);
            if (!undef(ea)) extensible_array_chunks(ea, d, !d.filters.empty(), unlim);
        } else if (idx == 5) {  // v2 B-tree: several unlimited dimensions (H5Dbtree2.c)
            if (q + 6 + O_ > n) fail(ATL_E_INVALID, "bad chunked layout");
            const uint64_t bt2 = addr(p + q + 6);
            if (!undef(bt2)) btree2_chunks(bt2, d, !d.filters.empty());
        } else {
            d.layout = -2;  // a chunk index this reader does not know
        }
        for (auto &c : d.chunks)
            if (c.size && (c.addr > size_ || c.size > size_ - c.addr))
                fail(ATL_E_INVALID, "a chunk of '%s' lies outside the file", name.c_str());
    }
}

void File::walk(uint64_t header_addr, const std::string &prefix, int depth) {
    if (depth > 8) return;
    std::vector<Msg> msgs;
    read_header(header_addr, msgs);
    std::vector<std::pair<std::string, uint64_t>> links;
    list_group(msgs, links);
    for (auto &l : links) {
        std::vector<Msg> m;
        read_header(l.second, m);
        Dataset d;
        const std::string full = prefix + l.first;
        parse_dataset(full, l.second, m, d);
        if (d.is_dataset) {
            addr_name_[l.second] = full;
            dsets_.push_back(std::move(d));
        } else if (l.second != header_addr) {
            bool is_group = false;
            for (auto &x : m) is_group |= (x.type == 0x11 || x.type == 0x02 || x.type == 0x06);
            if (is_group) walk(l.second, full + "/", depth + 1);
        }
    }
}

void File::resolve_dims() {
    for (auto &d : dsets_) {
        const int rank = int(d.shape.size());
        d.dims.assign(rank, std::string());
        const Attribute *dl = d.attr("DIMENSION_LIST");
        const uint64_t desc = 4 + uint64_t(O_) + 4;
        if (dl && dl->type.cls == TypeClass::VlenSeq && dl->nbytes >= desc * rank) {
            for (int i = 0; i < rank; ++i) {
                const uint8_t *p = nullptr;
                uint64_t n = 0;
                uint32_t cnt = 0;
                if (vlen_payload(dl->data + desc * i, &p, &n, &cnt) && cnt >= 1 && n >= uint64_t(O_)) {
                    auto it = addr_name_.find(rd(p, O_));
                    if (it != addr_name_.end()) d.dims[i] = it->second;
                }
            }
        }
        // fall back on a unique 1-d variable of the same length
        for (int i = 0; i < rank; ++i) {
            if (!d.dims[i].empty()) continue;
            if (rank == 1) {
                d.dims[i] = d.name;
                continue;
            }
            const Dataset *hit = nullptr;
            int hits = 0;
            for (auto &c : dsets_)
                if (c.shape.size() == 1 && c.shape[0] == d.shape[i]) {
                    hit = &c;
                    ++hits;
                }
            if (hits == 1) d.dims[i] = hit->name;
        }
    }
}

// ---- chunk payloads -------------------------------------------------------------------------------
int chunk_filters(const Dataset &d, const Chunk &c, uint64_t *payload_n, bool *deflate_out, bool *shuffled) {
    uint64_t n = c.size;
    *shuffled = false;
    int i_shuffle = -1, i_deflate = -1;
    bool deflate = false;
    for (int i = int(d.filters.size()) - 1; i >= 0; --i) {
        if (i < 32 && (c.mask >> i) & 1) continue;  // filter skipped for this chunk
        const int id = d.filters[i].id;
        if (id == 3) {  // fletcher32: trailing checksum
            if (n < 4 || deflate) {
                set_error("dataset '%s': unsupported filter order (fletcher32)", d.name.c_str());
                return ATL_E_UNSUPPORTED;
            }
            n -= 4;
        } else if (id == 1) {
            deflate = true;
            i_deflate = i;
        } else if (id == 2) {
            const uint32_t es = d.filters[i].params.empty() ? d.type.size : d.filters[i].params[0];
            if (es != d.type.size) {
                set_error("dataset '%s': shuffle element size %u differs from the type size %u", d.name.c_str(), es,
                          d.type.size);
                return ATL_E_UNSUPPORTED;
            }
            *shuffled = es > 1;
            i_shuffle = i;
        } else {
            set_error("dataset '%s': filter id %d is not supported (deflate, shuffle and fletcher32 are)",
                      d.name.c_str(), id);
            return ATL_E_UNSUPPORTED;
        }
    }
    if (i_shuffle >= 0 && i_deflate >= 0 && i_shuffle > i_deflate) {
        set_error("dataset '%s': shuffle after deflate is not supported", d.name.c_str());
        return ATL_E_UNSUPPORTED;
    }
    *payload_n = n;
    *deflate_out = deflate;
    return ATL_OK;
}

int chunk_inflate(const Dataset &d, const Chunk &c, const uint8_t *file_base, int fd, uint8_t *dst, uint64_t dst_n,
                  bool *shuffled) {
    const uint8_t *src = file_base + c.addr;
    uint64_t n = c.size;
    if (fd >= 0) {
        static thread_local std::vector<uint8_t> buf;
        if (buf.size() < n) buf.resize(size_t(n + n / 2));
        uint64_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, buf.data() + got, size_t(n - got), off_t(c.addr + got));
            if (r <= 0) {
                set_error("dataset '%s': read of %llu bytes at offset %llu failed", d.name.c_str(),
                          (unsigned long long)n, (unsigned long long)c.addr);
                return ATL_E_INVALID;
            }
            got += uint64_t(r);
        }
        src = buf.data();
    }
    bool deflate = false;
    {
        uint64_t payload = 0;
        const int rc = chunk_filters(d, c, &payload, &deflate, shuffled);
        if (rc) return rc;
        n = payload;
    }
    if (deflate) {
        static const bool use_fast = [] {
            const char *e = getenv("ATLITE_HIP_INFLATE");
            return !(e && strcmp(e, "zlib") == 0);
        }();
        if (use_fast && fast_inflate_zlib(src, n, dst, dst_n) == 0) return ATL_OK;
        uLongf out_n = uLongf(dst_n);  // zlib decides everything the fast decoder did not accept
        const int rc = uncompress(dst, &out_n, src, uLong(n));
        if (rc != Z_OK || out_n != dst_n) {
            set_error("dataset '%s': corrupt deflate stream (zlib rc %d, %llu of %llu bytes)", d.name.c_str(), rc,
                      (unsigned long long)out_n, (unsigned long long)dst_n);
            return ATL_E_INVALID;
        }
    } else {
        if (n != dst_n) {
            set_error("dataset '%s': stored chunk has %llu bytes, expected %llu", d.name.c_str(),
                      (unsigned long long)n, (unsigned long long)dst_n);
            return ATL_E_INVALID;
        }
        memcpy(dst, src, dst_n);
    }
    return ATL_OK;
}

}}  // namespace atl::h5
