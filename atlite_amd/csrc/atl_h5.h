// Minimal read-only HDF5 container parser for NetCDF-4 cutout files (no libhdf5 in this image).
//
// What atlite writes (atlite/data.py:139,246-248 -> xarray.to_netcdf, engine netcdf4): one root
// group, chunked datasets with the shuffle + deflate filters, dimension scales, a handful of
// attributes.  Covered here:
//   superblock v0-v3; object headers v1 and v2 (continuation blocks); groups as symbol tables
//   (B-tree v1 + local heap) and as link messages, compact or dense (fractal heap + B-tree v2);
//   attributes compact or dense; dataspace v1/v2; fixed-point / float / string / vlen-string /
//   reference datatypes; layouts compact, contiguous, chunked (v3 B-tree v1 index; v4 single-chunk,
//   implicit and fixed-array indexes); filter pipeline v1/v2 (deflate, shuffle, fletcher32);
//   global heap (vlen attribute payloads, DIMENSION_LIST).
// Not covered (-> ATL_E_UNSUPPORTED with a message): committed/shared datatypes, extensible-array
// and B-tree-v2 chunk indexes, szip/nbit/scaleoffset and third-party filters, external storage,
// virtual datasets.  Layout follows the "HDF5 File Format Specification Version 3.0".
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace atl { namespace h5 {

enum class TypeClass : int { Fixed = 0, Float = 1, String = 3, Reference = 7, VlenSeq = 90, VlenStr = 91, Other = 99 };

struct Datatype {
    TypeClass cls = TypeClass::Other;
    uint32_t size = 0;
    bool big_endian = false;
    bool is_signed = true;
    uint32_t base_size = 0;  // vlen sequences: element size of the base type
    TypeClass base_cls = TypeClass::Other;
};

struct Attribute {
    std::string name;
    Datatype type;
    std::vector<uint64_t> dims;     // empty = scalar
    const uint8_t *data = nullptr;  // points into the mapping
    uint64_t nbytes = 0;
};

struct Chunk {
    uint64_t addr = 0;   // file offset of the stored bytes; 0 with size 0 = never written
    uint64_t size = 0;   // stored (filtered) size
    uint32_t mask = 0;   // bit i set: filter i of the pipeline was skipped for this chunk
};

struct Filter {
    int id = 0;
    std::vector<uint32_t> params;
};

struct Dataset {
    std::string name;
    uint64_t header_addr = 0;
    Datatype type;
    std::vector<uint64_t> shape;
    int layout = -1;                 // 0 compact, 1 contiguous, 2 chunked
    std::vector<uint64_t> chunk;     // chunk shape (== shape for compact / contiguous)
    const uint8_t *compact = nullptr;
    uint64_t contiguous_addr = 0, contiguous_size = 0;
    std::vector<Filter> filters;     // in pipeline (write) order
    std::vector<uint64_t> grid;      // chunks per dimension
    std::vector<Chunk> chunks;       // row-major over grid
    std::vector<Attribute> attrs;
    std::vector<std::string> dims;   // dimension names (DIMENSION_LIST, else matched by length); may be empty
    bool is_dataset = false;
    const Attribute *attr(const std::string &n) const {
        for (auto &a : attrs)
            if (a.name == n) return &a;
        return nullptr;
    }
};

class File {
   public:
    File() = default;
    ~File();
    File(const File &) = delete;
    File &operator=(const File &) = delete;
    // 0 or ATL_E_*; message through atl::set_error
    int open(const char *path);
    const std::vector<Dataset> &datasets() const { return dsets_; }
    const Dataset *find(const std::string &name) const;
    const std::vector<Attribute> &global_attrs() const { return gattrs_; }
    const uint8_t *base() const { return map_; }
    int fd() const { return fd_; }
    uint64_t size() const { return size_; }
    // vlen payload (global heap object) of one vlen element descriptor {len, addr, index}
    bool vlen_payload(const uint8_t *desc, const uint8_t **p, uint64_t *n, uint32_t *count) const;
    int off_size() const { return O_; }

   private:
    struct Msg {
        int type;
        int flags;
        const uint8_t *p;
        uint32_t size;
    };
    const uint8_t *map_ = nullptr;
    int fd_ = -1;
    uint64_t size_ = 0;
    uint64_t base_ = 0;
    int O_ = 8, L_ = 8;
    std::vector<Dataset> dsets_;
    std::vector<Attribute> gattrs_;
    std::map<uint64_t, std::string> addr_name_;

    const uint8_t *at(uint64_t off, uint64_t n) const;  // bounds-checked, throws
    uint64_t rd(const uint8_t *p, int n) const;
    uint64_t addr(const uint8_t *p) const { return rd(p, O_); }
    bool undef(uint64_t a) const;
    void read_header(uint64_t a, std::vector<Msg> &out) const;
    void parse_v1_block(const uint8_t *p, uint64_t n, std::vector<Msg> &out, int &left, int depth) const;
    void parse_v2_block(const uint8_t *p, uint64_t n, bool track_order, std::vector<Msg> &out, int depth) const;
    void list_group(const std::vector<Msg> &msgs, std::vector<std::pair<std::string, uint64_t>> &out) const;
    void walk_group_btree(uint64_t node, const uint8_t *heap, uint64_t heap_n,
                          std::vector<std::pair<std::string, uint64_t>> &out, int depth) const;
    void parse_link(const uint8_t *p, uint64_t n, std::vector<std::pair<std::string, uint64_t>> &out) const;
    void fractal_objects(uint64_t heap_addr, uint64_t btree_addr, int rec_id_off, int id_len,
                         std::vector<std::pair<const uint8_t *, uint64_t>> &out) const;
    void btree2_records(uint64_t hdr, std::vector<const uint8_t *> &recs, int *rec_size) const;
    void parse_datatype(const uint8_t *p, uint64_t n, Datatype &t) const;
    void parse_dataspace(const uint8_t *p, uint64_t n, std::vector<uint64_t> &dims, bool *null_space,
                         std::vector<uint64_t> *max_dims = nullptr) const;
    bool parse_attribute(const uint8_t *p, uint64_t n, Attribute &a) const;
    void collect_attrs(const std::vector<Msg> &msgs, std::vector<Attribute> &out) const;
    void parse_dataset(const std::string &name, uint64_t a, const std::vector<Msg> &msgs, Dataset &d) const;
    void walk_chunk_btree(uint64_t node, int rank, Dataset &d, int depth) const;
    void fixed_array_chunks(uint64_t hdr, Dataset &d, bool filtered) const;
    void extensible_array_chunks(uint64_t hdr, Dataset &d, bool filtered, int unlim_dim) const;
    void btree2_chunks(uint64_t hdr, Dataset &d, bool filtered) const;
    void resolve_dims();
    void walk(uint64_t header_addr, const std::string &prefix, int depth);
};

// decode helpers shared by the host reader and the slab pipeline
// apply the reverse filter pipeline except shuffle: returns inflated bytes in dst (dst_n = chunk
// bytes); *shuffled tells the caller whether the payload is still byte-shuffled.
// The stored bytes come from pread(fd) into a per-thread buffer when fd >= 0 (many threads faulting
// pages of one mapping in contend on the address-space lock), else straight from the mapping.
int chunk_inflate(const Dataset &d, const Chunk &c, const uint8_t *file_base, int fd, uint8_t *dst, uint64_t dst_n,
                  bool *shuffled);
// what the reverse pipeline of one stored chunk amounts to: the first *payload_n stored bytes are a zlib stream
// (*deflate) or the chunk itself, still byte-shuffled or not (*shuffled); the fletcher32 trailer is cut off
int chunk_filters(const Dataset &d, const Chunk &c, uint64_t *payload_n, bool *deflate, bool *shuffled);

// zlib-wrapped DEFLATE stream -> exactly dst_n bytes (atl_inflate.cpp).  0 = done and Adler-32
// verified; non-zero = not handled (malformed, truncated, checksum or size mismatch): the caller must
// let zlib's own inflate decide.
int fast_inflate_zlib(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n);

// The serial half of the DEVICE decoder (atl_inflate_dev.h) executed on the host: a test entry (atl_inflate_probe which = 3).
// 0 = exactly dst_n bytes and the Adler-32 matches, else the decoder's dinf::Status.
int device_inflate_emulated(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n);
// ... its segment scheme (a stream's blocks decoded side by side: atl_inflate_dev.h): finder, count, chain, decode, resolve
int device_inflate_split_emulated(const uint8_t *src, uint64_t src_n, uint8_t *dst, uint64_t dst_n, int *n_segments);

}}  // namespace atl::h5
