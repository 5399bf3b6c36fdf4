// ref_index_reader.hip — reader for the reference's on-disk dense index, host code only.
//
// `<collection>/dense_hnsw/` as the Rust server writes it:
//   {k}.index   node records (models/serializer/hnsw/node.rs:19-101), 31 + 13*nb bytes each:
//                 u8 level | u32 version | u32 prop offset | u32 prop length | u32 metadata offset | u32 metadata length
//                 (0xFF x 8 = no metadata) | u32 parent ptr-offset | u32 child ptr-offset | u16 nb |
//                 nb x { u32 neighbour id | u32 neighbour ptr-offset | u8 metric tag | f32 value }   (0xFF x 13 = null slot;
//                 models/serializer/hnsw/neighbors.rs:22-61, metric_distance.rs:17-31); all little endian
//   nodes.ptr   one 8-byte entry per logical node: u32 record offset | u32 index-file id  (serializer/hnsw/latest_node.rs:18-44);
//               neighbour / parent / child links name a node by the byte offset of its entry in this file
//   prop.data   CBOR (serde_cbor 0.11) records {"id": u32, "value": Storage} (models/file_persist.rs:58-108), Storage externally
//               tagged: {"UnsignedByte": {"mag", "quant_vec": [u8..]}} | {"SubByte": {"mag", "quant_vec": [[u8..]..], "resolution"}} |
//               {"HalfPrecisionFP": {"mag", "quant_vec": [u16 bits..]}} | {"FullPrecisionFP": {"mag", "vec": [f32..]}}
// HNSWIndexData (hyper-parameters, root_vec_ptr_offset: indexes/hnsw/mod.rs:47-57) lives in LMDB on the Rust side: the host
// passes what the reader needs (the handle's cos_params + the root's ptr offset).
//
// What is loaded: the graph — per level the node ids and every node's neighbour ids in SLOT ORDER — through the same path as
// cos_index_upload_graph_level, and the root's stored code (row N).  The raw f32 vectors the exact rerank needs are not in
// these files (they live in the collection's embedding store): upload them first with cos_index_upload_vectors; with
// COS_LOAD_VERIFY_CODES the stored Storage of every vector is compared with what the device quantized from them.
// FORMAT PARITY UNPINNED: no file written by the reference is available in this image (no rustc); the reader is written from the
// serializer sources above and round-trips against tests/ref_index_writer.py, which is derived from the same sources.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

int32_t cos_push_level_ids(cos_index *ix, u32 level, std::vector<u32> &&node_ids, std::vector<u32> &&nbr_ids); // engine.hip
int32_t cos_set_root_code(cos_index *ix, const uint8_t *ref_code, float mag);                                 // engine.hip

namespace {

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    return got == out.size();
}
inline u32 rd32(const uint8_t *p) { u32 v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

// ---- minimal CBOR (RFC 8949) decoder: what serde_cbor emits for the prop records ---------------------------------
struct Cbor {
    const uint8_t *p, *end;
    bool ok = true;
    bool head(int &major, uint64_t &arg, int &info) {
        if (p >= end) return ok = false;
        const uint8_t b = *p++;
        major = b >> 5;
        info = b & 31;
        if (info < 24) arg = (uint64_t)info;
        else if (info <= 27) {
            const int nb = 1 << (info - 24);
            if (end - p < nb) return ok = false;
            arg = 0;
            for (int i = 0; i < nb; i++) arg = (arg << 8) | *p++; // big endian
        } else if (info == 31) arg = ~0ull; // indefinite length (serde_cbor does not emit it for these types)
        else return ok = false;
        return true;
    }
    // every element of an array / map takes at least one byte, so a declared length beyond the bytes that are left is a
    // corrupt or truncated record: refuse it BEFORE any allocation or loop sized by it
    bool fits(uint64_t elements) const { return elements <= (uint64_t)(end - p); }
    static constexpr int MAX_DEPTH = 32; // the prop records nest 4 deep; bounded so a crafted file cannot overflow the stack
    bool skip(int depth = 0) {
        int mj, info;
        uint64_t a;
        if (depth > MAX_DEPTH) return ok = false;
        if (!head(mj, a, info)) return false;
        switch (mj) {
        case 0: case 1: return true;
        case 2: case 3: if ((uint64_t)(end - p) < a) return ok = false; p += a; return true;
        case 4: if (!fits(a)) return ok = false; for (uint64_t i = 0; i < a; i++) if (!skip(depth + 1)) return false; return true;
        case 5: if (a > (1ull << 62) || !fits(2 * a)) return ok = false; for (uint64_t i = 0; i < 2 * a; i++) if (!skip(depth + 1)) return false; return true;
        case 6: return skip(depth + 1);
        default: return true; // simple / float: the argument bytes were consumed by head()
        }
    }
    bool uint(uint64_t &v) { int mj, info; return head(mj, v, info) && (mj == 0 || (ok = false)); }
    bool text(std::string &s) {
        int mj, info;
        uint64_t a;
        if (!head(mj, a, info) || mj != 3 || (uint64_t)(end - p) < a) return ok = false;
        s.assign((const char *)p, (size_t)a);
        p += a;
        return true;
    }
    bool map(uint64_t &n) { int mj, info; return head(mj, n, info) && ((mj == 5 && n <= (1ull << 62) && fits(2 * n)) || (ok = false)); }
    bool array(uint64_t &n) { int mj, info; return head(mj, n, info) && ((mj == 4 && fits(n)) || (ok = false)); }
    // f16 / f32 / f64 (serde_cbor writes the shortest lossless form) or an integer
    bool number(double &v) {
        int mj, info;
        uint64_t a;
        if (!head(mj, a, info)) return false;
        if (mj == 0) { v = (double)a; return true; }
        if (mj == 1) { v = -1.0 - (double)a; return true; }
        if (mj != 7) return ok = false;
        if (info == 25) { // IEEE half
            const u32 h = (u32)a, s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
            double r = e == 0 ? m * 5.9604644775390625e-08 : (e == 31 ? (m ? NAN : INFINITY) : (1.0 + m / 1024.0) * (double)(1ull << e) / 32768.0);
            v = s ? -r : r;
            return true;
        }
        if (info == 26) { u32 b = (u32)a; float f; memcpy(&f, &b, 4); v = f; return true; }
        if (info == 27) { memcpy(&v, &a, 8); return true; }
        return ok = false;
    }
};

struct PropValue {
    u32 id = 0;
    int storage = -1; // cos_storage
    u32 resolution = 0;
    float mag = 0.f;
    std::vector<uint8_t> code; // reference layout (cos_code_bytes)
};

// {"id": u32, "value": {<variant>: {...}}}
// max_code = the largest code this index could have stored (cos_code_bytes of f32 storage): anything longer is not a vector of it
bool parse_prop(const uint8_t *p, size_t len, bool want_code, size_t max_code, PropValue &out) {
    Cbor c{p, p + len};
    uint64_t n;
    if (!c.map(n)) return false;
    bool have_id = false;
    for (uint64_t i = 0; i < n && c.ok; i++) {
        std::string key;
        if (!c.text(key)) return false;
        if (key == "id") {
            uint64_t v;
            if (!c.uint(v)) return false;
            out.id = (u32)v;
            have_id = true;
        } else if (key == "value" && want_code) {
            uint64_t one;
            std::string variant;
            if (!c.map(one) || one != 1 || !c.text(variant)) return false;
            out.storage = variant == "UnsignedByte" ? COS_STORAGE_U8 : variant == "SubByte" ? COS_STORAGE_SUBBYTE
                          : variant == "HalfPrecisionFP" ? COS_STORAGE_F16 : variant == "FullPrecisionFP" ? COS_STORAGE_F32 : -1;
            if (out.storage < 0) return false;
            uint64_t nf;
            if (!c.map(nf)) return false;
            for (uint64_t f = 0; f < nf && c.ok; f++) {
                std::string fk;
                if (!c.text(fk)) return false;
                if (fk == "mag") { double d; if (!c.number(d)) return false; out.mag = (float)d; }
                else if (fk == "resolution") { uint64_t r; if (!c.uint(r)) return false; out.resolution = (u32)r; }
                else if (fk == "quant_vec" || fk == "vec") {
                    uint64_t m;
                    if (!c.array(m)) return false;       // m <= bytes left in the record (Cbor::fits)
                    if (m * (out.storage == COS_STORAGE_F32 ? 4 : out.storage == COS_STORAGE_F16 ? 2 : 1) > max_code) return false;
                    if (out.storage == COS_STORAGE_U8) {
                        out.code.resize(m);
                        for (uint64_t j = 0; j < m; j++) { uint64_t b; if (!c.uint(b)) return false; out.code[j] = (uint8_t)b; }
                    } else if (out.storage == COS_STORAGE_SUBBYTE) { // planes, plane-major
                        for (uint64_t pl = 0; pl < m; pl++) {
                            uint64_t pb;
                            if (!c.array(pb) || out.code.size() + pb > max_code) return false;
                            for (uint64_t j = 0; j < pb; j++) { uint64_t b; if (!c.uint(b)) return false; out.code.push_back((uint8_t)b); }
                        }
                    } else if (out.storage == COS_STORAGE_F16) { // half::f16 serialises as its u16 bits
                        out.code.resize(m * 2);
                        for (uint64_t j = 0; j < m; j++) { uint64_t b; if (!c.uint(b)) return false; const uint16_t h = (uint16_t)b; memcpy(&out.code[j * 2], &h, 2); }
                    } else {
                        out.code.resize(m * 4);
                        for (uint64_t j = 0; j < m; j++) { double d; if (!c.number(d)) return false; const float f = (float)d; memcpy(&out.code[j * 4], &f, 4); }
                    }
                } else if (!c.skip()) return false;
            }
        } else if (!c.skip()) return false;
    }
    return c.ok && have_id;
}

struct NodeRec {
    u32 id, ptr;
    std::vector<u32> nbr; // ids in slot order, COS_SLOT_EMPTY for null
};

struct ParsedDir {
    std::vector<std::vector<NodeRec>> levels; // per level, ascending id, root last
    PropValue root;                            // the root's stored Storage (when root_ptr_offset was given)
    bool have_root = false;
    std::vector<PropValue> level0_props;       // with want_codes: the stored Storage of every level-0 node (by position in levels[0])
};

// Parses the directory.  M_of(level) = neighbour slots per node; n_vectors = 0 skips the id range check.
static int32_t parse_dir_impl(const char *dir, u32 Ltop, u32 M, u32 M0, u32 n_vectors, bool have_root_ptr, u32 root_ptr_offset, bool want_codes, size_t max_code, ParsedDir &out) {
    const std::string root(dir);
    std::vector<uint8_t> ptrs, props;
    if (!read_file(root + "/nodes.ptr", ptrs)) return cos_fail(COS_ERR_INVALID, "cannot read %s/nodes.ptr", dir);
    if (!read_file(root + "/prop.data", props)) return cos_fail(COS_ERR_INVALID, "cannot read %s/prop.data", dir);
    std::vector<std::vector<uint8_t>> files;
    for (u32 k = 0;; k++) {
        std::vector<uint8_t> f;
        if (!read_file(root + "/" + std::to_string(k) + ".index", f)) break;
        files.push_back(std::move(f));
    }
    if (files.empty()) return cos_fail(COS_ERR_INVALID, "no 0.index under %s", dir);
    if (ptrs.size() % 8) return cos_fail(COS_ERR_INVALID, "nodes.ptr is not a whole number of 8-byte entries");
    const size_t n_ent = ptrs.size() / 8;
    if (have_root_ptr && ((size_t)root_ptr_offset / 8 >= n_ent || root_ptr_offset % 8))
        return cos_fail(COS_ERR_INVALID, "root ptr offset %u outside nodes.ptr", root_ptr_offset);
    out.levels.assign(Ltop + 1, {});
    std::vector<PropValue> l0props;
    for (size_t e = 0; e < n_ent; e++) {
        const u32 off = rd32(&ptrs[e * 8]), fid = rd32(&ptrs[e * 8 + 4]);
        if (off == 0xFFFFFFFFu || fid == 0xFFFFFFFFu) continue; // never written
        if (fid >= files.size()) return cos_fail(COS_ERR_INVALID, "nodes.ptr entry %zu names index file %u, which does not exist", e, fid);
        const std::vector<uint8_t> &F = files[fid];
        if ((size_t)off + 31 > F.size()) return cos_fail(COS_ERR_INVALID, "node record of entry %zu lies outside %u.index", e, fid);
        const uint8_t *r = &F[off];
        const u32 level = r[0];
        if (level > Ltop) return cos_fail(COS_ERR_INVALID, "entry %zu: level %u above num_layers %u", e, level, Ltop);
        const u32 prop_off = rd32(r + 5), prop_len = rd32(r + 9), meta_off = rd32(r + 13);
        if (meta_off != 0xFFFFFFFFu)
            return cos_fail(COS_ERR_UNIMPLEMENTED, "entry %zu carries prop_metadata: collections with a metadata schema (replica / pseudo nodes) are not loaded by this reader", e);
        const u32 nb = rd16(r + 29), Ml = level == 0 ? M0 : M;
        if (nb != Ml) return cos_fail(COS_ERR_INVALID, "entry %zu: %u neighbour slots on level %u, the index was created with %u", e, nb, level, Ml);
        if ((size_t)off + 31 + 13ull * nb > F.size()) return cos_fail(COS_ERR_INVALID, "neighbour array of entry %zu lies outside %u.index", e, fid);
        if ((size_t)prop_off + prop_len > props.size()) return cos_fail(COS_ERR_INVALID, "prop record of entry %zu lies outside prop.data", e);
        PropValue pv;
        const bool is_root_entry = have_root_ptr && (u32)(e * 8) == root_ptr_offset;
        const bool code_wanted = is_root_entry || (want_codes && level == 0);
        if (!parse_prop(&props[prop_off], prop_len, code_wanted, max_code, pv)) return cos_fail(COS_ERR_INVALID, "entry %zu: malformed CBOR prop record", e);
        if (n_vectors && pv.id != COS_ROOT_ID && pv.id >= n_vectors) return cos_fail(COS_ERR_INVALID, "entry %zu: internal id %u but only %u vectors are resident", e, pv.id, n_vectors);
        if (is_root_entry) {
            if (pv.id != COS_ROOT_ID) return cos_fail(COS_ERR_INVALID, "root ptr offset %u names node %u, not the root (u32::MAX)", root_ptr_offset, pv.id);
            if (level != Ltop) return cos_fail(COS_ERR_INVALID, "the root entry is on level %u, expected the top level %u", level, Ltop);
            out.root = pv;
            out.have_root = true;
        }
        NodeRec rec;
        rec.id = pv.id;
        rec.ptr = (u32)(e * 8);
        rec.nbr.resize(Ml);
        for (u32 j = 0; j < Ml; j++) {
            const uint8_t *q = r + 31 + 13ull * j;
            const u32 nid = rd32(q), nptr = rd32(q + 4);
            rec.nbr[j] = nptr == 0xFFFFFFFFu ? COS_SLOT_EMPTY : nid; // null slot = 13 x 0xFF (its id field reads u32::MAX = the ROOT id!)
        }
        if (want_codes && level == 0) { rec.ptr = (u32)l0props.size(); l0props.push_back(std::move(pv)); } // ptr reused as the prop index
        out.levels[level].push_back(std::move(rec));
    }
    if (have_root_ptr && !out.have_root) return cos_fail(COS_ERR_INVALID, "root entry not found");
    for (u32 l = 0; l <= Ltop; l++) {
        std::vector<NodeRec> &L = out.levels[l];
        std::sort(L.begin(), L.end(), [](const NodeRec &a, const NodeRec &b) { return a.id < b.id; });
        for (size_t i = 1; i < L.size(); i++)
            if (L[i].id == L[i - 1].id) return cos_fail(COS_ERR_INVALID, "level %u holds node %u twice", l, L[i].id);
        if (L.empty() || L.back().id != COS_ROOT_ID) return cos_fail(COS_ERR_INVALID, "level %u has no root node", l);
    }
    if (want_codes) {
        out.level0_props.resize(out.levels[0].size());
        for (size_t i = 0; i < out.levels[0].size(); i++) out.level0_props[i] = std::move(l0props[out.levels[0][i].ptr]);
    }
    return COS_OK;
}

// nothing may unwind through the extern "C" entry points (a Rust host aborts, or worse): allocation failures and anything else
// the standard containers throw on a damaged directory become COS_ERR_INVALID
int32_t parse_dir(const char *dir, u32 Ltop, u32 M, u32 M0, u32 n_vectors, bool have_root_ptr, u32 root_ptr_offset, bool want_codes, size_t max_code, ParsedDir &out) {
    try {
        return parse_dir_impl(dir, Ltop, M, M0, n_vectors, have_root_ptr, root_ptr_offset, want_codes, max_code, out);
    } catch (const std::exception &e) {
        return cos_fail(COS_ERR_INVALID, "reading %s failed: %s", dir, e.what());
    } catch (...) {
        return cos_fail(COS_ERR_INVALID, "reading %s failed", dir);
    }
}

} // namespace

// ---- host-only inspection (no device needed): what a maintainer's exporter / a test uses to look at a directory -------------
static int32_t cos_reference_dir_level_counts_impl(const char *dir, uint32_t num_layers, uint32_t neighbors_count, uint32_t level0_neighbors_count,
                                                  uint32_t *level_counts) {
    if (!dir || !level_counts || num_layers + 1 > (u32)MAX_LEVELS) return cos_fail(COS_ERR_INVALID, "bad argument");
    ParsedDir pd;
    int32_t rc = parse_dir(dir, num_layers, neighbors_count, level0_neighbors_count, 0, false, 0, false, 0, pd);
    if (rc) return rc;
    for (u32 l = 0; l <= num_layers; l++) level_counts[l] = (u32)pd.levels[l].size();
    return COS_OK;
}

static int32_t cos_reference_dir_read_level_impl(const char *dir, uint32_t num_layers, uint32_t neighbors_count, uint32_t level0_neighbors_count,
                                                uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids) {
    if (!dir || !node_ids || !nbr_ids || num_layers + 1 > (u32)MAX_LEVELS || level > num_layers) return cos_fail(COS_ERR_INVALID, "bad argument");
    ParsedDir pd;
    int32_t rc = parse_dir(dir, num_layers, neighbors_count, level0_neighbors_count, 0, false, 0, false, 0, pd);
    if (rc) return rc;
    const u32 Ml = level == 0 ? level0_neighbors_count : neighbors_count;
    const std::vector<NodeRec> &L = pd.levels[level];
    for (size_t i = 0; i < L.size(); i++) {
        node_ids[i] = L[i].id;
        memcpy(&nbr_ids[i * Ml], L[i].nbr.data(), (size_t)Ml * 4);
    }
    return COS_OK;
}

static int32_t cos_index_load_reference_dir_impl(cos_index *ix, const char *dir, uint32_t root_ptr_offset, uint32_t flags) {
    if (!ix || !dir) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload the raw vectors (cos_index_upload_vectors) before loading the graph");
    const u32 Ltop = ix->p.num_layers, n = ix->n;
    const bool verify = flags & COS_LOAD_VERIFY_CODES;
    const size_t cb = cos_code_bytes(ix->p.storage, ix->p.resolution, ix->p.dim);
    ParsedDir pd;
    int32_t rc = parse_dir(dir, Ltop, ix->p.neighbors_count, ix->p.level0_neighbors_count, n, true, root_ptr_offset, verify, (size_t)ix->p.dim * 4, pd);
    if (rc) return rc;
    auto storage_ok = [&](const PropValue &pv) {
        return pv.storage == (int)ix->p.storage && pv.code.size() == cb && (pv.storage != COS_STORAGE_SUBBYTE || pv.resolution == ix->p.resolution);
    };
    if (!storage_ok(pd.root)) return cos_fail(COS_ERR_STORAGE_MISMATCH, "the root's stored Storage does not match the index's storage type / dimension");
    if (pd.levels[0].size() != (size_t)n + 1) return cos_fail(COS_ERR_INVALID, "level 0 holds %zu nodes, %u vectors (+ root) are resident", pd.levels[0].size(), n);
    if (verify) {
        std::vector<uint8_t> dev_codes(((size_t)n + 1) * cb);
        std::vector<float> dev_mags((size_t)n + 1);
        rc = cos_index_download_codes(ix, dev_codes.data(), dev_mags.data());
        if (rc) return rc;
        for (size_t i = 0; i < pd.levels[0].size(); i++) {
            const PropValue &pv = pd.level0_props[i];
            if (pv.id == COS_ROOT_ID) continue;
            if (!storage_ok(pv)) return cos_fail(COS_ERR_STORAGE_MISMATCH, "vector %u: stored Storage does not match the index's storage type / dimension", pv.id);
            if (memcmp(&dev_codes[(size_t)pv.id * cb], pv.code.data(), cb) != 0 || memcmp(&dev_mags[pv.id], &pv.mag, 4) != 0)
                return cos_fail(COS_ERR_STORAGE_MISMATCH, "vector %u: the code stored by the reference differs from the device's quantization of the uploaded raw vector", pv.id);
        }
    }
    rc = cos_set_root_code(ix, pd.root.code.data(), pd.root.mag);
    if (rc) return rc;
    for (u32 l = 0; l <= Ltop; l++) {
        const std::vector<NodeRec> &L = pd.levels[l];
        const u32 Ml = ix->lv[l].M;
        std::vector<u32> ids(L.size()), nbr(L.size() * (size_t)Ml);
        for (size_t i = 0; i < L.size(); i++) {
            ids[i] = L[i].id;
            memcpy(&nbr[i * Ml], L[i].nbr.data(), (size_t)Ml * 4);
        }
        rc = cos_push_level_ids(ix, l, std::move(ids), std::move(nbr));
        if (rc) return rc;
    }
    return COS_OK;
}

extern "C" int32_t cos_reference_dir_level_counts(const char *dir, uint32_t num_layers, uint32_t neighbors_count, uint32_t level0_neighbors_count, uint32_t *level_counts) {
    try {
        return cos_reference_dir_level_counts_impl(dir, num_layers, neighbors_count, level0_neighbors_count, level_counts);
    } catch (const std::exception &e) {
        return cos_fail(COS_ERR_INVALID, "cos_reference_dir_level_counts: %s", e.what());
    } catch (...) {
        return cos_fail(COS_ERR_INVALID, "cos_reference_dir_level_counts failed");
    }
}

extern "C" int32_t cos_reference_dir_read_level(const char *dir, uint32_t num_layers, uint32_t neighbors_count, uint32_t level0_neighbors_count, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids) {
    try {
        return cos_reference_dir_read_level_impl(dir, num_layers, neighbors_count, level0_neighbors_count, level, node_ids, nbr_ids);
    } catch (const std::exception &e) {
        return cos_fail(COS_ERR_INVALID, "cos_reference_dir_read_level: %s", e.what());
    } catch (...) {
        return cos_fail(COS_ERR_INVALID, "cos_reference_dir_read_level failed");
    }
}

extern "C" int32_t cos_index_load_reference_dir(cos_index *ix, const char *dir, uint32_t root_ptr_offset, uint32_t flags) {
    try {
        return cos_index_load_reference_dir_impl(ix, dir, root_ptr_offset, flags);
    } catch (const std::exception &e) {
        return cos_fail(COS_ERR_INVALID, "cos_index_load_reference_dir: %s", e.what());
    } catch (...) {
        return cos_fail(COS_ERR_INVALID, "cos_index_load_reference_dir failed");
    }
}
