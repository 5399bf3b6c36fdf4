// shardset.hip — sharded serving behind the C ABI (BASELINE north_star: "the index shards by vector-ID range
// across the 8 GPUs of one node with a final RCCL all-gather of per-shard top-k over xGMI").
//
// The caller of the reference's search path is ONE process: IndexOps::batch_search (indexes/mod.rs:260-272).  A cos_shardset
// gives that process one call that fans a query batch out to S resident shards (cos_index handles, id_base = first id of the
// shard), runs walk + exact rerank on every shard's GPU, exchanges the packed per-shard records [ids | scores | counts] with
// ONE all-gather and merges them (merge_topk_kernel): cos_shardset_search_batch.  Two deployments share the code:
//   * single process, S devices   (the Rust host):  ranks 0..S-1 are all local; communicators from ncclCommInitAll;
//     the per-launch collective is S grouped ncclAllGather calls, one per device stream;
//   * one process per GPU         (bench.py under torchrun): every process owns one local shard and joins a world-size
//     communicator built from a ncclUniqueId the host side distributes (cos_shardset_unique_id -> broadcast -> create);
//     cos_shardset_exchange_device is the per-launch exchange, enqueued on the caller's stream.
// Shards that share a device (S = 2 on one GPU: the single-device parity test) cannot form an RCCL communicator; their
// records are gathered with device-to-device copies instead — the same data movement, no other difference.
//
// RCCL is resolved at run time (dlopen of librccl.so.1) the first time a communicator is needed, so libcosdata_hip.so itself
// has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi *rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("dlopen(librccl.so.1): ") + dlerror(); return; }
#define SYM(field, name)                                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));            \
    if (!api.field && api.error.empty()) api.error = std::string("librccl: missing symbol ") + name
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllGather, "ncclAllGather");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    });
    return &api;
}

#define RCCL_TRY(expr)                                                                                                        \
    do {                                                                                                                      \
        ncclResult_t _r = (expr);                                                                                             \
        if (_r != ncclSuccess) return cos_fail(COS_ERR_HIP, "%s: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

struct LocalShard {
    cos_index *ix = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    ncclComm_t comm = nullptr;
    // per-shard device buffers, grown on demand
    float *d_queries = nullptr;
    u32 *d_packed = nullptr;   // this shard's record [ids B*k | scores B*k | counts B]
    u32 *d_gathered = nullptr; // [world][words]
    int32_t *d_status = nullptr;
    size_t cap_q = 0, cap_words = 0, cap_gath = 0, cap_B = 0;
};

} // namespace

struct cos_shardset {
    std::vector<LocalShard> sh;
    u32 first_rank = 0, world = 0;
    bool use_rccl = false;
    std::mutex mu; // one search at a time per shard set (the collectives of two batches must not interleave)
    // merge output on shard 0's device
    u32 *d_out_ids = nullptr, *d_out_counts = nullptr;
    float *d_out_scores = nullptr;
    size_t cap_out = 0, cap_outB = 0;
};

extern "C" int32_t cos_shardset_unique_id(uint8_t *out) {
    if (!out) return cos_fail(COS_ERR_INVALID, "null argument");
    RcclApi *r = rccl();
    if (!r->error.empty()) return cos_fail(COS_ERR_HIP, "RCCL unavailable: %s", r->error.c_str());
    ncclUniqueId id;
    RCCL_TRY(r->GetUniqueId(&id));
    static_assert(sizeof(id) == COS_SHARDSET_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(out, &id, sizeof(id));
    return COS_OK;
}

extern "C" int32_t cos_shardset_destroy(cos_shardset *ss) {
    if (!ss) return COS_OK;
    for (LocalShard &s : ss->sh) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.comm) (void)rccl()->CommDestroy(s.comm);
        void *ptrs[] = {s.d_queries, s.d_packed, s.d_gathered, s.d_status};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (!ss->sh.empty()) (void)hipSetDevice(ss->sh[0].device);
    void *ptrs[] = {ss->d_out_ids, ss->d_out_counts, ss->d_out_scores};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete ss;
    return COS_OK;
}

extern "C" int32_t cos_shardset_create(cos_index *const *shards, uint32_t n_local, uint32_t first_rank, uint32_t world_size, const uint8_t *unique_id,
                                       cos_shardset **out) {
    if (!shards || !out || n_local == 0) return cos_fail(COS_ERR_INVALID, "null/empty shard list");
    *out = nullptr;
    if (world_size == 0) world_size = n_local;
    if ((uint64_t)first_rank + n_local > world_size) return cos_fail(COS_ERR_INVALID, "ranks [%u, %u) exceed world size %u", first_rank, first_rank + n_local, world_size);
    if (n_local != world_size && n_local != 1) return cos_fail(COS_ERR_UNIMPLEMENTED, "a process holds either every shard or exactly one");
    if (n_local != world_size && !unique_id) return cos_fail(COS_ERR_INVALID, "a multi-process shard set needs the ncclUniqueId of cos_shardset_unique_id");
    for (u32 s = 0; s < n_local; s++) {
        if (!shards[s]) return cos_fail(COS_ERR_INVALID, "shard %u is null", s);
        if (shards[s]->p.dim != shards[0]->p.dim) return cos_fail(COS_ERR_INVALID, "shards disagree on dim");
    }
    cos_shardset *ss = new cos_shardset();
    ss->first_rank = first_rank;
    ss->world = world_size;
    ss->sh.resize(n_local);
    bool distinct = true;
    for (u32 s = 0; s < n_local; s++) {
        ss->sh[s].ix = shards[s];
        ss->sh[s].device = shards[s]->p.device;
        for (u32 t = 0; t < s; t++) distinct &= ss->sh[t].device != ss->sh[s].device;
    }
    auto fail = [&](int32_t rc) { cos_shardset_destroy(ss); return rc; };
    for (LocalShard &s : ss->sh) {
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
            return fail(cos_fail(COS_ERR_HIP, "cannot create stream/event on device %d", s.device));
    }
    // RCCL communicators: needed whenever records cross a device boundary
    // (COS_SHARDSET_FORCE_RCCL=1: also for a world of one, so the RCCL call path can be exercised on a single-GPU box)
    const bool force = cosdev::tune_or(cosdev::TUNE_SHARDSET_FORCE_RCCL, 0) != 0 && world_size == 1;
    ss->use_rccl = (world_size > 1 && (n_local == 1 || distinct)) || force;
    if (ss->use_rccl) {
        RcclApi *r = rccl();
        if (!r->error.empty()) return fail(cos_fail(COS_ERR_HIP, "RCCL unavailable: %s", r->error.c_str()));
        if (n_local == world_size) {
            std::vector<int> devs(n_local);
            std::vector<ncclComm_t> comms(n_local);
            for (u32 s = 0; s < n_local; s++) devs[s] = ss->sh[s].device;
            ncclResult_t rr = r->CommInitAll(comms.data(), (int)n_local, devs.data());
            if (rr != ncclSuccess) return fail(cos_fail(COS_ERR_HIP, "ncclCommInitAll: %s", r->GetErrorString(rr)));
            for (u32 s = 0; s < n_local; s++) ss->sh[s].comm = comms[s];
        } else {
            ncclUniqueId id;
            memcpy(&id, unique_id, sizeof(id));
            if (hipSetDevice(ss->sh[0].device) != hipSuccess) return fail(cos_fail(COS_ERR_HIP, "hipSetDevice"));
            ncclResult_t rr = r->CommInitRank(&ss->sh[0].comm, (int)world_size, id, (int)first_rank);
            if (rr != ncclSuccess) return fail(cos_fail(COS_ERR_HIP, "ncclCommInitRank: %s", r->GetErrorString(rr)));
        }
    }
    *out = ss;
    return COS_OK;
}

// gathered[world][words] on every local shard's device <- every shard's packed record; enqueued on the shards' streams
static int32_t exchange_local(cos_shardset *ss, size_t words) {
    const u32 S = (u32)ss->sh.size();
    if (ss->use_rccl) {
        RcclApi *r = rccl();
        RCCL_TRY(r->GroupStart());
        for (LocalShard &s : ss->sh) {
            ncclResult_t rr = r->AllGather(s.d_packed, s.d_gathered, words, ncclUint32, s.comm, s.stream);
            if (rr != ncclSuccess) { (void)r->GroupEnd(); return cos_fail(COS_ERR_HIP, "ncclAllGather: %s", r->GetErrorString(rr)); }
        }
        RCCL_TRY(r->GroupEnd());
        return COS_OK;
    }
    // shards sharing a device (or a single shard): plain device copies into shard 0's gather buffer, ordered by events
    LocalShard &root = ss->sh[0];
    HIP_TRY(hipSetDevice(root.device));
    for (u32 s = 0; s < S; s++) {
        LocalShard &src = ss->sh[s];
        if (s) HIP_TRY(hipStreamWaitEvent(root.stream, src.done, 0));
        if (src.device == root.device)
            HIP_TRY(hipMemcpyAsync(root.d_gathered + (size_t)s * words, src.d_packed, words * 4, hipMemcpyDeviceToDevice, root.stream));
        else // a mixed set (e.g. 3 shards on 2 GPUs): explicit peer copy, valid with or without peer access between the two devices
            HIP_TRY(hipMemcpyPeerAsync(root.d_gathered + (size_t)s * words, root.device, src.d_packed, src.device, words * 4, root.stream));
    }
    return COS_OK;
}

template <typename T>
static hipError_t grow(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, need * sizeof(T));
    if (e == hipSuccess) cap = need;
    return e;
}

extern "C" int32_t cos_shardset_search_batch(cos_shardset *ss, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                             uint32_t *out_counts) {
    if (!ss || !queries || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (ss->sh.size() != ss->world) return cos_fail(COS_ERR_INVALID, "cos_shardset_search_batch needs every shard in this process; a process-per-GPU job uses cos_shardset_exchange_device");
    std::lock_guard<std::mutex> g(ss->mu);
    const u32 S = ss->world, dim = ss->sh[0].ix->p.dim;
    const size_t words = (size_t)B * (2 * (size_t)top_k + 1);
    struct DrainAll { // every return path — a later shard's error, a failed grow, a failed exchange — leaves no walk or collective
        cos_shardset *ss; // of this batch queued on any shard's stream: the next call (or destroy) starts from idle streams
        ~DrainAll() {
            for (LocalShard &s : ss->sh) { (void)hipSetDevice(s.device); (void)hipStreamSynchronize(s.stream); }
            (void)hipSetDevice(ss->sh[0].device);
        }
    } drain_all{ss};
    // 1. every shard: queries up, walk + exact rerank into its packed record (all shards run concurrently on their devices)
    for (LocalShard &s : ss->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(grow(s.d_queries, s.cap_q, (size_t)B * dim));
        HIP_TRY(grow(s.d_packed, s.cap_words, words));
        HIP_TRY(grow(s.d_gathered, s.cap_gath, words * S));
        HIP_TRY(grow(s.d_status, s.cap_B, (size_t)B));
        HIP_TRY(hipMemcpyAsync(s.d_queries, queries, (size_t)B * dim * 4, hipMemcpyHostToDevice, s.stream));
        int32_t rc = cos_search_batch_device(s.ix, s.d_queries, B, top_k, s.d_packed, reinterpret_cast<float *>(s.d_packed + (size_t)B * top_k),
                                             s.d_packed + 2 * (size_t)B * top_k, s.d_status, s.stream);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.done, s.stream));
    }
    // 2. ONE exchange step, 3. merge on shard 0's device
    int32_t rc = exchange_local(ss, words);
    if (rc) return rc;
    LocalShard &root = ss->sh[0];
    HIP_TRY(hipSetDevice(root.device));
    HIP_TRY(grow(ss->d_out_ids, ss->cap_out, (size_t)B * top_k));
    if (ss->cap_outB < (size_t)B * top_k) {
        if (ss->d_out_scores) (void)hipFree(ss->d_out_scores);
        if (ss->d_out_counts) (void)hipFree(ss->d_out_counts);
        ss->d_out_scores = nullptr; ss->d_out_counts = nullptr; ss->cap_outB = 0;
        HIP_TRY(hipMalloc((void **)&ss->d_out_scores, (size_t)B * top_k * 4));
        HIP_TRY(hipMalloc((void **)&ss->d_out_counts, (size_t)B * top_k * 4));
        ss->cap_outB = (size_t)B * top_k;
    }
    rc = cos_merge_topk_packed_device(root.d_gathered, S, B, top_k, ss->d_out_ids, ss->d_out_scores, ss->d_out_counts, root.device, root.stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_ids, ss->d_out_ids, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, root.stream));
    HIP_TRY(hipMemcpyAsync(out_scores, ss->d_out_scores, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, root.stream));
    HIP_TRY(hipMemcpyAsync(out_counts, ss->d_out_counts, (size_t)B * 4, hipMemcpyDeviceToHost, root.stream));
    // per-shard statuses: like the reference's collect::<Result<_>>, one failing query on any shard fails the call
    std::vector<int32_t> status((size_t)B);
    for (LocalShard &s : ss->sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipMemcpyAsync(status.data(), s.d_status, (size_t)B * 4, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
        for (u32 b = 0; b < B; b++)
            if (status[b] != COS_OK) {
                (void)hipSetDevice(root.device);
                (void)hipStreamSynchronize(root.stream);
                return cos_fail(status[b], "query %u failed on shard %u with status %d (zero-norm vector -> DistanceError::CalculationError)", b,
                                ss->first_rank + (u32)(&s - ss->sh.data()), status[b]);
            }
    }
    HIP_TRY(hipSetDevice(root.device));
    HIP_TRY(hipStreamSynchronize(root.stream));
    return COS_OK;
}

extern "C" int32_t cos_shardset_exchange_device(cos_shardset *ss, const uint32_t *d_packed_local, uint32_t B, uint32_t top_k, uint32_t *d_gathered,
                                                uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, void *stream) {
    if (!ss || !d_packed_local || !d_gathered || !d_out_ids || !d_out_scores || !d_out_counts || B == 0 || top_k == 0)
        return cos_fail(COS_ERR_INVALID, "bad argument");
    if (ss->sh.size() != 1) return cos_fail(COS_ERR_INVALID, "cos_shardset_exchange_device is the process-per-GPU entry point (one local shard)");
    LocalShard &s = ss->sh[0];
    HIP_TRY(hipSetDevice(s.device));
    const size_t words = (size_t)B * (2 * (size_t)top_k + 1);
    hipStream_t st = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> g(ss->mu); // collectives of one communicator are issued in one order on every rank
        if (ss->use_rccl) RCCL_TRY(rccl()->AllGather(d_packed_local, d_gathered, words, ncclUint32, s.comm, st));
        else HIP_TRY(hipMemcpyAsync(d_gathered, d_packed_local, words * 4, hipMemcpyDeviceToDevice, st)); // world of one
    }
    return cos_merge_topk_packed_device(d_gathered, ss->world, B, top_k, d_out_ids, d_out_scores, d_out_counts, s.device, st);
}
