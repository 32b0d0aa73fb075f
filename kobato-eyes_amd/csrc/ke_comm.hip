// ke_comm.hip -- the exchange steps of the multi-GPU path on RCCL (xGMI), behind the C ABI.
//
// One process per GPU.  Images are hash-partitioned (image i on rank i mod world); after local hashing ONE
// ncclAllGather shares the 64-bit hash shards (100 KB per rank at 100 000 images, 1 MB at 1 000 000 --
// latency-bound, so a single collective and no pipelining), a kernel puts the table back into corpus order, every rank
// scans its share of the tile triangle, and the per-rank edge lists (O(#near-duplicates)) are merged with one more
// all-gather of fixed-width records [count | first K edges] -- K follows the largest list seen, a second gather only
// when a list outgrows it -- followed by ONE device-to-host copy.  No all-reduce anywhere.
//
// RCCL is bound at first use (dlopen of the librccl that sits beside the process' HIP runtime): a process that already
// carries one -- PyTorch does -- is joined rather than given a second copy, and single-GPU users never load it.  A communicator made elsewhere
// (ncclComm_t) can be passed in as void*; ke_comm_create makes one from a 128-byte unique id the host distributes.
#include <dlfcn.h>

#include <algorithm>

#include "ke_internal.h"

namespace {

typedef int ncclResult;                      // ncclResult_t: 0 = ncclSuccess
typedef struct ncclComm *ncclCommPtr;        // ncclComm_t
struct KeNcclId { char internal[128]; };     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
enum { KE_NCCL_UINT8 = 1, KE_NCCL_UINT64 = 5 };   // ncclUint8, ncclUint64 (rccl.h: ncclDataType_t)

struct Rccl {
    void *handle = nullptr;
    ncclResult (*GetUniqueId)(KeNcclId *) = nullptr;
    ncclResult (*CommInitRank)(ncclCommPtr *, int, KeNcclId, int) = nullptr;
    ncclResult (*CommDestroy)(ncclCommPtr) = nullptr;
    ncclResult (*AllGather)(const void *, void *, size_t, int, ncclCommPtr, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    // The RCCL that belongs to the HIP runtime this process already runs on: a Python process may carry PyTorch's bundled
    // ROCm next to /opt/rocm, and an RCCL bound to the other copy finds no initialised device.  So look beside the loaded
    // libamdhip64 first, then wherever the loader finds one.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            names.push_back(dir + "librccl.so.1");
            names.push_back(dir + "librccl.so");
        }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string &name : names) {
        r.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        const char *e = dlerror();
        r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
        return &r;
    }
    auto sym = [&](const char *n) -> void * {
        void *p = dlsym(r.handle, n);
        if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return &r;
}

int need_rccl(ke_ctx *ctx, Rccl **out) {
    Rccl *r = rccl();
    if (!r->error.empty()) return ke_fail(ctx, KE_EUNSUPPORTED, "%s", r->error.c_str());
    *out = r;
    return KE_OK;
}

#define KE_NCCL(ctx, r, call)                                                                     \
    do {                                                                                          \
        ncclResult e_ = (call);                                                                   \
        if (e_ != 0)                                                                              \
            return ke_fail((ctx), KE_EHIP, "%s failed: %s", #call, (r)->GetErrorString ? (r)->GetErrorString(e_) : "?"); \
    } while (0)

// gathered[r * per + k] is corpus item r + k * world  ->  table[r + k * world]
__global__ void ke_interleave_u64(const uint64_t *__restrict__ gathered, int64_t per, int world, int64_t n_total,
                                  uint64_t *__restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    table[i] = gathered[(i % world) * per + i / world];
}

// record = [count : int64 | slots x ke_edge]; the count comes in as a kernel argument (no H2D copy on the step's path)
__global__ void ke_pack_edge_record(const ke_edge *__restrict__ edges, int64_t count, int64_t slots, uint8_t *__restrict__ record) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *reinterpret_cast<int64_t *>(record) = count;
    const int64_t m = count < slots ? count : slots;
    if (i < m * 3) {                                          // 24-byte records as three 8-byte words
        reinterpret_cast<uint64_t *>(record + 8)[i] = reinterpret_cast<const uint64_t *>(edges)[i];
    }
}

}  // namespace

KE_API int ke_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return KE_EINVAL;
    Rccl *r;
    KE_TRY(need_rccl(nullptr, &r));
    KeNcclId id;
    std::memset(&id, 0, sizeof id);
    if (r->GetUniqueId(&id) != 0) return ke_fail(nullptr, KE_EHIP, "ncclGetUniqueId failed");
    std::memcpy(id_out, id.internal, 128);
    return KE_OK;
}

KE_API int ke_comm_create(ke_ctx *ctx, const uint8_t *unique_id, int32_t world, int32_t rank, void **comm_out) {
    if (!ctx || !unique_id || !comm_out) return KE_EINVAL;
    if (world < 1 || rank < 0 || rank >= world) return ke_fail(ctx, KE_EINVAL, "bad rank %d of %d", rank, world);
    Rccl *r;
    KE_TRY(need_rccl(ctx, &r));
    KE_HIP(ctx, hipSetDevice(ctx->device));
    KeNcclId id;
    std::memcpy(id.internal, unique_id, 128);
    ncclCommPtr comm = nullptr;
    KE_NCCL(ctx, r, r->CommInitRank(&comm, world, id, rank));
    *comm_out = comm;
    return KE_OK;
}

KE_API int ke_comm_destroy(ke_ctx *ctx, void *comm) {
    if (!ctx) return KE_EINVAL;
    if (!comm) return KE_OK;
    Rccl *r;
    KE_TRY(need_rccl(ctx, &r));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KE_NCCL(ctx, r, r->CommDestroy((ncclCommPtr)comm));
    return KE_OK;
}

KE_API int ke_allgather_u64(ke_ctx *ctx, void *comm, int32_t world, const uint64_t *local, int64_t n_local, uint64_t *gathered) {
    if (!ctx || !comm) return KE_EINVAL;
    if (n_local < 0 || world < 1 || (n_local > 0 && (!local || !gathered))) return ke_fail(ctx, KE_EINVAL, "bad all-gather arguments");
    if (n_local == 0) return KE_OK;
    if (!ke_is_device_ptr(local) || !ke_is_device_ptr(gathered)) return ke_fail(ctx, KE_EINVAL, "ke_allgather_u64 works on device memory");
    Rccl *r;
    KE_TRY(need_rccl(ctx, &r));
    KE_HIP(ctx, hipSetDevice(ctx->device));
    KE_NCCL(ctx, r, r->AllGather(local, gathered, (size_t)n_local, KE_NCCL_UINT64, (ncclCommPtr)comm, ctx->stream));
    return KE_OK;
}

KE_API int ke_interleave_shards(ke_ctx *ctx, const uint64_t *gathered, int32_t world, int64_t n_total, uint64_t *table) {
    if (!ctx) return KE_EINVAL;
    if (world < 1 || n_total < 0 || (n_total > 0 && (!gathered || !table))) return ke_fail(ctx, KE_EINVAL, "bad interleave arguments");
    if (n_total == 0) return KE_OK;
    if (!ke_is_device_ptr(gathered) || !ke_is_device_ptr(table) || gathered == table)
        return ke_fail(ctx, KE_EINVAL, "ke_interleave_shards works on two distinct device arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t per = (n_total + world - 1) / world;
    hipLaunchKernelGGL(ke_interleave_u64, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, ctx->stream, gathered, per, world,
                       n_total, table);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

KE_API int ke_allgather_hashes(ke_ctx *ctx, void *comm, int32_t world, const uint64_t *local, int64_t n_total, uint64_t *table) {
    if (!ctx || !comm) return KE_EINVAL;
    if (world < 1 || n_total < 0 || (n_total > 0 && (!local || !table))) return ke_fail(ctx, KE_EINVAL, "bad all-gather arguments");
    if (n_total == 0) return KE_OK;
    if (!ke_is_device_ptr(local) || !ke_is_device_ptr(table)) return ke_fail(ctx, KE_EINVAL, "ke_allgather_hashes works on device memory");
    const int64_t per = (n_total + world - 1) / world;        // every rank sends `per` entries (the last ones may be padding)
    void *g;
    KE_TRY(ke_reserve(ctx, KE_BUF_COMM, (size_t)world * per * 8, &g));
    KE_TRY(ke_allgather_u64(ctx, comm, world, local, per, (uint64_t *)g));
    return ke_interleave_shards(ctx, (const uint64_t *)g, world, n_total, table);
}

KE_API int ke_allgather_edges(ke_ctx *ctx, void *comm, int32_t world, const ke_edge *local, int64_t n_local, ke_edge *merged_out,
                              int64_t capacity, int64_t *n_total_out, int64_t *counts_out) {
    if (!ctx || !comm || !n_total_out) return KE_EINVAL;
    if (world < 1 || n_local < 0 || capacity < 0 || (n_local > 0 && !local) || (capacity > 0 && !merged_out))
        return ke_fail(ctx, KE_EINVAL, "bad edge-gather arguments");
    if (n_local > 0 && !ke_is_device_ptr(local)) return ke_fail(ctx, KE_EINVAL, "local edges must be device memory (ke_hamming_scan's edges_out)");
    if (merged_out && ke_is_device_ptr(merged_out)) return ke_fail(ctx, KE_EINVAL, "merged edges are returned to a host array");
    Rccl *r;
    KE_TRY(need_rccl(ctx, &r));
    KE_HIP(ctx, hipSetDevice(ctx->device));
    *n_total_out = 0;
    std::vector<int64_t> counts((size_t)world, 0);
    for (int round = 0; round < 2; ++round) {
        const int64_t slots = ctx->edge_slots;
        const size_t width = 8 + (size_t)slots * sizeof(ke_edge);
        void *dev;
        KE_TRY(ke_reserve(ctx, KE_BUF_COMM_EDGES, (size_t)(world + 1) * width, &dev));
        uint8_t *send = (uint8_t *)dev, *recv = send + width;
        if (ctx->h_comm_bytes < (size_t)world * width) {       // pinned landing zone, grown with the record
            if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
            ctx->h_comm = nullptr;
            ctx->h_comm_bytes = 0;
            KE_HIP(ctx, hipHostMalloc(&ctx->h_comm, (size_t)world * width, hipHostMallocDefault));
            ctx->h_comm_bytes = (size_t)world * width;
        }
        const int64_t words = std::min(n_local, slots) * 3;
        hipLaunchKernelGGL(ke_pack_edge_record, dim3((unsigned)(std::max<int64_t>(words, 1) + 255) / 256), dim3(256), 0, ctx->stream, local,
                           n_local, slots, send);
        KE_HIP(ctx, hipGetLastError());
        KE_NCCL(ctx, r, r->AllGather(send, recv, width, KE_NCCL_UINT8, (ncclCommPtr)comm, ctx->stream));
        KE_HIP(ctx, hipMemcpyAsync(ctx->h_comm, recv, (size_t)world * width, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int64_t top = 0, total = 0;
        for (int k = 0; k < world; ++k) {
            std::memcpy(&counts[k], (const uint8_t *)ctx->h_comm + (size_t)k * width, 8);
            top = std::max(top, counts[k]);
            total += counts[k];
        }
        // the record follows the largest list seen (x1.25, power of two) so that the next step needs one collective
        int64_t want = 1024;
        while (want < top + top / 4) want *= 2;
        ctx->edge_slots = std::max(ctx->edge_slots, want);
        if (top > slots) continue;                            // some list did not fit this round's record: once more, wider
        *n_total_out = total;
        if (counts_out) std::memcpy(counts_out, counts.data(), (size_t)world * 8);
        int64_t at = 0;
        for (int k = 0; k < world; ++k) {
            const int64_t m = std::min(counts[k], std::max<int64_t>(capacity - at, 0));
            if (m > 0) std::memcpy(merged_out + at, (const uint8_t *)ctx->h_comm + (size_t)k * width + 8, (size_t)m * sizeof(ke_edge));
            at += counts[k];
        }
        return KE_OK;
    }
    return ke_fail(ctx, KE_EHIP, "edge gather did not converge");   // cannot happen: the second round's record holds the maximum
}
