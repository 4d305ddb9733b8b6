// Data-parallel plumbing of the hot path over RCCL / xGMI (include/ttsamd.h "ttsamd_dp_*").
//
// The reference is single-device (inference.py:23-24); utterances are independent, so the batch shards over the
// GPUs of one node with exactly two exchanges and no collective inside the model:
//   C1 once:     broadcast of the weights from the root (ncclBroadcast of a handle's device blobs, or of any
//                device buffer), one large transfer per blob — xGMI links are per peer, few big messages;
//   C2 per call: all-gather of the per-utterance lengths (a few hundred bytes), then a fan-in of the PACKED ragged
//                audio (valid samples only, no padding) to the root: grouped ncclSend / ncclRecv, i.e. 7 independent
//                point-to-point transfers over 7 xGMI links in parallel, no ring and no reduction.
// librccl is bound at run time (dlopen of the SONAME torch has already loaded, so a process never holds two
// copies); libttsamd.so itself does not depend on it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "kernels.hpp"

namespace ttsamd {

void hifigan_blobs(const void* h, void** f32, int64_t* n_f32, void** b16, int64_t* n_b16);
void fastpitch_blobs(const void* h, void** f32, int64_t* n_f32, void** b16, int64_t* n_b16);

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int32_t load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    for (const char* n : names) {
        so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (so) break;
    }
    TTS_REQUIRE(so, "dp: librccl.so.1 not found (%s)", dlerror());
    Rccl r;
    r.so = so;
#define TTS_SYM(field, name)                                                      \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(so, name));               \
    TTS_REQUIRE(r.field, "dp: librccl has no symbol %s", name)
    TTS_SYM(GetUniqueId, "ncclGetUniqueId");
    TTS_SYM(CommInitRank, "ncclCommInitRank");
    TTS_SYM(CommDestroy, "ncclCommDestroy");
    TTS_SYM(Broadcast, "ncclBroadcast");
    TTS_SYM(AllGather, "ncclAllGather");
    TTS_SYM(Send, "ncclSend");
    TTS_SYM(Recv, "ncclRecv");
    TTS_SYM(GroupStart, "ncclGroupStart");
    TTS_SYM(GroupEnd, "ncclGroupEnd");
    TTS_SYM(GetErrorString, "ncclGetErrorString");
#undef TTS_SYM
    g_rccl = r;
    return 0;
}

#define TTS_CHECK_NCCL(expr)                                                                                 \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) {                                                                             \
            set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__);        \
            return TTSAMD_EHIP;                                                                              \
        }                                                                                                    \
    } while (0)

struct DpComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

// out[off(b) + t] = wave[b][t], t < lens[b], off(b) = sum_{b' < b} lens[b']: the valid samples of a padded ragged
// batch back to back (what crosses xGMI and PCIe).  grid (ceil(n_max / 1024), B), 4 samples per thread.
__global__ __launch_bounds__(256) void pack_ragged_kernel(const float* __restrict__ wave, int64_t stride,
                                                         const int64_t* __restrict__ lens, int64_t n_max,
                                                         float* __restrict__ out) {
    const int b = blockIdx.y;
    const int64_t n = min(lens[b], n_max);
    const int64_t t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= n) return;
    int64_t off = 0;
    for (int i = 0; i < b; ++i) off += min(lens[i], n_max);   // uniform scalar loop, B is a few hundred at most
    const float* src = wave + (int64_t)b * stride + t0;
    float* dst = out + off + t0;
    if (t0 + 3 < n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
        for (int e = 0; e < 4 && t0 + e < n; ++e) dst[e] = src[e];
    }
}

}  // namespace
}  // namespace ttsamd

using namespace ttsamd;

extern "C" {

int32_t ttsamd_dp_unique_id(void* id128) {
    TTS_REQUIRE(id128, "dp_unique_id: null argument");
    TTS_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == TTSAMD_DP_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    TTS_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int32_t ttsamd_dp_init(int32_t rank, int32_t world, const void* id128, void** comm) {
    TTS_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "dp_init: bad argument (rank %d of %d)", rank,
                world);
    TTS_TRY(load_rccl());
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    auto* c = new DpComm();
    c->rank = rank;
    c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);   // binds the CURRENT hip device
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return TTSAMD_EHIP;
    }
    *comm = c;
    return 0;
}

int32_t ttsamd_dp_destroy(void* comm) {
    auto* c = (DpComm*)comm;
    if (!c) return 0;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

int32_t ttsamd_dp_rank(void* comm) { return comm ? ((DpComm*)comm)->rank : -1; }
int32_t ttsamd_dp_world(void* comm) { return comm ? ((DpComm*)comm)->world : 0; }

int32_t ttsamd_dp_broadcast(void* comm, void* buf, int64_t nbytes, int32_t root, void* stream) {
    auto* c = (DpComm*)comm;
    TTS_REQUIRE(c && (buf || nbytes == 0) && nbytes >= 0 && root >= 0 && root < c->world, "dp_broadcast: bad argument");
    if (nbytes == 0) return 0;
    TTS_CHECK_NCCL(g_rccl.Broadcast(buf, buf, (size_t)nbytes, ncclInt8, root, c->comm, (hipStream_t)stream));
    return 0;
}

int32_t ttsamd_dp_broadcast_weights(void* comm, int32_t kind, void* handle, int32_t root, void* stream) {
    auto* c = (DpComm*)comm;
    TTS_REQUIRE(c && handle && (kind == TTSAMD_DP_HIFIGAN || kind == TTSAMD_DP_FASTPITCH) && root >= 0 && root < c->world,
                "dp_broadcast_weights: bad argument");
    void *f32 = nullptr, *b16 = nullptr;
    int64_t n32 = 0, n16 = 0;
    if (kind == TTSAMD_DP_HIFIGAN) hifigan_blobs(handle, &f32, &n32, &b16, &n16);
    else fastpitch_blobs(handle, &f32, &n32, &b16, &n16);
    // every rank built its handle from tensors of the same shapes, so the packed blobs have the same size
    TTS_CHECK_NCCL(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.Broadcast(f32, f32, (size_t)n32 * 4, ncclInt8, root, c->comm, (hipStream_t)stream);
    ncclResult_t r2 = n16 > 0 ? g_rccl.Broadcast(b16, b16, (size_t)n16 * 2, ncclInt8, root, c->comm, (hipStream_t)stream)
                              : ncclSuccess;
    TTS_CHECK_NCCL(g_rccl.GroupEnd());
    TTS_CHECK_NCCL(r1);
    TTS_CHECK_NCCL(r2);
    return 0;
}

int32_t ttsamd_dp_allgather(void* comm, const void* send, void* recv, int64_t nbytes_per_rank, void* stream) {
    auto* c = (DpComm*)comm;
    TTS_REQUIRE(c && send && recv && nbytes_per_rank > 0, "dp_allgather: bad argument");
    TTS_CHECK_NCCL(g_rccl.AllGather(send, recv, (size_t)nbytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream));
    return 0;
}

int32_t ttsamd_dp_pack_audio(const float* wave, int64_t wave_stride, const int64_t* nsamples, int32_t batch,
                             int64_t n_max, float* packed, void* stream) {
    TTS_REQUIRE(wave && nsamples && packed && batch >= 1 && n_max >= 0 && wave_stride >= n_max,
                "dp_pack_audio: bad argument");
    if (n_max == 0) return 0;
    dim3 grid((unsigned)((n_max + 1023) / 1024), (unsigned)batch);
    hipLaunchKernelGGL(pack_ragged_kernel, grid, dim3(256), 0, (hipStream_t)stream, wave, wave_stride, nsamples, n_max,
                       packed);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t ttsamd_dp_gather_audio(void* comm, const float* packed, float* recv, const int64_t* counts,
                               const int64_t* offsets, int32_t root, void* stream) {
    auto* c = (DpComm*)comm;
    TTS_REQUIRE(c && counts && root >= 0 && root < c->world, "dp_gather_audio: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t mine = counts[c->rank];
    TTS_REQUIRE(mine >= 0 && (mine == 0 || packed), "dp_gather_audio: rank %d sends %lld floats from a null buffer", c->rank,
                (long long)mine);
    if (c->rank != root) {
        if (mine > 0) TTS_CHECK_NCCL(g_rccl.Send(packed, (size_t)mine, ncclFloat, root, c->comm, s));
        return 0;
    }
    TTS_REQUIRE(recv && offsets, "dp_gather_audio: the root needs recv and offsets");
    if (mine > 0 && recv + offsets[root] != packed)
        TTS_CHECK_HIP(hipMemcpyAsync(recv + offsets[root], packed, (size_t)mine * 4, hipMemcpyDeviceToDevice, s));
    TTS_CHECK_NCCL(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int r = 0; r < c->world; ++r) {
        if (r == root || counts[r] <= 0) continue;
        ncclResult_t rr = g_rccl.Recv(recv + offsets[r], (size_t)counts[r], ncclFloat, r, c->comm, s);
        if (rr != ncclSuccess) bad = rr;
    }
    TTS_CHECK_NCCL(g_rccl.GroupEnd());
    TTS_CHECK_NCCL(bad);
    return 0;
}

}  // extern "C"
