// yk_comm.hip — the multi-GPU gather of the tile maps behind the C-ABI (SURVEY 8(b) `yk_gather_maps`, 8(e)).
//
// The reference is single-process and single-threaded (include/YAIK.h:42-47, decoder/YAIK_API.cpp:59,68): nothing is translated here.
// Stripes / frames are independent given their pixels, so the ONLY exchange of the path is the concatenation of the per-rank tile maps
// on a root: grouped point-to-point transfers (ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd = ONE collective launch) on the
// handles' own streams, behind the kernels that produced the payloads.  xGMI is point-to-point (one direct link per peer): every sender
// uses its own link into the root, no ring.  Sizes differ per rank, so this is not ncclGather (equal counts); the receiver learns a
// rank's size from the 128-byte header of an EARLIER payload of that rank (yk_export_tile_maps_framed), never from a second collective.
//
// RCCL is bound at run time (dlopen librccl.so.1): a process that already holds a RCCL (PyTorch bundles one under the same SONAME) shares
// that instance, a plain C++ host gets /opt/rocm's; hosts without RCCL keep every other entry point of the library.
#include "yk_common.h"
#include <dlfcn.h>
#include <string.h>
#include <type_traits>
#include <mutex>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_rcclOnce;

const Rccl* rccl() {
    std::call_once(g_rcclOnce, [] {
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            g_rccl.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.so) break;
        }
        if (!g_rccl.so) { g_rccl.err = std::string("RCCL not found (librccl.so.1): ") + dlerror(); return; }
        bool ok = true;
        auto sym = [&](auto& fn, const char* n) { fn = reinterpret_cast<std::decay_t<decltype(fn)>>(dlsym(g_rccl.so, n)); if (!fn) { ok = false; g_rccl.err = std::string("RCCL symbol missing: ") + n; } };
        sym(g_rccl.GetUniqueId, "ncclGetUniqueId"); sym(g_rccl.CommInitRank, "ncclCommInitRank"); sym(g_rccl.CommInitAll, "ncclCommInitAll");
        sym(g_rccl.CommCount, "ncclCommCount"); sym(g_rccl.CommUserRank, "ncclCommUserRank"); sym(g_rccl.CommDestroy, "ncclCommDestroy");
        sym(g_rccl.GroupStart, "ncclGroupStart"); sym(g_rccl.GroupEnd, "ncclGroupEnd"); sym(g_rccl.Send, "ncclSend"); sym(g_rccl.Recv, "ncclRecv");
        sym(g_rccl.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(g_rccl.so); g_rccl.so = nullptr; }
    });
    return g_rccl.so ? &g_rccl : nullptr;
}

int failNccl(yk_ctx* c, const char* what, ncclResult_t r) {
    std::string msg = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return yk_fail(c, YK_ERR_COMM, msg.c_str());
}
#define YK_NCCL(c, call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return failNccl((c), #call, _r); } while (0)
}  // namespace

extern "C" {

int yk_device_alloc(yk_ctx* c, size_t bytes, void** dev) {
    if (!c || !dev) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMalloc(dev, bytes ? bytes : 16));
    return YK_OK;
}
void yk_device_free(yk_ctx* c, void* dev) { if (c && dev) { (void)hipSetDevice(c->device); (void)hipFree(dev); } }
int yk_device_download(yk_ctx* c, void* host, const void* dev, size_t bytes) {
    if (!c || (bytes && (!host || !dev))) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    if (bytes) YK_HIP(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_comm_available(void) { return rccl() ? 1 : 0; }

int yk_comm_unique_id(void* id128) {
    if (!id128) return YK_ERR_BAD_ARG;
    const Rccl* R = rccl(); if (!R) return YK_ERR_COMM;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return YK_ERR_COMM;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return YK_OK;
}

int yk_comm_init_rank(yk_ctx* c, const void* id128, int nRanks, int rank, void** comm) {
    if (!c || !id128 || !comm || nRanks < 1 || rank < 0 || rank >= nRanks) return YK_ERR_BAD_ARG;
    const Rccl* R = rccl(); if (!R) return yk_fail(c, YK_ERR_COMM, g_rccl.err.c_str());
    YK_HIP(c, hipSetDevice(c->device));
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    ncclComm_t cm = nullptr;
    YK_NCCL(c, R->CommInitRank(&cm, nRanks, id, rank));
    *comm = cm;
    return YK_OK;
}

int yk_comm_init_all(yk_ctx* const* ctxs, int n, void** comms) {
    if (!ctxs || !comms || n < 1 || n > 64) return YK_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) if (!ctxs[i]) return YK_ERR_BAD_ARG;
    const Rccl* R = rccl(); if (!R) return yk_fail(ctxs[0], YK_ERR_COMM, g_rccl.err.c_str());
    int devs[64];
    for (int i = 0; i < n; i++) {
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return yk_fail(ctxs[0], YK_ERR_BAD_ARG, "yk_comm_init_all: one handle per device (RCCL ranks need distinct GPUs)");
    }
    ncclComm_t cm[64];
    YK_NCCL(ctxs[0], R->CommInitAll(cm, n, devs));
    for (int i = 0; i < n; i++) comms[i] = cm[i];
    return YK_OK;
}

int yk_comm_ranks(void* comm, int* nRanks, int* rank) {
    const Rccl* R = rccl(); if (!R || !comm) return YK_ERR_BAD_ARG;
    int n = 0, r = 0;
    if (R->CommCount(static_cast<ncclComm_t>(comm), &n) != ncclSuccess || R->CommUserRank(static_cast<ncclComm_t>(comm), &r) != ncclSuccess) return YK_ERR_COMM;
    if (nRanks) *nRanks = n;
    if (rank) *rank = r;
    return YK_OK;
}

void yk_comm_destroy(void* comm) {
    const Rccl* R = rccl();
    if (R && comm) (void)R->CommDestroy(static_cast<ncclComm_t>(comm));
}

// One rank's share of the gather (one process per GPU): every rank but the root sends `sendBytes` from devSend; the root receives
// recvBytes[r] bytes of rank r at devRecv + recvOffsets[r] (its own payload is copied on the device).  All of it is one grouped launch
// on the handle's stream, i.e. behind the export kernel that filled devSend; nothing is synchronised here.
int yk_gather_maps(yk_ctx* c, void* comm, int root, const void* devSend, size_t sendBytes, void* devRecv, const size_t* recvBytes, const size_t* recvOffsets) {
    if (!c || !comm) return YK_ERR_BAD_ARG;
    const Rccl* R = rccl(); if (!R) return yk_fail(c, YK_ERR_COMM, g_rccl.err.c_str());
    int n = 0, me = 0;
    YK_NCCL(c, R->CommCount(static_cast<ncclComm_t>(comm), &n));
    YK_NCCL(c, R->CommUserRank(static_cast<ncclComm_t>(comm), &me));
    if (root < 0 || root >= n) return yk_fail(c, YK_ERR_BAD_ARG, "yk_gather_maps: root");
    if (sendBytes && !devSend) return yk_fail(c, YK_ERR_BAD_ARG, "yk_gather_maps: send buffer");
    if (me == root && (!devRecv || !recvBytes || !recvOffsets)) return yk_fail(c, YK_ERR_BAD_ARG, "yk_gather_maps: the root needs the receive buffer and the per-rank sizes");
    YK_HIP(c, hipSetDevice(c->device));
    ncclComm_t cm = static_cast<ncclComm_t>(comm);
    if (me == root) {
        if (recvBytes[me] < sendBytes) return yk_fail(c, YK_ERR_RANGE, "yk_gather_maps: the root's own slot is smaller than its payload");
        if (sendBytes) YK_HIP(c, hipMemcpyAsync(static_cast<uint8_t*>(devRecv) + recvOffsets[me], devSend, sendBytes, hipMemcpyDeviceToDevice, c->stream));
        YK_NCCL(c, R->GroupStart());
        for (int r = 0; r < n; r++)
            if (r != me && recvBytes[r]) {
                const ncclResult_t rr = R->Recv(static_cast<uint8_t*>(devRecv) + recvOffsets[r], recvBytes[r], ncclUint8, r, cm, c->stream);
                if (rr != ncclSuccess) { (void)R->GroupEnd(); return failNccl(c, "ncclRecv", rr); }
            }
        YK_NCCL(c, R->GroupEnd());
    } else if (sendBytes) {
        YK_NCCL(c, R->GroupStart());
        const ncclResult_t rr = R->Send(devSend, sendBytes, ncclUint8, root, cm, c->stream);
        if (rr != ncclSuccess) { (void)R->GroupEnd(); return failNccl(c, "ncclSend", rr); }
        YK_NCCL(c, R->GroupEnd());
    }
    return YK_OK;
}

// The same for ONE process that drives n devices (the C++ mirror's ConvertHotPathStripes; the reference itself is one process): all sends
// and the root's receives in one group, each on its handle's stream.  Sizes are known to the caller here, so there is no protocol at all.
int yk_gather_maps_all(yk_ctx* const* ctxs, void* const* comms, int n, int root, void* const* devSend, const size_t* sendBytes, void* devRecv, const size_t* recvOffsets) {
    if (!ctxs || !comms || !devSend || !sendBytes || !devRecv || !recvOffsets || n < 1 || root < 0 || root >= n) return YK_ERR_BAD_ARG;
    const Rccl* R = rccl(); if (!R) return yk_fail(ctxs[0], YK_ERR_COMM, g_rccl.err.c_str());
    yk_ctx* rc = ctxs[root];
    YK_HIP(rc, hipSetDevice(rc->device));
    if (sendBytes[root]) YK_HIP(rc, hipMemcpyAsync(static_cast<uint8_t*>(devRecv) + recvOffsets[root], devSend[root], sendBytes[root], hipMemcpyDeviceToDevice, rc->stream));
    YK_NCCL(rc, R->GroupStart());
    for (int r = 0; r < n; r++) {
        if (r == root || !sendBytes[r]) continue;
        ncclResult_t rr = R->Send(devSend[r], sendBytes[r], ncclUint8, root, static_cast<ncclComm_t>(comms[r]), ctxs[r]->stream);
        if (rr == ncclSuccess) rr = R->Recv(static_cast<uint8_t*>(devRecv) + recvOffsets[r], sendBytes[r], ncclUint8, r, static_cast<ncclComm_t>(comms[root]), rc->stream);
        if (rr != ncclSuccess) { (void)R->GroupEnd(); return failNccl(rc, "ncclSend / ncclRecv", rr); }
    }
    YK_NCCL(rc, R->GroupEnd());
    return YK_OK;
}

}  // extern "C"
