// yk_api.hip — the C-ABI of include/yaik_hip.h: handle lifetime, HBM buffers, launch order, result getters.
// Host-side glue only; every pixel is touched by the kernels in yk_stages.hip / yk_encode2.hip / yk_corners.hip / yk_partial.hip / yk_range1d.hip / yk_decode.hip.
#include "yk_common.h"
#ifdef YK_TEST_HOOKS
#include "../../include/yaik_hip_test.h"
#endif
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

int yk_fail(yk_ctx* c, int code, const char* what, hipError_t e) {
    if (c && c->err.empty()) {
        c->err = what ? what : "error";
        if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
    }
    return code;
}

void yk_rebase(yk_ctx* c, int f) {
    const YkFrameStrides& fs = c->fs; auto& B = c->B;
    c->curFrame = f;
    for (int i = 0; i < 4; i++) c->plane[i] = B.plane[i] ? B.plane[i] + (size_t)f * fs.plane : nullptr;
    c->keep = B.keep + (size_t)f * fs.keep; c->bounds = B.bounds + (size_t)f * 16;
    for (int i = 0; i < 7; i++) c->bitmap[i] = B.bitmap[i] + (size_t)f * fs.bitmap[i];
    c->coverage = B.coverage + (size_t)f * fs.coverage; c->tileDef = B.tileDef + (size_t)f * fs.tileDef;
    c->tileCount = B.tileCount + (size_t)f * fs.tileCount; c->slots = B.slots + (size_t)f * fs.slots;
    c->blockSums = B.blockSums + (size_t)f * fs.blockN; c->blockCnt = B.blockCnt + (size_t)f * fs.blockN;
    c->totals = B.totals + (size_t)f * 8; c->defsOut = B.defsOut + (size_t)f * fs.defsOut; c->nibOut = B.nibOut + (size_t)f * fs.nibOut;
}

static void yk_free_image(yk_ctx* c) {
    auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
    auto& B = c->B;
    F(B.keep); F(B.bounds); F(c->alphaUnitBox); F(c->alphaArrive);
    F(B.small);                                                 // bitmaps, coverage, tile records, run sums: one allocation
    for (int i = 0; i < 7; i++) B.bitmap[i] = nullptr;
    B.coverage = nullptr; B.bm0b = nullptr; B.tileInfo = nullptr; B.runSums = nullptr;
    F(B.tileDef); F(B.tileCount); F(B.slots);
    for (int i = 0; i < 3; i++) F(c->dst[i]);
    F(B.blockSums); F(B.blockCnt); F(B.totals); F(c->exportSizes); F(B.defsOut); F(B.nibOut);
    c->keep = nullptr; c->bounds = nullptr; for (int i = 0; i < 7; i++) c->bitmap[i] = nullptr;
    c->coverage = nullptr; c->tileDef = nullptr; c->tileCount = nullptr; c->slots = nullptr;
    c->blockSums = nullptr; c->blockCnt = nullptr; c->totals = nullptr; c->defsOut = nullptr; c->nibOut = nullptr;
    F(c->latticeOwner); F(c->cornerStream); F(c->cornerScratch); F(c->cornerEdgeIdx);
    F(c->preview); F(c->covCh); F(c->mapped3); F(c->ppBitmap); F(c->ppStream); F(c->ppScratch); c->ppBitmapBytes = c->ppBitmapCap = 0; c->ppStreamCap = c->ppStreamBytes = 0; c->ppScratchElems = 0;
    F(c->r1Slots); F(c->r1Params); F(c->r1Cnt); F(c->r1Pix); F(c->r1Type); c->r1Ready = false;
    F(c->pixCache); c->pixCacheValid = false;
    if (c->frameGraph) { (void)hipGraphExecDestroy(c->frameGraph); c->frameGraph = nullptr; }
    c->encoded = false; c->alphaDone = false; c->alphaFinished = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false;
}

// allocates every per-image array c->nFrames times (geometry fields are set) and points the handle at frame 0
static int yk_alloc_image(yk_ctx* c) {
    const size_t F = (size_t)c->nFrames;
    const size_t T8 = (size_t)c->tilesW * c->tilesH, MT = (size_t)c->mtW * c->mtH;
    auto& B = c->B; YkFrameStrides& fs = c->fs;
    auto up = [](size_t n, size_t a) { return (n + a - 1) / a * a; };
    fs.keep = up(MT + 4, 16);                                   // read as 4-byte words by yk_alpha_bbox_kernel
    YK_HIP(c, hipMalloc(&B.keep, fs.keep * F + 16));
    YK_HIP(c, hipMalloc(&B.bounds, 16 * sizeof(int32_t) * F));
    {
        const size_t nUnits = (size_t)((c->fullW / 4 + 63) / 64) * c->mtH, nGroups = (nUnits + 63) / 64;     // sized for the smallest unit yk_alpha_kernel may use (64 int4 per segment)
        YK_HIP(c, hipMalloc(&c->alphaUnitBox, (nUnits + nGroups) * F * 4 * sizeof(int) + 16));
        YK_HIP(c, hipMalloc(&c->alphaArrive, (nGroups + 1) * F * sizeof(uint32_t) + 16));
        YK_HIP(c, hipMemset(c->alphaArrive, 0, (nGroups + 1) * F * sizeof(uint32_t) + 16));
    }
    static const int sh[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    size_t cur = 0, oBm[7], oBm0b, oCov, oInfo, oRun;
    auto place = [&](size_t bytesPerFrame) { const size_t o = cur; cur = up(cur + bytesPerFrame * F + 16, 256); return o; };
    for (int i = 0; i < 7; i++) {
        const int bx = (sh[i][0] == 2) ? 32 : 64, by = (sh[i][1] == 2) ? 32 : 64;
        const size_t bits = (size_t)((c->fullW + bx - 1) / bx) * ((c->h + by - 1) / by) * ((bx >> sh[i][0]) * (by >> sh[i][1]));
        c->bitmapBytes[i] = bits >> 3;
        fs.bitmap[i] = up(c->bitmapBytes[i] + 4, 16);           // + the padding word pass 0 may touch
        oBm[i] = place(fs.bitmap[i]);
    }
    const size_t nB64 = (size_t)((c->fullW + 63) / 64) * ((c->h + 63) / 64);
    fs.bm0b = up(nB64 * 4, 16); oBm0b = place(fs.bm0b);
    fs.coverage = up(MT, 8); oCov = place(fs.coverage * sizeof(uint16_t));
    fs.tileInfo = up(T8, 2); oInfo = place(fs.tileInfo * sizeof(uint2));
    fs.runSums = up((T8 + 7) / 8, 4); oRun = place(fs.runSums * sizeof(uint32_t));
    if (cur >= ((size_t)1 << 32)) return yk_fail(c, YK_ERR_BAD_ARG, "image too large for 32-bit offsets into the small-output arena");
    YK_HIP(c, hipMalloc(&B.small, cur));
    YK_HIP(c, hipMemset(B.small, 0, cur));                      // allocation time only
    for (int i = 0; i < 7; i++) B.bitmap[i] = B.small + oBm[i];
    B.bm0b = B.small + oBm0b; B.coverage = reinterpret_cast<uint16_t*>(B.small + oCov);
    B.tileInfo = reinterpret_cast<uint2*>(B.small + oInfo); B.runSums = reinterpret_cast<uint32_t*>(B.small + oRun);
    fs.tileDef = 3 * T8; fs.tileCount = 3 * T8; fs.slots = 3 * T8 * YK_SLOT;
    YK_HIP(c, hipMalloc(&B.tileDef, fs.tileDef * F * sizeof(uint16_t) + 16));
    YK_HIP(c, hipMalloc(&B.tileCount, fs.tileCount * F + 16));
    YK_HIP(c, hipMalloc(&B.slots, fs.slots * F + 16));
    c->nScanBlocks = (int)((T8 + 1023) / 1024);
    fs.blockN = (size_t)c->nScanBlocks * 2;
    YK_HIP(c, hipMalloc(&B.blockSums, fs.blockN * F * sizeof(uint32_t)));
    YK_HIP(c, hipMalloc(&B.blockCnt, fs.blockN * F * sizeof(uint32_t)));
    YK_HIP(c, hipMemset(B.blockCnt, 0, fs.blockN * F * sizeof(uint32_t)));                   // synchronous (allocation time); kept zero between frames by the scan
    YK_HIP(c, hipMalloc(&B.totals, 8 * sizeof(uint32_t) * F));
    fs.defsOut = 3 * T8;
    YK_HIP(c, hipMalloc(&B.defsOut, fs.defsOut * F * sizeof(uint16_t) + 16));
    c->nibStride = T8 * YK_SLOT + 64;
    fs.nibOut = 3 * c->nibStride;
    YK_HIP(c, hipMalloc(&B.nibOut, fs.nibOut * F + 16));
    c->dstValid = false;
    yk_rebase(c, 0);
    return YK_OK;
}

static int yk_stage_fold(yk_ctx* c, int st) {               // waits for the recorded intervals of a stage and adds them up
    for (int k = 0; k < c->stN[st]; k++) {
        float t = 0;
        YK_HIP(c, hipEventSynchronize(c->stEv[st][k][1]));
        YK_HIP(c, hipEventElapsedTime(&t, c->stEv[st][k][0], c->stEv[st][k][1]));
        c->stAcc[st] += t; c->stCalls[st]++;
    }
    c->stN[st] = 0;
    return YK_OK;
}
// the ring is full: account the intervals that have already completed (no host wait) and, if none has, drop the oldest one -- a caller that keeps
// frames in flight is never stopped here (the averages of yk_stage_ms then miss that interval; callers that want every one query at most every
// YK_STAGE_RING intervals)
static int yk_stage_make_room(yk_ctx* c, int st) {
    int kept = 0;
    for (int k = 0; k < c->stN[st]; k++) {
        hipEvent_t e0 = c->stEv[st][k][0], e1 = c->stEv[st][k][1];
        if (hipEventQuery(e1) == hipSuccess) {
            float t = 0;
            if (hipEventElapsedTime(&t, e0, e1) == hipSuccess) { c->stAcc[st] += t; c->stCalls[st]++; }
        } else {
            (void)hipGetLastError();                                          // hipErrorNotReady is not an error here
            if (kept != k) { std::swap(c->stEv[st][kept][0], c->stEv[st][k][0]); std::swap(c->stEv[st][kept][1], c->stEv[st][k][1]); }
            kept++;
            continue;
        }
    }
    if (kept == YK_STAGE_RING) {                                              // nothing has completed: the oldest interval is dropped
        hipEvent_t d0 = c->stEv[st][0][0], d1 = c->stEv[st][0][1];
        for (int k = 1; k < kept; k++) { c->stEv[st][k - 1][0] = c->stEv[st][k][0]; c->stEv[st][k - 1][1] = c->stEv[st][k][1]; }
        c->stEv[st][kept - 1][0] = d0; c->stEv[st][kept - 1][1] = d1;
        kept--;
    }
    c->stN[st] = kept;
    return YK_OK;
}
int yk_stage_begin(yk_ctx* c, int st) {
    if (st < 0 || st >= YK_NUM_STAGES) return YK_ERR_BAD_ARG;
    if (c->stN[st] == YK_STAGE_RING) { int rc = yk_stage_make_room(c, st); if (rc) return rc; }
    hipEvent_t* ev = c->stEv[st][c->stN[st]];
    for (int i = 0; i < 2; i++) if (!ev[i]) YK_HIP(c, hipEventCreate(&ev[i]));
    YK_HIP(c, hipEventRecord(ev[0], c->stream));
    return YK_OK;
}
int yk_stage_end(yk_ctx* c, int st) {
    if (st < 0 || st >= YK_NUM_STAGES) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipEventRecord(c->stEv[st][c->stN[st]][1], c->stream));
    c->stN[st]++;
    return YK_OK;
}

extern "C" {

int yk_stage_ms(yk_ctx* c, int stage, float* msSum, int* intervals) {
    if (!c || stage < 0 || stage >= YK_NUM_STAGES) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    int rc = yk_stage_fold(c, stage); if (rc) return rc;
    if (msSum) *msSum = (float)c->stAcc[stage];
    if (intervals) *intervals = c->stCalls[stage];
    c->stAcc[stage] = 0; c->stCalls[stage] = 0;
    return YK_OK;
}

int yk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int yk_create(int device, yk_ctx** out) {
    if (!out) return YK_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return YK_ERR_NO_DEVICE;      // no CPU fallback, by design
    if (device < 0 || device >= n) return YK_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return YK_ERR_NO_DEVICE;
    yk_ctx* c = new yk_ctx();
    c->device = device;
    if (hipDeviceGetAttribute(&c->numCU, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->numCU <= 0) c->numCU = 256;
    if (hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) { delete c; return YK_ERR_HIP; }
    c->stream = c->ownStream;
    for (int r = 0; r < YK_EV_RING; r++) for (int i = 0; i < 5; i++) if (hipEventCreate(&c->evRing[r][i]) != hipSuccess) { delete c; return YK_ERR_HIP; }
    if (yk_qtab_get(c) != YK_OK) { yk_destroy(c); return YK_ERR_HIP; }
    *out = c;
    return YK_OK;
}

void yk_destroy(yk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    yk_free_image(c);
    yk_lut_destroy(c);
    yk_lut_dec_destroy(c);
    auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
    F(c->ownedPlanes); F(c->dPlanes); F(c->dMapRGB); F(c->dLatticeOwner); F(c->dTile4); F(c->dScratch); F(c->dLoaded);
    for (int r = 0; r < YK_EV_RING; r++) for (int i = 0; i < 5; i++) if (c->evRing[r][i]) (void)hipEventDestroy(c->evRing[r][i]);
    for (int st = 0; st < YK_NUM_STAGES; st++) for (int k = 0; k < YK_STAGE_RING; k++) for (int i = 0; i < 2; i++) if (c->stEv[st][k][i]) (void)hipEventDestroy(c->stEv[st][k][i]);
    if (c->frameGraph) (void)hipGraphExecDestroy(c->frameGraph);
    if (c->evHandoff) (void)hipEventDestroy(c->evHandoff);
    if (c->evFusedAfter) (void)hipEventDestroy(c->evFusedAfter);
    if (c->auxStream) (void)hipStreamDestroy(c->auxStream);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

const char* yk_last_error(const yk_ctx* c) { return c ? c->err.c_str() : "null handle"; }

int yk_set_stream(yk_ctx* c, void* s) {
    if (!c) return YK_ERR_BAD_ARG;
    hipStream_t next = s ? (hipStream_t)s : c->ownStream;
    if (next == c->stream) return YK_OK;
    // Work already queued on the old stream (uploads, the clears of yk_set_image, an encode) must stay ordered before whatever the
    // caller queues on the new one: the new stream waits, on the device, for an event recorded on the old stream.
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->evHandoff) YK_HIP(c, hipEventCreateWithFlags(&c->evHandoff, hipEventDisableTiming));
    // A caller's stream may be gone by now (the header asks for it to outlive this call, but a handle bound to a dead stream would fail
    // forever): if the old stream no longer takes an event there is nothing left on it to wait for, and the switch still happens.
    if (hipEventRecord(c->evHandoff, c->stream) == hipSuccess) YK_HIP(c, hipStreamWaitEvent(next, c->evHandoff, 0));
    else (void)hipGetLastError();
    c->stream = next;
    return YK_OK;
}

int yk_synchronize(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_set_image(yk_ctx* c, int fullW, int fullH, int nPlanes, int y0, int h, int haloRows) {
    if (!c) return YK_ERR_BAD_ARG;
    if (fullW < 8 || fullH < 8 || (fullW & 7) || (fullH & 7) || fullW > 32760 || fullH > 32760) return yk_fail(c, YK_ERR_BAD_ARG, "image size must be a multiple of 8 and fit s16 fields");
    if (nPlanes != 3 && nPlanes != 4) return yk_fail(c, YK_ERR_BAD_ARG, "nPlanes must be 3 or 4");
    if (y0 < 0 || h < 8 || (h & 7) || y0 + h > fullH || (y0 & 63)) return yk_fail(c, YK_ERR_BAD_ARG, "stripe rows must start on a multiple of 64 and lie inside the image");
    if (y0 + h < fullH && ((h & 63) || haloRows != 1)) return yk_fail(c, YK_ERR_BAD_ARG, "inner stripes need h % 64 == 0 and one halo row");
    if (y0 + h == fullH && haloRows != 0) return yk_fail(c, YK_ERR_BAD_ARG, "the last stripe has no halo row");
    YK_HIP(c, hipSetDevice(c->device));
    c->fusedAfter = nullptr;                                                 // an ordering request never outlives the image it was made for
    if (c->fullW == fullW && c->fullH == fullH && c->nPlanes == nPlanes && c->y0 == y0 && c->h == h && c->halo == haloRows && c->tileCount) {
        c->encoded = false; c->alphaDone = false; c->alphaFinished = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false;
        return YK_OK;
    }
    YK_HIP(c, hipStreamSynchronize(c->stream));
    yk_free_image(c);
    c->fullW = fullW; c->fullH = fullH; c->nPlanes = nPlanes; c->y0 = y0; c->h = h; c->halo = haloRows;
    c->tilesW = fullW / 8; c->tilesH = h / 8; c->mtW = (fullW + 15) / 16; c->mtH = (h + 15) / 16;
    c->nFrames = 1; c->fs.plane = 0;
    for (int i = 0; i < 4; i++) { c->B.plane[i] = nullptr; c->plane[i] = nullptr; }
    return yk_alloc_image(c);
}

int yk_set_batch(yk_ctx* c, int nFrames) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->tileCount) return yk_fail(c, YK_ERR_STATE, "yk_set_image first");
    if (nFrames < 1 || nFrames > 1024) return yk_fail(c, YK_ERR_BAD_ARG, "nFrames must be 1..1024");
    if (c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "batches hold whole images, not stripes");
    YK_HIP(c, hipSetDevice(c->device));
    if (nFrames != c->nFrames) {
        YK_HIP(c, hipStreamSynchronize(c->stream));
        const int32_t* keepPlanes[4] = { c->B.plane[0], c->B.plane[1], c->B.plane[2], c->B.plane[3] };
        const unsigned long long planeStride = c->fs.plane;
        yk_free_image(c);
        c->nFrames = nFrames;
        int rc = yk_alloc_image(c); if (rc) return rc;
        for (int i = 0; i < 4; i++) c->B.plane[i] = keepPlanes[i];
        c->fs.plane = planeStride;
        yk_rebase(c, 0);
    }
    return YK_OK;
}

int yk_select_frame(yk_ctx* c, int frame) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->tileCount) return yk_fail(c, YK_ERR_STATE, "yk_set_image first");
    if (frame < 0 || frame >= c->nFrames) return yk_fail(c, YK_ERR_BAD_ARG, "frame out of range");
    yk_rebase(c, frame);
    c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false; c->r1Ready = false;
    return YK_OK;
}

static int yk_check_planes(yk_ctx* c, const int32_t* const p[4], int strideElems) {
    if (!c || !p) return YK_ERR_BAD_ARG;
    if (!c->tileCount) return yk_fail(c, YK_ERR_STATE, "yk_set_image first");
    if (strideElems < c->fullW || (strideElems & 3)) return yk_fail(c, YK_ERR_BAD_ARG, "row pitch must be >= width and a multiple of 4 elements");
    for (int i = 0; i < c->nPlanes; i++) if (!p[i]) return yk_fail(c, YK_ERR_BAD_ARG, "null plane");
    return YK_OK;
}

// Precondition of the whole path (include/yaik_hip.h): plane samples lie in 0..255 -- the fused kernel keeps only their low byte, the
// reference reads the full int (framework.h:116-121), so an out-of-range sample would silently give different tiles.  One streaming pass over the bound
// planes (rows of this stripe incl. its halo) counts the samples outside; a workgroup adds its count with one atomic, and only when it has any.
__global__ __launch_bounds__(256) void yk_validate_kernel(const int32_t* __restrict__ plane, int strideElems, int w, int rows, unsigned long long* __restrict__ bad) {
    __shared__ unsigned s_bad;
    if (threadIdx.x == 0) s_bad = 0u;
    __syncthreads();
    const int vecPerRow = w >> 2;
    unsigned mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)vecPerRow * rows; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / vecPerRow), xv = (int)(i - (long long)y * vecPerRow);
        const int4 v = *reinterpret_cast<const int4*>(plane + (size_t)y * strideElems + (size_t)xv * 4);
        mine += ((v.x & ~255) ? 1u : 0u) + ((v.y & ~255) ? 1u : 0u) + ((v.z & ~255) ? 1u : 0u) + ((v.w & ~255) ? 1u : 0u);
    }
    if (mine) atomicAdd(&s_bad, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_bad) atomicAdd(bad, (unsigned long long)s_bad);
}

int yk_validate_planes(yk_ctx* c, size_t* nOutOfRange) {
    if (!c || !nOutOfRange) return YK_ERR_BAD_ARG;
    if (!c->B.plane[0]) return yk_fail(c, YK_ERR_STATE, "bind planes first");
    YK_HIP(c, hipSetDevice(c->device));
    unsigned long long* dBad = nullptr;
    YK_HIP(c, hipMalloc(&dBad, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(dBad, 0, sizeof(unsigned long long), c->stream);
    const int rows = c->h + c->halo;
    for (int f = 0; f < c->nFrames && e == hipSuccess; f++)
        for (int i = 0; i < c->nPlanes; i++)
            hipLaunchKernelGGL(yk_validate_kernel, dim3(c->numCU * 8), dim3(256), 0, c->stream, c->B.plane[i] + (size_t)f * c->fs.plane, c->strideElems, c->fullW, rows, dBad);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, dBad, sizeof bad, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dBad);
    if (e != hipSuccess) return yk_fail(c, YK_ERR_HIP, "yk_validate_planes", e);
    *nOutOfRange = (size_t)bad;
    return YK_OK;
}

int yk_upload_planes(yk_ctx* c, const int32_t* const hostPlanes[4], int strideElems) {
    int rc = yk_check_planes(c, hostPlanes, strideElems); if (rc) return rc;
    YK_HIP(c, hipSetDevice(c->device));
    const size_t rows = (size_t)c->h + c->halo, planeBytes = rows * c->fullW * sizeof(int32_t);
    if (c->ownedPlanesBytes < planeBytes * c->nPlanes) {
        if (c->ownedPlanes) { YK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->ownedPlanes); c->ownedPlanes = nullptr; }
        YK_HIP(c, hipMalloc(&c->ownedPlanes, planeBytes * c->nPlanes));
        c->ownedPlanesBytes = planeBytes * c->nPlanes;
    }
    if (c->nFrames != 1) return yk_fail(c, YK_ERR_STATE, "batches bind device planes (yk_bind_device_batch)");
    for (int i = 0; i < 4; i++) c->B.plane[i] = nullptr;
    for (int i = 0; i < c->nPlanes; i++) {
        int32_t* d = c->ownedPlanes + (size_t)i * rows * c->fullW;
        YK_HIP(c, hipMemcpy2DAsync(d, (size_t)c->fullW * 4, hostPlanes[i], (size_t)strideElems * 4, (size_t)c->fullW * 4, rows, hipMemcpyHostToDevice, c->stream));
        c->B.plane[i] = d;
    }
    c->fs.plane = 0; yk_rebase(c, 0);
    c->strideElems = c->fullW;
    c->encoded = false; c->alphaDone = false; c->alphaFinished = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false;
    // the 0..255 precondition is enforced where the boundary hands over host planes (a streaming pass over what was just copied: ~0.2 ms next to
    // ~48 ms of PCIe copy for an 8192 x 8192 RGBA image); callers that bind device memory check with yk_validate_planes when they cannot vouch for it
    size_t bad = 0;
    rc = yk_validate_planes(c, &bad); if (rc) return rc;
    if (bad) {
        for (int i = 0; i < 4; i++) c->B.plane[i] = nullptr;
        yk_rebase(c, 0);
        return yk_fail(c, YK_ERR_BAD_ARG, ("plane samples outside 0..255: " + std::to_string(bad) + " (the tile path is defined for 8-bit samples held in int32 planes)").c_str());
    }
    return YK_OK;
}

int yk_bind_device_planes(yk_ctx* c, const int32_t* const devPlanes[4], int strideElems) {
    int rc = yk_check_planes(c, devPlanes, strideElems); if (rc) return rc;
    if (c->nFrames != 1) return yk_fail(c, YK_ERR_STATE, "batches bind device planes with yk_bind_device_batch");
    for (int i = 0; i < 4; i++) c->B.plane[i] = nullptr;
    for (int i = 0; i < c->nPlanes; i++) {
        if (((uintptr_t)devPlanes[i]) & 15) return yk_fail(c, YK_ERR_BAD_ARG, "device planes must be 16-byte aligned");
        c->B.plane[i] = devPlanes[i];
    }
    c->fs.plane = 0; yk_rebase(c, 0);
    c->strideElems = strideElems;
    c->encoded = false; c->alphaDone = false; c->alphaFinished = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false;
    return YK_OK;
}

int yk_bind_device_batch(yk_ctx* c, const int32_t* const frame0Planes[4], int strideElems, size_t frameStrideElems) {
    int rc = yk_check_planes(c, frame0Planes, strideElems); if (rc) return rc;
    if (c->nFrames > 1 && (frameStrideElems < (size_t)strideElems * c->fullH || (frameStrideElems & 3))) return yk_fail(c, YK_ERR_BAD_ARG, "frame stride must cover a plane and be a multiple of 4 elements");
    for (int i = 0; i < 4; i++) c->B.plane[i] = nullptr;
    for (int i = 0; i < c->nPlanes; i++) {
        if (((uintptr_t)frame0Planes[i]) & 15) return yk_fail(c, YK_ERR_BAD_ARG, "device planes must be 16-byte aligned");
        c->B.plane[i] = frame0Planes[i];
    }
    c->fs.plane = frameStrideElems; yk_rebase(c, c->curFrame < c->nFrames ? c->curFrame : 0);
    c->strideElems = strideElems;
    c->encoded = false; c->alphaDone = false; c->alphaFinished = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false;
    return YK_OK;
}

// ---- alpha -----------------------------------------------------------------------------------------
int yk_alpha_reject(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->plane[0]) return yk_fail(c, YK_ERR_STATE, "bind planes first");
    if (c->nPlanes != 4) return yk_fail(c, YK_ERR_STATE, "image has no alpha plane");
    YK_HIP(c, hipSetDevice(c->device));
    hipEvent_t* ev = c->evRing[c->evHead % YK_EV_RING];
    YK_HIP(c, hipEventRecord(ev[0], c->stream));
    int rc = yk_launch_alpha(c); if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[1], c->stream));
    c->evAlphaInCur = true;
    c->alphaDone = true; c->alphaFinished = false;
    return YK_OK;
}

int yk_get_stripe_bbox(yk_ctx* c, int32_t bbox[4]) {
    if (!c || !bbox) return YK_ERR_BAD_ARG;
    if (!c->alphaDone) return yk_fail(c, YK_ERR_STATE, "yk_alpha_reject first");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(bbox, c->bounds + 8, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_alpha_finish(yk_ctx* c, const int32_t globalBBox[4]) {
    if (!c) return YK_ERR_BAD_ARG;
    if (c->nPlanes != 4) return YK_OK;
    if (!c->alphaDone) return yk_fail(c, YK_ERR_STATE, "yk_alpha_reject first");
    YK_HIP(c, hipSetDevice(c->device));
    int rc = yk_launch_alpha_finish(c, globalBBox); if (rc) return rc;
    c->alphaFinished = true;
    return YK_OK;
}

static int yk_fetch_alpha(yk_ctx* c, std::vector<uint8_t>& keep, int32_t b[5]) {
    if (!c->alphaFinished) return yk_fail(c, YK_ERR_STATE, "yk_alpha_finish first");
    YK_HIP(c, hipSetDevice(c->device));
    keep.resize((size_t)c->mtW * c->mtH);
    YK_HIP(c, hipMemcpyAsync(keep.data(), c->keep, keep.size(), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipMemcpyAsync(b, c->bounds + c->boundsOff, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    b[4] = (b[0] == 0 && b[1] == 0 && b[2] == c->fullW && b[3] == c->fullH) ? 1 : 0;      // "bbox == whole image -> every reject discarded" (EncoderContext.cpp:1294, :1400-1403)
    return YK_OK;
}

int yk_alpha_result(yk_ctx* c, int32_t bounds[4], int* hasChunk, int* remainingPixels, int32_t tileBBox[4]) {
    if (!c) return YK_ERR_BAD_ARG;
    if (c->nPlanes != 4) {                               // EncoderContext.cpp:1419-1426
        if (bounds) { bounds[0] = 0; bounds[1] = 0; bounds[2] = c->fullW; bounds[3] = c->fullH; }
        if (hasChunk) *hasChunk = 0;
        if (remainingPixels) *remainingPixels = c->fullW * c->h;
        return YK_OK;
    }
    std::vector<uint8_t> keep; int32_t b[5];
    int rc = yk_fetch_alpha(c, keep, b); if (rc) return rc;
    if (bounds) memcpy(bounds, b, 4 * sizeof(int32_t));
    if (hasChunk) *hasChunk = b[4] ? 0 : 1;
    if (remainingPixels) {
        if (b[4]) *remainingPixels = c->fullW * c->h;    // :1402 (this stripe's share)
        else { int n = 0; for (uint8_t k : keep) n += k ? 256 : 0; *remainingPixels = n; }   // :1323
    }
    if (tileBBox) { tileBBox[0] = b[0] >> 4; tileBBox[1] = b[1] >> 4; tileBBox[2] = (b[2] >> 4) - (b[0] >> 4); tileBBox[3] = (b[3] >> 4) - (b[1] >> 4); }
    return YK_OK;
}

int yk_alpha_bitmap(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c || !hostOut) return YK_ERR_BAD_ARG;
    if (c->nPlanes != 4) { if (nBytes) *nBytes = 0; return YK_OK; }
    std::vector<uint8_t> keep; int32_t b[5];
    int rc = yk_fetch_alpha(c, keep, b); if (rc) return rc;
    if (b[4] || b[2] < b[0]) { if (nBytes) *nBytes = 0; return YK_OK; }
    const int bx0 = b[0] >> 4, by0 = b[1] >> 4, tw = (b[2] >> 4) - bx0, th = (b[3] >> 4) - by0;
    const size_t sz = ((size_t)tw * th + 7) / 8;
    if (cap < sz) return yk_fail(c, YK_ERR_RANGE, "alpha bitmap buffer too small");
    memset(hostOut, 0, sz);
    const int my0 = c->y0 >> 4;
    for (int y = 0; y < th; y++) {                       // bit order of EncoderContext.cpp:1317-1327
        const int my = by0 + y - my0;
        if (my < 0 || my >= c->mtH) continue;
        for (int x = 0; x < tw; x++) {
            if (keep[(size_t)my * c->mtW + bx0 + x]) { const size_t bp = (size_t)y * tw + x; hostOut[bp >> 3] |= (uint8_t)(1u << (bp & 7)); }
        }
    }
    if (nBytes) *nBytes = sz;
    return YK_OK;
}

// ---- fused encode ------------------------------------------------------------------------------------
#ifdef YK_TEST_HOOKS                                     // include/yaik_hip_test.h: only in the test build of the library
int yk_set_kernel_version(yk_ctx* c, int version) {
    if (!c || (version != 1 && version != 2)) return YK_ERR_BAD_ARG;
    c->kernelVersion = version; return YK_OK;
}

int yk_set_ablation(yk_ctx* c, int flags) { if (!c) return YK_ERR_BAD_ARG; c->ablate = flags; return YK_OK; }
#endif

int yk_set_pixel_cache(yk_ctx* c, int enable) {
    if (!c) return YK_ERR_BAD_ARG;
    c->pixCacheOn = enable != 0;
    if (!enable) c->pixCacheValid = false;
    return YK_OK;
}

int yk_set_dst_fill(yk_ctx* c, int32_t fill) { if (!c) return YK_ERR_BAD_ARG; c->dstFill = fill; return YK_OK; }

int yk_encode_tiles(yk_ctx* c, int rejectFactor, int mode3BitOnly, int wantDst) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->plane[0]) return yk_fail(c, YK_ERR_STATE, "bind planes first");
    if (rejectFactor < 0 || rejectFactor > 64) return yk_fail(c, YK_ERR_BAD_ARG, "rejectFactor out of range");
    if (c->nPlanes == 4 && !c->alphaFinished) return yk_fail(c, YK_ERR_STATE, "RGBA image: run yk_alpha_reject + yk_alpha_finish first (MipPrefilter precedes the tile passes)");
    YK_HIP(c, hipSetDevice(c->device));
    if (wantDst) {
        const size_t n = (size_t)c->fullW * c->h;
        for (int p = 0; p < 3; p++) {
            if (!c->dst[p]) YK_HIP(c, hipMalloc(&c->dst[p], n * sizeof(int32_t)));
            if (c->dstFill == 0 || c->dstFill == -1) YK_HIP(c, hipMemsetAsync(c->dst[p], c->dstFill & 255, n * sizeof(int32_t), c->stream));
            else {
                std::vector<int32_t> f(n, c->dstFill);
                YK_HIP(c, hipMemcpyAsync(c->dst[p], f.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                YK_HIP(c, hipStreamSynchronize(c->stream));
            }
        }
    }
    hipEvent_t* ev = c->evRing[c->evHead % YK_EV_RING];
    if (!c->evAlphaInCur) { YK_HIP(c, hipEventRecord(ev[0], c->stream)); YK_HIP(c, hipEventRecord(ev[1], c->stream)); }   // no alpha stage: zero-length interval
    if (c->fusedAfter) { YK_HIP(c, hipStreamWaitEvent(c->stream, c->fusedAfter, 0)); c->fusedAfter = nullptr; }   // yk_order_fused_after
    YK_HIP(c, hipEventRecord(ev[2], c->stream));
    int rc = yk_launch_encode(c, rejectFactor, mode3BitOnly, wantDst); if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[3], c->stream));
    rc = yk_launch_pack(c); if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[4], c->stream));
    c->evAlphaInCur = false;
    c->evHead++;
    if (c->evHead - c->evTail > YK_EV_RING) c->evTail = c->evHead - YK_EV_RING;       // the oldest sets were overwritten
    c->encoded = true; c->dstValid = wantDst != 0; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false; c->nextCornerPass = 0; c->r1Ready = false;
    return YK_OK;
}

int yk_order_fused_after(yk_ctx* c, const yk_ctx* other) {
    if (!c || !other) return YK_ERR_BAD_ARG;
    if (c->device != other->device) return yk_fail(c, YK_ERR_BAD_ARG, "yk_order_fused_after: handles of one device");
    c->fusedAfter = nullptr;
    if (!other->evHead) return YK_OK;
    // Nothing of `other` is kept beyond this call (it may be destroyed before c encodes again): an auxiliary stream of c waits for the end of
    // other's fused kernel NOW, while that event certainly exists, and an event of c's own is recorded behind the wait; c's next fused
    // kernel waits for that one.
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->auxStream) YK_HIP(c, hipStreamCreateWithFlags(&c->auxStream, hipStreamNonBlocking));
    if (!c->evFusedAfter) YK_HIP(c, hipEventCreateWithFlags(&c->evFusedAfter, hipEventDisableTiming));
    static const int afterPack = getenv("YK_ORDER_AFTER_PACK") ? atoi(getenv("YK_ORDER_AFTER_PACK")) : 0;      // experiment switch
    YK_HIP(c, hipStreamWaitEvent(c->auxStream, other->evRing[(other->evHead - 1) % YK_EV_RING][afterPack ? 4 : 3], 0));
    YK_HIP(c, hipEventRecord(c->evFusedAfter, c->auxStream));
    c->fusedAfter = c->evFusedAfter;
    return YK_OK;
}

int yk_encode_batch(yk_ctx* c, int rejectFactor, int mode3BitOnly) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->B.plane[0]) return yk_fail(c, YK_ERR_STATE, "bind planes first");
    if (rejectFactor < 0 || rejectFactor > 64) return yk_fail(c, YK_ERR_BAD_ARG, "rejectFactor out of range");
    if (c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "batches hold whole images");
    if (c->kernelVersion != 2) return yk_fail(c, YK_ERR_STATE, "batches need kernel version 2");
    if (c->nFrames > 1 && c->fs.plane == 0) return yk_fail(c, YK_ERR_STATE, "yk_bind_device_batch first");
    YK_HIP(c, hipSetDevice(c->device));
    hipEvent_t* ev = c->evRing[c->evHead % YK_EV_RING];
    YK_HIP(c, hipEventRecord(ev[0], c->stream));
    int rc = YK_OK;
    if (c->nPlanes == 4) rc = yk_launch_alpha(c, true);                     // every frame's bbox kernel publishes its own bounds[0..4]
    if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[1], c->stream));
    if (c->fusedAfter) { YK_HIP(c, hipStreamWaitEvent(c->stream, c->fusedAfter, 0)); c->fusedAfter = nullptr; }   // yk_order_fused_after
    YK_HIP(c, hipEventRecord(ev[2], c->stream));
    rc = yk_launch_encode(c, rejectFactor, mode3BitOnly, 0, true); if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[3], c->stream));
    rc = yk_launch_pack(c, true); if (rc) return rc;
    YK_HIP(c, hipEventRecord(ev[4], c->stream));
    c->evAlphaInCur = false; c->evHead++;
    if (c->evHead - c->evTail > YK_EV_RING) c->evTail = c->evHead - YK_EV_RING;
    c->alphaDone = c->nPlanes == 4; c->alphaFinished = true;
    c->encoded = true; c->dstValid = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false; c->nextCornerPass = 0; c->r1Ready = false;
    return YK_OK;
}

int yk_encode_frame(yk_ctx* c, int rejectFactor, int mode3BitOnly) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->plane[0]) return yk_fail(c, YK_ERR_STATE, "bind planes first");
    if (rejectFactor < 0 || rejectFactor > 64) return yk_fail(c, YK_ERR_BAD_ARG, "rejectFactor out of range");
    if (c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "yk_encode_frame is the whole-image form (stripes need the bbox exchange between the stages)");
    YK_HIP(c, hipSetDevice(c->device));
    const unsigned long long key[12] = { (unsigned long long)(uintptr_t)c->plane[0], (unsigned long long)(uintptr_t)c->plane[1], (unsigned long long)(uintptr_t)c->plane[2],
                                         (unsigned long long)(uintptr_t)c->plane[3], (unsigned long long)c->fullW, (unsigned long long)c->fullH, (unsigned long long)c->strideElems,
                                         (unsigned long long)rejectFactor, (unsigned long long)(mode3BitOnly != 0), (unsigned long long)c->kernelVersion,
                                         (unsigned long long)c->ablate, (unsigned long long)(uintptr_t)c->stream };
    if (c->frameGraph && memcmp(key, c->frameGraphKey, sizeof key) != 0) { (void)hipGraphExecDestroy(c->frameGraph); c->frameGraph = nullptr; }
    if (!c->frameGraph) {
        hipGraph_t g = nullptr;
        YK_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = YK_OK;
        if (c->nPlanes == 4) rc = yk_launch_alpha(c);                       // bounds[0..4] are published by the bbox kernel (whole image)
        if (rc == YK_OK) rc = yk_launch_encode(c, rejectFactor, mode3BitOnly, 0);
        if (rc == YK_OK) rc = yk_launch_pack(c);
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != YK_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) return yk_fail(c, YK_ERR_HIP, "hipStreamEndCapture", e);
        const hipError_t e2 = hipGraphInstantiate(&c->frameGraph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { c->frameGraph = nullptr; return yk_fail(c, YK_ERR_HIP, "hipGraphInstantiate", e2); }
        memcpy(c->frameGraphKey, key, sizeof key);
    }
    hipEvent_t* ev = c->evRing[c->evHead % YK_EV_RING];                     // one interval for the whole frame (reported as "encode")
    if (c->fusedAfter) { YK_HIP(c, hipStreamWaitEvent(c->stream, c->fusedAfter, 0)); c->fusedAfter = nullptr; }   // yk_order_fused_after: the whole replay waits
    YK_HIP(c, hipEventRecord(ev[0], c->stream)); YK_HIP(c, hipEventRecord(ev[1], c->stream)); YK_HIP(c, hipEventRecord(ev[2], c->stream));
    YK_HIP(c, hipGraphLaunch(c->frameGraph, c->stream));
    YK_HIP(c, hipEventRecord(ev[3], c->stream)); YK_HIP(c, hipEventRecord(ev[4], c->stream));
    c->evAlphaInCur = false; c->evHead++;
    if (c->evHead - c->evTail > YK_EV_RING) c->evTail = c->evHead - YK_EV_RING;
    c->alphaDone = c->nPlanes == 4; c->alphaFinished = true;
    c->encoded = true; c->dstValid = false; c->cornersReady = false; c->ppActive = false; c->ppLastBit = 0; c->previewFresh = false; c->nextCornerPass = 0; c->r1Ready = false;
    return YK_OK;
}

size_t yk_gradient_bitmap_bytes(const yk_ctx* c, int pass) { return (c && pass >= 0 && pass < 7) ? c->bitmapBytes[pass] : 0; }
const uint8_t* yk_gradient_bitmap_device(const yk_ctx* c, int pass) { return (c && c->encoded && pass >= 0 && pass < 7) ? c->bitmap[pass] : nullptr; }

int yk_gradient_bitmap(yk_ctx* c, int pass, uint8_t* hostOut, size_t cap) {
    if (!c || !hostOut || pass < 0 || pass >= 7) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    if (cap < c->bitmapBytes[pass]) return yk_fail(c, YK_ERR_RANGE, "bitmap buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->bitmap[pass], c->bitmapBytes[pass], hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_gradient_counts(yk_ctx* c, int32_t counts[YK_NUM_PASSES]) {
    if (!c || !counts) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    for (int p = 0; p < 7; p++) {
        std::vector<uint8_t> b(c->bitmapBytes[p] + 8, 0);
        YK_HIP(c, hipMemcpyAsync(b.data(), c->bitmap[p], c->bitmapBytes[p], hipMemcpyDeviceToHost, c->stream));
        YK_HIP(c, hipStreamSynchronize(c->stream));
        int n = 0;
        for (size_t i = 0; i < c->bitmapBytes[p]; i++) n += __builtin_popcount(b[i]);
        counts[p] = n;
    }
    return YK_OK;
}

int yk_coverage(yk_ctx* c, uint16_t* hostOut, size_t capElems) {
    if (!c || !hostOut) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    const size_t n = (size_t)c->mtW * c->mtH;
    if (capElems < n) return yk_fail(c, YK_ERR_RANGE, "coverage buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->coverage, n * 2, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_range_sizes(yk_ctx* c, int plane, size_t* nDefs, size_t* nNibbles) {
    if (!c || plane < 0 || plane > 2) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    uint32_t t[2];
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(t, c->totals + plane * 2, sizeof t, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    if (nDefs) *nDefs = t[0];
    if (nNibbles) *nNibbles = t[1];
    return YK_OK;
}

int yk_range_streams(yk_ctx* c, int plane, uint16_t* hostDefs, size_t capDefs, uint8_t* hostNibbles, size_t capBytes) {
    size_t nd = 0, nn = 0;
    int rc = yk_range_sizes(c, plane, &nd, &nn); if (rc) return rc;
    const size_t nb = (nn + 1) / 2;                                       // closed to a whole byte (:4525-4527)
    if ((hostDefs && capDefs < nd) || (hostNibbles && capBytes < nb)) return yk_fail(c, YK_ERR_RANGE, "range stream buffer too small");
    const size_t T8 = (size_t)c->tilesW * c->tilesH;
    if (hostDefs && nd) YK_HIP(c, hipMemcpyAsync(hostDefs, c->defsOut + plane * T8, nd * 2, hipMemcpyDeviceToHost, c->stream));
    if (hostNibbles && nb) YK_HIP(c, hipMemcpyAsync(hostNibbles, c->nibOut + plane * c->nibStride, nb, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

const uint16_t* yk_range_defs_device(const yk_ctx* c, int plane) {
    return (c && c->encoded && plane >= 0 && plane < 3) ? c->defsOut + (size_t)plane * c->tilesW * c->tilesH : nullptr;
}
const uint8_t* yk_range_nibbles_device(const yk_ctx* c, int plane) {
    return (c && c->encoded && plane >= 0 && plane < 3) ? c->nibOut + (size_t)plane * c->nibStride : nullptr;
}

int yk_range_dst(yk_ctx* c, int plane, int32_t* hostOut, size_t capElems) {
    if (!c || !hostOut || plane < 0 || plane > 2) return YK_ERR_BAD_ARG;
    if (!c->encoded || !c->dstValid) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles(wantDst=1) first");
    const size_t n = (size_t)c->fullW * c->h;
    if (capElems < n) return yk_fail(c, YK_ERR_RANGE, "dst buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->dst[plane], n * 4, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

size_t yk_export_capacity(const yk_ctx* c) {
    if (!c || !c->tileCount) return 0;
    size_t n = 0;
    for (int i = 0; i < 7; i++) n += (c->bitmapBytes[i] + 15) & ~(size_t)15;
    n += ((size_t)c->mtW * c->mtH + 15) & ~(size_t)15;
    const size_t T8 = (size_t)c->tilesW * c->tilesH;
    n += 3 * (((T8 * 2 + 15) & ~(size_t)15) + ((T8 * YK_SLOT + 15) & ~(size_t)15));
    return n;
}

// One launch packs all sections: blockIdx.y = section.  The lengths of the six stream sections live on the device (scan totals),
// so every workgroup derives its section's offset itself and workgroup (0,0) publishes the 15 sizes for the host.
struct YkExportDesc {
    const uint8_t* src[14];
    unsigned long long fixedBytes[8];       // 7 bitmaps + keep flags
    const uint32_t* totals;                 // [3][2]: coded tiles, nibbles per plane
    unsigned long long* sizesOut;           // [16]: total payload bytes, then the 15 section sizes
};

__global__ __launch_bounds__(256) void yk_export_kernel(YkExportDesc d, uint8_t* __restrict__ dst) {
    const int sec = blockIdx.y;
    unsigned long long n[14], off = 0, mine = 0, myOff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) n[i] = d.fixedBytes[i];
#pragma unroll
    for (int p = 0; p < 3; p++) { n[8 + 2 * p] = (unsigned long long)d.totals[p * 2] * 2; n[9 + 2 * p] = ((unsigned long long)d.totals[p * 2 + 1] + 1) / 2; }
#pragma unroll
    for (int i = 0; i < 14; i++) { if (i == sec) { mine = n[i]; myOff = off; } off += (n[i] + 15) & ~15ULL; }
    if (sec == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        d.sizesOut[0] = off;
        for (int i = 0; i < 8; i++) d.sizesOut[1 + i] = n[i];
        for (int p = 0; p < 3; p++) { d.sizesOut[9 + 2 * p] = d.totals[p * 2]; d.sizesOut[10 + 2 * p] = d.totals[p * 2 + 1]; }
        d.sizesOut[15] = off;
    }
    const uint8_t* src = d.src[sec];
    uint8_t* o = dst + myOff;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // section offsets are multiples of 16; a source is too unless the tile grid is tiny and odd (defs of plane 1, 2)
    unsigned long long nVec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 ? (mine >> 4) : 0;
    for (size_t i = t0; i < nVec; i += stride) reinterpret_cast<uint4*>(o)[i] = reinterpret_cast<const uint4*>(src)[i];
    const unsigned long long padded = (mine + 15) & ~15ULL;
    for (unsigned long long i = (nVec << 4) + t0; i < padded; i += stride) o[i] = i < mine ? src[i] : 0;   // unaligned source / tail / zero padding
}

static int yk_export_launch(yk_ctx* c, void* devDst, size_t cap, unsigned long long* devMeta16) {
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    if (cap < yk_export_capacity(c)) return yk_fail(c, YK_ERR_RANGE, "export buffer smaller than yk_export_capacity");
    YK_HIP(c, hipSetDevice(c->device));
    const size_t T8 = (size_t)c->tilesW * c->tilesH;
    YkExportDesc d;
    for (int i = 0; i < 7; i++) { d.src[i] = c->bitmap[i]; d.fixedBytes[i] = c->bitmapBytes[i]; }
    d.src[7] = c->keep; d.fixedBytes[7] = (c->nPlanes == 4) ? (unsigned long long)c->mtW * c->mtH : 0;
    for (int p = 0; p < 3; p++) { d.src[8 + 2 * p] = reinterpret_cast<const uint8_t*>(c->defsOut + p * T8); d.src[9 + 2 * p] = c->nibOut + p * c->nibStride; }
    d.totals = c->totals; d.sizesOut = devMeta16;
    hipLaunchKernelGGL(yk_export_kernel, dim3(128, 14), dim3(256), 0, c->stream, d, (uint8_t*)devDst);
    YK_HIP(c, hipGetLastError());
    return YK_OK;
}

int yk_export_tile_maps(yk_ctx* c, void* devDst, size_t cap, uint64_t sizes[15]) {
    if (!c || !devDst || !sizes) return YK_ERR_BAD_ARG;
    if (!c->exportSizes) { YK_HIP(c, hipSetDevice(c->device)); YK_HIP(c, hipMalloc(&c->exportSizes, 16 * sizeof(unsigned long long))); }
    int rc = yk_export_launch(c, devDst, cap, c->exportSizes); if (rc) return rc;
    unsigned long long h[16];
    YK_HIP(c, hipMemcpyAsync(h, c->exportSizes, sizeof h, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));                 // the buffer is complete when this returns
    for (int i = 0; i < 15; i++) sizes[i] = h[1 + i];
    return YK_OK;
}

int yk_export_tile_maps_async(yk_ctx* c, void* devDst, size_t cap, void* devMeta16, void* consumerStream) {
    if (!c || !devDst || !devMeta16) return YK_ERR_BAD_ARG;
    int rc = yk_export_launch(c, devDst, cap, reinterpret_cast<unsigned long long*>(devMeta16)); if (rc) return rc;
    return yk_stream_handoff(c, consumerStream);
}

int yk_export_tile_maps_framed(yk_ctx* c, void* devDst, size_t cap, void* consumerStream) {
    if (!c || !devDst) return YK_ERR_BAD_ARG;
    if (cap < YK_EXPORT_HEADER_BYTES) return yk_fail(c, YK_ERR_RANGE, "export buffer smaller than its header");
    int rc = yk_export_launch(c, static_cast<uint8_t*>(devDst) + YK_EXPORT_HEADER_BYTES, cap - YK_EXPORT_HEADER_BYTES, reinterpret_cast<unsigned long long*>(devDst));
    if (rc) return rc;
    if (consumerStream == reinterpret_cast<void*>(-1)) return YK_OK;
    return yk_stream_handoff(c, consumerStream);
}

int yk_stream_handoff(yk_ctx* c, void* consumerStream) {
    if (!c) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->evHandoff) YK_HIP(c, hipEventCreateWithFlags(&c->evHandoff, hipEventDisableTiming));
    YK_HIP(c, hipEventRecord(c->evHandoff, c->stream));
    YK_HIP(c, hipStreamWaitEvent((hipStream_t)consumerStream, c->evHandoff, 0));
    return YK_OK;
}

int yk_stream_wait_for(yk_ctx* c, void* producerStream) {
    if (!c) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->evHandoff) YK_HIP(c, hipEventCreateWithFlags(&c->evHandoff, hipEventDisableTiming));
    YK_HIP(c, hipEventRecord(c->evHandoff, (hipStream_t)producerStream));
    YK_HIP(c, hipStreamWaitEvent(c->stream, c->evHandoff, 0));
    return YK_OK;
}

int yk_last_kernel_ms(yk_ctx* c, float* fusedEncodeMs, float* alphaMs, float* packMs) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->encoded || c->evHead == c->evTail) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipEventSynchronize(c->evRing[(c->evHead - 1) % YK_EV_RING][4]));
    double a = 0, e = 0, p = 0;
    const unsigned n = c->evHead - c->evTail;
    for (unsigned k = c->evTail; k != c->evHead; k++) {                         // average over the encodes since the last query
        hipEvent_t* ev = c->evRing[k % YK_EV_RING];
        float t = 0;
        YK_HIP(c, hipEventElapsedTime(&t, ev[0], ev[1])); a += t;
        YK_HIP(c, hipEventElapsedTime(&t, ev[2], ev[3])); e += t;
        YK_HIP(c, hipEventElapsedTime(&t, ev[3], ev[4])); p += t;
    }
    c->evTail = c->evHead;
    if (fusedEncodeMs) *fusedEncodeMs = (float)(e / n);
    if (alphaMs) *alphaMs = (float)(a / n);
    if (packMs) *packMs = (float)(p / n);
    return YK_OK;
}

}  // extern "C"
