// Internal header shared by the HIP translation units of libyaik_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include "../../include/yaik_hip.h"

#define YK_BLK      64      // pixels per workgroup block side (= the 64x64 swizzle block of include/YAIK_private.h:212)
#define YK_LSTRIDE  68      // LDS row pitch in 32-bit words (65 used, 16-byte aligned rows)
#define YK_LROWS    65
#define YK_EV_RING  64
#define YK_SLOT     32      // bytes of nibble slot per 8x8 tile-plane (64 nibbles)
#define YK_NUM_STAGES 7     // YK_STAGE_* of include/yaik_hip.h
#define YK_STAGE_RING 16

// Batches: one handle can hold nFrames images of one shape; every per-image array is allocated nFrames times back to back and
// the batch kernels address frame f at base + f * stride (elements of the array's own type).
struct YkFrameStrides {
    unsigned long long plane, keep, bitmap[7], coverage, tileDef, tileCount, slots, blockN, defsOut, nibOut;   // bounds: 16 ints, totals: 8 u32
    unsigned long long bm0b, tileInfo, runSums;   // the fused kernel's per-strip 16x16 bytes (u8), per-tile records (8 bytes each), run sums (u32)
};

struct YkEncodeParams {
    const int32_t* plane[4];
    int strideElems;
    int w, h;               // owned region (stripe): w = full width, h = owned rows
    int hAvail;             // rows physically present in the bound planes (h + halo)
    int y0;                 // first owned row in full-image coordinates
    int fullH;
    int rejectFactor, startMode, wantDst;
    int ablate;             // timing-only ablation switches (yk_set_ablation); 0 in every product run
    // alpha / bounds (device memory, written by the alpha kernels)
    const uint8_t* keep;    // per 16x16 macro-tile keep flag of this stripe, nullptr = no alpha plane
    const int32_t* bounds;  // [0..3] boundX0,Y0,X1,Y1 of the kept tiles (full-image pixels; yk_encode2_kernel derives the discard rule from them), [4] discardRejects for the first-generation kernel
    // outputs
    uint8_t* bitmap[7];
    uint16_t* coverage;     // per macro-tile, bit = cellY*4+cellX
    uint16_t* tileDef;      // [3][tilesW*tilesH]
    uint8_t*  tileCount;    // [3][tilesW*tilesH]
    uint8_t*  slots;        // [3][tilesW*tilesH][32]
    uint32_t* blockCnt;     // [ceil(tiles/1024)][2]: nibbles and coded tiles per scan block (atomics; nullptr = the scan counts itself)
    int32_t*  dst[3];
    int tilesW, tilesH, mtW, mtH;
    int xBB64, yBB64, xBB32, yBB32;
    int nFrames;            // 1 unless launched by yk_encode_batch
    const uint8_t* qtab;    // quantiser table of yk_encode2_kernel (yk_qtab_get)
    // yk_encode2_kernel's small outputs live in ONE allocation, so that lanes holding different outputs can share a store instruction (scalar base
    // + 32-bit lane offset): byte offsets, inside `small`, of frame 0 of the seven bitmaps (= bitmap[i]), of the strips' 16x16 bytes (4 bits each:
    // yk_scan2_kernel folds them into bitmap[0]), the coverage words (= coverage), the per-tile records {def0 | def1 << 16, def2 | count << 16}
    // and the per-run sums {nibbles / 16 | coded tiles << 16} (one per 8 consecutive tiles; widths that are multiples of 8 tiles only)
    uint8_t* small;
    uint32_t oBm[7], oBm0b, oCov, oInfo, oRun;
    // optional (yk_set_pixel_cache): the packed pixels (0x00BBGGRR) of every 4x4 cell the gradient passes left uncovered, for the live 1-D range
    // path behind the fused kernel, which then reads 4 bytes per such pixel instead of 12 from the planes: [strip (row-major)][cell row 0..3][lane]
    // 16-byte pieces, lane in this kernel's (Morton) order.  nullptr = not wanted.
    uint4* pixCache;
    YkFrameStrides fs;
};

// The layout of yk_ctx must NOT depend on YK_TEST_HOOKS: the product library and the test-hooks build of the same sources are loaded side by
// side by the tests, each owning the handles it created (yaik_amd/encoder.py keeps a handle with its library); no member below is conditional.
struct yk_ctx {
    int device = -1;
    int numCU = 256;             // compute units of the device (persistent grids are sized from it)
    const uint8_t* qtab = nullptr;   // per-device quantiser table (owned by the library, shared by all handles)
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // geometry
    int fullW = 0, fullH = 0, nPlanes = 0, y0 = 0, h = 0, halo = 0;
    int tilesW = 0, tilesH = 0, mtW = 0, mtH = 0;
    // batch: the pointers below are those of frame `curFrame`; B holds the allocations (frame 0), fs the per-frame strides
    int nFrames = 1, curFrame = 0;
    YkFrameStrides fs = {};
    struct Bases {
        const int32_t* plane[4]; uint8_t* keep; int32_t* bounds; uint8_t* bitmap[7]; uint16_t* coverage; uint16_t* tileDef; uint8_t* tileCount;
        uint8_t* slots; uint32_t* blockSums; uint32_t* blockCnt; uint32_t* totals; uint16_t* defsOut; uint8_t* nibOut;
        uint8_t* small;                 // one allocation: bitmap[0..6], bm0b, coverage, tileInfo, runSums (each nFrames times; the pointers above point into it)
        uint8_t* bm0b; uint2* tileInfo; uint32_t* runSums;
    } B = {};
    // input
    const int32_t* plane[4] = {nullptr, nullptr, nullptr, nullptr};
    int strideElems = 0;
    int32_t* ownedPlanes = nullptr; size_t ownedPlanesBytes = 0;
    // alpha
    uint8_t* keep = nullptr;            // mtW*mtH
    int32_t* bounds = nullptr;          // 16 ints: [0..4] the host-combined box of a striped image + its discard flag (yk_alpha_finish), [8..11] the box yk_alpha_kernel accumulates
    int* alphaUnitBox = nullptr;        // yk_alpha_kernel: one box {x0, y0, x1, y1} per work unit (tile row x 1024-pixel segment) and frame
    uint32_t* alphaArrive = nullptr;    // its arrival counters: per group of 64 units + one per frame (zero between launches)
    int boundsOff = 8;                  // where the image-wide box is: 8 (whole image, batch) or 0 (stripes, after yk_alpha_finish)
    bool alphaDone = false, alphaFinished = false;
    int32_t hostBounds[4] = {0, 0, 0, 0}; int hostDiscard = 1, hostHasChunk = 0;
    // encode outputs
    uint8_t* bitmap[7] = {}; size_t bitmapBytes[7] = {};
    uint16_t* coverage = nullptr;
    uint16_t* tileDef = nullptr; uint8_t* tileCount = nullptr; uint8_t* slots = nullptr;
    int32_t* dst[3] = {nullptr, nullptr, nullptr}; int32_t dstFill = -1; bool dstValid = false;
    // compaction
    uint32_t* blockSums = nullptr;      // [nBlocks][2]: exclusive prefix of (nibbles, coded tiles) per block of 1024 tiles, same for the 3 planes
    uint32_t* blockCnt = nullptr;       // [nBlocks][2]: the sums themselves, accumulated by the fused kernel, consumed (and cleared) by the scan
    uint32_t* totals = nullptr;         // [3][2] device
    unsigned long long* exportSizes = nullptr;   // [16] total + section sizes of the last yk_export_tile_maps
    hipEvent_t evHandoff = nullptr;              // yk_stream_handoff / yk_stream_wait_for
    hipEvent_t fusedAfter = nullptr;             // yk_order_fused_after: the next fused kernel waits for this event (= evFusedAfter, never another handle's)
    hipEvent_t evFusedAfter = nullptr; hipStream_t auxStream = nullptr;   // this handle's own: the event behind a wait for the other handle's fused kernel
    uint16_t* defsOut = nullptr;        // [3][T8]
    uint8_t*  nibOut = nullptr;         // [3][T8*32 + 8]
    size_t nibStride = 0;
    int nScanBlocks = 0;
    bool encoded = false;
    // corner streams
    uint32_t* latticeOwner = nullptr; size_t latticeElems = 0;
    uint8_t* cornerStream = nullptr; size_t cornerCap = 0;
    uint32_t* cornerScratch = nullptr; size_t cornerScratchElems = 0;
    uint32_t* cornerEdgeIdx = nullptr;  // [2][w/4+1]: emission index (in corners, within its pass) of the first / last lattice row
    bool cornersReady = false; int nextCornerPass = 0;
    size_t cornerOff[7] = {}, cornerBytes[7] = {};
    const uint32_t* cornerTotalsDev = nullptr; bool cornerTotalsPending = false;   // stream lengths still on the device (yk_corners_finish)
    const uint32_t* r1TotalsDev = nullptr; bool r1TotalsPending = false;           // the same for the 1-D path (yk_range1d_finish)
    // partial-plane gradient passes (FittingQuadSmooth with nullable planes): per-plane coverage (mapSmoothTile[p], u16 per 16x16 tile, bit = cell)
    // and per-plane "corner already emitted" flags per lattice point (mappedRGB[p]); allocated by the first partial pass after an encode
    uint16_t* covCh = nullptr; size_t covChStride = 0;     // [3][covChStride]
    uint8_t* mapped3 = nullptr;                            // bit p = plane p's corner at this lattice point has been emitted
    uint32_t* ppBitmap = nullptr; size_t ppBitmapBytes = 0; uint8_t* ppStream = nullptr; size_t ppStreamCap = 0, ppStreamBytes = 0;
    uint32_t* ppScratch = nullptr; size_t ppScratchElems = 0, ppBitmapCap = 0; int ppAccepted = 0; bool ppActive = false;
    int ppLastBit = 0, ppLastSx = 0, ppLastSy = 0;         // the last plane-subset pass (its bitmap is ppBitmap)
    int32_t* preview = nullptr; bool previewFresh = false; // FittingQuadSmooth's testOutput planes (3 x w*h int32), INT32_MIN where no tile wrote
    // (f)4 3-D LUT tiles (yk_lut3d.hip): pattern bank + the streams StartCorrelationSearch allocates
    struct YkLutState* lut = nullptr;
    struct YkLutDecState* lutDec = nullptr;      // decoder: the per-orientation tables YAIK_AssignLUT lays out
    // live 1-D range path (a15)
    uint8_t* r1Slots = nullptr; uint8_t* r1Params = nullptr; uint32_t* r1Cnt = nullptr; uint8_t* r1Pix = nullptr; uint8_t* r1Type = nullptr;
    uint32_t r1Tiles = 0, r1PixCount = 0; bool r1Ready = false;
    uint4* pixCache = nullptr; bool pixCacheOn = false, pixCacheValid = false;   // yk_set_pixel_cache: uncovered cells' packed pixels, fused kernel -> 1-D path
    uint32_t r1EndTiles[3] = {}, r1EndPix[3] = {};       // cumulative per plane (equal thirds unless a partial-plane pass ran)
    // decode
    int dw = 0, dh = 0; uint8_t* dPlanes = nullptr; size_t dPlaneSize = 0;
    uint8_t* dMapRGB = nullptr; uint32_t* dLatticeOwner = nullptr; uint8_t* dTile4 = nullptr; size_t dTile4Size = 0;
    uint8_t* dScratch = nullptr; size_t dScratchBytes = 0;
    uint8_t* dLoaded = nullptr;         // lattice point already popped from a colour stream (mapRGBMask)
    bool dSplit = false;
    bool dPlanesStale = false;          // the planes were not cleared for this image: cells tile4x4Mask does not mark hold the previous image (yk_dec_settle)
    // timing
    // timing events: a ring of YK_EV_RING sets {alpha begin, alpha end, encode begin, encode end, pack end} so that a caller can
    // run many frames back to back and read the per-kernel averages afterwards without synchronising every frame
    hipEvent_t evRing[YK_EV_RING][5] = {};
    unsigned evHead = 0, evTail = 0;      // sets [evTail, evHead) hold a completed encode; evCur = evHead % YK_EV_RING is being filled
    bool evAlphaInCur = false;
    float msEncode = 0, msAlpha = 0, msPack = 0;
    // stage timers (yk_stage_ms): HIP events around the kernel sections of the stages outside the fused encode, on the launch stream
    hipEvent_t stEv[YK_NUM_STAGES][YK_STAGE_RING][2] = {};
    int stN[YK_NUM_STAGES] = {};                 // event pairs recorded since the last fold
    double stAcc[YK_NUM_STAGES] = {}; int stCalls[YK_NUM_STAGES] = {};
    int ablate = 0;
    // whole-frame graph (yk_encode_frame): the stream operations of alpha stage + fused kernel + compaction, captured once per
    // (planes, shape, arguments) and replayed with one launch — for batches of small frames, where launches dominate
    hipGraphExec_t frameGraph = nullptr;
    unsigned long long frameGraphKey[12] = {};
    int kernelVersion = 2;              // 2 = yk_encode2_kernel; 1 = the registered cross-check launcher (tests/csrc/yk_encode_v1.hip)
};

int yk_fail(yk_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
#define YK_HIP(c, call) do { hipError_t _e = (call); if (_e != hipSuccess) return yk_fail((c), YK_ERR_HIP, #call, _e); } while (0)

// launchers implemented in the kernel TUs
void yk_rebase(yk_ctx* c, int frame);                    // point the working pointers at `frame`
int yk_stage_begin(yk_ctx* c, int stage);                // records the begin event of a new interval of `stage` on c->stream
int yk_stage_end(yk_ctx* c, int stage);
int yk_launch_alpha(yk_ctx* c, bool batch = false);
int yk_launch_alpha_finish(yk_ctx* c, const int32_t* globalBBox);
int yk_launch_encode(yk_ctx* c, int rejectFactor, int mode3BitOnly, int wantDst, bool batch = false);
int yk_launch_pack(yk_ctx* c, bool batch = false);
int yk_launch_corners(yk_ctx* c);
int yk_corners_finish(yk_ctx* c);                         // reads the corner streams' lengths back if that is still pending (synchronises)
void yk_lut_dec_destroy(yk_ctx* c);
void yk_lut_destroy(yk_ctx* c);                          // frees the 3-D LUT bank and streams (yk_lut3d.hip)
int yk_pp_activate(yk_ctx* c);                           // per-plane coverage / corner flags for the passes behind the RGB passes
int yk_launch_encode2(yk_ctx* c, const YkEncodeParams& P);
int yk_qtab_get(yk_ctx* c);                              // builds the device's quantiser table on first use, sets c->qtab
void yk_selftest_qtab_launch(yk_ctx* c, int* mismatches);
