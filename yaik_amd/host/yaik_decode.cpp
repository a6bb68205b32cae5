// See yaik_decode.h.  Control flow mirrors decoder/YAIK_API.cpp (Init :86-131, Pre :441-497, DecodeImage :600-1342): slot
// stack, first-error-wins sticky code, chunk state machine, "Pre must be followed by Decode".  The per-pixel work is done by
// the HIP kernels behind yk_decode_*; ZStd and PaletteDecompressor run on the host like in the reference.
#include "yaik_decode.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/yaik_hip.h"
#include "palette.h"
#include "yaik_format.h"
#include "zstd_dl.h"

using namespace yaikfmt;

namespace {
std::atomic<int> gError{YAIK_NO_ERROR};
void setError(YAIK_ERROR_CODE e) { int expected = YAIK_NO_ERROR; gError.compare_exchange_strong(expected, (int)e); }   // first error wins (:77-84)

struct Slot {
    yk_ctx* ctx = nullptr;
    void* srcCheck = nullptr; uint32_t srcLength = 0;
    int width = 0, height = 0; bool isRGBA = false;
};
struct Library {
    std::vector<Slot> slots;
    std::vector<Slot*> freeStack;
    std::mutex lock;                                    // the reference's stack is unguarded although Pre/Decode are documented thread-safe
    YAIK_SMemAlloc alloc;
    bool hasLut = false;                                // YAIK_AssignLUT handed a valid 3-D LUT file to every decode slot
};
Library* gLib = nullptr;
int gDevice = 0;

void* defaultAlloc(void*, size_t n) { return malloc(n); }
void  defaultFree(void*, void* p) { free(p); }

void defaultImageBuilder(YAIK_SDecodedImage*, YAIK_SCustomDataSource*) {}      // marker: the de-tile kernel writes outputImage directly

bool zexpand(const uint8_t* src, uint32_t n, uint32_t expected, std::vector<uint8_t>& out, size_t slack) {
    out.assign((size_t)expected + slack, 0);
    if (n == 0) return false;                                                   // DecompressData returns NULL for an empty stream (:517-519)
    if (!yaikzstd::decompress(out.data(), expected, src, n)) { setError(YAIK_INVALID_DECOMPRESSION); return false; }
    return true;
}
}

void YAIK_SetDevice(int device) { gDevice = device; }
static int gConsistentMarks = 0;
void YAIK_SetPartialPlaneMarks(int consistent) { gConsistentMarks = consistent ? 1 : 0; }

YAIK_LIB YAIK_Init(uint8_t maxDecodeThreadContext, YAIK_SMemAlloc* libraryMemAllocator) {
    if (maxDecodeThreadContext == 0) { setError(YAIK_INVALID_CONTEXT_COUNT); return nullptr; }
    if (gLib) { setError(YAIK_INIT_FAIL); return nullptr; }                     // one library per process, like the reference's gLibrary
    if (!yaikzstd::available()) { setError(YAIK_DECOMPRESSION_CREATE_FAIL); return nullptr; }
    Library* L = new Library();
    if (libraryMemAllocator) {
        if (!libraryMemAllocator->customAlloc || !libraryMemAllocator->customFree) { delete L; setError(YAIK_INVALID_CONTEXT_MEMALLOCATOR); return nullptr; }
        L->alloc = *libraryMemAllocator;
    }
    L->slots.resize(maxDecodeThreadContext);
    for (auto& s : L->slots) {
        if (yk_create(gDevice, &s.ctx) != YK_OK) {                              // no HIP device: refuse, there is no CPU decode path here
            for (auto& t : L->slots) if (t.ctx) yk_destroy(t.ctx);
            delete L; setError(YAIK_INIT_FAIL); return nullptr;
        }
        L->freeStack.push_back(&s);
    }
    gLib = L;
    return L;
}

// decoder/YAIK_API.cpp:133-415: only the 3-D table file ('LUL') is accepted ('LU2' is deprecated there too); the per-orientation tables are
// laid out in HBM by every decode slot's handle (yk_decode_assign_lut)
void YAIK_AssignLUT(YAIK_LIB lib, uint8_t* lutData, uint32_t lutDataLength) {
    if (!lib || lib != gLib) { setError(YAIK_INVALID_LIBRARYCTX); return; }
    if (!lutData || lutDataLength < sizeof(LUTHeader) || lutData[0] != 'L' || lutData[1] != 'U' || lutData[2] != 'L') { setError(YAIK_INVALID_LUT); return; }
    const uint32_t expected = ((uint32_t)lutData[5] + 1) * 3 * (64 + 32 + 16 + 8);
    if (expected != lutDataLength - sizeof(LUTHeader)) { setError(YAIK_INVALID_LUT); return; }
    for (auto& s : gLib->slots)
        if (yk_decode_assign_lut(s.ctx, lutData, lutDataLength) != YK_OK) { setError(YAIK_MALLOC_FAIL); return; }
    gLib->hasLut = true;
}

void YAIK_Release(YAIK_LIB lib) {
    if (!lib || lib != gLib) { setError(YAIK_RELEASE_EMPTY_LIBRARY); return; }
    for (auto& s : gLib->slots) if (s.ctx) yk_destroy(s.ctx);
    delete gLib; gLib = nullptr;
}

YAIK_ERROR_CODE YAIK_GetErrorCode() { return (YAIK_ERROR_CODE)gError.exchange(YAIK_NO_ERROR); }

bool YAIK_DecodeImagePre(YAIK_LIB lib, void* stream, uint32_t length, YAIK_SDecodedImage* info) {
    if (!info) { setError(YAIK_DECIMG_INVALIDCTX); return false; }
    info->hasAlpha = false; info->hasAlpha1Bit = false; info->outputImage = nullptr; info->width = 0; info->height = 0; info->internalTag = nullptr;
    if (!lib || lib != gLib) { setError(YAIK_INVALID_LIBRARYCTX); return false; }
    const FileHeader* h = (const FileHeader*)stream;
    if (!h || length <= sizeof(FileHeader)) { setError(YAIK_INVALID_STREAM); return false; }
    if (h->tag != TAG_FILE || (h->width & 15) || (h->height & 15) || h->width == 0 || h->height == 0) { setError(YAIK_INVALID_HEADER); return false; }
    Slot* s = nullptr;
    { std::lock_guard<std::mutex> g(gLib->lock); if (!gLib->freeStack.empty()) { s = gLib->freeStack.back(); gLib->freeStack.pop_back(); } }
    if (!s) { setError(YAIK_NO_EMPTYDECODE_SLOT); return false; }
    s->srcCheck = stream; s->srcLength = length; s->width = h->width; s->height = h->height; s->isRGBA = (h->infoMask & 1) != 0;
    info->width = h->width; info->height = h->height; info->hasAlpha = s->isRGBA; info->internalTag = s;
    info->customImageOutput = defaultImageBuilder; info->userContextCustomImage = nullptr;
    info->userMemoryAllocator.customAlloc = defaultAlloc; info->userMemoryAllocator.customFree = defaultFree; info->userMemoryAllocator.customContext = nullptr;
    return true;
}

bool YAIK_DecodeImage(void* stream, uint32_t length, YAIK_SDecodedImage* info) {
    if (!info || !info->internalTag || !gLib) { setError(YAIK_DECIMG_INVALIDCTX); return false; }
    Slot* s = (Slot*)info->internalTag;
    bool res = false;
    std::vector<uint8_t> bitmap, pal, rgb, types, pix;
    const YAIK_SMemAlloc& ua = info->userMemoryAllocator;
    do {
        if (!ua.customAlloc || !ua.customFree) { setError(YAIK_INVALID_CONTEXT_MEMALLOCATOR); break; }
        if (s->srcCheck != stream || s->srcLength != length) { setError(YAIK_DECIMG_DIFFSTREAM); break; }
        if (!info->outputImage) { setError(YAIK_DECIMG_BUFFERNOTSET); break; }
        const int w = s->width, h = s->height;
        if (yk_decode_begin(s->ctx, w, h) != YK_OK) { setError(YAIK_MALLOC_FAIL); break; }
        const uint8_t* p = (const uint8_t*)stream + sizeof(FileHeader);
        const uint8_t* const end = (const uint8_t*)stream + length;
        int state = 0; bool bad = false;
        while (!bad) {
            if (p + 4 > end) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
            HeaderBase hb; memcpy(&hb.tag, p, 4);
            if (hb.tag == TAG_END) break;
            if (p + sizeof(HeaderBase) > end) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
            memcpy(&hb, p, sizeof hb);
            const uint8_t* body = p + sizeof(HeaderBase);
            const uint8_t* endBlock = body + hb.length;
            if (endBlock > end || endBlock < body) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }     // never read past the stream (:724-728)
            switch (hb.tag) {
            case TAG_MIPMAP: {
                if (state != 0 || hb.length < sizeof(MipmapHeader)) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                MipmapHeader mh; memcpy(&mh, body, sizeof mh);
                if (mh.mipmapLevel != 4) { setError(YAIK_INVALID_MIPMAP_LEVEL); bad = true; break; }         // only level 4 is implemented (YAIK_Mipmap.cpp:53-54)
                const size_t need = ((size_t)mh.bbox.w * mh.bbox.h + 7) / 8;
                if (mh.bbox.w <= 0 || mh.bbox.h <= 0 || body + sizeof mh + need > endBlock) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                std::vector<uint8_t> mask((size_t)w * h / 8 + 64);
                if (yk_decode_mask(s->ctx, body + sizeof mh, mh.bbox.w, mh.bbox.h, mask.data(), mask.size()) != YK_OK) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                state = 1; break;
            }
            case TAG_GRADTILE: {
                if (state > 4) break;                                                                       // silently skipped after '1DTL' (:836)
                if (hb.length < sizeof(HeaderGradientTile)) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                state = 4;
                HeaderGradientTile gh; memcpy(&gh, body, sizeof gh);
                const int sx = gh.format & 7, sy = (gh.format >> 3) & 7;
                u32 bigX, bigY, bitCount;
                if (!swizzleSize(sx, sy, bigX, bigY, bitCount) || gh.plane < 1 || gh.plane > 7) { setError(YAIK_INVALID_PLANE_ID); bad = true; break; }
                const uint8_t* after = body + sizeof gh;
                if (after + (size_t)gh.streamBitmapSize + gh.streamRGBSizeZStd > endBlock) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                const uint32_t sizeBitmap = (uint32_t)(((w + bigX - 1) / bigX) * ((h + bigY - 1) / bigY) * bitCount / 8);
                if (!zexpand(after, gh.streamBitmapSize, sizeBitmap, bitmap, 0)) { bad = true; break; }
                if (!zexpand(after + gh.streamBitmapSize, gh.streamRGBSizeZStd, gh.streamRGBSizeCustomCompressor, pal, 128 * 3)) { bad = true; break; }
                const size_t slack = (size_t)((w + 3) >> 2) * ((h + 3) >> 2) * 12;                          // the reference's "secure buffer" (:901)
                rgb.assign((size_t)gh.streamRGBSizeUncompressed + slack, 0);
                if (!PaletteDecompressor(pal.data(), (int)gh.streamRGBSizeCustomCompressor, (int)gh.streamRGBSizeCustomCompressor + 128 * 3, rgb.data(),
                                         (int)gh.streamRGBSizeUncompressed, gh.colorCompression)) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                if (gh.plane != 7) {
                    // a chunk for one or two planes splits the masks per plane first (UpdateTileAndRGBMask, YAIK_API.cpp:876-878); only the 4x4
                    // decoders exist for such chunks, every other tile shape returns without touching anything (YAIK_Gradient.cpp:29-36)
                    if (yk_decode_split_masks(s->ctx) != YK_OK) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                    if (sx == 2 && sy == 2 && yk_decode_gradient_planes(s->ctx, gh.plane, gConsistentMarks, bitmap.data(), sizeBitmap, rgb.data(),
                                                                       gh.streamRGBSizeUncompressed) != YK_OK) { setError(YAIK_INVALID_STREAM); bad = true; }
                    break;
                }
                if (yk_decode_gradient(s->ctx, sx, sy, bitmap.data(), sizeBitmap, rgb.data(), gh.streamRGBSizeUncompressed) != YK_OK) { setError(YAIK_INVALID_STREAM); bad = true; }
                break;
            }
            case TAG_TILE1D: {
                if (state < 4) break;                                                                       // ignored before any gradient chunk (:962)
                if (hb.length < sizeof(Header1D)) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                state = 5;
                Header1D dh; memcpy(&dh, body, sizeof dh);
                const uint8_t* zt = body + sizeof dh; const uint8_t* zp = zt + dh.streamTypeCnt;
                if (zp + dh.streamPixelBit > endBlock) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                if (!zexpand(zt, dh.streamTypeCnt, dh.streamTypeUncmp, types, 64) || !zexpand(zp, dh.streamPixelBit, dh.streamPixelUncmp, pix, 64)) { bad = true; break; }
                if (yk_decode_1d(s->ctx, types.data(), dh.streamTypeUncmp, pix.data(), dh.streamPixelUncmp, dh.compressionRange) != YK_OK) { setError(YAIK_INVALID_STREAM); bad = true; }
                break;
            }
            case 0x4d504c41u: setError(YAIK_ALPHA_UNSUPPORTED_YET); bad = true; break;                      // 'ALPM': alpha value coder, off the path
            case TAG_TILE3D: {                                                                             // decoder/YAIK_API.cpp:999-1300
                if (state > 4) break;
                state = 4;
                if (hb.length < sizeof(HeaderTile3D)) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                if (!gLib->hasLut) { setError(YAIK_INVALID_LUT); bad = true; break; }                       // the reference would read a NULL table here
                HeaderTile3D th; memcpy(&th, body, sizeof th);
                // The counts come from an untrusted stream: all size arithmetic in 64 bits, and nothing is expanded before the counts are
                // plausible (a tile is at least 4x4 pixels; six colour bytes per tile; the reference leaves both checks a TODO, :1079).
                const uint64_t maxTiles = (uint64_t)(w / 4) * (uint64_t)(h / 4);
                if ((uint64_t)th.streamTypeCnt > maxTiles || (uint64_t)th.streamColorCnt != 6ull * (uint64_t)th.streamTypeCnt) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                const uint8_t* q = body + sizeof th;
                const uint32_t cmp[12] = { th.compr3BitSize, th.compr4BitSize, th.compr5BitSize, th.compr6BitSize, th.comprTypeSize, th.comprColorSize,
                                           th.sizeT16_8MapCmp, th.sizeT8_16MapCmp, th.sizeT8_8MapCmp, th.sizeT8_4MapCmp, th.sizeT4_8MapCmp, th.sizeT4_4MapCmp };
                const uint32_t raw[12] = { th.stream3BitCnt, th.stream4BitCnt, th.stream5BitCnt, th.stream6BitCnt, th.streamTypeCnt * 2, th.streamColorCnt,
                                           th.sizeT16_8Map, th.sizeT8_16Map, th.sizeT8_8Map, th.sizeT8_4Map, th.sizeT4_8Map, th.sizeT4_4Map };
                std::vector<uint8_t> part[12];
                const uint64_t maxStream = (uint64_t)w * (uint64_t)h * 3 + 4096;                            // no stream of an image can be longer than its pixels
                for (int k = 0; k < 12 && !bad; k++) if ((uint64_t)raw[k] > maxStream) { setError(YAIK_INVALID_STREAM); bad = true; }
                for (int k = 0; k < 12 && !bad; k++) {
                    if ((uint64_t)cmp[k] > (uint64_t)(endBlock - q)) { setError(YAIK_INVALID_TAG_ID); bad = true; break; }
                    if (raw[k] && !zexpand(q, cmp[k], raw[k], part[k], 256)) { bad = true; break; }
                    q += cmp[k];
                }
                if (bad) break;
                // what yk_decode_lut3d may read is what was really expanded, not what the header claims
                if (part[4].size() < (size_t)th.streamTypeCnt * 2 || part[5].size() < (size_t)th.streamTypeCnt * 6) { setError(YAIK_INVALID_STREAM); bad = true; break; }
                if (th.streamColorCnt) PaletteFullRangeRemapping(part[5].data(), (int)th.streamColorCnt, th.compressionRateColor);
                const uint8_t* maps[6]; size_t mapBytes[6]; const uint8_t* idx[4]; size_t idxBytes[4]; size_t used[6];
                for (int k = 0; k < 6; k++) { maps[k] = raw[6 + k] ? part[6 + k].data() : nullptr; mapBytes[k] = raw[6 + k]; }
                for (int k = 0; k < 4; k++) { idx[k] = raw[k] ? part[k].data() : nullptr; idxBytes[k] = raw[k]; }
                if (yk_decode_lut3d(s->ctx, maps, mapBytes, reinterpret_cast<const uint16_t*>(part[4].data()), th.streamTypeCnt, part[5].data(), idx, idxBytes, used) != YK_OK) {
                    setError(YAIK_INVALID_STREAM); bad = true;
                }
                break;
            }
            default: setError(YAIK_INVALID_TAG_ID); bad = true; break;
            }
            p = endBlock;
        }
        if (bad) break;
        if (info->customImageOutput && info->customImageOutput != defaultImageBuilder) {
            // custom builder: hand over the 8x8-tiled planes exactly like the reference (:1303-1318)
            const size_t planeSize = (size_t)(w / 8) * (h / 8) * 64;
            uint8_t* planes = (uint8_t*)ua.customAlloc(ua.customContext, planeSize * 3);
            if (!planes) { setError(YAIK_MALLOC_FAIL); break; }
            if (yk_decode_planes(s->ctx, planes, planes + planeSize, planes + 2 * planeSize, planeSize) == YK_OK) {
                YAIK_SCustomDataSource src;
                src.planeR = planes; src.planeG = planes + planeSize; src.planeB = planes + 2 * planeSize; src.planeA = nullptr;
                src.strideR = src.strideG = src.strideB = (w / 8) * 64; src.strideA = w;
                info->customImageOutput(info, &src);
                res = true;
            } else setError(YAIK_INVALID_STREAM);
            ua.customFree(ua.customContext, planes);
        } else {
            // default builder = the de-tile kernel.  The alpha value chunk ('ALPM') is not on this path, so like the reference's
            // pCtx->alphaChannel the alpha plane is NULL and internal_imageBuilderFunc takes its RGB branch (3 B/pixel) whether or
            // not the header says RGBA (YAIK_API.cpp:1316, YAIK_DefaultCallback.cpp:44,63); row padding is left untouched.
            res = yk_decode_output(s->ctx, info->outputImage, (size_t)info->outputImageStride, nullptr, w) == YK_OK;
            if (!res) setError(YAIK_INVALID_STREAM);
        }
    } while (false);
    { std::lock_guard<std::mutex> g(gLib->lock); gLib->freeStack.push_back(s); }     // the slot is released whatever happened (:1322-1337)
    info->internalTag = nullptr;
    return res;
}
