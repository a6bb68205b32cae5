// TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into, imported by or executed from the product path.
//
// Harness around the UNMODIFIED reference sources, which are compiled where they lie under
// /root/reference by oracle/Makefile (outputs only into oracle/_ref/).  It feeds raw int32 planes to
// the reference's own per-tile passes and decode loops and dumps every intermediate the parity tests
// need as a flat list of named blobs:
//
//   EncoderContext::MipPrefilter          encoder/EncoderContext.cpp:1257
//   EncoderContext::FittingQuadSmooth x7  encoder/EncoderContext.cpp:3710 (pass order :9057-9093)
//   DynamicTileEncoderTable + EncoderContext::DynamicTileEncode   :702, :4365
//   EncoderContext::DynamicTileCompressor                         :8398
//   DecompressGradient16x16..4x4          decoder/YAIK_Gradient.cpp:28..1208
//   Decompress1D                          decoder/YAIK_3DTile.cpp:24
//   Decompress1BitTiled                   decoder/YAIK_Mipmap.cpp:23   (on the 'MIPM' chunk MipPrefilter itself wrote)
//   internal_imageBuilderFunc             decoder/YAIK_DefaultCallback.cpp:24 (RGB at a padded outputImageStride; RGBA as the reference does it)
//
// The reference frees its raw streams before returning, so they are captured where they cross a
// translation-unit boundary, with the GNU linker's --wrap of ZSTD_compress (raw tile bitmaps, tile
// definitions, nibble stream, PaletteCompressor output).  The corner-colour stream is recovered by
// running the reference's own PaletteDecompressor on that output, exactly as the decoder does, which
// yields the de-quantised bytes the gradient decode loops consume.  No reference code is copied or altered.
//
// usage: ref_driver <in.bin> <out.blobs> [partial | lut3d <bank.bin>]
//   lut3d   : (f)4.  After the seven RGB passes: Load3DPattern for every pattern of a SYNTHETIC bank (bank.bin = the patterns in the file
//             format Load3DPattern reads, EncoderContext.cpp:7851-7867: u8 count, count x r, count x g, count x b, 6-bit values; written
//             one file per pattern into the scratch directory), StartCorrelationSearch, Correlation3DSearch x6 in Convert()'s order
//             (:9139-9199), EndCorrelationSearch (the '3DTL' chunk); blobs lut_*.  The reference's own 22-file bank is not in its repo.
//   partial : after the seven RGB passes also run the six partial-plane 4x4 passes in the order the reference's Convert() lists them
//             (RB, RG, GB, R, G, B; EncoderContext.cpp:9261-9415, disabled there by `if (0)` / `#if 0`), before the 1-D path, and
//             decode them with DecompressGradient4x4(planeBit) (blobs pp_*)
//   in.bin  : int32 w, h, nPlanes, then nPlanes planes of w*h int32 (row-major, values 0..255)
#define protected public
#define private public
#include "EncoderContext.h"
#undef protected
#undef private
#include "YAIK_functions.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

// ---- reference symbols that no header declares (EncoderContext.cpp:702, :8217-8218)
void DynamicTileEncoderTable();
int BitmapSwizzleMapSize(int TshiftX, int TshiftY, int imgW, int imgH);         // EncoderContext.cpp:7310
extern u8* streamType;
extern u8* pType;

// ---- captured cross-TU calls -------------------------------------------------------------------
struct Capture { int level; std::vector<u8> data; };
static std::vector<Capture> gZstd;

extern "C" size_t __real_ZSTD_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level);
extern "C" size_t __wrap_ZSTD_compress(void* dst, size_t dstCap, const void* src, size_t srcSize, int level) {
    Capture c; c.level = level;
    c.data.assign((const u8*)src, (const u8*)src + srcSize);
    gZstd.push_back(c);
    // Level is irrelevant to the captured raw bytes; use a fast one so the oracle runs in seconds.  YK_REF_KEEP_LEVEL=1 keeps
    // the reference's own levels (18 / 21) for the stage timings of tools/measure_cpu_ratio.py.
    static const bool keep = getenv("YK_REF_KEEP_LEVEL") != NULL;
    return __real_ZSTD_compress(dst, dstCap, src, srcSize, keep ? level : 1);
}

// ---- plumbing decoder/YAIK_API.cpp (not buildable here, see Makefile) would have supplied to YAIK_Mipmap.cpp / YAIK_Alpha.cpp
// (declared in decoder/YAIK_functions.h:22-24; definitions YAIK_API.cpp:27-57).  The loops under test stay unmodified.
static int gLastError = 0, gErrorCalls = 0;
void SetErrorCode(YAIK_ERROR_CODE error) { gLastError = (int)error; gErrorCalls++; }      // recorder, never aborts
u8*  AllocateMem(YAIK_SMemAlloc*, size_t size) { return (u8*)calloc(size ? size : 1, 1); }
void FreeMem(YAIK_SMemAlloc*, void* ptr) { free(ptr); }
void internal_imageBuilderFunc(struct YAIK_SDecodedImage* userInfo, struct YAIK_SCustomDataSource* sourceImageInternal);   // YAIK_DefaultCallback.cpp:24

#include <time.h>
static double nowSec() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }


// ---- blob writer -------------------------------------------------------------------------------
static FILE* gOut = NULL;
static void blob(const std::string& name, const void* data, size_t len) {
    u32 nl = (u32)name.size();
    unsigned long long dl = len;
    fwrite(&nl, 4, 1, gOut); fwrite(name.data(), 1, nl, gOut);
    fwrite(&dl, 8, 1, gOut); if (len) fwrite(data, 1, len, gOut);
}
static void blobPlane8(const std::string& name, Plane* p) {          // 0 / non-zero -> 0 / 255 etc, clamp to u8
    int n = p->GetWidth() * p->GetHeight();
    std::vector<u8> b(n);
    for (int i = 0; i < n; i++) b[i] = (u8)p->GetPixels()[i];
    blob(name, b.data(), n);
}
static void blobPlane16(const std::string& name, Plane* p) {
    int n = p->GetWidth() * p->GetHeight();
    std::vector<short> b(n);
    for (int i = 0; i < n; i++) b[i] = (short)p->GetPixels()[i];
    blob(name, b.data(), n * 2);
}
static std::string nm(const char* base, int a, int b = -1) {
    char buf[128];
    if (b >= 0) snprintf(buf, sizeof buf, "%s_%d_%d", base, a, b); else snprintf(buf, sizeof buf, "%s_%d", base, a);
    return buf;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_driver in.bin out.blobs [partial]\n"); return 2; }
    const bool partial = argc > 3 && !strcmp(argv[3], "partial");
    const bool lut3d = argc > 4 && !strcmp(argv[3], "lut3d");
    std::vector<u8> bankBytes;
    if (lut3d) {
        FILE* fb = fopen(argv[4], "rb"); if (!fb) { perror("bank"); return 2; }
        u8 tmp[4096]; size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, fb)) > 0) bankBytes.insert(bankBytes.end(), tmp, tmp + n);
        fclose(fb);
    }
    FILE* fi = fopen(argv[1], "rb");
    if (!fi) { perror("in"); return 2; }
    int hdr[3];
    if (fread(hdr, 4, 3, fi) != 3) return 2;
    int w = hdr[0], h = hdr[1], np = hdr[2];
    Image* img = Image::CreateImage(w, h, np, false);
    for (int p = 0; p < np; p++) {
        if (fread(img->GetPlane(p)->GetPixels(), 4, (size_t)w * h, fi) != (size_t)w * h) return 2;
    }
    fclose(fi);
    gOut = fopen(argv[2], "wb");
    if (!gOut) { perror("out"); return 2; }

    // The passes printf per tile (EncoderContext.cpp:4216, :8507); silence them.
    if (!freopen("/dev/null", "w", stdout)) return 2;

    char tmpl[] = "/tmp/yaikrefXXXXXX";
    char* dir = mkdtemp(tmpl);
    if (!dir || chdir(dir) != 0) return 2;          // debug PNGs the passes write land here

    EncoderContext* ctx = new EncoderContext();
    ctx->evaluateLUT = false; ctx->evaluateLUT2D = false; ctx->dumpImage = false; ctx->pStats = NULL;
    ctx->mapSmoothTile = NULL; ctx->mappedRGB = NULL;
    ctx->SetImageToEncode(img);
    ctx->outFile = fopen("chunks.bin", "wb+");
    ctx->fileOutSize = 0;

    int meta[3] = { w, h, np };
    blob("meta", meta, sizeof meta);

    double stage[3] = { 0, 0, 0 };                   // seconds: MipPrefilter, 7x FittingQuadSmooth, 3x DynamicTileEncode (4-bpp)
    std::vector<u8> mipChunk;                        // the 'MIPM' chunk as MipPrefilter wrote it (empty: none)
    // ---- a9 alpha tile-reject ----
    if (np == 4) {
        long before = ftell(ctx->outFile);
        double t0 = nowSec();
        ctx->MipPrefilter(true);
        stage[0] = nowSec() - t0;
        fflush(ctx->outFile);
        long after = ftell(ctx->outFile);
        std::vector<u8> chunk(after - before);
        fseek(ctx->outFile, before, SEEK_SET);
        if (!chunk.empty() && fread(chunk.data(), 1, chunk.size(), ctx->outFile) != chunk.size()) return 2;
        fseek(ctx->outFile, after, SEEK_SET);
        blob("mip_chunk", chunk.data(), chunk.size());
        mipChunk = chunk;
        int b[6] = { ctx->boundX0, ctx->boundY0, ctx->boundX1, ctx->boundY1, ctx->mipMapTileSize, ctx->remainingPixels };
        blob("mip_bounds", b, sizeof b);
        blobPlane8("mip_mask", ctx->mipmapMask);
    }

    // ---- a6 gradient fit, 7 passes in the shipped order ----
    static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    Image* preview = Image::CreateImage(w, h, 3, true);
    std::vector<std::vector<u8>> bitmaps(7), rgbdq(7);
    int counts[7];
    for (int i = 0; i < 7; i++) {
        gZstd.clear();
        fflush(ctx->outFile);
        long before = ftell(ctx->outFile);
        double t0 = nowSec();
        counts[i] = ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), preview, false, passes[i][0], passes[i][1]);
        stage[1] += nowSec() - t0;
        if (!gZstd.empty()) bitmaps[i] = gZstd[0].data;          // first ZSTD_compress of the pass = raw tile bitmap (:4266)
        else {
            // No tile accepted: the pass writes no chunk (:4239); its bitmap is all zero by construction (:3777).
            u32 bx, by, bc; HeaderGradientTile::getSwizzleSize(passes[i][0], passes[i][1], bx, by, bc);
            bitmaps[i].assign((((w + bx - 1) / bx) * ((h + by - 1) / by) * bc) >> 3, 0);
        }
        if (gZstd.size() >= 2) {
            // second ZSTD_compress of the pass = PaletteCompressor output (:4301). Expand it with the reference's own
            // PaletteDecompressor exactly as the decoder does (decoder/YAIK_API.cpp:896-910): padded input, sizes from the header.
            fflush(ctx->outFile);
            long after = ftell(ctx->outFile);
            fseek(ctx->outFile, before, SEEK_SET);
            HeaderBase hb; HeaderGradientTile hg;
            if (fread(&hb, sizeof hb, 1, ctx->outFile) != 1 || fread(&hg, sizeof hg, 1, ctx->outFile) != 1) return 2;
            fseek(ctx->outFile, after, SEEK_SET);
            std::vector<u8> pal(gZstd[1].data); size_t palSize = pal.size(); pal.resize(palSize + 128 * 3, 0);
            rgbdq[i].assign(hg.streamRGBSizeUncompressed, 0);
            bool ok = PaletteDecompressor(pal.data(), (int)palSize, (int)palSize + 128 * 3, rgbdq[i].data(), (int)hg.streamRGBSizeUncompressed, hg.colorCompression);
            if (!ok) { fprintf(stderr, "PaletteDecompressor failed pass %d\n", i); return 3; }
            blob(nm("grad_palette", i), gZstd[1].data.data(), gZstd[1].data.size());
        }
        blob(nm("grad_bitmap", i), bitmaps[i].data(), bitmaps[i].size());
        blob(nm("grad_rgbdq", i), rgbdq[i].data(), rgbdq[i].size());
    }
    blob("grad_counts", counts, sizeof counts);
    blobPlane8("smoothMap", ctx->smoothMap);
    blobPlane8("mipmapMask_post", ctx->mipmapMask);
    for (int p = 0; p < 3; p++) {
        blobPlane8(nm("mapSmoothTile", p), ctx->mapSmoothTile->GetPlane(p));
        blobPlane16(nm("preview", p), preview->GetPlane(p));
    }
    int bnd[4] = { ctx->boundX0, ctx->boundY0, ctx->boundX1, ctx->boundY1 };
    blob("bounds_post", bnd, sizeof bnd);

    // ---- a10-a13 range quantiser, both start modes (0 = 4-bpp allowed, 3 = 3-bpp only) ----
    DynamicTileEncoderTable();
    for (int m = 0; m < 2; m++) {
        for (int p = 0; p < 3; p++) {
            Plane* dst = new Plane(w, h);
            BoundingBox full = dst->GetRect();
            dst->Fill(full, -1);
            gZstd.clear();
            double t0 = nowSec();
            ctx->DynamicTileEncode(m == 1, img->GetPlane(p), dst, false, false, false, false);
            if (m == 0) stage[2] += nowSec() - t0;
            // :4519 tile definitions, :4533 nibble index stream
            blob(nm("plnt_defs", m, p), gZstd[0].data.data(), gZstd[0].data.size());
            blob(nm("plnt_idx", m, p), gZstd[1].data.data(), gZstd[1].data.size());
            blobPlane16(nm("plnt_dst", m, p), dst);
            delete dst;
        }
    }

    blob("stage_seconds", stage, sizeof stage);

    // ---- (f)4 3D-LUT tiles: Load3DPattern (:7851), Set3DPointCloud (:4744), Correlation3DSearch (:6245) with computeValues3D (:5807),
    // EndCorrelationSearch (:7366).  Runs where Convert() runs it: after the gradient passes, before the 1-D compressor.
    std::vector<Capture> lutZin; std::vector<u8> lutHeader, lutFile;
    if (lut3d) {
        int nPat = 0;
        for (size_t off = 0; off < bankBytes.size(); nPat++) {
            const size_t len = 1 + 3 * (size_t)bankBytes[off];
            char name[64]; snprintf(name, sizeof name, "pattern_%02d.lut", nPat);
            FILE* fp = fopen(name, "wb"); if (!fp) return 2;
            fwrite(&bankBytes[off], 1, len, fp); fclose(fp);
            ctx->Load3DPattern(name);
            off += len;
        }
        if (ctx->correlationPatternCount3D != nPat) return 4;
        for (int e = 0; e < nPat; e++) {                                  // the tables Set3DPointCloud built
            EncoderContext::EvalCtx3D& ev = ctx->correlationPattern3D[e];
            s16 f[(64 + 32 + 16 + 8) * 3];
            memcpy(f, ev.xFactor6Bit, 128); memcpy(f + 64, ev.yFactor6Bit, 128); memcpy(f + 128, ev.zFactor6Bit, 128);
            memcpy(f + 192, ev.xFactor5Bit, 64); memcpy(f + 224, ev.yFactor5Bit, 64); memcpy(f + 256, ev.zFactor5Bit, 64);
            memcpy(f + 288, ev.xFactor4Bit, 32); memcpy(f + 304, ev.yFactor4Bit, 32); memcpy(f + 320, ev.zFactor4Bit, 32);
            memcpy(f + 336, ev.xFactor3Bit, 16); memcpy(f + 344, ev.yFactor3Bit, 16); memcpy(f + 352, ev.zFactor3Bit, 16);
            blob(nm("lut_factors", e), f, sizeof f);
            blob(nm("lut_distanceField", e), ev.distanceField3D, sizeof ev.distanceField3D);
            std::vector<u8> pos(4 * 64 * 64 * 64);
            for (int i = 0; i < 64 * 64 * 64; i++) {
                pos[i] = (u8)ev.position6Bit3D[i]; pos[262144 + i] = (u8)ev.position5Bit3D[i];
                pos[2 * 262144 + i] = (u8)ev.position4Bit3D[i]; pos[3 * 262144 + i] = (u8)ev.position3Bit3D[i];
            }
            blob(nm("lut_positions", e), pos.data(), pos.size());
        }
        ctx->pStats = new EncoderStats();                                 // EndCorrelationSearch adds to it unconditionally (:7622)
        ctx->useYCoCg = false; ctx->isCaptureMode3D = false; ctx->testedLUT = nPat;
        ctx->StartCorrelationSearch(true);
        static const int lutPass[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };       // :9144-9199
        int cnt[6][6];
        for (int i = 0; i < 6; i++) {
            ctx->Correlation3DSearch(img, preview, lutPass[i][0], lutPass[i][1]);
            const int c6[6] = { ctx->streamTypeCnt, ctx->streamColorCnt, ctx->stream3BitCnt, ctx->stream4BitCnt, ctx->stream5BitCnt, ctx->stream6BitCnt };
            memcpy(cnt[i], c6, sizeof c6);
        }
        blob("lut_counts", cnt, sizeof cnt);                             // cumulative after each pass: tiles, colour bytes, 3/4/5/6-bit indices
        blob("lut_tileType", ctx->corr3D_tileStreamTileType, (size_t)ctx->streamTypeCnt * 2);
        blob("lut_color", ctx->corr3D_colorStream, ctx->streamColorCnt);
        blob("lut_idx3", ctx->corr3D_stream3Bit, ctx->stream3BitCnt);
        blob("lut_idx4", ctx->corr3D_stream4Bit, ctx->stream4BitCnt);
        blob("lut_idx5", ctx->corr3D_stream5Bit, ctx->stream5BitCnt);
        blob("lut_idx6", ctx->corr3D_stream6Bit, ctx->stream6BitCnt);
        u8* maps[6] = { ctx->corr3D_sizeT16_8Map, ctx->corr3D_sizeT8_16Map, ctx->corr3D_sizeT8_8Map, ctx->corr3D_sizeT8_4Map, ctx->corr3D_sizeT4_8Map, ctx->corr3D_sizeT4_4Map };
        for (int i = 0; i < 6; i++) blob(nm("lut_map", i), maps[i], BitmapSwizzleMapSize(lutPass[i][0], lutPass[i][1], w, h));
        for (int p = 0; p < 3; p++) blobPlane8(nm("lut_mapSmoothTile", p), ctx->mapSmoothTile->GetPlane(p));
        for (int p = 0; p < 3; p++) blobPlane16(nm("lut_preview", p), preview->GetPlane(p));
        gZstd.clear();
        fflush(ctx->outFile);
        const long before = ftell(ctx->outFile);
        ctx->EndCorrelationSearch(true, 7);                               // '3DTL' chunk: 6 maps, tile types, colours (CompressF), index streams (x3)
        fflush(ctx->outFile);
        const long after = ftell(ctx->outFile);
        std::vector<u8> chunk((size_t)(after - before));
        fseek(ctx->outFile, before, SEEK_SET);
        if (!chunk.empty() && fread(chunk.data(), 1, chunk.size(), ctx->outFile) != chunk.size()) return 2;
        fseek(ctx->outFile, after, SEEK_SET);
        blob("lut_chunk_header", chunk.data(), chunk.size() < 8 + sizeof(HeaderTile3D) ? chunk.size() : 8 + sizeof(HeaderTile3D));
        for (size_t k = 0; k < gZstd.size(); k++) blob(nm("lut_zin", (int)k), gZstd[k].data.data(), gZstd[k].data.size());
        lutZin.assign(gZstd.begin(), gZstd.end());
        lutHeader.resize(sizeof(HeaderTile3D));
        if (chunk.size() >= 8 + sizeof(HeaderTile3D)) memcpy(lutHeader.data(), chunk.data() + 8, sizeof(HeaderTile3D));
        // the decoder's LUT file 'LUL0' as RegisterAndCreate3DLut writes it (:7820-7847; that function itself only loads the reference's
        // hard-coded bank file names and returns at once when patterns are already loaded): header + BinarySave3D (:5452) per depth and pattern
        LUTHeader hd; memset(&hd, 0, sizeof hd);
        hd.lutH[0] = 'L'; hd.lutH[1] = 'U'; hd.lutH[2] = 'L'; hd.lutH[3] = '0'; hd.version = 0; hd.entryCount = (u8)(nPat - 1); hd.padding_extension[0] = 1;
        lutFile.resize(sizeof hd + (size_t)(64 + 32 + 16 + 8) * 3 * nPat);
        memcpy(lutFile.data(), &hd, sizeof hd);
        u8* fill = lutFile.data() + sizeof hd;
        for (int n = 0; n < 4; n++) for (int m = 0; m < nPat; m++) fill = ctx->correlationPattern3D[m].BinarySave3D(fill, 0, (EncoderContext::Mode)n);
        blob("lut_file", lutFile.data(), lutFile.size());
    }

    // ---- a6 with nullable planes: the partial-plane 4x4 passes (PlaneBit :3715, per-plane allow :3871-3875, per-plane paint :4031-4034)
    static const int ppMask[6] = { 5, 3, 6, 1, 2, 4 };                  // RB, RG, GB, R, G, B
    std::vector<std::vector<u8>> ppBitmaps(6), ppRgbdq(6);
    int ppCounts[6] = { 0, 0, 0, 0, 0, 0 };
    if (partial) {
        for (int i = 0; i < 6; i++) {
            gZstd.clear();
            fflush(ctx->outFile);
            long before = ftell(ctx->outFile);
            const int m = ppMask[i];
            ppCounts[i] = ctx->FittingQuadSmooth(3, (m & 1) ? img->GetPlane(0) : NULL, (m & 2) ? img->GetPlane(1) : NULL, (m & 4) ? img->GetPlane(2) : NULL,
                                                 preview, false, 2, 2);
            if (!gZstd.empty()) ppBitmaps[i] = gZstd[0].data;
            else ppBitmaps[i].assign((((w + 31) / 32) * ((h + 31) / 32) * 64) >> 3, 0);
            if (gZstd.size() >= 2) {
                fflush(ctx->outFile);
                long after = ftell(ctx->outFile);
                fseek(ctx->outFile, before, SEEK_SET);
                HeaderBase hb; HeaderGradientTile hg;
                if (fread(&hb, sizeof hb, 1, ctx->outFile) != 1 || fread(&hg, sizeof hg, 1, ctx->outFile) != 1) return 2;
                fseek(ctx->outFile, after, SEEK_SET);
                std::vector<u8> pal(gZstd[1].data); size_t palSize = pal.size(); pal.resize(palSize + 128 * 3, 0);
                // the palette codec works in RGB triples: with one or two planes the stream length need not be a multiple of 3 and the
                // reference then writes up to 2 bytes past `outputSize` (YAIK_GenericFunctions.cpp:131-137, :171) - give it the room
                ppRgbdq[i].assign(hg.streamRGBSizeUncompressed + 4, 0);
                if (!PaletteDecompressor(pal.data(), (int)palSize, (int)palSize + 128 * 3, ppRgbdq[i].data(), (int)hg.streamRGBSizeUncompressed, hg.colorCompression)) return 3;
                ppRgbdq[i].resize(hg.streamRGBSizeUncompressed);
                int hdr2[2] = { (int)hg.plane, (int)hg.format };
                blob(nm("pp_header", i), hdr2, sizeof hdr2);
                blob(nm("pp_palette", i), gZstd[1].data.data(), gZstd[1].data.size());
            }
            blob(nm("pp_bitmap", i), ppBitmaps[i].data(), ppBitmaps[i].size());
            blob(nm("pp_rgbdq", i), ppRgbdq[i].data(), ppRgbdq[i].size());
        }
        blob("pp_counts", ppCounts, sizeof ppCounts);
        blobPlane8("pp_smoothMap", ctx->smoothMap);
        for (int p = 0; p < 3; p++) blobPlane8(nm("pp_mapSmoothTile", p), ctx->mapSmoothTile->GetPlane(p));
    }

    // ---- a15 live 1-D range path ----
    size_t cap = (size_t)w * h * 3 + 64;
    std::vector<u8> pix(cap), types((size_t)(w / 8) * (h / 8) * 9 + 64);
    streamType = types.data(); pType = types.data();       // the reference's own 100000-byte global overflows beyond ~512x512
    u8* wr = pix.data();
    int ends[6];
    Image* out1d = Image::CreateImage(w, h, 3, true);
    for (int p = 0; p < 3; p++) {
        wr = ctx->DynamicTileCompressor(wr, img->GetPlane(p), ctx->mapSmoothTile->GetPlane(p), out1d->GetPlane(p));
        ends[p] = (int)(wr - pix.data());
        ends[3 + p] = (int)(pType - types.data());
    }
    blob("d1_pix", pix.data(), ends[2]);
    blob("d1_type", types.data(), ends[5]);
    blob("d1_ends", ends, sizeof ends);
    for (int p = 0; p < 3; p++) blobPlane16(nm("d1_out", p), out1d->GetPlane(p));
    // '1DTL' chunk of the three planes (EncoderContext.cpp:8524-8576), then everything the passes appended to outFile so far:
    // ['MIPM'] 7x['GTIL'] 6x'PLNT' '1DTL' exactly as the reference framed them (ZStd 1.3.4 payloads).
    ctx->GenerateDynamicTileChunk(pix.data(), ends[2]);
    {
        fflush(ctx->outFile);
        long all = ftell(ctx->outFile);
        std::vector<u8> file(all);
        fseek(ctx->outFile, 0, SEEK_SET);
        if (all && fread(file.data(), 1, file.size(), ctx->outFile) != file.size()) return 2;
        fseek(ctx->outFile, all, SEEK_SET);
        blob("chunks_file", file.data(), file.size());
    }

    // ---- a16/a17 decode loops on a hand-filled YAIK_Instance (allocation rule: decoder/YAIK_API.cpp:650-657, :855-874) ----
    YAIK_Instance inst; memset(&inst, 0, sizeof inst);
    inst.width = (u16)w; inst.height = (u16)h;
    inst.tileWidth = (u16)((w + 7) >> 3); inst.tileHeight = (u16)((h + 7) >> 3);
    int planeSize = inst.tileWidth * inst.tileHeight * 64;
    std::vector<u8> planes((size_t)planeSize * 3, 0);
    inst.planeR = planes.data(); inst.planeG = inst.planeR + planeSize; inst.planeB = inst.planeG + planeSize;
    inst.strideRGBMap = (u16)((w >> 2) + 1);
    int latt = inst.strideRGBMap * ((h >> 2) + 1);
    std::vector<u8> mapRGB((size_t)latt * 3, 0);
    int sizeMask = (latt + 7) >> 3;
    std::vector<u8> mapMask((size_t)sizeMask * 3, 0);
    inst.mapRGB = mapRGB.data(); inst.mapRGBMask = mapMask.data(); inst.sizeMapMask = sizeMask;
    inst.tile4x4MaskSize = ((((w + 15) >> 4) << 2) * (((h + 7) >> 3) << 1)) >> 3;
    std::vector<u8> t4((size_t)inst.tile4x4MaskSize * 3, 0);
    inst.tile4x4Mask = t4.data(); inst.singleRGB = true;

    typedef void (*GradFn)(YAIK_Instance*, u8*, u8*, u8*, u8*, u8*, u8);
    GradFn fns[7] = { DecompressGradient16x16, DecompressGradient16x8, DecompressGradient8x16, DecompressGradient8x8,
                      DecompressGradient8x4, DecompressGradient4x8, DecompressGradient4x4 };
    size_t slack = (size_t)((w + 3) >> 2) * ((h + 3) >> 2) * 12;     // "secure buffer" of YAIK_API.cpp:901
    for (int i = 0; i < 7; i++) {
        if (counts[i] == 0) continue;
        std::vector<u8> rgb(rgbdq[i]); rgb.resize(rgb.size() + slack, 0);
        std::vector<u8> bm(bitmaps[i]);
        fns[i](&inst, bm.data(), rgb.data(), inst.planeR, inst.planeG, inst.planeB, 7);
    }
    blob("dec_planes_grad", planes.data(), planes.size());
    blob("dec_tile4x4", t4.data(), inst.tile4x4MaskSize);
    if (lut3d && !lutFile.empty() && lutHeader.size() == sizeof(HeaderTile3D)) {
        // ---- (f)4 decode: Tile3D_16x8 .. Tile3D_4x4 (decoder/YAIK_3DTile.cpp:244-2140) in the order of the '3DTL' chunk reader
        // (decoder/YAIK_API.cpp:1002-1270), on the streams EndCorrelationSearch compressed.  The per-orientation tables are laid out by
        // YAIK_AssignLUT (YAIK_API.cpp:133-415), which is in the translation unit that cannot be built here: the layout below is harness
        // plumbing that follows its comment block and loops (per depth: [pattern][64 slots, 48 used: 6 axis orders x 8 flips][entry][3]).
        HeaderTile3D h3; memcpy(&h3, lutHeader.data(), sizeof h3);
        const int nPat = lutFile[5] + 1;
        std::vector<std::vector<u8>> tbl(4);
        const u8* stream = lutFile.data() + sizeof(LUTHeader);
        for (int bit = 3; bit <= 6; bit++) {
            const int len = 1 << bit;
            std::vector<u8>& T = tbl[bit - 3];
            T.assign((size_t)len * 3 * 64 * 256 + 256 * 3, 251);
            u8* pFill = T.data();
            for (int e = 0; e < nPat; e++) {
                const u8* o[3] = { stream, stream + len, stream + 2 * len };
                static const int axis[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {1,2,0}, {2,0,1}, {2,1,0} };
                for (int pat = 0; pat < 6; pat++)
                    for (int flip = 0; flip < 8; flip++)
                        for (int idx = 0; idx < len; idx++)
                            for (int c = 0; c < 3; c++) { const u8 v = o[axis[pat][c]][idx]; *pFill++ = ((flip >> c) & 1) ? (u8)(128 - v) : v; }
                memset(pFill, 251, (size_t)16 * 3 * len); pFill += (size_t)16 * 3 * len;
                stream += len * 3;
            }
        }
        u8* TBL[4] = { tbl[0].data(), tbl[1].data(), tbl[2].data(), tbl[3].data() };
        // the captured ZStd inputs, in EndCorrelationSearch's call order: 6 maps, [tile types], [colours], [3], [4], [5], [6] bit
        size_t zi = 6;
        std::vector<u8> maps[6];
        for (int k = 0; k < 6; k++) { maps[k] = lutZin[k].data; maps[k].resize(maps[k].size() + 64, 0); }
        std::vector<u8> tiles, colors, idxs[4];
        if (h3.streamTypeCnt) tiles = lutZin[zi++].data;
        if (h3.streamColorCnt) colors = lutZin[zi++].data;
        const u32 cnts[4] = { h3.stream3BitCnt, h3.stream4BitCnt, h3.stream5BitCnt, h3.stream6BitCnt };
        for (int k = 0; k < 4; k++) if (cnts[k]) idxs[k] = lutZin[zi++].data;
        tiles.resize(tiles.size() + 64, 0); colors.resize(colors.size() + 64, 0);
        for (int k = 0; k < 4; k++) idxs[k].resize(idxs[k].size() + 256, 0);
        PaletteFullRangeRemapping(colors.data(), h3.compressionRateColor, (int)h3.streamColorCnt);      // YAIK_API.cpp:1102
        TileParam tpm; memset(&tpm, 0, sizeof tpm);
        tpm.stream3Bit = idxs[0].data(); tpm.stream4Bit = idxs[1].data(); tpm.stream5Bit = idxs[2].data(); tpm.stream6Bit = idxs[3].data();
        tpm.tileStream = (u16*)tiles.data(); tpm.colorStream = colors.data();
        typedef void (*T3)(YAIK_Instance*, HeaderTile3D*, TileParam*, u8**);
        T3 t3[6] = { Tile3D_16x8, Tile3D_8x16, Tile3D_8x8, Tile3D_8x4, Tile3D_4x8, Tile3D_4x4 };
        for (int k = 0; k < 6; k++) {
            bool any = false; for (size_t i = 0; i + 64 < maps[k].size() + 0 && !any; i++) any = maps[k][i] != 0;
            if (!any) continue;                                           // CheckTileCount == 0 -> not called (YAIK_API.cpp:1111)
            tpm.currentMap = maps[k].data();
            t3[k](&inst, &h3, &tpm, TBL);
        }
        int used[6] = { (int)((u8*)tpm.tileStream - tiles.data()), (int)(tpm.colorStream - colors.data()), (int)(tpm.stream3Bit - idxs[0].data()),
                        (int)(tpm.stream4Bit - idxs[1].data()), (int)(tpm.stream5Bit - idxs[2].data()), (int)(tpm.stream6Bit - idxs[3].data()) };
        blob("lut_dec_consumed", used, sizeof used);
        blob("lut_dec_planes", planes.data(), planes.size());
        blob("lut_dec_tile4x4", t4.data(), inst.tile4x4MaskSize);
    }
    blob("dec_mapRGB", mapRGB.data(), mapRGB.size());
    blob("dec_mapRGBMask", mapMask.data(), sizeMask);

    // transition single -> per-plane masks (YAIK_API.cpp:530-544), then the three Decompress1D calls (:985-991)
    for (int p = 1; p < 3; p++) {
        memcpy(&mapMask[(size_t)sizeMask * p], mapMask.data(), sizeMask);
        memcpy(&t4[(size_t)inst.tile4x4MaskSize * p], t4.data(), inst.tile4x4MaskSize);
    }
    inst.singleRGB = false;
    if (partial) {                                                      // 'GTIL' chunks with plane != 7 come behind the split (YAIK_API.cpp:875-877)
        for (int i = 0; i < 6; i++) {
            if (ppCounts[i] == 0) continue;
            std::vector<u8> rgb(ppRgbdq[i]); rgb.resize(rgb.size() + slack, 0);
            std::vector<u8> bm(ppBitmaps[i]);
            DecompressGradient4x4(&inst, bm.data(), rgb.data(), inst.planeR, inst.planeG, inst.planeB, (u8)ppMask[i]);
        }
        blob("pp_dec_planes_grad", planes.data(), planes.size());
        blob("pp_dec_tile4x4", t4.data(), t4.size());
        blob("pp_dec_mapRGBMask", mapMask.data(), mapMask.size());
    }
    Header1D h1; memset(&h1, 0, sizeof h1);
    h1.compressionColor = (u8)ctx->colorCompression1D; h1.compressionRange = (u8)ctx->rangeCompression1D;
    std::vector<u8> pixPad(pix.begin(), pix.begin() + ends[2]); pixPad.resize(pixPad.size() + 64, 0);
    std::vector<u8> typPad(types.begin(), types.begin() + ends[5]); typPad.resize(typPad.size() + 64, 0);
    u8* tp = typPad.data(); u8* pp = pixPad.data();
    for (int p = 0; p < 3; p++) Decompress1D(&inst, &tp, &pp, (u8)p, &h1);
    int consumed[2] = { (int)(tp - typPad.data()), (int)(pp - pixPad.data()) };
    blob("dec_1d_consumed", consumed, sizeof consumed);
    blob("dec_planes_full", planes.data(), planes.size());

    // ---- a18 Decompress1BitTiled on the reference's own 'MIPM' payload (chunk reader: decoder/YAIK_API.cpp:732-749) ----
    if (mipChunk.size() > sizeof(HeaderBase) + sizeof(MipmapHeader)) {
        MipmapHeader mh; memcpy(&mh, mipChunk.data() + sizeof(HeaderBase), sizeof mh);
        std::vector<u8> payload(mipChunk.begin() + sizeof(HeaderBase) + sizeof(MipmapHeader), mipChunk.end());
        size_t payloadLen = payload.size(); payload.resize(payloadLen + 64, 0);
        gLastError = 0; gErrorCalls = 0;
        inst.mipMapMask = NULL;
        bool ok = Decompress1BitTiled(&inst, &mh, payload.data(), (u32)payloadLen);
        int info[8] = { mh.bbox.x, mh.bbox.y, mh.bbox.w, mh.bbox.h, (int)mh.mipmapLevel, ok ? 1 : 0, gLastError, gErrorCalls };
        blob("dec_mask_info", info, sizeof info);
        int mb[4] = { inst.maskBBox.x, inst.maskBBox.y, inst.maskBBox.w, inst.maskBBox.h };
        blob("dec_mask_bbox", mb, sizeof mb);
        if (ok && inst.mipMapMask) blob("dec_mask", inst.mipMapMask, ((size_t)inst.maskBBox.w * inst.maskBBox.h) >> 3);
        free(inst.mipMapMask); inst.mipMapMask = NULL;
    }

    // ---- a20 internal_imageBuilderFunc on the fully decoded tiled planes: RGB rows at a padded outputImageStride, and (4-plane
    // inputs) the RGBA branch exactly as the reference executes it, alpha = the source alpha plane as a linear 8-bit buffer ----
    {
        YAIK_SCustomDataSource src; memset(&src, 0, sizeof src);
        src.planeR = inst.planeR; src.planeG = inst.planeG; src.planeB = inst.planeB; src.planeA = NULL;
        src.strideR = src.strideG = src.strideB = inst.tileWidth * 64;
        YAIK_SDecodedImage user; memset(&user, 0, sizeof user);
        user.width = (u16)w; user.height = (u16)h; user.hasAlpha = false;
        int stride = w * 3 + 13;
        std::vector<u8> out((size_t)stride * h + 64, 0xA5);
        user.outputImage = out.data(); user.outputImageStride = stride;
        internal_imageBuilderFunc(&user, &src);
        int si[2] = { stride, 3 };
        blob("dec_rgb_out_info", si, sizeof si);
        blob("dec_rgb_out", out.data(), (size_t)stride * h);
        if (np == 4) {
            std::vector<u8> alpha((size_t)w * h);
            for (size_t i = 0; i < alpha.size(); i++) alpha[i] = (u8)img->GetPlane(3)->GetPixels()[i];
            src.planeA = alpha.data(); src.strideA = w; user.hasAlpha = true;
            int strideA = w * 4 + 20;
            std::vector<u8> outA((size_t)strideA * h + 64, 0xA5);
            user.outputImage = outA.data(); user.outputImageStride = strideA;
            internal_imageBuilderFunc(&user, &src);
            int sa[2] = { strideA, 4 };
            blob("dec_rgba_out_info", sa, sizeof sa);
            blob("dec_rgba_out", outA.data(), (size_t)strideA * h);
        }
    }

    fclose(gOut);
    fclose(ctx->outFile);
    // leave the scratch dir clean
    if (system("rm -f ./*.png ./EncoderDebug* chunks.bin") != 0) {}
    if (chdir("/") == 0) rmdir(dir);
    return 0;
}
