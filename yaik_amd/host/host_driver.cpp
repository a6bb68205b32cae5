// Small C++ program written the way a YAIK maintainer would drive the reference's EncoderContext for this path
// (encoder/ImageEncoder.cpp:158-213 / EncoderContext::Convert): same call sequence, same method names, but linked against
// the MI355X drop-in.  tests/test_gpu_host_mirror.py checks its output blob by blob.
//
// usage: host_driver <in.bin> <out.blobs> [mode3BitOnly] [out.yaik]
//        host_driver <in.bin> <out.blobs> lut <bank.bin>   3-D LUT tiles (Load3DPattern, Start/Correlation3DSearch x6/EndCorrelationSearch, :9117-9218)
//                                                   behind the gradient passes, the '3DTL' chunk decoded back after YAIK_AssignLUT (see run_lut)
//        host_driver <in.bin> <out.blobs> stripes <n>   the tile maps of the image encoded as n row stripes (ConvertHotPathStripes: stripe i on device
//                                                   i modulo the devices present; distinct devices are gathered by one RCCL transfer) next to the whole-image passes
//        host_driver <in.bin> <out.blobs> pp        the six plane-subset 4x4 passes of Convert() (:9261-9415) after the RGB passes, written
//                                                   as a .yaik stream and decoded back (see run_partial below)
// With the 4th argument the image is also converted to a .yaik stream (ConvertHotPath) and decoded back through the
// YAIK_* decoder API, the way an application would use the two libraries.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include "EncoderContext.h"
#include "../../include/yaik_hip.h"
#include "yaik_decode.h"
#include "chunks.h"
#include "palette.h"

static FILE* gOut;
static void blob(const std::string& name, const void* data, size_t len) {
    u32 nl = (u32)name.size(); unsigned long long dl = len;
    fwrite(&nl, 4, 1, gOut); fwrite(name.data(), 1, nl, gOut); fwrite(&dl, 8, 1, gOut); if (len) fwrite(data, 1, len, gOut);
}
static std::string nm(const char* b, int a, int c = -1) { char t[96]; if (c >= 0) snprintf(t, sizeof t, "%s_%d_%d", b, a, c); else snprintf(t, sizeof t, "%s_%d", b, a); return t; }

// FittingQuadSmooth with NULL planes, the way Convert() lists those calls (RB, RG, GB, R, G, B at 4x4), then the 1-D compressor on the
// per-plane maps; the chunks go to a .yaik stream which is decoded back with consistent tile marks (YAIK_SetPartialPlaneMarks)
static int run_partial(EncoderContext* ctx, Image* img, int w, int h, int np) {
    FILE* yf = tmpfile(); if (!yf) return 2;
    PaletteResetCodeBook();
    if (!yaikchunk::writeFileHeader(yf, w, h, np == 4)) return 2;
    ctx->outFile = yf;
    if (np == 4) ctx->MipPrefilter(true);
    static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    for (int i = 0; i < 7; i++) ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), nullptr, false, passes[i][0], passes[i][1]);
    static const int masks[6] = { 5, 3, 6, 1, 2, 4 };
    int counts[6];
    for (int i = 0; i < 6; i++) {
        const int m = masks[i];
        counts[i] = ctx->FittingQuadSmooth(3, (m & 1) ? img->GetPlane(0) : nullptr, (m & 2) ? img->GetPlane(1) : nullptr, (m & 4) ? img->GetPlane(2) : nullptr,
                                           nullptr, false, 2, 2);
        blob(nm("pp_bitmap", i), ctx->LastGradientBitmap().data(), ctx->LastGradientBitmap().size());
        blob(nm("pp_rgbraw", i), ctx->LastGradientRGBStream().data(), ctx->LastGradientRGBStream().size());
    }
    blob("pp_counts", counts, sizeof counts);
    std::vector<u8> pix((size_t)w * h * 3 + 64);
    u8* wr = pix.data();
    int ends[3];
    for (int p = 0; p < 3; p++) { wr = ctx->DynamicTileCompressor(wr, img->GetPlane(p), nullptr, nullptr); ends[p] = (int)(wr - pix.data()); }
    blob("d1_pix", pix.data(), (size_t)(wr - pix.data()));
    blob("d1_type", ctx->TileTypeStream1D().data(), ctx->TileTypeStream1D().size());
    blob("d1_pix_ends", ends, sizeof ends);
    ctx->GenerateDynamicTileChunk(pix.data(), (int)(wr - pix.data()));
    if (*ctx->LastError()) { fprintf(stderr, "%s\n", ctx->LastError()); return 4; }
    if (!yaikchunk::writeEndOfFile(yf)) return 2;
    ctx->outFile = nullptr;
    fflush(yf);
    const long n = ftell(yf);
    std::vector<u32> stream(((size_t)n + 3) / 4);
    fseek(yf, 0, SEEK_SET);
    if (fread(stream.data(), 1, (size_t)n, yf) != (size_t)n) return 2;
    fclose(yf);
    blob("yaik_file", stream.data(), (size_t)n);
    if ((w & 15) || (h & 15)) return 0;
    YAIK_LIB lib = YAIK_Init(1, nullptr);
    if (!lib) return 5;
    static std::vector<u8> tiled;
    for (int consistent = 0; consistent < 2; consistent++) {
        YAIK_SetPartialPlaneMarks(consistent);
        YAIK_SDecodedImage di;
        if (!YAIK_DecodeImagePre(lib, stream.data(), (u32)n, &di)) { fprintf(stderr, "Pre failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        std::vector<u8> outImg((size_t)di.width * di.height * 4);
        di.outputImage = outImg.data(); di.outputImageStride = di.width * 4;
        di.customImageOutput = [](YAIK_SDecodedImage* u, YAIK_SCustomDataSource* s) {
            const size_t planeSize = (size_t)(u->width / 8) * (u->height / 8) * 64;
            tiled.assign(s->planeR, s->planeR + planeSize);
            tiled.insert(tiled.end(), s->planeG, s->planeG + planeSize);
            tiled.insert(tiled.end(), s->planeB, s->planeB + planeSize);
        };
        if (!YAIK_DecodeImage(stream.data(), (u32)n, &di)) { fprintf(stderr, "Decode failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        blob(nm("yaik_planes_tiled", consistent), tiled.data(), tiled.size());
    }
    YAIK_SetPartialPlaneMarks(0);
    YAIK_Release(lib);
    return 0;
}

// The 3-D LUT part of Convert() against the mirror: bank files -> Load3DPattern, the six searches, '3DTL' chunk, LutFile -> YAIK_AssignLUT -> decode
static int run_lut(EncoderContext* ctx, Image* img, int w, int h, int np, const char* bankPath) {
    std::vector<u8> bank;
    { FILE* fb = fopen(bankPath, "rb"); if (!fb) return 2; u8 tmp[4096]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, fb)) > 0) bank.insert(bank.end(), tmp, tmp + n); fclose(fb); }
    char dirT[] = "/tmp/yaikhostXXXXXX";
    const char* dir = mkdtemp(dirT); if (!dir) return 2;
    int nPat = 0;
    for (size_t off = 0; off < bank.size(); nPat++) {
        const size_t len = 1 + 3 * (size_t)bank[off];
        const std::string name = std::string(dir) + "/pattern_" + std::to_string(nPat) + ".lut";
        FILE* fp = fopen(name.c_str(), "wb"); if (!fp) return 2;
        fwrite(&bank[off], 1, len, fp); fclose(fp);
        ctx->Load3DPattern(name.c_str());
        remove(name.c_str());
        off += len;
    }
    if (ctx->correlationPatternCount3D != nPat) { fprintf(stderr, "%s\n", ctx->LastError()); return 4; }
    FILE* yf = tmpfile(); if (!yf) return 2;
    PaletteResetCodeBook();
    if (!yaikchunk::writeFileHeader(yf, w, h, np == 4)) return 2;
    ctx->outFile = yf;
    if (np == 4) ctx->MipPrefilter(true);
    static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    for (int i = 0; i < 7; i++) ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), nullptr, false, passes[i][0], passes[i][1]);
    ctx->StartCorrelationSearch(true);
    static const int lp[6][2] = { {4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    int matched[6];
    for (int i = 0; i < 6; i++) { ctx->Correlation3DSearch(img, nullptr, lp[i][0], lp[i][1]); matched[i] = ctx->LastCorrelationMatches(); }
    blob("lut_matched", matched, sizeof matched);
    ctx->EndCorrelationSearch(true, 7);
    std::vector<u8> pix((size_t)w * h * 3 + 64);
    u8* wr = pix.data();
    for (int p = 0; p < 3; p++) wr = ctx->DynamicTileCompressor(wr, img->GetPlane(p), nullptr, nullptr);
    blob("d1_pix", pix.data(), (size_t)(wr - pix.data()));
    blob("d1_type", ctx->TileTypeStream1D().data(), ctx->TileTypeStream1D().size());
    ctx->GenerateDynamicTileChunk(pix.data(), (int)(wr - pix.data()));
    if (*ctx->LastError()) { fprintf(stderr, "%s\n", ctx->LastError()); return 4; }
    if (!yaikchunk::writeEndOfFile(yf)) return 2;
    ctx->outFile = nullptr;
    fflush(yf);
    const long n = ftell(yf);
    std::vector<u32> stream(((size_t)n + 3) / 4);
    fseek(yf, 0, SEEK_SET);
    if (fread(stream.data(), 1, (size_t)n, yf) != (size_t)n) return 2;
    fclose(yf);
    blob("yaik_file", stream.data(), (size_t)n);
    const std::string lutName = std::string(dir) + "/LutFile.lut";
    if (!ctx->Save3DLutFile(lutName.c_str())) { fprintf(stderr, "%s\n", ctx->LastError()); return 4; }
    std::vector<u8> lutFile;
    { FILE* fl = fopen(lutName.c_str(), "rb"); if (!fl) return 2; u8 tmp[4096]; size_t k; while ((k = fread(tmp, 1, sizeof tmp, fl)) > 0) lutFile.insert(lutFile.end(), tmp, tmp + k); fclose(fl); remove(lutName.c_str()); }
    rmdir(dir);
    blob("lut_file", lutFile.data(), lutFile.size());
    if ((w & 15) || (h & 15)) return 0;
    YAIK_LIB lib = YAIK_Init(1, nullptr);
    if (!lib) return 5;
    static std::vector<u8> tiled;
    YAIK_SDecodedImage di;
    std::vector<u8> outImg((size_t)w * h * 4);
    auto decodeOnce = [&]() -> int {
        if (!YAIK_DecodeImagePre(lib, stream.data(), (u32)n, &di)) return (int)YAIK_GetErrorCode();
        di.outputImage = outImg.data(); di.outputImageStride = di.width * 4;
        di.customImageOutput = [](YAIK_SDecodedImage* u, YAIK_SCustomDataSource* s) {
            const size_t planeSize = (size_t)(u->width / 8) * (u->height / 8) * 64;
            tiled.assign(s->planeR, s->planeR + planeSize);
            tiled.insert(tiled.end(), s->planeG, s->planeG + planeSize);
            tiled.insert(tiled.end(), s->planeB, s->planeB + planeSize);
        };
        return YAIK_DecodeImage(stream.data(), (u32)n, &di) ? 0 : (int)YAIK_GetErrorCode();
    };
    int codes[2];
    codes[0] = decodeOnce();                                          // no LUT assigned yet: the '3DTL' chunk must be refused (YAIK_INVALID_LUT)
    YAIK_AssignLUT(lib, lutFile.data(), (u32)lutFile.size());
    codes[1] = decodeOnce();
    blob("yaik_lut_codes", codes, sizeof codes);
    blob("yaik_planes_tiled", tiled.data(), tiled.size());
    {   // a crafted '3DTL' header: tile count 0x80000001 with colour count 6 passes a 32-bit "colours == 6 * tiles" test (the product wraps);
        // truncated counts must be refused with YAIK_INVALID_STREAM before anything is expanded or handed to the GPU
        int neg[4] = { -1, -1, -1, -1 };
        const u8* bytes = reinterpret_cast<const u8*>(stream.data());
        size_t at = 12;                                               // first chunk behind the 12-byte file header
        while (at + 8 <= (size_t)n) {
            u32 tag, len; memcpy(&tag, bytes + at, 4); memcpy(&len, bytes + at + 4, 4);
            if (tag == 0x4c544433u) break;                            // '3DTL'
            at += 8 + (((size_t)len + 3) & ~(size_t)3);
        }
        if (at + 8 + 76 <= (size_t)n) {
            std::vector<u32> bad(stream);
            u8* hb = reinterpret_cast<u8*>(bad.data()) + at + 8;
            const u32 colorCnt = 6, typeCnt = 0x80000001u;
            memcpy(hb, &colorCnt, 4); memcpy(hb + 4, &typeCnt, 4);
            bool pre = YAIK_DecodeImagePre(lib, bad.data(), (u32)n, &di);
            di.outputImage = outImg.data(); di.outputImageStride = di.width * 4;
            neg[0] = (pre && YAIK_DecodeImage(bad.data(), (u32)n, &di)) ? 1 : 0; neg[1] = (int)YAIK_GetErrorCode();
            std::vector<u32> bad2(stream);                            // an index stream that claims to be 3 GB long
            u8* hb2 = reinterpret_cast<u8*>(bad2.data()) + at + 8;
            const u32 huge = 0xC0000000u; memcpy(hb2 + 8, &huge, 4);
            pre = YAIK_DecodeImagePre(lib, bad2.data(), (u32)n, &di);
            di.outputImage = outImg.data(); di.outputImageStride = di.width * 4;
            neg[2] = (pre && YAIK_DecodeImage(bad2.data(), (u32)n, &di)) ? 1 : 0; neg[3] = (int)YAIK_GetErrorCode();
        }
        blob("yaik_lut_crafted", neg, sizeof neg);
    }
    YAIK_Release(lib);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: host_driver in.bin out.blobs [mode3]\n"); return 2; }
    const bool mode3 = argc > 3 && atoi(argv[3]) != 0;
    FILE* fi = fopen(argv[1], "rb"); if (!fi) return 2;
    int hdr[3]; if (fread(hdr, 4, 3, fi) != 3) return 2;
    const int w = hdr[0], h = hdr[1], np = hdr[2];
    Image* img = Image::CreateImage(w, h, np, false);
    for (int p = 0; p < np; p++) if (fread(img->GetPlane(p)->GetPixels(), 4, (size_t)w * h, fi) != (size_t)w * h) return 2;
    fclose(fi);
    gOut = fopen(argv[2], "wb"); if (!gOut) return 2;

    EncoderContext* ctx = new EncoderContext();
    if (!ctx->SetImageToEncode(img)) { fprintf(stderr, "%s\n", ctx->LastError()); return 3; }
    if (argc > 4 && std::string(argv[3]) == "lut") {
        const int rc = run_lut(ctx, img, w, h, np, argv[4]);
        fclose(gOut);
        ctx->SetImageToEncode(nullptr); ctx->Release(); delete ctx;
        return rc;
    }
    if (argc > 4 && std::string(argv[3]) == "stripes") {
        // the image as row stripes over the devices of the node, then the whole-image passes for comparison (tests/test_gpu_host_mirror.py)
        const int n = atoi(argv[4]);
        const int have = yk_device_count();
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; i++) devs[(size_t)i] = have > 0 ? i % have : 0;
        EncoderContext::StripeTileMaps maps;
        if (!ctx->ConvertHotPathStripes(devs.data(), n, mode3, &maps)) { fprintf(stderr, "ConvertHotPathStripes: %s\n", ctx->LastError()); return 4; }
        for (int i = 0; i < 7; i++) blob(nm("st_bitmap", i), maps.bitmap[i].data(), maps.bitmap[i].size());
        for (int p = 0; p < 3; p++) {
            blob(nm("st_defs", p), maps.defs[p].data(), maps.defs[p].size() * 2);
            blob(nm("st_nibbles", p), maps.nibbles[p].data(), maps.nibbles[p].size());
            unsigned long long nn = maps.nNibbles[p]; blob(nm("st_nn", p), &nn, sizeof nn);
        }
        int info[6] = { maps.bounds[0], maps.bounds[1], maps.bounds[2], maps.bounds[3], maps.stripes, maps.gatherRanks };
        blob("st_info", info, sizeof info);
        if (np == 4) ctx->MipPrefilter(true);
        static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
        for (int i = 0; i < 7; i++) {
            ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), nullptr, false, passes[i][0], passes[i][1]);
            blob(nm("wh_bitmap", i), ctx->LastGradientBitmap().data(), ctx->LastGradientBitmap().size());
        }
        for (int p = 0; p < 3; p++) {
            ctx->DynamicTileEncode(false, img->GetPlane(p), nullptr, false, false, false, false);
            blob(nm("wh_defs", p), ctx->LastTileDefs().data(), ctx->LastTileDefs().size() * 2);
            blob(nm("wh_nibbles", p), ctx->LastTileIndexStream().data(), ctx->LastTileIndexStream().size());
            unsigned long long nn = ctx->LastTileIndexCount(); blob(nm("wh_nn", p), &nn, sizeof nn);
        }
        int wb[4] = { ctx->boundX0, ctx->boundY0, ctx->boundX1, ctx->boundY1 };
        blob("wh_bounds", wb, sizeof wb);
        if (*ctx->LastError()) { fprintf(stderr, "%s\n", ctx->LastError()); return 4; }
        fclose(gOut);
        ctx->SetImageToEncode(nullptr); ctx->Release(); delete ctx;
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "pp") {
        const int rc = run_partial(ctx, img, w, h, np);
        fclose(gOut);
        ctx->SetImageToEncode(nullptr); ctx->Release(); delete ctx;
        return rc;
    }
    ctx->outFile = tmpfile();                                      // the passes append their chunks here, like the reference's outFile
    if (!ctx->outFile) return 2;
    if (np == 4) {
        ctx->MipPrefilter(true);
        int b[6] = { ctx->boundX0, ctx->boundY0, ctx->boundX1, ctx->boundY1, ctx->mipMapTileSize, ctx->remainingPixels };
        blob("mip_bounds", b, sizeof b);
        blob("_mip_bitmap", ctx->MipmapBitmap().data(), ctx->MipmapBitmap().size());
    }
    ctx->PrepareQuadSmooth();
    static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    Image* preview = Image::CreateImage(w, h, 3, true);
    int counts[7];
    for (int i = 0; i < 7; i++) {
        counts[i] = ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), preview, (i & 1) != 0 /* useYCoCg: ignored, like the reference (:3727) */, passes[i][0], passes[i][1]);
        blob(nm("grad_bitmap", i), ctx->LastGradientBitmap().data(), ctx->LastGradientBitmap().size());
        blob(nm("grad_rgbraw", i), ctx->LastGradientRGBStream().data(), ctx->LastGradientRGBStream().size());
    }
    blob("grad_counts", counts, sizeof counts);
    for (int p = 0; p < 3; p++) {                                  // testOutput as FittingQuadSmooth left it
        std::vector<short> p16((size_t)w * h);
        for (size_t i = 0; i < p16.size(); i++) p16[i] = (short)preview->GetPlane(p)->GetPixels()[i];
        blob(nm("preview", p), p16.data(), p16.size() * 2);
    }
    for (int p = 0; p < 3; p++) {
        Plane* dst = new Plane(w, h);
        BoundingBox full = dst->GetRect(); dst->Fill(full, -1);
        ctx->DynamicTileEncode(mode3, img->GetPlane(p), dst, false, false, false, false);
        blob(nm("plnt_defs", mode3 ? 1 : 0, p), ctx->LastTileDefs().data(), ctx->LastTileDefs().size() * 2);
        blob(nm("plnt_idx", mode3 ? 1 : 0, p), ctx->LastTileIndexStream().data(), ctx->LastTileIndexStream().size());
        std::vector<short> d16((size_t)w * h);
        for (size_t i = 0; i < d16.size(); i++) d16[i] = (short)dst->GetPixels()[i];
        blob(nm("plnt_dst", mode3 ? 1 : 0, p), d16.data(), d16.size() * 2);
        delete dst;
    }
    std::vector<u8> pix((size_t)w * h * 3 + 64);
    u8* wr = pix.data();
    for (int p = 0; p < 3; p++) wr = ctx->DynamicTileCompressor(wr, img->GetPlane(p), nullptr, nullptr);
    blob("d1_pix", pix.data(), (size_t)(wr - pix.data()));
    blob("d1_type", ctx->TileTypeStream1D().data(), ctx->TileTypeStream1D().size());
    ctx->GenerateDynamicTileChunk(pix.data(), (int)(wr - pix.data()));
    {
        fflush(ctx->outFile);
        const long all = ftell(ctx->outFile);
        std::vector<u8> file((size_t)all);
        fseek(ctx->outFile, 0, SEEK_SET);
        if (all && fread(file.data(), 1, file.size(), ctx->outFile) != file.size()) return 2;
        blob("chunks_file", file.data(), file.size());
        fclose(ctx->outFile); ctx->outFile = nullptr;
    }
    if (argc > 4) {
        // encode to a .yaik stream, then decode it back like an application would
        FILE* yf = fopen(argv[4], "wb+"); if (!yf) return 2;
        if (!ctx->ConvertHotPath(yf)) { fprintf(stderr, "ConvertHotPath: %s\n", ctx->LastError()); return 4; }
        fflush(yf);
        const long n = ftell(yf);
        std::vector<u32> stream(((size_t)n + 3) / 4);                // 4-byte aligned, as YAIK.h requires
        fseek(yf, 0, SEEK_SET);
        if (fread(stream.data(), 1, (size_t)n, yf) != (size_t)n) return 2;
        fclose(yf);
        blob("yaik_file", stream.data(), (size_t)n);
        {   // the same image through the threaded entropy stage: the file must be identical, byte for byte
            FILE* pf = tmpfile(); if (!pf) return 2;
            if (!ctx->ConvertHotPathParallel(pf, 4)) { fprintf(stderr, "ConvertHotPathParallel: %s\n", ctx->LastError()); return 4; }
            fflush(pf);
            const long pn = ftell(pf);
            std::vector<u8> pbytes((size_t)pn);
            fseek(pf, 0, SEEK_SET);
            if (pn && fread(pbytes.data(), 1, pbytes.size(), pf) != pbytes.size()) return 2;
            fclose(pf);
            blob("yaik_file_parallel", pbytes.data(), pbytes.size());
        }
        YAIK_LIB lib = YAIK_Init(1, nullptr);
        if (!lib) { fprintf(stderr, "YAIK_Init failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        YAIK_SDecodedImage di;
        if (!YAIK_DecodeImagePre(lib, stream.data(), (u32)n, &di)) { fprintf(stderr, "Pre failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        const int bpp = di.hasAlpha ? 4 : 3;
        std::vector<u8> outImg((size_t)di.width * di.height * bpp);
        di.outputImage = outImg.data(); di.outputImageStride = di.width * bpp;
        if (!YAIK_DecodeImage(stream.data(), (u32)n, &di)) { fprintf(stderr, "Decode failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        int dims[3] = { di.width, di.height, bpp };
        blob("yaik_dims", dims, sizeof dims);
        blob("yaik_image", outImg.data(), outImg.size());
        // second decode with a custom image builder: receives the 8x8-tiled planes
        static std::vector<u8> tiled;
        if (!YAIK_DecodeImagePre(lib, stream.data(), (u32)n, &di)) return 5;
        di.outputImage = outImg.data(); di.outputImageStride = di.width * bpp;
        di.customImageOutput = [](YAIK_SDecodedImage* u, YAIK_SCustomDataSource* s) {
            const size_t planeSize = (size_t)(u->width / 8) * (u->height / 8) * 64;
            tiled.assign(s->planeR, s->planeR + planeSize);
            tiled.insert(tiled.end(), s->planeG, s->planeG + planeSize);
            tiled.insert(tiled.end(), s->planeB, s->planeB + planeSize);
        };
        if (!YAIK_DecodeImage(stream.data(), (u32)n, &di)) { fprintf(stderr, "Decode (custom builder) failed: %d\n", (int)YAIK_GetErrorCode()); return 5; }
        blob("yaik_planes_tiled", tiled.data(), tiled.size());
        // error convention: a second Decode without Pre must fail with the sticky code, then read back as NO_ERROR
        const bool again = YAIK_DecodeImage(stream.data(), (u32)n, &di);
        int errs[3] = { again ? 1 : 0, (int)YAIK_GetErrorCode(), (int)YAIK_GetErrorCode() };
        blob("yaik_error_convention", errs, sizeof errs);
        // malformed streams must be refused with the reference's codes, never crash: wrong magic, unknown chunk tag, chunk running past the end
        int neg[6];
        {
            std::vector<u32> bad(stream); bad[0] ^= 0x01010101u;
            neg[0] = YAIK_DecodeImagePre(lib, bad.data(), (u32)n, &di) ? 1 : 0; neg[1] = (int)YAIK_GetErrorCode();
        }
        {
            std::vector<u32> bad(stream); bad[3] = 0x58585858u;           // first chunk tag (after the 12-byte file header) -> 'XXXX'
            bool ok = YAIK_DecodeImagePre(lib, bad.data(), (u32)n, &di);
            di.outputImage = outImg.data(); di.outputImageStride = di.width * bpp;
            neg[2] = (ok && YAIK_DecodeImage(bad.data(), (u32)n, &di)) ? 1 : 0; neg[3] = (int)YAIK_GetErrorCode();
        }
        {
            std::vector<u32> bad(stream); bad[4] = 0x7FFFFFF0u;           // first chunk length far beyond the stream
            bool ok = YAIK_DecodeImagePre(lib, bad.data(), (u32)n, &di);
            di.outputImage = outImg.data(); di.outputImageStride = di.width * bpp;
            neg[4] = (ok && YAIK_DecodeImage(bad.data(), (u32)n, &di)) ? 1 : 0; neg[5] = (int)YAIK_GetErrorCode();
        }
        blob("yaik_malformed", neg, sizeof neg);
        YAIK_Release(lib);
    }
    fclose(gOut);
    ctx->SetImageToEncode(nullptr);
    ctx->Release();
    delete ctx;
    return 0;
}
