// Small C++ program written the way a YAIK maintainer would drive the reference's EncoderContext for this path
// (encoder/ImageEncoder.cpp:158-213 / EncoderContext::Convert): same call sequence, same method names, but linked against
// the MI355X drop-in.  tests/test_gpu_host_mirror.py checks its output blob by blob.
//
// usage: host_driver <in.bin> <out.blobs> [mode3BitOnly]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "EncoderContext.h"

static FILE* gOut;
static void blob(const std::string& name, const void* data, size_t len) {
    u32 nl = (u32)name.size(); unsigned long long dl = len;
    fwrite(&nl, 4, 1, gOut); fwrite(name.data(), 1, nl, gOut); fwrite(&dl, 8, 1, gOut); if (len) fwrite(data, 1, len, gOut);
}
static std::string nm(const char* b, int a, int c = -1) { char t[96]; if (c >= 0) snprintf(t, sizeof t, "%s_%d_%d", b, a, c); else snprintf(t, sizeof t, "%s_%d", b, a); return t; }

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
        counts[i] = ctx->FittingQuadSmooth(3, img->GetPlane(0), img->GetPlane(1), img->GetPlane(2), preview, false, passes[i][0], passes[i][1]);
        blob(nm("grad_bitmap", i), ctx->LastGradientBitmap().data(), ctx->LastGradientBitmap().size());
        blob(nm("grad_rgbraw", i), ctx->LastGradientRGBStream().data(), ctx->LastGradientRGBStream().size());
    }
    blob("grad_counts", counts, sizeof counts);
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
    fclose(gOut);
    ctx->SetImageToEncode(nullptr);
    ctx->Release();
    delete ctx;
    return 0;
}
