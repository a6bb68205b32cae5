// Host side of the drop-in `EncoderContext` for the tile hot path: thin caller of the C-ABI (include/yaik_hip.h).
// Mirrors the reference's call contract (encoder/EncoderContext.cpp:1227-1427, 2784-2797, 3710-4363, 4365-4602, 8398-8522);
// every pixel is processed by the HIP kernels, nothing is computed here.
#include <cstring>
#include "EncoderContext.h"
#include <climits>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include "../../include/yaik_hip.h"
#include "chunks.h"
#include "yaik_format.h"
#include "palette.h"
#include "zstd_dl.h"

static const int kPass[7][2] = { {4, 4}, {4, 3}, {3, 4}, {3, 3}, {3, 2}, {2, 3}, {2, 2} };      // EncoderContext.cpp:9057-9093

EncoderContext::EncoderContext()
    : colorCompressionQuad(250), colorCompressionLUT3D(250), colorCompression1D(255), rangeCompression1D(15),
      mipMapTileSize(16), boundX0(0), boundY0(0), boundX1(0), boundY1(0), remainingPixels(0),
      dumpImage(false), evaluateLUT(false), evaluateLUT2D(false), outFile(nullptr), fileOutSize(0), device(0),
      original(nullptr), ctx(nullptr), bound(false), alphaDone(false), encoded(false), enc3(false), encDst(false), oneDReady(false),
      encReject(3), nextPass(0), nNibbles(0), cursor1d(0), mipHasChunk(false), lutMatched(0) { correlationPatternCount3D = 0; }

EncoderContext::~EncoderContext() { Release(); }

bool EncoderContext::fail(const char* what) {
    err = what;
    if (ctx) { const char* e = yk_last_error(ctx); if (e && *e) { err += ": "; err += e; } }
    printf("ERR %s\n", err.c_str());                    // the reference reports with printf and carries on (kassert, :24-28)
    return false;
}

void EncoderContext::Release() {
    ConvertHotPathFinish();
    if (ctx) { yk_destroy(ctx); ctx = nullptr; }
    if (original) { delete original; original = nullptr; }
    bound = alphaDone = encoded = oneDReady = false;
}

bool EncoderContext::SetImageToEncode(Image* newImage) {
    if (original) delete original;
    original = newImage;
    bound = alphaDone = encoded = oneDReady = false; nextPass = 0;
    if (!original) return false;
    if (!ctx && yk_create(device, &ctx) != YK_OK) return fail("no usable HIP device (this path has no CPU fallback)");
    // Convert() always runs the live 1-D path behind the gradient passes (:9451-9465): let the fused kernel leave it the uncovered cells' pixels,
    // so that the planes are read from HBM once on the whole path
    yk_set_pixel_cache(ctx, 1);
    const int w = original->GetWidth(), h = original->GetHeight(), n = original->HasAlpha() ? 4 : 3;
    if (yk_set_image(ctx, w, h, n, 0, h, 0) != YK_OK) return fail("yk_set_image");
    const int32_t* p[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int i = 0; i < n; i++) p[i] = original->GetPlane(i)->GetPixels();
    if (yk_upload_planes(ctx, p, w) != YK_OK) return fail("yk_upload_planes");
    bound = true;
    return true;
}

void EncoderContext::PrepareQuadSmooth() {}

void EncoderContext::CheckMipmapMask() {
    // the reference lazily creates an all-255 mask and full-image bounds when MipPrefilter did not run (:2784-2794)
    if (!alphaDone && original) { boundX0 = 0; boundY0 = 0; boundX1 = original->GetWidth(); boundY1 = original->GetHeight(); }
}

void EncoderContext::MipPrefilter(bool /*active*/) {
    if (!bound) { fail("MipPrefilter: SetImageToEncode first"); return; }
    mipBitmap.clear(); mipHasChunk = false;
    if (original->HasAlpha()) {
        if (yk_alpha_reject(ctx) != YK_OK || yk_alpha_finish(ctx, nullptr) != YK_OK) { fail("alpha reject"); return; }
    }
    int32_t b[4], tb[4]; int has = 0, rem = 0;
    if (yk_alpha_result(ctx, b, &has, &rem, tb) != YK_OK) { fail("yk_alpha_result"); return; }
    boundX0 = b[0]; boundY0 = b[1]; boundX1 = b[2]; boundY1 = b[3]; remainingPixels = rem; mipMapTileSize = 16; mipHasChunk = has != 0;
    if (has) {
        mipBitmap.resize(((size_t)tb[2] * tb[3] + 7) / 8 + 8);
        size_t nb = 0;
        if (yk_alpha_bitmap(ctx, mipBitmap.data(), mipBitmap.size(), &nb) != YK_OK) { fail("yk_alpha_bitmap"); return; }
        mipBitmap.resize(nb);
        if (outFile) {                                                     // 'MIPM' chunk, mipmapLevel = maxMipLevel + 1 = 4 (:1367-1396)
            const int tbi[4] = { tb[0], tb[1], tb[2], tb[3] };
            if (!yaikchunk::writeMipmap(outFile, tbi, 4, mipBitmap.data(), mipBitmap.size())) { fail("MipPrefilter: fwrite"); return; }
        }
    }
    alphaDone = true; encoded = false; nextPass = 0;
}

bool EncoderContext::ensureEncoded(int rejectFactor, bool mode3, bool wantDst) {
    if (encoded && encReject == rejectFactor && enc3 == mode3 && (encDst || !wantDst)) return true;
    if (original->HasAlpha() && !alphaDone) MipPrefilter(true);          // the reference's order: MipPrefilter precedes the tile passes
    else CheckMipmapMask();
    if (wantDst) yk_set_dst_fill(ctx, INT_MIN);
    if (yk_encode_tiles(ctx, rejectFactor, mode3 ? 1 : 0, wantDst ? 1 : 0) != YK_OK) return fail("yk_encode_tiles");
    encoded = true; encReject = rejectFactor; enc3 = mode3; encDst = wantDst; oneDReady = false;
    return true;
}

// testOutput (:4096-4104): the pass's accepted tiles write their blendC6Exp preview into the planes that took part
static bool copyPreview(yk_ctx* ctx, int pass, int planeBit, Image* testOutput, int w, int h) {
    std::vector<int32_t> buf((size_t)3 * w * h);
    if (yk_gradient_preview(ctx, pass, buf.data(), buf.size()) != YK_OK) return false;
    for (int n = 0; n < 3; n++) {
        if (!((planeBit >> n) & 1) || !testOutput->GetPlane(n)) continue;
        int* d = testOutput->GetPlane(n)->GetPixels();
        const int32_t* src = buf.data() + (size_t)n * w * h;
        for (size_t i = 0; i < (size_t)w * h; i++) if (src[i] != INT_MIN) d[i] = src[i];
    }
    return true;
}

int EncoderContext::FittingQuadSmooth(int rejectFactor, Plane* a, Plane* b, Plane* c, Image* testOutput, bool useYCoCg,
                                      int tileBitSizeX, int tileBitSizeY) {
    if (!bound) { fail("FittingQuadSmooth: SetImageToEncode first"); return 0; }
    (void)useYCoCg;      // ignored, like the reference: its only use (selecting YCoCgImg as the source) is commented out at EncoderContext.cpp:3727
    if ((a && a != original->GetPlane(0)) || (b && b != original->GetPlane(1)) || (c && c != original->GetPlane(2)) || (!a && !b && !c)) {
        fail("FittingQuadSmooth: srcA/B/C must be planes 0/1/2 of the image or NULL"); return 0;
    }
    const int planeBit = (a ? 1 : 0) | (b ? 2 : 0) | (c ? 4 : 0);           // PlaneBit (:3715)
    if (planeBit != 7) {
        // a pass over a subset of the planes (Convert()'s RB, RG, GB, R, G, B 4x4 passes, :9261-9415): it continues from the state the
        // seven RGB passes left, so those come first
        if (!encoded || nextPass != 7) { fail("FittingQuadSmooth: partial-plane passes follow the seven RGB passes"); return 0; }
        int tiles = 0;
        if (yk_gradient_partial_pass(ctx, rejectFactor, planeBit, tileBitSizeX, tileBitSizeY, &tiles) != YK_OK) { fail("yk_gradient_partial_pass"); return 0; }
        oneDReady = false;
        size_t nb = 0;
        yk_partial_bitmap(ctx, nullptr, 0, &nb); gradBitmap.resize(nb);
        if (nb && yk_partial_bitmap(ctx, gradBitmap.data(), nb, nullptr) != YK_OK) { fail("yk_partial_bitmap"); return 0; }
        yk_partial_corners(ctx, nullptr, 0, &nb); gradRgb.resize(nb);
        if (nb && yk_partial_corners(ctx, gradRgb.data(), nb, nullptr) != YK_OK) { fail("yk_partial_corners"); return 0; }
        if (outFile && !evaluateLUT) {
            std::string e;
            const long before = ftell(outFile);
            if (yaikchunk::writeGradientTile(outFile, original->GetWidth(), original->GetHeight(), tileBitSizeX, tileBitSizeY, gradBitmap.data(),
                                             gradBitmap.size(), gradRgb.data(), gradRgb.size(), colorCompressionQuad, planeBit, e) < 0) { fail(("FittingQuadSmooth: " + e).c_str()); return 0; }
            fileOutSize += (int)(ftell(outFile) - before);
        }
        if (testOutput && !copyPreview(ctx, 7, planeBit, testOutput, original->GetWidth(), original->GetHeight())) { fail("yk_gradient_preview"); return 0; }
        return tiles;
    }
    int pass = -1;
    for (int i = 0; i < 7; i++) if (kPass[i][0] == tileBitSizeX && kPass[i][1] == tileBitSizeY) pass = i;
    if (pass < 0 || pass != nextPass % 7) { fail("FittingQuadSmooth: passes must follow the shipped order 16x16,16x8,8x16,8x8,8x4,4x8,4x4"); return 0; }
    if (pass == 0) encoded = false;                                       // a new sequence re-encodes (planes may have changed)
    if (!ensureEncoded(rejectFactor, enc3, false)) return 0;
    nextPass = pass + 1;
    gradBitmap.resize(yk_gradient_bitmap_bytes(ctx, pass));
    if (yk_gradient_bitmap(ctx, pass, gradBitmap.data(), gradBitmap.size()) != YK_OK) { fail("yk_gradient_bitmap"); return 0; }
    size_t nb = 0;
    const size_t cap = (size_t)(original->GetWidth() / 4 + 1) * (original->GetHeight() / 4 + 2) * 3 + 16;
    gradRgb.resize(cap);
    if (yk_gradient_corners(ctx, pass, gradRgb.data(), cap, &nb) != YK_OK) { fail("yk_gradient_corners"); return 0; }
    gradRgb.resize(nb);
    if (outFile && !evaluateLUT) {                                         // 'GTIL' chunk (:4239-4347), full RGB pass: PlaneBit 7
        std::string e;
        const long before = ftell(outFile);
        if (yaikchunk::writeGradientTile(outFile, original->GetWidth(), original->GetHeight(), tileBitSizeX, tileBitSizeY, gradBitmap.data(),
                                         gradBitmap.size(), gradRgb.data(), gradRgb.size(), colorCompressionQuad, 7, e) < 0) { fail(("FittingQuadSmooth: " + e).c_str()); return 0; }
        fileOutSize += (int)(ftell(outFile) - before);
    }
    if (testOutput && !copyPreview(ctx, pass, 7, testOutput, original->GetWidth(), original->GetHeight())) { fail("yk_gradient_preview"); return 0; }
    int tiles = 0;
    for (u8 v : gradBitmap) tiles += __builtin_popcount(v);               // TileDone (:4362)
    return tiles;
}

int EncoderContext::DynamicTileEncode(bool mode3BitOnly, Plane* plane, Plane* dst, bool isCo, bool isCg, bool isHalfX, bool isHalfY) {
    if (!bound) { fail("DynamicTileEncode: SetImageToEncode first"); return 0; }
    if (isCo || isCg || isHalfX || isHalfY) { fail("DynamicTileEncode: only full-resolution RGB planes are on this path"); return 0; }
    int p = -1;
    for (int i = 0; i < 3; i++) if (plane == original->GetPlane(i)) p = i;
    if (p < 0) { fail("DynamicTileEncode: plane must be plane 0..2 of the image"); return 0; }
    if (!ensureEncoded(encoded ? encReject : 3, mode3BitOnly, dst != nullptr)) return 0;
    size_t nd = 0, nn = 0;
    if (yk_range_sizes(ctx, p, &nd, &nn) != YK_OK) { fail("yk_range_sizes"); return 0; }
    tileDefs.assign(nd, 0); tileIdx.assign((nn + 1) / 2, 0); nNibbles = nn;
    if (yk_range_streams(ctx, p, tileDefs.data(), nd, tileIdx.data(), tileIdx.size()) != YK_OK) { fail("yk_range_streams"); return 0; }
    if (dst) {                                                             // decoded values land in dst where a pixel was coded (:4448-4457)
        const size_t n = (size_t)original->GetWidth() * original->GetHeight();
        std::vector<int32_t> tmp(n);
        if (yk_range_dst(ctx, p, tmp.data(), n) != YK_OK) { fail("yk_range_dst"); return 0; }
        int* d = dst->GetPixels();
        for (size_t i = 0; i < n; i++) if (tmp[i] != INT_MIN) d[i] = tmp[i];
    }
    if (outFile) {                                                         // 'PLNT' chunk (:4516-4589); constraint = 8-aligned bound box (:4386-4391)
        BoundingBox cb;
        cb.x = (s16)((boundX0 >> 3) << 3); cb.y = (s16)((boundY0 >> 3) << 3);
        cb.w = (s16)((((boundX1 + 7) >> 3) << 3) - cb.x); cb.h = (s16)((((boundY1 + 7) >> 3) << 3) - cb.y);
        std::string e;
        if (!yaikchunk::writePlaneTile(outFile, cb, tileDefs.data(), tileDefs.size(), tileIdx.data(), tileIdx.size(), 0, false, false, e)) { fail(("DynamicTileEncode: " + e).c_str()); return 0; }
    }
    return 0;                                                              // the reference returns layerSize, which it never updates (:4407,:4601)
}

u8* EncoderContext::DynamicTileCompressor(u8* stream, Plane* src, Plane* /*map*/, Plane* /*debug*/) {
    if (!bound || !stream) { fail("DynamicTileCompressor: SetImageToEncode first"); return stream; }
    int p = -1;
    for (int i = 0; i < 3; i++) if (src == original->GetPlane(i)) p = i;
    if (p < 0) { fail("DynamicTileCompressor: src must be plane 0..2 of the image"); return stream; }
    if (!ensureEncoded(encoded ? encReject : 3, enc3, false)) return stream;
    if (!oneDReady) {
        if (yk_range1d_encode(ctx) != YK_OK) { fail("yk_range1d_encode"); return stream; }
        size_t np = 0, nt = 0;
        yk_range1d_streams(ctx, nullptr, 0, &np, nullptr, 0, &nt);
        pix1d.resize(np); type1d.resize(nt);
        if (yk_range1d_streams(ctx, pix1d.data(), np, nullptr, type1d.data(), nt, nullptr) != YK_OK) { fail("yk_range1d_streams"); return stream; }
        oneDReady = true;
    }
    size_t pixEnd[3] = {0, 0, 0};                                          // equal thirds unless a partial-plane pass ran
    if (yk_range1d_plane_ends(ctx, pixEnd, nullptr) != YK_OK) { fail("yk_range1d_plane_ends"); return stream; }
    const size_t from = p ? pixEnd[p - 1] : 0, per = pixEnd[p] - from;
    memcpy(stream, pix1d.data() + from, per);
    return stream + per;
}

// ---- (f)4 3-D LUT tiles ----------------------------------------------------------------------------------------------------------
void EncoderContext::Load3DPattern(const char* fileName) {
    FILE* f = fopen(fileName, "rb");
    if (!f) return;                                                        // a missing bank file is skipped silently (:7853-7854)
    u8 count = 0, r[256], g[256], b[256];
    bool ok = fread(&count, 1, 1, f) == 1 && fread(r, 1, count, f) == count && fread(g, 1, count, f) == count && fread(b, 1, count, f) == count;
    fclose(f);
    if (!ok) { fail("Load3DPattern: short file"); return; }
    if (!ctx) { fail("Load3DPattern: SetImageToEncode first (the bank lives on the GPU handle)"); return; }
    int idx = -1;
    if (yk_lut_load_pattern(ctx, r, g, b, count, &idx) != YK_OK) { fail("Load3DPattern: pattern refused (1..64 points of 6 bits, at most 64 patterns)"); return; }
    std::vector<u8> p((size_t)count * 3);
    for (int n = 0; n < count; n++) { p[n * 3] = r[n]; p[n * 3 + 1] = g[n]; p[n * 3 + 2] = b[n]; }
    lutPatterns.push_back(p);
    correlationPatternCount3D = idx + 1;
}

bool EncoderContext::Save3DLutFile(const char* fileName) {
    if (!ctx || correlationPatternCount3D == 0) return fail("Save3DLutFile: no pattern loaded");
    std::vector<u8> file(sizeof(yaikfmt::LUTHeader) + (size_t)(64 + 32 + 16 + 8) * 3 * correlationPatternCount3D, 0);
    yaikfmt::LUTHeader hd; memset(&hd, 0, sizeof hd);
    hd.lutH[0] = 'L'; hd.lutH[1] = 'U'; hd.lutH[2] = 'L'; hd.lutH[3] = '0'; hd.entryCount = (u8)(correlationPatternCount3D - 1); hd.padding_extension[0] = 1;
    memcpy(file.data(), &hd, sizeof hd);
    std::vector<int16_t> fac((size_t)correlationPatternCount3D * 4 * 3 * 64);
    for (int e = 0; e < correlationPatternCount3D; e++)
        if (yk_lut_pattern_tables(ctx, e, fac.data() + (size_t)e * 4 * 3 * 64, nullptr, nullptr) != YK_OK) return fail("yk_lut_pattern_tables");
    u8* w = file.data() + sizeof hd;
    for (int depth = 3; depth >= 0; depth--)                               // 3, 4, 5, 6 bit (BinarySave3D per depth and pattern, :7836-7841)
        for (int e = 0; e < correlationPatternCount3D; e++)
            for (int c = 0; c < 3; c++) for (int m = 0; m < (64 >> depth); m++) *w++ = (u8)fac[(((size_t)e * 4 + depth) * 3 + c) * 64 + m];
    FILE* f = fopen(fileName, "wb");
    if (!f) return fail("Save3DLutFile: cannot open the file");
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    fclose(f);
    return ok || fail("Save3DLutFile: fwrite");
}

void EncoderContext::StartCorrelationSearch(bool is3D) {
    if (!is3D) { fail("StartCorrelationSearch: the 2-D correlation mode is deprecated in the reference and not on this path"); return; }
    if (!bound) { fail("StartCorrelationSearch: SetImageToEncode first"); return; }
    if (!encoded || nextPass != 7) { fail("StartCorrelationSearch: the 3-D LUT search follows the seven RGB gradient passes"); return; }
    if (yk_lut_start(ctx) != YK_OK) { fail("yk_lut_start"); return; }
    oneDReady = false;
}

void EncoderContext::Correlation3DSearch(Image* input, Image* /*output*/, int tileShiftX, int tileShiftY) {
    if (!bound || input != original) { fail("Correlation3DSearch: input must be the image to encode"); return; }
    lutMatched = 0;
    if (yk_lut_search(ctx, tileShiftX, tileShiftY, &lutMatched) != YK_OK) { fail("yk_lut_search"); return; }
    oneDReady = false;
}

void EncoderContext::EndCorrelationSearch(bool is3D, u8 component) {
    if (!is3D || !bound) { fail("EndCorrelationSearch: 3-D search on a bound image only"); return; }
    std::vector<u8> buf[12];
    for (int which = 0; which < 12; which++) {
        size_t n = 0;
        if (yk_lut_stream(ctx, which, nullptr, 0, &n) != YK_OK) { fail("yk_lut_stream"); return; }
        buf[which].assign(n, 0);
        if (n && yk_lut_stream(ctx, which, buf[which].data(), n, nullptr) != YK_OK) { fail("yk_lut_stream"); return; }
    }
    if (!outFile) return;
    yaikchunk::Tile3DStreams st;
    for (int k = 0; k < 6; k++) { st.map[k] = buf[6 + k].data(); st.mapBytes[k] = buf[6 + k].size(); }
    st.tileType = reinterpret_cast<const u16*>(buf[0].data()); st.nTiles = buf[0].size() / 2; st.color = buf[1].data();
    for (int b = 0; b < 4; b++) { st.idx[b] = buf[2 + b].data(); st.nIdx[b] = buf[2 + b].size(); }
    std::string e;
    const long before = ftell(outFile);
    if (!yaikchunk::writeTile3D(outFile, st, colorCompressionLUT3D, component, e)) { fail(("EndCorrelationSearch: " + e).c_str()); return; }
    fileOutSize += (int)(ftell(outFile) - before);
}

void EncoderContext::GenerateDynamicTileChunk(u8* stream, int sizeStream) {
    if (!outFile || sizeStream <= 0) return;
    std::string e;
    if (!yaikchunk::writeTile1D(outFile, stream, (size_t)sizeStream, type1d.data(), type1d.size(), colorCompression1D, rangeCompression1D, e))
        fail(("GenerateDynamicTileChunk: " + e).c_str());
}

bool EncoderContext::ConvertHotPath(FILE* f) {
    if (!bound || !f) return fail("ConvertHotPath: SetImageToEncode and an open file first");
    FILE* saved = outFile; outFile = f; fileOutSize = 0; err.clear();
    PaletteResetCodeBook();                             // the reference converts one image per process: its code table starts zeroed
    const int w = original->GetWidth(), h = original->GetHeight();
    bool ok = yaikchunk::writeFileHeader(f, w, h, original->HasAlpha());
    if (ok && original->HasAlpha()) MipPrefilter(true);
    PrepareQuadSmooth();
    for (int i = 0; ok && i < 7; i++) {
        FittingQuadSmooth(3, original->GetPlane(0), original->GetPlane(1), original->GetPlane(2), nullptr, false, kPass[i][0], kPass[i][1]);
        ok = err.empty();
    }
    if (ok) {
        std::vector<u8> stream((size_t)w * h * 3 + 64);
        u8* wr = stream.data();
        for (int p = 0; p < 3; p++) wr = DynamicTileCompressor(wr, original->GetPlane(p), nullptr, nullptr);
        GenerateDynamicTileChunk(stream.data(), (int)(wr - stream.data()));
        ok = err.empty();
    }
    ok = ok && yaikchunk::writeEndOfFile(f);
    outFile = saved;
    return ok;
}

// ---- row stripes over the GPUs of the node -----------------------------------------------------------------------------------------------
bool EncoderContext::ConvertHotPathStripes(const int* devices, int nStripes, bool mode3BitOnly, StripeTileMaps* out) {
    if (!original || !devices || !out || nStripes < 1 || nStripes > 64) return fail("ConvertHotPathStripes: an image, 1..64 stripes and a result");
    const int w = original->GetWidth(), H = original->GetHeight(), np = original->HasAlpha() ? 4 : 3;
    struct Stripe { yk_ctx* c = nullptr; int y0 = 0, h = 0, halo = 0; void* send = nullptr; size_t bytes = 0; };
    std::vector<Stripe> st((size_t)nStripes);
    std::vector<void*> comms;
    void* recv = nullptr;
    bool ok = true;
    auto bad = [&](const char* what) { ok = false; return fail(what); };
    // the image's 64-row blocks dealt as evenly as possible (swizzle blocks never straddle stripes: per-pass bitmaps concatenate)
    const int blocks = (H + 63) / 64, base = blocks / nStripes, rem = blocks % nStripes;
    for (int i = 0; i < nStripes && ok; i++) {
        const int start = i * base + (i < rem ? i : rem), count = base + (i < rem ? 1 : 0);
        st[i].y0 = start * 64 < H ? start * 64 : H;
        const int y1 = (start + count) * 64 < H ? (start + count) * 64 : H;
        st[i].h = y1 - st[i].y0; st[i].halo = (y1 < H && st[i].h > 0) ? 1 : 0;
        if (st[i].h == 0) continue;                                      // more stripes than blocks: nothing to do for this one
        if (yk_create(devices[i], &st[i].c) != YK_OK) { bad("ConvertHotPathStripes: yk_create"); break; }
        const int32_t* p[4] = { nullptr, nullptr, nullptr, nullptr };
        for (int k = 0; k < np; k++) p[k] = original->GetPlane(k)->GetPixels() + (size_t)st[i].y0 * w;
        if (yk_set_image(st[i].c, w, H, np, st[i].y0, st[i].h, st[i].halo) != YK_OK || yk_upload_planes(st[i].c, p, w) != YK_OK) bad("ConvertHotPathStripes: binding a stripe");
    }
    int box[4] = { 9999999, 9999999, -1, -1 };
    if (ok && np == 4) {                                                 // MipPrefilter per stripe, its kept-tile box combined here (min / max)
        for (auto& s : st) if (s.c && yk_alpha_reject(s.c) != YK_OK) bad("ConvertHotPathStripes: yk_alpha_reject");
        for (auto& s : st) {
            int32_t b[4];
            if (!ok || !s.c) continue;
            if (yk_get_stripe_bbox(s.c, b) != YK_OK) { bad("ConvertHotPathStripes: yk_get_stripe_bbox"); break; }
            box[0] = b[0] < box[0] ? b[0] : box[0]; box[1] = b[1] < box[1] ? b[1] : box[1];
            box[2] = b[2] > box[2] ? b[2] : box[2]; box[3] = b[3] > box[3] ? b[3] : box[3];
        }
        const int32_t gb[4] = { box[0], box[1], box[2], box[3] };
        for (auto& s : st) if (ok && s.c && yk_alpha_finish(s.c, gb) != YK_OK) bad("ConvertHotPathStripes: yk_alpha_finish");
    }
    // every stripe's kernels are queued before anything is waited for: the devices work side by side
    for (auto& s : st) if (ok && s.c && yk_encode_tiles(s.c, 3, mode3BitOnly ? 1 : 0, 0) != YK_OK) bad("ConvertHotPathStripes: yk_encode_tiles");
    for (auto& s : st) {
        if (!ok || !s.c) continue;
        const size_t cap = yk_export_capacity(s.c) + YK_EXPORT_HEADER_BYTES;
        if (yk_device_alloc(s.c, cap, &s.send) != YK_OK || yk_export_tile_maps_framed(s.c, s.send, cap, reinterpret_cast<void*>(-1)) != YK_OK) bad("ConvertHotPathStripes: export");
    }
    std::vector<std::vector<u8>> payload((size_t)nStripes);
    if (ok) {
        for (int i = 0; i < nStripes && ok; i++) {                       // the 128-byte headers tell how much each stripe has to send
            if (!st[i].c) continue;
            unsigned long long hdr[16];
            if (yk_device_download(st[i].c, hdr, st[i].send, sizeof hdr) != YK_OK) { bad("ConvertHotPathStripes: header"); break; }
            st[i].bytes = YK_EXPORT_HEADER_BYTES + (size_t)hdr[0];
        }
        std::vector<int> live; bool distinct = true;
        for (int i = 0; i < nStripes; i++) if (st[i].c) { for (int j : live) if (devices[j] == devices[i]) distinct = false; live.push_back(i); }
        out->stripes = (int)live.size(); out->gatherRanks = 0;
        if (ok && live.size() > 1 && distinct && yk_comm_available()) {  // ONE grouped transfer onto the first stripe's device, one copy to the host
            const int n = (int)live.size();
            std::vector<yk_ctx*> cs; std::vector<void*> snd; std::vector<size_t> bytes, offs; size_t total = 0;
            for (int i : live) { cs.push_back(st[i].c); snd.push_back(st[i].send); bytes.push_back(st[i].bytes); offs.push_back(total); total += (st[i].bytes + 255) & ~(size_t)255; }
            comms.assign((size_t)n, nullptr);
            if (yk_comm_init_all(cs.data(), n, comms.data()) != YK_OK) bad(yk_last_error(cs[0]));
            if (ok && yk_device_alloc(cs[0], total, &recv) != YK_OK) bad("ConvertHotPathStripes: receive buffer");
            if (ok && yk_gather_maps_all(cs.data(), comms.data(), n, 0, snd.data(), bytes.data(), recv, offs.data()) != YK_OK) bad(yk_last_error(cs[0]));
            for (int k = 1; k < n && ok; k++) if (yk_synchronize(cs[k]) != YK_OK) bad("ConvertHotPathStripes: yk_synchronize");
            std::vector<u8> all(total);
            if (ok && yk_device_download(cs[0], all.data(), recv, total) != YK_OK) bad("ConvertHotPathStripes: download");
            for (int k = 0; k < n && ok; k++) payload[(size_t)live[k]].assign(all.begin() + (long)offs[k], all.begin() + (long)(offs[k] + bytes[k]));
            if (ok) { int cnt = 0; yk_comm_ranks(comms[0], &cnt, nullptr); out->gatherRanks = cnt; }
        } else {
            for (int i : live) {
                payload[(size_t)i].resize(st[i].bytes);
                if (ok && yk_device_download(st[i].c, payload[(size_t)i].data(), st[i].send, st[i].bytes) != YK_OK) bad("ConvertHotPathStripes: download");
            }
        }
    }
    if (ok) {                                                            // concatenate: bitmaps and definitions byte-wise, nibble streams with a 4-bit shift
        for (int k = 0; k < 7; k++) out->bitmap[k].clear();
        for (int p = 0; p < 3; p++) { out->defs[p].clear(); out->nibbles[p].clear(); out->nNibbles[p] = 0; }
        for (int i = 0; i < nStripes; i++) {
            if (payload[(size_t)i].empty()) continue;
            const u8* b = payload[(size_t)i].data();
            unsigned long long hdr[16]; memcpy(hdr, b, sizeof hdr);
            size_t off = YK_EXPORT_HEADER_BYTES;
            auto take = [&](size_t n) { const u8* v = b + off; off += (n + 15) & ~(size_t)15; return v; };
            for (int k = 0; k < 7; k++) { const size_t n = (size_t)hdr[1 + k]; const u8* v = take(n); out->bitmap[k].insert(out->bitmap[k].end(), v, v + n); }
            take((size_t)hdr[8]);                                        // keep flags: the stripes' own business
            for (int p = 0; p < 3; p++) {
                const size_t nd = (size_t)hdr[9 + 2 * p], nn = (size_t)hdr[10 + 2 * p];
                const u16* d = reinterpret_cast<const u16*>(take(nd * 2));
                out->defs[p].insert(out->defs[p].end(), d, d + nd);
                const u8* nb = take((nn + 1) / 2);
                std::vector<u8>& dst = out->nibbles[p];
                if ((out->nNibbles[p] & 1) == 0) dst.insert(dst.end(), nb, nb + (nn + 1) / 2);
                else for (size_t q = 0; q < nn; q++) {                  // the stream so far ends in a half byte
                    const u8 v = (u8)((nb[q >> 1] >> ((q & 1) * 4)) & 15);
                    if (((out->nNibbles[p] + q) & 1) == 0) dst.push_back(v); else dst.back() = (u8)(dst.back() | (v << 4));
                }
                out->nNibbles[p] += nn;
                dst.resize((out->nNibbles[p] + 1) / 2);
            }
        }
        if (np == 4) { out->bounds[0] = box[0]; out->bounds[1] = box[1]; out->bounds[2] = box[2]; out->bounds[3] = box[3]; }
        else { out->bounds[0] = 0; out->bounds[1] = 0; out->bounds[2] = w; out->bounds[3] = H; }
    }
    for (void* cm : comms) if (cm) yk_comm_destroy(cm);
    if (recv) yk_device_free(st[0].c ? st[0].c : nullptr, recv);
    for (auto& s : st) { if (s.c) { if (s.send) yk_device_free(s.c, s.send); yk_destroy(s.c); } }
    return ok;
}

// ---- threaded entropy stage -------------------------------------------------------------------------------------------------------
namespace {
// a few worker threads taking jobs from a queue; wait() returns when the queue has drained and every job has finished
class JobPool {
public:
    explicit JobPool(int n) { for (int i = 0; i < (n < 1 ? 1 : n); i++) workers.emplace_back([this] { run(); }); }
    ~JobPool() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); for (auto& t : workers) t.join(); }
    void push(std::function<void()> job) { { std::lock_guard<std::mutex> g(mu); jobs.push_back(std::move(job)); pending++; } cv.notify_one(); }
    void wait() { std::unique_lock<std::mutex> g(mu); done.wait(g, [this] { return pending == 0; }); }
private:
    void run() {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [this] { return stop || !jobs.empty(); }); if (jobs.empty()) return; job = std::move(jobs.front()); jobs.pop_front(); }
            job();
            { std::lock_guard<std::mutex> g(mu); if (--pending == 0) done.notify_all(); }
        }
    }
    std::vector<std::thread> workers; std::deque<std::function<void()>> jobs; std::mutex mu; std::condition_variable cv, done; int pending = 0; bool stop = false;
};
}

struct EncoderContext::EntropyStage {
    FILE* f = nullptr; int w = 0, h = 0, threads = 1, colorQuad = 250, color1D = 255, range1D = 15;
    std::vector<u8> bitmap[7], rgb[7], pix, type;
    std::thread worker; bool ok = true; std::string err;
    void run() {
        PaletteResetCodeBook();                         // like ConvertHotPath: the reference converts one image per process
        std::vector<u8> zBitmap[7], zRgb[7], pal[7], zPix, zType; u32 palSize[7] = {}; bool has[7] = {};
        std::mutex emu; auto bad = [&](const std::string& e) { std::lock_guard<std::mutex> g(emu); if (ok) { ok = false; err = e; } };
        {
            JobPool pool(threads);
            if (!pix.empty()) {                          // the largest stream first
                pool.push([&] { std::string e; if (!yaikchunk::compressStream(pix.data(), pix.size(), 18, zPix, e)) bad(e); });
                pool.push([&] { std::string e; if (!yaikchunk::compressStream(type.data(), type.size(), 18, zType, e)) bad(e); });
            }
            for (int i = 0; i < 7; i++) {
                has[i] = yaikchunk::gradientTileHasChunk(w, h, kPass[i][0], kPass[i][1], bitmap[i].data(), rgb[i].size());
                if (has[i]) pool.push([&, i] { std::string e; if (!yaikchunk::compressStream(bitmap[i].data(), bitmap[i].size(), 18, zBitmap[i], e)) bad(e); });
            }
            for (int i = 0; i < 7; i++) {                // PaletteCompressor in pass order on this thread; each result goes straight to a worker
                if (!has[i]) continue;
                pal[i].resize(rgb[i].size() * 3); palSize[i] = (u32)pal[i].size();
                if (!PaletteCompressor(rgb[i].data(), (int)rgb[i].size(), pal[i].data(), &palSize[i])) { bad("PaletteCompressor overflow"); break; }
                pal[i].resize(palSize[i]);
                pool.push([&, i] { std::string e; if (!yaikchunk::compressStream(pal[i].data(), pal[i].size(), 18, zRgb[i], e)) bad(e); });
            }
            pool.wait();
        }
        std::string e;
        for (int i = 0; ok && i < 7; i++)
            if (has[i] && !yaikchunk::emitGradientTile(f, w, h, kPass[i][0], kPass[i][1], bitmap[i].data(), rgb[i].size(), palSize[i], zBitmap[i], zRgb[i], colorQuad, 7, e)) bad(e);
        if (ok && !pix.empty() && !yaikchunk::emitTile1D(f, pix.size(), type.size(), zPix, zType, color1D, range1D, e)) bad(e);
        if (ok && !yaikchunk::writeEndOfFile(f)) bad("fwrite");
    }
};

bool EncoderContext::ConvertHotPathBegin(FILE* f, int threads) {
    if (!bound || !f) return fail("ConvertHotPathBegin: SetImageToEncode and an open file first");
    while ((int)stages.size() >= kMaxStages) retireOldestStage();
    if (!yaikzstd::available()) return fail(yaikzstd::lastError());      // also: the library is loaded before any worker asks for it
    FILE* saved = outFile; err.clear();
    EntropyStage* st = new EntropyStage();
    st->f = f; st->w = original->GetWidth(); st->h = original->GetHeight(); st->threads = threads;
    st->colorQuad = colorCompressionQuad; st->color1D = colorCompression1D; st->range1D = rangeCompression1D;
    bool ok = yaikchunk::writeFileHeader(f, st->w, st->h, original->HasAlpha());
    outFile = f; fileOutSize = 0;
    if (ok && original->HasAlpha()) MipPrefilter(true);                    // 'MIPM' is not compressed: written at once
    outFile = nullptr;                                                     // the passes only collect their raw streams
    PrepareQuadSmooth();
    for (int i = 0; ok && i < 7; i++) {
        FittingQuadSmooth(3, original->GetPlane(0), original->GetPlane(1), original->GetPlane(2), nullptr, false, kPass[i][0], kPass[i][1]);
        ok = err.empty();
        st->bitmap[i] = gradBitmap; st->rgb[i] = gradRgb;
    }
    if (ok) {
        st->pix.resize((size_t)st->w * st->h * 3 + 64);
        u8* wr = st->pix.data();
        for (int p = 0; p < 3; p++) wr = DynamicTileCompressor(wr, original->GetPlane(p), nullptr, nullptr);
        st->pix.resize((size_t)(wr - st->pix.data()));
        st->type = type1d;
        ok = err.empty();
    }
    outFile = saved;
    if (!ok) { delete st; return false; }
    stages.push_back(st);
    st->worker = std::thread([st] { st->run(); });
    return true;
}

void EncoderContext::retireOldestStage() {
    EntropyStage* st = stages.front();
    stages.erase(stages.begin());
    st->worker.join();
    if (!st->ok && stagesOk) { stagesOk = false; stagesErr = st->err; }
    delete st;
}

bool EncoderContext::ConvertHotPathFinish() {
    while (!stages.empty()) retireOldestStage();
    const bool ok = stagesOk;
    if (!ok) fail(("ConvertHotPath entropy stage: " + stagesErr).c_str());
    stagesOk = true; stagesErr.clear();
    return ok;
}
