#include "chunks.h"
#include <cstring>
#include "palette.h"
#include "yaik_format.h"
#include "zstd_dl.h"

using namespace yaikfmt;

namespace {
bool put(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
bool putChunk(FILE* f, u32 tag, const void* header, size_t headerBytes, const std::vector<const std::vector<u8>*>& parts) {
    size_t base = headerBytes;
    for (auto* p : parts) base += p->size();
    HeaderBase hb; hb.tag = tag; hb.length = (u32)(((base + 3) >> 2) << 2);              // rounded up to 4 (:4315-4316)
    const u8 pad[3] = { 0, 0, 0 };
    bool ok = put(f, &hb, sizeof hb) && put(f, header, headerBytes);
    for (auto* p : parts) ok = ok && put(f, p->data(), p->size());
    return ok && put(f, pad, hb.length - base);
}
bool zcompress(const void* src, size_t n, int level, std::vector<u8>& out, std::string& err) {
    if (!yaikzstd::available()) { err = yaikzstd::lastError(); return false; }
    out.resize(yaikzstd::compressBound(n) + 16);
    const size_t r = yaikzstd::compress(out.data(), out.size(), src, n, level);
    if (r == 0) { err = "ZSTD_compress failed"; return false; }
    out.resize(r);
    return true;
}
}

namespace yaikchunk {

bool writeFileHeader(FILE* f, int width, int height, bool hasAlpha) {
    FileHeader h; memset(&h, 0, sizeof h);
    h.tag = TAG_FILE; h.version = 1; h.width = (u16)width; h.height = (u16)height; h.infoMask = hasAlpha ? 1 : 0;
    return put(f, &h, sizeof h);
}

bool writeEndOfFile(FILE* f) { const u32 t = TAG_END; return put(f, &t, 4); }

bool writeMipmap(FILE* f, const int tb[4], int mipmapLevel, const u8* bits, size_t nBytes) {
    MipmapHeader h; memset(&h, 0, sizeof h);
    h.bbox.x = (s16)tb[0]; h.bbox.y = (s16)tb[1]; h.bbox.w = (s16)tb[2]; h.bbox.h = (s16)tb[3];
    h.version = 1; h.mipmapLevel = (u8)mipmapLevel;                                     // maxMipLevel + 1 (:1384)
    const std::vector<u8> payload(bits, bits + nBytes);
    return putChunk(f, TAG_MIPMAP, &h, sizeof h, { &payload });
}

void gradientExtent(int imgW, int imgH, int sx, int sy, const u8* bitmap, int out[4]) {
    u32 bigX, bigY, bitCount;
    out[0] = imgW; out[1] = imgH; out[2] = 0; out[3] = 0;
    if (!swizzleSize(sx, sy, bigX, bigY, bitCount)) return;
    const int xBB = (imgW + (int)bigX - 1) / (int)bigX, yBB = (imgH + (int)bigY - 1) / (int)bigY;
    const int perRow = (int)bigX >> sx, tx = 1 << sx, ty = 1 << sy;
    for (int by = 0; by < yBB; by++) for (int bx = 0; bx < xBB; bx++) {
        const size_t bit0 = ((size_t)by * xBB + bx) * bitCount;                          // bit index = block * bitCount + ty * perRow + tx (:3801-3805)
        for (u32 b = 0; b < bitCount; b++) {
            const size_t bit = bit0 + b;
            if (!((bitmap[bit >> 3] >> (bit & 7)) & 1)) continue;
            const int x = bx * (int)bigX + (int)(b % perRow) * tx, y = by * (int)bigY + (int)(b / perRow) * ty;
            if (out[0] > x) out[0] = x;
            if (out[1] > y) out[1] = y;
            if (out[2] < x + tx) out[2] = x + tx;
            if (out[3] < y + ty) out[3] = y + ty;
        }
    }
}

bool compressStream(const void* src, size_t n, int level, std::vector<u8>& out, std::string& err) { return zcompress(src, n, level, out, err); }

bool gradientTileHasChunk(int imgW, int imgH, int sx, int sy, const u8* bitmap, size_t rgbBytes) {
    int e[4];
    gradientExtent(imgW, imgH, sx, sy, bitmap, e);
    return e[2] > e[0] && e[3] > e[1] && rgbBytes > 0;                                   // (:4239)
}

bool emitGradientTile(FILE* f, int imgW, int imgH, int sx, int sy, const u8* bitmap, size_t rgbBytes, u32 palSize,
                      const std::vector<u8>& zBitmap, const std::vector<u8>& zRgb, int colorCompression, int planeBit, std::string& err) {
    int e[4];
    gradientExtent(imgW, imgH, sx, sy, bitmap, e);
    HeaderGradientTile h; memset(&h, 0, sizeof h);
    h.bbox.x = (s16)e[0]; h.bbox.y = (s16)e[1]; h.bbox.w = (s16)(e[2] - e[0]);
    h.bbox.h = (s16)(e[3] - e[0]);                                                       // maxY - minX, as the reference writes it (:4258)
    h.format = (u8)(sx | (sy << 3)); h.plane = (u8)planeBit;
    h.streamBitmapSize = (u32)zBitmap.size(); h.streamRGBSizeZStd = (u32)zRgb.size();
    h.streamRGBSizeCustomCompressor = palSize; h.streamRGBSizeUncompressed = (u32)rgbBytes;
    h.colorCompression = (u8)colorCompression;
    if (!putChunk(f, TAG_GRADTILE, &h, sizeof h, { &zBitmap, &zRgb })) { err = "fwrite"; return false; }
    return true;
}

int writeGradientTile(FILE* f, int imgW, int imgH, int sx, int sy, const u8* bitmap, size_t bitmapBytes, u8* rgb, size_t rgbBytes,
                      int colorCompression, int planeBit, std::string& err) {
    if (!gradientTileHasChunk(imgW, imgH, sx, sy, bitmap, rgbBytes)) return 0;
    std::vector<u8> zBitmap, zRgb, pal(rgbBytes * 3);
    if (!zcompress(bitmap, bitmapBytes, 18, zBitmap, err)) return -1;                    // CompressStream level 18 (:3697)
    u32 palSize = (u32)pal.size();
    if (!PaletteCompressor(rgb, (int)rgbBytes, pal.data(), &palSize)) { err = "PaletteCompressor overflow"; return -1; }
    pal.resize(palSize);
    if (!zcompress(pal.data(), pal.size(), 18, zRgb, err)) return -1;
    return emitGradientTile(f, imgW, imgH, sx, sy, bitmap, rgbBytes, palSize, zBitmap, zRgb, colorCompression, planeBit, err) ? 1 : -1;
}

bool writePlaneTile(FILE* f, const BoundingBox& constraint, const u16* defs, size_t nDefs, const u8* idx, size_t idxBytes,
                    int planeType, bool halfX, bool halfY, std::string& err) {
    std::vector<u8> zDefs, zIdx;
    if (!zcompress(defs, nDefs * sizeof(u16), 21, zDefs, err) || !zcompress(idx, idxBytes, 21, zIdx, err)) return false;   // (:4519, :4533)
    PlaneTile h; memset(&h, 0, sizeof h);
    h.bbox = constraint; h.version = 1;
    h.streamSizeTileMap = (u32)zDefs.size(); h.streamSizeTileStream = (u32)zIdx.size(); h.expectedSizeTileStream = (u32)idxBytes;
    h.format = (u8)((planeType << 2) | (halfX ? 1 : 0) | (halfY ? 2 : 0));
    if (!putChunk(f, TAG_PLANE, &h, sizeof h, { &zDefs, &zIdx })) { err = "fwrite"; return false; }
    return true;
}

bool emitTile1D(FILE* f, size_t pixBytes, size_t typeBytes, const std::vector<u8>& zPix, const std::vector<u8>& zType, int compressionColor,
                int compressionRange, std::string& err) {
    Header1D h; memset(&h, 0, sizeof h);
    h.version = 0; h.compressionColor = (u8)compressionColor; h.compressionRange = (u8)compressionRange;
    h.streamPixelBit = (u32)zPix.size(); h.streamPixelUncmp = (u32)pixBytes;
    h.streamTypeCnt = (u32)zType.size(); h.streamTypeUncmp = (u32)typeBytes;
    if (!putChunk(f, TAG_TILE1D, &h, sizeof h, { &zType, &zPix })) { err = "fwrite"; return false; }      // type first (:8566-8568)
    return true;
}

bool writeTile1D(FILE* f, const u8* pix, size_t pixBytes, const u8* type, size_t typeBytes, int compressionColor, int compressionRange,
                 std::string& err) {
    if (pixBytes == 0) return true;                                                      // no chunk for an empty stream (:8525)
    std::vector<u8> zPix, zType;
    if (!zcompress(pix, pixBytes, 18, zPix, err) || !zcompress(type, typeBytes, 18, zType, err)) return false;
    return emitTile1D(f, pixBytes, typeBytes, zPix, zType, compressionColor, compressionRange, err);
}

bool writeTile3D(FILE* f, const Tile3DStreams& st, int colorCompression, int component, std::string& err) {
    HeaderTile3D h; memset(&h, 0, sizeof h);
    for (int k = 0; k < 6; k++) if (st.mapBytes[k] > 65535) { err = "'3DTL' keeps tile-map sizes in 16 bits (HeaderTile3D, YAIK_private.h:316-328): image too large for this chunk"; return false; }
    std::vector<u8> zMap[6], zType, zColor, zIdx[4];
    for (int k = 0; k < 6; k++) if (!zcompress(st.map[k], st.mapBytes[k], 18, zMap[k], err)) return false;
    h.component = (u8)component;
    // header fields in the reference's naming; the MAP ORDER in the payload is 16x8, 8x16, 8x8, 8x4, 4x8, 4x4 (:7651-7662)
    h.sizeT16_8Map = (u16)st.mapBytes[0]; h.sizeT16_8MapCmp = (u16)zMap[0].size();
    h.sizeT8_16Map = (u16)st.mapBytes[1]; h.sizeT8_16MapCmp = (u16)zMap[1].size();
    h.sizeT8_8Map = (u16)st.mapBytes[2]; h.sizeT8_8MapCmp = (u16)zMap[2].size();
    h.sizeT8_4Map = (u16)st.mapBytes[3]; h.sizeT8_4MapCmp = (u16)zMap[3].size();
    h.sizeT4_8Map = (u16)st.mapBytes[4]; h.sizeT4_8MapCmp = (u16)zMap[4].size();
    h.sizeT4_4Map = (u16)st.mapBytes[5]; h.sizeT4_4MapCmp = (u16)zMap[5].size();
    h.streamTypeCnt = (u32)st.nTiles; h.streamColorCnt = (u32)st.nTiles * 6;
    if (st.nTiles) {
        if (!zcompress(st.tileType, st.nTiles * 2, 18, zType, err)) return false;
        std::vector<u8> col(st.color, st.color + st.nTiles * 6);
        for (auto& v : col) v = (u8)((v * colorCompression + 127) / 255);                // CompressF (:7494-7497)
        if (!zcompress(col.data(), col.size(), 18, zColor, err)) return false;
        h.compressionRateColor = (u8)colorCompression;
    }
    h.comprTypeSize = (u32)zType.size(); h.comprColorSize = (u32)zColor.size();
    u32* cnt[4] = { &h.stream3BitCnt, &h.stream4BitCnt, &h.stream5BitCnt, &h.stream6BitCnt };
    u32* cmp[4] = { &h.compr3BitSize, &h.compr4BitSize, &h.compr5BitSize, &h.compr6BitSize };
    for (int b = 0; b < 4; b++) {
        *cnt[b] = (u32)st.nIdx[b];
        if (st.nIdx[b]) {
            std::vector<u8> x3(st.idx[b], st.idx[b] + st.nIdx[b]);
            for (auto& v : x3) v = (u8)(v * 3);                                           // index -> interleaved entry offset (:7526-7529)
            if (!zcompress(x3.data(), x3.size(), 18, zIdx[b], err)) return false;
        }
        *cmp[b] = (u32)zIdx[b].size();
    }
    if (!putChunk(f, TAG_TILE3D, &h, sizeof h, { &zIdx[0], &zIdx[1], &zIdx[2], &zIdx[3], &zType, &zColor, &zMap[0], &zMap[1], &zMap[2], &zMap[3], &zMap[4], &zMap[5] })) { err = "fwrite"; return false; }
    return true;
}

}  // namespace yaikchunk
