// CPU-only companion of the host library (no HIP): exercises the entropy stage and the chunk framing on named blob files, so
// that they can be pinned against the reference's own chunk stream without a GPU.
//   entropy_tool parse <file.yaik|chunks.bin> <w> <h> <out.blobs>     split a chunk stream into headers + decompressed payloads
//   entropy_tool write <streams.blobs> <out.yaik>                       frame raw pass outputs in the order the passes run
// Blob file = repeated { u32 nameLen, name, u64 dataLen, data } (same container as host_driver.cpp).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>
#include "chunks.h"
#include "palette.h"
#include "yaik_format.h"
#include "zstd_dl.h"

using namespace yaikfmt;
typedef std::vector<u8> Bytes;

static FILE* gOut;
static void blob(const std::string& name, const void* data, size_t len) {
    u32 nl = (u32)name.size(); unsigned long long dl = len;
    fwrite(&nl, 4, 1, gOut); fwrite(name.data(), 1, nl, gOut); fwrite(&dl, 8, 1, gOut); if (len) fwrite(data, 1, len, gOut);
}
static bool readAll(const char* path, Bytes& out) {
    FILE* f = fopen(path, "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f); return ok;
}
static std::map<std::string, Bytes> readBlobs(const char* path) {
    std::map<std::string, Bytes> m; Bytes all;
    if (!readAll(path, all)) return m;
    size_t p = 0;
    while (p + 4 <= all.size()) {
        u32 nl; memcpy(&nl, &all[p], 4); p += 4;
        std::string name((const char*)&all[p], nl); p += nl;
        unsigned long long dl; memcpy(&dl, &all[p], 8); p += 8;
        m[name] = Bytes(all.begin() + p, all.begin() + p + dl); p += dl;
    }
    return m;
}
static std::string nm(const char* b, int i, const char* suffix) { char t[96]; snprintf(t, sizeof t, "%s%d_%s", b, i, suffix); return t; }

static int parse(const char* path, int w, int h, const char* outPath) {
    Bytes f; if (!readAll(path, f)) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    gOut = fopen(outPath, "wb"); if (!gOut) return 2;
    size_t p = 0;
    if (f.size() >= sizeof(FileHeader)) {
        FileHeader fh; memcpy(&fh, f.data(), sizeof fh);
        if (fh.tag == TAG_FILE) { int v[4] = { fh.version, fh.width, fh.height, fh.infoMask }; blob("file_header", v, sizeof v); w = fh.width; h = fh.height; p = sizeof fh; }
    }
    int i = 0, terminated = 0;
    while (p + 4 <= f.size()) {
        u32 tag; memcpy(&tag, &f[p], 4);
        if (tag == TAG_END) { terminated = 1; break; }
        HeaderBase hb; memcpy(&hb, &f[p], sizeof hb);
        const u8* body = &f[p + sizeof hb];
        blob(nm("c", i, "tag"), &hb.tag, 4);
        int len = (int)hb.length; blob(nm("c", i, "length_mod4"), &(len = len & 3), 4);
        if (hb.tag == TAG_MIPMAP) {
            MipmapHeader mh; memcpy(&mh, body, sizeof mh);
            int v[6] = { mh.bbox.x, mh.bbox.y, mh.bbox.w, mh.bbox.h, mh.version, mh.mipmapLevel };
            blob(nm("c", i, "hdr"), v, sizeof v);
            blob(nm("c", i, "bits"), body + sizeof mh, ((size_t)mh.bbox.w * mh.bbox.h + 7) / 8);
        } else if (hb.tag == TAG_GRADTILE) {
            HeaderGradientTile gh; memcpy(&gh, body, sizeof gh);
            int v[9] = { gh.bbox.x, gh.bbox.y, gh.bbox.w, gh.bbox.h, (int)gh.streamRGBSizeCustomCompressor, (int)gh.streamRGBSizeUncompressed,
                         gh.colorCompression, gh.format, gh.plane };                    // `version` is uninitialised in the reference: not compared
            blob(nm("c", i, "hdr"), v, sizeof v);
            u32 bx, by, bc; swizzleSize(gh.format & 7, (gh.format >> 3) & 7, bx, by, bc);
            Bytes bm((size_t)((w + bx - 1) / bx) * ((h + by - 1) / by) * bc / 8), pal(gh.streamRGBSizeCustomCompressor + 128 * 3), rgb(gh.streamRGBSizeUncompressed);
            const u8* z = body + sizeof gh;
            if (!yaikzstd::decompress(bm.data(), bm.size(), z, gh.streamBitmapSize)) { fprintf(stderr, "chunk %d: bitmap does not expand\n", i); return 3; }
            if (!yaikzstd::decompress(pal.data(), gh.streamRGBSizeCustomCompressor, z + gh.streamBitmapSize, gh.streamRGBSizeZStd)) { fprintf(stderr, "chunk %d: rgb does not expand\n", i); return 3; }
            if (!PaletteDecompressor(pal.data(), (int)gh.streamRGBSizeCustomCompressor, (int)pal.size(), rgb.data(), (int)rgb.size(), gh.colorCompression)) { fprintf(stderr, "chunk %d: palette\n", i); return 3; }
            blob(nm("c", i, "bitmap"), bm.data(), bm.size());
            blob(nm("c", i, "palette"), pal.data(), gh.streamRGBSizeCustomCompressor);
            blob(nm("c", i, "rgb"), rgb.data(), rgb.size());
        } else if (hb.tag == TAG_PLANE) {
            PlaneTile ph; memcpy(&ph, body, sizeof ph);
            int v[7] = { ph.bbox.x, ph.bbox.y, ph.bbox.w, ph.bbox.h, (int)ph.expectedSizeTileStream, ph.version, ph.format };
            blob(nm("c", i, "hdr"), v, sizeof v);
            // the tile-map size is not in the header: expand into a bound and keep what came out
            Bytes idx(ph.expectedSizeTileStream);
            const u8* z = body + sizeof ph;
            Bytes defs((size_t)(w / 8 + 1) * (h / 8 + 1) * 2 + 64);
            size_t got = 0;
            if (!yaikzstd::decompressAny(defs.data(), defs.size(), z, ph.streamSizeTileMap, &got)) { fprintf(stderr, "chunk %d: defs\n", i); return 3; }
            blob(nm("c", i, "defs"), defs.data(), got);
            if (!yaikzstd::decompress(idx.data(), idx.size(), z + ph.streamSizeTileMap, ph.streamSizeTileStream)) { fprintf(stderr, "chunk %d: idx\n", i); return 3; }
            blob(nm("c", i, "idx"), idx.data(), idx.size());
        } else if (hb.tag == TAG_TILE1D) {
            Header1D dh; memcpy(&dh, body, sizeof dh);
            int v[5] = { (int)dh.streamPixelUncmp, (int)dh.streamTypeUncmp, dh.compressionColor, dh.compressionRange, dh.version };
            blob(nm("c", i, "hdr"), v, sizeof v);
            Bytes ty(dh.streamTypeUncmp), px(dh.streamPixelUncmp);
            const u8* z = body + sizeof dh;
            if (!yaikzstd::decompress(ty.data(), ty.size(), z, dh.streamTypeCnt) || !yaikzstd::decompress(px.data(), px.size(), z + dh.streamTypeCnt, dh.streamPixelBit)) { fprintf(stderr, "chunk %d: 1d\n", i); return 3; }
            blob(nm("c", i, "type"), ty.data(), ty.size());
            blob(nm("c", i, "pix"), px.data(), px.size());
        } else { fprintf(stderr, "chunk %d: unknown tag %08x\n", i, hb.tag); return 3; }
        p += sizeof hb + hb.length; i++;
    }
    int tail[2] = { i, terminated };
    blob("chunk_count_terminated", tail, sizeof tail);
    fclose(gOut);
    return 0;
}

static int writeFile(const char* inPath, const char* outPath) {
    auto b = readBlobs(inPath);
    if (!b.count("meta")) { fprintf(stderr, "no meta blob\n"); return 2; }
    int meta[3]; memcpy(meta, b["meta"].data(), 12);
    const int w = meta[0], h = meta[1], np = meta[2];
    FILE* f = fopen(outPath, "wb"); if (!f) return 2;
    std::string err;
    bool ok = true;
    if (b.count("with_file_header")) ok = yaikchunk::writeFileHeader(f, w, h, np == 4);
    if (b.count("_mip_bitmap") && b.count("_mip_tile_bbox") && b["_mip_has_chunk"][0]) {
        s16 tb[4]; memcpy(tb, b["_mip_tile_bbox"].data(), 8);
        const int t[4] = { tb[0], tb[1], tb[2], tb[3] };
        ok = ok && yaikchunk::writeMipmap(f, t, 4, b["_mip_bitmap"].data(), b["_mip_bitmap"].size());
    }
    static const int passes[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    PaletteResetCodeBook();
    for (int i = 0; ok && i < 7; i++) {
        char k1[32], k2[32]; snprintf(k1, sizeof k1, "grad_bitmap_%d", i); snprintf(k2, sizeof k2, "grad_rgbraw_%d", i);
        Bytes& bm = b[k1]; Bytes& rgb = b[k2];
        ok = yaikchunk::writeGradientTile(f, w, h, passes[i][0], passes[i][1], bm.data(), bm.size(), rgb.data(), rgb.size(), 250, 7, err) >= 0;
    }
    if (ok && b.count("bounds_post")) {
        int bd[4]; memcpy(bd, b["bounds_post"].data(), 16);
        BoundingBox cb; cb.x = (s16)((bd[0] >> 3) << 3); cb.y = (s16)((bd[1] >> 3) << 3);
        cb.w = (s16)((((bd[2] + 7) >> 3) << 3) - cb.x); cb.h = (s16)((((bd[3] + 7) >> 3) << 3) - cb.y);
        for (int m = 0; ok && m < 2; m++) for (int p = 0; ok && p < 3; p++) {
            char k1[32], k2[32]; snprintf(k1, sizeof k1, "plnt_defs_%d_%d", m, p); snprintf(k2, sizeof k2, "plnt_idx_%d_%d", m, p);
            if (!b.count(k1)) continue;
            ok = yaikchunk::writePlaneTile(f, cb, (const u16*)b[k1].data(), b[k1].size() / 2, b[k2].data(), b[k2].size(), 0, false, false, err);
        }
    }
    if (ok && b.count("d1_pix")) ok = yaikchunk::writeTile1D(f, b["d1_pix"].data(), b["d1_pix"].size(), b["d1_type"].data(), b["d1_type"].size(), 255, 15, err);
    if (ok && b.count("with_file_header")) ok = yaikchunk::writeEndOfFile(f);
    fclose(f);
    if (!ok) { fprintf(stderr, "write failed: %s\n", err.c_str()); return 3; }
    return 0;
}

// entropy_tool palbench <nColours>: times PaletteCompressor on a synthetic corner stream (6-bit quantised random walk)
static int palbench(int n) {
    Bytes in((size_t)n * 3), out((size_t)n * 9 + 64);
    unsigned s = 12345; int c[3] = { 100, 120, 90 };
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
        s = s * 1664525u + 1013904223u;
        c[k] += (int)((s >> 24) % 9) - 4; if (c[k] < 0) c[k] = 0; if (c[k] > 250) c[k] = 250;
        in[(size_t)i * 3 + k] = (u8)(c[k] & ~3);
    }
    PaletteResetCodeBook();
    u32 sz = (u32)out.size();
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    const bool ok = PaletteCompressor(in.data(), (int)in.size(), out.data(), &sz);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    unsigned h = 2166136261u; for (u32 i = 0; i < sz; i++) h = (h ^ out[i]) * 16777619u;
    printf("PaletteCompressor: %d colours -> %u bytes in %.3f s (%.2f Mcolours/s) ok=%d fnv=%08x\n", n, sz, dt, n / dt / 1e6, (int)ok, h);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (!yaikzstd::available()) { fprintf(stderr, "%s\n", yaikzstd::lastError()); return 4; }
    if (argc == 6 && !strcmp(argv[1], "parse")) return parse(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5]);
    if (argc == 4 && !strcmp(argv[1], "write")) return writeFile(argv[2], argv[3]);
    if (argc == 3 && !strcmp(argv[1], "palbench")) return palbench(atoi(argv[2]));
    fprintf(stderr, "usage: entropy_tool parse <file> <w> <h> <out.blobs> | entropy_tool write <streams.blobs> <out.yaik>\n");
    return 2;
}
