// Wall-clock of EncoderContext::ConvertHotPath (sequential entropy stage) against ConvertHotPathBegin/Finish (entropy stages of several
// images in flight on host threads while the next images go through the GPU passes) on synthetic frames.   usage: convert_bench [W=4096] [images=3] [threads=8]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "EncoderContext.h"

static Image* synth(int W, unsigned seed) {
    Image* img = Image::CreateImage(W, W, 4, false);
    unsigned s = seed;
    for (int y = 0; y < W; y++)
        for (int x = 0; x < W; x++) {
            s = s * 1664525u + 1013904223u;
            const int k = ((x >> 6) + (y >> 6)) & 3;
            const int base[3] = { 255 * x / W, 255 * y / W, 255 * (x + y) / (2 * W) };
            for (int c = 0; c < 3; c++) {
                int v = base[c];
                if (k == 2) v = (v + ((s >> (8 + 4 * c)) & 7)) & 255;
                if (k == 3) v = (s >> (8 + 8 * c)) & 255;
                img->GetPlane(c)->GetPixels()[(size_t)y * W + x] = v;
            }
            const bool frame = x < W / 8 || x >= W - W / 16 || y < W / 16 || y >= W - W / 8, hole = (((x >> 7) + (y >> 7)) % 5) == 0;
            img->GetPlane(3)->GetPixels()[(size_t)y * W + x] = (frame || hole) ? 0 : 255;
        }
    return img;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 8, T = argc > 3 ? atoi(argv[3]) : 2;
    EncoderContext ctx;
    std::vector<long> sizes[2];
    double secs[2] = { 0, 0 }, gpuSide = 0;
    for (int mode = 0; mode < 2; mode++) {
        std::vector<FILE*> files;
        // images are generated outside the timed region
        std::vector<Image*> imgs; for (int i = 0; i < N; i++) imgs.push_back(synth(W, 1000u + i));
        const double t0 = now();
        for (int i = 0; i < N; i++) {
            FILE* f = tmpfile(); if (!f) return 2;
            files.push_back(f);
            const double g0 = now();
            if (!ctx.SetImageToEncode(imgs[i])) { fprintf(stderr, "%s\n", ctx.LastError()); return 3; }
            const bool ok = mode == 0 ? ctx.ConvertHotPath(f) : ctx.ConvertHotPathBegin(f, T);
            if (mode == 1) gpuSide += now() - g0;
            if (!ok) { fprintf(stderr, "%s\n", ctx.LastError()); return 4; }
        }
        if (!ctx.ConvertHotPathFinish()) { fprintf(stderr, "%s\n", ctx.LastError()); return 4; }
        secs[mode] = now() - t0;
        for (FILE* f : files) { fflush(f); sizes[mode].push_back(ftell(f)); fclose(f); }
        ctx.SetImageToEncode(nullptr);
    }
    bool same = sizes[0] == sizes[1];
    printf("%d x %dx%d RGBA -> .yaik: sequential entropy stage %.2f s (%.2f s per image), entropy stages of up to 8 images in flight (%d ZStd workers each) while the next images' "
           "upload and GPU passes run %.2f s (%.2f s per image, of which upload + GPU + download %.3f s); file sizes %s (%ld bytes first image)\n",
           N, W, W, secs[0], secs[0] / N, T, secs[1], secs[1] / N, gpuSide / N, same ? "identical" : "DIFFER", sizes[0].empty() ? 0L : sizes[0][0]);
    return same ? 0 : 1;
}
