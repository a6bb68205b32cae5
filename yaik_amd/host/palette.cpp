#include "palette.h"
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace {
struct Code { int ref, r, g, b; };
// One table per THREAD: within a thread it behaves like the reference's process-global table (rows carry over from call to call);
// entropy stages of different images, each on its own thread and starting from PaletteResetCodeBook like a fresh process, do not
// see each other (EncoderContext::ConvertHotPathBegin).
thread_local std::vector<Code> gCodes(100000, Code{0, 0, 0, 0});     // the reference's table has 100000 rows (:3216) and overflows beyond; this one grows
thread_local int gCodeCount = 0;
thread_local std::unordered_map<u32, int> gRowOf;       // delta -> row, this call only (the reference scans linearly: same row found)

inline u32 deltaKey(int dr, int dg, int db) { return (u32)(dr + 256) | ((u32)(dg + 256) << 10) | ((u32)(db + 256) << 20); }
void registerCode(int dr, int dg, int db) {
    const u32 key = deltaKey(dr, dg, db);
    auto it = gRowOf.find(key);
    if (it != gRowOf.end()) { gCodes[it->second].ref++; return; }
    if (gCodeCount == (int)gCodes.size()) gCodes.resize(gCodes.size() * 2, Code{0, 0, 0, 0});
    gRowOf.emplace(key, gCodeCount);
    gCodes[gCodeCount++] = Code{0, dr, dg, db};
}
int findCode(int dr, int dg, int db) {                  // rows 0..63 only, stale or not
    for (int n = 0; n < 64; n++) if (gCodes[n].r == dr && gCodes[n].g == dg && gCodes[n].b == db) return n;
    return -1;
}
}

void PaletteResetCodeBook() { std::fill(gCodes.begin(), gCodes.end(), Code{0, 0, 0, 0}); gCodeCount = 0; }

bool PaletteCompressor(u8* input, int size, u8* output, u32* maxSizeInOut) {
    const int entries = size / 3;
    const u32 cap = *maxSizeInOut;
    u32 si = 0; bool error = false;
    auto put = [&](int v) { if (si < cap) output[si++] = (u8)v; else error = true; };

    // pass 1: every colour votes for the delta to its nearest predecessor among the previous 64 (first minimum wins)
    gCodeCount = 0; gRowOf.clear();
    registerCode(0, 0, 0);
    for (int n = 1; n < entries; n++) {
        const u8* pix = input + n * 3;
        int best = 999999999, bR = 0, bG = 0, bB = 0;
        for (int prev = std::max(0, n - 64); prev < n; prev++) {
            const int dR = pix[0] - input[prev * 3], dG = pix[1] - input[prev * 3 + 1], dB = pix[2] - input[prev * 3 + 2];
            const int d = dR * dR + dG * dG + dB * dB;
            if (d < best) { best = d; bR = dR; bG = dG; bB = dB; }
        }
        registerCode(bR, bG, bB);
    }
    // rows 1.. ordered by votes, ties in registration order (the reference's qsort is glibc's stable merge sort here); row 0 stays
    if (gCodeCount > 2) std::stable_sort(gCodes.begin() + 1, gCodes.begin() + gCodeCount, [](const Code& a, const Code& b) { return a.ref > b.ref; });

    const int finalCount = std::min(gCodeCount, 128);
    put(finalCount);
    for (int n = 0; n < finalCount; n++) { put(gCodes[n].r); put(gCodes[n].g); put(gCodes[n].b); }
    if (entries > 0) { put(input[0]); put(input[1]); put(input[2]); }

    // pass 2: per colour, in priority order: code of the delta to the previous colour; else the lowest-numbered code reaching
    // it from one of the 64 colours before that (back-reference byte + code); else explicit deltas / absolute bytes
    for (int n = 1; n < entries && !error; n++) {
        const u8* pix = input + n * 3;
        bool done = false; int bestIdx = 999, bestDist = 0;
        for (int prev = n - 1; prev >= std::max(0, n - 65); prev--) {
            const int idx = findCode(pix[0] - input[prev * 3], pix[1] - input[prev * 3 + 1], pix[2] - input[prev * 3 + 2]);
            if (idx < 0) continue;
            if (prev == n - 1) { put(idx & 0x7F); done = true; break; }
            const int distance = (n - prev) - 2;
            if (distance < 64 && idx < bestIdx) { bestIdx = idx; bestDist = distance; done = true; }
        }
        if (bestIdx != 999) { put(0xC0 | (bestDist & 0x3F)); put(bestIdx & 0x7F); }
        if (!done) {
            const int dR = pix[0] - pix[-3], dG = pix[1] - pix[-2], dB = pix[2] - pix[-1];
            const int mask = (dR ? 1 : 0) | (dG ? 2 : 0) | (dB ? 4 : 0);
            const bool fits = dR >= -128 && dR <= 127 && dG >= -128 && dG <= 127 && dB >= -128 && dB <= 127;
            put((fits ? 0x80 : 0x88) | mask);
            if (dR) put(fits ? dR : pix[0]);
            if (dG) put(fits ? dG : pix[1]);
            if (dB) put(fits ? dB : pix[2]);
        }
    }
    *maxSizeInOut = error ? 0 : si;
    return !error;
}

void PaletteFullRangeRemapping(u8* data, int size, u8 originalRange) {
    const int inv = originalRange ? ((255 << 16) / originalRange) : (255 << 16);
    for (int i = 0; i < size; i++) data[i] = (u8)((data[i] * inv) >> 16);
}

bool PaletteDecompressor(u8* input, int inputSize, int inputBufferSize, u8* output, int outputSize, u8 colorCompression) {
    if (inputSize < 4 || outputSize < 3) return false;
    const u8* in = input;
    const int bookSize = *in++;
    if (1 + bookSize * 3 + 3 > inputSize) return false;
    const u8* book = in; in += bookSize * 3;
    const u8* inEnd = input + inputBufferSize;          // the caller over-allocates by 128*3 bytes (decoder/YAIK_API.cpp:893)
    u8* wr = output; u8* const lastRGB = output + outputSize - 3;
    *wr++ = *in++; *wr++ = *in++; *wr++ = *in++;
    const u8* last = output;
    while (wr <= lastRGB) {
        if (in + 4 > inEnd) return false;
        const int c = *in++;
        if (c & 0x80) {
            if (c & 0x40) {                             // back-reference: the next code applies to an older colour
                last = wr - ((c & 0x3F) + 2) * 3;
                if (last < output) return false;
            } else {
                const int kind = (c >> 3) & 7;
                if (kind == 0) {                        // signed deltas of the flagged channels
                    wr[0] = (u8)(last[0] + ((c & 1) ? *in++ : 0)); wr[1] = (u8)(last[1] + ((c & 2) ? *in++ : 0)); wr[2] = (u8)(last[2] + ((c & 4) ? *in++ : 0));
                } else if (kind == 1) {                 // absolute bytes of the flagged channels
                    wr[0] = (c & 1) ? *in++ : last[0]; wr[1] = (c & 2) ? *in++ : last[1]; wr[2] = (c & 4) ? *in++ : last[2];
                } else return false;
                last = wr; wr += 3;
            }
        } else {
            const u8* code = book + (c & 0x7F) * 3;
            wr[0] = (u8)(last[0] + code[0]); wr[1] = (u8)(last[1] + code[1]); wr[2] = (u8)(last[2] + code[2]);
            last = wr; wr += 3;
        }
    }
    PaletteFullRangeRemapping(output, outputSize, colorCompression);
    return true;
}
