// Host-side mirror of the reference's operator surface for the tile hot path (encoder/framework.h:74-225 in KLab/YAIK):
// `Plane` (int32 2-D array), `Image` (3-4 planes), `BoundingBox`.  Same class and method names, argument meaning and
// ownership rules, so code written against the reference's framework.h compiles against this header for the path;
// everything outside the path (resampling, PNG I/O, YCoCg, histograms) is intentionally absent.
// Written from the interface description; no reference code is reproduced.
#pragma once
#include <cstdint>
#include <cstring>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef int16_t s16;
typedef int32_t s32;

struct BoundingBox { s16 x, y, w, h; };                 // include/YAIK_private.h:15-20

class Plane {                                           // encoder/framework.h:74-127
public:
    Plane(int w_, int h_) : w(w_), h(h_) { pixels = new int[(size_t)w_ * h_]; }
    ~Plane() { delete[] pixels; }
    inline int  GetWidth() { return w; }
    inline int  GetHeight() { return h; }
    inline int* GetPixels() { return pixels; }
    inline int  GetIndex(int x, int y) { return x + y * w; }
    inline BoundingBox GetRect() { BoundingBox r; r.x = 0; r.y = 0; r.w = (s16)w; r.h = (s16)h; return r; }
    void SetPixel(int x, int y, int v) { pixels[x + y * w] = v; }
    void Fill(BoundingBox& r, int v) {
        for (int y = r.y; y < r.y + r.h; y++) for (int x = r.x; x < r.x + r.w; x++) pixels[x + y * w] = v;
    }
    void Clear() { memset(pixels, 0, sizeof(int) * (size_t)w * h); }
    Plane* Clone() { Plane* p = new Plane(w, h); memcpy(p->pixels, pixels, sizeof(int) * (size_t)w * h); return p; }
    // clamp-to-edge read that also reports whether the coordinate was outside (framework.h:116-121)
    int GetPixelValue(int x, int y, bool& isOutside) {
        isOutside = (x < 0) || (x >= w) || (y < 0) || (y >= h);
        const int cx = x < 0 ? 0 : (x >= w ? w - 1 : x), cy = y < 0 ? 0 : (y >= h ? h - 1 : y);
        return pixels[cx + cy * w];
    }
private:
    int* pixels;
    int w, h;
};
typedef Plane* TPlane;

class Image {                                           // encoder/framework.h:137-225
public:
    ~Image() { for (int i = 0; i < 4; i++) planes[i] = nullptr; }      // planes are NOT deleted, like the reference (:151-157)
    static Image* CreateImage(int w, int h, int channelCount, bool fill) {
        Image* r = new Image(); r->w = w; r->h = h; r->planeCount = channelCount;
        for (int c = 0; c < channelCount && c < 4; c++) { r->planes[c] = new Plane(w, h); if (fill) r->planes[c]->Clear(); }
        return r;
    }
    inline int GetWidth() { return w; }
    inline int GetHeight() { return h; }
    inline TPlane GetPlane(int i) { return planes[i]; }
    bool HasAlpha() { return planeCount == 4; }
    void Clear() { for (int i = 0; i < 4; i++) if (planes[i]) planes[i]->Clear(); }
    bool ReplacePlane(int index, TPlane np) {           // false if the size differs from the plane being replaced (:181-196)
        if (!np) return false;
        TPlane old = planes[index];
        if (old && (np->GetWidth() != old->GetWidth() || np->GetHeight() != old->GetHeight())) return false;
        planes[index] = np; return true;
    }
    void GetPixel(int x, int y, int* rgb, bool& isOutside) {
        isOutside = (x < 0) || (x >= w) || (y < 0) || (y >= h);
        const int cx = x < 0 ? 0 : (x >= w ? w - 1 : x), cy = y < 0 ? 0 : (y >= h ? h - 1 : y);
        for (int c = 0; c < 3; c++) rgb[c] = planes[c]->GetPixels()[cx + cy * w];
    }
    void SetPixel(int x, int y, int r, int g, int b) {
        planes[0]->SetPixel(x, y, r); planes[1]->SetPixel(x, y, g); planes[2]->SetPixel(x, y, b);
    }
private:
    Image() : planeCount(0), w(0), h(0) { for (int i = 0; i < 4; i++) planes[i] = nullptr; }
    Plane* planes[4];
    int planeCount, w, h;
};
