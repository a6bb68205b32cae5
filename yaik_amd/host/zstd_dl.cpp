#include "zstd_dl.h"
#include <dlfcn.h>
#include <string>

namespace {
typedef size_t (*fn_compress)(void*, size_t, const void*, size_t, int);
typedef size_t (*fn_decompress)(void*, size_t, const void*, size_t);
typedef size_t (*fn_bound)(size_t);
typedef unsigned (*fn_iserr)(size_t);
struct Lib {
    void* h = nullptr; fn_compress c = nullptr; fn_decompress d = nullptr; fn_bound b = nullptr; fn_iserr e = nullptr;
    std::string err;
    Lib() {
        const char* names[] = { "libzstd.so.1", "libzstd.so" };
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) { err = "libzstd.so.1 not found"; return; }
        c = (fn_compress)dlsym(h, "ZSTD_compress"); d = (fn_decompress)dlsym(h, "ZSTD_decompress");
        b = (fn_bound)dlsym(h, "ZSTD_compressBound"); e = (fn_iserr)dlsym(h, "ZSTD_isError");
        if (!c || !d || !b || !e) { err = "libzstd lacks the simple API"; h = nullptr; }
    }
};
Lib& lib() { static Lib l; return l; }
}

namespace yaikzstd {
bool available() { return lib().h != nullptr; }
const char* lastError() { return lib().err.c_str(); }
size_t compressBound(size_t n) { return available() ? lib().b(n) : 0; }
size_t compress(void* dst, size_t cap, const void* src, size_t n, int level) {
    if (!available()) return 0;
    const size_t r = lib().c(dst, cap, src, n, level);
    return lib().e(r) ? 0 : r;
}
bool decompressAny(void* dst, size_t cap, const void* src, size_t n, size_t* outSize) {
    if (!available()) return false;
    const size_t r = lib().d(dst, cap, src, n);
    if (lib().e(r)) return false;
    *outSize = r;
    return true;
}
bool decompress(void* dst, size_t expected, const void* src, size_t n) {
    if (!available()) return false;
    const size_t r = lib().d(dst, expected, src, n);
    return !lib().e(r) && r == expected;
}
}
