// Drop-in mirror of the hot-path part of the reference's `EncoderContext` (encoder/EncoderContext.h:183-370 in KLab/YAIK):
// same method names, argument meaning and call order for MipPrefilter -> 7x FittingQuadSmooth -> DynamicTileEncode x3
// (-> DynamicTileCompressor x3), executed on an MI355X through the C-ABI of include/yaik_hip.h.
//
// Differences a caller sees:
//   * with `outFile` set the passes append the same chunks as the reference ('MIPM', 'GTIL', 'PLNT', '1DTL'; PaletteCompressor
//     and ZStd run on host cores, see chunks.h); with `outFile == NULL` they write nothing.  Either way the raw streams each
//     pass hands to the entropy stage are exposed through the Last*() accessors (bitmap, corner stream, tile defs, nibbles);
//   * the seven FittingQuadSmooth calls must come in the shipped order (4,4)(4,3)(3,4)(3,3)(3,2)(2,3)(2,2) with the image's
//     planes 0,1,2 (EncoderContext.cpp:9057-9093): the first one launches the fused kernel, the others return its cached results;
//   * errors follow the reference's convention (message on stdout, neutral return value) plus LastError().
// The methods are public here: this class is the surface of the path, not of Convert().
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#include "framework.h"

struct yk_ctx;

struct EncoderContext {
public:
    EncoderContext();
    ~EncoderContext();

    // same tunables as the reference constructor defaults (EncoderContext.h:185-228)
    int colorCompressionQuad, colorCompressionLUT3D, colorCompression1D, rangeCompression1D;
    // mipmap-mask bounding box in pixels + bookkeeping the range quantiser reads (EncoderContext.h:246-253)
    int mipMapTileSize, boundX0, boundY0, boundX1, boundY1, remainingPixels;
    bool dumpImage, evaluateLUT, evaluateLUT2D;
    FILE* outFile;                                      // chunks are appended here when set (the reference requires it; NULL = streams only)
    int fileOutSize;

    bool SetImageToEncode(Image* newImage);             // takes ownership, deletes the previous image (EncoderContext.cpp:1227-1233)
    void Release();

    void CheckMipmapMask();                             // EncoderContext.cpp:2784
    void PrepareQuadSmooth();                           // empty in the reference too (:2796)
    void MipPrefilter(bool active);                     // :1257
    int  FittingQuadSmooth(int rejectFactor, Plane* a, Plane* b, Plane* c, Image* testOutput, bool useYCoCg,
                           int tileBitSizeX, int tileBitSizeY);                           // :3710, returns TileDone
    int  DynamicTileEncode(bool mode3BitOnly, Plane* plane, Plane* dst, bool isCo, bool isCg, bool isHalfX, bool isHalfY);   // :4365
    u8*  DynamicTileCompressor(u8* stream, Plane* src, Plane* map, Plane* debug);         // :8398, returns the advanced cursor
    // (f)4 3-D LUT tiles, where Convert() runs them (:9117-9218): after the gradient passes, before the 1-D compressor
    void Load3DPattern(const char* fileName);                                             // :7851, one 'Bank3D' .lut file (u8 count, r[], g[], b[])
    bool Save3DLutFile(const char* fileName);                                             // the decoder's 'LUL0' file RegisterAndCreate3DLut writes (:7820-7847)
    void StartCorrelationSearch(bool is3D);                                               // :7316
    void Correlation3DSearch(Image* input, Image* output, int tileShiftX, int tileShiftY);// :6245
    void EndCorrelationSearch(bool is3D, u8 component);                                   // :7366, appends the '3DTL' chunk to outFile
    int  correlationPatternCount3D;
    int  LastCorrelationMatches() const { return lutMatched; }                            // "MATCHED TILE" of the last Correlation3DSearch
    void GenerateDynamicTileChunk(u8* stream, int sizeStream);                            // :8524, '1DTL' chunk of the three planes' streams
    // The chunk sequence of Convert() restricted to this path (:9007-9016, :9057-9093, :9451-9470, :9779-9781): file header,
    // ['MIPM' for RGBA], 7x 'GTIL', '1DTL', terminator — a stream the reference's YAIK_DecodeImage accepts.  Takes no ownership of f.
    bool ConvertHotPath(FILE* f);
    // The same file, byte for byte, with the entropy stage threaded: PaletteCompressor keeps its pass order on one
    // thread (its code book carries over from chunk to chunk), the sixteen ZStd streams (7 bitmaps, 7 colour streams, the 1-D pixel and
    // type streams) are compressed by `threads` workers, the chunks are written in file order.  ConvertHotPathBegin returns as soon as
    // every raw stream is on the host: the GPU and this object are then free for SetImageToEncode + the passes of the NEXT image while
    // the previous files are still being compressed (each stage has its own thread and its own PaletteCompressor code book, started
    // empty like a fresh process; up to kMaxStages images in flight, Begin waits for the oldest beyond that); ConvertHotPathFinish
    // waits for all of them.  `f` belongs to its stage between Begin and Finish.
    bool ConvertHotPathBegin(FILE* f, int threads);
    bool ConvertHotPathFinish();
    bool ConvertHotPathParallel(FILE* f, int threads) { return ConvertHotPathBegin(f, threads) && ConvertHotPathFinish(); }
    // The tile maps of the image, encoded as row stripes on several GPUs of the node by THIS process (the reference is one
    // process, include/YAIK.h:42-47): stripe i = a band of 64-row blocks + one halo row (the BL / BR corner samples at y + T,
    // EncoderContext.cpp:3853-3856) on devices[i]; the image-wide kept-tile box of MipPrefilter is combined on the host (no collective);
    // the per-stripe maps are concatenated on the first device by ONE grouped RCCL transfer (yk_gather_maps_all) and come back in one copy.
    // Stripes that share a device (a one-GPU box) are simply read back one by one.  The result equals the whole-image encode bit for bit:
    // 7 swizzled bitmaps, and per plane the tile definitions and the nibble stream of DynamicTileEncode.
    struct StripeTileMaps {
        std::vector<u8> bitmap[7];
        std::vector<u16> defs[3];
        std::vector<u8> nibbles[3];
        size_t nNibbles[3] = { 0, 0, 0 };
        int bounds[4] = { 0, 0, 0, 0 };                 // boundX0, boundY0, boundX1, boundY1 of the whole image
        int stripes = 0, gatherRanks = 0;               // ranks of the RCCL communicator the gather used (0 = no transfer between devices)
    };
    bool ConvertHotPathStripes(const int* devices, int nStripes, bool mode3BitOnly, StripeTileMaps* out);

    // raw streams of the last call of each kind (what the reference compresses and writes, before entropy coding)
    const std::vector<u8>&  LastGradientBitmap() const { return gradBitmap; }             // pFillBitMap (:3775)
    const std::vector<u8>&  LastGradientRGBStream() const { return gradRgb; }             // rgbStream (:3783)
    const std::vector<u16>& LastTileDefs() const { return tileDefs; }                     // streamTileDef (:4419)
    const std::vector<u8>&  LastTileIndexStream() const { return tileIdx; }               // streamTileIdx (:4421), closed to a byte
    size_t                  LastTileIndexCount() const { return nNibbles; }
    const std::vector<u8>&  MipmapBitmap() const { return mipBitmap; }                    // 'MIPM' payload (:1317-1327)
    bool                    MipmapHasChunk() const { return mipHasChunk; }
    const std::vector<u8>&  TileTypeStream1D() const { return type1d; }                   // streamType (:8217)
    const char*             LastError() const { return err.c_str(); }
    int                     device;                                                        // HIP device ordinal, set before SetImageToEncode

private:
    struct EntropyStage;                                // the streams of one image + the thread that compresses and writes them
    enum { kMaxStages = 8 };
    std::vector<EntropyStage*> stages;
    bool stagesOk = true; std::string stagesErr;
    void retireOldestStage();
    bool ensureEncoded(int rejectFactor, bool mode3, bool wantDst);
    bool fail(const char* what);
    Image* original;
    yk_ctx* ctx;
    bool bound, alphaDone, encoded, enc3, encDst, oneDReady;
    int encReject, nextPass;
    std::vector<u8> gradBitmap, gradRgb, tileIdx, mipBitmap, pix1d, type1d;
    std::vector<u16> tileDefs;
    size_t nNibbles, cursor1d;
    bool mipHasChunk;
    int lutMatched;
    std::vector<std::vector<u8>> lutPatterns;          // as loaded (count x 3, interleaved): the factor tables of the LUT file come from them
    std::string err;
};
