/* yaik_hip.h — C-ABI of the MI355X (gfx950) implementation of the YAIK per-tile hot path.
 *
 * This is the drop-in boundary: plain C, opaque handle, caller-owned pointers and sizes, no C++ or
 * torch types.  Every entry point names the reference interface it replaces (paths relative to the
 * KLab/YAIK tree).  The reference has no FFI of its own — the passes are C++ member functions called
 * in-process — so the boundary sits exactly where `EncoderContext` touches pixels; the C++ mirror of
 * that class in yaik_amd/host/ (same method names and argument meaning) is a thin caller of these
 * functions, and INTEGRATION.md shows the binding a YAIK maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative yk_status on failure (first error text is kept
 *     in the handle: yk_last_error).  Nothing throws.  There is NO CPU fallback: without a usable
 *     HIP device yk_create fails.
 *   - one handle per host thread / GPU; handles are not thread-safe.
 *   - "device pointer" = HBM address valid on the handle's device (e.g. torch tensor data_ptr()).
 *   - planes are int32, row-major, values in [0,255]: the reference `Plane` layout
 *     (encoder/framework.h:74-127, idx = x + y*w).  Width and height must be multiples of 8
 *     (Image::LoadPNG enforces the same, encoder/Image.cpp:206).
 *   - all launches go to the handle's stream; getters that copy to host synchronise that stream.
 */
#ifndef YAIK_HIP_H
#define YAIK_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct yk_ctx yk_ctx;

enum yk_status {
    YK_OK = 0,
    YK_ERR_NO_DEVICE = -1,      /* no HIP device / runtime: the product path refuses to run */
    YK_ERR_BAD_ARG = -2,
    YK_ERR_HIP = -3,            /* a HIP call failed; see yk_last_error */
    YK_ERR_STATE = -4,          /* call order violated (e.g. encode before binding planes) */
    YK_ERR_RANGE = -5,          /* output buffer too small */
    YK_ERR_COMM = -6            /* RCCL missing or a RCCL call failed; see yk_last_error */
};

/* number of gradient passes and their (tileShiftX, tileShiftY) in the shipped order
 * 16x16,16x8,8x16,8x8,8x4,4x8,4x4 (encoder/EncoderContext.cpp:9057-9093) */
#define YK_NUM_PASSES 7

/* ---- lifetime ------------------------------------------------------------------------------ */
int         yk_create(int device, yk_ctx** out);
void        yk_destroy(yk_ctx* c);
const char* yk_last_error(const yk_ctx* c);
/* NULL = the handle's own non-blocking stream (the default).  That stream is NOT ordered against the null stream or any
 * other stream of the process: fence hand-overs of buffers on the host, or pass the producer's / consumer's stream here.
 * A stream handle is only meaningful inside the HIP runtime instance that created it (a process that loads this library
 * before a framework's bundled copy of the runtime ends up with two instances; load the framework first). */
/* The stream being replaced should still exist when this is called (the new stream is ordered behind what the old one holds); if it was
 * destroyed meanwhile the switch still happens, without that ordering. */
int         yk_set_stream(yk_ctx* c, void* hipStream);
int         yk_synchronize(yk_ctx* c);
/* Device-side ordering against another stream of the SAME runtime instance (0 = the null stream), instead of a host fence:
 * yk_stream_wait_for: everything queued on the handle's stream from now on waits for what `producerStream` holds now;
 * yk_stream_handoff:  everything queued on `consumerStream` from now on waits for what the handle's stream holds now. */
int         yk_stream_wait_for(yk_ctx* c, void* producerStream);
int         yk_stream_handoff(yk_ctx* c, void* consumerStream);
int         yk_device_count(void);                          /* no device initialisation side effects beyond hipGetDeviceCount */

/* ---- image binding:  EncoderContext::SetImageToEncode (encoder/EncoderContext.cpp:1227) ------
 * A handle works on one image or one row stripe of an image.  fullW/fullH are the whole image,
 * [y0, y0+h) the rows this handle owns (y0 and h multiples of 64 unless the stripe is the last one);
 * for a whole image pass y0 = 0, h = fullH.  The planes handed in must contain the owned rows plus
 * `haloRows` (0 or 1) further row: FittingQuadSmooth samples the BL/BR corners at y+T
 * (EncoderContext.cpp:3855-3856); the last stripe clamps like Plane::GetPixelValue instead. */
int yk_set_image(yk_ctx* c, int fullW, int fullH, int nPlanes, int y0, int h, int haloRows);
/* host planes -> HBM copy owned by the handle (strideElems = row pitch in int32 elements) */
int yk_upload_planes(yk_ctx* c, const int32_t* const hostPlanes[4], int strideElems);
/* zero-copy: planes already resident in HBM */
int yk_bind_device_planes(yk_ctx* c, const int32_t* const devPlanes[4], int strideElems);
/* The path is defined for samples in 0..255 held in int32 planes (the fused kernel keeps their low byte; the reference reads the whole int,
 * encoder/framework.h:116-121).  yk_upload_planes checks what it copied and fails with YK_ERR_BAD_ARG (no planes bound afterwards) when a sample
 * lies outside; callers that bind device memory and cannot vouch for its contents call yk_validate_planes: *nOutOfRange = samples outside 0..255
 * in the bound planes (all frames of a batch; one streaming pass, synchronises). */
int yk_validate_planes(yk_ctx* c, size_t* nOutOfRange);

/* ---- a9  alpha tile-reject:  EncoderContext::MipPrefilter (EncoderContext.cpp:1257-1427) -----
 * stage 1 (per stripe): per aligned 16x16 block "all 256 alphas == 0" + bounding box of kept blocks. */
int yk_alpha_reject(yk_ctx* c);
/* bbox of kept blocks of THIS stripe in full-image pixels {x0,y0,x1,y1}; {9999999,9999999,-1,-1} if none. Synchronises. */
int yk_get_stripe_bbox(yk_ctx* c, int32_t bbox[4]);
/* stage 2: the image-wide bbox (min/max over stripes; for a whole image pass yk_get_stripe_bbox's result or NULL
 * to use the handle's own).  Applies the reference rule "bbox == whole image -> all rejects discarded, no chunk"
 * (:1294, :1400-1403) and fixes boundX0..Y1 for the range quantiser.  Images without alpha: call with NULL or skip. */
int yk_alpha_finish(yk_ctx* c, const int32_t globalBBox[4]);
/* results: bounds[4] = boundX0,boundY0,boundX1,boundY1; hasChunk; remainingPixels (this stripe);
 * tileBBox[4] = MipmapHeader.bbox in 16-px tiles (x,y,w,h).  Synchronises. */
int yk_alpha_result(yk_ctx* c, int32_t bounds[4], int* hasChunk, int* remainingPixels, int32_t tileBBox[4]);
/* 'MIPM' payload: 1 bit per 16x16 tile inside tileBBox, row-major, LSB first, 1 = kept (:1317-1327).
 * For a stripe: only the bits of the owned rows are set (OR the stripes' buffers). cap >= (w*h+7)/8 of tileBBox. */
int yk_alpha_bitmap(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes);

/* ---- a6 + a10..a13  fused tile encode ----------------------------------------------------------
 * One launch does what 7x EncoderContext::FittingQuadSmooth(rejectFactor, R,G,B, .., sx, sy)
 * (EncoderContext.cpp:3710-4363) followed by DynamicTileEncode(mode3BitOnly, plane, dst, ..) for the three
 * planes (EncoderContext.cpp:4365-4602) compute, reading every input sample once.
 *   rejectFactor : the reference passes 3 (:9042)
 *   mode3BitOnly : DynamicTileEncode's first argument (Stats.startMode = 3, :4412)
 *   wantDst      : also produce the decoded-value planes `dst` (:4448-4457); costs 12 B/pixel of writes */
int yk_encode_tiles(yk_ctx* c, int rejectFactor, int mode3BitOnly, int wantDst);
/* The whole frame with ONE launch: alpha reject (RGBA) + fused kernel + stream compaction captured as a hipGraph the first
 * time and replayed afterwards (re-captured when the bound planes, the shape or the arguments change).  Same results as
 * yk_alpha_reject + yk_alpha_finish(NULL) + yk_encode_tiles(.., wantDst 0); for batches of small frames, where the ~8 stream
 * operations per frame cost more host time than the kernels take.  Whole images only (a stripe needs the host bbox step). */
int yk_encode_frame(yk_ctx* c, int rejectFactor, int mode3BitOnly);

/* ---- batches of equally shaped images (new; BASELINE config 4: 256 x 2048x2048 frames) --------------------------------
 * A frame of 2048x2048 is one round of waves: alone it cannot fill the chip.  A handle can hold nFrames images of one shape
 * (yk_set_image for the shape, then yk_set_batch) whose planes lie at frame0Planes[p] + f * frameStrideElems; yk_encode_batch
 * runs alpha reject + fused kernel + compaction over ALL frames with one launch per kernel (the grid simply spans
 * nFrames x strips), with the same per-frame results as yk_encode_frame.  yk_select_frame chooses the frame every getter,
 * yk_export_tile_maps, the corner streams and the 1-D path act on (default 0).  Whole images only, kernel version 2. */
int yk_set_batch(yk_ctx* c, int nFrames);
int yk_bind_device_batch(yk_ctx* c, const int32_t* const frame0Planes[4], int strideElems, size_t frameStrideElems);
int yk_encode_batch(yk_ctx* c, int rejectFactor, int mode3BitOnly);
int yk_select_frame(yk_ctx* c, int frame);

/* ---- streams of frames on several handles (new) --------------------------------------------------------------------------
 * With two handles (two streams) in flight the HBM-bound alpha / compaction kernels of one frame run under the fused kernel of
 * the other.  Two fused kernels sharing the chip only slow each other down, so a caller that alternates handles can order them:
 * the NEXT yk_encode_tiles of c launches its fused kernel after the fused kernel most recently launched on `other` has finished
 * (a stream-wait on an event, no host synchronisation; the alpha stage and the compaction of c are not held back).  Same device;
 * `other` must stay alive until that encode of c has been issued (the event belongs to it).  The request is consumed by the next
 * yk_encode_tiles / yk_encode_batch / yk_encode_frame of c (the last one holds its whole replay back) and dropped by yk_set_image. */
int yk_order_fused_after(yk_ctx* c, const yk_ctx* other);

/* gradient results (valid after yk_encode_tiles) -------------------------------------------------
 * swizzled 1-bit tile bitmap of pass p, byte-exact `pFillBitMap` (:3775-3777, bit rule :3801-3805,:4026;
 * size = HeaderGradientTile::getBitmapSwizzleSize/8, include/YAIK_private.h:278-286) */
size_t yk_gradient_bitmap_bytes(const yk_ctx* c, int pass);
int    yk_gradient_bitmap(yk_ctx* c, int pass, uint8_t* hostOut, size_t cap);
const uint8_t* yk_gradient_bitmap_device(const yk_ctx* c, int pass);
/* accepted-tile count per pass = FittingQuadSmooth's return value (TileDone, :4362). Synchronises. */
int    yk_gradient_counts(yk_ctx* c, int32_t counts[YK_NUM_PASSES]);
/* coverage after the 7 passes: 1 bit per 4x4 cell, u16 per 16x16 macro-tile (bit = cellY*4+cellX), row-major
 * macro-tiles.  Equals smoothMap/mapSmoothTile != 0 sampled per cell (:4029-4037). */
int    yk_coverage(yk_ctx* c, uint16_t* hostOut, size_t capElems);
/* corner-colour stream of pass p = `rgbStream` (:4113-4132): CompressF(Round6(corner),250) bytes of every corner
 * not yet in `mappedRGB`, in scan order, de-duplicated across passes.  Call for p = 0..6 in order. */
int    yk_gradient_corners(yk_ctx* c, int pass, uint8_t* hostOut, size_t cap, size_t* nBytes);
/* pass's corner stream where it lies in HBM (valid until the handle's next encode) and its length; builds the streams on first use */
int yk_gradient_corners_device(yk_ctx* c, int pass, const uint8_t** dev, size_t* nBytes);
/* (re)builds the seven corner streams on the device without copying anything out (yk_gradient_corners does it on first use) */
int    yk_gradient_corners_run(yk_ctx* c);
/* Row stripes (new; SURVEY §8e "corner dedup across stripe-boundary lattice rows ... on the root"): a stripe handle
 * de-duplicates inside its own rows, but its first and last lattice rows (y = y0 and y = y0 + h) are shared with the
 * neighbouring stripes.  For those two rows, n = w/4 + 1 points each (first row, then last row): keys[i] = the stripe-local
 * first-toucher key pass << 27 | bitIndex << 2 | corner (0xFFFFFFFF = untouched), index[i] = position, in corners, of that
 * point's colour inside the stripe's stream of that pass.  The root drops the later of two emissions of one point:
 * yaik_amd/distributed.py::merge_corner_streams.  capElems >= 2 * (w/4 + 1). */
int    yk_gradient_corner_edges(yk_ctx* c, uint32_t* hostKeys, uint32_t* hostIndex, size_t capElems);

/* ---- a6 with nullable planes: EncoderContext::FittingQuadSmooth(rejectFactor, srcA, srcB, srcC, ..., tileShiftX, tileShiftY) where one or
 * two of the planes are NULL (EncoderContext.cpp:3710-4363; PlaneBit :3715; the reference's call sites are the six 4x4 passes RB, RG, GB, R,
 * G, B of Convert(), :9261-9415, all compiled out or under `if (0)`).  planeBit: bit0/1/2 = srcA/srcB/srcC present (7 = a plain RGB pass).
 * Runs ONE more pass after yk_encode_tiles, on the state the seven RGB passes left (further calls continue from each other):
 *   - a tile is tried when its top-left pixel is uncovered in every PRESENT plane (:3871-3875); absent planes read 0 and never reject;
 *   - accepted tiles paint the per-plane coverage of the present planes and the common coverage (smoothMap; :4029-4037), so yk_coverage
 *     grows and yk_coverage_plane(p) tells what plane p still has to code in the 1-D path - yk_range1d_encode uses the per-plane
 *     coverage from the first partial pass on, exactly like DynamicTileCompressor(.., mapSmoothTile->GetPlane(p), ..) (:9451-9460);
 *   - the corner stream holds one byte per PRESENT plane whose mappedRGB[p] has not seen the lattice point (:4001-4021, :4113-4132).
 * The 2-D tile maps of yk_encode_tiles are not touched (the reference's partial passes precede only the 1-D compressor).
 * Single images only (nFrames == 1, whole image: y0 == 0).  *tilesAccepted = the function's return value (TileDone).  Synchronises. */
int    yk_gradient_partial_pass(yk_ctx* c, int rejectFactor, int planeBit, int tileShiftX, int tileShiftY, int* tilesAccepted);
/* testOutput of FittingQuadSmooth (:3960-3971, :4096-4104): the tiles pass `pass` accepted (0..6 = the RGB passes, 7 = the last plane-subset
 * pass) write blendC6Exp - the rounded bilinear blend of their Round6P corners - into three int32 preview planes the handle keeps per
 * encode (INT32_MIN where no tile wrote yet).  Calls accumulate; hostOut (may be NULL) receives the three planes, w*h each.  A debugging
 * aid of the encoder, not what the decoder reconstructs.  Single whole images. */
int    yk_gradient_preview(yk_ctx* c, int pass, int32_t* hostOut, size_t capElems);
/* bitmap (swizzled like yk_gradient_bitmap, size :3770-3777) and corner stream of the LAST partial pass */
int    yk_partial_bitmap(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes);
int    yk_partial_corners(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes);
/* mapSmoothTile->GetPlane(plane) != 0 per 4x4 cell, same layout as yk_coverage (equal to it until the first partial pass) */
int    yk_coverage_plane(yk_ctx* c, int plane, uint16_t* hostOut, size_t capElems);

/* ---- (f)4  3-D LUT tiles: EncoderContext::Load3DPattern (EncoderContext.cpp:7851), EvalCtx3D::Set3DPointCloud (:4744),
 * StartCorrelationSearch (:7316), Correlation3DSearch (:6245) with computeValues3D (:5807).  Runs where Convert() runs it (:9117-9218):
 * after the gradient passes, before the 1-D compressor, whose per-plane maps it shrinks (a matched tile covers all three planes of
 * mapSmoothTile, not smoothMap: yk_coverage_plane grows, yk_coverage does not).
 * yk_lut_load_pattern: one pattern of the bank = the contents of one 'Bank3D' .lut file (u8 count, r[], g[], b[], 6-bit values), at most 64
 * points (more make the reference overrun its tables, :7907-7917) and 64 patterns; *index = its number.  The bank belongs to the handle
 * and survives yk_set_image (the first pattern reserves the whole bank's tables in HBM: 64 MB of nearest-entry tables + 0.3 MB).  yk_lut_start allocates and clears the streams; yk_lut_search runs ONE tile shape (shiftX, shiftY) in
 * {(4,3),(3,4),(3,3),(3,2),(2,3),(2,2)} - the reference calls them in that order - and appends to the streams; *matched = tiles taken.
 * yk_lut_stream(which): 0 = corr3D_tileStreamTileType (u16: orientation | pattern << 6 | bitMode << 14, :6559), 1 = corr3D_colorStream
 * (box lo RGB, hi RGB per tile, before EndCorrelationSearch's CompressF :7494), 2..5 = corr3D_stream3Bit..6Bit (entry numbers, before the
 * x 3 of :7526), 6..11 = the tile maps of 16x8, 8x16, 8x8, 8x4, 4x8, 4x4 (BitmapSwizzleMapSize bytes, :7310).  Single whole images. */
int yk_lut_clear(yk_ctx* c);
int yk_lut_load_pattern(yk_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int count, int* index);
int yk_lut_pattern_tables(yk_ctx* c, int pattern, int16_t* factors, uint16_t* distanceField, uint8_t* positions);
int yk_lut_start(yk_ctx* c);
int yk_lut_search(yk_ctx* c, int tileShiftX, int tileShiftY, int* matched);
int yk_lut_stream(yk_ctx* c, int which, uint8_t* hostOut, size_t cap, size_t* nBytes);

/* range-quantiser results per plane (valid after yk_encode_tiles) ---------------------------------
 * tileDefs = `streamTileDef` u16 EncodeTileType(type,range,base) of tiles with >= 1 valid pixel, LeftRightOrder
 * (:4419,:4434-4438); nibbles = `streamTileIdx`, low nibble first (:1180-1184), closed to a whole byte (:4525). */
int yk_range_sizes(yk_ctx* c, int plane, size_t* nDefs, size_t* nNibbles);        /* synchronises */
int yk_range_streams(yk_ctx* c, int plane, uint16_t* hostDefs, size_t capDefs, uint8_t* hostNibbles, size_t capBytes);
const uint16_t* yk_range_defs_device(const yk_ctx* c, int plane);
const uint8_t*  yk_range_nibbles_device(const yk_ctx* c, int plane);
/* `dst` plane (int32 w*h of this stripe; untouched where no valid pixel; pre-filled with `fill`) */
int yk_range_dst(yk_ctx* c, int plane, int32_t* hostOut, size_t capElems);
int yk_set_dst_fill(yk_ctx* c, int32_t fill);

/* ---- a15  live 1-D range path:  3x EncoderContext::DynamicTileCompressor(stream, plane, mapSmoothTile[plane], debug)
 * (EncoderContext.cpp:8398-8522, call sites :9451-9465) on what the gradient passes left uncovered.  Produces the two
 * streams GenerateDynamicTileChunk hands to ZStd (:8524-8576): pixel bytes (planes R,G,B appended, 1 B per pixel) and the
 * per-tile parameter bytes color0,minCol,delta (`streamType`, :8503-8505).  colorCompression1D = 255, rangeCompression1D = 15. */
int yk_range1d_encode(yk_ctx* c);
/* enable != 0: from the next yk_encode_tiles on, the fused kernel also leaves the packed pixels of every 4x4 cell it did not cover (4 bytes per
 * pixel, only those cells) in a cache the 1-D path reads instead of the int32 planes: every input sample is then read from HBM once on the whole
 * path.  Costs the fused kernel four 16-byte stores per lane of an uncovered cell, so it is off unless a caller runs the 1-D path
 * (EncoderContext::ConvertHotPath turns it on).  The results of yk_range1d_encode are the same either way.  Not for batches. */
int yk_set_pixel_cache(yk_ctx* c, int enable);
int yk_range1d_streams(yk_ctx* c, uint8_t* hostPix, size_t capPix, size_t* nPix, uint8_t* hostType, size_t capType, size_t* nType);
/* the two streams where they lie in HBM (valid until the handle's next yk_range1d_encode / yk_set_image) and their lengths; synchronises once for the lengths */
int yk_range1d_streams_device(yk_ctx* c, const uint8_t** devPix, size_t* nPix, const uint8_t** devType, size_t* nType);
/* where plane p's share of the two streams ends (bytes, cumulative): the cursor DynamicTileCompressor returns after plane p (:8521) and
 * the end of its streamType entries.  Equal thirds unless a partial-plane pass gave the planes different coverage. */
int yk_range1d_plane_ends(yk_ctx* c, size_t pixEnd[3], size_t typeEnd[3]);

/* ---- tile-map export for the multi-GPU gather (new; the reference is single-process) -----------------------
 * Packs this handle's results into ONE caller-owned HBM buffer (device-to-device copies on the handle's stream) so
 * that a single RCCL gather can concatenate the per-stripe / per-frame tile maps.  Layout, every section padded to
 * 16 bytes: 7 gradient bitmaps | keep flags (1 byte per 16x16 tile, RGBA only) | per plane: tile defs (u16), nibbles.
 * sizes[0..6] bitmap bytes, sizes[7] keep bytes, sizes[8+2p] = nDefs(p), sizes[9+2p] = nNibbles(p), sizes[14] = total
 * bytes written.  One kernel packs all sections; the call returns after the stream has been synchronised, i.e. the buffer
 * is complete and may be handed to another runtime instance / RCCL.  cap >= yk_export_capacity.  The buffer is WRITTEN on the handle's
 * stream: work other streams still have queued on it (its fill, a previous consumer) must have finished or be ordered before that stream
 * (yk_stream_wait_for) -- the handle's stream is not ordered against any other. */
size_t yk_export_capacity(const yk_ctx* c);
int    yk_export_tile_maps(yk_ctx* c, void* devDst, size_t cap, uint64_t sizes[15]);
/* The same without a host synchronisation, for pipelines that keep the size table on the device: devMeta16 (device, 16 x u64)
 * receives {total payload bytes, sizes[0..14]}; work queued afterwards on `consumerStream` (same runtime instance, 0 = null
 * stream; e.g. the stream a RCCL collective is enqueued behind) sees the finished buffer and table (yk_stream_handoff). */
int    yk_export_tile_maps_async(yk_ctx* c, void* devDst, size_t cap, void* devMeta16, void* consumerStream);
/* The framed form the gather moves: devDst[0..127] = 16 x u64 header {payload bytes, sizes[0..14]}, the sections behind it (cap >= 128 +
 * yk_export_capacity).  A receiver needs nothing but the buffer: the size table travels inside it, and the length to transfer for a LATER
 * payload of the same rank is agreed from this header on both sides -- no second collective.  No host synchronisation; work queued
 * afterwards on consumerStream (0 = null stream, (void*)-1 = no hand-over: the gather then goes to the handle's own stream) sees the buffer. */
#define YK_EXPORT_HEADER_BYTES 128
int    yk_export_tile_maps_framed(yk_ctx* c, void* devDst, size_t cap, void* consumerStream);

/* ---- the multi-GPU gather of the tile maps (SURVEY 8(b) `yk_gather_maps`, 8(e); new: the reference is one process,
 * include/YAIK.h:42-47) ------------------------------------------------------------------------------------------------
 * Row stripes / frames are independent given their pixels; the only exchange is the concatenation of the per-rank tile maps on a root.
 * RCCL over xGMI, bound at run time (librccl.so.1); communicators are opaque pointers owned by the caller.  A gather is ONE grouped launch
 * of point-to-point transfers (every sender uses its own direct link into the root) on the handles' streams, behind the kernels that
 * produced the payloads: nothing is synchronised, the host never waits for the frame it just queued.
 *   one process per GPU:  id on rank 0 (yk_comm_unique_id), passed to the others by the launcher's own means, yk_comm_init_rank on
 *                         every rank, then yk_gather_maps per image / frame;
 *   one process, n GPUs:  yk_comm_init_all over n handles (one per device), then yk_gather_maps_all. */
/* HBM scratch for hosts that do not link the HIP runtime themselves (the C++ mirror is plain g++): buffers on the handle's device;
 * yk_device_download copies to the host on the handle's stream and synchronises it. */
int  yk_device_alloc(yk_ctx* c, size_t bytes, void** dev);
void yk_device_free(yk_ctx* c, void* dev);
int  yk_device_download(yk_ctx* c, void* host, const void* dev, size_t bytes);
int  yk_comm_available(void);                                           /* 1 when RCCL could be loaded */
int  yk_comm_unique_id(void* id128);                                    /* 128 bytes (ncclUniqueId) */
int  yk_comm_init_rank(yk_ctx* c, const void* id128, int nRanks, int rank, void** comm);
int  yk_comm_init_all(yk_ctx* const* ctxs, int n, void** comms);        /* comms[n] out */
int  yk_comm_ranks(void* comm, int* nRanks, int* rank);                 /* what RCCL reports for the communicator */
void yk_comm_destroy(void* comm);
/* Rank side (one process per GPU): every rank but `root` sends sendBytes from devSend; the root receives recvBytes[r] bytes of rank r at
 * devRecv + recvOffsets[r] (its own payload is copied there on the device; recvBytes / recvOffsets / devRecv may be NULL elsewhere).
 * Sender and root must name the same count for a rank: derive it on both sides from that rank's previous header. */
/* COUNTS MUST AGREE: rank r's sendBytes has to equal recvBytes[r] as the root posts it.  The library cannot check this across ranks, and an
 * ncclSend / ncclRecv pair with different counts is undefined in RCCL (a hang or a truncated payload).  Callers either exchange the sizes first
 * or use a fixed capacity per rank with the true length inside the payload (the framed export's 128-byte header carries it; that is what
 * yaik_amd/distributed.py and EncoderContext::ConvertHotPathStripes do). */
int  yk_gather_maps(yk_ctx* c, void* comm, int root, const void* devSend, size_t sendBytes,
                    void* devRecv, const size_t* recvBytes, const size_t* recvOffsets);
/* One process driving n devices: rank r's sendBytes[r] bytes land at devRecv + recvOffsets[r] on the root's device. */
int  yk_gather_maps_all(yk_ctx* const* ctxs, void* const* comms, int n, int root, void* const* devSend, const size_t* sendBytes,
                        void* devRecv, const size_t* recvOffsets);

/* ---- decode side: the loops behind YAIK_DecodeImage's chunk switch (decoder/YAIK_API.cpp:731-1303) ----
 * Buffers mirror YAIK_Instance (include/YAIK_private.h:26-54): planeR/G/B u8 in 8x8 tiles, mapRGB lattice,
 * tile4x4Mask.  Width/height multiples of 16 (the reference loops mis-stride otherwise, YAIK_Gradient.cpp:15). */
int yk_decode_begin(yk_ctx* c, int w, int h);
/* DecompressGradient16x16 .. 4x4 (decoder/YAIK_Gradient.cpp:28,203,401,599,800,999,1208), planeBit 7.
 * bitmap = swizzled tile bitmap, rgb = corner stream AFTER PaletteDecompressor (0..255). Host pointers. */
int yk_decode_gradient(yk_ctx* c, int tileShiftX, int tileShiftY, const uint8_t* bitmap, size_t bitmapBytes,
                       const uint8_t* rgb, size_t rgbBytes);
/* The same with both streams already in HBM on the handle's device (e.g. straight from an encoder handle: yk_gradient_bitmap_device,
 * yk_gradient_corners_device) and no host synchronisation.  remapRange > 0 applies PaletteFullRangeRemapping(range) to the colour stream on
 * the way in (decoder/YAIK_GenericFunctions.cpp:128-137; the encoder's streams are CompressF(.., 250) values), 0 takes it as it is. */
/* ORDERING is the caller's: the yk_decode_*_device entry points read the given device memory on THIS handle's stream and nothing orders that
 * stream behind the producer's.  When the streams come from an encoder handle (yk_gradient_corners_device, yk_range1d_streams_device,
 * yk_gradient_bitmap_device), order the two first: yk_synchronize(producer) on the host, or a device-side wait (yk_stream_handoff(producer,
 * consumerStream) / yk_stream_wait_for(c, producerStream) with the streams the caller gave the handles through yk_set_stream).  The pointers go
 * stale with the producer's next encode / yk_set_image. */
int yk_decode_gradient_device(yk_ctx* c, int tileShiftX, int tileShiftY, const uint8_t* devBitmap, size_t bitmapBytes,
                              const uint8_t* devRgb, size_t rgbBytes, int remapRange);
/* All 'GTIL' chunks of a file at once, streams in HBM: the same result as yk_decode_gradient_device for pass 0 .. nPasses-1 in that order
 * (first toucher of every lattice point over ALL passes in one launch, one scan, one launch popping the colours, then ONE render launch
 * in which a workgroup owns a 64x64 block of the image and walks the passes in call order: 5 launches for seven passes instead of 35 + 21 copies / clears).  Up to seven passes per
 * call and 2^25 tile slots per pass; beyond that the call runs the passes one after the other. */
int yk_decode_gradient_all_device(yk_ctx* c, int nPasses, const int* tileShiftX, const int* tileShiftY, const uint8_t* const* devBitmap,
                                  const size_t* bitmapBytes, const uint8_t* const* devRgb, const size_t* rgbBytes, int remapRange);
/* DecompressGradient4x4 with a plane subset (decoder/YAIK_Gradient.cpp:1208-1226 -> 4x4R / G / RG / B / RB / GB, :1420-2732): planeBit
 * 1..6 (bit 0 = R, 1 = G, 2 = B; 7 forwards to yk_decode_gradient).  Like YAIK_API.cpp:875-877 the masks are split per plane first
 * (UpdateTileAndRGBMask).  Only the 4x4 size has partial-plane loops in the reference.  consistentMarks = 0 reproduces what those loops
 * do to tile4x4Mask, defects included (the R / G / B loops never mark; GB / RB put the B marks at tile4x4Mask + (size >> 1), :1678,
 * :1924) -- byte-identical state to the reference decoder; 1 marks every present plane's own mask, which is what the ENCODER's
 * per-plane coverage and therefore the 1-D streams behind such passes assume (with 0 the reference's own Decompress1D runs off them). */
int yk_decode_gradient_planes(yk_ctx* c, int planeBit, int consistentMarks, const uint8_t* bitmap, size_t bitmapBytes,
                              const uint8_t* rgb, size_t rgbBytes);
/* UpdateTileAndRGBMask alone (decoder/YAIK_API.cpp:530-544): what a plane-subset chunk of any OTHER tile shape still does before its
 * decoder returns without work (YAIK_Gradient.cpp:29-36).  Idempotent. */
int yk_decode_split_masks(yk_ctx* c);
/* ---- (f)4 decode: YAIK_AssignLUT (decoder/YAIK_API.cpp:133-415) and the '3DTL' chunk, Tile3D_16x8 .. Tile3D_4x4 (decoder/YAIK_3DTile.cpp:244-2140,
 * chunk reader YAIK_API.cpp:1002-1270).  yk_decode_assign_lut takes the decoder's LUT file ('LUL0': LUTHeader + per depth 3..6 and pattern the
 * x, y, z entry lists) and lays out the 48 orientation tables per pattern and depth in HBM; it belongs to the handle until replaced.
 * yk_decode_lut3d runs the six tile shapes in chunk order on host streams: maps[k] / mapBytes[k] (16x8, 8x16, 8x8, 8x4, 4x8, 4x4; NULL or 0 =
 * absent), tiles (u16 per tile), colors (6 bytes per tile AFTER PaletteFullRangeRemapping), idx[f] / idxBytes[f] = the 3 / 4 / 5 / 6 bit index
 * streams as stored (entry number x 3).  Pixels of 4x4 cells tile4x4Mask already marks are skipped (and consume no index), every cell of a
 * decoded tile is marked afterwards.  consumed[6] = bytes used of tiles, colors and the four index streams.  Must come before '1DTL' /
 * plane-subset chunks (single-plane masks), like in the file. */
int yk_decode_assign_lut(yk_ctx* c, const uint8_t* lutFile, size_t lutBytes);
int yk_decode_lut3d(yk_ctx* c, const uint8_t* const maps[6], const size_t mapBytes[6], const uint16_t* tiles, size_t nTiles, const uint8_t* colors,
                    const uint8_t* const idx[4], const size_t idxBytes[4], size_t consumed[6]);
/* Decompress1D x3 planes (decoder/YAIK_3DTile.cpp:24-240) on the '1DTL' streams (type: 3 B/tile, pix: 1 B/pixel) */
int yk_decode_1d(yk_ctx* c, const uint8_t* typeStream, size_t typeBytes, const uint8_t* pixStream, size_t pixBytes,
                 int compressionRange);
/* streams already in HBM (yk_range1d_streams_device of an encoder handle on the same device), no host synchronisation */
int yk_decode_1d_device(yk_ctx* c, const uint8_t* devType, size_t typeBytes, const uint8_t* devPix, size_t pixBytes, int compressionRange);
/* Decompress1BitTiled (decoder/YAIK_Mipmap.cpp:23-154): 1 bit / 16x16 tile -> swizzled 1 bit / pixel mask */
int yk_decode_mask(yk_ctx* c, const uint8_t* bits, int tileBBoxW, int tileBBoxH, uint8_t* hostOut, size_t cap);
/* 8x8-tiled u8 planes exactly as YAIK_SCustomDataSource hands them to imageBuilderFunc (include/YAIK.h:205-224) */
int yk_decode_planes(yk_ctx* c, uint8_t* hostR, uint8_t* hostG, uint8_t* hostB, size_t capEach);
const uint8_t* yk_decode_planes_device(yk_ctx* c, size_t* planeSize);
/* internal_imageBuilderFunc (decoder/YAIK_DefaultCallback.cpp:24-191): de-tile into interleaved rows at outputImageStride.
 * Only the pixel bytes of a row are written; the rest of each outputImageStride-sized row is left untouched, like the reference
 * (include/YAIK.h:190: the stride places the image inside a larger user buffer).
 * hostAlpha == NULL -> RGB, 3 B/pixel, byte-identical to the reference (pinned by the compiled reference: tests/golden, blob dec_rgb_out).
 * With a linear 8-bit alpha plane (strideA bytes per row) yk_decode_output writes RGBA 4 B/pixel as include/YAIK.h documents.
 * The reference's own RGBA branch does something else (:45-62: the alpha store never advances dst and the alpha row cursors are
 * not moved per tile row): rows of w RGB triples + one alpha byte.  yk_decode_output_reference_rgba reproduces exactly that
 * (blob dec_rgba_out) for callers that need byte identity with the reference's output rather than a usable RGBA image. */
int yk_decode_output(yk_ctx* c, uint8_t* hostOut, size_t outputImageStride, const uint8_t* hostAlpha, int strideA);
int yk_decode_output_reference_rgba(yk_ctx* c, uint8_t* hostOut, size_t outputImageStride, const uint8_t* hostAlpha, int strideA);
int yk_decode_tile4x4(yk_ctx* c, uint8_t* hostOut, size_t cap);
/* the three planes of tile4x4Mask back to back (planes 1 and 2 are meaningful once a partial-plane pass has split the masks) */
int yk_decode_tile4x4_planes(yk_ctx* c, uint8_t* hostOut, size_t cap);

/* ---- timing hooks for bench.py: HIP events on the handle's stream around every alpha stage / fused kernel / compaction.
 * Returns the averages over the yk_encode_tiles calls since the previous query (a ring of 64 event sets, older ones are
 * dropped), so a caller can queue many frames back to back and read the per-kernel times once, without a sync per frame.
 * Synchronises with the most recent encode. */
int yk_last_kernel_ms(yk_ctx* c, float* fusedEncodeMs, float* alphaMs, float* packMs);
/* Device time of the stages outside the fused encode, from HIP events recorded on the launch stream around the stage's KERNELS
 * (host<->device copies of the decode entry points are outside the intervals).  Returns the sum of the intervals recorded since
 * the last query of that stage and their number, and resets both. */
enum { YK_STAGE_CORNERS = 0,       /* yk_gradient_corners: lattice clear + owner / count / scan / emit kernels of the 7 passes */
       YK_STAGE_RANGE1D = 1,       /* yk_range1d_encode: yk_range1d_kernel (the dominant kernel of the live 1-D path) */
       YK_STAGE_RANGE1D_PACK = 2,  /* yk_range1d_encode: scans + yk_range1d_pack_kernel */
       YK_STAGE_DEC_GRADIENT = 3,  /* yk_decode_gradient: owner / corner / scan / render kernels of one pass per interval */
       YK_STAGE_DEC_1D = 4,        /* yk_decode_1d: count / scans / yk_dec1d_kernel */
       YK_STAGE_DEC_DETILE = 5,    /* yk_decode_output: yk_dec_detile_kernel */
       YK_STAGE_LUT3D = 6 };       /* yk_lut_search: yk_lut_search_kernel (one interval per tile shape) */
int yk_stage_ms(yk_ctx* c, int stage, float* msSum, int* intervals);

/* ---- diagnostics: the MEASURED HBM roof of this device (roofline fractions are quoted against the 8 TB/s specification
 * AND against what the part really streams).  Two hand-written kernels with 16-byte accesses, eight loads per lane in flight, on a
 * scratch pair of `bytes` each (allocated and freed inside the call; >= 1 MiB): *copyGBs = best of `reps` of dst[i] = src[i], read + write
 * bytes counted (the figure MI355X_MICROARCH.md quotes: 6.29 TB/s); *readGBs = best of `reps` of a read-only stream.  Synchronises the
 * handle's stream.  No part of the tile path. */
int yk_measure_roof(yk_ctx* c, size_t bytes, int reps, double* copyGBs, double* readGBs);

#ifdef __cplusplus
}
#endif
#endif /* YAIK_HIP_H */
