/* orbx.h -- C ABI of the MI355X-native ORB extractor (drop-in for ORB_SLAM3::ORBextractor).
 *
 * The reference has no FFI layer: its boundary is the C++ class used by Frame/Tracking
 * (include/ORBextractor.h:49-83, src/ORBextractor.cc:468-571,1534-1659).  The C++ facade
 * orb-slam3_amd/facade/ORBextractor.h re-exposes that class over these entry points.
 *
 * Conventions: no exceptions cross this ABI; negative return = error (ORBX_E_*); all output buffers are
 * caller-owned; a handle is NOT thread-safe (one per concurrent caller, exactly like the reference's
 * stateful extractor); the device is selected at create time.  There is NO CPU fallback:
 * every entry point that computes fails with ORBX_E_HIP when no gfx950 device is usable.
 */
#ifndef ORBX_H_
#define ORBX_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbx orbx_t;

/* == cv::KeyPoint memory layout, 28 bytes (replaces std::vector<cv::KeyPoint>& of ORBextractor.h:57-59) */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orbx_kp_t;

enum {
    ORBX_OK = 0,
    ORBX_E_EMPTY = -1,        /* empty image: the reference's `return -1` (ORBextractor.cc:1538-1539) */
    ORBX_E_INVALID = -2,      /* bad argument */
    ORBX_E_CAPACITY = -3,     /* caller buffer / configured batch too small */
    ORBX_E_TOO_SMALL = -4,    /* a pyramid level is narrower than one 35-px FAST cell (reference divides by 0) */
    ORBX_E_HIP = -5,          /* HIP runtime error or no device; orbx_last_error() has the text */
    ORBX_E_UNSUPPORTED = -6,  /* size limit of the on-chip quadtree / packed coordinates exceeded */
    ORBX_E_INTERNAL = -7      /* device-side overflow flag raised (never expected) */
};

/* flags for image / result pointers */
enum { ORBX_HOST = 0, ORBX_DEVICE = 1 };

/* replaces ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.cc:468-571).  max_w/max_h/max_batch size the device-resident pyramid and scratch. */
int orbx_create(orbx_t** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                int device_id, int max_w, int max_h, int max_batch);
void orbx_destroy(orbx_t*);
const char* orbx_last_error(void);

/* upper bound of keypoints one image can return (>= nfeatures + 3*nlevels, see DistributeOctTree) */
int orbx_max_keypoints(const orbx_t*);

/* replaces ORBextractor::operator() (ORBextractor.cc:1534-1659) for one image.
 * img: host pointer, row stride in bytes.  kps/desc: host buffers with room for `cap` entries
 * (desc is cap x 32 bytes).  Returns n >= 0 keypoints, writes *mono_index (the reference's return value).
 * Synchronous.  With per-stage timing off (orbx_set_stage_timing(o, 0), what the C++ facade does) the third and later calls with
 * an unchanged image size replay the whole frame -- staging copy, both streams' kernels, result packing -- as one captured graph;
 * ORBX_ONE_GRAPH=0 in the environment keeps every call on the eager path. */
int orbx_extract(orbx_t*, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                 orbx_kp_t* kps, uint8_t* desc, int cap, int* mono_index);

/* frame-parallel batch (all images w x h, same stride).  imgs[i] are host or device pointers
 * (img_space = ORBX_HOST / ORBX_DEVICE).  lap01 = 2 ints per image or NULL for {0,0}.
 * Results stay on the device (orbx_result_*); n_out/mono_out (host, may be NULL) receive per-image counts
 * -- passing them forces a stream sync.  Returns 0 or ORBX_E_*. */
int orbx_extract_batch(orbx_t*, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                       const int* lap01, int* n_out, int* mono_out);
/* enqueue only (no sync, no D2H): the timed body of bench.py.  Same arguments as above minus outputs. */
int orbx_extract_batch_async(orbx_t*, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                             const int* lap01);
int orbx_sync(orbx_t*);

/* device-resident results of the last batch: kps = [max_batch][cap] orbx_kp_t, desc = [max_batch][cap][32],
 * counts = [max_batch] int32 (n), monos = [max_batch] int32; cap = orbx_max_keypoints(). */
int orbx_result_device(const orbx_t*, const orbx_kp_t** kps, const uint8_t** desc, const int32_t** counts,
                       const int32_t** monos, int* cap);
/* copy image `i` of the last batch to host buffers; returns n or ORBX_E_* */
int orbx_result_fetch(orbx_t*, int i, orbx_kp_t* kps, uint8_t* desc, int cap, int* mono_index);
/* all frames of the last batch at once: kps / desc are [nimg][cap_per_img] (one device-to-host copy each when cap_per_img ==
 * orbx_max_keypoints()); n_out / mono_out [nimg].  Returns nimg. */
int orbx_result_fetch_all(orbx_t*, orbx_kp_t* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out);

/* mvImagePyramid back-door (include/ORBextractor.h:83; used by Frame::ComputeStereoMatches, Frame.cc:1168) */
int orbx_level_size(const orbx_t*, int level, int* w, int* h);
int orbx_level_image(orbx_t*, int frame, int level, int blurred, uint8_t* dst, int dst_stride);
/* all levels of one frame of the last batch at once: dst[l] / dst_stride[l] per level (dst[l] = NULL skips a level).  One copy of the
 * frame's pyramid slab out of HBM plus host row copies; level 0 of a host-image call comes from the staging buffer.  This is what
 * the facade fills mvImagePyramid (include/ORBextractor.h:83) with after operator(). */
int orbx_pyramid_fetch(orbx_t*, int frame, uint8_t* const* dst, const int* dst_stride);
/* the same without host copies: ptr[l] / pitch[l] point into pinned memory of the handle (level 0: the staging buffer of a host
 * image, NULL for a device-resident frame), valid until the next extraction or pyramid call on this handle -- the lifetime the
 * reference's mvImagePyramid has (the extractor overwrites it on every call). */
int orbx_pyramid_map(orbx_t*, int frame, const uint8_t** ptr, int* pitch);
/* GetScaleFactors & co (include/ORBextractor.h:61-79) */
void orbx_scale_tables(const orbx_t*, float* sf, float* inv_sf, float* sig2, float* inv_sig2);
int orbx_features_per_level(const orbx_t*, int* nfeat);

/* stage introspection for parity tests: FAST candidates handed to the quadtree for (frame, level), in
 * the reference's vToDistributeKeys order, as (x,y,response) int triples relative to (16,16). */
int orbx_level_candidates(orbx_t*, int frame, int level, int32_t* xyr, int cap);
/* keypoints chosen by the quadtree for (frame, level) in list order: (x,y,response) in level coordinates */
int orbx_level_selected(orbx_t*, int frame, int level, int32_t* xyr, int cap);

/* timing: HIP events of the most recent async batch, recorded on the stream each kernel group is launched on [ms]:
 * 0 pyramid (resize chain, stream 2), 1 FAST (level 0 + levels 1.., stream 1), 2 quadtree, 3 slots, 4 blur (stream 2),
 * 5 orient+descriptor, 6 total, 7 wall span from the first resize/FAST launch to the end of the last FAST launch
 * (the two streams overlap, so 0+1 double-counts; 7 is the figure the roofline uses).  ms8 holds 8 floats. */
int orbx_last_timings(orbx_t*, float* ms8);
/* 1 when the Gaussian blur is scheduled inside that pass (the default: the matrix-core blur runs directly behind the resize
 * chain, beside FAST): ms8[7] then ends with the later of FAST and blur, and the pass's algorithmic traffic includes the
 * blur's read + write of every level (2 * sum of level pixels).  0 with ORBX_BLUR_V2 / ORBX_BLUR_LATE. */
int orbx_blur_in_pass(const orbx_t*);
/* stage-boundary events are optional: with `on` = 0 only the dependency events are recorded (a few microseconds less per
 * batch); orbx_last_timings / orbx_mean_timings then fill ms8[6] (total), ms8[7] (the pass: to the later of FAST and an in-pass blur)
 * and ms8[1] (first launch -> end of the last FAST launch, i.e. pyramid+FAST without the blur's tail) and zero the rest.
 * Default on.  Restarts the timing ring. */
int orbx_set_stage_timing(orbx_t*, int on);
/* mean of the same 8 figures over the most recent (<= 32) enqueued batches; *nsamples = how many were averaged.
 * Lets a caller keep enqueueing without a host sync per batch and still report HIP-event kernel times. */
int orbx_mean_timings(orbx_t*, float* ms8, int* nsamples);
/* stream ordering without host syncs: make `other_stream` (a hipStream_t, e.g. orbm_stream()) wait for everything
 * enqueued so far on this extractor / make this extractor's next batch wait for `other_stream` (a consumer still
 * reading the result buffers of the previous batch). */
int orbx_stream_wait_results(orbx_t*, void* other_stream);
int orbx_stream_wait_other(orbx_t*, void* other_stream);
/* finer form of orbx_stream_wait_other for a consumer that only READS the result block (keypoints, descriptors, counts) of
 * the previous batch on `reader_stream`: the next batch starts at once and waits for the reader's work enqueued so far only
 * before its first kernel that writes the result block -- a matcher on its own stream then runs beside the next batch's
 * pyramid / FAST phase.  Call after enqueueing the reader's work and before the next orbx_extract_batch_async. */
int orbx_guard_results(orbx_t*, void* reader_stream);
/* HIP-graph replay of an enqueue sequence.  Between orbx_capture_begin(o, slot) and orbx_capture_end(o) every *_async /
 * enqueue-only call on this handle -- orbx_extract_batch_async on device-resident 16-byte aligned images,
 * orbx_result_download_async, the stream-ordering helpers, and matcher calls whose stream is the extractor's
 * (orbm_set_stream(m, orbx_stream(o))) -- is recorded instead of run; orbx_graph_launch(o, slot) then replays the whole
 * two-stream fork/join (plus the copy stream) with ONE runtime call.  The same batch (image pointers, size, lapping areas)
 * must have been enqueued once eagerly before, so that geometry and per-frame tables are in place.  slot in [0, 8): every
 * slot keeps its own time stamps -- a captured sequence marks start / end of FAST / end of blur / end of batch with a one-lane
 * wall-clock kernel, because event records stamp nothing on replay -- and orbx_mean_timings averages the latest replay of each
 * launched slot (total and pass span only; the per-stage figures need an eager batch).  A change of image
 * size drops all graphs.  No allocation, host sync or host-image staging may happen inside a capture. */
int orbx_capture_begin(orbx_t*, int slot);
int orbx_capture_end(orbx_t*);
int orbx_graph_launch(orbx_t*, int slot);
/* results -> host (the reference's consumers read mvKeys / mDescriptors on the host, src/Frame.cc:357-366).  The handle owns
 * a ring of EIGHT result blocks; a batch writes the current one (orbx_set_result_block(o, 0..7), default 0; orbx_result_device /
 * orbx_result_fetch* refer to it).  A block is one allocation holding kps [max_batch][cap], desc [max_batch][cap][32],
 * counts [max_batch] and monos [max_batch] at the byte offsets orbx_result_block_layout reports (they change with the image
 * size).  orbx_result_download_async hands the current block to the handle's copy thread: it waits for the batch that fills
 * the block, moves it with ONE transfer (copy engine, copy stream) into a PINNED host block of the same layout
 * (orbx_host_alloc(bytes)) and marks the block free again.  The call itself returns at once; a caller that walks the ring batch
 * by batch gets the copy of batch i beside the kernels of batch i+1.  Whoever is about to rewrite a block whose copy has not
 * landed -- orbx_extract_batch_async, orbx_graph_launch, another download of the same block -- waits for it on the host first.
 * Never inside a capture.  orbx_download_sync (or orbx_sync) returns when every requested copy has landed. */
/* caller-owned DEVICE buffers that belong to a block's batch (what a matcher computed from it: mvuRight, mvDepth, match lists)
 * can ride along: after orbx_block_attach(o, block, dev, bytes, &off) every download of that block also copies `bytes` from
 * `dev` to host_block + off (off >= the block's own size; attachments follow each other, 256-byte aligned, so the pinned block
 * must be that much larger).  A change of image size, or orbx_block_detach_all, drops the attachments. */
int orbx_block_attach(orbx_t*, int block, const void* dev, size_t bytes, size_t* host_offset);
int orbx_block_detach_all(orbx_t*);
int orbx_result_block_layout(const orbx_t*, size_t* off_kps, size_t* off_desc, size_t* off_counts, size_t* off_monos, size_t* bytes);
int orbx_result_download_async(orbx_t*, void* host_block);
int orbx_set_result_block(orbx_t*, int block);
int orbx_download_sync(orbx_t*);
/* two caller-placed time stamps on the extractor's stream (which = 0 / 1) and the time between them: the GPU-side wall time
 * of a run of batches or graph replays, gaps included */
int orbx_mark(orbx_t*, int which);
int orbx_mark_elapsed_ms(orbx_t*, float* ms);
void* orbx_host_alloc(size_t bytes);   /* pinned host memory */
void orbx_host_free(void*);
/* SURVEY 8(f).4 image ingest -- replaces cv::cvtColor(im, gray, COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) in Tracking::GrabImage*
 * (src/Tracking.cc:1264-1290, 1339-1348, 1393-1402).  nimg interleaved 8-bit colour images (src_space = ORBX_HOST | ORBX_DEVICE)
 * are converted into the caller's DEVICE buffers dst[i] (dst_stride bytes per row; a multiple of 16 lets orbx_extract_batch*
 * use them in place).  gray = (R*RY + G*GY + B*BY + half) >> coef_bits; coef_bits = 14: OpenCV 3.x coefficients
 * (4899, 9617, 1868), 15: OpenCV 4.x (9798, 19235, 3735).  The three ingest entry points only ENQUEUE on the extractor's
 * stream (ordered before the next orbx_extract_batch_async; no allocation, no host sync): call orbx_sync before reading
 * their output from the host, and keep device inputs alive until then. */
int orbx_gray_from_color(orbx_t*, const uint8_t* const* src, int src_space, int nimg, int w, int h, int src_stride,
                         int channels, int blue_first, int coef_bits, uint8_t* const* dst, int dst_stride);
/* SURVEY 8(f).4 stereo rectification -- replaces cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of the stereo examples
 * (Examples/Stereo/stereo_euroc.cc:168-169) for 8-bit single-channel images with CV_32F maps, BORDER_CONSTANT 0.
 * src[i] / dst[i] / mapx / mapy are DEVICE pointers (the maps are uploaded once, e.g. with orbx_dev_alloc + orbx_memcpy_h2d);
 * maps and destination are dw x dh.  Enqueued on the extractor's stream. */
int orbx_remap_linear(orbx_t*, const uint8_t* const* src, int nimg, int sw, int sh, int src_stride, const float* mapx, const float* mapy,
                      int dw, int dh, uint8_t* const* dst, int dst_stride);
/* SURVEY 8(f).4  cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))->apply(im, im) of the TUM-VI examples
 * (Examples/Monocular/mono_tum_vi.cc:101-109) for 8-bit single-channel DEVICE images (src[i] may equal dst[i]: the tile LUTs
 * are finished before any pixel is rewritten).  OpenCV 4.x semantics (see oracle/orbref.h: orbref_clahe). */
int orbx_clahe(orbx_t*, const uint8_t* const* src, int nimg, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
               uint8_t* const* dst, int dst_stride);
/* algorithmic bytes of the pyramid+FAST pass for one frame of the current geometry */
int64_t orbx_algorithmic_bytes(const orbx_t*, int64_t* fused_lower_bound);
void* orbx_stream(const orbx_t*);     /* hipStream_t the kernels are launched on */

/* plain device-memory helpers so callers need no HIP bindings of their own */
void* orbx_dev_alloc(size_t bytes);
void orbx_dev_free(void*);
int orbx_memcpy_h2d(void* dst, const void* src, size_t bytes);
int orbx_memcpy_d2h(void* dst, const void* src, size_t bytes);
int orbx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
