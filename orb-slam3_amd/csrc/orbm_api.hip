// orbm_api.hip -- C ABI (include/orbm.h) over the gfx950 matcher kernels.  No CPU fallback for the
// device entry points; orbm_hamming / orbm_three_maxima are the reference's scalar helpers and stay scalar.
#include "../../include/orbm.h"
#include "orbm_kernels.hip.h"

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace orbmk;

// A/B switches and test knobs exist only in the -DORBX_AB build (see orbx_api.hip)
#ifdef ORBX_AB
static inline const char* ab_env(const char* name) { return getenv(name); }
#else
static inline const char* ab_env(const char*) { return nullptr; }
#endif

static thread_local std::string g_merr;
static void set_merr(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_merr = buf;
}
#define MHIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_merr("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return ORBM_E_HIP; } } while (0)

struct orbm {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t ownStream = nullptr;                           // the stream created with the handle (orbm_set_stream may point `stream` elsewhere)
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    bool timed = false, gridFirst = false;
    // Scratch arena of the single-frame entry points: ONE device block and a pinned host mirror of the same size.  Uploads are
    // staged in the mirror and sent with one asynchronous copy before the kernel, results come back with one copy after it
    // (a dozen hipMalloc / small pageable copies per call cost over a millisecond).
    uint8_t* arDev = nullptr; uint8_t* arPin = nullptr;
    size_t arCap = 0, arOff = 0, arUp = 0, arOutLo = (size_t)-1, arOutHi = 0;   // bump offset; [0, arUp) staged uploads; small outputs in [arOutLo, arOutHi)
    uint8_t* scr = nullptr; size_t scrCap = 0;                 // grow-only device scratch of the batched (enqueue-only) entry points
    std::vector<uint8_t*> scrOld;                              // superseded scratch blocks: graphs captured earlier still hold their addresses, so they live as long as the handle
    int* hStatus = nullptr;                                    // pinned, device-visible status word of the enqueue-only entry points (capacity overflows); read by orbm_sync
};

// scratch of the enqueue-only entry points: grows outside a capture only (an allocation cannot be recorded into a graph)
static uint8_t* batch_scratch(orbm* m, size_t bytes) {
    if (bytes <= m->scrCap && m->scr) return m->scr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(m->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) return nullptr;
    // grow by ADDING: a step graph captured through orbx_capture_begin keeps the old block's address in its kernel nodes, and
    // a later eager call with more pairs must not turn that graph's next replay into a use-after-free
    uint8_t* nb = nullptr;
    if (hipMalloc((void**)&nb, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (m->scr) m->scrOld.push_back(m->scr);
    m->scr = nb; m->scrCap = bytes;
    return m->scr;
}

// time stamps: under stream capture (the caller replays this enqueue sequence as a HIP graph, orbx_capture_begin) an event
// record stamps nothing on replay, so it is skipped there: orbm_last_timing keeps reporting the last eagerly enqueued call
static hipError_t rec_time(orbm* m, hipEvent_t ev) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(m->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) return hipSuccess;
    return hipEventRecord(ev, m->stream);
}

// A frame kept in HBM across searches (orbm_frame_create): keypoints, descriptors, mvuRight and the 64 x 48 grid as CSR.
struct orbm_dframe {
    int device = 0, n = 0;
    uint8_t* block = nullptr;                                  // owned allocation: [kps | desc | uright] (host-created frames) + [grid_start | grid_idx]
    const KpIn* dKps = nullptr; const uint8_t* dDesc = nullptr; const float* dUr = nullptr;
    int *dGs = nullptr, *dGi = nullptr;
    std::vector<orbm_kp_t> hkps;                               // host copy: the replay reads angle / octave of the matched keypoint
    float min_x = 0, min_y = 0, inv_w = 0, inv_h = 0;
};

static inline size_t grid_cs_lds(int n) { return (size_t)(2 * GRID_CELLS + 1) * sizeof(int) + (size_t)std::max(n, 1) * sizeof(unsigned short) + 16; }

namespace { int knn2_host(orbm* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride, const int32_t* nt,
                          int npairs, int32_t* idx2, int32_t* dist2); }

extern "C" {

const char* orbm_last_error(void) { return g_merr.c_str(); }

int orbm_create(orbm_t** out, int device_id) {
    if (!out) return ORBM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_merr("no HIP device: the MI355X matcher has no CPU fallback"); return ORBM_E_HIP; }
    if (device_id < 0 || device_id >= ndev) { set_merr("device %d out of range", device_id); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(device_id));
    orbm* m = new orbm;
    m->device = device_id;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->e0) != hipSuccess ||
        hipEventCreate(&m->e1) != hipSuccess || hipEventCreate(&m->e2) != hipSuccess ||
        hipHostMalloc((void**)&m->hStatus, 64, hipHostMallocDefault) != hipSuccess) { set_merr("stream/event creation failed"); orbm_destroy(m); return ORBM_E_HIP; }
    *m->hStatus = 0;
    m->ownStream = m->stream;
    *out = m;
    return ORBM_OK;
}

void orbm_destroy(orbm_t* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->ownStream) { (void)hipStreamSynchronize(m->ownStream); (void)hipStreamDestroy(m->ownStream); }
    if (m->arDev) (void)hipFree(m->arDev);
    if (m->scr) (void)hipFree(m->scr);
    for (uint8_t* p : m->scrOld) (void)hipFree(p);
    if (m->hStatus) (void)hipHostFree(m->hStatus);
    if (m->arPin) (void)hipHostFree(m->arPin);
    if (m->e0) (void)hipEventDestroy(m->e0);
    if (m->e1) (void)hipEventDestroy(m->e1);
    if (m->e2) (void)hipEventDestroy(m->e2);
    delete m;
}

int orbm_sync(orbm_t* m) {
    if (!m) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    if (m->hStatus && *(volatile int*)m->hStatus) {             // set by a kernel of an enqueue-only call since the last sync
        const int st = *(volatile int*)m->hStatus;
        *(volatile int*)m->hStatus = 0;
        set_merr("a batched call overflowed a device-side list (status 0x%x: 1 = stereo row lists): results of that call are incomplete", st);
        return ORBM_E_CAPACITY;
    }
    return ORBM_OK;
}

void* orbm_stream(const orbm_t* m) { return m ? (void*)m->stream : nullptr; }

int orbm_set_stream(orbm_t* m, void* stream) {
    if (!m) return ORBM_E_INVALID;
    hipStream_t ns = stream ? (hipStream_t)stream : m->ownStream;
    if (ns == m->stream) return ORBM_OK;                        // nothing to drain (and safe while that stream is being captured)
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    m->stream = ns;
    return ORBM_OK;
}

int orbm_hamming(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32); memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

void orbm_three_maxima(const int* sz, int L, int* ind3) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; ++i) {
        const int s = sz[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
        else if (s > max3) { max3 = s; i3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if (max3 < 0.1f * (float)max1) i3 = -1;
    ind3[0] = i1; ind3[1] = i2; ind3[2] = i3;
}

static int knn2_launch(orbm_t* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                       const int32_t* nt, int npairs, int32_t* idx2, int32_t* dist2, double ratio, uint8_t* good);

int orbm_knn2_batch_async(orbm_t* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                          const int32_t* nt, int npairs, int max_nt, int32_t* idx2, int32_t* dist2) {
    (void)max_nt;
    return knn2_launch(m, q, q_stride, nq, t, t_stride, nt, npairs, idx2, dist2, 0.0, nullptr);
}

int orbm_knn2_ratio_batch_async(orbm_t* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                                const int32_t* nt, int npairs, double ratio, int32_t* idx2, int32_t* dist2, uint8_t* good) {
    if (!good) return ORBM_E_INVALID;
    return knn2_launch(m, q, q_stride, nq, t, t_stride, nt, npairs, idx2, dist2, ratio, good);
}

static int knn2_launch(orbm_t* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                       const int32_t* nt, int npairs, int32_t* idx2, int32_t* dist2, double ratio, uint8_t* good) {
    if (!m || !q || !t || !nq || !nt || !idx2 || !dist2 || npairs < 1 || q_stride < 1 || t_stride < 1) return ORBM_E_INVALID;
    if (t_stride >= (1 << 22)) { set_merr("knn2: more than 2^22 train descriptors per pair"); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(m->device));
    m->gridFirst = false;
    MHIPCHK(rec_time(m, m->e0));
    // matrix-core kernel unless the train set is beyond its 19-bit row field (or ORBM_KNN2_VALU asks for the popcount kernel: A/B)
    const bool forceValu = ab_env("ORBM_KNN2_VALU") != nullptr;   // read per call: tests flip it
    if (t_stride <= KM_MAX_NT && !forceValu)
        hipLaunchKernelGGL(k_knn2_mfma, dim3((q_stride + 255) / 256, npairs), dim3(256), 0, m->stream, q, q_stride, nq, t, t_stride, nt, idx2, dist2, ratio, good);
    else
        hipLaunchKernelGGL(k_knn2, dim3((q_stride + 63) / 64, npairs), dim3(256), 0, m->stream, q, q_stride, nq, t, t_stride, nt, idx2, dist2, ratio, good);
    MHIPCHK(rec_time(m, m->e1));
    MHIPCHK(hipGetLastError());
    m->timed = true;
    return ORBM_OK;
}

int orbm_knn2_batch(orbm_t* m, int space, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                    const int32_t* nt, int npairs, int32_t* idx2, int32_t* dist2) {
    if (!m) return ORBM_E_INVALID;
    if (space == ORBM_DEVICE) {
        int rc = orbm_knn2_batch_async(m, q, q_stride, nq, t, t_stride, nt, npairs, t_stride, idx2, dist2);
        if (rc) return rc;
        return orbm_sync(m);
    }
    if (!q || !t || !nq || !nt || !idx2 || !dist2 || npairs < 1 || q_stride < 1 || t_stride < 1) return ORBM_E_INVALID;
    return knn2_host(m, q, q_stride, nq, t, t_stride, nt, npairs, idx2, dist2);
}

int orbm_last_timing(orbm_t* m, float* ms) {
    if (!m || !m->timed) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    MHIPCHK(hipEventElapsedTime(ms, m->gridFirst ? m->e2 : m->e0, m->e1));
    return ORBM_OK;
}

}  // extern "C"

// =================================================================================================
// Flattened searches: GPU distance phase + host replay of the reference's sequential bookkeeping
// =================================================================================================
extern "C" int orbx_internal_levels(void* o, int frame, int* nlevels, const uint8_t** ptr, int* pitch, int* w, int* h,
                                    float* sf, float* isf, int* device);

namespace {

// Arena sub-allocations.  A DevBuf is an (offset, size) view, resolved against the arena's current blocks when used, so the
// arena may grow (reallocate) while a function is still collecting its buffers.  Protocol inside one entry point:
//   arena_reset(m); UP(...) / AL(...) for every buffer (uploads first); ARENA_FLUSH(m); launch; sync; ARENA_FETCH(m); read .host()
struct DevBuf {
    orbm* m = nullptr; size_t off = 0, bytes = 0; bool ok = false;
    template <class T> T* as() { return ok ? (T*)(m->arDev + off) : nullptr; }
    void* ptr() { return as<void>(); }
    const void* host() const { return m->arPin + off; }
    bool alloc(orbm* mm, size_t n) {
        m = mm; bytes = n;
        const size_t start = (m->arOff + 255) & ~(size_t)255, end = start + std::max<size_t>(n, 4);
        if (end > m->arCap) {                               // grow: new blocks, staged bytes move over, old blocks go
            const size_t ncap = std::max(end * 2, (size_t)1 << 20);
            uint8_t *nd = nullptr, *np = nullptr;
            if (hipMalloc((void**)&nd, ncap) != hipSuccess || hipHostMalloc((void**)&np, ncap, hipHostMallocDefault) != hipSuccess) {
                if (nd) (void)hipFree(nd);
                return false;
            }
            if (m->arPin && m->arOff) memcpy(np, m->arPin, m->arOff);
            (void)hipStreamSynchronize(m->stream);
            if (m->arDev) (void)hipFree(m->arDev);
            if (m->arPin) (void)hipHostFree(m->arPin);
            m->arDev = nd; m->arPin = np; m->arCap = ncap;
        }
        off = start; m->arOff = end; ok = true;
        return true;
    }
    bool upload(orbm* mm, const void* src, size_t n) {
        if (!alloc(mm, n)) return false;
        if (n) memcpy(m->arPin + off, src, n);
        m->arUp = m->arOff;
        return true;
    }
};
static inline void arena_reset(orbm* m) { m->arOff = 0; m->arUp = 0; m->arOutLo = (size_t)-1; m->arOutHi = 0; }
#define UP(buf, src, bytes) do { if (!(buf).upload(m, (src), (bytes))) { set_merr("device upload failed (%zu B)", (size_t)(bytes)); return ORBM_E_HIP; } } while (0)
#define AL(buf, bytes) do { if (!(buf).alloc(m, (bytes))) { set_merr("device allocation failed (%zu B)", (size_t)(bytes)); return ORBM_E_HIP; } \
                            m->arOutLo = std::min(m->arOutLo, (buf).off); m->arOutHi = std::max(m->arOutHi, m->arOff); } while (0)
// large output read back selectively by the caller (not part of the ARENA_FETCH range): allocate these LAST
#define AL_BIG(buf, bytes) do { if (!(buf).alloc(m, (bytes))) { set_merr("device allocation failed (%zu B)", (size_t)(bytes)); return ORBM_E_HIP; } } while (0)
#define UPIO(buf, src, bytes) do { UP(buf, src, bytes); m->arOutLo = std::min(m->arOutLo, (buf).off); m->arOutHi = std::max(m->arOutHi, m->arOff); } while (0)   /* in/out buffer */
// one host-to-device copy for everything staged so far / one device-to-host copy of everything behind the uploads
#define ARENA_FLUSH(m) do { if ((m)->arUp) MHIPCHK(hipMemcpyAsync((m)->arDev, (m)->arPin, (m)->arUp, hipMemcpyHostToDevice, (m)->stream)); } while (0)
#define ARENA_FETCH(m) do { if ((m)->arOutHi > (m)->arOutLo) MHIPCHK(hipMemcpy((m)->arPin + (m)->arOutLo, (m)->arDev + (m)->arOutLo, (m)->arOutHi - (m)->arOutLo, hipMemcpyDeviceToHost)); } while (0)

// host-array form of orbm_knn2_batch (Frame::ComputeStereoFishEyeMatches hands over host descriptors): through the arena
int knn2_host(orbm* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride, const int32_t* nt,
              int npairs, int32_t* idx2, int32_t* dist2) {
    MHIPCHK(hipSetDevice(m->device));
    const size_t qb = (size_t)npairs * q_stride * 32, tb = (size_t)npairs * t_stride * 32, ob = (size_t)npairs * q_stride * 2 * sizeof(int);
    DevBuf dq, dt, dnq, dnt, di, dd;
    arena_reset(m);
    UP(dq, q, qb); UP(dt, t, tb); UP(dnq, nq, sizeof(int) * npairs); UP(dnt, nt, sizeof(int) * npairs);
    AL(di, ob); AL(dd, ob);
    ARENA_FLUSH(m);
    int rc = orbm_knn2_batch_async(m, dq.as<uint8_t>(), q_stride, dnq.as<int32_t>(), dt.as<uint8_t>(), t_stride, dnt.as<int32_t>(), npairs, t_stride,
                                   di.as<int32_t>(), dd.as<int32_t>());
    if (rc) return rc;
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    memcpy(idx2, di.host(), ob); memcpy(dist2, dd.host(), ob);
    return ORBM_OK;
}

struct RotHist {                                             // rotation-consistency histogram (e.g. ORBmatcher.cc:459-466)
    std::vector<int> bins[ORBM_HISTO_LENGTH];
    void add(float a1, float a2, float factor, int idx) {
        float rot = a1 - a2;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == ORBM_HISTO_LENGTH) bin = 0;
        if (bin >= 0 && bin < ORBM_HISTO_LENGTH) bins[bin].push_back(idx);
    }
    void maxima(int* ind) const {
        int sz[ORBM_HISTO_LENGTH];
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) sz[i] = (int)bins[i].size();
        orbm_three_maxima(sz, ORBM_HISTO_LENGTH, ind);
    }
};

// runs k_window for nq windows against frame f; candidate lists come back in host vectors
int window_pass(orbm* m, const orbm_frame_t* f, int nq, const float* qx, const float* qy, const float* qr,
                const int32_t* minl, const int32_t* maxl, const float* qur, const float* qer, const uint8_t* qdesc,
                int& cap, std::vector<int>& cnt, std::vector<int>& idx, std::vector<int>& dist, bool retry = false, const orbm_dframe* df = nullptr) {
    // `cap` in: capacity of a window on the device; out: row stride of idx / dist (= the longest candidate list, >= 1)
    cnt.assign(nq, 0);
    if (nq == 0 || f->n == 0) return ORBM_OK;
    if (const char* e = retry ? nullptr : ab_env("ORBM_WINDOW_CAP")) cap = std::max(1, std::min(cap, atoi(e)));   // test knob: a small first capacity forces the retry below
    MHIPCHK(hipSetDevice(m->device));
    DevBuf dk, dd, du, dgs, dgi, dqx, dqy, dqr, dmin, dmax, dqu, dqe, dqd, dcnt, didx, ddist, dovf, dpack;
    arena_reset(m);
    if (!df) {                                                  // host frame: it travels with every call (a resident frame does not)
        UP(dk, f->kps, sizeof(KpIn) * f->n); UP(dd, f->desc, (size_t)32 * f->n);
        if (f->uright) UP(du, f->uright, sizeof(float) * f->n);
        UP(dgs, f->grid_start, sizeof(int) * (ORBM_GRID_COLS * ORBM_GRID_ROWS + 1));
        UP(dgi, f->grid_idx, sizeof(int) * f->n);
    }
    UP(dqx, qx, sizeof(float) * nq); UP(dqy, qy, sizeof(float) * nq); UP(dqr, qr, sizeof(float) * nq);
    UP(dmin, minl, sizeof(int) * nq); UP(dmax, maxl, sizeof(int) * nq);
    if (qur) UP(dqu, qur, sizeof(float) * nq);
    if (qer) UP(dqe, qer, sizeof(float) * nq);
    UP(dqd, qdesc, (size_t)32 * nq);
    AL(dcnt, sizeof(int) * nq); AL(dovf, sizeof(int));
    AL_BIG(didx, sizeof(int) * (size_t)nq * cap); AL_BIG(ddist, sizeof(int) * (size_t)nq * cap);   // [nq][cap]: only the used columns come back
    AL_BIG(dpack, 2 * sizeof(int) * (size_t)nq * cap);       // packed copy of the used columns (reserved now: the arena must not grow after the launch)
    MHIPCHK(hipMemsetAsync(dovf.ptr(), 0, sizeof(int), m->stream));
    MHIPCHK(rec_time(m, m->e0));
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_window, dim3((nq + 3) / 4), dim3(256), 0, m->stream, df ? df->dKps : dk.as<KpIn>(), df ? df->dDesc : dd.as<uint8_t>(),
                       df ? (f->uright ? df->dUr : nullptr) : (f->uright ? du.as<float>() : nullptr), df ? df->dGs : dgs.as<int>(), df ? df->dGi : dgi.as<int>(),
                       f->min_x, f->min_y, f->inv_w, f->inv_h,
                       nq, dqx.as<float>(), dqy.as<float>(), dqr.as<float>(), dmin.as<int>(), dmax.as<int>(),
                       qur ? dqu.as<float>() : nullptr, qer ? dqe.as<float>() : nullptr, dqd.as<uint8_t>(), cap,
                       dcnt.as<int>(), didx.as<int>(), ddist.as<int>(), dovf.as<int>());
    MHIPCHK(rec_time(m, m->e1));
    m->timed = true;
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    int ovf = 0;
    memcpy(&ovf, dovf.host(), sizeof(int));
    memcpy(cnt.data(), dcnt.host(), sizeof(int) * nq);
    const int devCap = cap;
    int maxc = 0;
    for (int i = 0; i < nq; ++i) maxc = std::max(maxc, std::min(cnt[i], devCap));
    const size_t pn = (size_t)nq * std::max(maxc, 1);
    idx.resize(pn); dist.resize(pn);
    if (maxc > 0) {                                         // pack the used columns on the device: one contiguous copy, stride maxc
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((pn + 255) / 256)), dim3(256), 0, m->stream, didx.as<int>(), ddist.as<int>(), nq, devCap, maxc,
                           dpack.as<int>(), dpack.as<int>() + pn);
        MHIPCHK(hipGetLastError());
        MHIPCHK(hipMemcpyAsync(m->arPin + dpack.off, m->arDev + dpack.off, 2 * sizeof(int) * pn, hipMemcpyDeviceToHost, m->stream));
        MHIPCHK(hipStreamSynchronize(m->stream));
        memcpy(idx.data(), dpack.host(), sizeof(int) * pn);
        memcpy(dist.data(), (const int*)dpack.host() + pn, sizeof(int) * pn);
    }
    if (ovf) {
        // Frame::GetFeaturesInArea is unbounded (Frame.cc:784-871): a window that holds more than the default capacity (dense
        // frames, the 100-px initialisation window) is simply run again with room for every keypoint of the frame
        if (devCap < f->n) {
            cap = f->n;
            return window_pass(m, f, nq, qx, qy, qr, minl, maxl, qur, qer, qdesc, cap, cnt, idx, dist, true, df);
        }
        set_merr("a search window returned more than %d candidates", devCap);
        return ORBM_E_CAPACITY;
    }
    // Everything below indexes host arrays with what the device reported: counts beyond the row stride or keypoint indices
    // beyond the frame must stop here, not in a caller's replay loop (round 2's ORBM_WINDOW_CAP=4 run died in such a loop,
    // profiles/NOTES.md "18:59 segfault").
    for (int i = 0; i < nq; ++i)
        if (cnt[i] < 0 || cnt[i] > devCap) { set_merr("window %d reports %d candidates for a capacity of %d without the overflow flag", i, cnt[i], devCap); return ORBM_E_HIP; }
    for (int i = 0; i < nq; ++i)
        for (int c = 0; c < cnt[i]; ++c)
            if ((unsigned)idx[(size_t)i * maxc + c] >= (unsigned)f->n) { set_merr("window %d candidate %d: keypoint index %d outside the frame (%d keypoints)", i, c, idx[(size_t)i * maxc + c], f->n); return ORBM_E_HIP; }
    cap = std::max(maxc, 1);
    return ORBM_OK;
}

// k_window_topk against a resident frame: per window the candidate count and the WT_K smallest (distance, position) keys.
// One upload of the queries, one kernel, one download.
int window_topk_pass(orbm* m, const orbm_dframe* df, bool stereo_gate, int nq, const float* qx, const float* qy, const float* qr,
                     const int32_t* minl, const int32_t* maxl, const float* qur, const float* qer, const uint8_t* qdesc,
                     const int** cnt, const unsigned int** keys) {
    // results stay in the arena's pinned mirror (valid until the handle's next call): *cnt [nq], *keys [nq][WT_K]
    *cnt = nullptr; *keys = nullptr;
    if (nq == 0 || df->n == 0) return ORBM_OK;
    MHIPCHK(hipSetDevice(m->device));
    DevBuf dqx, dqy, dqr, dmin, dmax, dqu, dqe, dqd, dcnt, dkeys;
    arena_reset(m);
    UP(dqx, qx, sizeof(float) * nq); UP(dqy, qy, sizeof(float) * nq); UP(dqr, qr, sizeof(float) * nq);
    UP(dmin, minl, sizeof(int) * nq); UP(dmax, maxl, sizeof(int) * nq);
    if (qur) UP(dqu, qur, sizeof(float) * nq);
    if (qer) UP(dqe, qer, sizeof(float) * nq);
    UP(dqd, qdesc, (size_t)32 * nq);
    AL(dcnt, sizeof(int) * nq); AL(dkeys, sizeof(unsigned int) * (size_t)nq * WT_K);
    static const bool prof = getenv("ORBM_PROFILE") != nullptr;
    static const bool copies = ab_env("ORBM_TOPK_COPIES") != nullptr;     // A/B: staged copies instead of zero-copy
    auto T = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = prof ? T() : 0;
    if (prof) MHIPCHK(rec_time(m, m->e0));                       // a latency path: no event records unless someone is looking
    // Zero-copy: the queries sit in the arena's pinned host mirror, which the GPU can address; the kernel reads them over PCIe
    // and writes counts and keys straight back into it.  ~110 KB in all: two copy commands would cost more in latency (~10 us
    // each) than the link takes to move the bytes, and this is a latency path (Tracking waits for the result).
    auto hp = [&](DevBuf& b) { return (void*)(m->arPin + b.off); };
    if (copies) ARENA_FLUSH(m);
#define QP(buf, T_) (copies ? (buf).as<T_>() : (T_*)hp(buf))
    hipLaunchKernelGGL(k_window_topk, dim3((nq + 3) / 4), dim3(256), 0, m->stream, df->dKps, df->dDesc, stereo_gate ? df->dUr : nullptr, df->dGs, df->dGi,
                       df->min_x, df->min_y, df->inv_w, df->inv_h, nq, QP(dqx, float), QP(dqy, float), QP(dqr, float), QP(dmin, int), QP(dmax, int),
                       qur ? QP(dqu, float) : nullptr, qer ? QP(dqe, float) : nullptr, QP(dqd, uint8_t), QP(dcnt, int), QP(dkeys, unsigned int));
#undef QP
    if (prof) { MHIPCHK(rec_time(m, m->e1)); m->timed = true; }
    MHIPCHK(hipGetLastError());
    if (copies) MHIPCHK(hipMemcpyAsync(m->arPin + m->arOutLo, m->arDev + m->arOutLo, m->arOutHi - m->arOutLo, hipMemcpyDeviceToHost, m->stream));
    const double t1 = prof ? T() : 0;
    MHIPCHK(hipStreamSynchronize(m->stream));
    if (prof) fprintf(stderr, "orbm profile:   enqueue %.1f us, wait %.1f us (%s)\n", t1 - t0, T() - t1, copies ? "staged copies" : "zero-copy");
    *cnt = (const int*)dcnt.host(); *keys = (const unsigned int*)dkeys.host();
    {   // the replays index the frame's arrays with the low 20 bits of every returned key: check them once, branch-free
        const int* c_ = *cnt; const unsigned int* k_ = *keys;
        unsigned bad = 0;
        for (int i = 0; i < nq; ++i) {
            const int nc = std::min(std::max(c_[i], 0), (int)WT_K);
            bad |= (unsigned)(c_[i] < 0);
            for (int c = 0; c < nc; ++c) bad |= (unsigned)((k_[(size_t)i * WT_K + c] & 0xFFFFFu) >= (unsigned)df->n);
        }
        if (bad) { set_merr("top-K window pass returned a keypoint index outside the resident frame (%d keypoints)", df->n); *cnt = nullptr; *keys = nullptr; return ORBM_E_HIP; }
    }
    return ORBM_OK;
}

// merge-join of two FeatureVector CSRs (ORBmatcher.cc:343-518 / 1448-1595) -> jobs + GPU distances
struct JoinJobs { std::vector<int> q, l2, len, off, node_b; int total = 0; };

int bucket_pass(orbm* m, const uint8_t* d1, int n1, const uint8_t* d2, int n2, const int32_t* idx2, int nidx2,
                const JoinJobs& J, std::vector<int>& dist) {
    dist.assign(J.total, 0);
    if (J.total == 0) return ORBM_OK;
    MHIPCHK(hipSetDevice(m->device));
    DevBuf b1, b2, bi, bq, bl, bo, bout;
    arena_reset(m);
    UP(b1, d1, (size_t)32 * n1); UP(b2, d2, (size_t)32 * n2); UP(bi, idx2, sizeof(int) * nidx2);
    const int nj = (int)J.q.size();
    UP(bq, J.q.data(), sizeof(int) * nj); UP(bl, J.l2.data(), sizeof(int) * nj); UP(bo, J.off.data(), sizeof(int) * nj);
    AL(bout, sizeof(int) * (size_t)J.total);
    MHIPCHK(rec_time(m, m->e0));
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_pairdist, dim3((J.total + 255) / 256), dim3(256), 0, m->stream, b1.as<uint8_t>(), b2.as<uint8_t>(),
                       bi.as<int>(), nj, bq.as<int>(), bl.as<int>(), bo.as<int>(), J.total, bout.as<int>());
    MHIPCHK(rec_time(m, m->e1));
    m->timed = true;
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    memcpy(dist.data(), bout.host(), sizeof(int) * (size_t)J.total);
    return ORBM_OK;
}

}  // namespace

extern "C" {

int orbm_grid_build(orbm_t* m, const orbm_kp_t* kps, int n, float min_x, float min_y, float inv_w, float inv_h,
                    int32_t* grid_start, int32_t* grid_idx) {
    if (!m || !grid_start || !grid_idx || n < 0 || (n > 0 && !kps)) return ORBM_E_INVALID;
    if (n > 65535) { set_merr("grid build supports at most 65535 keypoints"); return ORBM_E_CAPACITY; }
    MHIPCHK(hipSetDevice(m->device));
    int n2 = 64; while (n2 < n) n2 <<= 1;
    if ((size_t)n2 * 4 > 150 * 1024) { set_merr("too many keypoints for the LDS sort"); return ORBM_E_CAPACITY; }
    DevBuf dk, dgs, dgi, dpl;
    arena_reset(m);
    UP(dk, kps, sizeof(KpIn) * n);
    AL(dgs, sizeof(int) * (ORBM_GRID_COLS * ORBM_GRID_ROWS + 1)); AL(dgi, sizeof(int) * (n + 1)); AL(dpl, sizeof(int));
    ARENA_FLUSH(m);
    if (n <= GRID_CS_MAX)
        hipLaunchKernelGGL(k_grid_build_cs, dim3(1), dim3(256), grid_cs_lds(n), m->stream, dk.as<KpIn>(), n, min_x, min_y, inv_w, inv_h,
                           dgs.as<int>(), dgi.as<int>(), dpl.as<int>());
    else {
        MHIPCHK(hipFuncSetAttribute((const void*)k_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 4));
        hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(256), (size_t)n2 * 4, m->stream, dk.as<KpIn>(), n, n2, min_x, min_y, inv_w, inv_h,
                           dgs.as<int>(), dgi.as<int>(), dpl.as<int>());
    }
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    int placed = 0;
    memcpy(&placed, dpl.host(), sizeof(int));
    memcpy(grid_start, dgs.host(), sizeof(int) * (ORBM_GRID_COLS * ORBM_GRID_ROWS + 1));
    if (placed) memcpy(grid_idx, dgi.host(), sizeof(int) * placed);
    return placed;
}

int orbm_window_candidates(orbm_t* m, const orbm_frame_t* f, int nq, const float* qx, const float* qy, const float* qr,
                           const int32_t* min_level, const int32_t* max_level, const float* q_ur, const float* q_er_max,
                           const uint8_t* qdesc, int cap, int32_t* out_cnt, int32_t* out_idx, int32_t* out_dist) {
    if (!m || !f || nq < 0 || cap < 1) return ORBM_E_INVALID;
    std::vector<int> cnt, idx, dist;
    int stride = cap;                                        // in: window capacity; out: row stride of the packed lists
    int rc = window_pass(m, f, nq, qx, qy, qr, min_level, max_level, q_ur, q_er_max, qdesc, stride, cnt, idx, dist);
    if (rc && rc != ORBM_E_CAPACITY) return rc;
    for (int i = 0; i < nq; ++i) {
        out_cnt[i] = cnt[i];
        const int c = std::min(cnt[i], cap);
        if (c > 0 && !idx.empty()) {
            memcpy(out_idx + (size_t)i * cap, &idx[(size_t)i * stride], sizeof(int) * c);
            memcpy(out_dist + (size_t)i * cap, &dist[(size_t)i * stride], sizeof(int) * c);
        }
    }
    return rc;
}

// Candidate lists as the claim replays see them: either the full [queries][stride] lists of window_pass, or the K best
// (distance, position) keys of window_topk_pass, which are a prefix of the list in exactly the order that matters to a
// `dist < bestDist` scan (see k_window_topk).  trunc(i): the list of query i was cut.
struct CandLists {
    const int* cnt = nullptr; const int* idx = nullptr; const int* dist = nullptr; int stride = 0;
    const unsigned int* keys = nullptr;                                 // dist << 20 | keypoint index, in (distance, visiting order) rank
    int count(int i) const { return keys ? std::min(cnt[i], (int)WT_K) : std::min(cnt[i], stride); }   // (never beyond a row: window_pass has checked)
    bool trunc(int i) const { return keys && cnt[i] > WT_K; }
    int index(int i, int c) const { return keys ? (int)(keys[(size_t)i * WT_K + c] & 0xFFFFFu) : idx[(size_t)i * stride + c]; }
    int distance(int i, int c) const { return keys ? (int)(keys[(size_t)i * WT_K + c] >> 20) : dist[(size_t)i * stride + c]; }
};

// M4 body.  cur: the frame's view (host arrays, or the resident frame's host-side copy of what the replay reads); df: resident frame or NULL
static int search_by_projection_frame_impl(orbm_t* m, const orbm_frame_t* cur, const orbm_dframe* df, const uint8_t* cur_blocked, const float* sf,
                                           int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                           const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                           float th, int bForward, int bBackward, float mbf, int check_ori, int32_t* match) {
    // windows exactly as ORBmatcher.cc:2543-2549; invalid queries get an empty window (r < 0)
    std::vector<float> qr(nq), qur(nq), qer(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; qur[i] = 0; qer[i] = -1.f; continue; }
        const int o = octave[i];
        const float radius = th * sf[o];
        qr[i] = radius;
        if (bForward) { minl[i] = o; maxl[i] = -1; }
        else if (bBackward) { minl[i] = 0; maxl[i] = o; }
        else { minl[i] = o - 1; maxl[i] = o + 1; }
        qur[i] = u[i] - mbf * invzc[i];                                    // :2571
        qer[i] = cur->uright ? radius : -1.f;
    }
    // sequential replay of the claims (ORBmatcher.cc:2553-2612); false = a cut list ran out of unblocked candidates
    int nmatches = 0;
    auto replay = [&](const CandLists& Lc) -> bool {
        nmatches = 0;
        RotHist rh;
        const float factor = ORBM_HISTO_LENGTH / 360.0f;
        std::vector<uint8_t> blocked(cur_blocked, cur_blocked + cur->n);
        for (int i = 0; i < cur->n; ++i) match[i] = -1;
        for (int i = 0; i < nq; ++i) {
            const int nc = valid[i] ? Lc.count(i) : 0;
            if (nc == 0) continue;
            int bestDist = 256, bestIdx2 = -1;
            for (int c = 0; c < nc; ++c) {
                const int i2 = Lc.index(i, c);
                if (blocked[i2]) continue;
                const int d = Lc.distance(i, c);
                if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
            }
            if (bestIdx2 < 0 && Lc.trunc(i)) return false;
            if (bestDist <= ORBM_TH_HIGH) {
                match[bestIdx2] = i;
                if (mp_obs[i]) blocked[bestIdx2] = 1;
                nmatches++;
                if (check_ori) rh.add(angle[i], cur->kps[bestIdx2].angle, factor, bestIdx2);
            }
        }
        if (check_ori) {
            int ind[3];
            rh.maxima(ind);
            for (int b = 0; b < ORBM_HISTO_LENGTH; ++b)
                if (b != ind[0] && b != ind[1] && b != ind[2])
                    for (int k : rh.bins[b]) { match[k] = ORBM_MATCH_PRUNED; nmatches--; }
        }
        return true;
    };
    std::vector<int> cnt, idx, dist;
    if (df) {
        const int* kc = nullptr; const unsigned int* kk = nullptr;
        static const bool prof = getenv("ORBM_PROFILE") != nullptr;
        auto T = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = prof ? T() : 0;
        int rc = window_topk_pass(m, df, cur->uright != nullptr, nq, u, v, qr.data(), minl.data(), maxl.data(), qur.data(), qer.data(), qdesc, &kc, &kk);
        if (rc) return rc;
        const double t1 = prof ? T() : 0;
        CandLists Lk; Lk.cnt = kc; Lk.keys = kk;
        const bool done = !kc || replay(Lk);
        if (!kc) { for (int i = 0; i < cur->n; ++i) match[i] = -1; nmatches = 0; }
        if (prof) fprintf(stderr, "orbm profile: top-K pass %.1f us, replay %.1f us%s\n", t1 - t0, T() - t1, done ? "" : " (fallback)");
        if (done) return nmatches;
    }
    int cap = std::max(1, std::min(cur->n, 2048));      // becomes the row stride of idx / dist after window_pass
    int rc = window_pass(m, cur, nq, u, v, qr.data(), minl.data(), maxl.data(), qur.data(), qer.data(), qdesc, cap, cnt, idx, dist, false, df);
    if (rc) return rc;
    CandLists Lf; Lf.cnt = cnt.data(); Lf.idx = idx.data(); Lf.dist = dist.data(); Lf.stride = cap;
    replay(Lf);
    return nmatches;
}

int orbm_search_by_projection_frame(orbm_t* m, const orbm_frame_t* cur, const uint8_t* cur_blocked, const float* sf,
                                    int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                    const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                    float th, int bForward, int bBackward, float mbf, int check_ori, int32_t* match) {
    if (!m || !cur || nq < 0) return ORBM_E_INVALID;
    return search_by_projection_frame_impl(m, cur, nullptr, cur_blocked, sf, nq, valid, u, v, invzc, octave, angle, qdesc, mp_obs, th, bForward, bBackward, mbf,
                                           check_ori, match);
}

static int search_by_projection_points_impl(orbm_t* m, const orbm_frame_t* f, const orbm_dframe* df, const uint8_t* blocked_in, const float* sf,
                                            int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                            const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                            float th, float nnratio, int32_t* match) {
    const bool bFactor = th != 1.0;
    std::vector<float> qr(nq), qer(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!in_view[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; qer[i] = -1.f; continue; }
        float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;                       // RadiusByViewingCos (:242-249)
        if (bFactor) r *= th;
        qr[i] = r * sf[level[i]];
        minl[i] = level[i] - 1; maxl[i] = level[i];
        qer[i] = f->uright ? r * sf[level[i]] : -1.f;                      // :107-117
    }
    int nmatches = 0;
    auto replay = [&](const CandLists& Lc) -> bool {
        nmatches = 0;
        std::vector<uint8_t> blocked(blocked_in, blocked_in + f->n);
        for (int i = 0; i < f->n; ++i) match[i] = -1;
        for (int iMP = 0; iMP < nq; ++iMP) {
            const int nc = in_view[iMP] ? Lc.count(iMP) : 0;
            if (nc == 0) continue;
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, seen = 0;
            for (int c = 0; c < nc; ++c) {
                const int k = Lc.index(iMP, c);
                if (blocked[k]) continue;
                ++seen;
                const int d = Lc.distance(iMP, c);
                if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = f->kps[k].octave; bestIdx = k; }
                else if (d < bestDist2) { bestLevel2 = f->kps[k].octave; bestDist2 = d; }
            }
            if (seen < 2 && Lc.trunc(iMP)) return false;                    // best AND second must come from the unblocked candidates
            if (bestDist <= ORBM_TH_HIGH) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                    match[bestIdx] = iMP;
                    if (mp_obs[iMP]) blocked[bestIdx] = 1;
                    nmatches++;
                }
            }
        }
        return true;
    };
    std::vector<int> cnt, idx, dist;
    if (df) {
        const int* kc = nullptr; const unsigned int* kk = nullptr;
        int rc = window_topk_pass(m, df, f->uright != nullptr, nq, px, py, qr.data(), minl.data(), maxl.data(), pxr, qer.data(), qdesc, &kc, &kk);
        if (rc) return rc;
        if (!kc) { for (int i = 0; i < f->n; ++i) match[i] = -1; return 0; }
        CandLists Lk; Lk.cnt = kc; Lk.keys = kk;
        if (replay(Lk)) return nmatches;
    }
    int cap = std::max(1, std::min(f->n, 2048));      // becomes the row stride of idx / dist after window_pass
    int rc = window_pass(m, f, nq, px, py, qr.data(), minl.data(), maxl.data(), pxr, qer.data(), qdesc, cap, cnt, idx, dist, false, df);
    if (rc) return rc;
    CandLists Lf; Lf.cnt = cnt.data(); Lf.idx = idx.data(); Lf.dist = dist.data(); Lf.stride = cap;
    replay(Lf);
    return nmatches;
}

int orbm_search_by_projection_points(orbm_t* m, const orbm_frame_t* f, const uint8_t* blocked_in, const float* sf,
                                     int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                     const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                     float th, float nnratio, int32_t* match) {
    if (!m || !f || nq < 0) return ORBM_E_INVALID;
    return search_by_projection_points_impl(m, f, nullptr, blocked_in, sf, nq, in_view, px, py, pxr, view_cos, level, qdesc, mp_obs, th, nnratio, match);
}

// ---- frames resident in HBM: what Tracking's 2-4 searches per frame work on (Tracking.cc:3002-3211, 3867-3891) ----
void orbm_frame_destroy(orbm_dframe_t* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->block) (void)hipFree(f->block);
    delete f;
}

int orbm_frame_create(orbm_t* m, int space, int n, const orbm_kp_t* kps, const uint8_t* desc, const float* uright,
                      float min_x, float min_y, float inv_w, float inv_h, orbm_dframe_t** out) {
    if (!m || !out || n < 0 || (n > 0 && (!kps || !desc))) return ORBM_E_INVALID;
    *out = nullptr;
    if (n > 65535) { set_merr("a resident frame holds at most 65535 keypoints"); return ORBM_E_CAPACITY; }
    int n2 = 64; while (n2 < n) n2 <<= 1;
    if ((size_t)n2 * 4 > 150 * 1024) { set_merr("too many keypoints for the LDS sort"); return ORBM_E_CAPACITY; }
    MHIPCHK(hipSetDevice(m->device));
    orbm_dframe* f = new orbm_dframe;
    f->device = m->device; f->n = n; f->min_x = min_x; f->min_y = min_y; f->inv_w = inv_w; f->inv_h = inv_h;
    const size_t ncells = ORBM_GRID_COLS * ORBM_GRID_ROWS + 1;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t bKps = space == ORBM_HOST ? al(sizeof(KpIn) * (size_t)std::max(n, 1)) : 0, bDesc = space == ORBM_HOST ? al((size_t)32 * std::max(n, 1)) : 0,
                 bUr = space == ORBM_HOST && uright ? al(sizeof(float) * (size_t)std::max(n, 1)) : 0, bGs = al(sizeof(int) * ncells), bGi = al(sizeof(int) * (size_t)(n + 1)) + 256;
    if (hipMalloc((void**)&f->block, bKps + bDesc + bUr + bGs + bGi) != hipSuccess) { set_merr("hipMalloc failed"); delete f; return ORBM_E_HIP; }
    uint8_t* p = f->block;
    f->hkps.resize(std::max(n, 1));
    hipError_t e = hipSuccess;
    if (space == ORBM_HOST) {
        f->dKps = (const KpIn*)p; p += bKps; f->dDesc = p; p += bDesc;
        if (uright) { f->dUr = (const float*)p; p += bUr; }
        if (n > 0) {
            memcpy(f->hkps.data(), kps, sizeof(orbm_kp_t) * n);
            e = hipMemcpyAsync((void*)f->dKps, kps, sizeof(KpIn) * n, hipMemcpyHostToDevice, m->stream);
            if (e == hipSuccess) e = hipMemcpyAsync((void*)f->dDesc, desc, (size_t)32 * n, hipMemcpyHostToDevice, m->stream);
            if (e == hipSuccess && uright) e = hipMemcpyAsync((void*)f->dUr, uright, sizeof(float) * n, hipMemcpyHostToDevice, m->stream);
        }
    } else {                                                     // adopt the caller's device arrays (an extractor's result block): no copy of them
        f->dKps = (const KpIn*)kps; f->dDesc = desc; f->dUr = uright;
        if (n > 0) e = hipMemcpyAsync(f->hkps.data(), kps, sizeof(orbm_kp_t) * n, hipMemcpyDeviceToHost, m->stream);
    }
    f->dGs = (int*)p; p += bGs; f->dGi = (int*)p;
    int* dPlaced = f->dGi + (n + 1);
    if (e == hipSuccess && n > 0) {
        if (n <= GRID_CS_MAX)
            hipLaunchKernelGGL(k_grid_build_cs, dim3(1), dim3(256), grid_cs_lds(n), m->stream, f->dKps, n, min_x, min_y, inv_w, inv_h, f->dGs, f->dGi, dPlaced);
        else {
            if (n2 * 4 > 48 * 1024) e = hipFuncSetAttribute((const void*)k_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 4);
            if (e == hipSuccess) hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(256), (size_t)n2 * 4, m->stream, f->dKps, n, n2, min_x, min_y, inv_w, inv_h, f->dGs, f->dGi, dPlaced);
        }
        if (e == hipSuccess) e = hipGetLastError();
    } else if (e == hipSuccess) e = hipMemsetAsync(f->dGs, 0, sizeof(int) * ncells, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) { set_merr("resident frame setup failed: %s", hipGetErrorString(e)); orbm_frame_destroy(f); return ORBM_E_HIP; }
    *out = f;
    return ORBM_OK;
}

int orbm_frame_size(const orbm_dframe_t* f) { return f ? f->n : ORBM_E_INVALID; }

static orbm_frame_t resident_view(const orbm_dframe* f, bool stereo_gate) {
    orbm_frame_t v;
    v.n = f->n; v.kps = f->hkps.data(); v.desc = nullptr; v.uright = stereo_gate && f->dUr ? (const float*)f->hkps.data() /* non-NULL marker: only tested, never read */ : nullptr;
    v.min_x = f->min_x; v.min_y = f->min_y; v.inv_w = f->inv_w; v.inv_h = f->inv_h; v.grid_start = nullptr; v.grid_idx = nullptr;
    return v;
}

int orbm_search_by_projection_frame_resident(orbm_t* m, const orbm_dframe_t* cur, const uint8_t* cur_blocked, const float* sf,
                                             int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                             const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                             float th, int bForward, int bBackward, float mbf, int check_ori, int32_t* match) {
    if (!m || !cur || nq < 0 || cur->device != m->device) return ORBM_E_INVALID;
    const orbm_frame_t view = resident_view(cur, true);
    return search_by_projection_frame_impl(m, &view, cur, cur_blocked, sf, nq, valid, u, v, invzc, octave, angle, qdesc, mp_obs, th, bForward, bBackward, mbf,
                                           check_ori, match);
}

int orbm_search_by_projection_points_resident(orbm_t* m, const orbm_dframe_t* f, const uint8_t* blocked_in, const float* sf,
                                              int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                              const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                              float th, float nnratio, int32_t* match) {
    if (!m || !f || nq < 0 || f->device != m->device) return ORBM_E_INVALID;
    const orbm_frame_t view = resident_view(f, true);
    return search_by_projection_points_impl(m, &view, f, blocked_in, sf, nq, in_view, px, py, pxr, view_cos, level, qdesc, mp_obs, th, nnratio, match);
}

int orbm_search_for_initialization(orbm_t* m, const orbm_frame_t* F1, const orbm_frame_t* F2, float* prev, int windowSize,
                                   float nnratio, int check_ori, int32_t* vnMatches12) {
    if (!m || !F1 || !F2) return ORBM_E_INVALID;
    const int n1 = F1->n;
    std::vector<float> qx(n1), qy(n1), qr(n1);
    std::vector<int> minl(n1), maxl(n1);
    for (int i = 0; i < n1; ++i) {
        const int l1 = F1->kps[i].octave;
        qx[i] = prev[2 * i]; qy[i] = prev[2 * i + 1];
        if (l1 > 0) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; }           // :825
        else { qr[i] = (float)windowSize; minl[i] = l1; maxl[i] = l1; }
    }
    int cap = std::max(1, std::min(F2->n, 4096));      // becomes the row stride of idx / dist after window_pass
    std::vector<int> cnt, idx, dist;
    int rc = window_pass(m, F2, n1, qx.data(), qy.data(), qr.data(), minl.data(), maxl.data(), nullptr, nullptr, F1->desc, cap, cnt, idx, dist);
    if (rc) return rc;
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) vnMatches12[i] = -1;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;
    std::vector<int> vMatchedDistance(F2->n, INT_MAX), vnMatches21(F2->n, -1);
    for (int i1 = 0; i1 < n1; ++i1) {
        if (F1->kps[i1].octave > 0 || cnt[i1] == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < cnt[i1]; ++c) {
            const int i2 = idx[(size_t)i1 * cap + c], d = dist[(size_t)i1 * cap + c];
            if (vMatchedDistance[i2] <= d) continue;
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= ORBM_TH_LOW && bestDist < (float)bestDist2 * nnratio) {
            if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
            vnMatches12[i1] = bestIdx2; vnMatches21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist;
            nmatches++;
            if (check_ori) rh.add(F1->kps[i1].angle, F2->kps[bestIdx2].angle, factor, i1);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int i1 : rh.bins[b]) if (vnMatches12[i1] >= 0) { vnMatches12[i1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = F2->kps[vnMatches12[i1]].x; prev[2 * i1 + 1] = F2->kps[vnMatches12[i1]].y; }
    return nmatches;
}

static void join_nodes(int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1, int nn2, const int32_t* nodes2,
                       const int32_t* start2, JoinJobs& J) {
    int a = 0, b = 0;                                                       // lower_bound jumps == this linear merge on sorted keys
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            const int len = start2[b + 1] - start2[b];
            for (int i = start1[a]; i < start1[a + 1]; ++i) {
                J.q.push_back(idx1[i]); J.l2.push_back(start2[b]); J.len.push_back(len); J.off.push_back(J.total); J.node_b.push_back(b);
                J.total += len;
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) ++a; else ++b;
    }
}

static int triangulation_impl(orbm_t* m, bool legacy, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                  int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                  int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                  int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                  const float* F12, float epx, float epy, const float* sf2, const float* sigma2_2,
                                  int bOnlyStereo, int bCoarse, int check_ori, int32_t* vMatches12,
                                  orbm_pair_gate_fn gate = nullptr, void* gate_user = nullptr) {
    if (!m || n1 < 0 || n2 < 0) return ORBM_E_INVALID;
    std::vector<uint8_t> vbMatched2(n2, 0);                                 // only maintained by the legacy overload (:1319)
    JoinJobs J;
    join_nodes(nn1, nodes1, start1, idx1, nn2, nodes2, start2, J);
    std::vector<int> dist;
    int rc = bucket_pass(m, desc1, n1, desc2, n2, idx2, start2[nn2], J, dist);
    if (rc) return rc;
    for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
    int nmatches = 0;
    RotHist rh;
    const float factor = legacy ? ORBM_HISTO_LENGTH / 360.0f : 1.0f / ORBM_HISTO_LENGTH;   // :1166 vs :1441 (sic)
    for (size_t j = 0; j < J.q.size(); ++j) {
        const int i1 = J.q[j];
        if (has_mp1[i1]) continue;
        const bool bStereo1 = uright1 && uright1[i1] >= 0;
        if (bOnlyStereo && !bStereo1) continue;
        const orbm_kp_t& kp1 = kps1[i1];
        int bestDist = ORBM_TH_LOW, bestIdx2 = -1;
        for (int c = 0; c < J.len[j]; ++c) {
            const int i2 = idx2[J.l2[j] + c];
            if (vbMatched2[i2] || has_mp2[i2]) continue;
            const bool bStereo2 = uright2 && uright2[i2] >= 0;
            if (bOnlyStereo && !bStereo2) continue;
            const int d = dist[J.off[j] + c];
            if (d > ORBM_TH_LOW || d > bestDist) continue;
            const orbm_kp_t& kp2 = kps2[i2];
            if (!bStereo1 && !bStereo2 && sf2) {                            // sf2 == NULL: pKF1->mpCamera2 set (:1517) / M12, which has no such gate
                const float distex = epx - kp2.x, distey = epy - kp2.y;
                if (distex * distex + distey * distey < 100 * sf2[kp2.octave]) continue;
            }
            bool epi = false;
            if (gate) epi = gate(gate_user, i1, i2) != 0;                   // the caller's camera model, called where the reference calls it (:1552 / :1729)
            else {   // Pinhole::epipolarConstrain_ (Pinhole.cpp:281-295)
                const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
                const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
                const float c2 = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
                const float num = a * kp2.x + b * kp2.y + c2;
                const float den = a * a + b * b;
                if (den != 0) { const float dsqr = num * num / den; epi = dsqr < 3.84 * sigma2_2[kp2.octave]; }
            }
            if (epi || bCoarse) { bestIdx2 = i2; bestDist = d; }
        }
        if (bestIdx2 >= 0) {
            vMatches12[i1] = bestIdx2;
            if (legacy) vbMatched2[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(kp1.angle, kps2[bestIdx2].angle, factor, i1);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int i : rh.bins[b]) { if (legacy) vbMatched2[vMatches12[i]] = 0; vMatches12[i] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_search_for_triangulation(orbm_t* m, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                  int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                  int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                  int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                  const float* F12, float epx, float epy, const float* sf2, const float* sigma2_2,
                                  int bOnlyStereo, int bCoarse, int check_ori, int32_t* vMatches12) {
    return triangulation_impl(m, false, n1, kps1, desc1, has_mp1, uright1, nn1, nodes1, start1, idx1, n2, kps2, desc2, has_mp2, uright2,
                              nn2, nodes2, start2, idx2, F12, epx, epy, sf2, sigma2_2, bOnlyStereo, bCoarse, check_ori, vMatches12);
}

int orbm_search_for_triangulation_legacy(orbm_t* m, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                  int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                  int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                  int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                  const float* F12, float epx, float epy, const float* sf2, const float* sigma2_2,
                                  int bOnlyStereo, int bCoarse, int check_ori, int32_t* vMatches12) {
    return triangulation_impl(m, true, n1, kps1, desc1, has_mp1, uright1, nn1, nodes1, start1, idx1, n2, kps2, desc2, has_mp2, uright2,
                              nn2, nodes2, start2, idx2, F12, epx, epy, sf2, sigma2_2, bOnlyStereo, bCoarse, check_ori, vMatches12);
}

int orbm_search_for_triangulation_gated(orbm_t* m, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                                        int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                                        int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2,
                                        int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                                        orbm_pair_gate_fn gate, void* user, int check_ori, int32_t* vMatches12) {
    if (!gate) { set_merr("orbm_search_for_triangulation_gated needs a gate callback"); return ORBM_E_INVALID; }
    return triangulation_impl(m, false, n1, kps1, desc1, has_mp1, nullptr, nn1, nodes1, start1, idx1, n2, kps2, desc2, has_mp2, nullptr,
                              nn2, nodes2, start2, idx2, nullptr, 0.f, 0.f, nullptr, nullptr, 0, 0, check_ori, vMatches12, gate, user);
}

int orbm_search_by_projection_kf(orbm_t* m, const orbm_frame_t* cur, const uint8_t* blocked_in, const float* sf,
                                 int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                 const float* angle, const uint8_t* qdesc, float th, int ORBdist, int check_ori, int32_t* match) {
    if (!m || !cur || nq < 0) return ORBM_E_INVALID;
    std::vector<float> qr(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; continue; }
        qr[i] = th * sf[level[i]]; minl[i] = level[i] - 1; maxl[i] = level[i] + 1;      // :2775-2778
    }
    int cap = std::max(1, std::min(cur->n, 2048));      // becomes the row stride of idx / dist after window_pass
    std::vector<int> cnt, idx, dist;
    orbm_frame_t f = *cur; f.uright = nullptr;                              // this overload has no stereo gate
    int rc = window_pass(m, &f, nq, u, v, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, cap, cnt, idx, dist);
    if (rc) return rc;
    int nmatches = 0;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;                        // :2736
    std::vector<uint8_t> taken(blocked_in, blocked_in + cur->n);
    for (int i = 0; i < cur->n; ++i) match[i] = -1;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i] || cnt[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < cnt[i]; ++c) {
            const int i2 = idx[(size_t)i * cap + c];
            if (taken[i2]) continue;                                        // :2793
            const int d = dist[(size_t)i * cap + c];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            match[bestIdx2] = i; taken[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], cur->kps[bestIdx2].angle, factor, bestIdx2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int k : rh.bins[b]) { match[k] = ORBM_MATCH_PRUNED; nmatches--; }
    }
    return nmatches;
}

int orbm_search_by_projection_sim3(orbm_t* m, const orbm_frame_t* kf, const uint8_t* matched_in, const float* sf,
                                   int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                   const uint8_t* qdesc, int th, float ratioHamming, int32_t* match) {
    if (!m || !kf || nq < 0) return ORBM_E_INVALID;
    std::vector<float> qr(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; continue; }
        qr[i] = th * sf[level[i]]; minl[i] = level[i] - 1; maxl[i] = level[i];          // :600-601, :625-627
    }
    int cap = std::max(1, std::min(kf->n, 2048));      // becomes the row stride of idx / dist after window_pass
    std::vector<int> cnt, idx, dist;
    orbm_frame_t f = *kf; f.uright = nullptr;
    int rc = window_pass(m, &f, nq, u, v, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, cap, cnt, idx, dist);
    if (rc) return rc;
    int nmatches = 0;
    std::vector<uint8_t> matched(matched_in, matched_in + kf->n);
    for (int i = 0; i < kf->n; ++i) match[i] = -1;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i] || cnt[i] == 0) continue;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < cnt[i]; ++c) {
            const int k = idx[(size_t)i * cap + c];
            if (matched[k]) continue;                                        // :621-622
            const int d = dist[(size_t)i * cap + c];
            if (d < bestDist) { bestDist = d; bestIdx = k; }
        }
        if (bestDist <= ORBM_TH_LOW * ratioHamming) { match[bestIdx] = i; matched[bestIdx] = 1; nmatches++; }
    }
    return nmatches;
}

int orbm_fuse(orbm_t* m, const orbm_frame_t* kf, const float* sf, const float* inv_sigma2,
              int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const int32_t* level,
              const uint8_t* qdesc, float th, int chi2_gate, int32_t* best_idx) {
    if (!m || !kf || nq < 0) return ORBM_E_INVALID;
    std::vector<float> qr(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; continue; }
        qr[i] = th * sf[level[i]]; minl[i] = level[i] - 1; maxl[i] = level[i];
    }
    int cap = std::max(1, std::min(kf->n, 2048));      // becomes the row stride of idx / dist after window_pass
    std::vector<int> cnt, idx, dist;
    orbm_frame_t f = *kf; f.uright = nullptr;                               // the chi2 gate below is not a window gate
    int rc = window_pass(m, &f, nq, u, v, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, cap, cnt, idx, dist);
    if (rc) return rc;
    int nFused = 0;
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1;
        if (!valid[i] || cnt[i] == 0) continue;
        int bestDist = chi2_gate ? 256 : INT_MAX, bestIdx = -1;
        for (int c = 0; c < cnt[i]; ++c) {
            const int k = idx[(size_t)i * cap + c];
            if (chi2_gate) {                                                 // :1925-1949
                const orbm_kp_t& kp = kf->kps[k];
                if (kf->uright && kf->uright[k] >= 0) {
                    const float ex = u[i] - kp.x, ey = v[i] - kp.y, er = ur[i] - kf->uright[k];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kp.octave] > 7.8) continue;
                } else {
                    const float ex = u[i] - kp.x, ey = v[i] - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kp.octave] > 5.99) continue;
                }
            }
            const int d = dist[(size_t)i * cap + c];
            if (d < bestDist) { bestDist = d; bestIdx = k; }
        }
        if (bestDist <= ORBM_TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    return nFused;
}

static int sim3_one_way(orbm* m, const orbm_frame_t* src, const orbm_frame_t* dst, const float* sf_dst, const uint8_t* valid,
                        const float* u, const float* v, const int32_t* level, const uint8_t* qdesc, float th, std::vector<int>& vnMatch) {
    const int nq = src->n;
    vnMatch.assign(nq, -1);
    std::vector<float> qr(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; continue; }
        qr[i] = th * sf_dst[level[i]]; minl[i] = level[i] - 1; maxl[i] = level[i];
    }
    int cap = std::max(1, std::min(dst->n, 2048));      // becomes the row stride of idx / dist after window_pass
    std::vector<int> cnt, idx, dist;
    orbm_frame_t f = *dst; f.uright = nullptr;
    int rc = window_pass(m, &f, nq, u, v, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, cap, cnt, idx, dist);
    if (rc) return rc;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int c = 0; c < cnt[i]; ++c) {
            const int d = dist[(size_t)i * cap + c];
            if (d < bestDist) { bestDist = d; bestIdx = idx[(size_t)i * cap + c]; }
        }
        if (bestDist <= ORBM_TH_HIGH) vnMatch[i] = bestIdx;
    }
    return ORBM_OK;
}

int orbm_search_by_sim3(orbm_t* m, const orbm_frame_t* kf1, const orbm_frame_t* kf2, const float* sf1, const float* sf2,
                        const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* qdesc1,
                        const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* qdesc2,
                        float th, int32_t* matches12) {
    if (!m || !kf1 || !kf2) return ORBM_E_INVALID;
    std::vector<int> vnMatch1, vnMatch2;
    int rc = sim3_one_way(m, kf1, kf2, sf2, valid1, u1, v1, level1, qdesc1, th, vnMatch1);
    if (rc) return rc;
    rc = sim3_one_way(m, kf2, kf1, sf1, valid2, u2, v2, level2, qdesc2, th, vnMatch2);
    if (rc) return rc;
    int nFound = 0;
    for (int i1 = 0; i1 < kf1->n; ++i1) {
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    return nFound;
}

int orbm_grid_build_batch_async(orbm_t* m, const orbm_kp_t* kps, const int32_t* counts, int nframes, int cap,
                                float min_x, float min_y, float inv_w, float inv_h, int32_t* grid_start, int32_t* grid_idx) {
    if (!m || !kps || !counts || !grid_start || !grid_idx || nframes < 1 || cap < 1 || cap > 65535) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    int n2 = 64; while (n2 < cap) n2 <<= 1;
    if ((size_t)n2 * 4 > 150 * 1024) { set_merr("too many keypoints per frame for the LDS sort"); return ORBM_E_CAPACITY; }
    if (cap > GRID_CS_MAX && n2 * 4 > 48 * 1024) MHIPCHK(hipFuncSetAttribute((const void*)k_grid_build_batch, hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 4));
    MHIPCHK(rec_time(m, m->e2));
    m->gridFirst = true;                                                    // orbm_last_timing then spans grid build + the next kernel
    if (cap <= GRID_CS_MAX)
        hipLaunchKernelGGL(k_grid_build_batch_cs, dim3(nframes), dim3(256), grid_cs_lds(cap), m->stream, (const KpIn*)kps, counts, cap,
                           min_x, min_y, inv_w, inv_h, grid_start, grid_idx);
    else
        hipLaunchKernelGGL(k_grid_build_batch, dim3(nframes), dim3(256), (size_t)n2 * 4, m->stream, (const KpIn*)kps, counts, cap, n2,
                           min_x, min_y, inv_w, inv_h, grid_start, grid_idx);
    MHIPCHK(hipGetLastError());
    return ORBM_OK;
}

int orbm_track_window_batch_async(orbm_t* m, const orbm_kp_t* kps, const uint8_t* desc, const int32_t* counts, int cap,
                                  const int32_t* grid_start, const int32_t* grid_idx,
                                  float min_x, float min_y, float inv_w, float inv_h,
                                  int q_first, int t_first, int npairs, float th, const float* sf, int nlevels,
                                  float dx, float dy, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
    if (!m || !kps || !desc || !counts || !grid_start || !grid_idx || !best_idx || !best_dist || !second_dist || npairs < 1 ||
        nlevels < 1 || nlevels > 12 || !sf) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    ScaleTab st;
    for (int i = 0; i < 12; ++i) st.sf[i] = i < nlevels ? sf[i] : sf[nlevels - 1];
    MHIPCHK(rec_time(m, m->e0));
    hipLaunchKernelGGL(k_track_window, dim3((cap + 3) / 4, npairs), dim3(256), 0, m->stream, (const KpIn*)kps, desc, counts, cap,
                       grid_start, grid_idx, min_x, min_y, inv_w, inv_h, q_first, t_first, th, st, dx, dy, best_idx, best_dist, second_dist);
    MHIPCHK(rec_time(m, m->e1));
    MHIPCHK(hipGetLastError());
    m->timed = true;
    return ORBM_OK;
}

int orbm_search_by_projection_batch_async(orbm_t* m, const orbm_kp_t* kps, const uint8_t* desc, const int32_t* counts, int cap,
                                          const int32_t* grid_start, const int32_t* grid_idx,
                                          float min_x, float min_y, float inv_w, float inv_h,
                                          int q_first, int t_first, int npairs, float th, const float* sf, int nlevels,
                                          float dx, float dy, const uint8_t* t_blocked, const uint8_t* q_obs, int check_orientation,
                                          int32_t* match, int32_t* nmatches) {
    if (!m || !kps || !desc || !counts || !grid_start || !grid_idx || !match || !nmatches || npairs < 1 || cap < 1 || cap > 65535 ||
        nlevels < 1 || nlevels > 12 || !sf || q_first < 0 || t_first < 0) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    ScaleTab st;
    for (int i = 0; i < 12; ++i) st.sf[i] = i < nlevels ? sf[i] : sf[nlevels - 1];
    const size_t lds = (size_t)(2 * ((cap + 31) >> 5) + 32 + cap) * sizeof(unsigned);   // blocked bits, rotation histogram, observed bits, proposal tags
    if (lds > 64 * 1024) { set_merr("SearchByProjection batch: %d keypoint slots per frame need %zu B of LDS (limit 64 KB, ~16 000 slots)", cap, lds); return ORBM_E_INVALID; }
    // scratch of the handle: per query the window population and its TK_K best candidates, per pair the (slot, bin) list of the assignments
    const size_t rows = (size_t)npairs * cap;
    const size_t bCnt = (rows * sizeof(int) + 255) & ~(size_t)255, bKeys = (rows * TK_K * sizeof(unsigned) + 255) & ~(size_t)255,
                 bAcc = (rows * sizeof(unsigned) + 255) & ~(size_t)255, bEnt = rows * sizeof(uint4);
    uint8_t* scr = batch_scratch(m, bCnt + bKeys + bAcc + bEnt);
    if (!scr) { set_merr("SearchByProjection batch scratch of %zu B unavailable (inside a capture, run the call once eagerly first)", bCnt + bKeys + bAcc + bEnt); return ORBM_E_HIP; }
    int* topCnt = (int*)scr; unsigned* topKeys = (unsigned*)(scr + bCnt); unsigned* acc = (unsigned*)(scr + bCnt + bKeys);
    uint4* ent = (uint4*)(scr + bCnt + bKeys + bAcc);                      // the searched frames' grid entries, packed (k_track_pack)
    const float factor = ORBM_HISTO_LENGTH / 360.0f;                        // ORBmatcher.cc:2478
    MHIPCHK(rec_time(m, m->e0));
#ifdef ORBX_AB
    if (ab_env("ORBM_TOPK_WAVE"))                                           // A/B: a wave per query
        hipLaunchKernelGGL(k_track_topk, dim3((cap + 3) / 4, npairs), dim3(256), 0, m->stream, (const KpIn*)kps, desc, counts, cap,
                           grid_start, grid_idx, min_x, min_y, inv_w, inv_h, q_first, t_first, th, st, dx, dy, factor, topCnt, topKeys);
    else
#endif
    {
        hipLaunchKernelGGL(k_track_pack, dim3((cap + 255) / 256, npairs), dim3(256), 0, m->stream, (const KpIn*)kps, cap, grid_start, grid_idx, t_first, ent);
        hipLaunchKernelGGL(k_track_topk16, dim3((cap + 15) / 16, npairs), dim3(256), 0, m->stream, (const KpIn*)kps, desc, counts, cap,
                           grid_start, ent, min_x, min_y, inv_w, inv_h, q_first, t_first, th, st, dx, dy, factor, topCnt, topKeys);
    }
#ifdef ORBX_AB
    if (ab_env("ORBM_CLAIM_V1"))                                            // A/B: eight queries per step
        hipLaunchKernelGGL(k_track_claim, dim3(npairs), dim3(64), lds, m->stream, (const KpIn*)kps, desc, counts, cap,
                           grid_start, grid_idx, min_x, min_y, inv_w, inv_h, q_first, t_first, th, st, dx, dy, factor, topCnt, topKeys,
                           t_blocked, q_obs, check_orientation, acc, match, nmatches);
    else
#endif
    hipLaunchKernelGGL(k_track_claim64, dim3(npairs), dim3(64), lds, m->stream, (const KpIn*)kps, desc, counts, cap,
                       grid_start, grid_idx, min_x, min_y, inv_w, inv_h, q_first, t_first, th, st, dx, dy, factor, topCnt, topKeys,
                       t_blocked, q_obs, check_orientation, acc, match, nmatches);
    MHIPCHK(rec_time(m, m->e1));
    MHIPCHK(hipGetLastError());
    m->timed = true;
    return ORBM_OK;
}

// ---- DBoW2 vocabulary ----
struct orbm_vocab {
    int k = 0, L = 0, nnodes = 0, nwords = 0, device = 0;
    // level-major layout (k_bow_transform2): node rows renumbered breadth-first, the children of a node contiguous in child order
    unsigned* dInfo = nullptr;                                 // [nnodes] first child << 5 | child count
    int *dOrig = nullptr, *dWord = nullptr;                    // [nnodes] the vocabulary's own node id; word id (leaves)
    uint8_t* dDesc = nullptr;
    double* dWeight = nullptr;
};

void orbm_vocab_destroy(orbm_vocab_t* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    void* ptrs[] = {v->dInfo, v->dOrig, v->dWord, v->dDesc, v->dWeight};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete v;
}

int orbm_vocab_create(orbm_t* m, orbm_vocab_t** out, int k, int L, int nnodes, const int32_t* parent, const uint8_t* is_leaf,
                      const uint8_t* desc, const double* weight) {
    if (!m || !out || nnodes < 2 || !parent || !is_leaf || !desc || !weight) return ORBM_E_INVALID;
    *out = nullptr;
    // children lists in node-id order, word ids in leaf order: what loadFromTextFile builds (:1385-1420)
    std::vector<int> cstart(nnodes + 1, 0), cidx(nnodes - 1), word(nnodes, 0);
    for (int i = 1; i < nnodes; ++i) {
        if (parent[i] < 0 || parent[i] >= i) { set_merr("vocabulary node %d has parent %d (must precede it)", i, parent[i]); return ORBM_E_INVALID; }
        cstart[parent[i] + 1]++;
    }
    for (int i = 0; i < nnodes; ++i) cstart[i + 1] += cstart[i];
    std::vector<int> fill(cstart.begin(), cstart.end() - 1);
    int nwords = 0;
    for (int i = 1; i < nnodes; ++i) { cidx[fill[parent[i]]++] = i; if (is_leaf[i]) word[i] = nwords++; }
    for (int i = 1; i < nnodes; ++i) {
        if ((cstart[i + 1] == cstart[i]) != (is_leaf[i] != 0)) { set_merr("vocabulary node %d: leaf flag disagrees with its children", i); return ORBM_E_INVALID; }
        if (cstart[i + 1] - cstart[i] > 31) { set_merr("vocabulary node %d has %d children (at most 31 supported; DBoW2 allows k <= 20)", i, cstart[i + 1] - cstart[i]); return ORBM_E_INVALID; }
    }
    if (cstart[1] - cstart[0] > 31) { set_merr("vocabulary root has more than 31 children"); return ORBM_E_INVALID; }
    // level-major renumbering: breadth-first from the root, children of a node consecutive in their child-list order
    std::vector<int> order; order.reserve(nnodes);             // new row -> vocabulary node id
    std::vector<int> newOf(nnodes, -1);
    order.push_back(0); newOf[0] = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const int id = order[h];
        for (int c = cstart[id]; c < cstart[id + 1]; ++c) { newOf[cidx[c]] = (int)order.size(); order.push_back(cidx[c]); }
    }
    if ((int)order.size() != nnodes) { set_merr("vocabulary is not one tree (%zu of %d nodes reachable from the root)", order.size(), nnodes); return ORBM_E_INVALID; }
    std::vector<unsigned> info(nnodes);
    std::vector<int> wordN(nnodes);
    std::vector<double> weightN(nnodes);
    std::vector<uint8_t> descN((size_t)32 * nnodes);
    for (int r = 0; r < nnodes; ++r) {
        const int id = order[r];
        const int cnt = cstart[id + 1] - cstart[id];
        info[r] = cnt ? ((unsigned)newOf[cidx[cstart[id]]] << 5) | (unsigned)cnt : 0u;
        wordN[r] = word[id]; weightN[r] = weight[id];
        memcpy(&descN[(size_t)32 * r], desc + (size_t)32 * id, 32);
    }
    MHIPCHK(hipSetDevice(m->device));
    orbm_vocab* v = new orbm_vocab;
    v->k = k; v->L = L; v->nnodes = nnodes; v->nwords = nwords; v->device = m->device;
    bool ok = hipMalloc((void**)&v->dInfo, sizeof(unsigned) * nnodes) == hipSuccess && hipMalloc((void**)&v->dOrig, sizeof(int) * nnodes) == hipSuccess &&
              hipMalloc((void**)&v->dWord, sizeof(int) * nnodes) == hipSuccess && hipMalloc((void**)&v->dDesc, (size_t)32 * nnodes) == hipSuccess &&
              hipMalloc((void**)&v->dWeight, sizeof(double) * nnodes) == hipSuccess;
    ok = ok && hipMemcpy(v->dInfo, info.data(), sizeof(unsigned) * nnodes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->dOrig, order.data(), sizeof(int) * nnodes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->dWord, wordN.data(), sizeof(int) * nnodes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->dDesc, descN.data(), (size_t)32 * nnodes, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->dWeight, weightN.data(), sizeof(double) * nnodes, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { set_merr("vocabulary upload failed"); orbm_vocab_destroy(v); return ORBM_E_HIP; }
    *out = v;
    return ORBM_OK;
}

int orbm_vocab_load_text(orbm_t* m, orbm_vocab_t** out, const char* path) {
    if (!m || !out || !path) return ORBM_E_INVALID;
    FILE* f = fopen(path, "r");
    if (!f) { set_merr("cannot open vocabulary %s", path); return ORBM_E_INVALID; }
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        fclose(f); set_merr("vocabulary %s: not a correct text file", path); return ORBM_E_INVALID;       // :1359-1363
    }
    std::vector<int> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    for (;;) {
        int pid, isleaf;
        if (fscanf(f, "%d %d", &pid, &isleaf) != 2) break;
        uint8_t d[32];
        bool good = true;
        for (int i = 0; i < 32; ++i) { int b; if (fscanf(f, "%d", &b) != 1) { good = false; break; } d[i] = (uint8_t)b; }
        double w;
        if (!good || fscanf(f, "%lf", &w) != 1) break;
        parent.push_back(pid); leaf.push_back(isleaf > 0); weight.push_back(w);
        desc.insert(desc.end(), d, d + 32);
    }
    fclose(f);
    return orbm_vocab_create(m, out, k, L, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data());
}

int orbm_vocab_info(const orbm_vocab_t* v, int* k, int* L, int* nnodes, int* nwords) {
    if (!v) return ORBM_E_INVALID;
    *k = v->k; *L = v->L; *nnodes = v->nnodes; *nwords = v->nwords;
    return ORBM_OK;
}

int orbm_bow_transform(orbm_t* m, const orbm_vocab_t* v, const uint8_t* desc, int n, int levelsup,
                       int32_t* word_id, int32_t* node_id, double* weight) {
    if (!m || !v || n < 0 || (n > 0 && (!desc || !word_id || !node_id || !weight))) return ORBM_E_INVALID;
    if (n == 0) return ORBM_OK;
    if (v->device != m->device) { set_merr("vocabulary lives on another device"); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(m->device));
    DevBuf dd, dw, dn, dwt;
    arena_reset(m);
    UP(dd, desc, (size_t)32 * n); AL(dw, sizeof(int) * n); AL(dn, sizeof(int) * n); AL(dwt, sizeof(double) * n);
    m->gridFirst = false;
    MHIPCHK(rec_time(m, m->e0));
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_bow_transform2, dim3((unsigned)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, m->stream, dd.as<uint8_t>(), n, v->dInfo,
                       v->dDesc, v->dOrig, v->dWord, v->dWeight, v->L, levelsup, dw.as<int>(), dn.as<int>(), dwt.as<double>());
    MHIPCHK(rec_time(m, m->e1));
    m->timed = true;
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    memcpy(word_id, dw.host(), sizeof(int) * n);
    memcpy(node_id, dn.host(), sizeof(int) * n);
    memcpy(weight, dwt.host(), sizeof(double) * n);
    return ORBM_OK;
}

int orbm_bow_vectors(int n, const int32_t* word_id, const int32_t* node_id, const double* weight,
                     int32_t* bow_ids, double* bow_vals, int* nbow,
                     int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_idx, int* nfv) {
    // BowVector::addWeight in feature order (BowVector.cpp:34-46), then L1 normalisation in word order (:62-84);
    // FeatureVector::addFeature (FeatureVector.cpp:34-46).  Sorted vectors instead of std::map, same summation order.
    std::vector<std::pair<int, int>> byWord, byNode;             // (key, feature)
    for (int i = 0; i < n; ++i) if (weight[i] > 0) { byWord.push_back(std::make_pair(word_id[i], i)); byNode.push_back(std::make_pair(node_id[i], i)); }
    std::stable_sort(byWord.begin(), byWord.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    std::stable_sort(byNode.begin(), byNode.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    int b = 0;
    for (size_t i = 0; i < byWord.size();) {
        size_t j = i;
        double acc = weight[byWord[i].second];
        for (j = i + 1; j < byWord.size() && byWord[j].first == byWord[i].first; ++j) acc += weight[byWord[j].second];
        bow_ids[b] = byWord[i].first; bow_vals[b] = acc; ++b;
        i = j;
    }
    double norm = 0.0;
    for (int i = 0; i < b; ++i) norm += fabs(bow_vals[i]);
    if (norm > 0.0) for (int i = 0; i < b; ++i) bow_vals[i] /= norm;
    *nbow = b;
    int c = 0, o = 0;
    for (size_t i = 0; i < byNode.size();) {
        size_t j = i;
        fv_nodes[c] = byNode[i].first; fv_start[c] = o;
        for (j = i; j < byNode.size() && byNode[j].first == byNode[i].first; ++j) fv_idx[o++] = byNode[j].second;
        ++c;
        i = j;
    }
    fv_start[c] = o;
    *nfv = c;
    return ORBM_OK;
}

int orbm_search_by_projection_frame_fisheye(orbm_t* m, const orbm_frame_t* cur_l, const orbm_frame_t* cur_r,
                                            const uint8_t* blocked_l_in, const uint8_t* blocked_r_in, const float* sf,
                                            int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const float* vr,
                                            const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                            float th, int bForward, int bBackward, int check_ori, int32_t* match_l, int32_t* match_r) {
    if (!m || !cur_l || !cur_r || nq < 0) return ORBM_E_INVALID;
    std::vector<float> qr(nq);
    std::vector<int> minl(nq), maxl(nq);
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) { qr[i] = -1.f; minl[i] = 0; maxl[i] = -1; continue; }
        const int o = octave[i];
        qr[i] = th * sf[o];
        if (bForward) { minl[i] = o; maxl[i] = -1; } else if (bBackward) { minl[i] = 0; maxl[i] = o; } else { minl[i] = o - 1; maxl[i] = o + 1; }
    }
    orbm_frame_t fl = *cur_l, fr = *cur_r; fl.uright = nullptr; fr.uright = nullptr;     // no stereo gate when Nleft != -1 (:2569)
    int capL = std::max(1, std::min(fl.n, 2048)), capR = std::max(1, std::min(fr.n, 2048));      // row strides after window_pass
    std::vector<int> cntL, idxL, distL, cntR, idxR, distR;
    int rc = window_pass(m, &fl, nq, u, v, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, capL, cntL, idxL, distL);
    if (rc) return rc;
    rc = window_pass(m, &fr, nq, ur, vr, qr.data(), minl.data(), maxl.data(), nullptr, nullptr, qdesc, capR, cntR, idxR, distR);
    if (rc) return rc;
    int nmatches = 0;
    const int Nleft = fl.n;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;
    std::vector<uint8_t> bl(blocked_l_in, blocked_l_in + fl.n), br(blocked_r_in, blocked_r_in + fr.n);
    for (int i = 0; i < fl.n; ++i) match_l[i] = -1;
    for (int i = 0; i < fr.n; ++i) match_r[i] = -1;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i] || cntL[i] == 0) continue;                            // an empty left window skips the right block too (:2551)
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < cntL[i]; ++c) {
            const int i2 = idxL[(size_t)i * capL + c];
            if (bl[i2]) continue;
            const int d = distL[(size_t)i * capL + c];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= ORBM_TH_HIGH) {
            match_l[bestIdx2] = i;
            if (mp_obs[i]) bl[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], fl.kps[bestIdx2].angle, factor, bestIdx2);
        }
        bestDist = 256; bestIdx2 = -1;
        for (int c = 0; c < cntR[i]; ++c) {
            const int i2 = idxR[(size_t)i * capR + c];
            if (br[i2]) continue;
            const int d = distR[(size_t)i * capR + c];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= ORBM_TH_HIGH) {
            match_r[bestIdx2] = i;
            if (mp_obs[i]) br[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], fr.kps[bestIdx2].angle, factor, bestIdx2 + Nleft);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int k : rh.bins[b]) { if (k < Nleft) match_l[k] = ORBM_MATCH_PRUNED; else match_r[k - Nleft] = ORBM_MATCH_PRUNED; nmatches--; }
    }
    return nmatches;
}

int orbm_search_by_projection_points_fisheye(orbm_t* m, const orbm_frame_t* f_l, const orbm_frame_t* f_r,
                                             const uint8_t* blocked_l_in, const uint8_t* blocked_r_in,
                                             const int32_t* l2r, const int32_t* r2l, const float* sf,
                                             int nq, const uint8_t* in_view, const float* px, const float* py, const float* view_cos, const int32_t* level,
                                             const uint8_t* in_view_r, const float* pxr, const float* pyr, const float* view_cos_r, const int32_t* level_r,
                                             const uint8_t* qdesc, const uint8_t* mp_obs, float th, float nnratio, int32_t* match_l, int32_t* match_r) {
    if (!m || !f_l || !f_r || nq < 0) return ORBM_E_INVALID;
    const bool bFactor = th != 1.0;
    std::vector<float> qrl(nq), qrr(nq);
    std::vector<int> minL(nq), maxL(nq), minR(nq), maxR(nq);
    for (int i = 0; i < nq; ++i) {
        if (!in_view[i]) { qrl[i] = -1.f; minL[i] = 0; maxL[i] = -1; }
        else {
            float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            qrl[i] = r * sf[level[i]]; minL[i] = level[i] - 1; maxL[i] = level[i];
        }
        if (!in_view_r[i] || level_r[i] == -1) { qrr[i] = -1.f; minR[i] = 0; maxR[i] = -1; }
        else {
            const float r = view_cos_r[i] > 0.998 ? 2.5f : 4.0f;             // the right block applies no th factor (:174)
            qrr[i] = r * sf[level_r[i]]; minR[i] = level_r[i] - 1; maxR[i] = level_r[i];
        }
    }
    orbm_frame_t fl = *f_l, fr = *f_r; fl.uright = nullptr; fr.uright = nullptr;
    int capL = std::max(1, std::min(fl.n, 2048)), capR = std::max(1, std::min(fr.n, 2048));      // row strides after window_pass
    std::vector<int> cntL, idxL, distL, cntR, idxR, distR;
    int rc = window_pass(m, &fl, nq, px, py, qrl.data(), minL.data(), maxL.data(), nullptr, nullptr, qdesc, capL, cntL, idxL, distL);
    if (rc) return rc;
    rc = window_pass(m, &fr, nq, pxr, pyr, qrr.data(), minR.data(), maxR.data(), nullptr, nullptr, qdesc, capR, cntR, idxR, distR);
    if (rc) return rc;
    int nmatches = 0;
    std::vector<uint8_t> bl(blocked_l_in, blocked_l_in + fl.n), br(blocked_r_in, blocked_r_in + fr.n);
    for (int i = 0; i < fl.n; ++i) match_l[i] = -1;
    for (int i = 0; i < fr.n; ++i) match_r[i] = -1;
    for (int iMP = 0; iMP < nq; ++iMP) {
        if (!in_view[iMP] && !in_view_r[iMP]) continue;
        if (in_view[iMP] && cntL[iMP] > 0) {
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int c = 0; c < cntL[iMP]; ++c) {
                const int k = idxL[(size_t)iMP * capL + c];
                if (bl[k]) continue;
                const int d = distL[(size_t)iMP * capL + c];
                if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = fl.kps[k].octave; bestIdx = k; }
                else if (d < bestDist2) { bestLevel2 = fl.kps[k].octave; bestDist2 = d; }
            }
            if (bestDist <= ORBM_TH_HIGH) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;          // :148-149 (skips the right block)
                if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                    match_l[bestIdx] = iMP;
                    if (mp_obs[iMP]) bl[bestIdx] = 1;
                    if (l2r[bestIdx] != -1) { match_r[l2r[bestIdx]] = iMP; if (mp_obs[iMP]) br[l2r[bestIdx]] = 1; nmatches++; }
                    nmatches++;
                }
            }
        }
        if (in_view_r[iMP] && level_r[iMP] != -1) {
            if (cntR[iMP] == 0) continue;
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int c = 0; c < cntR[iMP]; ++c) {
                const int k = idxR[(size_t)iMP * capR + c];
                if (br[k]) continue;
                const int d = distR[(size_t)iMP * capR + c];
                if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = fr.kps[k].octave; bestIdx = k; }
                else if (d < bestDist2) { bestLevel2 = fr.kps[k].octave; bestDist2 = d; }
            }
            if (bestDist <= ORBM_TH_HIGH) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                if (r2l[bestIdx] != -1) { match_l[r2l[bestIdx]] = iMP; if (mp_obs[iMP]) bl[r2l[bestIdx]] = 1; nmatches++; }
                match_r[bestIdx] = iMP;
                if (mp_obs[iMP]) br[bestIdx] = 1;
                nmatches++;
            }
        }
    }
    return nmatches;
}

int orbm_search_by_bow_fisheye(orbm_t* m, int nkf, const orbm_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                               int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                               int nf, int nleft, const orbm_kp_t* kps_f, const uint8_t* desc_f,
                               int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                               float nnratio, int check_ori, int32_t* f_match) {
    if (!m || nkf < 0 || nf < 0) return ORBM_E_INVALID;
    JoinJobs J;
    join_nodes(nnk, nodes_k, start_k, idx_k, nnf, nodes_f, start_f, J);
    std::vector<int> dist;
    int rc = bucket_pass(m, desc_kf, nkf, desc_f, nf, idx_f, start_f[nnf], J, dist);
    if (rc) return rc;
    for (int i = 0; i < nf; ++i) f_match[i] = -1;
    int nmatches = 0;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;
    for (size_t j = 0; j < J.q.size(); ++j) {
        const int iKF = J.q[j];
        if (!kf_good[iKF]) continue;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (int c = 0; c < J.len[j]; ++c) {
            const int iF = idx_f[J.l2[j] + c];
            if (f_match[iF] >= 0) continue;
            const int d = dist[J.off[j] + c];
            if (iF < nleft && d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = iF; }
            else if (iF < nleft && d < bestDist2) bestDist2 = d;
            if (iF >= nleft && d < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = d; bestIdxFR = iF; }
            else if (iF >= nleft && d < bestDist2R) bestDist2R = d;
        }
        if (bestDist1 <= ORBM_TH_LOW) {
            if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                f_match[bestIdxF] = iKF;
                if (check_ori) rh.add(kps_kf[iKF].angle, kps_f[bestIdxF].angle, factor, bestIdxF);
                nmatches++;
            }
            if (bestDist1R <= ORBM_TH_LOW) {                                 // nested; ratio test disabled by `|| true` (:471-473)
                f_match[bestIdxFR] = iKF;
                if (check_ori) rh.add(kps_kf[iKF].angle, kps_f[bestIdxFR].angle, factor, bestIdxFR);
                nmatches++;
            }
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int i : rh.bins[b]) { f_match[i] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_search_by_bow_kf(orbm_t* m, int n1, const orbm_kp_t* kps1, const uint8_t* desc1, const uint8_t* good1,
                          int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1,
                          int n2, const orbm_kp_t* kps2, const uint8_t* desc2, const uint8_t* good2,
                          int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2,
                          float nnratio, int check_ori, int32_t* vpMatches12) {
    if (!m || n1 < 0 || n2 < 0) return ORBM_E_INVALID;
    JoinJobs J;
    join_nodes(nn1, nodes1, start1, idx1, nn2, nodes2, start2, J);
    std::vector<int> dist;
    int rc = bucket_pass(m, desc1, n1, desc2, n2, idx2, start2[nn2], J, dist);
    if (rc) return rc;
    for (int i = 0; i < n1; ++i) vpMatches12[i] = -1;
    std::vector<uint8_t> vbMatched2(n2, 0);
    int nmatches = 0;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;                        // :978
    for (size_t j = 0; j < J.q.size(); ++j) {
        const int i1 = J.q[j];
        if (!good1[i1]) continue;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int c = 0; c < J.len[j]; ++c) {
            const int i2 = idx2[J.l2[j] + c];
            if (vbMatched2[i2] || !good2[i2]) continue;
            const int d = dist[J.off[j] + c];
            if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist1 < ORBM_TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            vpMatches12[i1] = bestIdx2;
            vbMatched2[bestIdx2] = 1;
            if (check_ori) rh.add(kps1[i1].angle, kps2[bestIdx2].angle, factor, i1);
            nmatches++;
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int i : rh.bins[b]) { vpMatches12[i] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_search_by_bow(orbm_t* m, int nkf, const orbm_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                       int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                       int nf, const orbm_kp_t* kps_f, const uint8_t* desc_f,
                       int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                       float nnratio, int check_ori, int32_t* f_match) {
    if (!m || nkf < 0 || nf < 0) return ORBM_E_INVALID;
    JoinJobs J;
    join_nodes(nnk, nodes_k, start_k, idx_k, nnf, nodes_f, start_f, J);
    std::vector<int> dist;
    int rc = bucket_pass(m, desc_kf, nkf, desc_f, nf, idx_f, start_f[nnf], J, dist);
    if (rc) return rc;
    for (int i = 0; i < nf; ++i) f_match[i] = -1;
    int nmatches = 0;
    RotHist rh;
    const float factor = ORBM_HISTO_LENGTH / 360.0f;                        // :334
    for (size_t j = 0; j < J.q.size(); ++j) {                              // jobs are in the reference's (node, iKF) order
        const int iKF = J.q[j];
        if (!kf_good[iKF]) continue;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int c = 0; c < J.len[j]; ++c) {
            const int iF = idx_f[J.l2[j] + c];
            if (f_match[iF] >= 0) continue;                                 // :385 -- order-dependent gate
            const int d = dist[J.off[j] + c];
            if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = iF; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist1 <= ORBM_TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            f_match[bestIdxF] = iKF;
            if (check_ori) rh.add(kps_kf[iKF].angle, kps_f[bestIdxF].angle, factor, bestIdxF);
            nmatches++;
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int i : rh.bins[b]) { f_match[i] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_stereo_matches(orbm_t* m, void* left, int frame_l, void* right, int frame_r,
                        int nl, const orbm_kp_t* kl, const uint8_t* dl, int nr, const orbm_kp_t* kr, const uint8_t* dr,
                        float mb, float mbf, float* uright, float* depth) {
    if (!m || !left || !right || nl < 0 || nr < 0 || nr > 65535) return ORBM_E_INVALID;
    for (int i = 0; i < nl; ++i) { uright[i] = -1.0f; depth[i] = -1.0f; }
    if (nl == 0 || nr == 0) return 0;
    StereoLevels lv;
    memset(&lv, 0, sizeof lv);
    int nlev = 0, nlevR = 0, devL = 0, devR = 0, wl[12], hl[12], hr[12];
    float sfR[12], isfR[12];
    if (orbx_internal_levels(left, frame_l, &nlev, lv.L, lv.pitchL, wl, hl, lv.sf, lv.isf, &devL) ||
        orbx_internal_levels(right, frame_r, &nlevR, lv.R, lv.pitchR, lv.wR, hr, sfR, isfR, &devR)) {
        set_merr("extractor handles hold no pyramid for the requested batch slot"); return ORBM_E_INVALID;
    }
    if (nlev != nlevR || devL != m->device || devR != m->device) { set_merr("extractors and matcher must share one device and level count"); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(m->device));
    DevBuf bkl, bdl, bkr, bdr, bur, bde, bsad;
    arena_reset(m);
    UP(bkl, kl, sizeof(KpIn) * nl); UP(bdl, dl, (size_t)32 * nl); UP(bkr, kr, sizeof(KpIn) * nr); UP(bdr, dr, (size_t)32 * nr);
    AL(bur, sizeof(float) * nl); AL(bde, sizeof(float) * nl); AL(bsad, sizeof(int) * nl);
    MHIPCHK(rec_time(m, m->e0));
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_stereo, dim3((nl + 3) / 4), dim3(256), 0, m->stream, bkl.as<KpIn>(), bdl.as<uint8_t>(), nl, bkr.as<KpIn>(),
                       bdr.as<uint8_t>(), nr, lv, mb, mbf, bur.as<float>(), bde.as<float>(), bsad.as<int>());
    MHIPCHK(rec_time(m, m->e1));
    m->timed = true;
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    std::vector<int> sad(nl);
    memcpy(uright, bur.host(), sizeof(float) * nl);
    memcpy(depth, bde.host(), sizeof(float) * nl);
    memcpy(sad.data(), bsad.host(), sizeof(int) * nl);
    // median cut (Frame.cc:1261-1275)
    std::vector<std::pair<int, int>> vDistIdx;
    for (int i = 0; i < nl; ++i) if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
    if (vDistIdx.empty()) return 0;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
        if (vDistIdx[i].first < thDist) break;
        uright[vDistIdx[i].second] = -1; depth[vDistIdx[i].second] = -1; --kept;
    }
    return kept;
}

// ---- batched, device-resident forms of M15 / 8(f).1 / M10 (config C3 of bench.py: nothing leaves HBM between the extraction and
// the match lists) ------------------------------------------------------------------------------------------
extern "C" int orbx_internal_batch_layout(void* o, const uint8_t* const** l0tab, int* l0pitch, const uint8_t** pyr, size_t* frameBytes,
                                          int* nlevels, int* off, int* pitch, int* w, int* h, float* sf, float* isf, int* device, int* maxBatch, int* kpCap);

int orbm_stereo_batch_async(orbm_t* m, void* extractor, int first_l, int first_r, int npairs, const orbm_kp_t* kps, const uint8_t* desc,
                            const int32_t* counts, int cap, float mb, float mbf, float* uright, float* depth, int32_t* sad, int32_t* kept) {
    if (!m || !extractor || !kps || !desc || !counts || !uright || !depth || !sad || !kept || npairs < 1 || first_l < 0 || first_r < 0 || cap < 1 || cap > 65535) return ORBM_E_INVALID;
    StereoBatchLayout B;
    memset(&B, 0, sizeof B);
    int nlev = 0, dev = 0, maxBatch = 0, kpCap = 0, hl[12] = {};
    if (orbx_internal_batch_layout(extractor, &B.l0, &B.l0pitch, &B.pyr, &B.frameBytes, &nlev, B.off, B.pitch, B.w, hl, B.sf, B.isf, &dev, &maxBatch, &kpCap)) {
        set_merr("the extractor handle holds no geometry"); return ORBM_E_INVALID;
    }
    if (dev != m->device || cap != kpCap || first_l + npairs > maxBatch || first_r + npairs > maxBatch) { set_merr("extractor / matcher mismatch (device, capacity or batch range)"); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(m->device));
    int n2 = 64; while (n2 < cap) n2 <<= 1;
    if (n2 * 4 > 48 * 1024) MHIPCHK(hipFuncSetAttribute((const void*)k_stereo_cut, hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 4));
    // vRowIndices (Frame.cc:1064-1083) of every right image as CSR in the handle's scratch: [npairs][nrows + 1] starts + [npairs][rowCap] indices
    // A keypoint enters rows floor(y - r) .. ceil(y + r), r = 2 * sf[octave] (Frame.cc:1071-1077): at most floor(2 r) + 3 rows (and never
    // more than the image has), so [cap] keypoints of the coarsest level bound every right image's list.
    float sfMax = 1.f;
    for (int l = 0; l < nlev; ++l) sfMax = std::max(sfMax, B.sf[l]);
    const int nrows = hl[0];
    const long long perKp = std::min<long long>((long long)floorf(4.f * sfMax) + 3, nrows);
    if (perKp * cap > 0x7fffffffLL) { set_merr("stereo row lists too large"); return ORBM_E_INVALID; }
    const int rowCap = (int)(perKp * cap);
    const size_t bStart = ((size_t)npairs * (nrows + 1) * sizeof(int) + 255) & ~(size_t)255, bIdx = ((size_t)npairs * rowCap * sizeof(unsigned short) + 255) & ~(size_t)255;
    uint8_t* scr = batch_scratch(m, bStart + bIdx + 256);
    if (!scr) { set_merr("stereo scratch of %zu B unavailable (inside a capture, run the call once eagerly first)", bStart + bIdx + 256); return ORBM_E_HIP; }
    int* rowStart = (int*)scr; unsigned short* rowIdx = (unsigned short*)(scr + bStart);
    int* rowErr = m->hStatus;                                    // cannot happen with the bound above; if it ever does, orbm_sync reports ORBM_E_CAPACITY
    const size_t ldsRows = (size_t)(2 * nrows + 2) * sizeof(int);
    if (ldsRows > 48 * 1024) MHIPCHK(hipFuncSetAttribute((const void*)k_stereo_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsRows));
    m->gridFirst = false;
    MHIPCHK(rec_time(m, m->e0));
    hipLaunchKernelGGL(k_stereo_rows, dim3(npairs), dim3(256), ldsRows, m->stream, (const KpIn*)kps, counts, cap, first_r, B, nrows, rowCap, rowStart, rowIdx, rowErr);
    hipLaunchKernelGGL(k_stereo_batch, dim3((cap + 256 / ST_GS - 1) / (256 / ST_GS), npairs), dim3(256), 0, m->stream, (const KpIn*)kps, desc, counts, cap, first_l, first_r, B, mb, mbf,
                       uright, depth, sad, rowStart, rowIdx, nrows, rowCap);
    hipLaunchKernelGGL(k_stereo_cut, dim3(npairs), dim3(256), 0, m->stream, counts, cap, first_l, n2, sad, uright, depth, kept);
    MHIPCHK(rec_time(m, m->e1));
    m->timed = true;
    MHIPCHK(hipGetLastError());
    return ORBM_OK;
}

int orbm_bow_nodes_batch_async(orbm_t* m, const orbm_vocab_t* v, const uint8_t* desc, int nrows, int levelsup, int32_t* node_id) {
    if (!m || !v || !desc || !node_id || nrows < 1) return ORBM_E_INVALID;
    if (v->device != m->device) { set_merr("vocabulary lives on another device"); return ORBM_E_INVALID; }
    MHIPCHK(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_bow_transform2, dim3((unsigned)(((size_t)nrows * 16 + 255) / 256)), dim3(256), 0, m->stream, desc, nrows, v->dInfo, v->dDesc, v->dOrig,
                       v->dWord, v->dWeight, v->L, levelsup, (int*)nullptr, node_id, (double*)nullptr);
    MHIPCHK(hipGetLastError());
    return ORBM_OK;
}

int orbm_triangulation_batch_async(orbm_t* m, int npairs, int cap,
                                   const orbm_kp_t* kps1, const uint8_t* desc1, const int32_t* counts1, const int32_t* node1, const float* uright1,
                                   const orbm_kp_t* kps2, const uint8_t* desc2, const int32_t* counts2, const int32_t* node2, const float* uright2,
                                   const float* F12, float epx, float epy, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                   int only_stereo, int coarse, int32_t* matches12, int32_t* nmatches) {
    if (!m || npairs < 1 || cap < 1 || cap > 65535 || !kps1 || !desc1 || !counts1 || !node1 || !kps2 || !desc2 || !counts2 || !node2 || !F12 || !scale_factors2 ||
        !level_sigma2_2 || nlevels < 1 || nlevels > 12 || !matches12 || !nmatches) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    TriParams P;
    for (int i = 0; i < 9; ++i) P.F12[i] = F12[i];
    P.epx = epx; P.epy = epy; P.onlyStereo = only_stereo; P.coarse = coarse;
    for (int i = 0; i < 12; ++i) { P.sf2[i] = scale_factors2[std::min(i, nlevels - 1)]; P.sigma2[i] = level_sigma2_2[std::min(i, nlevels - 1)]; }
    const size_t bS = ((size_t)npairs * 257 * sizeof(int) + 255) & ~(size_t)255, bI = ((size_t)npairs * cap * sizeof(unsigned short) + 255) & ~(size_t)255;
    uint8_t* scr = batch_scratch(m, bS + bI);
    if (!scr) { set_merr("triangulation scratch of %zu B unavailable (inside a capture, run the call once eagerly first)", bS + bI); return ORBM_E_HIP; }
    int* bStart = (int*)scr; unsigned short* bIdx = (unsigned short*)(scr + bS);
    MHIPCHK(hipMemsetAsync(nmatches, 0, sizeof(int) * npairs, m->stream));
    hipLaunchKernelGGL(k_tri_buckets, dim3(npairs), dim3(256), 0, m->stream, counts2, node2, cap, bStart, bIdx);
    hipLaunchKernelGGL(k_triangulate_batch, dim3((cap + 255) / 256, npairs), dim3(256), 0, m->stream, (const KpIn*)kps1, desc1, counts1, node1, uright1,
                       (const KpIn*)kps2, desc2, counts2, node2, uright2, cap, P, matches12, nmatches, bStart, bIdx);
    MHIPCHK(hipGetLastError());
    return ORBM_OK;
}

// ---- SURVEY 8(f).2 / 8(f).3 ------------------------------------------------------------------------------
static bool fill_undist(UndistParams& P, const float* k, const float* dist, int ndist, const float* newk) {
    for (double& d : P.k) d = 0.0;
    for (int i = 0; i < ndist && i < 14; ++i) P.k[i] = (double)dist[i];
    P.fx = k[0]; P.fy = k[1]; P.cx = k[2]; P.cy = k[3];
    P.nfx = newk[0]; P.nfy = newk[1]; P.ncx = newk[2]; P.ncy = newk[3];
    return ndist < 1 || dist[0] == 0.0f;                         // Frame.cc:928: first coefficient zero -> no undistortion
}

int orbm_undistort_keypoints(orbm_t* m, int space, const orbm_kp_t* kps, int n, const float* k, const float* dist, int ndist,
                             const float* newk, orbm_kp_t* out) {
    if (!m || n < 0 || !k || !newk || ndist < 0 || ndist > 14 || (ndist > 0 && !dist) || (n > 0 && (!kps || !out))) return ORBM_E_INVALID;
    if (n == 0) return 0;
    MHIPCHK(hipSetDevice(m->device));
    UndistParams P;
    const int pass = fill_undist(P, k, dist, ndist, newk) ? 1 : 0;
    if (space == ORBM_DEVICE) {
        hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, m->stream, (const KpIn*)kps, n, P, pass, (KpIn*)out);
        MHIPCHK(hipGetLastError());
        return n;
    }
    DevBuf di, dout;
    arena_reset(m);
    UP(di, kps, sizeof(KpIn) * n); AL(dout, sizeof(KpIn) * n);
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, m->stream, di.as<KpIn>(), n, P, pass, dout.as<KpIn>());
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    memcpy(out, dout.host(), sizeof(KpIn) * n);
    return n;
}

int orbm_image_bounds(orbm_t* m, int cols, int rows, const float* k, const float* dist, int ndist, const float* newk, float* bounds) {
    if (!m || !k || !newk || !bounds || ndist < 0 || ndist > 14 || (ndist > 0 && !dist)) return ORBM_E_INVALID;
    if (ndist < 1 || dist[0] == 0.0f) { bounds[0] = 0.f; bounds[1] = (float)cols; bounds[2] = 0.f; bounds[3] = (float)rows; return ORBM_OK; }
    orbm_kp_t c[4], u[4];
    memset(c, 0, sizeof(c));
    c[1].x = (float)cols; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
    int rc = orbm_undistort_keypoints(m, ORBM_HOST, c, 4, k, dist, ndist, newk, u);
    if (rc < 0) return rc;
    bounds[0] = std::min(u[0].x, u[2].x); bounds[1] = std::max(u[1].x, u[3].x);      // Frame.cc:1005-1008
    bounds[2] = std::min(u[0].y, u[1].y); bounds[3] = std::max(u[2].y, u[3].y);
    return ORBM_OK;
}

int orbm_is_in_frustum(orbm_t* m, int space, int n, const float* pw, const float* normal, const float* min_dist, const float* max_dist,
                       const float* rcw, const float* tcw, const float* ow, const float* k, const float* bounds,
                       float bf, float viewing_cos_limit, float log_scale_factor, int n_scale_levels,
                       uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, float* depth, int32_t* level, float* view_cos) {
    if (!m || n < 0 || !rcw || !tcw || !ow || !k || !bounds || n_scale_levels < 1) return ORBM_E_INVALID;
    if (n == 0) return 0;
    if (!pw || !normal || !min_dist || !max_dist || !in_view || !proj_x || !proj_y || !proj_xr || !depth || !level || !view_cos) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    FrustumParams F;
    memcpy(F.rcw, rcw, sizeof(F.rcw)); memcpy(F.tcw, tcw, sizeof(F.tcw)); memcpy(F.ow, ow, sizeof(F.ow));
    memcpy(F.k, k, sizeof(F.k)); memcpy(F.bounds, bounds, sizeof(F.bounds));
    F.bf = bf; F.cosLimit = viewing_cos_limit; F.logSF = log_scale_factor; F.nLevels = n_scale_levels;
    const dim3 grid((n + 255) / 256), block(256);
    if (space == ORBM_DEVICE) {
        hipLaunchKernelGGL(k_frustum, grid, block, 0, m->stream, n, pw, normal, min_dist, max_dist, F, in_view, proj_x, proj_y, proj_xr,
                           depth, level, view_cos);
        MHIPCHK(hipGetLastError());
        return 0;
    }
    DevBuf dp, dn, dmin, dmax, div, dx, dy, dxr, dd, dl, dvc;
    arena_reset(m);
    UP(dp, pw, sizeof(float) * 3 * n); UP(dn, normal, sizeof(float) * 3 * n); UP(dmin, min_dist, sizeof(float) * n); UP(dmax, max_dist, sizeof(float) * n);
    AL(div, n); AL(dx, sizeof(float) * n); AL(dy, sizeof(float) * n);
    // the reference leaves mTrackProjXR / mTrackDepth / mnTrackScaleLevel / mTrackViewCos untouched for rejected points:
    // in/out buffers that start from the caller's values
    UPIO(dxr, proj_xr, sizeof(float) * n); UPIO(dd, depth, sizeof(float) * n); UPIO(dl, level, sizeof(int) * n); UPIO(dvc, view_cos, sizeof(float) * n);
    ARENA_FLUSH(m);
    hipLaunchKernelGGL(k_frustum, grid, block, 0, m->stream, n, dp.as<float>(), dn.as<float>(), dmin.as<float>(), dmax.as<float>(), F,
                       div.as<uint8_t>(), dx.as<float>(), dy.as<float>(), dxr.as<float>(), dd.as<float>(), dl.as<int>(), dvc.as<float>());
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipStreamSynchronize(m->stream));
    ARENA_FETCH(m);
    memcpy(in_view, div.host(), n); memcpy(proj_x, dx.host(), sizeof(float) * n);
    memcpy(proj_y, dy.host(), sizeof(float) * n); memcpy(proj_xr, dxr.host(), sizeof(float) * n);
    memcpy(depth, dd.host(), sizeof(float) * n); memcpy(level, dl.host(), sizeof(int) * n);
    memcpy(view_cos, dvc.host(), sizeof(float) * n);
    int cnt = 0;
    for (int i = 0; i < n; ++i) cnt += in_view[i] ? 1 : 0;
    return cnt;
}

}  // extern "C"
