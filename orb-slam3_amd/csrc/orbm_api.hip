// orbm_api.hip -- C ABI (include/orbm.h) over the gfx950 matcher kernels.  No CPU fallback for the
// device entry points; orbm_hamming / orbm_three_maxima are the reference's scalar helpers and stay scalar.
#include "../../include/orbm.h"
#include "orbm_kernels.hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace orbmk;

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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool timed = false;
};

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
        hipEventCreate(&m->e1) != hipSuccess) { set_merr("stream/event creation failed"); orbm_destroy(m); return ORBM_E_HIP; }
    *out = m;
    return ORBM_OK;
}

void orbm_destroy(orbm_t* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    if (m->e0) (void)hipEventDestroy(m->e0);
    if (m->e1) (void)hipEventDestroy(m->e1);
    delete m;
}

int orbm_sync(orbm_t* m) {
    if (!m) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    return ORBM_OK;
}

void* orbm_stream(const orbm_t* m) { return m ? (void*)m->stream : nullptr; }

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

int orbm_knn2_batch_async(orbm_t* m, const uint8_t* q, int q_stride, const int32_t* nq, const uint8_t* t, int t_stride,
                          const int32_t* nt, int npairs, int max_nt, int32_t* idx2, int32_t* dist2) {
    (void)max_nt;
    if (!m || !q || !t || !nq || !nt || !idx2 || !dist2 || npairs < 1 || q_stride < 1 || t_stride < 1) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipEventRecord(m->e0, m->stream));
    hipLaunchKernelGGL(k_knn2, dim3((q_stride + 63) / 64, npairs), dim3(256), 0, m->stream, q, q_stride, nq, t, t_stride, nt, idx2, dist2);
    MHIPCHK(hipEventRecord(m->e1, m->stream));
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
    MHIPCHK(hipSetDevice(m->device));
    const size_t qb = (size_t)npairs * q_stride * 32, tb = (size_t)npairs * t_stride * 32, ob = (size_t)npairs * q_stride * 2 * sizeof(int);
    uint8_t *dq = nullptr, *dt = nullptr; int *dnq = nullptr, *dnt = nullptr, *di = nullptr, *dd = nullptr;
    int rc = ORBM_OK;
    do {
        if (hipMalloc((void**)&dq, qb) != hipSuccess || hipMalloc((void**)&dt, tb) != hipSuccess || hipMalloc((void**)&dnq, sizeof(int) * npairs) != hipSuccess ||
            hipMalloc((void**)&dnt, sizeof(int) * npairs) != hipSuccess || hipMalloc((void**)&di, ob) != hipSuccess || hipMalloc((void**)&dd, ob) != hipSuccess) {
            set_merr("hipMalloc failed"); rc = ORBM_E_HIP; break;
        }
        if (hipMemcpy(dq, q, qb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dt, t, tb, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dnq, nq, sizeof(int) * npairs, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dnt, nt, sizeof(int) * npairs, hipMemcpyHostToDevice) != hipSuccess) { set_merr("H2D failed"); rc = ORBM_E_HIP; break; }
        rc = orbm_knn2_batch_async(m, dq, q_stride, dnq, dt, t_stride, dnt, npairs, t_stride, di, dd);
        if (rc) break;
        rc = orbm_sync(m);
        if (rc) break;
        if (hipMemcpy(idx2, di, ob, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(dist2, dd, ob, hipMemcpyDeviceToHost) != hipSuccess) { set_merr("D2H failed"); rc = ORBM_E_HIP; }
    } while (0);
    void* ps[] = {dq, dt, dnq, dnt, di, dd};
    for (void* p : ps) if (p) (void)hipFree(p);
    return rc;
}

int orbm_last_timing(orbm_t* m, float* ms) {
    if (!m || !m->timed) return ORBM_E_INVALID;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    MHIPCHK(hipEventElapsedTime(ms, m->e0, m->e1));
    return ORBM_OK;
}

}  // extern "C"
