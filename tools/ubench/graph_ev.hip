// graph_ev.hip -- does a two-stream fork/join captured into a hipGraph keep (a) its cross-stream overlap and
// (b) timing events recorded with hipEventRecordExternal?  And what does one replay cost the host?
// build: hipcc --offload-arch=gfx950 -O3 -o graph_ev graph_ev.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned* p, int iters) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
    if (v == 0x12345678u) p[0] = v;
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1, t2;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
    unsigned* d; CK(hipMalloc(&d, 64));
    auto enqueue = [&](bool ext) -> int {
        if (ext) CK(hipEventRecordWithFlags(t0, s0, hipEventRecordExternal)); else CK(hipEventRecord(t0, s0));
        CK(hipEventRecord(fork, s0));
        CK(hipStreamWaitEvent(s1, fork, 0));
        for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s1, d, 20000);      // chain on s1 (few CUs)
        if (ext) CK(hipEventRecordWithFlags(t1, s1, hipEventRecordExternal)); else CK(hipEventRecord(t1, s1));
        CK(hipEventRecord(join, s1));
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, d, 40000);      // beside it on s0
        CK(hipStreamWaitEvent(s0, join, 0));
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, d, 1000);
        if (ext) CK(hipEventRecordWithFlags(t2, s0, hipEventRecordExternal)); else CK(hipEventRecord(t2, s0));
        return 0;
    };
    for (int i = 0; i < 3; ++i) if (enqueue(false)) return 1;
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
    float a = 0, b = 0;
    {
        auto h0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 50; ++i) if (enqueue(false)) return 1;
        auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        auto h2 = std::chrono::steady_clock::now();
        CK(hipEventElapsedTime(&a, t0, t1)); CK(hipEventElapsedTime(&b, t0, t2));
        printf("eager : host enqueue %.1f us/step, wall %.1f us/step, events t0->t1 %.1f us, t0->t2 %.1f us\n",
               std::chrono::duration<double, std::micro>(h1 - h0).count() / 50, std::chrono::duration<double, std::micro>(h2 - h0).count() / 50, a * 1e3, b * 1e3);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (enqueue(true)) return 1;
    CK(hipStreamEndCapture(s0, &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    {
        auto h0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s0));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s0));
        auto h2 = std::chrono::steady_clock::now();
        hipError_t e1 = hipEventElapsedTime(&a, t0, t1), e2 = hipEventElapsedTime(&b, t0, t2);
        printf("graph : %zu nodes, host launch %.1f us/step, wall %.1f us/step, events (%s,%s) t0->t1 %.1f us, t0->t2 %.1f us\n", nn,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / 50, std::chrono::duration<double, std::micro>(h2 - h0).count() / 50,
               hipGetErrorString(e1), hipGetErrorString(e2), a * 1e3, b * 1e3);
    }
    return 0;
}
