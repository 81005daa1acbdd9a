// graph_ev2.hip -- which precondition does hipEventRecordWithFlags(..., hipEventRecordExternal) have under capture?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); printf("%-70s -> %s\n", #x, hipGetErrorString(e)); } while (0)
__global__ void nop(unsigned* p) { if (threadIdx.x == 9999) p[0] = 1; }
int main() {
    hipStream_t s0, s2; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t fresh, used, f2, dep, dep2; hipEventCreate(&fresh); hipEventCreate(&used); hipEventCreate(&f2);
    hipEventCreateWithFlags(&dep, hipEventDisableTiming); hipEventCreateWithFlags(&dep2, hipEventDisableTiming);
    unsigned* d; hipMalloc(&d, 1 << 20); void* h; hipHostMalloc(&h, 1 << 20, 0);
    hipEventRecord(used, s0); hipStreamSynchronize(s0);
    hipGraph_t g;
    printf("--- case A: fresh event, first op of the capture\n");
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecordWithFlags(fresh, s0, hipEventRecordExternal));
    hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s0, d);
    CK(hipStreamEndCapture(s0, &g));
    printf("--- case B: fork to s2 with a D2H copy first, then external records of a used and a fresh event\n");
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(dep, s0));
    CK(hipStreamWaitEvent(s2, dep, 0));
    CK(hipMemcpyAsync(h, d, 1 << 20, hipMemcpyDeviceToHost, s2));
    CK(hipEventRecord(dep2, s2));
    CK(hipSetDevice(0));
    CK(hipEventRecordWithFlags(used, s0, hipEventRecordExternal));
    CK(hipEventRecordWithFlags(f2, s0, hipEventRecordExternal));
    hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s0, d);
    CK(hipStreamWaitEvent(s0, dep2, 0));
    CK(hipStreamEndCapture(s0, &g));
    return 0;
}
