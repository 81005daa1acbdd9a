// Device -> pinned host copy beside a compute-saturated GPU: hipMemcpyAsync (the runtime's copy kernel) vs hsa_amd_memory_async_copy (SDMA engine).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/hsa_copy tools/ubench/hsa_copy.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
__global__ void spin(float* p, int iters) {
    float a = p[threadIdx.x], b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.9999f + a * 1e-9f; }
    if (a == 12345.f) p[0] = a;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 48u << 20;
    void *dev, *pin; float* scratch;
    CK(hipMalloc(&dev, N)); CK(hipHostMalloc(&pin, N, 0)); CK(hipMalloc((void**)&scratch, 1 << 20));
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hsa_init());
    hsa_amd_pointer_info_t pi{}; pi.size = sizeof pi;
    CK(hsa_amd_pointer_info(dev, &pi, nullptr, nullptr, nullptr));
    const hsa_agent_t gpu = pi.agentOwner;
    hsa_amd_pointer_info_t ph{}; ph.size = sizeof ph;
    CK(hsa_amd_pointer_info(pin, &ph, nullptr, nullptr, nullptr));
    const hsa_agent_t cpu = ph.agentOwner;
    hsa_signal_t sig; CK(hsa_signal_create(1, 0, nullptr, &sig));
    for (int load = 0; load < 2; ++load) {
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                if (load) hipLaunchKernelGGL(spin, dim3(256 * 32), dim3(256), 0, sk, scratch, 400000);   // ~ms of saturated VALU
                const double t0 = now();
                if (mode == 0) { CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc)); }
                else {
                    hsa_signal_store_relaxed(sig, 1);
                    CK(hsa_amd_memory_async_copy(pin, cpu, dev, gpu, N, 0, nullptr, sig));
                    while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
                }
                const double dt = now() - t0;
                if (dt < best) best = dt;
                CK(hipStreamSynchronize(sk));
            }
            printf("%s, %s: %.1f GB/s\n", load ? "GPU busy" : "GPU idle", mode ? "hsa_amd_memory_async_copy" : "hipMemcpyAsync", N / best / 1e9);
        }
    }
    return 0;
}
