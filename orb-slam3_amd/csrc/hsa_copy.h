// Results-to-host copies through hsa_amd_memory_async_copy (a DMA engine) for the extractor's copy thread.
// hipMemcpyAsync is executed by this runtime as a copy KERNEL, which beside the saturated extraction moves ~19 GB/s -- less
// than the stereo step produces (21 GB/s): C3 213 k -> 231 k frames/s with the DMA engine, results identical.  The HSA runtime
// is the one HIP already loaded (dlopen by soname, no link dependency).  ORBX_DL_HSA=0: A/B switch.
//
// Host-only C++ (function pointers + one completion signal), so that tests/hsa_copy_stub.cpp can drive the failure paths with
// stubbed entry points on a machine without a GPU.
#pragma once
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <dlfcn.h>
#include <chrono>
#include <cstddef>
#include <cstdint>

struct HsaCopy {
    enum Result {
        DONE = 0,        // every piece copied
        REFUSED = 1,     // a pointer HSA does not know (e.g. a pageable destination): NOTHING was issued, the caller's own copy path takes the job
        PARTIAL = 2,     // a piece was refused after earlier ones had been issued; those have completed, the caller copies everything again
        TIMEOUT = 3      // issued copies did not complete within the deadline (a DMA that faulted never signals): an error, not a fallback
    };
    bool ok = false;
    hsa_status_t (*init)() = nullptr;
    hsa_status_t (*ptrinfo)(const void*, hsa_amd_pointer_info_t*, void* (*)(size_t), uint32_t*, hsa_agent_t**) = nullptr;
    hsa_status_t (*sigcreate)(hsa_signal_value_t, uint32_t, const hsa_agent_t*, hsa_signal_t*) = nullptr;
    void (*sigstore)(hsa_signal_t, hsa_signal_value_t) = nullptr;
    void (*sigsub)(hsa_signal_t, hsa_signal_value_t) = nullptr;
    hsa_signal_value_t (*sigwait)(hsa_signal_t, hsa_signal_condition_t, hsa_signal_value_t, uint64_t, hsa_wait_state_t) = nullptr;
    hsa_status_t (*copy)(void*, hsa_agent_t, const void*, hsa_agent_t, size_t, uint32_t, const hsa_signal_t*, hsa_signal_t) = nullptr;
    hsa_status_t (*sigdestroy)(hsa_signal_t) = nullptr;
    hsa_signal_t sig{};
    double timeoutMs = 20000.0;              // ORBX_DL_TIMEOUT_MS
    uint64_t waitSlice = 2000000;            // one blocking wait, in HSA timestamp ticks (the deadline is checked on the host clock between slices)
    static const int MAXP = 32;

    ~HsaCopy() { if (ok && sigdestroy) (void)sigdestroy(sig); }

    bool load() {
        void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        init = (decltype(init))dlsym(h, "hsa_init"); ptrinfo = (decltype(ptrinfo))dlsym(h, "hsa_amd_pointer_info");
        sigcreate = (decltype(sigcreate))dlsym(h, "hsa_signal_create"); sigstore = (decltype(sigstore))dlsym(h, "hsa_signal_store_relaxed");
        sigsub = (decltype(sigsub))dlsym(h, "hsa_signal_subtract_relaxed");
        sigwait = (decltype(sigwait))dlsym(h, "hsa_signal_wait_scacquire"); copy = (decltype(copy))dlsym(h, "hsa_amd_memory_async_copy");
        sigdestroy = (decltype(sigdestroy))dlsym(h, "hsa_signal_destroy");
        if (!init || !ptrinfo || !sigcreate || !sigstore || !sigsub || !sigwait || !copy) return false;
        if (init() != HSA_STATUS_SUCCESS || sigcreate(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return false;
        return ok = true;
    }

    // waits until the signal drops below 1 or the deadline passes
    bool wait_done() {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (sigwait(sig, HSA_SIGNAL_CONDITION_LT, 1, waitSlice, HSA_WAIT_STATE_BLOCKED) < 1) return true;
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeoutMs) return false;
        }
    }

    // n pieces, all device -> pinned host
    Result run(void* const* dst, const void* const* src, const size_t* bytes, int n) {
        if (n < 1) return DONE;
        if (n > MAXP) return REFUSED;
        hsa_agent_t as[MAXP], ad[MAXP];
        // every pointer is looked up BEFORE the first copy is issued: a refusal then leaves nothing in flight
        for (int i = 0; i < n; ++i) {
            hsa_amd_pointer_info_t ps{}, pd{}; ps.size = sizeof ps; pd.size = sizeof pd;
            if (ptrinfo(src[i], &ps, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || ptrinfo(dst[i], &pd, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
                ps.type == HSA_EXT_POINTER_TYPE_UNKNOWN || pd.type == HSA_EXT_POINTER_TYPE_UNKNOWN)
                return REFUSED;
            as[i] = ps.agentOwner; ad[i] = pd.agentOwner;
        }
        sigstore(sig, n);                    // nothing in flight: a plain store cannot race with a completion
        int issued = 0;
        for (; issued < n; ++issued)
            if (copy(dst[issued], ad[issued], src[issued], as[issued], bytes[issued], 0, nullptr, sig) != HSA_STATUS_SUCCESS) break;
        // pieces that were not issued never signal: take them off ATOMICALLY (the issued ones may be completing right now,
        // a store of `issued` could land after their decrements and leave the signal above zero for ever)
        if (issued < n) sigsub(sig, n - issued);
        if (issued > 0 && !wait_done()) return TIMEOUT;
        return issued == n ? DONE : PARTIAL;
    }
};
