// CPU-side unit test of the copy thread's DMA path (orb-slam3_amd/csrc/hsa_copy.h) with stubbed HSA entry points:
// refusal before anything is issued, refusal of a later piece while earlier pieces complete concurrently (the
// signal must still reach zero: un-issued pieces are subtracted atomically, not stored), and a DMA that never
// completes (finite deadline -> TIMEOUT instead of a hung copy thread).  No GPU, no HSA runtime needed.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../orb-slam3_amd/csrc/hsa_copy.h"

static std::atomic<long> g_sig{0};
static int g_unknownAt = -1;                 // pointer-info lookups from this index on report "unknown"
static int g_lookups = 0;
static int g_failCopyAt = -1;                // the copy call with this index is refused
static int g_copies = 0;
static bool g_neverComplete = false;
static bool g_completeLate = false;          // completions arrive from another thread after the issue loop is over
static std::vector<std::thread> g_dma;

static hsa_status_t s_ptrinfo(const void*, hsa_amd_pointer_info_t* info, void* (*)(size_t), uint32_t*, hsa_agent_t**) {
    const int k = g_lookups++ / 2;           // two lookups (src, dst) per piece
    info->type = (g_unknownAt >= 0 && k >= g_unknownAt) ? HSA_EXT_POINTER_TYPE_UNKNOWN : HSA_EXT_POINTER_TYPE_HSA;
    return HSA_STATUS_SUCCESS;
}
static void s_store(hsa_signal_t, hsa_signal_value_t v) { g_sig.store(v); }
static void s_sub(hsa_signal_t, hsa_signal_value_t v) { g_sig.fetch_sub(v); }
static hsa_signal_value_t s_wait(hsa_signal_t, hsa_signal_condition_t, hsa_signal_value_t, uint64_t, hsa_wait_state_t) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    return g_sig.load();
}
static hsa_status_t s_copy(void* dst, hsa_agent_t, const void* src, hsa_agent_t, size_t n, uint32_t, const hsa_signal_t*, hsa_signal_t) {
    const int k = g_copies++;
    if (k == g_failCopyAt) return HSA_STATUS_ERROR;
    if (g_neverComplete) return HSA_STATUS_SUCCESS;
    if (g_completeLate) g_dma.emplace_back([=] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); std::memcpy(dst, src, n); g_sig.fetch_sub(1); });
    else { std::memcpy(dst, src, n); g_sig.fetch_sub(1); }          // completes before the issue loop goes on: the racy case of a plain store
    return HSA_STATUS_SUCCESS;
}

#define CHECK(c) do { if (!(c)) { std::printf("hsa_copy_stub: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    HsaCopy h;
    h.ptrinfo = s_ptrinfo; h.sigstore = s_store; h.sigsub = s_sub; h.sigwait = s_wait; h.copy = s_copy;
    h.timeoutMs = 200.0;
    char a[3][16], b[3][16];
    void* dst[3] = {b[0], b[1], b[2]}; const void* src[3] = {a[0], a[1], a[2]}; size_t nb[3] = {16, 16, 16};
    auto reset = [&] { g_sig = 0; g_unknownAt = g_failCopyAt = -1; g_lookups = g_copies = 0; g_neverComplete = g_completeLate = false;
                       for (int i = 0; i < 3; ++i) { std::memset(a[i], 'A' + i, 16); std::memset(b[i], 0, 16); } };
    reset();
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::DONE && g_copies == 3 && g_sig == 0 && b[2][5] == 'C');
    reset(); g_completeLate = true;
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::DONE && g_sig == 0 && b[0][0] == 'A' && b[1][0] == 'B' && b[2][0] == 'C');
    for (auto& t : g_dma) t.join(); g_dma.clear();
    // an attachment pointer HSA does not know: refused before ANY copy is in flight
    reset(); g_unknownAt = 1;
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::REFUSED && g_copies == 0);
    // piece 1 refused by the copy call after piece 0 was issued and HAS ALREADY COMPLETED: the signal was 3, is 2 after the
    // completion; a store of `issued` (= 1) would leave it at 1 for ever, the subtraction of the 2 un-issued pieces ends at 0
    reset(); g_failCopyAt = 1;
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::PARTIAL && g_sig == 0 && b[0][0] == 'A' && b[1][0] == 0);
    // the same with the completion still in flight when the subtraction happens
    reset(); g_failCopyAt = 2; g_completeLate = true;
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::PARTIAL && g_sig == 0 && b[1][0] == 'B');
    for (auto& t : g_dma) t.join(); g_dma.clear();
    // refused at once: nothing issued, nothing to wait for
    reset(); g_failCopyAt = 0;
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::PARTIAL && g_sig == 0);
    // a DMA that never signals: the wait gives up after the deadline
    reset(); g_neverComplete = true;
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(h.run(dst, src, nb, 3) == HsaCopy::TIMEOUT);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(ms >= 200.0 && ms < 5000.0);
    CHECK(h.run(dst, src, nb, 0) == HsaCopy::DONE && h.run(dst, src, nb, 33) == HsaCopy::REFUSED);
    std::printf("hsa_copy_stub ok\n");
    return 0;
}
