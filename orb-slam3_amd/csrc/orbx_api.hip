// orbx_api.hip -- C ABI (include/orbx.h) over the gfx950 extractor kernels.
// Host side: replays the reference constructor tables and per-image geometry in the same float arithmetic
// (ORBextractor.cc:468-571, 1038-1106, 1664-1672), owns the device-resident pyramid/scratch, launches the
// pipeline on one HIP stream.  No CPU fallback: every failure surfaces as ORBX_E_*.
#include "../../include/orbx.h"
#include "orbx_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <array>
#include <deque>
#include <map>
#include <dlfcn.h>
#include "hsa_copy.h"
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace orbxk;

static thread_local std::string g_err;
static void set_err(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return ORBX_E_HIP; } } while (0)

static const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

// A/B switches (first-generation kernels, scheduling variants, test knobs) exist only in the -DORBX_AB build (liborbslam3_amd_ab.so, what
// tests/ab runs against); in the product library they read as unset and the kernels behind them are not compiled in.
#ifdef ORBX_AB
static inline const char* ab_env(const char* name) { return getenv(name); }
#define AB_LAUNCH(...) hipLaunchKernelGGL(__VA_ARGS__)
#else
static inline const char* ab_env(const char*) { return nullptr; }
#define AB_LAUNCH(...) ((void)0)                              /* (the flag that guards the call is never set in this build) */
#endif

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

struct orbx {
    int nfeatures, nlevels, iniTh, minTh, device, maxW, maxH, maxBatch;
    double scaleFactor;
    std::vector<float> sf, invsf, sig2, invsig2;
    std::vector<int> nfeat;
    Umax umax;
    // geometry of the current image size
    int curW = 0, curH = 0;
    Geom g;
    std::vector<CellInfo> cells;
    std::vector<BlurTask> tiles;
    std::vector<Blur3Task> tiles3;                            // k_blur3 (matrix-core blur) tasks: B3_CHUNK-tile walks, then one-tile tasks
    size_t nTiles3Walk = 0;                                   // (small batches take the one-tile list: more, shorter workgroups)
    std::vector<u32> b3Th, b3Tv;                              // its weight fragments: [strip][2][64][4], [tile row][64][4]
    BlurSel blurSel;
    std::vector<StripInfo> strips;
    StripInfo* dStrips = nullptr; size_t capStrips = 0;
    std::vector<CellAux> aux; std::vector<F4Item> f4items;    // k_fast4: per-cell scalars and per-shape item tables
    CellAux* dAux = nullptr; size_t capAux = 0;
    F4Item* dF4Items = nullptr; size_t capF4Items = 0;
    bool fastV3 = false;                                      // ORBX_FAST_V3: A/B, k_fast3 instead of k_fast4
    bool fastV1 = false;
    // k_fast3 launch groups: level 0 (needs no resize), the fine levels, the coarse levels.  Each group sizes its own LDS
    // (tile of its tallest cell row + survivor queues), because occupancy -- 20 vs 28 waves per CU -- is worth ~15 %.
    struct F3Group { int strip0 = 0, nstrips = 0, tile = 0, qcap = 0, lastLevel = 0, pitch = 0; size_t lds = 0; bool v4 = false; };   // pitch: 176 / 208 = compile-time tile pitch of the group, 0 = per strip
    hipEvent_t evLvl[12] = {};                                // "pyramid level l is resized" for the levels that end a FAST group
    std::vector<F3Group> f3g;
    std::vector<int> stripTile, stripQ;                       // per strip: tile bytes, worst-case queue entries
    std::vector<RzTab> xt, yt;
    std::vector<RzX4> x4;
    std::vector<RzTask> rzTasks[12];
    RzX4* dX4 = nullptr; RzTask* dRzTasks = nullptr; size_t capX4 = 0, capRzTasks = 0;
    int rzTaskOff[12] = {}; bool rzStream[12] = {};
    int64_t algBytes = 0, fusedBytes = 0;
    size_t qtLds = 0, qt2Lds = 0;
    int qtFuseD = 3;
    int maxIni = 1;                                            // most quadtree roots of any level
    int qt2Cap = 0, qt2Sort = 0;
    bool qtV1 = false, odV1 = false, serial = false, blurV2 = false, blurEarly = true;
    bool blurTiled = true;                                     // k_blur3 writes / k_orient_desc2 reads the blurred levels as 16 x 8-px tiles (ORBX_BLUR_ROWMAJOR: A/B)
    bool qtWide = false;                                       // 1024-thread quadtree workgroups (large frames / feature counts)
    int qtWideForce = -1;                                      // ORBX_QT_WIDE=0/1: A/B switch
    // device
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;
    static const int kRing = 32;                             // timing-event sets kept for orbx_mean_timings
    hipEvent_t evr[kRing][12] = {};
    hipEvent_t* ev = evr[0];
    hipEvent_t evDone = nullptr;
    hipEvent_t evGuard = nullptr; bool guardPending = false;   // orbx_guard_results: a reader of the result block on another stream
    // results -> host: copy stream + "copy may start" / "copy done" events (orbx_result_download_async)
    // results -> host: a ring of result blocks (keypoints, descriptors, counts, mono indices).  A batch writes the CURRENT block
    // (orbx_set_result_block), so the copy of batch i's block to the host runs beside batch i+1, which writes another block.
    // dKps / dDesc / dN / dMono below always alias the current block.
    // The copies are issued by a helper thread: it waits (on the host) for the batch that fills the block, then starts a
    // DEPENDENCY-FREE hipMemcpyAsync on the copy stream.  A copy that the runtime has to order behind unfinished kernels
    // (hipStreamWaitEvent + hipMemcpyAsync) stalls the enqueueing thread for ~7 ms every ten or so calls on the HIP 7.0
    // runtime bench.py runs on (tools/d2h_probe.py), which would starve the GPU whenever it hits the thread that launches
    // the batches; a copy kernel has no such stalls but its PCIe writes slow the bandwidth-bound kernels beside it by 5-20 %.
    // Flow control is host-side: a block is `busy` from orbx_result_download_async until its copy has landed, and whoever is
    // about to rewrite a busy block (orbx_extract_batch_async / orbx_graph_launch) waits for it first.
    static const int kBlocks = 8;                              // 7 batches of slack: a copy call that stalls for several milliseconds never idles the GPU
    hipStream_t stream3 = nullptr;
    hipEvent_t evBatchDone[kBlocks] = {};
    std::thread dlThread; std::mutex dlMu; std::condition_variable dlCv;
    struct DlReq { int block; void* host; };
    struct Attach { const void* dev; size_t bytes, hostOff; };   // caller buffers that travel with a block (orbx_block_attach)
    std::vector<Attach> attach[kBlocks];
    std::deque<DlReq> dlQueue; bool dlBusy[kBlocks] = {}; bool dlStop = false; int dlError = 0; bool dlKernel = false; int dlGrid = 16;
    // A block is ONE allocation -- [kps | desc | counts | monos] at offKps.. -- so that it reaches the host with a single copy
    // (several back-to-back hipMemcpyAsync on one stream block the calling thread for ~0.4 ms each on this runtime).
    struct ResBlock { u8* base = nullptr; KpOut* kps = nullptr; u8* desc = nullptr; int *n = nullptr, *mono = nullptr; } rb[kBlocks];
    size_t offKps = 0, offDesc = 0, offN = 0, offMono = 0, blockBytes = 0;
    int curBlock = 0;
    // HIP-graph replay of an enqueue sequence (orbx_capture_begin / _end / orbx_graph_launch): per slot the instantiated graph
    // and its own timing events (recorded as external event nodes, so a replay refreshes them)
    struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int nimg = 0, block = 0; long launches = 0; };
    unsigned long long* dStamps = nullptr; int wallClockKHz = 0;   // k_stamp slots [kSlots][4]
    static const int kSlots = 8;                               // public graph slots (orbx_capture_begin)
    static const int kAllSlots = kSlots + 1;                   // + the single-frame call's own graph (orbx_extract)
    GraphSlot gs[kAllSlots];
    int oneW = 0, oneH = 0, oneEager = 0;                      // geometry the single-frame graph was captured for; eager calls seen since
    bool oneOff = false;                                       // capture failed once (or ORBX_ONE_GRAPH=0): stay eager
    double oneEagerUs = 0;                                     // enqueue-to-synchronised time of the last eager single-frame call
    int capSlot = -1;                                          // >= 0 while the streams are being captured
    bool capFailed = false;                                    // an enqueue failed inside the open capture
    bool graphMode = false;                                    // the most recent batch came from a graph replay
    hipEvent_t evDepc[12] = {};                                // capture mode: dependency twins of the timing events
    hipEvent_t evMark[2] = {};                                 // orbx_mark: caller-placed time stamps on the extractor's stream
    long nEnq = 0;
    u8 *dPyr = nullptr, *dBlur = nullptr, *dL0 = nullptr;
    const u8** dL0Ptr = nullptr;
    Blur3Task* dTiles3 = nullptr; size_t capTiles3 = 0;
    u32 *dB3Th = nullptr, *dB3Tv = nullptr; size_t capB3Th = 0, capB3Tv = 0;
    CellInfo* dCells = nullptr; BlurTask* dTiles = nullptr; RzTab *dXt = nullptr, *dYt = nullptr;
    u32 *dCandCnt = nullptr, *dCandEnt = nullptr, *dSel = nullptr, *dSelCnt = nullptr;
    u16* dKpNode = nullptr;
    u32* dDense = nullptr;
    int maxCells = 0;
    KpOut* dKps = nullptr; u8* dDesc = nullptr; KpWork* dWork = nullptr;
    int *dN = nullptr, *dMono = nullptr, *dLap = nullptr, *dErr = nullptr;
    bool stageTiming = true;                                   // record the stage-boundary events (orbx_set_stage_timing)
    std::vector<int> hN, hMono; bool countsValid = false;      // per-frame counts of the last batch, fetched once
    u8* hPinned = nullptr; size_t capPinned = 0; hipEvent_t evH2D = nullptr;   // pinned staging of host images
    u8* dIngest = nullptr; size_t capIngest = 0;               // grow-only scratch of the ingest entry points (pointer tables, staged colour images, CLAHE LUTs)
    std::vector<const u8*> upPtr; std::vector<int> upLap;      // what dL0Ptr / dLap currently hold
    u8* hLvl = nullptr; size_t capLvlHost = 0;                // pinned landing buffer of orbx_level_image
    u8* hPyr = nullptr; size_t capPyrHost = 0;                // pinned landing buffer of orbx_pyramid_fetch
    u8* hOne = nullptr; size_t capOne = 0;                    // pinned landing buffer of the single-frame call (k_fetch_one)
    u32 *dOvf = nullptr, *dOvfList = nullptr;                  // k_fast3 queue overflow list -> k_fast_fix
    size_t capOvfList = 0;
    int f3QcapForce = 0;                                       // ORBX_FAST_QCAP: test knob, forces a small queue
    bool f3NoFixedPitch = false;                               // ORBX_FAST_PITCH0: A/B, every group on the per-strip-pitch kernel
    int8_t* dPattern = nullptr;
    u32* dOdW = nullptr;                                       // IC_Angle byte weights (k_orient_desc2)
    size_t capL0 = 0, capPyr = 0, capCells = 0, capTiles = 0, capXt = 0, capYt = 0, capCandCnt = 0, capCandEnt = 0, capSel = 0;
    int lastBatch = 0;
    int l0pitch = 0, lastL0Pitch = 0;
    std::vector<const u8*> hL0Ptr;
    bool l0Staged = false;                                     // the last batch's level 0 was uploaded FROM hPinned (host images): its host copy is still there
    std::vector<int> hLap;
    bool timed = false;
};

extern "C" int orbx_set_result_block(orbx_t* o, int block);
static int dl_drain(orbx* o);
static int dl_wait_block(orbx* o, int b);

template <class T> static int ensure(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(hipMalloc((void**)p, need * sizeof(T)));
    *cap = need;
    return 0;
}

// ---- geometry (replays ORBextractor.cc:1669, 1053-1106, 695-699 in the reference's float arithmetic)
static int build_geometry(orbx* o, int w, int h) {
    if (o->curW == w && o->curH == h) return 0;
    if (w > o->maxW || h > o->maxH) { set_err("image %dx%d exceeds the configured maximum %dx%d", w, h, o->maxW, o->maxH); return ORBX_E_CAPACITY; }
    if (o->capSlot >= 0) { set_err("the image size must not change while a graph is being captured"); return ORBX_E_INVALID; }
    // Everything below tears the current geometry down before it can fail (too small / unsupported / out of memory): mark the
    // handle as having NO geometry first, so that a failed rebuild can never be mistaken for a valid one by the next call
    // with the previous size, and nothing refers to freed result buffers.
    o->curW = o->curH = 0; o->lastBatch = 0; o->countsValid = false; o->timed = false; o->graphMode = false;
    if (o->stream) { (void)hipStreamSynchronize(o->stream); (void)hipStreamSynchronize(o->stream2); (void)dl_drain(o); }
    for (auto& G : o->gs) {                                    // graphs captured for the old geometry hold its pointers and grids
        if (G.exec) (void)hipGraphExecDestroy(G.exec);
        if (G.graph) (void)hipGraphDestroy(G.graph);
        G.exec = nullptr; G.graph = nullptr; G.launches = 0; G.nimg = 0;
    }
    o->oneW = o->oneH = 0; o->oneEager = 0;
    o->upPtr.clear(); o->upLap.clear();
    for (auto& v : o->attach) v.clear();                       // the block layout is about to change
    Geom& g = o->g;
    memset(&g, 0, sizeof g);
    const int L = o->nlevels;
    g.nlevels = L; g.w0 = w; g.h0 = h; g.iniTh = o->iniTh; g.minTh = o->minTh; g.lowTh = std::min(o->iniTh, o->minTh);
    std::vector<Blur3Task> tiles3one;                         // one-tile k_blur3 tasks, appended behind the walks
    o->cells.clear(); o->tiles.clear(); o->tiles3.clear(); o->b3Th.clear(); o->b3Tv.clear(); o->strips.clear(); o->f3g.clear(); o->stripTile.clear(); o->stripQ.clear(); o->xt.clear(); o->yt.clear(); o->x4.clear(); for (auto& v : o->rzTasks) v.clear();
    size_t off = 0, boff = 0;
    int totalSlots = 0, totalSel = 0, maxN = 0;
    int64_t sumAll = 0, sumSrc = 0, sumDst = 0;
    for (int l = 0; l < L; ++l) {
        LevelDesc& D = g.lv[l];
        const float scale = o->invsf[l];
        D.w = cv_round_f((float)w * scale); D.h = cv_round_f((float)h * scale);
        if (D.w > 4095 || D.h > 4095) { set_err("level %d is %dx%d: packed candidates hold 12-bit coordinates", l, D.w, D.h); return ORBX_E_UNSUPPORTED; }
        D.pitch = align_up(D.w, 64);
        D.off = (int)off;
        off += (size_t)align_up(D.pitch * D.h, 256);
        // blurred copy: row-major twin of the level, or 16 x 8-px tiles (one more tile per tile row than the pitch needs: the
        // descriptor kernel's 48-byte patch rows may start up to 10 bytes before the right edge)
        D.btpr = D.pitch / 16 + 1;
        D.boff = o->blurTiled ? (int)boff : D.off;
        boff += (size_t)align_up(D.btpr * ((D.h + 7) / 8) * 128, 256);
        D.sf = o->sf[l];
        D.patch = (float)(int)(31 * o->sf[l]);
        D.N = o->nfeat[l];
        sumAll += (int64_t)D.w * D.h;
        if (l < L - 1) sumSrc += (int64_t)D.w * D.h;
        if (l > 0) sumDst += (int64_t)D.w * D.h;
        // FAST cell grid
        const int minB = 16, maxBX = D.w - 16, maxBY = D.h - 16;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const float W = 35;
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) { set_err("level %d (%dx%d) is smaller than one FAST cell", l, D.w, D.h); return ORBX_E_TOO_SMALL; }
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        D.cellBase = (int)o->cells.size();
        D.slotBase = totalSlots;
        D.slotCap = ((wCell + 1) / 2) * ((hCell + 1) / 2);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minB + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minB + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                CellInfo c;
                c.level = (short)l; c.x0 = (short)iniX; c.y0 = (short)iniY;
                c.cw = (short)((int)maxX - (int)iniX); c.ch = (short)((int)maxY - (int)iniY);
                c.addx = (short)(j * wCell); c.addy = (short)(i * hCell); c.pad = 0;
                c.cnt = (int)o->cells.size();
                c.slot = totalSlots;
                if ((int)c.cw * c.ch > ORBX_FAST_TILE) { set_err("FAST cell %dx%d exceeds the LDS tile", c.cw, c.ch); return ORBX_E_UNSUPPORTED; }
                totalSlots += D.slotCap;
                o->cells.push_back(c);
            }
        }
        D.nCells = (int)o->cells.size() - D.cellBase;
        // strips for k_fast3: runs of <= 8 cells of one cell-row (one wavefront per cell); the tile pitch is a
        // multiple of 16 bytes (<= 512) and starts >= 4 bytes left of the strip at a 16-byte aligned column
        for (int ci = D.cellBase; ci < (int)o->cells.size();) {
            const CellInfo& f = o->cells[ci];
            const int Hs = f.ch;
            if (Hs > 127) { set_err("FAST cell of %d rows exceeds the strip kernel's 127", Hs); return ORBX_E_UNSUPPORTED; }
            const int xal = (f.x0 - 4) & ~15;
            int n = 0, needed = 0;
            while (ci + n < (int)o->cells.size() && n < F3_NT / 64) {
                const CellInfo& c = o->cells[ci + n];
                if (c.y0 != f.y0) break;
                const int nd = c.x0 + c.cw - 3 - xal + 8;
                if (nd > 512) break;
                needed = nd; ++n;
            }
            if (n == 0) { set_err("FAST cell %dx%d does not fit the strip tile", f.cw, f.ch); return ORBX_E_UNSUPPORTED; }
            const CellInfo& last = o->cells[ci + n - 1];
            StripInfo st;
            st.level = (short)l; st.ncell = (short)n; st.x0 = f.x0; st.y0 = f.y0;
            st.w = (short)(last.x0 + last.cw - f.x0); st.h = (short)Hs; st.xal = (short)xal;
            st.lp = (short)align_up(needed, 16); st.cell0 = ci;
            int qworst = 64;
            for (int k = 0; k < n; ++k) {
                const CellInfo& c = o->cells[ci + k];
                qworst = std::max(qworst, align_up(std::max(0, c.cw - 6) * std::max(0, c.ch - 6), 64));
            }
            o->stripTile.push_back((int)st.lp * Hs); o->stripQ.push_back(qworst);
            o->strips.push_back(st);
            ci += n;
        }
        if (totalSlots - D.slotBase >= (1 << 24)) { set_err("level %d has too many candidate slots", l); return ORBX_E_UNSUPPORTED; }
        // quadtree roots (ORBextractor.cc:695-699)
        D.qtW = maxBX - minB; D.qtH = maxBY - minB;
        D.nIni = (int)std::round(static_cast<float>(D.qtW) / D.qtH);
        if (D.nIni < 1 || D.nIni > 16) { set_err("aspect ratio of level %d unsupported (nIni=%d; the reference divides by zero for nIni=0)", l, D.nIni); return ORBX_E_UNSUPPORTED; }
        D.hX = static_cast<float>(D.qtW) / D.nIni;
        D.selBase = totalSel;
        D.selCap = std::max(D.N + 3, 4 * D.nIni) + 1;
        totalSel += D.selCap;
        maxN = std::max(maxN, std::max(D.N, 4 * D.nIni));
        // blur tasks: one wavefront per (248-px column strip, BL_R-row block); right-edge reflect-101 selectors
        {   // strips of output dwords: the first stores from lane 0 (63 dwords), the others from lane 1 (62), and a strip whose
            // lane 63 is the image's last dword stores that one too (k_blur2's doStore)
            const int gl = (D.w - 1) >> 2;
            for (int ty = 0; ty < D.h; ty += BL_R)
                for (int g0 = 0; g0 <= gl;) {
                    o->tiles.push_back(BlurTask{(short)l, (short)g0, (short)ty, 0});
                    const int lead = g0 > 0 ? 1 : 0;
                    int lastOut = g0 - lead + 62;
                    if (g0 - lead + 63 == gl) lastOut = gl;
                    g0 = lastOut + 1;
                }
        }
        {   // k_blur3: one wavefront per 32-px column strip and chunk of up to B3_CHUNK 26-row tiles; weight fragments per
            // strip (pass 1, reflect-101 folded at the left / right edge) and per tile row (pass 2, folded at top / bottom)
            const int ntile = (D.h + B3_ROWS - 1) / B3_ROWS, nstrip = (D.w + 31) / 32;
            const int th0 = (int)(o->b3Th.size() / 512), tv0 = (int)(o->b3Tv.size() / 256);
            o->b3Th.resize(o->b3Th.size() + (size_t)nstrip * 512);
            o->b3Tv.resize(o->b3Tv.size() + (size_t)ntile * 256);
            for (int sx = 0; sx < nstrip; ++sx) b3_build_th(sx * 32, D.w, o->b3Th.data() + (size_t)(th0 + sx) * 512);
            for (int t = 0; t < ntile; ++t) b3_build_tv(t * B3_ROWS, D.h, o->b3Tv.data() + (size_t)(tv0 + t) * 256);
            for (int t0 = 0; t0 < ntile; t0 += B3_CHUNK)
                for (int sx = 0; sx < nstrip; sx += 4)                  // a workgroup = four adjacent strips
                    o->tiles3.push_back(Blur3Task{(short)l, (short)(sx * 32), (short)t0, (short)std::min(B3_CHUNK, ntile - t0), th0 + sx, tv0 + t0});
            for (int t0 = 0; t0 < ntile; ++t0)
                for (int sx = 0; sx < nstrip; sx += 4)
                    tiles3one.push_back(Blur3Task{(short)l, (short)(sx * 32), (short)t0, 1, th0 + sx, tv0 + t0});
        }
        {
            static const u32 kSelB[4] = {0x05060700u, 0x07000100u, 0x01020100u, 0x03020100u};   // k = (w-1)&3 valid bytes-1
            static const u32 kSelC[4] = {0x04040404u, 0x04040506u, 0x05060700u, 0x07000102u};
            const int k = (D.w - 1) & 3;
            o->blurSel.selB[l] = kSelB[k]; o->blurSel.selC[l] = kSelC[k];
        }
        // resize taps from level l-1
        D.rzx = (int)o->xt.size(); D.rzy = (int)o->yt.size();
        if (l > 0) {
            const LevelDesc& S = g.lv[l - 1];
            const double scale_x = 1.0 / ((double)D.w / S.w), scale_y = 1.0 / ((double)D.h / S.h);
            auto sat = [](float v) { int r = (int)lrintf(v); return (short)std::min(32767, std::max(-32768, r)); };
            for (int dx = 0; dx < D.w; ++dx) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floorf(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
                o->xt.push_back(RzTab{sx, sat((1.f - fx) * 2048.f), sat(fx * 2048.f)});
            }
            for (int dy = 0; dy < D.h; ++dy) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floorf(fy);
                fy -= sy;
                o->yt.push_back(RzTab{sy, sat((1.f - fy) * 2048.f), sat(fy * 2048.f)});
            }
            // per-dword tables of the streaming kernel: it loads 8 bytes from the first tap, so every tap pair must start
            // within 6 bytes of it (true for scale factors up to 1.5; otherwise the level uses k_resize)
            while (o->x4.size() * 4 < (size_t)D.rzx) o->x4.push_back(RzX4{});
            bool ok = (D.rzx % 4) == 0 && S.w >= 8;
            for (int gx = 0; gx * 4 < D.w && ok; ++gx) {
                RzX4 e{};
                int sx[4];
                for (int i = 0; i < 4; ++i) {
                    const int dx = std::min(gx * 4 + i, D.w - 1);
                    const RzTab& tb = o->xt[D.rzx + dx];
                    if (tb.s < 0 || tb.a0 < 0 || tb.a1 < 0) { ok = false; break; }
                    if (tb.s + 1 < S.w) {
                        sx[i] = tb.s;
                        e.a[i] = (u32)(unsigned short)(2 * tb.a0) | ((u32)(unsigned short)(2 * tb.a1) << 16);   // doubled taps (<= 4096): k_resize2 keeps (H >> 4) << 5
                    } else {                                           // clamped last column (a1 == 0): pair (w-2, w-1), weight on the second
                        sx[i] = tb.s - 1;
                        e.a[i] = (u32)(unsigned short)(2 * tb.a0) << 16;
                    }
                }
                if (!ok) break;
                e.bs = *std::min_element(sx, sx + 4);
                for (int i = 0; i < 4; ++i) {
                    const int off = sx[i] - e.bs;
                    if (off < 0 || off > 6) { ok = false; break; }
                    e.o[i] = (u8)off;
                }
                o->x4.push_back(e);
            }
            // rows: strictly increasing source rows without clamping, and a bounded source span per task
            for (int dy = 0; dy < D.h && ok; ++dy) {
                const int sy = o->yt[D.rzy + dy].s;
                if (sy < 0 || sy + 1 >= S.h || (dy > 0 && sy <= o->yt[D.rzy + dy - 1].s)) ok = false;
            }
            for (int ty = 0; ty < D.h && ok; ty += RZ_R) {
                const int ye = std::min(ty + RZ_R, D.h);
                if (o->yt[D.rzy + ye - 1].s + 2 - o->yt[D.rzy + ty].s > RZ_SRC) ok = false;
            }
            o->rzStream[l] = ok;
            if (ok)
                for (int ty = 0; ty < D.h; ty += RZ_R)
                    for (int gx = 0; gx * 4 < D.w; gx += 64) o->rzTasks[l].push_back(RzTask{(short)l, (short)gx, (short)ty, 0});
        }
        while (o->xt.size() % 4) o->xt.push_back(RzTab{0, 0, 0});      // keep every level's tap offset a multiple of 4
    }
    g.pyrFrameBytes = off;
    g.blurTiled = o->blurTiled ? 1 : 0;
    g.blrFrameBytes = o->blurTiled ? boff : off;
    g.totalCells = (int)o->cells.size();
    g.totalSlots = totalSlots;
    g.totalSel = totalSel;
    g.kpCap = totalSel;
    g.nodeCap = 2 * maxN + 64;
    int sc = 1; while (sc < maxN + 8) sc <<= 1;
    g.sortCap = sc;
    if (g.nodeCap > 65000) { set_err("nfeatures per level %d too large for 16-bit node ids", maxN); return ORBX_E_UNSUPPORTED; }
    o->qtLds = (size_t)g.sortCap * 8 + (size_t)g.nodeCap * 56 + 64;
    {   // k_quadtree2: list capacity max(N+3, 4*nIni) (+slack), power-of-two sort buffer
        o->qt2Cap = maxN + 8;
        int sc2 = 1; while (sc2 < o->qt2Cap) sc2 <<= 1;
        o->qt2Sort = sc2;
        o->maxCells = 0;
        for (int l = 0; l < L; ++l) o->maxCells = std::max(o->maxCells, g.lv[l].nCells);
        o->qtWide = o->qtWideForce >= 0 ? o->qtWideForce != 0 : maxN >= 512;   // many features per level: 1024-thread workgroups
        o->maxIni = 1;
        for (int l = 0; l < L; ++l) o->maxIni = std::max(o->maxIni, g.lv[l].nIni);
        // + the fused first iterations' histograms (4 + 16 + 64 (+ 256) words per root), path table and per-node paths.  A fourth fused
        // iteration only happens when some level wants more than ~512 nodes (128 + 3 * 128 <= N), and 12 path bits hold 16 roots * 4^4 / 16
        o->qtFuseD = (maxN >= 512 && o->maxIni <= 4) ? 4 : 3;
        const size_t hw = o->qtFuseD == 4 ? 340 : 84, tw = o->qtFuseD == 4 ? 256 : 64;
        o->qt2Lds = (size_t)sc2 * 8 + (size_t)o->qt2Cap * (16 * 2 + 16 + 4 + 4 + 8 + 2 + 1 + 1) + (size_t)(o->maxCells + 1) * 4 + 64 +
                    (size_t)o->maxIni * (hw * 4 + tw * 2) + (size_t)o->qt2Cap * 4 + 8;
        if (o->qt2Lds > 160 * 1024 - 512) { set_err("quadtree for %d features/level needs %zu B of LDS (> 160 KiB)", maxN, o->qt2Lds); return ORBX_E_UNSUPPORTED; }
    }
    if (o->qtLds > 160 * 1024 - 512) { set_err("quadtree for %d features/level needs %zu B of LDS (> 160 KiB)", maxN, o->qtLds); return ORBX_E_UNSUPPORTED; }
    o->algBytes = sumSrc + sumDst + sumAll;      // read L0..L6 + write L1..L7 + read L0..L7
    o->fusedBytes = sumSrc + sumDst;
    o->l0pitch = align_up(w, 64);

    const size_t B = (size_t)o->maxBatch;
    if (ensure(&o->dPyr, &o->capPyr, off * B)) return ORBX_E_HIP;
    { size_t c2 = 0; u8* old = o->dBlur; if (old) (void)hipFree(old); o->dBlur = nullptr; if (ensure(&o->dBlur, &c2, g.blrFrameBytes * B)) return ORBX_E_HIP; }
    if (ensure(&o->dL0, &o->capL0, (size_t)o->l0pitch * h * B)) return ORBX_E_HIP;
    if (ensure(&o->dCells, &o->capCells, o->cells.size())) return ORBX_E_HIP;
    if (ensure(&o->dTiles, &o->capTiles, o->tiles.size())) return ORBX_E_HIP;
    o->nTiles3Walk = o->tiles3.size();
    o->tiles3.insert(o->tiles3.end(), tiles3one.begin(), tiles3one.end());
    if (ensure(&o->dTiles3, &o->capTiles3, o->tiles3.size())) return ORBX_E_HIP;
    if (ensure(&o->dB3Th, &o->capB3Th, o->b3Th.size())) return ORBX_E_HIP;
    if (ensure(&o->dB3Tv, &o->capB3Tv, o->b3Tv.size())) return ORBX_E_HIP;
    if (ensure(&o->dStrips, &o->capStrips, o->strips.size())) return ORBX_E_HIP;
    if (ensure(&o->dXt, &o->capXt, std::max<size_t>(1, o->xt.size()))) return ORBX_E_HIP;
    if (ensure(&o->dYt, &o->capYt, std::max<size_t>(1, o->yt.size()))) return ORBX_E_HIP;
    {
        std::vector<RzTask> all;
        for (int l = 0; l < L; ++l) { o->rzTaskOff[l] = (int)all.size(); all.insert(all.end(), o->rzTasks[l].begin(), o->rzTasks[l].end()); }
        if (ensure(&o->dX4, &o->capX4, std::max<size_t>(1, o->x4.size()))) return ORBX_E_HIP;
        if (ensure(&o->dRzTasks, &o->capRzTasks, std::max<size_t>(1, all.size()))) return ORBX_E_HIP;
        if (!o->x4.empty()) HIPCHK(hipMemcpy(o->dX4, o->x4.data(), o->x4.size() * sizeof(RzX4), hipMemcpyHostToDevice));
        if (!all.empty()) HIPCHK(hipMemcpy(o->dRzTasks, all.data(), all.size() * sizeof(RzTask), hipMemcpyHostToDevice));
    }
    if (ensure(&o->dCandCnt, &o->capCandCnt, (size_t)g.totalCells * B)) return ORBX_E_HIP;
    HIPCHK(hipMemset(o->dCandCnt, 0, sizeof(u32) * (size_t)g.totalCells * B));   // a cell no kernel has written yet is an EMPTY cell, not whatever the allocation held
    { size_t c2 = 0; if (o->dKpNode) (void)hipFree(o->dKpNode); o->dKpNode = nullptr; if (ensure(&o->dKpNode, &c2, (size_t)g.totalSlots * B)) return ORBX_E_HIP; }
    if (ensure(&o->dCandEnt, &o->capCandEnt, (size_t)g.totalSlots * B)) return ORBX_E_HIP;
    { size_t c2 = 0; if (o->dDense) (void)hipFree(o->dDense); o->dDense = nullptr; if (ensure(&o->dDense, &c2, (size_t)g.totalSlots * B)) return ORBX_E_HIP; }
    if (ensure(&o->dSel, &o->capSel, (size_t)g.totalSel * B)) return ORBX_E_HIP;
    {
        for (auto& R : o->rb) { if (R.base) (void)hipFree(R.base); R = orbx::ResBlock(); }
        if (o->dWork) (void)hipFree(o->dWork);
        o->dKps = nullptr; o->dDesc = nullptr; o->dN = nullptr; o->dMono = nullptr; o->dWork = nullptr;
        o->offKps = 0;
        o->offDesc = (sizeof(KpOut) * g.kpCap * B + 255) & ~(size_t)255;
        o->offN = o->offDesc + (((size_t)32 * g.kpCap * B + 255) & ~(size_t)255);
        o->offMono = o->offN + ((sizeof(int) * B + 255) & ~(size_t)255);
        o->blockBytes = o->offMono + ((sizeof(int) * B + 255) & ~(size_t)255);
        for (auto& R : o->rb) {
            HIPCHK(hipMalloc((void**)&R.base, o->blockBytes));
            HIPCHK(hipMemset(R.base, 0, o->blockBytes));
            R.kps = (KpOut*)(R.base + o->offKps); R.desc = R.base + o->offDesc; R.n = (int*)(R.base + o->offN); R.mono = (int*)(R.base + o->offMono);
        }
        (void)orbx_set_result_block(o, o->curBlock);
        HIPCHK(hipMalloc((void**)&o->dWork, sizeof(KpWork) * g.kpCap * B));
    }
    HIPCHK(hipMemcpy(o->dCells, o->cells.data(), o->cells.size() * sizeof(CellInfo), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(o->dTiles, o->tiles.data(), o->tiles.size() * sizeof(BlurTask), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(o->dTiles3, o->tiles3.data(), o->tiles3.size() * sizeof(Blur3Task), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(o->dB3Th, o->b3Th.data(), o->b3Th.size() * sizeof(u32), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(o->dB3Tv, o->b3Tv.data(), o->b3Tv.size() * sizeof(u32), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(o->dStrips, o->strips.data(), o->strips.size() * sizeof(StripInfo), hipMemcpyHostToDevice));
    if (!o->xt.empty()) HIPCHK(hipMemcpy(o->dXt, o->xt.data(), o->xt.size() * sizeof(RzTab), hipMemcpyHostToDevice));
    if (!o->yt.empty()) HIPCHK(hipMemcpy(o->dYt, o->yt.data(), o->yt.size() * sizeof(RzTab), hipMemcpyHostToDevice));
    {   // FAST launch groups.  Boundaries: [level 0] [levels 1..k] [levels k+1..], k = the smallest level after which at most
        // 15 % of the cells remain: the coarse levels have few cells, but the tallest cell rows (an image 102 rows high is
        // cut into 2 cell rows of 51) and the densest corners, so they keep the worst-case queue.  The other groups bound the
        // queue so that 7 workgroups fit a CU; a cell that overflows it is redone by k_fast_fix.
        const int nS = (int)o->strips.size();
        std::vector<int> cellsAfter(g.nlevels + 1, 0);
        for (int l = g.nlevels - 1; l >= 0; --l) cellsAfter[l] = cellsAfter[l + 1] + g.lv[l].nCells;
        int k = g.nlevels - 1;
        while (k > 0 && cellsAfter[k] * 100 <= 15 * g.totalCells) --k;      // levels > k hold <= 15 % of the cells
        int b0 = 0, b1 = 0;
        for (const StripInfo& stp : o->strips) { if (stp.level == 0) ++b0; if (stp.level <= k) ++b1; }
        const int bounds[4] = {0, b0, std::max(b0, b1), nS};
        size_t maxLds = 0;
        for (int gi = 0; gi < 3; ++gi) {
            orbx::F3Group G;
            G.strip0 = bounds[gi]; G.nstrips = bounds[gi + 1] - bounds[gi];
            if (G.nstrips <= 0) continue;
            int qworst = 64, maxLp = 0, maxH = 0;
            for (int si = G.strip0; si < G.strip0 + G.nstrips; ++si) {
                G.tile = std::max(G.tile, o->stripTile[si]); qworst = std::max(qworst, o->stripQ[si]);
                maxLp = std::max(maxLp, (int)o->strips[si].lp); maxH = std::max(maxH, (int)o->strips[si].h);
            }
            // one compile-time tile pitch for the whole group when its strips allow it (35..40-px cells, 4 per strip: <= 208 bytes)
            G.pitch = o->f3NoFixedPitch ? 0 : maxLp <= 176 ? 176 : maxLp <= 208 ? 208 : 0;
            if (G.pitch) G.tile = G.pitch * maxH;
            G.tile = align_up(G.tile, 16);
            G.lastLevel = o->strips[G.strip0 + G.nstrips - 1].level;
            G.qcap = qworst;
            if (gi < 2) {
                int wgs = 7;
                if (const char* e = ab_env("ORBX_FAST_WGS")) wgs = std::max(1, atoi(e));
                const int budget = (160 * 1024 / wgs - 2 * G.tile) / (2 * (F3_NT / 64));
                G.qcap = std::min(qworst, std::max(wgs > 7 ? 256 : 512, budget / 64 * 64));
            } else {
                // coarse levels: dense corners (over half of a cell's pixels survive on the synthetic stream), so only a mild
                // bound -- 5 workgroups per CU instead of 4 -- and only if the queue still holds >= 3/4 of the worst case
                const int budget = (160 * 1024 / 5 - 2 * G.tile) / (2 * (F3_NT / 64));
                const int q5 = budget / 64 * 64;
                if (q5 < qworst && q5 * 4 >= qworst * 3) G.qcap = q5;
            }
            if (o->f3QcapForce > 0) G.qcap = std::min(G.qcap, align_up(o->f3QcapForce, 64));
            G.lds = (size_t)2 * G.tile + (size_t)(F3_NT / 64) * G.qcap * 2;
            if (const char* e = ab_env("ORBX_FAST_LDSPAD")) G.lds += (size_t)atoi(e);   // experiment: occupancy sensitivity
            if (G.lds > 160 * 1024 - 256) { set_err("FAST strip needs %zu B of LDS", G.lds); return ORBX_E_UNSUPPORTED; }
            maxLds = std::max(maxLds, G.lds);
            o->f3g.push_back(G);
        }
        // k_fast4 tables: per cell the scalars a wave needs, per distinct cell shape (pitch, window width, window height, column
        // phase) the lane -> (tile bytes, valid pixels, queue-entry base) list of every quick-pass iteration
        o->aux.assign(o->cells.size(), CellAux{});
        o->f4items.clear();
        std::map<std::array<int, 4>, u32> shapes;
        for (orbx::F3Group& G : o->f3g) {
            G.v4 = !o->fastV3;
            for (int si = G.strip0; si < G.strip0 + G.nstrips && G.v4; ++si) {
                const StripInfo& stp = o->strips[si];
                const int Pb = G.pitch ? G.pitch : (int)stp.lp;
                for (int k = 0; k < stp.ncell && G.v4; ++k) {
                    const CellInfo& c = o->cells[stp.cell0 + k];
                    CellAux& A = o->aux[stp.cell0 + k];
                    A.slot = c.slot; A.cnt = c.cnt;
                    const int cx0 = c.x0 + 3 - stp.xal, vw = c.cw - 6, vh = stp.h - 6;
                    if (vw <= 0 || vh <= 0) { A.nit = 0; continue; }
                    const int a = cx0 & 3, s4 = cx0 - a, ncol = (a + vw + 7) / 8, nitems = vh * ncol, nit = (nitems + 63) / 64;
                    // limits of the 16-bit fields (entry: 7-bit row, 7-bit xs; item: 16-bit tile offset); beyond them the group stays on k_fast3
                    if (ncol > 16 || vh > 127 || (vh - 1) * Pb + 8 * (ncol - 1) > 65535 || 3 * Pb + s4 > 65535 || s4 < 4) { G.v4 = false; break; }
                    const std::array<int, 4> key = {Pb, vw, vh, a};
                    auto itS = shapes.find(key);
                    if (itS == shapes.end()) {
                        const u32 t0 = (u32)o->f4items.size();
                        o->f4items.resize(t0 + (size_t)nit * 64);
                        for (int i = 0; i < nit * 64; ++i) {
                            F4Item& I = o->f4items[t0 + i];
                            I.mask = 0; I.offq = 0;
                            if (i >= nitems) continue;                  // idle lane of the last iteration: reads (row 0, column 0), keeps nothing
                            const int row = i / ncol, col = i % ncol;
                            for (int px = 0; px < 8; ++px) { const int xs = 8 * col + px; if (xs >= a && xs < a + vw) I.mask |= 1u << (4 * px + 3); }
                            I.offq = (u32)(row * Pb + 8 * col) | ((u32)((row << 9) | (col << 5)) << 16);
                        }
                        itS = shapes.emplace(key, t0).first;
                    }
                    A.tab = itS->second;
                    A.base = (u16)(3 * Pb + s4); A.nit = (u16)nit;
                    A.outx = (short)(c.x0 + 3 - 16 - a); A.outy = (short)(stp.y0 + 3 - 16);
                    A.xlo = (u16)a; A.xhi = (u16)(a + vw - 1);
                }
            }
        }
        if (o->f4items.empty()) o->f4items.push_back(F4Item{0, 0});
        if (ensure(&o->dAux, &o->capAux, o->aux.size())) return ORBX_E_HIP;
        if (ensure(&o->dF4Items, &o->capF4Items, o->f4items.size())) return ORBX_E_HIP;
        HIPCHK(hipMemcpy(o->dAux, o->aux.data(), o->aux.size() * sizeof(CellAux), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(o->dF4Items, o->f4items.data(), o->f4items.size() * sizeof(F4Item), hipMemcpyHostToDevice));
        HIPCHK(hipFuncSetAttribute((const void*)k_fast4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
        HIPCHK(hipFuncSetAttribute((const void*)k_fast4<176>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
        HIPCHK(hipFuncSetAttribute((const void*)k_fast4<208>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
#ifdef ORBX_AB
        HIPCHK(hipFuncSetAttribute((const void*)k_fast3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
        HIPCHK(hipFuncSetAttribute((const void*)k_fast3<176>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
        HIPCHK(hipFuncSetAttribute((const void*)k_fast3<208>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds));
#endif
    }
    if (ensure(&o->dOvfList, &o->capOvfList, (size_t)g.totalCells * B)) return ORBX_E_HIP;
#ifdef ORBX_AB
    HIPCHK(hipFuncSetAttribute((const void*)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o->qtLds));
#endif
    HIPCHK(hipFuncSetAttribute((const void*)k_quadtree2<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o->qt2Lds));
    HIPCHK(hipFuncSetAttribute((const void*)k_quadtree2<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o->qt2Lds));
    o->curW = w; o->curH = h;
    return 0;
}

// stage-boundary events exist only for the per-stage timings; the dependency / span events (0, 2, 6, 8, 9) are always recorded
// Eager mode: one HIP event serves as time stamp and as dependency.  Under stream capture an event record is only a dependency
// marker -- nothing is stamped on replay (external event-record nodes would be, but the HIP 7.0 runtime that PyTorch ships and
// that bench.py therefore runs on rejects hipEventRecordExternal) -- so a captured sequence marks the four span boundaries
// (start, end of FAST, end of blur, end of batch) with k_stamp and takes its dependencies from twin events.
static hipError_t rec_ev(orbx* o, int i, hipStream_t s, bool dep) {
    if (o->capSlot < 0) return hipEventRecord(o->ev[i], s);
    const int k = i == 0 ? 0 : i == 2 ? 1 : i == 9 ? 2 : i == 6 ? 3 : -1;
    if (k >= 0) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, s, o->dStamps + (size_t)o->capSlot * 4 + k);
    if (!dep) return hipSuccess;
    return hipEventRecord(o->evDepc[i], s);
}
static inline hipEvent_t dep_ev(orbx* o, int i) { return o->capSlot < 0 ? o->ev[i] : o->evDepc[i]; }
#define STAGE_EV(i, stream) do { if (o->stageTiming && o->capSlot < 0) HIPCHK(hipEventRecord(o->ev[i], stream)); } while (0)

// ---- helper thread of the results-to-host copies (see struct orbx)
// ---- The results-to-host copy goes through hsa_amd_memory_async_copy (a DMA engine): hsa_copy.h.  If the HSA runtime cannot be had, or a
// destination is not pinned memory it knows, the copy falls back to hipMemcpyAsync.
static void dl_worker(orbx* o) {
    (void)hipSetDevice(o->device);
    HsaCopy hsa;
    bool useHsa = !(getenv("ORBX_DL_HSA") && atoi(getenv("ORBX_DL_HSA")) == 0) && !o->dlKernel && hsa.load();
    if (getenv("ORBX_DL_TIMEOUT_MS")) hsa.timeoutMs = atof(getenv("ORBX_DL_TIMEOUT_MS"));
    for (;;) {
        orbx::DlReq r;
        {
            std::unique_lock<std::mutex> lk(o->dlMu);
            o->dlCv.wait(lk, [&] { return o->dlStop || !o->dlQueue.empty(); });
            if (o->dlQueue.empty()) return;                      // dlStop and nothing left to do
            r = o->dlQueue.front();
        }
        hipError_t e = hipEventSynchronize(o->evBatchDone[r.block]);      // host-side wait: the copy below has no device-side dependency
        bool done = false;
        if (e == hipSuccess && useHsa && o->attach[r.block].size() < 31) {
            void* dst[32]; const void* src[32]; size_t nb[32];
            int n = 0;
            dst[n] = r.host; src[n] = o->rb[r.block].base; nb[n] = o->blockBytes; ++n;
            for (const orbx::Attach& a : o->attach[r.block]) { dst[n] = (u8*)r.host + a.hostOff; src[n] = a.dev; nb[n] = a.bytes; ++n; }
            const HsaCopy::Result r_ = hsa.run(dst, src, nb, n);
            done = r_ == HsaCopy::DONE;
            if (r_ == HsaCopy::TIMEOUT) e = hipErrorLaunchTimeOut;   // issued DMA never completed: ORBX_E_HIP at the next orbx_sync / block wait, no silent fallback
            if (!done) useHsa = false;                           // e.g. a pageable destination: the runtime's own copy handles it (all pieces again)
        }
        if (e == hipSuccess && !done) {
            if (o->dlKernel) {
                const size_t n16 = o->blockBytes / 16;               // blockBytes is a multiple of 256
                const unsigned grid = (unsigned)std::min<size_t>(o->dlGrid, (n16 + 255) / 256);
                AB_LAUNCH(k_copy_out, dim3(grid), dim3(256), 0, o->stream3, (v4u_t*)r.host, (const v4u_t*)o->rb[r.block].base, n16);
                (void)grid;
                e = hipGetLastError();
            } else
                e = hipMemcpyAsync(r.host, o->rb[r.block].base, o->blockBytes, hipMemcpyDeviceToHost, o->stream3);
            for (const orbx::Attach& a : o->attach[r.block])
                if (e == hipSuccess) e = hipMemcpyAsync((u8*)r.host + a.hostOff, a.dev, a.bytes, hipMemcpyDeviceToHost, o->stream3);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(o->stream3);
        {
            std::lock_guard<std::mutex> lk(o->dlMu);
            if (e != hipSuccess && !o->dlError) o->dlError = (int)e;
            o->dlQueue.pop_front();
            o->dlBusy[r.block] = false;
        }
        o->dlCv.notify_all();
    }
}
// host-side wait until block b's pending copy (if any) has landed
static int dl_wait_block(orbx* o, int b) {
    std::unique_lock<std::mutex> lk(o->dlMu);
    o->dlCv.wait(lk, [&] { return !o->dlBusy[b]; });
    if (o->dlError) { set_err("results-to-host copy failed: %s", hipGetErrorString((hipError_t)o->dlError)); o->dlError = 0; return ORBX_E_HIP; }
    return ORBX_OK;
}
static int dl_drain(orbx* o) {
    std::unique_lock<std::mutex> lk(o->dlMu);
    o->dlCv.wait(lk, [&] { return o->dlQueue.empty(); });
    if (o->dlError) { set_err("results-to-host copy failed: %s", hipGetErrorString((hipError_t)o->dlError)); o->dlError = 0; return ORBX_E_HIP; }
    return ORBX_OK;
}
static void dl_stop(orbx* o) {
    { std::lock_guard<std::mutex> lk(o->dlMu); o->dlStop = true; }
    o->dlCv.notify_all();
    if (o->dlThread.joinable()) o->dlThread.join();
}

extern "C" {

const char* orbx_last_error(void) { return g_err.c_str(); }

int orbx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orbx_create(orbx_t** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                int device_id, int max_w, int max_h, int max_batch) {
    if (!out) return ORBX_E_INVALID;
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > 12 || !(scale_factor > 1.0f) || max_batch < 1 || max_w < 1 || max_h < 1 ||
        ini_th < 1 || ini_th > 255 || min_th < 1 || min_th > 255) {
        set_err("invalid extractor parameters"); return ORBX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err("no HIP device: the MI355X extractor has no CPU fallback"); return ORBX_E_HIP; }
    if (device_id < 0 || device_id >= ndev) { set_err("device %d out of range (%d devices)", device_id, ndev); return ORBX_E_INVALID; }
    HIPCHK(hipSetDevice(device_id));
    orbx* o = new orbx;
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->iniTh = ini_th; o->minTh = min_th; o->device = device_id;
    o->maxW = max_w; o->maxH = max_h; o->maxBatch = max_batch;
    o->fastV1 = ab_env("ORBX_FAST_V1") != nullptr;
    o->qtV1 = ab_env("ORBX_QT_V1") != nullptr;
    // default: the matrix-core blur (k_blur3) directly behind the resize chain, i.e. beside FAST (which is VALU/LDS-bound, while
    // k_blur3 is memory + MFMA).  A/B switches: ORBX_BLUR_V2 = the VALU blur (k_blur2) beside the quadtree as before;
    // ORBX_BLUR_LATE = k_blur3 but beside the quadtree.
    o->blurV2 = ab_env("ORBX_BLUR_V2") != nullptr;
    o->odV1 = ab_env("ORBX_OD_V1") != nullptr;
    if (const char* e = ab_env("ORBX_FAST_QCAP")) o->f3QcapForce = atoi(e);
    o->f3NoFixedPitch = ab_env("ORBX_FAST_PITCH0") != nullptr;
    o->fastV3 = ab_env("ORBX_FAST_V3") != nullptr;
    if (const char* e = ab_env("ORBX_QT_WIDE")) o->qtWideForce = atoi(e) != 0 ? 1 : 0;
    o->blurEarly = !o->blurV2 && ab_env("ORBX_BLUR_LATE") == nullptr;
    o->blurTiled = !o->blurV2 && !o->odV1 && ab_env("ORBX_BLUR_ROWMAJOR") == nullptr;
    o->dlKernel = ab_env("ORBX_DL_KERNEL") != nullptr;          // A/B: results-to-host copy by k_copy_out instead of the copy engine
    if (const char* e = ab_env("ORBX_DL_GRID")) o->dlGrid = std::max(1, atoi(e));
    if (const char* e = ab_env("ORBX_ONE_GRAPH")) o->oneOff = atoi(e) == 0;   // A/B: single-frame calls stay eager
    o->serial = ab_env("ORBX_SERIAL") != nullptr;            // A/B switch: simple per-cell reference kernel
    o->scaleFactor = scale_factor;                              // double member initialised from float (ORBextractor.h:96)
    const int L = nlevels;
    o->sf.resize(L); o->sig2.resize(L); o->invsf.resize(L); o->invsig2.resize(L); o->nfeat.resize(L);
    o->sf[0] = 1.0f; o->sig2[0] = 1.0f;
    for (int i = 1; i < L; ++i) { o->sf[i] = (float)(o->sf[i - 1] * o->scaleFactor); o->sig2[i] = o->sf[i] * o->sf[i]; }
    for (int i = 0; i < L; ++i) { o->invsf[i] = 1.0f / o->sf[i]; o->invsig2[i] = 1.0f / o->sig2[i]; }
    float factor = (float)(1.0f / o->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; ++l) { o->nfeat[l] = cv_round_f(nDesired); sum += o->nfeat[l]; nDesired *= factor; }
    o->nfeat[L - 1] = std::max(nfeatures - sum, 0);
    {   // umax (ORBextractor.cc:542-570)
        int* um = o->umax.v;
        int v, v0, vmax = (int)floorf(15 * sqrtf(2.f) / 2 + 1), vmin = (int)ceilf(15 * sqrtf(2.f) / 2);
        for (v = 0; v <= vmax; ++v) um[v] = (int)lrint(std::sqrt(225.0 - v * v));
        for (v = 15, v0 = 0; v >= vmin; --v) { while (um[v0] == um[v0 + 1]) ++v0; um[v] = v0; ++v0; }
    }
    int rc = ORBX_OK;
    do {
        if (hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&o->stream2, hipStreamNonBlocking) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipStreamCreate failed"); break; }
        for (auto& set : o->evr) for (auto& e : set) if (rc == ORBX_OK && hipEventCreate(&e) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        for (auto& e : o->evLvl) if (rc == ORBX_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        if (rc == ORBX_OK && hipEventCreateWithFlags(&o->evDone, hipEventDisableTiming) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        if (rc == ORBX_OK && hipEventCreateWithFlags(&o->evGuard, hipEventDisableTiming) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        if (rc == ORBX_OK && hipStreamCreateWithFlags(&o->stream3, hipStreamNonBlocking) != hipSuccess) { rc = ORBX_E_HIP; set_err("copy stream creation failed"); }
        for (auto& e : o->evBatchDone) if (rc == ORBX_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        for (auto& e : o->evDepc) if (rc == ORBX_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipEventCreate failed"); }
        if (rc == ORBX_OK && (hipMalloc((void**)&o->dStamps, sizeof(unsigned long long) * orbx::kAllSlots * 4) != hipSuccess ||
                              hipDeviceGetAttribute(&o->wallClockKHz, hipDeviceAttributeWallClockRate, device_id) != hipSuccess || o->wallClockKHz <= 0)) { rc = ORBX_E_HIP; set_err("time stamp setup failed"); }
        if (rc) break;
        const size_t B = max_batch;
        if (hipMalloc((void**)&o->dL0Ptr, sizeof(u8*) * B) != hipSuccess || hipMalloc((void**)&o->dSelCnt, sizeof(u32) * 12 * B) != hipSuccess ||
            hipMalloc((void**)&o->dLap, sizeof(int) * 2 * B) != hipSuccess || hipMalloc((void**)&o->dErr, sizeof(int)) != hipSuccess ||
            hipMalloc((void**)&o->dPattern, 1024) != hipSuccess || hipMalloc((void**)&o->dOdW, sizeof(u32) * OD_WTAB) != hipSuccess || hipMalloc((void**)&o->dOvf, 2 * sizeof(u32)) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipMalloc failed"); break; }
        {   // IC_Angle weights for v_dot4: entry [|v|][j] packs, for the 4 bytes of dword j of a 32-byte patch row (it starts at
            // cx-15) at distance |v| from the centre, (u + 16) where the pixel lies inside the disc (|u| <= umax[|v|]) and 0
            // elsewhere; u = 4j + b - 15.  Row 16 is all zero (lanes past the 31st row).
            std::vector<u32> wt(OD_WTAB, 0u);
            for (int va = 0; va < 16; ++va)
                for (int j = 0; j < 8; ++j) {
                    u32 wv = 0;
                    for (int b = 0; b < 4; ++b) {
                        const int u = 4 * j + b - 15;
                        if (std::abs(u) <= o->umax.v[va]) wv |= (u32)(u + 16) << (8 * b);
                    }
                    wt[va * 8 + j] = wv;
                }
            if (hipMemcpy(o->dOdW, wt.data(), sizeof(u32) * OD_WTAB, hipMemcpyHostToDevice) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipMemcpy failed"); break; }
        }
        if (hipMemcpy(o->dPattern, kPattern, 1024, hipMemcpyHostToDevice) != hipSuccess || hipMemset(o->dErr, 0, sizeof(int)) != hipSuccess || hipMemset(o->dOvf, 0, 2 * sizeof(u32)) != hipSuccess) { rc = ORBX_E_HIP; set_err("hipMemcpy failed"); break; }
        o->hL0Ptr.resize(B); o->hLap.resize(2 * B);
        rc = build_geometry(o, max_w, max_h);
    } while (0);
    if (rc) { orbx_destroy(o); return rc; }
    *out = o;
    return ORBX_OK;
}

void orbx_destroy(orbx_t* o) {
    if (!o) return;
    (void)hipSetDevice(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    if (o->stream2) (void)hipStreamSynchronize(o->stream2);
    dl_stop(o);                                                  // pending copies finish first (their batches are done by now)
    if (o->stream3) (void)hipStreamSynchronize(o->stream3);
    for (auto& G : o->gs) {
        if (G.exec) (void)hipGraphExecDestroy(G.exec);
        if (G.graph) (void)hipGraphDestroy(G.graph);
    }
    for (auto& e : o->evDepc) if (e) (void)hipEventDestroy(e);
    for (auto& e : o->evMark) if (e) (void)hipEventDestroy(e);
    for (auto& e : o->evBatchDone) if (e) (void)hipEventDestroy(e);
    if (o->stream3) (void)hipStreamDestroy(o->stream3);
    void* ptrs[] = {o->dPyr, o->dBlur, o->dL0, (void*)o->dL0Ptr, o->dCells, o->dTiles, o->dTiles3, o->dB3Th, o->dB3Tv, o->dStrips, o->dAux, o->dF4Items, o->dX4, o->dRzTasks, o->dXt, o->dYt, o->dCandCnt, o->dCandEnt,
                    o->dSel, o->dSelCnt, o->dKpNode, o->dDense, o->rb[0].base, o->rb[1].base, o->rb[2].base, o->rb[3].base, o->rb[4].base, o->rb[5].base, o->rb[6].base, o->rb[7].base, o->dWork, o->dLap, o->dErr, o->dPattern, o->dOvf, o->dOvfList, o->dOdW, (void*)o->dStamps};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& set : o->evr) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    if (o->evDone) (void)hipEventDestroy(o->evDone);
    if (o->evGuard) (void)hipEventDestroy(o->evGuard);
    if (o->evH2D) (void)hipEventDestroy(o->evH2D);
    if (o->hPinned) (void)hipHostFree(o->hPinned);
    if (o->hOne) (void)hipHostFree(o->hOne);
    if (o->hPyr) (void)hipHostFree(o->hPyr);
    if (o->hLvl) (void)hipHostFree(o->hLvl);
    if (o->dIngest) (void)hipFree(o->dIngest);
    for (auto& e : o->evLvl) if (e) (void)hipEventDestroy(e);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    if (o->stream2) (void)hipStreamDestroy(o->stream2);
    delete o;
}

int orbx_max_keypoints(const orbx_t* o) { return o ? o->g.kpCap : ORBX_E_INVALID; }

static int extract_batch_async_impl(orbx_t* o, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                                    const int* lap01);
int orbx_extract_batch_async(orbx_t* o, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                             const int* lap01) {
    const int rc = extract_batch_async_impl(o, imgs, img_space, nimg, w, h, stride, lap01);
    if (rc && o && o->capSlot >= 0) o->capFailed = true;         // orbx_capture_end then discards the half-recorded graph
    return rc;
}
static int extract_batch_async_impl(orbx_t* o, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                                    const int* lap01) {
    if (!o || !imgs || nimg < 1) return ORBX_E_INVALID;
    if (nimg > o->maxBatch) { set_err("batch %d exceeds max_batch %d", nimg, o->maxBatch); return ORBX_E_CAPACITY; }
    if (w <= 0 || h <= 0) return ORBX_E_EMPTY;
    for (int i = 0; i < nimg; ++i) if (!imgs[i]) return ORBX_E_EMPTY;
    if (stride < w) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    int rc = build_geometry(o, w, h);
    if (rc) return rc;
    const Geom& g = o->g;
    hipStream_t st = o->stream;
    const bool capturing = o->capSlot >= 0;
    if (!capturing) { rc = dl_wait_block(o, o->curBlock); if (rc) return rc; }   // this block may still be on its way to the host
    int l0pitch;
    bool aligned = (stride % 16) == 0;
    for (int i = 0; i < nimg && aligned; ++i) aligned = ((uintptr_t)imgs[i] % 16) == 0;
    if (capturing && !(img_space == ORBX_DEVICE && aligned)) {
        set_err("graph capture takes device-resident, 16-byte aligned images only (staging copies are per-call host work)"); return ORBX_E_INVALID;
    }
    if (img_space == ORBX_DEVICE && aligned) {
        for (int i = 0; i < nimg; ++i) o->hL0Ptr[i] = imgs[i];
        l0pitch = stride;
        o->l0Staged = false;
    } else if (img_space == ORBX_DEVICE) {                    // kernels use 16-byte row loads: stage misaligned inputs
        o->l0Staged = false;                                  // dL0 is filled device-to-device: hPinned (if any) holds an OLDER host batch
        for (int i = 0; i < nimg; ++i) {
            u8* d = o->dL0 + (size_t)i * o->l0pitch * h;
            HIPCHK(hipMemcpy2DAsync(d, o->l0pitch, imgs[i], stride, w, h, hipMemcpyDeviceToDevice, st));
            o->hL0Ptr[i] = d;
        }
        l0pitch = o->l0pitch;
    } else {
        // host images: packed into one pinned staging buffer (CPU memcpy), then ONE asynchronous copy for the whole batch --
        // per-image copies from pageable memory run at ~3 GB/s and block the calling thread
        const size_t imgBytes = (size_t)o->l0pitch * h, need = imgBytes * nimg;
        if (need > o->capPinned) {
            if (o->hPinned) (void)hipHostFree(o->hPinned);
            o->hPinned = nullptr; o->capPinned = 0;
            o->oneW = o->oneH = 0;                                // the single-frame graph copies out of the old buffer
            HIPCHK(hipHostMalloc((void**)&o->hPinned, need, hipHostMallocDefault));
            o->capPinned = need;
        }
        if (!o->evH2D) HIPCHK(hipEventCreateWithFlags(&o->evH2D, hipEventDisableTiming));
        else HIPCHK(hipEventSynchronize(o->evH2D));               // the previous batch's DMA out of the staging buffer is done
        for (int i = 0; i < nimg; ++i) {
            u8* p = o->hPinned + imgBytes * i;
            if (stride == o->l0pitch) memcpy(p, imgs[i], (size_t)stride * h);
            else for (int y = 0; y < h; ++y) memcpy(p + (size_t)y * o->l0pitch, imgs[i] + (size_t)y * stride, (size_t)w);
            o->hL0Ptr[i] = o->dL0 + imgBytes * i;
        }
        HIPCHK(hipMemcpyAsync(o->dL0, o->hPinned, need, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(o->evH2D, st));
        l0pitch = o->l0pitch;
        o->l0Staged = true;
    }
    for (int i = 0; i < nimg; ++i) { o->hLap[2 * i] = lap01 ? lap01[2 * i] : 0; o->hLap[2 * i + 1] = lap01 ? lap01[2 * i + 1] : 0; }
    // the per-frame pointer and lapping tables are re-uploaded only when they change (a streaming caller cycling through
    // the same device buffers pays for them once: two tiny copies cost ~25 us of stream time per batch)
    if (capturing && (o->upPtr.size() != (size_t)nimg || memcmp(o->upPtr.data(), o->hL0Ptr.data(), sizeof(u8*) * nimg) != 0 ||
                      o->upLap.size() != (size_t)(2 * nimg) || memcmp(o->upLap.data(), o->hLap.data(), sizeof(int) * 2 * nimg) != 0)) {
        set_err("graph capture: enqueue the same batch (same image pointers and lapping areas) once outside the capture first"); return ORBX_E_INVALID;
    }
    if (o->upPtr.size() != (size_t)nimg || memcmp(o->upPtr.data(), o->hL0Ptr.data(), sizeof(u8*) * nimg) != 0) {
        HIPCHK(hipMemcpyAsync((void*)o->dL0Ptr, o->hL0Ptr.data(), sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
        o->upPtr.assign(o->hL0Ptr.begin(), o->hL0Ptr.begin() + nimg);
    }
    if (o->upLap.size() != (size_t)(2 * nimg) || memcmp(o->upLap.data(), o->hLap.data(), sizeof(int) * 2 * nimg) != 0) {
        HIPCHK(hipMemcpyAsync(o->dLap, o->hLap.data(), sizeof(int) * 2 * nimg, hipMemcpyHostToDevice, st));
        o->upLap.assign(o->hLap.begin(), o->hLap.begin() + 2 * nimg);
    }

    o->lastL0Pitch = l0pitch;
    if (!capturing) { o->ev = o->evr[o->nEnq % orbx::kRing]; ++o->nEnq; }
    // Two HIP streams (events order them): FAST is VALU-bound, the resize chain is latency-bound and the quadtree is
    // LDS-latency-bound -- so level-0 FAST (needs no resize) runs beside the resize chain and the blur beside the quadtree.
    //   s0: FAST(L0) --wait pyramid--> FAST(L1..) -> quadtree -> slots --wait blur--> orientation+descriptors
    //   s1: resize L1..L7 -> [pyramid ready] --wait FAST--> blur -> [blur ready]
    hipStream_t s1 = o->serial ? o->stream : o->stream2;          // ORBX_SERIAL=1: single stream, clean per-kernel timings
    HIPCHK(rec_ev(o, 0, st, true));
    HIPCHK(hipStreamWaitEvent(s1, dep_ev(o, 0), 0));             // inputs uploaded; previous batch's readers of the pyramid are done
    STAGE_EV(7, s1);
    for (int l = 1; l < g.nlevels; ++l) {
        if (o->rzStream[l]) {
            const int nt = (int)o->rzTasks[l].size();
            hipLaunchKernelGGL(k_resize2, dim3((nt + 3) / 4, nimg), dim3(256), 0, s1, g, o->dL0Ptr, l0pitch, o->dPyr,
                               o->dRzTasks + o->rzTaskOff[l], nt, o->dX4, o->dYt);
        } else {                                                 // general fallback (large scale factors)
            dim3 grid((g.lv[l].w + 255) / 256, (g.lv[l].h + 3) / 4, nimg), block(64, 4);
            hipLaunchKernelGGL(k_resize, grid, block, 0, s1, g, o->dL0Ptr, l0pitch, o->dPyr, l, o->dXt, o->dYt);
        }
        if (!o->fastV1 && l < g.nlevels - 1)
            for (const orbx::F3Group& G : o->f3g) if (G.lastLevel == l) HIPCHK(hipEventRecord(o->evLvl[l], s1));
    }
    HIPCHK(rec_ev(o, 8, s1, true));                              // pyramid ready
    if (o->fastV1) {
        HIPCHK(hipStreamWaitEvent(st, dep_ev(o, 8), 0));
        STAGE_EV(1, st);
        AB_LAUNCH(k_fast, dim3(g.totalCells, nimg), dim3(256), 0, st, g, o->dL0Ptr, l0pitch, o->dPyr, o->dCells,
                  o->dCandCnt, o->dCandEnt, o->dErr);
        STAGE_EV(10, st);
    } else {
        // every group waits only for the pyramid levels it reads: the fine levels start while the coarse ones are still
        // being resized (the 7 resizes are a dependent chain of small kernels, ~160 us)
        bool first = true;
        for (const orbx::F3Group& G : o->f3g) {
            const bool lvl0 = G.lastLevel == 0;
            if (!lvl0) {
                if (first) STAGE_EV(10, st);   // level-0 FAST done
                HIPCHK(hipStreamWaitEvent(st, G.lastLevel < g.nlevels - 1 ? o->evLvl[G.lastLevel] : dep_ev(o, 8), 0));
                if (first) STAGE_EV(1, st);
                first = false;
            }
            if (G.v4) {
                auto kern = G.pitch == 176 ? k_fast4<176> : G.pitch == 208 ? k_fast4<208> : k_fast4<0>;
                hipLaunchKernelGGL(kern, dim3((unsigned)G.nstrips, nimg), dim3(F3_NT), G.lds, st, g, o->dL0Ptr, l0pitch, o->dPyr,
                                   o->dAux, o->dF4Items, o->dStrips + G.strip0, o->dCandCnt, o->dCandEnt, o->dErr, G.tile, G.qcap, o->dOvf, o->dOvfList);
            } else {
#ifdef ORBX_AB
                auto kern = G.pitch == 176 ? k_fast3<176> : G.pitch == 208 ? k_fast3<208> : k_fast3<0>;
                hipLaunchKernelGGL(kern, dim3((unsigned)G.nstrips, nimg), dim3(F3_NT), G.lds, st, g, o->dL0Ptr, l0pitch, o->dPyr,
                                   o->dCells, o->dStrips + G.strip0, o->dCandCnt, o->dCandEnt, o->dErr, G.tile, G.qcap, o->dOvf, o->dOvfList);
#else
                set_err("FAST group outside k_fast4's field limits (cells wider than 125 px or taller than 127 rows)"); return ORBX_E_UNSUPPORTED;
#endif
            }
        }
        if (first) {                                             // single-level extractor
            STAGE_EV(10, st);
            STAGE_EV(1, st);
        }
        // (already satisfied -- the last FAST group waited for the last level -- but NOT removable: without this edge hipGraph places
        // the blur branch and the FAST branch of the captured step differently and the two no longer overlap: step 1.83 -> 2.07 ms)
        HIPCHK(hipStreamWaitEvent(st, dep_ev(o, 8), 0));         // later stages read every level
        // (grid sized to the batch: a single frame's launch of 512 workgroups that find nothing to do costs 18 us, of 8 workgroups 8 us)
        hipLaunchKernelGGL(k_fast_fix, dim3((unsigned)std::min(512, std::max(8, 4 * nimg))), dim3(256), 0, st, g, o->dL0Ptr, l0pitch, o->dPyr, o->dCells, o->dCandCnt, o->dCandEnt,
                           o->dErr, o->dOvf, o->dOvfList);
    }
    HIPCHK(rec_ev(o, 2, st, true));
    // blur (VALU + HBM) runs beside the quadtree (LDS-latency bound), not beside FAST (VALU bound)
    if (!o->serial && !o->blurEarly) HIPCHK(hipStreamWaitEvent(s1, dep_ev(o, 2), 0));
    STAGE_EV(11, s1);
    if (o->blurV2)
        AB_LAUNCH(k_blur2, dim3((unsigned)(o->tiles.size() + 3) / 4, nimg), dim3(256), 0, s1, g, o->dL0Ptr, l0pitch, o->dPyr, o->dBlur,
                  o->dTiles, (int)o->tiles.size(), o->blurSel);
    else {
        // big batches: workgroups walk B3_CHUNK tiles (prologue amortised); small ones: one tile per workgroup (8x the workgroups,
        // an eighth of the latency -- the single-frame path)
        const bool walk = (size_t)nimg * o->nTiles3Walk >= 2048;
        const size_t first = walk ? 0 : o->nTiles3Walk, count = walk ? o->nTiles3Walk : o->tiles3.size() - o->nTiles3Walk;
        hipLaunchKernelGGL(k_blur3, dim3((unsigned)count, nimg), dim3(256), 0, s1, g, o->dL0Ptr, l0pitch, o->dPyr, o->dBlur,
                           o->dTiles3 + first, (const uint4*)o->dB3Th, (const uint4*)o->dB3Tv);
    }
    HIPCHK(rec_ev(o, 9, s1, true));                              // blur ready
    if (o->qtV1)
        AB_LAUNCH(k_quadtree, dim3(nimg, g.nlevels), dim3(256), o->qtLds, st, g, o->dCells, o->dCandCnt, o->dCandEnt,
                  o->dKpNode, o->dSel, o->dSelCnt, o->dErr);
    else {
        Geom g2 = g; g2.nodeCap = o->qt2Cap; g2.sortCap = o->qt2Sort;
        // few workgroups (small batches, the single-frame call): the kernel's time is one workgroup's latency, and 1024 threads walk
        // a level's candidates four times faster than 256 (one frame: 0.276 -> 0.237 ms end to end); many workgroups: 256 threads
        // (shorter barriers, more workgroups per CU: 0.24 vs 0.56 ms per 512 frames)
        const bool wide = o->qtWideForce >= 0 ? o->qtWide : (o->qtWide || nimg * g.nlevels <= 512);
        if (wide)
            hipLaunchKernelGGL(k_quadtree2<1024>, dim3(nimg, g.nlevels), dim3(1024), o->qt2Lds, st, g2, o->dCandCnt, o->dCandEnt,
                               o->dDense, o->dKpNode, o->dSel, o->dSelCnt, o->dErr, o->maxCells, o->maxIni, o->qtFuseD);
        else
            hipLaunchKernelGGL(k_quadtree2<256>, dim3(nimg, g.nlevels), dim3(256), o->qt2Lds, st, g2, o->dCandCnt, o->dCandEnt,
                               o->dDense, o->dKpNode, o->dSel, o->dSelCnt, o->dErr, o->maxCells, o->maxIni, o->qtFuseD);
    }
    STAGE_EV(3, st);
    if (o->guardPending) {                                       // k_slots is the first writer of the result block (orbx_guard_results)
        HIPCHK(hipStreamWaitEvent(st, o->evGuard, 0));
        o->guardPending = false;
    }

    hipLaunchKernelGGL(k_slots, dim3(nimg), dim3(256), 0, st, g, o->dSel, o->dSelCnt, o->dLap, o->dKps, o->dWork, o->dN, o->dMono);
    STAGE_EV(4, st);
    HIPCHK(hipStreamWaitEvent(st, dep_ev(o, 9), 0));
    STAGE_EV(5, st);
    if (o->odV1)
        AB_LAUNCH(k_orient_desc, dim3((g.kpCap + 3) / 4, nimg), dim3(256), 0, st, g, o->dL0Ptr, l0pitch, o->dPyr, o->dBlur,
                  o->dWork, o->dN, o->dKps, o->dDesc, o->dPattern, o->umax);
    else
        hipLaunchKernelGGL(k_orient_desc2, dim3((g.kpCap + 15) / 16, nimg), dim3(256), 0, st, g, o->dL0Ptr, l0pitch, o->dPyr, o->dBlur,
                           o->dWork, o->dN, o->dKps, o->dDesc, o->dPattern, o->dOdW);
    HIPCHK(rec_ev(o, 6, st, false));
    HIPCHK(hipGetLastError());
    if (capturing) { o->gs[o->capSlot].nimg = nimg; o->gs[o->capSlot].block = o->curBlock; return ORBX_OK; }   // nothing has run: the handle's state changes at orbx_graph_launch
    o->lastBatch = nimg;
    o->timed = true;
    o->graphMode = false;
    o->countsValid = false;
    return ORBX_OK;
}

// ---- HIP-graph replay.  Between orbx_capture_begin and orbx_capture_end every *_async call on this handle (and on anything
// whose stream is the extractor's: orbm_set_stream(m, orbx_stream(o))) is recorded instead of run -- the two-stream fork/join
// of the extraction, the match kernels behind it, the results-to-host copies on the copy stream -- and one orbx_graph_launch
// replays the lot with a single runtime call.
int orbx_capture_begin(orbx_t* o, int slot) {
    if (!o || slot < 0 || slot >= orbx::kSlots) return ORBX_E_INVALID;
    if (o->capSlot >= 0) { set_err("a capture is already open"); return ORBX_E_INVALID; }
    if (!o->curW) { set_err("run one batch eagerly first: capture needs the geometry and the per-frame tables in place"); return ORBX_E_INVALID; }
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream)); HIPCHK(hipStreamSynchronize(o->stream2));
    { const int rc = dl_drain(o); if (rc) return rc; }
    orbx::GraphSlot& G = o->gs[slot];
    if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
    if (G.graph) { (void)hipGraphDestroy(G.graph); G.graph = nullptr; }
    G.launches = 0; G.nimg = 0;
    HIPCHK(hipStreamBeginCapture(o->stream, hipStreamCaptureModeThreadLocal));
    o->capSlot = slot; o->capFailed = false;
    return ORBX_OK;
}

int orbx_capture_end(orbx_t* o) {
    if (!o || o->capSlot < 0) return ORBX_E_INVALID;
    orbx::GraphSlot& G = o->gs[o->capSlot];
    hipStream_t st = o->stream;
    // forked streams must be joined before the capture ends: a download enqueued behind the last batch has no consumer yet
    hipError_t e = hipSuccess;
    if (o->guardPending) { e = hipStreamWaitEvent(st, o->evGuard, 0); o->guardPending = false; }
    o->capSlot = -1;
    hipGraph_t graph = nullptr;
    hipError_t e2 = hipStreamEndCapture(st, &graph);
    if (e != hipSuccess || e2 != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        set_err("stream capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return ORBX_E_HIP;
    }
    if (o->capFailed || G.nimg < 1) {
        (void)hipGraphDestroy(graph);
        G.nimg = 0;
        set_err(o->capFailed ? "an enqueue failed inside the capture: nothing recorded" : "the capture holds no extraction");
        return ORBX_E_INVALID;
    }
    G.graph = graph;
    HIPCHK(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
    return ORBX_OK;
}

int orbx_graph_launch(orbx_t* o, int slot) {
    if (!o || slot < 0 || slot >= orbx::kSlots || !o->gs[slot].exec || o->capSlot >= 0) return ORBX_E_INVALID;
    orbx::GraphSlot& G = o->gs[slot];
    if (G.nimg > 0) { const int rc = dl_wait_block(o, G.block); if (rc) return rc; }   // the block this graph rewrites may still be on its way to the host
    HIPCHK(hipGraphLaunch(G.exec, o->stream));
    ++G.launches;
    if (G.nimg > 0) { const int rc = orbx_set_result_block(o, G.block); if (rc) return rc; }
    if (G.nimg > 0) { o->lastBatch = G.nimg; o->countsValid = false; o->timed = true; o->graphMode = true; }
    return ORBX_OK;
}

// ---- results -> host.  The reference's consumers are host-side (Frame.cc:357-366 fills mvKeys / mDescriptors), so SURVEY 8(d)
// counts this copy in the metric.  The copy runs on its own stream behind everything enqueued so far on the extractor's stream,
// i.e. beside the NEXT batch's pyramid / FAST phase; that batch waits for it only before its first kernel that rewrites the
// result block.  Destinations should be pinned (orbx_host_alloc) -- pageable memory makes the copy synchronous.
int orbx_result_download_async(orbx_t* o, void* host_block) {
    if (!o || !host_block || !o->curW) return ORBX_E_INVALID;
    if (o->capSlot >= 0) { set_err("the copy to the host is enqueued eagerly, behind the graph launch, not captured"); return ORBX_E_INVALID; }
    HIPCHK(hipSetDevice(o->device));
    const int b = o->curBlock;
    int rc = dl_wait_block(o, b);                                // an earlier copy of this very block
    if (rc) return rc;
    HIPCHK(hipEventRecord(o->evBatchDone[b], o->stream));        // "the batch that fills the block is done"
    {
        std::lock_guard<std::mutex> lk(o->dlMu);
        if (!o->dlThread.joinable()) o->dlThread = std::thread(dl_worker, o);
        o->dlBusy[b] = true;
        o->dlQueue.push_back(orbx::DlReq{b, host_block});
    }
    o->dlCv.notify_all();
    return ORBX_OK;
}

// Caller-owned device buffers that belong to a block's batch (a matcher's outputs for that batch: mvuRight, mvDepth, match lists)
// travel to the host with it: they land behind the block proper, at *host_offset of the pinned host block.
int orbx_block_attach(orbx_t* o, int block, const void* dev, size_t bytes, size_t* host_offset) {
    if (!o || block < 0 || block >= orbx::kBlocks || !dev || bytes == 0 || !host_offset || !o->curW) return ORBX_E_INVALID;
    const int rc = dl_wait_block(o, block);
    if (rc) return rc;
    size_t off = o->blockBytes;
    for (const orbx::Attach& a : o->attach[block]) off = a.hostOff + ((a.bytes + 255) & ~(size_t)255);
    o->attach[block].push_back(orbx::Attach{dev, bytes, off});
    *host_offset = off;
    return ORBX_OK;
}
int orbx_block_detach_all(orbx_t* o) {
    if (!o) return ORBX_E_INVALID;
    const int rc = dl_drain(o);
    for (auto& v : o->attach) v.clear();
    return rc;
}

int orbx_result_block_layout(const orbx_t* o, size_t* off_kps, size_t* off_desc, size_t* off_counts, size_t* off_monos, size_t* bytes) {
    if (!o || !o->curW) return ORBX_E_INVALID;
    if (off_kps) *off_kps = o->offKps;
    if (off_desc) *off_desc = o->offDesc;
    if (off_counts) *off_counts = o->offN;
    if (off_monos) *off_monos = o->offMono;
    if (bytes) *bytes = o->blockBytes;
    return ORBX_OK;
}

int orbx_set_result_block(orbx_t* o, int block) {
    if (!o || block < 0 || block >= orbx::kBlocks) return ORBX_E_INVALID;
    if (block != o->curBlock) o->countsValid = false;
    o->curBlock = block;
    o->dKps = o->rb[block].kps; o->dDesc = o->rb[block].desc; o->dN = o->rb[block].n; o->dMono = o->rb[block].mono;
    return ORBX_OK;
}

int orbx_download_sync(orbx_t* o) {
    if (!o) return ORBX_E_INVALID;
    return dl_drain(o);
}

// caller-placed time stamps on the extractor's stream (which = 0 / 1), e.g. around a run of graph replays: the GPU-side wall
// time of the whole run, gaps between batches included
int orbx_mark(orbx_t* o, int which) {
    if (!o || which < 0 || which > 1 || o->capSlot >= 0) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    if (!o->evMark[which]) HIPCHK(hipEventCreate(&o->evMark[which]));
    HIPCHK(hipEventRecord(o->evMark[which], o->stream));
    return ORBX_OK;
}
int orbx_mark_elapsed_ms(orbx_t* o, float* ms) {
    if (!o || !ms || !o->evMark[0] || !o->evMark[1]) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipEventSynchronize(o->evMark[1]));
    HIPCHK(hipEventElapsedTime(ms, o->evMark[0], o->evMark[1]));
    return ORBX_OK;
}

void* orbx_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void orbx_host_free(void* p) { if (p) (void)hipHostFree(p); }

int orbx_sync(orbx_t* o) {
    if (!o) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipStreamSynchronize(o->stream2));
    { const int rc = dl_drain(o); if (rc) return rc; }
    int e = 0;
    HIPCHK(hipMemcpy(&e, o->dErr, sizeof(int), hipMemcpyDeviceToHost));
    if (e) { set_err("device-side overflow flag %d", e); (void)hipMemset(o->dErr, 0, sizeof(int)); return ORBX_E_INTERNAL; }
    return ORBX_OK;
}

// one sync + error check + two small copies per BATCH (not per frame): the counts every fetch needs
static int fetch_counts(orbx* o) {
    if (o->countsValid) return ORBX_OK;
    int rc = orbx_sync(o);
    if (rc) return rc;
    o->hN.resize(o->lastBatch); o->hMono.resize(o->lastBatch);
    if (o->lastBatch > 0) {
        HIPCHK(hipMemcpy(o->hN.data(), o->dN, sizeof(int) * o->lastBatch, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(o->hMono.data(), o->dMono, sizeof(int) * o->lastBatch, hipMemcpyDeviceToHost));
    }
    o->countsValid = true;
    return ORBX_OK;
}

int orbx_extract_batch(orbx_t* o, const uint8_t* const* imgs, int img_space, int nimg, int w, int h, int stride,
                       const int* lap01, int* n_out, int* mono_out) {
    int rc = orbx_extract_batch_async(o, imgs, img_space, nimg, w, h, stride, lap01);
    if (rc) return rc;
    rc = fetch_counts(o);
    if (rc) return rc;
    if (n_out) memcpy(n_out, o->hN.data(), sizeof(int) * nimg);
    if (mono_out) memcpy(mono_out, o->hMono.data(), sizeof(int) * nimg);
    return ORBX_OK;
}

int orbx_result_device(const orbx_t* o, const orbx_kp_t** kps, const uint8_t** desc, const int32_t** counts,
                       const int32_t** monos, int* cap) {
    if (!o) return ORBX_E_INVALID;
    if (kps) *kps = (const orbx_kp_t*)o->dKps;
    if (desc) *desc = o->dDesc;
    if (counts) *counts = o->dN;
    if (monos) *monos = o->dMono;
    if (cap) *cap = o->g.kpCap;
    return ORBX_OK;
}

int orbx_result_fetch(orbx_t* o, int i, orbx_kp_t* kps, uint8_t* desc, int cap, int* mono_index) {
    if (!o || i < 0 || i >= o->lastBatch) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    int rc = fetch_counts(o);
    if (rc) return rc;
    const int n = o->hN[i], mono = o->hMono[i];
    if (n > cap) { set_err("%d keypoints exceed caller capacity %d", n, cap); return ORBX_E_CAPACITY; }
    if (n > 0) {
        if (kps) HIPCHK(hipMemcpy(kps, o->dKps + (size_t)i * o->g.kpCap, sizeof(KpOut) * n, hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, o->dDesc + (size_t)i * o->g.kpCap * 32, (size_t)32 * n, hipMemcpyDeviceToHost));
    }
    if (mono_index) *mono_index = mono;
    return n;
}

int orbx_result_fetch_all(orbx_t* o, orbx_kp_t* kps, uint8_t* desc, int cap_per_img, int* n_out, int* mono_out) {
    if (!o || o->lastBatch < 1 || cap_per_img < 1) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    int rc = fetch_counts(o);
    if (rc) return rc;
    const int B = o->lastBatch, dc = o->g.kpCap;
    for (int i = 0; i < B; ++i) if (o->hN[i] > cap_per_img) { set_err("%d keypoints exceed caller capacity %d", o->hN[i], cap_per_img); return ORBX_E_CAPACITY; }
    if (cap_per_img == dc) {                                    // same layout as the device buffers: one copy each
        if (kps) HIPCHK(hipMemcpy(kps, o->dKps, sizeof(KpOut) * (size_t)dc * B, hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, o->dDesc, (size_t)32 * dc * B, hipMemcpyDeviceToHost));
    } else {
        const size_t rows = (size_t)std::min(cap_per_img, dc);
        if (kps) HIPCHK(hipMemcpy2D(kps, sizeof(KpOut) * cap_per_img, o->dKps, sizeof(KpOut) * dc, sizeof(KpOut) * rows, B, hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy2D(desc, (size_t)32 * cap_per_img, o->dDesc, (size_t)32 * dc, 32 * rows, B, hipMemcpyDeviceToHost));
    }
    if (n_out) memcpy(n_out, o->hN.data(), sizeof(int) * B);
    if (mono_out) memcpy(mono_out, o->hMono.data(), sizeof(int) * B);
    return B;
}

// ---- the single-frame call.  A frame's ~17 kernels, event records and waits cost the calling thread ~0.15 ms to enqueue -- as long
// as the GPU needs to run them -- so from the third call with the same geometry on (and per-stage timing off) the whole sequence
// [image staging buffer -> HBM, lapping area -> HBM, extraction on both streams, k_fetch_one] is replayed as ONE hipGraphLaunch;
// per call the host only copies the image into the pinned staging buffer and reads the packed results back out of pinned memory.
static size_t one_desc_off(int kc) { return (16 + (size_t)kc * 28 + 15) & ~(size_t)15; }

static int one_landing(orbx* o) {                                // pinned landing buffer [header | kps | desc | lapping area]
    const int kc = o->g.kpCap;
    const size_t need = one_desc_off(kc) + (size_t)kc * 32 + 16;
    if (need > o->capOne) {
        if (o->hOne) (void)hipHostFree(o->hOne);
        o->hOne = nullptr; o->capOne = 0;
        o->oneW = o->oneH = 0;                                   // a captured graph would hold the old pointer
        HIPCHK(hipHostMalloc((void**)&o->hOne, need, hipHostMallocDefault));
        o->capOne = need;
    }
    return ORBX_OK;
}

static int one_results(orbx* o, orbx_kp_t* kps, uint8_t* desc, int cap, int* mono_index) {
    const int kc = o->g.kpCap;
    const u32* hd = (const u32*)o->hOne;
    const int n = (int)hd[0], mono = (int)hd[1], e = (int)hd[2];
    if (e) { set_err("device-side overflow flag %d", e); (void)hipMemset(o->dErr, 0, sizeof(int)); return ORBX_E_INTERNAL; }
    o->hN.assign(1, n); o->hMono.assign(1, mono); o->countsValid = true;
    if (n > cap) { set_err("%d keypoints exceed caller capacity %d", n, cap); return ORBX_E_CAPACITY; }
    if (n > 0) {
        if (kps) memcpy(kps, hd + 4, sizeof(KpOut) * (size_t)n);
        if (desc) memcpy(desc, o->hOne + one_desc_off(kc), (size_t)32 * n);
    }
    if (mono_index) *mono_index = mono;
    return n;
}

static void one_launch_fetch(orbx* o) {
    const int kc = o->g.kpCap;
    hipLaunchKernelGGL(k_fetch_one, dim3(8), dim3(256), 0, o->stream, o->dKps, o->dDesc, o->dN, o->dMono, o->dErr, 0, kc, (u32*)o->hOne,
                       (int)(one_desc_off(kc) / 4));
}

// (The four one-lane k_stamp kernels and the 8-byte lapping-area copy stay in this graph although nothing reads the stamps: without
// them hipGraph places the nodes so that the replay is no faster than the eager path -- 0.193 instead of 0.170 ms, measured.)
// capture the sequence for the current geometry into the private slot; on any failure the handle stays on the eager path for good
static void one_capture(orbx* o, int w, int h) {
    // the captured call below takes level 0 as a device image (dL0); the frame in it was uploaded from hPinned by the eager call
    // that led here and the capture does not change that, so the "host copy still staged" state survives this function
    struct KeepStaged { orbx* o; bool v; ~KeepStaged() { o->l0Staged = v; } } keepStaged{o, o->l0Staged};
    const int slot = orbx::kSlots;
    orbx::GraphSlot& G = o->gs[slot];
    if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
    if (G.graph) { (void)hipGraphDestroy(G.graph); G.graph = nullptr; }
    G.launches = 0; G.nimg = 0;
    o->oneW = o->oneH = 0;
    if (hipStreamSynchronize(o->stream) != hipSuccess || hipStreamSynchronize(o->stream2) != hipSuccess || dl_drain(o)) { o->oneOff = true; return; }
    const int kc = o->g.kpCap;
    int* lapPinned = (int*)(o->hOne + one_desc_off(kc) + (size_t)kc * 32);
    const size_t imgBytes = (size_t)o->l0pitch * h;
    if (hipStreamBeginCapture(o->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); o->oneOff = true; return; }
    o->capSlot = slot; o->capFailed = false;
    bool ok = hipMemcpyAsync(o->dL0, o->hPinned, imgBytes, hipMemcpyHostToDevice, o->stream) == hipSuccess &&
              hipMemcpyAsync(o->dLap, lapPinned, 2 * sizeof(int), hipMemcpyHostToDevice, o->stream) == hipSuccess;
    if (ok) {
        const uint8_t* dimg = o->dL0;
        const int lap[2] = {lapPinned[0], lapPinned[1]};
        o->upLap.assign(lap, lap + 2);                           // the captured copy above uploads them on every replay
        ok = orbx_extract_batch_async(o, &dimg, ORBX_DEVICE, 1, w, h, o->l0pitch, lap) == ORBX_OK;
    }
    if (ok) { one_launch_fetch(o); ok = hipGetLastError() == hipSuccess; }
    o->capSlot = -1;
    hipGraph_t graph = nullptr;
    const hipError_t e2 = hipStreamEndCapture(o->stream, &graph);
    if (!ok || e2 != hipSuccess || !graph || o->capFailed || G.nimg != 1) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        G.exec = nullptr; G.nimg = 0; o->oneOff = true;
        return;
    }
    // How hipGraph places a small graph's nodes differs from one instantiation to the next (a 16-frame step: 0.24 or 0.32 ms), so the
    // replay is timed against the eager call it is meant to beat -- on the frame still in the staging buffer, results land in the
    // landing buffer and are simply overwritten -- and re-instantiated, at most twice, if it does not.
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); G.exec = nullptr; break; }
        double best = 1e30;
        bool ran = true;
        for (int r = 0; r < 3 && ran; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            ran = hipGraphLaunch(G.exec, o->stream) == hipSuccess && hipStreamSynchronize(o->stream) == hipSuccess;
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        if (ran && (o->oneEagerUs <= 0 || best < 0.85 * o->oneEagerUs)) break;   // (the eager figure includes ~25 us of staging memcpy the replay here does not)
        (void)hipGraphExecDestroy(G.exec); G.exec = nullptr;
        (void)hipGetLastError();
    }
    if (!G.exec) { (void)hipGraphDestroy(graph); G.nimg = 0; o->oneOff = true; return; }
    G.graph = graph;
    o->oneW = w; o->oneH = h;
}

int orbx_extract(orbx_t* o, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                 orbx_kp_t* kps, uint8_t* desc, int cap, int* mono_index) {
    if (!o) return ORBX_E_INVALID;
    if (!img || w <= 0 || h <= 0) return ORBX_E_EMPTY;
    const int lap[2] = {lap0, lap1};
    HIPCHK(hipSetDevice(o->device));
    const bool graphOk = !o->oneOff && !o->stageTiming && !o->serial && o->capSlot < 0 && !o->guardPending;   // (a pending orbx_guard_results wait is eager-path business)
    // (the graph's kernels read the device-side image-pointer table as it was at capture: a batch call in between may have rewritten it)
    const bool tableOk = o->upPtr.size() == 1 && o->upPtr[0] == o->dL0;
    if (graphOk && tableOk && o->gs[orbx::kSlots].exec && o->oneW == w && o->oneH == h && o->curW == w && o->curH == h && stride >= w) {
        // ---- replay
        orbx::GraphSlot& G = o->gs[orbx::kSlots];
        { const int rc = dl_wait_block(o, G.block); if (rc) return rc; }
        const int kc = o->g.kpCap;
        if (o->evH2D) HIPCHK(hipEventSynchronize(o->evH2D));     // an asynchronous batch of host images may still be leaving the staging buffer
        u8* p = o->hPinned;
        if (stride == o->l0pitch) memcpy(p, img, (size_t)stride * h);
        else for (int y = 0; y < h; ++y) memcpy(p + (size_t)y * o->l0pitch, img + (size_t)y * stride, (size_t)w);
        int* lapPinned = (int*)(o->hOne + one_desc_off(kc) + (size_t)kc * 32);
        lapPinned[0] = lap0; lapPinned[1] = lap1;
        HIPCHK(hipGraphLaunch(G.exec, o->stream));
        ++G.launches;
        { const int rc = orbx_set_result_block(o, G.block); if (rc) return rc; }
        o->hL0Ptr[0] = o->dL0; o->lastL0Pitch = o->l0pitch; o->l0Staged = true;
        o->hLap[0] = lap0; o->hLap[1] = lap1; o->upLap.assign(lap, lap + 2);
        o->lastBatch = 1; o->countsValid = false; o->timed = true; o->graphMode = true;
        HIPCHK(hipStreamSynchronize(o->stream));
        return one_results(o, kps, desc, cap, mono_index);
    }
    const auto tEager = std::chrono::steady_clock::now();
    int rc = orbx_extract_batch_async(o, &img, ORBX_HOST, 1, w, h, stride, lap);
    if (rc) return rc;
    // results: one packing kernel into pinned host memory and one synchronisation (k_fetch_one) instead of the general path's
    // stream sync + error flag + counts + monos + keypoints + descriptors as separate synchronous copies
    rc = one_landing(o);
    if (rc) return rc;
    one_launch_fetch(o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(o->stream));
    o->oneEagerUs = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tEager).count();
    { const int drc = dl_drain(o); if (drc) return drc; }
    const int n = one_results(o, kps, desc, cap, mono_index);
    if (n >= 0 && graphOk) {
        if (o->oneW == w && o->oneH == h) o->oneEager = 0;
        else if (++o->oneEager >= 2) {                           // same geometry twice in a row on the eager path: worth a graph
            int* lapPinned = (int*)(o->hOne + one_desc_off(o->g.kpCap) + (size_t)o->g.kpCap * 32);
            lapPinned[0] = lap0; lapPinned[1] = lap1;
            one_capture(o, w, h);
            o->oneEager = 0;
        }
    }
    return n;
}

int orbx_level_size(const orbx_t* o, int level, int* w, int* h) {
    if (!o || level < 0 || level >= o->nlevels || !o->curW) return ORBX_E_INVALID;
    *w = o->g.lv[level].w; *h = o->g.lv[level].h;
    return ORBX_OK;
}

int orbx_level_image(orbx_t* o, int frame, int level, int blurred, uint8_t* dst, int dst_stride) {
    if (!o || frame < 0 || frame >= o->lastBatch || level < 0 || level >= o->nlevels) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    const LevelDesc& D = o->g.lv[level];
    if (level == 0 && !blurred) {
        // level 0 lives wherever the caller's image (or its staged copy) is
        const u8* src = o->hL0Ptr[frame];
        int pitch = (src >= o->dL0 && src < o->dL0 + o->capL0) ? o->l0pitch : 0;
        if (!pitch) { set_err("level 0 of a device-resident input is the caller's own buffer"); return ORBX_E_INVALID; }
        HIPCHK(hipMemcpy2D(dst, dst_stride, src, pitch, D.w, D.h, hipMemcpyDeviceToHost));
        return ORBX_OK;
    }
    if (blurred && o->g.blurTiled) {                             // de-tile on the host (a debug / test accessor)
        const size_t nb = (size_t)D.btpr * ((D.h + 7) / 8) * 128;
        std::vector<u8> raw(nb);
        HIPCHK(hipMemcpy(raw.data(), o->dBlur + (size_t)frame * o->g.blrFrameBytes + D.boff, nb, hipMemcpyDeviceToHost));
        for (int y = 0; y < D.h; ++y)
            for (int x = 0; x < D.w; ++x)
                dst[(size_t)y * dst_stride + x] = raw[((size_t)(y >> 3) * D.btpr + (x >> 4)) * 128 + (size_t)(y & 7) * 16 + (x & 15)];
        return ORBX_OK;
    }
    const u8* base = blurred ? o->dBlur + (size_t)frame * o->g.blrFrameBytes + D.boff : o->dPyr + (size_t)frame * o->g.pyrFrameBytes + D.off;
    // one contiguous copy of the level (rows with their pitch) into pinned memory, then host row copies: a 2-D copy into pageable
    // memory goes row by row through the runtime (1.2 ms for a 752x480 level)
    const size_t nb = (size_t)D.pitch * D.h;
    if (nb > o->capLvlHost) {
        if (o->hLvl) (void)hipHostFree(o->hLvl);
        o->hLvl = nullptr; o->capLvlHost = 0;
        HIPCHK(hipHostMalloc((void**)&o->hLvl, nb, hipHostMallocDefault));
        o->capLvlHost = nb;
    }
    HIPCHK(hipMemcpy(o->hLvl, base, nb, hipMemcpyDeviceToHost));
    for (int y = 0; y < D.h; ++y) memcpy(dst + (size_t)y * dst_stride, o->hLvl + (size_t)y * D.pitch, (size_t)D.w);
    return ORBX_OK;
}

// every level of one frame of the last batch in one go (the facade's mvImagePyramid, ORBextractor.h:83): levels 1.. leave HBM as ONE
// copy of the frame's pyramid slab into pinned memory, level 0 of a host image comes back out of the staging buffer it was uploaded
// from -- instead of one synchronous 2-D copy per level (8 x ~35 us, more than the extraction of a 752x480 frame itself).
int orbx_pyramid_fetch(orbx_t* o, int frame, uint8_t* const* dst, const int* dst_stride) {
    if (!o || !dst || !dst_stride || frame < 0 || frame >= o->lastBatch || !o->curW) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    const Geom& g = o->g;
    if (g.pyrFrameBytes > o->capPyrHost) {
        if (o->hPyr) (void)hipHostFree(o->hPyr);
        o->hPyr = nullptr; o->capPyrHost = 0;
        HIPCHK(hipHostMalloc((void**)&o->hPyr, g.pyrFrameBytes, hipHostMallocDefault));
        o->capPyrHost = g.pyrFrameBytes;
    }
    if (g.nlevels > 1) HIPCHK(hipMemcpyAsync(o->hPyr, o->dPyr + (size_t)frame * g.pyrFrameBytes, g.pyrFrameBytes, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));                     // (also: everything the batch enqueued has run)
    for (int l = 0; l < g.nlevels; ++l) {
        const LevelDesc& D = g.lv[l];
        if (!dst[l]) continue;
        if (l == 0) {
            const u8* src = o->hL0Ptr[frame];
            const bool staged = o->l0Staged && o->hPinned && src >= o->dL0 && src < o->dL0 + o->capL0 && (size_t)(src - o->dL0) + (size_t)o->l0pitch * D.h <= o->capPinned;
            if (staged) {                                        // the host copy of what was uploaded is still in the staging buffer
                const u8* hp = o->hPinned + (src - o->dL0);
                for (int y = 0; y < D.h; ++y) memcpy(dst[0] + (size_t)y * dst_stride[0], hp + (size_t)y * o->l0pitch, (size_t)D.w);
            } else {
                const int rc = orbx_level_image(o, frame, 0, 0, dst[0], dst_stride[0]);
                if (rc) return rc;
            }
            continue;
        }
        const u8* hp = o->hPyr + D.off;
        for (int y = 0; y < D.h; ++y) memcpy(dst[l] + (size_t)y * dst_stride[l], hp + (size_t)y * D.pitch, (size_t)D.w);
    }
    return ORBX_OK;
}

// zero-copy form of orbx_pyramid_fetch: the frame's pyramid slab lands in the handle's pinned buffer and the caller gets a pointer
// and a row pitch per level (level 0: the staging buffer the image was uploaded from; NULL when the frame was device-resident).
// The memory belongs to the handle and is overwritten by the next extraction / fetch -- exactly the lifetime of the reference's
// mvImagePyramid, which the extractor overwrites on every call (ORBextractor.cc:1672).
int orbx_pyramid_map(orbx_t* o, int frame, const uint8_t** ptr, int* pitch) {
    if (!o || !ptr || !pitch || frame < 0 || frame >= o->lastBatch || !o->curW) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    const Geom& g = o->g;
    if (g.pyrFrameBytes > o->capPyrHost) {
        if (o->hPyr) (void)hipHostFree(o->hPyr);
        o->hPyr = nullptr; o->capPyrHost = 0;
        HIPCHK(hipHostMalloc((void**)&o->hPyr, g.pyrFrameBytes, hipHostMallocDefault));
        o->capPyrHost = g.pyrFrameBytes;
    }
    if (g.nlevels > 1) HIPCHK(hipMemcpyAsync(o->hPyr, o->dPyr + (size_t)frame * g.pyrFrameBytes, g.pyrFrameBytes, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    const u8* src = o->hL0Ptr[frame];
    const bool staged = o->l0Staged && o->hPinned && src >= o->dL0 && src < o->dL0 + o->capL0 && (size_t)(src - o->dL0) + (size_t)o->l0pitch * g.lv[0].h <= o->capPinned;
    ptr[0] = staged ? o->hPinned + (src - o->dL0) : nullptr; pitch[0] = staged ? o->l0pitch : 0;
    for (int l = 1; l < g.nlevels; ++l) { ptr[l] = o->hPyr + g.lv[l].off; pitch[l] = g.lv[l].pitch; }
    return ORBX_OK;
}

void orbx_scale_tables(const orbx_t* o, float* sf, float* inv_sf, float* sig2, float* inv_sig2) {
    for (int i = 0; i < o->nlevels; ++i) {
        if (sf) sf[i] = o->sf[i];
        if (inv_sf) inv_sf[i] = o->invsf[i];
        if (sig2) sig2[i] = o->sig2[i];
        if (inv_sig2) inv_sig2[i] = o->invsig2[i];
    }
}

int orbx_features_per_level(const orbx_t* o, int* nfeat) {
    if (!o) return ORBX_E_INVALID;
    for (int i = 0; i < o->nlevels; ++i) nfeat[i] = o->nfeat[i];
    return o->nlevels;
}

int orbx_level_candidates(orbx_t* o, int frame, int level, int32_t* xyr, int cap) {
    if (!o || frame < 0 || frame >= o->lastBatch || level < 0 || level >= o->nlevels) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    const Geom& g = o->g;
    const LevelDesc& D = g.lv[level];
    std::vector<u32> cnt(D.nCells), ent((size_t)D.nCells * D.slotCap);
    HIPCHK(hipMemcpy(cnt.data(), o->dCandCnt + (size_t)frame * g.totalCells + D.cellBase, sizeof(u32) * D.nCells, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ent.data(), o->dCandEnt + (size_t)frame * g.totalSlots + D.slotBase, sizeof(u32) * ent.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < D.nCells; ++c)
        for (u32 k = 0; k < cnt[c]; ++k, ++n)
            if (n < cap) {
                const u32 e = ent[(size_t)c * D.slotCap + k];
                xyr[3 * n] = (int)(e & 0xFFF); xyr[3 * n + 1] = (int)((e >> 12) & 0xFFF); xyr[3 * n + 2] = (int)(e >> 24);
            }
    return n;
}

int orbx_level_selected(orbx_t* o, int frame, int level, int32_t* xyr, int cap) {
    if (!o || frame < 0 || frame >= o->lastBatch || level < 0 || level >= o->nlevels) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    const Geom& g = o->g;
    const LevelDesc& D = g.lv[level];
    u32 n = 0;
    HIPCHK(hipMemcpy(&n, o->dSelCnt + frame * g.nlevels + level, sizeof(u32), hipMemcpyDeviceToHost));
    std::vector<u32> e(n);
    if (n) HIPCHK(hipMemcpy(e.data(), o->dSel + (size_t)frame * g.totalSel + D.selBase, sizeof(u32) * n, hipMemcpyDeviceToHost));
    for (u32 i = 0; i < n && (int)i < cap; ++i) {
        xyr[3 * i] = (int)(e[i] & 0xFFF) + 16; xyr[3 * i + 1] = (int)((e[i] >> 12) & 0xFFF) + 16; xyr[3 * i + 2] = (int)(e[i] >> 24);
    }
    return (int)n;
}

static int timings_of(orbx* o, hipEvent_t* ev, float* ms7) {
    float f0 = 0, f1 = 0;
    if (!o->stageTiming) {                                   // only the dependency / span events exist: total and pyramid+FAST span
        for (int i = 0; i < 6; ++i) ms7[i] = 0.f;
        HIPCHK(hipEventElapsedTime(&ms7[6], ev[0], ev[6]));
        HIPCHK(hipEventElapsedTime(&ms7[7], ev[0], ev[2]));
        ms7[1] = ms7[7];                                     // start -> end of the last FAST launch: the pyramid+FAST span proper
        if (o->blurEarly && !o->serial) {
            float b = 0;
            HIPCHK(hipEventElapsedTime(&b, ev[0], ev[9]));
            ms7[7] = std::max(ms7[7], b);
        }
        return ORBX_OK;
    }
    HIPCHK(hipEventElapsedTime(&ms7[0], ev[7], ev[8]));      // resize chain (stream 2)
    HIPCHK(hipEventElapsedTime(&f0, ev[0], ev[10]));         // FAST level 0 (fastV1: whole FAST is ev1->ev10)
    HIPCHK(hipEventElapsedTime(&f1, ev[1], o->fastV1 ? ev[10] : ev[2]));
    ms7[1] = o->fastV1 ? f1 : f0 + f1;
    HIPCHK(hipEventElapsedTime(&ms7[2], ev[2], ev[3]));
    HIPCHK(hipEventElapsedTime(&ms7[3], ev[3], ev[4]));
    HIPCHK(hipEventElapsedTime(&ms7[4], ev[11], ev[9]));     // blur (stream 2, beside the quadtree)
    HIPCHK(hipEventElapsedTime(&ms7[5], ev[5], ev[6]));
    HIPCHK(hipEventElapsedTime(&ms7[6], ev[0], ev[6]));
    HIPCHK(hipEventElapsedTime(&ms7[7], ev[0], ev[2]));      // wall span of the pyramid+FAST pass (both streams)
    if (o->blurEarly && !o->serial) {                        // the blur runs inside the pass: it ends with the later of FAST and blur
        float b = 0;
        HIPCHK(hipEventElapsedTime(&b, ev[0], ev[9]));
        ms7[7] = std::max(ms7[7], b);
    }
    return ORBX_OK;
}

int orbx_last_timings(orbx_t* o, float* ms7) {   // 8 floats, see include/orbx.h
    if (!o || !o->timed) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipStreamSynchronize(o->stream2));
    if (o->graphMode) return orbx_mean_timings(o, ms7, nullptr);
    return timings_of(o, o->ev, ms7);
}

int orbx_mean_timings(orbx_t* o, float* ms8, int* nsamples) {
    if (!o || !o->timed) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipStreamSynchronize(o->stream2));
    int n = 0;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (o->graphMode) {                                      // replays refresh their slot's own stamps: the latest replay of every slot
        unsigned long long st[orbx::kAllSlots * 4];
        HIPCHK(hipMemcpy(st, o->dStamps, sizeof st, hipMemcpyDeviceToHost));
        const double msPerTick = 1.0 / (double)o->wallClockKHz;
        for (int k = 0; k < orbx::kAllSlots; ++k) {
            const orbx::GraphSlot& G = o->gs[k];
            if (!G.exec || G.launches < 1 || G.nimg < 1) continue;
            const unsigned long long* t = st + 4 * k;
            unsigned long long passEnd = t[1];
            if (o->blurEarly && !o->serial && t[2] > passEnd) passEnd = t[2];
            acc[1] += (double)(t[1] - t[0]) * msPerTick;     // start -> end of the last FAST launch (no per-stage events in a replay)
            acc[6] += (double)(t[3] - t[0]) * msPerTick;
            acc[7] += (double)(passEnd - t[0]) * msPerTick;
            ++n;
        }
    } else {
        n = (int)std::min<long>(o->nEnq, orbx::kRing);
        for (int k = 0; k < n; ++k) {
            float t[8];
            int rc = timings_of(o, o->evr[(o->nEnq - 1 - k) % orbx::kRing], t);
            if (rc) return rc;
            for (int i = 0; i < 8; ++i) acc[i] += t[i];
        }
    }
    for (int i = 0; i < 8; ++i) ms8[i] = n ? (float)(acc[i] / n) : 0.f;
    if (nsamples) *nsamples = n;
    return ORBX_OK;
}

int orbx_blur_in_pass(const orbx_t* o) { return o && o->blurEarly && !o->serial ? 1 : 0; }

int orbx_set_stage_timing(orbx_t* o, int on) {
    if (!o) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream)); HIPCHK(hipStreamSynchronize(o->stream2));
    o->stageTiming = on != 0;
    o->nEnq = 0; o->timed = false;                            // the ring restarts: sets recorded under the other mode are not mixed in
    return ORBX_OK;
}

int orbx_stream_wait_results(orbx_t* o, void* other_stream) {
    if (!o) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipEventRecord(o->evDone, o->stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)other_stream, o->evDone, 0));
    return ORBX_OK;
}

int orbx_guard_results(orbx_t* o, void* reader_stream) {
    if (!o) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipEventRecord(o->evGuard, (hipStream_t)reader_stream));
    o->guardPending = true;
    return ORBX_OK;
}

int orbx_stream_wait_other(orbx_t* o, void* other_stream) {
    if (!o) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipEventRecord(o->evDone, (hipStream_t)other_stream));
    HIPCHK(hipStreamWaitEvent(o->stream, o->evDone, 0));
    return ORBX_OK;
}

// Scratch of the ingest entry points: grow-only, so a call needs no allocation, no free and no host sync (the kernels and the
// small table uploads are ordered by the extractor's stream; a later call reuses the block only behind them in that stream).
static u8* ingest_scratch(orbx* o, size_t bytes) {
    if (bytes <= o->capIngest && o->dIngest) return o->dIngest;
    (void)hipStreamSynchronize(o->stream);
    if (o->dIngest) (void)hipFree(o->dIngest);
    o->dIngest = nullptr; o->capIngest = 0;
    if (hipMalloc((void**)&o->dIngest, bytes) != hipSuccess) { set_err("hipMalloc of %zu B failed", bytes); return nullptr; }
    o->capIngest = bytes;
    return o->dIngest;
}

int orbx_gray_from_color(orbx_t* o, const uint8_t* const* src, int src_space, int nimg, int w, int h, int src_stride,
                         int channels, int blue_first, int coef_bits, uint8_t* const* dst, int dst_stride) {
    if (!o || !src || !dst || nimg < 1 || w < 1 || h < 1 || (channels != 3 && channels != 4) || (coef_bits != 14 && coef_bits != 15) ||
        src_stride < w * channels || dst_stride < w) return ORBX_E_INVALID;
    for (int i = 0; i < nimg; ++i) if (!src[i] || !dst[i]) return ORBX_E_EMPTY;
    HIPCHK(hipSetDevice(o->device));
    const int ry = coef_bits == 14 ? 4899 : 9798, gy = coef_bits == 14 ? 9617 : 19235, by = coef_bits == 14 ? 1868 : 3735;
    const int c0 = blue_first ? by : ry, c2 = blue_first ? ry : by;
    hipStream_t st = o->stream;
    const size_t imgBytes = (size_t)src_stride * h, tab = (sizeof(u8*) * nimg + 255) & ~(size_t)255;
    u8* sc = ingest_scratch(o, 2 * tab + (src_space != ORBX_DEVICE ? imgBytes * nimg : 0));
    if (!sc) return ORBX_E_HIP;
    const u8** dS = (const u8**)sc; u8** dD = (u8**)(sc + tab); u8* stage = sc + 2 * tab;
    std::vector<const u8*> hs(nimg);
    for (int i = 0; i < nimg; ++i) {
        if (src_space != ORBX_DEVICE) {
            HIPCHK(hipMemcpyAsync(stage + imgBytes * i, src[i], imgBytes, hipMemcpyHostToDevice, st));
            hs[i] = stage + imgBytes * i;
        } else hs[i] = src[i];
    }
    HIPCHK(hipMemcpyAsync((void*)dS, hs.data(), sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)dD, dst, sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gray, dim3((w + 1023) / 1024, h, nimg), dim3(256), 0, st, dS, w, h, src_stride, channels, c0, gy, c2, coef_bits, dD, dst_stride);
    HIPCHK(hipGetLastError());
    return ORBX_OK;
}

int orbx_remap_linear(orbx_t* o, const uint8_t* const* src, int nimg, int sw, int sh, int src_stride, const float* mapx, const float* mapy,
                      int dw, int dh, uint8_t* const* dst, int dst_stride) {
    if (!o || !src || !dst || !mapx || !mapy || nimg < 1 || sw < 1 || sh < 1 || dw < 1 || dh < 1 || src_stride < sw || dst_stride < dw) return ORBX_E_INVALID;
    for (int i = 0; i < nimg; ++i) if (!src[i] || !dst[i]) return ORBX_E_EMPTY;
    HIPCHK(hipSetDevice(o->device));
    hipStream_t st = o->stream;
    const size_t tab = (sizeof(u8*) * nimg + 255) & ~(size_t)255;
    u8* sc = ingest_scratch(o, 2 * tab);
    if (!sc) return ORBX_E_HIP;
    const u8** dS = (const u8**)sc; u8** dD = (u8**)(sc + tab);
    HIPCHK(hipMemcpyAsync((void*)dS, src, sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)dD, dst, sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_remap, dim3((dw + 1023) / 1024, dh, nimg), dim3(256), 0, st, dS, sw, sh, src_stride, mapx, mapy, dw, dh, dD, dst_stride);
    HIPCHK(hipGetLastError());
    return ORBX_OK;
}

int orbx_clahe(orbx_t* o, const uint8_t* const* src, int nimg, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
               uint8_t* const* dst, int dst_stride) {
    if (!o || !src || !dst || nimg < 1 || w < 1 || h < 1 || tiles_x < 1 || tiles_y < 1 || src_stride < w || dst_stride < w) return ORBX_E_INVALID;
    for (int i = 0; i < nimg; ++i) if (!src[i] || !dst[i]) return ORBX_E_EMPTY;
    ClaheGeom G;
    G.w = w; G.h = h; G.tilesX = tiles_x; G.tilesY = tiles_y;
    int ew = w, eh = h;                                         // clahe.cpp extends to the right / bottom when the size is not a multiple
    if (w % tiles_x != 0 || h % tiles_y != 0) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    if (ew > 2 * w - 1 || eh > 2 * h - 1) { set_err("CLAHE tile grid %dx%d too fine for a %dx%d image", tiles_x, tiles_y, w, h); return ORBX_E_UNSUPPORTED; }
    G.tw = ew / tiles_x; G.th = eh / tiles_y;
    const int area = G.tw * G.th;
    G.clip = 0;
    if (clip_limit > 0.0) { G.clip = (int)(clip_limit * area / 256); G.clip = std::max(G.clip, 1); }
    G.lutScale = (float)255 / area; G.invTw = 1.0f / G.tw; G.invTh = 1.0f / G.th;
    HIPCHK(hipSetDevice(o->device));
    hipStream_t st = o->stream;
    const size_t tab = (sizeof(u8*) * nimg + 255) & ~(size_t)255;
    u8* sc = ingest_scratch(o, 2 * tab + (size_t)nimg * tiles_x * tiles_y * 256);
    if (!sc) return ORBX_E_HIP;
    const u8** dS = (const u8**)sc; u8** dD = (u8**)(sc + tab); u8* dLut = sc + 2 * tab;
    HIPCHK(hipMemcpyAsync((void*)dS, src, sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)dD, dst, sizeof(u8*) * nimg, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y, nimg), dim3(256), 0, st, dS, src_stride, G, dLut);
    hipLaunchKernelGGL(k_clahe_apply, dim3((w + 1023) / 1024, h, nimg), dim3(256), 0, st, dS, src_stride, G, dLut, dD, dst_stride);
    HIPCHK(hipGetLastError());
    return ORBX_OK;
}

int64_t orbx_algorithmic_bytes(const orbx_t* o, int64_t* fused_lower_bound) {
    if (!o) return ORBX_E_INVALID;
    if (fused_lower_bound) *fused_lower_bound = o->fusedBytes;
    return o->algBytes;
}

// internal (not part of include/orbx.h): device pointers of the pyramid levels of batch slot `frame`, used by the
// stereo matcher living in the same library (orbm_stereo_matches)
int orbx_internal_levels(orbx_t* o, int frame, int* nlevels, const uint8_t** ptr, int* pitch, int* w, int* h,
                         float* sf, float* isf, int* device) {
    if (!o || frame < 0 || frame >= o->lastBatch) return ORBX_E_INVALID;
    HIPCHK(hipSetDevice(o->device));
    HIPCHK(hipStreamSynchronize(o->stream));
    *nlevels = o->nlevels; *device = o->device;
    for (int l = 0; l < o->nlevels; ++l) {
        const LevelDesc& D = o->g.lv[l];
        if (l == 0) {
            ptr[0] = o->hL0Ptr[frame];
            pitch[0] = (ptr[0] >= o->dL0 && ptr[0] < o->dL0 + o->capL0) ? o->l0pitch : o->lastL0Pitch;
        } else { ptr[l] = o->dPyr + (size_t)frame * o->g.pyrFrameBytes + D.off; pitch[l] = D.pitch; }
        w[l] = D.w; h[l] = D.h; sf[l] = o->sf[l]; isf[l] = o->invsf[l];
    }
    return ORBX_OK;
}

// internal (not part of include/orbx.h): where the batch's pyramids live, for the batched stereo matcher in the same library
// (orbm_stereo_batch_async).  No sync: the caller orders its kernels behind the extraction on a stream.
int orbx_internal_batch_layout(orbx_t* o, const uint8_t* const** l0tab, int* l0pitch, const uint8_t** pyr, size_t* frameBytes,
                               int* nlevels, int* off, int* pitch, int* w, int* h, float* sf, float* isf, int* device, int* maxBatch, int* kpCap) {
    if (!o || !o->curW) return ORBX_E_INVALID;
    *l0tab = o->dL0Ptr; *l0pitch = o->lastL0Pitch ? o->lastL0Pitch : o->l0pitch; *pyr = o->dPyr; *frameBytes = o->g.pyrFrameBytes;
    *nlevels = o->nlevels; *device = o->device; *maxBatch = o->maxBatch; *kpCap = o->g.kpCap;
    for (int l = 0; l < o->nlevels; ++l) { const LevelDesc& D = o->g.lv[l]; off[l] = D.off; pitch[l] = D.pitch; w[l] = D.w; h[l] = D.h; sf[l] = o->sf[l]; isf[l] = o->invsf[l]; }
    return ORBX_OK;
}

void* orbx_stream(const orbx_t* o) { return o ? (void*)o->stream : nullptr; }

void* orbx_dev_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_err("hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void orbx_dev_free(void* p) { if (p) (void)hipFree(p); }
int orbx_memcpy_h2d(void* dst, const void* src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return ORBX_OK; }
int orbx_memcpy_d2h(void* dst, const void* src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return ORBX_OK; }

}  // extern "C"
