// orbref_extract.cpp -- CPU ORACLE for ORBextractor (TEST INFRASTRUCTURE ONLY, see orbref.h).
//
// Restates, function by function, the reference's src/ORBextractor.cc and the OpenCV 4.x
// primitives it calls.  PARITY UNPINNED (no golden vectors exist in the reference; OpenCV absent).
// Built with -ffp-contract=off: all float expressions are evaluated exactly as written.
//
// Normative choices where the reference itself is under-determined (DESIGN.md "Oracle"):
//   * DistributeOctTree tie-break (reference sorts on heap pointers, ORBextractor.cc:920-927):
//     equal-count nodes are ordered by creation sequence; the most recently created splits first.
//   * GaussianBlur: OpenCV >= 4.3 bit-exact fixed-point kernel [18,34,48,56,48,34,18]/256.
//   * cos/sin of the keypoint angle: (float)cos((double)angle) (glibc cosf is not correctly rounded
//     by contract; the double evaluation rounded once is).
#include "orbref.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <list>
#include <vector>

namespace {

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;   // ORBextractor.cc:76-78

const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

inline int cv_round(float v)  { return (int)lrintf(v); }   // round-half-even (default FP env)
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(float v)  { return (int)floorf(v); }
inline int cv_ceil(float v)   { return (int)ceilf(v); }

struct Img {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    void alloc(int W, int H) { w = W; h = H; d.assign((size_t)W * H, 0); }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
};

struct Cand { int x, y, r; };          // integer-valued floats in the reference
struct LevelKp { int x, y, r; float angle; };

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2 (OpenCV core/mathfuncs_core)
// ---------------------------------------------------------------------------------------------
float fast_atan2(float y, float x) {
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------------------------------------
// cv::resize INTER_LINEAR, CV_8UC1, fixed point
// ---------------------------------------------------------------------------------------------
inline short sat_short_round(float v) {
    int r = cv_round(v);
    return (short)std::min(32767, std::max(-32768, r));
}

void resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> xa(2 * dw), ya(2 * dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        xa[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ya[2 * dy] = sat_short_round((1.f - fy) * 2048.f);
        ya[2 * dy + 1] = sat_short_round(fy * 2048.f);
    }
    std::vector<int> h0(dw), h1(dw);
    int prev0 = -2, prev1 = -2;   // cached source rows held in h0/h1
    auto hrow = [&](int sy, std::vector<int>& out) {
        const uint8_t* s = src + (size_t)sy * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int s1 = (sx + 1 < sw) ? s[sx + 1] : 0;    // second tap has weight 0 when clamped
            out[dx] = s[sx] * xa[2 * dx] + s1 * xa[2 * dx + 1];
        }
    };
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        if (sy0 == prev1) { std::swap(h0, h1); std::swap(prev0, prev1); }
        if (sy0 != prev0) { hrow(sy0, h0); prev0 = sy0; }
        if (sy1 == prev0) { h1 = h0; prev1 = sy1; }
        else if (sy1 != prev1) { hrow(sy1, h1); prev1 = sy1; }
        const int b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
        uint8_t* d = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2;
            d[dx] = (uint8_t)std::min(255, std::max(0, v));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), CV_8UC1 bit-exact path
// ---------------------------------------------------------------------------------------------
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

void gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    static const int K[7] = {18, 34, 48, 56, 48, 34, 18};
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint16_t* o = hbuf.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            if (x >= 3 && x + 3 < w) for (int k = 0; k < 7; ++k) acc += K[k] * s[x + k - 3];
            else for (int k = 0; k < 7; ++k) acc += K[k] * s[reflect101(x + k - 3, w)];
            o[x] = (uint16_t)acc;                       // Q8.8, max 255*256
        }
    }
    for (int y = 0; y < h; ++y) {
        const uint16_t* r[7];
        for (int k = 0; k < 7; ++k) r[k] = hbuf.data() + (size_t)reflect101(y + k - 3, h) * w;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; ++k) acc += (uint32_t)K[k] * r[k][x];
            d[x] = (uint8_t)((acc + 32768u) >> 16);      // Q16.16 -> u8, round half up
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST TYPE_9_16 with non-max suppression (OpenCV features2d/fast.cpp)
// ---------------------------------------------------------------------------------------------
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16>: max(threshold, A, B) - 1
int corner_score(const uint8_t* p, const int* off, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; ++k) d[k] = v - p[off[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], std::min(d[k + 2], d[k + 3]));
        if (a <= a0) continue;
        for (int t = 4; t <= 8; ++t) a = std::min(a, d[k + t]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(std::max(d[k + 1], d[k + 2]), std::max(d[k + 3], std::max(d[k + 4], d[k + 5])));
        if (b >= b0) continue;
        for (int t = 6; t <= 8; ++t) b = std::max(b, d[k + t]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

struct FastKp { int x, y, score; };

void fast9_16_nms(const uint8_t* img, int w, int h, int stride, int threshold, std::vector<FastKp>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    int off[25];
    for (int k = 0; k < 25; ++k) off[k] = kRingDy[k & 15] * stride + kRingDx[k & 15];
    uint8_t tab[512];
    for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    std::vector<uint8_t> sbuf((size_t)3 * w, 0);
    std::vector<int> cbuf((size_t)3 * (w + 1), 0);
    uint8_t* buf[3] = {sbuf.data(), sbuf.data() + w, sbuf.data() + 2 * w};
    int* cpb[3] = {cbuf.data() + 1, cbuf.data() + (w + 1) + 1, cbuf.data() + 2 * (w + 1) + 1};

    for (int i = 3; i < h - 2; ++i) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpb[(i - 3) % 3];
        std::memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; ++j, ++ptr) {
                const int v = ptr[0];
                const uint8_t* t = tab + 255 - v;
                int d = t[ptr[off[0]]] | t[ptr[off[8]]];
                if (d == 0) continue;
                d &= t[ptr[off[2]]] | t[ptr[off[10]]];
                d &= t[ptr[off[4]]] | t[ptr[off[12]]];
                d &= t[ptr[off[6]]] | t[ptr[off[14]]];
                if (d == 0) continue;
                d &= t[ptr[off[1]]] | t[ptr[off[9]]];
                d &= t[ptr[off[3]]] | t[ptr[off[11]]];
                d &= t[ptr[off[5]]] | t[ptr[off[13]]];
                d &= t[ptr[off[7]]] | t[ptr[off[15]]];
                bool corner = false;
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < 25; ++k) {
                        if (ptr[off[k]] < vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < 25; ++k) {
                        if (ptr[off[k]] > vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score(ptr, off, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        const int* cp = cpb[(i - 4 + 3) % 3];
        const int nc = cp[-1];
        for (int k = 0; k < nc; ++k) {
            int j = cp[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back({j, i - 1, score});
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ExtractorNode / DistributeOctTree (ORBextractor.cc:602-674, 688-1034)
// ---------------------------------------------------------------------------------------------
struct Node {
    std::vector<int> keys;                 // indices into the candidate array, order preserved
    int ULx = 0, ULy = 0, URx = 0, URy = 0, BLx = 0, BLy = 0, BRx = 0, BRy = 0;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;                          // creation sequence (normative tie-break, replaces heap address)
};

void divide_node(const Node& p, const Cand* c, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (int k : p.keys) {
        const float x = (float)c[k].x, y = (float)c[k].y;
        if (x < n1.URx) { if (y < n1.BRy) n1.keys.push_back(k); else n3.keys.push_back(k); }
        else if (y < n1.BRy) n2.keys.push_back(k);
        else n4.keys.push_back(k);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

typedef std::pair<int, Node*> SizeNode;
bool size_node_less(const SizeNode& a, const SizeNode& b) {
    if (a.first != b.first) return a.first < b.first;
    return a.second->seq < b.second->seq;
}

// returns indices (into c) of the selected candidates, in the reference's output order
void distribute_oct_tree(const Cand* c, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& result) {
    result.clear();
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return;   // reference divides by zero here (aspect < 0.5); oracle rejects (documented)
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; ++i) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        ini[i] = &lNodes.back();
    }
    for (int i = 0; i < n; ++i) ini[(int)((float)c[i].x / hX)]->keys.push_back(i);

    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = lNodes.erase(lit);
        else ++lit;
    }

    bool finish = false;
    std::vector<SizeNode> vSize;
    auto push_child = [&](Node& ch, int* nToExpand) {
        if (ch.keys.empty()) return;
        ch.seq = seq++;
        lNodes.push_front(ch);
        if (ch.keys.size() > 1) {
            if (nToExpand) ++*nToExpand;
            vSize.push_back(std::make_pair((int)ch.keys.size(), &lNodes.front()));
            lNodes.front().lit = lNodes.begin();
        }
    };
    while (!finish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSize.clear();
        while (lit != lNodes.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, c, n1, n2, n3, n4);
            push_child(n1, &nToExpand); push_child(n2, &nToExpand);
            push_child(n3, &nToExpand); push_child(n4, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            finish = true;
        } else if ((int)lNodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeNode> prev = vSize;
                vSize.clear();
                std::sort(prev.begin(), prev.end(), size_node_less);
                for (int j = (int)prev.size() - 1; j >= 0; --j) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, c, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    lNodes.erase(prev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) finish = true;
            }
        }
    }
    result.reserve(lNodes.size());
    for (auto& nd : lNodes) {
        int best = nd.keys[0];
        int maxR = c[best].r;
        for (size_t k = 1; k < nd.keys.size(); ++k)
            if (c[nd.keys[k]].r > maxR) { best = nd.keys[k]; maxR = c[best].r; }
        result.push_back(best);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// the extractor object
// ---------------------------------------------------------------------------------------------
struct orbref {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;                        // include/ORBextractor.h:96 (double member from float arg)
    std::vector<float> sf, invsf, sig2, invsig2;
    std::vector<int> nfeat;
    int umax[16];
    std::vector<Img> pyr, blurred;
    std::vector<std::vector<Cand>> cands;
    std::vector<std::vector<LevelKp>> lkps;
    double ms[6] = {0, 0, 0, 0, 0, 0};
};

extern "C" {

orbref_t* orbref_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    if (nlevels < 1 || nfeatures < 0) return nullptr;
    orbref* o = new orbref;
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->iniTh = ini_th; o->minTh = min_th;
    o->scaleFactor = scale_factor;
    o->sf.resize(nlevels); o->sig2.resize(nlevels); o->invsf.resize(nlevels); o->invsig2.resize(nlevels);
    o->sf[0] = 1.0f; o->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        o->sf[i] = (float)(o->sf[i - 1] * o->scaleFactor);          // float*double -> float (ORBextractor.cc:488)
        o->sig2[i] = o->sf[i] * o->sf[i];
    }
    for (int i = 0; i < nlevels; ++i) { o->invsf[i] = 1.0f / o->sf[i]; o->invsig2[i] = 1.0f / o->sig2[i]; }
    o->nfeat.resize(nlevels);
    float factor = (float)(1.0f / o->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; ++level) {
        o->nfeat[level] = cv_round(nDesired);
        sum += o->nfeat[level];
        nDesired *= factor;
    }
    o->nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax (ORBextractor.cc:542-570)
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    o->pyr.resize(nlevels); o->blurred.resize(nlevels); o->cands.resize(nlevels); o->lkps.resize(nlevels);
    return o;
}

void orbref_destroy(orbref_t* o) { delete o; }

void orbref_tables(const orbref_t* o, float* sf, float* inv_sf, float* sig2, float* inv_sig2, int* nfeat, int* umax) {
    for (int i = 0; i < o->nlevels; ++i) {
        if (sf) sf[i] = o->sf[i];
        if (inv_sf) inv_sf[i] = o->invsf[i];
        if (sig2) sig2[i] = o->sig2[i];
        if (inv_sig2) inv_sig2[i] = o->invsig2[i];
        if (nfeat) nfeat[i] = o->nfeat[i];
    }
    if (umax) for (int i = 0; i < 16; ++i) umax[i] = o->umax[i];
}

static int ic_angle_moments(const Img& im, int cx, int cy, const int* umax, int* m01_out, int* m10_out) {
    int m_01 = 0, m_10 = 0;
    const int step = im.w;
    const uint8_t* center = im.row(cy) + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    *m01_out = m_01; *m10_out = m_10;
    return 0;
}

int orbref_extract(orbref_t* o, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                   orbref_kp_t* kps, uint8_t* desc, int cap, int* mono_index) {
    if (!img || w <= 0 || h <= 0) return -1;                               // ORBextractor.cc:1538-1539
    const int L = o->nlevels;
    // ---- ComputePyramid (ORBextractor.cc:1664-1717); the 19-px border is never read, so not kept
    double t0 = now_ms();
    for (int level = 0; level < L; ++level) {
        float scale = o->invsf[level];
        int sw = cv_round((float)w * scale), sh = cv_round((float)h * scale);
        // FAST grid needs width-32 >= 35 (nCols>=1) -> reference divides by zero below that
        if (sw - 2 * (EDGE_THRESHOLD - 3) < 35 || sh - 2 * (EDGE_THRESHOLD - 3) < 35) return -3;
        o->pyr[level].alloc(sw, sh);
        if (level == 0) {
            for (int y = 0; y < h; ++y) std::memcpy(o->pyr[0].row(y), img + (size_t)y * stride, w);
        } else {
            const Img& p = o->pyr[level - 1];
            resize_linear(p.d.data(), p.w, p.h, p.w, o->pyr[level].d.data(), sw, sh, sw);
        }
    }
    double t1 = now_ms(); o->ms[0] += t1 - t0;

    // ---- ComputeKeyPointsOctTree (ORBextractor.cc:1038-1185)
    const float W = 35;
    std::vector<FastKp> cell;
    std::vector<int> sel;
    int total = 0;
    for (int level = 0; level < L; ++level) {
        double ta = now_ms();
        const Img& im = o->pyr[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<Cand>& vToDistribute = o->cands[level];
        vToDistribute.clear();
        vToDistribute.reserve((size_t)o->nfeatures * 10);
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
                const uint8_t* sub = im.row(y0) + x0;
                fast9_16_nms(sub, cw, ch, im.w, o->iniTh, cell);
                if (cell.empty()) fast9_16_nms(sub, cw, ch, im.w, o->minTh, cell);
                for (const FastKp& k : cell) vToDistribute.push_back({k.x + j * wCell, k.y + i * hCell, k.score});
            }
        }
        double tb = now_ms(); o->ms[1] += tb - ta;
        distribute_oct_tree(vToDistribute.data(), (int)vToDistribute.size(), minBorderX, maxBorderX,
                            minBorderY, maxBorderY, o->nfeat[level], sel);
        std::vector<LevelKp>& lk = o->lkps[level];
        lk.clear();
        for (int idx : sel)
            lk.push_back({vToDistribute[idx].x + minBorderX, vToDistribute[idx].y + minBorderY, vToDistribute[idx].r, -1.f});
        total += (int)lk.size();
        o->ms[2] += now_ms() - tb;
    }
    // ---- computeOrientation (ORBextractor.cc:580-591, 91-138)
    double t2 = now_ms();
    for (int level = 0; level < L; ++level)
        for (LevelKp& k : o->lkps[level]) {
            int m01, m10;
            ic_angle_moments(o->pyr[level], k.x, k.y, o->umax, &m01, &m10);
            k.angle = fast_atan2((float)m01, (float)m10);
        }
    o->ms[3] += now_ms() - t2;

    if (total > cap) return -2;
    // ---- descriptors + output ordering (ORBextractor.cc:1590-1655)
    int monoIndex = 0, stereoIndex = total - 1;
    for (int level = 0; level < L; ++level) {
        std::vector<LevelKp>& lk = o->lkps[level];
        if (lk.empty()) { o->blurred[level].alloc(0, 0); continue; }
        double ta = now_ms();
        const Img& im = o->pyr[level];
        Img& bl = o->blurred[level];
        bl.alloc(im.w, im.h);
        gauss7(im.d.data(), im.w, im.h, im.w, bl.d.data(), im.w);
        double tb = now_ms(); o->ms[4] += tb - ta;
        const float scale = o->sf[level];
        const float patch = (float)(int)(PATCH_SIZE * o->sf[level]);     // int scaledPatchSize (ORBextractor.cc:1163)
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        for (const LevelKp& k : lk) {
            uint8_t d[32];
            float angle = k.angle * factorPI;
            float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
            const uint8_t* center = bl.row(k.y) + k.x;
            const int step = bl.w;
            const int8_t* p = kPattern;
            for (int i = 0; i < 32; ++i) {
                int val = 0;
                for (int bit = 0; bit < 8; ++bit, p += 4) {
                    float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
                    int t0 = center[cv_round(x0 * b + y0 * a) * step + cv_round(x0 * a - y0 * b)];
                    int t1 = center[cv_round(x1 * b + y1 * a) * step + cv_round(x1 * a - y1 * b)];
                    val |= (t0 < t1) << bit;
                }
                d[i] = (uint8_t)val;
            }
            orbref_kp_t kp;
            kp.x = (float)k.x; kp.y = (float)k.y;
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            kp.size = patch; kp.angle = k.angle; kp.response = (float)k.r; kp.octave = level; kp.class_id = -1;
            int slot;
            if (kp.x >= lap0 && kp.x <= lap1) slot = stereoIndex--; else slot = monoIndex++;
            kps[slot] = kp;
            std::memcpy(desc + (size_t)slot * 32, d, 32);
        }
        o->ms[5] += now_ms() - tb;
    }
    if (mono_index) *mono_index = monoIndex;
    return total;
}

int orbref_level_size(const orbref_t* o, int level, int* w, int* h) {
    if (level < 0 || level >= o->nlevels) return -1;
    *w = o->pyr[level].w; *h = o->pyr[level].h; return 0;
}
static int copy_img(const Img& im, uint8_t* dst, int dst_stride) {
    for (int y = 0; y < im.h; ++y) std::memcpy(dst + (size_t)y * dst_stride, im.row(y), im.w);
    return 0;
}
int orbref_level_image(const orbref_t* o, int level, uint8_t* dst, int dst_stride) {
    if (level < 0 || level >= o->nlevels) return -1;
    return copy_img(o->pyr[level], dst, dst_stride);
}
int orbref_level_blurred(const orbref_t* o, int level, uint8_t* dst, int dst_stride) {
    if (level < 0 || level >= o->nlevels) return -1;
    if (o->blurred[level].w == 0) return 1;
    return copy_img(o->blurred[level], dst, dst_stride);
}
int orbref_level_candidates(const orbref_t* o, int level, int32_t* xyr, int cap) {
    if (level < 0 || level >= o->nlevels) return -1;
    const auto& c = o->cands[level];
    int n = (int)c.size();
    for (int i = 0; i < n && i < cap; ++i) { xyr[3 * i] = c[i].x; xyr[3 * i + 1] = c[i].y; xyr[3 * i + 2] = c[i].r; }
    return n;
}
int orbref_level_keypoints(const orbref_t* o, int level, int32_t* xyr, float* angle, int cap) {
    if (level < 0 || level >= o->nlevels) return -1;
    const auto& c = o->lkps[level];
    int n = (int)c.size();
    for (int i = 0; i < n && i < cap; ++i) {
        xyr[3 * i] = c[i].x; xyr[3 * i + 1] = c[i].y; xyr[3 * i + 2] = c[i].r;
        if (angle) angle[i] = c[i].angle;
    }
    return n;
}
void orbref_stage_ms(const orbref_t* o, double* out6) { for (int i = 0; i < 6; ++i) out6[i] = o->ms[i]; }
void orbref_stage_reset(orbref_t* o) { for (int i = 0; i < 6; ++i) o->ms[i] = 0; }

float orbref_fast_atan2(float y, float x) { return fast_atan2(y, x); }

int orbref_fast(const uint8_t* img, int w, int h, int stride, int threshold, int32_t* xys, int cap) {
    std::vector<FastKp> out;
    fast9_16_nms(img, w, h, stride, threshold, out);
    int n = (int)out.size();
    for (int i = 0; i < n && i < cap; ++i) { xys[3 * i] = out[i].x; xys[3 * i + 1] = out[i].y; xys[3 * i + 2] = out[i].score; }
    return n;
}

int orbref_fast_score(const uint8_t* img, int stride, int x, int y) {
    int off[25];
    for (int k = 0; k < 25; ++k) off[k] = kRingDy[k & 15] * stride + kRingDx[k & 15];
    return corner_score(img + (size_t)y * stride + x, off, -1000);   // threshold below any A,B -> max(A,B)-1
}

void orbref_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    resize_linear(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orbref_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    gauss7(src, w, h, sstride, dst, dstride);
}
int orbref_distribute(const int32_t* xyr, int n, int minX, int maxX, int minY, int maxY, int N, int32_t* out_idx, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; ++i) c[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    std::vector<int> sel;
    distribute_oct_tree(c.data(), n, minX, maxX, minY, maxY, N, sel);
    int m = (int)sel.size();
    for (int i = 0; i < m && i < cap; ++i) out_idx[i] = sel[i];
    return m;
}
const int8_t* orbref_pattern(void) { return kPattern; }

}  // extern "C"

// internal accessor for the stereo restatement (orbref_frame.cpp): level image of the last extract call
extern "C" const uint8_t* orbref_level_data(const orbref_t* o, int level, int* w, int* h) {
    if (!o || level < 0 || level >= o->nlevels) return nullptr;
    *w = o->pyr[level].w; *h = o->pyr[level].h;
    return o->pyr[level].d.data();
}
extern "C" int orbref_nlevels(const orbref_t* o) { return o->nlevels; }
