// orbref_frame.cpp -- CPU ORACLE for the Frame grid / ORBmatcher searches on flattened arrays
// (TEST INFRASTRUCTURE ONLY, see orbref.h).  Each function keeps the reference's loop structure:
// GetFeaturesInArea -> DescriptorDistance -> best/second bookkeeping -> rotation histogram.
// PARITY UNPINNED (no golden vectors in the reference).  Built with -ffp-contract=off.
#include "orbref.h"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

extern "C" const uint8_t* orbref_level_data(const orbref_t* o, int level, int* w, int* h);
extern "C" int orbref_nlevels(const orbref_t* o);

namespace {
const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;     // ORBmatcher.cc:36-38
const int GC = ORBREF_GRID_COLS, GR = ORBREF_GRID_ROWS;

void features_in_area(const orbref_frame_t* f, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) {
    out.clear();                                               // Frame.cc:784-871
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - f->min_x - factorX) * f->inv_w));
    if (nMinCellX >= GC) return;
    const int nMaxCellX = std::min(GC - 1, (int)std::ceil((x - f->min_x + factorX) * f->inv_w));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - f->min_y - factorY) * f->inv_h));
    if (nMinCellY >= GR) return;
    const int nMaxCellY = std::min(GR - 1, (int)std::ceil((y - f->min_y + factorY) * f->inv_h));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GR + iy;
            for (int j = f->grid_start[c]; j < f->grid_start[c + 1]; ++j) {
                const int k = f->grid_idx[j];
                const orbref_kp_t& kp = f->kps[k];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) out.push_back(k);
            }
        }
}

struct RotHist {
    std::vector<int> bins[HISTO_LENGTH];
    void add(float a1, float a2, float factor, int idx) {
        float rot = a1 - a2;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        bins[bin].push_back(idx);
    }
    void maxima(int* ind) const {
        int sz[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; ++i) sz[i] = (int)bins[i].size();
        orbref_three_maxima(sz, HISTO_LENGTH, ind);
    }
};
}  // namespace

extern "C" {

int orbref_grid_build(const orbref_kp_t* kps, int n, float min_x, float min_y, float inv_w, float inv_h,
                      int32_t* grid_start, int32_t* grid_idx) {
    std::vector<std::vector<int>> cells((size_t)GC * GR);
    int placed = 0;
    for (int i = 0; i < n; ++i) {
        const int posX = (int)std::round((kps[i].x - min_x) * inv_w);       // Frame.cc:888-889 (round, not floor)
        const int posY = (int)std::round((kps[i].y - min_y) * inv_h);
        if (posX < 0 || posX >= GC || posY < 0 || posY >= GR) continue;
        cells[(size_t)posX * GR + posY].push_back(i);
        ++placed;
    }
    int o = 0;
    for (int c = 0; c < GC * GR; ++c) {
        grid_start[c] = o;
        for (int k : cells[c]) grid_idx[o++] = k;
    }
    grid_start[GC * GR] = o;
    return placed;
}

int orbref_features_in_area(const orbref_frame_t* f, float x, float y, float r, int min_level, int max_level, int32_t* out, int cap) {
    std::vector<int> v;
    features_in_area(f, x, y, r, min_level, max_level, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

int orbref_search_by_projection_frame(const orbref_frame_t* cur, const uint8_t* cur_blocked, const float* sf,
                                      int nq, const uint8_t* valid, const float* u, const float* v, const float* invzc,
                                      const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                      float th, int bForward, int bBackward, float mbf, int check_ori, int32_t* match) {
    int nmatches = 0;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // ORBmatcher.cc:2480
    std::vector<uint8_t> blocked(cur_blocked, cur_blocked + cur->n);
    for (int i = 0; i < cur->n; ++i) match[i] = -1;
    std::vector<int> vIndices2;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) continue;
        const int nLastOctave = octave[i];
        const float radius = th * sf[nLastOctave];
        if (bForward) features_in_area(cur, u[i], v[i], radius, nLastOctave, -1, vIndices2);
        else if (bBackward) features_in_area(cur, u[i], v[i], radius, 0, nLastOctave, vIndices2);
        else features_in_area(cur, u[i], v[i], radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (blocked[i2]) continue;                                        // :2565-2567
            if (cur->uright && cur->uright[i2] > 0) {                         // :2569-2576
                const float ur = u[i] - mbf * invzc[i];
                const float er = std::fabs(ur - cur->uright[i2]);
                if (er > radius) continue;
            }
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, cur->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            if (match[bestIdx2] < 0) {} // (re-assignment of an unblocked slot simply overwrites, as in the reference)
            match[bestIdx2] = i;
            if (mp_obs[i]) blocked[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], cur->kps[bestIdx2].angle, factor, bestIdx2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < HISTO_LENGTH; ++b)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int idx : rh.bins[b]) { match[idx] = -2; nmatches--; }   // assigned, then culled: the reference NULLs the slot (:2700-2708, :2843-2847)
    }
    return nmatches;
}

int orbref_search_by_projection_points(const orbref_frame_t* f, const uint8_t* blocked_in, const float* sf,
                                       int nq, const uint8_t* in_view, const float* px, const float* py, const float* pxr,
                                       const float* view_cos, const int32_t* level, const uint8_t* qdesc, const uint8_t* mp_obs,
                                       float th, float nnratio, int32_t* match) {
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> blocked(blocked_in, blocked_in + f->n);
    for (int i = 0; i < f->n; ++i) match[i] = -1;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nq; ++iMP) {
        if (!in_view[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;                        // RadiusByViewingCos :242-249
        if (bFactor) r *= th;
        features_in_area(f, px[iMP], py[iMP], r * sf[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (blocked[idx]) continue;
            if (f->uright && f->uright[idx] > 0) {
                const float er = std::fabs(pxr[iMP] - f->uright[idx]);
                if (er > r * sf[nPredictedLevel]) continue;
            }
            const int dist = orbref_hamming(qdesc + 32 * (size_t)iMP, f->desc + 32 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f->kps[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f->kps[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                match[bestIdx] = iMP;
                if (mp_obs[iMP]) blocked[bestIdx] = 1;
                nmatches++;
            }
        }
    }
    return nmatches;
}

int orbref_search_for_initialization(const orbref_frame_t* F1, const orbref_frame_t* F2, float* prev, int windowSize,
                                     float nnratio, int check_ori, int32_t* vnMatches12) {
    int nmatches = 0;
    for (int i = 0; i < F1->n; ++i) vnMatches12[i] = -1;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // :812
    std::vector<int> vMatchedDistance(F2->n, INT_MAX), vnMatches21(F2->n, -1), vIndices2;
    for (int i1 = 0; i1 < F1->n; ++i1) {
        const int level1 = F1->kps[i1].octave;
        if (level1 > 0) continue;
        features_in_area(F2, prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
        if (vIndices2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const int dist = orbref_hamming(F1->desc + 32 * (size_t)i1, F2->desc + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) rh.add(F1->kps[i1].angle, F2->kps[bestIdx2].angle, factor, i1);
            }
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < HISTO_LENGTH; ++b) {
            if (b == ind[0] || b == ind[1] || b == ind[2]) continue;
            for (int idx1 : rh.bins[b])
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < F1->n; ++i1)
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = F2->kps[vnMatches12[i1]].x; prev[2 * i1 + 1] = F2->kps[vnMatches12[i1]].y; }
    return nmatches;
}

static bool epipolar_constrain(const float* F12, const orbref_kp_t& kp1, const orbref_kp_t& kp2, float unc) {
    // Pinhole::epipolarConstrain_ (Pinhole.cpp:281-295) with F12 precomputed by the caller
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * unc;
}

int orbref_search_for_triangulation(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                    int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1v,
                                    int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                    int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2v,
                                    const float* F12, float epx, float epy, const float* sf2, const float* sigma2_2,
                                    int bOnlyStereo, int bCoarse, int check_ori, int32_t* vMatches12) {
    int nmatches = 0;
    std::vector<bool> vbMatched2(n2, false);                                  // never set in this overload (:1567)
    for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
    RotHist rh;
    const float factor = 1.0f / HISTO_LENGTH;                                 // :1441 (sic)
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = start1[a]; i1 < start1[a + 1]; ++i1) {
                const int idx1 = idx1v[i1];
                if (has_mp1[idx1]) continue;
                const bool bStereo1 = uright1 && uright1[idx1] >= 0;
                if (bOnlyStereo && !bStereo1) continue;
                const orbref_kp_t& kp1 = kps1[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = start2[b]; i2 < start2[b + 1]; ++i2) {
                    const int idx2 = idx2v[i2];
                    if (vbMatched2[idx2] || has_mp2[idx2]) continue;
                    const bool bStereo2 = uright2 && uright2[idx2] >= 0;
                    if (bOnlyStereo && !bStereo2) continue;
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orbref_kp_t& kp2 = kps2[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = epx - kp2.x, distey = epy - kp2.y;
                        if (distex * distex + distey * distey < 100 * sf2[kp2.octave]) continue;
                    }
                    if (epipolar_constrain(F12, kp1, kp2, sigma2_2[kp2.octave]) || bCoarse) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) rh.add(kp1.angle, kps2[bestIdx2].angle, factor, idx1);
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            a = (int)(std::lower_bound(nodes1, nodes1 + nn1, nodes2[b]) - nodes1);
        } else {
            b = (int)(std::lower_bound(nodes2, nodes2 + nn2, nodes1[a]) - nodes2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { vMatches12[i] = -1; nmatches--; }
        }
    }
    (void)n1;
    return nmatches;
}

int orbref_search_for_triangulation_gated(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                                          int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1v,
                                          int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2,
                                          int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2v,
                                          orbref_pair_gate_fn gate, void* user, int check_ori, int32_t* vMatches12) {
    // ORBmatcher.cc:1632-1821 (and :1388-1629 with mpCamera2): vbMatched2 is declared but never set in either
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
    RotHist rh;
    const float factor = 1.0f / HISTO_LENGTH;                                 // :1441, :1672 (sic)
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = start1[a]; i1 < start1[a + 1]; ++i1) {
                const int idx1 = idx1v[i1];
                if (has_mp1[idx1]) continue;                                  // :1683-1685
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = start2[b]; i2 < start2[b + 1]; ++i2) {
                    const int idx2 = idx2v[i2];
                    if (has_mp2[idx2]) continue;                              // :1706-1707
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;           // :1713-1715
                    if (gate(user, idx1, idx2)) { bestIdx2 = idx2; bestDist = dist; }   // :1729-1733 / :1552-1556
                }
                if (bestIdx2 >= 0) {
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) rh.add(kps1[idx1].angle, kps2[bestIdx2].angle, factor, idx1);
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            a = (int)(std::lower_bound(nodes1, nodes1 + nn1, nodes2[b]) - nodes1);
        } else {
            b = (int)(std::lower_bound(nodes2, nodes2 + nn2, nodes1[a]) - nodes2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { vMatches12[i] = -1; nmatches--; }
        }
    }
    (void)n2;
    return nmatches;
}

int orbref_search_by_bow(int nkf, const orbref_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                         int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                         int nf, const orbref_kp_t* kps_f, const uint8_t* desc_f,
                         int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                         float nnratio, int check_ori, int32_t* f_match) {
    for (int i = 0; i < nf; ++i) f_match[i] = -1;
    int nmatches = 0;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // :334
    int a = 0, b = 0;
    while (a < nnk && b < nnf) {
        if (nodes_k[a] == nodes_f[b]) {
            for (int iKF = start_k[a]; iKF < start_k[a + 1]; ++iKF) {
                const int realIdxKF = idx_k[iKF];
                if (!kf_good[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = start_f[b]; iF < start_f[b + 1]; ++iF) {
                    const int realIdxF = idx_f[iF];
                    if (f_match[realIdxF] >= 0) continue;                    // :385
                    const int dist = orbref_hamming(desc_kf + 32 * (size_t)realIdxKF, desc_f + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        f_match[bestIdxF] = realIdxKF;
                        if (check_ori) rh.add(kps_kf[realIdxKF].angle, kps_f[bestIdxF].angle, factor, bestIdxF);
                        nmatches++;
                    }
                }
            }
            ++a; ++b;
        } else if (nodes_k[a] < nodes_f[b]) {
            a = (int)(std::lower_bound(nodes_k, nodes_k + nnk, nodes_f[b]) - nodes_k);
        } else {
            b = (int)(std::lower_bound(nodes_f, nodes_f + nnf, nodes_k[a]) - nodes_f);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { f_match[i] = -1; nmatches--; }
        }
    }
    (void)nkf;
    return nmatches;
}


int orbref_search_by_projection_kf(const orbref_frame_t* cur, const uint8_t* blocked_in, const float* sf,
                                   int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                   const float* angle, const uint8_t* qdesc, float th, int ORBdist, int check_ori, int32_t* match) {
    int nmatches = 0;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // ORBmatcher.cc:2736
    std::vector<uint8_t> taken(blocked_in, blocked_in + cur->n);
    for (int i = 0; i < cur->n; ++i) match[i] = -1;
    std::vector<int> vIndices2;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * sf[nPredictedLevel];
        features_in_area(cur, u[i], v[i], radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (taken[i2]) continue;                                          // :2793 (any MapPoint, observations not consulted)
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, cur->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
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
        for (int b = 0; b < HISTO_LENGTH; ++b)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int idx : rh.bins[b]) { match[idx] = -2; nmatches--; }   // assigned, then culled: the reference NULLs the slot (:2700-2708, :2843-2847)
    }
    return nmatches;
}

int orbref_search_by_bow_kf(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* good1,
                            int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1v,
                            int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* good2,
                            int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2v,
                            float nnratio, int check_ori, int32_t* vpMatches12) {
    for (int i = 0; i < n1; ++i) vpMatches12[i] = -1;
    std::vector<bool> vbMatched2(n2, false);
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // :978
    int nmatches = 0, a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = start1[a]; i1 < start1[a + 1]; ++i1) {
                const int idx1 = idx1v[i1];
                if (!good1[idx1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = start2[b]; i2 < start2[b + 1]; ++i2) {
                    const int idx2 = idx2v[i2];
                    if (vbMatched2[idx2] || !good2[idx2]) continue;
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {                                     // strict (:1047)
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        vpMatches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (check_ori) rh.add(kps1[idx1].angle, kps2[bestIdx2].angle, factor, idx1);
                        nmatches++;
                    }
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            a = (int)(std::lower_bound(nodes1, nodes1 + nn1, nodes2[b]) - nodes1);
        } else {
            b = (int)(std::lower_bound(nodes2, nodes2 + nn2, nodes1[a]) - nodes2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { vpMatches12[i] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbref_search_for_triangulation_legacy(int n1, const orbref_kp_t* kps1, const uint8_t* desc1, const uint8_t* has_mp1, const float* uright1,
                                    int nn1, const int32_t* nodes1, const int32_t* start1, const int32_t* idx1v,
                                    int n2, const orbref_kp_t* kps2, const uint8_t* desc2, const uint8_t* has_mp2, const float* uright2,
                                    int nn2, const int32_t* nodes2, const int32_t* start2, const int32_t* idx2v,
                                    const float* F12, float epx, float epy, const float* sf2, const float* sigma2_2,
                                    int bOnlyStereo, int bCoarse, int check_ori, int32_t* vMatches12) {
    int nmatches = 0;
    std::vector<bool> vbMatched2(n2, false);
    for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;                               // :1166
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int i1 = start1[a]; i1 < start1[a + 1]; ++i1) {
                const int idx1 = idx1v[i1];
                if (has_mp1[idx1]) continue;
                const bool bStereo1 = uright1 && uright1[idx1] >= 0;
                if (bOnlyStereo && !bStereo1) continue;
                const orbref_kp_t& kp1 = kps1[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = start2[b]; i2 < start2[b + 1]; ++i2) {
                    const int idx2 = idx2v[i2];
                    if (vbMatched2[idx2] || has_mp2[idx2]) continue;
                    const bool bStereo2 = uright2 && uright2[idx2] >= 0;
                    if (bOnlyStereo && !bStereo2) continue;
                    const int dist = orbref_hamming(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orbref_kp_t& kp2 = kps2[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = epx - kp2.x, distey = epy - kp2.y;
                        if (distex * distex + distey * distey < 100 * sf2[kp2.octave]) continue;
                    }
                    if (epipolar_constrain(F12, kp1, kp2, sigma2_2[kp2.octave]) || bCoarse) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    vMatches12[idx1] = bestIdx2;
                    vbMatched2[bestIdx2] = true;                              // :1319
                    nmatches++;
                    if (check_ori) rh.add(kp1.angle, kps2[bestIdx2].angle, factor, idx1);
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) {
            a = (int)(std::lower_bound(nodes1, nodes1 + nn1, nodes2[b]) - nodes1);
        } else {
            b = (int)(std::lower_bound(nodes2, nodes2 + nn2, nodes1[a]) - nodes2);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { vbMatched2[vMatches12[i]] = false; vMatches12[i] = -1; nmatches--; }   // :1366
        }
    }
    return nmatches;
}


int orbref_search_by_projection_sim3(const orbref_frame_t* kf, const uint8_t* matched_in, const float* sf,
                                     int nq, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                     const uint8_t* qdesc, int th, float ratioHamming, int32_t* match) {
    int nmatches = 0;
    std::vector<uint8_t> matched(matched_in, matched_in + kf->n);
    for (int i = 0; i < kf->n; ++i) match[i] = -1;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nq; ++iMP) {
        if (!valid[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        const float radius = th * sf[nPredictedLevel];
        features_in_area(kf, u[iMP], v[iMP], radius, -1, -1, vIndices);       // KeyFrame::GetFeaturesInArea: no level filter
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            if (matched[idx]) continue;
            const int kpLevel = kf->kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orbref_hamming(qdesc + 32 * (size_t)iMP, kf->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW * ratioHamming) { match[bestIdx] = iMP; matched[bestIdx] = 1; nmatches++; }
    }
    return nmatches;
}

int orbref_fuse(const orbref_frame_t* kf, const float* sf, const float* inv_sigma2,
                int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const int32_t* level,
                const uint8_t* qdesc, float th, int chi2_gate, int32_t* best_idx) {
    int nFused = 0;
    std::vector<int> vIndices;
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1;
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * sf[nPredictedLevel];
        features_in_area(kf, u[i], v[i], radius, -1, -1, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = chi2_gate ? 256 : INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const orbref_kp_t& kp = kf->kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2_gate) {
                if (kf->uright && kf->uright[idx] >= 0) {
                    const float ex = u[i] - kp.x, ey = v[i] - kp.y, er = ur[i] - kf->uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u[i] - kp.x, ey = v[i] - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, kf->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    return nFused;
}


static void sim3_one_way(const orbref_frame_t* src, const orbref_frame_t* dst, const float* sf_dst, const uint8_t* valid,
                         const float* u, const float* v, const int32_t* level, const uint8_t* qdesc, float th, std::vector<int>& vnMatch) {
    vnMatch.assign(src->n, -1);
    std::vector<int> vIndices;
    for (int i = 0; i < src->n; ++i) {
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * sf_dst[nPredictedLevel];
        features_in_area(dst, u[i], v[i], radius, -1, -1, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const int oct = dst->kps[idx].octave;
            if (oct < nPredictedLevel - 1 || oct > nPredictedLevel) continue;
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, dst->desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
    }
}

int orbref_search_by_sim3(const orbref_frame_t* kf1, const orbref_frame_t* kf2, const float* sf1, const float* sf2,
                          const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1, const uint8_t* qdesc1,
                          const uint8_t* valid2, const float* u2, const float* v2, const int32_t* level2, const uint8_t* qdesc2,
                          float th, int32_t* matches12) {
    std::vector<int> vnMatch1, vnMatch2;
    sim3_one_way(kf1, kf2, sf2, valid1, u1, v1, level1, qdesc1, th, vnMatch1);   // :2247-2318
    sim3_one_way(kf2, kf1, sf1, valid2, u2, v2, level2, qdesc2, th, vnMatch2);   // :2321-2392
    int nFound = 0;
    for (int i1 = 0; i1 < kf1->n; ++i1) {
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    return nFound;
}


int orbref_search_by_projection_frame_fisheye(const orbref_frame_t* cur_l, const orbref_frame_t* cur_r,
                                              const uint8_t* blocked_l_in, const uint8_t* blocked_r_in, const float* sf,
                                              int nq, const uint8_t* valid, const float* u, const float* v, const float* ur, const float* vr,
                                              const int32_t* octave, const float* angle, const uint8_t* qdesc, const uint8_t* mp_obs,
                                              float th, int bForward, int bBackward, int check_ori, int32_t* match_l, int32_t* match_r) {
    int nmatches = 0;
    const int Nleft = cur_l->n;
    RotHist rh;                                                               // holds global indices (right ones offset by Nleft)
    const float factor = HISTO_LENGTH / 360.0f;
    std::vector<uint8_t> bl(blocked_l_in, blocked_l_in + cur_l->n), br(blocked_r_in, blocked_r_in + cur_r->n);
    for (int i = 0; i < cur_l->n; ++i) match_l[i] = -1;
    for (int i = 0; i < cur_r->n; ++i) match_r[i] = -1;
    std::vector<int> vIndices2;
    for (int i = 0; i < nq; ++i) {
        if (!valid[i]) continue;
        const int nLastOctave = octave[i];
        const float radius = th * sf[nLastOctave];
        auto area = [&](const orbref_frame_t* f, float x, float y) {
            if (bForward) features_in_area(f, x, y, radius, nLastOctave, -1, vIndices2);
            else if (bBackward) features_in_area(f, x, y, radius, 0, nLastOctave, vIndices2);
            else features_in_area(f, x, y, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        };
        area(cur_l, u[i], v[i]);
        if (vIndices2.empty()) continue;                                      // :2551 -- also skips the right-camera block
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (bl[i2]) continue;
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, cur_l->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            match_l[bestIdx2] = i;
            if (mp_obs[i]) bl[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], cur_l->kps[bestIdx2].angle, factor, bestIdx2);
        }
        area(cur_r, ur[i], vr[i]);                                            // :2615-2627
        bestDist = 256; bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (br[i2]) continue;
            const int dist = orbref_hamming(qdesc + 32 * (size_t)i, cur_r->desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            match_r[bestIdx2] = i;
            if (mp_obs[i]) br[bestIdx2] = 1;
            nmatches++;
            if (check_ori) rh.add(angle[i], cur_r->kps[bestIdx2].angle, factor, bestIdx2 + Nleft);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int b = 0; b < HISTO_LENGTH; ++b)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int idx : rh.bins[b]) { if (idx < Nleft) match_l[idx] = -2; else match_r[idx - Nleft] = -2; nmatches--; }
    }
    return nmatches;
}

int orbref_search_by_projection_points_fisheye(const orbref_frame_t* f_l, const orbref_frame_t* f_r,
                                               const uint8_t* blocked_l_in, const uint8_t* blocked_r_in,
                                               const int32_t* l2r, const int32_t* r2l, const float* sf,
                                               int nq, const uint8_t* in_view, const float* px, const float* py, const float* view_cos, const int32_t* level,
                                               const uint8_t* in_view_r, const float* pxr, const float* pyr, const float* view_cos_r, const int32_t* level_r,
                                               const uint8_t* qdesc, const uint8_t* mp_obs, float th, float nnratio, int32_t* match_l, int32_t* match_r) {
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> bl(blocked_l_in, blocked_l_in + f_l->n), br(blocked_r_in, blocked_r_in + f_r->n);
    for (int i = 0; i < f_l->n; ++i) match_l[i] = -1;
    for (int i = 0; i < f_r->n; ++i) match_r[i] = -1;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nq; ++iMP) {
        if (!in_view[iMP] && !in_view_r[iMP]) continue;
        if (in_view[iMP]) {
            const int nPredictedLevel = level[iMP];
            float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            features_in_area(f_l, px[iMP], py[iMP], r * sf[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
            if (!vIndices.empty()) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int idx : vIndices) {
                    if (bl[idx]) continue;
                    const int dist = orbref_hamming(qdesc + 32 * (size_t)iMP, f_l->desc + 32 * (size_t)idx);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f_l->kps[idx].octave; bestIdx = idx; }
                    else if (dist < bestDist2) { bestLevel2 = f_l->kps[idx].octave; bestDist2 = dist; }
                }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;          // :148-149 skips the right block too
                    if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                        match_l[bestIdx] = iMP;
                        if (mp_obs[iMP]) bl[bestIdx] = 1;
                        if (l2r[bestIdx] != -1) {                                                 // :152-157
                            match_r[l2r[bestIdx]] = iMP;
                            if (mp_obs[iMP]) br[l2r[bestIdx]] = 1;
                            nmatches++;
                        }
                        nmatches++;
                    }
                }
            }
        }
        if (in_view_r[iMP]) {                                                                     // :170-236
            const int nPredictedLevel = level_r[iMP];
            if (nPredictedLevel != -1) {
                const float r = view_cos_r[iMP] > 0.998 ? 2.5f : 4.0f;                            // no th factor here (:174)
                features_in_area(f_r, pxr[iMP], pyr[iMP], r * sf[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
                if (vIndices.empty()) continue;
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int idx : vIndices) {
                    if (br[idx]) continue;
                    const int dist = orbref_hamming(qdesc + 32 * (size_t)iMP, f_r->desc + 32 * (size_t)idx);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f_r->kps[idx].octave; bestIdx = idx; }
                    else if (dist < bestDist2) { bestLevel2 = f_r->kps[idx].octave; bestDist2 = dist; }
                }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (r2l[bestIdx] != -1) {                                                     // :222-226
                        match_l[r2l[bestIdx]] = iMP;
                        if (mp_obs[iMP]) bl[r2l[bestIdx]] = 1;
                        nmatches++;
                    }
                    match_r[bestIdx] = iMP;
                    if (mp_obs[iMP]) br[bestIdx] = 1;
                    nmatches++;
                }
            }
        }
    }
    return nmatches;
}

int orbref_search_by_bow_fisheye(int nkf, const orbref_kp_t* kps_kf, const uint8_t* desc_kf, const uint8_t* kf_good,
                         int nnk, const int32_t* nodes_k, const int32_t* start_k, const int32_t* idx_k,
                         int nf, int nleft, const orbref_kp_t* kps_f, const uint8_t* desc_f,
                         int nnf, const int32_t* nodes_f, const int32_t* start_f, const int32_t* idx_f,
                         float nnratio, int check_ori, int32_t* f_match) {
    for (int i = 0; i < nf; ++i) f_match[i] = -1;
    int nmatches = 0;
    RotHist rh;
    const float factor = HISTO_LENGTH / 360.0f;
    int a = 0, b = 0;
    while (a < nnk && b < nnf) {
        if (nodes_k[a] == nodes_f[b]) {
            for (int iKF = start_k[a]; iKF < start_k[a + 1]; ++iKF) {
                const int realIdxKF = idx_k[iKF];
                if (!kf_good[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
                for (int iF = start_f[b]; iF < start_f[b + 1]; ++iF) {
                    const int realIdxF = idx_f[iF];
                    if (f_match[realIdxF] >= 0) continue;
                    const int dist = orbref_hamming(desc_kf + 32 * (size_t)realIdxKF, desc_f + 32 * (size_t)realIdxF);
                    if (realIdxF < nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (realIdxF < nleft && dist < bestDist2) bestDist2 = dist;
                    if (realIdxF >= nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                    else if (realIdxF >= nleft && dist < bestDist2R) bestDist2R = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        f_match[bestIdxF] = realIdxKF;
                        if (check_ori) rh.add(kps_kf[realIdxKF].angle, kps_f[bestIdxF].angle, factor, bestIdxF);
                        nmatches++;
                    }
                    if (bestDist1R <= TH_LOW) {                                                   // nested, ratio test disabled (:471-473)
                        f_match[bestIdxFR] = realIdxKF;
                        if (check_ori) rh.add(kps_kf[realIdxKF].angle, kps_f[bestIdxFR].angle, factor, bestIdxFR);
                        nmatches++;
                    }
                }
            }
            ++a; ++b;
        } else if (nodes_k[a] < nodes_f[b]) {
            a = (int)(std::lower_bound(nodes_k, nodes_k + nnk, nodes_f[b]) - nodes_k);
        } else {
            b = (int)(std::lower_bound(nodes_f, nodes_f + nnf, nodes_k[a]) - nodes_f);
        }
    }
    if (check_ori) {
        int ind[3];
        rh.maxima(ind);
        for (int bb = 0; bb < HISTO_LENGTH; ++bb) {
            if (bb == ind[0] || bb == ind[1] || bb == ind[2]) continue;
            for (int i : rh.bins[bb]) { f_match[i] = -1; nmatches--; }
        }
    }
    (void)nkf;
    return nmatches;
}

int orbref_stereo_matches(const orbref_t* left, const orbref_t* right,
                          int N, const orbref_kp_t* kl, const uint8_t* dl, int Nr, const orbref_kp_t* kr, const uint8_t* dr,
                          float mb, float mbf, float* mvuRight, float* mvDepth) {
    const int L_ = orbref_nlevels(left);
    std::vector<float> sf(L_), isf(L_);
    orbref_tables(left, sf.data(), isf.data(), nullptr, nullptr, nullptr, nullptr);
    for (int i = 0; i < N; ++i) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    int w0, nRows;
    orbref_level_data(left, 0, &w0, &nRows);
    std::vector<std::vector<int>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; ++iR) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * sf[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r), minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);          // reference indexes unchecked (rows exist for real keypoints)
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; ++iL) {
        const orbref_kp_t& kpL = kl[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<int>& vCandidates = vRowIndices[(int)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        for (int iR : vCandidates) {
            const orbref_kp_t& kpR = kr[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orbref_hamming(dl + 32 * (size_t)iL, dr + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = isf[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor), scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, Lh = 5;
            int lw, lh, rw, rh_;
            const uint8_t* IL = orbref_level_data(left, kpL.octave, &lw, &lh);
            const uint8_t* IR = orbref_level_data(right, kpL.octave, &rw, &rh_);
            int bestDistS = INT_MAX, bestincR = 0;
            float vDists[11];
            const float iniu = scaleduR0 + Lh - w, endu = scaleduR0 + Lh + w + 1;
            if (iniu < 0 || endu >= rw) continue;
            for (int incR = -Lh; incR <= Lh; ++incR) {
                int sad = 0;
                for (int dy = -w; dy <= w; ++dy)
                    for (int dx = -w; dx <= w; ++dx)
                        sad += std::abs((int)IL[((int)scaledvL + dy) * lw + (int)scaleduL + dx] -
                                        (int)IR[((int)scaledvL + dy) * rw + (int)scaleduR0 + incR + dx]);
                const float dist = (float)(double)sad;                         // cv::norm(NORM_L1) -> double -> float
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[Lh + incR] = dist;
            }
            if (bestincR == -Lh || bestincR == Lh) continue;
            const float dist1 = vDists[Lh + bestincR - 1], dist2 = vDists[Lh + bestincR], dist3 = vDists[Lh + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = sf[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;                                            // reference indexes an empty vector here (UB)
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1; mvDepth[vDistIdx[i].second] = -1; --kept;
    }
    return kept;
}

// ---------------------------------------------------------------------------------------------------------
// cv::undistortPoints restated (OpenCV imgproc undistort.dispatch.cpp, cvUndistortPointsInternal): double arithmetic,
// 5 iterations, R = I so the final rotation/projection is  x' = P00*x + P01*y + P02  with P01 = 0, w = 1/(0*x + 0*y + 1).
// ---------------------------------------------------------------------------------------------------------
static void undistort_point(double u, double v, const double k[14], double fx, double fy, double cx, double cy,
                            double nfx, double nfy, double ncx, double ncy, float* ox, float* oy) {
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = nfx * x + 0.0 * y + ncx, yy = 0.0 * x + nfy * y + ncy, ww = 1.0 / (0.0 * x + 0.0 * y + 1.0);
    *ox = (float)(xx * ww); *oy = (float)(yy * ww);
}

static void load_dist(const float* dist, int ndist, double k[14]) {
    for (int i = 0; i < 14; ++i) k[i] = 0.0;
    for (int i = 0; i < ndist && i < 14; ++i) k[i] = (double)dist[i];
}

int orbref_undistort_keypoints(const orbref_kp_t* kps, int n, const float* kk, const float* dist, int ndist, const float* newk,
                               orbref_kp_t* out) {
    if (n < 0 || !kk || !newk || (ndist > 0 && !dist)) return -2;
    for (int i = 0; i < n; ++i) out[i] = kps[i];
    if (ndist < 1 || dist[0] == 0.0f) return n;                      // Frame.cc:928-932
    double k[14]; load_dist(dist, ndist, k);
    for (int i = 0; i < n; ++i)
        undistort_point((double)kps[i].x, (double)kps[i].y, k, kk[0], kk[1], kk[2], kk[3], newk[0], newk[1], newk[2], newk[3],
                        &out[i].x, &out[i].y);
    return n;
}

int orbref_image_bounds(int cols, int rows, const float* kk, const float* dist, int ndist, const float* newk, float* bounds) {
    if (!kk || !newk || !bounds) return -2;
    if (ndist < 1 || dist[0] == 0.0f) { bounds[0] = 0.f; bounds[1] = (float)cols; bounds[2] = 0.f; bounds[3] = (float)rows; return 0; }
    double k[14]; load_dist(dist, ndist, k);
    const float cxs[4] = {0.f, (float)cols, 0.f, (float)cols}, cys[4] = {0.f, 0.f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; ++i)
        undistort_point((double)cxs[i], (double)cys[i], k, kk[0], kk[1], kk[2], kk[3], newk[0], newk[1], newk[2], newk[3], &ux[i], &uy[i]);
    bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]);
    bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    return 0;
}

int orbref_is_in_frustum(int n, const float* pw, const float* normal, const float* min_dist, const float* max_dist,
                         const float* rcw, const float* tcw, const float* ow, const float* kk, const float* bounds,
                         float bf, float viewing_cos_limit, float log_scale_factor, int n_scale_levels,
                         uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, float* depth, int32_t* level, float* view_cos) {
    if (n < 0) return -2;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        in_view[i] = 0; proj_x[i] = -1.f; proj_y[i] = -1.f;
        const float* P = pw + 3 * i;
        float Pc[3];
        for (int r = 0; r < 3; ++r) {                               // Matx33f * Matx31f: float accumulation from 0, k ascending
            float s = 0.f;
            for (int c = 0; c < 3; ++c) s += rcw[3 * r + c] * P[c];
            Pc[r] = s + tcw[r];
        }
        double n2 = 0.0;                                            // cv::norm(Matx): double accumulation, sqrt in double
        for (int c = 0; c < 3; ++c) n2 += (double)Pc[c] * (double)Pc[c];
        const float Pc_dist = (float)std::sqrt(n2);
        const float PcZ = Pc[2];
        const float invz = 1.0f / PcZ;
        if (PcZ < 0.0f) continue;
        const float u = kk[0] * Pc[0] / Pc[2] + kk[2], v = kk[1] * Pc[1] / Pc[2] + kk[3];   // Pinhole.cpp:57-60
        if (u < bounds[0] || u > bounds[1]) continue;
        if (v < bounds[2] || v > bounds[3]) continue;
        proj_x[i] = u; proj_y[i] = v;
        const float maxD = 1.2f * max_dist[i], minD = 0.8f * min_dist[i];
        float PO[3];
        for (int c = 0; c < 3; ++c) PO[c] = P[c] - ow[c];
        double d2 = 0.0;
        for (int c = 0; c < 3; ++c) d2 += (double)PO[c] * (double)PO[c];
        const float dist = (float)std::sqrt(d2);
        if (dist < minD || dist > maxD) continue;
        float dot = 0.f;
        for (int c = 0; c < 3; ++c) dot += PO[c] * normal[3 * i + c];
        const float vc = dot / dist;
        if (vc < viewing_cos_limit) continue;
        const float ratio = max_dist[i] / dist;                    // PredictScale (MapPoint.cc:725-740): float log, float divide
        int ns = (int)std::ceil(std::log(ratio) / log_scale_factor);
        if (ns < 0) ns = 0; else if (ns >= n_scale_levels) ns = n_scale_levels - 1;
        in_view[i] = 1; proj_xr[i] = u - bf * invz; depth[i] = Pc_dist; level[i] = ns; view_cos[i] = vc;
        ++cnt;
    }
    return cnt;
}

int orbref_gray_from_color(const uint8_t* src, int w, int h, int src_stride, int channels, int blue_first, int coef_bits,
                           uint8_t* dst, int dst_stride) {
    if (!src || !dst || w < 1 || h < 1 || (channels != 3 && channels != 4) || (coef_bits != 14 && coef_bits != 15)) return -2;
    const int ry = coef_bits == 14 ? 4899 : 9798, gy = coef_bits == 14 ? 9617 : 19235, by = coef_bits == 14 ? 1868 : 3735;
    const int c0 = blue_first ? by : ry, c2 = blue_first ? ry : by, half = 1 << (coef_bits - 1);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * src_stride;
        uint8_t* d = dst + (size_t)y * dst_stride;
        for (int x = 0; x < w; ++x, s += channels) d[x] = (uint8_t)((s[0] * c0 + s[1] * gy + s[2] * c2 + half) >> coef_bits);
    }
    return 0;
}

int orbref_remap_linear(const uint8_t* src, int sw, int sh, int src_stride, const float* mapx, const float* mapy,
                        int dw, int dh, uint8_t* dst, int dst_stride) {
    if (!src || !mapx || !mapy || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return -2;
    auto at = [&](int x, int y) -> int { return (x >= 0 && x < sw && y >= 0 && y < sh) ? src[(size_t)y * src_stride + x] : 0; };
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const int sx = (int)lrintf(mapx[(size_t)y * dw + x] * 32.f), sy = (int)lrintf(mapy[(size_t)y * dw + x] * 32.f);   // cvRound(map * INTER_TAB_SIZE)
            const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            const int v = at(ix, iy) * w00 + at(ix + 1, iy) * w01 + at(ix, iy + 1) * w10 + at(ix + 1, iy + 1) * w11;
            dst[(size_t)y * dst_stride + x] = (uint8_t)((v + (1 << 14)) >> 15);
        }
    return 0;
}

static inline int reflect101_idx(int p, int n) {              // cv::borderInterpolate(BORDER_REFLECT_101)
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

int orbref_clahe(const uint8_t* src, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
                 uint8_t* dst, int dst_stride) {
    if (!src || !dst || w < 1 || h < 1 || tiles_x < 1 || tiles_y < 1) return -2;
    int ew = w, eh = h;                                         // extended size (clahe.cpp: copyMakeBorder when not divisible)
    if (w % tiles_x != 0 || h % tiles_y != 0) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / 256); clip = std::max(clip, 1); }
    const float lutScale = (float)255 / area;
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            int hist[256] = {0};
            for (int y = ty * th; y < (ty + 1) * th; ++y)
                for (int x = tx * tw; x < (tx + 1) * tw; ++x)
                    hist[src[(size_t)reflect101_idx(y, h) * src_stride + reflect101_idx(x, w)]]++;
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; ++i) if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
                const int redistBatch = clipped / 256;
                int residual = clipped - redistBatch * 256;
                for (int i = 0; i < 256; ++i) hist[i] += redistBatch;
                if (residual != 0) {
                    const int step = std::max(256 / residual, 1);
                    for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            int sum = 0;
            uint8_t* tl = &lut[((size_t)ty * tiles_x + tx) * 256];
            for (int i = 0; i < 256; ++i) {
                sum += hist[i];
                const int v = (int)lrintf((float)sum * lutScale);
                tl[i] = (uint8_t)std::min(255, std::max(0, v));
            }
        }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; ++y) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = (int)std::floor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles_y - 1);
        for (int x = 0; x < w; ++x) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = (int)std::floor(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles_x - 1);
            const int v = src[(size_t)y * src_stride + x];
            const float l11 = lut[((size_t)ty1 * tiles_x + tx1) * 256 + v], l12 = lut[((size_t)ty1 * tiles_x + tx2) * 256 + v];
            const float l21 = lut[((size_t)ty2 * tiles_x + tx1) * 256 + v], l22 = lut[((size_t)ty2 * tiles_x + tx2) * 256 + v];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            const int r = (int)lrintf(res);
            dst[(size_t)y * dst_stride + x] = (uint8_t)std::min(255, std::max(0, r));
        }
    }
    return 0;
}

}  // extern "C"
