// orbref_match.cpp -- CPU ORACLE for the ORBmatcher primitives (TEST INFRASTRUCTURE ONLY, see orbref.h).
#include "orbref.h"
#include <climits>
#include <cstring>

extern "C" {

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2911-2931): SWAR popcount over 8 x 32-bit words
int orbref_hamming(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:2863-2905) on bin sizes
void orbref_three_maxima(const int* s_, int L, int* ind3) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3_ = -1;
    for (int i = 0; i < L; ++i) {
        const int s = s_[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3_ = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3_ = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3_ = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3_ = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3_ = -1; }
    ind3[0] = ind1; ind3[1] = ind2; ind3[2] = ind3_;
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by Frame::ComputeStereoFishEyeMatches
// (src/Frame.cc:1440-1480).  Tie order: ascending train index (normative).
void orbref_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2, int32_t* dist2) {
    for (int i = 0; i < nq; ++i) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; ++j) {
            int d = orbref_hamming(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1;
        dist2[2 * i] = i0 < 0 ? -1 : b0; dist2[2 * i + 1] = i1 < 0 ? -1 : b1;
    }
}

}  // extern "C"
