// orbref_bow.cpp -- CPU ORACLE for the DBoW2 vocabulary transform (TEST INFRASTRUCTURE ONLY, see orbref.h).
// Restates Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h (loadFromTextFile :1338-1440, transform :1125-1262),
// FORB::distance (FORB.cpp:81-101) and BowVector/FeatureVector (BowVector.cpp, FeatureVector.cpp) with std::map,
// exactly as the vendored code does.  PARITY UNPINNED (ORBvoc.txt is a missing blob; tests use synthetic trees).
#include "orbref.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct VNode {
    int id = 0, parent = 0, word_id = 0;
    double weight = 0;
    std::vector<int> children;
    uint8_t d[32];
    bool isLeaf() const { return children.empty(); }
};
struct orbref_vocab { int k = 0, L = 0; std::vector<VNode> nodes; std::vector<int> words; };

extern "C" {

orbref_vocab_t* orbref_vocab_load_text(const char* path) {
    std::ifstream f(path);
    if (!f.good()) return nullptr;
    orbref_vocab* v = new orbref_vocab;
    std::string s;
    std::getline(f, s);
    std::stringstream ss; ss << s;
    int n1 = 0, n2 = 0;
    ss >> v->k; ss >> v->L; ss >> n1; ss >> n2;
    if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { delete v; return nullptr; }
    v->nodes.resize(1);
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        if (snode.empty()) continue;                          // (the reference parses the trailing empty line into a junk node)
        std::stringstream sn; sn << snode;
        const int nid = (int)v->nodes.size();
        v->nodes.resize(nid + 1);
        VNode& nd = v->nodes[nid];
        nd.id = nid;
        int pid, leaf;
        sn >> pid; nd.parent = pid;
        v->nodes[pid].children.push_back(nid);
        sn >> leaf;
        for (int i = 0; i < 32; ++i) { int b; sn >> b; v->nodes[nid].d[i] = (uint8_t)b; }
        sn >> v->nodes[nid].weight;
        if (leaf > 0) { v->nodes[nid].word_id = (int)v->words.size(); v->words.push_back(nid); }
    }
    return v;
}
// The same tree from arrays (node 0 = root; parent[i] < i): what loadFromTextFile builds line by line (:1385-1420) -- children in
// node-id order, word ids in leaf order -- without a 150 MB text file for an ORBvoc-sized tree (k = 10, L = 6: 1 111 111 nodes).
orbref_vocab_t* orbref_vocab_create(int k, int L, int nnodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight) {
    if (nnodes < 2) return nullptr;
    orbref_vocab* v = new orbref_vocab;
    v->k = k; v->L = L;
    v->nodes.resize(nnodes);
    for (int nid = 1; nid < nnodes; ++nid) {
        VNode& nd = v->nodes[nid];
        nd.id = nid; nd.parent = parent[nid];
        if (nd.parent < 0 || nd.parent >= nid) { delete v; return nullptr; }
        v->nodes[nd.parent].children.push_back(nid);
        memcpy(nd.d, desc + 32 * (size_t)nid, 32);
        nd.weight = weight[nid];
        if (is_leaf[nid]) { nd.word_id = (int)v->words.size(); v->words.push_back(nid); }
    }
    return v;
}
void orbref_vocab_destroy(orbref_vocab_t* v) { delete v; }
int orbref_vocab_info(const orbref_vocab_t* v, int* k, int* L, int* nnodes, int* nwords) {
    *k = v->k; *L = v->L; *nnodes = (int)v->nodes.size(); *nwords = (int)v->words.size(); return 0;
}

int orbref_bow_transform(const orbref_vocab_t* v, const uint8_t* desc, int n, int levelsup,
                         int32_t* word_id, int32_t* node_id, double* weight) {
    for (int i = 0; i < n; ++i) {
        const uint8_t* feature = desc + 32 * (size_t)i;
        const int nid_level = v->L - levelsup;
        int nid = 0;
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const std::vector<int>& nodes = v->nodes[final_id].children;
            final_id = nodes[0];
            double best_d = orbref_hamming(feature, v->nodes[final_id].d);
            for (size_t c = 1; c < nodes.size(); ++c) {
                const int id = nodes[c];
                const double d = orbref_hamming(feature, v->nodes[id].d);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!v->nodes[final_id].isLeaf());
        word_id[i] = v->nodes[final_id].word_id;
        weight[i] = v->nodes[final_id].weight;
        node_id[i] = nid;
    }
    return 0;
}

int orbref_bow_vectors(int n, const int32_t* word_id, const int32_t* node_id, const double* weight,
                       int32_t* bow_ids, double* bow_vals, int* nbow,
                       int32_t* fv_nodes, int32_t* fv_start, int32_t* fv_idx, int* nfv) {
    std::map<int, double> bow;                                 // BowVector::addWeight (BowVector.cpp:32-44)
    std::map<int, std::vector<unsigned>> fv;                   // FeatureVector::addFeature (FeatureVector.cpp:34-46)
    for (int i = 0; i < n; ++i) {
        if (weight[i] > 0) {
            auto it = bow.lower_bound(word_id[i]);
            if (it != bow.end() && !(bow.key_comp()(word_id[i], it->first))) it->second += weight[i];
            else bow.insert(it, std::make_pair(word_id[i], weight[i]));
            fv[node_id[i]].push_back((unsigned)i);
        }
    }
    double norm = 0.0;                                         // BowVector::normalize(L1) (BowVector.cpp:58-80)
    for (auto& kv : bow) norm += std::fabs(kv.second);
    if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
    int b = 0;
    for (auto& kv : bow) { bow_ids[b] = kv.first; bow_vals[b] = kv.second; ++b; }
    *nbow = b;
    int c = 0, o = 0;
    for (auto& kv : fv) { fv_nodes[c] = kv.first; fv_start[c] = o; for (unsigned k : kv.second) fv_idx[o++] = (int)k; ++c; }
    fv_start[c] = o;
    *nfv = c;
    return 0;
}

}  // extern "C"
