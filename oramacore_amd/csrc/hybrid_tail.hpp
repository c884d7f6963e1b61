// hybrid_tail.hpp — the device form of the one-call hybrid search's tail (hybrid_tail.hip).
//   src/collection_manager/sides/read/index/embedding_field.rs:264-276, token_score.rs:393-422, search.rs:482
#pragma once

#include "common.hpp"

namespace orama {

constexpr uint32_t kHybridTailMaxVec = 512;  // vector hits (`limit`) the device form takes; beyond: the host form

struct HybridTailArgs {
    // the vector leg's answer, still on the device: [ids | distances | n] of the scan's top-`limit` rows
    const uint64_t* v_ids = nullptr;
    const float* v_dist = nullptr;
    const uint32_t* v_n = nullptr;
    uint32_t limit = 0;
    float min_similarity = 0.0f;
    int rescale_e5 = 0;
    // DocumentId -> local document index of the postings store
    const uint64_t* docs = nullptr;  // sorted ids (non-dense stores)
    uint64_t n_docs = 0, dense_base = 0;
    int dense = 0;
    // the vector map (device scratch, `limit` entries each): documents in order of their first passing hit, summed scores,
    // local indices; full-text score / presence of every vector hit (range_score_docs_kernel)
    uint64_t* vdoc = nullptr;
    float* vsc = nullptr;
    uint32_t* vlocal = nullptr;
    float* vft = nullptr;
    uint32_t* vpresent = nullptr;
    // state[0] = entries of the vector map, state[1] = a hit is not a document of the index, state[2] = merged entries
    uint32_t* state = nullptr;
    // the full-text leg's raw candidates (the batch's top-k outputs of query 0) and result words
    const uint64_t* cand_id = nullptr;
    const float* cand_score = nullptr;
    const uint32_t* cand_n = nullptr;
    uint32_t k_asked = 0, top_k = 0;
    const uint32_t *res_count = nullptr, *res_max_key = nullptr, *res_min_inv = nullptr, *res_overflow = nullptr;
    // merged entries for K4 (k_asked + limit of each) and the words the host reads back
    float* e_score = nullptr;
    uint64_t* e_doc = nullptr;
    uint32_t* out_flag = nullptr;            // 0 = answered; 1 = the candidates cannot prove the answer, 2 = foreign hit, 4 = overflow
    unsigned long long* out_count = nullptr;
};

int launch_hybrid_vec_epilogue(const HybridTailArgs& a, hipStream_t stream);
int launch_hybrid_merge(const HybridTailArgs& a, hipStream_t stream);

}  // namespace orama
