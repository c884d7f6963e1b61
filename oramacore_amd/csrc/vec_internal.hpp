// vec_internal.hpp — what the fused hybrid entry point needs from the vector store (library-internal).
#pragma once

#include "common.hpp"

struct orama_vec;

namespace orama {

// Shared (read) lock on a vector store for the lifetime of the object.
class VecSharedLock {
   public:
    explicit VecSharedLock(orama_vec* v);
    ~VecSharedLock();
    VecSharedLock(const VecSharedLock&) = delete;
    VecSharedLock& operator=(const VecSharedLock&) = delete;

   private:
    orama_vec* v_;
};

orama_ctx* vec_ctx(orama_vec* v);
uint32_t vec_dim(orama_vec* v);
uint64_t vec_rows(orama_vec* v);
bool vec_rows_are_f32(orama_vec* v);  // plain fp32 storage (no fp16 rows: the selections behind its scans only write their outputs)
// Enqueue scan + top-k for q queries resident at d_queries on stream s (caller holds a VecSharedLock).
int vec_search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s);

// Two-stage exact search of a store with an fp16 shadow (vec_store.hip): usable for these host queries?
bool vec_two_stage_usable(orama_vec* v, const float* queries, uint32_t q, uint32_t k);
// ... and in its device form (the query test runs on the device)?
bool vec_two_stage_device_usable(orama_vec* v, uint32_t q, uint32_t k);
// Runs it on sc's stream with sc2 as the shadow stage's scratch; BLOCKS until the answers are in the device outputs.
// Queries whose candidate list could not be proven complete are re-answered by the plain scan (also blocking).
int vec_two_stage_search(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k,
                         const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n);


// The same in two halves, for a caller with other work to enqueue beside the scan (the one-call hybrid search): begin()
// only enqueues (stage 1, the completeness proof, the re-rank and the selection) and keeps the shadow store's read lock;
// finish() blocks, re-answers the unproven queries with the plain scan and releases the lock.
class VecTwoStage {
   public:
    VecTwoStage() = default;
    ~VecTwoStage();
    VecTwoStage(const VecTwoStage&) = delete;
    VecTwoStage& operator=(const VecTwoStage&) = delete;
    int begin(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k, const uint64_t* d_allow,
              uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n);
    // The plan's DEVICE form (vec_two_stage_device_usable must hold): the unproven queries are re-answered on the device, behind
    // everything else on the stream — nothing for the host to decide; finish() then only drains the stream and unlocks.
    int begin_device(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k,
                     const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n);
    int finish(bool* reran = nullptr);  // *reran: an unproven query was answered again by the plain scan (outputs rewritten after the caller's read-back was enqueued)

   private:
    void unlock();
    orama_vec* v_ = nullptr;
    void* locked_ = nullptr;  // the shadow store's std::shared_mutex while its read lock is held
    Scratch *sc_ = nullptr, *sc2_ = nullptr;
    const float* d_queries_ = nullptr;
    uint32_t q_ = 0, k_ = 0;
    const uint64_t* d_allow_ = nullptr;
    uint64_t allow_bits_ = 0;
    uint64_t* d_out_ids_ = nullptr;
    float* d_out_dist_ = nullptr;
    uint32_t* d_out_n_ = nullptr;
    bool flags_pending_ = false, device_form_ = false;
};

}  // namespace orama
