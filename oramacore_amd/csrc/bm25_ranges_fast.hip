// bm25_ranges_fast.hip — K3r's scoring launch for the plain top-k batch with compact key lists, round-6 form ("K3r-f").
//
// Same contract as range_score_compact_kernel (bm25_ranges.hip: one workgroup per (query, range), scores bit-identical to
// BM25Scorer::add / get_scores — bm25.rs:369-428 — `count` exact, survivors appended behind the query's cursor), different
// bookkeeping: a MaxScore cut over the query's LISTS, with the frequent terms' documents read as bitmap words.
//
//   * BACKGROUND LISTS.  A document's score is at most the sum over its lists of c_l = idf (k+1) (the limit of idf (k+1) S / (k + S)).
//     Once the query's first ranges have published a floor (a score at least topk documents reach), the lists of smallest c_l whose
//     bounds SUM to less than the floor cannot lift a document over it by themselves: a document that occurs in those lists only
//     is counted, never scored.  The store keeps, for its longest lists (>= n_docs / 128 postings, every normalised tf a tame
//     number), a bitmap of its documents + the popcount prefix of the bitmap's words (orama_post::d_acc): a background list's
//     part of a range is `width / 32` coalesced words OR-ed into an LDS bitmap — no gather, no registers, ~6 vector instructions
//     per WORD where a gathered posting costs ~150.  The long lists of frequent terms are most of a query's postings.
//   * FOREGROUND LISTS (all the others; every list while no floor is published) are gathered as before.  A foreground posting
//     marks its document with a RETURNING bitmap atomic: the bit was set before, or the background bitmap holds the document ->
//     the document has several contributions and goes into a second bitmap (`multi`); otherwise it is a singleton, scored on
//     the spot (the fold of ONE contribution).  Foreground postings of multi documents are compacted into a dense LDS list.
//   * Only the multi documents (3 % of the documents) get ranks (popcount prefix over `multi`), presence masks, cells and the fold
//     in (token, reference) order; one lane per multi document looks its background contributions up — bitmap word, position =
//     word prefix + set bits below, normalised tf — and appends them to the list before the cells are laid out.
//   * A foreground list whose own bound is under the floor (no bitmap for it, or the sum rule left it out): its singletons are
//     marked and counted, not scored (a wave iteration whose singletons are all of that kind skips division and ordering).
//   * `count` = documents touched (background words by popcount, foreground first touches) - documents found not to be in the map.
//
// Exact: bounds hold in f32 with margins for every rounding (see seg_cup), `applied` (bm25.rs: S normal, the term a number) holds
// for every background posting by the store's tame-tf guarantee and the boost check, a floor only ever drops keys that cannot be
// among the topk.  Threshold queries that need more than one token and filtered batches take no background lists.
//
// Compiled with -ffp-contract=off (see bm25_kernels.hip).
#include "bm25_ranges.hpp"

#include "bm25_ranges_dev.hpp"

namespace orama {

namespace {

constexpr int kThreads = (int)kRangeThreads;
constexpr int kWaves = kThreads / 64;
static_assert(kRangeCap / kThreads == 8, "a lane carries at most 8 postings through the phases");
constexpr uint32_t kBitWords = kRangeMaxWidth / 32;
constexpr int kWordsPerThread = kBitWords / kThreads;
constexpr uint32_t kBlkShift = kThreads == 256 ? 5 : 6, kBlocks = kRangeCap >> kBlkShift;
constexpr uint32_t kCells = 4 * kThreads;     // contributions of multi documents one range may hold ...
constexpr uint32_t kMultiMax = 2 * kThreads;  // ... and how many such documents: more raise `overflow` (narrower ranges), as in the round-5 body
constexpr int kCellRounds = kCells / kThreads, kMultiRounds = kMultiMax / kThreads;
constexpr uint32_t kRefs = 32;                // references of a query (32-bit presence masks: the launcher checks)
static_assert(kRangeCap % kThreads == 0 && kBitWords % kThreads == 0 && kCells % kThreads == 0 && kMultiMax % kThreads == 0, "whole threads");
static_assert(kRangeMaxWidth <= 0x10000u && kRangeMaxRefs <= 64 && kMultiMax <= 1024, "local document: 16 bits, reference: 6 bits, multi rank: 10 bits of a word");
static_assert(kBlocks <= 64, "the block table is built by one wave");
static_assert(kCells == kBitWords && kMultiMax * 2 == kBitWords, "cells take the multi bitmap's place, cell bases + documents the rank table's");

// 19.3 KB: eight workgroups per CU, like the round-5 body (its speed is the workgroups resident per CU).
struct FastLds {
    uint32_t bits[kBitWords];    // A: documents touched by a foreground posting.  From B on: the multi list's words (document | reference << 16 | rank << 22)
    uint32_t multi[kBitWords];   // A-C.b2: documents with several contributions.  From C.d on: the cells (f32)
    float mp_val[kCells];        // the multi list's normalised tf
    union {
        uint32_t bgu[kBitWords];  // prologue-A: union of the background lists' bitmap words
        struct {
            uint16_t mrank[kBitWords];  // C.a-b2: exclusive popcount prefix of `multi`.  From C.c on: [kMultiMax] first cell | [kMultiMax] local document
            uint32_t mmask[kMultiMax];  // per multi document (by rank): the references that hold it; after the fold: its score word
        } c;
    } u;
    unsigned long long seg_pos[kRefs];    // first posting of each reference inside this range
    unsigned long long seg_begin[kRefs];  // first posting of each reference's list
    unsigned long long seg_acc[kRefs];    // RangeSeg::acc_off
    uint32_t seg_off[kRefs + 1];
    uint32_t seg_key[kRefs];   // token << 10 | rank
    uint32_t seg_pkb[kRefs];   // kept | token << 25 | reference << 17
    uint32_t seg_ub[kRefs];    // ordered(what no singleton of this list can exceed); ~0: no such bound
    float seg_cup[kRefs];      // that bound as a number when the list may go to the background, +inf otherwise
    float seg_boost[kRefs];
    uint16_t blk_run[kBlocks];
    float idf[kMaxTokens];
    uint32_t wave_floor[kWaves], wave_pub[kWaves], scan_tot[kWaves];
    uint32_t red[4];           // [1] documents touched, [2] of which not in the map
    uint32_t list_cursor, cell_cursor;
    uint32_t pub_floor;
    __device__ __forceinline__ uint32_t* mp_key() { return bits; }
    __device__ __forceinline__ float* cellv() { return reinterpret_cast<float*>(multi); }
    __device__ __forceinline__ uint16_t* md_cb() { return u.c.mrank; }
    __device__ __forceinline__ uint16_t* md_dl() { return u.c.mrank + kMultiMax; }
};
static_assert(kThreads != 256 || sizeof(FastLds) <= 20480, "eight workgroups per CU");

struct FastRange {
    uint32_t qi, cap, doc0, n_words;
    uint32_t bg;  // references whose lists are background lists for this workgroup (one bit each)
    bool publish;
    uint32_t pub_slot;
};

// the jshare-th largest of the wave's 64 values, to its leading 24 bits (a lower bound of it); 0: fewer than jshare lanes hold one
__device__ __forceinline__ uint32_t wave_nth_largest(uint32_t v, uint32_t jshare) {
    uint32_t w = 0u;
#pragma unroll
    for (int bit = 31; bit >= 8; --bit) {
        const uint32_t t = w | (1u << bit);
        w = (uint32_t)__popcll(__ballot(v >= t)) >= jshare ? t : w;
    }
    return w;
}

template <int NITER, bool FILTER>
__device__ __forceinline__ void fast_body(const RangeBatch& b, const RangeQuery& q, const FastRange& rg, FastLds& L, uint32_t my_docs) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t cap = rg.cap, qi = rg.qi, doc0 = rg.doc0, n_words = rg.n_words;
    const uint32_t pub_floor = L.pub_floor;
    const bool bg_any = rg.bg != 0u;

    // ---- A. gather the foreground postings; mark the document; flag postings of lists whose singletons cannot reach the floor
    uint32_t pk[NITER];  // [kept:1 | token:6 | below-floor list:1 | pad:1 | reference:6 | pad:1 | local document:16]
    float pv[NITER];     // normalised tf (boost included)
#pragma unroll
    for (int g = 0; g < NITER; g += 4) {
        constexpr int G = 4;
        unsigned long long pos[G];
        uint32_t run[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int n = g + j;
            if (n >= NITER) break;
            const uint32_t e = min(threadIdx.x + n * kThreads, cap - 1u);
            uint32_t lo = L.blk_run[e >> kBlkShift];
            while (L.seg_off[lo + 1] <= e) ++lo;  // (runs are ~100 postings: almost always zero steps; background lists are empty runs)
            run[j] = lo;
            pos[j] = L.seg_pos[lo] + (e - L.seg_off[lo]);
        }
        uint32_t doc[G];
        float ntf[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (g + j >= NITER) break;
            doc[j] = b.post_doc[pos[j]];
            ntf[j] = b.post_ntf[pos[j]];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int n = g + j;
            if (n >= NITER) break;
            bool kept = threadIdx.x + n * kThreads < cap;
            if constexpr (FILTER) {  // collect_contributions_with_filter: filtered docs never reach the scorer
                const uint64_t id = b.docs ? b.docs[doc[j]] : b.dense_base + doc[j];
                kept = kept && id < b.allow_bits && ((b.allow[id >> 6] >> (id & 63)) & 1ull);
            }
            const uint32_t dl = doc[j] - doc0;
            const float val = L.seg_boost[run[j]] * ntf[j];
            // S = 0.0 + val is a positive normal number under 2^100: the term idf (k+1) S / (k + S) then is a number in
            // [0, idf (k+1) (1 + 3 ulp)] (seg_ub's conditions on idf and k), i.e. `applied` holds and the list's bound applies
            const bool tame = (__builtin_bit_cast(uint32_t, val) - 0x00800000u) < 0x71000000u;
            const bool under = tame && L.seg_ub[run[j]] < pub_floor;
            pk[n] = kept ? (L.seg_pkb[run[j]] | (under ? 0x1000000u : 0u) | dl) : 0u;
            pv[n] = val;
            // (both atomics by every lane; a dropped posting ORs nothing into a word of its own)
            const uint32_t wd = kept ? dl >> 5 : threadIdx.x, bit = kept ? 1u << (dl & 31u) : 0u;
            const uint32_t old = atomicOr(&L.bits[wd], bit);
            const uint32_t inbg = bg_any ? L.u.bgu[wd] : 0u;
            atomicOr(&L.multi[wd], (old | inbg) & bit);  // (every later posting of the document sets it again: idempotent)
            my_docs += (kept && ((old | inbg) & bit) == 0u) ? 1u : 0u;  // the first touch of a document no background list holds
        }
    }
    __syncthreads();  // `bits` and the background union are dead from here on

    // ---- B. singletons are scored by their posting's lane; foreground postings of multi documents go to the dense list
    const float k1 = q.k + 1.0f;
    const bool thr_ok = !(q.use_threshold && 1u < q.threshold);  // a singleton holds ONE token
    uint32_t* const mp_key = L.mp_key();
    uint32_t my_bad = 0;  // documents this lane found NOT to be in the map
    uint32_t ko[NITER];
#if ORAMA_COMPARISON_KERNELS
    uint32_t st_skipped = 0, st_iters = 0, st_under = 0, st_kept = 0;  // ORAMA_K3R_DBG=16 with ORAMA_K3R_STATS=1
#endif
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        const uint32_t dl = pk[n] & 0xffffu;
        const bool kept = (pk[n] >> 31) != 0u;
        const bool is_multi = kept && ((L.multi[dl >> 5] >> (dl & 31u)) & 1u);
        const bool under = ((pk[n] >> 24) & 1u) != 0u;
        ko[n] = 0u;
#if ORAMA_COMPARISON_KERNELS
        if (__ballot(kept) != 0ull) {
            ++st_iters;
            if (__ballot(kept && !is_multi && !under) == 0ull) ++st_skipped;
        }
        st_under += (kept && under) ? 1u : 0u;
        st_kept += kept ? 1u : 0u;
#endif
        if (__ballot(is_multi) != 0ull) {  // (wave-uniform; two or three lanes of most iterations)
            if (is_multi) {
                const uint32_t slot = atomicAdd(&L.list_cursor, 1u);
                if (slot < kCells) {  // (beyond: the range overflows, nothing of it is used)
                    mp_key[slot] = dl | (((pk[n] >> 17) & 63u) << 16);
                    L.mp_val[slot] = pv[n];
                }
            }
        }
        if (__ballot(kept && !is_multi && !under) != 0ull) {  // (wave-uniform) a singleton that may reach the floor: every lane evaluates
            const float sum = 0.0f + 1.0f * pv[n];                                      // Iterator::sum() from 0.0, weight 1.0
            const float term = L.idf[(pk[n] >> 25) & 63u] * k1 * sum / (q.k + sum);      // bm25f_score, bm25.rs:124-126
            const bool applied = f32_is_normal(sum) && term == term;
            const float score = 0.0f + term * 1.0f;                                     // entry(key).or_insert(0.0) += term * boost 1.0
            const bool single = kept && !is_multi;
            const bool in_map = single && applied && thr_ok;
            my_bad += (single && !in_map) ? 1u : 0u;
            ko[n] = in_map ? f32_to_ordered(score) : 0u;
        } else {
            // every singleton of this wave iteration belongs to a list under the floor: in the map unless the threshold asks for
            // more than one token (S is a positive normal number, the term a number), never a survivor — nothing to evaluate
            my_bad += (kept && !is_multi && !thr_ok) ? 1u : 0u;
        }
    }
#if ORAMA_COMPARISON_KERNELS
    if (b.debug & 16u) {
        if (lane == 0) {
            atomicAdd(&b.results[qi].pad0[1], st_skipped);
            atomicAdd(&b.results[qi].pad0[3], st_iters);
            if (threadIdx.x == 0 && pub_floor) atomicAdd(&b.results[qi].pad0[0], 1u);
            if (threadIdx.x == 0) atomicMax(&b.results[qi].score_floor, pub_floor);
            if (threadIdx.x == 0) atomicAdd(&b.results[qi].pad0[2], (uint32_t)__popc(rg.bg));
        }
        atomicAdd(&b.results[qi].pad0[4], st_under);
        atomicAdd(&b.results[qi].pad0[5], st_kept);
    }
#endif
    uint32_t best = ko[0];
#pragma unroll
    for (int n = 1; n < NITER; ++n) best = max(best, ko[n]);
    const uint32_t kq = q.topk, jshare = (kq + kWaves - 1u) / kWaves;
    // the workgroup's own floor: the jshare-th best of each wave's lane bests — at least topk of this range's documents reach the
    // smallest of the four (bm25_ranges.hip, phase 5)
    const uint32_t wb = (kq != 0u && jshare <= 64u) ? wave_nth_largest(best, jshare) : 0u;
    if (lane == 0) L.wave_floor[wave] = wb;
    if (rg.publish) {  // (workgroup-uniform: one of the query's first ranges) the 4th largest lane best of each wave
        const uint32_t w4 = wave_nth_largest(best, 4u);
        if (lane == 0) L.wave_pub[wave] = w4;
    }
    __syncthreads();
    uint32_t n_cells = L.list_cursor;
    uint32_t n_multi = 0;
    if (n_cells > kCells) {
        if (threadIdx.x == 0) {
            b.results[qi].overflow = 1;
            atomicMax(&b.results[qi].pad1[0], (n_cells * 16u + kCells - 1u) / kCells);
        }
        return;
    }

    // ---- C. the multi documents (workgroup-uniform)
    if (n_cells != 0u) {
        uint16_t* const mrank = L.u.c.mrank;
        uint32_t* const mmask = L.u.c.mmask;
        // a. their ranks: exclusive popcount prefix over the words of `multi`
        {
            uint32_t cnt[kWordsPerThread], sum = 0;
#pragma unroll
            for (int n = 0; n < kWordsPerThread; ++n) {
                const uint32_t w = threadIdx.x * kWordsPerThread + n;
                cnt[n] = w < n_words ? (uint32_t)__popc(L.multi[w]) : 0u;
                sum += cnt[n];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(incl, off, 64);
                if ((int)lane >= off) incl += y;
            }
            if (lane == 63) L.scan_tot[wave] = incl;
#pragma unroll
            for (int t = 0; t < kMultiRounds; ++t) mmask[threadIdx.x + t * kThreads] = 0u;
            __syncthreads();
            uint32_t excl = incl - sum;
#pragma unroll
            for (int w = 0; w < kWaves - 1; ++w) excl += (uint32_t)w < wave ? L.scan_tot[w] : 0u;
#pragma unroll
            for (int n = 0; n < kWordsPerThread; ++n) {
                const uint32_t w = threadIdx.x * kWordsPerThread + n;
                if (w < n_words) mrank[w] = (uint16_t)excl;
                excl += cnt[n];
            }
#pragma unroll
            for (int w = 0; w < kWaves; ++w) n_multi += L.scan_tot[w];
        }
        if (n_multi > kMultiMax) {
            if (threadIdx.x == 0) {
                b.results[qi].overflow = 1;
                atomicMax(&b.results[qi].pad1[0], (n_multi * 16u + kMultiMax - 1u) / kMultiMax);
            }
            return;
        }
        __syncthreads();
        // b. every listed (foreground) posting: the rank of its document, its reference into the document's presence mask
#pragma unroll
        for (int j = 0; j < kCellRounds; ++j) {
            const uint32_t i = threadIdx.x + j * kThreads;
            if (j * kThreads + wave * 64u < n_cells) {  // (wave-uniform)
                if (i < n_cells) {
                    const uint32_t key = mp_key[i], dl = key & 0xffffu, ref = key >> 16;
                    const uint32_t rank = (uint32_t)mrank[dl >> 5] + (uint32_t)__popc(L.multi[dl >> 5] & ((1u << (dl & 31u)) - 1u));
                    mp_key[i] = key | (rank << 22);
                    atomicOr(&mmask[rank], 1u << ref);
                }
            }
        }
        __syncthreads();
        if (bg_any) {  // (workgroup-uniform)
            // b2. one lane per multi document: the document of rank r (the word whose prefix range holds r, then the set bit of that
            // order), and what the background lists contribute to it — bitmap word, position in the list, normalised tf
            const uint32_t w0 = doc0 >> 5;
            for (uint32_t r = threadIdx.x; r < n_multi; r += kThreads) {
                uint32_t lo = 0, hi = n_words;  // the last word whose prefix is <= r
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)mrank[mid] <= r) lo = mid; else hi = mid;
                }
                uint32_t v = L.multi[lo];
                for (uint32_t skip = r - (uint32_t)mrank[lo]; skip != 0; --skip) v &= v - 1u;
                const uint32_t bitpos = (uint32_t)__ffs((int)v) - 1u, dl = lo * 32u + bitpos;
                for (uint32_t m = rg.bg; m != 0; m &= m - 1u) {
                    const uint32_t i = (uint32_t)__ffs((int)m) - 1u;
                    const uint32_t* acc = b.post_acc + (L.seg_acc[i] - 1u);
                    const uint32_t gw = w0 + lo;
                    const uint32_t word = gw < b.acc_words ? acc[gw] : 0u;
                    if ((word >> bitpos) & 1u) {
                        const unsigned long long at = L.seg_begin[i] + acc[b.acc_words + gw] + (uint32_t)__popc(word & ((1u << bitpos) - 1u));
                        const uint32_t slot = atomicAdd(&L.list_cursor, 1u);
                        if (slot < kCells) {
                            mp_key[slot] = dl | (i << 16) | (r << 22);
                            L.mp_val[slot] = L.seg_boost[i] * b.post_ntf[at];
                        }
                        atomicOr(&mmask[r], 1u << i);
                    }
                }
            }
            __syncthreads();
            n_cells = L.list_cursor;
            if (n_cells > kCells) {
                if (threadIdx.x == 0) {
                    b.results[qi].overflow = 1;
                    atomicMax(&b.results[qi].pad1[0], (n_cells * 16u + kCells - 1u) / kCells);
                }
                return;
            }
        }
        // (the rank table is dead: cell bases and documents take its place)
        // c. the contribution of a document's FIRST reference asks for its cells — one per contribution, in reference order
        uint16_t* const md_cb = L.md_cb();
        uint16_t* const md_dl = L.md_dl();
#pragma unroll
        for (int j = 0; j < kCellRounds; ++j) {
            const uint32_t i = threadIdx.x + j * kThreads;
            if (j * kThreads + wave * 64u < n_cells) {
                if (i < n_cells) {
                    const uint32_t key = mp_key[i], rank = key >> 22, ref = (key >> 16) & 63u;
                    const uint32_t m = mmask[rank];
                    if ((m & ((1u << ref) - 1u)) == 0u) {
                        md_cb[rank] = (uint16_t)atomicAdd(&L.cell_cursor, (uint32_t)__popc(m));
                        md_dl[rank] = (uint16_t)(key & 0xffffu);
                    }
                }
            }
        }
        __syncthreads();  // (the multi bitmap is dead: the cells take its place)
        // d. every listed contribution parks its normalised tf in its cell
        float* const cellv = L.cellv();
#pragma unroll
        for (int j = 0; j < kCellRounds; ++j) {
            const uint32_t i = threadIdx.x + j * kThreads;
            if (j * kThreads + wave * 64u < n_cells) {
                if (i < n_cells) {
                    const uint32_t key = mp_key[i], rank = key >> 22, ref = (key >> 16) & 63u;
                    const uint32_t below = mmask[rank] & ((1u << ref) - 1u);
                    cellv[(uint32_t)md_cb[rank] + (uint32_t)__popc(below)] = L.mp_val[i];
                }
            }
        }
        __syncthreads();
        // e. one lane per document: its cells in bit order are its contributions in (token, reference) order — folded exactly
        // as BM25Scorer::add / get_scores would; the score word replaces the mask
        for (uint32_t i = threadIdx.x; i < n_multi; i += kThreads) {
            uint32_t m = mmask[i];
            DocFold f;
            for (uint32_t c = md_cb[i]; m != 0; ++c) {
                const uint32_t ref = (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1;
                f.add(L.seg_key[ref] >> 10, cellv[c], L.idf, q.k, k1);
            }
            const bool in_map = f.finish(L.idf, q.k, k1, q.use_threshold, q.threshold);
            my_bad += in_map ? 0u : 1u;
            mmask[i] = (in_map && f.score == f.score) ? f32_to_ordered(f.score) : 0u;
        }
    }

    // ---- D. the floor, the survivors, their append behind the query's cursor, the published word
    if (my_docs) atomicAdd(&L.red[1], my_docs);
    if (my_bad) atomicAdd(&L.red[2], my_bad);
    __syncthreads();
    uint32_t* const mscore = L.u.c.mmask;
    uint32_t lb = L.wave_floor[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) lb = min(lb, L.wave_floor[w]);
    const uint32_t floor_w = kq == 0u ? 0xffffffffu : max(max(lb, pub_floor), 1u);
    unsigned long long sm[NITER + kMultiRounds];
    uint32_t tot = 0u;
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        sm[n] = __ballot(ko[n] >= floor_w);
        tot += (uint32_t)__popcll(sm[n]);
    }
#pragma unroll
    for (int t = 0; t < kMultiRounds; ++t) {
        const uint32_t i = threadIdx.x + t * kThreads;
        sm[NITER + t] = (uint32_t)t * kThreads < n_multi ? __ballot(i < n_multi && mscore[i] >= floor_w) : 0ull;
        tot += (uint32_t)__popcll(sm[NITER + t]);
    }
    // one cursor bump per WAVE (bm25_ranges.hip: a workgroup-wide bump put two barriers and a returning global atomic in a row
    // on every workgroup's critical path)
    uint32_t at = 0u;
    if (lane == 0 && tot) at = atomicAdd(&b.results[qi].n_keys, tot);
    at = __shfl(at, 0, 64);
    if (threadIdx.x == 0 && L.red[1] != L.red[2]) atomicAdd(&b.results[qi].count, L.red[1] - L.red[2]);
    unsigned long long* const lst = b.keys + q.key_off;
    const unsigned long long below_me = (1ull << lane) - 1ull;
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        if (ko[n] >= floor_w)
            lst[at + (uint32_t)__popcll(sm[n] & below_me)] = ((unsigned long long)ko[n] << 32) | (unsigned long long)(~(doc0 + (pk[n] & 0xffffu)));
        at += (uint32_t)__popcll(sm[n]);
    }
#pragma unroll
    for (int t = 0; t < kMultiRounds; ++t) {
        if ((sm[NITER + t] >> lane) & 1ull) {
            const uint32_t i = threadIdx.x + t * kThreads;
            lst[at + (uint32_t)__popcll(sm[NITER + t] & below_me)] =
                ((unsigned long long)mscore[i] << 32) | (unsigned long long)(~(doc0 + (uint32_t)L.md_dl()[i]));
        }
        at += (uint32_t)__popcll(sm[NITER + t]);
    }
    if (rg.publish && wave == 0) {
        // what this range publishes for the floor of the query's later ranges: a score word 4 of its documents reach — the larger
        // of (the 4th largest lane best of its best wave: single-term documents) and (the 4th largest score among its documents
        // that hold SEVERAL of the query's terms: the query's k-th best lives among those) — bm25_ranges.hip, same rule
        uint32_t v = L.wave_pub[0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v = max(v, L.wave_pub[w]);
        if (n_multi >= 4u) {
            uint32_t mbest = 0u;
            for (uint32_t i = lane; i < n_multi; i += 64u) mbest = max(mbest, mscore[i]);
            v = max(v, wave_nth_largest(mbest, 4u));
        }
        if (lane == 0 && v) atomicMax(&b.score_pub[(size_t)qi * kScorePubRanges + rg.pub_slot], v);
    }
}

template <bool FILTER>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void range_score_fast_kernel(RangeBatch b) {
    __shared__ FastLds L;
    // (query, range) of this workgroup: the batch is scored in kRangeStripes passes over its queries (bm25_ranges.hip)
    uint32_t lo = 0, hi = kRangeStripes * kRangeBatchMax;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (b.stripe_start[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const uint32_t qi = lo % kRangeBatchMax, stripe = lo / kRangeBatchMax;
    const RangeQuery q = b.queries[qi];
    const uint32_t r = (uint32_t)((uint64_t)stripe * q.n_ranges / kRangeStripes) + (blockIdx.x - b.stripe_start[lo]);
    if (r >= q.n_ranges) return;
    const uint32_t ns = q.seg_end - q.seg_begin;  // (<= 32: the launcher checks)
    const RangeSeg* segs = b.segs + q.seg_begin;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    FastRange rg;
    rg.qi = qi;
    rg.n_words = (min(q.width, kRangeMaxWidth) + 31u) >> 5;
    rg.doc0 = r * q.width;
    rg.bg = 0u;
    rg.publish = r < kScorePubRanges;  // the publishers: the query's first ranges — the ones the striped grid scores first
    rg.pub_slot = r;
    const bool thr_ok = !(q.use_threshold && 1u < q.threshold);
    // background lists need whole bitmap words per range, every document in the map (no filter, no multi-token threshold)
    const bool bg_possible = !FILTER && thr_ok && b.post_acc != nullptr && (q.width & 31u) == 0u && q.width <= kRangeMaxWidth;

    // ---- 0. wave 0: the range's run of every reference, each list's bound; wave 1: the published floor; waves 1-3: clear the bitmaps
    uint32_t run_len = 0;  // (wave 0, lane = reference) postings of the reference inside this range
    if (wave == 0) {
        if (lane < 4) L.red[lane] = 0;
        if (lane == 0) {
            L.list_cursor = 0u;
            L.cell_cursor = 0u;
            L.seg_off[0] = 0;
        }
        static_assert(kMaxTokens == 64, "one lane per token");
        L.idf[lane] = lane < q.n_tokens ? b.idf[(size_t)qi * kMaxTokens + lane] : 0.0f;
        const float k1 = q.k + 1.0f;
        const uint32_t* row = b.bounds + q.bounds_base + (uint64_t)r * ns;
        if (lane < ns) {
            const uint32_t i = lane;
            const uint32_t b0 = row[i], b1 = row[ns + i];
            const RangeSeg sg = segs[i];
            const uint32_t tok = sg.tok_rank >> 10;
            L.seg_pos[i] = sg.post_begin + b0;
            L.seg_begin[i] = sg.post_begin;
            L.seg_acc[i] = sg.acc_off;
            L.seg_key[i] = sg.tok_rank;
            L.seg_pkb[i] = 0x80000000u | (tok << 25) | (i << 17);
            L.seg_boost[i] = sg.boost;
            // what no singleton of this list can exceed.  Its score is fl(fl(fl(idf k1) S) / fl(k + S)) with S a positive
            // normal number under 2^100 (checked per posting / guaranteed per list): for 0 <= k < 1e30 and c = fl(idf k1) in
            // [0, 1e8] every intermediate is a finite number, fl(k + S) >= S (1 - u), so the score is at most
            // c (1 + u)^2 / (1 - u) < c (1 + 2^-20) — the factor below.  Anything else (negative or huge k, NaN or huge idf): no bound.
            // (idf[tok]: this wave's own LDS write above — LDS operations of one wave complete in order)
            const float c = L.idf[tok & 63u] * k1;
            const bool bounded = q.k >= 0.0f && q.k < 1e30f && c >= 0.0f && c <= 1e8f;
            const float c_up = c * 1.000001f;
            L.seg_ub[i] = bounded ? f32_to_ordered(c_up) : 0xffffffffu;
            // a background candidate: a dense list with a bitmap (every normalised tf within [2^-60, 2^60]: the store's guarantee)
            // under a boost within [2^-40, 2^39] — every contribution a positive normal number under 2^100
            const bool cand = bg_possible && bounded && sg.acc_off != 0ull && sg.boost >= 0x1p-40f && sg.boost <= 0x1p39f;
            L.seg_cup[i] = cand ? c_up : __builtin_huge_valf();
            run_len = b1 - b0;
        }
    } else {
        for (uint32_t w = threadIdx.x - 64u; w < rg.n_words; w += kThreads - 64u) {
            L.bits[w] = 0u;
            L.multi[w] = 0u;
        }
        if (wave == 1) {
            // the floor the query's first ranges have published so far: the ceil(topk / 4)-th largest of their words (each vouches
            // for 4 documents at or above its word; unpublished = 0), to its leading 24 bits; 0: not enough yet
            const uint32_t v = __hip_atomic_load(&b.score_pub[(size_t)qi * kScorePubRanges + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t need = (q.topk + 3u) / 4u;
            const uint32_t fl = (q.topk != 0u && need <= kScorePubRanges) ? wave_nth_largest(v, need) : 0u;
            if (lane == 0) L.pub_floor = fl;
        }
    }
    __syncthreads();
    // ---- 0b. every wave: the background lists — the candidates of smallest bound whose bounds sum to less than the floor (a
    // document that occurs in those lists only scores at most the sum: it cannot be among the topk).  The sum carries a margin
    // for its own roundings and the fold's (32 additions each, 1 + 70 u < 1.00001).
    uint32_t my_docs = 0;
    {
        const uint32_t pub_floor = L.pub_floor;
        if (bg_possible && pub_floor != 0u) {  // (workgroup-uniform)
            const float floor_f = ordered_to_f32(pub_floor);
            const float c = lane < ns ? L.seg_cup[lane] : __builtin_huge_valf();
            float below = 0.0f;  // the bounds of the candidates ordered before this one, and its own
            for (uint32_t j = 0; j < ns; ++j) {
                const float cj = L.seg_cup[j];
                below += (cj < c || (cj == c && j <= lane)) ? cj : 0.0f;
            }
            rg.bg = (uint32_t)__ballot(lane < ns && c < __builtin_huge_valf() && below * 1.00001f < floor_f);
        }
    }
    if (wave == 0) {
        // the foreground runs' offsets among the gathered postings (a background list gathers nothing), the block table
        uint32_t x = ((rg.bg >> lane) & 1u) ? 0u : run_len;
        if (lane >= ns) x = 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {  // inclusive scan of the run lengths
            const uint32_t y = __shfl_up(x, off, 64);
            if ((int)lane >= off) x += y;
        }
        if (lane < ns) L.seg_off[lane + 1] = x;
        const uint32_t carry = __shfl(x, 63, 64);
        // block table: the run that holds gathered posting 32 * lane (the wave's own LDS writes above are visible to it)
        if (carry <= kRangeCap && (lane << kBlkShift) < carry) {
            const uint32_t e = lane << kBlkShift;
            uint32_t lo2 = 0, hi2 = ns;
            while (hi2 - lo2 > 1) {
                const uint32_t mid = (lo2 + hi2) >> 1;
                if (L.seg_off[mid] <= e) lo2 = mid; else hi2 = mid;
            }
            L.blk_run[lane] = (uint16_t)lo2;
        }
    } else if (rg.bg != 0u) {
        // the background lists' documents of this range, as words: OR-ed into one bitmap, counted
        const uint32_t w0 = rg.doc0 >> 5;
        for (uint32_t w = threadIdx.x - 64u; w < rg.n_words; w += kThreads - 64u) {
            uint32_t acc_w = 0u;
            if (w0 + w < b.acc_words)
                for (uint32_t m = rg.bg; m != 0; m &= m - 1u) {
                    const uint32_t i = (uint32_t)__ffs((int)m) - 1u;
                    acc_w |= b.post_acc[(L.seg_acc[i] - 1u) + w0 + w];
                }
            L.u.bgu[w] = acc_w;
            my_docs += (uint32_t)__popc(acc_w);
        }
    }
    __syncthreads();
    const uint32_t cap = L.seg_off[ns];
    rg.cap = cap;
    if (cap == 0) {
        // nothing to gather: an empty range, or every list of it in the background — those documents are counted
        if (rg.bg != 0u) {
            if (my_docs) atomicAdd(&L.red[1], my_docs);
            __syncthreads();
            if (threadIdx.x == 0 && L.red[1]) atomicAdd(&b.results[qi].count, L.red[1]);
        }
        return;
    }
    if (cap > kRangeCap) {
        // the query is rerun with smaller ranges (a compact list has no slots to clear: nothing of this range is appended)
        if (threadIdx.x == 0) {
            b.results[qi].overflow = 1;
            atomicMax(&b.results[qi].pad1[0], (cap * 16u + kRangeCap - 1u) / kRangeCap);
        }
        return;
    }
    // rounds of the per-posting phases (workgroup-uniform): one straight-line body per round count that occurs
    const uint32_t n_iter = (cap + kThreads - 1) / kThreads;
    if (n_iter <= 1) fast_body<1, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter <= 2) fast_body<2, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter <= 3) fast_body<3, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter <= 4) fast_body<4, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter == 5) fast_body<5, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter == 6) fast_body<6, FILTER>(b, q, rg, L, my_docs);
    else if (n_iter == 7) fast_body<7, FILTER>(b, q, rg, L, my_docs);
    else fast_body<8, FILTER>(b, q, rg, L, my_docs);
}

}  // namespace

int launch_range_score_fast(orama_ctx* ctx, const RangeBatch& b, hipStream_t stream) {
    const uint32_t grid = b.range_start[b.n_queries];
    ORAMA_REQUIRE(b.compact_keys && b.post_ntf && !b.map_idx && !b.omc_dense && !b.any_minmax && b.max_refs <= kRefs,
                  "internal: the fast range scorer takes plain top-k batches with compact key lists");
    ORAMA_REQUIRE(b.score_pub, "internal: compact key lists without their published-score table");
    ORAMA_REQUIRE(b.stripe_start && b.stripe_total == grid, "internal: stripe table not filled");
    (void)ctx;
    if (b.allow) hipLaunchKernelGGL(range_score_fast_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, b);
    else hipLaunchKernelGGL(range_score_fast_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, b);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
