// bm25_ranges.hip — K3r: BM25F over resident postings, partitioned by document range, batched over queries.
//
// K3 (bm25_kernels.hip) scatters every posting into a per-document record in HBM (random 64-byte read-modify-writes
// into a table of n_docs records) and walks the records again to finalise: ~130 B of random HBM traffic per posting
// and ~20 small launches per query.  Posting lists are sorted by document (orama_post_build checks it), so the
// union of a query's lists can be cut anywhere in DOCUMENT space:
//
//   range_bounds_kernel   one lower-bound search per (list, range boundary): where does each list cross the range boundaries
//                         (range = `width` consecutive local documents, chosen per query so that a range
//                         holds ~1 280 postings on average).  bounds[list][r] = postings of the list in ranges < r.
//   range_score_kernel    one workgroup per (query, range) pair that exists (1-D grid).  SORT-FREE since round 4:
//                           1. gather the <= 2 048 postings of the range (each list contributes one contiguous run:
//                              coalesced), mark each posting's local document in an LDS bitmap (one bit per document of the range);
//                           2. popcount prefix over the bitmap words = dense rank of every touched document;
//                           3. every posting ORs its token into its document's presence mask (one 64-bit word per rank);
//                           4. a document whose mask has ONE token of a one-list token is a singleton (97 % of the documents
//                              of a 12-token query): its posting's thread scores it on the spot — score = 0.0 + idf (k+1) S / (k + S) —
//                              and writes its key.  The other documents get popcount(mask) cells (LDS cursor, one
//                              returning atomic per such document), their postings add their normalised tf into the cell of
//                              their token — lists of one token in reference order, a barrier only between ranks of a token —
//                              and one thread per document folds the cells with tokens ascending.
//                         Those are exactly the additions BM25Scorer::add / get_scores perform (bm25.rs:369-428) in the
//                         same order, so scores are bit-identical to K3, to the round-3 merge-tree form of this kernel
//                         (bm25_ranges_merge.hip, comparison builds) and to the CPU restatement.  Output: one 64-bit key
//                         ordered(score) << 32 | ~document per scored document, at the slot of one of its postings
//                         (slot base of a range = sum of its bounds: no cursor, no atomics), 0 elsewhere.
//   launch_keys_topk      (select.hip) exact top-k over the key lists of the whole batch.
//
// Round 3's form sorted the postings of a range by (document, token, rank) with a merge tree of 64-bit keys: 8.6 VALU
// wave-instructions per posting (profiles/r03_k3r_sq_counters.md), every SIMD issuing for the whole launch.  Nothing is
// sorted here: a posting costs one run lookup, two LDS atomics without return, one rank computation and — for the 97 % —
// one IEEE division.  The normalised tf's query-independent part tf / (1 - b + b len / avglen) is stored per posting by the
// store (same operations, same bits; recomputed when the average lengths move), which removes two more divisions.
// The plain top-k search — no score map, OMC multipliers or min / max in the batch — runs an instantiation of its own
// (PLAIN): none of those branches compiled in, and the singleton's score evaluated by every lane and SELECTED instead of
// branched to; the general form nests four divergent branches per posting, and the scalar unit (exec masks, branches) was as
// busy as the vector units.  2.50 vector + 2.41 scalar wave-instructions per posting, 62 registers: eight workgroups per CU,
// 3.5 us per C4-shaped query (profiles/r04_k3r_sq_counters_v6.md).
//
// HBM traffic: 8 B per posting read by the score kernel + 8 B per posting written and read once by the top-k (the
// bounds searches touch ~17 elements per list and range).
// Nothing is sized by n_docs: a query needs 8 B per referenced posting of scratch instead of K3's 136 B per
// document of the index.  df (token_score.rs:262-275): known on the host when no filter applies and every token has
// one list; otherwise the same kernel runs once in counting mode first.
// A range that holds more than 2048 postings (documents of a term clustered in id space) raises the query's
// `overflow` flag; the host reruns that query with 8x smaller ranges (always terminates: a range of one document
// holds at most one posting per list).
//
// Compiled with -ffp-contract=off (see bm25_kernels.hip).
#include "bm25_ranges.hpp"

#include "bm25_ranges_dev.hpp"

namespace orama {

namespace {

constexpr int kBoundsThreads = 256;

// bounds[query][range][reference] = lower bound of the range's first document in the reference's (document-sorted) list.
// SIXTEEN lanes per entry search together: every round they probe 16 evenly spaced elements of the window at once and keep
// the sixteenth that holds the answer — 4 dependent loads for a list of 50 K postings where a binary search makes 17.  The
// launch is a chain of dependent loads and nothing else, and it sits at the head of every query: 14 us alone, 44 us beside the
// 4.3 ms vector scan of a hybrid query (profiles/r03_hybrid_tail_timeline_after.log), with one thread per entry.
// (The very first form walked every referenced posting — one thread per posting, 19 M threads per 32-query batch.)
// kBoundsLanes = 16 for small batches (the launch is latency: one query, or the few of a hybrid search); big batches hide the
// latency behind their own parallelism and pay for the 16 probes per round instead (1.22 against 0.55 us per query at 32
// queries per launch, profiles/r04_bounds_lanes.log): they search with ONE lane per entry, a plain binary search.
template <uint32_t kBoundsLanes>
__global__ __launch_bounds__(kBoundsThreads) void range_bounds_kernel(RangeBatch b) {
    const uint32_t qi = blockIdx.y;
    // the query's result words start at zero: cleared here, by the first launch of the set, instead of by a fill command
    // of their own in front of it
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < sizeof(RangeResult) / 4; i += kBoundsThreads) reinterpret_cast<uint32_t*>(&b.results[qi])[i] = 0u;
        if (b.score_pub && threadIdx.x < kScorePubRanges) b.score_pub[(size_t)qi * kScorePubRanges + threadIdx.x] = 0u;
    }
    const RangeQuery q = b.queries[qi];
    const uint32_t ns = q.seg_end - q.seg_begin;
    const uint64_t entries = (uint64_t)ns * (q.n_ranges + 1u);
    const uint32_t sub = threadIdx.x & (kBoundsLanes - 1u);
    const uint32_t group_shift = (threadIdx.x & 63u) & ~(kBoundsLanes - 1u);  // first lane of this group inside its wave
    const uint64_t e = (uint64_t)blockIdx.x * (kBoundsThreads / kBoundsLanes) + threadIdx.x / kBoundsLanes;
    const bool live = e < entries;  // (whole groups: the ballots below are executed by every lane of the wave)
    const uint32_t r = live ? (uint32_t)(e / ns) : 0u, i = live ? (uint32_t)(e - (uint64_t)r * ns) : 0u;
    const RangeSeg* sg = b.segs + q.seg_begin + i;
    const uint32_t len = live ? sg->len : 0u;
    uint32_t lo = 0, hi = len;
    if (r >= q.n_ranges) lo = len, hi = len;
    if (r == 0) hi = 0;
    const uint32_t target = r * q.width;  // first document of range r (n_docs < 2^32)
    const uint32_t* pd = b.post_doc + (live ? sg->post_begin : 0ull);
    if constexpr (kBoundsLanes == 1) {  // one lane per entry: binary search
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pd[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (live) b.bounds[q.bounds_base + e] = lo;
        return;
    }
    // the answer lies in [lo, hi]; every round cuts the window into 16 pieces (wave-uniform loop: groups that are done idle)
    for (;;) {
        const uint32_t span = hi - lo;
        if (__ballot(span > 0) == 0ull) break;
        const uint32_t step = (span + kBoundsLanes - 1u) / kBoundsLanes;  // >= 1 while span > 0
        const uint32_t idx = lo + (sub + 1u) * step - 1u;                 // the last element of this lane's piece
        const bool below = span > 0 && idx < hi && pd[idx] < target;      // ... lies before the boundary: so does the whole piece
        const uint32_t m = (uint32_t)(__ballot(below) >> group_shift) & ((1u << kBoundsLanes) - 1u);
        const uint32_t c = (uint32_t)__popc(m);  // pieces entirely below (the probes are monotone: a prefix of the lanes)
        if (span > 0) {
            const uint32_t nlo = lo + c * step;
            // the boundary lies in piece c: [nlo, nlo + step - 1] — its last element is not below, so the answer is <= that index
            hi = c == kBoundsLanes ? hi : min(hi, nlo + step - 1u);
            lo = min(nlo, hi);
        }
    }
    if (live && sub == 0) b.bounds[q.bounds_base + e] = lo;
}

// ---------------------------------------------------------------------------------------------- the scoring launch
// The launch is bound by VALU ISSUE (profiles/r04_k3r_sq_counters_v*.md: SQ_ACTIVE_INST_VALU = the whole launch on every
// SIMD, one wave instruction per 4 cycles), so the kernel is written for few vector instructions per posting:
//   * 4 fat waves per workgroup (a lane carries up to 8 postings through the phases in registers): per-wave fixed work —
//     tables, scans, reductions — is paid once per 320 postings instead of once per 160;
//   * wave 0 alone loads the range's tables, scans the run lengths and builds a 32-element block -> run table (the other
//     waves clear the bitmap and the presence masks meanwhile): a posting finds its run with one table read instead of a
//     binary search of its own;
//   * the rarely taken paths (3 % of the documents of a 12-token query have more than one posting) are kept out of the
//     per-posting code: a multi-posting document's postings only park their value in a cell; the documents themselves are
//     listed densely and folded + reported by as many lanes as there are such documents;
//   * reductions through LDS atomics without return (one instruction) instead of cross-lane shuffles (~30).
constexpr int kThreads = (int)kRangeThreads;
constexpr int kWaves = kThreads / 64;
static_assert(kRangeCap / kThreads == 8, "a lane carries at most 8 postings through the phases (score_body<.., 8>)");
constexpr uint32_t kBitWords = kRangeMaxWidth / 32;        // bitmap words of the widest range
constexpr int kWordsPerThread = kBitWords / kThreads;
constexpr uint32_t kBlkShift = kThreads == 256 ? 5 : 6, kBlocks = kRangeCap >> kBlkShift;  // run lookup table: one entry per 32 (64) gathered postings
constexpr uint32_t kCells = 4 * kThreads;                  // cells of the multi-posting documents of one range ...
constexpr uint32_t kMultiMax = 2 * kThreads;               // ... and how many such documents: more raise `overflow` (narrower ranges)
static_assert(kRangeCap % kThreads == 0 && kBitWords % kThreads == 0, "phase loops are unrolled over whole threads");
static_assert(kRangeCap <= 0x8000u && kRangeMaxWidth <= 0x10000u, "posting slot and local document share one 32-bit word");
static_assert(kBlocks <= 64, "the block table is built by one wave");

template <bool WIDE>
struct MaskOf {
    typedef uint32_t type;
};
template <>
struct MaskOf<true> {
    typedef unsigned long long type;
};
__device__ __forceinline__ uint32_t mask_popc(uint32_t m) { return (uint32_t)__popc(m); }
__device__ __forceinline__ uint32_t mask_popc(unsigned long long m) { return (uint32_t)__popcll(m); }
__device__ __forceinline__ uint32_t mask_first(uint32_t m) { return (uint32_t)__ffs((int)m) - 1u; }
__device__ __forceinline__ uint32_t mask_first(unsigned long long m) { return (uint32_t)__ffsll((long long)m) - 1u; }

// LDS of one scoring workgroup.  WIDE: queries of more than 32 references (64-bit presence masks).  Kept under 20 KB for the
// common instantiation: the kernel's waves spend most of their cycles waiting (dependent LDS and global loads, barriers),
// so the workgroups resident per CU set its speed (profiles/r04_k3r_sq_counters_v2.md: at 39.6 KB — 16 waves per CU — the
// launch took as long as with twice the vector instructions).
template <bool WIDE>
struct ScoreLds {
    typedef typename MaskOf<WIDE>::type mask_t;
    // region A, phases 1-3: the range's document bitmap (one bit per document) + the exclusive popcount prefix of its
    // words (u16: at most 2 048 documents are touched).
    // Phases 4-6: the multi-posting documents — their cells (one f32 per posting) and per document its first cell, its
    // presence mask and the (slot, local document) of the posting that lends it its key slot.
    static constexpr uint32_t kRegionWords = kCells + kMultiMax * 2 + kMultiMax / 2 + (WIDE ? kMultiMax : 0);
    uint32_t region_a[kRegionWords];
    mask_t dmask[kRangeCap];                    // per touched document (by rank): REFERENCES (lists) that hold it; after phase 4: first cell
    unsigned long long seg_pos[kRangeMaxRefs];  // first posting of each reference inside this range
    uint32_t seg_off[kRangeMaxRefs + 1];        // start of each reference's run among the gathered postings
    uint32_t seg_key[kRangeMaxRefs];            // token << 10 | rank
    uint32_t seg_pkb[kRangeMaxRefs];            // the constant part of a kept posting's packed word: kept | token << 25 | reference << 17
    float seg_boost[kRangeMaxRefs], seg_avg[kRangeMaxRefs];
    uint16_t blk_run[kBlocks];                  // run that holds gathered posting 32 i
    float idf[kMaxTokens];
    uint32_t df_lds[kMaxTokens];
    uint32_t wave_tot[kWaves];
    uint32_t red[4];                            // slot base, count, max key, ~min key
    uint32_t cell_cursor;                       // cells handed out | multi documents << 16
    uint32_t pub_floor;                         // compact key lists: the floor the query's published scores gave when the workgroup started
    __device__ __forceinline__ uint32_t* bitmap() { return region_a; }
    __device__ __forceinline__ uint16_t* word_rank() { return reinterpret_cast<uint16_t*>(region_a + kBitWords); }
    __device__ __forceinline__ float* cellv() { return reinterpret_cast<float*>(region_a); }
    __device__ __forceinline__ uint32_t* md_own() { return region_a + kCells; }
    __device__ __forceinline__ mask_t* md_mask() { return reinterpret_cast<mask_t*>(region_a + kCells + kMultiMax); }
    __device__ __forceinline__ uint16_t* md_cb() { return reinterpret_cast<uint16_t*>(region_a + kCells + kMultiMax + kMultiMax * (WIDE ? 2 : 1)); }
};
static_assert((kCells + kMultiMax * 2 + kMultiMax / 2) * 4 >= kBitWords * 4 + kBitWords * 2, "bitmap + prefix fit region A");
static_assert(kRangeMaxRefs <= 64, "one bit per reference in the presence masks");

// What phases 1-6 need to know about their workgroup (all workgroup-uniform).
struct ScoreRange {
    uint32_t qi, cap, slot_base, doc0, n_words;
    bool count_each;
    bool publish = false;   // compact key lists: this range publishes its score word (ScoreRange::pub_slot) for the query's other ranges
    uint32_t pub_slot = 0;
};

// Phases 1-6 for a workgroup whose lanes carry NITER postings each (NITER * 256 >= cap): compiled per NITER so that the
// per-posting loops are straight-line code — a run-time round count inside one body made the compiler shuffle the whole
// register arrays at every round's branch (a third of the vector instructions of the first form of this kernel).  Rounds
// past the end (e >= cap) read the last posting again and are not `kept`.
// PLAIN: a batch without score maps, OMC multipliers and min / max tracking whose store holds the pre-divided tf (every plain
// top-k search at the default b, filtered or not): those branches — per posting, per round — are not compiled in.
template <bool DF_ONLY, bool WIDE, int NITER, bool PLAIN = false, bool COMPACT = false>
__device__ __forceinline__ void score_body(const RangeBatch& b, const RangeQuery& q, const ScoreRange& rg, ScoreLds<WIDE>& L) {
    typedef typename MaskOf<WIDE>::type mask_t;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t cap = rg.cap, qi = rg.qi, doc0 = rg.doc0, n_words = rg.n_words;
    uint32_t* const bitmap = L.bitmap();
    uint16_t* const word_rank = L.word_rank();
    float* const cellv = L.cellv();
    uint32_t* const md_own = L.md_own();
    mask_t* const md_mask = L.md_mask();
    uint16_t* const md_cb = L.md_cb();

    // ---- 1. gather: posting e belongs to the run whose [seg_off[i], seg_off[i+1]) holds it
    uint32_t pk[NITER];  // [kept:1 | token:6 | run:8 | pad:1 | local document:16]
    float pv[NITER];     // normalised tf (boost included)
    const float one_minus_b = 1.0f - b.b;
    // (rounds in groups of four: eight rounds of addresses, documents and values alive at once cost the registers of a
    // resident wave per SIMD)
#pragma unroll
    for (int g = 0; g < NITER; g += 4) {
        constexpr int G = 4;
        unsigned long long pos[G];
        uint32_t run[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int n = g + j;
            if (n >= NITER) break;
            if (!rg.count_each) L.dmask[threadIdx.x + n * kThreads] = (mask_t)0;  // (touched documents <= postings; first used in phase 3)
            const uint32_t e = min(threadIdx.x + n * kThreads, cap - 1u);
            uint32_t lo = L.blk_run[e >> kBlkShift];
            while (L.seg_off[lo + 1] <= e) ++lo;  // (runs are ~100 postings: almost always zero steps)
            run[j] = lo;
            pos[j] = L.seg_pos[lo] + (e - L.seg_off[lo]);
        }
        uint32_t doc[G], val[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (g + j >= NITER) break;
            doc[j] = b.post_doc[pos[j]];
            val[j] = 0u;
            if (!DF_ONLY) val[j] = (PLAIN || b.post_ntf) ? __builtin_bit_cast(uint32_t, b.post_ntf[pos[j]]) : b.post_val[pos[j]];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int n = g + j;
            if (n >= NITER) break;
            bool kept = threadIdx.x + n * kThreads < cap;
            if (b.allow) {  // collect_contributions_with_filter: filtered docs never reach the scorer
                const uint64_t id = b.docs ? b.docs[doc[j]] : b.dense_base + doc[j];  // dense ids: no table lookup
                kept = kept && id < b.allow_bits && ((b.allow[id >> 6] >> (id & 63)) & 1ull);
            }
            const uint32_t dl = doc[j] - doc0;
            pk[n] = kept ? (L.seg_pkb[run[j]] | dl) : 0u;
            pv[n] = 0.0f;
            if (rg.count_each) {
                if (kept) atomicAdd(&L.df_lds[(pk[n] >> 25) & 63u], 1u);
            } else {
                if (kept) atomicOr(&bitmap[dl >> 5], 1u << (dl & 31u));
                if (!DF_ONLY) {
                    const float pre = (PLAIN || b.post_ntf) ? __builtin_bit_cast(float, val[j]) : ntf_pre_of(val[j], one_minus_b, b.b, L.seg_avg[run[j]]);
                    pv[n] = L.seg_boost[run[j]] * pre;
                }
            }
        }
    }
    __syncthreads();
    if (rg.count_each) {
        if (threadIdx.x < q.n_tokens && L.df_lds[threadIdx.x]) atomicAdd(&b.results[qi].df[threadIdx.x], L.df_lds[threadIdx.x]);
        return;
    }

    // ---- 2. rank of every touched document = set bits before its own: exclusive popcount prefix over the bitmap words
    {
        uint32_t cnt[kWordsPerThread], sum = 0;
#pragma unroll
        for (int n = 0; n < kWordsPerThread; ++n) {
            const uint32_t w = threadIdx.x * kWordsPerThread + n;  // consecutive words per thread: one scan
            cnt[n] = w < n_words ? (uint32_t)__popc(bitmap[w]) : 0u;
            sum += cnt[n];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += y;
        }
        if (lane == 63) L.wave_tot[wave] = incl;
        __syncthreads();
        uint32_t excl = incl - sum;
#pragma unroll
        for (int w = 0; w < kWaves - 1; ++w) excl += (uint32_t)w < wave ? L.wave_tot[w] : 0u;
#pragma unroll
        for (int n = 0; n < kWordsPerThread; ++n) {
            const uint32_t w = threadIdx.x * kWordsPerThread + n;
            if (w < n_words) word_rank[w] = (uint16_t)excl;
            excl += cnt[n];
        }
    }
    __syncthreads();
    uint32_t n_touched = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) n_touched += L.wave_tot[w];

    // ---- 3. every kept posting: rank of its document, its REFERENCE (list) into the document's presence mask.  One bit per
    // list, not per token: a document is then at most once behind each bit, every posting owns its cell, and the cells of a
    // document in bit order are its contributions in (token, reference) order — queries with several lists per token
    // (several fields, prefix / typo expansions) need no special path.
    uint32_t prk[(NITER + 1) / 2] = {0};  // the ranks, two per register (a rank is < 2 048)
#define PRANK(n) ((prk[(n) >> 1] >> (16 * ((n) & 1))) & 0xffffu)
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        const uint32_t dl = pk[n] & 0xffffu;
        const uint32_t rank = (uint32_t)word_rank[dl >> 5] + (uint32_t)__popc(bitmap[dl >> 5] & ((1u << (dl & 31u)) - 1u));
        prk[n >> 1] |= rank << (16 * (n & 1));
        if (pk[n] >> 31) atomicOr(&L.dmask[rank], (mask_t)1 << ((pk[n] >> 17) & 63u));
    }
    __syncthreads();
    mask_t pm[NITER];  // the document's presence mask (0: dropped posting)
#pragma unroll
    for (int n = 0; n < NITER; ++n) pm[n] = (pk[n] >> 31) ? L.dmask[PRANK(n)] : (mask_t)0;
    if (DF_ONLY) {
        // corpus_docs.len(): distinct (token, document) pairs among the kept postings (token_score.rs:262-275): a posting
        // counts when no EARLIER list of its token holds the document (the lists of a token are consecutive references)
#pragma unroll
        for (int n = 0; n < NITER; ++n) {
            if (!(pk[n] >> 31)) continue;
            const uint32_t run = (pk[n] >> 17) & 63u, first = run - (L.seg_key[run] & 1023u);
            const mask_t earlier = (((mask_t)1 << run) - 1) & ~(((mask_t)1 << first) - 1);
            if ((pm[n] & earlier) == 0) atomicAdd(&L.df_lds[(pk[n] >> 25) & 63u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < q.n_tokens && L.df_lds[threadIdx.x]) atomicAdd(&b.results[qi].df[threadIdx.x], L.df_lds[threadIdx.x]);
        return;
    }

    // ---- 4. cells for the documents with more than one posting: one cell per posting and a place in the dense list of such
    // documents, both handed out by ONE returning LDS atomic (cells in the low half, list position in the high half) to the
    // posting of the document's FIRST reference — there is exactly one — which also lends the document its key slot.
    // Every posting has read its document's mask (above); the mask's slot then takes the first cell for the others.
    __syncthreads();  // (the bitmap and its prefix are dead from here on: region A holds the multi-document tables)
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        const mask_t m = pm[n];
        const mask_t below = m & (((mask_t)1 << ((pk[n] >> 17) & 63u)) - 1);
        if ((m & (m - 1)) != 0 && below == 0) {
            const uint32_t cells = mask_popc(m);
            const uint32_t got = atomicAdd(&L.cell_cursor, cells | (1u << 16));
            const uint32_t cb = got & 0xffffu, mi = got >> 16;
            if (cb + cells <= kCells && mi < kMultiMax) {  // (beyond: the range overflows, nothing of it is used)
                L.dmask[PRANK(n)] = (mask_t)cb;
                md_mask[mi] = m;
                md_cb[mi] = (uint16_t)cb;
                md_own[mi] = (threadIdx.x + n * kThreads) | ((pk[n] & 0xffffu) << 16);
            }
        }
    }
    __syncthreads();
    const uint32_t n_cells = L.cell_cursor & 0xffffu, n_multi = L.cell_cursor >> 16;
    if (n_cells > kCells || n_multi > kMultiMax) {
        // more multi-posting documents than the tables hold (terms that occur together in most of their documents): like a
        // range of too many postings, the query is rerun with narrower ranges; its slots must be empty meanwhile
        if (!COMPACT)  // (a compact list has no slots to clear: nothing of this range is appended)
            for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
                b.keys[q.key_off + rg.slot_base + e] = 0ull;
                if (b.map_idx) b.map_idx[rg.slot_base + e] = 0xffffffffu;
            }
        if (threadIdx.x == 0) {
            b.results[qi].overflow = 1;
            // how much too much, in 1/16: what the host narrows the ranges by (the tables take kMultiMax documents / kCells cells)
            atomicMax(&b.results[qi].pad1[0], max((n_multi * 16u + kMultiMax - 1u) / kMultiMax, (n_cells * 16u + kCells - 1u) / kCells));
        }
        return;
    }

    const float k1 = q.k + 1.0f;
    unsigned long long* out = b.keys + q.key_off + rg.slot_base;
    uint32_t my_count = 0, my_max = 0u, my_min_inv = 0u;
    // the document of slot e is final: its key (0 = not in the map or NaN), its map entry in score-map mode
    auto report = [&](uint32_t e, uint32_t dl, float score, bool in_map) {
        unsigned long long out_key = 0ull;
        uint32_t map_doc = 0xffffffffu;  // the document whose map entry this slot holds (score-map mode)
        float map_score = 0.0f;
        if (in_map) {
            const uint32_t doc = doc0 + dl;
            if constexpr (!PLAIN) {
                if (q.track_minmax && score == score) {  // hybrid: min / max of the full-text scores (before any OMC)
                    const uint32_t ord = f32_to_ordered(score);
                    my_max = max(my_max, ord);
                    my_min_inv = max(my_min_inv, ~ord);
                }
                if (b.omc_dense) score = score * b.omc_dense[doc];
            }
            ++my_count;
            map_doc = doc;
            map_score = score;
            if (score == score)  // a NaN score stays in the map (count) and is never selected
                out_key = ((unsigned long long)f32_to_ordered(score) << 32) | (unsigned long long)(~doc);
        }
        out[e] = out_key;
        if constexpr (PLAIN) {
            (void)map_doc;
            (void)map_score;
        } else if (b.map_idx) {
            // score-map mode (a batch of ONE query): slot = position of the entry in the map's candidate list, the
            // per-document table points back at it (ScoreMapDev, facets.hip) — NaN scores included, they count
            const uint32_t pos = rg.slot_base + e;
            b.map_idx[pos] = map_doc;
            if (map_doc != 0xffffffffu) {
                b.map_score[pos] = map_score;
                b.map_emit[map_doc] = ((unsigned long long)b.map_epoch << 32) | pos;
            }
        }
    };

    // ---- 5. a singleton is scored and reported by its posting's lane; a posting of any other document parks its normalised
    // tf in its own cell and leaves its slot empty — except the lender's, which phase 6 writes
    if constexpr (PLAIN && COMPACT) {
        // COMPACT key list (round 5).  Round 4 wrote one 8-byte slot per posting — empty or not — and the top-k read them all
        // back: 2.2 x the algorithmic bytes over the chain (profiles/r04_pmc_k3r_*.json), the top-k bound by READING slots.  Here
        // a lane keeps the score words of its postings in registers (the normalised tf's registers: dead by now), the workgroup
        // derives a FLOOR — the j-th best of each wave's 64 lane bests, j = ceil(topk / 4): at least topk of this range's
        // documents reach the smallest of the four, so the query's topk-th best does too — raises it to what earlier
        // workgroups of the query published, appends the keys at or above it behind ONE bump of the query's cursor, and
        // publishes its own.  A floor never cuts the answer (>= topk documents are at or above it; ties on the score word
        // pass), `count` comes from the presence masks as before: same answers, bit for bit.
        uint32_t ko[NITER];
#pragma unroll
        for (int n = 0; n < NITER; ++n) {
            const uint32_t e = threadIdx.x + n * kThreads;
            const mask_t m = pm[n];
            const bool single = m != 0 && (m & (m - 1)) == 0;
            const float sum = 0.0f + 1.0f * pv[n];                                      // Iterator::sum() from 0.0, weight 1.0
            const float term = L.idf[(pk[n] >> 25) & 63u] * k1 * sum / (q.k + sum);      // bm25f_score, bm25.rs:124-126
            const bool applied = f32_is_normal(sum) && term == term;
            const float score = 0.0f + term * 1.0f;                                     // entry(key).or_insert(0.0) += term * boost 1.0
            const bool in_map = single && applied && !(q.use_threshold && 1u < q.threshold);
            if (m != 0 && !single) {
                const uint32_t below = mask_popc((mask_t)(m & (((mask_t)1 << ((pk[n] >> 17) & 63u)) - 1)));
                cellv[(uint32_t)L.dmask[PRANK(n)] + below] = pv[n];
            }
            my_count += in_map ? 1u : 0u;
            ko[n] = (in_map && e < cap) ? f32_to_ordered(score) : 0u;
        }
        uint32_t best = ko[0];
#pragma unroll
        for (int n = 1; n < NITER; ++n) best = max(best, ko[n]);
        const uint32_t kq = q.topk, jshare = (kq + kWaves - 1u) / kWaves;
        uint32_t wb = 0u;
        if (kq != 0u && jshare <= 64u) {
            // the jshare-th largest of the wave's lane bests, to its leading 24 bits (15 mantissa bits: a LOWER bound of it, 0.003 %
            // below at most) in 24 dependent ballot steps — one compare each, the counting is scalar: the launch is bound by
            // latency, and this chain sits on every workgroup's critical path (0: fewer than jshare lanes hold a key).
            // Tried and dropped (profiles/r05_k3r_compact_ab_v*.log): whole keys — score word, then document — as floors (ties
            // are not what lets keys through: same survivors, +0.7 us per query); a second chain for the 2 jshare-th largest, two
            // waves vouching for all topk (the waves of a range see different lists: 9 -> 6 % survivors, +0.2 us per query, the
            // top-k no faster).
#pragma unroll
            for (int bit = 31; bit >= 8; --bit) {
                const uint32_t t = wb | (1u << bit);
                wb = (uint32_t)__popcll(__ballot(best >= t)) >= jshare ? t : wb;
            }
        }
        if (lane == 0) L.wave_tot[wave] = wb;  // (the prefix sums of phase 2 are consumed)
        if (rg.publish) {
            // (workgroup-uniform: one of the query's first ranges) what this range publishes for the others' floor: the 4th
            // largest lane best of its best wave — 4 documents at or above it (the smallest of the four wave MAXIMA was
            // published first: the wave that saw no rare term drags it under the local floors, profiles/r05_k3r_compact_ab_v8.log)
            uint32_t w4 = 0u;
#pragma unroll
            for (int bit = 31; bit >= 8; --bit) {
                const uint32_t t = w4 | (1u << bit);
                w4 = (uint32_t)__popcll(__ballot(best >= t)) >= 4u ? t : w4;
            }
            if (lane == 0) L.df_lds[wave] = w4;  // (df_lds: the counting modes' table, unused here)
        }
        if (n_cells != 0u) {  // (workgroup-uniform) ---- 6. the multi-posting documents: folded as below, their score word parked in LDS
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n_multi; i += kThreads) {
                mask_t m = md_mask[i];
                DocFold f;
                for (uint32_t c = md_cb[i]; m != 0; ++c) {
                    const uint32_t ref = mask_first(m);
                    m &= m - 1;
                    f.add(L.seg_key[ref] >> 10, cellv[c], L.idf, q.k, k1);
                }
                const bool in_map = f.finish(L.idf, q.k, k1, q.use_threshold, q.threshold);
                my_count += in_map ? 1u : 0u;
                md_mask[i] = (mask_t)((in_map && f.score == f.score) ? f32_to_ordered(f.score) : 0u);
            }
        }
        if (my_count) atomicAdd(&L.red[1], my_count);
        __syncthreads();
        uint32_t lb = L.wave_tot[0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) lb = min(lb, L.wave_tot[w]);
        // (L.pub_floor: what the query's published scores gave when this workgroup started — range_score_main, wave 1)
        const uint32_t floor_w = kq == 0u ? 0xffffffffu : max(max(lb, L.pub_floor), 1u);
        constexpr int kMultiRounds = kMultiMax / kThreads;
        // (the multi-posting documents' score words are read from LDS where they are used — counted here, written below —
        // instead of being carried in registers across the barriers: the kernel's 64-register budget, eight workgroups per CU)
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
            sm[NITER + t] = (uint32_t)t * kThreads < n_multi ? __ballot(i < n_multi && (uint32_t)md_mask[i] >= floor_w) : 0ull;
            tot += (uint32_t)__popcll(sm[NITER + t]);
        }
        // one cursor bump per WAVE, no barrier behind the floor's: a workgroup-wide bump (LDS offsets, barrier, one global atomic,
        // barrier) put two barriers and a returning global atomic in a row on every workgroup's critical path — 3.5 -> 4.3 us per
        // query (profiles/r05_k3r_compact_ab_v2.log); the cursors of different queries live 1 KB apart, ~1 500 bumps each
        uint32_t at = 0u;
        if (lane == 0 && tot) at = atomicAdd(&b.results[qi].n_keys, tot);
        at = __shfl(at, 0, 64);
        if (threadIdx.x == 0) {
            if (L.red[1]) atomicAdd(&b.results[qi].count, L.red[1]);
            if (b.debug & 16u) {  // ORAMA_K3R_DBG=16 (with ORAMA_K3R_STATS=1): how many workgroups found a published floor, the largest one, the largest local one
                if (L.pub_floor) atomicAdd(&b.results[qi].pad0[0], 1u);
                if (L.pub_floor > lb) atomicAdd(&b.results[qi].pad0[1], 1u);
                atomicMax(&b.results[qi].score_floor, L.pub_floor);
                atomicMax(&b.results[qi].pad0[2], lb);
            }
        }
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
                    ((unsigned long long)(uint32_t)md_mask[i] << 32) | (unsigned long long)(~(doc0 + (md_own[i] >> 16)));
            }
            at += (uint32_t)__popcll(sm[NITER + t]);
        }
        if (rg.publish && wave == 0) {
            // what this range publishes for the floor of the query's later ranges: a score word 4 of its documents reach — the
            // larger of (the 4th largest lane best of its best wave: single-term documents) and (the 4th largest score among its
            // documents that hold SEVERAL of the query's terms, parked in LDS by phase 6: the query's k-th best lives among those,
            // a floor made of single-term keys alone let 3-6 % of the postings through).  One atomicMax per publisher and address,
            // performed at the memory side: visible to the other XCDs' agent-scope loads.
            uint32_t v = L.df_lds[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) v = max(v, L.df_lds[w]);
            if (n_multi >= 4u) {
                uint32_t mbest = 0u;  // this lane's best of the multi-term scores lane, lane + 64, ...
                for (uint32_t i = lane; i < n_multi; i += 64u) mbest = max(mbest, (uint32_t)md_mask[i]);
                uint32_t m4 = 0u;
#pragma unroll
                for (int bit = 31; bit >= 8; --bit) {
                    const uint32_t t = m4 | (1u << bit);
                    m4 = (uint32_t)__popcll(__ballot(mbest >= t)) >= 4u ? t : m4;
                }
                v = max(v, m4);
            }
            if (lane == 0 && v) atomicMax(&b.score_pub[(size_t)qi * kScorePubRanges + rg.pub_slot], v);
        }
        return;
    }
    if constexpr (PLAIN) {
        // Straight-line form of the loop below for the plain search: every lane evaluates the singleton's fold (the additions
        // DocFold::add + finish perform for ONE contribution, in their order) and selects; only the rare posting of a document
        // with several postings branches.  The general loop nests four divergent branches per posting: the scalar unit — exec
        // masks, branches — was as busy as the vector units (profiles/r04_k3r_sq_counters_v5.md).
#pragma unroll
        for (int n = 0; n < NITER; ++n) {
            const uint32_t e = threadIdx.x + n * kThreads;
            const uint32_t dl = pk[n] & 0xffffu;
            const mask_t m = pm[n];
            const bool single = m != 0 && (m & (m - 1)) == 0;
            const float sum = 0.0f + 1.0f * pv[n];                                      // Iterator::sum() from 0.0, weight 1.0
            const float term = L.idf[(pk[n] >> 25) & 63u] * k1 * sum / (q.k + sum);      // bm25f_score, bm25.rs:124-126
            const bool applied = f32_is_normal(sum) && term == term;
            const float score = 0.0f + term * 1.0f;                                     // entry(key).or_insert(0.0) += term * boost 1.0
            const bool in_map = single && applied && !(q.use_threshold && 1u < q.threshold);  // (one token: popcount(mask) = 1)
            bool lends = false;
            if (m != 0 && !single) {
                const uint32_t below = mask_popc((mask_t)(m & (((mask_t)1 << ((pk[n] >> 17) & 63u)) - 1)));
                lends = below == 0;  // the posting of the document's first reference: it asked for the cells in phase 4
                cellv[(uint32_t)L.dmask[PRANK(n)] + below] = pv[n];
            }
            const uint32_t doc = doc0 + dl;
            const unsigned long long key = in_map ? ((unsigned long long)f32_to_ordered(score) << 32) | (unsigned long long)(~doc) : 0ull;
            my_count += in_map ? 1u : 0u;
            if (e < cap && !lends) out[e] = key;
        }
    } else {
#pragma unroll
    for (int n = 0; n < NITER; ++n) {
        const uint32_t e = threadIdx.x + n * kThreads;
        const uint32_t dl = pk[n] & 0xffffu;
        const mask_t m = pm[n];
        float score = 0.0f;
        bool in_map = false, lends = false;
        if ((m & (m - 1)) == 0) {
            if (m != 0) {  // (a dropped posting reports an empty slot)
                DocFold f;
                f.add((pk[n] >> 25) & 63u, pv[n], L.idf, q.k, k1);
                in_map = f.finish(L.idf, q.k, k1, q.use_threshold, q.threshold);
                score = f.score;
            }
        } else {
            const uint32_t below = mask_popc((mask_t)(m & (((mask_t)1 << ((pk[n] >> 17) & 63u)) - 1)));
            lends = below == 0;  // the posting of the document's first reference: it asked for the cells in phase 4
            cellv[(uint32_t)L.dmask[PRANK(n)] + below] = pv[n];
        }
        if (e < cap && !lends) report(e, dl, score, in_map);
    }
    }
    if (n_cells != 0u) {  // (workgroup-uniform)
        __syncthreads();
        // ---- 6. one lane per multi-posting document: its cells in bit order are its contributions in (token, reference)
        // order — folded exactly as BM25Scorer::add / get_scores would, reported in the slot lent to it
        for (uint32_t i = threadIdx.x; i < n_multi; i += kThreads) {
            mask_t m = md_mask[i];
            DocFold f;
            for (uint32_t c = md_cb[i]; m != 0; ++c) {
                const uint32_t ref = mask_first(m);
                m &= m - 1;
                f.add(L.seg_key[ref] >> 10, cellv[c], L.idf, q.k, k1);
            }
            const bool in_map = f.finish(L.idf, q.k, k1, q.use_threshold, q.threshold);
            const uint32_t o = md_own[i];
            report(o & 0xffffu, o >> 16, f.score, in_map);
        }
    }

    if (my_count) atomicAdd(&L.red[1], my_count);
    if constexpr (!PLAIN) {
        if (q.track_minmax) {
            if (my_max) atomicMax(&L.red[2], my_max);
            if (my_min_inv) atomicMax(&L.red[3], my_min_inv);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (L.red[1]) atomicAdd(&b.results[qi].count, L.red[1]);
        if (L.red[2]) atomicMax(&b.results[qi].max_key, L.red[2]);
        if (L.red[3]) atomicMax(&b.results[qi].min_inv, L.red[3]);
    }
#undef PRANK
}

// DF_ONLY: the counting pass (corpus_docs.len() per token under a filter / with several lists per token).
template <bool DF_ONLY, bool WIDE, bool PLAIN = false, bool COMPACT = false>
__device__ __forceinline__ void range_score_main(const RangeBatch& b) {
    __shared__ ScoreLds<WIDE> L;

    // (query, range) of this workgroup: the batch's pairs laid end to end
    uint32_t qi = 0, r = 0;
    if constexpr (COMPACT) {
        // compact key lists: the batch is scored in kRangeStripes passes over its queries (RangeBatch::stripe_start), so that a
        // query's ranges are scored THROUGHOUT the launch: its first ranges have published their scores by the time most of
        // the others start.  (Query after query, the ~400 workgroups of a query are dispatched together and none finds a
        // floor; range-major — workgroup w = range w / n of query w % n — ties the launch to the LARGEST query of the batch:
        // 3.5 -> 6.7 us per query on unsorted batches, profiles/r05_k3r_compact_ab_v5.log.)
        uint32_t lo = 0, hi = kRangeStripes * kRangeBatchMax;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (b.stripe_start[mid] <= blockIdx.x) lo = mid; else hi = mid;
        }
        // (entries of equal value = empty (stripe, query) cells: the LAST of them is the one that holds the workgroup)
        qi = lo % kRangeBatchMax;
        const uint32_t stripe = lo / kRangeBatchMax;
        r = (uint32_t)((uint64_t)stripe * b.queries[qi].n_ranges / kRangeStripes) + (blockIdx.x - b.stripe_start[lo]);
    } else {
        uint32_t lo = 0, hi = b.n_queries;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (b.range_start[mid] <= blockIdx.x) lo = mid; else hi = mid;
        }
        qi = lo;
        r = blockIdx.x - b.range_start[qi];
    }
    const RangeQuery q = b.queries[qi];
    if (r >= q.n_ranges) return;
    if (DF_ONLY && !q.want_df) return;
    const uint32_t ns = q.seg_end - q.seg_begin;
    const RangeSeg* segs = b.segs + q.seg_begin;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    ScoreRange rg;
    rg.qi = qi;
    rg.count_each = DF_ONLY && q.want_df == 2u;  // every token has ONE list: a kept posting is its own (token, document) pair
    rg.n_words = (min(q.width, kRangeMaxWidth) + 31u) >> 5;
    rg.doc0 = r * q.width;
    if constexpr (COMPACT) {
        // the publishers: the query's first kScorePubRanges ranges — the ones the range-major grid scores first
        rg.publish = r < kScorePubRanges;
        rg.pub_slot = r;
    }

    // ---- 0. wave 0: the range's run of every reference, their offsets among the gathered postings, the block table;
    //         the other waves: clear the bitmap
    if (wave == 0) {
        if (lane < 4) L.red[lane] = 0;
        if (lane == 0) {
            L.cell_cursor = 0u;
            L.seg_off[0] = 0;
        }
        L.df_lds[lane] = 0;
        static_assert(kMaxTokens == 64, "one lane per token");
        if (!DF_ONLY) L.idf[lane] = lane < q.n_tokens ? b.idf[(size_t)qi * kMaxTokens + lane] : 0.0f;
        // (the bounds of a range are contiguous over the references, and their address does not depend on the reference
        // table: both loads are issued together); slot base = postings of the query in earlier ranges
        uint32_t carry = 0, base_part = 0;
        const uint32_t* row = b.bounds + q.bounds_base + (uint64_t)r * ns;
        for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint32_t x = 0;
            if (i < ns) {
                const uint32_t b0 = row[i], b1 = row[ns + i];
                const RangeSeg sg = segs[i];
                L.seg_pos[i] = sg.post_begin + b0;
                L.seg_key[i] = sg.tok_rank;
                L.seg_pkb[i] = 0x80000000u | ((sg.tok_rank >> 10) << 25) | (i << 17);
                L.seg_boost[i] = sg.boost;
                L.seg_avg[i] = sg.avg_len;
                base_part += b0;
                x = b1 - b0;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {  // inclusive scan of the run lengths, 64 references at a time
                const uint32_t y = __shfl_up(x, off, 64);
                if ((int)lane >= off) x += y;
            }
            if (i < ns) L.seg_off[i + 1] = carry + x;
            carry += __shfl(x, 63, 64);
        }
        if (base_part) atomicAdd(&L.red[0], base_part);
        // block table: the run that holds gathered posting 32 * lane (the wave's own LDS writes above are visible to it:
        // LDS operations of one wave complete in order)
        if (carry <= kRangeCap && (lane << kBlkShift) < carry) {
            const uint32_t e = lane << kBlkShift;
            uint32_t lo = 0, hi = ns;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (L.seg_off[mid] <= e) lo = mid; else hi = mid;
            }
            L.blk_run[lane] = (uint16_t)lo;
        }
    } else if (!rg.count_each) {
        for (uint32_t w = threadIdx.x - 64u; w < rg.n_words; w += kThreads - 64u) L.bitmap()[w] = 0u;
        if constexpr (COMPACT) {
            if (wave == 1) {
                // the floor the query's first ranges have published so far: the ceil(topk / 4)-th largest of their words (each
                // vouches for 4 documents at or above its word; unpublished = 0), to its leading 24 bits; 0: not enough yet
                const uint32_t v = __hip_atomic_load(&b.score_pub[(size_t)qi * kScorePubRanges + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t need = (q.topk + 3u) / 4u;
                uint32_t fl = 0u;
                if (q.topk != 0u && need <= kScorePubRanges) {
#pragma unroll
                    for (int bit = 31; bit >= 8; --bit) {
                        const uint32_t t = fl | (1u << bit);
                        fl = (uint32_t)__popcll(__ballot(v >= t)) >= need ? t : fl;
                    }
                }
                if (lane == 0) L.pub_floor = fl;
            }
        }
    }
    __syncthreads();
    const uint32_t cap = L.seg_off[ns];
    if (cap == 0) return;
    rg.cap = cap;
    rg.slot_base = L.red[0];
    if (cap > kRangeCap) {
        // the query is rerun with smaller ranges; its slots still reach the batch's top-k, so they must be empty
        if (!DF_ONLY && !COMPACT)
            for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
                b.keys[q.key_off + rg.slot_base + e] = 0ull;
                if (b.map_idx) b.map_idx[rg.slot_base + e] = 0xffffffffu;
            }
        if (threadIdx.x == 0) {
            b.results[qi].overflow = 1;
            atomicMax(&b.results[qi].pad1[0], (cap * 16u + kRangeCap - 1u) / kRangeCap);  // (postings against the 2 048 a range takes, in 1/16)
        }
        return;
    }
    // rounds of the per-posting phases (workgroup-uniform): the bodies that exist are 2, 4, 5, 6, 7 and 8 rounds — a range of
    // the targeted ~1 536 postings takes 6 or 7 (a round count without a body of its own runs the next one: whole rounds of
    // clamped, unkept postings)
    const uint32_t n_iter = (cap + kThreads - 1) / kThreads;
    if (DF_ONLY) {
        if (n_iter <= 4) score_body<DF_ONLY, WIDE, 4, PLAIN, COMPACT>(b, q, rg, L);
        else score_body<DF_ONLY, WIDE, 8, PLAIN, COMPACT>(b, q, rg, L);
    } else if (n_iter <= 2) score_body<DF_ONLY, WIDE, 2, PLAIN, COMPACT>(b, q, rg, L);
    else if (n_iter <= 4) score_body<DF_ONLY, WIDE, 4, PLAIN, COMPACT>(b, q, rg, L);
    else if (n_iter == 5) score_body<DF_ONLY, WIDE, 5, PLAIN, COMPACT>(b, q, rg, L);
    else if (n_iter == 6) score_body<DF_ONLY, WIDE, 6, PLAIN, COMPACT>(b, q, rg, L);
    else if (n_iter == 7) score_body<DF_ONLY, WIDE, 7, PLAIN, COMPACT>(b, q, rg, L);
    else score_body<DF_ONLY, WIDE, 8, PLAIN, COMPACT>(b, q, rg, L);
}

template <bool DF_ONLY, bool WIDE, bool PLAIN = false>
__global__ __launch_bounds__(kThreads) void range_score_kernel(RangeBatch b) {
    range_score_main<DF_ONLY, WIDE, PLAIN, false>(b);
}
// The plain search with compact key lists.  Eight workgroups per CU are what makes this launch fast (it waits — dependent LDS
// and global loads, barriers — most of its cycles): the register allocator is held to the 64 VGPRs that buys.
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void range_score_compact_kernel(RangeBatch b) {
    range_score_main<false, false, true, true>(b);
}

// Hybrid path: the full-text score of given documents.  One WAVE per document: lane i looks the document up in the
// query's i-th reference (i + 64, ... for more than 64) — a binary search inside the window the bounds table gives for
// the document's range, a handful of postings — and parks the normalised tf in LDS; lane 0 then folds what was found in
// (token, reference) order with DocFold, the additions of the range kernel.  (One thread per document walking all its
// references took 46 us for 100 documents x 12 lists of ~50 K postings: 200 dependent loads in a row, after the scan,
// on the critical path of a hybrid query; this form takes one search's worth.)
constexpr int kDocsThreads = 256;
__global__ __launch_bounds__(kDocsThreads) void range_score_docs_kernel(RangeBatch b, uint32_t qi, const uint32_t* __restrict__ docs,
                                                                        uint32_t n, float* __restrict__ out_score,
                                                                        uint32_t* __restrict__ out_present,
                                                                        const uint32_t* __restrict__ n_dev) {
    constexpr uint32_t kDocWaves = kDocsThreads / 64;
    __shared__ float idf[kMaxTokens];
    __shared__ float found_ntf[kDocWaves][kRangeMaxRefs];
    __shared__ unsigned long long found_mask[kDocWaves][kRangeMaxRefs / 64];
    const RangeQuery q = b.queries[qi];
    for (uint32_t t = threadIdx.x; t < kMaxTokens; t += kDocsThreads) idf[t] = t < q.n_tokens ? b.idf[(size_t)qi * kMaxTokens + t] : 0.0f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * kDocWaves + w;
    if (n_dev) n = min(n, *n_dev);  // (the device tail: the grid was sized for `limit`, the vector map holds *n_dev documents)
    if (j >= n) return;  // (wave-uniform; no block-wide barrier below)
    const uint32_t doc = docs[j];
    bool allowed = true;
    if (b.allow) {
        const uint64_t id = b.docs ? b.docs[doc] : b.dense_base + doc;
        allowed = id < b.allow_bits && ((b.allow[id >> 6] >> (id & 63)) & 1ull);
    }
    const float k1 = q.k + 1.0f, one_minus_b = 1.0f - b.b;
    const uint32_t ns = q.seg_end - q.seg_begin;
    const uint32_t r = doc / q.width;
    const uint32_t* bnd = b.bounds + q.bounds_base + (uint64_t)r * ns;
    for (uint32_t base = 0; base < ns; base += 64) {
        const uint32_t i = base + lane;
        bool hit = false;
        float ntf = 0.0f;
        if (allowed && i < ns && r < q.n_ranges) {
            const RangeSeg sg = b.segs[q.seg_begin + i];
            const uint32_t* pd = b.post_doc + sg.post_begin;
            uint32_t lo = bnd[i], hi = bnd[ns + i];  // the reference's postings inside the document's range
            const uint32_t end = hi;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pd[mid] < doc) lo = mid + 1; else hi = mid;
            }
            if (lo < end && pd[lo] == doc) {
                const float pre = b.post_ntf ? b.post_ntf[sg.post_begin + lo]
                                             : ntf_pre_of(b.post_val[sg.post_begin + lo], one_minus_b, b.b, sg.avg_len);
                ntf = sg.boost * pre;
                hit = true;
            }
        }
        if (i < kRangeMaxRefs) found_ntf[w][i] = ntf;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) found_mask[w][base >> 6] = m;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
    if (lane == 0) {
        DocFold f;
        for (uint32_t i = 0; i < ns; ++i)
            if ((found_mask[w][i >> 6] >> (i & 63)) & 1ull) f.add(b.segs[q.seg_begin + i].tok_rank >> 10, found_ntf[w][i], idf, q.k, k1);
        const bool present = f.finish(idf, q.k, k1, q.use_threshold, q.threshold);
        out_score[j] = f.score;
        out_present[j] = present ? 1u : 0u;
    }
}

// post_ntf of every posting: one thread per posting finds its list (upper-bound search in the offsets: the list table of
// a 2^20-term vocabulary stays in L2) and divides exactly as the scoring kernels would.
__global__ __launch_bounds__(256) void ntf_precompute_kernel(const uint32_t* __restrict__ post_val, float* __restrict__ post_ntf,
                                                             const uint64_t* __restrict__ list_off, const float* __restrict__ list_avg,
                                                             uint32_t n_lists, uint64_t n_postings, float b) {
    const float one_minus_b = 1.0f - b;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_postings; i += (uint64_t)gridDim.x * 256) {
        uint32_t lo = 0, hi = n_lists;  // the last list with list_off[l] <= i is the one that holds posting i (empty lists share an offset)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (list_off[mid] <= i) lo = mid; else hi = mid;
        }
        post_ntf[i] = ntf_pre_of(post_val[i], one_minus_b, b, list_avg[lo]);
    }
}

#if ORAMA_COMPARISON_KERNELS
// Dense-list accelerators (orama_post::d_acc, RangeSeg::acc_off; read by bm25_ranges_fast.hip only).  acc_bits_kernel: every posting of an accelerated list sets its
// document's bit (grid.y = the list).  acc_scan_kernel: one workgroup per list — the exclusive popcount prefix of the bitmap's
// words (a posting's position in its list = prefix of its word + set bits below it in the word) and the smallest / largest
// normalised tf of the list (NaN if any posting's is).
__global__ __launch_bounds__(256) void acc_bits_kernel(const uint32_t* __restrict__ post_doc, const uint64_t* __restrict__ list_off,
                                                       const uint32_t* __restrict__ acc_list, uint32_t acc_words, uint32_t* __restrict__ acc) {
    const uint32_t i = blockIdx.y, l = acc_list[i];
    const uint64_t begin = list_off[l], end = list_off[l + 1];
    uint32_t* bits = acc + (size_t)i * 2 * acc_words;
    for (uint64_t j = begin + (uint64_t)blockIdx.x * 256 + threadIdx.x; j < end; j += (uint64_t)gridDim.x * 256) {
        const uint32_t d = post_doc[j];
        if ((d >> 5) < acc_words) atomicOr(&bits[d >> 5], 1u << (d & 31u));
    }
}
constexpr int kAccScanThreads = 1024;
__global__ __launch_bounds__(kAccScanThreads) void acc_scan_kernel(const float* __restrict__ post_ntf, const uint64_t* __restrict__ list_off,
                                                                  const uint32_t* __restrict__ acc_list, uint32_t acc_words,
                                                                  uint32_t* __restrict__ acc, float* __restrict__ minmax) {
    __shared__ uint32_t part[kAccScanThreads];
    __shared__ float red_lo[kAccScanThreads / 64], red_hi[kAccScanThreads / 64];
    __shared__ uint32_t red_nan;
    const uint32_t i = blockIdx.x, l = acc_list[i], t = threadIdx.x;
    const uint32_t* bits = acc + (size_t)i * 2 * acc_words;
    uint32_t* wrank = acc + (size_t)i * 2 * acc_words + acc_words;
    const uint32_t per = (acc_words + kAccScanThreads - 1) / kAccScanThreads;
    const uint32_t w0 = min(t * per, acc_words), w1 = min(w0 + per, acc_words);
    uint32_t sum = 0;
    for (uint32_t w = w0; w < w1; ++w) sum += (uint32_t)__popc(bits[w]);
    part[t] = sum;
    if (t == 0) red_nan = 0u;
    __syncthreads();
    for (int off = 1; off < kAccScanThreads; off <<= 1) {  // (Hillis-Steele over 1 024 partial sums: the launch runs once per build)
        const uint32_t v = t >= (uint32_t)off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t w = w0; w < w1; ++w) {
        wrank[w] = run;
        run += (uint32_t)__popc(bits[w]);
    }
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
    bool nan = false;
    for (uint64_t j = list_off[l] + t; j < list_off[l + 1]; j += kAccScanThreads) {
        const float v = post_ntf[j];
        nan |= v != v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if (nan) atomicOr(&red_nan, 1u);
    if ((t & 63u) == 0) red_lo[t >> 6] = lo, red_hi[t >> 6] = hi;
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < kAccScanThreads / 64; ++w) lo = fminf(lo, red_lo[w]), hi = fmaxf(hi, red_hi[w]);
        minmax[2 * i] = red_nan ? __builtin_nanf("") : lo;
        minmax[2 * i + 1] = red_nan ? __builtin_nanf("") : hi;
    }
}
#endif

}  // namespace

#if ORAMA_COMPARISON_KERNELS
int launch_acc_build(const uint32_t* post_doc, const float* post_ntf, const uint64_t* d_list_off, const uint32_t* d_acc_list, uint32_t n_acc,
                     uint32_t acc_words, uint32_t* d_acc, float* d_minmax, hipStream_t stream) {
    if (n_acc == 0) return ORAMA_OK;
    ORAMA_HIP_TRY(hipMemsetAsync(d_acc, 0, (size_t)n_acc * 2 * acc_words * 4, stream));
    hipLaunchKernelGGL(acc_bits_kernel, dim3(512, n_acc), dim3(256), 0, stream, post_doc, d_list_off, d_acc_list, acc_words, d_acc);
    hipLaunchKernelGGL(acc_scan_kernel, dim3(n_acc), dim3(kAccScanThreads), 0, stream, post_ntf, d_list_off, d_acc_list, acc_words, d_acc, d_minmax);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
#endif

int launch_range_bounds(orama_ctx* ctx, const RangeBatch& b, hipStream_t stream) {
    if (b.total_postings == 0 || b.n_queries == 0 || b.max_bound_entries == 0) {
        if (b.n_queries) ORAMA_HIP_TRY(hipMemsetAsync(b.results, 0, (size_t)b.n_queries * sizeof(RangeResult), stream));
        return ORAMA_OK;
    }
    ORAMA_REQUIRE(b.n_queries <= kRangeBatchMax, "bm25 ranges: batch too large");
    ProfScope prof(&ctx->prof, "bm25_range_bounds", stream);
    const uint32_t lanes = b.n_queries <= 4 ? 16u : 1u;     // lanes that search one entry together
    const uint64_t per_block = kBoundsThreads / lanes;      // entries a workgroup searches
    const uint64_t blocks = (b.max_bound_entries + per_block - 1) / per_block;
    ORAMA_SUPPORT(blocks < 0x7fffffffull, "bm25 ranges: batch references too many postings");
    if (lanes == 16u) hipLaunchKernelGGL(range_bounds_kernel<16>, dim3((uint32_t)blocks, b.n_queries), dim3(kBoundsThreads), 0, stream, b);
    else hipLaunchKernelGGL(range_bounds_kernel<1>, dim3((uint32_t)blocks, b.n_queries), dim3(kBoundsThreads), 0, stream, b);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_range_score(orama_ctx* ctx, const RangeBatch& b, bool df_only, hipStream_t stream) {
    if (b.total_postings == 0 || b.n_queries == 0 || b.max_ranges == 0) return ORAMA_OK;
    ORAMA_REQUIRE(b.n_queries <= kRangeBatchMax, "bm25 ranges: batch too large");
#if ORAMA_COMPARISON_KERNELS
    if (ctx->k3r_merge) return launch_range_score_merge(ctx, b, df_only, stream);
#endif
    ProfScope prof(&ctx->prof, df_only ? "bm25_range_df" : "bm25_range_score", stream);
    const uint32_t grid = b.range_start[b.n_queries];
    ORAMA_REQUIRE(grid >= b.max_ranges, "internal: range_start table not filled");
    // 64-bit presence masks when a query of the batch has more than 32 references
    const bool wide = b.max_refs > 32;
    if (df_only) {
        if (wide) hipLaunchKernelGGL((range_score_kernel<true, true>), dim3(grid), dim3(kThreads), 0, stream, b);
        else hipLaunchKernelGGL((range_score_kernel<true, false>), dim3(grid), dim3(kThreads), 0, stream, b);
    } else {
        // (plain: the pre-divided tf at hand, nothing of the batch asks for a score map, OMC multipliers or min / max)
        const bool plain = !b.map_idx && !b.omc_dense && !b.any_minmax && b.post_ntf;
        ORAMA_REQUIRE(!b.compact_keys || (plain && !wide), "internal: compact key lists need the plain scoring launch");
        if (wide) hipLaunchKernelGGL((range_score_kernel<false, true>), dim3(grid), dim3(kThreads), 0, stream, b);
#if ORAMA_COMPARISON_KERNELS
        else if (plain && b.compact_keys && ctx->k3r_fast) return launch_range_score_fast(ctx, b, stream);
#endif
        else if (plain && b.compact_keys) {
            ORAMA_REQUIRE(b.score_pub, "internal: compact key lists without their published-score table");
            ORAMA_REQUIRE(b.stripe_start && b.stripe_total == grid, "internal: stripe table not filled");
            hipLaunchKernelGGL(range_score_compact_kernel, dim3(grid), dim3(kThreads), 0, stream, b);
        }
        else if (plain) hipLaunchKernelGGL((range_score_kernel<false, false, true>), dim3(grid), dim3(kThreads), 0, stream, b);
        else hipLaunchKernelGGL((range_score_kernel<false, false>), dim3(grid), dim3(kThreads), 0, stream, b);
    }
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_range_score_docs(orama_ctx* ctx, const RangeBatch& b, uint32_t qi, const uint32_t* d_doc, uint32_t n, float* d_out_score,
                            uint32_t* d_out_present, hipStream_t stream, const uint32_t* d_n) {
    (void)ctx;
    if (n == 0) return ORAMA_OK;
    hipLaunchKernelGGL(range_score_docs_kernel, dim3((n + kDocsThreads / 64 - 1) / (kDocsThreads / 64)), dim3(kDocsThreads), 0, stream, b, qi,
                       d_doc, n, d_out_score, d_out_present, d_n);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_ntf_precompute(const uint32_t* post_val, float* post_ntf, const uint64_t* d_list_off, const float* d_list_avg,
                          uint32_t n_lists, uint64_t n_postings, float b, hipStream_t stream) {
    if (n_postings == 0 || n_lists == 0) return ORAMA_OK;
    const uint64_t blocks = std::min<uint64_t>((n_postings + 255) / 256, 256ull * 64);
    hipLaunchKernelGGL(ntf_precompute_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, post_val, post_ntf, d_list_off, d_list_avg,
                       n_lists, n_postings, b);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
