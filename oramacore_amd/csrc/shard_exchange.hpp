// shard_exchange.hpp — host-level exchanges over a shard group for calls that run in PHASES on every shard (the batched
// full-text search over a sharded index, fulltext.hip): hold one lane of the group for the whole call, sum words over every
// shard of the index, gather one block per shard.  Implemented in shard_group.hip on the group's own communicators
// (RCCL, or nothing at all when every shard lives in this process).  SURVEY §8(e): df per token is an index-wide
// quantity (token_score.rs:262-275), the per-shard top-k lists are gathered and merged (sort.rs:260-279), counts are summed
// (search.rs:482).
#pragma once

#include <cstddef>
#include <cstdint>

struct orama_shard_group;

namespace orama {

// One lane of the group for the duration of a multi-phase call (collectives of concurrent calls must reach every rank in
// the same order: with ranks in other processes the group has one lane, i.e. one such call at a time).  Re-entrant on the
// holding thread: orama_shard_post_search called inside takes the same lane.
class ShardCall {
   public:
    explicit ShardCall(orama_shard_group* g) : g_(g) {}
    ShardCall(const ShardCall&) = delete;
    ShardCall& operator=(const ShardCall&) = delete;
    int init();
    ~ShardCall();

   private:
    orama_shard_group* g_;
    void* lease_ = nullptr;
};

// Shards of the index (all processes) / held by this process / global index of local shard 0.
void shard_group_shape(orama_shard_group* g, uint32_t* world, uint32_t* n_local, uint32_t* rank0);

// inout[0 .. count): this PROCESS's contribution (already summed over its local shards) -> the sum over every process of
// the group.  One all-reduce; a no-op when every shard lives in this process.  Values must stay below 2^31.
int shard_sum_u32(orama_shard_group* g, uint32_t* inout, size_t count);

// local_blocks: the blocks of this process's shards, `block_bytes` each, in local shard order -> out: the blocks of EVERY
// shard of the index in shard order (world x block_bytes).  One all-gather; a copy when every shard lives in this process.
int shard_gather_blocks(orama_shard_group* g, const void* local_blocks, size_t block_bytes, void* out);

}  // namespace orama
