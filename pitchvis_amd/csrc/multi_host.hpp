// multi_host.hpp — frame sharding of one stream over several devices: contiguous frame ranges, each with
// a halo of `window_union - hop` preceding samples, kernel tables replicated per device, no collective on the data path.
// The reference's only data-parallel driver gives every rayon worker its own Vqt and its own stream
// (pitchvis_train/src/train.rs:146-155); here the workers are devices and they share ONE stream, so the planner also says
// which samples a worker must hold.  Pure host arithmetic: used by pvq_plan_shard and by pvq_vqt_analyze_batch_multi.
#pragma once

#include <cstddef>
#include <cstdint>

namespace pvq {

struct ShardPlan {
    uint64_t first_frame;   // global index of the shard's first frame
    uint64_t n_frames;
    uint64_t sample_begin;  // first sample of the global stream the shard must hold (relative to the stream's first hop, i.e. after its own n_lead)
    uint64_t sample_end;    // one past the last
    uint64_t n_lead;        // history samples inside [sample_begin, sample_end) that precede the shard's first hop
};

// Contiguous split of `n_frames_total` frames over `world` shards; the first n_frames_total % world shards take one extra frame.
// Returns false for rank >= world or world == 0.
bool plan_shard(uint64_t n_frames_total, uint64_t hop, uint64_t window_union, uint32_t rank, uint32_t world, ShardPlan* out);

}  // namespace pvq
