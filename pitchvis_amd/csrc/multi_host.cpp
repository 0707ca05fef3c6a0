#include "multi_host.hpp"

namespace pvq {

bool plan_shard(uint64_t n_frames_total, uint64_t hop, uint64_t window_union, uint32_t rank, uint32_t world, ShardPlan* out) {
    if (!out || world == 0 || rank >= world) return false;
    const uint64_t base = n_frames_total / world, extra = n_frames_total % world;
    const uint64_t n = base + (rank < extra ? 1 : 0);
    const uint64_t first = (uint64_t)rank * base + (rank < extra ? rank : extra);
    const uint64_t hop_begin = first * hop;                              // first new sample of this shard
    const uint64_t halo = window_union > hop ? window_union - hop : 0;   // samples before hop_begin its first frame reads
    const uint64_t begin = hop_begin > halo ? hop_begin - halo : 0;
    out->first_frame = first;
    out->n_frames = n;
    out->sample_begin = begin;
    out->sample_end = hop_begin + n * hop;
    out->n_lead = hop_begin - begin;
    return true;
}

}  // namespace pvq
