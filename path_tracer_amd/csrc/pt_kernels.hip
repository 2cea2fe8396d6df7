// gfx950 (MI355X, wave64) kernels of the wavefront path tracer.
//
//   k_generate      main.rs:186-199   seed draw, shuffled-scrambled Sobol jitter, camera ray
//   k_closest       tlas.rs:66-110 + blas.rs:214-256 + boundingbox.rs:115-131 + primitive.rs:117-178
//                   persistent-threads ordered traversal; BVH staged in LDS; per-lane stack in LDS;
//                   ballot/mbcnt refill of idle lanes from the ray queue; material binning of the hits;
//                   paths that end at the hit or miss are finished here (integrator.rs:207-214, 263-266)
//   k_any           tlas.rs:111-144 + blas.rs:257-294   (shadow rays; finishes paths that died owing one explicit-light estimate)
//   k_shade_surface<class>, k_shade_terminal   integrator.rs:163-270 split per material class; the NEE of the previous bounce is
//                   resolved first; BSDF-sampled NEE rays that miss the lights' root box are answered on the spot
//                   (LDS-resident scenes: the Lambertian pass also walks its own explicit-light shadow rays, inline_any: no shadow queue, no k_any launch)
//   k_accumulate    integrator.rs:272-280 + accumulate.wgsl:20-23 in sample order
//
// Arithmetic: pt_math.h / pt_materials.h.  Built with -ffp-contract=off; v_min/v_max are used in the slab test only
// where they provably equal the SSE select semantics of the reference (see slab()).
#include "pt_kernels.h"
#include "pt_materials.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace pt {
namespace {

#ifndef PT_STEPS_PER_ROUND
#define PT_STEPS_PER_ROUND 8
#endif
#ifndef PT_REFILL_BELOW
#define PT_REFILL_BELOW 40
#endif
#ifndef PT_STEPS_PER_ROUND_GLOBAL_BVH
#define PT_STEPS_PER_ROUND_GLOBAL_BVH 4
#endif
#ifndef PT_REFILL_BELOW_GLOBAL_BVH
#define PT_REFILL_BELOW_GLOBAL_BVH 56
#endif
#ifndef PT_STEPS_ANY
#define PT_STEPS_ANY 4 // any-hit rays are short: same-box sweep 3 / 4 / 5 / 6 / 8 / 10 / 12: 70.8 / 67.6 / 69.0 / 69.4 / 69.9 / 69.3 / 69.6 ms per frame
#endif
#ifndef PT_REFILL_BELOW_ANY
#define PT_REFILL_BELOW_ANY PT_REFILL_BELOW
#endif
// waves per SIMD the compiler must leave room for in the traversal kernels (second parameter of __launch_bounds__): FIVE, for every variant.
// BVH in global memory: the kernels wait on L2 — one more DEPENDENT load per branch level, an L1 hit, costs the closest-hit kernel 10-13 %
// (experiments/r04_latency_probe.patch, round 4) — so a fifth wave pays, provided it is not bought with scratch: round 2 asked the compiler for 5 or
// 6 waves of the kernel as it was (123-127 VGPRs wanted: 72-164 B of scratch) and got +-0 / -15 %; round 4 first took the register peak away — the
// closest-hit leaves test ONE triangle at a time instead of two side by side (123 -> 105 VGPRs) — and then 96 VGPRs cost 8-20 B of scratch, all of it
// in the service sections.  Same-box A/B, whole frame at 64 spp, atrium / 82 k mesh / 328 k mesh / spheres: pairs at 4 waves 325 / 73.6 / 123.5 / 54.4 ms,
// single at 4 waves 309 / 70.1 / 119.6 / 52.9, single at 5 waves 281 / 67.2 / 115.0 / 49.6 (-14 / -9 / -7 / -9 %); 6 waves (80 VGPRs, 64-88 B of
// scratch, 13 stack levels in LDS) 315 / 71.7 / 124.0 / 54.4; any-hit alone at 6: 295 / 68.3 / 114.7 / 51.4.
// BVH in LDS: these kernels run at 0.80-0.87 of VALU issue at four waves, and the same change fills most of the rest: Cornell 256 spp 66.3 -> 61.4 ms,
// its 1/8 share 10.1 -> 9.4 ms, mixed materials 77.4 -> 72.2 ms (k_trace_fused then carries 48 B of scratch).  Pairs at five waves 63.3 / 9.55 / 74.8; single
// at four 65.1 / 10.1 / 75.6; closest-hit four + any-hit five 64.5 / 9.7 / 74.7; closest-hit five + any-hit six 62.9 / 9.5 / 74.5; six for both 66.4 / 10.0 / 77.3;
// world + NEE launches unfused at five 62.1 / 9.5 / 73.5.
#ifndef PT_WAVES_LDS_BVH
#define PT_WAVES_LDS_BVH 5
#endif
#ifndef PT_WAVES_LDS_BVH_ANY
#define PT_WAVES_LDS_BVH_ANY 5
#endif
#ifndef PT_WAVES_GLOBAL_BVH
#define PT_WAVES_GLOBAL_BVH 5
#endif
#ifndef PT_WAVES_GLOBAL_BVH_ANY
#define PT_WAVES_GLOBAL_BVH_ANY 5
#endif
// branch levels expanded per traversal step of k_closest (1 = one node per step)
#ifndef PT_BRANCH_LEVELS
#define PT_BRANCH_LEVELS 4
#endif
#ifndef PT_BRANCH_LEVELS_ANY
#define PT_BRANCH_LEVELS_ANY 4
#endif
#ifndef PT_BRANCH_LEVELS_INLINE
#define PT_BRANCH_LEVELS_INLINE 6 // ... of the shadow walk inside the Lambertian shading pass (inline_any; no refill there, so fewer, longer steps: same-box 1 / 2 / 3 / 4 / 6 / 8: Cornell frame 59.05 / 57.9 / 57.8 / 57.4 / 56.95 / 56.9 ms, one pipeline 60.8 / 59.9 / 59.9 / 59.3 / 59.0 / 59.3)
#endif
// frames with fewer local pixels than this use four lanes per pixel in k_accumulate (k_accumulate<FEW_PIXELS>)
#ifndef PT_ACC_QUAD_BELOW
#define PT_ACC_QUAD_BELOW (1u << 30) // (whole 1080p frame: 0.675 -> 0.55 ms; beyond 2^30 pixels the thread index would overflow)
#endif
// PT_STEP_STATS (variant builds, tools/step_stats.py): per traversal step of k_closest, how many lanes take each section
#ifndef PT_STEP_STATS
#define PT_STEP_STATS 0
#endif
// PT_WAVE_TIMES (variant builds, tools/wave_times.py): when the waves of a k_closest launch start, first have rays, find the queue empty, end
#ifndef PT_WAVE_TIMES
#define PT_WAVE_TIMES 0
#endif
// a wave takes part in a launch only if the queue holds this many 64-ray generations for it (fetch_plan)
#ifndef PT_MIN_GENERATIONS
#define PT_MIN_GENERATIONS 1
#endif
// striped shade-queue tails: see stripes_for
#ifndef PT_STRIPE_DIV
#define PT_STRIPE_DIV 64
#endif
#ifndef PT_REGIONS_PER_STRIPE
#define PT_REGIONS_PER_STRIPE 256
#endif
#ifndef PT_REGION_DIV
#define PT_REGION_DIV 16
#endif
// tapered chunks: 64-ray chunks per wave in a queue's last level (0 = chunks of one size; chunk_of)
#ifndef PT_TAPER
#define PT_TAPER 2
#endif
#ifndef PT_TAPER_ANY
#define PT_TAPER_ANY 0
#endif
#ifndef PT_RESERVICE
#define PT_RESERVICE 1
#endif
#ifndef PT_CLAIM_AHEAD
#define PT_CLAIM_AHEAD 0 // rays left in a chunk when the next one is claimed; 0 = as soon as the chunk is taken up (same-box A/B of 64 / 128 / 256: +0.3 ms for a 1/8 share, +3 ms per frame)
#endif
// threads per shading workgroup (a workgroup makes one reservation per queue and iteration: block_append4)
// waves per SIMD the surface shading kernels must leave room for (1: whatever the registers they want allow — 102-130 VGPRs: four, GGX with volumes three).
// Five: the Lambertian kernel fits 96 VGPRs without scratch.  Round 3 measured that at FOUR traversal waves: the pass's own launches 22.39 -> 22.26 ms, the
// two-pipeline frame +1 ms.  Round 4, traversal kernels at five waves: one pipeline 62.35 -> 62.63 ms (nothing), but the default two-pipeline frame
// 61.2 -> 59.0 ms and mixed materials 72.8 -> 70.0 ms: the fifth wave lets one pipeline's shading pass run beside the other's traversal.  Six (80 VGPRs, 60 B
// of scratch): 64.4 / 76.0 ms.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 5
#endif
#ifndef PT_SHADE_WAVES_DIEL
#define PT_SHADE_WAVES_DIEL 5   // (16 B of scratch)
#endif
#ifndef PT_SHADE_WAVES_GGX
#define PT_SHADE_WAVES_GGX 5    // (64 B of scratch, and still better than four waves: mixed materials 71.5 -> 70.3 ms)
#endif
#ifndef PT_SHADE_WAVES_VOLUMES
#define PT_SHADE_WAVES_VOLUMES 4 // kernels of scenes with participating media (five: 32-104 B of scratch; media scene 47.2 -> 48.7 ms)
#endif
constexpr int shade_waves(uint32_t qclass, bool volumes)
{
    return volumes ? PT_SHADE_WAVES_VOLUMES : (qclass == Q_GGX ? PT_SHADE_WAVES_GGX : (qclass == Q_DIELECTRIC ? PT_SHADE_WAVES_DIEL : PT_SHADE_WAVES));
}
// LDS-resident scenes: the Lambertian shading pass traces its explicit-light shadow ray itself (inline_any) instead of queueing it for k_any<SHADOW>: same box,
// Cornell frame 58.45 -> 57.5 ms, one pipeline 62.1 -> 59.8, 1/4 and 1/8 share 16.4 / 9.4 -> 15.6 / 8.95, mixed materials 64 spp 21.1 -> 20.0.  At four waves per SIMD
// (PT_SHADE_WAVES_INLINE, 128 VGPRs) it wins on one pipeline only (61.7) and loses the two-pipeline overlap (61.5); five waves need the walk AFTER the record store.
#ifndef PT_INLINE_SHADOW
#define PT_INLINE_SHADOW 1
#endif
#ifndef PT_SHADE_WAVES_INLINE
#define PT_SHADE_WAVES_INLINE 5
#endif
#ifndef PT_SHADE_THREADS
#define PT_SHADE_THREADS 256
#endif
#ifndef PT_CHUNK_MAX
#define PT_CHUNK_MAX 1024
#endif
#ifndef PT_CHUNK_DIV
#define PT_CHUNK_DIV 4
#endif
// scenes whose BVH stays in global memory have long rays: smaller chunks, shorter launch tails (same-box A/B on the 82 k / 328 k meshes: -3 % / -7 %)
#ifndef PT_CHUNK_DIV_GLOBAL_BVH
#define PT_CHUNK_DIV_GLOBAL_BVH 16
#endif
// traversal steps between refill checks, and: refill idle lanes when at most this many lanes are still traversing.  Scenes whose BVH
// stays in global memory (long rays, every step waits on L2) want their lanes refilled sooner and more often: same-box A/B of
// 32…62 lanes x 4…16 steps on the 82 k / 328 k meshes: 56 lanes x 4 steps -6 % / -5 % against the LDS scenes' 40 x 8.
template <bool LDS_SCENE> struct Refill
{
    static constexpr int kSteps = LDS_SCENE ? PT_STEPS_PER_ROUND : PT_STEPS_PER_ROUND_GLOBAL_BVH;
    static constexpr int kStepsAny = LDS_SCENE ? PT_STEPS_ANY : PT_STEPS_PER_ROUND_GLOBAL_BVH;
    static constexpr int kBelow = LDS_SCENE ? PT_REFILL_BELOW : PT_REFILL_BELOW_GLOBAL_BVH;
    static constexpr int kBelowAny = LDS_SCENE ? PT_REFILL_BELOW_ANY : PT_REFILL_BELOW_GLOBAL_BVH;
};

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f)
{
    const uint32_t q = __umulhi(f.magic, n);
    const uint32_t t = ((n - q) >> 1) + q;
    return f.one ? n : (t >> f.shift);
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// Persistent-thread work fetch.  A single queue-head word sustains only ~88 dequeues/us on MI355X — 16 k claims take 186 us, which
// used to be most of a 1 M-ray launch — so:
//  * the first chunk of every wave is static (wave w owns rays [w*chunk, (w+1)*chunk)): a launch with few rays costs no atomics
//    at all, and only as many workgroups as the queue has 64-ray chunks take part (the others leave before staging anything);
//  * the rest of the queue is cut into kQueueHeads partitions with one claim cursor each, every cursor in a cache line of its own
//    (QueueHeads, pt_types.h).  A wave that runs dry looks at all cursors with ONE 64-lane load, claims a chunk with one atomic
//    from the first partition at or after its home partition that has one left, and so drifts on to other partitions (steals)
//    when its own is empty.  "Nothing left anywhere" is seen by that load: no wave ends with an atomic that finds the queue empty;
//  * with claims this cheap the chunks can be small (n / 4 per wave, 64..1024 rays; same-box sweep: 512/8 the same, 256/16 and
//    2048/2 worse): the launch ends about one small chunk after the last claim instead of after the slowest wave's two big ones.
struct FetchPlan
{
    uint32_t n;        // rays (queue slots) in the launch
    uint32_t chunk;    // rays per static chunk, and the most a dynamic claim takes; multiple of 64
    uint32_t n_static; // slots [0, n_static) are owned statically, chunk by chunk, by the participating waves
    uint32_t psize;    // slots per dynamic partition, multiple of 64
    uint32_t lvl;      // the last `lvl` slots of a partition are handed out 64 at a time, the 2 lvl before them 128 at a time, ... (chunk_of)
    uint32_t blocks;   // workgroups that take part
};
// The chunks of a partition, in the order they are handed out: big ones first, small ones last, so that what the waves still hold
// when the queue runs dry — each its current chunk and the one it claimed ahead — is little.  (Per-wave time records of a 17 M-ray
// launch, tools/wave_times.py: with 1024-ray chunks throughout the waves end anywhere in the last 590 us of 1510, an average wave
// idles for the last 18 % of the launch; with 128-ray chunks throughout they end within 90 us of each other, but every chunk
// boundary costs ~5 us of a wave's time — a refill cut short — and the launch is no shorter.)  The cursor of a partition counts
// chunks, not slots.  false: the partition has no chunk i.
__device__ __forceinline__ bool chunk_of(const FetchPlan& pl, uint32_t i, uint32_t& start, uint32_t& len)
{
    uint32_t rem = pl.psize, base = 0u;
    const uint32_t e64 = min(rem, pl.lvl); rem -= e64;
    const uint32_t e128 = min(rem, 2u * pl.lvl); rem -= e128;
    const uint32_t e256 = min(rem, 4u * pl.lvl); rem -= e256;
    const uint32_t e512 = min(rem, 8u * pl.lvl); rem -= e512;
    const uint32_t e1024 = rem;
#define PT_LEVEL(E, C)                                                                                     \
    {                                                                                                      \
        const uint32_t c = min((uint32_t)(C), pl.chunk), cnt = ((E) + c - 1u) / c;                          \
        if (i < cnt) { start = base + i * c; len = min(c, (E) - i * c); return true; }                      \
        i -= cnt;                                                                                          \
        base += (E);                                                                                       \
    }
    PT_LEVEL(e1024, 1024u) PT_LEVEL(e512, 512u) PT_LEVEL(e256, 256u) PT_LEVEL(e128, 128u) PT_LEVEL(e64, 64u)
#undef PT_LEVEL
    return false;
}
__device__ __forceinline__ FetchPlan fetch_plan(uint32_t n, uint32_t chunk_div, uint32_t taper)
{
    FetchPlan pl;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t need_waves = (n + 63u) >> 6;
    pl.blocks = min(gridDim.x, (need_waves + wpb * (uint32_t)PT_MIN_GENERATIONS - 1u) / (wpb * (uint32_t)PT_MIN_GENERATIONS));
    const uint32_t waves = max(pl.blocks, 1u) * wpb;
    uint32_t c = n / (waves * chunk_div);
    c = c < 64u ? 64u : (c > (uint32_t)PT_CHUNK_MAX ? (uint32_t)PT_CHUNK_MAX : c);
    pl.chunk = (c + 63u) & ~63u;
    pl.n = n;
    const uint64_t st = (uint64_t)waves * pl.chunk;
    pl.n_static = st >= n ? n : (uint32_t)st;
    const uint32_t units = (n - pl.n_static + 63u) >> 6;
    pl.psize = ((units + kQueueHeads - 1u) / kQueueHeads) << 6;
    // over all partitions the 64-ray level holds PT_TAPER chunks per wave (`taper`; 2 = the current one and the one claimed ahead)
    pl.lvl = taper ? ((taper * waves * 64u / kQueueHeads + 63u) & ~63u) : 0u;
    return pl;
}
struct WaveRange
{
    uint32_t cur, end;
    uint32_t home;    // partition this wave tries first
    bool drained;     // the queue has no chunk left for this wave
    // The claim for the chunk AFTER the current one is issued when the current one is taken up and its answer is looked at only
    // when the current one runs out, so the atomic's round trip overlaps the traversal instead of stalling the wave (a same-box
    // A/B of synchronous claims: every halving of the chunk size cost ~2 % of the frame).  A claim that succeeded owns its chunk:
    // the wave always looks at the answer before it leaves.
    bool nx_valid;    // a claim is outstanding
    uint32_t nx_p;    // ... on this partition
    uint32_t nx_got;  // ... and this is the atomic's return value (lane 0): the chunk's index in the partition
};
__device__ __forceinline__ void prefetch_claim(WaveRange& wr, uint32_t* heads, const FetchPlan& pl)
{
    wr.nx_valid = true;
    wr.nx_p = wr.home;
    if (lane_id() == 0u) wr.nx_got = atomicAdd(heads + wr.home * kHeadStrideWords, 1u);
}
// chunk `idx` of partition p, clipped to the queue; false if nothing of it exists
__device__ __forceinline__ bool take_claim(WaveRange& wr, const FetchPlan& pl, uint32_t p, uint32_t idx)
{
    const uint32_t dyn = pl.n - pl.n_static;
    uint32_t start, len;
    if (!chunk_of(pl, idx, start, len)) return false;
    const uint64_t off = (uint64_t)p * pl.psize + start;
    if (off >= dyn) return false;
    const uint64_t stop = min(off + len, (uint64_t)dyn);
    wr.cur = pl.n_static + (uint32_t)off;
    wr.end = pl.n_static + (uint32_t)stop;
    return true;
}
__device__ __forceinline__ WaveRange first_range(const FetchPlan& pl, uint32_t* heads)
{
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: keep the range in SGPRs
    const uint64_t lo = (uint64_t)wave * pl.chunk;
    WaveRange wr;
    wr.cur = lo >= pl.n_static ? pl.n : (uint32_t)lo;
    wr.end = lo >= pl.n_static ? pl.n : (uint32_t)min((uint64_t)pl.n_static, lo + pl.chunk);
    wr.home = (wave * 7u) & (kQueueHeads - 1u); // neighbouring waves start on different cursors
    wr.drained = pl.n_static >= pl.n;            // nothing is handed out dynamically
    wr.nx_valid = false;
    wr.nx_p = 0u;
    wr.nx_got = 0u;
    if (!wr.drained && (PT_CLAIM_AHEAD == 0 || wr.end - wr.cur <= (uint32_t)PT_CLAIM_AHEAD)) prefetch_claim(wr, heads, pl);
    return wr;
}
// returns how many of the wave's idle lanes receive a ray; lane i (rank r among idle lanes) gets ray first + r
__device__ __forceinline__ uint32_t claim_rays(WaveRange& wr, uint32_t* heads, const FetchPlan& pl, uint32_t n_idle, uint32_t& first)
{
    if (wr.cur >= wr.end && !wr.drained)
    {
        bool have = false;
        if (wr.nx_valid)
        {
            const uint32_t got = __builtin_amdgcn_readfirstlane(wr.nx_got);
            wr.nx_valid = false;
            have = take_claim(wr, pl, wr.nx_p, got);
        }
        const uint32_t l = lane_id();
        while (!have) // the home partition is empty: look at all cursors; at most kQueueHeads rounds (a failed claim = an empty partition)
        {
            const uint32_t h = __hip_atomic_load(heads + l * kHeadStrideWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t st_l = 0u, len_l = 0u;
            const bool more = chunk_of(pl, h, st_l, len_l) && (uint64_t)l * pl.psize + st_l < (uint64_t)(pl.n - pl.n_static);
            const uint64_t m = __ballot(more);
            if (m == 0ull) { wr.drained = true; wr.cur = wr.end = pl.n; break; }
            const uint64_t rot = wr.home == 0u ? m : ((m >> wr.home) | (m << (64u - wr.home)));
            const uint32_t p = (wr.home + (uint32_t)__builtin_ctzll(rot)) & (kQueueHeads - 1u);
            uint32_t got = 0;
            if (l == 0u) got = atomicAdd(heads + p * kHeadStrideWords, 1u);
            got = __builtin_amdgcn_readfirstlane(got);
            wr.home = p;
            have = take_claim(wr, pl, p, got);
        }
        if (!wr.drained && PT_CLAIM_AHEAD == 0) prefetch_claim(wr, heads, pl);
    }
    const uint32_t take = min(n_idle, wr.end - wr.cur);
    first = wr.cur;
    wr.cur += take;
    // The next chunk is claimed when the current one has PT_CLAIM_AHEAD rays left — a couple of refills, time enough for the atomic's
    // round trip — not when it is taken up: what a wave owns when the queue runs dry is then about one chunk, not two, and the
    // launch's tail is that much shorter.
    if (PT_CLAIM_AHEAD != 0 && !wr.nx_valid && !wr.drained && wr.end - wr.cur <= (uint32_t)PT_CLAIM_AHEAD) prefetch_claim(wr, heads, pl);
    return take;
}

// exact count of what a wave processed, added to the cursor line of its home partition (64 addresses per queue instead of one)
__device__ __forceinline__ void add_tally(uint32_t* heads, uint32_t per_lane, uint32_t word)
{
    uint32_t total = per_lane;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (total != 0u && lane_id() == 0u) atomicAdd(heads + ((wave * 7u) & (kQueueHeads - 1u)) * kHeadStrideWords + word, total);
}

// Queue appends.  A returning atomic on ONE queue-tail word sustains ~88 operations per microsecond on MI355X (price list
// "dequeue"), and a wavefront renderer wants ~10^10 appended entries per second, so nobody appends entry by entry:
//  * a TRAVERSAL WAVE (k_closest -> shade queues) RESERVES a private region of the output queue with one atomic, fills it locally
//    and reserves the next one when it runs out.  Whatever is left of its last region when the wave exits is filled with HOLE
//    markers that consumers skip; these queues' extents therefore count slots, not entries.  The surface classes' queues hand their
//    regions out over up to 64 striped tail words (wave_reserve_striped below); the terminal queue, which sees little traffic, keeps
//    one tail and regions of 64 slots, or slots_in / (waves * 16) clamped to [64, 8192] when an environment map sends every miss there (region_size);
//  * a SHADING WORKGROUP (-> ray queues, terminal queue) reserves exactly what it appends, once per queue and iteration
//    (block_append4): those queues are dense.  Holes are not free — a hole is an idle lane until the next refill — and were measured
//    to cost more than the reservations they save, see block_append4.
// (Tried: regions shared by the four waves of a traversal workgroup through an LDS cursor: tail atomics / 4, but the per-retirement
// LDS atomic cost 2 ms per frame.)
//
// Capacity: a reservation that would pass the queue's capacity is diverted to the queue's dump area (kQueueDumpSlots slots past the
// capacity, shared by everybody who overflows) and raises the batch's overflow flag: nothing is ever stored out of bounds, the
// host turns the flag into PT_ERR_LIMIT.  Consumers clamp the slot count they read to the capacity.
enum : uint32_t { PRIMARY_MISS = 0xffu /* PathState::occl: the camera ray left the scene at once (radiance = ambient) */ };
enum : uint32_t { HOLE = 0xffffffffu, PATH_ENDS = 0x80000000u /* shadow-ray path ids: see k_any */ };
struct Region { uint32_t cur, end; };
// `least`: the most one reservation has to take in one go (64 for a wave, 256 for a workgroup)
__device__ __forceinline__ uint32_t region_size(uint32_t n_in, uint32_t producers, uint32_t least)
{
    uint32_t r = n_in / (max(producers, 1u) * (uint32_t)PT_REGION_DIV);
    r = r < least ? least : (r > kQueueDumpSlots ? (uint32_t)kQueueDumpSlots : r);
    return (r + 63u) & ~63u;
}
// one thread: next region of `rsize` slots, or the dump area when the queue is full
__device__ __forceinline__ uint32_t next_region(const Region& rg, uint32_t* counter, uint32_t rsize, uint32_t cap, uint32_t* overflow)
{
    if (rg.end > cap) return cap; // already diverted: stay in the dump area and leave the counter alone (it must not wrap)
    const uint32_t nb = atomicAdd(counter, rsize);
    if (cap < rsize || nb > cap - rsize)
    {
        __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return cap;
    }
    return nb;
}
// wave-uniform: take `total` slots; entries ranked below `left` go to base0 + rank, the others to base1 + (rank - left)
struct Placement { uint32_t base0, left, base1; };
__device__ __forceinline__ uint32_t place(const Placement& p, uint32_t rank) { return rank < p.left ? p.base0 + rank : p.base1 + (rank - p.left); }
__device__ __forceinline__ Placement wave_reserve(Region& rg, uint32_t* counter, uint32_t total, uint32_t rsize, uint32_t cap, uint32_t* overflow)
{
    Placement p{rg.cur, rg.end - rg.cur, 0u};
    if (total > p.left)
    {
        uint32_t nb = 0;
        if (lane_id() == 0u) nb = next_region(rg, counter, rsize, cap, overflow);
        nb = __shfl(nb, 0);
        p.base1 = nb;
        rg.cur = nb + (total - p.left);
        rg.end = nb + rsize;
    }
    else rg.cur += total;
    return p;
}

// ---- striped tails (pt_types.h): the same region scheme with the reservations spread over up to 64 tail words
struct Stripes { uint32_t log_r, log_k; }; // region = 1 << log_r slots, 1 << log_k stripes
// `n_key` = slots of the bounce's closest-hit input queue: what the producers (k_closest) and the consumers (k_shade_surface) of the
// bounce's shade queues both know.  Regions of n_key / (4096 * PT_STRIPE_DIV) slots rounded down to a power of two in [64, 8192]
// (4096 = the traversal waves of a full launch); about PT_REGIONS_PER_STRIPE regions per stripe, so a short queue has one tail
// and no gaps at all, a long one 64.
__device__ __forceinline__ Stripes stripes_for(uint32_t n_key)
{
    const uint32_t want = n_key / (4096u * (uint32_t)PT_STRIPE_DIV);
    Stripes st;
    st.log_r = want < 64u ? 6u : min(13u, 31u - (uint32_t)__clz(want));
    const uint32_t per = (n_key >> st.log_r) / (uint32_t)PT_REGIONS_PER_STRIPE;
    st.log_k = per == 0u ? 0u : min(6u, 31u - (uint32_t)__clz(per));
    return st;
}
// one thread: the next region of stripe `rot & (K - 1)`, or the dump area when the queue is full
__device__ __forceinline__ uint32_t next_region_striped(const Region& rg, uint32_t* tails, const Stripes st, uint32_t rot, uint32_t cap, uint32_t* overflow)
{
    if (rg.end > cap) return cap; // already diverted: stay in the dump area
    const uint32_t k = rot & ((1u << st.log_k) - 1u);
    const uint32_t r = atomicAdd(tails + k * kTailStrideWords, 1u);
    const uint64_t base = (((uint64_t)r << st.log_k) | k) << st.log_r;
    if (base + (1ull << st.log_r) > (uint64_t)cap)
    {
        __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return cap;
    }
    return (uint32_t)base;
}
// wave-uniform, like wave_reserve; `rot` moves on with every reservation so that a producer visits all stripes in turn
__device__ __forceinline__ Placement wave_reserve_striped(Region& rg, uint32_t* tails, const Stripes st, uint32_t& rot, uint32_t total, uint32_t cap, uint32_t* overflow)
{
    Placement p{rg.cur, rg.end - rg.cur, 0u};
    if (total > p.left)
    {
        uint32_t nb = 0;
        if (lane_id() == 0u) nb = next_region_striped(rg, tails, st, rot, cap, overflow);
        nb = __builtin_amdgcn_readfirstlane(nb);
        rot += 1u;
        p.base1 = nb;
        rg.cur = nb + (total - p.left);
        rg.end = nb + (1u << st.log_r);
    }
    else rg.cur += total;
    return p;
}
// consumer side.  Lane l of the calling wave loads tail l; returns the queue's extent in slots (clamped to cap), wave-uniform
__device__ __forceinline__ uint32_t stripe_load(const uint32_t* tails, const Stripes st, uint32_t cap, uint32_t& my_tail)
{
    const uint32_t l = lane_id();
    my_tail = l < (1u << st.log_k) ? tails[l * kTailStrideWords] : 0u;
    uint64_t e = my_tail != 0u ? (((((uint64_t)(my_tail - 1u)) << st.log_k) | l) + 1ull) << st.log_r : 0ull;
    e = e > (uint64_t)cap ? (uint64_t)cap : e;
    uint32_t ext = (uint32_t)e;
    for (int off = 32; off > 0; off >>= 1) ext = max(ext, (uint32_t)__shfl_xor((int)ext, off));
    return ext;
}
// does slot `idx` lie in a region that some producer reserved?  `tail_of` = the 64 tails (LDS)
__device__ __forceinline__ bool stripe_valid(const uint32_t* tail_of, const Stripes st, uint32_t idx)
{
    const uint32_t g = idx >> st.log_r;
    return (g >> st.log_k) < tail_of[g & ((1u << st.log_k) - 1u)];
}

// Block-wide queue append for up to four queues at once: the workgroup reserves EXACTLY what it appends, one atomic per queue and
// call (two barriers).  Regions per workgroup, as the traversal waves use them, were tried first (region = slots_in / (workgroups * 16),
// at least 256): fewer atomics, but whatever 3072 workgroups leave of their last regions are holes in the next launch's queue — for
// a 0.14 M-ray bounce of a 1/8 share 0.4 M of them — and holes cost more than the atomics even where those run at the one-word limit
// (~88 per microsecond is what a full-speed shading launch makes: 22 entries per ns / 256).  Same-box A/B, exact against regions:
// whole frame 69.6 -> 69.0 ms, 1/2 share 36.9 -> 35.4, 1/4 19.3 -> 18.6, 1/8 10.8 -> 10.2 ms, 1-spp frame 1.40 -> 1.22 ms, mixed
// materials 14.5 -> 12.8 ms, 82 k mesh 14.1 -> 13.4 ms.
struct BlockAppend
{
    uint32_t wave_cnt[4][PT_SHADE_THREADS / 64];
    uint32_t wave_rank[4][PT_SHADE_THREADS / 64]; // rank of the wave's first entry inside the workgroup's tile
    uint32_t base[4];
};
__device__ __forceinline__ void block_append4(BlockAppend& sh, uint32_t* const counters[4], const bool pred[4], const uint32_t caps[4], uint32_t* overflow, uint32_t pos[4])
{
    const uint32_t wid = threadIdx.x >> 6;
    uint64_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        m[q] = __ballot(pred[q]);
        if (lane_id() == 0u) sh.wave_cnt[q][wid] = (uint32_t)__popcll(m[q]);
    }
    __syncthreads();
    if (threadIdx.x < 4u)
    {
        const uint32_t q = threadIdx.x;
        const uint32_t nw = blockDim.x >> 6;
        uint32_t total = 0;
        for (uint32_t w = 0; w < nw; ++w) { sh.wave_rank[q][w] = total; total += sh.wave_cnt[q][w]; }
        uint32_t base = caps[q]; // a full queue: the entries go to its dump area (kQueueDumpSlots >= a workgroup's 256), nothing is stored out of bounds
        if (total != 0u)
        {
            const uint32_t nb = atomicAdd(counters[q], total);
            if (nb > caps[q] || total > caps[q] - nb) __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else base = nb;
        }
        sh.base[q] = base;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) pos[q] = sh.base[q] + sh.wave_rank[q][wid] + mbcnt64(m[q]);
    __syncthreads(); // the shared tables are rewritten by the next iteration
}

__device__ __forceinline__ f3 xyz(const f4& v) { return f3{v.x, v.y, v.z}; }
__device__ __forceinline__ float asf(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t asu(float f) { return __float_as_uint(f); }

// Streaming accesses of the shading pass: rays and hit records are written once and read once, by the next kernel, long after the
// ~200 GB in between have flushed every cache; "nt" accesses keep them from displacing the records and scene tables in L2
// (same-box A/B: -0.6 ms per frame; the 64-byte path records must NOT be streamed: their four words merge in L2, +4 ms without).
typedef float nt4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(f4* p, f4 v) { __builtin_nontemporal_store(nt4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4_t*>(p)); }
__device__ __forceinline__ f4 nt_load(const f4* p) { const nt4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt4_t*>(p)); return f4{v.x, v.y, v.z, v.w}; }


// AABB::intersect / intersect_t (boundingbox.rs:97-131):
//   t0 = (min - o) * inv, t1 = (max - o) * inv;  t_small = min(max(t0, EPS), max(t1, EPS));  t_big = max(min(t0, t_max), min(t1, t_max));
//   hit iff max_element(t_small) <= min_element(t_big), t_enter = max_element(t_small)      (glam min/max = SSE selects: b wins on NaN)
// evaluated with half the min/max work by picking, per axis, the plane the ray reaches first.  Subtraction and the multiply by
// inv are monotone, so for inv >= 0 (incl. +inf) t0 <= t1 and for inv < 0 (incl. -inf) t1 <= t0 whenever both are numbers; then
// t_small = max(t_near, EPS) and t_big = min(t_far, t_max) exactly.  A NaN arises only as 0 * inf (origin on the plane, zero
// direction component): the reference's selects turn that axis into (EPS, t_max), i.e. no constraint, and so do IEEE
// maxNum/minNum (v_max3_f32/v_min3_f32), which drop a NaN operand.  If both planes give NaN the axis is unconstrained on both
// sides as well.  Zero signs cannot reach the result (t_enter >= EPS; t_big is only compared).  Requires t_max not NaN (a NaN
// t_max fails every reference box test; call sites handle it).
__device__ __forceinline__ bool slab(const uint4 w0, const uint4 w1, const f3 o, const f3 inv, const float t_max, float& t_enter)
{
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    // (near, far) pairs: one v_pk_add_f32 + one v_pk_mul_f32 per axis
    typedef float pair_t __attribute__((ext_vector_type(2)));
    const pair_t px = (pair_t{asf(nx ? w1.x : w0.x), asf(nx ? w0.x : w1.x)} - pair_t{o.x, o.x}) * pair_t{inv.x, inv.x};
    const pair_t py = (pair_t{asf(ny ? w1.y : w0.y), asf(ny ? w0.y : w1.y)} - pair_t{o.y, o.y}) * pair_t{inv.y, inv.y};
    const pair_t pz = (pair_t{asf(nz ? w1.z : w0.z), asf(nz ? w0.z : w1.z)} - pair_t{o.z, o.z}) * pair_t{inv.z, inv.z};
    const float tnx = px.x, tfx = px.y, tny = py.x, tfy = py.y, tnz = pz.x, tfz = pz.y;
    const float ts = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), PT_EPSILON);
    const float tb = fminf(fminf(fminf(tfx, tfy), tfz), t_max);
    t_enter = ts;
    return ts <= tb;
}

__device__ __forceinline__ bool sign_differs(float a, float b) // a.signum() != b.signum()
{
    return __builtin_isunordered(a, b) || (((asu(a) ^ asu(b)) >> 31) != 0u);
}

// Triangle::intersect_naive on the pre-translated origin (primitive.rs:117-155)
__device__ __forceinline__ bool tri_planes(const uint4 p0, const uint4 p1, const uint4 p2, const f3 o, const f3 d, const float t_max,
                                           const float t_est, float& td, float& ud, float& vd, float& det)
{
    const f3 mo = fma3(d, bc3(t_est), o); // ray.at(t_estimate)
    const float t_min = PT_EPSILON - t_est, t_mx = t_max - t_est;
    const f4 n0{asf(p0.x), asf(p0.y), asf(p0.z), asf(p0.w)};
    det = dot3(d, xyz(n0));
    td = -dot4(f4{mo.x, mo.y, mo.z, -1.0f}, n0);
    const bool out_t = sign_differs(td - det * t_min, det * t_mx - td);
    const f3 p = det * mo + td * d;
    const f4 p4{p.x, p.y, p.z, det};
    ud = dot4(p4, f4{asf(p1.x), asf(p1.y), asf(p1.z), asf(p1.w)});
    const bool out_u = sign_differs(ud, det - ud);
    vd = dot4(p4, f4{asf(p2.x), asf(p2.y), asf(p2.z), asf(p2.w)});
    const bool out_v = sign_differs(vd, det - ud - vd);
    // the three rejections of primitive.rs:122-140 without early exits: the traversal loop is more sensitive to exec-mask operations
    // than to the ~25 VALU operations a rejected triangle would have skipped (same-box A/B: -0.9 ms per frame)
    return !(out_t | out_u | out_v);
}

// the same test in two parts, for leaves that evaluate two triangles side by side: everything that does not depend on t_max ...
struct TriEval { float td, ud, vd, det; bool uv_ok; };
__device__ __forceinline__ TriEval tri_eval(const uint4* tp, const f3 mo, const f3 d)
{
    TriEval e;
    const uint4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
    const f4 n0{asf(p0.x), asf(p0.y), asf(p0.z), asf(p0.w)};
    e.det = dot3(d, xyz(n0));
    e.td = -dot4(f4{mo.x, mo.y, mo.z, -1.0f}, n0);
    const f3 p = e.det * mo + e.td * d;
    const f4 p4{p.x, p.y, p.z, e.det};
    e.ud = dot4(p4, f4{asf(p1.x), asf(p1.y), asf(p1.z), asf(p1.w)});
    e.vd = dot4(p4, f4{asf(p2.x), asf(p2.y), asf(p2.z), asf(p2.w)});
    const bool out_u = sign_differs(e.ud, e.det - e.ud), out_v = sign_differs(e.vd, e.det - e.ud - e.vd);
    e.uv_ok = !out_u && !out_v;
    return e;
}
// ... and the range test against the current t_max (primitive.rs:122-123)
__device__ __forceinline__ bool tri_in_range(const TriEval& e, float t_min, float t_mx) { return !sign_differs(e.td - e.det * t_min, e.det * t_mx - e.td); }

struct Blob
{
    const uint4* nodes; // 2 words per node
    const uint4* tris;  // 3 words per triangle
    const uint4* inst;  // INST_WORDS words per instance
    const uint2* leaves; // big-leaf table {first, count} (NODE_TRIS_BIG)
};
// triangles of a leaf link
__device__ __forceinline__ void leaf_range(const Blob& bl, uint32_t kind, uint32_t payload, uint32_t& first, uint32_t& count)
{
    if (kind == NODE_TRIS) { first = payload & LEAF_FIRST_MASK; count = (payload >> LEAF_FIRST_BITS) + 1u; }
    else { const uint2 lf = bl.leaves[payload]; first = lf.x; count = lf.y; }
}

struct LaneRay { f3 o, d, inv; };

// Ray::transform(inv_matrix)  ray.rs:22-28
//
// Identity instances (every Cornell model) take a shortcut that is bit-identical to the full arithmetic.  glam's
// inverse of the identity has matrix3 = I (+0 zeros) and translation = -0, so for a finite vector v
//   ((1*vx + 0*vy) + 0*vz) [+ -0]  ==  vx                      when vx != 0
//                                  ==  zero signed by sign(vx) & sign(vy) & sign(vz)   when vx == +-0
// (a sum of zeros is -0 only if every addend is -0); 1/v'x is then the world-space reciprocal or +-inf.
//
// `root0` / `root1`: the instance record's copy of the BLAS root node (DInstance::root_node).  EAGER (BVH in global memory): the six
// words a visit needs — matrix rows, meta, root node — are loaded TOGETHER, one memory round trip instead of three dependent ones
// (meta -> rows; meta.root -> root node), and pinned so that the compiler cannot sink the row loads into the branch that uses them.
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <bool EAGER, bool ROOT_BOX>
__device__ __forceinline__ LaneRay to_object(const Blob& bl, uint32_t inst, const LaneRay& w, bool ray_finite, uint4& root0, uint4& root1)
{
    const uint4* ip = bl.inst + INST_WORDS * inst;
    uint4 r0, r1, r2;
    if (EAGER)
    {
        r0 = ip[0]; r1 = ip[1]; r2 = ip[2];
    }
    const uint4 meta = ip[INST_META_WORD];
    if (ROOT_BOX) { root0 = ip[INST_ROOT_WORD]; root1 = ip[INST_ROOT_WORD + 1u]; }
    else root0.w = reinterpret_cast<const uint32_t*>(ip + INST_ROOT_WORD)[3];
    if (EAGER) { pin(r0); pin(r1); pin(r2); }
    LaneRay r;
    if ((meta.w & INSTANCE_IDENTITY) != 0u && ray_finite)
    {
        const uint32_t so = asu(w.o.x) & asu(w.o.y) & asu(w.o.z) & 0x80000000u;
        const uint32_t sd = asu(w.d.x) & asu(w.d.y) & asu(w.d.z) & 0x80000000u;
        r.o.x = w.o.x != 0.0f ? w.o.x : asf(so);
        r.o.y = w.o.y != 0.0f ? w.o.y : asf(so);
        r.o.z = w.o.z != 0.0f ? w.o.z : asf(so);
        r.d.x = w.d.x != 0.0f ? w.d.x : asf(sd);
        r.d.y = w.d.y != 0.0f ? w.d.y : asf(sd);
        r.d.z = w.d.z != 0.0f ? w.d.z : asf(sd);
        r.inv.x = w.d.x != 0.0f ? w.inv.x : asf(sd | 0x7f800000u);
        r.inv.y = w.d.y != 0.0f ? w.inv.y : asf(sd | 0x7f800000u);
        r.inv.z = w.d.z != 0.0f ? w.inv.z : asf(sd | 0x7f800000u);
        return r;
    }
    if (!EAGER) { r0 = ip[0]; r1 = ip[1]; r2 = ip[2]; }
    r.o.x = ((asf(r0.x) * w.o.x + asf(r0.y) * w.o.y) + asf(r0.z) * w.o.z) + asf(r0.w);
    r.o.y = ((asf(r1.x) * w.o.x + asf(r1.y) * w.o.y) + asf(r1.z) * w.o.z) + asf(r1.w);
    r.o.z = ((asf(r2.x) * w.o.x + asf(r2.y) * w.o.y) + asf(r2.z) * w.o.z) + asf(r2.w);
    r.d.x = (asf(r0.x) * w.d.x + asf(r0.y) * w.d.y) + asf(r0.z) * w.d.z;
    r.d.y = (asf(r1.x) * w.d.x + asf(r1.y) * w.d.y) + asf(r1.z) * w.d.z;
    r.d.z = (asf(r2.x) * w.d.x + asf(r2.y) * w.d.y) + asf(r2.z) * w.d.z;
    r.inv = rcp3(r.d);
    return r;
}

template <bool LDS_SCENE>
__device__ __forceinline__ Blob stage_scene(const SceneView& sv, const uint4* __restrict__ gblob, uint4* smem, uint32_t& words)
{
    Blob b;
    const uint4* base = gblob;
    words = 0;
    if (LDS_SCENE)
    {
        words = sv.blob_bytes >> 4;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) smem[i] = gblob[i];
        __syncthreads();
        base = smem;
    }
    b.nodes = base;
    b.tris = base + 2u * sv.n_nodes;
    b.inst = base + 2u * sv.n_nodes + 3u * sv.n_tris;
    b.leaves = reinterpret_cast<const uint2*>(base + 2u * sv.n_nodes + 3u * sv.n_tris + INST_WORDS * sv.n_instances);
    return b;
}

// CLOSEST_PRIMARY = CLOSEST_WORLD for bounce 0: every ray starts at the eye (only directions are stored, ray index == path id)
// and a miss ends the path on the spot (integrator.rs:263-266 with accumulated = 0, path_weight = 1).
enum { CLOSEST_WORLD = 0, CLOSEST_LIGHTS = 1, CLOSEST_HOOK = 2, CLOSEST_PRIMARY = 3 };

struct ClosestOut
{
    f4* hits;              // HOOK: dense by ray index; WORLD/PRIMARY: only rays whose path goes to the terminal queue; LIGHTS: by ray index (slot)
    f4* q_base;            // surface classes' hit records in queue order: see WavefrontBuffers::q_shade_base
    uint32_t q_stride, q_class_slot;
    uint2* q_term;         // terminal queue entries {ray index, path id}
    uint32_t* n_shade;     // counters row: n_shade[Q_COUNT]
    uint4* wave_times;     // PT_WAVE_TIMES builds: [kWaveTimeSlots] records of this launch, else unused
    uint32_t* tails;       // the row's striped tails [Q_COUNT][kTailWordsPerQueue] (surface classes; the terminal queue keeps n_shade[Q_TERMINAL])
    uint32_t cap_shade, cap_term; // queue capacities (slots)
    uint32_t class_mask;   // shade classes present in the scene (bit Q_TERMINAL always set)
    uint32_t* overflow;    // batch-wide "a queue was full" flag
    // CLOSEST_LIGHTS (fused NEE chain): world root for the follow-up any-hit, result codes by path id
    uint32_t world_root;
    uint8_t* occl;
    // CLOSEST_PRIMARY
    f3 eye;
    f4* radiance;
    f4* first_pos;
    uint32_t* first_id;
    // environment-map builds of the frame only (a primary miss is then shaded by the terminal pass, but its id / position defaults are written here):
    // how a path id splits into (sample, pixel) — RenderParams::blk_* — and from which batch-local sample on id / position are kept
    uint32_t keep_s_id, keep_s_pos, blk_log, n_blk, blk_last, act_pixels;
    FastDiv div_blk_paths, div_blk_last;
    uint32_t finalize_miss;               // 0 when an environment map is set: misses then go to the terminal queue like any bounce
    // CLOSEST_WORLD: paths that end at this hit or miss are finished here
    const DPathRec* rec;
    uint32_t enable_nee;
};

// Per-lane traversal stack of (node, t_enter) entries.  The position `sp` is opaque to the traversal loop:
//  * Stack8<false> (the whole stack fits the LDS budget): sp IS the entry's byte offset inside the workgroup's dynamic LDS, so a push
//    or pop is one ds_write_b64 / ds_read_b64 with no address arithmetic beyond the add that moves sp;
//  * Stack8<true> (BVH deeper than PT_STACK_LDS_LEVELS): sp is the level; the deepest levels live in global memory
//    ([level][global lane], coalesced) so that occupancy does not collapse.  LDS and global accesses stay separate instructions
//    (no flat addressing).
template <bool SPILL>
struct Stack8;
template <>
struct Stack8<false>
{
    char* base;             // start of dynamic LDS
    uint32_t bottom, step;  // this lane's level-0 offset; bytes per level (8 * blockDim.x)
    __device__ __forceinline__ static Stack8 make(uint4* smem, uint32_t blob_words, const SceneView&)
    {
        return Stack8{reinterpret_cast<char*>(smem), blob_words * 16u + threadIdx.x * 8u, blockDim.x * 8u};
    }
    __device__ __forceinline__ uint32_t empty() const { return bottom; }
    __device__ __forceinline__ uint32_t up(uint32_t sp) const { return sp + step; }
    __device__ __forceinline__ uint32_t down(uint32_t sp) const { return sp - step; }
    __device__ __forceinline__ uint2 get(uint32_t sp) const { return *reinterpret_cast<const uint2*>(base + sp); }
    __device__ __forceinline__ void put(uint32_t sp, uint2 v) const { *reinterpret_cast<uint2*>(base + sp) = v; }
};
template <>
struct Stack8<true>
{
    uint2* lds;
    uint2* spill;
    uint32_t stride, lds_levels, spill_stride;
    __device__ __forceinline__ static Stack8 make(uint4* smem, uint32_t blob_words, const SceneView& sv)
    {
        return Stack8{reinterpret_cast<uint2*>(smem + blob_words) + threadIdx.x,
                      reinterpret_cast<uint2*>(sv.stack_spill) + (blockIdx.x * blockDim.x + threadIdx.x), blockDim.x, sv.stack_lds, gridDim.x * blockDim.x};
    }
    __device__ __forceinline__ uint32_t empty() const { return 0u; }
    __device__ __forceinline__ uint32_t up(uint32_t sp) const { return sp + 1u; }
    __device__ __forceinline__ uint32_t down(uint32_t sp) const { return sp - 1u; }
    __device__ __forceinline__ uint2 get(uint32_t sp) const
    {
        uint2 v;
        if (sp < lds_levels) v = lds[__umul24(sp, stride)];
        else v = spill[(size_t)(sp - lds_levels) * spill_stride];
        return v;
    }
    __device__ __forceinline__ void put(uint32_t sp, uint2 v) const
    {
        if (sp < lds_levels) lds[__umul24(sp, stride)] = v;
        else spill[(size_t)(sp - lds_levels) * spill_stride] = v;
    }
};

#if PT_WAVE_TIMES
__device__ uint4* g_any_times = nullptr; // where the shadow-ray launch in flight puts its per-wave time records (set in-stream by launch_trace_shadow)
#endif
// ------------------------------------------------------------------------------------------------ closest hit
// a traversal kernel's single argument (one struct, so that it starts at offset 0 of the kernel-argument segment)
struct ClosestKArgs
{
    SceneView sv;
    const uint4* gblob;
    const f4* ra;
    const f4* rb;
    const uint32_t* n_ptr;
    uint32_t* heads;
    uint32_t root, cap_in;
    ClosestOut out;
};
typedef const __attribute__((address_space(4))) ClosestKArgs* ClosestKArgsPtr;
typedef const __attribute__((address_space(4))) ClosestOut* ClosestOutPtr;
// (the empty asm keeps the compiler from hoisting the loads made through the pointer out of the section that makes them)
__device__ __forceinline__ ClosestOutPtr launder_args(ClosestOutPtr p)
{
    asm volatile("" : "+s"(p));
    return p;
}
// (the kernel's body as a function of the staged scene: k_closest is one launch of it, k_trace_fused runs it before another)
template <int BVH, int MODE, bool SPILL>
__device__ __forceinline__ void closest_body(const SceneView& sv, const Blob& bl, const uint32_t blob_words, uint4* smem, const uint32_t root, const f4* __restrict__ ra,
                                             const f4* __restrict__ rb, const uint32_t* __restrict__ n_ptr, const uint32_t cap_in,
                                             uint32_t* __restrict__ heads, const ClosestOutPtr outp)
{
    constexpr bool LDS_SCENE = BVH != 0; // BVH: 0 in global memory, 1 in LDS
    const FetchPlan plan = fetch_plan(min(*n_ptr, cap_in), LDS_SCENE ? (uint32_t)PT_CHUNK_DIV : (uint32_t)PT_CHUNK_DIV_GLOBAL_BVH, (uint32_t)PT_TAPER);
    if (blockIdx.x >= plan.blocks) return; // a short queue keeps only as many workgroups as it has 64-ray chunks
    // per-lane stack of (node, t_enter), [level][thread] in LDS (conflict-free ds_read_b64 / ds_write_b64)
    const Stack8<SPILL> stk = Stack8<SPILL>::make(smem, blob_words, sv);
    const uint32_t prim_bits = sv.prim_bits;

    bool active = false, pending = false, ray_finite = false;
    uint32_t ray_idx = 0, pid = 0;
    LaneRay w{}, ob{};
    float t_max = 0.0f, bt = 0.0f;
    float hud = 0.0f, hvd = 0.0f, hdet = 1.0f; // best hit's (u, v) numerators and determinant: divided once, when the ray retires (primitive.rs:158-160)
    uint32_t bid = MISS_ID, sp = stk.empty(), blas_base = 0, inst = 0;
    bool in_blas = false;
    bool any_phase = false;   // CLOSEST_LIGHTS: the lights-TLAS hit exists, now any-hit against the world (integrator.rs:103)
    uint32_t chain_code = 0u; // 0 light visible, 1 blocked, 2 no light on the ray
    WaveRange wr = first_range(plan, heads);
    // material binning (CLOSEST_WORLD / PRIMARY): a retiring ray's hit record goes straight to its class's shade queue, into a region
    // of that queue this wave has reserved (wave_reserve: one atomic per region, none per entry)
    uint32_t light_hits = 0, valid_rays = 0;
#if PT_STEP_STATS
    uint32_t st_iter = 0, st_lane_active = 0, st_lane_inst = 0, st_lane_branch = 0, st_lane_leaf = 0, st_wave_inst = 0, st_wave_branch = 0, st_wave_leaf = 0;
#endif
#if PT_WAVE_TIMES
    const uint32_t tw_start = (uint32_t)wall_clock64();
    uint32_t tw_first = 0u, tw_drained = 0u;
#endif
    Region bin_region[Q_COUNT];
#pragma unroll
    for (uint32_t c = 0; c < Q_COUNT; ++c) bin_region[c] = Region{0u, 0u};
    // the terminal queue's regions.  Without an environment map its entries are rare (paths that also cast a BSDF-sampled NEE ray): the
    // smallest region a wave can use keeps the queue's extent, and what k_shade_terminal reads, small (whole frame: its first launch
    // 185 -> ~30 us).  With one, every miss goes there: regions sized like any busy queue's.
    // The launch description is read through the kernel-argument segment where it is needed (the service section and the epilogue:
    // s_load, scalar cache) instead of living in scalar registers across the traversal loop, which has none to spare (66 were spilled
    // to vector lanes and read back with v_readlane in every service)
    const uint32_t lights_world_root = MODE == CLOSEST_LIGHTS ? outp->world_root : 0u;
    const uint32_t rsize = outp->finalize_miss != 0u ? 64u : region_size(plan.n, plan.blocks * (blockDim.x >> 6), 64u);
    const Stripes stripes = stripes_for(plan.n);
    uint32_t stripe_rot = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (;;)
    {
        uint64_t act = __ballot(active);
        const bool no_more = wr.drained && wr.cur >= wr.end;
#if PT_WAVE_TIMES
        if (no_more && tw_drained == 0u) tw_drained = (uint32_t)wall_clock64() | 1u;
#endif
        const bool service = no_more ? (act == 0ull) : (__popcll(act) <= Refill<LDS_SCENE>::kBelow);
        if (service)
        {
            const ClosestOutPtr out = launder_args(outp);
            // ---- retire finished lanes
            const uint64_t pm = __ballot(pending);
            if (pm != 0ull)
            {
                if (MODE == CLOSEST_WORLD || MODE == CLOSEST_PRIMARY)
                {
                    uint64_t qm = pm;
                    if (MODE == CLOSEST_PRIMARY)
                    {
                        // a primary miss is a finished path: accumulated = 0 + 0.006 * 1 (integrator.rs:265), defaults of :156-157
                        const bool missed = pending && bid == MISS_ID && out->finalize_miss != 0u;
                        // one byte per camera ray instead of a 16-byte radiance record for the (usually many) rays that leave at once:
                        // k_accumulate reads PRIMARY_MISS as radiance (0.006, 0.006, 0.006)
                        if (pending) out->occl[ray_idx] = missed ? (uint8_t)PRIMARY_MISS : (uint8_t)0u;
                        // (a plain miss leaves nothing else behind: k_accumulate reads PRIMARY_MISS as id 255, position r.at(1e5), integrator.rs:156-157)
                        if (!missed && pending && bid == MISS_ID)
                        {
                            // environment map present: the terminal pass shades the miss; defaults of integrator.rs:156-157 still apply
                            const FastDiv dbp{out->div_blk_paths.magic, out->div_blk_paths.shift, out->div_blk_paths.one, out->div_blk_paths.d};
                            const FastDiv dbl{out->div_blk_last.magic, out->div_blk_last.shift, out->div_blk_last.one, out->div_blk_last.d};
                            const uint32_t b = fastdiv(ray_idx, dbp), rem = ray_idx - b * (out->act_pixels << out->blk_log);
                            const bool short_blk = b + 1u == out->n_blk;
                            const uint32_t k = short_blk ? fastdiv(rem, dbl) : (rem >> out->blk_log);
                            const uint32_t sl = (b << out->blk_log) + (rem - k * (short_blk ? out->blk_last : (1u << out->blk_log)));
                            if (sl >= out->keep_s_id) out->first_id[(sl - out->keep_s_id) * out->act_pixels + k] = 255u;
                            if (sl == out->keep_s_pos)
                            {
                                const f3 far = fma3(w.d, bc3(1e5f), w.o);
                                out->first_pos[k] = f4{far.x, far.y, far.z, 1e5f};
                            }
                        }
                        qm = __ballot(pending && !missed);
                        pending = pending && !missed;
                    }
                    if (MODE == CLOSEST_WORLD)
                    {
                        // A path that ends here — the ray left the scene (integrator.rs:263-266, no environment map) or found a light
                        // (:207-214) — is finished in this kernel, whose memory pipes are idle, instead of a terminal-queue round trip:
                        // add what the last bounce still owes (explicit-light estimate; a blocked shadow ray has zeroed it), then the
                        // ambient term or the emission.  The rare path that also cast a BSDF-sampled NEE ray goes to the queue as before.
                        bool ends = false, emissive = false;
                        uint32_t mat_id = 0;
                        if (pending)
                        {
                            if (bid == MISS_ID) ends = out->finalize_miss != 0u;
                            else
                            {
                                const uint4 meta = bl.inst[INST_WORDS * (bid >> prim_bits) + INST_META_WORD];
                                emissive = ends = (meta.w & 0xffu) == (uint32_t)Q_TERMINAL;
                                mat_id = meta.z;
                            }
                        }
                        if (ends)
                        {
                            const DPathRec& rec = out->rec[pid];
                            const f4 acc4 = rec.acc;
                            const uint32_t flags = asu(acc4.w);
                            if (!(flags & FLAG_BSDF_CAST))
                            {
                                f3 acc = xyz(acc4);
                                if (flags & FLAG_NEE_PENDING) acc = acc + xyz(rec.nee_pw) * (xyz(rec.nee_e) + f3{0.0f, 0.0f, 0.0f}); // integrator.rs:231-234
                                const f3 pw = xyz(rec.pw);
                                if (!emissive) acc = acc + f3{0.006f, 0.006f, 0.006f} * pw;
                                else if (!out->enable_nee || (flags & FLAG_LAST_DELTA))
                                {
                                    const DMaterial& m = sv.materials[mat_id];
                                    acc = fma3(f3{m.colour[0], m.colour[1], m.colour[2]}, pw, acc);
                                }
                                out->radiance[pid] = f4{acc.x, acc.y, acc.z, 0.0f};
                                pending = false;
                            }
                        }
                        qm = __ballot(pending);
                    }
                    if (qm != 0ull)
                    {
                        // bin by shade class.  Surface classes get the whole hit record in queue order (the shading pass then reads
                        // linearly and never gathers by ray index); terminal entries stay {ray index, path id} + hits[ray index].
                        uint32_t cls = Q_COUNT;
                        if (pending) cls = bid != MISS_ID ? (bl.inst[INST_WORDS * (bid >> prim_bits) + INST_META_WORD].w & 0xffu) : (uint32_t)Q_TERMINAL;
                        const f4 hit{bt, hud / hdet, hvd / hdet, asf(bid)};
#pragma unroll
                        for (uint32_t c = 0; c < Q_COUNT; ++c)
                        {
                            if (!((out->class_mask >> c) & 1u)) continue;
                            const uint64_t m = __ballot(cls == c);
                            if (m == 0ull) continue;
                            const Placement pl = c == Q_TERMINAL
                                ? wave_reserve(bin_region[c], out->n_shade + c, (uint32_t)__popcll(m), rsize, out->cap_term, out->overflow)
                                : wave_reserve_striped(bin_region[c], out->tails + c * kTailWordsPerQueue, stripes, stripe_rot, (uint32_t)__popcll(m), out->cap_shade, out->overflow);
                            if (cls == c)
                            {
                                const uint32_t pos = place(pl, mbcnt64(m));
                                if (c == Q_TERMINAL)
                                {
                                    out->hits[ray_idx] = hit;
                                    out->q_term[pos] = make_uint2(ray_idx, pid);
                                }
                                else
                                {
                                    f4* const qa = out->q_base + (size_t)(3u * ((out->q_class_slot >> (4u * c)) & 0xfu)) * out->q_stride + pos;
                                    nt_store(qa, f4{w.d.x, w.d.y, w.d.z, asf(pid)});
                                    nt_store(qa + out->q_stride, hit);
                                    if (MODE != CLOSEST_PRIMARY) nt_store(qa + 2u * (size_t)out->q_stride, f4{w.o.x, w.o.y, w.o.z, 0.0f});
                                }
                            }
                        }
                    }
                }
                else if (MODE == CLOSEST_LIGHTS)
                {
                    if (pending)
                    {
                        out->occl[pid] = (uint8_t)chain_code;
                        if (chain_code == 0u) out->hits[ray_idx] = f4{bt, hud / hdet, hvd / hdet, asf(bid)}; // only a visible light is ever read back
                    }
                }
                else
                {
                    if (pending) out->hits[ray_idx] = f4{bt, hud / hdet, hvd / hdet, asf(bid)};
                }
                pending = false;
            }
            if (no_more) break;
            // ---- refill idle lanes from the wave's private range
            const uint64_t idle = ~act;
            uint32_t first;
            const uint32_t take = claim_rays(wr, heads, plan, (uint32_t)__popcll(idle), first);
            const uint32_t rank = mbcnt64(idle);
            if (!active && rank < take)
            {
                const uint32_t mine = first + rank;
                const f4 b = rb[mine];
                ray_idx = mine;
                pid = asu(b.w);
                if (pid != HOLE) {
                valid_rays += 1u;
                if (MODE == CLOSEST_PRIMARY)
                {
                    w.o = f3{out->eye.x, out->eye.y, out->eye.z};
                    t_max = asf(0x7f800000u);
                }
                else
                {
                    const f4 a = ra[mine]; // (asking for both words of the ray together: +-0, and 8 B more scratch at five waves)
                    w.o = xyz(a);
                    t_max = a.w;
                }
                w.d = xyz(b);
                w.inv = rcp3(w.d);
                ray_finite = finite3(w.o) && finite3(w.d);
                bid = MISS_ID;
                any_phase = false;
                chain_code = 2u;
                bt = asf(0x7f800000u);
                hud = 0.0f;
                hvd = 0.0f;
                hdet = 1.0f;
                in_blas = false;
                sp = stk.empty();
                // TLAS::intersect: root box test, then (root, 0.0)   tlas.rs:68-74
                float te;
                const uint4 root0 = bl.nodes[2u * root];
                const bool ok = (t_max == t_max) && slab(root0, bl.nodes[2u * root + 1u], w.o, w.inv, t_max, te);
                if (ok)
                {
                    stk.put(sp, make_uint2(root0.w, 0u));    // entries are (link, t_enter); the root goes in with t_enter 0
                    sp = stk.up(sp);
                    active = true;
                }
                else { pending = true; }
                } // not a hole
            }
#if PT_WAVE_TIMES
            if (tw_first == 0u && take != 0u) tw_first = (uint32_t)wall_clock64() | 1u;
#endif
            act = __ballot(active);
            if (act == 0ull) continue; // retires the lanes that missed the root box, then refills again or exits
            // the wave's chunk ended inside this refill: go round again at once for the next chunk instead of stepping with idle lanes
            if (PT_RESERVICE && take < (uint32_t)__popcll(idle) && !wr.drained && (uint32_t)__popcll(act) <= (uint32_t)Refill<LDS_SCENE>::kBelow) continue;
        }

#pragma unroll 1
        for (int it = 0; it < Refill<LDS_SCENE>::kSteps; ++it)
        {
#if PT_STEP_STATS
            {
                const uint64_t am = __ballot(active);
                if (am != 0ull) { st_iter += 1u; st_lane_active += (uint32_t)__popcll(am); }
            }
#endif
            if (!active) continue;
            if (in_blas && sp == blas_base) in_blas = false; // BLAS::intersect returned  blas.rs:255
            if (MODE == CLOSEST_LIGHTS && any_phase)
            {
                // TLAS::any_intersect on the world with t_max = light_t * (1 - EPS)   tlas.rs:111-144, blas.rs:257-294; entries are
                // (link, entry distance) of nodes whose box was already met, as in k_any
                if (sp == stk.empty()) { active = false; pending = true; chain_code = 0u; continue; }
                sp = stk.down(sp);
                const uint2 e = stk.get(sp);
                uint32_t link = e.x;
                float t_enter = asf(e.y);
                if ((link >> NODE_KIND_SHIFT) == NODE_INSTANCE)
                {
                    uint4 r0, r1;
                    ob = to_object<!LDS_SCENE, true>(bl, link & NODE_PAYLOAD_MASK, w, ray_finite, r0, r1);
                    in_blas = true;
                    blas_base = sp;
                    if (!slab(r0, r1, ob.o, ob.inv, t_max, t_enter)) continue;
                    link = r0.w;
                }
                const uint32_t kkind = link >> NODE_KIND_SHIFT, kpay = link & NODE_PAYLOAD_MASK;
                if (kkind == NODE_BRANCH)
                {
                    const uint4* cp = bl.nodes + 2u * kpay;
                    const uint4 l0 = cp[0], l1 = cp[1], r0 = cp[2], r1 = cp[3];
                    const f3 so = in_blas ? ob.o : w.o, sinv = in_blas ? ob.inv : w.inv;
                    float tl, tr;
                    const bool hl = slab(l0, l1, so, sinv, t_max, tl);
                    const bool hr = slab(r0, r1, so, sinv, t_max, tr);
                    if (hl) { stk.put(sp, make_uint2(l0.w, asu(tl))); sp = stk.up(sp); }
                    if (hr) { stk.put(sp, make_uint2(r0.w, asu(tr))); sp = stk.up(sp); }
                }
                else
                {
                    uint32_t first, count;
                    leaf_range(bl, kkind, kpay, first, count);
                    for (uint32_t k = 0; k < count; ++k)
                    {
                        const uint4* tp = bl.tris + 3u * (first + k);
                        float td, ud, vd, det;
                        if (tri_planes(tp[0], tp[1], tp[2], ob.o, ob.d, t_max, t_enter, td, ud, vd, det))
                        {
                            active = false;
                            pending = true;
                            chain_code = 1u;
                            break;
                        }
                    }
                }
                continue;
            }
            if (sp == stk.empty())
            {
                if (MODE == CLOSEST_LIGHTS && bid != MISS_ID)
                {
                    // integrator.rs:100-103: the light was hit; the same ray now asks the world for any blocker before it
                    light_hits += 1u;
                    any_phase = true;
                    in_blas = false;
                    t_max = bt * (1.0f - PT_EPSILON);
                    float te = 0.0f;
                    const uint4 wr0 = bl.nodes[2u * lights_world_root];
                    // NaN t_max: every box test fails -> visible; so does a ray that misses the world's root box (tlas.rs:118-121)
                    if (t_max == t_max && slab(wr0, bl.nodes[2u * lights_world_root + 1u], w.o, w.inv, t_max, te)) { stk.put(sp, make_uint2(wr0.w, asu(te))); sp = stk.up(sp); }
                    else { active = false; pending = true; chain_code = 0u; }
                    continue;
                }
                active = false;
                pending = true;
                continue;
            }
            sp = stk.down(sp);
            const uint2 e = stk.get(sp);
            if (asf(e.y) > t_max) continue;                  // tlas.rs:80-83 / blas.rs:222-225
            uint32_t link = e.x;
            float t_est = asf(e.y);
#if PT_STEP_STATS
            { const bool x = (link >> NODE_KIND_SHIFT) == NODE_INSTANCE; const uint64_t m = __ballot(x); st_lane_inst += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_inst += 1u; }
#endif
            if ((link >> NODE_KIND_SHIFT) == NODE_INSTANCE)
            {
                // TLAS leaf: transform the ray, run the BLAS with the current t_max  tlas.rs:88-99.  BLAS::intersect pushes its root
                // with t_enter = 0 and no box test (blas.rs:217) and pops it at once: that pop happens here, in the same step.
                uint4 root0, root1;
                inst = link & NODE_PAYLOAD_MASK;
                ob = to_object<!LDS_SCENE, false>(bl, inst, w, ray_finite, root0, root1);
                in_blas = true;
                blas_base = sp;
                if (0.0f > t_max) continue;                  // the root's pop test  blas.rs:222-225
                link = root0.w;
                t_est = 0.0f;
            }
            uint32_t kind = link >> NODE_KIND_SHIFT, payload = link & NODE_PAYLOAD_MASK;
#if PT_STEP_STATS
            { const bool x = kind == NODE_BRANCH; const uint64_t m = __ballot(x); st_lane_branch += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_branch += 1u; }
#endif
            // push_to_stack  blas.rs:133-162.  A near child that is itself a branch is expanded in the SAME step (up to PT_BRANCH_LEVELS
            // levels) instead of going through the stack: the kernel pays per wave-step far more than per section of a step
            // (profiles/r02_step_stats_cornell.md), and the reference would pop exactly that child next (its pop test t_enter > t_max
            // cannot fire: the box was just met within t_max).
#pragma unroll 1
            for (int lvl = 0; lvl < PT_BRANCH_LEVELS && kind == NODE_BRANCH; ++lvl)
            {
                const uint4* cp = bl.nodes + 2u * payload; // the children are one contiguous 64-byte record pair
                const uint4 l0 = cp[0], l1 = cp[1], r0 = cp[2], r1 = cp[3];
                const f3 o = in_blas ? ob.o : w.o, inv = in_blas ? ob.inv : w.inv;
                float tl, tr;
                const bool hl = slab(l0, l1, o, inv, t_max, tl);
                const bool hr = slab(r0, r1, o, inv, t_max, tr);
                // both hit: the farther child goes underneath (ties: left underneath, right popped first); one hit: that child
                const bool left_near = tl < tr;
                const uint2 le = make_uint2(l0.w, asu(tl)), re = make_uint2(r0.w, asu(tr));
                if (hl && hr) { stk.put(sp, left_near ? re : le); sp = stk.up(sp); }
                kind = NODE_INSTANCE; // nothing more to do in this step unless the near child says otherwise
                if (hl || hr)
                {
                    // the entry the reference pops next.  A triangle leaf is dealt with in this very step, a branch in the next level of
                    // this loop; an instance (or a branch beyond the last level) goes on the stack
                    const uint2 near = (hl && (left_near || !hr)) ? le : re;
                    const uint32_t nk = near.x >> NODE_KIND_SHIFT;
                    if ((nk & 1u) != 0u || (nk == NODE_BRANCH && lvl + 1 < PT_BRANCH_LEVELS))
                    {
                        kind = nk;
                        payload = near.x & NODE_PAYLOAD_MASK;
                        t_est = asf(near.y);
                    }
                    else { stk.put(sp, near); sp = stk.up(sp); }
                }
            }
#if PT_STEP_STATS
            { const bool x = (kind & 1u) != 0u; const uint64_t m = __ballot(x); st_lane_leaf += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_leaf += 1u; }
#endif
            if (kind & 1u)
            {
                uint32_t first, count;
                leaf_range(bl, kind, payload, first, count);
                // blas.rs:230-251, in leaf order, each triangle against the t_max the one before may have lowered.  (Rounds 1-3 evaluated two triangles
                // side by side — -2.2 ms per Cornell frame at four waves per SIMD — which cost 18 VGPRs; without it the kernels fit five waves: round 4.)
                const f3 mo = fma3(ob.d, bc3(t_est), ob.o);  // ray.at(t_estimate)  primitive.rs:150
                const float t_min = PT_EPSILON - t_est;
                auto accept = [&](const TriEval& e, uint32_t tri) {
                    if (e.uv_ok && tri_in_range(e, t_min, t_max - t_est))
                    {
                        // primitive.rs:158-170: (t,u,v) = xyz / det ; t += t_estimate
                        bt = e.td / e.det + t_est;
                        hud = e.ud;
                        hvd = e.vd;
                        hdet = e.det;
                        t_max = bt;                          // a NaN here rejects everything that follows (tri_in_range is unordered)
                        bid = (inst << prim_bits) | tri;
                    }
                };
                for (uint32_t k = 0; k < count; ++k) accept(tri_eval(bl.tris + 3u * (first + k), mo, ob.d), first + k);
                if (bt != bt) { sp = stk.empty(); in_blas = false; } // NaN t_max: nothing else can be accepted anywhere
            }
        }
    }
    if (MODE == CLOSEST_WORLD || MODE == CLOSEST_PRIMARY)
    {
        const ClosestOutPtr out = launder_args(outp);
        // hand back what is left of this wave's regions as holes (a hole is a path id of HOLE)
        const f4 hole{0.0f, 0.0f, 0.0f, asf(HOLE)};
        for (uint32_t i = bin_region[Q_TERMINAL].cur + lane_id(); i < bin_region[Q_TERMINAL].end; i += 64u) out->q_term[i] = make_uint2(HOLE, 0u);
#pragma unroll
        for (uint32_t c = 1; c < Q_COUNT; ++c)
        {
            f4* const qa = out->q_base + (size_t)(3u * ((out->q_class_slot >> (4u * c)) & 0xfu)) * out->q_stride;
            for (uint32_t i = bin_region[c].cur + lane_id(); i < bin_region[c].end; i += 64u) qa[i] = hole;
        }
    }
#if PT_WAVE_TIMES
    // one record per wave (100 MHz ticks, low 32 bits): start, first rays, queue found empty, end; the host reduces them
    // (pt_last_batch_step_stats in a PT_WAVE_TIMES build, tools/wave_times.py)
    if (lane_id() == 0u && outp->wave_times)
    {
        const uint32_t tw_end = (uint32_t)wall_clock64();
        if (tw_drained == 0u) tw_drained = tw_end;
        if (tw_first == 0u) tw_first = tw_start;
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wave < kWaveTimeSlots) outp->wave_times[wave] = make_uint4(tw_start, tw_first, tw_drained, tw_end | 1u);
    }
#endif
#if PT_STEP_STATS
    // words 8..15 of the cursor lines: wave-steps executed, lanes active in them, lanes taking the instance / branch / leaf section, and
    // (words 13..15) wave-steps in which at least one lane took that section; tools/step_stats.py reads these.
    {
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        uint32_t* line = heads + ((wave * 7u) & (kQueueHeads - 1u)) * kHeadStrideWords;
        if (lane_id() == 0u) { atomicAdd(line + 8, st_iter); atomicAdd(line + 9, st_lane_active); }
        uint32_t a = st_lane_inst, b = st_lane_branch, c2 = st_lane_leaf;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); c2 += __shfl_xor(c2, off); }
        if (lane_id() == 0u) { atomicAdd(line + 10, a); atomicAdd(line + 11, b); atomicAdd(line + 12, c2); }
        a = st_wave_inst; b = st_wave_branch; c2 = st_wave_leaf;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); c2 += __shfl_xor(c2, off); }
        if (lane_id() == 0u) { atomicAdd(line + 13, a); atomicAdd(line + 14, b); atomicAdd(line + 15, c2); }
    }
#endif
    if (MODE != CLOSEST_HOOK) add_tally(heads, valid_rays, HEAD_TALLY0);
    // any-hit casts of integrator.rs:103
    if (MODE == CLOSEST_LIGHTS) add_tally(heads, light_hits, HEAD_TALLY1);
}
template <int BVH, int MODE, bool SPILL>
__global__ void __launch_bounds__(256, BVH != 0 ? PT_WAVES_LDS_BVH : PT_WAVES_GLOBAL_BVH) k_closest(const ClosestKArgs a)
{
    constexpr bool LDS_SCENE = BVH != 0;
    extern __shared__ uint4 smem[];
    // (workgroups the queue has no 64-ray chunk for leave before staging anything)
    if (blockIdx.x >= fetch_plan(min(*a.n_ptr, a.cap_in), LDS_SCENE ? (uint32_t)PT_CHUNK_DIV : (uint32_t)PT_CHUNK_DIV_GLOBAL_BVH, (uint32_t)PT_TAPER).blocks) return;
    uint32_t blob_words;
    const Blob bl = stage_scene<LDS_SCENE>(a.sv, a.gblob, smem, blob_words);
    const ClosestOutPtr outp = &((ClosestKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr())->out;
    closest_body<BVH, MODE, SPILL>(a.sv, bl, blob_words, smem, a.root, a.ra, a.rb, a.n_ptr, a.cap_in, a.heads, outp);
}

// ------------------------------------------------------------------------------------------------ any hit
enum { ANY_SHADOW = 0, ANY_HOOK = 2 };

// TLAS::any_intersect / BLAS::any_intersect (tlas.rs:111-144, blas.rs:257-294) pop a node, test ITS box and push both children
// untested.  The answer is a disjunction over the triangles of every leaf whose box the ray meets, each tested with that
// box's own entry distance, so neither the visiting order nor the moment a box is tested can change it.  Here a node's box is
// tested when its parent is expanded (the instance's BLAS root right after the ray transform) and only nodes that were hit go
// on the stack, with their entry distance: a missed child costs a slab test instead of a full traversal step.
template <int BVH, int MODE, bool SPILL>
__device__ __forceinline__ void any_body(const SceneView& sv, const Blob& bl, const uint32_t blob_words, uint4* smem, const uint32_t root, const f4* __restrict__ ra,
                                         const f4* __restrict__ rb, const uint32_t* __restrict__ n_ptr, const uint32_t cap_in,
                                         uint32_t* __restrict__ heads, uint32_t* __restrict__ occluded,
                                         f4* __restrict__ radiance)
{
    constexpr bool LDS_SCENE = BVH != 0; // BVH: 0 in global memory, 1 in LDS
    const FetchPlan plan = fetch_plan(min(*n_ptr, cap_in), LDS_SCENE ? (uint32_t)PT_CHUNK_DIV : (uint32_t)PT_CHUNK_DIV_GLOBAL_BVH, (uint32_t)PT_TAPER_ANY);
    if (blockIdx.x >= plan.blocks) return;
    const Stack8<SPILL> stk = Stack8<SPILL>::make(smem, blob_words, sv); // entries (node, entry distance of its box)

    bool active = false, ray_finite = false;
    uint32_t out_idx = 0, valid_rays = 0;
    LaneRay w{}, ob{};
    float t_max = 0.0f;
    uint32_t sp = stk.empty(), blas_base = 0;
    // ANY_SHADOW: `occluded` is PathState::rec: a blocked shadow ray erases the path's explicit-light candidate (integrator.rs:55-56,73)
    // and a visible one leaves it alone, so the next shading pass needs no separate visibility word; ANY_HOOK: one word per ray
    // A shadow ray whose path id carries PATH_ENDS belongs to a path that died in the shading pass with nothing else owed: its
    // radiance is completed here (accumulated += path_weight * (explicit + 0), integrator.rs:231-234) instead of in a terminal pass.
    bool path_ends = false;
    auto put_result_at = [&](uint32_t idx, bool ends, uint32_t v) {
        if (MODE == ANY_SHADOW)
        {
            DPathRec* rec = reinterpret_cast<DPathRec*>(occluded) + idx;
            if (ends)
            {
                const f3 e = v != 0u ? f3{0.0f, 0.0f, 0.0f} : xyz(rec->nee_e);
                const f3 acc = xyz(rec->acc) + xyz(rec->nee_pw) * (e + f3{0.0f, 0.0f, 0.0f});
                radiance[idx] = f4{acc.x, acc.y, acc.z, 0.0f};
            }
            else if (v != 0u)
            {
                float* e = reinterpret_cast<float*>(&rec->nee_e);
                e[0] = 0.0f; e[1] = 0.0f; e[2] = 0.0f;
            }
        }
        else occluded[idx] = v;
    };
    auto put_result = [&](uint32_t v) { put_result_at(out_idx, path_ends, v); };
    bool in_blas = false;
#if PT_WAVE_TIMES
    const uint32_t tw_start = (uint32_t)wall_clock64();
    uint32_t tw_drained = 0u;
#endif
#if PT_STEP_STATS
    uint32_t st_iter = 0, st_lane_active = 0, st_lane_inst = 0, st_lane_branch = 0, st_lane_leaf = 0, st_wave_inst = 0, st_wave_branch = 0, st_wave_leaf = 0;
#endif
    WaveRange wr = first_range(plan, heads);

    for (;;)
    {
        uint64_t act = __ballot(active);
        const bool no_more = wr.drained && wr.cur >= wr.end;
#if PT_WAVE_TIMES
        if (no_more && tw_drained == 0u) tw_drained = (uint32_t)wall_clock64() | 1u;
#endif
        const bool service = no_more ? (act == 0ull) : (__popcll(act) <= Refill<LDS_SCENE>::kBelowAny);
        if (service)
        {
            if (no_more) break;
            const uint64_t idle = ~act;
            uint32_t first;
            const uint32_t take = claim_rays(wr, heads, plan, (uint32_t)__popcll(idle), first);
            const uint32_t rank = mbcnt64(idle);
            if (!active && rank < take)
            {
                const uint32_t mine = first + rank;
                const f4 a = ra[mine], b = rb[mine];
                const uint32_t tag = asu(b.w);
                if (tag != HOLE) {
                valid_rays += 1u;
                path_ends = MODE == ANY_SHADOW && (tag & PATH_ENDS) != 0u;
                out_idx = (MODE == ANY_HOOK) ? mine : (tag & ~PATH_ENDS);
                w.o = xyz(a);
                w.d = xyz(b);
                w.inv = rcp3(w.d);
                ray_finite = finite3(w.o) && finite3(w.d);
                t_max = a.w;
                in_blas = false;
                sp = stk.empty();
                // the TLAS root's own box (tlas.rs:118-121); a NaN t_max fails every reference box test -> not occluded
                float te;
                const uint4 root0 = bl.nodes[2u * root];
                const bool ok = (t_max == t_max) && slab(root0, bl.nodes[2u * root + 1u], w.o, w.inv, t_max, te);
                if (ok)
                {
                    stk.put(sp, make_uint2(root0.w, asu(te)));
                    sp = stk.up(sp);
                    active = true;
                }
                else put_result(0u);
                } // not a hole
            }
            act = __ballot(active);
            if (act == 0ull) continue;
            if (PT_RESERVICE && take < (uint32_t)__popcll(idle) && !wr.drained && (uint32_t)__popcll(act) <= (uint32_t)Refill<LDS_SCENE>::kBelowAny) continue;
        }

#pragma unroll 1
        for (int it = 0; it < Refill<LDS_SCENE>::kStepsAny; ++it)
        {
#if PT_STEP_STATS
            {
                const uint64_t am = __ballot(active);
                if (am != 0ull) { st_iter += 1u; st_lane_active += (uint32_t)__popcll(am); }
            }
#endif
            if (!active) continue;
            if (in_blas && sp == blas_base) in_blas = false;
            if (sp == stk.empty())
            {
                active = false;
                put_result(0u);
                continue;
            }
            sp = stk.down(sp);
            const uint2 e = stk.get(sp);                     // (link, entry distance) of a node whose box the ray meets
            uint32_t link = e.x;
            float t_enter = asf(e.y);
#if PT_STEP_STATS
            { const bool x = (link >> NODE_KIND_SHIFT) == NODE_INSTANCE; const uint64_t m = __ballot(x); st_lane_inst += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_inst += 1u; }
#endif
            if ((link >> NODE_KIND_SHIFT) == NODE_INSTANCE)
            {
                // TLAS leaf: transform the ray; the BLAS root's box is the first thing BLAS::any_intersect tests  blas.rs:262-264
                uint4 r0, r1;
                ob = to_object<!LDS_SCENE, true>(bl, link & NODE_PAYLOAD_MASK, w, ray_finite, r0, r1);
                in_blas = true;
                blas_base = sp;
                if (!slab(r0, r1, ob.o, ob.inv, t_max, t_enter)) continue;
                link = r0.w;
            }
            uint32_t kind = link >> NODE_KIND_SHIFT, payload = link & NODE_PAYLOAD_MASK;
#if PT_STEP_STATS
            { const bool x = kind == NODE_BRANCH; const uint64_t m = __ballot(x); st_lane_branch += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_branch += 1u; }
#endif
            // up to PT_BRANCH_LEVELS_ANY levels per step: the child that would be popped next (right if met, else left) is expanded or
            // tested at once instead of going through the stack (the kernel pays per wave-step, profiles/r02_step_stats_cornell.md)
#pragma unroll 1
            for (int lvl = 0; lvl < PT_BRANCH_LEVELS_ANY && kind == NODE_BRANCH; ++lvl)
            {
                const uint4* cp = bl.nodes + 2u * payload;
                const uint4 l0 = cp[0], l1 = cp[1], r0 = cp[2], r1 = cp[3];
                const f3 o = in_blas ? ob.o : w.o, inv = in_blas ? ob.inv : w.inv;
                float tl, tr;
                const bool hl = slab(l0, l1, o, inv, t_max, tl);
                const bool hr = slab(r0, r1, o, inv, t_max, tr);
                if (hl && hr) { stk.put(sp, make_uint2(l0.w, asu(tl))); sp = stk.up(sp); }   // left then right: right is popped first
                kind = NODE_INSTANCE; // nothing more in this step unless the next child says otherwise
                if (hl || hr)
                {
                    const uint2 next = hr ? make_uint2(r0.w, asu(tr)) : make_uint2(l0.w, asu(tl));
                    const uint32_t nk = next.x >> NODE_KIND_SHIFT;
                    if ((nk & 1u) != 0u || (nk == NODE_BRANCH && lvl + 1 < PT_BRANCH_LEVELS_ANY))
                    {
                        kind = nk;
                        payload = next.x & NODE_PAYLOAD_MASK;
                        t_enter = asf(next.y);
                    }
                    else { stk.put(sp, next); sp = stk.up(sp); }
                }
            }
#if PT_STEP_STATS
            { const bool x = (kind & 1u) != 0u; const uint64_t m = __ballot(x); st_lane_leaf += x ? 1u : 0u; if (m != 0ull && lane_id() == (uint32_t)__builtin_ctzll(m)) st_wave_leaf += 1u; }
#endif
            if (kind & 1u)
            {
                uint32_t first, count;
                leaf_range(bl, kind, payload, first, count);
                for (uint32_t k = 0; k < count; ++k)         // intersect_bool  primitive.rs:181-189
                {
                    const uint4* tp = bl.tris + 3u * (first + k);
                    float td, ud, vd, det;
                    if (tri_planes(tp[0], tp[1], tp[2], ob.o, ob.d, t_max, t_enter, td, ud, vd, det))
                    {
                        active = false;
                        put_result(1u);
                        break;
                    }
                }
            }
        }
    }
#if PT_STEP_STATS
    // words 8..15 of the shadow queue's cursor lines, as k_closest's (tools/step_stats.py any): wave-steps, lanes active, lanes taking the
    // instance / branch / leaf section, wave-steps in which some lane took it
    if (MODE == ANY_SHADOW)
    {
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        uint32_t* line = heads + ((wave * 7u) & (kQueueHeads - 1u)) * kHeadStrideWords;
        if (lane_id() == 0u) { atomicAdd(line + 8, st_iter); atomicAdd(line + 9, st_lane_active); }
        uint32_t x = st_lane_inst, y = st_lane_branch, z = st_lane_leaf;
        for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off); y += __shfl_xor(y, off); z += __shfl_xor(z, off); }
        if (lane_id() == 0u) { atomicAdd(line + 10, x); atomicAdd(line + 11, y); atomicAdd(line + 12, z); }
        x = st_wave_inst; y = st_wave_branch; z = st_wave_leaf;
        for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off); y += __shfl_xor(y, off); z += __shfl_xor(z, off); }
        if (lane_id() == 0u) { atomicAdd(line + 13, x); atomicAdd(line + 14, y); atomicAdd(line + 15, z); }
    }
#endif
    if (MODE == ANY_SHADOW) add_tally(heads, valid_rays, HEAD_TALLY0);
#if PT_WAVE_TIMES
    if (MODE == ANY_SHADOW && lane_id() == 0u && g_any_times)
    {
        const uint32_t tw_end = (uint32_t)wall_clock64();
        if (tw_drained == 0u) tw_drained = tw_end;
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wave < kWaveTimeSlots) g_any_times[wave] = make_uint4(tw_start, tw_start, tw_drained, tw_end | 1u);
    }
#endif
}
template <int BVH, int MODE, bool SPILL>
__global__ void __launch_bounds__(256, BVH != 0 ? PT_WAVES_LDS_BVH_ANY : PT_WAVES_GLOBAL_BVH_ANY) k_any(const SceneView sv, const uint4* __restrict__ gblob, const uint32_t root, const f4* __restrict__ ra,
                                              const f4* __restrict__ rb, const uint32_t* __restrict__ n_ptr, const uint32_t cap_in,
                                              uint32_t* __restrict__ heads, uint32_t* __restrict__ occluded,
                                              f4* __restrict__ radiance)
{
    constexpr bool LDS_SCENE = BVH != 0;
    extern __shared__ uint4 smem[];
    if (blockIdx.x >= fetch_plan(min(*n_ptr, cap_in), LDS_SCENE ? (uint32_t)PT_CHUNK_DIV : (uint32_t)PT_CHUNK_DIV_GLOBAL_BVH, (uint32_t)PT_TAPER_ANY).blocks) return;
    uint32_t blob_words;
    const Blob bl = stage_scene<LDS_SCENE>(sv, gblob, smem, blob_words);
    any_body<BVH, MODE, SPILL>(sv, bl, blob_words, smem, root, ra, rb, n_ptr, cap_in, heads, occluded, radiance);
}

// One launch for two of the three traversals between two shading passes: the world closest-hit rays of bounce b, then the (few)
// BSDF-sampled NEE rays the shading pass of bounce b - 1 cast.  Neither needs the other's results — the world launch finishes only
// paths that cast no BSDF-sampled ray — so a wave that finds the first queue drained goes on to the second without waiting for
// anybody: the NEE launch's own grid, its launch tail and the side stream's fork / join disappear from every bounce.
// (The shadow rays of bounce b - 1 cannot join them: a path that ends at the world hit still owes its explicit-light estimate, and
// k_closest<WORLD> reads what the shadow-ray launch left of it.  Tried all the same, with such paths sent through the terminal queue
// instead: what the two saved launch tails return, the terminal pass takes.)
struct FusedArgs
{
    const f4 *wa, *wb;          // world closest-hit rays of this bounce
    const uint32_t* wn;
    uint32_t* wheads;
    const f4 *la, *lb;          // BSDF-sampled NEE rays of the bounce before
    const uint32_t* ln;
    uint32_t* lheads;
    uint32_t world_root, lights_root, cap_in;
};
struct FusedKArgs
{
    SceneView sv;
    const uint4* gblob;
    FusedArgs fa;
    ClosestOut wout, lout;
};
template <int BVH, bool SPILL>
__global__ void __launch_bounds__(256, BVH != 0 ? PT_WAVES_LDS_BVH : PT_WAVES_GLOBAL_BVH) k_trace_fused(const FusedKArgs a)
{
    constexpr bool LDS_SCENE = BVH != 0;
    extern __shared__ uint4 smem[];
    const uint32_t cdiv = LDS_SCENE ? (uint32_t)PT_CHUNK_DIV : (uint32_t)PT_CHUNK_DIV_GLOBAL_BVH;
    const uint32_t need = max(fetch_plan(min(*a.fa.wn, a.fa.cap_in), cdiv, (uint32_t)PT_TAPER).blocks, fetch_plan(min(*a.fa.ln, a.fa.cap_in), cdiv, (uint32_t)PT_TAPER).blocks);
    if (blockIdx.x >= need) return;
    uint32_t blob_words;
    const Blob bl = stage_scene<LDS_SCENE>(a.sv, a.gblob, smem, blob_words);
    typedef const __attribute__((address_space(4))) FusedKArgs* FusedKArgsPtr;
    const FusedKArgsPtr k = (FusedKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    closest_body<BVH, CLOSEST_WORLD, SPILL>(a.sv, bl, blob_words, smem, a.fa.world_root, a.fa.wa, a.fa.wb, a.fa.wn, a.fa.cap_in, a.fa.wheads, &k->wout);
    closest_body<BVH, CLOSEST_LIGHTS, SPILL>(a.sv, bl, blob_words, smem, a.fa.lights_root, a.fa.la, a.fa.lb, a.fa.ln, a.fa.cap_in, a.fa.lheads, &k->lout);
}

// ------------------------------------------------------------------------------------------------ path bookkeeping
struct PixelId { uint32_t gpixel, sample, lpixel, gx, gy; };

__device__ __forceinline__ uint32_t global_row(const RenderParams& rp, uint32_t ly)
{
    const uint32_t strip = fastdiv(ly, rp.div_strip_rows);
    return (strip * rp.world_size + rp.rank) * rp.strip_rows + (ly - strip * rp.strip_rows);
}
// path id <-> (batch-local sample, index of the pixel in the active rectangle): RenderParams::blk_log
struct PidParts { uint32_t s, k; };
__device__ __forceinline__ PidParts pid_split(const RenderParams& rp, uint32_t pid)
{
    const uint32_t b = fastdiv(pid, rp.div_blk_paths), rem = pid - b * (rp.act_pixels << rp.blk_log);
    const bool short_blk = b + 1u == rp.n_blk;   // (the last block; blk_last == 1 << blk_log when it is a full one: both branches then agree)
    const uint32_t k = short_blk ? fastdiv(rem, rp.div_blk_last) : (rem >> rp.blk_log);
    return PidParts{(b << rp.blk_log) + (rem - k * (short_blk ? rp.blk_last : (1u << rp.blk_log))), k};
}
__device__ __forceinline__ uint32_t pid_join(const RenderParams& rp, uint32_t s, uint32_t k)
{
    const uint32_t b = s >> rp.blk_log;
    return b * (rp.act_pixels << rp.blk_log) + k * (b + 1u == rp.n_blk ? rp.blk_last : (1u << rp.blk_log)) + (s - (b << rp.blk_log));
}
__device__ __forceinline__ PixelId path_pixel(const RenderParams& rp, uint32_t pid, uint32_t* s_local = nullptr, uint32_t* k_out = nullptr)
{
    const PidParts pp = pid_split(rp, pid);
    if (s_local) *s_local = pp.s;
    if (k_out) *k_out = pp.k;
    const uint32_t r = fastdiv(pp.k, rp.div_act_w), x = rp.act_x0 + (pp.k - r * rp.act_w);
    const uint32_t ly = rp.act_ly0 + r;
    const uint32_t gy = global_row(rp, ly);
    return PixelId{gy * rp.width + x, rp.first_sample + pp.s, ly * rp.width + x, x, gy};
}

// direction of the camera ray of (pixel gx, gy; sample): main.rs:193-199 + Camera::create_ray camera.rs:94-105
__device__ __forceinline__ f3 camera_ray_dir(const RenderParams& rp, const CameraView& cam, uint32_t gx, uint32_t gy, uint32_t sample)
{
    Stream rng{stream_key(rp.seed, gy * rp.width + gx, sample), 0u};
    const uint32_t seed = rng.u32();                                       // main.rs:193 (the stream's draw 0)
    float jx, jy;
    ss_sobol(rp.n_sobol, sample, seed, &jx, &jy);                          // main.rs:194
    const float ox = jx - 0.5f, oy = jy - 0.5f;
    const float u = ((float)gx + ox) / (float)rp.width;                    // main.rs:196
    const float v = ((float)gy + oy) / (float)rp.height;                   // main.rs:197
    // Camera::create_ray  camera.rs:94-105  (Mat4::project_point3, then normalise)
    const float nx = u * 2.0f - 1.0f, ny = v * 2.0f - 1.0f, nz = 0.0f;
    const float* M = cam.ray_matrix;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        float t = M[i] * nx;
        t = M[4 + i] * ny + t;
        t = M[8 + i] * nz + t;
        t = M[12 + i] + t;
        r[i] = t;
    }
    const float rw = 1.0f / r[3];
    const f3 eye{cam.eye[0], cam.eye[1], cam.eye[2]};
    return unit3(f3{r[0] * rw, r[1] * rw, r[2] * rw} - eye);
}

// main.rs:186-199
__global__ void __launch_bounds__(256) k_generate(const RenderParams rp, const CameraView cam, const RayQueue rq, Counters* ctr)
{
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid == 0u) ctr[0].n_closest = rp.n_paths;
    if (pid >= rp.n_paths) return;
    const PixelId px = path_pixel(rp, pid);
    const f3 dir = camera_ray_dir(rp, cam, px.gx, px.gy, px.sample);
    rq.b[pid] = f4{dir.x, dir.y, dir.z, asf(pid)};
    // no state is initialised: bounce 0 knows path_weight = 1, accumulated = 0, one draw consumed (integrator.rs:153-161)
}

struct ShadeIO
{
    PathState st;
    RayQueue rq_in, rq_out, rq_shadow, rq_lchain_prev, rq_lchain;
    f4* lchain_nb;            // this bounce's BSDF-sampled rays: bsdf rgb | weakening, by rq_lchain slot
    const f4* lchain_nb_prev; // last bounce's
    const f4* lchain_hit;     // last bounce's light hits, by rq_lchain_prev slot
    const f4* hits;           // terminal pass only: hits of the rays that went to the terminal queue, by ray index
    const uint2* entries;     // terminal pass: {ray index | ENTRY_DEAD, path id}
    ShadeQueue q_in;          // surface pass: this class's hit records in queue order
    const uint32_t* tails_in; // ... and the queue's striped tails
    uint2* q_term_next;
    uint32_t cap_slots, cap_slots_term, cap_slots_shade; // queue capacities (slots)
    Counters* ctr;     // row of this bounce
    uint32_t* lchain_heads; // cursor lines of this bounce's BSDF-sampled NEE queue (the culled tally goes there)
    uint32_t* shadow_heads; // cursor lines of this bounce's shadow-ray queue (INLINE: the tally of shadow rays traced by the shading pass itself)
    Counters* ctr_next;
    f4 primary_a;      // bounce 0: origin.xyz | +inf of every primary ray
    EnvView env;       // equirect environment for misses (w == 0: constant ambient, integrator.rs:263-266)
};

// MIS power heuristic  integrator.rs:22
__device__ __forceinline__ float mis2(float f, float g) { return sq(f) / (sq(f) + sq(g)); }

// shading normal of a hit: Triangle::get_normal + face-forward in object space (primitive.rs:57-63,161-165),
// then the deferred instance transform (tlas.rs:105)
__device__ __forceinline__ f3 hit_normal(const SceneView& sv, uint32_t inst, uint32_t tri, float u, float v, f3 dir_world, bool& front)
{
    const DTriVerts tv = sv.tri_shade[tri];
    const float wgt = 1.0f - u - v;
    const m33 nm{xyz(tv.a), xyz(tv.b), xyz(tv.c)};
    f3 n = unit3(mul(nm, f3{wgt, u, v}));
    const DInstance& in = sv.instances[inst];
    const f3 d_obj{(in.inv[0] * dir_world.x + in.inv[1] * dir_world.y) + in.inv[2] * dir_world.z,
                   (in.inv[4] * dir_world.x + in.inv[5] * dir_world.y) + in.inv[6] * dir_world.z,
                   (in.inv[8] * dir_world.x + in.inv[9] * dir_world.y) + in.inv[10] * dir_world.z};
    front = dot3(d_obj, n) < 0.0f;
    if (!front) n = -n;
    return f3{(in.fwd[0] * n.x + in.fwd[1] * n.y) + in.fwd[2] * n.z, (in.fwd[4] * n.x + in.fwd[5] * n.y) + in.fwd[6] * n.z,
              (in.fwd[8] * n.x + in.fwd[9] * n.y) + in.fwd[10] * n.z};
}

// add the previous bounce's direct-light estimate: accumulated += path_weight * (explicit + bsdf)   integrator.rs:231-234
__device__ __forceinline__ void resolve_nee(const SceneView& sv, const ShadeIO& io, uint32_t pid, const DPathRec& rec, f3& acc, uint32_t& flags)
{
    if (!(flags & FLAG_NEE_PENDING)) return;
    const f4 e4 = rec.nee_e;
    const f4 pw4 = rec.nee_pw;
    f3 e = xyz(e4);
    // (a blocked explicit shadow ray has already zeroed nee_e in the record: k_any<SHADOW>)
    f3 s{0.0f, 0.0f, 0.0f};
    if (flags & FLAG_BSDF_CAST)
    {
        if (io.st.occl[pid] == 0u && pw4.w > 0.0f)                       // integrator.rs:100,103,108 (0 = light hit and visible)
        {
            const uint32_t slot = asu(e4.w); // of the BSDF-sampled ray in last bounce's rq_lchain
            const f4 lh = io.lchain_hit[slot];
            const uint32_t lid = asu(lh.w);
            const f4 b4 = io.lchain_nb_prev[slot];
            const uint32_t inst = lid >> sv.prim_bits, tri = lid & ((1u << sv.prim_bits) - 1u);
            const DInstance& in = sv.instances[inst];
            const DMaterial& lm = sv.materials[in.material];
            const f3 emitted{lm.colour[0], lm.colour[1], lm.colour[2]};
            const DTriIsect ti = sv.tri_isect[tri];
            const float area = 0.5f * len3(xyz(ti.n0));                    // primitive.rs:94
            const float sample_pdf = (area * len3(emitted) / sv.light_weight_sum) / area; // light_sampler.rs:39, integrator.rs:111
            const f3 dir = xyz(io.rq_lchain_prev.b[slot]);
            bool ff;
            const f3 ln = hit_normal(sv, inst, tri, lh.y, lh.z, dir, ff);
            const float cosine = fabsf(dot3(dir, ln));
            const float light_pdf = sample_pdf * (lh.x * lh.x / cosine);   // integrator.rs:115
            const float weight = mis2(pw4.w, light_pdf);
            s = emitted * weight * b4.w * xyz(b4) / pw4.w;                 // integrator.rs:119-123
        }
    }
    acc = acc + xyz(pw4) * (e + s);
    flags &= ~(FLAG_NEE_PENDING | FLAG_BSDF_CAST);
}

// ------------------------------------------------------------------------------------------------ shading
// Q_TERMINAL: misses (integrator.rs:254-269), emissive hits (:207-214) and paths that already ended but still owe an NEE resolve.
__global__ void __launch_bounds__(256) k_shade_terminal(const SceneView sv, const RenderParams rp, const ShadeIO io, const uint32_t bounce)
{
    const uint32_t n = min(io.ctr->n_shade[Q_TERMINAL], io.cap_slots_term);
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x)
    {
        const uint2 e2 = io.entries[idx];
        const uint32_t entry = e2.x, pid = e2.y;
        if (entry == HOLE) continue;
        const bool dead = (entry & ENTRY_DEAD) != 0u;
        f4 ra{}, rb{}, hit{};
        if (!dead)
        {
            hit = io.hits[entry];
            if (asu(hit.w) != MISS_ID) // a miss needs no origin, and its direction only for the environment lookup
            {
                ra = bounce == 0u ? io.primary_a : io.rq_in.a[entry];
                rb = io.rq_in.b[entry];
            }
            else if (io.env.w != 0u) rb = io.rq_in.b[entry];
        }
        f3 acc{0.0f, 0.0f, 0.0f};
        uint32_t flags = 0u;
        if (bounce != 0u)
        {
            const DPathRec& rec = io.st.rec[pid];
            const f4 acc4 = rec.acc;
            acc = xyz(acc4);
            flags = asu(acc4.w);
            resolve_nee(sv, io, pid, rec, acc, flags);
        }
        if (!dead)
        {
            const f3 pw = bounce == 0u ? f3{1.0f, 1.0f, 1.0f} : xyz(io.st.rec[pid].pw);
            const uint32_t hid = asu(hit.w);
            if (hid == MISS_ID)
            {
                if (io.env.w != 0u) acc = acc + env_lookup(io.env, xyz(rb)) * pw;      // integrator.rs:256-262
                else acc = acc + f3{0.006f, 0.006f, 0.006f} * pw;                      // integrator.rs:263-266
            }
            else
            {
                const uint32_t inst = hid >> sv.prim_bits;
                const DInstance& in = sv.instances[inst];
                if (bounce == 0u)                                                     // integrator.rs:181-185
                {
                    const PidParts pp = pid_split(rp, pid);
                    if (pp.s == rp.keep_s_pos)
                    {
                        const f3 p = fma3(xyz(rb), bc3(hit.x), xyz(ra));
                        io.st.first_pos[pp.k] = f4{p.x, p.y, p.z, hit.x};
                    }
                    if (pp.s >= rp.keep_s_id) io.st.first_id[(pp.s - rp.keep_s_id) * rp.act_pixels + pp.k] = in.blas & 0xffu;
                }
                const DMaterial& m = sv.materials[in.material];
                if (!rp.enable_nee || (flags & FLAG_LAST_DELTA) || bounce == 0u)       // integrator.rs:209-212
                    acc = fma3(f3{m.colour[0], m.colour[1], m.colour[2]}, pw, acc);
            }
        }
        io.st.radiance[pid] = f4{acc.x, acc.y, acc.z, 0.0f}; // every entry of this queue is a finished path
    }
}

// TLAS::any_intersect for ONE ray per lane, run by the whole wave until its last ray is done (no refill): the shading pass of an LDS-resident scene
// answers its own explicit-light shadow ray with it (k_shade_surface<.., INLINE>).  Steps as in any_body: a node's box is tested when its parent is
// expanded, only nodes that were met go on the stack; the answer is a disjunction, so the order cannot change it.
__device__ __forceinline__ bool inline_any(const Blob& bl, const Stack8<false>& stk, const uint32_t root, const f4 ra, const f4 rb, const bool want)
{
    bool active = false, blocked = false, in_blas = false, ray_finite = false;
    LaneRay w{}, ob{};
    const float t_max = ra.w;
    uint32_t sp = stk.empty(), blas_base = 0u;
    if (want)
    {
        w.o = xyz(ra);
        w.d = xyz(rb);
        w.inv = rcp3(w.d);
        ray_finite = finite3(w.o) && finite3(w.d);
        float te;
        const uint4 root0 = bl.nodes[2u * root];
        if ((t_max == t_max) && slab(root0, bl.nodes[2u * root + 1u], w.o, w.inv, t_max, te))
        {
            stk.put(sp, make_uint2(root0.w, asu(te)));
            sp = stk.up(sp);
            active = true;
        }
    }
    while (__ballot(active) != 0ull)
    {
        if (!active) continue;
        if (in_blas && sp == blas_base) in_blas = false;
        if (sp == stk.empty()) { active = false; continue; }
        sp = stk.down(sp);
        const uint2 e = stk.get(sp);
        uint32_t link = e.x;
        float t_enter = asf(e.y);
        if ((link >> NODE_KIND_SHIFT) == NODE_INSTANCE)
        {
            uint4 r0, r1;
            ob = to_object<false, true>(bl, link & NODE_PAYLOAD_MASK, w, ray_finite, r0, r1);
            in_blas = true;
            blas_base = sp;
            if (!slab(r0, r1, ob.o, ob.inv, t_max, t_enter)) continue;
            link = r0.w;
        }
        uint32_t kind = link >> NODE_KIND_SHIFT, payload = link & NODE_PAYLOAD_MASK;
#pragma unroll 1
        for (int lvl = 0; lvl < PT_BRANCH_LEVELS_INLINE && kind == NODE_BRANCH; ++lvl)
        {
            const uint4* cp = bl.nodes + 2u * payload;
            const uint4 l0 = cp[0], l1 = cp[1], r0 = cp[2], r1 = cp[3];
            const f3 o = in_blas ? ob.o : w.o, inv = in_blas ? ob.inv : w.inv;
            float tl, tr;
            const bool hl = slab(l0, l1, o, inv, t_max, tl);
            const bool hr = slab(r0, r1, o, inv, t_max, tr);
            if (hl && hr) { stk.put(sp, make_uint2(l0.w, asu(tl))); sp = stk.up(sp); }
            kind = NODE_INSTANCE;
            if (hl || hr)
            {
                const uint2 next = hr ? make_uint2(r0.w, asu(tr)) : make_uint2(l0.w, asu(tl));
                const uint32_t nk = next.x >> NODE_KIND_SHIFT;
                if ((nk & 1u) != 0u || (nk == NODE_BRANCH && lvl + 1 < PT_BRANCH_LEVELS_INLINE))
                {
                    kind = nk;
                    payload = next.x & NODE_PAYLOAD_MASK;
                    t_enter = asf(next.y);
                }
                else { stk.put(sp, next); sp = stk.up(sp); }
            }
        }
        if (kind & 1u)
        {
            uint32_t first, count;
            leaf_range(bl, kind, payload, first, count);
            for (uint32_t k = 0; k < count; ++k)
            {
                const uint4* tp = bl.tris + 3u * (first + k);
                float td, ud, vd, det;
                if (tri_planes(tp[0], tp[1], tp[2], ob.o, ob.d, t_max, t_enter, td, ud, vd, det))
                {
                    active = false;
                    blocked = true;
                    break;
                }
            }
        }
    }
    return blocked;
}

// Surface classes.  One kernel per queue class; RNG draws in program order of integrator.rs:231-251.
// The launch description as ONE kernel argument (offset 0 of the kernel-argument segment), so that a section of the kernel can read its
// fields from there when it runs (shade_args) instead of holding them in scalar registers across the whole iteration.
struct ShadeKArgs { SceneView sv; RenderParams rp; ShadeIO io; uint32_t bounce; const uint4* gblob; uint32_t world_root; };
typedef const __attribute__((address_space(4))) ShadeKArgs* ShadeKArgsPtr;
__device__ __forceinline__ const ShadeKArgs& shade_args()
{
    ShadeKArgsPtr p = (ShadeKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));        // loads through p cannot move above this point
    return *(const ShadeKArgs*)p;
}
#define PT_SHADE_ARGS                                                                                                              \
    const ShadeKArgs& ka_ = shade_args();                                                                                          \
    [[maybe_unused]] const SceneView& sv = ka_.sv;                                                                                 \
    [[maybe_unused]] const RenderParams& rp = ka_.rp;                                                                              \
    [[maybe_unused]] const ShadeIO& io = ka_.io;
template <uint32_t QCLASS, bool VOLUMES, bool INLINE = false>
__global__ void __launch_bounds__(PT_SHADE_THREADS, INLINE ? PT_SHADE_WAVES_INLINE : shade_waves(QCLASS, VOLUMES)) k_shade_surface(const ShadeKArgs kargs)
{
    extern __shared__ uint4 smem_dyn[];
    // Only what the loop header needs is taken from the argument here; each section of an iteration re-reads the launch description from
    // the kernel-argument segment (PT_SHADE_ARGS: scalar loads that hit the constant cache) instead of keeping ~130 words of it in ~100
    // scalar registers for the whole iteration: spilled scalars 81 -> 0 (every class), static VALU 1947 -> 1747 (no v_readlane / v_writelane).
    const ShadeIO& io = kargs.io;
    const uint32_t bounce = kargs.bounce;
    __shared__ BlockAppend sh_append;
    __shared__ uint32_t sh_tail[kTailStripes];
    __shared__ uint32_t sh_extent;
    // the queue's extent and the regions the shorter stripes never reached (stripe_valid); keyed like the producer's (k_closest of this bounce)
    const Stripes stripes = stripes_for(min(io.ctr->n_closest, io.cap_slots));
    if (threadIdx.x < kTailStripes)
    {
        uint32_t mine;
        const uint32_t ext = stripe_load(io.tails_in, stripes, io.cap_slots_shade, mine);
        sh_tail[threadIdx.x] = mine;
        if (threadIdx.x == 0u)
        {
            sh_extent = ext;
            if (blockIdx.x == 0u) io.ctr->n_shade[QCLASS] = ext; // for the host's per-bounce table only
        }
    }
    // INLINE: the BVH blob and the per-lane stacks in dynamic LDS, as in the traversal kernels (stage_scene ends with the barrier)
    if (INLINE) { uint32_t words; (void)stage_scene<true>(kargs.sv, kargs.gblob, smem_dyn, words); }
    else __syncthreads();
    uint32_t traced = 0u; // INLINE: wave-uniform (a scalar register): shadow rays this wave has answered
    const uint32_t n = sh_extent;
    const uint32_t total = ((n + blockDim.x - 1u) / blockDim.x) * blockDim.x; // whole blocks take part in the queue appends
    uint32_t culled = 0;
    // Everything this pass needs of a hit arrives in queue order (ShadeQueue): direction | path id, t u v | hit id, origin.  The first
    // word of the NEXT iteration is fetched one iteration ahead, so the path-record load that depends on its path id starts as soon
    // as an iteration begins instead of one memory round trip later.
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const f4 hole_a{0.0f, 0.0f, 0.0f, asf(HOLE)};
    f4 a_next = idx < n ? nt_load(io.q_in.a + idx) : hole_a;
    for (; idx < total; idx += stride)
    {
        const f4 rb = a_next;
        if (!INLINE) a_next = (idx + stride) < n ? nt_load(io.q_in.a + idx + stride) : hole_a; // (INLINE: requested after the shadow walk, which wants the registers)
        bool valid = idx < n && stripe_valid(sh_tail, stripes, idx) && asu(rb.w) != HOLE;
        bool want_shadow = false, want_lchain = false, want_next = false, want_dead = false, ends_with_shadow = false, cull_now = false;
        f4 sh_a{}, sh_b{}, lc_a{}, lc_b{}, nx_a{}, nx_b{};
        uint32_t pid = 0, flags = 0;
        f3 acc{}, pw{};
        f4 nee_e{}, nee_pw{}, nee_b{};
        uint32_t draws = 0;
        if (valid)
        {
            PT_SHADE_ARGS
            pid = asu(rb.w);
            const f4 hit = nt_load(io.q_in.b + idx);
            const f4 ra = bounce == 0u ? io.primary_a : nt_load(io.q_in.c + idx);
            f4 pw4{1.0f, 1.0f, 1.0f, asf(1u)}; // bounce 0: path_weight = 1, accumulated = 0, the seed draw consumed
            acc = f3{0.0f, 0.0f, 0.0f};
            flags = 0u;
            if (bounce != 0u)
            {
                const DPathRec& rec = io.st.rec[pid];
                const f4 acc4 = rec.acc;
                pw4 = rec.pw;
                acc = xyz(acc4);
                flags = asu(acc4.w);
                resolve_nee(sv, io, pid, rec, acc, flags);
            }
            pw = xyz(pw4);

            const f3 ro = xyz(ra), rd = xyz(rb);
            const uint32_t hid = asu(hit.w);
            const uint32_t inst = hid >> sv.prim_bits, tri = hid & ((1u << sv.prim_bits) - 1u);
            const DInstance& in = sv.instances[inst];
            MatView mat = load_material(sv.materials, in.material);
            // the queue class fixes the material kind: let the compiler drop the other materials' code from this kernel
            if (QCLASS == Q_LAMBERT) mat.kind = (VOLUMES && mat.kind == MAT_EMISSIVE) ? (uint32_t)MAT_EMISSIVE : (uint32_t)MAT_LAMBERTIAN;
            else if (QCLASS == Q_SPECULAR) mat.kind = MAT_SPECULAR;
            else if (QCLASS == Q_DIELECTRIC) mat.kind = MAT_DIELECTRIC;
            else mat.kind = mat.kind == MAT_GGX_DIELECTRIC ? (uint32_t)MAT_GGX_DIELECTRIC : (uint32_t)MAT_GGX_METAL;
            bool front;
            const f3 normal = hit_normal(sv, inst, tri, hit.y, hit.z, rd, front);
            const f3 p = fma3(rd, bc3(hit.x), ro);                                     // r.at(hit_info.t)
            uint32_t s_local, k_pix;
            const PixelId px = path_pixel(rp, pid, &s_local, &k_pix);
            if (bounce == 0u)                                                          // integrator.rs:181-185
            {
                if (s_local == rp.keep_s_pos) io.st.first_pos[k_pix] = f4{p.x, p.y, p.z, hit.x};
                if (s_local >= rp.keep_s_id) io.st.first_id[(s_local - rp.keep_s_id) * rp.act_pixels + k_pix] = in.blas & 0xffu;
            }
            Stream rng{stream_key(rp.seed, px.gpixel, px.sample), asu(pw4.w)};
            const f3 wi = -rd;                                                         // integrator.rs:187
            const bool is_delta = mat_is_delta(mat.kind);

            // ---- participating media  integrator.rs:189-205: free-flight scattering in every volume the path is inside
            bool scattered = false;
            float s_t = 0.0f;
            f3 s_dir{0.0f, 0.0f, 0.0f};
            uint32_t vst = 0xffffffffu, vst_in = 0xffffffffu;
            if (VOLUMES)
            {
                if (bounce != 0u) vst = io.st.vstack[pid];
                vst_in = vst;
                if (vst != 0xffffffffu)
                {
                    for (uint32_t slot = 0; slot < 4u; ++slot)
                    {
                        const uint32_t vm = (vst >> (8u * slot)) & 0xffu;
                        if (vm == 0xffu) continue;
                        const DMaterial& dm = sv.materials[vm];
                        if (dm.vol_flags & 2u)                                         // VolumeScatter::scatter  volume.rs:83-97
                        {
                            const float t = -ln_det(rng.f32()) / dm.vol_c;
                            if (!(t > hit.x))
                            {
                                const f3 d = hg_direction(dm.vol_g, rng, rd);
                                if (!scattered || total_order_key(t) < total_order_key(s_t)) { s_t = t; s_dir = d; } // min_by(total_cmp), first wins
                                scattered = true;
                            }
                        }
                    }
                    const float dist = scattered ? s_t : hit.x;
                    f3 wgt{1.0f, 1.0f, 1.0f};
                    for (uint32_t slot = 0; slot < 4u; ++slot)
                    {
                        const uint32_t vm = (vst >> (8u * slot)) & 0xffu;
                        if (vm == 0xffu) continue;
                        const DMaterial& dm = sv.materials[vm];
                        if (dm.vol_flags & 1u) wgt = wgt * beer_lambert(f3{dm.vol_abs[0], dm.vol_abs[1], dm.vol_abs[2]}, dist);
                    }
                    pw = pw * wgt;
                }
            }
            f3 ndir{0.0f, 0.0f, 0.0f}, next_o = p;
            bool alive = true;
            if (scattered)
            {
                // integrator.rs:196-200: the path continues from inside the medium; no surface interaction this iteration
                flags |= FLAG_LAST_DELTA;
                next_o = fma3(rd, bc3(s_t), ro);
                ndir = s_dir;
            }
            else if (mat.kind == MAT_EMISSIVE)
            {
                // only reached when the scene has volumes (emissive hits are then shaded here, after the media)  integrator.rs:207-214
                if (!rp.enable_nee || (flags & FLAG_LAST_DELTA) || bounce == 0u) acc = fma3(mat.colour, pw, acc);
                alive = false;
            }
            else
            {
            if (VOLUMES && sv.materials[in.material].has_volume != 0u)     // integrator.rs:217-227
            {
                const uint32_t me = in.material & 0xffu;
                int found = -1, empty = -1;
                for (int slot = 0; slot < 4; ++slot)
                {
                    const uint32_t vm = (vst >> (8 * slot)) & 0xffu;
                    if (vm == me && found < 0) found = slot;
                    if (vm == 0xffu && empty < 0) empty = slot;
                }
                if (front) { if (found < 0 && empty >= 0) vst = (vst & ~(0xffu << (8 * empty))) | (me << (8 * empty)); }
                else if (found >= 0)
                {
                    // remove and close the gap so that insertion order is kept
                    uint32_t out_v = 0xffffffffu;
                    int k = 0;
                    for (int slot = 0; slot < 4; ++slot)
                    {
                        const uint32_t vm = (vst >> (8 * slot)) & 0xffu;
                        if (slot == found || vm == 0xffu) continue;
                        out_v = (out_v & ~(0xffu << (8 * k))) | (vm << (8 * k));
                        ++k;
                    }
                    vst = out_v;
                }
            }
            if (rp.enable_nee && !is_delta)                                            // integrator.rs:231
            {
                // ---- estimate_direct_explicit  integrator.rs:25-74
                {
                    PT_SHADE_ARGS
                    const float x = rng.f32();                                         // light_sampler.rs:33
                    // binary_search_by(total_cmp): for a non-decreasing cdf Ok(i)/Err(i) = number of entries below x
                    uint32_t li = 0, hi = sv.n_lights;
                    const int32_t xk = total_order_key(x);
                    while (li < hi)
                    {
                        const uint32_t mid = (li + hi) >> 1;
                        if (total_order_key(sv.lights[mid].cdf) < xk) li = mid + 1u; else hi = mid;
                    }
                    if (li >= sv.n_lights) li = sv.n_lights - 1u;                       // clamp (reference would index out of bounds)
                    const DLight L = sv.lights[li];
                    float lu = rng.f32();                                              // primitive.rs:81-88
                    float lv = rng.f32();
                    if (lu + lv > 1.0f) { lu = 1.0f - lu; lv = 1.0f - lv; }
                    const float lw = 1.0f - lu - lv;
                    const DTriVerts lp = sv.tri_pos[L.tri], ln = sv.tri_shade[L.tri];
                    const f3 point = mul(m33{xyz(lp.a), xyz(lp.b), xyz(lp.c)}, f3{lw, lu, lv});
                    const f3 lnormal = unit3(mul(m33{xyz(ln.a), xyz(ln.b), xyz(ln.c)}, f3{lw, lu, lv}));
                    const f3 d = point - p;
                    const float dist2 = len_sq(d);
                    const float dist = sqrtf(dist2);
                    const f3 dir = unit3(d);
                    f3 ce{0.0f, 0.0f, 0.0f};
                    if (dot3(dir, normal) > 0.0f)                                      // integrator.rs:55
                    {
                        const BsdfSample bp = mat_bsdf_pdf(mat, wi, dir, normal, front);
                        const DTriIsect ti = sv.tri_isect[L.tri];
                        const float sample_pdf = L.pdf / (0.5f * len3(xyz(ti.n0)));    // integrator.rs:61
                        const float cosine = fabsf(dot3(dir, lnormal));
                        const float light_pdf = sample_pdf * (dist2 / cosine);         // integrator.rs:65
                        const float weight = mis2(light_pdf, bp.pdf);
                        const DMaterial& lm = sv.materials[L.material];
                        ce = f3{lm.colour[0], lm.colour[1], lm.colour[2]} * weight * mat_weakening(mat.kind, dir, normal) * bp.bsdf / light_pdf;
                        want_shadow = true;
                        sh_a = f4{p.x, p.y, p.z, (1.0f - PT_EPSILON) * dist};          // integrator.rs:56
                        sh_b = f4{dir.x, dir.y, dir.z, asf(pid)};
                    }
                    nee_e = f4{ce.x, ce.y, ce.z, 0.0f};
                }
                // ---- estimate_direct_bsdf  integrator.rs:77-130
                {
                    PT_SHADE_ARGS
                    const f3 dir = mat_scatter(mat, rng, rd, normal, front);
                    if (dot3(dir, normal) > 0.0f)                                      // integrator.rs:96
                    {
                        const BsdfSample bp = mat_bsdf_pdf(mat, wi, dir, normal, front);
                        // scene.lights.intersect(ray) starts with the root box of the lights TLAS (tlas.rs:68-74).  That test is done
                        // here, with the very arithmetic k_closest uses at refill: a ray that fails it (nearly all of them: the lights
                        // are small) finds no light, contributes nothing (integrator.rs:100-102) and never becomes a queue entry.
                        float te;
                        const uint4* const lr = reinterpret_cast<const uint4*>(sv.nodes + sv.lights_root);
                        const bool may_hit = slab(lr[0], lr[1], p, rcp3(dir), asf(0x7f800000u), te);
                        if (may_hit)
                        {
                            nee_b = f4{bp.bsdf.x, bp.bsdf.y, bp.bsdf.z, mat_weakening(mat.kind, dir, normal)};
                            nee_pw.w = bp.pdf;
                            want_lchain = true;
                            lc_a = f4{p.x, p.y, p.z, asf(0x7f800000u)};
                            lc_b = f4{dir.x, dir.y, dir.z, asf(pid)};
                            flags |= FLAG_BSDF_CAST;
                        }
                        else if (!INLINE) culled += 1u;
                        else cull_now = true;
                    }
                }
                nee_pw.x = pw.x; nee_pw.y = pw.y; nee_pw.z = pw.z;
                flags |= FLAG_NEE_PENDING;
            }

            // ---- continuation  integrator.rs:236-251
            ndir = mat_scatter(mat, rng, rd, normal, front);
            const BsdfSample info = mat_bsdf_pdf(mat, wi, ndir, normal, front);
            alive = !(info.pdf < 0.0f);                                                // MIN_PDF = 0  integrator.rs:243
            if (alive)
            {
                pw = pw * (mat_weakening(mat.kind, ndir, normal) * info.bsdf / info.pdf); // integrator.rs:249
                flags = is_delta ? (flags | FLAG_LAST_DELTA) : (flags & ~FLAG_LAST_DELTA);
            }
            } // surface interaction
            if (alive)
            {
                {
                const uint32_t nb = bounce + 1u;
                if (nb > rp.max_bounces) alive = false;                                 // for b in 0..=max_bounces
                else if (nb > 3u)                                                       // Russian roulette of the next iteration  integrator.rs:166-177
                {
                    const float survive = min_num(hmax3(pw), 0.9999f);
                    if (rng.f32() > survive) alive = false;
                    else pw = pw / survive;
                }
                }
            }
            if (alive && VOLUMES && (vst != vst_in || bounce == 0u)) io.st.vstack[pid] = vst;
            draws = rng.k;
            if (alive)
            {
                want_next = true;
                nx_a = f4{next_o.x, next_o.y, next_o.z, asf(0x7f800000u)};
                nx_b = f4{ndir.x, ndir.y, ndir.z, asf(pid)};
            }
            else if (flags & FLAG_NEE_PENDING)
            {
                // the path is over but still owes this bounce's direct light (integrator.rs:231-234)
                if (flags & FLAG_BSDF_CAST) want_dead = true;            // both estimates pending: the terminal pass adds them
                else if (want_shadow) { ends_with_shadow = true; sh_b.w = asf(pid | PATH_ENDS); } // the shadow-ray kernel finishes it
                else acc = acc + xyz(nee_pw) * (xyz(nee_e) + f3{0.0f, 0.0f, 0.0f});  // nothing was cast: explicit = bsdf = 0
            }
        }
        // ---- block-aggregated queue appends (every thread of the block reaches this): one atomic per queue per block
        PT_SHADE_ARGS
        uint32_t* const ctrs[4] = {&io.ctr->n_shadow, &io.ctr->n_lchain, &io.ctr_next->n_closest, &io.ctr_next->n_shade[Q_TERMINAL]};
        const bool preds[4] = {want_shadow && !INLINE, want_lchain, want_next, want_dead};
        const uint32_t caps[4] = {io.cap_slots, io.cap_slots, io.cap_slots, io.cap_slots_term};
        uint32_t pos[4];
        block_append4(sh_append, ctrs, preds, caps, &io.ctr->overflow, pos);
        if (want_shadow && !INLINE) { nt_store(io.rq_shadow.a + pos[0], sh_a); nt_store(io.rq_shadow.b + pos[0], sh_b); }
        if (want_lchain) { io.rq_lchain.a[pos[1]] = lc_a; io.rq_lchain.b[pos[1]] = lc_b; io.lchain_nb[pos[1]] = nee_b; nee_e.w = asf(pos[1]); }
        if (want_next) { nt_store(io.rq_out.a + pos[2], nx_a); nt_store(io.rq_out.b + pos[2], nx_b); }
        if (want_dead) io.q_term_next[pos[3]] = make_uint2(pid | ENTRY_DEAD, pid);
        const bool write_rec = valid && (want_next || want_dead || ends_with_shadow);
        if (valid && !write_rec) io.st.radiance[pid] = f4{acc.x, acc.y, acc.z, 0.0f}; // the path ended here, nothing owed
        // The 64-byte record is stored by the four lanes of a quad together, one whole record per store instruction (each lane a
        // 16-byte word): L2 then sees one full-sector write per record instead of four partial ones.  4x4 transposes inside the quad.
        {
            float m[4][4] = {{pw.x, pw.y, pw.z, asf(draws)}, {acc.x, acc.y, acc.z, asf(flags)}, {nee_e.x, nee_e.y, nee_e.z, nee_e.w}, {nee_pw.x, nee_pw.y, nee_pw.z, nee_pw.w}};
            const uint32_t q = threadIdx.x & 3u;
#pragma unroll
            for (int c = 0; c < 4; ++c) // component c of the four words: m[word][c]
            {
                float a0 = m[0][c], a1 = m[1][c], a2 = m[2][c], a3 = m[3][c];
                // exchange with lane ^ 1
                { const float t = (q & 1u) ? a0 : a1; const float r = asf(__builtin_amdgcn_mov_dpp(asu(t), 0xB1, 0xF, 0xF, true)); if (q & 1u) a0 = r; else a1 = r; }
                { const float t = (q & 1u) ? a2 : a3; const float r = asf(__builtin_amdgcn_mov_dpp(asu(t), 0xB1, 0xF, 0xF, true)); if (q & 1u) a2 = r; else a3 = r; }
                // exchange with lane ^ 2
                { const float t = (q & 2u) ? a0 : a2; const float r = asf(__builtin_amdgcn_mov_dpp(asu(t), 0x4E, 0xF, 0xF, true)); if (q & 2u) a0 = r; else a2 = r; }
                { const float t = (q & 2u) ? a1 : a3; const float r = asf(__builtin_amdgcn_mov_dpp(asu(t), 0x4E, 0xF, 0xF, true)); if (q & 2u) a1 = r; else a3 = r; }
                m[0][c] = a0; m[1][c] = a1; m[2][c] = a2; m[3][c] = a3; // now m[k][c] = component c of word q of quad member k
            }
            const uint32_t wr = write_rec ? 1u : 0u;
#define PT_QUAD_STORE(K, SEL)                                                                                                   \
            {                                                                                                                      \
                const uint32_t pid_k = __builtin_amdgcn_mov_dpp(pid, SEL, 0xF, 0xF, true); /* quad_perm [K,K,K,K] */               \
                const uint32_t wr_k = __builtin_amdgcn_mov_dpp(wr, SEL, 0xF, 0xF, true);                                           \
                if (wr_k != 0u) reinterpret_cast<f4*>(io.st.rec + pid_k)[q] = f4{m[K][0], m[K][1], m[K][2], m[K][3]};             \
            }
            PT_QUAD_STORE(0, 0x00) PT_QUAD_STORE(1, 0x55) PT_QUAD_STORE(2, 0xAA) PT_QUAD_STORE(3, 0xFF)
#undef PT_QUAD_STORE
        }
        if (INLINE)
        {
            // the explicit-light shadow ray is answered here, after the record is on its way: what k_any<SHADOW> does to the record (put_result_at), same arithmetic
            // (the stack's base is recomputed here, behind an opaque move, rather than kept in a register across the iteration)
            // ... and so is the description of the staged scene (re-read from the kernel-argument segment like every other section's)
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const ShadeKArgs& ka2 = shade_args();
            const uint32_t blob_words = ka2.sv.blob_bytes >> 4;
            Blob bl;
            bl.nodes = smem_dyn;
            bl.tris = smem_dyn + 2u * ka2.sv.n_nodes;
            bl.inst = smem_dyn + 2u * ka2.sv.n_nodes + 3u * ka2.sv.n_tris;
            bl.leaves = reinterpret_cast<const uint2*>(smem_dyn + 2u * ka2.sv.n_nodes + 3u * ka2.sv.n_tris + INST_WORDS * ka2.sv.n_instances);
            const Stack8<false> stk{reinterpret_cast<char*>(smem_dyn), blob_words * 16u + tid * 8u, blockDim.x * 8u};
            traced += (uint32_t)__popcll(__ballot(want_shadow)); // (both counted here, where the whole wave is: wave-uniform values in scalar registers)
            culled += (uint32_t)__popcll(__ballot(cull_now));
            const bool blocked = inline_any(bl, stk, ka2.world_root, sh_a, sh_b, want_shadow);
            if (want_shadow)
            {
                DPathRec* rec = shade_args().io.st.rec + pid;
                if (ends_with_shadow)
                {
                    const f3 e = blocked ? f3{0.0f, 0.0f, 0.0f} : xyz(rec->nee_e);
                    const f3 a2 = xyz(rec->acc) + xyz(rec->nee_pw) * (e + f3{0.0f, 0.0f, 0.0f});
                    shade_args().io.st.radiance[pid] = f4{a2.x, a2.y, a2.z, 0.0f};
                }
                else if (blocked)
                {
                    float* e = reinterpret_cast<float*>(&rec->nee_e);
                    e[0] = 0.0f; e[1] = 0.0f; e[2] = 0.0f;
                }
            }
            a_next = (idx + stride) < n ? nt_load(shade_args().io.q_in.a + idx + stride) : hole_a;
        }
    }
    if (!INLINE) add_tally(io.lchain_heads, culled, HEAD_TALLY2);
    else if (lane_id() == 0u)
    {
        // (both counts are wave-uniform here)
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const uint32_t line = ((wave * 7u) & (kQueueHeads - 1u)) * kHeadStrideWords;
        if (culled != 0u) atomicAdd(io.lchain_heads + line + HEAD_TALLY2, culled);
        if (traced != 0u) atomicAdd(io.shadow_heads + line + HEAD_TALLY0, traced);
    }
}

#undef PT_SHADE_ARGS

// integrator.rs:272-280: finite check, clamp_length_max(100), alpha 1
__device__ __forceinline__ f3 finalise(f3 acc)
{
    if (!finite3(acc)) return f3{0.0f, 0.0f, 0.0f};
    return clamp_len_max(acc, 100.0f);
}

// accumulate.wgsl:20-23 applied once per sample, in sample order; id history shift main.rs:206.  A pixel outside the active rectangle
// (RenderParams::act_*) never had a path: each of its samples is the miss result of integrator.rs:263-266 — radiance 0.006, id 255,
// position r.at(1e5) of that sample's camera ray (:156-157) — added sample by sample like any other.
// The sum must be taken in sample order (float addition does not associate), the loads and the per-sample finite check / clamp need not
// wait for each other.  !FEW_PIXELS: one thread per LOCAL pixel, eight samples' loads in flight.  FEW_PIXELS (one rank's share of a
// sharded frame: the active pixels fill ~250 workgroups, one wave per SIMD, and the kernel waits on memory latency): FOUR lanes per
// pixel; lane q loads and finalises samples q, q + 4, ... and all four lanes then add the quad's values in sample order (DPP quad
// broadcasts; the sums are redundant, the loads are not): four times the loads in flight.
template <bool FEW_PIXELS>
__global__ void __launch_bounds__(256) k_accumulate(const RenderParams rp, const CameraView cam, const PathState st, f4* accum, f4* position, uint32_t* id,
                                                     const uint32_t write_position, const uint32_t add_to_accum)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lp = FEW_PIXELS ? tid >> 2 : tid, q = FEW_PIXELS ? (tid & 3u) : 0u;
    if (lp >= rp.local_pixels) return;
    f4 a = accum[lp];
    uint32_t idv = id[lp];
    const uint32_t ly = fastdiv(lp, rp.div_width), x = lp - ly * rp.width;
    if (x - rp.act_x0 >= rp.act_w || ly - rp.act_ly0 >= rp.act_rows)
    {
        for (uint32_t s = 0; s < rp.batch_samples; ++s) a = f4{a.x + 0.006f, a.y + 0.006f, a.z + 0.006f, a.w + 1.0f};
        for (uint32_t s = rp.batch_samples >= 2u ? rp.batch_samples - 2u : 0u; s < rp.batch_samples; ++s) idv = (idv << 16) | 255u;
        if (q != 0u) return;
        if (add_to_accum) accum[lp] = a;
        id[lp] = idv;
        if (write_position)
        {
            const f3 d = camera_ray_dir(rp, cam, x, global_row(rp, ly), rp.first_sample + rp.batch_samples - 1u);
            const f3 far = fma3(d, bc3(1e5f), f3{cam.eye[0], cam.eye[1], cam.eye[2]});
            position[lp] = f4{far.x, far.y, far.z, 1e5f};
        }
        return;
    }
    const uint32_t k = (ly - rp.act_ly0) * rp.act_w + (x - rp.act_x0);
    if (FEW_PIXELS)
    {
        constexpr uint32_t G = 4; // samples per lane and round: 16 per pixel
        for (uint32_t s0 = 0; s0 < rp.batch_samples; s0 += 4u * G)
        {
            uint32_t oc[G];
            f4 rad[G];
#pragma unroll
            for (uint32_t j = 0; j < G; ++j)
            {
                const uint32_t s = s0 + 4u * j + q;
                const uint32_t pid = pid_join(rp, s, k);
                oc[j] = s < rp.batch_samples ? (uint32_t)st.occl[pid] : (uint32_t)PRIMARY_MISS;
                // (a primary miss's record is stale memory inside the allocation: read and dropped, so that the load need not wait for the byte)
                rad[j] = s < rp.batch_samples ? st.radiance[pid] : f4{};
            }
            f3 c[G];
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) c[j] = finalise(oc[j] != PRIMARY_MISS ? xyz(rad[j]) : f3{0.006f, 0.006f, 0.006f});
#pragma unroll
            for (uint32_t j = 0; j < G; ++j)
            {
#define PT_QUAD_ADD(SEL, QQ)                                                                                                           \
                if (s0 + 4u * j + (QQ) < rp.batch_samples)                                                                          \
                {                                                                                                                      \
                    const float cx = asf(__builtin_amdgcn_mov_dpp(asu(c[j].x), SEL, 0xF, 0xF, true));                                 \
                    const float cy = asf(__builtin_amdgcn_mov_dpp(asu(c[j].y), SEL, 0xF, 0xF, true));                                 \
                    const float cz = asf(__builtin_amdgcn_mov_dpp(asu(c[j].z), SEL, 0xF, 0xF, true));                                 \
                    a = f4{a.x + cx, a.y + cy, a.z + cz, a.w + 1.0f};                                                                  \
                }
                PT_QUAD_ADD(0x00, 0u) PT_QUAD_ADD(0x55, 1u) PT_QUAD_ADD(0xAA, 2u) PT_QUAD_ADD(0xFF, 3u)
#undef PT_QUAD_ADD
            }
        }
        if (q != 0u) return;
        // (id << 16) | new once per sample: only the batch's last two samples are kept (RenderParams::keep_s_id); a camera ray that left the
        // scene at once wrote nothing but its PRIMARY_MISS byte: id 255 (integrator.rs:157)
        for (uint32_t s = rp.keep_s_id; s < rp.batch_samples; ++s)
        {
            const bool miss = st.occl[pid_join(rp, s, k)] == PRIMARY_MISS;
            idv = (idv << 16) | (miss ? 255u : st.first_id[(s - rp.keep_s_id) * rp.act_pixels + k]);
        }
    }
    else
    {
        constexpr uint32_t G = 8;
        for (uint32_t s0 = 0; s0 < rp.batch_samples; s0 += G)
        {
            uint32_t oc[G];
            f4 rad[G];
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) oc[j] = (s0 + j) < rp.batch_samples ? (uint32_t)st.occl[pid_join(rp, s0 + j, k)] : (uint32_t)PRIMARY_MISS;
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) rad[j] = oc[j] != PRIMARY_MISS ? st.radiance[pid_join(rp, s0 + j, k)] : f4{0.006f, 0.006f, 0.006f, 0.0f};
#pragma unroll
            for (uint32_t j = 0; j < G; ++j)
            {
                if ((s0 + j) >= rp.batch_samples) break;
                const f3 c = finalise(xyz(rad[j]));
                a = f4{a.x + c.x, a.y + c.y, a.z + c.z, a.w + 1.0f};
                // (id << 16) | new once per sample: only the last two samples survive in 32 bits; a PRIMARY_MISS is id 255 (integrator.rs:157)
                if (s0 + j >= rp.keep_s_id) idv = (idv << 16) | (oc[j] == PRIMARY_MISS ? 255u : st.first_id[(s0 + j - rp.keep_s_id) * rp.act_pixels + k]);
            }
        }
    }
    if (add_to_accum) accum[lp] = a;
    id[lp] = idv;
    if (write_position)
    {
        // the batch's last sample.  A camera ray that left the scene at once has no record: r.at(1e5) of that sample's ray (integrator.rs:156)
        if (st.occl[pid_join(rp, rp.keep_s_pos, k)] == PRIMARY_MISS)
        {
            const f3 d = camera_ray_dir(rp, cam, x, global_row(rp, ly), rp.first_sample + rp.keep_s_pos);
            const f3 far = fma3(d, bc3(1e5f), f3{cam.eye[0], cam.eye[1], cam.eye[2]});
            position[lp] = f4{far.x, far.y, far.z, 1e5f};
        }
        else position[lp] = st.first_pos[k];
    }
}

// per-sample radiance (pt_render_samples): out[sample * local_pixels + local pixel]
__global__ void __launch_bounds__(256) k_store_samples(const RenderParams rp, const PathState st, f4* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.local_pixels * rp.batch_samples) return;
    const uint32_t s = i / rp.local_pixels, lp = i - s * rp.local_pixels;
    const uint32_t ly = fastdiv(lp, rp.div_width), x = lp - ly * rp.width;
    f3 c{0.006f, 0.006f, 0.006f};
    if (x - rp.act_x0 < rp.act_w && ly - rp.act_ly0 < rp.act_rows)
    {
        const uint32_t pid = pid_join(rp, s, (ly - rp.act_ly0) * rp.act_w + (x - rp.act_x0));
        c = finalise(st.occl[pid] == PRIMARY_MISS ? f3{0.006f, 0.006f, 0.006f} : xyz(st.radiance[pid]));
    }
    out[i] = f4{c.x, c.y, c.z, 1.0f};
}

// ------------------------------------------------------------------------------------------------ probes
__global__ void k_sobol_probe(uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ss_sobol(n_points, index[i], seed[i], &out[2 * i], &out[2 * i + 1]);
}
__global__ void k_math_probe(int fn, uint32_t n, const float* a, const float* b, float* o0, float* o1, uint64_t seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn)
    {
    case 0: sincos_det(a[i], &o0[i], &o1[i]); break;
    case 1: o0[i] = exp_det(a[i]); break;
    case 2: o0[i] = ln_det(a[i]); break;
    case 3: o0[i] = hypot_det(a[i], b[i]); break;
    case 4: o0[i] = a[i] / b[i]; break;
    case 5: o0[i] = sqrtf(a[i]); break;
    case 6: o0[i] = tan_det(a[i]); break;
    case 8: o0[i] = atan2_det(a[i], b[i]); break;
    case 9: o0[i] = asin_det(a[i]); break;
    case 7:
    {
        Stream r{stream_key(seed, asu(a[i]), asu(b[i])), 0u};
        o0[i] = asf(r.u32());
        o1[i] = r.f32();
        break;
    }
    }
}
__global__ void k_material_probe(const SceneView sv, int material, uint32_t n, const float* incoming, const float* normal, const uint8_t* front,
                                 const uint32_t* pixel, const uint32_t* sample, uint32_t draws, uint64_t seed, float* out9)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MatView m = load_material(sv.materials, (uint32_t)material);
    Stream rng{stream_key(seed, pixel[i], sample[i]), draws};
    const f3 in{incoming[3 * i], incoming[3 * i + 1], incoming[3 * i + 2]}, nn{normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
    const bool ff = front[i] != 0;
    const f3 wo = mat_scatter(m, rng, in, nn, ff);
    const BsdfSample bp = mat_bsdf_pdf(m, -in, wo, nn, ff);
    float* o = out9 + 9 * i;
    o[0] = wo.x; o[1] = wo.y; o[2] = wo.z;
    o[3] = bp.bsdf.x; o[4] = bp.bsdf.y; o[5] = bp.bsdf.z;
    o[6] = bp.pdf;
    o[7] = mat_weakening(m.kind, wo, nn);
    o[8] = (float)(rng.k - draws);
}

} // namespace

// ================================================================================================ launchers
void launch_generate(hipStream_t s, const RenderParams& rp, const CameraView& cam, const WavefrontBuffers& wb)
{
    const uint32_t blocks = (rp.n_paths + 255u) / 256u;
    hipLaunchKernelGGL(k_generate, dim3(blocks), dim3(256), 0, s, rp, cam, wb.rq[0], wb.counters);
}

// A persistent grid must be RESIDENT: workgroups that the device cannot hold at once start only when others have left, and by then
// the queue's dynamic part is gone and the late-comers' static chunks are the launch's tail.  So the grid is what the kernel's
// registers and LDS allow per CU (asked of the runtime once per kernel variant and LDS size), times the CU count, capped by
// tl.grid_blocks (the spill area is sized for that).
#ifndef PT_RESIDENT_GRID
#define PT_RESIDENT_GRID 1
#endif
template <typename K>
static uint32_t resident_grid(K kernel, const TraceLaunch& tl, size_t lds)
{
#if PT_RESIDENT_GRID
    struct Key { const void* f; uint32_t threads; size_t lds; int dev; int per_cu; };
    static std::mutex mu;
    static std::vector<Key> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (const Key& k : cache)
        if (k.f == (const void*)kernel && k.threads == tl.block_threads && k.lds == lds && k.dev == dev) return std::min<uint32_t>(tl.grid_blocks, tl.n_cus * (uint32_t)k.per_cu);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)tl.block_threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    cache.push_back(Key{(const void*)kernel, tl.block_threads, lds, dev, per_cu});
    return std::min<uint32_t>(tl.grid_blocks, tl.n_cus * (uint32_t)per_cu);
#else
    return tl.grid_blocks;
#endif
}

template <int MODE>
static void launch_closest_impl(hipStream_t s, const TraceLaunch& tl, uint32_t root, const RayQueue& rq, const uint32_t* n_ptr, uint32_t cap_in,
                                uint32_t* heads, const ClosestOut& out)
{
    const bool spill = tl.scene.stack_entries > tl.scene.stack_lds;
    const dim3 block(tl.block_threads);
    const uint4* blob = (const uint4*)tl.blob;
    const size_t lds = trace_lds_bytes(tl);
    const ClosestKArgs ka{tl.scene, blob, rq.a, rq.b, n_ptr, heads, root, cap_in, out};
#define PT_LAUNCH1(K) hipLaunchKernelGGL(K, dim3(resident_grid(K, tl, lds)), block, lds, s, ka)
    if (tl.lds_scene && !spill) PT_LAUNCH1((k_closest<1, MODE, false>));
    else if (tl.lds_scene) PT_LAUNCH1((k_closest<1, MODE, true>));
    else if (!spill) PT_LAUNCH1((k_closest<0, MODE, false>));
    else PT_LAUNCH1((k_closest<0, MODE, true>));
#undef PT_LAUNCH1
}
template <int MODE>
static void launch_any_impl(hipStream_t s, const TraceLaunch& tl, uint32_t root, const RayQueue& rq, const uint32_t* n_ptr, uint32_t cap_in,
                            uint32_t* heads, uint32_t* occluded, f4* radiance = nullptr)
{
    const size_t lds = trace_lds_bytes(tl);
    const bool spill = tl.scene.stack_entries > tl.scene.stack_lds;
    const dim3 block(tl.block_threads);
    const uint4* blob = (const uint4*)tl.blob;
#define PT_LAUNCH(K) hipLaunchKernelGGL(K, dim3(resident_grid(K, tl, lds)), block, lds, s, tl.scene, blob, root, rq.a, rq.b, n_ptr, cap_in, heads, occluded, radiance)
    if (tl.lds_scene && !spill) PT_LAUNCH((k_any<1, MODE, false>));
    else if (tl.lds_scene) PT_LAUNCH((k_any<1, MODE, true>));
    else if (!spill) PT_LAUNCH((k_any<0, MODE, false>));
    else PT_LAUNCH((k_any<0, MODE, true>));
#undef PT_LAUNCH
}

static uint32_t* row_heads(const WavefrontBuffers& wb, uint32_t row, uint32_t which)
{
    return wb.heads + ((size_t)row * HEADS_PER_ROW + which) * kHeadWordsPerQueue;
}

void launch_trace_world(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b, const RenderParams& rp, const CameraView& cam,
                        const EnvView& env)
{
    Counters* row = wb.counters + b;
    ClosestOut out{};
    out.hits = wb.hits;
    out.q_base = wb.q_shade_base;
    out.q_stride = wb.q_stride;
    out.q_class_slot = wb.q_class_slot;
    out.q_term = wb.q_term[b & 1u];
    out.n_shade = row->n_shade;
    out.tails = wb.tails + (size_t)b * Q_COUNT * kTailWordsPerQueue;
    out.wave_times = wb.wave_times ? wb.wave_times + (size_t)b * kWaveTimeSlots : nullptr;
    out.cap_shade = wb.cap_slots_shade;
    out.cap_term = wb.cap_slots_term;
    out.class_mask = wb.class_mask | (1u << Q_TERMINAL);
    out.overflow = &row->overflow;
    if (b == 0u)
    {
        out.eye = f3{cam.eye[0], cam.eye[1], cam.eye[2]};
        out.radiance = wb.st.radiance;
        out.occl = wb.st.occl;
        out.first_pos = wb.st.first_pos;
        out.first_id = wb.st.first_id;
        out.keep_s_id = rp.keep_s_id; out.keep_s_pos = rp.keep_s_pos;
        out.blk_log = rp.blk_log; out.n_blk = rp.n_blk; out.blk_last = rp.blk_last; out.act_pixels = rp.act_pixels;
        out.div_blk_paths = rp.div_blk_paths; out.div_blk_last = rp.div_blk_last;
        out.finalize_miss = env.w == 0u ? 1u : 0u;
        launch_closest_impl<CLOSEST_PRIMARY>(s, tl, tl.scene.world_root, wb.rq[0], &row->n_closest, wb.cap_slots, row_heads(wb, b, HEADS_CLOSEST), out);
    }
    else
    {
        out.rec = wb.st.rec;
        out.radiance = wb.st.radiance;
        out.enable_nee = rp.enable_nee;
        out.finalize_miss = env.w == 0u ? 1u : 0u;
        launch_closest_impl<CLOSEST_WORLD>(s, tl, tl.scene.world_root, wb.rq[b & 1u], &row->n_closest, wb.cap_slots, row_heads(wb, b, HEADS_CLOSEST), out);
    }
}
// world closest hit of bounce b (b >= 1) + the BSDF-sampled NEE rays of bounce b - 1 in one launch (k_trace_fused)
void launch_trace_fused(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b, const RenderParams& rp, const EnvView& env)
{
    Counters* row = wb.counters + b;
    Counters* prev = wb.counters + (b - 1u);
    ClosestOut wout{};
    wout.hits = wb.hits;
    wout.q_base = wb.q_shade_base;
    wout.q_stride = wb.q_stride;
    wout.q_class_slot = wb.q_class_slot;
    wout.q_term = wb.q_term[b & 1u];
    wout.n_shade = row->n_shade;
    wout.tails = wb.tails + (size_t)b * Q_COUNT * kTailWordsPerQueue;
    wout.wave_times = nullptr;
    wout.cap_shade = wb.cap_slots_shade;
    wout.cap_term = wb.cap_slots_term;
    wout.class_mask = wb.class_mask | (1u << Q_TERMINAL);
    wout.overflow = &row->overflow;
    wout.rec = wb.st.rec;
    wout.radiance = wb.st.radiance;
    wout.enable_nee = rp.enable_nee;
    wout.finalize_miss = env.w == 0u ? 1u : 0u;
    ClosestOut lout{};
    lout.hits = wb.lchain_hit;
    lout.world_root = tl.scene.world_root;
    lout.occl = wb.st.occl;
    FusedArgs fa{};
    fa.wa = wb.rq[b & 1u].a; fa.wb = wb.rq[b & 1u].b; fa.wn = &row->n_closest; fa.wheads = row_heads(wb, b, HEADS_CLOSEST);
    fa.la = wb.rq_lchain[(b - 1u) & 1u].a; fa.lb = wb.rq_lchain[(b - 1u) & 1u].b; fa.ln = &prev->n_lchain; fa.lheads = row_heads(wb, b - 1u, HEADS_LCHAIN);
    fa.world_root = tl.scene.world_root; fa.lights_root = tl.scene.lights_root; fa.cap_in = wb.cap_slots;
    const size_t lds = trace_lds_bytes(tl);
    const bool spill = tl.scene.stack_entries > tl.scene.stack_lds;
    const dim3 block(tl.block_threads);
    const uint4* blob = (const uint4*)tl.blob;
    const FusedKArgs ka{tl.scene, blob, fa, wout, lout};
#define PT_LAUNCH(K) hipLaunchKernelGGL(K, dim3(resident_grid(K, tl, lds)), block, lds, s, ka)
    if (tl.lds_scene && !spill) PT_LAUNCH((k_trace_fused<1, false>));
    else if (tl.lds_scene) PT_LAUNCH((k_trace_fused<1, true>));
    else if (!spill) PT_LAUNCH((k_trace_fused<0, false>));
    else PT_LAUNCH((k_trace_fused<0, true>));
#undef PT_LAUNCH
}
void launch_trace_shadow(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b)
{
    Counters* row = wb.counters + b;
#if PT_WAVE_TIMES
    {
        uint4* where = wb.wave_times_any ? wb.wave_times_any + (size_t)b * kWaveTimeSlots : nullptr;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_any_times), &where, sizeof(where), 0, hipMemcpyHostToDevice, s);
    }
#endif
    launch_any_impl<ANY_SHADOW>(s, tl, tl.scene.world_root, wb.rq_shadow, &row->n_shadow, wb.cap_slots, row_heads(wb, b, HEADS_SHADOW),
                                reinterpret_cast<uint32_t*>(wb.st.rec), wb.st.radiance);
}
void launch_trace_lchain(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b)
{
    Counters* row = wb.counters + b;
    ClosestOut out{};
    out.hits = wb.lchain_hit;
    out.n_shade = nullptr;
    out.world_root = tl.scene.world_root;
    out.occl = wb.st.occl;
    launch_closest_impl<CLOSEST_LIGHTS>(s, tl, tl.scene.lights_root, wb.rq_lchain[b & 1u], &row->n_lchain, wb.cap_slots, row_heads(wb, b, HEADS_LCHAIN), out);
}
// The Lambertian shading pass walks its own shadow rays when the scene's BVH and a workgroup's stacks take no more LDS than five workgroups per CU can share
// (the pass keeps its five waves per SIMD), nothing spills from the stacks, the scene has no media (those kernels are short of registers as it is) and the
// traversal workgroups have the shading pass's shape (the stack layout is per thread of a workgroup).
bool shade_traces_shadow(const TraceLaunch& tl)
{
    return PT_INLINE_SHADOW != 0 && tl.lds_scene && tl.scene.stack_entries <= tl.scene.stack_lds && !tl.scene.has_volumes && tl.block_threads == PT_SHADE_THREADS &&
           trace_lds_bytes(tl) + 1024 <= 32 * 1024;
}
void launch_shade(hipStream_t s, uint32_t qclass, const SceneView& sv, const RenderParams& rp, const WavefrontBuffers& wb, uint32_t b,
                  uint32_t grid_blocks, const CameraView& cam, const EnvView& env, const TraceLaunch* tl)
{
    ShadeIO io{};
    io.env = env;
    io.primary_a = f4{cam.eye[0], cam.eye[1], cam.eye[2], __builtin_inff()};
    io.st = wb.st;
    io.rq_in = wb.rq[b & 1u];
    io.rq_out = wb.rq[(b + 1u) & 1u];
    io.rq_shadow = wb.rq_shadow;
    io.rq_lchain = wb.rq_lchain[b & 1u];
    io.rq_lchain_prev = wb.rq_lchain[(b + 1u) & 1u];
    io.lchain_nb = wb.lchain_nb[b & 1u];
    io.lchain_nb_prev = wb.lchain_nb[(b + 1u) & 1u];
    io.lchain_hit = wb.lchain_hit;
    io.hits = wb.hits;
    io.entries = wb.q_term[b & 1u];
    if (qclass != Q_TERMINAL)
    {
        io.q_in = shade_queue(wb, qclass);
        io.tails_in = wb.tails + ((size_t)b * Q_COUNT + qclass) * kTailWordsPerQueue;
    }
    io.q_term_next = wb.q_term[(b + 1u) & 1u];
    io.cap_slots = wb.cap_slots;
    io.cap_slots_term = wb.cap_slots_term;
    io.cap_slots_shade = wb.cap_slots_shade;
    io.ctr = wb.counters + b;
    io.lchain_heads = row_heads(wb, b, HEADS_LCHAIN);
    io.shadow_heads = row_heads(wb, b, HEADS_SHADOW);
    io.ctr_next = wb.counters + b + 1u;
    const uint32_t surface_blocks = (grid_blocks * 256u + PT_SHADE_THREADS - 1u) / PT_SHADE_THREADS; // grid_blocks is in units of 256 threads
    const bool inl = tl && shade_traces_shadow(*tl);
    const ShadeKArgs ka{sv, rp, io, b, inl ? (const uint4*)tl->blob : nullptr, inl ? tl->scene.world_root : 0u};
    const size_t lds = inl ? trace_lds_bytes(*tl) : 0;
    switch (qclass)
    {
    case Q_TERMINAL: hipLaunchKernelGGL(k_shade_terminal, dim3(grid_blocks), dim3(256), 0, s, sv, rp, io, b); break;
    case Q_LAMBERT:
        if (inl) hipLaunchKernelGGL((k_shade_surface<Q_LAMBERT, false, true>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), lds, s, ka);
        else if (sv.has_volumes) hipLaunchKernelGGL((k_shade_surface<Q_LAMBERT, true>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        else hipLaunchKernelGGL((k_shade_surface<Q_LAMBERT, false>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        break;
    case Q_SPECULAR:
        if (sv.has_volumes) hipLaunchKernelGGL((k_shade_surface<Q_SPECULAR, true>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        else hipLaunchKernelGGL((k_shade_surface<Q_SPECULAR, false>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        break;
    case Q_DIELECTRIC:
        if (sv.has_volumes) hipLaunchKernelGGL((k_shade_surface<Q_DIELECTRIC, true>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        else hipLaunchKernelGGL((k_shade_surface<Q_DIELECTRIC, false>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        break;
    case Q_GGX:
        if (sv.has_volumes) hipLaunchKernelGGL((k_shade_surface<Q_GGX, true>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        else hipLaunchKernelGGL((k_shade_surface<Q_GGX, false>), dim3(surface_blocks), dim3(PT_SHADE_THREADS), 0, s, ka);
        break;
    default: break;
    }
}

void launch_accumulate(hipStream_t s, const RenderParams& rp, const CameraView& cam, const WavefrontBuffers& wb, f4* accum, f4* position, uint32_t* id,
                       uint32_t write_position, uint32_t add_to_accum)
{
    const uint32_t blocks = (rp.local_pixels + 255u) / 256u;
    if (rp.local_pixels < (uint32_t)PT_ACC_QUAD_BELOW) hipLaunchKernelGGL(k_accumulate<true>, dim3((rp.local_pixels * 4u + 255u) / 256u), dim3(256), 0, s, rp, cam, wb.st, accum, position, id, write_position, add_to_accum);
    else hipLaunchKernelGGL(k_accumulate<false>, dim3(blocks), dim3(256), 0, s, rp, cam, wb.st, accum, position, id, write_position, add_to_accum);
}
void launch_store_samples(hipStream_t s, const RenderParams& rp, const WavefrontBuffers& wb, f4* out)
{
    const uint32_t blocks = (uint32_t)(((uint64_t)rp.local_pixels * rp.batch_samples + 255u) / 256u);
    hipLaunchKernelGGL(k_store_samples, dim3(blocks), dim3(256), 0, s, rp, wb.st, out);
}

void launch_trace_rays_closest(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* n_and_heads, f4* hits)
{
    ClosestOut out{};
    out.hits = hits;
    launch_closest_impl<CLOSEST_HOOK>(s, tl, root, rq, n_and_heads, n, n_and_heads + 32, out);
}
void launch_trace_rays_any(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* n_and_heads, uint32_t* occluded)
{
    launch_any_impl<ANY_HOOK>(s, tl, root, rq, n_and_heads, n, n_and_heads + 32, occluded);
}
void launch_sobol_probe(hipStream_t s, uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out_xy)
{
    hipLaunchKernelGGL(k_sobol_probe, dim3((n + 255u) / 256u), dim3(256), 0, s, n_points, n, index, seed, out_xy);
}
void launch_math_probe(hipStream_t s, int fn, uint32_t n, const float* a, const float* b, float* o0, float* o1, uint64_t seed)
{
    hipLaunchKernelGGL(k_math_probe, dim3((n + 255u) / 256u), dim3(256), 0, s, fn, n, a, b, o0, o1, seed);
}
void launch_material_probe(hipStream_t s, const SceneView& sv, int material, uint32_t n, const float* incoming, const float* normal,
                           const uint8_t* front, const uint32_t* pixel, const uint32_t* sample, uint32_t draws, uint64_t seed, float* out9)
{
    hipLaunchKernelGGL(k_material_probe, dim3((n + 255u) / 256u), dim3(256), 0, s, sv, material, n, incoming, normal, front, pixel, sample, draws, seed,
                       out9);
}

} // namespace pt
