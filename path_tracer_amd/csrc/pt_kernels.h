// Launch interface of the gfx950 kernels (pt_kernels.hip).  No launcher allocates, frees or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include "pt_materials.h"
#include "pt_types.h"

namespace pt {

struct TraceLaunch
{
    SceneView scene;
    const void* blob;        // contiguous nodes | tri_isect | instances (device), copied to LDS when lds_scene
    bool lds_scene;
    uint32_t grid_blocks;    // upper bound of the persistent grid (the launchers shrink it to what is resident at once)
    uint32_t n_cus;
    uint32_t block_threads;  // 64..256
};

// dynamic LDS of a traversal workgroup: the staged BVH blob (LDS scenes) + the per-lane (node, t_enter) stacks
inline size_t trace_lds_bytes(bool lds_scene, uint32_t blob_bytes, uint32_t stack_lds, uint32_t block_threads)
{
    return (lds_scene ? (size_t)blob_bytes : 0) + (size_t)stack_lds * block_threads * 8;
}
inline size_t trace_lds_bytes(const TraceLaunch& tl) { return trace_lds_bytes(tl.lds_scene, tl.scene.blob_bytes, tl.scene.stack_lds, tl.block_threads); }

struct WavefrontBuffers
{
    PathState st;
    RayQueue rq[2];       // world closest-hit rays, double buffered by bounce parity
    RayQueue rq_shadow;   // explicit-light shadow rays
    RayQueue rq_lchain[2]; // BSDF-sampled NEE rays, double buffered by bounce parity (the next shading pass re-reads directions)
    f4* lchain_nb[2];     // per BSDF-sampled ray, same slots and parity: bsdf rgb of the sampled direction | weakening
    f4* lchain_hit;       // per BSDF-sampled ray: its lights-TLAS closest hit t,u,v | id, written only when the light is visible
    f4* hits;             // world closest hits of rays whose path goes to the terminal queue, by ray index (and the unit hooks' output)
    // surface classes: self-contained hit records in queue order (ShadeQueue).  One allocation for all classes present in the scene:
    // class q's arrays a, b, c start at q_shade_base + (3 * slot(q) + {0,1,2}) * q_stride, slot(q) = nibble q of q_class_slot
    // (one base pointer instead of fifteen: the traversal kernels are short of scalar registers)
    f4* q_shade_base;
    uint32_t q_stride;     // slots per array (capacity + dump area)
    uint32_t q_class_slot;
    uint2* q_term[2];     // terminal queue {ray index | path id + ENTRY_DEAD, path id}, double buffered by bounce parity
    Counters* counters;   // [max_bounces + 2]
    uint32_t* heads;      // [max_bounces + 2][HEADS_PER_ROW][kHeadWordsPerQueue] claim cursors of the ray queues
    uint4* wave_times;    // PT_WAVE_TIMES builds only: [max_bounces + 2][kWaveTimeSlots] (k_closest launches), else nullptr
    uint4* wave_times_any; // ... the same for the shadow-ray launches
    uint32_t* tails;      // [max_bounces + 2][Q_COUNT][kTailWordsPerQueue] striped tails of the shade queues (pt_types.h)
    uint32_t cap_slots;   // capacity of every ray queue (each allocated with kQueueDumpSlots more)
    uint32_t cap_slots_shade; // capacity of the surface shade queues (= cap_slots except in the overflow test)
    uint32_t cap_slots_term;
    uint32_t class_mask;  // bit q: the scene has instances of shade class q
};

inline ShadeQueue shade_queue(const WavefrontBuffers& wb, uint32_t q)
{
    const size_t slot = (wb.q_class_slot >> (4u * q)) & 0xfu;
    f4* a = wb.q_shade_base + 3u * slot * (size_t)wb.q_stride;
    return ShadeQueue{a, a + wb.q_stride, a + 2u * (size_t)wb.q_stride};
}

void launch_generate(hipStream_t s, const RenderParams& rp, const CameraView& cam, const WavefrontBuffers& wb);
// closest hit against the world TLAS for bounce `b`: reads rq[b&1], writes hits + shade queues of row b
void launch_trace_world(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b, const RenderParams& rp, const CameraView& cam,
                        const EnvView& env);
// the same for bounce b >= 1 together with the BSDF-sampled NEE launch of bounce b - 1, as ONE launch (a wave goes from queue to queue)
void launch_trace_fused(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b, const RenderParams& rp, const EnvView& env);
// NEE rays produced by the shading of bounce `b` (counter row b)
void launch_trace_shadow(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b);
// BSDF-sampled NEE rays: closest hit against the lights TLAS, then (same kernel, same lane) any-hit against the world
void launch_trace_lchain(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b);
// shading of bounce b for one queue class
// (tl: the scene's traversal launch description; with it the Lambert / GGX passes of an LDS-resident scene may trace their own shadow rays)
void launch_shade(hipStream_t s, uint32_t qclass, const SceneView& sv, const RenderParams& rp, const WavefrontBuffers& wb, uint32_t b,
                  uint32_t grid_blocks, const CameraView& cam, const EnvView& env, const TraceLaunch* tl = nullptr);
// true: the shading pass answers the explicit-light shadow rays itself and nothing is queued for launch_trace_shadow
bool shade_traces_shadow(const TraceLaunch& tl);
// accum[pixel] += sum over batch samples in order of (finalised rgb, 1); position/id of the last samples
void launch_accumulate(hipStream_t s, const RenderParams& rp, const CameraView& cam, const WavefrontBuffers& wb, f4* accum, f4* position, uint32_t* id,
                       uint32_t write_position, uint32_t add_to_accum);
void launch_store_samples(hipStream_t s, const RenderParams& rp, const WavefrontBuffers& wb, f4* out);

// after the path (pt_post.hip)
void launch_post_accumulate(hipStream_t s, uint32_t n, const f4* input, f4* accum);
void launch_post_velocity(hipStream_t s, int w, int h, const f4* position, const float* m16, float* velocity_xy);
void launch_post_reproject(hipStream_t s, int w, int h, const f4* input, const f4* accum, const float* velocity_xy, const uint32_t* id, f4* output);
void launch_post_tonemap(hipStream_t s, uint32_t n, const f4* accum, f4* out);
void launch_post_rgb8(hipStream_t s, uint32_t n, const f4* accum, uint8_t* out);
void launch_post_deinterleave(hipStream_t s, uint32_t w, uint32_t h, uint32_t world, uint32_t strip, uint32_t pad_rows, const f4* parts, f4* full);

// unit hooks
// n_and_heads: word 0 = number of rays, words [32, 32 + kHeadWordsPerQueue) = zeroed claim cursors
void launch_trace_rays_closest(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* n_and_heads, f4* hits);
void launch_trace_rays_any(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* n_and_heads, uint32_t* occluded);
void launch_sobol_probe(hipStream_t s, uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out_xy);
void launch_math_probe(hipStream_t s, int fn, uint32_t n, const float* a, const float* b, float* o0, float* o1, uint64_t seed);
void launch_material_probe(hipStream_t s, const SceneView& sv, int material, uint32_t n, const float* incoming, const float* normal,
                           const uint8_t* front, const uint32_t* pixel, const uint32_t* sample, uint32_t draws, uint64_t seed, float* out9);

} // namespace pt
