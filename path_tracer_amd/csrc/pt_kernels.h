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
    uint32_t grid_blocks;    // persistent grid
    uint32_t block_threads;  // 64..256
};

struct WavefrontBuffers
{
    PathState st;
    RayQueue rq[2];       // world closest-hit rays, double buffered by bounce parity
    RayQueue rq_shadow;   // explicit-light shadow rays
    RayQueue rq_lchain[2]; // BSDF-sampled NEE rays, double buffered by bounce parity (the next shading pass re-reads directions)
    f4* lchain_nb[2];     // per BSDF-sampled ray, same slots and parity: bsdf rgb of the sampled direction | weakening
    f4* lchain_hit;       // per BSDF-sampled ray: its lights-TLAS closest hit t,u,v | id, written only when the light is visible
    f4* hits;             // world closest hits, dense by ray index
    uint2* q_shade[Q_COUNT]; // entries {ray index, path id}; Q_TERMINAL slot unused (see q_term)
    uint2* q_term[2];     // terminal queue {ray index | path id + ENTRY_DEAD, path id}, double buffered by bounce parity
    Counters* counters;   // [max_bounces + 2]
};

void launch_generate(hipStream_t s, const RenderParams& rp, const CameraView& cam, const WavefrontBuffers& wb);
// closest hit against the world TLAS for bounce `b`: reads rq[b&1], writes hits + shade queues of row b
void launch_trace_world(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b, const RenderParams& rp, const CameraView& cam,
                        const EnvView& env);
// NEE rays produced by the shading of bounce `b` (counter row b)
void launch_trace_shadow(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b);
// BSDF-sampled NEE rays: closest hit against the lights TLAS, then (same kernel, same lane) any-hit against the world
void launch_trace_lchain(hipStream_t s, const TraceLaunch& tl, const WavefrontBuffers& wb, uint32_t b);
// shading of bounce b for one queue class
void launch_shade(hipStream_t s, uint32_t qclass, const SceneView& sv, const RenderParams& rp, const WavefrontBuffers& wb, uint32_t b,
                  uint32_t grid_blocks, const CameraView& cam, const EnvView& env);
// accum[pixel] += sum over batch samples in order of (finalised rgb, 1); position/id of the last samples
void launch_accumulate(hipStream_t s, const RenderParams& rp, const WavefrontBuffers& wb, f4* accum, f4* position, uint32_t* id,
                       uint32_t write_position, uint32_t add_to_accum);
void launch_store_samples(hipStream_t s, const RenderParams& rp, const WavefrontBuffers& wb, f4* out);

// after the path (pt_post.hip)
void launch_post_accumulate(hipStream_t s, uint32_t n, const f4* input, f4* accum);
void launch_post_velocity(hipStream_t s, int w, int h, const f4* position, const float* m16, float* velocity_xy);
void launch_post_reproject(hipStream_t s, int w, int h, const f4* input, const f4* accum, const float* velocity_xy, const uint32_t* id, f4* output);
void launch_post_tonemap(hipStream_t s, uint32_t n, const f4* accum, f4* out);
void launch_post_rgb8(hipStream_t s, uint32_t n, const f4* accum, uint8_t* out);

// unit hooks
void launch_trace_rays_closest(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* head, f4* hits);
void launch_trace_rays_any(hipStream_t s, const TraceLaunch& tl, uint32_t root, RayQueue rq, uint32_t n, uint32_t* head, uint32_t* occluded);
void launch_sobol_probe(hipStream_t s, uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out_xy);
void launch_math_probe(hipStream_t s, int fn, uint32_t n, const float* a, const float* b, float* o0, float* o1, uint64_t seed);
void launch_material_probe(hipStream_t s, const SceneView& sv, int material, uint32_t n, const float* incoming, const float* normal,
                           const uint8_t* front, const uint32_t* pixel, const uint32_t* sample, uint32_t draws, uint64_t seed, float* out9);

} // namespace pt
