/* libptmi — MI355X-native wavefront path-tracing core: C-ABI drop-in boundary.
 *
 * The reference (CouncilmanJeremyJamm/path_tracer, Rust) has no FFI today; its per-pixel integration loop is called
 * in-crate from the rayon closure at src/main.rs:181-207.  This header is the seam a maintainer would bind with an
 * `extern "C"` block + build.rs: plain pointers and sizes, int status codes, no C++ / torch types,
 * nothing thrown across the boundary.  Each entry point cites the reference interface it replaces
 * (paths relative to /root/reference).
 *
 * Threading: one thread at a time per pt_ctx.  All device memory is owned by the ctx; caller buffers are caller-owned.
 */
#ifndef PT_API_H
#define PT_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

enum pt_status
{
    PT_OK = 0,
    PT_ERR_ARG = -1,      /* bad argument / capacity */
    PT_ERR_HIP = -2,      /* HIP runtime failure (no device, OOM, launch failure) */
    PT_ERR_STATE = -3,    /* call order (render before build, no camera, NEE without lights ...) */
    PT_ERR_NONRIGID = -4, /* model matrix carries scale: model.rs:40-44 asserts scale == (1,1,1) */
    PT_ERR_LIMIT = -5,    /* scene exceeds a packing limit of the device layout */
    PT_ERR_IO = -6,       /* file cannot be read / written */
    PT_ERR_PARSE = -7,    /* malformed OBJ (the reference panics) */
    PT_ERR_NCCL = -8      /* RCCL failure (library missing, communicator or collective error) */
};

/* Material enum of material.rs:80-89; GGX splits into its two GGXModel variants (material.rs:176-184). */
enum pt_material_kind
{
    PT_LAMBERTIAN = 0,     /* Lambertian::new(albedo)                           material.rs:99  */
    PT_EMISSIVE = 1,       /* Emissive::new(emitted)                            material.rs:126 */
    PT_SPECULAR = 2,       /* Specular::new(colour)                             material.rs:146 */
    PT_GGX_METAL = 3,      /* GGX::new_metal(colour, roughness)                 material.rs:290 */
    PT_GGX_DIELECTRIC = 4, /* GGX::new_dielectric(colour, roughness, ior, vol)  material.rs:305 */
    PT_DIELECTRIC = 5      /* Dielectric::new(colour, ior, vol)                 material.rs:475 */
};

typedef struct pt_material_desc
{
    int32_t kind;             /* pt_material_kind */
    float colour[3];          /* albedo / emitted / colour */
    float roughness;          /* linear roughness (GGX): alpha = clamp(roughness^2, 1e-4, 0.9999) */
    float ior;
    int32_t has_volume;       /* Option<Volume>: Volume::new(absorption, k, c, g)  material/volume.rs:136 */
    float vol_absorption[3];
    float vol_k, vol_c, vol_g;
} pt_material_desc;

/* Compile-time constants of src/main.rs:43-51 and src/integrator.rs:10-11 made run-time. */
typedef struct pt_config
{
    uint32_t width, height;   /* IMAGE_WIDTH / IMAGE_HEIGHT   main.rs:44-45 */
    uint32_t max_bounces;     /* MAX_BOUNCES (inclusive)      main.rs:49, integrator.rs:163 */
    uint32_t n_sobol;         /* NUM_POINTS of SobolSampler<N> main.rs:48 (0 => 512) */
    uint32_t enable_nee;      /* ENABLE_NEE                   main.rs:51 */
    uint64_t seed;            /* key of the counter-based per-(pixel,sample) WyRand streams */
    /* multi-GPU row sharding (one process per GPU): rows are dealt to ranks in strips of `strip_rows` */
    uint32_t rank, world_size, strip_rows;
    /* wavefront batch: samples per pixel resident at once (0 => auto) */
    uint32_t batch_spp;
    int32_t device;           /* HIP device ordinal, -1 => current */
    uint32_t flags;           /* PT_FLAG_* */
    /* tuning / test knobs; 0 => the library's default */
    uint32_t stack_lds_levels; /* traversal-stack levels kept in LDS (default 14); deeper levels of deep BVHs spill to HBM */
    uint32_t queue_slack;      /* 0 (the default): shade / terminal queues hold one slot per path + 1/8 + 4 M slots (producers reserve queue
                                  regions and return unused tails as holes).  Non-zero: the slack is EXACTLY this many 1/1024ths of the path
                                  count and the 4 M-slot constant is dropped — so 128 is NOT the default and can be too little for small
                                  batches.  A queue that would overflow is never written past its end: the render returns PT_ERR_LIMIT, and
                                  a render that fails leaves the accumulation RESET (it may have received incomplete samples).  Tests set
                                  bit 31 to see that: the low 16 bits are then the WHOLE capacity in 1/1024ths of the path count. */
    uint32_t pipelines;        /* wavefront batches in flight on separate HIP streams (default 2; 1 = strictly one after another) */
    uint32_t reserved;
} pt_config;

enum
{
    PT_FLAG_TIMING = 1u,       /* bracket every world closest-hit launch with HIP events on the launch stream (ms_trace_closest) */
    PT_FLAG_NO_LDS_SCENE = 2u, /* force BVH reads from global memory even when the scene fits LDS */
    PT_FLAG_TIMING_ALL = 4u,   /* bracket every kernel launch (adds ~10 us of idle per launch; diagnostic) */
    PT_FLAG_NO_PRIMARY_CULL = 8u /* generate and trace the camera rays of EVERY pixel, also where they provably miss the scene's bounds */
};

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
pt_ctx* pt_create(const pt_config* cfg);
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(pt_ctx* ctx);
int pt_set_config(pt_ctx* ctx, const pt_config* cfg); /* change image size / bounces / seed / sharding between renders */

/* ---- Scene::new(Vec<Model>)  src/scene.rs:21 ------------------------------------------------------------------ */
/* returns the material index (>= 0) or a negative pt_status */
int pt_add_material(pt_ctx* ctx, const pt_material_desc* desc);
/* Model::new(path, material, matrices)  model.rs:36, with the OBJ file replaced by its triangle soup
 * (positions / normals: n_tris * 3 vertices * xyz, the Vec<Vertex> load_obj returns, blas.rs:44-131);
 * affine3x4_rowmajor: n_instances rigid transforms.  Returns the model (= BLAS) index or a negative pt_status. */
int pt_add_model(pt_ctx* ctx, const float* positions_xyz, const float* normals_xyz, uint32_t n_tris, int material,
                 const float* affine3x4_rowmajor, uint32_t n_instances);
/* Model::new(path, ...) with the reference's own OBJ reader, load_obj blas.rs:44-131 (v / vn / f with v/vt/vn references,
 * negative indices, fan triangulation, face-normal fallback for normal index 0).  PT_ERR_IO: file unreadable;
 * PT_ERR_PARSE: where the reference would panic. */
int pt_add_model_obj(pt_ctx* ctx, const char* path, int material, const float* affine3x4_rowmajor, uint32_t n_instances);
/* triangle soup of a model as loaded (n_tris*9 floats each); pass cap_tris = 0 to query *n_tris */
int pt_model_vertices(pt_ctx* ctx, int model, float* positions_xyz, float* normals_xyz, uint32_t cap_tris, uint32_t* n_tris);
/* BLAS (SAH sweep, blas_bvh.rs:62-136) + world/light TLAS (agglomerative, tlas_bvh.rs:85-138) + LightSampler
 * (light_sampler.rs:41-61) on the host; flattened for the device.  No GPU is touched until the first render/trace.
 * The trees are the reference's node for node; the SAH sweep of a large model forks its subtrees onto up to 16 host threads
 * (environment PTMI_BUILD_THREADS=n overrides, 1 = the caller's thread only; PTMI_DEBUG_BUILD=1 prints stage times on stderr).
 * PT_ERR_LIMIT: the scene exceeds a packing limit, or the host ran out of memory building it (pt_add_model* likewise): no C++
 * exception crosses this boundary. */
int pt_build(pt_ctx* ctx);

/* ---- Camera::new / create_ray  src/camera.rs:17-31, 94-105 ---------------------------------------------------- */
int pt_set_camera(pt_ctx* ctx, const float eye[3], const float target[3], float fov_y_deg, float aspect);
/* Camera::input (src/camera.rs:56-92) without winit: the caller translates its window events.  PT_EV_MOUSE_MOTION carries
 * DeviceEvent::MouseMotion's delta (a = delta.0, b = delta.1) -> update_rotation (camera.rs:41-54); PT_EV_KEY_W/S/A/D are the
 * pressed-key arms -> update_origin (camera.rs:33-39); a, b are ignored for keys.  dt = seconds since the previous event
 * (main.rs:144).  Returns 1 where the reference returns true (event consumed, camera changed), 0 for any other event code,
 * negative pt_status on error.  The next pt_render / pt_frame uses the moved camera. */
enum pt_event { PT_EV_MOUSE_MOTION = 0, PT_EV_KEY_W = 1, PT_EV_KEY_S = 2, PT_EV_KEY_A = 3, PT_EV_KEY_D = 4 };
int pt_camera_input(pt_ctx* ctx, int event, float a, float b, float dt);
int pt_camera_angles(pt_ctx* ctx, float pitch_yaw[2]); /* the private yaw / pitch fields (camera.rs:8-9), for tests */
int pt_camera_matrices(pt_ctx* ctx, float cam_to_world_3x4[12], float inv_proj_4x4_colmajor[16]);
/* ImageHelper (src/image_helper.rs:13-17) as used on a miss, integrator.rs:256-262: equirect environment, LINEAR rgb,
 * width*height*3 floats, row-major (the gamma-2.2 decode of load_image :25-33 is the caller's).  NULL / 0 restores the
 * reference's `Err` branch: constant ambient 0.006 (integrator.rs:263-266). */
int pt_set_environment(pt_ctx* ctx, uint32_t width, uint32_t height, const float* rgb_linear);
int pt_create_ray(pt_ctx* ctx, float s, float t, float o[3], float d[3]); /* host evaluation, for tests */

/* ---- the per-frame pixel loop  src/main.rs:181-207 + accumulate.wgsl:20-23 ------------------------------------- */
/* Renders samples [first_sample, first_sample+n_samples) of every local pixel on the GPU and adds them, in sample
 * order, to the ctx-owned device accumulation buffer (sum rgb, n).  Outputs (any may be NULL) are host buffers of
 * local_rows*width entries, row-major, row 0 = bottom (camera.rs:96):
 *   data_rgba  : the accumulation buffer after this call (acc.rgb / acc.w is the displayed mean, shader.wgsl:63)
 *   position   : first-hit xyz + t of the LAST sample (main.rs:205)
 *   id         : (id << 16) | new_id applied once per sample (main.rs:206); in/out
 * Blocking.  On failure every launch already enqueued has been waited for and the accumulation buffer and id history are reset to zero
 * (batches pipelined behind a failed one may have added incomplete samples). */
int pt_render(pt_ctx* ctx, uint32_t first_sample, uint32_t n_samples, float* data_rgba, float* position_xyzt, uint32_t* id);
/* The rectangle of pixels camera rays are generated for (host computation, no GPU): columns [rect[0], rect[0]+rect[1]) and LOCAL rows
 * [rect[2], rect[2]+rect[3]) of this rank.  Every camera ray of a pixel outside it misses the world TLAS's root box (returned in
 * root_box as min xyz, max xyz when non-NULL), which is all TLAS::intersect would find out (tlas.rs:68-72): such pixels receive the miss
 * result of integrator.rs:263-266 without a path.  The whole frame when an environment map is set, when the box reaches behind the
 * image plane, or with PT_FLAG_NO_PRIMARY_CULL. */
int pt_active_pixels(pt_ctx* ctx, uint32_t rect[4], float root_box[6]);
/* same, results stay on the device (no host copy); *_dev may be NULL or device pointers of the sizes above */
int pt_render_device(pt_ctx* ctx, uint32_t first_sample, uint32_t n_samples);
int pt_reset_accumulation(pt_ctx* ctx);
/* device pointer of the accumulation buffer (float4 per local pixel) so that the caller (torch.distributed / RCCL
 * gather) can read it without a host round trip; *n_pixels = local pixel count */
int pt_accum_device_ptr(pt_ctx* ctx, void** dev_ptr, uint64_t* n_pixels);
int pt_read_accumulation(pt_ctx* ctx, float* data_rgba);
/* accumulation, last sample's first-hit position and id history as they lie on the device (any pointer may be NULL), e.g. after pt_render_device */
int pt_read_frame(pt_ctx* ctx, float* data_rgba, float* position_xyzt, uint32_t* id);
/* checkpoint / resume across contexts and processes (the progressive accumulation of src/shaders/accumulate.wgsl:20-23):
 * restores what pt_render / pt_read_frame returned — accumulation, and optionally (NULL = leave alone) the last sample's first-hit position and the
 * id history — so that pt_render(first_sample = samples already in `data_rgba`, ...) continues the frame bit for bit */
int pt_write_accumulation(pt_ctx* ctx, const float* data_rgba, const float* position_xyzt, const uint32_t* id);
/* per-sample radiance (rgb,1) of the last pt_render* call's final batch is not kept; this renders n_samples and
 * writes them un-accumulated: out[(s * local_pixels + pixel) * 4] (test hook for bit-exact comparison per sample) */
int pt_render_samples(pt_ctx* ctx, uint32_t first_sample, uint32_t n_samples, float* samples_rgba);
/* rows owned by this rank: local row ly <-> global row rows[ly] */
int pt_local_rows(pt_ctx* ctx, uint32_t* n_rows, uint32_t* rows, uint32_t cap);
/* HIP stream the library launches on (a hipStream_t); pass NULL to return to the library's own stream */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);
int pt_synchronize(pt_ctx* ctx);

/* ---- after the path: State::update / State::render  src/state.rs:505-586, 629-667 ------------------------------ */
/* One iteration of the reference's event loop (main.rs:179-218): the pixel loop for ONE sample (sample index = frame_index),
 * then State::update on the device-resident accumulation texture.  last_inv_projection NULL or equal (f32 ==, state.rs:549) to
 * the current camera's: accumulate.wgsl (acc += (rgb,1)); otherwise velocity.wgsl + compute.wgsl (temporal reprojection) with the PREVIOUS frame's
 * (matrix * inv_projection).inverse(), column-major (main.rs:128,213-216; pt_inv_projection).  data / position / id as the
 * reference hands them to state.update: this frame's (rgb,1), first-hit xyz|t, id history.  Single-rank contexts only. */
int pt_frame(pt_ctx* ctx, uint32_t frame_index, const float* last_inv_projection, float* data_rgba, float* position_xyzt, uint32_t* id);
int pt_inv_projection(pt_ctx* ctx, float out16_colmajor[16]); /* (cam.matrix * cam.inv_projection).inverse() of the current camera */
/* State::render: GT tonemap of accumulation.rgb / accumulation.w (shader.wgsl:3-33,59-64), rgba f32, alpha 1, host buffer */
int pt_present(pt_ctx* ctx, float* rgba);
/* ImageHelper::write_image (src/image_helper.rs:37-58): accumulation.rgb / accumulation.w through tonemapping.rs's GT curve,
 * gamma 1/2.2, x255, Rust `as u8`; W*H*3 bytes in framebuffer row order.  pt_write_image also encodes them as an 8-bit RGB PNG
 * (image::save_buffer(.., ColorType::Rgb8)); an unwritable path returns PT_ERR_IO. */
int pt_present_rgb8(pt_ctx* ctx, uint8_t* rgb);
int pt_write_image(pt_ctx* ctx, const char* path);
/* the same kernels on caller images (host pointers, row-major w*h; rgba f32, velocity 2 x f32, id u32): unit hooks */
int pt_post_velocity(pt_ctx* ctx, uint32_t w, uint32_t h, const float* position, const float* last_inv_projection, float* velocity);
int pt_post_reproject(pt_ctx* ctx, uint32_t w, uint32_t h, const float* input, const float* accum, const float* velocity, const uint32_t* id,
                      float* output);
int pt_post_tonemap(pt_ctx* ctx, uint32_t w, uint32_t h, const float* accum, float* out);
int pt_post_rgb8(pt_ctx* ctx, uint32_t w, uint32_t h, const float* accum, uint8_t* rgb);

/* ---- unit hooks: TLAS::intersect / any_intersect  src/tlas.rs:66, 111 ------------------------------------------ */
/* which: 0 world TLAS, 1 lights TLAS.  Host SoA in, host SoA out.  miss => inst = prim = 0xffffffff, t = +inf.
 * inst = TLAS leaf index in allocation order, prim = triangle index inside its BLAS (load order). */
int pt_trace_closest(pt_ctx* ctx, int which, uint32_t n, const float* o_xyz, const float* d_xyz, const float* t_max, float* t,
                     float* u, float* v, uint32_t* inst, uint32_t* prim);
int pt_trace_any(pt_ctx* ctx, int which, uint32_t n, const float* o_xyz, const float* d_xyz, const float* t_max, uint8_t* hit);
/* get_ss_sobol(index, seed)  src/sampling.rs:97 evaluated on the device */
int pt_ss_sobol(pt_ctx* ctx, uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out_xy);
/* device arithmetic probes: fn 0 sin/cos, 1 exp, 2 ln, 3 hypot(a,b), 4 a/b, 5 sqrt, 7 k-th f32 draw of stream (a=pixel bits,b=sample bits) */
int pt_math_batch(pt_ctx* ctx, int fn, uint32_t n, const float* a, const float* b, float* out0, float* out1);
/* MaterialTrait::scatter_direction + get_bsdf_pdf + get_weakening on the device for n (incoming, normal, front) tuples,
 * drawing from stream (pixel[i], sample[i]) at draws_consumed; out[i*9..] = wo xyz, bsdf rgb, pdf, weakening, draws */
int pt_material_eval(pt_ctx* ctx, int material, uint32_t n, const float* incoming_xyz, const float* normal_xyz, const uint8_t* front,
                     const uint32_t* pixel, const uint32_t* sample, uint32_t draws_consumed, float* out9);

/* ---- host-builder introspection (CPU only; compared against the oracle's builders) ----------------------------- */
int pt_blas_count(pt_ctx* ctx);
int pt_blas_dump(pt_ctx* ctx, int blas, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b,
                 uint32_t* n_prim_ids, uint32_t* prim_ids, uint32_t cap_nodes, uint32_t cap_ids);
int pt_tlas_dump(pt_ctx* ctx, int which, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b,
                 uint32_t cap_nodes);
/* the leaves' `matrix` / `inv_matrix` (tlas_bvh.rs:36-41; inv_matrix = matrix.inverse(), :99) in leaf allocation order — the index a
 * leaf's `a` holds in pt_tlas_dump —, each as 12 floats, rows of the 3x4 */
int pt_tlas_instances(pt_ctx* ctx, int which, uint32_t* n, float* matrix12, float* inv_matrix12, uint32_t cap);
int pt_light_cdf(pt_ctx* ctx, uint32_t* n, float* pdf, float* cdf, uint32_t* blas, uint32_t* prim, float* max_weight, uint32_t cap);
int pt_triangle_dump(pt_ctx* ctx, int blas, uint32_t prim, float out36[36]);

/* ---- several GPUs, one process  (src/main.rs:72,181-207: ONE process, all pixels independent) ---------------------- */
/* The reference fans its pixel loop out over a thread pool inside one process.  pt_multi is that for GPUs: N contexts, one per
 * device, rows dealt to them in strips of pt_config.strip_rows (rank/world_size of `cfg` are overwritten), one host thread per
 * device while rendering, then ONE RCCL gather (ncclCommInitAll + ncclGather, xGMI) of the strip framebuffers to devices[0] and a
 * de-interleave kernel there.  No collective runs while rendering; the counter-based RNG is keyed by the GLOBAL pixel, so the frame
 * is bit-identical for any device count.
 *   devices NULL: 0 .. n_devices-1.  The same device may be listed several times (contexts then share it and the gather is a
 *   device-to-device copy instead of RCCL: RCCL refuses duplicate devices) — that is how one-GPU boxes test the assembly.
 * Scene: make the Scene::new calls (pt_add_material / pt_add_model* / pt_build / pt_set_camera / pt_set_environment /
 * pt_camera_input) on pt_multi_ctx(m, 0); pt_multi_render replicates the built scene, camera and environment to the other
 * contexts whenever they changed.  Every other pt_* call on a member context is allowed but sees only that rank's rows. */
typedef struct pt_multi pt_multi;
pt_multi* pt_multi_create(const pt_config* cfg, const int32_t* devices, uint32_t n_devices);
void pt_multi_destroy(pt_multi* m);
const char* pt_multi_last_error(pt_multi* m);
pt_ctx* pt_multi_ctx(pt_multi* m, uint32_t rank);
/* pt_render for the whole frame: samples [first_sample, first_sample + n_samples) on every device, gather, and (data_rgba != NULL)
 * the accumulated frame, height*width*4 floats, row 0 = bottom, copied to the host.  Blocking. */
int pt_multi_render(pt_multi* m, uint32_t first_sample, uint32_t n_samples, float* data_rgba);
/* the gathered frame of the last pt_multi_render on devices[0] (float4 per pixel) */
int pt_multi_framebuffer_device_ptr(pt_multi* m, void** dev_ptr);
int pt_multi_reset_accumulation(pt_multi* m);
int pt_multi_write_image(pt_multi* m, const char* path); /* pt_write_image of the gathered frame */
/* 1 if the last gather went through RCCL, 0 if it was device-to-device copies (duplicate devices) */
int pt_multi_used_rccl(pt_multi* m);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
typedef struct pt_stats
{
    uint64_t rays_closest;       /* world.intersect call sites      integrator.rs:179 */
    uint64_t rays_any;           /* world.any_intersect call sites  integrator.rs:56,103 */
    uint64_t rays_light_closest; /* lights.intersect call sites     integrator.rs:100 */
    uint64_t paths;
    uint64_t launches_trace_closest; /* world closest-hit kernel launches in the timed region */
    double ms_trace_closest;         /* their summed duration (HIP events on the launch stream; PT_FLAG_TIMING) */
    double ms_trace_any, ms_trace_light, ms_shade, ms_generate, ms_accumulate;
    double ms_total;                 /* wall of the render calls since the last reset, host clock around stream sync */
    uint64_t scene_bytes;            /* nodes + triangles + instances resident for traversal */
    uint32_t lds_scene;              /* 1 if traversal reads the BVH from LDS */
    uint32_t stack_entries;
    uint64_t state_bytes;            /* wavefront state + queues resident in HBM */
    uint64_t rays_light_closest_traced; /* of rays_light_closest, those that went through the lights TLAS (the others miss its root box:
                                        the shading pass answers them with that one slab test, tlas.rs:68-74) */
    uint64_t rays_primary_culled;    /* of rays_closest, camera rays of pixels whose every ray misses the world's root box: answered by
                                        the host's projection of that box onto the image plane, never generated */
} pt_stats;
int pt_get_stats(pt_ctx* ctx, pt_stats* out);
int pt_multi_get_stats(pt_multi* m, pt_stats* sum);   /* counters summed over the devices; ms_total = the slowest device's */
/* counter rows of the LAST wavefront batch, 16 words per bounce (diagnostic): [0] closest-queue slots, [2] shadow-queue slots,
 * [4] NEE-chain slots, [6] NEE-chain rays traced, [7] of those hitting a light, [8..12] shade-queue slots (terminal, lambert,
 * specular, dielectric, ggx), [13] closest rays traced, [14] shadow rays traced.  Slots include the holes producers return. */
int pt_last_batch_counters(pt_ctx* ctx, uint32_t* rows16, uint32_t cap_rows, uint32_t* n_rows);
/* diagnostic (tools/shade_access_bench.py): path ids of slots [first, first + count) of a surface class's shade queue as the LAST batch's
 * last bounce left it (the order its shading pass read them in); 0xffffffff marks a slot no ray took.  qclass: 1 lambert .. 4 ggx */
int pt_last_batch_shade_pids(pt_ctx* ctx, uint32_t qclass, uint32_t first, uint32_t count, uint32_t* pids);
/* diagnostic, meaningful only with a -DPT_STEP_STATS=1 build of the kernels (tools/step_stats.py): per bounce, 8 words: traversal
 * wave-steps of the world closest-hit kernel, lanes active in them, lanes taking the instance / branch / triangle-leaf section, wave-steps
 * in which some lane took the instance / branch / leaf section */
int pt_last_batch_step_stats(pt_ctx* ctx, uint32_t* rows8, uint32_t cap_rows, uint32_t* n_rows);
int pt_reset_stats(pt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
