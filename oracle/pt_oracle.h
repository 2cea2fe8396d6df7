/* ORACLE — TEST INFRASTRUCTURE ONLY (see pto_math.h header).  PARITY UNPINNED.
 *
 * C entry points of the CPU restatement of the reference's per-pixel
 * integration loop.  Used by tests/ (as the checker), by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product
 * (path_tracer_amd/, include/pt_api.h) never includes or links this.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pto_ctx pto_ctx;

enum { PTO_LAMBERTIAN = 0, PTO_EMISSIVE = 1, PTO_SPECULAR = 2, PTO_GGX_METAL = 3, PTO_GGX_DIELECTRIC = 4, PTO_DIELECTRIC = 5 };

typedef struct pto_volume_desc
{
    float absorption[3]; /* Volume::new(absorption, k, c, g)  volume.rs:136 */
    float k, c, g;
    int present;
} pto_volume_desc;

typedef struct pto_render_cfg
{
    uint32_t width, height;
    uint32_t first_sample, n_samples;
    uint32_t max_bounces;  /* inclusive, integrator.rs:163 */
    uint32_t n_sobol;      /* SobolSampler<N> table size, main.rs:48,131 */
    uint64_t seed;
    uint32_t enable_nee;   /* main.rs:51 */
    uint32_t threads;      /* 0 => hardware_concurrency()-1, main.rs:72 */
    uint32_t row_begin, row_end; /* rows [row_begin,row_end) are rendered; 0,0 => all */
} pto_render_cfg;

/* counters[]: 0 closest-hit casts (world), 1 any-hit casts, 2 closest-hit casts (lights TLAS),
 * 3 nodes visited, 4 triangle tests, 5 paths, 6 nodes visited by world closest only,
 * 7 triangle tests by world closest only */
enum { PTO_N_COUNTERS = 8 };

pto_ctx* pto_create(void);
void pto_destroy(pto_ctx*);
int pto_add_material(pto_ctx*, int kind, const float colour[3], float roughness, float ior, const pto_volume_desc* vol);
/* positions/normals: n_tris*3 vertices * xyz; affines: n_inst row-major 3x4 */
int pto_add_model(pto_ctx*, const float* positions, const float* normals, uint32_t n_tris, int material,
                  const float* affines, uint32_t n_inst);
/* Model::new(path, ..) through load_obj, blas.rs:44-131 */
int pto_add_model_obj(pto_ctx*, const char* path, int material, const float* affines, uint32_t n_inst);
int pto_model_vertices(pto_ctx*, int model, float* positions, float* normals, uint32_t cap_tris, uint32_t* n_tris);
int pto_build(pto_ctx*);
int pto_set_camera(pto_ctx*, const float eye[3], const float target[3], float fov_y_deg, float aspect);
/* equirect environment (linear RGB, w*h*3 floats), NULL/0 => constant ambient branch integrator.rs:263-266 */
int pto_set_environment(pto_ctx*, uint32_t w, uint32_t h, const float* rgb);
int pto_camera_matrices(pto_ctx*, float cam_to_world_3x4[12], float inv_proj_4x4[16], float ray_matrix_4x4[16]);
int pto_create_ray(pto_ctx*, float s, float t, float o[3], float d[3]);
int pto_inv_projection(pto_ctx*, float out16_colmajor[16]);
int pto_camera_move(pto_ctx*, float dx, float dz, float dt);      /* Camera::update_origin   camera.rs:33-39 */
int pto_camera_rotate(pto_ctx*, float dx, float dy, float dt);    /* Camera::update_rotation camera.rs:41-54 */
int pto_camera_angles(pto_ctx*, float pitch_yaw[2]);
int pto_primary_ray(pto_ctx*, const pto_render_cfg*, uint32_t pixel, uint32_t sample, float o[3], float d[3]);

int pto_render(pto_ctx*, const pto_render_cfg*, float* accum_rgba, float* position_xyzt, uint32_t* id, uint64_t* counters);
/* per-sample radiance (rgb1) without accumulation: out[(s*H*W + pixel)*4] */
int pto_render_samples(pto_ctx*, const pto_render_cfg*, float* samples_rgba);
int pto_integrate(pto_ctx*, const pto_render_cfg*, const float o[3], const float d[3], uint32_t pixel, uint32_t sample,
                  uint32_t draws_consumed, float colour[4], float position[4], uint8_t* id);

/* which: 0 = world TLAS, 1 = lights TLAS.  hit_*: t,u,v ; inst = TLAS leaf index (allocation order), prim = triangle
 * index inside its BLAS (original order); miss => inst = prim = 0xffffffff */
int pto_trace_closest(pto_ctx*, int which, uint32_t n, const float* o, const float* d, const float* tmax, float* t, float* u,
                      float* v, uint32_t* inst, uint32_t* prim, float* normal_xyz, uint8_t* front);
int pto_trace_any(pto_ctx*, int which, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* hit);

/* canonical BVH dump used to compare the product's host builder with this one.
 * BLAS b: nodes[n][8] = {min xyz, max xyz, kind(0 branch,1 leaf) as float bits, unused}; links[n][2] = (left,right) or
 * (first,count) into prim_ids.  TLAS which: same with leaf links = (instance index, blas index). */
int pto_blas_count(pto_ctx*);
int pto_blas_dump(pto_ctx*, int which, int blas, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a,
                  uint32_t* b, uint32_t* n_prim_ids, uint32_t* prim_ids, uint32_t cap_nodes, uint32_t cap_ids);
int pto_tlas_instances(pto_ctx*, int which, uint32_t* n, float* matrix12, float* inv_matrix12, uint32_t cap);
int pto_tlas_dump(pto_ctx*, int which, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b,
                  uint32_t cap_nodes);
int pto_light_cdf(pto_ctx*, uint32_t* n, float* pdf, float* cdf, uint32_t* blas, uint32_t* prim, float* max_weight, uint32_t cap);
int pto_triangle_dump(pto_ctx*, int which, int blas, uint32_t prim, float out36[36]);

/* after the path (State::update / State::render): accumulate.wgsl, velocity.wgsl, compute.wgsl, shader.wgsl's tonemap.
 * All images row-major w*h; input/accum/output rgba f32, velocity 2 f32, id u32; last_inv_proj 4x4 column-major. */
int pto_post_accumulate(uint32_t w, uint32_t h, const float* input, float* accum);
int pto_post_velocity(uint32_t w, uint32_t h, const float* position, const float* last_inv_proj, float* velocity);
int pto_post_reproject(uint32_t w, uint32_t h, const float* input, const float* accum, const float* velocity, const uint32_t* id, float* output);
int pto_post_tonemap(uint32_t w, uint32_t h, const float* accum, float* out);
int pto_post_rgb8(uint32_t w, uint32_t h, const float* accum, uint8_t* out_rgb);   /* image_helper.rs:41-48, tonemapping.rs */

/* math / sampler hooks for known-answer and device-math parity tests */
void pto_ss_sobol_raw(uint32_t n_points, uint32_t index, uint32_t seed, uint32_t out_shuffled_x_y[3]);
void pto_ss_sobol(uint32_t n_points, uint32_t index, uint32_t seed, float out[2]);
uint32_t pto_sobol_dim1(uint32_t index);
uint32_t pto_low_bias_hash(uint32_t x);
uint32_t pto_lk_hash(uint32_t x, uint32_t seed);
uint64_t pto_wyrand(uint64_t seed, uint32_t k);
uint64_t pto_stream_state0(uint64_t seed, uint32_t pixel, uint32_t sample);
void pto_math_batch(int fn, uint32_t n, const float* a, const float* b, float* out0, float* out1);
/* material hook: evaluates scatter_direction (consuming draws from (seed,pixel,sample) stream at draws_consumed) and
 * get_bsdf_pdf(wi=-incoming, wo=scattered) ; out = wo[3], bsdf[3], pdf, weakening, draws used */
int pto_material_eval(pto_ctx*, int material, const float incoming[3], const float normal[3], int front_facing, uint64_t seed,
                      uint32_t pixel, uint32_t sample, uint32_t draws_consumed, float out[9]);

#ifdef __cplusplus
}
#endif
#endif
