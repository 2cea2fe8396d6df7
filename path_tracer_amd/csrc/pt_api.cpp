// libptmi C-ABI (include/pt_api.h): context, device residency, the wavefront render loop and the unit hooks.
// Host scene construction lives in pt_scene.cpp, kernels in pt_kernels.hip.  No CPU fallback exists: every compute
// entry point needs a HIP device and reports PT_ERR_HIP without one.
#include "../../include/pt_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pt_kernels.h"
#include "pt_png.h"
#include "pt_scene.h"

using namespace pt;

#ifndef PT_SHADE_BLOCKS_PER_CU
#define PT_SHADE_BLOCKS_PER_CU 16  // persistent shading workgroups per CU (same-box sweeps: rounds 1-3 over 4, 6, 8, 12, 16 -> 12; round 4, with the shadow walk inside the Lambertian pass, 5 / 10 / 12 / 15 / 16 / 20: one-pipeline Cornell frame 60.9 / 59.6 / 60.4 / 59.3 / 59.9 / 58.9 ms, 1/4 share 16.6 (12) / 16.1 / 15.9 / 16.0, global-BVH scenes +-0)
#endif
#ifndef PT_JOIN_LATE
#define PT_JOIN_LATE 0 // same-box A/B: joining the side stream only before the shading pass costs +2 ms per frame (the two traversal kernels fight for wave slots), 0.1 ms less for a 1/8 share
#endif
#ifndef PT_INTERLEAVE_FIRST
#define PT_INTERLEAVE_FIRST 1 // the first batches of the pipelines are enqueued bounce by bounce across the pipelines (render_common)
#endif
#ifndef PT_WAVE_TIMES
#define PT_WAVE_TIMES 0
#endif
#ifndef PT_SIDE_MIN_PATHS
#define PT_SIDE_MIN_PATHS (4u << 20)
#endif
#ifndef PT_SIDE_PRIORITY
#define PT_SIDE_PRIORITY 0
#endif
#ifndef PT_SPLIT_MIN_PATHS
#define PT_SPLIT_MIN_PATHS (24u << 20) // BVHs in global memory: a request that fits is cut in two for two pipelines from this many paths on (see render_common)
#endif
#ifndef PT_SPLIT_MIN_PATHS_LDS
#define PT_SPLIT_MIN_PATHS_LDS (24u << 20)
#endif
#ifndef PT_PIPES4_MIN_PATHS
#define PT_PIPES4_MIN_PATHS (~0ull) // four pipelines by default: never (see render_common); pt_config.pipelines = 4 asks for them
#endif
// samples per block of path ids (power of two; 1 = sample-major path ids as in rounds 1-3): RenderParams::blk_log.  Same-box A/B of 1 / 4 / 8 / 16 / 64, ms per
// frame: Cornell 256 spp 67.6 / 67.4 / 66.7 / 67.3 / 67.6, atrium 64 spp 288.4 / 283.8 / 283.1 / 281.4 / 281.4, three spheres 49.95 / 49.2 / 49.0 / 48.7 / 48.7, mixed
// materials 79.05 / 78.3 / 77.1 / 77.2 / 77.3, 82 k / 328 k meshes +-0: the camera-ray launch gains 7-10 % (Cornell 4.61 -> 4.30 ms, atrium 33.2 -> 30.0 ms), the rest ~1 %.
#ifndef PT_PID_BLOCK
#define PT_PID_BLOCK 16
#endif
#ifndef PT_FUSED_TRACE
#define PT_FUSED_TRACE 1
#endif
#ifndef PT_TRACE_BLOCKS_PER_CU_MAX
#define PT_TRACE_BLOCKS_PER_CU_MAX 8
#endif

namespace {

enum TimeCat { T_GEN = 0, T_WORLD, T_ANY, T_LIGHT, T_SHADE, T_ACCUM, T_COUNT };

struct DevBuf
{
    void* p = nullptr;
    size_t bytes = 0;
};

} // namespace

struct pt_ctx
{
    pt_config cfg{};
    HostScene scene;
    std::string err;
    std::mutex mu;

    bool dev_ready = false;
    int device = 0;
    int n_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr; // `stream` is what pipeline 0 launches on (the caller's, after pt_set_stream)
    hipEvent_t ev_start = nullptr;                      // render start on `stream`: what the other pipelines wait for

    // One wavefront pipeline: the state of ONE batch in flight and the streams it is launched on.  Batches are independent until they
    // add their samples to the frame, so consecutive batches alternate between pipelines: the tail of one batch's launches (a few
    // waves finishing the longest rays) overlaps the body of the other's instead of leaving the device idle.
    struct Pipe
    {
        hipStream_t own_stream = nullptr, side_stream = nullptr; // side: the (tiny) BSDF-sampled NEE launch runs beside the shadow-ray launch
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr; // ev_done: this pipeline's last batch has been accumulated
        size_t cap_paths = 0, cap_slots = 0, cap_slots_term = 0;
        uint32_t cap_rows = 0;
        std::vector<DevBuf> pool;
        WavefrontBuffers wb{};
        Counters* h_counters = nullptr; // pinned
        uint32_t* h_heads = nullptr;    // pinned mirror of the claim-cursor lines (they carry the exact ray tallies)
        struct Ev { hipEvent_t a, b; int cat; };
        std::vector<Ev> ev_pool;
        size_t ev_used = 0;
        uint32_t slack_cfg = 0;         // pt_config.queue_slack the pool was sized with
        uint32_t pixels_cfg = 0;        // local pixel count the pool was sized with (first-hit buffers)
        size_t state_bytes = 0;
        bool busy = false;              // a batch has been launched and not yet harvested
        uint32_t busy_rows = 0;
        uint64_t busy_paths = 0, busy_culled = 0;
    };
    static constexpr int kMaxPipes = 4;
    Pipe pipe[kMaxPipes];
    hipStream_t pipe_stream(int i) const { return i == 0 ? stream : pipe[i].own_stream; }

    // scene residency
    bool scene_uploaded = false;
    DevBuf d_spill;
    size_t spill_region_words = 0; // 8-byte words per traversal launch's spill area (d_spill holds two per pipeline)
    DevBuf d_blob, d_tri_shade, d_tri_pos, d_tri_orig, d_materials, d_lights, d_env;
    std::vector<f4> h_env;
    uint32_t env_w = 0, env_h = 0;
    bool env_uploaded = false;
    SceneView sv{};
    bool lds_scene = false;
    uint32_t block_threads = 256, trace_blocks = 1024;
    bool class_present[Q_COUNT] = {true, false, false, false, false};

    // frame
    std::vector<uint32_t> rows; // local row -> global row
    uint32_t local_pixels = 0;
    uint32_t pad_rows = 0;      // rows of the largest strip set (rank 0's)
    uint64_t scene_version = 0; // bumped by pt_build / camera / environment changes (pt_multi replicates on change)
    DevBuf d_accum, d_position, d_id;
    DevBuf d_input, d_velocity, d_output; // State::update textures (pt_frame)

    // stats
    pt_stats stats{};
    int last_pipe = 0; // pipeline whose counters pt_last_batch_counters reports
};

namespace {

int fail(pt_ctx* c, int code, const std::string& msg)
{
    c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                                       \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) return fail((c), PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

void dev_free(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b = DevBuf();
}
int dev_alloc(pt_ctx* c, DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return PT_OK;
    dev_free(b);
    bytes = std::max<size_t>(bytes, 16);
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return PT_OK;
}

void compute_rows(pt_ctx* c)
{
    const pt_config& g = c->cfg;
    c->rows.clear();
    const uint32_t world = std::max(1u, g.world_size), strip = std::max(1u, g.strip_rows);
    for (uint32_t y = 0; y < g.height; ++y)
        if ((y / strip) % world == g.rank) c->rows.push_back(y);
    c->local_pixels = (uint32_t)c->rows.size() * g.width;
    // rank 0 owns the most rows: every rank's strip framebuffer is allocated that large so that one equal-count gather moves them all
    uint32_t r0 = 0;
    for (uint32_t y = 0; y < g.height; ++y) r0 += ((y / strip) % world == 0u) ? 1u : 0u;
    c->pad_rows = r0;
}

int normalise_config(pt_ctx* c, const pt_config* in)
{
    pt_config g = *in;
    if (g.width == 0 || g.height == 0) return fail(c, PT_ERR_ARG, "width/height must be non-zero");
    if (g.n_sobol == 0) g.n_sobol = 512;
    if (g.world_size == 0) g.world_size = 1;
    if (g.strip_rows == 0) g.strip_rows = 4;
    if (g.rank >= g.world_size) return fail(c, PT_ERR_ARG, "rank >= world_size");
    if ((uint64_t)g.width * g.height > 0x7fffffffull) return fail(c, PT_ERR_ARG, "image too large");
    c->cfg = g;
    compute_rows(c);
    return PT_OK;
}

int ensure_device(pt_ctx* c)
{
    if (c->dev_ready)
    {
        // the caller's thread may have switched devices since (one process can hold several contexts)
        HIPCHK(c, hipSetDevice(c->device));
        return PT_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(c, PT_ERR_HIP, "no HIP device available (libptmi has no CPU path)");
    if (c->cfg.device >= 0) HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipGetDevice(&c->device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    for (int i = 0; i < pt_ctx::kMaxPipes; ++i)
    {
        pt_ctx::Pipe& pp = c->pipe[i];
        if (i > 0) HIPCHK(c, hipStreamCreateWithFlags(&pp.own_stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&pp.side_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&pp.ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&pp.ev_join, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&pp.ev_done, hipEventDisableTiming));
    }
    if (!c->stream) c->stream = c->own_stream;
    c->dev_ready = true;
    return PT_OK;
}

int upload_scene(pt_ctx* c)
{
    if (c->scene_uploaded) return ensure_device(c);
    if (!c->scene.built) return fail(c, PT_ERR_STATE, "pt_build has not been called");
    int r = ensure_device(c);
    if (r) return r;
    const FlatScene& f = c->scene.flat;
    const size_t nb = f.nodes.size() * sizeof(DNode), tb = f.tri_isect.size() * sizeof(DTriIsect), ib = f.instances.size() * sizeof(DInstance);
    const size_t lb = (f.big_leaves.size() * 4 + 15) / 16 * 16; // big-leaf table {first, count}, usually empty
    std::vector<uint8_t> blob(nb + tb + ib + lb);
    std::memcpy(blob.data(), f.nodes.data(), nb);
    std::memcpy(blob.data() + nb, f.tri_isect.data(), tb);
    std::memcpy(blob.data() + nb + tb, f.instances.data(), ib);
    if (lb) std::memcpy(blob.data() + nb + tb + ib, f.big_leaves.data(), f.big_leaves.size() * 4);
    if ((r = dev_alloc(c, c->d_blob, blob.size()))) return r;
    HIPCHK(c, hipMemcpy(c->d_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        int rr = dev_alloc(c, b, bytes);
        if (rr) return rr;
        if (bytes) HIPCHK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return PT_OK;
    };
    if ((r = up(c->d_tri_shade, f.tri_shade.data(), f.tri_shade.size() * sizeof(DTriVerts)))) return r;
    if ((r = up(c->d_tri_pos, f.tri_pos.data(), f.tri_pos.size() * sizeof(DTriVerts)))) return r;
    if ((r = up(c->d_tri_orig, f.tri_orig.data(), f.tri_orig.size() * 4))) return r;
    if ((r = up(c->d_materials, f.materials.data(), f.materials.size() * sizeof(DMaterial)))) return r;
    if ((r = up(c->d_lights, f.lights.data(), f.lights.size() * sizeof(DLight)))) return r;

    SceneView& sv = c->sv;
    uint8_t* base = (uint8_t*)c->d_blob.p;
    sv.nodes = (const DNode*)base;
    sv.tri_isect = (const DTriIsect*)(base + nb);
    sv.instances = (const DInstance*)(base + nb + tb);
    sv.tri_shade = (const DTriVerts*)c->d_tri_shade.p;
    sv.tri_pos = (const DTriVerts*)c->d_tri_pos.p;
    sv.tri_orig = (const uint32_t*)c->d_tri_orig.p;
    sv.materials = (const DMaterial*)c->d_materials.p;
    sv.lights = (const DLight*)c->d_lights.p;
    sv.n_nodes = (uint32_t)f.nodes.size();
    sv.n_tris = (uint32_t)f.tri_isect.size();
    sv.n_instances = (uint32_t)f.instances.size();
    sv.n_materials = (uint32_t)f.materials.size();
    sv.n_lights = (uint32_t)f.lights.size();
    sv.world_root = f.world_root;
    sv.lights_root = f.lights_root;
    sv.prim_bits = f.prim_bits;
    sv.light_weight_sum = f.light_weight_sum;
    sv.blob_bytes = (uint32_t)blob.size();
    sv.stack_entries = f.stack_entries;
    sv.has_volumes = f.has_volumes ? 1u : 0u;

    // launch geometry of the traversal kernels: BVH in LDS when it is small, per-lane stacks always in LDS
    c->lds_scene = blob.size() <= 48 * 1024 && !(c->cfg.flags & PT_FLAG_NO_LDS_SCENE);
    uint32_t threads = 256;
#ifndef PT_STACK_LDS_LEVELS
#define PT_STACK_LDS_LEVELS 14
#endif
    sv.stack_lds = std::min<uint32_t>(sv.stack_entries, c->cfg.stack_lds_levels ? c->cfg.stack_lds_levels : PT_STACK_LDS_LEVELS); // deeper levels spill to global memory
    // (the same helper the launchers size their dynamic LDS with: the budget check and the launch cannot disagree)
    auto lds_need = [&](uint32_t t) { return trace_lds_bytes(c->lds_scene, sv.blob_bytes, sv.stack_lds, t); };
    while (threads > 64 && lds_need(threads) > 64 * 1024) threads >>= 1;
    if (lds_need(threads) > 160 * 1024) return fail(c, PT_ERR_LIMIT, "BVH too deep for the LDS traversal stack");
    c->block_threads = threads;
    const size_t lds = lds_need(threads);
    uint32_t per_cu = (uint32_t)std::min<size_t>((160 * 1024) / std::max<size_t>(lds, 1), 2048 / threads);
    per_cu = std::max(1u, std::min(per_cu, (uint32_t)PT_TRACE_BLOCKS_PER_CU_MAX));
    c->trace_blocks = (uint32_t)c->n_cus * per_cu;
    sv.stack_spill = nullptr;
    if (sv.stack_entries > sv.stack_lds)
    {
        // two regions per pipeline: the BSDF-sampled NEE launch runs on the side stream BESIDE the shadow-ray launch (nee_launches),
        // pipelines run beside each other, and the spill slots are indexed by lane only, so concurrent traversal kernels must not share them
        const size_t lanes = (size_t)c->trace_blocks * threads;
        c->spill_region_words = (size_t)(sv.stack_entries - sv.stack_lds) * lanes;
        if ((r = dev_alloc(c, c->d_spill, 2 * pt_ctx::kMaxPipes * c->spill_region_words * 8))) return r;
        sv.stack_spill = (uint64_t*)c->d_spill.p;
    }

    for (uint32_t q = 0; q < Q_COUNT; ++q) c->class_present[q] = (q == Q_TERMINAL);
    for (const DInstance& in : f.instances) c->class_present[in.qclass & 0xffu] = true;
    c->scene_uploaded = true;
    c->stats.scene_bytes = blob.size();
    c->stats.lds_scene = c->lds_scene;
    c->stats.stack_entries = sv.stack_entries;
    return PT_OK;
}

TraceLaunch trace_launch(pt_ctx* c, int pipe = 0, bool side_stream = false)
{
    TraceLaunch tl;
    tl.scene = c->sv;
    if (tl.scene.stack_spill) tl.scene.stack_spill += (size_t)(2 * pipe + (side_stream ? 1 : 0)) * c->spill_region_words;
    tl.blob = c->d_blob.p;
    tl.lds_scene = c->lds_scene;
    tl.grid_blocks = c->trace_blocks;
    tl.n_cus = (uint32_t)c->n_cus;
    tl.block_threads = c->block_threads;
    return tl;
}

// Camera rays are generated only where they can hit something: the world TLAS's root box is projected onto the image plane (its
// eight corners through the inverse of the camera's ray matrix, in binary64) and the pixel rectangle that contains the projection,
// grown by a margin of three pixels, is the ACTIVE rectangle; every camera ray of a pixel outside it misses the root box — which is
// all TLAS::intersect would find out (tlas.rs:68-72) — so its sample is the miss result of integrator.rs:263-266 and k_accumulate
// adds exactly that.  Why this is safe: the root box is convex and in front of the camera, a projective map keeps its image inside
// the hull (hence the bounding rectangle) of the projected corners, a jittered sample lies within half a pixel of its pixel centre,
// and a ray that clears the box by more than two pixels of angle (> 1e-3 rad at 1080p) fails the binary32 slab test by a margin six
// orders above its rounding error.  Not done when an environment map is set (a miss then needs its own direction), when a corner
// is not safely in front of the camera, or with PT_FLAG_NO_PRIMARY_CULL.
struct ActiveRect { uint32_t x0, w, ly0, rows; };
ActiveRect active_rect(pt_ctx* c)
{
    const uint32_t W = c->cfg.width, H = c->cfg.height, local_rows = (uint32_t)c->rows.size();
    const ActiveRect full{0u, W, 0u, local_rows};
    if ((c->cfg.flags & PT_FLAG_NO_PRIMARY_CULL) || c->env_w || !c->scene.built || !c->scene.camera.set) return full;
    const FlatScene& f = c->scene.flat;
    if (f.world_root == MISS_ID || f.world_root >= f.nodes.size()) return full;
    const DNode& root = f.nodes[f.world_root];
    // M maps NDC (nx, ny, 0, 1) to a homogeneous world point on the image plane (k_generate); invert it in binary64
    double m[16], inv[16];
    for (int i = 0; i < 16; ++i) m[i] = c->scene.camera.ray_matrix[i];
    {
        // Gauss-Jordan on the column-major 4x4 (treated as a[row][col] = m[col * 4 + row])
        double a[4][8];
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) { a[r][k] = m[k * 4 + r]; a[r][4 + k] = r == k ? 1.0 : 0.0; }
        for (int col = 0; col < 4; ++col)
        {
            int piv = col;
            for (int r = col + 1; r < 4; ++r)
                if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
            if (std::fabs(a[piv][col]) < 1e-300) return full;
            for (int k = 0; k < 8; ++k) std::swap(a[col][k], a[piv][k]);
            const double d = a[col][col];
            for (int k = 0; k < 8; ++k) a[col][k] /= d;
            for (int r = 0; r < 4; ++r)
                if (r != col)
                {
                    const double fct = a[r][col];
                    for (int k = 0; k < 8; ++k) a[r][k] -= fct * a[col][k];
                }
        }
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) inv[k * 4 + r] = a[r][4 + k];
    }
    const double eye[3] = {c->scene.camera.matrix.t.x, c->scene.camera.matrix.t.y, c->scene.camera.matrix.t.z};
    // view axis = direction of the central ray
    double ctr[4];
    for (int r = 0; r < 4; ++r) ctr[r] = m[12 + r];
    if (!(std::fabs(ctr[3]) > 1e-300)) return full;
    double fwd[3] = {ctr[0] / ctr[3] - eye[0], ctr[1] / ctr[3] - eye[1], ctr[2] / ctr[3] - eye[2]};
    const double fl = std::sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]);
    if (!(fl > 0.0) || !std::isfinite(fl)) return full;
    double ext = 0.0;
    for (int k = 0; k < 3; ++k) { fwd[k] /= fl; ext = std::max(ext, std::fabs((double)root.mx[k] - (double)root.mn[k])); }
    double px_lo = 1e300, px_hi = -1e300, py_lo = 1e300, py_hi = -1e300;
    for (int corner = 0; corner < 8; ++corner)
    {
        const double X[3] = {corner & 1 ? root.mx[0] : root.mn[0], corner & 2 ? root.mx[1] : root.mn[1], corner & 4 ? root.mx[2] : root.mn[2]};
        if (!std::isfinite(X[0]) || !std::isfinite(X[1]) || !std::isfinite(X[2])) return full;
        const double depth = (X[0] - eye[0]) * fwd[0] + (X[1] - eye[1]) * fwd[1] + (X[2] - eye[2]) * fwd[2];
        if (!(depth > 1e-3 * ext + 1e-6)) return full; // the box reaches (nearly) behind the camera plane: no rectangle bounds its image
        double q[4];
        for (int r = 0; r < 4; ++r) q[r] = inv[r] * X[0] + inv[4 + r] * X[1] + inv[8 + r] * X[2] + inv[12 + r];
        if (!(std::fabs(q[3]) > 1e-300)) return full;
        const double nx = q[0] / q[3], ny = q[1] / q[3];
        if (!std::isfinite(nx) || !std::isfinite(ny)) return full;
        const double px = (nx + 1.0) * 0.5 * W, py = (ny + 1.0) * 0.5 * H; // continuous pixel coordinates: pixel gx spans [gx - 0.5, gx + 0.5]
        px_lo = std::min(px_lo, px); px_hi = std::max(px_hi, px);
        py_lo = std::min(py_lo, py); py_hi = std::max(py_hi, py);
    }
    const double margin = 3.0;
    const double gx0 = std::floor(px_lo - 0.5 - margin), gx1 = std::ceil(px_hi + 0.5 + margin) + 1.0;
    const double gy0 = std::floor(py_lo - 0.5 - margin), gy1 = std::ceil(py_hi + 0.5 + margin) + 1.0;
    const uint32_t x0 = (uint32_t)std::min<double>(std::max(gx0, 0.0), W), x1 = (uint32_t)std::min<double>(std::max(gx1, 0.0), W);
    const uint32_t y0 = (uint32_t)std::min<double>(std::max(gy0, 0.0), H), y1 = (uint32_t)std::min<double>(std::max(gy1, 0.0), H);
    // local rows are in ascending global order: those inside [y0, y1) are contiguous
    uint32_t ly0 = 0, n = 0;
    while (ly0 < local_rows && c->rows[ly0] < y0) ++ly0;
    while (ly0 + n < local_rows && c->rows[ly0 + n] < y1) ++n;
    return ActiveRect{x0, x1 > x0 ? x1 - x0 : 0u, ly0, n};
}

int ensure_frame(pt_ctx* c)
{
    int r;
    const size_t px = std::max<size_t>(std::max<size_t>(c->local_pixels, (size_t)c->pad_rows * c->cfg.width), 1);
    const bool fresh = c->d_accum.bytes < px * 16;
    if ((r = dev_alloc(c, c->d_accum, px * 16))) return r;
    if ((r = dev_alloc(c, c->d_position, px * 16))) return r;
    if ((r = dev_alloc(c, c->d_id, px * 4))) return r;
    if (fresh)
    {
        HIPCHK(c, hipMemsetAsync(c->d_accum.p, 0, c->d_accum.bytes, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_position.p, 0, c->d_position.bytes, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_id.p, 0, c->d_id.bytes, c->stream));
    }
    return PT_OK;
}

void free_pipe_pool(pt_ctx::Pipe& pp)
{
    for (DevBuf& b : pp.pool) dev_free(b);
    pp.pool.clear();
    if (pp.h_counters) { (void)hipHostFree(pp.h_counters); pp.h_counters = nullptr; }
    if (pp.h_heads) { (void)hipHostFree(pp.h_heads); pp.h_heads = nullptr; }
    pp.cap_paths = 0;
    pp.cap_rows = 0;
    pp.wb = WavefrontBuffers{};
}

int ensure_wavefront(pt_ctx* c, int pipe, size_t n_paths, uint32_t rows)
{
    pt_ctx::Pipe& pp = c->pipe[pipe];
    const bool vstack_ok = !c->sv.has_volumes || pp.wb.st.vstack != nullptr;
    bool queues_ok = true; // a scene edit may have introduced a material class the pool has no shade queue for
    for (uint32_t q = 1; q < Q_COUNT; ++q) queues_ok = queues_ok && (!c->class_present[q] || ((pp.wb.class_mask >> q) & 1u));
    if (pp.cap_paths >= n_paths && pp.cap_rows >= rows && vstack_ok && queues_ok && pp.slack_cfg == c->cfg.queue_slack && pp.pixels_cfg == c->local_pixels) return PT_OK;
    free_pipe_pool(pp);
    pp.slack_cfg = c->cfg.queue_slack;
    pp.pixels_cfg = c->local_pixels;
    n_paths = std::max<size_t>(n_paths, 64);
    // The ray queues are dense — a shading workgroup reserves exactly what it appends (block_append4) and a path has at most one ray
    // per queue and bounce — so they hold n_paths slots.  The shade queues and the terminal queue hold slots, not entries: a
    // traversal wave reserves regions (64…8192 slots: wave_reserve, wave_reserve_striped) and leaves the tail of its last one as
    // holes, and the shorter stripes leave gaps; both stay below n/8 + 8192 * 64 per queue.  The slack is a sizing rule, not a safety
    // margin: a producer that finds a queue full diverts to the queue's dump area and the batch fails with PT_ERR_LIMIT
    // (pt_config.queue_slack shrinks the slack so that tests can see exactly that).
    const size_t frac = c->cfg.queue_slack ? (c->cfg.queue_slack & 0xffffu) : 128;
    const size_t fixed = c->cfg.queue_slack ? 0 : ((size_t)4 << 20);
    const size_t n_slots = n_paths;
    const size_t n_slots_term = n_paths + n_paths * frac / 1024 + fixed;
    // test mode: the surface shade queues are SMALLER than the batch (the ray queues cannot be: the camera rays of a batch fill one)
    const size_t n_slots_shade = (c->cfg.queue_slack & 0x80000000u) ? std::max<size_t>(n_paths * frac / 1024, 64) : n_paths + n_paths * frac / 1024 + fixed;
    if (n_slots_term + kQueueDumpSlots >= (1ull << 32)) return fail(c, PT_ERR_LIMIT, "batch too large for 32-bit queue slots");
    pp.cap_slots = n_slots;
    pp.cap_slots_term = n_slots_term;
    size_t total = 0;
    auto take = [&](size_t bytes, void** out) -> int {
        DevBuf b;
        int r = dev_alloc(c, b, bytes);
        if (r) return r;
        pp.pool.push_back(b);
        *out = b.p;
        total += b.bytes;
        return PT_OK;
    };
    int r;
    WavefrontBuffers& w = pp.wb;
    const size_t qs = n_slots + kQueueDumpSlots, qs_term = n_slots_term + kQueueDumpSlots; // allocated slots incl. the dump area
#define TAKE(ptr, bytes)                                  \
    if ((r = take((bytes), (void**)&(ptr)))) return r;
    TAKE(w.st.rec, n_paths * sizeof(DPathRec));
    TAKE(w.st.radiance, n_paths * 16);
    TAKE(w.st.occl, n_paths);
    if (c->sv.has_volumes) { TAKE(w.st.vstack, n_paths * 4); }
    else w.st.vstack = nullptr;
    // first-hit position / id: only the batch's last sample / last two samples are ever read (RenderParams::keep_*_from)
    TAKE(w.st.first_pos, (size_t)std::max<uint32_t>(c->local_pixels, 64) * 16);
    TAKE(w.st.first_id, (size_t)std::max<uint32_t>(c->local_pixels, 64) * 2 * 4);
    for (int k = 0; k < 2; ++k)
    {
        TAKE(w.rq[k].a, qs * 16);
        TAKE(w.rq[k].b, qs * 16);
        TAKE(w.rq_lchain[k].a, qs * 16);
        TAKE(w.rq_lchain[k].b, qs * 16);
        TAKE(w.lchain_nb[k], qs * 16);
        TAKE(w.q_term[k], qs_term * 8);
    }
    TAKE(w.rq_shadow.a, qs * 16);
    TAKE(w.rq_shadow.b, qs * 16);
    TAKE(w.hits, qs * 16);
    TAKE(w.lchain_hit, qs * 16);
    w.class_mask = 0;
    w.q_class_slot = 0;
    uint32_t n_classes = 0;
    for (uint32_t q = 1; q < Q_COUNT; ++q)
    {
        if (!c->class_present[q]) continue;
        w.class_mask |= 1u << q;
        w.q_class_slot |= n_classes << (4u * q);
        ++n_classes;
    }
    w.q_stride = (uint32_t)(n_slots_shade + kQueueDumpSlots);
    TAKE(w.q_shade_base, (size_t)std::max(n_classes, 1u) * 3 * w.q_stride * 16);
    TAKE(w.counters, (size_t)rows * sizeof(Counters));
    TAKE(w.heads, (size_t)rows * HEADS_PER_ROW * kHeadWordsPerQueue * 4);
    TAKE(w.tails, (size_t)rows * Q_COUNT * kTailWordsPerQueue * 4);
#if PT_WAVE_TIMES
    TAKE(w.wave_times, (size_t)rows * kWaveTimeSlots * 16);
    TAKE(w.wave_times_any, (size_t)rows * kWaveTimeSlots * 16);
#endif
#undef TAKE
    w.cap_slots = (uint32_t)n_slots;
    w.cap_slots_shade = (uint32_t)n_slots_shade;
    w.cap_slots_term = (uint32_t)n_slots_term;
    HIPCHK(c, hipHostMalloc((void**)&pp.h_counters, (size_t)rows * sizeof(Counters), hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&pp.h_heads, (size_t)rows * HEADS_PER_ROW * kHeadWordsPerQueue * 4, hipHostMallocDefault));
    pp.cap_paths = n_paths;
    pp.cap_rows = rows;
    pp.state_bytes = total;
    c->stats.state_bytes = 0;
    for (const pt_ctx::Pipe& q : c->pipe) c->stats.state_bytes += q.cap_paths ? q.state_bytes : 0;
    return PT_OK;
}

struct Timer
{
    pt_ctx* c;
    pt_ctx::Pipe& pp;
    hipStream_t stream;
    int cat;
    bool on;
    size_t slot = 0;
    Timer(pt_ctx* c_, pt_ctx::Pipe& pp_, hipStream_t s_, int cat_)
        : c(c_), pp(pp_), stream(s_), cat(cat_), on((c_->cfg.flags & PT_FLAG_TIMING_ALL) != 0 || (cat_ == T_WORLD && (c_->cfg.flags & PT_FLAG_TIMING) != 0))
    {
        if (!on) return;
        if (pp.ev_used == pp.ev_pool.size())
        {
            pt_ctx::Pipe::Ev e;
            (void)hipEventCreate(&e.a);
            (void)hipEventCreate(&e.b);
            pp.ev_pool.push_back(e);
        }
        slot = pp.ev_used++;
        pp.ev_pool[slot].cat = cat;
        (void)hipEventRecord(pp.ev_pool[slot].a, stream);
    }
    ~Timer()
    {
        if (on) (void)hipEventRecord(pp.ev_pool[slot].b, stream);
    }
};

void harvest_events(pt_ctx* c, pt_ctx::Pipe& pp)
{
    double ms[T_COUNT] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < pp.ev_used; ++i)
    {
        float t = 0;
        if (hipEventElapsedTime(&t, pp.ev_pool[i].a, pp.ev_pool[i].b) == hipSuccess) ms[pp.ev_pool[i].cat] += t;
        if (pp.ev_pool[i].cat == T_WORLD) c->stats.launches_trace_closest++;
    }
    pp.ev_used = 0;
    c->stats.ms_generate += ms[T_GEN];
    c->stats.ms_trace_closest += ms[T_WORLD];
    c->stats.ms_trace_any += ms[T_ANY];
    c->stats.ms_trace_light += ms[T_LIGHT];
    c->stats.ms_shade += ms[T_SHADE];
    c->stats.ms_accumulate += ms[T_ACCUM];
}

// what a launched batch leaves behind: overflow flags and exact ray tallies.  Blocks until the pipeline's stream is idle.
int harvest_batch(pt_ctx* c, int pipe)
{
    pt_ctx::Pipe& pp = c->pipe[pipe];
    if (!pp.busy) return PT_OK;
    pp.busy = false;
    HIPCHK(c, hipStreamSynchronize(c->pipe_stream(pipe)));
    HIPCHK(c, hipGetLastError());
    const uint32_t rows = pp.busy_rows;
    for (uint32_t r = 0; r < rows; ++r)
        // a producer that found a queue full has diverted its entries to the queue's dump area (nothing was written out of bounds)
        // and raised this flag: the batch's results are incomplete
        if (pp.h_counters[r].overflow)
        {
            pp.ev_used = 0; // the abandoned batch's event pairs must not be added to the next render's timers
            return fail(c, PT_ERR_LIMIT, "a wavefront queue was full (pt_config.queue_slack too small for this scene); the batch was abandoned, no memory was overwritten");
        }
    // exact tallies live beside the claim cursors, one per cursor line (64 addresses per queue instead of one: a launch of a few
    // thousand waves that each add to ONE word spends ~50 us on that alone)
    for (uint32_t r = 0; r < rows; ++r)
    {
        const uint32_t* hrow = pp.h_heads + (size_t)r * HEADS_PER_ROW * kHeadWordsPerQueue;
        for (uint32_t g = 0; g < kQueueHeads; ++g)
        {
            const uint32_t* cl = hrow + HEADS_CLOSEST * kHeadWordsPerQueue + g * kHeadStrideWords;
            const uint32_t* sh = hrow + HEADS_SHADOW * kHeadWordsPerQueue + g * kHeadStrideWords;
            const uint32_t* lc = hrow + HEADS_LCHAIN * kHeadWordsPerQueue + g * kHeadStrideWords;
            c->stats.rays_closest += cl[HEAD_TALLY0];
            c->stats.rays_any += (uint64_t)sh[HEAD_TALLY0] + lc[HEAD_TALLY1];
            c->stats.rays_light_closest_traced += lc[HEAD_TALLY0];
            c->stats.rays_light_closest += (uint64_t)lc[HEAD_TALLY0] + lc[HEAD_TALLY2]; // casts of integrator.rs:100, whoever answered them
        }
    }
    c->stats.paths += pp.busy_paths;
    // camera rays of pixels outside the active rectangle: cast (integrator.rs:179) and answered by the projection of the world's root box
    c->stats.rays_closest += pp.busy_culled;
    c->stats.rays_primary_culled += pp.busy_culled;
    c->last_pipe = pipe;
    harvest_events(c, pp);
    return PT_OK;
}

// one wavefront batch: samples [first, first+count) of every local pixel, launched on pipeline `pipe` (nothing here waits for the
// device except the emptiness probes of very long bounce budgets).  `after`: event the batch's accumulation must wait for (the
// previous batch's accumulation: samples are added to the frame in sample order), or null.
// One batch's launches, cut at the bounces so that the batches of several pipelines can be enqueued side by side (render_common):
// batch_begin, batch_bounce(0..max_bounces), batch_end.  Everything is asynchronous on the pipeline's streams except the
// every-fourth-bounce look at the counters that long bounce budgets use to stop early.
enum : uint32_t { kEagerRows = 18u }; // bookkeeping rows cleared when a batch begins (bounces 0..16); deeper ones are cleared as the bounces are enqueued
struct BatchRun
{
    pt_ctx* c = nullptr;
    int pipe = 0;
    RenderParams rp{};
    CameraView cam{};
    EnvView env{};
    TraceLaunch tl{}, tl_side{};
    hipStream_t s = nullptr;
    uint32_t rows = 0, shade_blocks = 1, last_row = 0, count = 0, cleared_rows = 0;
    bool nee = false, side_busy = false, stopped = false, write_position = false, aux_with_samples = false, fused = false;
    bool no_shadow_queue = false; // every shadow ray of the scene is answered inside the shading pass (shade_traces_shadow): launch_trace_shadow has nothing to do
    int nee_err = PT_OK;
    f4* samples_out = nullptr;
    hipEvent_t after = nullptr;
};

int batch_begin(BatchRun& br, pt_ctx* c, int pipe, uint32_t first_sample, uint32_t count, bool write_position, f4* samples_out, bool aux_with_samples, hipEvent_t after)
{
    pt_ctx::Pipe& pp = c->pipe[pipe];
    const pt_config& g = c->cfg;
    br = BatchRun{};
    br.c = c;
    br.pipe = pipe;
    br.count = count;
    br.write_position = write_position;
    br.samples_out = samples_out;
    br.aux_with_samples = aux_with_samples;
    br.after = after;
    RenderParams& rp = br.rp;
    rp.width = g.width;
    rp.height = g.height;
    rp.local_rows = (uint32_t)c->rows.size();
    rp.local_pixels = c->local_pixels;
    rp.rank = g.rank;
    rp.world_size = g.world_size;
    rp.strip_rows = g.strip_rows;
    rp.first_sample = first_sample;
    rp.batch_samples = count;
    const ActiveRect ar = active_rect(c);
    rp.act_x0 = ar.x0; rp.act_w = ar.w; rp.act_ly0 = ar.ly0; rp.act_rows = ar.rows;
    rp.act_pixels = ar.w * ar.rows;
    rp.n_paths = rp.act_pixels * count;
    rp.max_bounces = g.max_bounces;
    rp.n_sobol = g.n_sobol;
    rp.enable_nee = g.enable_nee;
    rp.keep_s_id = count >= 2 ? count - 2 : 0u;
    rp.keep_s_pos = count - 1;
    rp.seed = g.seed;
    // path ids in blocks of PT_PID_BLOCK samples (RenderParams::blk_log); a batch shorter than a block is one short block
    rp.blk_log = 0;
    while ((2u << rp.blk_log) <= (uint32_t)PT_PID_BLOCK && (2u << rp.blk_log) <= count) ++rp.blk_log;
    rp.n_blk = (count + (1u << rp.blk_log) - 1u) >> rp.blk_log;
    rp.blk_last = count - ((rp.n_blk - 1u) << rp.blk_log);
    rp.div_blk_paths = fastdiv_make(rp.act_pixels << rp.blk_log);
    rp.div_blk_last = fastdiv_make(rp.blk_last);
    rp.div_act_w = fastdiv_make(rp.act_w);
    rp.div_width = fastdiv_make(rp.width);
    rp.div_strip_rows = fastdiv_make(rp.strip_rows);
    br.rows = g.max_bounces + 2;
    br.last_row = br.rows - 1;
    br.s = c->pipe_stream(pipe);
    const WavefrontBuffers& wb = pp.wb;
    br.tl = trace_launch(c, pipe, false);
    br.tl_side = trace_launch(c, pipe, true);
    std::memcpy(br.cam.ray_matrix, c->scene.camera.ray_matrix, 64);
    br.cam.eye[0] = c->scene.camera.matrix.t.x;
    br.cam.eye[1] = c->scene.camera.matrix.t.y;
    br.cam.eye[2] = c->scene.camera.matrix.t.z;
    if (c->env_w)
    {
        br.env.data = (const f4*)c->d_env.p; // uploaded by the caller (ensure_environment)
        br.env.w = c->env_w;
        br.env.h = c->env_h;
    }
    // per-bounce bookkeeping (64 B of counters + 24 KB of claim cursors + 40 KB of striped tails per row): the first kEagerRows rows are
    // cleared here, the rows of deeper bounces one bounce ahead of their first use (batch_bounce) — the reference's default budget of
    // 1024 bounces would otherwise cost 67 MB of memset and a 25 MB read-back per batch, also for an interactive 1-spp frame
    br.cleared_rows = std::min<uint32_t>(br.rows, kEagerRows);
    HIPCHK(c, hipMemsetAsync(wb.counters, 0, (size_t)br.cleared_rows * sizeof(Counters), br.s));
    HIPCHK(c, hipMemsetAsync(wb.heads, 0, (size_t)br.cleared_rows * HEADS_PER_ROW * kHeadWordsPerQueue * 4, br.s));
    HIPCHK(c, hipMemsetAsync(wb.tails, 0, (size_t)br.cleared_rows * Q_COUNT * kTailWordsPerQueue * 4, br.s));
#if PT_WAVE_TIMES
    HIPCHK(c, hipMemsetAsync(wb.wave_times, 0, (size_t)br.rows * kWaveTimeSlots * 16, br.s));
    HIPCHK(c, hipMemsetAsync(wb.wave_times_any, 0, (size_t)br.rows * kWaveTimeSlots * 16, br.s));
#endif
    if (rp.n_paths) { Timer t(c, pp, br.s, T_GEN); launch_generate(br.s, rp, br.cam, wb); }
    br.shade_blocks = (uint32_t)std::max<size_t>(1, std::min<size_t>(((size_t)rp.n_paths + 255) / 256, (size_t)c->n_cus * PT_SHADE_BLOCKS_PER_CU));
    br.nee = g.enable_nee != 0;
    // PT_FUSED_TRACE: the BSDF-sampled NEE rays of a bounce ride in the next bounce's world closest-hit launch (k_trace_fused) instead of
    // a launch of their own on the side stream.  Same-box A/B: LDS-resident scenes whole frame +-0, 1/4 share 18.45 -> 18.3 ms, 1-spp
    // frame 1.24 -> 1.18 ms; BVHs in global memory lose 3 % (82 k mesh 14.2 -> 14.65 ms, 328 k 25.4 -> 26.3 ms: their NEE rays are long and
    // used to overlap the shadow-ray launch), so only the former.  Per-launch event timing (PT_FLAG_TIMING_ALL) keeps the launches apart.
    br.fused = PT_FUSED_TRACE != 0 && c->lds_scene && !(g.flags & PT_FLAG_TIMING_ALL);
    // the Lambertian shading pass of an LDS-resident scene walks its own shadow rays; GGX surfaces (the only other class that casts them) still queue theirs
    br.no_shadow_queue = shade_traces_shadow(br.tl) && !c->class_present[Q_GGX];
    return PT_OK;
}

// The BSDF-sampled NEE launch of a bounce has almost nothing to do since shading answers the rays that miss the lights' root box
// (it is all launch latency and tail).  It runs on a side stream beside the shadow-ray launch; the main stream waits for it before
// the next closest-hit launch (waiting only before the next SHADING pass, the first reader of its results, was measured: +2 ms per
// frame, the two traversal kernels fight for wave slots).
void batch_nee_launches(BatchRun& br, uint32_t row)
{
    pt_ctx* c = br.c;
    pt_ctx::Pipe& pp = c->pipe[br.pipe];
    const WavefrontBuffers& wb = pp.wb;
    const bool timing_all = (c->cfg.flags & PT_FLAG_TIMING_ALL) != 0; // per-launch events want one stream
    // a small batch (an interactive 1-spp frame) gains nothing from the second stream and pays ~20 us per bounce for the two
    // cross-stream event waits: its two launches go out one after the other
    if (timing_all || br.rp.n_paths < (uint32_t)PT_SIDE_MIN_PATHS)
    {
        if (!br.no_shadow_queue) { Timer t(c, pp, br.s, T_ANY); launch_trace_shadow(br.s, br.tl, wb, row); }
        { Timer t(c, pp, br.s, T_LIGHT); launch_trace_lchain(br.s, br.tl, wb, row); }
        return;
    }
    if (hipEventRecord(pp.ev_fork, br.s) != hipSuccess || hipStreamWaitEvent(pp.side_stream, pp.ev_fork, 0) != hipSuccess) br.nee_err = PT_ERR_HIP;
    launch_trace_lchain(pp.side_stream, br.tl_side, wb, row);
    if (hipEventRecord(pp.ev_join, pp.side_stream) != hipSuccess) br.nee_err = PT_ERR_HIP;
    br.side_busy = true;
    if (!br.no_shadow_queue) launch_trace_shadow(br.s, br.tl, wb, row);
}

void batch_join_side(BatchRun& br)
{
    if (!br.side_busy) return;
    if (hipStreamWaitEvent(br.s, br.c->pipe[br.pipe].ev_join, 0) != hipSuccess) br.nee_err = PT_ERR_HIP;
    br.side_busy = false;
}

int batch_bounce(BatchRun& br, uint32_t b)
{
    if (br.stopped) return PT_OK;
    pt_ctx* c = br.c;
    pt_ctx::Pipe& pp = c->pipe[br.pipe];
    const pt_config& g = c->cfg;
    const WavefrontBuffers& wb = pp.wb;
    hipStream_t s = br.s;
    // bounce b touches rows b and b + 1 (the shading pass appends to the next bounce's queues)
    while (br.cleared_rows < std::min<uint32_t>(br.rows, b + 2u))
    {
        const uint32_t row = br.cleared_rows++;
        HIPCHK(c, hipMemsetAsync(wb.counters + row, 0, sizeof(Counters), s));
        HIPCHK(c, hipMemsetAsync(wb.heads + (size_t)row * HEADS_PER_ROW * kHeadWordsPerQueue, 0, (size_t)HEADS_PER_ROW * kHeadWordsPerQueue * 4, s));
        HIPCHK(c, hipMemsetAsync(wb.tails + (size_t)row * Q_COUNT * kTailWordsPerQueue, 0, (size_t)Q_COUNT * kTailWordsPerQueue * 4, s));
    }
    if (b > 0 && br.nee && br.fused)
    {
        // the shadow rays of the bounce before, then ONE launch for this bounce's world closest hit and the bounce before's BSDF-sampled NEE rays
        if (!br.no_shadow_queue) { Timer t(c, pp, s, T_ANY); launch_trace_shadow(s, br.tl, wb, b - 1); }
        Timer t(c, pp, s, T_WORLD);
        launch_trace_fused(s, br.tl, wb, b, br.rp, br.env);
    }
    else
    {
        if (b > 0 && br.nee) batch_nee_launches(br, b - 1);
#if !PT_JOIN_LATE
        batch_join_side(br);
#endif
        { Timer t(c, pp, s, T_WORLD); launch_trace_world(s, br.tl, wb, b, br.rp, br.cam, br.env); }
        batch_join_side(br);
    }
    for (uint32_t q = 0; q < Q_COUNT; ++q)
        if (c->class_present[q]) { Timer t(c, pp, s, T_SHADE); launch_shade(s, q, c->sv, br.rp, wb, b, br.shade_blocks, br.cam, br.env, &br.tl); }
    // long bounce budgets (reference default MAX_BOUNCES = 1024): stop once no path is left
    if (g.max_bounces > 16 && b >= 8 && (b % 4) == 0 && b < g.max_bounces)
    {
        HIPCHK(c, hipMemcpyAsync(pp.h_counters + b + 1, wb.counters + b + 1, sizeof(Counters), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const Counters& nx = pp.h_counters[b + 1];
        if (nx.n_closest == 0) { br.last_row = b + 1; br.stopped = true; }
    }
    return PT_OK;
}

int batch_end(BatchRun& br)
{
    pt_ctx* c = br.c;
    pt_ctx::Pipe& pp = c->pipe[br.pipe];
    const pt_config& g = c->cfg;
    const WavefrontBuffers& wb = pp.wb;
    hipStream_t s = br.s;
    const RenderParams& rp = br.rp;
    if (br.last_row == br.rows - 1) br.last_row = g.max_bounces + 1;
    // the last shading pass may still owe direct-light estimates: trace them, then a resolve-only terminal pass
    if (br.nee)
    {
        batch_nee_launches(br, br.last_row - 1);
        batch_join_side(br);
        { Timer t(c, pp, s, T_SHADE); launch_shade(s, Q_TERMINAL, c->sv, rp, wb, br.last_row, br.shade_blocks, br.cam, br.env); }
    }
    if (br.nee_err) return fail(c, PT_ERR_HIP, "stream fork/join failed");
    if (br.after && hipStreamWaitEvent(s, br.after, 0) != hipSuccess) return fail(c, PT_ERR_HIP, "hipStreamWaitEvent");
    if (br.samples_out)
    {
        launch_store_samples(s, rp, wb, br.samples_out);
        // pt_frame: the frame's own colour goes to the input texture, position / id history are still updated
        if (br.aux_with_samples) launch_accumulate(s, rp, br.cam, wb, (f4*)c->d_accum.p, (f4*)c->d_position.p, (uint32_t*)c->d_id.p, 1u, 0u);
    }
    else
    {
        Timer t(c, pp, s, T_ACCUM);
        launch_accumulate(s, rp, br.cam, wb, (f4*)c->d_accum.p, (f4*)c->d_position.p, (uint32_t*)c->d_id.p, br.write_position ? 1u : 0u, 1u);
    }
    HIPCHK(c, hipEventRecord(pp.ev_done, s));
    // only the rows the batch used come back (all of them were cleared: last_row + 1 <= cleared_rows)
    const uint32_t used_rows = std::min<uint32_t>(br.rows, std::min<uint32_t>(br.last_row + 1u, br.cleared_rows));
    HIPCHK(c, hipMemcpyAsync(pp.h_counters, wb.counters, (size_t)used_rows * sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(pp.h_heads, wb.heads, (size_t)used_rows * HEADS_PER_ROW * kHeadWordsPerQueue * 4, hipMemcpyDeviceToHost, s));
    pp.busy = true;
    pp.busy_rows = used_rows;
    pp.busy_paths = (uint64_t)rp.local_pixels * br.count;
    pp.busy_culled = (uint64_t)(rp.local_pixels - rp.act_pixels) * br.count;
    return PT_OK;
}

int launch_batch(pt_ctx* c, int pipe, uint32_t first_sample, uint32_t count, bool write_position, f4* samples_out, bool aux_with_samples, hipEvent_t after)
{
    BatchRun br;
    int r;
    if ((r = batch_begin(br, c, pipe, first_sample, count, write_position, samples_out, aux_with_samples, after))) return r;
    for (uint32_t b = 0; b <= c->cfg.max_bounces; ++b)
        if ((r = batch_bounce(br, b))) return r;
    return batch_end(br);
}

int ensure_environment(pt_ctx* c)
{
    if (!c->env_w || c->env_uploaded) return PT_OK;
    int er = dev_alloc(c, c->d_env, c->h_env.size() * sizeof(f4));
    if (er) return er;
    HIPCHK(c, hipMemcpy(c->d_env.p, c->h_env.data(), c->h_env.size() * sizeof(f4), hipMemcpyHostToDevice));
    c->env_uploaded = true;
    return PT_OK;
}

// one batch on pipeline 0, start to finish
int run_batch(pt_ctx* c, uint32_t first_sample, uint32_t count, bool write_position, f4* samples_out, bool aux_with_samples = false)
{
    int r;
    if ((r = ensure_environment(c))) return r;
    if ((r = launch_batch(c, 0, first_sample, count, write_position, samples_out, aux_with_samples, nullptr))) return r;
    return harvest_batch(c, 0);
}

int precheck(pt_ctx* c)
{
    if (!c->scene.built) return fail(c, PT_ERR_STATE, "pt_build has not been called");
    if (!c->scene.camera.set) return fail(c, PT_ERR_STATE, "pt_set_camera has not been called");
    if (c->cfg.enable_nee && c->scene.flat.lights.empty()) return fail(c, PT_ERR_STATE, "NEE is enabled but the scene has no emissive model");
    return PT_OK;
}

int render_common(pt_ctx* c, uint32_t first_sample, uint32_t n_samples, float* samples_out)
{
    int r;
    if ((r = precheck(c))) return r;
    if (n_samples == 0 || c->local_pixels == 0) return PT_OK;
    if ((r = upload_scene(c))) return r;
    if ((r = ensure_frame(c))) return r;
    if ((r = ensure_environment(c))) return r;
    // Batches: by default the whole request stays resident when HBM allows; otherwise (or with pt_config.batch_spp) it is cut into
    // equal batches that alternate between `pipelines` wavefront pipelines on their own HIP streams, so that one batch's launch
    // tails overlap the other's launches.  ~PT_BYTES_PER_PATH of wavefront state per path; path ids are 29-bit.
    uint32_t want_pipes = samples_out ? 1u : std::min<uint32_t>(c->cfg.pipelines ? c->cfg.pipelines : 2u, (uint32_t)pt_ctx::kMaxPipes);
    size_t max_paths = (size_t)96 << 20;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        {
            size_t held = 0;
            for (const pt_ctx::Pipe& pp : c->pipe)
                for (const DevBuf& b : pp.pool) held += b.bytes;
            // bytes of wavefront state per path (ensure_wavefront): 323 in records, ray queues and terminal queues + 54 per surface class present
            // in the scene (its 48-byte shade queue with an eighth of slack) + 4 with volumes.  (Rounds 1-3 assumed 410 whatever the scene:
            // right for one class, but the four-class atrium then asked for 268 GiB of a 268.2 GiB device.)
            uint32_t n_classes = 0;
            for (uint32_t q = 1; q < Q_COUNT; ++q) n_classes += c->class_present[q] ? 1u : 0u;
            const double per_path = 330.0 + 56.0 * std::max(n_classes, 1u) + (c->sv.has_volumes ? 4.0 : 0.0) + 24.0;
            max_paths = (size_t)((double)(free_b + held) * 0.85 / per_path);
        }
        max_paths = std::min<size_t>(std::max<size_t>(max_paths, 1u << 20), (1ull << 29) - 1);
    }
    const ActiveRect ar = active_rect(c);
    const size_t act_pixels = std::max<size_t>((size_t)ar.w * ar.rows, 1);
    uint32_t batch = c->cfg.batch_spp ? c->cfg.batch_spp : (uint32_t)std::max<size_t>(1, max_paths / act_pixels);
    batch = std::min(batch, n_samples);
    uint32_t n_batches = (n_samples + batch - 1) / batch;
    // (A request that fits at once runs as ONE batch on one pipeline.  Cutting a small one — a rank's share of a sharded frame — in two
    // for two pipelines used to hide its launch tails (-8 %); since the tails were shortened at the source (striped tails, tapered
    // chunks) it costs 5 % instead: twice the launches, and two persistent kernels fighting for the same wave slots.)
    // ... with one exception: BVHs in global memory.  Their rays are long, the launches end in long tails, and smaller batches on several
    // pipelines overlap them (same box, 82 k-triangle mesh at 1080p unless noted):
    //   32 spp   one batch 43.9 ms   2 x 16 on two pipelines 40.8   4 x 8 on four 44.9
    //   128 spp                      2 x 64: 149.1                  4 x 32: 149.3
    //   512 spp  one batch 629.5     2 x 256: 580 (4 x 128 on TWO pipelines: 583-588)   3 x 171 on three: 572   4 x 128 on four: 561
    //   328 k mesh, 1024 spp         2 x 512: 1903                  3 x 342: 1889       4 x 256: 1846
    //   8 spp (4 M paths) 14.21 -> 14.26: nothing; three spheres, 8 spp (7.8 M paths): 10.46 -> 9.95 (four pipelines: 11.3)
    // so (rounds 2-3, FOUR waves per SIMD): two pipelines from 6 M paths on.  Round 4, FIVE waves per SIMD — the kernels hide more of their own latency, a second
    // persistent grid has less to fill — same box, one pipeline against the split: 82 k mesh 64 spp 67.2 -> 64.9 ms, 512 spp 497 -> 475; 328 k mesh 64 spp 115.1 -> 113.5,
    // 1024 spp (bench.py) 1663 -> 1637; atrium 256 spp 1127-1141 -> 1112-1114, but 64 spp 280.7 -> 288.2 and 8 spp 38.7 -> 39.5; spheres +-0: the split went off ...
    // ... and came back once the surface shading kernels ran at five waves too (one pipeline's shading beside the other's traversal), same box, one batch against
    // two halves: atrium 16 / 64 spp 74.1 -> 71.8 / 281.6 -> 272.7 ms, 82 k mesh 16 / 64 spp 20.4 -> 20.3 / 64.7 -> 62.7, 328 k mesh 64 spp 113.5 -> 108.6, three spheres
    // 16 / 64 spp 15.1 -> 15.2 / 49.4 -> 47.7: from PT_SPLIT_MIN_PATHS = 24 M paths on, as for LDS scenes.  (Requests that do not fit always alternated between two
    // pipelines.)  Four (from PT_PIPES4_MIN_PATHS on) did not hold up under bench.py: 82 k mesh at 512 spp
    // 573 -> 588 ms, 328 k mesh at 1024 spp 1979 -> 1959 ms, three spheres at 4096^2 x 1024 spp 12.15 -> 12.43 s; off.
    const uint64_t total_paths = (uint64_t)n_samples * act_pixels;
    if (!samples_out && !c->cfg.pipelines && !c->lds_scene && total_paths >= (uint64_t)PT_PIPES4_MIN_PATHS) want_pipes = (uint32_t)pt_ctx::kMaxPipes;
    // (Round 4: with five waves per SIMD in the traversal AND the surface shading kernels one pipeline's shading pass runs beside the other's traversal, and the split pays
    // from a quarter of the headline frame on: whole frame 62.4 -> 59.0 ms, rank 0's half 32.4 -> 30.5, its quarter 17.1 -> 16.4, its eighth (16.6 M paths) 9.44 -> 9.38:
    // from PT_SPLIT_MIN_PATHS_LDS = 24 M paths on.)  Round 3:
    // LDS-resident BVHs gain only on very large requests (with round 3's launch structure): the 133 M-path headline frame 68.3 -> 67.2 ms
    // (bench.py, three interleaved repeats), mixed materials at 66 M paths 42.9 -> 42.3 ms, but rank 0's half of the sharded frame (66 M
    // paths) 35.0 -> 35.7 ms and its quarter +-0: from PT_SPLIT_MIN_PATHS_LDS paths on.
    if (!c->cfg.batch_spp && n_batches < want_pipes && want_pipes >= 2 && total_paths >= (c->lds_scene ? (uint64_t)PT_SPLIT_MIN_PATHS_LDS : (uint64_t)PT_SPLIT_MIN_PATHS))
    {
        n_batches = std::min<uint32_t>(want_pipes, n_samples);
        batch = (n_samples + n_batches - 1) / n_batches;
    }
    uint32_t n_pipes = std::min(want_pipes, n_batches);
    if (!c->cfg.batch_spp && n_batches > 1 && (uint64_t)batch * act_pixels * n_pipes > max_paths)
    {
        // the request does not fit at once: the pipelines share the memory
        batch = (uint32_t)std::max<size_t>(1, max_paths / n_pipes / act_pixels);
        n_batches = (n_samples + batch - 1) / batch;
    }
    if (!c->cfg.batch_spp && n_batches > 1)
    {
        // pools that an earlier request left (a 64-spp warm-up before a 4096-spp render, say) are kept when they are nearly large enough:
        // rounding n_samples / n_batches up differently must not cost a reallocation of hundreds of GB (seconds) inside a render
        size_t cap_min = ~(size_t)0;
        for (uint32_t i = 0; i < n_pipes; ++i) cap_min = std::min(cap_min, c->pipe[i].cap_paths);
        const size_t want = (size_t)batch * act_pixels;
        if (cap_min >= act_pixels && cap_min < want && cap_min * 4 >= want * 3)
        {
            batch = (uint32_t)(cap_min / act_pixels);
            n_batches = (n_samples + batch - 1) / batch;
        }
    }
    batch = (n_samples + n_batches - 1) / n_batches;
    if ((uint64_t)batch * act_pixels >= (1ull << 29)) return fail(c, PT_ERR_ARG, "batch too large (path ids are 29-bit)");
    // pipelines this call does not use give their memory back
    for (int i = (int)n_pipes; i < pt_ctx::kMaxPipes; ++i)
        if (!c->pipe[i].busy && c->pipe[i].cap_paths) free_pipe_pool(c->pipe[i]);
    if (std::getenv("PTMI_DEBUG_BATCH"))
        std::fprintf(stderr, "[ptmi] render %u spp: max_paths %zu act_pixels %zu batch %u x %u on %u pipelines; pools before: %zu %zu paths\n", n_samples, max_paths, act_pixels, batch,
                     n_batches, n_pipes, c->pipe[0].cap_paths, c->pipe[1].cap_paths);
    // ... and a pipeline that kept a larger pool from an earlier request (a whole frame resident on pipeline 0, say) gives it back when
    // the pipelines have to share the budget: its old pool plus the others' new ones could exceed what max_paths was computed from
    const size_t need_paths = (size_t)batch * act_pixels;
    if (n_pipes > 1)
        for (uint32_t i = 0; i < n_pipes; ++i)
            if (!c->pipe[i].busy && c->pipe[i].cap_paths > need_paths + need_paths / 4) free_pipe_pool(c->pipe[i]);
    for (uint32_t i = 0; i < n_pipes; ++i)
        if ((r = ensure_wavefront(c, (int)i, need_paths, c->cfg.max_bounces + 2))) return r;
    DevBuf d_samples;
    if (samples_out && (r = dev_alloc(c, d_samples, (size_t)batch * c->local_pixels * 16))) return r;
    const auto t0 = std::chrono::steady_clock::now();
    // the other pipelines start after whatever the caller queued on pipeline 0's stream (accumulation resets, uploads)
    if (n_pipes > 1)
    {
        HIPCHK(c, hipEventRecord(c->ev_start, c->stream));
        for (uint32_t i = 1; i < n_pipes; ++i) HIPCHK(c, hipStreamWaitEvent(c->pipe_stream((int)i), c->ev_start, 0));
    }
    hipEvent_t prev_done = nullptr;
    int err = PT_OK;
    // The first n_pipes batches find every pipeline idle; their launches are enqueued bounce by bounce ACROSS the pipelines:
    // enqueueing a whole batch takes the host about a millisecond, which the second pipeline of a short request (one rank's share
    // of a sharded frame is two batches) would otherwise spend idle, and the first would run its tail alone at the end.  Later
    // batches go out whole, each as soon as its pipeline's previous batch is done, which keeps the pipelines out of step.
    uint32_t done = 0, k = 0;
    if (n_pipes > 1 && PT_INTERLEAVE_FIRST)
    {
        BatchRun run[pt_ctx::kMaxPipes];
        for (uint32_t i = 0; i < n_pipes && done < n_samples && !err; ++i, ++k)
        {
            const uint32_t cnt = std::min(batch, n_samples - done);
            if ((err = harvest_batch(c, (int)i))) break;
            err = batch_begin(run[i], c, (int)i, first_sample + done, cnt, done + cnt == n_samples, nullptr, false, nullptr);
            done += cnt;
        }
        for (uint32_t b = 0; b <= c->cfg.max_bounces && !err; ++b)
            for (uint32_t i = 0; i < k && !err; ++i) err = batch_bounce(run[i], b);
        for (uint32_t i = 0; i < k && !err; ++i)
        {
            run[i].after = prev_done; // accumulation stays in sample order
            if ((err = batch_end(run[i]))) break;
            prev_done = c->pipe[i].ev_done;
        }
    }
    for (; done < n_samples && !err; done += batch, ++k)
    {
        const int pi = (int)(k % n_pipes);
        const uint32_t cnt = std::min(batch, n_samples - done);
        if ((err = harvest_batch(c, pi))) break; // the pipeline's previous batch must be done before its buffers are reused
        if ((err = launch_batch(c, pi, first_sample + done, cnt, done + cnt == n_samples, samples_out ? (f4*)d_samples.p : nullptr, false, prev_done))) break;
        prev_done = c->pipe[pi].ev_done;
        if (samples_out)
        {
            if ((err = harvest_batch(c, pi))) break;
            hipError_t e = hipMemcpy(samples_out + (size_t)done * c->local_pixels * 4, d_samples.p, (size_t)cnt * c->local_pixels * 16, hipMemcpyDeviceToHost);
            if (e != hipSuccess) err = fail(c, PT_ERR_HIP, hipGetErrorString(e));
        }
    }
    for (uint32_t i = 0; i < n_pipes; ++i)
    {
        const int hr = harvest_batch(c, (int)i);
        if (hr && !err) err = hr;
    }
    // whatever the caller queues next on pipeline 0's stream comes after the last accumulation (harvest_batch has waited for every
    // pipeline, so this is already true for the host; the event keeps stream order explicit for callers that share the stream)
    if (!err && prev_done && n_pipes > 1) HIPCHK(c, hipStreamWaitEvent(c->stream, prev_done, 0));
    if (err)
    {
        // Launches of the failed batch (and of batches pipelined behind it) may still be running on the pipelines' streams, and on
        // PT_ERR_LIMIT some batches have added incomplete samples to the frame: wait for everything, then drop the partial sums so that
        // nothing stale can be read back as a result (pt_api.h: the accumulation is reset by a failed render).
        for (int i = 0; i < pt_ctx::kMaxPipes; ++i)
        {
            (void)hipStreamSynchronize(c->pipe_stream(i));
            if (c->pipe[i].side_stream) (void)hipStreamSynchronize(c->pipe[i].side_stream);
            c->pipe[i].busy = false;
            c->pipe[i].ev_used = 0; // (batches that were never harvested: their timers are void as well)
        }
        (void)hipGetLastError();
        if (c->d_accum.p) (void)hipMemsetAsync(c->d_accum.p, 0, c->d_accum.bytes, c->stream);
        if (c->d_id.p) (void)hipMemsetAsync(c->d_id.p, 0, c->d_id.bytes, c->stream);
        (void)hipStreamSynchronize(c->stream);
        dev_free(d_samples);
        return err;
    }
    dev_free(d_samples);
    c->stats.ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

} // namespace

// ====================================================================================================== C-ABI
extern "C" {

pt_ctx* pt_create(const pt_config* cfg)
{
    if (!cfg) return nullptr;
    pt_ctx* c = new (std::nothrow) pt_ctx();
    if (!c) return nullptr;
    if (normalise_config(c, cfg) != PT_OK)
    {
        delete c;
        return nullptr;
    }
    return c;
}

void pt_destroy(pt_ctx* c)
{
    if (!c) return;
    if (c->dev_ready)
    {
        (void)hipStreamSynchronize(c->stream);
        for (int i = 0; i < pt_ctx::kMaxPipes; ++i)
        {
            pt_ctx::Pipe& pp = c->pipe[i];
            (void)hipStreamSynchronize(c->pipe_stream(i));
            free_pipe_pool(pp);
            for (auto& e : pp.ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
            if (pp.own_stream) (void)hipStreamDestroy(pp.own_stream);
            if (pp.side_stream) (void)hipStreamDestroy(pp.side_stream);
            if (pp.ev_fork) (void)hipEventDestroy(pp.ev_fork);
            if (pp.ev_join) (void)hipEventDestroy(pp.ev_join);
            if (pp.ev_done) (void)hipEventDestroy(pp.ev_done);
        }
        DevBuf* bufs[] = {&c->d_input, &c->d_velocity, &c->d_output, &c->d_blob, &c->d_tri_shade, &c->d_tri_pos, &c->d_tri_orig, &c->d_materials, &c->d_lights, &c->d_env, &c->d_spill, &c->d_accum, &c->d_position, &c->d_id};
        for (DevBuf* b : bufs) dev_free(*b);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    }
    delete c;
}

const char* pt_last_error(pt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int pt_set_config(pt_ctx* c, const pt_config* cfg)
{
    if (!c || !cfg) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    const uint32_t old_px = c->local_pixels;
    const int old_dev = c->cfg.device;
    const pt_config old = c->cfg;
    int r = normalise_config(c, cfg);
    if (r) return r;
    if (c->cfg.stack_lds_levels != old.stack_lds_levels || ((c->cfg.flags ^ old.flags) & PT_FLAG_NO_LDS_SCENE)) c->scene_uploaded = false;
    if (c->dev_ready && cfg->device != old_dev && cfg->device >= 0) return fail(c, PT_ERR_STATE, "device cannot change after first use");
    if (c->local_pixels != old_px)
    {
        dev_free(c->d_accum);
        dev_free(c->d_position);
        dev_free(c->d_id);
    }
    return PT_OK;
}

int pt_add_material(pt_ctx* c, const pt_material_desc* d)
{
    if (!c || !d) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r = c->scene.add_material(d->kind, d->colour, d->roughness, d->ior, d->has_volume != 0, d->vol_absorption, d->vol_k, d->vol_c, d->vol_g);
    if (r < 0) return fail(c, PT_ERR_ARG, "bad material description");
    c->scene_uploaded = false;
    return r;
}

int pt_add_model(pt_ctx* c, const float* positions, const float* normals, uint32_t n_tris, int material, const float* affines, uint32_t n_inst)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    try { r = c->scene.add_model(positions, normals, n_tris, material, affines, n_inst); }
    catch (const std::exception& e) { return fail(c, PT_ERR_LIMIT, std::string("model does not fit in host memory: ") + e.what()); }
    if (r == -4) return fail(c, PT_ERR_NONRIGID, "Model matrix can only contain translation and rotation");
    if (r < 0) return fail(c, PT_ERR_ARG, "bad model description");
    c->scene_uploaded = false;
    return r;
}

int pt_add_model_obj(pt_ctx* c, const char* path, int material, const float* affines, uint32_t n_inst)
{
    if (!c || !path) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    std::string err;
    int r;
    try { r = c->scene.add_model_obj(path, material, affines, n_inst, &err); }
    catch (const std::exception& e) { return fail(c, PT_ERR_LIMIT, std::string("model does not fit in host memory: ") + e.what()); }
    if (r == -4) return fail(c, PT_ERR_NONRIGID, "Model matrix can only contain translation and rotation");
    if (r == -6) return fail(c, PT_ERR_IO, err);
    if (r == -7) return fail(c, PT_ERR_PARSE, err);
    if (r < 0) return fail(c, PT_ERR_ARG, "bad model description");
    c->scene_uploaded = false;
    return r;
}

int pt_model_vertices(pt_ctx* c, int model, float* positions, float* normals, uint32_t cap_tris, uint32_t* n_tris)
{
    if (!c || !n_tris) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (model < 0 || model >= (int)c->scene.models.size()) return fail(c, PT_ERR_ARG, "model index");
    const HostModel& m = c->scene.models[model];
    *n_tris = m.n_tris;
    if (cap_tris == 0) return PT_OK;
    if (cap_tris < m.n_tris || !positions || !normals) return fail(c, PT_ERR_ARG, "capacity");
    std::memcpy(positions, m.positions.data(), (size_t)m.n_tris * 36);
    std::memcpy(normals, m.normals.data(), (size_t)m.n_tris * 36);
    return PT_OK;
}

int pt_build(pt_ctx* c)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->scene.materials.size() > 255) return fail(c, PT_ERR_LIMIT, "at most 255 materials (volume stacks hold 8-bit material indices)");
    std::string err;
    int r;
    // the host builders allocate (and fork threads): a failed allocation is an error code for the caller, not an exception through a C ABI
    try { r = c->scene.build(&err); }
    catch (const std::exception& e) { c->scene.built = false; return fail(c, PT_ERR_LIMIT, std::string("scene build failed on the host: ") + e.what()); }
    c->scene_uploaded = false;
    c->scene_version++;
    if (r) return fail(c, r == -5 ? PT_ERR_LIMIT : PT_ERR_STATE, err);
    return PT_OK;
}

int pt_set_camera(pt_ctx* c, const float eye[3], const float target[3], float fov_y_deg, float aspect)
{
    if (!c || !eye || !target) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->scene.set_camera(eye, target, fov_y_deg, aspect);
    c->scene_version++;
    return PT_OK;
}

int pt_camera_input(pt_ctx* c, int event, float a, float b, float dt)                   // Camera::input  camera.rs:56-92
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->scene.camera.set) return fail(c, PT_ERR_STATE, "pt_set_camera has not been called");
    switch (event)
    {
    case PT_EV_MOUSE_MOTION: c->scene.camera_rotate(a, b, dt); c->scene_version++; return 1;
    case PT_EV_KEY_W: c->scene.camera_move(0.0f, 1.0f, dt); c->scene_version++; return 1;
    case PT_EV_KEY_S: c->scene.camera_move(0.0f, -1.0f, dt); c->scene_version++; return 1;
    case PT_EV_KEY_A: c->scene.camera_move(-1.0f, 0.0f, dt); c->scene_version++; return 1;
    case PT_EV_KEY_D: c->scene.camera_move(1.0f, 0.0f, dt); c->scene_version++; return 1;
    default: return 0;
    }
}

int pt_camera_angles(pt_ctx* c, float pitch_yaw[2])
{
    if (!c || !pitch_yaw) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->scene.camera.set) return fail(c, PT_ERR_STATE, "pt_set_camera has not been called");
    pitch_yaw[0] = c->scene.camera.pitch;
    pitch_yaw[1] = c->scene.camera.yaw;
    return PT_OK;
}

int pt_set_environment(pt_ctx* c, uint32_t width, uint32_t height, const float* rgb_linear)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->env_uploaded = false;
    c->scene_version++;
    c->h_env.clear();
    c->env_w = c->env_h = 0;
    if (!rgb_linear || width == 0 || height == 0) return PT_OK;
    c->h_env.resize((size_t)width * height);
    for (size_t i = 0; i < c->h_env.size(); ++i) c->h_env[i] = f4{rgb_linear[3 * i], rgb_linear[3 * i + 1], rgb_linear[3 * i + 2], 0.0f};
    c->env_w = width;
    c->env_h = height;
    return PT_OK;
}

int pt_camera_matrices(pt_ctx* c, float m34[12], float inv_proj[16])
{
    if (!c || !c->scene.camera.set) return PT_ERR_STATE;
    const xf34& a = c->scene.camera.matrix;
    const float rows[12] = {a.m.c0.x, a.m.c1.x, a.m.c2.x, a.t.x, a.m.c0.y, a.m.c1.y, a.m.c2.y, a.t.y, a.m.c0.z, a.m.c1.z, a.m.c2.z, a.t.z};
    if (m34) std::memcpy(m34, rows, sizeof(rows));
    if (inv_proj) std::memcpy(inv_proj, c->scene.camera.inv_proj, 64);
    return PT_OK;
}

int pt_create_ray(pt_ctx* c, float s, float t, float o[3], float d[3])
{
    if (!c || !c->scene.camera.set) return PT_ERR_STATE;
    c->scene.create_ray(s, t, o, d);
    return PT_OK;
}

int pt_active_pixels(pt_ctx* c, uint32_t rect[4], float root_box[6])
{
    if (!c || !rect) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = precheck(c))) return r;
    const ActiveRect ar = active_rect(c);
    rect[0] = ar.x0; rect[1] = ar.w; rect[2] = ar.ly0; rect[3] = ar.rows;
    if (root_box)
    {
        const FlatScene& f = c->scene.flat;
        if (f.world_root == MISS_ID) return fail(c, PT_ERR_STATE, "empty scene");
        const DNode& n = f.nodes[f.world_root];
        for (int k = 0; k < 3; ++k) { root_box[k] = n.mn[k]; root_box[3 + k] = n.mx[k]; }
    }
    return PT_OK;
}

int pt_render_device(pt_ctx* c, uint32_t first_sample, uint32_t n_samples)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return render_common(c, first_sample, n_samples, nullptr);
}

int pt_render(pt_ctx* c, uint32_t first_sample, uint32_t n_samples, float* data, float* position, uint32_t* id)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = precheck(c))) return r;
    if ((r = upload_scene(c))) return r;
    if ((r = ensure_frame(c))) return r;
    const size_t px = c->local_pixels;
    if (id && px) HIPCHK(c, hipMemcpyAsync(c->d_id.p, id, px * 4, hipMemcpyHostToDevice, c->stream));
    if ((r = render_common(c, first_sample, n_samples, nullptr))) return r;
    if (data && px) HIPCHK(c, hipMemcpyAsync(data, c->d_accum.p, px * 16, hipMemcpyDeviceToHost, c->stream));
    if (position && px) HIPCHK(c, hipMemcpyAsync(position, c->d_position.p, px * 16, hipMemcpyDeviceToHost, c->stream));
    if (id && px) HIPCHK(c, hipMemcpyAsync(id, c->d_id.p, px * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_render_samples(pt_ctx* c, uint32_t first_sample, uint32_t n_samples, float* samples)
{
    if (!c || !samples) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return render_common(c, first_sample, n_samples, samples);
}

int pt_reset_accumulation(pt_ctx* c)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    HIPCHK(c, hipMemsetAsync(c->d_accum.p, 0, c->d_accum.bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_id.p, 0, c->d_id.bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_accum_device_ptr(pt_ctx* c, void** p, uint64_t* n)
{
    if (!c || !p) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    *p = c->d_accum.p;
    if (n) *n = c->local_pixels;
    return PT_OK;
}

int pt_read_accumulation(pt_ctx* c, float* data)
{
    if (!c || !data) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    HIPCHK(c, hipMemcpyAsync(data, c->d_accum.p, (size_t)c->local_pixels * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// the frame's whole state as it lies on the device (any pointer may be null): what pt_write_accumulation takes back
int pt_read_frame(pt_ctx* c, float* data, float* position, uint32_t* id)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    const size_t px = c->local_pixels;
    if (data && px) HIPCHK(c, hipMemcpyAsync(data, c->d_accum.p, px * 16, hipMemcpyDeviceToHost, c->stream));
    if (position && px) HIPCHK(c, hipMemcpyAsync(position, c->d_position.p, px * 16, hipMemcpyDeviceToHost, c->stream));
    if (id && px) HIPCHK(c, hipMemcpyAsync(id, c->d_id.p, px * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// Restore what pt_render / pt_read_frame returned into this context, so that a render stopped in one process continues in
// another: accumulation (sum of rgb, sample count), and optionally the last sample's first-hit position and the id history
// (accumulate.wgsl:20-23, main.rs:204-206).  Local pixels, row-major, like every other framebuffer of the API.
int pt_write_accumulation(pt_ctx* c, const float* data, const float* position, const uint32_t* id)
{
    if (!c || !data) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the frame buffers may still be written by a render in flight
    HIPCHK(c, hipMemcpyAsync(c->d_accum.p, data, (size_t)c->local_pixels * 16, hipMemcpyHostToDevice, c->stream));
    if (position) HIPCHK(c, hipMemcpyAsync(c->d_position.p, position, (size_t)c->local_pixels * 16, hipMemcpyHostToDevice, c->stream));
    if (id) HIPCHK(c, hipMemcpyAsync(c->d_id.p, id, (size_t)c->local_pixels * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_local_rows(pt_ctx* c, uint32_t* n_rows, uint32_t* rows, uint32_t cap)
{
    if (!c || !n_rows) return PT_ERR_ARG;
    *n_rows = (uint32_t)c->rows.size();
    if (rows)
        for (uint32_t i = 0; i < cap && i < c->rows.size(); ++i) rows[i] = c->rows[i];
    return PT_OK;
}

int pt_set_stream(pt_ctx* c, void* stream)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = stream ? (hipStream_t)stream : c->own_stream;
    return PT_OK;
}

int pt_synchronize(pt_ctx* c)
{
    if (!c) return PT_ERR_ARG;
    if (!c->dev_ready) return PT_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// ---- after the path: State::update / State::render
int pt_inv_projection(pt_ctx* c, float out16[16])
{
    if (!c || !out16) return PT_ERR_ARG;
    if (!c->scene.camera.set) return fail(c, PT_ERR_STATE, "pt_set_camera has not been called");
    c->scene.inv_projection(out16);
    return PT_OK;
}

int pt_frame(pt_ctx* c, uint32_t frame_index, const float* last_inv_projection, float* data, float* position, uint32_t* id)
{
    if (!c) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = precheck(c))) return r;
    if (c->cfg.world_size != 1) return fail(c, PT_ERR_STATE, "pt_frame needs the whole frame on one rank (3x3 neighbourhoods cross row strips)");
    if ((r = upload_scene(c))) return r;
    if ((r = ensure_frame(c))) return r;
    const size_t px = c->local_pixels;
    if ((r = dev_alloc(c, c->d_input, px * 16)) || (r = dev_alloc(c, c->d_velocity, px * 8)) || (r = dev_alloc(c, c->d_output, px * 16))) return r;
    if (id) HIPCHK(c, hipMemcpyAsync(c->d_id.p, id, px * 4, hipMemcpyHostToDevice, c->stream));
    if ((r = ensure_wavefront(c, 0, px, c->cfg.max_bounces + 2))) return r;
    if ((r = run_batch(c, frame_index, 1, true, (f4*)c->d_input.p, true))) return r;   // main.rs:181-207, one sample per pixel
    hipStream_t s = c->stream;
    const int w = (int)c->cfg.width, h = (int)c->cfg.height;
    bool moved = false;                                                                                                    // state.rs:549 `inv_projection == last_inv_projection`
    if (last_inv_projection)
    {
        float cur[16];
        c->scene.inv_projection(cur);
        for (int i = 0; i < 16; ++i) moved |= !(cur[i] == last_inv_projection[i]);
    }
    if (!moved) launch_post_accumulate(s, (uint32_t)px, (const f4*)c->d_input.p, (f4*)c->d_accum.p);           // state.rs:561-566
    else
    {
        launch_post_velocity(s, w, h, (const f4*)c->d_position.p, last_inv_projection, (float*)c->d_velocity.p);            // state.rs:569-572
        launch_post_reproject(s, w, h, (const f4*)c->d_input.p, (const f4*)c->d_accum.p, (const float*)c->d_velocity.p,
                              (const uint32_t*)c->d_id.p, (f4*)c->d_output.p);                                              // state.rs:574-578
        HIPCHK(c, hipMemcpyAsync(c->d_accum.p, c->d_output.p, px * 16, hipMemcpyDeviceToDevice, s));                        // state.rs:583
    }
    if (data) HIPCHK(c, hipMemcpyAsync(data, c->d_input.p, px * 16, hipMemcpyDeviceToHost, s));
    if (position) HIPCHK(c, hipMemcpyAsync(position, c->d_position.p, px * 16, hipMemcpyDeviceToHost, s));
    if (id) HIPCHK(c, hipMemcpyAsync(id, c->d_id.p, px * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    return PT_OK;
}

int pt_present(pt_ctx* c, float* rgba)
{
    if (!c || !rgba) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    const size_t px = c->local_pixels;
    if ((r = dev_alloc(c, c->d_output, px * 16))) return r;
    launch_post_tonemap(c->stream, (uint32_t)px, (const f4*)c->d_accum.p, (f4*)c->d_output.p);
    HIPCHK(c, hipMemcpyAsync(rgba, c->d_output.p, px * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

static int present_rgb8_locked(pt_ctx* c, std::vector<uint8_t>& host)
{
    int r;
    if ((r = ensure_device(c))) return r;
    if ((r = ensure_frame(c))) return r;
    const size_t px = c->local_pixels;
    if ((r = dev_alloc(c, c->d_output, px * 16))) return r;
    launch_post_rgb8(c->stream, (uint32_t)px, (const f4*)c->d_accum.p, (uint8_t*)c->d_output.p);
    host.resize(px * 3);
    HIPCHK(c, hipMemcpyAsync(host.data(), c->d_output.p, px * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_present_rgb8(pt_ctx* c, uint8_t* rgb)
{
    if (!c || !rgb) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    std::vector<uint8_t> host;
    int r = present_rgb8_locked(c, host);
    if (r) return r;
    std::memcpy(rgb, host.data(), host.size());
    return PT_OK;
}

int pt_write_image(pt_ctx* c, const char* path)
{
    if (!c || !path) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->cfg.world_size != 1) return fail(c, PT_ERR_STATE, "pt_write_image needs the whole frame on one rank (gather first)");
    std::vector<uint8_t> host;
    int r = present_rgb8_locked(c, host);
    if (r) return r;
    std::string err;
    if (!write_png_rgb8(path, host.data(), c->cfg.width, c->cfg.height, &err)) return fail(c, PT_ERR_IO, err.c_str());
    return PT_OK;
}

namespace {
struct TmpBufs
{
    std::vector<DevBuf> b;
    ~TmpBufs() { for (DevBuf& x : b) dev_free(x); }
    int up(pt_ctx* c, const void* src, size_t bytes, void** out)
    {
        DevBuf d;
        int r = dev_alloc(c, d, bytes);
        if (r) return r;
        b.push_back(d);
        if (src) { hipError_t e = hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice); if (e != hipSuccess) return fail(c, PT_ERR_HIP, hipGetErrorString(e)); }
        *out = d.p;
        return PT_OK;
    }
};
} // namespace

int pt_post_velocity(pt_ctx* c, uint32_t w, uint32_t h, const float* position, const float* last_inv_projection, float* velocity)
{
    if (!c || !position || !last_inv_projection || !velocity || !w || !h) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    TmpBufs t;
    void *dp, *dv;
    const size_t px = (size_t)w * h;
    if ((r = t.up(c, position, px * 16, &dp)) || (r = t.up(c, nullptr, px * 8, &dv))) return r;
    launch_post_velocity(c->stream, (int)w, (int)h, (const f4*)dp, last_inv_projection, (float*)dv);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(velocity, dv, px * 8, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_post_reproject(pt_ctx* c, uint32_t w, uint32_t h, const float* input, const float* accum, const float* velocity, const uint32_t* id, float* output)
{
    if (!c || !input || !accum || !velocity || !id || !output || !w || !h) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    TmpBufs t;
    void *di, *da, *dv, *did, *dout;
    const size_t px = (size_t)w * h;
    if ((r = t.up(c, input, px * 16, &di)) || (r = t.up(c, accum, px * 16, &da)) || (r = t.up(c, velocity, px * 8, &dv)) ||
        (r = t.up(c, id, px * 4, &did)) || (r = t.up(c, nullptr, px * 16, &dout)))
        return r;
    launch_post_reproject(c->stream, (int)w, (int)h, (const f4*)di, (const f4*)da, (const float*)dv, (const uint32_t*)did, (f4*)dout);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(output, dout, px * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_post_tonemap(pt_ctx* c, uint32_t w, uint32_t h, const float* accum, float* out)
{
    if (!c || !accum || !out || !w || !h) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    TmpBufs t;
    void *da, *dout;
    const size_t px = (size_t)w * h;
    if ((r = t.up(c, accum, px * 16, &da)) || (r = t.up(c, nullptr, px * 16, &dout))) return r;
    launch_post_tonemap(c->stream, (uint32_t)px, (const f4*)da, (f4*)dout);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout, px * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_post_rgb8(pt_ctx* c, uint32_t w, uint32_t h, const float* accum, uint8_t* rgb)
{
    if (!c || !accum || !rgb || !w || !h) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    TmpBufs t;
    void *da, *dout;
    const size_t px = (size_t)w * h;
    if ((r = t.up(c, accum, px * 16, &da)) || (r = t.up(c, nullptr, px * 3, &dout))) return r;
    launch_post_rgb8(c->stream, (uint32_t)px, (const f4*)da, (uint8_t*)dout);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(rgb, dout, px * 3, hipMemcpyDeviceToHost));
    return PT_OK;
}

// ---- unit hooks
static int upload_rays(pt_ctx* c, uint32_t n, const float* o, const float* d, const float* tmax, DevBuf& da, DevBuf& db, DevBuf& dhead)
{
    std::vector<f4> a(n), b(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        a[i] = f4{o[3 * i], o[3 * i + 1], o[3 * i + 2], tmax ? tmax[i] : __builtin_inff()};
        b[i] = f4{d[3 * i], d[3 * i + 1], d[3 * i + 2], from_bits(i)};
    }
    int r;
    const size_t head_bytes = (size_t)(32 + kHeadWordsPerQueue) * 4; // word 0 = n, then the queue's zeroed claim cursors
    if ((r = dev_alloc(c, da, (size_t)n * 16)) || (r = dev_alloc(c, db, (size_t)n * 16)) || (r = dev_alloc(c, dhead, head_bytes))) return r;
    HIPCHK(c, hipMemcpyAsync(da.p, a.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db.p, b.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(dhead.p, 0, head_bytes, c->stream));
    HIPCHK(c, hipMemcpyAsync(dhead.p, &n, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_trace_closest(pt_ctx* c, int which, uint32_t n, const float* o, const float* d, const float* tmax, float* t, float* u, float* v,
                     uint32_t* inst, uint32_t* prim)
{
    if (!c || !o || !d || !t || !u || !v || !inst || !prim) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = upload_scene(c))) return r;
    if (n == 0) return PT_OK;
    const uint32_t root = which ? c->sv.lights_root : c->sv.world_root;
    if (root == MISS_ID)
    {
        for (uint32_t i = 0; i < n; ++i) { t[i] = __builtin_inff(); u[i] = v[i] = 0; inst[i] = prim[i] = MISS_ID; }
        return PT_OK;
    }
    DevBuf da, db, dh, dhit;
    r = upload_rays(c, n, o, d, tmax, da, db, dh);
    if (!r) r = dev_alloc(c, dhit, (size_t)n * 16);
    if (!r)
    {
        launch_trace_rays_closest(c->stream, trace_launch(c), root, RayQueue{(f4*)da.p, (f4*)db.p}, n, (uint32_t*)dh.p, (f4*)dhit.p);
        std::vector<f4> h(n);
        hipError_t e = hipMemcpyAsync(h.data(), dhit.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) r = fail(c, PT_ERR_HIP, hipGetErrorString(e));
        else
        {
            const FlatScene& f = c->scene.flat;
            const uint32_t mask = (1u << f.prim_bits) - 1u;
            for (uint32_t i = 0; i < n; ++i)
            {
                const uint32_t id = bits(h[i].w);
                t[i] = h[i].x; u[i] = h[i].y; v[i] = h[i].z;
                if (id == MISS_ID) { inst[i] = prim[i] = MISS_ID; t[i] = __builtin_inff(); u[i] = v[i] = 0; }
                else
                {
                    inst[i] = (id >> f.prim_bits) - f.inst_base[which ? 1 : 0];
                    prim[i] = f.tri_orig[id & mask];
                }
            }
        }
    }
    dev_free(da); dev_free(db); dev_free(dh); dev_free(dhit);
    return r;
}

int pt_trace_any(pt_ctx* c, int which, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* hit)
{
    if (!c || !o || !d || !tmax || !hit) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = upload_scene(c))) return r;
    if (n == 0) return PT_OK;
    const uint32_t root = which ? c->sv.lights_root : c->sv.world_root;
    if (root == MISS_ID) { std::memset(hit, 0, n); return PT_OK; }
    DevBuf da, db, dh, docc;
    r = upload_rays(c, n, o, d, tmax, da, db, dh);
    if (!r) r = dev_alloc(c, docc, (size_t)n * 4);
    if (!r)
    {
        launch_trace_rays_any(c->stream, trace_launch(c), root, RayQueue{(f4*)da.p, (f4*)db.p}, n, (uint32_t*)dh.p, (uint32_t*)docc.p);
        std::vector<uint32_t> h(n);
        hipError_t e = hipMemcpyAsync(h.data(), docc.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) r = fail(c, PT_ERR_HIP, hipGetErrorString(e));
        else
            for (uint32_t i = 0; i < n; ++i) hit[i] = h[i] ? 1 : 0;
    }
    dev_free(da); dev_free(db); dev_free(dh); dev_free(docc);
    return r;
}

int pt_ss_sobol(pt_ctx* c, uint32_t n_points, uint32_t n, const uint32_t* index, const uint32_t* seed, float* out)
{
    if (!c || !index || !seed || !out || n_points == 0) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    DevBuf di, ds, dout;
    if ((r = dev_alloc(c, di, (size_t)n * 4)) || (r = dev_alloc(c, ds, (size_t)n * 4)) || (r = dev_alloc(c, dout, (size_t)n * 8))) return r;
    HIPCHK(c, hipMemcpy(di.p, index, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(ds.p, seed, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_sobol_probe(c->stream, n_points, n, (const uint32_t*)di.p, (const uint32_t*)ds.p, (float*)dout.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    dev_free(di); dev_free(ds); dev_free(dout);
    return PT_OK;
}

int pt_math_batch(pt_ctx* c, int fn, uint32_t n, const float* a, const float* b, float* o0, float* o1)
{
    if (!c || !a || !o0) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int r;
    if ((r = ensure_device(c))) return r;
    DevBuf da, db, d0, d1;
    if ((r = dev_alloc(c, da, (size_t)n * 4)) || (r = dev_alloc(c, db, (size_t)n * 4)) || (r = dev_alloc(c, d0, (size_t)n * 4)) ||
        (r = dev_alloc(c, d1, (size_t)n * 4)))
        return r;
    HIPCHK(c, hipMemcpy(da.p, a, (size_t)n * 4, hipMemcpyHostToDevice));
    if (b) HIPCHK(c, hipMemcpy(db.p, b, (size_t)n * 4, hipMemcpyHostToDevice));
    else HIPCHK(c, hipMemset(db.p, 0, (size_t)n * 4));
    HIPCHK(c, hipMemset(d1.p, 0, (size_t)n * 4));
    launch_math_probe(c->stream, fn, n, (const float*)da.p, (const float*)db.p, (float*)d0.p, (float*)d1.p, c->cfg.seed);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(o0, d0.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (o1) HIPCHK(c, hipMemcpy(o1, d1.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    dev_free(da); dev_free(db); dev_free(d0); dev_free(d1);
    return PT_OK;
}

int pt_material_eval(pt_ctx* c, int material, uint32_t n, const float* incoming, const float* normal, const uint8_t* front, const uint32_t* pixel,
                     const uint32_t* sample, uint32_t draws, float* out9)
{
    if (!c || !incoming || !normal || !front || !pixel || !sample || !out9) return PT_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (material < 0 || material >= (int)c->scene.materials.size()) return fail(c, PT_ERR_ARG, "material index");
    int r;
    if ((r = upload_scene(c))) return r;
    DevBuf di, dn, df, dp, ds, dout;
    if ((r = dev_alloc(c, di, (size_t)n * 12)) || (r = dev_alloc(c, dn, (size_t)n * 12)) || (r = dev_alloc(c, df, n)) ||
        (r = dev_alloc(c, dp, (size_t)n * 4)) || (r = dev_alloc(c, ds, (size_t)n * 4)) || (r = dev_alloc(c, dout, (size_t)n * 36)))
        return r;
    HIPCHK(c, hipMemcpy(di.p, incoming, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dn.p, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(df.p, front, n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dp.p, pixel, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(ds.p, sample, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_material_probe(c->stream, c->sv, material, n, (const float*)di.p, (const float*)dn.p, (const uint8_t*)df.p, (const uint32_t*)dp.p,
                          (const uint32_t*)ds.p, draws, c->cfg.seed, (float*)dout.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out9, dout.p, (size_t)n * 36, hipMemcpyDeviceToHost));
    dev_free(di); dev_free(dn); dev_free(df); dev_free(dp); dev_free(ds); dev_free(dout);
    return PT_OK;
}

// ---- host-builder introspection
int pt_blas_count(pt_ctx* c) { return (c && c->scene.built) ? (int)c->scene.blas.size() : PT_ERR_STATE; }

int pt_blas_dump(pt_ctx* c, int blas, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b,
                 uint32_t* n_prim_ids, uint32_t* prim_ids, uint32_t cap_nodes, uint32_t cap_ids)
{
    if (!c || !c->scene.built) return PT_ERR_STATE;
    if (blas < 0 || blas >= (int)c->scene.blas.size()) return PT_ERR_ARG;
    const HostBlas& bl = c->scene.blas[blas];
    *n_nodes = (uint32_t)bl.nodes.size();
    *root = bl.root;
    *n_prim_ids = (uint32_t)bl.prim_ids.size();
    if (bl.nodes.size() > cap_nodes || bl.prim_ids.size() > cap_ids) return PT_ERR_ARG;
    for (size_t i = 0; i < bl.nodes.size(); ++i)
    {
        std::memcpy(boxes6 + i * 6, &bl.nodes[i].box, 24);
        kind[i] = bl.nodes[i].kind == NODE_BRANCH ? 0 : 1;
        a[i] = bl.nodes[i].a;
        b[i] = bl.nodes[i].b;
    }
    std::memcpy(prim_ids, bl.prim_ids.data(), bl.prim_ids.size() * 4);
    return PT_OK;
}

int pt_tlas_dump(pt_ctx* c, int which, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b, uint32_t cap)
{
    if (!c || !c->scene.built) return PT_ERR_STATE;
    const HostTlas& t = which ? c->scene.lights : c->scene.world;
    *n_nodes = (uint32_t)t.nodes.size();
    *root = t.root;
    if (t.nodes.size() > cap) return PT_ERR_ARG;
    for (size_t i = 0; i < t.nodes.size(); ++i)
    {
        std::memcpy(boxes6 + i * 6, &t.nodes[i].box, 24);
        kind[i] = t.nodes[i].kind == NODE_BRANCH ? 0 : 1;
        a[i] = t.nodes[i].a;
        b[i] = t.nodes[i].b;
    }
    return PT_OK;
}

// TLASNodeType::Leaf { matrix, inv_matrix } (tlas_bvh.rs:36-41, 92-101) of every leaf, in arena order: rows of the two 3x4 matrices
int pt_tlas_instances(pt_ctx* c, int which, uint32_t* n, float* matrix12, float* inv_matrix12, uint32_t cap)
{
    if (!c || !c->scene.built || !n) return PT_ERR_STATE;
    const HostTlas& t = which ? c->scene.lights : c->scene.world;
    *n = (uint32_t)t.instances.size();
    if (t.instances.size() > cap || !matrix12 || !inv_matrix12) return PT_ERR_ARG;
    auto rows = [](const xf34& x, float* o) {
        const float r[12] = {x.m.c0.x, x.m.c1.x, x.m.c2.x, x.t.x, x.m.c0.y, x.m.c1.y, x.m.c2.y, x.t.y, x.m.c0.z, x.m.c1.z, x.m.c2.z, x.t.z};
        std::memcpy(o, r, sizeof(r));
    };
    for (size_t i = 0; i < t.instances.size(); ++i) { rows(t.instances[i].fwd, matrix12 + 12 * i); rows(t.instances[i].inv, inv_matrix12 + 12 * i); }
    return PT_OK;
}

int pt_light_cdf(pt_ctx* c, uint32_t* n, float* pdf, float* cdf, uint32_t* blas, uint32_t* prim, float* max_weight, uint32_t cap)
{
    if (!c || !c->scene.built) return PT_ERR_STATE;
    const auto& L = c->scene.light_items;
    *n = (uint32_t)L.size();
    *max_weight = c->scene.light_weight_sum;
    if (L.size() > cap) return PT_ERR_ARG;
    for (size_t i = 0; i < L.size(); ++i) { pdf[i] = L[i].pdf; cdf[i] = L[i].cdf; blas[i] = L[i].blas; prim[i] = L[i].prim; }
    return PT_OK;
}

int pt_triangle_dump(pt_ctx* c, int blas, uint32_t prim, float out36[36])
{
    if (!c || !c->scene.built) return PT_ERR_STATE;
    if (blas < 0 || blas >= (int)c->scene.blas.size() || prim >= c->scene.blas[blas].tris.size()) return PT_ERR_ARG;
    const HostTriangle& t = c->scene.blas[blas].tris[prim];
    std::memcpy(out36, &t.n0, 16);
    std::memcpy(out36 + 4, &t.n1, 16);
    std::memcpy(out36 + 8, &t.n2, 16);
    std::memcpy(out36 + 12, t.p, 36);
    std::memcpy(out36 + 21, t.n, 36);
    for (int i = 30; i < 36; ++i) out36[i] = 0;
    return PT_OK;
}

// ====================================================================================================== several devices, one process
} // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <thread>

namespace {
// RCCL is loaded on first use: a single-device user of libptmi never maps it
struct Rccl
{
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string* err)
    {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) { *err = std::string("librccl.so cannot be loaded: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Gather = (decltype(Gather))dlsym(lib, "ncclGather");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather || !GetErrorString) { *err = "librccl.so lacks an expected symbol"; return false; }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
} // namespace

struct pt_multi
{
    std::vector<pt_ctx*> ctx;
    std::vector<int> devices;
    std::string err;
    bool distinct = true;                 // no device listed twice: the gather can go through RCCL
    std::vector<ncclComm_t> comms;
    uint64_t replicated_version = ~0ull;  // scene_version of ctx[0] the other contexts hold
    DevBuf d_parts, d_full;               // on devices[0]: gathered padded strips, assembled frame
    bool used_rccl = false;
};

namespace {
int mfail(pt_multi* m, int code, const std::string& msg)
{
    m->err = msg;
    return code;
}
#define MHIPCHK(m, call)                                                                                      \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) return mfail((m), PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Scene::new happened on rank 0's context: every other context gets a copy of the built host scene, camera and environment
int multi_replicate(pt_multi* m)
{
    pt_ctx* c0 = m->ctx[0];
    std::lock_guard<std::mutex> lk0(c0->mu); // rank 0's scene must not change under the copies (pt_add_model / pt_build from another thread)
    if (m->replicated_version == c0->scene_version) return PT_OK;
    for (size_t i = 1; i < m->ctx.size(); ++i)
    {
        pt_ctx* c = m->ctx[i];
        if (c == c0) continue;
        std::lock_guard<std::mutex> lk(c->mu);
        c->scene = c0->scene;
        c->scene_uploaded = false;
        c->h_env = c0->h_env;
        c->env_w = c0->env_w;
        c->env_h = c0->env_h;
        c->env_uploaded = false;
        c->scene_version = c0->scene_version;
    }
    m->replicated_version = c0->scene_version;
    return PT_OK;
}
} // namespace

extern "C" {

pt_multi* pt_multi_create(const pt_config* cfg, const int32_t* devices, uint32_t n_devices)
{
    if (!cfg || n_devices == 0 || n_devices > 64) return nullptr;
    pt_multi* m = new (std::nothrow) pt_multi();
    if (!m) return nullptr;
    for (uint32_t i = 0; i < n_devices; ++i)
    {
        pt_config g = *cfg;
        g.rank = i;
        g.world_size = n_devices;
        g.device = devices ? devices[i] : (int32_t)i;
        pt_ctx* c = pt_create(&g);
        if (!c) { pt_multi_destroy(m); return nullptr; }
        m->ctx.push_back(c);
        m->devices.push_back(g.device);
        for (uint32_t j = 0; j < i; ++j) m->distinct = m->distinct && m->devices[j] != g.device;
    }
    return m;
}

void pt_multi_destroy(pt_multi* m)
{
    if (!m) return;
    if (!m->comms.empty())
    {
        for (ncclComm_t k : m->comms) (void)g_rccl.CommDestroy(k);
    }
    if (!m->ctx.empty() && m->ctx[0]->dev_ready)
    {
        (void)hipSetDevice(m->ctx[0]->device);
        dev_free(m->d_parts);
        dev_free(m->d_full);
    }
    for (pt_ctx* c : m->ctx) pt_destroy(c);
    delete m;
}

const char* pt_multi_last_error(pt_multi* m) { return m ? m->err.c_str() : "null pt_multi"; }
pt_ctx* pt_multi_ctx(pt_multi* m, uint32_t rank) { return (m && rank < m->ctx.size()) ? m->ctx[rank] : nullptr; }
int pt_multi_used_rccl(pt_multi* m) { return m ? (m->used_rccl ? 1 : 0) : PT_ERR_ARG; }

int pt_multi_reset_accumulation(pt_multi* m)
{
    if (!m) return PT_ERR_ARG;
    for (pt_ctx* c : m->ctx)
    {
        int r = pt_reset_accumulation(c);
        if (r) return mfail(m, r, pt_last_error(c));
    }
    return PT_OK;
}

int pt_multi_render(pt_multi* m, uint32_t first_sample, uint32_t n_samples, float* data_rgba)
{
    if (!m) return PT_ERR_ARG;
    int r;
    if ((r = multi_replicate(m))) return r;
    const size_t n = m->ctx.size();
    // one host thread per device (the reference: one rayon pool, src/main.rs:72); no collective while rendering
    std::vector<int> rc(n, PT_OK);
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < n; ++i) th.emplace_back([&, i] { rc[i] = pt_render_device(m->ctx[i], first_sample, n_samples); });
        rc[0] = pt_render_device(m->ctx[0], first_sample, n_samples);
        for (std::thread& t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i)
        if (rc[i]) return mfail(m, rc[i], std::string("device ") + std::to_string(m->devices[i]) + ": " + pt_last_error(m->ctx[i]));

    // ---- gather: every rank's padded strip framebuffer to devices[0], then de-interleave there
    pt_ctx* c0 = m->ctx[0];
    const pt_config& g = c0->cfg;
    const size_t part_px = (size_t)c0->pad_rows * g.width, full_px = (size_t)g.width * g.height;
    for (pt_ctx* c : m->ctx)
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if ((r = ensure_device(c)) || (r = ensure_frame(c))) return mfail(m, r, pt_last_error(c));
    }
    MHIPCHK(m, hipSetDevice(c0->device));
    if ((r = dev_alloc(c0, m->d_parts, std::max<size_t>(part_px * n, 1) * 16)) || (r = dev_alloc(c0, m->d_full, std::max<size_t>(full_px, 1) * 16))) return mfail(m, r, pt_last_error(c0));
    m->used_rccl = false;
    if (m->distinct)
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        std::string e;
        if (!g_rccl.load(&e)) return mfail(m, PT_ERR_NCCL, e);
        if (m->comms.empty())
        {
            m->comms.resize(n);
            const ncclResult_t nr = g_rccl.CommInitAll(m->comms.data(), (int)n, m->devices.data());
            if (nr != ncclSuccess) { m->comms.clear(); return mfail(m, PT_ERR_NCCL, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr)); }
        }
        ncclResult_t nr = g_rccl.GroupStart();
        for (size_t i = 0; i < n && nr == ncclSuccess; ++i)
        {
            pt_ctx* c = m->ctx[i];
            if (hipSetDevice(c->device) != hipSuccess) { nr = ncclUnhandledCudaError; break; }
            nr = g_rccl.Gather(c->d_accum.p, i == 0 ? m->d_parts.p : nullptr, part_px * 4, ncclFloat, 0, m->comms[i], c->stream);
        }
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return mfail(m, PT_ERR_NCCL, std::string("ncclGather: ") + g_rccl.GetErrorString(nr));
        m->used_rccl = true;
        for (size_t i = 1; i < n; ++i)
        {
            MHIPCHK(m, hipSetDevice(m->ctx[i]->device));
            MHIPCHK(m, hipStreamSynchronize(m->ctx[i]->stream));
        }
        MHIPCHK(m, hipSetDevice(c0->device));
    }
    else
    {
        // contexts sharing a device (or RCCL ruled out): plain device-to-device copies on devices[0]'s stream
        for (size_t i = 0; i < n; ++i)
            MHIPCHK(m, hipMemcpyAsync((uint8_t*)m->d_parts.p + i * part_px * 16, m->ctx[i]->d_accum.p, part_px * 16, hipMemcpyDeviceToDevice, c0->stream));
    }
    launch_post_deinterleave(c0->stream, g.width, g.height, (uint32_t)n, g.strip_rows, c0->pad_rows, (const f4*)m->d_parts.p, (f4*)m->d_full.p);
    if (data_rgba) MHIPCHK(m, hipMemcpyAsync(data_rgba, m->d_full.p, full_px * 16, hipMemcpyDeviceToHost, c0->stream));
    MHIPCHK(m, hipStreamSynchronize(c0->stream));
    MHIPCHK(m, hipGetLastError());
    return PT_OK;
}

int pt_multi_write_image(pt_multi* m, const char* path)
{
    if (!m || !path) return PT_ERR_ARG;
    if (!m->d_full.p) return mfail(m, PT_ERR_STATE, "pt_multi_render has not been called");
    pt_ctx* c0 = m->ctx[0];
    const size_t px = (size_t)c0->cfg.width * c0->cfg.height;
    MHIPCHK(m, hipSetDevice(c0->device));
    DevBuf d_rgb;
    int r = dev_alloc(c0, d_rgb, px * 3);
    if (r) return mfail(m, r, pt_last_error(c0));
    launch_post_rgb8(c0->stream, (uint32_t)px, (const f4*)m->d_full.p, (uint8_t*)d_rgb.p);   // ImageHelper::write_image  image_helper.rs:37-58
    std::vector<uint8_t> host(px * 3);
    hipError_t e = hipMemcpyAsync(host.data(), d_rgb.p, px * 3, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
    dev_free(d_rgb);
    if (e != hipSuccess) return mfail(m, PT_ERR_HIP, hipGetErrorString(e));
    std::string err;
    if (!write_png_rgb8(path, host.data(), c0->cfg.width, c0->cfg.height, &err)) return mfail(m, PT_ERR_IO, err);
    return PT_OK;
}

int pt_multi_framebuffer_device_ptr(pt_multi* m, void** dev_ptr)
{
    if (!m || !dev_ptr) return PT_ERR_ARG;
    if (!m->d_full.p) return mfail(m, PT_ERR_STATE, "pt_multi_render has not been called");
    *dev_ptr = m->d_full.p;
    return PT_OK;
}

int pt_multi_get_stats(pt_multi* m, pt_stats* sum)
{
    if (!m || !sum) return PT_ERR_ARG;
    *sum = pt_stats();
    for (pt_ctx* c : m->ctx)
    {
        const pt_stats& s = c->stats;
        sum->rays_closest += s.rays_closest; sum->rays_any += s.rays_any; sum->rays_light_closest += s.rays_light_closest;
        sum->rays_light_closest_traced += s.rays_light_closest_traced; sum->rays_primary_culled += s.rays_primary_culled; sum->paths += s.paths;
        sum->launches_trace_closest += s.launches_trace_closest; sum->ms_trace_closest += s.ms_trace_closest;
        sum->ms_trace_any += s.ms_trace_any; sum->ms_trace_light += s.ms_trace_light; sum->ms_shade += s.ms_shade;
        sum->ms_generate += s.ms_generate; sum->ms_accumulate += s.ms_accumulate;
        sum->ms_total = std::max(sum->ms_total, s.ms_total);
        sum->state_bytes += s.state_bytes;
        sum->scene_bytes = s.scene_bytes; sum->lds_scene = s.lds_scene; sum->stack_entries = s.stack_entries;
    }
    return PT_OK;
}

int pt_get_stats(pt_ctx* c, pt_stats* out)
{
    if (!c || !out) return PT_ERR_ARG;
    *out = c->stats;
    return PT_OK;
}

int pt_last_batch_counters(pt_ctx* c, uint32_t* rows16, uint32_t cap_rows, uint32_t* n_rows)
{
    if (!c || !rows16 || !n_rows) return PT_ERR_ARG;
    const pt_ctx::Pipe& pp = c->pipe[c->last_pipe];
    if (!pp.h_counters || !pp.h_heads) return PT_ERR_STATE;
    const uint32_t rows = std::min(cap_rows, c->cfg.max_bounces + 2);
    const uint32_t have = std::min(rows, pp.busy_rows); // the batch read back only the rows it used; the others are zero
    std::memset(rows16, 0, (size_t)rows * sizeof(Counters));
    std::memcpy(rows16, pp.h_counters, (size_t)have * sizeof(Counters));
    for (uint32_t r = 0; r < have; ++r)
    {
        uint32_t* o = rows16 + 16 * r;
        o[1] = o[3] = o[5] = 0;
        o[6] = o[7] = o[13] = o[14] = o[15] = 0;
        const uint32_t* hrow = pp.h_heads + (size_t)r * HEADS_PER_ROW * kHeadWordsPerQueue;
        for (uint32_t g = 0; g < kQueueHeads; ++g)
        {
            const uint32_t* cl = hrow + HEADS_CLOSEST * kHeadWordsPerQueue + g * kHeadStrideWords;
            const uint32_t* sh = hrow + HEADS_SHADOW * kHeadWordsPerQueue + g * kHeadStrideWords;
            const uint32_t* lc = hrow + HEADS_LCHAIN * kHeadWordsPerQueue + g * kHeadStrideWords;
            o[6] += lc[HEAD_TALLY0]; o[7] += lc[HEAD_TALLY1]; o[13] += cl[HEAD_TALLY0]; o[14] += sh[HEAD_TALLY0]; o[15] += lc[HEAD_TALLY2];
        }
    }
    *n_rows = rows;
    return PT_OK;
}

// diagnostic (tools/shade_access_bench.py): the path ids of the LAST batch's last bounce in the order its shading pass read them — word 3 of
// array `a` of the class's shade queue, which nothing overwrites after the batch's last traversal launch.  HOLE (0xffffffff) = a slot no ray took.
int pt_last_batch_shade_pids(pt_ctx* c, uint32_t qclass, uint32_t first, uint32_t count, uint32_t* pids)
{
    if (!c || !pids || qclass == Q_TERMINAL || qclass >= Q_COUNT) return PT_ERR_ARG;
    const pt_ctx::Pipe& pp = c->pipe[c->last_pipe];
    if (!pp.wb.q_shade_base || !((pp.wb.class_mask >> qclass) & 1u)) return PT_ERR_STATE;
    if ((uint64_t)first + count > pp.wb.q_stride) return PT_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const ShadeQueue q = shade_queue(pp.wb, qclass);
    HIPCHK(c, hipMemcpy2D(pids, 4, reinterpret_cast<const uint8_t*>(q.a + first) + 12, 16, 4, count, hipMemcpyDeviceToHost));
    return PT_OK;
}

// diagnostic (PT_STEP_STATS variant builds, tools/step_stats.py): words 8..15 of the closest-hit cursor lines of the last batch, summed
// over the 64 lines, one row of 8 words per bounce
int pt_last_batch_step_stats(pt_ctx* c, uint32_t* rows8, uint32_t cap_rows, uint32_t* n_rows)
{
    if (!c || !rows8 || !n_rows) return PT_ERR_ARG;
    const pt_ctx::Pipe& pp = c->pipe[c->last_pipe];
    if (!pp.h_heads) return PT_ERR_STATE;
    const uint32_t rows = std::min(cap_rows, c->cfg.max_bounces + 2);
#if PT_WAVE_TIMES
    // diagnostic build: the eight words are a reduction of the per-wave time records of the row's k_closest launch instead:
    // ~min start, max end, sum of lifetimes, sum of (queue empty - start), waves, ~min and max of "queue empty", sum of (first rays - start)
    {
        std::vector<uint32_t> rec((size_t)rows * kWaveTimeSlots * 4);
        const char* which = std::getenv("PTMI_WAVE_TIMES_KERNEL"); // "any": the shadow-ray launches instead of the closest-hit ones
        const uint4* src = (which && which[0] == 'a') ? pp.wb.wave_times_any : pp.wb.wave_times;
        if (hipMemcpy(rec.data(), src, rec.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return PT_ERR_HIP;
        for (uint32_t r = 0; r < rows; ++r)
        {
            uint32_t* o = rows8 + 8 * r;
            std::memset(o, 0, 32);
            uint32_t t0 = 0; bool any = false;
            for (uint32_t w = 0; w < kWaveTimeSlots; ++w)
            {
                const uint32_t* q = &rec[((size_t)r * kWaveTimeSlots + w) * 4];
                if (q[3] == 0u) continue;
                if (!any || (int32_t)(q[0] - t0) < 0) t0 = q[0];
                any = true;
            }
            if (!any) continue;
            uint32_t max_end = 0, min_dr = 0xffffffffu, max_dr = 0;
            for (uint32_t w = 0; w < kWaveTimeSlots; ++w)
            {
                const uint32_t* q = &rec[((size_t)r * kWaveTimeSlots + w) * 4];
                if (q[3] == 0u) continue;
                max_end = std::max(max_end, q[3] - t0);
                min_dr = std::min(min_dr, q[2] - t0);
                max_dr = std::max(max_dr, q[2] - t0);
                o[2] += q[3] - q[0];
                o[3] += q[2] - q[0];
                o[4] += 1u;
                o[7] += q[1] - q[0];
            }
            o[0] = ~t0; o[1] = t0 + max_end; o[5] = ~(t0 + min_dr); o[6] = t0 + max_dr;
            if (std::getenv("PTMI_WAVE_TIMES_DUMP"))
            {
                std::vector<uint32_t> ends;
                for (uint32_t w = 0; w < kWaveTimeSlots; ++w)
                {
                    const uint32_t* q = &rec[((size_t)r * kWaveTimeSlots + w) * 4];
                    if (q[3] != 0u) ends.push_back(q[3] - t0);
                }
                std::sort(ends.begin(), ends.end());
                std::fprintf(stderr, "row %u ends (us) p1 %.0f p10 %.0f p25 %.0f p50 %.0f p75 %.0f p90 %.0f p99 %.0f max %.0f\n", r, ends[ends.size() / 100] * 0.01, ends[ends.size() / 10] * 0.01,
                             ends[ends.size() / 4] * 0.01, ends[ends.size() / 2] * 0.01, ends[ends.size() * 3 / 4] * 0.01, ends[ends.size() * 9 / 10] * 0.01, ends[ends.size() * 99 / 100] * 0.01, ends.back() * 0.01);
            }
        }
        *n_rows = rows;
        return PT_OK;
    }
#endif
    // PTMI_STEP_STATS_BASE=16: a second group of eight words, for instrumented builds that fill it
    const char* base_env = std::getenv("PTMI_STEP_STATS_BASE");
    const uint32_t base = base_env && std::atoi(base_env) == 16 ? 16u : 8u;
    // PTMI_STEP_STATS_QUEUE=shadow: the shadow-ray launches' statistics (k_any) instead of the closest-hit launches'
    const char* q_env = std::getenv("PTMI_STEP_STATS_QUEUE");
    const uint32_t which_queue = q_env && q_env[0] == 's' ? (uint32_t)HEADS_SHADOW : (uint32_t)HEADS_CLOSEST;
    for (uint32_t r = 0; r < rows; ++r)
    {
        uint32_t* o = rows8 + 8 * r;
        std::memset(o, 0, 32);
        if (r >= pp.busy_rows) continue; // not read back: the batch ended before this bounce
        const uint32_t* hrow = pp.h_heads + (size_t)r * HEADS_PER_ROW * kHeadWordsPerQueue + which_queue * kHeadWordsPerQueue;
        for (uint32_t g = 0; g < kQueueHeads; ++g)
            for (uint32_t k = 0; k < 8; ++k) o[k] += hrow[g * kHeadStrideWords + base + k];
    }
    *n_rows = rows;
    return PT_OK;
}

int pt_reset_stats(pt_ctx* c)
{
    if (!c) return PT_ERR_ARG;
    pt_stats keep = c->stats;
    c->stats = pt_stats();
    c->stats.scene_bytes = keep.scene_bytes;
    c->stats.lds_scene = keep.lds_scene;
    c->stats.stack_entries = keep.stack_entries;
    c->stats.state_bytes = keep.state_bytes;
    return PT_OK;
}

} // extern "C"
