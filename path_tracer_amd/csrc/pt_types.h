// Flattened scene + wavefront state layouts shared by the host builder and the gfx950 kernels.
//
// HBM layout (all arrays 16-byte aligned, read through 16-byte vector loads):
//   nodes      : 32 B  { min.xyz, a | max.xyz, b }   TLAS(world) | TLAS(lights) | BLAS 0 | BLAS 1 ...   absolute indices
//                b>>30 = kind: 0 branch (a = left, b&mask = right), 1 triangle leaf (a = first triangle, b&mask = count),
//                             2 instance leaf (a = instance record)
//   tri_isect  : 48 B  Havel-Herout planes n0|d0, n1|d1, n2|d2 (primitive.rs:20-26), BLAS-leaf order
//   tri_shade  : 48 B  vertex normals  (3 x float4, w unused)
//   tri_pos    : 48 B  vertex positions (3 x float4, w unused)  - only light sampling reads it
//   instances  : 144 B inverse 3x4 | blas root, blas id, material, class | copy of the BLAS root node | forward 3x4
//   materials  : 48 B
//   lights     : 16 B  { triangle, material, pdf, cdf }
// Traversal touches nodes + tri_isect + instances only ("scene blob"); when that fits it is staged in LDS.
#pragma once
#include <stdint.h>
#include "pt_math.h"

namespace pt {

enum : uint32_t
{
    NODE_BRANCH = 0u,   // link payload: index of the left child; the right child is the next node
    NODE_TRIS = 1u,     // link payload: (count - 1) << 24 | first triangle (leaf order)
    NODE_INSTANCE = 2u, // link payload: instance index
    NODE_TRIS_BIG = 3u, // link payload: index into the big-leaf table {first, count} (leaves over 64 triangles / scenes over 2^24)
    NODE_KIND_SHIFT = 30u,
    NODE_PAYLOAD_MASK = 0x3fffffffu,
    LEAF_FIRST_BITS = 24u,
    LEAF_FIRST_MASK = 0x00ffffffu,
    LEAF_MAX_COUNT = 64u,
    MISS_ID = 0xffffffffu
};

// One BVH node.  `link` (kind << 30 | payload) is everything a traversal needs once the node's box has been tested: it is what
// gets pushed on the stack, so a popped entry never has to go back to the node it came from.  Within a tree the nodes are laid
// out breadth-first with the two children of a branch adjacent: a branch is expanded with one contiguous 64-byte read.
struct alignas(16) DNode
{
    float mn[3];
    uint32_t link;
    float mx[3];
    uint32_t aux;   // branch: right child (= left + 1); leaf: triangle count; instance: unused
};
static_assert(sizeof(DNode) == 32, "node is two 16-byte words");

struct alignas(16) DTriIsect { f4 n0, n1, n2; };
static_assert(sizeof(DTriIsect) == 48, "");
struct alignas(16) DTriVerts { f4 a, b, c; };

// One TLAS leaf.  What a traversal needs when it enters the instance is the FIRST 96 bytes, one contiguous run of six 16-byte words that
// are loaded together: the inverse matrix, the meta word and a COPY of the BLAS root node (its link is what BLAS::intersect starts from,
// blas.rs:216-217; its box is what BLAS::any_intersect tests first, blas.rs:262-264).  Without the copy an instance visit is three
// dependent memory round trips (meta -> matrix rows, meta.root -> root node); on BVHs in global memory 88 % of the atrium's wave-steps
// execute the instance section for ~10 of 64 lanes (profiles/r04_step_stats_atrium.md) and every lane of the wave waits for them.
struct alignas(16) DInstance
{
    float inv[12];  // rows of the inverse instance matrix (ray -> object space)       tlas_bvh.rs:41-43
    uint32_t root;  // absolute node index of the BLAS root
    uint32_t blas;  // BLAS (= model) index in its TLAS arena; world: first_id source   integrator.rs:184
    uint32_t material;
    uint32_t qclass; // bits 7:0 shade-queue class of the material (Q_*), bit 8 INSTANCE_IDENTITY
    DNode root_node; // nodes[root]
    float fwd[12];  // rows of the instance matrix (object normal -> world)            tlas.rs:105
};
static_assert(sizeof(DInstance) == 144, "");
enum : uint32_t { INST_WORDS = 9u, INST_META_WORD = 3u, INST_ROOT_WORD = 4u }; // 16-byte words per record; where meta and the root node copy start

enum : uint32_t { MAT_LAMBERTIAN = 0, MAT_EMISSIVE = 1, MAT_SPECULAR = 2, MAT_GGX_METAL = 3, MAT_GGX_DIELECTRIC = 4, MAT_DIELECTRIC = 5 };

struct alignas(16) DMaterial
{
    float colour[3];
    uint32_t kind;
    float alpha;       // GGX a
    float ior;
    uint32_t has_volume;
    uint32_t vol_flags; // bit0 absorption, bit1 scatter
    float vol_abs[3];   // absorption * k
    float vol_c;
    float vol_g;
    float pad[3];
};
static_assert(sizeof(DMaterial) == 64, "");

struct alignas(16) DLight
{
    uint32_t tri;      // absolute triangle index (leaf order)
    uint32_t material;
    float pdf;         // weight / sum                                               light_sampler.rs:47
    float cdf;         // running sum of pdf                                         light_sampler.rs:50-56
};

// shade-queue classes (one shading kernel each)
enum : uint32_t { Q_TERMINAL = 0, Q_LAMBERT = 1, Q_SPECULAR = 2, Q_DIELECTRIC = 3, Q_GGX = 4, Q_COUNT = 5 };
enum : uint32_t { ENTRY_DEAD = 0x80000000u };
enum : uint32_t { INSTANCE_IDENTITY = 0x100u }; // inverse matrix is bit-exactly glam's inverse of the identity (I, translation -0)

// per-bounce counter row (zeroed once per batch).  n_* count queue SLOTS: producers reserve regions and return the unused
// tails as holes, so the exact ray counts are tallied separately by the consumers (valid_*).
struct Counters
{
    uint32_t n_closest;      // rays in the world closest-hit queue of this bounce
    uint32_t overflow;       // non-zero: some producer found a queue full (its entries went to the queue's dump area, nothing was
                             // written out of bounds); the batch is void and the host reports PT_ERR_LIMIT
    uint32_t n_shadow;       // explicit-light shadow rays produced by this bounce's shading
    uint32_t spare0;
    uint32_t n_lchain;       // BSDF-sampled NEE rays (lights TLAS closest hit, then world any hit)
    uint32_t spare1;
    uint32_t spare2, spare3;
    uint32_t n_shade[Q_COUNT];  // [Q_TERMINAL]: the terminal queue's tail (slots).  Surface classes: their queues' tails are striped (kTailStripes
                                // words each, WavefrontBuffers::tails); the shading pass writes the extent it derived from them here for the host's tables
    uint32_t spare4, spare5, spare6; // (exact ray tallies live in the claim-cursor lines: HEAD_TALLY*)
};
static_assert(sizeof(Counters) == 64, "");

struct SceneView
{
    const DNode* nodes;
    const DTriIsect* tri_isect;
    const DTriVerts* tri_shade;
    const DTriVerts* tri_pos;
    const uint32_t* tri_orig;     // leaf-order index -> (blas-local) load-order primitive id
    const DInstance* instances;
    const DMaterial* materials;
    const DLight* lights;
    uint32_t n_nodes, n_tris, n_instances, n_materials, n_lights;
    uint32_t world_root, lights_root; // absolute node indices (MISS_ID: empty TLAS)
    uint32_t prim_bits;               // hit id = (instance << prim_bits) | triangle
    float light_weight_sum;           // LightSampler::max                         light_sampler.rs:43
    uint32_t has_volumes;             // some material carries Some(Volume): per-path volume stacks are live
    uint32_t blob_bytes;              // nodes + tri_isect + instances, multiple of 16
    uint32_t stack_entries;           // per-lane traversal stack capacity (exact bound)
    uint32_t stack_lds;               // closest-hit levels kept in LDS; deeper levels spill to `stack_spill` (deep BLASes only)
    uint64_t* stack_spill;            // 8-byte (node, t_enter) records [level - stack_lds][global lane], null when stack_entries <= stack_lds
};
struct CameraView
{
    float ray_matrix[16]; // (matrix * inv_projection), column-major                camera.rs:98
    float eye[3];         // matrix.translation
    float pad;
};

// Division by a launch-invariant divisor (Granlund-Montgomery / libdivide "branch-free" form): n / d = (((n - q) >> 1) + q) >> shift
// with q = mulhi(magic, n), exact for every 32-bit n and d >= 2; d == 1 is flagged.  Built on the host (fastdiv_make).
struct FastDiv
{
    uint32_t magic, shift, one, d;
};
inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f{0u, 0u, d <= 1u ? 1u : 0u, d};
    if (d <= 1u) return f;
    uint32_t l = 31u;
    while (!((d >> l) & 1u)) --l;                 // floor(log2 d)
    if ((d & (d - 1u)) == 0u) { f.magic = 0u; f.shift = l - 1u; return f; }
    const uint64_t num = (uint64_t)1 << (32u + l);
    uint64_t m = num / d;
    const uint64_t rem = num - m * d;
    m += m;
    const uint64_t twice_rem = rem + rem;
    if (twice_rem >= d) m += 1u;
    f.magic = (uint32_t)(m + 1u);
    f.shift = l;
    return f;
}

struct RenderParams
{
    uint32_t width, height;
    uint32_t local_rows, local_pixels;
    uint32_t rank, world_size, strip_rows;
    uint32_t first_sample;   // global index of batch-local sample 0
    uint32_t batch_samples;
    uint32_t n_paths;        // act_pixels * batch_samples
    uint32_t max_bounces, n_sobol, enable_nee;
    uint32_t keep_s_id;      // batch-local samples >= this are the batch's last two: their first-hit id is kept (id history, main.rs:206)
    uint32_t keep_s_pos;     // the batch's last sample (batch_samples - 1): its first-hit position is kept (main.rs:205)
    // PATH IDS are dealt in BLOCKS of 2^blk_log samples: block b holds samples [b << blk_log, ...) of every active pixel, and inside a block the
    // samples of ONE pixel are consecutive:  pid = b * (act_pixels << blk_log) + k * (samples in block b) + (sample - (b << blk_log)),  k = the pixel's index
    // in the active rectangle (row-major).  64 consecutive path ids — what a traversal wave fetches — are then 2^blk_log samples of 64 >> blk_log
    // neighbouring pixels instead of one sample of 64 pixels in a row: camera rays eight times closer together, and the shadow rays of a bounce-0 hit
    // leave one point for one light.  blk_log = 0 is the sample-major order of rounds 1-3.  The batch's last block may be short (blk_last samples).
    uint32_t blk_log, n_blk, blk_last;
    // Camera rays are generated only for the ACTIVE pixels: a rectangle of columns [act_x0, act_x0 + act_w) and LOCAL rows
    // [act_ly0, act_ly0 + act_rows) outside which every camera ray provably misses the scene's bounds (the host projects the world
    // TLAS's root box onto the image plane, with a margin; pt_api.cpp).  Path id = sample * act_pixels + (row - act_ly0) * act_w +
    // (column - act_x0) in round 1-3's sample-major order (now: pid_join); pixels outside the rectangle get the miss result (integrator.rs:263-266) in k_accumulate.
    uint32_t act_x0, act_w, act_ly0, act_rows, act_pixels;
    uint32_t pad[1];
    uint64_t seed;
    FastDiv div_blk_paths, div_blk_last, div_act_w, div_width, div_strip_rows; // div_blk_paths: by act_pixels << blk_log
};

// wavefront state, one slot per path (pid = s_local * local_pixels + local_pixel).  What a shading pass reads and writes together is
// ONE 64-byte record: in later bounces only a sparse subset of paths is alive, and a record costs one memory sector per live path
// where a structure of arrays costs one sector per field.
struct DPathRec
{
    f4 pw;     // path_weight.xyz | draws consumed (bits)
    f4 acc;    // accumulated.xyz | flags (bits): bit16 last_delta, bit17 nee pending, bit18 bsdf ray cast
    f4 nee_e;  // explicit-light candidate contribution (integrator.rs:69-70) | slot of the BSDF-sampled ray in rq_lchain (bits)
    f4 nee_pw; // path_weight at NEE time | bsdf pdf of the BSDF-sampled direction
};
static_assert(sizeof(DPathRec) == 64, "path state layout");
struct PathState
{
    DPathRec* rec;
    // BSDF-sampled NEE ray of the path's last bounce (the few that pass the lights' root box): 0 reached a light, 1 blocked before
    // it, 2 no light on the ray; written by k_closest<LIGHTS>.  At bounce 0 the byte says whether the camera ray left the scene at
    // once (0xff: the path's radiance is the ambient term and no radiance record is written) — k_closest<PRIMARY> writes it for
    // every path of the batch.  Dense bytes.  (The explicit shadow ray's result needs no word: a
    // blocked ray zeroes DPathRec::nee_e.)
    uint8_t* occl;
    uint32_t* vstack;   // volume stack (integrator.rs:161): four material indices, one per byte, 0xff = empty, insertion order; null without volumes
    f4* radiance;       // finished paths: accumulated.xyz (what integrate() returns before the finite check), dense by path id
    f4* first_pos;      // first-hit xyz | t        (main.rs:205) of the batch's LAST sample (RenderParams::keep_s_pos): index = the pixel's index in the active rectangle
    uint32_t* first_id; // of the batch's last TWO samples (id history, main.rs:206): index = (sample - keep_s_id) * act_pixels + pixel index
};
enum : uint32_t { FLAG_BOUNCE_MASK = 0xffffu, FLAG_LAST_DELTA = 1u << 16, FLAG_NEE_PENDING = 1u << 17, FLAG_BSDF_CAST = 1u << 18 };

// dense ray queue: A = origin.xyz | t_max, B = direction.xyz | path id (bits)
struct RayQueue
{
    f4* a;
    f4* b;
};

// Claim cursors of a ray queue.  A returning atomic on ONE word sustains only ~88 operations per microsecond on MI355X, so the part of
// a queue that is handed out dynamically is cut into kQueueHeads partitions, each with a cursor in a cache line of its own; a wave
// looks at all cursors with one 64-lane load, takes a chunk from the first partition at or after its home that has one left, and
// moves on (steals) when its home runs dry.  Zeroed with the counters once per batch.
enum : uint32_t { kQueueHeads = 64u, kHeadStrideWords = 32u, kHeadWordsPerQueue = kQueueHeads * kHeadStrideWords };
enum : uint32_t { HEADS_CLOSEST = 0, HEADS_SHADOW = 1, HEADS_LCHAIN = 2, HEADS_PER_ROW = 3 };
// words of a cursor's cache line: the cursor itself, then exact tallies that the waves whose home partition this is add to
// (rays actually traced — the queue counters count slots, holes included —, BSDF-sampled NEE rays that hit a light, BSDF-sampled NEE
// rays the shading pass culled against the lights' root box)
enum : uint32_t { HEAD_CURSOR = 0, HEAD_TALLY0 = 1, HEAD_TALLY1 = 2, HEAD_TALLY2 = 3 };
// every queue is allocated with this many slots past its capacity: a producer that finds the queue full diverts its writes there
// (see wave_reserve) instead of past the end
enum : uint32_t { kQueueDumpSlots = 8192u };
// Striped queue tails (the surface classes' shade queues).  A returning atomic on ONE tail word sustains ~88 reservations per
// microsecond; a 1/8 share of the frame wants ~70.  So a queue's regions are handed out by up to kTailStripes tail words, each in a
// cache line of its own: region r of stripe k is global region r * K + k, i.e. slots [(r * K + k) << log_r, ... + (1 << log_r)).
// Producers rotate through the stripes, so the stripes stay within a few regions of each other and the queue stays dense up to the
// regions the shorter stripes never reached; the consumer reads the K tails once and skips those regions (stripe_valid).  Region
// size and stripe count are powers of two derived from the size of the bounce's input queue (stripes_for), which producer and
// consumer both know.
enum : uint32_t { kWaveTimeSlots = 8192u }; // PT_WAVE_TIMES diagnostic builds: per-wave records of a k_closest launch
enum : uint32_t { kTailStripes = 64u, kTailStrideWords = 32u, kTailWordsPerQueue = kTailStripes * kTailStrideWords };

// Shade queue of one surface class: what the shading pass needs of a hit, written by the traversal kernel in queue order and read
// back linearly (no gather by ray index):  a = direction.xyz | path id,  b = t, u, v | hit id,  c = origin.xyz | unused.
// `c` is not written at bounce 0 (every camera ray starts at the eye).  A slot whose path id is HOLE is skipped.
struct ShadeQueue
{
    f4* a;
    f4* b;
    f4* c;
};

} // namespace pt
