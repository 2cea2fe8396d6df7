// Host scene construction for libptmi (CPU, once per scene).  Follows the reference's builders so that the trees —
// and therefore traversal order, tie-breaking and the per-leaf t_enter estimates that enter triangle-test rounding —
// are the reference's own:
//   Triangle::new            src/tlas/tlas_bvh/blas/primitive.rs:31-54
//   BLASNode::generate_blas  src/tlas/tlas_bvh/blas/blas_bvh.rs:62-136   (sorted sweep SAH, 64 bins)
//   TLASNode::generate_tlas  src/tlas/tlas_bvh.rs:85-138                  (agglomerative clustering)
//   LightSampler::new        src/scene/light_sampler.rs:41-61
//   Camera::new              src/camera.rs:17-31
#include "pt_scene.h"

#include <algorithm>
#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <numeric>
#include <system_error>
#include <thread>
#include <utility>

namespace pt {

namespace {

const float kInf = __builtin_inff();

HostBox box_empty() { return HostBox{bc3(kInf), bc3(-kInf)}; }                          // AABB::identity  boundingbox.rs:59-65
HostBox box_join(const HostBox& a, const HostBox& b) { return HostBox{min3(a.mn, b.mn), max3(a.mx, b.mx)}; } // boundingbox.rs:134-139
float box_area(const HostBox& b)                                                        // boundingbox.rs:90-95
{
    f3 v = b.mx - b.mn;
    return 2.0f * dot3(v, f3{v.z, v.x, v.y});
}
int box_longest_axis(const HostBox& b)                                                  // boundingbox.rs:71-88
{
    f3 l = b.mx - b.mn;
    float m = hmax3(l);
    return l.x == m ? 0 : (l.y == m ? 1 : 2);
}
HostBox box_transform(const HostBox& b, const xf34& m)                                  // boundingbox.rs:51-57 (two corners only)
{
    f3 p = xf_point(m, b.mn), q = xf_point(m, b.mx);
    return HostBox{min3(p, q), max3(p, q)};
}
float axis_of(const f3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

HostTriangle make_triangle(const float* p9, const float* n9)                            // primitive.rs:31-54
{
    HostTriangle t;
    for (int k = 0; k < 3; ++k)
    {
        t.p[k] = f3{p9[k * 3], p9[k * 3 + 1], p9[k * 3 + 2]};
        t.n[k] = f3{n9[k * 3], n9[k * 3 + 1], n9[k * 3 + 2]};
    }
    f3 ab = t.p[1] - t.p[0], ac = t.p[2] - t.p[0];
    f3 n0 = cross3(ab, ac);
    float d0 = dot3(n0, t.p[0]);
    float scale = len_sq(n0);
    f3 n1 = cross3(ac, n0) / scale;
    float d1 = -dot3(n1, t.p[0]);
    f3 n2 = cross3(n0, ab) / scale;
    float d2 = -dot3(n2, t.p[0]);
    t.n0 = f4{n0.x, n0.y, n0.z, d0};
    t.n1 = f4{n1.x, n1.y, n1.z, d1};
    t.n2 = f4{n2.x, n2.y, n2.z, d2};
    return t;
}
HostBox triangle_box(const HostTriangle& t)                                             // primitive.rs:97-103
{
    return HostBox{min3(min3(t.p[0], t.p[1]), t.p[2]), max3(max3(t.p[0], t.p[1]), t.p[2])};
}

m33 inverse33(const m33& m) // glam Mat3A::inverse
{
    f3 r0 = cross3(m.c1, m.c2), r1 = cross3(m.c2, m.c0), r2 = cross3(m.c0, m.c1);
    float inv_det = 1.0f / dot3(m.c2, r2);
    f3 a = r0 * inv_det, b = r1 * inv_det, c = r2 * inv_det;
    return m33{f3{a.x, b.x, c.x}, f3{a.y, b.y, c.y}, f3{a.z, b.z, c.z}};
}
xf34 inverse_affine(const xf34& a) // glam Affine3A::inverse
{
    m33 mi = inverse33(a.m);
    return xf34{mi, -mul(mi, a.t)};
}

// host threads of Scene::new: the cores up to 16; PTMI_BUILD_THREADS overrides (1 = the caller's thread only)
unsigned build_threads()
{
    unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    if (const char* e = std::getenv("PTMI_BUILD_THREADS")) threads = (unsigned)std::max(1, std::min(atoi(e), 64));
    return threads;
}
// fn(lo, hi) over [0, n) in equal slices, one per thread (independent elements only)
template <class F>
void parallel_slices(size_t n, F fn)
{
    const size_t t = std::min<size_t>(build_threads(), n / 4096 + 1);
    if (t <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(t);
    pool.reserve(t);                                   // before any thread exists: emplace_back below never reallocates (a bad_alloc there would
                                                       // unwind past joinable threads and terminate)
    size_t started = 1;
    try
    {
        for (; started < t; ++started)
            pool.emplace_back([&fn, &failed, started, n, t] {
                try { fn(n * started / t, n * (started + 1) / t); }
                catch (...) { failed[started] = std::current_exception(); }              // handed to the caller after the join
            });
    }
    catch (...) {}                                     // fewer threads than asked for (std::system_error, or no memory for one): the caller does the rest
    try
    {
        fn((size_t)0, n / t);
        if (started < t) fn(n * started / t, n);
    }
    catch (...) { failed[0] = std::current_exception(); }
    for (std::thread& th : pool) th.join();
    for (const std::exception_ptr& e : failed)
        if (e) std::rethrow_exception(e);
}

// SAH sweep builder over a contiguous span of (box, primitive) records.  The tree is the reference's, node for node (tests/test_host.py
// compares the arena with the oracle's direct restatement); what differs is how it is reached:
//  * the reference folds a box over each side of every candidate split (64 x n joins per node, blas_bvh.rs:96-110).  The left boxes
//    are running prefixes of one forward fold - the same joins in the same order.  The right boxes come from one reverse fold
//    join(item, suffix): `a < b ? a : b` keeps the later operand on a tie, so with the item first a +0 / -0 tie resolves to the later
//    item exactly as the forward fold over the same items does, and all other finite or infinite values do not depend on the order.
//    A NaN does (the forward fold forgets everything before it): a span holding one takes the direct evaluation - add_model
//    refuses non-finite vertices, so that branch only keeps the builder free of preconditions;
//  * the two children of a large node are built by two threads into arenas of their own and spliced in the reference's post-order.
struct SweepItem { HostBox box; uint32_t prim; };
struct alignas(128) SweepArena { std::vector<HostNode> nodes; std::vector<uint32_t> prim_ids; };   // two threads' arenas never share a cache line
struct SweepBuilder
{
    std::vector<SweepItem>& items;
    static constexpr size_t kBins = 64;             // DESIRED_BINS       blas_bvh.rs:13
    static constexpr float kTraversal = 1.0f;       // TRAVERSAL_COST     blas_bvh.rs:15
    static constexpr float kIntersect = 2.0f;       // INTERSECTION_COST  blas_bvh.rs:16
    static constexpr size_t kForkSpan = 4096;       // smaller nodes are not worth a thread

    std::vector<uint64_t> keys;                     // sort scratch, indexed like `items` (threads work on disjoint spans)
    std::vector<SweepItem> moved;
    explicit SweepBuilder(std::vector<SweepItem>& it) : items(it), keys(it.size()), moved(it.size()) {}

    HostBox span_box(size_t lo, size_t hi) const                                        // blas_bvh.rs:28-33
    {
        HostBox b = box_empty();
        for (size_t i = lo; i < hi; ++i) b = box_join(b, items[i].box);
        return b;
    }
    // The reference's sort is stable and compares box minima with total_cmp (glidesort, blas_bvh.rs:86-91).  A stable sort by key
    // is a sort by (key, position), and those pairs are distinct: an ordinary in-place sort of 64-bit words, no allocation per node.
    void sort_span(size_t lo, size_t hi, int axis)
    {
        for (size_t i = lo; i < hi; ++i)
            keys[i] = ((uint64_t)((uint32_t)total_order_key(axis_of(items[i].box.mn, axis)) ^ 0x80000000u) << 32) | (uint64_t)(i - lo);
        std::sort(keys.begin() + lo, keys.begin() + hi);
        for (size_t i = lo; i < hi; ++i) moved[i] = items[lo + (size_t)(uint32_t)keys[i]];
        std::copy(moved.begin() + lo, moved.begin() + hi, items.begin() + lo);
    }
    static bool has_nan(const HostBox& b)
    {
        auto nan = [](float x) { return x != x; };
        return nan(b.mn.x) || nan(b.mn.y) || nan(b.mn.z) || nan(b.mx.x) || nan(b.mx.y) || nan(b.mx.z);
    }
    static uint32_t emit(SweepArena& out, const HostNode& n) { out.nodes.push_back(n); return (uint32_t)out.nodes.size() - 1; }
    // append a finished subtree (arena-local links) to `out`; returns the subtree root's index there
    static uint32_t splice(SweepArena& out, const SweepArena& sub, uint32_t sub_root)
    {
        const uint32_t node_base = (uint32_t)out.nodes.size(), prim_base = (uint32_t)out.prim_ids.size();
        out.prim_ids.insert(out.prim_ids.end(), sub.prim_ids.begin(), sub.prim_ids.end());
        out.nodes.reserve(out.nodes.size() + sub.nodes.size());
        for (HostNode n : sub.nodes)
        {
            if (n.kind == NODE_BRANCH) { n.a += node_base; n.b += node_base; }
            else n.a += prim_base;
            out.nodes.push_back(n);
        }
        return sub_root + node_base;
    }

    // returns (node id in `out`); depth_out = nodes on the longest path below and including this node
    uint32_t run(SweepArena& out, size_t lo, size_t hi, int parent_axis, uint32_t* depth_out, int fork_levels)
    {
        const size_t span = hi - lo;
        if (span == 1)                                                                   // LeafSingle  blas_bvh.rs:67-75
        {
            uint32_t first = (uint32_t)out.prim_ids.size();
            out.prim_ids.push_back(items[lo].prim);
            *depth_out = 1;
            return emit(out, HostNode{items[lo].box, NODE_TRIS, first, 1});
        }
        HostBox bb = box_empty();
        bool any_order = true;
        for (size_t i = lo; i < hi; ++i)
        {
            bb = box_join(bb, items[i].box);
            any_order = any_order && !has_nan(items[i].box);
        }
        const float bb_sa = box_area(bb);
        const int axis = box_longest_axis(bb);
        if (axis != parent_axis) sort_span(lo, hi, axis);                                // blas_bvh.rs:86-91
        const size_t bin = std::max<size_t>(span / kBins, 1);
        const size_t n_candidates = span / bin - 1;                                      // <= 2 * kBins - 2
        float right_area[2 * kBins];
        if (any_order)
        {
            HostBox suffix = box_empty();
            for (size_t i = span; i-- > bin;)
            {
                suffix = box_join(items[lo + i].box, suffix);               // ties (a zero of either sign) keep the later item, as the forward fold does
                if (i % bin == 0 && i / bin <= n_candidates) right_area[i / bin - 1] = box_area(suffix);
            }
        }
        size_t best_j = 0, done = 0;
        float best_cost = 0.0f;
        HostBox prefix = box_empty();
        for (size_t c = 0; c < n_candidates; ++c)                                        // blas_bvh.rs:96-110
        {
            const size_t j = (c + 1) * bin;
            for (; done < j; ++done) prefix = box_join(prefix, items[lo + done].box);
            const float la = box_area(prefix), ra = any_order ? right_area[c] : box_area(span_box(lo + j, hi));
            const float cost = kTraversal + ((float)j * la + (float)(span - j) * ra) * kIntersect / bb_sa;
            // min_by(total_cmp): first of equal minima wins
            if (c == 0 || total_order_key(cost) < total_order_key(best_cost)) { best_cost = cost; best_j = j; }
        }
        const float leaf_cost = kIntersect * (float)span;                                // blas_bvh.rs:112
        if (leaf_cost < best_cost)                                                       // Leaf{primitive_ids}  blas_bvh.rs:114-122
        {
            uint32_t first = (uint32_t)out.prim_ids.size();
            for (size_t i = lo; i < hi; ++i) out.prim_ids.push_back(items[i].prim);
            *depth_out = 1;
            return emit(out, HostNode{bb, NODE_TRIS, first, (uint32_t)span});
        }
        uint32_t dl = 0, dr = 0, left = 0, right = 0;                                    // blas_bvh.rs:125-133
        bool forked = false;
        if (fork_levels > 0 && span >= kForkSpan)
        {
            SweepArena la, ra;
            try
            {
                la.nodes.reserve(2 * best_j); la.prim_ids.reserve(best_j);
                ra.nodes.reserve(2 * (span - best_j)); ra.prim_ids.reserve(span - best_j);
                std::exception_ptr left_failed;
                std::thread t([&] {
                    try { left = run(la, lo, lo + best_j, axis, &dl, fork_levels - 1); }
                    catch (...) { left_failed = std::current_exception(); }              // e.g. bad_alloc: rethrown in the parent after the join
                });
                forked = true;
                try { right = run(ra, lo + best_j, hi, axis, &dr, fork_levels - 1); }
                catch (...) { t.join(); throw; }
                t.join();
                if (left_failed) std::rethrow_exception(left_failed);
            }
            catch (const std::system_error&) {}                                          // no thread to be had: build the children in turn
            if (forked)
            {
                left = splice(out, la, left);
                right = splice(out, ra, right);
            }
        }
        if (!forked)
        {
            left = run(out, lo, lo + best_j, axis, &dl, 0);
            right = run(out, lo + best_j, hi, axis, &dr, 0);
        }
        *depth_out = 1 + std::max(dl, dr);
        return emit(out, HostNode{bb, NODE_BRANCH, left, right});
    }
};

uint32_t tree_depth(const std::vector<HostNode>& nodes, uint32_t root)
{
    // children always precede parents in arena order
    std::vector<uint32_t> d(nodes.size(), 1);
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].kind == NODE_BRANCH) d[i] = 1 + std::max(d[nodes[i].a], d[nodes[i].b]);
    return nodes.empty() ? 0 : d[root];
}

// ---- 4x4 helpers for the camera (column-major float[16]) ----
void mat4_mul(const float* a, const float* b, float* out) // glam Mat4 * Mat4 (mul_vec4 per column)
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
        {
            float v = a[0 * 4 + r] * b[c * 4 + 0];
            v = v + a[1 * 4 + r] * b[c * 4 + 1];
            v = v + a[2 * 4 + r] * b[c * 4 + 2];
            v = v + a[3 * 4 + r] * b[c * 4 + 3];
            out[c * 4 + r] = v;
        }
}
void mat4_inverse(const float* m, float* out) // glam Mat4::inverse (cofactor expansion)
{
    const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    const float m30 = m[12], m31 = m[13], m32 = m[14], m33_ = m[15];
    const float c00 = m22 * m33_ - m32 * m23, c02 = m12 * m33_ - m32 * m13, c03 = m12 * m23 - m22 * m13;
    const float c04 = m21 * m33_ - m31 * m23, c06 = m11 * m33_ - m31 * m13, c07 = m11 * m23 - m21 * m13;
    const float c08 = m21 * m32 - m31 * m22, c10 = m11 * m32 - m31 * m12, c11 = m11 * m22 - m21 * m12;
    const float c12 = m20 * m33_ - m30 * m23, c14 = m10 * m33_ - m30 * m13, c15 = m10 * m23 - m20 * m13;
    const float c16 = m20 * m32 - m30 * m22, c18 = m10 * m32 - m30 * m12, c19 = m10 * m22 - m20 * m12;
    const float c20 = m20 * m31 - m30 * m21, c22 = m10 * m31 - m30 * m11, c23 = m10 * m21 - m20 * m11;
    const float fac0[4] = {c00, c00, c02, c03}, fac1[4] = {c04, c04, c06, c07}, fac2[4] = {c08, c08, c10, c11};
    const float fac3[4] = {c12, c12, c14, c15}, fac4[4] = {c16, c16, c18, c19}, fac5[4] = {c20, c20, c22, c23};
    const float v0[4] = {m10, m00, m00, m00}, v1[4] = {m11, m01, m01, m01}, v2[4] = {m12, m02, m02, m02}, v3[4] = {m13, m03, m03, m03};
    const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    float inv[16];
    for (int i = 0; i < 4; ++i)
    {
        inv[0 + i] = ((v1[i] * fac0[i] - v2[i] * fac1[i]) + v3[i] * fac2[i]) * sa[i];
        inv[4 + i] = ((v0[i] * fac0[i] - v2[i] * fac3[i]) + v3[i] * fac4[i]) * sb[i];
        inv[8 + i] = ((v0[i] * fac1[i] - v1[i] * fac3[i]) + v3[i] * fac5[i]) * sa[i];
        inv[12 + i] = ((v0[i] * fac2[i] - v1[i] * fac4[i]) + v2[i] * fac5[i]) * sb[i];
    }
    const float d0 = m[0] * inv[0], d1 = m[1] * inv[4], d2 = m[2] * inv[8], d3 = m[3] * inv[12];
    const float det = d0 + d1 + d2 + d3;
    const float rcp = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * rcp;
}
f3 unit3_recip(f3 v) { float r = 1.0f / sqrtf(dot3(v, v)); return v * r; } // glam scalar Vec3::normalize

} // namespace

int HostScene::add_material(int kind, const float colour[3], float roughness, float ior, bool has_volume, const float vol_abs[3], float k,
                            float c, float g)
{
    if (kind < 0 || kind > (int)MAT_DIELECTRIC) return -1;
    DMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.kind = (uint32_t)kind;
    m.colour[0] = colour[0]; m.colour[1] = colour[1]; m.colour[2] = colour[2];
    m.ior = ior;
    if (kind == (int)MAT_GGX_METAL || kind == (int)MAT_GGX_DIELECTRIC) m.alpha = clamp_rs(sq(roughness), 0.0001f, 0.9999f); // material.rs:294,309
    if (has_volume && (kind == (int)MAT_GGX_DIELECTRIC || kind == (int)MAT_DIELECTRIC))
    {
        m.has_volume = 1;
        if (k != 0.0f) { m.vol_flags |= 1u; m.vol_abs[0] = vol_abs[0] * k; m.vol_abs[1] = vol_abs[1] * k; m.vol_abs[2] = vol_abs[2] * k; } // volume.rs:112,138
        if (c != 0.0f) { m.vol_flags |= 2u; m.vol_c = c; m.vol_g = clamp_rs(g, -0.999f, 0.999f); }                                       // volume.rs:27,139
    }
    materials.push_back(m);
    built = false;
    return (int)materials.size() - 1;
}

int HostScene::add_model(const float* positions, const float* normals, uint32_t n_tris, int material, const float* affines, uint32_t n_inst)
{
    if (!positions || !normals || n_tris == 0 || material < 0 || material >= (int)materials.size() || (n_inst && !affines)) return -1;
    // Vertex positions must be numbers: a NaN or infinite coordinate makes every box it touches NaN / infinite, the builders'
    // surface-area comparisons (tlas_bvh.rs:56-83, blas_bvh.rs:93-110) then have no minimum — the reference's `min_by(partial_cmp)
    // .unwrap()` panics — and no traversal could return anything meaningful.  (Found by the host sanitizer build: the TLAS builder
    // indexed its work list with "no partner".)
    for (size_t i = 0; i < (size_t)n_tris * 9; ++i)
        if (!finite_f(positions[i])) return -1;
    HostModel m;
    m.n_tris = n_tris;
    m.material = material;
    m.positions.assign(positions, positions + (size_t)n_tris * 9);
    m.normals.assign(normals, normals + (size_t)n_tris * 9);
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        const float* r = affines + (size_t)i * 12;
        xf34 x{m33{f3{r[0], r[4], r[8]}, f3{r[1], r[5], r[9]}, f3{r[2], r[6], r[10]}}, f3{r[3], r[7], r[11]}};
        // model.rs:40-44: to_scale_rotation_translation().0 == Vec3::ONE or panic
        const float det = dot3(x.m.c2, cross3(x.m.c0, x.m.c1));
        const float sx = len3(x.m.c0) * signum_rs(det), sy = len3(x.m.c1), sz = len3(x.m.c2);
        if (!(sx == 1.0f && sy == 1.0f && sz == 1.0f)) return -4;
        m.matrices.push_back(x);
    }
    models.push_back(std::move(m));
    built = false;
    return (int)models.size() - 1;
}

// load_obj  src/tlas/tlas_bvh/blas.rs:44-131: `v`, `vn` (normalised on load), `f` with v/vt/vn references (1-based, negative =
// relative to the end), fan triangulation, un-normalised face normal when the normal reference is 0; every other keyword is
// skipped.  Where the reference panics (missing third reference field, unparsable number, index out of range) this returns
// an error; blank lines are skipped (the reference would panic on `tokens[0]`).
namespace {
bool parse_f32(const std::string& t, float* out)
{
    if (t.empty() || t.find_first_of("xX") != std::string::npos) return false;
    char* end = nullptr;
    *out = strtof(t.c_str(), &end);
    return end && *end == 0;
}
bool parse_index(const std::string& t, size_t count, size_t* out) // usize, or `len + isize` (blas.rs:85-92)
{
    if (t.empty()) return false;
    char* end = nullptr;
    if (t[0] != '-' && t[0] != '+')
    {
        unsigned long long v = strtoull(t.c_str(), &end, 10);
        if (!end || *end != 0) return false;
        *out = (size_t)v;
        return true;
    }
    long long v = strtoll(t.c_str(), &end, 10);
    if (!end || *end != 0) return false;
    *out = (size_t)((long long)count + v);
    return true;
}
} // namespace

int HostScene::add_model_obj(const char* path, int material, const float* affines, uint32_t n_inst, std::string* err)
{
    if (!path) return -1;
    FILE* fp = fopen(path, "rb");
    if (!fp) { if (err) *err = std::string("cannot open ") + path; return -6; }
    std::vector<f3> positions{f3{0, 0, 0}}, normals{f3{0, 0, 0}}; // index 0 is a dummy: OBJ indices are 1-based  blas.rs:46-47
    std::vector<float> out_p, out_n;
    std::string line;
    int c;
    size_t line_no = 0;
    bool eof = false;
    while (!eof)
    {
        line.clear();
        while ((c = fgetc(fp)) != EOF && c != '\n') line.push_back((char)c);
        if (c == EOF) eof = true;
        ++line_no;
        std::vector<std::string> tok;
        size_t i = 0;
        while (i < line.size())
        {
            while (i < line.size() && isspace((unsigned char)line[i])) ++i;
            size_t j = i;
            while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
            if (j > i) tok.push_back(line.substr(i, j - i));
            i = j;
        }
        if (tok.empty()) continue;
        auto bad = [&](const char* what) {
            if (err) *err = std::string(path) + ":" + std::to_string(line_no) + ": " + what;
            fclose(fp);
            return -7;
        };
        if (tok[0] == "v" || tok[0] == "vn")
        {
            f3 v;
            if (tok.size() < 4 || !parse_f32(tok[1], &v.x) || !parse_f32(tok[2], &v.y) || !parse_f32(tok[3], &v.z)) return bad("expected three numbers");
            if (tok[0] == "v" && (!finite_f(v.x) || !finite_f(v.y) || !finite_f(v.z))) return bad("vertex coordinate is not a finite number");
            if (tok[0] == "v") positions.push_back(v);
            else normals.push_back(unit3(v));                                                  // blas.rs:74
        }
        else if (tok[0] == "f")
        {
            std::vector<std::pair<size_t, size_t>> refs;
            for (size_t k = 1; k < tok.size(); ++k)
            {
                std::vector<std::string> idx;
                size_t a = 0;
                for (;;)
                {
                    size_t b = tok[k].find('/', a);
                    idx.push_back(tok[k].substr(a, b == std::string::npos ? std::string::npos : b - a));
                    if (b == std::string::npos) break;
                    a = b + 1;
                }
                size_t vi, ni;
                if (idx.size() < 3) return bad("face reference needs v/vt/vn");                 // indices[2] panics in the reference
                if (!parse_index(idx[0], positions.size(), &vi) || !parse_index(idx[2], normals.size(), &ni)) return bad("bad face index");
                if (vi >= positions.size() || ni >= normals.size()) return bad("face index out of range");
                refs.push_back({vi, ni});
            }
            if (refs.size() < 3) continue;                                                       // `1..(len - 1)` is empty
            for (size_t k = 1; k + 1 < refs.size(); ++k)                                         // blas.rs:97-119
            {
                const std::pair<size_t, size_t> tri[3] = {refs[0], refs[k], refs[k + 1]};
                for (const auto& r : tri)
                {
                    const f3 p = positions[r.first];
                    f3 n;
                    if (r.second != 0) n = normals[r.second];
                    else n = cross3(positions[tri[1].first] - positions[tri[0].first], positions[tri[2].first] - positions[tri[0].first]);
                    out_p.insert(out_p.end(), {p.x, p.y, p.z});
                    out_n.insert(out_n.end(), {n.x, n.y, n.z});
                }
            }
        }
    }
    fclose(fp);
    if (out_p.empty()) { if (err) *err = std::string(path) + ": no faces"; return -7; }
    return add_model(out_p.data(), out_n.data(), (uint32_t)(out_p.size() / 9), material, affines, n_inst);
}

void HostScene::build_blas(HostBlas& out, const HostModel& m)                           // BLAS::new  blas.rs:174-201
{
    out = HostBlas();
    out.material = m.material;
    out.tris.resize(m.n_tris);
    std::vector<SweepItem> items(m.n_tris);
    parallel_slices(m.n_tris, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i)
        {
            out.tris[i] = make_triangle(&m.positions[i * 9], &m.normals[i * 9]);
            items[i] = SweepItem{triangle_box(out.tris[i]), (uint32_t)i};
        }
    });
    const auto tt0 = std::chrono::steady_clock::now();
    SweepBuilder sb{items};
    SweepArena arena;
    arena.nodes.reserve(2 * items.size());
    arena.prim_ids.reserve(items.size());
    int fork_levels = 0;                                                                 // 2^levels builder threads
    for (unsigned t = build_threads(); t > 1; t >>= 1) ++fork_levels;
    out.root = sb.run(arena, 0, items.size(), 4, &out.depth, fork_levels);               // last_split_axis = 4  blas.rs:191
    out.nodes.assign(arena.nodes.begin(), arena.nodes.end());                            // exact size; the arena was reserved for the worst case
    out.prim_ids = std::move(arena.prim_ids);
    if (m.n_tris >= 1024 && std::getenv("PTMI_DEBUG_BUILD"))
        fprintf(stderr, "[ptmi]  BLAS of %u triangles: %zu nodes, depth %u, SAH sweep %.1f ms on up to %d threads\n", m.n_tris, out.nodes.size(), out.depth,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(), 1 << fork_levels);
}

void HostScene::build_tlas(HostTlas& out, const std::vector<uint32_t>& model_ids)       // TLAS::new tlas.rs:24-53
{
    out = HostTlas();
    out.models = model_ids;
    std::vector<uint32_t> open; // the `nodes` worklist of tlas_bvh.rs:88
    for (uint32_t a = 0; a < model_ids.size(); ++a)
    {
        const uint32_t mi = model_ids[a];
        const HostBox root_box = blas[mi].nodes[blas[mi].root].box;
        for (const xf34& mat : models[mi].matrices)
        {
            const uint32_t inst = (uint32_t)out.instances.size();
            out.instances.push_back(HostInstance{a, mi, mat, inverse_affine(mat)});
            out.nodes.push_back(HostNode{box_transform(root_box, mat), NODE_INSTANCE, inst, a});
            open.push_back((uint32_t)out.nodes.size() - 1);
        }
    }
    if (open.empty()) return;
    auto best_partner = [&](size_t self) -> size_t {                                     // find_best_match  tlas_bvh.rs:56-83
        float best = kInf;
        size_t idx = SIZE_MAX;
        for (size_t i = 0; i < open.size(); ++i)
        {
            if (i == self) continue;
            const float sa = box_area(box_join(out.nodes[open[self]].box, out.nodes[open[i]].box));
            if (sa < best) { best = sa; idx = i; }
        }
        if (idx == SIZE_MAX) idx = self == 0 ? 1 : 0; // every joined area is infinite (boxes ~1e19 apart overflow binary32): any partner
        return idx;
    };
    auto take = [&](size_t i) { uint32_t v = open[i]; open[i] = open.back(); open.pop_back(); return v; }; // Vec::swap_remove
    size_t a = 0, b = open.size() > 1 ? best_partner(0) : SIZE_MAX;
    while (open.size() > 1)                                                              // tlas_bvh.rs:110-136
    {
        const size_t c = best_partner(b);
        if (a == c)
        {
            uint32_t n1, n2;
            if (a > b) { n1 = take(a); n2 = take(b); } else { n1 = take(b); n2 = take(a); }
            a = open.size();
            out.nodes.push_back(HostNode{box_join(out.nodes[n1].box, out.nodes[n2].box), NODE_BRANCH, n1, n2});
            open.push_back((uint32_t)out.nodes.size() - 1);
            b = open.size() > 1 ? best_partner(a) : SIZE_MAX;
        }
        else { a = b; b = c; }
    }
    out.root = open[0];
    out.depth = tree_depth(out.nodes, out.root);
}

void HostScene::build_lights()                                                          // tlas.rs:55-64, blas.rs:203-212, light_sampler.rs:41-61
{
    light_items.clear();
    light_weight_sum = 0;
    std::vector<float> weights;
    for (uint32_t a = 0; a < lights.models.size(); ++a)
    {
        const HostBlas& bl = blas[lights.models[a]];
        const DMaterial& mat = materials[bl.material];
        const float emitted_len = len3(f3{mat.colour[0], mat.colour[1], mat.colour[2]});
        for (uint32_t p = 0; p < bl.tris.size(); ++p)
        {
            const float area = 0.5f * len3(f3{bl.tris[p].n0.x, bl.tris[p].n0.y, bl.tris[p].n0.z}); // primitive.rs:94
            weights.push_back(area * emitted_len);
            light_items.push_back(HostLight{a, p, 0, 0});
        }
    }
    float sum = 0.0f;
    for (float w : weights) sum = sum + w;
    light_weight_sum = sum;
    float running = 0.0f;
    for (size_t i = 0; i < light_items.size(); ++i)
    {
        light_items[i].pdf = weights[i] / sum;
        running += light_items[i].pdf;
        light_items[i].cdf = running;
    }
}

int HostScene::build(std::string* err)                                                  // Scene::new  scene.rs:21-35
{
    if (models.empty()) { if (err) *err = "no models"; return -3; }
    const bool dbg = std::getenv("PTMI_DEBUG_BUILD") != nullptr;                          // stage times of Scene::new on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    blas.resize(models.size());
    for (size_t i = 0; i < models.size(); ++i) build_blas(blas[i], models[i]);
    const auto t1 = now();
    std::vector<uint32_t> all(models.size()), emissive;
    std::iota(all.begin(), all.end(), 0u);
    for (uint32_t i = 0; i < models.size(); ++i)
        if (materials[models[i].material].kind == MAT_EMISSIVE) emissive.push_back(i);
    build_tlas(world, all);
    build_tlas(lights, emissive);
    build_lights();
    const auto t2 = now();
    int r = flatten(err);
    if (dbg) fprintf(stderr, "[ptmi] scene build: BLAS %.1f ms, TLAS + lights %.1f ms, flatten %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
    built = (r == 0);
    return r;
}

int HostScene::flatten(std::string* err)
{
    FlatScene f;
    f.materials = materials;
    for (const DMaterial& m : materials) f.has_volumes = f.has_volumes || m.has_volume != 0;
    // absolute node layout: world TLAS | lights TLAS | BLAS 0 | BLAS 1 | ...
    const uint32_t world_base = 0;
    const uint32_t lights_base = (uint32_t)world.nodes.size();
    std::vector<uint32_t> blas_base(blas.size());
    uint32_t cursor = lights_base + (uint32_t)lights.nodes.size();
    uint32_t tri_cursor = 0;
    f.tri_base.resize(blas.size());
    for (size_t i = 0; i < blas.size(); ++i)
    {
        blas_base[i] = cursor;
        cursor += (uint32_t)blas[i].nodes.size();
        f.tri_base[i] = tri_cursor;
        tri_cursor += (uint32_t)blas[i].prim_ids.size();
    }
    if (cursor > NODE_PAYLOAD_MASK || tri_cursor > NODE_PAYLOAD_MASK) { if (err) *err = "scene too large for 30-bit node payloads"; return -5; }
    f.nodes.resize(cursor);
    f.inst_base = {0u, (uint32_t)world.instances.size()};

    auto put_box = [](DNode& d, const HostBox& b) {
        d.mn[0] = b.mn.x; d.mn[1] = b.mn.y; d.mn[2] = b.mn.z;
        d.mx[0] = b.mx.x; d.mx[1] = b.mx.y; d.mx[2] = b.mx.z;
    };
    // Device order of one tree: the children of a branch in adjacent slots (left, right: one 64-byte fetch serves both box tests);
    // the top kBreadthFirstLevels levels breadth-first (they are hot in every cache anyway), every subtree below them depth-first
    // (a pair, then the left child's subtree, then the right's), so that a descending ray's next pair is often in the line it
    // just touched.  Against breadth-first throughout: 82 k / 328 k meshes and the three spheres about -1 % (same-box A/B of 1 / 6 /
    // 10 / 14 top levels: alike).
    constexpr int kBreadthFirstLevels = 6;
    auto layout = [](const std::vector<HostNode>& nodes, uint32_t root, uint32_t base, std::vector<uint32_t>& where) -> bool {
        where.assign(nodes.size(), MISS_ID);
        if (root == MISS_ID) return nodes.empty();
        where[root] = base;
        uint32_t next = base + 1u;
        std::vector<std::pair<uint32_t, int>> frontier{{root, 0}}, deep;
        for (size_t k = 0; k < frontier.size(); ++k)
        {
            const HostNode& n = nodes[frontier[k].first];
            if (n.kind != NODE_BRANCH) continue;
            if (frontier[k].second >= kBreadthFirstLevels) { deep.push_back(frontier[k]); continue; }
            where[n.a] = next;
            where[n.b] = next + 1u;
            next += 2u;
            frontier.push_back({n.a, frontier[k].second + 1});
            frontier.push_back({n.b, frontier[k].second + 1});
        }
        for (const auto& d : deep)
        {
            std::vector<uint32_t> stack{d.first};
            while (!stack.empty())
            {
                const uint32_t i = stack.back();
                stack.pop_back();
                const HostNode& n = nodes[i];
                if (n.kind != NODE_BRANCH) continue;
                where[n.a] = next;
                where[n.b] = next + 1u;
                next += 2u;
                stack.push_back(n.b);
                stack.push_back(n.a);
            }
        }
        return next == base + nodes.size();
    };
    auto leaf_link = [&](uint32_t first, uint32_t count) -> uint32_t {
        if (count >= 1u && count <= LEAF_MAX_COUNT && first <= LEAF_FIRST_MASK)
            return (NODE_TRIS << NODE_KIND_SHIFT) | ((count - 1u) << LEAF_FIRST_BITS) | first;
        f.big_leaves.push_back(first);
        f.big_leaves.push_back(count);
        return (NODE_TRIS_BIG << NODE_KIND_SHIFT) | (uint32_t)(f.big_leaves.size() / 2u - 1u);
    };
    std::vector<uint32_t> slot_of;
    auto put_tlas = [&](const HostTlas& t, uint32_t base, uint32_t inst_base) -> bool {
        if (!layout(t.nodes, t.root, base, slot_of)) return false;
        for (size_t i = 0; i < t.nodes.size(); ++i)
        {
            DNode& d = f.nodes[slot_of[i]];
            put_box(d, t.nodes[i].box);
            if (t.nodes[i].kind == NODE_BRANCH) { d.link = (NODE_BRANCH << NODE_KIND_SHIFT) | slot_of[t.nodes[i].a]; d.aux = slot_of[t.nodes[i].b]; }
            else { d.link = (NODE_INSTANCE << NODE_KIND_SHIFT) | (inst_base + t.nodes[i].a); d.aux = 0u; }
        }
        return true;
    };
    if (!put_tlas(world, world_base, f.inst_base[0]) || !put_tlas(lights, lights_base, f.inst_base[1]))
    {
        if (err) *err = "internal: TLAS arena holds nodes outside the tree";
        return -5;
    }
    f.world_root = world.root == MISS_ID ? MISS_ID : world_base;   // the root is the first node of its tree
    f.lights_root = lights.root == MISS_ID ? MISS_ID : lights_base;

    uint32_t max_blas_depth = 0;
    std::vector<uint32_t> blas_root_at(blas.size(), MISS_ID);
    f.tri_isect.reserve(tri_cursor);
    f.tri_shade.reserve(tri_cursor);
    f.tri_pos.reserve(tri_cursor);
    f.tri_orig.reserve(tri_cursor);
    for (size_t i = 0; i < blas.size(); ++i)
    {
        const HostBlas& bl = blas[i];
        max_blas_depth = std::max(max_blas_depth, bl.depth);
        if (!layout(bl.nodes, bl.root, blas_base[i], slot_of)) { if (err) *err = "internal: BLAS arena holds nodes outside the tree"; return -5; }
        blas_root_at[i] = blas_base[i];
        for (size_t n = 0; n < bl.nodes.size(); ++n)
        {
            DNode& d = f.nodes[slot_of[n]];
            put_box(d, bl.nodes[n].box);
            if (bl.nodes[n].kind == NODE_BRANCH) { d.link = (NODE_BRANCH << NODE_KIND_SHIFT) | slot_of[bl.nodes[n].a]; d.aux = slot_of[bl.nodes[n].b]; }
            else { d.link = leaf_link(f.tri_base[i] + bl.nodes[n].a, bl.nodes[n].b); d.aux = bl.nodes[n].b; }
        }
        const size_t at = f.tri_orig.size(); // triangles stored in leaf order
        f.tri_isect.resize(at + bl.prim_ids.size());
        f.tri_shade.resize(at + bl.prim_ids.size());
        f.tri_pos.resize(at + bl.prim_ids.size());
        f.tri_orig.resize(at + bl.prim_ids.size());
        parallel_slices(bl.prim_ids.size(), [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k)
            {
                const uint32_t id = bl.prim_ids[k];
                const HostTriangle& t = bl.tris[id];
                f.tri_isect[at + k] = DTriIsect{t.n0, t.n1, t.n2};
                f.tri_shade[at + k] = DTriVerts{f4{t.n[0].x, t.n[0].y, t.n[0].z, 0}, f4{t.n[1].x, t.n[1].y, t.n[1].z, 0}, f4{t.n[2].x, t.n[2].y, t.n[2].z, 0}};
                f.tri_pos[at + k] = DTriVerts{f4{t.p[0].x, t.p[0].y, t.p[0].z, 0}, f4{t.p[1].x, t.p[1].y, t.p[1].z, 0}, f4{t.p[2].x, t.p[2].y, t.p[2].z, 0}};
                f.tri_orig[at + k] = id;
            }
        });
    }
    auto put_instances = [&](const HostTlas& t) {
        for (const HostInstance& hi : t.instances)
        {
            DInstance d;
            std::memset(&d, 0, sizeof(d));
            const xf34* src[2] = {&hi.inv, &hi.fwd};
            float* dst[2] = {d.inv, d.fwd};
            for (int k = 0; k < 2; ++k)
            {
                const xf34& x = *src[k];
                const float rows[12] = {x.m.c0.x, x.m.c1.x, x.m.c2.x, x.t.x, x.m.c0.y, x.m.c1.y, x.m.c2.y, x.t.y, x.m.c0.z, x.m.c1.z, x.m.c2.z, x.t.z};
                std::memcpy(dst[k], rows, sizeof(rows));
            }
            d.root = blas_root_at[hi.model];
            d.root_node = f.nodes[d.root];
            d.blas = hi.blas;
            d.material = (uint32_t)blas[hi.model].material;
            switch (materials[d.material].kind)
            {
            case MAT_LAMBERTIAN: d.qclass = Q_LAMBERT; break;
            case MAT_SPECULAR: d.qclass = Q_SPECULAR; break;
            case MAT_DIELECTRIC: d.qclass = Q_DIELECTRIC; break;
            case MAT_GGX_METAL:
            case MAT_GGX_DIELECTRIC: d.qclass = Q_GGX; break;
            default: d.qclass = Q_TERMINAL; break;
            }
            // with participating media an emissive hit is preceded by the volume interaction (integrator.rs:189-205), which may
            // scatter the path onwards: such hits are shaded by a surface kernel instead of the terminal pass
            if (f.has_volumes && materials[d.material].kind == MAT_EMISSIVE) d.qclass = Q_LAMBERT;
            {
                // bit pattern of Affine3A::IDENTITY.inverse(): unit matrix with +0 zeros, translation -0
                static const uint32_t ident[12] = {0x3f800000u, 0, 0, 0x80000000u, 0, 0x3f800000u, 0, 0x80000000u, 0, 0, 0x3f800000u, 0x80000000u};
                if (std::memcmp(d.inv, ident, sizeof(ident)) == 0) d.qclass |= INSTANCE_IDENTITY;
            }
            f.instances.push_back(d);
        }
    };
    put_instances(world);
    put_instances(lights);

    // leaf-order position of every load-order primitive, per model
    std::vector<std::vector<uint32_t>> where(blas.size());
    for (size_t i = 0; i < blas.size(); ++i)
    {
        where[i].assign(blas[i].tris.size(), 0);
        for (size_t k = 0; k < blas[i].prim_ids.size(); ++k) where[i][blas[i].prim_ids[k]] = f.tri_base[i] + (uint32_t)k;
    }
    for (const HostLight& l : light_items)
    {
        const uint32_t mi = lights.models[l.blas];
        f.lights.push_back(DLight{where[mi][l.prim], (uint32_t)blas[mi].material, l.pdf, l.cdf});
    }
    f.light_weight_sum = light_weight_sum;

    uint32_t pb = 1;
    while ((1ull << pb) < (uint64_t)std::max<uint32_t>(tri_cursor, 2)) ++pb;
    f.prim_bits = pb;
    if ((uint64_t)f.instances.size() >= (1ull << (32 - pb)) - 1) { if (err) *err = "instance x triangle id does not fit 32 bits"; return -5; }
    // Exact bound of the shared traversal stack.  A node at depth d (root = 1) is popped with at most d-1 pending siblings
    // below it and pushes at most two children, so a tree of depth D never holds more than D entries; while a BLAS is
    // being traversed its TLAS leaf has been popped, leaving at most D_tlas - 1 TLAS entries underneath.
    const uint32_t tlas_depth = std::max(world.depth, lights.depth);
    f.stack_entries = std::max(tlas_depth, (tlas_depth > 0 ? tlas_depth - 1 : 0) + max_blas_depth);
    flat = std::move(f);
    return 0;
}

namespace {
// ---- the few glam 0.23 quaternion routines Camera needs (camera.rs:23, 46-49), in glam's operation order
struct quat { float x, y, z, w; };

quat quat_mul(quat l, quat r)                                                            // Quat::mul_quat, SSE2 (rtm::quat_mul) lane formulas
{
    return quat{(l.w * r.x + l.x * r.w) + (l.y * r.z + -(l.z * r.y)),
                (l.w * r.y + -(l.x * r.z)) + (l.y * r.w + l.z * r.x),
                (l.w * r.z + l.x * r.y) + (-(l.y * r.x) + l.z * r.w),
                (l.w * r.w + -(l.x * r.x)) + (-(l.y * r.y) + -(l.z * r.z))};
}

quat quat_from_axes(f3 ax, f3 ay, f3 az)                                                 // Quat::from_rotation_axes (from_mat3)
{
    const float m00 = ax.x, m01 = ax.y, m02 = ax.z, m10 = ay.x, m11 = ay.y, m12 = ay.z, m20 = az.x, m21 = az.y, m22 = az.z;
    if (m22 <= 0.0f)
    {
        const float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f)
        {
            const float four_xsq = omm22 - dif10, inv4x = 0.5f / std::sqrt(four_xsq);
            return quat{four_xsq * inv4x, (m01 + m10) * inv4x, (m02 + m20) * inv4x, (m12 - m21) * inv4x};
        }
        const float four_ysq = omm22 + dif10, inv4y = 0.5f / std::sqrt(four_ysq);
        return quat{(m01 + m10) * inv4y, four_ysq * inv4y, (m12 + m21) * inv4y, (m20 - m02) * inv4y};
    }
    const float sum10 = m11 + m00, opm22 = 1.0f + m22;
    if (sum10 <= 0.0f)
    {
        const float four_zsq = opm22 - sum10, inv4z = 0.5f / std::sqrt(four_zsq);
        return quat{(m02 + m20) * inv4z, (m12 + m21) * inv4z, four_zsq * inv4z, (m01 - m10) * inv4z};
    }
    const float four_wsq = opm22 + sum10, inv4w = 0.5f / std::sqrt(four_wsq);
    return quat{(m12 - m21) * inv4w, (m20 - m02) * inv4w, (m01 - m10) * inv4w, four_wsq * inv4w};
}

void euler_yxz_of(quat q, float* first, float* second)                                   // EulerRot::YXZ convert_quat (first, second)
{
    *first = atan2_det(2.0f * (q.x * q.z + q.w * q.y), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z);
    float v = -2.0f * (q.y * q.z - q.w * q.x);
    v = v < -1.0f ? -1.0f : v;                                                           // arc_clamp
    v = v > 1.0f ? 1.0f : v;
    *second = asin_det(v);
}

m33 mat3_of_quat(quat q)                                                                 // Mat3A::from_quat
{
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    return m33{f3{1.0f - (yy + zz), xy + wz, xz - wy}, f3{xy - wz, 1.0f - (xx + zz), yz + wx}, f3{xz + wy, yz - wx, 1.0f - (xx + yy)}};
}

} // namespace

void HostScene::set_camera(const float eye_[3], const float target_[3], float fov_deg, float aspect)     // Camera::new  camera.rs:17-31
{
    const f3 eye{eye_[0], eye_[1], eye_[2]}, target{target_[0], target_[1], target_[2]};
    // Affine3A::look_at_rh(eye, center, Y) == look_to_lh(eye, -(center - eye), Y)
    const f3 up{0.0f, 1.0f, 0.0f};
    const f3 fwd = unit3_recip(-(target - eye));
    const f3 side = unit3_recip(cross3(up, fwd));
    const f3 upv = cross3(fwd, side);
    const xf34 view{m33{f3{side.x, upv.x, fwd.x}, f3{side.y, upv.y, fwd.y}, f3{side.z, upv.z, fwd.z}},
                    f3{-dot3(side, eye), -dot3(upv, eye), -dot3(fwd, eye)}};
    camera.matrix = inverse_affine(view);
    // Mat4::perspective_infinite_rh(fov.to_radians(), aspect, 1.0)
    const float fov = fov_deg * (3.14159265358979323846f / 180.0f);
    const float fl = 1.0f / tan_det(0.5f * fov);
    const float proj[16] = {fl / aspect, 0, 0, 0, 0, fl, 0, 0, 0, 0, -1.0f, -1.0f, 0, 0, -1.0f, 0};
    mat4_inverse(proj, camera.inv_proj);
    refresh_ray_matrix();
    // (pitch, yaw, _) = matrix.to_scale_rotation_translation().1.to_euler(EulerRot::YXZ)   camera.rs:23
    const m33& r = camera.matrix.m;
    const float det = dot3(r.c2, cross3(r.c0, r.c1));                                   // Mat3A::determinant
    const float sgn = det != det ? det : (std::signbit(det) ? -1.0f : 1.0f);            // f32::signum
    const float sx = 1.0f / (std::sqrt(dot3(r.c0, r.c0)) * sgn), sy = 1.0f / std::sqrt(dot3(r.c1, r.c1)), sz = 1.0f / std::sqrt(dot3(r.c2, r.c2));
    const quat q = quat_from_axes(r.c0 * sx, r.c1 * sy, r.c2 * sz);
    euler_yxz_of(q, &camera.pitch, &camera.yaw);
    camera.set = true;
}

void HostScene::refresh_ray_matrix()
{
    const xf34& m = camera.matrix;
    const float m4[16] = {m.m.c0.x, m.m.c0.y, m.m.c0.z, 0, m.m.c1.x, m.m.c1.y, m.m.c1.z, 0, m.m.c2.x, m.m.c2.y, m.m.c2.z, 0, m.t.x, m.t.y, m.t.z, 1.0f};
    mat4_mul(m4, camera.inv_proj, camera.ray_matrix);
}

void HostScene::camera_move(float dx, float dz, float dt)                                // Camera::update_origin  camera.rs:33-39
{
    const float sensitivity = 5.0e5f;
    const f3 step = mul(camera.matrix.m, f3{dx, 0.0f, -dz});                             // transform_vector3a
    camera.matrix.t = camera.matrix.t + (step * dt) * sensitivity;
    refresh_ray_matrix();
}

void HostScene::camera_rotate(float dx, float dy, float dt)                              // Camera::update_rotation  camera.rs:41-54
{
    const float sensitivity = 1.0e4f;
    camera.yaw -= (dy * dt) * sensitivity;
    camera.pitch -= (dx * dt) * sensitivity;
    // Quat::from_euler(EulerRot::YXZ, pitch, yaw, 0.0) = (rot_y(pitch) * rot_x(yaw)) * rot_z(0)
    float s, c;
    sincos_det(camera.pitch * 0.5f, &s, &c);
    const quat qy{0.0f, s, 0.0f, c};
    sincos_det(camera.yaw * 0.5f, &s, &c);
    const quat qx{s, 0.0f, 0.0f, c};
    sincos_det(0.0f * 0.5f, &s, &c);
    const quat qz{0.0f, 0.0f, s, c};
    camera.matrix.m = mat3_of_quat(quat_mul(quat_mul(qy, qx), qz));                      // Affine3A::from_rotation_translation
    refresh_ray_matrix();
}

void HostScene::inv_projection(float out16[16]) const { mat4_inverse(camera.ray_matrix, out16); }

void HostScene::create_ray(float s, float t, float o[3], float d[3]) const                // Camera::create_ray  camera.rs:94-105
{
    const float* M = camera.ray_matrix;
    const float nx = s * 2.0f - 1.0f, ny = t * 2.0f - 1.0f, nz = 0.0f;
    float r[4];
    for (int i = 0; i < 4; ++i) // Mat4::project_point3
    {
        float v = M[i] * nx;
        v = M[4 + i] * ny + v;
        v = M[8 + i] * nz + v;
        v = M[12 + i] + v;
        r[i] = v;
    }
    const float rw = 1.0f / r[3];
    const f3 point{r[0] * rw, r[1] * rw, r[2] * rw};
    const f3 dir = unit3(point - camera.matrix.t);
    o[0] = camera.matrix.t.x; o[1] = camera.matrix.t.y; o[2] = camera.matrix.t.z;
    d[0] = dir.x; d[1] = dir.y; d[2] = dir.z;
}

} // namespace pt
