// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see pto_math.h).
//
// CPU restatement of the reference's per-pixel integration loop
// (CouncilmanJeremyJamm/path_tracer @ /root/reference, Rust).  Every function
// cites the reference file:line it follows.  Control flow, stack discipline,
// RNG draw order and floating-point operation order mirror the source text.
//
// Documented deviations:
//   B-1  Sobol index = true sample index (reference passes c.w as u32 = 0/1).
//   B-8  LightSampler::sample index clamped to len-1 (reference would panic).
//   B-13 volume stack = insertion-ordered small vector keyed by material index
//        (reference: pointer-keyed hash set with address-dependent order).
//   RNG  thread-local entropy-seeded WyRand -> counter-based WyRand stream per
//        (seed, pixel, sample); libm sin_cos/tan/exp/ln -> det_* of pto_math.h.
#include "pt_oracle.h"
#include "pto_math.h"

#include <algorithm>
#include <atomic>
#include <fstream>
#include <sstream>
#include <string>
#include <memory>
#include <thread>
#include <vector>

using namespace pto;

namespace {

const float EPSILON = 5e-04f;          // utility.rs:4
const float INF = INFINITY;            // utility.rs:5
const float PI_F = 3.14159265358979323846f;
const float TAU_F = 6.28318530717958647692f;
const float FRAC_1_PI = 0.318309886183790671537767526745028724f;

struct Counters
{
    uint64_t c[PTO_N_COUNTERS] = {0, 0, 0, 0, 0, 0, 0, 0};
    int mode = -1; // which counter class the current traversal feeds (0 world closest, 1 any, 2 lights closest)
    void node() { c[3]++; if (mode == 0) c[6]++; }
    void tri() { c[4]++; if (mode == 0) c[7]++; }
};

// ------------------------------------------------------------------ ray.rs
struct Ray
{
    V3 o, d, inv;
    static Ray make(V3 o, V3 d) { return Ray{o, d, recip(d)}; }                 // ray.rs:10-18
    V3 at(float t) const { return vfma(d, splat(t), o); }                        // ray.rs:20 (mul_add)
    Ray transform(const Affine& inv_m) const                                     // ray.rs:22-28
    {
        return make(transform_point(inv_m, o), transform_vector(inv_m, d));
    }
};

// ------------------------------------------------------------------ boundingbox.rs
struct AABB
{
    V3 mn, mx;
    static AABB identity() { return AABB{splat(INF), splat(-INF)}; }             // boundingbox.rs:59-65
    AABB transform(const Affine& m) const                                        // boundingbox.rs:51-57 (corner-only)
    {
        V3 a = transform_point(m, mn), b = transform_point(m, mx);
        return AABB{vmin(a, b), vmax(a, b)};
    }
    float surface_area() const                                                   // boundingbox.rs:90-95
    {
        V3 v = mx - mn;
        return 2.0f * dot(v, V3{v.z, v.x, v.y});
    }
    uint8_t longest_axis() const                                                 // boundingbox.rs:71-88
    {
        V3 l = mx - mn;
        float m = max_element(l);
        if (l.x == m) return 0;
        if (l.y == m) return 1;
        return 2;
    }
    bool intersect(const Ray& r, float t_max) const                              // boundingbox.rs:97-113
    {
        V3 t0 = (mn - r.o) * r.inv, t1 = (mx - r.o) * r.inv;
        V3 tmin_v = splat(EPSILON), tmax_v = splat(t_max);
        V3 t_smaller = vmin(vmax(t0, tmin_v), vmax(t1, tmin_v));
        V3 t_bigger = vmax(vmin(t0, tmax_v), vmin(t1, tmax_v));
        return max_element(t_smaller) <= min_element(t_bigger);
    }
    bool intersect_t(const Ray& r, float t_max, float* t_out) const              // boundingbox.rs:115-131
    {
        V3 t0 = (mn - r.o) * r.inv, t1 = (mx - r.o) * r.inv;
        V3 tmin_v = splat(EPSILON), tmax_v = splat(t_max);
        V3 t_smaller = vmin(vmax(t0, tmin_v), vmax(t1, tmin_v));
        V3 t_bigger = vmax(vmin(t0, tmax_v), vmin(t1, tmax_v));
        float t = max_element(t_smaller);
        *t_out = t;
        return t <= min_element(t_bigger);
    }
};
AABB surrounding_box(const AABB& a, const AABB& b) { return AABB{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; } // boundingbox.rs:134-139

// ------------------------------------------------------------------ model.rs:5-11
struct HitInfo
{
    V3 normal;
    float u, v, t;
    bool front_facing;
};

// ------------------------------------------------------------------ primitive.rs
struct Triangle
{
    M3 positions, normals;
    V4 n0, n1, n2;

    static Triangle make(const V3 p[3], const V3 n[3])                           // primitive.rs:31-54
    {
        V3 ab = p[1] - p[0], ac = p[2] - p[0];
        V3 n0 = cross(ab, ac);
        float d0 = dot(n0, p[0]);
        float scale = length_squared(n0);
        V3 n1 = cross(ac, n0) / scale;
        float d1 = -dot(n1, p[0]);
        V3 n2 = cross(n0, ab) / scale;
        float d2 = -dot(n2, p[0]);
        Triangle t;
        t.positions = M3{p[0], p[1], p[2]};
        t.normals = M3{n[0], n[1], n[2]};
        t.n0 = V4{n0.x, n0.y, n0.z, d0};
        t.n1 = V4{n1.x, n1.y, n1.z, d1};
        t.n2 = V4{n2.x, n2.y, n2.z, d2};
        return t;
    }
    V3 get_normal(float u, float v) const                                        // primitive.rs:57-63
    {
        float w = 1.0f - u - v;
        return normalize(mul(normals, V3{w, u, v}));
    }
    V3 get_position(float u, float v) const                                      // primitive.rs:66-70
    {
        float w = 1.0f - u - v;
        return mul(positions, V3{w, u, v});
    }
    void random_point(Rng& rng, V3* p, V3* n) const                              // primitive.rs:77-91
    {
        float u = rng.next_f32();
        float v = rng.next_f32();
        if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
        *p = get_position(u, v);
        *n = get_normal(u, v);
    }
    float area() const { return 0.5f * length(V3{n0.x, n0.y, n0.z}); }           // primitive.rs:94
    AABB create_bounding_box() const                                             // primitive.rs:97-103
    {
        V3 mn = vmin(vmin(positions.c0, positions.c1), positions.c2);
        V3 mx = vmax(vmax(positions.c0, positions.c1), positions.c2);
        return AABB{mn, mx};
    }
    // Havel-Herout, primitive.rs:117-144.  Returns (t*det, u*det, v*det, det).
    bool intersect_naive(V3 origin, V3 direction, float t_min, float t_max, V4* out) const
    {
        float det = dot(direction, V3{n0.x, n0.y, n0.z});
        float td = -dot4(V4{origin.x, origin.y, origin.z, -1.0f}, n0);
        if (signum_differs(td - det * t_min, det * t_max - td)) return false;
        V3 p3 = det * origin + td * direction;
        V4 p{p3.x, p3.y, p3.z, det};
        float ud = dot4(p, n1);
        if (signum_differs(ud, det - ud)) return false;
        float vd = dot4(p, n2);
        if (signum_differs(vd, det - ud - vd)) return false;
        *out = V4{td, ud, vd, det};
        return true;
    }
    bool intersect(const Ray& ray, float t_max, float t_estimate, HitInfo* hi) const // primitive.rs:147-178
    {
        V3 moved = ray.at(t_estimate);
        V4 tuvd;
        if (!intersect_naive(moved, ray.d, EPSILON - t_estimate, t_max - t_estimate, &tuvd)) return false;
        V3 q = V3{tuvd.x, tuvd.y, tuvd.z} / tuvd.w;
        V3 n = get_normal(q.y, q.z);
        bool face_forward = dot(ray.d, n) < 0.0f;
        hi->normal = face_forward ? n : -n;
        hi->u = q.y;
        hi->v = q.z;
        hi->t = q.x + t_estimate;
        hi->front_facing = face_forward;
        return true;
    }
    bool intersect_bool(const Ray& ray, float t_max, float t_estimate) const      // primitive.rs:181-189
    {
        V3 moved = ray.at(t_estimate);
        V4 tuvd;
        return intersect_naive(moved, ray.d, EPSILON - t_estimate, t_max - t_estimate, &tuvd);
    }
};

// ------------------------------------------------------------------ volume.rs
struct Volume
{
    bool has_absorption = false, has_scatter = false;
    V3 absorption{0, 0, 0}; // absorption * k                                   volume.rs:112
    float c = 0, g = 0;     // g clamped to +-0.999                             volume.rs:27
    V3 get_transmission(float dist) const                                        // volume.rs:113
    {
        V3 e = -absorption * dist;
        return V3{det_exp(e.x), det_exp(e.y), det_exp(e.z)};
    }
    V3 scatter_direction(Rng& rng, V3 incoming) const                            // volume.rs:32-60
    {
        float u0 = rng.next_f32();
        float u1 = rng.next_f32();
        float phi = 2.0f * PI_F * u0;
        float z;
        if (g == 0.0f) { z = 1.0f - 2.0f * u1; }
        else
        {
            float x = (1.0f - g * g) / (1.0f + g * (1.0f - 2.0f * u1));
            z = (1.0f + g * g - x * x) / (2.0f * g);
        }
        float sine, cosine;
        det_sincos(phi, &sine, &cosine);
        float r = std::sqrt(1.0f - z * z);
        float x = r * cosine, y = r * sine;
        return mul(generate_onb(-incoming), V3{x, y, z});
    }
    // volume.rs:83-97 (scatter_pdf result is discarded by the integrator's `(t, dir, _)` pattern)
    bool scatter(Rng& rng, V3 incoming, float t_max, float* t_out, V3* dir) const
    {
        float t = -det_ln(rng.next_f32()) / c;
        if (t > t_max) return false;
        *dir = scatter_direction(rng, incoming);
        *t_out = t;
        return true;
    }
};

// ------------------------------------------------------------------ material.rs
struct BsdfPdf { V3 bsdf; float pdf; };

V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }                      // utility.rs:21
V3 refract(V3 i, V3 n, float eta)                                                // utility.rs:23-36
{
    float n_dot_i = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - n_dot_i * n_dot_i);
    if (k <= 0.0f) return splat(NAN);
    return eta * i - (eta * n_dot_i + std::sqrt(k)) * n;
}
V3 random_cosine_vector(Rng& rng)                                                // utility.rs:7-19
{
    float r = std::sqrt(rng.next_f32());
    float z = std::sqrt(1.0f - r * r);
    float phi = TAU_F * rng.next_f32();
    float s, c;
    det_sincos(phi, &s, &c);
    return V3{c * r, s * r, z};
}
M3 generate_onb_ggx(V3 v)                                                        // onb.rs:9-27
{
    if (v.z > 0.99999f) return M3{V3{1, 0, 0}, V3{-0.0f, -1.0f, -0.0f}, V3{0, 0, 1}};
    V3 t1 = normalize(cross(v, V3{0, 0, 1}));
    V3 t2 = cross(t1, v);
    return M3{t1, t2, v};
}

struct Material
{
    int kind = PTO_LAMBERTIAN;
    V3 colour{0, 0, 0}; // albedo / emitted / colour
    float a = 0;        // GGX alpha = clamp(roughness^2, 1e-4, 0.9999)            material.rs:294,309
    float ior = 1;
    bool has_volume = false;
    Volume volume;

    bool is_delta() const { return kind == PTO_SPECULAR || kind == PTO_DIELECTRIC; }   // material.rs:151,494
    bool is_emissive() const { return kind == PTO_EMISSIVE; }                           // material.rs:131
    V3 get_emitted() const { return kind == PTO_EMISSIVE ? colour : V3{0, 0, 0}; }      // material.rs:51,135
    float get_weakening(V3 wo, V3 n) const { return is_delta() ? 1.0f : std::fabs(dot(wo, n)); } // material.rs:67-77
    const Volume* get_volume() const                                                    // material.rs:62,452-459,529
    {
        return ((kind == PTO_GGX_DIELECTRIC || kind == PTO_DIELECTRIC) && has_volume) ? &volume : nullptr;
    }

    // ---- GGX helpers, material.rs:189-284
    float ggx_d(V3 h) const
    {
        if (h.z <= 0.0f) return 0.0f;
        float cosine_sq = h.z * h.z;
        float tan_sq = std::sqrt(1.0f - cosine_sq) / cosine_sq; // as written at material.rs:197
        float x = (a * a) + tan_sq;
        return a * a / (PI_F * cosine_sq * cosine_sq * x * x);
    }
    static float ggx_f(float v_dot_h, float f0) { return mul_add(powi5(1.0f - v_dot_h), 1.0f - f0, f0); } // material.rs:205
    static V3 ggx_f_vector(float v_dot_h, V3 f0)                                                            // material.rs:207
    {
        V3 one_minus{1.0f - f0.x, 1.0f - f0.y, 1.0f - f0.z};
        return f0 + (one_minus * powi5(1.0f - v_dot_h));
    }
    float ggx_g1(V3 v, V3 h) const                                                                           // material.rs:210-221
    {
        if (v.z * dot(h, v) <= 0.0f) return 0.0f;
        float tan_squared = powi_m2(v.z) - 1.0f;
        return 2.0f / (1.0f + std::sqrt(1.0f + a * a * tan_squared));
    }
    float ggx_g(V3 wi, V3 wo, V3 h) const { return ggx_g1(wi, h) * ggx_g1(wo, h); }                           // material.rs:224
    float ggx_g_uncorrelated(V3 wi, V3 wo) const                                                             // material.rs:227-244
    {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float a_squared = a * a;
        float x = 2.0f * wi.z * wo.z;
        float y = 1.0f - a_squared;
        float z = wo.z * det_hypot(a, wi.z * std::sqrt(y));
        float w = wi.z * det_hypot(a, wo.z * std::sqrt(y));
        return x / (z + w);
    }
    V3 ggx_half_vector(Rng& rng, V3 incoming, V3 normal) const                                               // material.rs:248-284
    {
        M3 onb_a = generate_onb(normal);
        V3 _v = mul(transpose(onb_a), -incoming);
        V3 v = normalize(_v * V3{a, a, 1.0f});
        M3 onb_b = generate_onb_ggx(v);
        float u1 = rng.next_f32();
        float u2 = rng.next_f32();
        float _a = 1.0f / (1.0f + v.z);
        bool condition = u2 < _a;
        float r = rs_min(std::sqrt(u1), 0.9999f);
        float phi = condition ? (PI_F * u2 / _a) : (PI_F + ((u2 - _a) / (1.0f - _a)) * PI_F);
        float sn, cs;
        det_sincos(phi, &sn, &cs);
        float p1 = r * cs;
        float p2 = r * sn * (condition ? 1.0f : v.z);
        V3 _h = mul(onb_b, V3{p1, p2, std::sqrt(1.0f - p1 * p1 - p2 * p2)});
        return mul(onb_a, normalize(_h * V3{a, a, 1.0f}));
    }
    static float dielectric_f(float cosine, float eta)                                                       // material.rs:477-489
    {
        if (eta * eta * (1.0f - cosine * cosine) > 1.0f) return 1.0f;
        float f0 = powi2((eta - 1.0f) / (eta + 1.0f));
        return mul_add(powi5(1.0f - cosine), 1.0f - f0, f0);
    }

    V3 scatter_direction(Rng& rng, V3 incoming, V3 normal, bool front_facing) const
    {
        switch (kind)
        {
        case PTO_LAMBERTIAN: return mul(generate_onb(normal), random_cosine_vector(rng));   // material.rs:104-107
        case PTO_EMISSIVE: return V3{0, 0, 0};                                               // material.rs:133
        case PTO_SPECULAR: return reflect(incoming, normal);                                 // material.rs:153
        case PTO_GGX_METAL:                                                                  // material.rs:317-325
        {
            V3 h = ggx_half_vector(rng, incoming, normal);
            return reflect(incoming, h);
        }
        case PTO_GGX_DIELECTRIC:                                                             // material.rs:326-346
        {
            V3 h = ggx_half_vector(rng, incoming, normal);
            float eta = front_facing ? (1.0f / ior) : ior;
            float f0 = powi2((eta - 1.0f) / (eta + 1.0f));
            float f = ggx_f(-dot(incoming, h), f0);
            V3 refracted = refract(incoming, h, eta);
            bool ray_reflected = is_nan(refracted) || rng.next_f32() < f; // conditional draw, material.rs:335
            return ray_reflected ? reflect(incoming, h) : refracted;
        }
        case PTO_DIELECTRIC:                                                                 // material.rs:496-509
        {
            float eta = front_facing ? (1.0f / ior) : ior;
            float cosine = -dot(incoming, normal);
            if (rng.next_f32() < dielectric_f(cosine, eta)) return reflect(incoming, normal);
            return refract(incoming, normal, eta);
        }
        }
        return V3{0, 0, 0};
    }

    BsdfPdf get_bsdf_pdf(V3 incoming, V3 outgoing, const HitInfo& hi) const
    {
        switch (kind)
        {
        case PTO_LAMBERTIAN:                                                                 // material.rs:109-115
        {
            float cosine = dot(outgoing, hi.normal);
            return BsdfPdf{colour * FRAC_1_PI, cosine * FRAC_1_PI};
        }
        case PTO_EMISSIVE: return BsdfPdf{colour, 1.0f};                                     // material.rs:134
        case PTO_SPECULAR: return BsdfPdf{colour, 1.0f};                                     // material.rs:155
        case PTO_DIELECTRIC:                                                                 // material.rs:511-527
        {
            float cosine = -dot(incoming, outgoing);
            float eta = hi.front_facing ? (1.0f / ior) : ior;
            float f = dielectric_f(cosine, eta);
            if (dot(outgoing, hi.normal) > 0.0f) return BsdfPdf{splat(f), f};
            float bsdf = (1.0f - f) / (eta * eta);
            return BsdfPdf{colour * bsdf, 1.0f - f};
        }
        case PTO_GGX_METAL:
        case PTO_GGX_DIELECTRIC:                                                             // material.rs:349-450
        {
            const bool transmissive = kind == PTO_GGX_DIELECTRIC;
            M3 onb_inv = transpose(generate_onb(hi.normal));
            V3 wi = mul(onb_inv, outgoing);
            V3 wo = mul(onb_inv, incoming);
            bool ray_transmitted = wi.z < 0.0f;
            V3 h;
            if (transmissive && ray_transmitted)
            {
                float eta = hi.front_facing ? ior : (1.0f / ior);
                V3 _h = normalize(eta * wi + wo);
                h = _h * rs_signum(_h.z);
            }
            else { h = normalize(wi + wo); }
            float i_dot_h = dot(wi, h), o_dot_h = dot(wo, h);
            float d = ggx_d(h);
            float f, g;
            if (!transmissive) { f = 1.0f; g = ggx_g_uncorrelated(wi, wo); }
            else
            {
                float eta = hi.front_facing ? ior : (1.0f / ior);
                float f0 = powi2((eta - 1.0f) / (eta + 1.0f));
                f = ggx_f(std::fabs(i_dot_h), f0);
                g = ggx_g(wi, wo, h);
            }
            if (ray_transmitted)
            {
                if (!transmissive) return BsdfPdf{V3{0, 0, 0}, 0.0f};                        // BsdfPdf::invalid()
                float eta = hi.front_facing ? ior : (1.0f / ior);
                float x = std::fabs(i_dot_h * o_dot_h);
                float y = std::fabs(wi.z * wo.z);
                float z = (1.0f - f) * g * d;
                float w = (eta * i_dot_h) + o_dot_h;
                float btdf = (x * z) / (y * w * w);
                float ja = std::fabs(o_dot_h), jb = w;
                float jacobian = ja / (jb * jb);
                float pdf = d * (1.0f - f) * std::fabs(h.z) * jacobian;
                return BsdfPdf{colour * btdf * eta * eta, pdf};
            }
            float brdf = f * g * d / (4.0f * std::fabs(wi.z * wo.z));
            float jacobian = 1.0f / (4.0f * std::fabs(o_dot_h));
            float pdf = d * h.z * f * jacobian;
            V3 tint = transmissive ? V3{1, 1, 1} : ggx_f_vector(std::fabs(i_dot_h), colour);
            return BsdfPdf{brdf * tint, pdf};
        }
        }
        return BsdfPdf{V3{0, 0, 0}, 0.0f};
    }
};

// ------------------------------------------------------------------ blas_bvh.rs
const size_t DESIRED_BINS = 64;       // blas_bvh.rs:13
const float TRAVERSAL_COST = 1.0f;    // blas_bvh.rs:15
const float INTERSECTION_COST = 2.0f; // blas_bvh.rs:16

struct PrimitiveInfo { AABB box; uint32_t id; };
AABB boxes_union(const PrimitiveInfo* p, size_t n)                                // blas_bvh.rs:28-33
{
    AABB a = AABB::identity();
    for (size_t i = 0; i < n; ++i) a = surrounding_box(a, p[i].box);
    return a;
}
enum { NODE_BRANCH = 0, NODE_LEAF_SINGLE = 1, NODE_LEAF = 2 };
struct BLASNode
{
    AABB box;
    int type;
    uint32_t left = 0, right = 0, prim = 0;
    std::vector<uint32_t> prims;
};

uint32_t generate_blas(std::vector<BLASNode>& arena, PrimitiveInfo* info, size_t span, uint8_t last_split_axis) // blas_bvh.rs:62-136
{
    if (span == 1)
    {
        BLASNode n;
        n.box = info[0].box;
        n.type = NODE_LEAF_SINGLE;
        n.prim = info[0].id;
        arena.push_back(n);
        return (uint32_t)arena.size() - 1;
    }
    AABB bb = boxes_union(info, span);
    float bb_sa = bb.surface_area();
    uint8_t split_axis = bb.longest_axis();
    if (split_axis != last_split_axis)
    {
        // glidesort::sort_by = stable sort; comparator = total_cmp on minimum[axis] (boundingbox.rs:67)
        std::stable_sort(info, info + span, [split_axis](const PrimitiveInfo& a, const PrimitiveInfo& b) {
            const float* pa = &a.box.mn.x;
            const float* pb = &b.box.mn.x;
            return total_key(pa[split_axis]) < total_key(pb[split_axis]);
        });
    }
    size_t bin_size = std::max<size_t>(span / DESIRED_BINS, 1);
    size_t num_bins = (span / bin_size) - 1;
    size_t best_split = 0;
    float best_sah = 0;
    bool have = false;
    for (size_t i = 0; i < num_bins; ++i)
    {
        size_t j = (i + 1) * bin_size;
        AABB l = boxes_union(info, j), r = boxes_union(info + j, span - j);
        float sah = TRAVERSAL_COST + (((float)j) * l.surface_area() + ((float)(span - j)) * r.surface_area()) * INTERSECTION_COST / bb_sa;
        // Iterator::min_by keeps the FIRST of equal minima; ordering is f32::total_cmp
        if (!have || total_key(sah) < total_key(best_sah)) { best_sah = sah; best_split = j; have = true; }
    }
    float no_split_sah = INTERSECTION_COST * (float)span;
    if (no_split_sah < best_sah)
    {
        BLASNode n;
        n.box = bb;
        n.type = NODE_LEAF;
        for (size_t i = 0; i < span; ++i) n.prims.push_back(info[i].id);
        arena.push_back(n);
        return (uint32_t)arena.size() - 1;
    }
    uint32_t left = generate_blas(arena, info, best_split, split_axis);
    uint32_t right = generate_blas(arena, info + best_split, span - best_split, split_axis);
    BLASNode n;
    n.box = bb;
    n.type = NODE_BRANCH;
    n.left = left;
    n.right = right;
    arena.push_back(n);
    return (uint32_t)arena.size() - 1;
}

// ------------------------------------------------------------------ blas.rs
struct StackEntry { uint32_t id; float t; };
struct Scratch
{
    std::vector<StackEntry> tlas_stack, blas_stack;
    std::vector<uint32_t> tlas_ids, blas_ids;
    Counters* ctr = nullptr;
};

// blas.rs:133-162: test both children, push far first; on a tie (t_l >= t_r) left is pushed first.
template <class Node>
void push_to_stack(const Ray& r, float t_max, std::vector<StackEntry>& stack, const std::vector<Node>& arena, uint32_t left,
                   uint32_t right, Counters* ctr)
{
    float tl, tr;
    bool il = arena[left].box.intersect_t(r, t_max, &tl);
    bool ir = arena[right].box.intersect_t(r, t_max, &tr);
    ctr->node();
    ctr->node();
    if (il && ir)
    {
        if (tl < tr) { stack.push_back({right, tr}); stack.push_back({left, tl}); }
        else { stack.push_back({left, tl}); stack.push_back({right, tr}); }
    }
    else if (il) { stack.push_back({left, tl}); }
    else if (ir) { stack.push_back({right, tr}); }
}

struct BLAS
{
    std::vector<Triangle> primitives;
    int material = 0;
    uint32_t root = 0;
    std::vector<BLASNode> nodes;

    void build(const float* positions, const float* normals, uint32_t n_tris, int mat)   // blas.rs:174-201
    {
        material = mat;
        primitives.reserve(n_tris);
        for (uint32_t i = 0; i < n_tris; ++i)
        {
            V3 p[3], n[3];
            for (int k = 0; k < 3; ++k)
            {
                p[k] = V3{positions[(i * 3 + k) * 3 + 0], positions[(i * 3 + k) * 3 + 1], positions[(i * 3 + k) * 3 + 2]};
                n[k] = V3{normals[(i * 3 + k) * 3 + 0], normals[(i * 3 + k) * 3 + 1], normals[(i * 3 + k) * 3 + 2]};
            }
            primitives.push_back(Triangle::make(p, n));
        }
        std::vector<PrimitiveInfo> info(n_tris);
        for (uint32_t i = 0; i < n_tris; ++i) info[i] = PrimitiveInfo{primitives[i].create_bounding_box(), i};
        root = generate_blas(nodes, info.data(), info.size(), 4);
    }

    bool intersect(Scratch& sc, const Ray& r, float t_max, HitInfo* out, uint32_t* prim_out) const // blas.rs:214-256
    {
        auto& stack = sc.blas_stack;
        stack.clear();
        stack.push_back({root, 0.0f}); // root pushed without a box test
        bool found = false;
        while (!stack.empty())
        {
            StackEntry e = stack.back();
            stack.pop_back();
            if (e.t > t_max) continue;
            const BLASNode& n = nodes[e.id];
            if (n.type == NODE_BRANCH) { push_to_stack(r, t_max, stack, nodes, n.left, n.right, sc.ctr); }
            else if (n.type == NODE_LEAF_SINGLE)
            {
                HitInfo hi;
                sc.ctr->tri();
                if (primitives[n.prim].intersect(r, t_max, e.t, &hi)) { t_max = hi.t; *out = hi; *prim_out = n.prim; found = true; }
            }
            else
            {
                for (uint32_t id : n.prims)
                {
                    HitInfo hi;
                    sc.ctr->tri();
                    if (primitives[id].intersect(r, t_max, e.t, &hi)) { t_max = hi.t; *out = hi; *prim_out = id; found = true; }
                }
            }
        }
        return found;
    }
    bool any_intersect(Scratch& sc, const Ray& r, float t_max) const                      // blas.rs:257-294
    {
        auto& stack = sc.blas_ids;
        stack.clear();
        stack.push_back(root);
        while (!stack.empty())
        {
            uint32_t cur = stack.back();
            stack.pop_back();
            float t_enter;
            sc.ctr->node();
            if (!nodes[cur].box.intersect_t(r, t_max, &t_enter)) continue;
            const BLASNode& n = nodes[cur];
            if (n.type == NODE_BRANCH) { stack.push_back(n.left); stack.push_back(n.right); }
            else if (n.type == NODE_LEAF_SINGLE)
            {
                sc.ctr->tri();
                if (primitives[n.prim].intersect_bool(r, t_max, t_enter)) return true;
            }
            else
            {
                for (uint32_t id : n.prims)
                {
                    sc.ctr->tri();
                    if (primitives[id].intersect_bool(r, t_max, t_enter)) return true;
                }
            }
        }
        return false;
    }
};

// ------------------------------------------------------------------ tlas_bvh.rs / tlas.rs
struct TLASNode
{
    AABB box;
    bool leaf = false;
    uint32_t blas = 0, instance = 0;
    Affine matrix, inv_matrix;
    uint32_t left = 0, right = 0;
};

struct ModelDesc
{
    std::vector<float> positions, normals;
    uint32_t n_tris;
    int material;
    std::vector<Affine> matrices;
};

struct TLAS
{
    std::vector<BLAS> blas;
    std::vector<TLASNode> nodes;
    uint32_t root = 0;
    bool empty = true;

    static size_t find_best_match(const std::vector<TLASNode>& arena, const std::vector<uint32_t>& list, size_t a_index) // tlas_bvh.rs:56-83
    {
        const AABB& a = arena[list[a_index]].box;
        float best_sa = INF;
        size_t best_index = SIZE_MAX;
        for (size_t i = 0; i < list.size(); ++i)
        {
            if (i == a_index) continue;
            float sa = surrounding_box(a, arena[list[i]].box).surface_area();
            if (sa < best_sa) { best_index = i; best_sa = sa; }
        }
        return best_index;
    }
    static uint32_t swap_remove(std::vector<uint32_t>& v, size_t i)
    {
        uint32_t x = v[i];
        v[i] = v.back();
        v.pop_back();
        return x;
    }
    void build(const std::vector<const ModelDesc*>& models)                               // tlas.rs:24-53 + tlas_bvh.rs:85-138
    {
        blas.resize(models.size());
        for (size_t i = 0; i < models.size(); ++i)
            blas[i].build(models[i]->positions.data(), models[i]->normals.data(), models[i]->n_tris, models[i]->material);
        std::vector<uint32_t> list;
        for (size_t i = 0; i < models.size(); ++i)
        {
            const AABB bb = blas[i].nodes[blas[i].root].box;
            for (const Affine& m : models[i]->matrices)
            {
                TLASNode n;
                n.box = bb.transform(m);
                n.leaf = true;
                n.blas = (uint32_t)i;
                n.instance = (uint32_t)nodes.size();
                n.matrix = m;
                n.inv_matrix = inverse(m);
                nodes.push_back(n);
                list.push_back((uint32_t)nodes.size() - 1);
            }
        }
        empty = list.empty();
        if (empty) return;
        size_t a = 0;
        size_t b = list.size() > 1 ? find_best_match(nodes, list, a) : SIZE_MAX;
        while (list.size() > 1)
        {
            size_t c = find_best_match(nodes, list, b);
            if (a == c)
            {
                uint32_t n1, n2;
                if (a > b) { n1 = swap_remove(list, a); n2 = swap_remove(list, b); }
                else { n1 = swap_remove(list, b); n2 = swap_remove(list, a); }
                a = list.size();
                TLASNode n;
                n.box = surrounding_box(nodes[n1].box, nodes[n2].box);
                n.leaf = false;
                n.left = n1;
                n.right = n2;
                nodes.push_back(n);
                list.push_back((uint32_t)nodes.size() - 1);
                b = list.size() > 1 ? find_best_match(nodes, list, a) : SIZE_MAX;
            }
            else { a = b; b = c; }
        }
        root = list[0];
    }

    // tlas.rs:66-110
    bool intersect(Scratch& sc, const Ray& r, float t_max, HitInfo* out, uint32_t* blas_out, uint32_t* prim_out, uint32_t* inst_out) const
    {
        if (empty) return false;
        sc.ctr->node();
        if (!nodes[root].box.intersect(r, t_max)) return false;
        auto& stack = sc.tlas_stack;
        stack.clear();
        stack.push_back({root, 0.0f});
        bool found = false;
        HitInfo best{};
        uint32_t best_node = 0, best_prim = 0;
        while (!stack.empty())
        {
            StackEntry e = stack.back();
            stack.pop_back();
            if (e.t > t_max) continue;
            const TLASNode& n = nodes[e.id];
            if (!n.leaf) { push_to_stack(r, t_max, stack, nodes, n.left, n.right, sc.ctr); }
            else
            {
                Ray ray = r.transform(n.inv_matrix);
                HitInfo hi;
                uint32_t prim;
                if (blas[n.blas].intersect(sc, ray, t_max, &hi, &prim))
                {
                    t_max = hi.t;
                    best = hi;
                    best_node = e.id;
                    best_prim = prim;
                    found = true;
                }
            }
        }
        if (!found) return false;
        best.normal = transform_vector(nodes[best_node].matrix, best.normal); // deferred normal transform, tlas.rs:105
        *out = best;
        *blas_out = nodes[best_node].blas;
        *prim_out = best_prim;
        *inst_out = nodes[best_node].instance;
        return true;
    }
    bool any_intersect(Scratch& sc, const Ray& r, float t_max) const                      // tlas.rs:111-144
    {
        if (empty) return false;
        auto& stack = sc.tlas_ids;
        stack.clear();
        stack.push_back(root);
        while (!stack.empty())
        {
            uint32_t cur = stack.back();
            stack.pop_back();
            sc.ctr->node();
            if (!nodes[cur].box.intersect(r, t_max)) continue;
            const TLASNode& n = nodes[cur];
            if (!n.leaf) { stack.push_back(n.left); stack.push_back(n.right); }
            else
            {
                Ray ray = r.transform(n.inv_matrix);
                if (blas[n.blas].any_intersect(sc, ray, t_max)) return true;
            }
        }
        return false;
    }
};

// ------------------------------------------------------------------ light_sampler.rs / scene.rs
struct LightItem { uint32_t blas, prim; float pdf; };
struct LightSampler
{
    std::vector<LightItem> lights;
    std::vector<float> cdf;
    float max = 0;
    void build(const TLAS& lights_tlas, const std::vector<Material>& mats)                // tlas.rs:55-64, blas.rs:203-212, light_sampler.rs:41-61
    {
        std::vector<LightItem> data;
        for (size_t b = 0; b < lights_tlas.blas.size(); ++b)
            for (size_t p = 0; p < lights_tlas.blas[b].primitives.size(); ++p)
            {
                float w = lights_tlas.blas[b].primitives[p].area() * length(mats[lights_tlas.blas[b].material].get_emitted());
                data.push_back({(uint32_t)b, (uint32_t)p, w});
            }
        float m = 0.0f;
        for (auto& d : data) m = m + d.pdf;
        max = m;
        float state = 0.0f;
        for (auto& d : data)
        {
            lights.push_back({d.blas, d.prim, d.pdf / max});
            state += lights.back().pdf;
            cdf.push_back(state);
        }
    }
    // light_sampler.rs:31-37: binary_search_by(total_cmp) -> Ok(i) | Err(i); for a strictly increasing cdf both equal the number
    // of entries below x.  Clamped to len-1 (deviation B-8).
    const LightItem& sample(Rng& rng) const
    {
        float x = rng.next_f32();
        size_t idx = 0;
        while (idx < cdf.size() && total_key(cdf[idx]) < total_key(x)) ++idx;
        if (idx >= lights.size()) idx = lights.size() - 1;
        return lights[idx];
    }
    float get_sample_pdf(const Triangle& tri, const Material& m) const { return tri.area() * length(m.get_emitted()) / max; } // light_sampler.rs:39
};

// ------------------------------------------------------------------ sampling.rs
const uint32_t DIRECTIONS[32] = { // sampling.rs:4-8
    0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000, 0x80800000, 0xc0c00000, 0xa0a00000,
    0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000, 0x80008000, 0xc000c000, 0xa000a000, 0xf000f000, 0x88008800, 0xcc00cc00,
    0xaa00aa00, 0xff00ff00, 0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc, 0xaaaaaaaa, 0xffffffff,
};
uint32_t reverse_bits(uint32_t x)
{
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}
uint32_t sobol_dim1(uint32_t index)                                               // sampling.rs:24-30
{
    uint32_t x = 0;
    for (int bit = 0; bit < 32; ++bit) x ^= ((index >> bit) & 1u) * DIRECTIONS[bit];
    return x;
}
uint32_t lk_hash(uint32_t x, uint32_t seed)                                       // sampling.rs:53-68
{
    x ^= x * 0x3d20adeau;
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
uint32_t scramble_base2(uint32_t x, uint32_t seed) { return reverse_bits(lk_hash(reverse_bits(x), seed)); } // sampling.rs:71
uint32_t low_bias_hash(uint32_t x)                                                // sampling.rs:76-91
{
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0xd35a2d97u;
    x ^= x >> 15;
    return x;
}
// sampling.rs:97-114; table entry i = (reverse_bits(i), sobol(i)) (sampling.rs:38-45) is recomputed instead of stored
void ss_sobol_raw(uint32_t n_points, uint32_t index, uint32_t seed, uint32_t out[3])
{
    uint32_t x_seed = low_bias_hash(seed);
    uint32_t y_seed = low_bias_hash(seed + 1u);
    uint32_t shuffle_seed = low_bias_hash(seed + 2u);
    uint32_t shuffled_index = scramble_base2(index, shuffle_seed);
    uint32_t slot = shuffled_index % n_points;
    out[0] = shuffled_index;
    out[1] = scramble_base2(reverse_bits(slot), x_seed);
    out[2] = scramble_base2(sobol_dim1(slot), y_seed);
}
void ss_sobol(uint32_t n_points, uint32_t index, uint32_t seed, float out[2])
{
    uint32_t raw[3];
    ss_sobol_raw(n_points, index, seed, raw);
    out[0] = (float)raw[1] / 4294967296.0f; // u32::MAX as f32 == 2^32
    out[1] = (float)raw[2] / 4294967296.0f;
}

// ------------------------------------------------------------------ camera.rs
struct M4 { V4 c[4]; };
V4 mul4(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
V4 add4(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
V4 mulv(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
V4 subv(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
V4 mat4_mul_vec(const M4& m, V4 v) // glam Mat4::mul_vec4
{
    V4 res = mul4(m.c[0], v.x);
    res = add4(res, mul4(m.c[1], v.y));
    res = add4(res, mul4(m.c[2], v.z));
    res = add4(res, mul4(m.c[3], v.w));
    return res;
}
M4 mat4_mul(const M4& a, const M4& b) { return M4{{mat4_mul_vec(a, b.c[0]), mat4_mul_vec(a, b.c[1]), mat4_mul_vec(a, b.c[2]), mat4_mul_vec(a, b.c[3])}}; }
M4 mat4_inverse(const M4& s) // glam Mat4::inverse (scalar form of the cofactor expansion)
{
    float m00 = s.c[0].x, m01 = s.c[0].y, m02 = s.c[0].z, m03 = s.c[0].w;
    float m10 = s.c[1].x, m11 = s.c[1].y, m12 = s.c[1].z, m13 = s.c[1].w;
    float m20 = s.c[2].x, m21 = s.c[2].y, m22 = s.c[2].z, m23 = s.c[2].w;
    float m30 = s.c[3].x, m31 = s.c[3].y, m32 = s.c[3].z, m33 = s.c[3].w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    V4 fac0{coef00, coef00, coef02, coef03}, fac1{coef04, coef04, coef06, coef07}, fac2{coef08, coef08, coef10, coef11};
    V4 fac3{coef12, coef12, coef14, coef15}, fac4{coef16, coef16, coef18, coef19}, fac5{coef20, coef20, coef22, coef23};
    V4 vec0{m10, m00, m00, m00}, vec1{m11, m01, m01, m01}, vec2{m12, m02, m02, m02}, vec3{m13, m03, m03, m03};
    V4 inv0 = add4(subv(mulv(vec1, fac0), mulv(vec2, fac1)), mulv(vec3, fac2));
    V4 inv1 = add4(subv(mulv(vec0, fac0), mulv(vec2, fac3)), mulv(vec3, fac4));
    V4 inv2 = add4(subv(mulv(vec0, fac1), mulv(vec1, fac3)), mulv(vec3, fac5));
    V4 inv3 = add4(subv(mulv(vec0, fac2), mulv(vec1, fac4)), mulv(vec2, fac5));
    V4 sign_a{1.0f, -1.0f, 1.0f, -1.0f}, sign_b{-1.0f, 1.0f, -1.0f, 1.0f};
    M4 inv{{mulv(inv0, sign_a), mulv(inv1, sign_b), mulv(inv2, sign_a), mulv(inv3, sign_b)}};
    V4 col0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
    V4 dot0 = mulv(s.c[0], col0);
    float dot1 = dot0.x + dot0.y + dot0.z + dot0.w;
    float rcp = 1.0f / dot1;
    return M4{{mul4(inv.c[0], rcp), mul4(inv.c[1], rcp), mul4(inv.c[2], rcp), mul4(inv.c[3], rcp)}};
}
// glam scalar Vec3::normalize = self * length_recip (camera.rs:19 converts to glam::Vec3)
V3 normalize_recip(V3 v) { float r = 1.0f / std::sqrt(dot(v, v)); return v * r; }

struct Camera
{
    Affine matrix;    // camera-to-world
    M4 inv_projection;
    M4 ray_matrix;    // matrix * inv_projection (camera.rs:98), constant per frame
    float yaw = 0, pitch = 0; // camera.rs:8-9 (named as the reference names them: pitch turns about Y, yaw about X)
    bool set = false;
    void init(V3 origin, V3 target, float fov_deg, float aspect)                  // camera.rs:17-31
    {
        // Affine3A::look_at_rh(eye, center, up) = look_to_lh(eye, eye - center, up)
        V3 up{0, 1, 0};
        V3 f = normalize_recip(-(target - origin));
        V3 s = normalize_recip(cross(up, f));
        V3 u = cross(f, s);
        Affine view{M3{V3{s.x, u.x, f.x}, V3{s.y, u.y, f.y}, V3{s.z, u.z, f.z}}, V3{-dot(s, origin), -dot(u, origin), -dot(f, origin)}};
        matrix = inverse(view);
        // Mat4::perspective_infinite_rh(fov.to_radians(), aspect, 1.0).inverse()
        float fov = fov_deg * (PI_F / 180.0f);
        float fl = 1.0f / det_tan(0.5f * fov);
        M4 proj{{V4{fl / aspect, 0, 0, 0}, V4{0, fl, 0, 0}, V4{0, 0, -1.0f, -1.0f}, V4{0, 0, -1.0f, 0}}};
        inv_projection = mat4_inverse(proj);
        M4 m4{{V4{matrix.m.c0.x, matrix.m.c0.y, matrix.m.c0.z, 0}, V4{matrix.m.c1.x, matrix.m.c1.y, matrix.m.c1.z, 0},
               V4{matrix.m.c2.x, matrix.m.c2.y, matrix.m.c2.z, 0}, V4{matrix.t.x, matrix.t.y, matrix.t.z, 1.0f}}};
        ray_matrix = mat4_mul(m4, inv_projection);
        // camera.rs:23  let (pitch, yaw, _) = matrix.to_scale_rotation_translation().1.to_euler(glam::EulerRot::YXZ);
        // glam 0.23 Affine3A::to_scale_rotation_translation: scale = (|x|*signum(det), |y|, |z|); rotation = Quat::from_mat3(axes * scale.recip())
        float det = dot(matrix.m.c2, cross(matrix.m.c0, matrix.m.c1));
        V3 scale{length(matrix.m.c0) * rs_signum(det), length(matrix.m.c1), length(matrix.m.c2)};
        V3 inv_scale{1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z};
        V4 q = quat_from_mat3(matrix.m.c0 * inv_scale.x, matrix.m.c1 * inv_scale.y, matrix.m.c2 * inv_scale.z);
        // glam euler.rs, EulerRot::YXZ on a quaternion: first, second (third is dropped)
        pitch = det_atan2(2.0f * (q.x * q.z + q.w * q.y), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z);
        yaw = det_asin(rs_clamp(-2.0f * (q.y * q.z - q.w * q.x), -1.0f, 1.0f));
        set = true;
    }
    void update_ray_matrix()
    {
        M4 m4{{V4{matrix.m.c0.x, matrix.m.c0.y, matrix.m.c0.z, 0}, V4{matrix.m.c1.x, matrix.m.c1.y, matrix.m.c1.z, 0},
               V4{matrix.m.c2.x, matrix.m.c2.y, matrix.m.c2.z, 0}, V4{matrix.t.x, matrix.t.y, matrix.t.z, 1.0f}}};
        ray_matrix = mat4_mul(m4, inv_projection);
    }
    static V4 quat_from_mat3(V3 x_axis, V3 y_axis, V3 z_axis)                      // glam Quat::from_rotation_axes
    {
        float m00 = x_axis.x, m01 = x_axis.y, m02 = x_axis.z, m10 = y_axis.x, m11 = y_axis.y, m12 = y_axis.z, m20 = z_axis.x, m21 = z_axis.y, m22 = z_axis.z;
        if (m22 <= 0.0f)
        {
            float dif10 = m11 - m00, omm22 = 1.0f - m22;
            if (dif10 <= 0.0f) { float f = omm22 - dif10, i = 0.5f / std::sqrt(f); return V4{f * i, (m01 + m10) * i, (m02 + m20) * i, (m12 - m21) * i}; }
            float f = omm22 + dif10, i = 0.5f / std::sqrt(f);
            return V4{(m01 + m10) * i, f * i, (m12 + m21) * i, (m20 - m02) * i};
        }
        float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) { float f = opm22 - sum10, i = 0.5f / std::sqrt(f); return V4{(m02 + m20) * i, (m12 + m21) * i, f * i, (m01 - m10) * i}; }
        float f = opm22 + sum10, i = 0.5f / std::sqrt(f);
        return V4{(m12 - m21) * i, (m20 - m02) * i, (m01 - m10) * i, f * i};
    }
    static V4 quat_mul(V4 lhs, V4 rhs)                                             // glam sse2 Quat::mul_quat (rtm::quat_mul)
    {
        auto lanes = [](V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; };
        V4 r_wzyx{rhs.w, rhs.z, rhs.y, rhs.x}, r_zwxy{rhs.z, rhs.w, rhs.x, rhs.y}, r_yxwz{rhs.y, rhs.x, rhs.w, rhs.z};
        V4 t0 = mul4(rhs, lhs.w);
        V4 t1 = lanes(mul4(r_wzyx, lhs.x), V4{1.0f, -1.0f, 1.0f, -1.0f});
        V4 t2 = lanes(mul4(r_zwxy, lhs.y), V4{1.0f, 1.0f, -1.0f, -1.0f});
        V4 t3 = lanes(mul4(r_yxwz, lhs.z), V4{-1.0f, 1.0f, 1.0f, -1.0f});
        return add4(add4(t0, t1), add4(t2, t3));
    }
    void update_origin(float dx, float dz, float dt)                               // camera.rs:33-39
    {
        const float sensitivity = 5.0e5f;
        matrix.t = matrix.t + transform_vector(matrix, V3{dx, 0.0f, -dz}) * dt * sensitivity;
        update_ray_matrix();
    }
    void update_rotation(float dx, float dy, float dt)                             // camera.rs:41-54
    {
        const float sensitivity = 1.0e4f;
        yaw -= dy * dt * sensitivity;
        pitch -= dx * dt * sensitivity;
        // Quat::from_euler(YXZ, a, b, c) = rot_y(a) * rot_x(b) * rot_z(c); rot_*(t) has sin(t/2) on its axis, cos(t/2) in w
        float sa, ca, sb, cb, sc, cc;
        det_sincos(pitch * 0.5f, &sa, &ca); det_sincos(yaw * 0.5f, &sb, &cb); det_sincos(0.0f * 0.5f, &sc, &cc);
        V4 q = quat_mul(quat_mul(V4{0, sa, 0, ca}, V4{sb, 0, 0, cb}), V4{0, 0, sc, cc});
        // Affine3A::from_rotation_translation -> Mat3A::from_quat
        float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        matrix.m = M3{V3{1.0f - (yy + zz), xy + wz, xz - wy}, V3{xy - wz, 1.0f - (xx + zz), yz + wx}, V3{xz + wy, yz - wx, 1.0f - (xx + yy)}};
        update_ray_matrix();
    }
    Ray create_ray(float s, float t) const                                        // camera.rs:94-105
    {
        V3 ndc{s * 2.0f - 1.0f, t * 2.0f - 1.0f, 0.0f};
        // Mat4::project_point3
        V4 res = mul4(ray_matrix.c[0], ndc.x);
        res = add4(mul4(ray_matrix.c[1], ndc.y), res);
        res = add4(mul4(ray_matrix.c[2], ndc.z), res);
        res = add4(ray_matrix.c[3], res);
        float rw = 1.0f / res.w;
        V3 point{res.x * rw, res.y * rw, res.z * rw};
        V3 dir = normalize(point - matrix.t);
        return Ray::make(matrix.t, dir);
    }
};

// ------------------------------------------------------------------ scene.rs
struct Scene
{
    TLAS world, lights;
    LightSampler light_sampler;
    std::vector<Material> materials;
    bool has_lights = false;
};

} // namespace

// image_helper.rs:13-17,61-88 (linear RGB; the gamma-2.2 decode of load_image :25-33 is the caller's business)
struct EnvMap
{
    std::vector<V3> data;
    uint32_t w = 0, h = 0;
    V3 get_pixel(uint32_t x, uint32_t y) const { return data[(size_t)(y % h) * w + (x % w)]; }     // image_helper.rs:61-68
    static uint32_t sat_u32(float f) { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xffffffffu : (uint32_t)f); } // Rust `as u32`
    V3 get_pixel_bilinear(float u, float v) const                                                  // image_helper.rs:71-88
    {
        float x = (float)w * u, y = (float)h * v;
        uint32_t x0 = sat_u32(x), y0 = sat_u32(y);
        float xf = x - std::trunc(x), yf = y - std::trunc(y);
        V3 c00 = get_pixel(x0, y0), c01 = get_pixel(x0, y0 + 1u), c10 = get_pixel(x0 + 1u, y0), c11 = get_pixel(x0 + 1u, y0 + 1u);
        return (((1.0f - xf) * (1.0f - yf)) * c00 + ((1.0f - xf) * yf) * c01 + (xf * (1.0f - yf)) * c10) + (xf * yf) * c11;
    }
};

struct pto_ctx
{
    EnvMap env;
    std::vector<Material> materials;
    std::vector<ModelDesc> models;
    std::unique_ptr<Scene> scene;
    Camera camera;
};

namespace {

// ------------------------------------------------------------------ integrator.rs
const float MIN_PDF = 0.0f; // integrator.rs:10
float mis_heuristic(float f, float g) { return powi2(f) / (powi2(f) + powi2(g)); } // integrator.rs:22, POWER = 2

struct Tracer
{
    const Scene* scene;
    Scratch sc;
    Counters ctr;
    Tracer(const Scene* s) : scene(s) { sc.ctr = &ctr; }
    bool world_intersect(const Ray& r, float t_max, HitInfo* hi, uint32_t* blas, uint32_t* prim, uint32_t* inst)
    {
        ctr.c[0]++;
        ctr.mode = 0;
        return scene->world.intersect(sc, r, t_max, hi, blas, prim, inst);
    }
    bool world_any(const Ray& r, float t_max)
    {
        ctr.c[1]++;
        ctr.mode = 1;
        return scene->world.any_intersect(sc, r, t_max);
    }
    bool lights_intersect(const Ray& r, float t_max, HitInfo* hi, uint32_t* blas, uint32_t* prim, uint32_t* inst)
    {
        ctr.c[2]++;
        ctr.mode = 2;
        return scene->lights.intersect(sc, r, t_max, hi, blas, prim, inst);
    }
};

V3 estimate_direct_explicit(Tracer& tr, Rng& rng, const Ray& incoming_ray, const HitInfo& hi, const Material& mat) // integrator.rs:25-74
{
    const Scene& scene = *tr.scene;
    V3 incoming = -incoming_ray.d;
    const LightItem& li = scene.light_sampler.sample(rng);                       // scene.rs:37-45
    const BLAS& lb = scene.lights.blas[li.blas];
    const Triangle& light = lb.primitives[li.prim];
    const Material& light_material = scene.materials[lb.material];
    float pdf = li.pdf;
    V3 point, light_normal;
    light.random_point(rng, &point, &light_normal);
    V3 o = incoming_ray.at(hi.t);
    V3 d = point - o;
    float distance_squared = length_squared(d);
    float distance = std::sqrt(distance_squared);
    Ray outgoing = Ray::make(o, normalize(d));
    if (dot(outgoing.d, hi.normal) > 0.0f && !tr.world_any(outgoing, (1.0f - EPSILON) * distance))
    {
        BsdfPdf bp = mat.get_bsdf_pdf(incoming, outgoing.d, hi);
        float sample_pdf = pdf / light.area();
        float cosine = std::fabs(dot(outgoing.d, light_normal));
        float light_pdf = sample_pdf * (distance_squared / cosine);
        float weight = mis_heuristic(light_pdf, bp.pdf);
        return light_material.get_emitted() * weight * mat.get_weakening(outgoing.d, hi.normal) * bp.bsdf / light_pdf;
    }
    return V3{0, 0, 0};
}

V3 estimate_direct_bsdf(Tracer& tr, Rng& rng, const Ray& incoming_ray, const HitInfo& hi, const Material& mat) // integrator.rs:77-130
{
    const Scene& scene = *tr.scene;
    V3 incoming = -incoming_ray.d;
    Ray outgoing = Ray::make(incoming_ray.at(hi.t), mat.scatter_direction(rng, incoming_ray.d, hi.normal, hi.front_facing));
    if (dot(outgoing.d, hi.normal) > 0.0f)
    {
        HitInfo lhi;
        uint32_t lblas, lprim, linst;
        if (tr.lights_intersect(outgoing, INF, &lhi, &lblas, &lprim, &linst))
        {
            if (!tr.world_any(outgoing, lhi.t * (1.0f - EPSILON)))
            {
                BsdfPdf bp = mat.get_bsdf_pdf(incoming, outgoing.d, hi);
                if (bp.pdf > MIN_PDF)
                {
                    const BLAS& lb = scene.lights.blas[lblas];
                    const Triangle& light = lb.primitives[lprim];
                    const Material& light_material = scene.materials[lb.material];
                    float sample_pdf = scene.light_sampler.get_sample_pdf(light, light_material) / light.area();
                    float cosine = std::fabs(dot(outgoing.d, lhi.normal));
                    float light_pdf = sample_pdf * (lhi.t * lhi.t / cosine);
                    float weight = mis_heuristic(bp.pdf, light_pdf);
                    return light_material.get_emitted() * weight * mat.get_weakening(outgoing.d, hi.normal) * bp.bsdf / bp.pdf;
                }
            }
        }
    }
    return V3{0, 0, 0};
}

struct Sample { V4 colour; V4 position; uint8_t id; };

// integrator.rs:143-281 (env is Err in this build: the PNG is not in the repository -> constant ambient branch :263-266)
Sample integrate(Tracer& tr, Ray r, Rng& rng, uint32_t max_bounces, bool enable_nee, const EnvMap* env)
{
    const Scene& scene = *tr.scene;
    V3 accumulated{0, 0, 0};
    V3 path_weight{1, 1, 1};
    V3 p0 = r.at(1e5f);
    V4 position{p0.x, p0.y, p0.z, 1e5f};
    uint8_t first_id = 255;
    bool last_delta = false;
    std::vector<int> volume_stack; // material indices, insertion ordered (deviation B-13)

    for (uint32_t b = 0;; ++b)
    {
        if (b > 3)                                                                // integrator.rs:166-177
        {
            float survive_prob = rs_min(max_element(path_weight), 0.9999f);
            if (rng.next_f32() > survive_prob) break;
            path_weight = path_weight / survive_prob;
        }
        HitInfo hi;
        uint32_t blas_id, prim_id, inst_id;
        if (tr.world_intersect(r, INF, &hi, &blas_id, &prim_id, &inst_id))       // integrator.rs:179
        {
            const int mat_index = scene.world.blas[blas_id].material;
            const Material& material = scene.materials[mat_index];
            if (b == 0)
            {
                V3 p = r.at(hi.t);
                position = V4{p.x, p.y, p.z, hi.t};
                first_id = (uint8_t)blas_id;
            }
            V3 wi = -r.d;

            // participating media, integrator.rs:189-205
            bool scattered = false;
            float best_t = 0;
            V3 best_dir{0, 0, 0};
            for (int vm : volume_stack)
            {
                const Volume& v = scene.materials[vm].volume;
                if (!v.has_scatter) continue;
                float t;
                V3 dir;
                if (v.scatter(rng, r.d, hi.t, &t, &dir))
                {
                    if (!scattered || total_key(t) < total_key(best_t)) { best_t = t; best_dir = dir; }
                    scattered = true;
                }
            }
            {
                float dist = scattered ? best_t : hi.t;
                V3 w{1, 1, 1};
                for (int vm : volume_stack)
                {
                    const Volume& v = scene.materials[vm].volume;
                    if (v.has_absorption) w = w * v.get_transmission(dist);
                }
                path_weight = path_weight * w;
            }
            if (scattered)
            {
                last_delta = true;
                r = Ray::make(r.at(best_t), best_dir);
                if (b == max_bounces) break;
                continue;
            }

            if (material.is_emissive())                                           // integrator.rs:207-214
            {
                if (!enable_nee || last_delta || b == 0) accumulated = vfma(material.get_emitted(), path_weight, accumulated);
                break;
            }
            if (const Volume* v = material.get_volume())                           // integrator.rs:217-227
            {
                (void)v;
                auto it = std::find(volume_stack.begin(), volume_stack.end(), mat_index);
                if (hi.front_facing) { if (it == volume_stack.end()) volume_stack.push_back(mat_index); }
                else if (it != volume_stack.end()) volume_stack.erase(it);
            }
            bool is_delta = material.is_delta();
            if (enable_nee && !is_delta)                                           // integrator.rs:231-234
            {
                V3 e = estimate_direct_explicit(tr, rng, r, hi, material);
                V3 s = estimate_direct_bsdf(tr, rng, r, hi, material);
                accumulated = accumulated + path_weight * (e + s);
            }
            r = Ray::make(r.at(hi.t), material.scatter_direction(rng, r.d, hi.normal, hi.front_facing)); // integrator.rs:236-239
            BsdfPdf info = material.get_bsdf_pdf(wi, r.d, hi);
            if (info.pdf < MIN_PDF) break;                                         // integrator.rs:243-247
            path_weight = path_weight * (material.get_weakening(r.d, hi.normal) * info.bsdf / info.pdf); // integrator.rs:249
            last_delta = is_delta;
        }
        else
        {
            if (env && env->w)                                                     // integrator.rs:256-262
            {
                float u = mul_add(det_atan2(r.d.x, r.d.z), FRAC_1_PI * 0.5f, 0.5f);
                float v = mul_add(det_asin(r.d.y), -FRAC_1_PI, 0.5f);
                accumulated = accumulated + env->get_pixel_bilinear(u, v) * path_weight;
            }
            else accumulated = accumulated + V3{0.006f, 0.006f, 0.006f} * path_weight;  // integrator.rs:263-266
            break;
        }
        if (b == max_bounces) break;                                               // `for b in 0..=max_bounces`
    }
    Sample s;
    s.position = position;
    s.id = first_id;
    if (is_finite(accumulated))                                                    // integrator.rs:272-280
    {
        V3 c = clamp_length_max(accumulated, 100.0f);
        s.colour = V4{c.x, c.y, c.z, 1.0f};
    }
    else { s.colour = V4{0, 0, 0, 1.0f}; }
    return s;
}

// main.rs:186-207 pixel closure (seed draw, Sobol jitter, u/v, camera ray)
Ray primary_ray(const Camera& cam, const pto_render_cfg& cfg, uint32_t pixel, uint32_t sample, Rng* rng_out)
{
    uint32_t x = pixel % cfg.width, y = pixel / cfg.width;
    Rng rng{stream_state0(cfg.seed, pixel, sample), 0};
    uint32_t seed = rng.next_u32();                                                // main.rs:193
    float off[2];
    ss_sobol(cfg.n_sobol, sample, seed, off);                                      // main.rs:194 (index: deviation B-1)
    float ox = off[0] - 0.5f, oy = off[1] - 0.5f;
    float u = ((float)x + ox) / (float)cfg.width;                                  // main.rs:196
    float v = ((float)y + oy) / (float)cfg.height;                                 // main.rs:197
    *rng_out = rng;
    return cam.create_ray(u, v);
}

} // namespace

// ====================================================================== C API
extern "C" {

pto_ctx* pto_create(void) { return new pto_ctx(); }
void pto_destroy(pto_ctx* c) { delete c; }

int pto_add_material(pto_ctx* c, int kind, const float colour[3], float roughness, float ior, const pto_volume_desc* vol)
{
    Material m;
    m.kind = kind;
    m.colour = V3{colour[0], colour[1], colour[2]};
    m.ior = ior;
    if (kind == PTO_GGX_METAL || kind == PTO_GGX_DIELECTRIC) m.a = rs_clamp(powi2(roughness), 0.0001f, 0.9999f);
    if (vol && vol->present)
    {
        m.has_volume = true;
        if (vol->k != 0.0f) { m.volume.has_absorption = true; m.volume.absorption = V3{vol->absorption[0], vol->absorption[1], vol->absorption[2]} * vol->k; }
        if (vol->c != 0.0f) { m.volume.has_scatter = true; m.volume.c = vol->c; m.volume.g = rs_clamp(vol->g, -0.999f, 0.999f); }
    }
    c->materials.push_back(m);
    return (int)c->materials.size() - 1;
}

int pto_add_model(pto_ctx* c, const float* positions, const float* normals, uint32_t n_tris, int material, const float* affines,
                  uint32_t n_inst)
{
    if (material < 0 || material >= (int)c->materials.size() || n_tris == 0) return -1;
    ModelDesc m;
    m.positions.assign(positions, positions + (size_t)n_tris * 9);
    m.normals.assign(normals, normals + (size_t)n_tris * 9);
    m.n_tris = n_tris;
    m.material = material;
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        const float* a = affines + i * 12;
        Affine f;
        f.m.c0 = V3{a[0], a[4], a[8]};
        f.m.c1 = V3{a[1], a[5], a[9]};
        f.m.c2 = V3{a[2], a[6], a[10]};
        f.t = V3{a[3], a[7], a[11]};
        // model.rs:40-44: to_scale_rotation_translation().0 must equal (1,1,1) exactly
        float det = dot(f.m.c2, cross(f.m.c0, f.m.c1));
        V3 scale{length(f.m.c0) * rs_signum(det), length(f.m.c1), length(f.m.c2)};
        if (!(scale.x == 1.0f && scale.y == 1.0f && scale.z == 1.0f)) return -4;
        m.matrices.push_back(f);
    }
    c->models.push_back(std::move(m));
    return (int)c->models.size() - 1;
}

// load_obj, blas.rs:44-131.  Returns 0, -6 (cannot open) or -7 (where the reference would panic).
static int load_obj(const char* path, std::vector<float>& out_p, std::vector<float>& out_n)
{
    std::ifstream in(path);
    if (!in) return -6;
    std::vector<V3> normals{V3{0, 0, 0}}, positions{V3{0, 0, 0}};                      // blas.rs:46-47
    struct VertexRef { size_t vertex, normal; };                                       // model.rs:20-24
    auto to_f32 = [](const std::string& t, float* v) {
        if (t.empty() || t.find('x') != std::string::npos || t.find('X') != std::string::npos) return false;
        size_t used = 0;
        try { *v = std::stof(t, &used); } catch (...) { return false; }
        return used == t.size();
    };
    auto to_index = [](const std::string& t, size_t len, size_t* v) {                  // blas.rs:85-92
        if (t.empty()) return false;
        size_t used = 0;
        try
        {
            if (t[0] == '-' || t[0] == '+') { long long k = std::stoll(t, &used); *v = (size_t)((long long)len + k); }
            else *v = (size_t)std::stoull(t, &used);
        }
        catch (...) { return false; }
        return used == t.size();
    };
    std::string line;
    while (std::getline(in, line))
    {
        std::istringstream ss(line);
        std::vector<std::string> tokens;
        for (std::string t; ss >> t;) tokens.push_back(t);
        if (tokens.empty()) continue;                                                   // (reference: tokens[0] would panic)
        if (tokens[0] == "v" || tokens[0] == "vn")                                      // blas.rs:60-75
        {
            V3 v;
            if (tokens.size() < 4 || !to_f32(tokens[1], &v.x) || !to_f32(tokens[2], &v.y) || !to_f32(tokens[3], &v.z)) return -7;
            if (tokens[0] == "v") positions.push_back(v);
            else normals.push_back(normalize(v));
        }
        else if (tokens[0] == "f")                                                      // blas.rs:76-120
        {
            std::vector<VertexRef> refs;
            for (size_t k = 1; k < tokens.size(); ++k)
            {
                std::vector<std::string> indices;
                std::string field;
                std::istringstream fs(tokens[k]);
                while (std::getline(fs, field, '/')) indices.push_back(field);
                if (!tokens[k].empty() && tokens[k].back() == '/') indices.push_back("");
                if (indices.size() < 3) return -7;
                VertexRef r;
                if (!to_index(indices[0], positions.size(), &r.vertex) || !to_index(indices[2], normals.size(), &r.normal)) return -7;
                if (r.vertex >= positions.size() || r.normal >= normals.size()) return -7;
                refs.push_back(r);
            }
            for (size_t i = 1; i + 1 < refs.size(); ++i)
            {
                const VertexRef* tri[3] = {&refs[0], &refs[i], &refs[i + 1]};
                for (const VertexRef* vr : tri)
                {
                    V3 position = positions[vr->vertex];
                    V3 normal;
                    if (vr->normal != 0) normal = normals[vr->normal];
                    else
                    {
                        V3 u = positions[tri[1]->vertex] - positions[tri[0]->vertex];
                        V3 v = positions[tri[2]->vertex] - positions[tri[0]->vertex];
                        normal = cross(u, v);
                    }
                    out_p.push_back(position.x); out_p.push_back(position.y); out_p.push_back(position.z);
                    out_n.push_back(normal.x); out_n.push_back(normal.y); out_n.push_back(normal.z);
                }
            }
        }
    }
    return out_p.empty() ? -7 : 0;
}

int pto_add_model_obj(pto_ctx* c, const char* path, int material, const float* affines, uint32_t n_inst)
{
    std::vector<float> p, n;
    int r = load_obj(path, p, n);
    if (r) return r;
    return pto_add_model(c, p.data(), n.data(), (uint32_t)(p.size() / 9), material, affines, n_inst);
}

int pto_model_vertices(pto_ctx* c, int model, float* positions, float* normals, uint32_t cap_tris, uint32_t* n_tris)
{
    if (model < 0 || model >= (int)c->models.size()) return -1;
    const ModelDesc& m = c->models[model];
    *n_tris = m.n_tris;
    if (cap_tris == 0) return 0;
    if (cap_tris < m.n_tris) return -1;
    std::memcpy(positions, m.positions.data(), (size_t)m.n_tris * 36);
    std::memcpy(normals, m.normals.data(), (size_t)m.n_tris * 36);
    return 0;
}

int pto_build(pto_ctx* c)                                                          // scene.rs:21-35
{
    auto sc = std::make_unique<Scene>();
    sc->materials = c->materials;
    std::vector<const ModelDesc*> all, lights;
    for (auto& m : c->models)
    {
        all.push_back(&m);
        if (c->materials[m.material].is_emissive()) lights.push_back(&m);
    }
    if (all.empty()) return -1;
    sc->world.build(all);
    sc->lights.build(lights);
    sc->has_lights = !lights.empty();
    if (sc->has_lights) sc->light_sampler.build(sc->lights, sc->materials);
    c->scene = std::move(sc);
    return 0;
}

int pto_set_camera(pto_ctx* c, const float eye[3], const float target[3], float fov_y_deg, float aspect)
{
    c->camera.init(V3{eye[0], eye[1], eye[2]}, V3{target[0], target[1], target[2]}, fov_y_deg, aspect);
    return 0;
}

int pto_camera_move(pto_ctx* c, float dx, float dz, float dt) { if (!c->camera.set) return -1; c->camera.update_origin(dx, dz, dt); return 0; }
int pto_camera_rotate(pto_ctx* c, float dx, float dy, float dt) { if (!c->camera.set) return -1; c->camera.update_rotation(dx, dy, dt); return 0; }
int pto_camera_angles(pto_ctx* c, float out[2]) { out[0] = c->camera.pitch; out[1] = c->camera.yaw; return 0; }

int pto_set_environment(pto_ctx* c, uint32_t w, uint32_t h, const float* rgb)
{
    c->env = EnvMap();
    if (!rgb || w == 0 || h == 0) return 0;
    c->env.w = w; c->env.h = h;
    c->env.data.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) c->env.data[i] = V3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    return 0;
}

int pto_camera_matrices(pto_ctx* c, float m34[12], float ip[16], float rm[16])
{
    if (!c->camera.set) return -1;
    const Affine& a = c->camera.matrix;
    float rows[12] = {a.m.c0.x, a.m.c1.x, a.m.c2.x, a.t.x, a.m.c0.y, a.m.c1.y, a.m.c2.y, a.t.y, a.m.c0.z, a.m.c1.z, a.m.c2.z, a.t.z};
    std::memcpy(m34, rows, sizeof(rows));
    std::memcpy(ip, &c->camera.inv_projection, 64); // column-major
    std::memcpy(rm, &c->camera.ray_matrix, 64);
    return 0;
}

int pto_inv_projection(pto_ctx* c, float out16[16])                            // (cam.matrix * cam.inv_projection).inverse()  main.rs:128
{
    if (!c->camera.set) return -1;
    M4 inv = mat4_inverse(c->camera.ray_matrix);
    std::memcpy(out16, &inv, 64);
    return 0;
}

int pto_create_ray(pto_ctx* c, float s, float t, float o[3], float d[3])
{
    if (!c->camera.set) return -1;
    Ray r = c->camera.create_ray(s, t);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
    d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
    return 0;
}

int pto_primary_ray(pto_ctx* c, const pto_render_cfg* cfg, uint32_t pixel, uint32_t sample, float o[3], float d[3])
{
    if (!c->camera.set) return -1;
    Rng rng;
    Ray r = primary_ray(c->camera, *cfg, pixel, sample, &rng);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
    d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
    return 0;
}

static int render_impl(pto_ctx* c, const pto_render_cfg* cfg, float* accum, float* position, uint32_t* id, uint64_t* counters,
                       float* samples_out)
{
    if (!c->scene || !c->camera.set) return -3;
    if (cfg->enable_nee && !c->scene->has_lights) return -3;
    const uint32_t W = cfg->width, H = cfg->height;
    uint32_t r0 = cfg->row_begin, r1 = cfg->row_end;
    if (r0 == 0 && r1 == 0) r1 = H;
    unsigned nt = cfg->threads ? cfg->threads : std::max(1u, std::thread::hardware_concurrency() - 1); // main.rs:72
    std::atomic<uint32_t> next_row{r0};
    std::vector<Counters> ctrs(nt);
    auto worker = [&](unsigned tid) {
        Tracer tr(c->scene.get());
        for (;;)
        {
            uint32_t y = next_row.fetch_add(1);
            if (y >= r1) break;
            for (uint32_t x = 0; x < W; ++x)
            {
                uint32_t pixel = y * W + x;
                // accumulate.wgsl:20-23: acc += (rgb, 1), one sample after the other
                V4 acc{0, 0, 0, 0};
                if (accum) acc = V4{accum[pixel * 4 + 0], accum[pixel * 4 + 1], accum[pixel * 4 + 2], accum[pixel * 4 + 3]};
                uint32_t idv = id ? id[pixel] : 0;
                for (uint32_t s = 0; s < cfg->n_samples; ++s)
                {
                    uint32_t sample = cfg->first_sample + s;
                    Rng rng;
                    Ray ray = primary_ray(c->camera, *cfg, pixel, sample, &rng);
                    Sample sm = integrate(tr, ray, rng, cfg->max_bounces, cfg->enable_nee != 0, &c->env);
                    tr.ctr.c[5]++;
                    acc = V4{acc.x + sm.colour.x, acc.y + sm.colour.y, acc.z + sm.colour.z, acc.w + 1.0f};
                    idv = (idv << 16) | (uint32_t)sm.id;                           // main.rs:206
                    if (position && s + 1 == cfg->n_samples) std::memcpy(position + pixel * 4, &sm.position, 16);
                    if (samples_out) std::memcpy(samples_out + ((size_t)s * W * H + pixel) * 4, &sm.colour, 16);
                }
                if (accum) std::memcpy(accum + pixel * 4, &acc, 16);
                if (id) id[pixel] = idv;
            }
        }
        ctrs[tid] = tr.ctr;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (counters)
        for (int k = 0; k < PTO_N_COUNTERS; ++k)
        {
            counters[k] = 0;
            for (auto& ct : ctrs) counters[k] += ct.c[k];
        }
    return 0;
}

int pto_render(pto_ctx* c, const pto_render_cfg* cfg, float* accum, float* position, uint32_t* id, uint64_t* counters)
{
    return render_impl(c, cfg, accum, position, id, counters, nullptr);
}
int pto_render_samples(pto_ctx* c, const pto_render_cfg* cfg, float* samples)
{
    return render_impl(c, cfg, nullptr, nullptr, nullptr, nullptr, samples);
}

int pto_integrate(pto_ctx* c, const pto_render_cfg* cfg, const float o[3], const float d[3], uint32_t pixel, uint32_t sample,
                  uint32_t draws_consumed, float colour[4], float position[4], uint8_t* id)
{
    if (!c->scene) return -3;
    Tracer tr(c->scene.get());
    Rng rng{stream_state0(cfg->seed, pixel, sample), draws_consumed};
    Ray r = Ray::make(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    Sample s = integrate(tr, r, rng, cfg->max_bounces, cfg->enable_nee != 0, &c->env);
    std::memcpy(colour, &s.colour, 16);
    std::memcpy(position, &s.position, 16);
    *id = s.id;
    return 0;
}

int pto_trace_closest(pto_ctx* c, int which, uint32_t n, const float* o, const float* d, const float* tmax, float* t, float* u, float* v,
                      uint32_t* inst, uint32_t* prim, float* normal, uint8_t* front)
{
    if (!c->scene) return -3;
    Tracer tr(c->scene.get());
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    for (uint32_t i = 0; i < n; ++i)
    {
        Ray r = Ray::make(V3{o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, V3{d[i * 3], d[i * 3 + 1], d[i * 3 + 2]});
        HitInfo hi;
        uint32_t b, p, in;
        if (tl.intersect(tr.sc, r, tmax[i], &hi, &b, &p, &in))
        {
            t[i] = hi.t; u[i] = hi.u; v[i] = hi.v; inst[i] = in; prim[i] = p;
            if (normal) { normal[i * 3] = hi.normal.x; normal[i * 3 + 1] = hi.normal.y; normal[i * 3 + 2] = hi.normal.z; }
            if (front) front[i] = hi.front_facing;
        }
        else
        {
            t[i] = INF; u[i] = 0; v[i] = 0; inst[i] = 0xffffffffu; prim[i] = 0xffffffffu;
            if (normal) { normal[i * 3] = normal[i * 3 + 1] = normal[i * 3 + 2] = 0; }
            if (front) front[i] = 0;
        }
    }
    return 0;
}

int pto_trace_any(pto_ctx* c, int which, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* hit)
{
    if (!c->scene) return -3;
    Tracer tr(c->scene.get());
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    for (uint32_t i = 0; i < n; ++i)
    {
        Ray r = Ray::make(V3{o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, V3{d[i * 3], d[i * 3 + 1], d[i * 3 + 2]});
        hit[i] = tl.any_intersect(tr.sc, r, tmax[i]) ? 1 : 0;
    }
    return 0;
}

int pto_blas_count(pto_ctx* c) { return c->scene ? (int)c->scene->world.blas.size() : -3; }

int pto_blas_dump(pto_ctx* c, int which, int blas, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a,
                  uint32_t* b, uint32_t* n_prim_ids, uint32_t* prim_ids, uint32_t cap_nodes, uint32_t cap_ids)
{
    if (!c->scene) return -3;
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    if (blas < 0 || blas >= (int)tl.blas.size()) return -1;
    const BLAS& bl = tl.blas[blas];
    *n_nodes = (uint32_t)bl.nodes.size();
    *root = bl.root;
    uint32_t ids = 0;
    for (size_t i = 0; i < bl.nodes.size(); ++i)
    {
        const BLASNode& n = bl.nodes[i];
        uint32_t cnt = n.type == NODE_BRANCH ? 0 : (n.type == NODE_LEAF_SINGLE ? 1 : (uint32_t)n.prims.size());
        if (i < cap_nodes)
        {
            std::memcpy(boxes6 + i * 6, &n.box, 24);
            kind[i] = n.type == NODE_BRANCH ? 0 : 1;
            a[i] = n.type == NODE_BRANCH ? n.left : ids;
            b[i] = n.type == NODE_BRANCH ? n.right : cnt;
        }
        for (uint32_t k = 0; k < cnt; ++k)
        {
            if (ids < cap_ids) prim_ids[ids] = n.type == NODE_LEAF_SINGLE ? n.prim : n.prims[k];
            ids++;
        }
    }
    *n_prim_ids = ids;
    return (bl.nodes.size() <= cap_nodes && ids <= cap_ids) ? 0 : -1;
}

int pto_tlas_dump(pto_ctx* c, int which, uint32_t* n_nodes, uint32_t* root, float* boxes6, uint32_t* kind, uint32_t* a, uint32_t* b,
                  uint32_t cap_nodes)
{
    if (!c->scene) return -3;
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    *n_nodes = (uint32_t)tl.nodes.size();
    *root = tl.root;
    for (size_t i = 0; i < tl.nodes.size() && i < cap_nodes; ++i)
    {
        const TLASNode& n = tl.nodes[i];
        std::memcpy(boxes6 + i * 6, &n.box, 24);
        kind[i] = n.leaf ? 1 : 0;
        a[i] = n.leaf ? n.instance : n.left;
        b[i] = n.leaf ? n.blas : n.right;
    }
    return tl.nodes.size() <= cap_nodes ? 0 : -1;
}

// TLASNodeType::Leaf { matrix, inv_matrix }  tlas_bvh.rs:36-41,92-101 — leaves in allocation order (their `instance` numbers)
int pto_tlas_instances(pto_ctx* c, int which, uint32_t* n, float* matrix12, float* inv_matrix12, uint32_t cap)
{
    if (!c->scene) return -3;
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    uint32_t count = 0;
    for (const TLASNode& nd : tl.nodes) count += nd.leaf ? 1u : 0u;
    *n = count;
    if (count > cap) return -1;
    auto rows = [](const Affine& x, float* o) {
        const float r[12] = {x.m.c0.x, x.m.c1.x, x.m.c2.x, x.t.x, x.m.c0.y, x.m.c1.y, x.m.c2.y, x.t.y, x.m.c0.z, x.m.c1.z, x.m.c2.z, x.t.z};
        std::memcpy(o, r, sizeof(r));
    };
    for (const TLASNode& nd : tl.nodes)
        if (nd.leaf && nd.instance < count) { rows(nd.matrix, matrix12 + 12 * (size_t)nd.instance); rows(nd.inv_matrix, inv_matrix12 + 12 * (size_t)nd.instance); }
    return 0;
}

int pto_light_cdf(pto_ctx* c, uint32_t* n, float* pdf, float* cdf, uint32_t* blas, uint32_t* prim, float* max_weight, uint32_t cap)
{
    if (!c->scene) return -3;
    const LightSampler& ls = c->scene->light_sampler;
    *n = (uint32_t)ls.lights.size();
    *max_weight = ls.max;
    for (size_t i = 0; i < ls.lights.size() && i < cap; ++i)
    {
        pdf[i] = ls.lights[i].pdf; cdf[i] = ls.cdf[i]; blas[i] = ls.lights[i].blas; prim[i] = ls.lights[i].prim;
    }
    return ls.lights.size() <= cap ? 0 : -1;
}

int pto_triangle_dump(pto_ctx* c, int which, int blas, uint32_t prim, float out36[36])
{
    if (!c->scene) return -3;
    const TLAS& tl = which ? c->scene->lights : c->scene->world;
    if (blas < 0 || blas >= (int)tl.blas.size() || prim >= tl.blas[blas].primitives.size()) return -1;
    const Triangle& t = tl.blas[blas].primitives[prim];
    std::memcpy(out36, &t.n0, 16);
    std::memcpy(out36 + 4, &t.n1, 16);
    std::memcpy(out36 + 8, &t.n2, 16);
    std::memcpy(out36 + 12, &t.positions, 36);
    std::memcpy(out36 + 21, &t.normals, 36);
    for (int i = 30; i < 36; ++i) out36[i] = 0;
    return 0;
}

// ====================================================================== after the path: State::update + State::render
// Restatement of src/shaders/{accumulate,velocity,compute,shader}.wgsl as dispatched by src/state.rs:505-586,629-667.
// WGSL leaves several things to the GPU (bilinear filter weights are fixed-point in hardware, mat*vec summation order, pow/exp
// precision, out-of-range float->int casts): they are DEFINED here — and identically in libptmi's post kernels — as exact
// binary32 arithmetic in the order written below, pow(x,c) = exp(c ln x) with the deterministic routines, saturating casts,
// out-of-bounds textureLoad = 0.  The reference's own output therefore differs from this by its GPU's sampler rounding.
namespace {
struct Img { const float* p; int w, h; V4 at(int x, int y) const { const float* q = p + ((size_t)y * w + x) * 4; return V4{q[0], q[1], q[2], q[3]}; } };
V4 v4add(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
V4 v4scale(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
int sat_i32(float f) { return f != f ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (-2147483647 - 1) : (int)f)); }
// textureSampleLevel(tex, linear clamp-to-edge sampler, uv, 0)
V4 sample_bilinear(const Img& t, float u, float v)
{
    float x = u * (float)t.w - 0.5f, y = v * (float)t.h - 0.5f;
    float fx0 = std::floor(x), fy0 = std::floor(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = sat_i32(fx0), y0 = sat_i32(fy0);
    auto cl = [](int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
    int xa = cl(x0, t.w), xb = cl(x0 < 2147483647 ? x0 + 1 : x0, t.w), ya = cl(y0, t.h), yb = cl(y0 < 2147483647 ? y0 + 1 : y0, t.h);
    V4 top = v4add(v4scale(t.at(xa, ya), 1.0f - fx), v4scale(t.at(xb, ya), fx));
    V4 bot = v4add(v4scale(t.at(xa, yb), 1.0f - fx), v4scale(t.at(xb, yb), fx));
    return v4add(v4scale(top, 1.0f - fy), v4scale(bot, fy));
}
V3 w_divide(V4 v) { float d = v.w > 1.0f ? v.w : 1.0f; return V3{v.x / d, v.y / d, v.z / d}; } // v.xyz / max(v.w, 1.0)
V3 rgb_to_ycocg(V3 c) { return V3{(0.25f * c.x + 0.5f * c.y) + -0.25f * c.z, (0.5f * c.x + 0.0f * c.y) + 0.5f * c.z, (0.25f * c.x + -0.5f * c.y) + -0.25f * c.z}; }   // compute.wgsl:64-71 (columns)
V3 ycocg_to_rgb(V3 c) { return V3{(1.0f * c.x + 1.0f * c.y) + 1.0f * c.z, (1.0f * c.x + 0.0f * c.y) + -1.0f * c.z, (-1.0f * c.x + 1.0f * c.y) + -1.0f * c.z}; }       // compute.wgsl:73-80
float fmax_w(float a, float b) { return a > b ? a : b; } // WGSL max/min on non-NaN data
float fmin_w(float a, float b) { return a < b ? a : b; }
V3 clip_aabb(V3 mn, V3 mx, V3 q)                                                     // compute.wgsl:82-101
{
    V3 p_clip = 0.5f * (mx + mn), e_clip = 0.5f * (mx - mn);
    V3 v_clip = q - p_clip;
    V3 v_unit{v_clip.x / e_clip.x, v_clip.y / e_clip.y, v_clip.z / e_clip.z};
    float ma = fmax_w(std::fabs(v_unit.x), fmax_w(std::fabs(v_unit.y), std::fabs(v_unit.z)));
    if (ma > 1.0f) return p_clip + v_clip / ma;
    return q;
}
V3 sample_catmull_rom(const Img& tex, float uvx, float uvy)                            // compute.wgsl:16-62
{
    const float sx = (float)tex.w, sy = (float)tex.h;
    float spx = uvx * sx + 0.5f, spy = uvy * sy + 0.5f;
    float t1x = std::floor(spx - 0.5f) + 0.5f, t1y = std::floor(spy - 0.5f) + 0.5f;
    float fx = spx - t1x, fy = spy - t1y;
    auto w0 = [](float f) { return f * (-0.5f + f * (1.0f - 0.5f * f)); };
    auto w1 = [](float f) { return 1.0f + f * f * (-2.5f + 1.5f * f); };
    auto w2 = [](float f) { return f * (0.5f + f * (2.0f - 1.5f * f)); };
    auto w3 = [](float f) { return f * f * (-0.5f + 0.5f * f); };
    float w0x = w0(fx), w1x = w1(fx), w2x = w2(fx), w3x = w3(fx), w0y = w0(fy), w1y = w1(fy), w2y = w2(fy), w3y = w3(fy);
    float w12x = w1x + w2x, w12y = w1y + w2y;
    float o12x = w2x / (w1x + w2x), o12y = w2y / (w1y + w2y);
    float p0x = (t1x - 1.0f) / sx, p0y = (t1y - 1.0f) / sy, p3x = (t1x + 2.0f) / sx, p3y = (t1y + 2.0f) / sy;
    float p12x = (t1x + o12x) / sx, p12y = (t1y + o12y) / sy;
    V3 c{0, 0, 0};
    auto tap = [&](float u, float v, float wa, float wb) { c = c + w_divide(sample_bilinear(tex, u, v)) * wa * wb; };
    tap(p0x, p0y, w0x, w0y); tap(p12x, p0y, w12x, w0y); tap(p3x, p0y, w3x, w0y);
    tap(p0x, p12y, w0x, w12y); tap(p12x, p12y, w12x, w12y); tap(p3x, p12y, w3x, w12y);
    tap(p0x, p3y, w0x, w3y); tap(p12x, p3y, w12x, w3y); tap(p3x, p3y, w3x, w3y);
    return c;
}
float det_pow(float x, float c) { return det_exp(c * det_ln(x)); }
float gt_tonemap_wgsl(float x, float p, float a, float m, float l, float c, float b)       // shader.wgsl:3-33
{
    float l0 = (p - m) * l / a;
    float tt = rs_clamp((x - 0.0f) / (m - 0.0f), 0.0f, 1.0f);                                // smoothstep(0, m, x)
    float w0 = 1.0f - tt * tt * (3.0f - 2.0f * tt);
    float w2 = x >= m + l0 ? 1.0f : 0.0f;                                                   // step(m + l0, x)
    float w1 = 1.0f - w0 - w2;
    float toe = m * det_pow(x / m, c) + b;
    float lin = m + a * (x - m);
    float s0 = m + l0, s1 = m + a * l0, c2 = a * p / (p - s1);
    float sh = p - (p - s1) * det_exp(-c2 * (x - s0) / p);
    float r = (toe * w0 + lin * w1) + sh * w2;
    return fmax_w(r, 0.0f);
}
} // namespace

namespace {
float gt_tonemap_rs(float x, float p, float a, float m, float l, float c, float b)          // tonemapping.rs:66-96
{
    if (x < 0.0f) return b;
    float l0 = (p - m) * l / a;
    auto smoothstep = [](float x, float e0, float e1) {                                       // :36-52
        if (x < e0) return 0.0f;
        if (x > e1) return 1.0f;
        float q = (x - e0) / (e1 - e0), r = 3.0f - 2.0f * q;
        return q * q * r;
    };
    auto lerp = [](float x, float e0, float e1) {                                             // :20-34
        if (x < e0) return 0.0f;
        if (x > e1) return 1.0f;
        return (x - e0) / (e1 - e0);
    };
    float w0 = 1.0f - smoothstep(x, 0.0f, m);
    float w2 = lerp(x, m + l0, m + l0);
    float w1 = 1.0f - w0 - w2;
    float t = (m * det_pow(x / m, c) + b) * w0;                                               // gt_toe :5
    float u = (m + a * (x - m)) * w1;                                                         // gt_linear :2
    float s0 = m + l0, s1 = m + a * l0, c2 = a * p / (p - s1);
    float v = (p - (p - s1) * det_exp(-c2 * (x - s0) / p)) * w2;                              // gt_shoulder :8-16
    return t + u + v;
}
uint8_t as_u8(float f) { return f != f ? 0 : (f >= 255.0f ? 255 : (f <= 0.0f ? 0 : (uint8_t)f)); } // Rust `as u8`
} // namespace

// ImageHelper::write_image's byte conversion (image_helper.rs:41-48) applied to accumulation.rgb / accumulation.w
extern "C" int pto_post_rgb8(uint32_t w, uint32_t h, const float* accum, uint8_t* out)
{
    const float g = 1.0f / 2.2f;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int k = 0; k < 3; ++k)
            out[3 * i + k] = as_u8(det_pow(gt_tonemap_rs(accum[4 * i + k] / accum[4 * i + 3], 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f), g) * 255.0f);
    return 0;
}

extern "C" int pto_post_accumulate(uint32_t w, uint32_t h, const float* input, float* accum)           // accumulate.wgsl:20-23
{
    for (size_t i = 0; i < (size_t)w * h; ++i)
    {
        accum[4 * i] += input[4 * i]; accum[4 * i + 1] += input[4 * i + 1]; accum[4 * i + 2] += input[4 * i + 2]; accum[4 * i + 3] += 1.0f;
    }
    return 0;
}

extern "C" int pto_post_velocity(uint32_t w, uint32_t h, const float* position, const float* last_inv_proj, float* velocity) // velocity.wgsl:16-39
{
    const float* M = last_inv_proj; // column-major
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x)
        {
            const float* P = position + ((size_t)y * w + x) * 4;
            float cu = ((float)x + 0.5f) / (float)w, cv = ((float)y + 0.5f) / (float)h;
            V4 r;
            float* rr = &r.x;
            for (int i = 0; i < 4; ++i) rr[i] = ((M[i] * P[0] + M[4 + i] * P[1]) + M[8 + i] * P[2]) + M[12 + i] * 1.0f;
            V3 d = w_divide(r);
            float pu = d.x * 0.5f + 0.5f, pv = d.y * 0.5f + 0.5f;
            velocity[((size_t)y * w + x) * 2] = cu - pu;
            velocity[((size_t)y * w + x) * 2 + 1] = cv - pv;
        }
    return 0;
}

extern "C" int pto_post_reproject(uint32_t w, uint32_t h, const float* input, const float* accum, const float* velocity, const uint32_t* id,
                                  float* output)                                                                // compute.wgsl:103-212
{
    Img in{input, (int)w, (int)h}, acc{accum, (int)w, (int)h};
    const float dx = (float)w, dy = (float)h;
    for (int cy = 0; cy < (int)h; ++cy)
        for (int cx = 0; cx < (int)w; ++cx)
        {
            V4 cur4 = in.at(cx, cy);
            V3 current{cur4.x, cur4.y, cur4.z};
            V3 m1{0, 0, 0}, m2{0, 0, 0};
            float closest_depth = 1e20f;
            int vx = 0, vy = 0;
            int x0 = cx - 1 > 0 ? cx - 1 : 0, y0 = cy - 1 > 0 ? cy - 1 : 0;
            int x1 = cx + 1 < (int)w - 1 ? cx + 1 : (int)w - 1, y1 = cy + 1 < (int)h - 1 ? cy + 1 : (int)h - 1;
            int n = (x1 + 1 - x0) * (y1 + 1 - y0);
            for (int x = x0; x <= x1; ++x)
                for (int y = y0; y <= y1; ++y)
                {
                    V4 dd = in.at(x, y);
                    V3 d = rgb_to_ycocg(V3{dd.x, dd.y, dd.z});
                    m1 = m1 + d;
                    m2 = m2 + d * d;
                    if (dd.w < closest_depth) { closest_depth = dd.w; vx = x; vy = y; }
                }
            float cu = ((float)cx + 0.5f) / dx, cv = ((float)cy + 0.5f) / dy;
            float pu = cu - velocity[((size_t)vy * w + vx) * 2], pv = cv - velocity[((size_t)vy * w + vx) * 2 + 1];
            int px = sat_i32(std::floor(pu * dx)), py = sat_i32(std::floor(pv * dy));
            bool oob = px < 0 || py < 0 || px >= (int)w || py >= (int)h;
            uint32_t current_id = id[(size_t)cy * w + cx] & 0xffffu;
            uint32_t old_id = oob ? 0u : ((id[(size_t)py * w + px] >> 16) & 0xffffu);
            float* out = output + ((size_t)cy * w + cx) * 4;
            if (current_id != old_id || oob)
            {
                float c0x = (float)cx / dx, c0y = (float)cy / dy, c1x = c0x + 1.0f / dx, c1y = c0y + 1.0f / dy;
                V4 a = sample_bilinear(in, c0x, c0y), b = sample_bilinear(in, c0x, c1y), c = sample_bilinear(in, c1x, c0y), d = sample_bilinear(in, c1x, c1y);
                V4 s = v4add(v4add(v4add(a, b), c), d);
                out[0] = s.x / 4.0f; out[1] = s.y / 4.0f; out[2] = s.z / 4.0f; out[3] = s.w / 4.0f;
            }
            else
            {
                float fn = (float)n;
                V3 mu = m1 / fn;
                V3 var = m2 / fn - mu * mu;
                V3 sigma{std::sqrt(var.x), std::sqrt(var.y), std::sqrt(var.z)};
                V3 mn = mu - 1.0f * sigma, mx = mu + 1.0f * sigma;
                V3 prev = sample_catmull_rom(acc, pu, pv);
                V3 cl = ycocg_to_rgb(clip_aabb(mn, mx, rgb_to_ycocg(prev)));
                V3 o = cl * (1.0f - 0.15f) + current * 0.15f;                              // mix(clamped, current, 0.15)
                out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = 1.0f;
            }
        }
    return 0;
}

extern "C" int pto_post_tonemap(uint32_t w, uint32_t h, const float* accum, float* out)                       // shader.wgsl:59-64
{
    for (size_t i = 0; i < (size_t)w * h; ++i)
    {
        for (int k = 0; k < 3; ++k) out[4 * i + k] = gt_tonemap_wgsl(accum[4 * i + k] / accum[4 * i + 3], 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f);
        out[4 * i + 3] = 1.0f;
    }
    return 0;
}

void pto_ss_sobol_raw(uint32_t n_points, uint32_t index, uint32_t seed, uint32_t out[3]) { ss_sobol_raw(n_points, index, seed, out); }
void pto_ss_sobol(uint32_t n_points, uint32_t index, uint32_t seed, float out[2]) { ss_sobol(n_points, index, seed, out); }
uint32_t pto_sobol_dim1(uint32_t index) { return sobol_dim1(index); }
uint32_t pto_low_bias_hash(uint32_t x) { return low_bias_hash(x); }
uint32_t pto_lk_hash(uint32_t x, uint32_t seed) { return lk_hash(x, seed); }
uint64_t pto_wyrand(uint64_t seed, uint32_t k) { return wy_mix(seed + (uint64_t)(k + 1) * WY_INC); }
uint64_t pto_stream_state0(uint64_t seed, uint32_t pixel, uint32_t sample) { return stream_state0(seed, pixel, sample); }

// fn: 0 sin/cos(a) -> out0,out1 ; 1 exp(a) ; 2 ln(a) ; 3 hypot(a,b) ; 4 a/b ; 5 sqrt(a) ; 6 tan(a)
void pto_math_batch(int fn, uint32_t n, const float* a, const float* b, float* out0, float* out1)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        switch (fn)
        {
        case 0: det_sincos(a[i], &out0[i], &out1[i]); break;
        case 1: out0[i] = det_exp(a[i]); break;
        case 2: out0[i] = det_ln(a[i]); break;
        case 3: out0[i] = det_hypot(a[i], b[i]); break;
        case 4: out0[i] = a[i] / b[i]; break;
        case 5: out0[i] = std::sqrt(a[i]); break;
        case 6: out0[i] = det_tan(a[i]); break;
        case 8: out0[i] = det_atan2(a[i], b[i]); break;
        case 9: out0[i] = det_asin(a[i]); break;
        }
    }
}

int pto_material_eval(pto_ctx* c, int material, const float incoming[3], const float normal[3], int front_facing, uint64_t seed,
                      uint32_t pixel, uint32_t sample, uint32_t draws_consumed, float out[9])
{
    if (material < 0 || material >= (int)c->materials.size()) return -1;
    const Material& m = c->materials[material];
    Rng rng{stream_state0(seed, pixel, sample), draws_consumed};
    V3 in{incoming[0], incoming[1], incoming[2]}, n{normal[0], normal[1], normal[2]};
    V3 wo = m.scatter_direction(rng, in, n, front_facing != 0);
    HitInfo hi{n, 0, 0, 0, front_facing != 0};
    BsdfPdf bp = m.get_bsdf_pdf(-in, wo, hi);
    out[0] = wo.x; out[1] = wo.y; out[2] = wo.z;
    out[3] = bp.bsdf.x; out[4] = bp.bsdf.y; out[5] = bp.bsdf.z;
    out[6] = bp.pdf;
    out[7] = m.get_weakening(wo, n);
    out[8] = (float)(rng.k - draws_consumed);
    return 0;
}

} // extern "C"
