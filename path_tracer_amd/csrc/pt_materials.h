// MaterialTrait implementations for the shading kernels (src/tlas/tlas_bvh/blas/primitive/material.rs,
// src/utility.rs, material/onb.rs).  Each routine consumes RNG draws in the reference's program order.
#pragma once
#include "pt_types.h"

namespace pt {

#define PT_PI 3.14159265358979323846f
#define PT_TAU 6.28318530717958647692f
#define PT_FRAC_1_PI 0.318309886183790671537767526745028724f
#define PT_EPSILON 5e-04f /* utility.rs:4 */

struct BsdfSample { f3 bsdf; float pdf; };

PT_HD f3 reflect_rs(f3 i, f3 n) { return i - 2.0f * dot3(n, i) * n; }                           // utility.rs:21
PT_HD f3 refract_rs(f3 i, f3 n, float eta)                                                    // utility.rs:23-36
{
    float ndi = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k <= 0.0f) { float q = from_bits(0x7fc00000u); return f3{q, q, q}; }
    return eta * i - (eta * ndi + sqrtf(k)) * n;
}
PT_HD f3 cosine_vector(Stream& rng)                                                          // utility.rs:7-19
{
    float r = sqrtf(rng.f32());
    float z = sqrtf(1.0f - r * r);
    float phi = PT_TAU * rng.f32();
    float s, c;
    sincos_det(phi, &s, &c);
    return f3{c * r, s * r, z};
}
PT_HD m33 onb_ggx(f3 v)                                                                      // onb.rs:9-27
{
    if (v.z > 0.99999f) return m33{f3{1.0f, 0.0f, 0.0f}, f3{-0.0f, -1.0f, -0.0f}, f3{0.0f, 0.0f, 1.0f}};
    f3 t1 = unit3(cross3(v, f3{0.0f, 0.0f, 1.0f}));
    f3 t2 = cross3(t1, v);
    return m33{t1, t2, v};
}
PT_HD m33 transpose33(const m33& m) { return m33{f3{m.c0.x, m.c1.x, m.c2.x}, f3{m.c0.y, m.c1.y, m.c2.y}, f3{m.c0.z, m.c1.z, m.c2.z}}; }

struct MatView
{
    uint32_t kind;
    f3 colour;
    float alpha, ior;
};
PT_HD MatView load_material(const DMaterial* mats, uint32_t idx)
{
    const DMaterial& m = mats[idx];
    return MatView{m.kind, f3{m.colour[0], m.colour[1], m.colour[2]}, m.alpha, m.ior};
}
PT_HD bool mat_is_delta(uint32_t kind) { return kind == MAT_SPECULAR || kind == MAT_DIELECTRIC; }    // material.rs:151,494
PT_HD float mat_weakening(uint32_t kind, f3 wo, f3 n) { return mat_is_delta(kind) ? 1.0f : fabsf(dot3(wo, n)); } // material.rs:67-77

// ---- GGX pieces, material.rs:189-284
PT_HD float ggx_d(float a, f3 h)
{
    if (h.z <= 0.0f) return 0.0f;
    float c2 = h.z * h.z;
    float tan_sq = sqrtf(1.0f - c2) / c2; // as written, material.rs:197
    float x = (a * a) + tan_sq;
    return a * a / (PT_PI * c2 * c2 * x * x);
}
PT_HD float schlick(float v_dot_h, float f0) { return fma_rs(pow5(1.0f - v_dot_h), 1.0f - f0, f0); }  // material.rs:205
PT_HD f3 schlick_rgb(float v_dot_h, f3 f0)                                                            // material.rs:207
{
    f3 om{1.0f - f0.x, 1.0f - f0.y, 1.0f - f0.z};
    return f0 + (om * pow5(1.0f - v_dot_h));
}
PT_HD float ggx_g1(float a, f3 v, f3 h)                                                               // material.rs:210-221
{
    if (v.z * dot3(h, v) <= 0.0f) return 0.0f;
    float tan2 = inv_sq(v.z) - 1.0f;
    return 2.0f / (1.0f + sqrtf(1.0f + a * a * tan2));
}
PT_HD float ggx_g_uncorrelated(float a, f3 wi, f3 wo)                                                 // material.rs:227-244
{
    if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
    float a2 = a * a;
    float x = 2.0f * wi.z * wo.z;
    float y = 1.0f - a2;
    float z = wo.z * hypot_det(a, wi.z * sqrtf(y));
    float w = wi.z * hypot_det(a, wo.z * sqrtf(y));
    return x / (z + w);
}
PT_HD f3 ggx_half_vector(float a, Stream& rng, f3 incoming, f3 normal)                                // material.rs:248-284
{
    m33 onb_a = onb_from_normal(normal);
    f3 v_ = mul(transpose33(onb_a), -incoming);
    f3 v = unit3(v_ * f3{a, a, 1.0f});
    m33 onb_b = onb_ggx(v);
    float u1 = rng.f32();
    float u2 = rng.f32();
    float a_ = 1.0f / (1.0f + v.z);
    bool cond = u2 < a_;
    float r = min_num(sqrtf(u1), 0.9999f);
    float phi = cond ? (PT_PI * u2 / a_) : (PT_PI + ((u2 - a_) / (1.0f - a_)) * PT_PI);
    float sn, cs;
    sincos_det(phi, &sn, &cs);
    float p1 = r * cs;
    float p2 = r * sn * (cond ? 1.0f : v.z);
    f3 h_ = mul(onb_b, f3{p1, p2, sqrtf(1.0f - p1 * p1 - p2 * p2)});
    return mul(onb_a, unit3(h_ * f3{a, a, 1.0f}));
}
PT_HD float dielectric_fresnel(float cosine, float eta)                                               // material.rs:477-489
{
    if (eta * eta * (1.0f - cosine * cosine) > 1.0f) return 1.0f;
    float f0 = sq((eta - 1.0f) / (eta + 1.0f));
    return fma_rs(pow5(1.0f - cosine), 1.0f - f0, f0);
}

// ---- participating media, material/volume.rs
// VolumeScatter::scatter_direction (Henyey-Greenstein)  volume.rs:32-60
PT_HD f3 hg_direction(float g, Stream& rng, f3 incoming)
{
    float u0 = rng.f32();
    float u1 = rng.f32();
    float phi = 2.0f * PT_PI * u0;
    float z;
    if (g == 0.0f) z = 1.0f - 2.0f * u1;
    else
    {
        float x = (1.0f - g * g) / (1.0f + g * (1.0f - 2.0f * u1));
        z = (1.0f + g * g - x * x) / (2.0f * g);
    }
    float sn, cs;
    sincos_det(phi, &sn, &cs);
    float r = sqrtf(1.0f - z * z);
    return mul(onb_from_normal(-incoming), f3{r * cs, r * sn, z});
}
// VolumeAbsorption::get_transmission  volume.rs:113
PT_HD f3 beer_lambert(f3 absorption, float dist)
{
    f3 e = (-absorption) * dist;
    return f3{exp_det(e.x), exp_det(e.y), exp_det(e.z)};
}
// Equirect lookup of the environment on a miss  integrator.rs:256-262, image_helper.rs:61-88
struct EnvView { const f4* data; uint32_t w, h, pad0, pad1; };
PT_HD uint32_t sat_u32(float f) { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xffffffffu : (uint32_t)f); } // Rust `as u32`
PT_HD f3 env_lookup(const EnvView& env, f3 d)
{
    float u = fma_rs(atan2_det(d.x, d.z), PT_FRAC_1_PI * 0.5f, 0.5f);
    float v = fma_rs(asin_det(d.y), -PT_FRAC_1_PI, 0.5f);
    float x = (float)env.w * u, y = (float)env.h * v;
    uint32_t x0 = sat_u32(x), y0 = sat_u32(y);
    float xf = x - truncf(x), yf = y - truncf(y);
    const uint32_t xa = x0 % env.w, xb = (x0 + 1u) % env.w, ya = y0 % env.h, yb = (y0 + 1u) % env.h;
    const f4 c00 = env.data[(size_t)ya * env.w + xa], c01 = env.data[(size_t)yb * env.w + xa];
    const f4 c10 = env.data[(size_t)ya * env.w + xb], c11 = env.data[(size_t)yb * env.w + xb];
    const f3 a{c00.x, c00.y, c00.z}, b{c01.x, c01.y, c01.z}, c{c10.x, c10.y, c10.z}, e{c11.x, c11.y, c11.z};
    return (((1.0f - xf) * (1.0f - yf)) * a + ((1.0f - xf) * yf) * b + (xf * (1.0f - yf)) * c) + (xf * yf) * e;
}

// MaterialTrait::scatter_direction
PT_HD f3 mat_scatter(const MatView& m, Stream& rng, f3 incoming, f3 normal, bool front)
{
    switch (m.kind)
    {
    case MAT_LAMBERTIAN: return mul(onb_from_normal(normal), cosine_vector(rng));                     // material.rs:104-107
    case MAT_SPECULAR: return reflect_rs(incoming, normal);                                           // material.rs:153
    case MAT_GGX_METAL: { f3 h = ggx_half_vector(m.alpha, rng, incoming, normal); return reflect_rs(incoming, h); } // material.rs:325
    case MAT_GGX_DIELECTRIC:                                                                          // material.rs:326-346
    {
        f3 h = ggx_half_vector(m.alpha, rng, incoming, normal);
        float eta = front ? (1.0f / m.ior) : m.ior;
        float f0 = sq((eta - 1.0f) / (eta + 1.0f));
        float f = schlick(-dot3(incoming, h), f0);
        f3 refracted = refract_rs(incoming, h, eta);
        bool reflected = anynan3(refracted);
        if (!reflected) reflected = rng.f32() < f; // the draw happens only when refraction is possible, material.rs:335
        return reflected ? reflect_rs(incoming, h) : refracted;
    }
    case MAT_DIELECTRIC:                                                                              // material.rs:496-509
    {
        float eta = front ? (1.0f / m.ior) : m.ior;
        float cosine = -dot3(incoming, normal);
        if (rng.f32() < dielectric_fresnel(cosine, eta)) return reflect_rs(incoming, normal);
        return refract_rs(incoming, normal, eta);
    }
    default: return f3{0.0f, 0.0f, 0.0f};                                                             // Emissive, material.rs:133
    }
}

// MaterialTrait::get_bsdf_pdf(incoming, outgoing, hit)
PT_HD BsdfSample mat_bsdf_pdf(const MatView& m, f3 incoming, f3 outgoing, f3 normal, bool front)
{
    switch (m.kind)
    {
    case MAT_LAMBERTIAN:                                                                              // material.rs:109-115
    {
        float cosine = dot3(outgoing, normal);
        return BsdfSample{m.colour * PT_FRAC_1_PI, cosine * PT_FRAC_1_PI};
    }
    case MAT_EMISSIVE:
    case MAT_SPECULAR: return BsdfSample{m.colour, 1.0f};                                             // material.rs:134,155
    case MAT_DIELECTRIC:                                                                              // material.rs:511-527
    {
        float cosine = -dot3(incoming, outgoing);
        float eta = front ? (1.0f / m.ior) : m.ior;
        float f = dielectric_fresnel(cosine, eta);
        if (dot3(outgoing, normal) > 0.0f) return BsdfSample{bc3(f), f};
        float b = (1.0f - f) / (eta * eta);
        return BsdfSample{m.colour * b, 1.0f - f};
    }
    default:                                                                                          // GGX, material.rs:349-450
    {
        const bool transmissive = m.kind == MAT_GGX_DIELECTRIC;
        const float a = m.alpha;
        m33 onb_inv = transpose33(onb_from_normal(normal));
        f3 wi = mul(onb_inv, outgoing);
        f3 wo = mul(onb_inv, incoming);
        bool transmitted = wi.z < 0.0f;
        float eta = front ? m.ior : (1.0f / m.ior);
        f3 h;
        if (transmissive && transmitted) { f3 h_ = unit3(eta * wi + wo); h = h_ * signum_rs(h_.z); }
        else { h = unit3(wi + wo); }
        float i_dot_h = dot3(wi, h), o_dot_h = dot3(wo, h);
        float d = ggx_d(a, h);
        float f, g;
        if (!transmissive) { f = 1.0f; g = ggx_g_uncorrelated(a, wi, wo); }
        else
        {
            float f0 = sq((eta - 1.0f) / (eta + 1.0f));
            f = schlick(fabsf(i_dot_h), f0);
            g = ggx_g1(a, wi, h) * ggx_g1(a, wo, h);                                                   // material.rs:224
        }
        if (transmitted)
        {
            if (!transmissive) return BsdfSample{f3{0.0f, 0.0f, 0.0f}, 0.0f};                          // BsdfPdf::invalid()
            float x = fabsf(i_dot_h * o_dot_h);
            float y = fabsf(wi.z * wo.z);
            float z = (1.0f - f) * g * d;
            float w = (eta * i_dot_h) + o_dot_h;
            float btdf = (x * z) / (y * w * w);
            float jac = fabsf(o_dot_h) / (w * w);
            float pdf = d * (1.0f - f) * fabsf(h.z) * jac;
            return BsdfSample{m.colour * btdf * eta * eta, pdf};
        }
        float brdf = f * g * d / (4.0f * fabsf(wi.z * wo.z));
        float jac = 1.0f / (4.0f * fabsf(o_dot_h));
        float pdf = d * h.z * f * jac;
        f3 tint = transmissive ? f3{1.0f, 1.0f, 1.0f} : schlick_rgb(fabsf(i_dot_h), m.colour);
        return BsdfSample{brdf * tint, pdf};
    }
    }
}

} // namespace pt
