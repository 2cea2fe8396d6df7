// Arithmetic shared by the host scene builder and the gfx950 kernels.
//
// The reference's float behaviour comes from glam 0.23 (SSE2 Vec3A), nanorand 0.7 WyRand and Rust core f32 intrinsics
// (SURVEY.md Appendix A).  Everything here is an explicit IEEE-754 binary32 operation in the reference's order;
// fused multiply-add appears only where the reference calls mul_add (src/ray.rs:20, src/integrator.rs:211,
// material.rs:205,487).  Build with -ffp-contract=off and without fast-math.
//
// sin/cos/exp/ln are this library's own polynomial routines (the reference calls the host libm, which cannot be
// reproduced on a GPU); hypot follows glibc's hypotf (binary64 sqrt of the exact squares).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PT_HD inline
#endif

namespace pt {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

PT_HD uint32_t bits(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_HD bool isnan_f(float a) { return a != a; }
PT_HD bool finite_f(float a) { return (bits(a) & 0x7f800000u) != 0x7f800000u; }

// glam Vec3A::min/max == SSE minps/maxps: second operand wins on NaN and on +-0 ties
PT_HD float min_sse(float a, float b) { return a < b ? a : b; }
PT_HD float max_sse(float a, float b) { return a > b ? a : b; }
// Rust f32::min (minNum)
PT_HD float min_num(float a, float b) { return isnan_f(a) ? b : (isnan_f(b) ? a : (a < b ? a : b)); }
PT_HD float clamp_rs(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
PT_HD float signum_rs(float a) { return isnan_f(a) ? a : from_bits((bits(a) & 0x80000000u) | 0x3f800000u); }
// `a.signum() != b.signum()`  (primitive.rs:122,131,138)
PT_HD bool sign_mismatch(float a, float b) { return isnan_f(a) || isnan_f(b) || (((bits(a) ^ bits(b)) >> 31) != 0u); }
// key that orders like f32::total_cmp
PT_HD int32_t total_order_key(float f) { int32_t i = (int32_t)bits(f); return i ^ (int32_t)(((uint32_t)(i >> 31)) >> 1); }

PT_HD float sq(float x) { return x * x; }                                   // powi(2)
PT_HD float pow5(float x) { float a = x * x; float b = a * a; return x * b; } // powi(5), LLVM binary decomposition
PT_HD float inv_sq(float x) { return 1.0f / (x * x); }                      // powi(-2)
PT_HD float fma_rs(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

PT_HD f3 bc3(float a) { return f3{a, a, a}; }
PT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
PT_HD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
PT_HD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
PT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PT_HD f3 min3(f3 a, f3 b) { return f3{min_sse(a.x, b.x), min_sse(a.y, b.y), min_sse(a.z, b.z)}; }
PT_HD f3 max3(f3 a, f3 b) { return f3{max_sse(a.x, b.x), max_sse(a.y, b.y), max_sse(a.z, b.z)}; }
PT_HD f3 rcp3(f3 a) { return f3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }
PT_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }                          // glam dot3
PT_HD float dot4(f4 a, f4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }            // glam dot4
PT_HD f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_HD float len_sq(f3 a) { return dot3(a, a); }
PT_HD float len3(f3 a) { return sqrtf(dot3(a, a)); }
PT_HD f3 unit3(f3 a) { float l = sqrtf(dot3(a, a)); return f3{a.x / l, a.y / l, a.z / l}; }            // Vec3A::normalize
PT_HD float hmax3(f3 a) { return max_sse(max_sse(a.x, a.z), max_sse(a.y, a.z)); }                      // Vec3A::max_element
PT_HD float hmin3(f3 a) { return min_sse(min_sse(a.x, a.z), min_sse(a.y, a.z)); }
PT_HD bool finite3(f3 a) { return finite_f(a.x) && finite_f(a.y) && finite_f(a.z); }
PT_HD bool anynan3(f3 a) { return isnan_f(a.x) || isnan_f(a.y) || isnan_f(a.z); }
PT_HD f3 fma3(f3 a, f3 b, f3 c) { return f3{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z)}; }
PT_HD f3 clamp_len_max(f3 v, float m)                                                                  // Vec3A::clamp_length_max
{
    float l2 = dot3(v, v);
    if (l2 > m * m) { float s = (1.0f / sqrtf(l2)) * m; return v * s; }
    return v;
}

// 3x3 by columns; rigid 3x4
struct m33 { f3 c0, c1, c2; };
PT_HD f3 mul(const m33& m, f3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }                    // Mat3A * Vec3A
struct xf34 { m33 m; f3 t; };
PT_HD f3 xf_point(const xf34& a, f3 p) { return mul(a.m, p) + a.t; }                                   // transform_point3a
PT_HD f3 xf_vector(const xf34& a, f3 v) { return mul(a.m, v); }                                        // transform_vector3a

// Vec3A::any_orthonormal_pair -> Mat3A::from_cols(c0, c1, n)   (material/onb.rs:1-7)
PT_HD m33 onb_from_normal(f3 n)
{
    float sign = from_bits((bits(n.z) & 0x80000000u) | 0x3f800000u); // copysign(1, n.z)
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    return m33{f3{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x}, f3{b, sign + n.y * n.y * a, -n.y}, n};
}

// ---- deterministic sin/cos/exp/ln ---------------------------------------------------------------------------------
PT_HD void sincos_det(float x, float* s, float* c)
{
    const float magic = 12582912.0f;
    float kf = (x * 0.636619772367581343f + magic) - magic;
    int q = (int)kf;
    float r = __builtin_fmaf(kf, -1.57073974609375f, x);
    r = __builtin_fmaf(kf, -5.657970905303955078125e-05f, r);
    r = __builtin_fmaf(kf, -9.920936294705029468e-10f, r);
    float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    float sr = __builtin_fmaf(ps * r2, r, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
    float sv = (q & 1) ? cr : sr;
    float cv = (q & 1) ? sr : cr;
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}
PT_HD float tan_det(float x) { float s, c; sincos_det(x, &s, &c); return s / c; }
PT_HD float exp_det(float x)
{
    if (isnan_f(x)) return x;
    if (x > 88.72283f) return from_bits(0x7f800000u);
    if (x < -103.0f) return 0.0f;
    const float magic = 12582912.0f;
    float kf = (x * 1.44269504088896341f + magic) - magic;
    int k = (int)kf;
    float r = __builtin_fmaf(kf, -0.693359375f, x);
    r = __builtin_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float e = __builtin_fmaf(p * r, r, r) + 1.0f;
    int k1 = k / 2, k2 = k - k1;
    return e * from_bits((uint32_t)(k1 + 127) << 23) * from_bits((uint32_t)(k2 + 127) << 23);
}
PT_HD float ln_det(float x)
{
    if (isnan_f(x) || x < 0.0f) return from_bits(0x7fc00000u);
    if (x == 0.0f) return from_bits(0xff800000u);
    if (bits(x) == 0x7f800000u) return x;
    uint32_t u = bits(x);
    int e = 0;
    if (u < 0x00800000u) { x = x * 8388608.0f; u = bits(x); e = -23; }
    e += (int)(u >> 23) - 126;
    float m = from_bits((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float ef = (float)e;
    y = __builtin_fmaf(ef, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(ef, 0.693359375f, r);
}
// atan2 / asin (Cephes atanf / asinf layout): equirect env lookup, integrator.rs:258-259
PT_HD float atan_pos_det(float x)
{
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y + __builtin_fmaf(p * z, x, x);
}
PT_HD float atan2_det(float y, float x)
{
    if (isnan_f(x) || isnan_f(y)) return from_bits(0x7fc00000u);
    const float pi = 3.14159265358979323846f, pio2 = 1.5707963267948966f;
    if (x == 0.0f)
    {
        if (y == 0.0f) return (bits(x) >> 31) ? ((bits(y) >> 31) ? -pi : pi) : y;
        return (bits(y) >> 31) ? -pio2 : pio2;
    }
    float a = atan_pos_det(fabsf(y / x));
    if (bits(x) >> 31) a = pi - a;
    return (bits(y) >> 31) ? -a : a;
}
PT_HD float asin_det(float x)
{
    float a = fabsf(x);
    if (!(a <= 1.0f)) return from_bits(0x7fc00000u);
    if (a < 1.0e-4f) return x;
    bool big = a > 0.5f;
    float z, t;
    if (big) { z = 0.5f * (1.0f - a); t = sqrtf(z); }
    else { t = a; z = t * t; }
    float p = __builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    float r = __builtin_fmaf(p * z, t, t);
    if (big) { r = r + r; r = 1.5707963267948966f - r; }
    return (bits(x) >> 31) ? -r : r;
}
PT_HD float hypot_det(float a, float b) { double da = (double)a, db = (double)b; return (float)sqrt(da * da + db * db); }

// ---- counter-based WyRand (nanorand 0.7.0 constants) --------------------------------------------------------------
PT_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}
PT_HD uint64_t wyrand_at(uint64_t s0, uint32_t k) // k-th output (k = 0 first) of a WyRand seeded with s0
{
    uint64_t s = s0 + (uint64_t)(k + 1u) * 0xa0761d6478bd642fULL;
    uint64_t m = s ^ 0xe7037ed1a0b428dbULL;
    return mulhi64(s, m) ^ (s * m);
}
PT_HD uint64_t stream_key(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * ((((uint64_t)sample) << 32) | (uint64_t)pixel);
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
struct Stream
{
    uint64_t s0;
    uint32_t k;
    PT_HD uint32_t u32() { uint32_t v = (uint32_t)wyrand_at(s0, k); k += 1u; return v; }
    PT_HD float f32() { return (float)u32() / 4294967296.0f; } // (u32 as f32) / (u32::MAX as f32), range [0,1]
};

// ---- shuffled-scrambled Sobol (src/sampling.rs) -------------------------------------------------------------------
PT_HD uint32_t rev32(uint32_t x) { return __builtin_bitreverse32(x); }
PT_HD uint32_t sobol_direction(int bit) // DIRECTIONS[bit], sampling.rs:4-8
{
    constexpr uint32_t kDirections[32] = {
        0x80000000u, 0xc0000000u, 0xa0000000u, 0xf0000000u, 0x88000000u, 0xcc000000u, 0xaa000000u, 0xff000000u,
        0x80800000u, 0xc0c00000u, 0xa0a00000u, 0xf0f00000u, 0x88880000u, 0xcccc0000u, 0xaaaa0000u, 0xffff0000u,
        0x80008000u, 0xc000c000u, 0xa000a000u, 0xf000f000u, 0x88008800u, 0xcc00cc00u, 0xaa00aa00u, 0xff00ff00u,
        0x80808080u, 0xc0c0c0c0u, 0xa0a0a0a0u, 0xf0f0f0f0u, 0x88888888u, 0xccccccccu, 0xaaaaaaaau, 0xffffffffu};
    return kDirections[bit];
}
PT_HD uint32_t sobol_dim1(uint32_t index)                                                               // sampling.rs:24-30
{
    uint32_t x = 0;
    for (int bit = 0; bit < 32 && (index >> bit) != 0u; ++bit)
        if ((index >> bit) & 1u) x ^= sobol_direction(bit);
    return x;
}
PT_HD uint32_t lk_hash(uint32_t x, uint32_t seed)                                                       // sampling.rs:53-68
{
    x ^= x * 0x3d20adeau;
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
PT_HD uint32_t owen_scramble(uint32_t x, uint32_t seed) { return rev32(lk_hash(rev32(x), seed)); }       // sampling.rs:71
PT_HD uint32_t low_bias32(uint32_t x)                                                                   // sampling.rs:76-91
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0xd35a2d97u;
    x ^= x >> 15;
    return x;
}
PT_HD void ss_sobol(uint32_t n_points, uint32_t index, uint32_t seed, float* px, float* py)              // sampling.rs:97-114
{
    uint32_t xs = low_bias32(seed), ys = low_bias32(seed + 1u), ss = low_bias32(seed + 2u);
    uint32_t slot = owen_scramble(index, ss) % n_points;
    uint32_t x = owen_scramble(rev32(slot), xs);       // table entry .x = reverse_bits(i)   sampling.rs:41
    uint32_t y = owen_scramble(sobol_dim1(slot), ys);  // table entry .y = sobol(i)          sampling.rs:42
    *px = (float)x / 4294967296.0f;
    *py = (float)y / 4294967296.0f;
}

} // namespace pt
