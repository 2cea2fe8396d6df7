// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
// shipped product; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link or call it.
//
// PARITY UNPINNED: the reference (CouncilmanJeremyJamm/path_tracer, Rust) has
// no tests, golden vectors or fixtures, cannot be compiled here (no rustc, 211
// crates, assets absent) and is entropy-seeded, so nothing in it pins outputs.
// This file restates, from the reference's source text, the arithmetic of the
// third-party crates the hot path relies on:
//   * glam 0.23.0 (Cargo.lock), SSE2 backend: Vec3A/Vec4/Mat3A/Affine3A/Mat4
//   * nanorand 0.7.0 WyRand
//   * Rust core f32 intrinsics (signum, min, clamp, powi, total_cmp, mul_add)
// The crate sources are NOT under /root/reference; their semantics below are
// the published algorithms as called from the reference's own call sites.
//
// Rules that make CPU == GPU bit-exact achievable:
//   * every op is an explicit IEEE-754 binary32 op in a fixed order;
//   * fused multiply-add ONLY where the reference calls mul_add
//     (ray.rs:20, integrator.rs:211, material.rs:205,487);
//   * compile with -ffp-contract=off, no -ffast-math;
//   * sin/cos/tan/exp/ln are deterministic polynomial routines defined here
//     (the reference uses the system libm, which no GPU can reproduce);
//     hypot follows glibc's hypotf: sqrt in binary64 of the exact squares.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pto {

// ---------------------------------------------------------------- scalars
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// SSE minps/maxps lane semantics, which glam's Vec3A::min/max inherit:
// min(a,b) = a < b ? a : b  (returns b when either is NaN, and b on +-0 ties).
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline float sse_max(float a, float b) { return a > b ? a : b; }

// Rust f32::min / f32::max (IEEE minNum/maxNum): NaN loses.
static inline float rs_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
static inline float rs_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
// Rust f32::clamp
static inline float rs_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// `a.signum() != b.signum()` as used at primitive.rs:122,131,138:
// signum(+0)=+1, signum(-0)=-1, signum(NaN)=NaN and NaN != anything.
static inline bool signum_differs(float a, float b)
{
    if (a != a || b != b) return true;
    return (f2u(a) >> 31) != (f2u(b) >> 31);
}
static inline float rs_signum(float a) { return (a != a) ? a : u2f((f2u(a) & 0x80000000u) | 0x3f800000u); }

// f32::total_cmp key: monotone map of the bit pattern to a signed integer.
static inline int32_t total_key(float f)
{
    int32_t i = (int32_t)f2u(f);
    return i ^ (int32_t)(((uint32_t)(i >> 31)) >> 1);
}

// f32::powi with constant exponents, as LLVM expands it (binary decomposition).
static inline float powi2(float x) { return x * x; }
static inline float powi5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }
static inline float powi_m2(float x) { return 1.0f / (x * x); }

static inline float mul_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------- deterministic libm stand-ins
// sin/cos by 3-term Cody-Waite reduction to [-pi/4, pi/4] and Cephes-style
// minimax polynomials; every step is an explicit IEEE op, so the same text
// gives the same bits on x86 and on gfx950.  Valid for |x| < 2^20.
static inline void det_sincos(float x, float* s, float* c)
{
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float MAGIC = 12582912.0f; // 1.5 * 2^23: add/sub rounds to nearest integer
    float kf = (x * TWO_OVER_PI + MAGIC) - MAGIC;
    int q = (int)kf;
    float r = __builtin_fmaf(kf, -1.57073974609375f, x);
    r = __builtin_fmaf(kf, -5.657970905303955078125e-05f, r);
    r = __builtin_fmaf(kf, -9.920936294705029468e-10f, r);
    float r2 = r * r;
    // sin(r) = r + r*r2*(S1 + r2*(S2 + r2*S3))
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    float sr = __builtin_fmaf(ps * r2, r, r);
    // cos(r) = 1 - r2/2 + r2*r2*(C1 + r2*(C2 + r2*C3))
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
    float sv = (q & 1) ? cr : sr;
    float cv = (q & 1) ? sr : cr;
    if (q & 2) sv = -sv;
    if ((q + 1) & 2) cv = -cv;
    *s = sv;
    *c = cv;
}
static inline float det_tan(float x) { float s, c; det_sincos(x, &s, &c); return s / c; }

// exp(x): x = k*ln2 + r, degree-6 polynomial, scale by 2^k via exponent bits.
static inline float det_exp(float x)
{
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.0f) return 0.0f;
    const float MAGIC = 12582912.0f;
    float kf = (x * 1.44269504088896341f + MAGIC) - MAGIC;
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
    // 2^k in two steps so that subnormal results stay defined
    int k1 = k / 2, k2 = k - k1;
    return e * u2f((uint32_t)(k1 + 127) << 23) * u2f((uint32_t)(k2 + 127) << 23);
}

// ln(x) for finite x > 0 (Cephes logf layout); ln(0) = -inf, ln(<0) = NaN.
static inline float det_ln(float x)
{
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t u = f2u(x);
    int e = 0;
    if (u < 0x00800000u) { x = x * 8388608.0f; u = f2u(x); e = -23; }
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u); // [0.5, 1)
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

// atan / atan2 / asin (Cephes atanf / asinf layout, explicit fma): used by the equirect env lookup, integrator.rs:258-259
static inline float det_atan_pos(float x) // x >= 0
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
static inline float det_atan2(float y, float x)
{
    if (x != x || y != y) return NAN;
    const float PI_ = 3.14159265358979323846f, PIO2_ = 1.5707963267948966f;
    if (x == 0.0f)
    {
        if (y == 0.0f) return (f2u(x) >> 31) ? ((f2u(y) >> 31) ? -PI_ : PI_) : y;
        return (f2u(y) >> 31) ? -PIO2_ : PIO2_;
    }
    float a = det_atan_pos(std::fabs(y / x));
    if (f2u(x) >> 31) a = PI_ - a;
    return (f2u(y) >> 31) ? -a : a;
}
static inline float det_asin(float x)
{
    float a = std::fabs(x);
    if (!(a <= 1.0f)) return NAN;
    if (a < 1.0e-4f) return x;
    bool big = a > 0.5f;
    float z, t;
    if (big) { z = 0.5f * (1.0f - a); t = std::sqrt(z); }
    else { t = a; z = t * t; }
    float p = __builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    float r = __builtin_fmaf(p * z, t, t);
    if (big) { r = r + r; r = 1.5707963267948966f - r; }
    return (f2u(x) >> 31) ? -r : r;
}

// glibc hypotf: sqrt of the exact binary64 sum of squares, rounded once to binary32.
static inline float det_hypot(float a, float b)
{
    double da = (double)a, db = (double)b;
    return (float)std::sqrt(da * da + db * db);
}

// ---------------------------------------------------------------- Vec3A
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 splat(float a) { return V3{a, a, a}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 vmin(V3 a, V3 b) { return V3{sse_min(a.x, b.x), sse_min(a.y, b.y), sse_min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return V3{sse_max(a.x, b.x), sse_max(a.y, b.y), sse_max(a.z, b.z)}; }
static inline V3 recip(V3 a) { return V3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }
// glam sse2 dot3: (x*x' + y*y') + z*z', three roundings after three products
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glam sse2 dot4: (x*x' + z*z') + (y*y' + w*w')
static inline float dot4(V4 a, V4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }
static inline V3 cross(V3 a, V3 b)
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length_squared(V3 a) { return dot(a, a); }
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return V3{a.x / l, a.y / l, a.z / l}; }
// glam sse2 max_element/min_element shuffle order: ((x,z),(y,z)) then pair
static inline float max_element(V3 a) { float m1 = sse_max(a.x, a.z), m2 = sse_max(a.y, a.z); return sse_max(m1, m2); }
static inline float min_element(V3 a) { float m1 = sse_min(a.x, a.z), m2 = sse_min(a.y, a.z); return sse_min(m1, m2); }
static inline bool is_finite(V3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }
static inline bool is_nan(V3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
// Vec3A::mul_add(self, a, b) = self*a + b fused per lane (FMA is on: target-cpu=native)
static inline V3 vfma(V3 a, V3 b, V3 c)
{
    return V3{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z)};
}
// glam 0.23 clamp_length_max: self * (length_sq.sqrt().recip() * max)
static inline V3 clamp_length_max(V3 v, float m)
{
    float l2 = dot(v, v);
    if (l2 > m * m) { float s = (1.0f / std::sqrt(l2)) * m; return v * s; }
    return v;
}

// ---------------------------------------------------------------- Mat3A / Affine3A
struct M3 { V3 c0, c1, c2; }; // columns
static inline V3 mul(const M3& m, V3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
static inline M3 transpose(const M3& m)
{
    return M3{V3{m.c0.x, m.c1.x, m.c2.x}, V3{m.c0.y, m.c1.y, m.c2.y}, V3{m.c0.z, m.c1.z, m.c2.z}};
}
// glam Mat3A::inverse: cross products, det = z_axis . (x_axis x y_axis), multiply by 1/det, transpose
static inline M3 inverse(const M3& m)
{
    V3 t0 = cross(m.c1, m.c2), t1 = cross(m.c2, m.c0), t2 = cross(m.c0, m.c1);
    float det = dot(m.c2, t2);
    float inv = 1.0f / det;
    return transpose(M3{t0 * inv, t1 * inv, t2 * inv});
}
struct Affine { M3 m; V3 t; };
static inline V3 transform_point(const Affine& a, V3 p) { return mul(a.m, p) + a.t; }
static inline V3 transform_vector(const Affine& a, V3 v) { return mul(a.m, v); }
static inline Affine inverse(const Affine& a)
{
    M3 mi = inverse(a.m);
    return Affine{mi, -mul(mi, a.t)};
}
static inline Affine affine_identity() { return Affine{M3{V3{1, 0, 0}, V3{0, 1, 0}, V3{0, 0, 1}}, V3{0, 0, 0}}; }

// Vec3A::any_orthonormal_pair (Duff et al. 2017), onb.rs:5
static inline M3 generate_onb(V3 n)
{
    float sign = std::copysign(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    V3 c0{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
    V3 c1{b, sign + n.y * n.y * a, -n.y};
    return M3{c0, c1, n};
}

// ---------------------------------------------------------------- WyRand (nanorand 0.7.0), counter form
// next(): s += INC; t = s * (s ^ XOR) as u128; out = hi ^ lo.  The state advance is a constant add,
// so draw k (k = 0,1,..) of a stream with initial state s0 is mix(s0 + (k+1)*INC).
static const uint64_t WY_INC = 0xa0761d6478bd642fULL;
static const uint64_t WY_XOR = 0xe7037ed1a0b428dbULL;
static inline uint64_t wy_mix(uint64_t s)
{
    __uint128_t t = (__uint128_t)s * (__uint128_t)(s ^ WY_XOR);
    return (uint64_t)(t >> 64) ^ (uint64_t)t;
}
// per-(pixel, sample) stream key; pixel = y*W + x is the GLOBAL pixel index, so the
// image does not depend on tiling or GPU count.
static inline uint64_t stream_state0(uint64_t global_seed, uint32_t pixel, uint32_t sample)
{
    uint64_t z = global_seed + 0x9E3779B97F4A7C15ULL * ((((uint64_t)sample) << 32) | (uint64_t)pixel);
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
struct Rng
{
    uint64_t s0;
    uint32_t k; // draws consumed
    uint64_t next_u64() { k += 1; return wy_mix(s0 + (uint64_t)k * WY_INC); }
    // generate::<u32>() = first four native-endian bytes of the 64-bit output = low half on x86
    uint32_t next_u32() { return (uint32_t)next_u64(); }
    // generate::<f32>() = (u32 as f32) / (u32::MAX as f32); u32::MAX as f32 == 2^32; range [0,1]
    float next_f32() { return (float)next_u32() / 4294967296.0f; }
};

} // namespace pto
