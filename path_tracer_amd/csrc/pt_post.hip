// The step after the path: what the reference's State::update / State::render do to a finished frame
// (src/state.rs:505-586, 629-667) as gfx950 kernels working on the framebuffers that are already in HBM.
//
//   k_post_accumulate   src/shaders/accumulate.wgsl:20-23     static camera: accumulation += (rgb, 1)
//   k_post_velocity     src/shaders/velocity.wgsl:16-39       per-pixel motion vector from the first-hit position
//   k_post_reproject    src/shaders/compute.wgsl:103-212      3x3 YCoCg variance clip, Catmull-Rom history, id disocclusion, 15 % blend
//   k_post_tonemap      src/shaders/shader.wgsl:3-33,59-64    Uchimura "GT" curve on accumulation.rgb / accumulation.w
//   k_post_rgb8         src/image_helper.rs:41-48 + src/image_helper/tonemapping.rs   the 8-bit gamma-2.2 image write_image saves
//
// WGSL leaves bilinear filter weights, mat*vec summation order, pow/exp precision and out-of-range casts to the GPU.  Here
// they are exact binary32 operations in the written order, pow(x, c) = exp(c ln x) with pt_math.h's routines, saturating
// casts, out-of-bounds textureLoad = 0 — the same definitions the oracle uses, so the two agree bit for bit.
#include <hip/hip_runtime.h>

#include "pt_kernels.h"

namespace pt {
namespace {

struct Tex
{
    const f4* p;
    int w, h;
    __device__ __forceinline__ f4 at(int x, int y) const { return p[(size_t)y * w + x]; }
};
__device__ __forceinline__ f4 add4(f4 a, f4 b) { return f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ f4 mul4(f4 a, float s) { return f4{a.x * s, a.y * s, a.z * s, a.w * s}; }
__device__ __forceinline__ int to_i32_sat(float f)
{
    return isnan_f(f) ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (-2147483647 - 1) : (int)f));
}
__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

// textureSampleLevel(tex, sampler{linear, clamp-to-edge}, uv, 0)  (sampler: src/state.rs:169-178)
__device__ f4 bilinear(const Tex& t, float u, float v)
{
    const float x = u * (float)t.w - 0.5f, y = v * (float)t.h - 0.5f;
    const float bx = floorf(x), by = floorf(y);
    const float fx = x - bx, fy = y - by;
    const int x0 = to_i32_sat(bx), y0 = to_i32_sat(by);
    const int xa = clampi(x0, t.w), xb = clampi(x0 < 2147483647 ? x0 + 1 : x0, t.w);
    const int ya = clampi(y0, t.h), yb = clampi(y0 < 2147483647 ? y0 + 1 : y0, t.h);
    const f4 top = add4(mul4(t.at(xa, ya), 1.0f - fx), mul4(t.at(xb, ya), fx));
    const f4 bot = add4(mul4(t.at(xa, yb), 1.0f - fx), mul4(t.at(xb, yb), fx));
    return add4(mul4(top, 1.0f - fy), mul4(bot, fy));
}
__device__ __forceinline__ f3 w_divide(f4 v) // v.xyz / max(v.w, 1.0)
{
    const float d = v.w > 1.0f ? v.w : 1.0f;
    return f3{v.x / d, v.y / d, v.z / d};
}
__device__ __forceinline__ f3 to_ycocg(f3 c) // compute.wgsl:64-71 (mat3x3 columns times vector)
{
    return f3{(0.25f * c.x + 0.5f * c.y) + -0.25f * c.z, (0.5f * c.x + 0.0f * c.y) + 0.5f * c.z, (0.25f * c.x + -0.5f * c.y) + -0.25f * c.z};
}
__device__ __forceinline__ f3 from_ycocg(f3 c) // compute.wgsl:73-80
{
    return f3{(1.0f * c.x + 1.0f * c.y) + 1.0f * c.z, (1.0f * c.x + 0.0f * c.y) + -1.0f * c.z, (-1.0f * c.x + 1.0f * c.y) + -1.0f * c.z};
}
__device__ __forceinline__ float maxf_w(float a, float b) { return a > b ? a : b; }
__device__ f3 clip_to_box(f3 mn, f3 mx, f3 q) // clip_aabb  compute.wgsl:82-101
{
    const f3 centre = 0.5f * (mx + mn), extent = 0.5f * (mx - mn);
    const f3 v = q - centre;
    const f3 unit{v.x / extent.x, v.y / extent.y, v.z / extent.z};
    const float m = maxf_w(fabsf(unit.x), maxf_w(fabsf(unit.y), fabsf(unit.z)));
    if (m > 1.0f) return centre + v / m;
    return q;
}
__device__ f3 catmull_rom(const Tex& tex, float ux, float uy) // sample_catmull_rom  compute.wgsl:16-62
{
    const float sx = (float)tex.w, sy = (float)tex.h;
    const float px = ux * sx + 0.5f, py = uy * sy + 0.5f;
    const float t1x = floorf(px - 0.5f) + 0.5f, t1y = floorf(py - 0.5f) + 0.5f;
    const float fx = px - t1x, fy = py - t1y;
    const float w0x = fx * (-0.5f + fx * (1.0f - 0.5f * fx)), w0y = fy * (-0.5f + fy * (1.0f - 0.5f * fy));
    const float w1x = 1.0f + fx * fx * (-2.5f + 1.5f * fx), w1y = 1.0f + fy * fy * (-2.5f + 1.5f * fy);
    const float w2x = fx * (0.5f + fx * (2.0f - 1.5f * fx)), w2y = fy * (0.5f + fy * (2.0f - 1.5f * fy));
    const float w3x = fx * fx * (-0.5f + 0.5f * fx), w3y = fy * fy * (-0.5f + 0.5f * fy);
    const float w12x = w1x + w2x, w12y = w1y + w2y;
    const float o12x = w2x / (w1x + w2x), o12y = w2y / (w1y + w2y);
    const float p0x = (t1x - 1.0f) / sx, p0y = (t1y - 1.0f) / sy, p3x = (t1x + 2.0f) / sx, p3y = (t1y + 2.0f) / sy;
    const float p12x = (t1x + o12x) / sx, p12y = (t1y + o12y) / sy;
    const float us[3] = {p0x, p12x, p3x}, vs[3] = {p0y, p12y, p3y}, wu[3] = {w0x, w12x, w3x}, wv[3] = {w0y, w12y, w3y};
    f3 c{0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) c = c + w_divide(bilinear(tex, us[i], vs[j])) * wu[i] * wv[j];
    return c;
}
__device__ __forceinline__ float pow_det(float x, float c) { return exp_det(c * ln_det(x)); }
__device__ float gt_curve(float x, float p, float a, float m, float l, float c, float b) // gt_tonemap  shader.wgsl:3-33
{
    const float l0 = (p - m) * l / a;
    const float t = clamp_rs((x - 0.0f) / (m - 0.0f), 0.0f, 1.0f); // smoothstep(0, m, x)
    const float w0 = 1.0f - t * t * (3.0f - 2.0f * t);
    const float w2 = x >= m + l0 ? 1.0f : 0.0f;                    // step(m + l0, x)
    const float w1 = 1.0f - w0 - w2;
    const float toe = m * pow_det(x / m, c) + b;
    const float lin = m + a * (x - m);
    const float s0 = m + l0, s1 = m + a * l0, c2 = a * p / (p - s1);
    const float shoulder = p - (p - s1) * exp_det(-c2 * (x - s0) / p);
    const float r = (toe * w0 + lin * w1) + shoulder * w2;
    return maxf_w(r, 0.0f);
}

// tonemapping.rs:1-96 — the CPU-side curve differs from shader.wgsl in its branches (x < 0 -> b, branchy smoothstep, and the
// shoulder weight gt_lerp(x, e, e) which is 0/0 = NaN at x == e exactly); restated as written.
__device__ float gt_curve_rs(float x, float p, float a, float m, float l, float c, float b)
{
    if (x < 0.0f) return b;
    const float l0 = (p - m) * l / a;
    float sm;                                                      // gt_smoothstep(x, 0, m)  :36-52
    if (x < 0.0f) sm = 0.0f;
    else if (x > m) sm = 1.0f;
    else { const float q = (x - 0.0f) / (m - 0.0f); const float r = 3.0f - 2.0f * q; sm = q * q * r; }
    const float w0 = 1.0f - sm;
    const float e = m + l0;
    float w2;                                                      // gt_lerp(x, e, e)  :20-34
    if (x < e) w2 = 0.0f;
    else if (x > e) w2 = 1.0f;
    else w2 = (x - e) / (e - e);
    const float w1 = 1.0f - w0 - w2;
    const float t = (m * pow_det(x / m, c) + b) * w0;              // gt_toe
    const float u = (m + a * (x - m)) * w1;                        // gt_linear
    const float s0 = m + l0, s1 = m + a * l0, c2 = a * p / (p - s1);
    const float v = (p - (p - s1) * exp_det(-c2 * (x - s0) / p)) * w2; // gt_shoulder
    return t + u + v;
}
__device__ __forceinline__ uint32_t to_u8_sat(float f) { return isnan_f(f) ? 0u : (f >= 255.0f ? 255u : (f <= 0.0f ? 0u : (uint32_t)f)); } // Rust `as u8`

__global__ void __launch_bounds__(256) k_post_accumulate(uint32_t n, const f4* __restrict__ input, f4* __restrict__ accum)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4 a = accum[i], c = input[i];
    accum[i] = f4{a.x + c.x, a.y + c.y, a.z + c.z, a.w + 1.0f};
}

struct Mat4 { float m[16]; };

__global__ void __launch_bounds__(256) k_post_velocity(int w, int h, const f4* __restrict__ position, const Mat4 M, float2* __restrict__ velocity)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int x = i % w, y = i / w;
    const f4 P = position[i];
    const float cu = ((float)x + 0.5f) / (float)w, cv = ((float)y + 0.5f) / (float)h;
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = ((M.m[k] * P.x + M.m[4 + k] * P.y) + M.m[8 + k] * P.z) + M.m[12 + k] * 1.0f;
    const f3 d = w_divide(f4{r[0], r[1], r[2], r[3]});
    velocity[i] = make_float2(cu - (d.x * 0.5f + 0.5f), cv - (d.y * 0.5f + 0.5f));
}

__global__ void __launch_bounds__(256) k_post_reproject(int w, int h, const f4* __restrict__ input, const f4* __restrict__ accum,
                                                         const float2* __restrict__ velocity, const uint32_t* __restrict__ id, f4* __restrict__ output)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int cx = i % w, cy = i / w;
    const Tex in{input, w, h}, acc{accum, w, h};
    const f4 cur4 = in.at(cx, cy);
    const f3 current{cur4.x, cur4.y, cur4.z};
    f3 m1{0.0f, 0.0f, 0.0f}, m2{0.0f, 0.0f, 0.0f};
    float closest_depth = 1e20f;
    int vx = 0, vy = 0;
    const int x0 = cx - 1 > 0 ? cx - 1 : 0, y0 = cy - 1 > 0 ? cy - 1 : 0;
    const int x1 = cx + 1 < w - 1 ? cx + 1 : w - 1, y1 = cy + 1 < h - 1 ? cy + 1 : h - 1;
    const int n = (x1 + 1 - x0) * (y1 + 1 - y0);
    for (int x = x0; x <= x1; ++x)
        for (int y = y0; y <= y1; ++y)
        {
            const f4 dd = in.at(x, y);
            const f3 d = to_ycocg(f3{dd.x, dd.y, dd.z});
            m1 = m1 + d;
            m2 = m2 + d * d;
            if (dd.w < closest_depth) { closest_depth = dd.w; vx = x; vy = y; }
        }
    const float dx = (float)w, dy = (float)h;
    const float cu = ((float)cx + 0.5f) / dx, cv = ((float)cy + 0.5f) / dy;
    const float2 vel = velocity[(size_t)vy * w + vx];
    const float pu = cu - vel.x, pv = cv - vel.y;
    const int px = to_i32_sat(floorf(pu * dx)), py = to_i32_sat(floorf(pv * dy));
    const bool oob = px < 0 || py < 0 || px >= w || py >= h;
    const uint32_t current_id = id[i] & 0xffffu;
    const uint32_t old_id = oob ? 0u : ((id[(size_t)py * w + px] >> 16) & 0xffffu);
    if (current_id != old_id || oob)
    {
        // disocclusion: average of four bilinear taps at the texel corners  compute.wgsl:170-181
        const float c0x = (float)cx / dx, c0y = (float)cy / dy, c1x = c0x + 1.0f / dx, c1y = c0y + 1.0f / dy;
        const f4 s = add4(add4(add4(bilinear(in, c0x, c0y), bilinear(in, c0x, c1y)), bilinear(in, c1x, c0y)), bilinear(in, c1x, c1y));
        output[i] = f4{s.x / 4.0f, s.y / 4.0f, s.z / 4.0f, s.w / 4.0f};
        return;
    }
    const float fn = (float)n;
    const f3 mu = m1 / fn;
    const f3 var = m2 / fn - mu * mu;
    const f3 sigma{sqrtf(var.x), sqrtf(var.y), sqrtf(var.z)};
    const f3 mn = mu - 1.0f * sigma, mx = mu + 1.0f * sigma;
    const f3 prev = catmull_rom(acc, pu, pv);
    const f3 clamped = from_ycocg(clip_to_box(mn, mx, to_ycocg(prev)));
    const f3 o = clamped * (1.0f - 0.15f) + current * 0.15f; // mix(clamped, current, 0.15)
    output[i] = f4{o.x, o.y, o.z, 1.0f};
}

__global__ void __launch_bounds__(256) k_post_tonemap(uint32_t n, const f4* __restrict__ accum, f4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4 a = accum[i];
    out[i] = f4{gt_curve(a.x / a.w, 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f), gt_curve(a.y / a.w, 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f),
                gt_curve(a.z / a.w, 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f), 1.0f};
}

// one thread per pixel writes 3 bytes; a wave's 192 bytes are contiguous
__global__ void __launch_bounds__(256) k_post_rgb8(uint32_t n, const f4* __restrict__ accum, uint8_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f4 a = accum[i];
    const float g = 1.0f / 2.2f;
    const float ch[3] = {a.x / a.w, a.y / a.w, a.z / a.w};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[(size_t)i * 3u + k] = (uint8_t)to_u8_sat(pow_det(gt_curve_rs(ch[k], 1.0f, 1.0f, 0.22f, 0.4f, 1.33f, 0.0f), g) * 255.0f);
}

// strips gathered from `world` ranks (rank r's rows, padded to pad_rows, at parts + r * pad_rows * w) -> the whole frame, row-major:
// global row y belongs to rank (y / strip) % world and is that rank's local row (y / strip / world) * strip + y % strip  (compute_rows)
__global__ void __launch_bounds__(256) k_post_deinterleave(uint32_t w, uint32_t h, uint32_t world, uint32_t strip, uint32_t pad_rows,
                                                            const f4* __restrict__ parts, f4* __restrict__ full)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t y = i / w, x = i - y * w;
    const uint32_t st = y / strip, rank = st % world, ly = (st / world) * strip + (y - st * strip);
    full[i] = parts[((size_t)rank * pad_rows + ly) * w + x];
}

} // namespace

void launch_post_deinterleave(hipStream_t s, uint32_t w, uint32_t h, uint32_t world, uint32_t strip, uint32_t pad_rows, const f4* parts, f4* full)
{
    hipLaunchKernelGGL(k_post_deinterleave, dim3((w * h + 255u) / 256u), dim3(256), 0, s, w, h, world, strip, pad_rows, parts, full);
}
void launch_post_rgb8(hipStream_t s, uint32_t n, const f4* accum, uint8_t* out)
{
    hipLaunchKernelGGL(k_post_rgb8, dim3((n + 255u) / 256u), dim3(256), 0, s, n, accum, out);
}
void launch_post_accumulate(hipStream_t s, uint32_t n, const f4* input, f4* accum)
{
    hipLaunchKernelGGL(k_post_accumulate, dim3((n + 255u) / 256u), dim3(256), 0, s, n, input, accum);
}
void launch_post_velocity(hipStream_t s, int w, int h, const f4* position, const float* m16, float* velocity_xy)
{
    Mat4 M;
    for (int i = 0; i < 16; ++i) M.m[i] = m16[i];
    hipLaunchKernelGGL(k_post_velocity, dim3((w * h + 255) / 256), dim3(256), 0, s, w, h, position, M, reinterpret_cast<float2*>(velocity_xy));
}
void launch_post_reproject(hipStream_t s, int w, int h, const f4* input, const f4* accum, const float* velocity_xy, const uint32_t* id, f4* output)
{
    hipLaunchKernelGGL(k_post_reproject, dim3((w * h + 255) / 256), dim3(256), 0, s, w, h, input, accum, reinterpret_cast<const float2*>(velocity_xy), id, output);
}
void launch_post_tonemap(hipStream_t s, uint32_t n, const f4* accum, f4* out)
{
    hipLaunchKernelGGL(k_post_tonemap, dim3((n + 255u) / 256u), dim3(256), 0, s, n, accum, out);
}

} // namespace pt
