// Microbenchmark of k_shade_surface's memory ACCESS PATTERN: the same bytes per shaded hit — 48-byte hit record in,
// 64-byte path record in and out, 32-byte continuation ray and (70 % of the lanes) 32-byte shadow ray out — with the path record
//   mode 0: addressed by PATH ID (today: DPathRec[pid], pids in the order a real bounce's shade queue holds them),
//   mode 1: addressed by QUEUE SLOT (what "the state travels with the ray" would give the shading pass: everything linear).
// Same launch shape as the real pass (persistent workgroups of 256 threads, 12 per CU asked for, four waves per SIMD enforced through
// the LDS footprint), `alu` dependent FMAs per hit standing in for the ~1700 VALU instructions of the real kernel.
// Built and driven by tools/shade_access_bench.py; not part of libptmi.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float nt4_t __attribute__((ext_vector_type(4)));
struct f4 { float x, y, z, w; };
__device__ __forceinline__ f4 nt_load(const f4* p) { const nt4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt4_t*>(p)); return f4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void nt_store(f4* p, f4 v) { __builtin_nontemporal_store(nt4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4_t*>(p)); }

template <int MODE>
__global__ void __launch_bounds__(256) k_access(const f4* __restrict__ qa, const f4* __restrict__ qb, const f4* __restrict__ qc, f4* __restrict__ rec,
                                                f4* __restrict__ ray_a, f4* __restrict__ ray_b, f4* __restrict__ sh_a, f4* __restrict__ sh_b, uint32_t n, int alu)
{
    extern __shared__ uint32_t occupancy_ballast[];  // 40 KB per workgroup: four workgroups (= four waves per SIMD) per CU, as the real kernel's registers allow
    if (threadIdx.x == 0xffffu) occupancy_ballast[0] = 1u;
    const uint32_t tiles = (n + 255u) / 256u;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const uint32_t i = t * 256u + threadIdx.x;
        if (i >= n) continue;
        const f4 a = nt_load(qa + i), b = nt_load(qb + i), c = nt_load(qc + i);
        const uint32_t pid = __float_as_uint(a.w);
        if (pid == 0xffffffffu) continue;                     // a hole
        f4* r = rec + 4u * (size_t)(MODE == 0 ? pid : i);
        f4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        float x = a.x + b.x + c.x + r0.x + r1.y + r2.z + r3.w;
        for (int k = 0; k < alu; ++k) x = __builtin_fmaf(x, 0.999f, b.y);   // dependent chain: stands in for the shading arithmetic
        r0.x = x; r1.y = x; r2.z = x; r3.w = x;
        r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
        nt_store(ray_a + i, f4{x, b.y, b.z, c.x});
        nt_store(ray_b + i, f4{a.x, a.y, a.z, a.w});
        if ((i * 2654435761u) >> 24 < 179u) { nt_store(sh_a + i, f4{x, c.y, c.z, 1.0f}); nt_store(sh_b + i, f4{a.y, a.z, a.x, a.w}); }
    }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// pids: n entries (0xffffffff = hole) all < n_rec.  Returns 0 and the median of `reps` timed launches in ms_out[mode] for mode 0 and 1.
extern "C" int shade_access_run(const uint32_t* pids, uint32_t n, uint32_t n_rec, int alu, int reps, float* ms_out, double* bytes_out)
{
    int dev = 0, cus = 0;
    CHK(hipGetDevice(&dev));
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    f4 *qa, *qb, *qc, *rec, *ra, *rb, *sa, *sb;
    const size_t qbytes = (size_t)n * 16, rbytes = (size_t)std::max(n_rec, n) * 64;
    CHK(hipMalloc(&qa, qbytes)); CHK(hipMalloc(&qb, qbytes)); CHK(hipMalloc(&qc, qbytes)); CHK(hipMalloc(&rec, rbytes));
    CHK(hipMalloc(&ra, qbytes)); CHK(hipMalloc(&rb, qbytes)); CHK(hipMalloc(&sa, qbytes)); CHK(hipMalloc(&sb, qbytes));
    CHK(hipMemset(qa, 0, qbytes)); CHK(hipMemset(qb, 0, qbytes)); CHK(hipMemset(qc, 0, qbytes)); CHK(hipMemset(rec, 0, rbytes));
    CHK(hipMemcpy2D(reinterpret_cast<uint8_t*>(qa) + 12, 16, pids, 4, 4, n, hipMemcpyHostToDevice));
    uint64_t live = 0;
    for (uint32_t i = 0; i < n; ++i) live += pids[i] != 0xffffffffu;
    *bytes_out = (double)n * 48 + (double)live * (64 + 64 + 32 + 0.7 * 32);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const dim3 grid((unsigned)cus * 12u), block(256);
    for (int mode = 0; mode < 2; ++mode)
    {
        float best[16];
        for (int r = 0; r < reps + 1 && r < 16; ++r)
        {
            CHK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_access<0>, grid, block, 40 * 1024, 0, qa, qb, qc, rec, ra, rb, sa, sb, n, alu);
            else hipLaunchKernelGGL(k_access<1>, grid, block, 40 * 1024, 0, qa, qb, qc, rec, ra, rb, sa, sb, n, alu);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&best[r], e0, e1));
        }
        // median of the timed launches (the first is a warm-up)
        for (int x = 1; x <= reps; ++x) for (int y = x + 1; y <= reps; ++y) if (best[y] < best[x]) { float tmp = best[x]; best[x] = best[y]; best[y] = tmp; }
        ms_out[mode] = best[1 + (reps - 1) / 2];
    }
    (void)hipFree(qa); (void)hipFree(qb); (void)hipFree(qc); (void)hipFree(rec); (void)hipFree(ra); (void)hipFree(rb); (void)hipFree(sa); (void)hipFree(sb);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}
