"""hipMalloc / hipFree wall time against size on the GPU box (why the wavefront pools are kept, not resized: see DESIGN section 6, cold start)."""
import ctypes as C, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
p = C.c_void_p()
hip.hipMalloc(C.byref(p), C.c_size_t(1 << 20)); hip.hipFree(p)
for gib in (1, 8, 32, 64, 128, 200):
    for rep in range(2):
        t0 = time.perf_counter(); r = hip.hipMalloc(C.byref(p), C.c_size_t(gib << 30)); t1 = time.perf_counter()
        hip.hipFree(p); t2 = time.perf_counter()
        print(f"{gib} GiB: hipMalloc rc {r} {1e3*(t1-t0):.1f} ms, hipFree {1e3*(t2-t1):.1f} ms", flush=True)
# many pieces
t0 = time.perf_counter(); ps = []
for i in range(32):
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), C.c_size_t(4 << 30)); ps.append(q)
t1 = time.perf_counter()
for q in ps: hip.hipFree(q)
print(f"32 x 4 GiB: {1e3*(t1-t0):.1f} ms, free {1e3*(time.perf_counter()-t1):.1f} ms")
