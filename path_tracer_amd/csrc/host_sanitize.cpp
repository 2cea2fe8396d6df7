// CPU-only driver of libptmi's HOST code (pt_scene.cpp: OBJ reader, SAH / agglomerative builders, flattening, camera; pt_png.cpp: PNG
// encoder) for the sanitizer build `make -C path_tracer_amd/csrc host-asan` (g++ -fsanitize=address,undefined; no HIP involved: GPU
// ASan does not exist on this pool, and these files are where user-supplied bytes enter the library).  tests/test_host_sanitizer.py
// feeds it valid, malformed and byte-mutated OBJ files and a camera random walk; any sanitizer report fails the test.
//   host_sanitize obj <file>...        Model::new(path) + Scene::new for each file (parse errors are an expected outcome)
//   host_sanitize walk <events> <seed> Camera::input random walk with create_ray after every event
//   host_sanitize png <w> <h> <file>   encode a test image
//   host_sanitize soup <n> <seed>      Scene::new over n random triangles (PTMI_BUILD_THREADS forks the SAH sweep: also built with
//                                      -fsanitize=thread by `make host-tsan`); prints a checksum of the BLAS arena
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_png.h"
#include "pt_scene.h"

using namespace pt;

static const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

static int light_scene(HostScene& sc)
{
    const float white[3] = {0.7f, 0.7f, 0.7f}, emit[3] = {10.0f, 10.0f, 10.0f}, zero[3] = {0, 0, 0};
    const int m0 = sc.add_material(0, white, 0.0f, 1.0f, false, zero, 0, 0, 0);
    const int m1 = sc.add_material(1, emit, 0.0f, 1.0f, false, zero, 0, 0, 0);
    const float quad[18] = {-50, 300, -50, 50, 300, -50, 50, 300, 50, -50, 300, -50, 50, 300, 50, -50, 300, 50};
    const float nrm[18] = {0, -1, 0, 0, -1, 0, 0, -1, 0, 0, -1, 0, 0, -1, 0, 0, -1, 0};
    if (sc.add_model(quad, nrm, 2, m1, kIdentity, 1) < 0) return -1;
    return m0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "obj")
    {
        int parsed = 0, rejected = 0;
        for (int i = 2; i < argc; ++i)
        {
            HostScene sc;
            const int mat = light_scene(sc);
            if (mat < 0) return 3;
            std::string err;
            const int r = sc.add_model_obj(argv[i], mat, kIdentity, 1, &err);
            if (r < 0) { ++rejected; continue; }
            if (sc.build(&err) != 0) { ++rejected; continue; }
            const float eye[3] = {0, 50, 1000}, tgt[3] = {0, 50, 0};
            sc.set_camera(eye, tgt, 60.0f, 1.5f);
            float o[3], d[3];
            sc.create_ray(0.25f, 0.75f, o, d);
            ++parsed;
        }
        std::printf("{\"parsed\": %d, \"rejected\": %d}\n", parsed, rejected);
        return 0;
    }
    if (cmd == "walk" && argc >= 4)
    {
        HostScene sc;
        const float eye[3] = {0, 50, 1000}, tgt[3] = {0, 50, 0};
        sc.set_camera(eye, tgt, 60.0f, 16.0f / 9.0f);
        uint64_t s = std::strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) & 0xffff) / 65535.0f; };
        double acc = 0;
        for (int e = 0, n = std::atoi(argv[2]); e < n; ++e)
        {
            const float dt = 1e-6f + rnd() * 5e-6f;
            switch ((int)(rnd() * 5.0f) % 5)
            {
            case 0: sc.camera_rotate(rnd() * 8.0f - 4.0f, rnd() * 8.0f - 4.0f, dt); break;
            case 1: sc.camera_move(0.0f, 1.0f, dt); break;
            case 2: sc.camera_move(0.0f, -1.0f, dt); break;
            case 3: sc.camera_move(-1.0f, 0.0f, dt); break;
            default: sc.camera_move(1.0f, 0.0f, dt); break;
            }
            float o[3], d[3], m[16];
            sc.create_ray(rnd(), rnd(), o, d);
            sc.inv_projection(m);
            acc += d[0] + d[1] + d[2] + m[5];
        }
        std::printf("{\"checksum\": %.6f}\n", acc);
        return 0;
    }
    if (cmd == "soup" && argc >= 4)
    {
        HostScene sc;
        const int mat = light_scene(sc);
        if (mat < 0) return 3;
        const uint32_t n = (uint32_t)std::atoi(argv[2]);
        uint64_t s = std::strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) & 0xffff) / 65535.0f; };
        std::vector<float> p((size_t)n * 9), nr((size_t)n * 9, 0.0f);
        for (uint32_t i = 0; i < n; ++i)
        {
            const float c[3] = {rnd() * 200.0f - 100.0f, rnd() * 200.0f - 100.0f, rnd() * 200.0f - 100.0f};
            for (int k = 0; k < 9; ++k) p[(size_t)i * 9 + k] = c[k % 3] + rnd() * 4.0f - 2.0f;
            for (int k = 0; k < 3; ++k) nr[(size_t)i * 9 + k * 3 + 1] = 1.0f;
        }
        if (sc.add_model(p.data(), nr.data(), n, mat, kIdentity, 1) < 0) return 4;
        std::string err;
        if (sc.build(&err) != 0) { std::printf("{\"error\": \"%s\"}\n", err.c_str()); return 0; }
        const HostBlas& b = sc.blas.back();
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint32_t v) { h = (h ^ v) * 1099511628211ull; };
        for (const HostNode& nd : b.nodes)
        {
            uint32_t w[6];
            std::memcpy(w, &nd.box, sizeof(w));
            for (uint32_t v : w) mix(v);
            mix(nd.kind); mix(nd.a); mix(nd.b);
        }
        for (uint32_t v : b.prim_ids) mix(v);
        std::printf("{\"nodes\": %zu, \"depth\": %u, \"arena\": \"%016llx\"}\n", b.nodes.size(), b.depth, (unsigned long long)h);
        return 0;
    }
    if (cmd == "png" && argc >= 5)
    {
        const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]);
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)((i * 2654435761u) >> 24);
        std::string err;
        if (!write_png_rgb8(argv[4], rgb.data(), w, h, &err)) { std::printf("{\"error\": \"%s\"}\n", err.c_str()); return 0; }
        std::printf("{\"bytes\": %zu}\n", rgb.size());
        return 0;
    }
    return 2;
}
