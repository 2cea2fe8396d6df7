// Headless driver: src/main.rs's run() without the window — scene set-up (:75-137), then the event loop's MainEventsCleared
// body (:179-216) for a number of frames, optionally with scripted camera input, and ImageHelper::write_image at the end.
// Everything goes through include/ptmi.hpp, i.e. the C-ABI of libptmi.
//
//   examples/headless [--width W] [--height H] [--frames N] [--bounces B] [--move] [--models DIR] [--out file.png]
//   examples/headless --gpus N [--devices a,b,..] --spp S ...   the same scene, S samples per pixel on N GPUs of this process (pt_multi)
//   examples/headless --render FIRST COUNT [--load-state in.bin] [--save-state out.bin] ...   samples [FIRST, FIRST + COUNT) accumulated
//       without the temporal pass; the frame's state (accumulation, first-hit position, id history) can be saved and picked up by another
//       process: a long render stopped and continued (pt_read_frame / pt_write_accumulation)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptmi.hpp"

using namespace ptmi;

int main(int argc, char** argv)
{
    uint32_t width = 1920, height = 1080, frames = 64, bounces = 8; // IMAGE_WIDTH/HEIGHT main.rs:44-45; the reference's MAX_BOUNCES is 1024
    bool move = false;
    uint32_t gpus = 0, spp = 64;
    std::vector<int32_t> devices;
    std::string models_dir = "models/cornell", out = "", load_state = "", save_state = "";
    bool render_mode = false;
    uint32_t render_first = 0, render_count = 0;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--width") width = (uint32_t)std::atoi(next("--width"));
        else if (a == "--height") height = (uint32_t)std::atoi(next("--height"));
        else if (a == "--frames") frames = (uint32_t)std::atoi(next("--frames"));
        else if (a == "--bounces") bounces = (uint32_t)std::atoi(next("--bounces"));
        else if (a == "--models") models_dir = next("--models");
        else if (a == "--out") out = next("--out");
        else if (a == "--move") move = true;
        else if (a == "--gpus") gpus = (uint32_t)std::atoi(next("--gpus"));
        else if (a == "--spp") spp = (uint32_t)std::atoi(next("--spp"));
        else if (a == "--render") { render_mode = true; render_first = (uint32_t)std::atoi(next("--render")); render_count = (uint32_t)std::atoi(next("--render")); }
        else if (a == "--load-state") load_state = next("--load-state");
        else if (a == "--save-state") save_state = next("--save-state");
        else if (a == "--devices")
        {
            const std::string list = next("--devices");
            for (size_t p0 = 0; p0 < list.size();) { const size_t p1 = list.find(',', p0); devices.push_back(std::atoi(list.substr(p0, p1 - p0).c_str())); if (p1 == std::string::npos) break; p0 = p1 + 1; }
        }
        else if (a == "--help" || a == "-h")
        {
            std::printf("usage: %s [--width W] [--height H] [--frames N] [--bounces B] [--move] [--models DIR] [--out file.png]\n", argv[0]);
            return 0;
        }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try
    {
        // Materials  main.rs:77-92
        const Material diffuse_gray = Lambertian::New({0.73f, 0.73f, 0.73f});
        const Material diffuse_green = Lambertian::New({0.12f, 0.45f, 0.15f});
        const Material diffuse_red = Lambertian::New({0.65f, 0.05f, 0.05f});
        const Material light = Emissive::New(Vec3A::splat(15.0f));

        // Models and BVHs  main.rs:94-117 (the two blocks the reference has commented out stand in for its dragon, whose file it does not ship)
        const std::vector<Affine3A> one{Affine3A::IDENTITY()};
        const Scene scene = Scene::New({
            Model::New(models_dir + "/cb_light.obj", light, one),
            Model::New(models_dir + "/cb_main.obj", diffuse_gray, one),
            Model::New(models_dir + "/cb_right.obj", diffuse_red, one),
            Model::New(models_dir + "/cb_left.obj", diffuse_green, one),
            Model::New(models_dir + "/cb_box_tall.obj", diffuse_gray, one),
            Model::New(models_dir + "/cb_box_short.obj", diffuse_gray, one),
        });

        // Camera  main.rs:119-128
        const Vec3A look_from{0.0f, 50.0f, 1000.0f}, look_at{0.0f, 50.0f, 0.0f};
        const Camera cam = Camera::New(look_from, look_at, 60.0f, (float)width / (float)height, 0.0f, 950.0f);
        if (gpus > 0 || !devices.empty())
        {
            // several GPUs, one process: rows dealt to the devices in strips, one RCCL gather of the framebuffer (pt_multi)
            if (devices.empty()) for (uint32_t d = 0; d < gpus; ++d) devices.push_back((int32_t)d);
            MultiRenderer multi(scene, cam, width, height, bounces, devices);
            multi.render(0, 1);                                  // scene upload, RCCL communicator set-up
            multi.reset_accumulation();
            const auto m0 = std::chrono::steady_clock::now();
            multi.render(0, spp);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - m0).count();
            const pt_stats st = multi.stats();
            const double rays = (double)st.rays_closest + (double)st.rays_any + (double)st.rays_light_closest;
            std::printf("{\"gpus\": %zu, \"rccl\": %s, \"spp\": %u, \"width\": %u, \"height\": %u, \"ms\": %.3f, \"Mray_per_s\": %.1f}\n", devices.size(),
                        multi.used_rccl() ? "true" : "false", spp, width, height, 1e3 * sec, rays * (double)spp / (double)(spp + 1) / sec / 1e6);
            if (!out.empty()) multi.write_image(out);
            return 0;
        }
        Renderer renderer(scene, cam, width, height, bounces);
        if (render_mode)
        {
            // checkpoint / resume: the state file is {width, height, data rgba, position xyzt, id} of the frame as it lies on the device
            if (!load_state.empty())
            {
                std::FILE* f = std::fopen(load_state.c_str(), "rb");
                uint32_t wh[2] = {0, 0};
                Frame fr;
                const size_t px = (size_t)width * height;
                fr.data.resize(px * 4); fr.position.resize(px * 4); fr.id.resize(px);
                const bool ok = f && std::fread(wh, 4, 2, f) == 2 && wh[0] == width && wh[1] == height && std::fread(fr.data.data(), 4, px * 4, f) == px * 4 &&
                                std::fread(fr.position.data(), 4, px * 4, f) == px * 4 && std::fread(fr.id.data(), 4, px, f) == px;
                if (f) std::fclose(f);
                if (!ok) { std::fprintf(stderr, "cannot read a %ux%u frame state from %s\n", width, height, load_state.c_str()); return 1; }
                renderer.write_accumulation(fr);
            }
            renderer.render(render_first, render_count);
            if (!save_state.empty())
            {
                const Frame fr = renderer.read_frame();
                std::FILE* f = std::fopen(save_state.c_str(), "wb");
                const uint32_t wh[2] = {width, height};
                const size_t px = (size_t)width * height;
                const bool ok = f && std::fwrite(wh, 4, 2, f) == 2 && std::fwrite(fr.data.data(), 4, px * 4, f) == px * 4 && std::fwrite(fr.position.data(), 4, px * 4, f) == px * 4 &&
                                std::fwrite(fr.id.data(), 4, px, f) == px;
                if (f) std::fclose(f);
                if (!ok) { std::fprintf(stderr, "cannot write %s\n", save_state.c_str()); return 1; }
            }
            std::printf("{\"first_sample\": %u, \"samples\": %u, \"width\": %u, \"height\": %u}\n", render_first, render_count, width, height);
            if (!out.empty()) renderer.write_image(out);
            return 0;
        }
        Mat4 last_inv_proj = renderer.inv_projection();

        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t frame = 0; frame < frames; ++frame)
        {
            if (move && frame >= frames / 2)
            {
                // what Camera::input would receive from winit: a key held down and a slow mouse drag, dt = 1/60 s scaled to the
                // reference's sensitivities (camera.rs:35,43)
                renderer.input(PT_EV_KEY_W, 0.0f, 0.0f, 2.0e-6f);
                renderer.input(PT_EV_MOUSE_MOTION, 1.0f, 0.25f, 1.0e-6f);
            }
            renderer.frame(frame, last_inv_proj);            // the pixel loop + state.update   main.rs:181-215
            last_inv_proj = renderer.inv_projection();       // main.rs:216
        }
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const pt_stats st = renderer.stats();
        const double rays = (double)st.rays_closest + (double)st.rays_any + (double)st.rays_light_closest;
        std::printf("{\"frames\": %u, \"width\": %u, \"height\": %u, \"ms_per_frame\": %.3f, \"Mray_per_s\": %.1f}\n", frames, width, height,
                    1e3 * seconds / (frames ? frames : 1), rays / seconds / 1e6);
        if (!out.empty()) renderer.write_image(out);         // ImageHelper::write_image
    }
    catch (const Error& e)
    {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
