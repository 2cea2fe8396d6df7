// ptmi.hpp — C++17 host side above the C-ABI (include/pt_api.h), mirroring the reference's own setup and per-frame calls so that a
// driver reads like src/main.rs.  Header-only; link with -lptmi.  The reference is Rust: `Type::new(..)` becomes `Type::New(..)`
// (`new` is a C++ keyword), `Result`/`unwrap` panics become ptmi::Error.  Everything here is plumbing: no arithmetic of the path.
//
//   reference (file:line)                                   here
//   Volume::new(absorption, k, c, g)      volume.rs:136      ptmi::Volume::New
//   Lambertian::new(albedo)               material.rs:99     ptmi::Lambertian::New
//   Emissive::new(emitted)                material.rs:126    ptmi::Emissive::New
//   Specular::new(colour)                 material.rs:146    ptmi::Specular::New
//   GGX::new_metal / new_dielectric       material.rs:290,305 ptmi::GGX::NewMetal / NewDielectric
//   Dielectric::new(colour, ior, volume)  material.rs:475    ptmi::Dielectric::New
//   Model::new(path, material, matrices)  model.rs:36        ptmi::Model::New            (+ Model::FromTriangles for triangle soups)
//   Scene::new(models)                    scene.rs:21        ptmi::Scene::New
//   Camera::new(origin, target, fov, aspect, _, _)  camera.rs:17   ptmi::Camera::New
//   Camera::input(event, window, dt)      camera.rs:56       ptmi::Renderer::input
//   the pixel loop + state.update(..)     main.rs:181-216    ptmi::Renderer::frame
//   (cam.matrix * cam.inv_projection).inverse()  main.rs:128 ptmi::Renderer::inv_projection
//   state.render()                        state.rs:629       ptmi::Renderer::present
//   ImageHelper::write_image              image_helper.rs:37 ptmi::Renderer::write_image
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pt_api.h"

namespace ptmi {

struct Error : std::runtime_error
{
    int code;
    Error(int c, const std::string& what) : std::runtime_error("libptmi error " + std::to_string(c) + ": " + what), code(c) {}
};

struct Vec3A
{
    float x, y, z;
    static Vec3A splat(float v) { return {v, v, v}; }
};

// row-major 3x4, as pt_add_model takes it (glam::Affine3A)
struct Affine3A
{
    std::array<float, 12> m;
    static Affine3A IDENTITY() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}; }
};
using Mat4 = std::array<float, 16>; // column-major, as glam::Mat4

struct Volume
{
    Vec3A absorption;
    float k, c, g;
    static Volume New(Vec3A absorption, float k, float c, float g) { return {absorption, k, c, g}; }
};

struct Material
{
    pt_material_desc d{};
    bool operator==(const Material& o) const
    {
        return d.kind == o.d.kind && d.colour[0] == o.d.colour[0] && d.colour[1] == o.d.colour[1] && d.colour[2] == o.d.colour[2] &&
               d.roughness == o.d.roughness && d.ior == o.d.ior && d.has_volume == o.d.has_volume &&
               (!d.has_volume || (d.vol_absorption[0] == o.d.vol_absorption[0] && d.vol_absorption[1] == o.d.vol_absorption[1] &&
                                  d.vol_absorption[2] == o.d.vol_absorption[2] && d.vol_k == o.d.vol_k && d.vol_c == o.d.vol_c && d.vol_g == o.d.vol_g));
    }
};
namespace detail {
inline Material material(int kind, Vec3A colour, float roughness = 0.0f, float ior = 1.0f, const std::optional<Volume>& v = std::nullopt)
{
    Material m;
    m.d.kind = kind;
    m.d.colour[0] = colour.x; m.d.colour[1] = colour.y; m.d.colour[2] = colour.z;
    m.d.roughness = roughness;
    m.d.ior = ior;
    if (v)
    {
        m.d.has_volume = 1;
        m.d.vol_absorption[0] = v->absorption.x; m.d.vol_absorption[1] = v->absorption.y; m.d.vol_absorption[2] = v->absorption.z;
        m.d.vol_k = v->k; m.d.vol_c = v->c; m.d.vol_g = v->g;
    }
    return m;
}
} // namespace detail
struct Lambertian { static Material New(Vec3A albedo) { return detail::material(PT_LAMBERTIAN, albedo); } };
struct Emissive { static Material New(Vec3A emitted) { return detail::material(PT_EMISSIVE, emitted); } };
struct Specular { static Material New(Vec3A colour) { return detail::material(PT_SPECULAR, colour); } };
struct GGX
{
    static Material NewMetal(Vec3A colour, float roughness) { return detail::material(PT_GGX_METAL, colour, roughness); }
    static Material NewDielectric(Vec3A colour, float roughness, float ior, std::optional<Volume> volume)
    {
        return detail::material(PT_GGX_DIELECTRIC, colour, roughness, ior, volume);
    }
};
struct Dielectric
{
    static Material New(Vec3A colour, float ior, std::optional<Volume> volume) { return detail::material(PT_DIELECTRIC, colour, 0.0f, ior, volume); }
};

struct Model
{
    std::string path;                     // OBJ file read by the library's load_obj (blas.rs:44-131) ...
    std::vector<float> positions, normals; // ... or a triangle soup, 9 floats per triangle each
    Material material;
    std::vector<Affine3A> matrices;
    static Model New(std::string file_path, Material material, std::vector<Affine3A> matrices) { return {std::move(file_path), {}, {}, material, std::move(matrices)}; }
    static Model FromTriangles(std::vector<float> positions, std::vector<float> normals, Material material, std::vector<Affine3A> matrices)
    {
        return {"", std::move(positions), std::move(normals), material, std::move(matrices)};
    }
};

struct Scene
{
    std::vector<Model> models;
    static Scene New(std::vector<Model> models) { return {std::move(models)}; }
};

struct Camera
{
    Vec3A origin, target;
    float fov, aspect_ratio;
    static Camera New(Vec3A origin, Vec3A target, float fov, float aspect_ratio, float /*aperture*/, float /*focus*/) { return {origin, target, fov, aspect_ratio}; }
};

struct Frame
{
    std::vector<float> data, position; // W*H*4 each: this frame's (rgb, 1) and first-hit (xyz, t)   main.rs:135-136
    std::vector<uint32_t> id;          // W*H: (old << 16) | new                                        main.rs:137,206
};

// Scene::new on a context: materials in first-use order, models, pt_build
inline void upload(pt_ctx* ctx_, const Scene& scene)
{
    auto check = [&](int r) { if (r < 0) throw Error(r, pt_last_error(ctx_)); return r; };
        std::vector<Material> mats; // distinct materials in first-use order
        for (const Model& m : scene.models)
        {
            size_t idx = 0;
            while (idx < mats.size() && !(mats[idx] == m.material)) ++idx;
            if (idx == mats.size()) { mats.push_back(m.material); check(pt_add_material(ctx_, &m.material.d)); }
            const float* mat = m.matrices.empty() ? nullptr : m.matrices[0].m.data();
            const uint32_t n_inst = (uint32_t)m.matrices.size();
            if (!m.path.empty()) check(pt_add_model_obj(ctx_, m.path.c_str(), (int)idx, mat, n_inst));
            else check(pt_add_model(ctx_, m.positions.data(), m.normals.data(), (uint32_t)(m.positions.size() / 9), (int)idx, mat, n_inst));
        }
        check(pt_build(ctx_));
}

// owns a pt_ctx: Scene::new + the wavefront state of main.rs's loop
class Renderer
{
public:
    Renderer(const Scene& scene, const Camera& cam, uint32_t width, uint32_t height, uint32_t max_bounces, uint32_t n_sobol = 512, bool enable_nee = true,
             uint64_t seed = 0x5EED5EEDull, int device = -1)
    {
        pt_config cfg{};
        cfg.width = width; cfg.height = height; cfg.max_bounces = max_bounces; cfg.n_sobol = n_sobol; cfg.enable_nee = enable_nee ? 1u : 0u;
        cfg.seed = seed; cfg.rank = 0; cfg.world_size = 1; cfg.strip_rows = 4; cfg.batch_spp = 0; cfg.device = device; cfg.flags = 0;
        ctx_ = pt_create(&cfg);
        if (!ctx_) throw Error(PT_ERR_ARG, "pt_create failed (bad configuration)");
        width_ = width; height_ = height;
        upload(ctx_, scene);
        set_camera(cam);
    }
    ~Renderer() { if (ctx_) pt_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void set_camera(const Camera& c)
    {
        const float eye[3] = {c.origin.x, c.origin.y, c.origin.z}, tgt[3] = {c.target.x, c.target.y, c.target.z};
        check(pt_set_camera(ctx_, eye, tgt, c.fov, c.aspect_ratio));
    }
    // Camera::input: true where the reference's returns true
    bool input(pt_event event, float a, float b, float dt) { return check(pt_camera_input(ctx_, event, a, b, dt)) == 1; }
    Mat4 inv_projection() const { Mat4 m{}; check(pt_inv_projection(ctx_, m.data())); return m; }
    void set_environment(uint32_t w, uint32_t h, const float* rgb_linear) { check(pt_set_environment(ctx_, w, h, rgb_linear)); }

    // one MainEventsCleared iteration: the pixel loop for sample `frame_index`, then State::update  (main.rs:179-216)
    void frame(uint32_t frame_index, const Mat4& last_inv_projection, Frame* out = nullptr)
    {
        if (out)
        {
            out->data.resize((size_t)width_ * height_ * 4); out->position.resize((size_t)width_ * height_ * 4); out->id.resize((size_t)width_ * height_);
            check(pt_frame(ctx_, frame_index, last_inv_projection.data(), out->data.data(), out->position.data(), out->id.data()));
        }
        else check(pt_frame(ctx_, frame_index, last_inv_projection.data(), nullptr, nullptr, nullptr));
    }
    // n_samples per pixel accumulated without the temporal pass (what the loop converges to for a camera at rest)
    void render(uint32_t first_sample, uint32_t n_samples) { check(pt_render_device(ctx_, first_sample, n_samples)); check(pt_synchronize(ctx_)); }
    void reset_accumulation() { check(pt_reset_accumulation(ctx_)); }
    Frame read_frame() const
    {
        Frame f;
        f.data.resize((size_t)width_ * height_ * 4); f.position.resize((size_t)width_ * height_ * 4); f.id.resize((size_t)width_ * height_);
        check(pt_read_frame(ctx_, f.data.data(), f.position.data(), f.id.data()));
        return f;
    }
    // checkpoint / resume: put a frame's state (as frame() / pt_render returned it) back, e.g. in another process
    void write_accumulation(const Frame& f)
    {
        // pt_write_accumulation copies width * height texels from each pointer: a frame saved at another size must not be read past its end
        const size_t px = (size_t)width_ * height_;
        if (f.data.size() != px * 4 || (!f.position.empty() && f.position.size() != px * 4) || (!f.id.empty() && f.id.size() != px))
            throw Error(PT_ERR_ARG, "write_accumulation: the frame was not saved at this renderer's " + std::to_string(width_) + "x" + std::to_string(height_));
        check(pt_write_accumulation(ctx_, f.data.data(), f.position.empty() ? nullptr : f.position.data(), f.id.empty() ? nullptr : f.id.data()));
    }
    std::vector<float> present() const { std::vector<float> v((size_t)width_ * height_ * 4); check(pt_present(ctx_, v.data())); return v; }
    std::vector<uint8_t> present_rgb8() const { std::vector<uint8_t> v((size_t)width_ * height_ * 3); check(pt_present_rgb8(ctx_, v.data())); return v; }
    void write_image(const std::string& path) const { check(pt_write_image(ctx_, path.c_str())); }
    pt_stats stats() const { pt_stats s{}; check(pt_get_stats(ctx_, &s)); return s; }
    pt_ctx* handle() const { return ctx_; }

private:
    int check(int r) const
    {
        if (r < 0) throw Error(r, pt_last_error(ctx_));
        return r;
    }
    pt_ctx* ctx_ = nullptr;
    uint32_t width_ = 0, height_ = 0;
};

// owns a pt_multi: the same pixel loop fanned out over several GPUs of one process (the reference fans it out over the threads of
// one rayon pool, main.rs:72,181); rows are dealt to the devices in strips, one RCCL gather per render
class MultiRenderer
{
public:
    MultiRenderer(const Scene& scene, const Camera& cam, uint32_t width, uint32_t height, uint32_t max_bounces, const std::vector<int32_t>& devices,
                  uint32_t n_sobol = 512, bool enable_nee = true, uint64_t seed = 0x5EED5EEDull)
    {
        pt_config cfg{};
        cfg.width = width; cfg.height = height; cfg.max_bounces = max_bounces; cfg.n_sobol = n_sobol; cfg.enable_nee = enable_nee ? 1u : 0u;
        cfg.seed = seed; cfg.strip_rows = 4; cfg.device = -1;
        m_ = pt_multi_create(&cfg, devices.data(), (uint32_t)devices.size());
        if (!m_) throw Error(PT_ERR_ARG, "pt_multi_create failed (bad configuration)");
        width_ = width; height_ = height;
        pt_ctx* c0 = pt_multi_ctx(m_, 0);
        upload(c0, scene);
        const float eye[3] = {cam.origin.x, cam.origin.y, cam.origin.z}, tgt[3] = {cam.target.x, cam.target.y, cam.target.z};
        if (pt_set_camera(c0, eye, tgt, cam.fov, cam.aspect_ratio) < 0) throw Error(PT_ERR_STATE, pt_last_error(c0));
    }
    ~MultiRenderer() { if (m_) pt_multi_destroy(m_); }
    MultiRenderer(const MultiRenderer&) = delete;
    MultiRenderer& operator=(const MultiRenderer&) = delete;
    // samples [first_sample, first_sample + n_samples) of every pixel on all devices, gathered on the first; `out` (W*H*4) optional
    void render(uint32_t first_sample, uint32_t n_samples, std::vector<float>* out = nullptr)
    {
        if (out) out->resize((size_t)width_ * height_ * 4);
        check(pt_multi_render(m_, first_sample, n_samples, out ? out->data() : nullptr));
    }
    void reset_accumulation() { check(pt_multi_reset_accumulation(m_)); }
    void write_image(const std::string& path) { check(pt_multi_write_image(m_, path.c_str())); }
    bool used_rccl() const { return pt_multi_used_rccl(m_) == 1; }
    pt_stats stats() const { pt_stats s{}; check(pt_multi_get_stats(m_, &s)); return s; }
    pt_multi* handle() const { return m_; }

private:
    int check(int r) const
    {
        if (r < 0) throw Error(r, pt_multi_last_error(m_));
        return r;
    }
    pt_multi* m_ = nullptr;
    uint32_t width_ = 0, height_ = 0;
};

} // namespace ptmi
