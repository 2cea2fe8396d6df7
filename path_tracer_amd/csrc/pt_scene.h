// Host side of Scene::new (src/scene.rs:21-35): BLAS / TLAS builders, light sampler, camera matrices and the
// flattening into the device layout of pt_types.h.  Runs once per scene on the CPU, no GPU involved.
#pragma once
#include <string>
#include <vector>

#include "pt_types.h"

namespace pt {

struct HostBox { f3 mn, mx; };

struct HostNode
{
    HostBox box;
    uint32_t kind; // NODE_BRANCH / NODE_TRIS / NODE_INSTANCE
    uint32_t a, b; // branch: left,right (arena-local); tris: first,count into prim_ids; instance: instance index, blas index
};

struct HostTriangle
{
    f3 p[3], n[3];
    f4 n0, n1, n2; // primitive.rs:20-26
};

struct HostBlas
{
    std::vector<HostTriangle> tris;   // load order (Id<Triangle>)
    std::vector<HostNode> nodes;      // arena order (children before parents)
    std::vector<uint32_t> prim_ids;   // leaf contents, in leaf order
    uint32_t root = 0;
    uint32_t depth = 0;               // nodes on the longest root->leaf path
    int material = 0;
};

struct HostInstance
{
    uint32_t blas;  // arena index inside the owning TLAS
    uint32_t model; // model index (== world blas index)
    xf34 fwd, inv;
};

struct HostTlas
{
    std::vector<HostNode> nodes;
    std::vector<HostInstance> instances; // leaf allocation order
    uint32_t root = MISS_ID;
    uint32_t depth = 0;
    std::vector<uint32_t> models;        // arena index -> model index
};

struct HostModel
{
    std::vector<float> positions, normals; // n_tris * 9
    uint32_t n_tris = 0;
    int material = 0;
    std::vector<xf34> matrices;
};

struct HostLight { uint32_t blas, prim; float pdf, cdf; };

struct HostCamera
{
    bool set = false;
    float yaw = 0, pitch = 0; // as the reference names them: pitch is the Y angle, yaw the X angle of EulerRot::YXZ (camera.rs:23)
    xf34 matrix;          // camera-to-world
    float inv_proj[16];   // column-major
    float ray_matrix[16]; // matrix * inv_projection, column-major
};

struct FlatScene
{
    std::vector<DNode> nodes;
    std::vector<DTriIsect> tri_isect;
    std::vector<DTriVerts> tri_shade, tri_pos;
    std::vector<uint32_t> tri_orig;
    std::vector<DInstance> instances;
    std::vector<uint32_t> big_leaves;   // {first, count} pairs of the leaves NODE_TRIS cannot encode
    std::vector<DMaterial> materials;
    std::vector<DLight> lights;
    std::vector<uint32_t> tri_base;     // per model: absolute index of its first triangle
    std::vector<uint32_t> inst_base;    // [0] world instances start, [1] lights instances start
    uint32_t world_root = MISS_ID, lights_root = MISS_ID;
    uint32_t prim_bits = 0;
    uint32_t stack_entries = 0;
    float light_weight_sum = 0;
    bool has_volumes = false;
};

class HostScene
{
public:
    std::vector<DMaterial> materials;
    std::vector<HostModel> models;
    std::vector<HostBlas> blas;         // one per model
    HostTlas world, lights;
    std::vector<HostLight> light_items;
    float light_weight_sum = 0;
    HostCamera camera;
    FlatScene flat;
    bool built = false;

    int add_material(int kind, const float colour[3], float roughness, float ior, bool has_volume, const float vol_abs[3], float k, float c,
                     float g);
    int add_model(const float* positions, const float* normals, uint32_t n_tris, int material, const float* affines, uint32_t n_inst);
    // load_obj (blas.rs:44-131) + add_model; returns the model index, -1 bad argument, -4 non-rigid, -6 unreadable file, -7 parse error
    int add_model_obj(const char* path, int material, const float* affines, uint32_t n_inst, std::string* err);
    int build(std::string* err);
    void set_camera(const float eye[3], const float target[3], float fov_deg, float aspect);
    void create_ray(float s, float t, float o[3], float d[3]) const;
    void inv_projection(float out16[16]) const; // (matrix * inv_projection).inverse()  main.rs:128
    void camera_move(float dx, float dz, float dt);   // Camera::update_origin    camera.rs:33-39
    void camera_rotate(float dx, float dy, float dt); // Camera::update_rotation  camera.rs:41-54

private:
    void build_blas(HostBlas& out, const HostModel& m);
    void build_tlas(HostTlas& out, const std::vector<uint32_t>& model_ids);
    void build_lights();
    int flatten(std::string* err);
    void refresh_ray_matrix();
};

} // namespace pt
