// Host-side scene description: everything the tracer consumes, parsed once into plain arrays.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace ptx
{

struct Material
{
    double Ka[3] = {0, 0, 0}, Kd[3] = {0, 0, 0}, Ks[3] = {0, 0, 0};
    double Ni = 1.0, Ns = 1.0, illum = 0.0;
    std::string name;
};

struct Light
{
    int type = 0;               // 0 square, 1 sphere
    double radiance[3] = {0, 0, 0};
    double p[4][3] = {};        // square: corners (scaled)
    double center[3] = {0, 0, 0};
    double radius = 0;          // sphere (scaled)
};

struct Overrides
{
    int width = 0, height = 0, spp = 0, depth = 0;
    std::string model;
    bool full_region = false;
};

struct Scene
{
    // geometry: one entry per triangle, vertices duplicated per face like the reference's tVertex
    std::vector<float> verts;   // [F][3][3]  positions, already (float)((v + disp) * scale)
    std::vector<float> normals; // [F][3][3]  vertex normals (NaN when the OBJ carries none)
    std::vector<int32_t> mat_ids;
    std::vector<int32_t> vert_ids; // [F][3] OBJ vertex index (0-based) of each corner
    std::vector<Material> mats;
    std::vector<Light> lights;
    bool has_normals = false;

    // camera
    int width = 0, height = 0;
    double fov_deg = 0, near_plane = 0; // (double)(float) of the conf values
    double view[16] = {};               // row-major, (double)(float) glm::lookAt
    double cam_pos[3] = {0, 0, 0};
    double raster2world[16] = {};       // V^-1 * mat3 * mat2 * mat1

    // tracer
    int depth = 1, spp = 1;
    bool enable_indirect = true, enable_bary_normal = true, enable_path_tracing = true;
    int region[4] = {0, 0, 0, 0};
    int accel_type = 0;
    double world_scale = 1.0;
    std::string result_path, model_path;

    int n_tris() const { return static_cast<int>(mat_ids.size()); }
    int n_square_lights() const
    {
        int n = 0;
        for (const auto &l : lights)
            n += l.type == 0;
        return n;
    }
    // recomputes raster2world from view / fov / near / W / H
    void update_camera();
};

// throws std::runtime_error (message starts with "io:", "parse:" or "arg:")
void load_conf(const std::string &conf_path, const std::string &root_dir, const Overrides &ov, Scene &out);
void load_obj(const std::string &path, double scale, const double disp[3], Scene &out);
void look_at_float(const double pos[3], const double front[3], const double up[3], double view16[16]);

void finalize_image(const double *sum, int64_t n_pixels, double *out);
void write_ppm(const std::string &path, const double *rgb, int width, int height);

} // namespace ptx
