// Scene front end: reference-format conf -> Scene.  Host only, no HIP.
//
// Behaviour follows (paths relative to the reference root):
//   src/scene/cRTScene.cpp:6-28            Scene block (ModelName, WorldScale, ModelDisplacement)
//   src/geometry/cMeshLoader.cpp:48-201    OBJ/MTL through tinyobjloader: fan triangulation, face
//                                          ids running in file order, material id per face (-1 when
//                                          none), vertex = (float)((v + disp) * scale)
//   src/render/camera/cBaseCamera.cpp:57-90,132-156, cArcballCamera.cpp:30-34   camera
//   src/tracing/cPathTracer.cpp:148-182,215-259   raster->world matrix, PathTracer block
//   src/tracing/cLight.cpp:266-358         Light block (positions and radii times WorldScale)
#include "scene.hpp"
#include "sum_order.hpp"
#include "json.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace ptx
{

namespace
{

std::string read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw std::runtime_error("io: cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string dirname_of(const std::string &p)
{
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}

std::string join_path(const std::string &root, const std::string &rel)
{
    if (!rel.empty() && rel[0] == '/')
        return rel;
    if (root.empty())
        return rel;
    return root + "/" + rel;
}

// whitespace tokenizer for OBJ/MTL lines
std::vector<std::string> split_ws(const std::string &line)
{
    std::vector<std::string> out;
    size_t i = 0, n = line.size();
    while (i < n)
    {
        while (i < n && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r'))
            i++;
        size_t j = i;
        while (j < n && !(line[j] == ' ' || line[j] == '\t' || line[j] == '\r'))
            j++;
        if (j > i)
            out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

float to_float(const std::string &s) { return static_cast<float>(std::strtod(s.c_str(), nullptr)); }

std::string rest_of_line(const std::vector<std::string> &tok)
{
    std::string s;
    for (size_t i = 1; i < tok.size(); i++)
    {
        if (i > 1)
            s += " ";
        s += tok[i];
    }
    return s;
}

void load_mtl(const std::string &path, std::vector<Material> &mats)
{
    std::ifstream f(path);
    if (!f)
        return; // tinyobj only warns when the .mtl is missing; faces then get material id -1
    std::string line;
    Material *cur = nullptr;
    while (std::getline(f, line))
    {
        auto tok = split_ws(line);
        if (tok.empty() || tok[0][0] == '#')
            continue;
        const std::string &k = tok[0];
        if (k == "newmtl")
        {
            mats.emplace_back();
            cur = &mats.back();
            cur->name = rest_of_line(tok);
            continue;
        }
        if (!cur)
            continue;
        auto rgb = [&](double *dst) {
            for (int i = 0; i < 3 && 1 + i < static_cast<int>(tok.size()); i++)
                dst[i] = static_cast<double>(to_float(tok[1 + i]));
        };
        if (k == "Ka")
            rgb(cur->Ka);
        else if (k == "Kd")
            rgb(cur->Kd);
        else if (k == "Ks")
            rgb(cur->Ks);
        else if (k == "Ni" && tok.size() > 1)
            cur->Ni = static_cast<double>(to_float(tok[1]));
        else if (k == "Ns" && tok.size() > 1)
            cur->Ns = static_cast<double>(to_float(tok[1]));
        else if (k == "illum" && tok.size() > 1)
            cur->illum = static_cast<double>(std::atoi(tok[1].c_str()));
    }
}

int fix_index(int idx, int n)
{
    if (idx > 0)
        return idx - 1;
    if (idx < 0)
        return n + idx;
    return -1;
}

// 4x4 inverse (row-major): adjugate by cofactor expansion -- every cofactor the sum of six signed
// triple products, taken left to right -- divided by the determinant along row 0.  The CPU
// restatement the parity tests compare with uses the same expansion, term by term: camera rays are
// compared bit for bit, and the raster -> world matrix is where they start.
void invert4(const double *m, double *inv)
{
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = PTX_SUM4(m[0] * a[0], m[1] * a[4], m[2] * a[8], m[3] * a[12]);
    double id = 1.0 / det;
    for (int i = 0; i < 16; i++)
        inv[i] = a[i] * id;
}

void mul4(const double *a, const double *b, double *o)
{
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
        {
            r[i * 4 + j] = PTX_SUM4(a[i * 4 + 0] * b[0 * 4 + j], a[i * 4 + 1] * b[1 * 4 + j], a[i * 4 + 2] * b[2 * 4 + j],
                                    a[i * 4 + 3] * b[3 * 4 + j]);
        }
    std::memcpy(o, r, sizeof(r));
}

} // namespace

// glm::lookAt in float (right-handed): f = normalize(center - eye), s = normalize(f x up),
// u = s x f; rows (s, -s.e), (u, -u.e), (-f, f.e).  glm::normalize multiplies by 1/sqrt(dot).
void look_at_float(const double pos[3], const double front[3], const double up[3], double view16[16])
{
    float e[3], fr[3], upv[3], c[3], f[3], s[3], u[3];
    for (int i = 0; i < 3; i++)
    {
        e[i] = static_cast<float>(pos[i]);
        fr[i] = static_cast<float>(front[i]);
        upv[i] = static_cast<float>(up[i]);
        c[i] = e[i] + fr[i];
    }
    auto dot = [](const float *a, const float *b) {
        float t0 = a[0] * b[0], t1 = a[1] * b[1], t2 = a[2] * b[2];
        return (t0 + t1) + t2;
    };
    auto normalize = [&](float *v) {
        float k = 1.0f / std::sqrt(dot(v, v));
        v[0] *= k, v[1] *= k, v[2] *= k;
    };
    auto cross = [](const float *x, const float *y, float *o) {
        float a0 = x[1] * y[2], a1 = y[1] * x[2];
        float b0 = x[2] * y[0], b1 = y[2] * x[0];
        float c0 = x[0] * y[1], c1 = y[0] * x[1];
        o[0] = a0 - a1, o[1] = b0 - b1, o[2] = c0 - c1;
    };
    for (int i = 0; i < 3; i++)
        f[i] = c[i] - e[i];
    normalize(f);
    cross(f, upv, s);
    normalize(s);
    cross(s, f, u);
    float m[16] = {s[0], s[1], s[2], -dot(s, e), u[0], u[1], u[2], -dot(u, e),
                   -f[0], -f[1], -f[2], dot(f, e), 0, 0, 0, 1};
    for (int i = 0; i < 16; i++)
        view16[i] = static_cast<double>(m[i]);
}

void Scene::update_camera()
{
    const double W = width, H = height;
    const double m1[16] = {1.0 / W, 0, 0, 0.5 / W, 0, 1.0 / H, 0, 0.5 / H, 0, 0, 1, 0, 0, 0, 0, 1};
    const double m2[16] = {2, 0, 0, -1, 0, -2, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1};
    const double tn = std::tan(fov_deg / 180.0 * M_PI / 2);
    double m3[16] = {0};
    m3[0] = width * 1.0 / height * tn * near_plane;
    m3[5] = tn * near_plane;
    m3[10] = 0, m3[11] = -near_plane;
    m3[15] = 1;
    double m4[16], t[16];
    invert4(view, m4);
    mul4(m4, m3, t);
    mul4(t, m2, t);
    mul4(t, m1, raster2world);
}

void load_obj(const std::string &path, double scale, const double disp[3], Scene &out)
{
    std::ifstream f(path);
    if (!f)
        throw std::runtime_error("io: cannot open OBJ " + path);
    std::vector<float> v, vn;
    std::vector<std::array<int, 3>> fv, fn;
    std::vector<int> fm;
    int cur_mat = -1;
    std::string line;
    const std::string base = dirname_of(path);
    out.mats.clear();
    while (std::getline(f, line))
    {
        auto tok = split_ws(line);
        if (tok.empty() || tok[0][0] == '#')
            continue;
        const std::string &k = tok[0];
        if (k == "v" && tok.size() >= 4)
        {
            for (int i = 0; i < 3; i++)
                v.push_back(to_float(tok[1 + i]));
        }
        else if (k == "vn" && tok.size() >= 4)
        {
            for (int i = 0; i < 3; i++)
                vn.push_back(to_float(tok[1 + i]));
        }
        else if (k == "mtllib")
            load_mtl(join_path(base, rest_of_line(tok)), out.mats);
        else if (k == "usemtl")
        {
            const std::string name = rest_of_line(tok);
            cur_mat = -1;
            for (size_t i = 0; i < out.mats.size(); i++)
                if (out.mats[i].name == name)
                {
                    cur_mat = static_cast<int>(i);
                    break;
                }
        }
        else if (k == "f")
        {
            std::vector<int> vi, ni;
            for (size_t t = 1; t < tok.size(); t++)
            {
                const std::string &w = tok[t];
                size_t s1 = w.find('/');
                int a = std::atoi(w.substr(0, s1).c_str());
                int c = 0;
                if (s1 != std::string::npos)
                {
                    size_t s2 = w.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < w.size())
                        c = std::atoi(w.substr(s2 + 1).c_str());
                }
                vi.push_back(fix_index(a, static_cast<int>(v.size() / 3)));
                ni.push_back(c == 0 ? -1 : fix_index(c, static_cast<int>(vn.size() / 3)));
            }
            if (vi.size() < 3)
                continue;
            for (size_t t = 2; t < vi.size(); t++) // fan triangulation: (0, t-1, t)
            {
                fv.push_back({vi[0], vi[t - 1], vi[t]});
                fn.push_back({ni[0], ni[t - 1], ni[t]});
                fm.push_back(cur_mat);
            }
        }
    }
    const size_t F = fv.size();
    const int nv = static_cast<int>(v.size() / 3), nn = static_cast<int>(vn.size() / 3);
    out.verts.assign(F * 9, 0.f);
    out.normals.assign(F * 9, std::numeric_limits<float>::quiet_NaN());
    out.mat_ids.assign(fm.begin(), fm.end());
    out.vert_ids.assign(F * 3, -1);
    out.has_normals = nn > 0;
    for (size_t i = 0; i < F; i++)
        for (int j = 0; j < 3; j++)
        {
            int a = fv[i][j];
            if (a < 0 || a >= nv)
                throw std::runtime_error("parse: OBJ vertex index out of range in " + path);
            out.vert_ids[i * 3 + j] = a;
            for (int c = 0; c < 3; c++)
            {
                // tinyobj::real_t vx = (attrib.vertices[..] + disp[c]) * scale  -> rounded to float
                double w = (static_cast<double>(v[a * 3 + c]) + disp[c]) * scale;
                if (!std::isfinite(static_cast<float>(w)))
                    throw std::runtime_error("parse: OBJ vertex coordinate is not finite in " + path);
                out.verts[i * 9 + j * 3 + c] = static_cast<float>(w);
            }
            int b = fn[i][j];
            if (nn > 0 && b >= 0 && b < nn)
                for (int c = 0; c < 3; c++)
                    out.normals[i * 9 + j * 3 + c] = vn[b * 3 + c];
        }
    out.model_path = path;
}

void load_conf(const std::string &conf_path, const std::string &root_dir, const Overrides &ov, Scene &sc)
{
    Json root;
    try
    {
        root = Json::parse(read_file(conf_path));
    }
    catch (const std::runtime_error &e)
    {
        std::string m = e.what();
        if (m.rfind("io:", 0) == 0)
            throw;
        throw std::runtime_error("parse: " + conf_path + ": " + m);
    }

    const Json &scene = root["Scene"];
    if (scene.isNull())
        throw std::runtime_error("parse: conf has no Scene block");
    sc.world_scale = scene["WorldScale"].asDouble();
    double disp[3] = {0, 0, 0};
    if (scene["ModelDisplacement"].isNull())
        throw std::runtime_error("parse: Scene.ModelDisplacement missing");
    for (int i = 0; i < 3; i++)
        disp[i] = scene["ModelDisplacement"][i].asDouble();
    std::string model = ov.model.empty() ? scene["ModelName"].asString() : ov.model;
    load_obj(join_path(root_dir, model), sc.world_scale, disp, sc);

    const Json &win = root["MainWindowInfo"];
    sc.width = static_cast<int>(win["Width"].asFloat());
    sc.height = static_cast<int>(win["Height"].asFloat());
    if (ov.width > 0)
        sc.width = ov.width;
    if (ov.height > 0)
        sc.height = ov.height;
    if (sc.width <= 0 || sc.height <= 0)
        throw std::runtime_error("parse: MainWindowInfo Width/Height must be positive");

    const Json &cam = root["Camera"];
    double pos[3] = {0, 0, 1}, up[3] = {0, 1, 0}, front[3] = {0, 0, -1};
    for (int i = 0; i < 3; i++)
    {
        pos[i] = cam["CameraPos"][i].asDouble();
        up[i] = cam["CameraUp"][i].asDouble();
        front[i] = cam["CameraFront"][i].asDouble();
    }
    if (cam["Type"].asString() == "Arcball")
    {
        double g[3], n2 = 0;
        for (int i = 0; i < 3; i++)
        {
            g[i] = cam["CameraFocus"][i].asDouble() - pos[i];
            n2 += g[i] * g[i];
        }
        double n = std::sqrt(n2);
        for (int i = 0; i < 3; i++)
            front[i] = g[i] / n;
    }
    sc.fov_deg = static_cast<double>(cam["CameraFov"].asFloat());
    sc.near_plane = static_cast<double>(cam["CameraNear"].asFloat());
    for (int i = 0; i < 3; i++)
        sc.cam_pos[i] = pos[i];
    look_at_float(pos, front, up, sc.view);
    sc.update_camera();

    const Json &pt = root["PathTracer"];
    if (pt.isNull())
        throw std::runtime_error("parse: conf has no PathTracer block");
    sc.result_path = pt["ResultPath"].asString();
    sc.depth = ov.depth > 0 ? ov.depth : pt["Depth"].asInt();
    sc.spp = ov.spp > 0 ? ov.spp : pt["Samples"].asInt();
    sc.enable_indirect = pt["EnableIndirectLight"].asBool();
    sc.enable_bary_normal = pt["EnableBarycentricNormal"].asBool();
    sc.enable_path_tracing = pt["EnablePathTracing"].isNull() ? true : pt["EnablePathTracing"].asBool();
    const Json &reg = pt["DrawRegion"];
    if (reg.isNull() || reg.size() != 4)
        throw std::runtime_error("parse: PathTracer.DrawRegion must have 4 entries");
    for (int i = 0; i < 4; i++)
        sc.region[i] = reg[i].asInt();
    if (ov.full_region || ov.width > 0 || ov.height > 0)
    {
        sc.region[0] = 0, sc.region[1] = 0, sc.region[2] = sc.height, sc.region[3] = sc.width;
    }
    if (sc.depth <= 0 || sc.spp <= 0)
        throw std::runtime_error("parse: PathTracer.Depth and Samples must be positive");
    const std::string accel = pt["RayCastAccel"]["Type"].asString();
    if (accel == "None")
        sc.accel_type = 0;
    else if (accel == "AABB")
        sc.accel_type = 1;
    else if (accel == "OCTREE")
        sc.accel_type = 2;
    else
        throw std::runtime_error("parse: BuildAccelStruct invalid type " + accel);

    const Json &lj = root["Light"];
    const int nl = lj["LightNum"].asInt();
    if (sc.world_scale < 1e-5 || nl == 0 || lj["LightType"].isNull() || lj["LightInfo"].isNull() ||
        lj["LightRadiance"].isNull())
        throw std::runtime_error("parse: BuildLight light parse failed");
    sc.lights.clear();
    for (int i = 0; i < nl; i++)
    {
        Light L;
        const std::string type = lj["LightType"][i].asString();
        const Json &info = lj["LightInfo"][i];
        for (int c = 0; c < 3; c++)
            L.radiance[c] = lj["LightRadiance"][i][c].asDouble();
        if (type == "square")
        {
            L.type = 0;
            for (int k = 0; k < 4; k++)
            {
                if (info[k].isNull())
                    throw std::runtime_error("parse: BuildLight: sub point fetched fail");
                for (int c = 0; c < 3; c++)
                    L.p[k][c] = info[k][c].asDouble() * sc.world_scale;
            }
        }
        else if (type == "sphere")
        {
            L.type = 1;
            for (int c = 0; c < 3; c++)
                L.center[c] = info[c].asDouble() * sc.world_scale;
            L.radius = info[3].asDouble() * sc.world_scale;
        }
        else
            throw std::runtime_error("parse: BuildLight: Unsupported type " + type);
        sc.lights.push_back(L);
    }
}

void finalize_image(const double *sum, int64_t n_pixels, double *out)
{
    for (int64_t i = 0; i < n_pixels; i++)
    {
        double r = sum[i * 3], g = sum[i * 3 + 1], b = sum[i * 3 + 2];
        double m = r > g ? r : g;
        m = m > b ? m : b;
        m = m > 0 ? m : 0; // the reference's 4th colour component is 0 and takes part in maxCoeff
        if (m > 1)
            r /= m, g /= m, b /= m;
        out[i * 3] = r, out[i * 3 + 1] = g, out[i * 3 + 2] = b;
    }
}

void write_ppm(const std::string &path, const double *rgb, int width, int height)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f)
        throw std::runtime_error("io: cannot write " + path);
    std::fprintf(f, "P3\n%d %d\n%d\n", width, height, 255);
    const int64_t n = static_cast<int64_t>(width) * height;
    for (int64_t i = 0; i < n; i++)
        std::fprintf(f, "%d %d %d ", static_cast<int>(rgb[i * 3] * 255), static_cast<int>(rgb[i * 3 + 1] * 255),
                     static_cast<int>(rgb[i * 3 + 2] * 255));
    std::fclose(f);
}

} // namespace ptx
