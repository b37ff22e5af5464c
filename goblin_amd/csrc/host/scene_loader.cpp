// libgoblin_host.so -- scene front end: Goblin JSON + OBJ -> gbl_scene_desc.
//
// Host-side mirror of the reference's ContextLoader for the hot-path subset
// (/root/reference/src/GoblinContextLoader.cpp:33-504).  Same keys, same
// defaults, same int-vs-float strictness of ParamSet, same "first definition of
// a name wins" map semantics (SceneCache::add* use std::map::insert,
// GoblinScene.cpp:130-158).  Where the reference silently substitutes its
// magenta/unit-sphere error objects for a missing name (GoblinScene.cpp:170-226)
// this loader fails with GBL_ERR_INVALID and a message instead -- there is no
// sphere primitive on the device path to stand in with.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/goblin_hip.h"
#include "json_lite.h"
#include "../abi_guard.h"

namespace {
thread_local std::string g_last_error;
}

namespace gbl_host_detail {
gbl_status fail(gbl_status code, const std::string& msg) {   // shared with image_io.cpp
    g_last_error = msg;
    return code;
}
}  // namespace gbl_host_detail

namespace {
using gbl_host_detail::fail;

// ---------------------------------------------------------------------------
// ParamSet: a strictly typed bag (GoblinParamSet.cpp:100-161,
// GoblinContextLoader.cpp:33-65).  An integer literal is only visible to
// get_int, a float literal only to get_float; 2/3/4-element arrays become
// vectors with elements narrowed to float; nested objects are ignored.
// ---------------------------------------------------------------------------
struct Vec {
    float v[4] = {0, 0, 0, 0};
};

struct Params {
    std::map<std::string, bool> bools;
    std::map<std::string, int> ints;
    std::map<std::string, float> floats;
    std::map<std::string, std::string> strings;
    std::map<std::string, Vec> vec2s, vec3s, vec4s;

    explicit Params(const gbl_json::Value* obj = nullptr) {
        if (!obj || obj->kind != gbl_json::Value::Object) return;
        for (const auto& kv : obj->obj) {
            const gbl_json::Value& v = kv.second;
            switch (v.kind) {
                case gbl_json::Value::Bool: bools.insert({kv.first, v.b}); break;
                case gbl_json::Value::Int: ints.insert({kv.first, static_cast<int>(v.i)}); break;
                case gbl_json::Value::Float: floats.insert({kv.first, static_cast<float>(v.d)}); break;
                case gbl_json::Value::String: strings.insert({kv.first, v.s}); break;
                case gbl_json::Value::Array: {
                    size_t n = v.arr.size();
                    if (n < 2 || n > 4) break;
                    Vec vec;
                    bool ok = true;
                    for (size_t i = 0; i < n; ++i) {
                        if (!v.arr[i].is_number()) ok = false;
                        else vec.v[i] = v.arr[i].as_float();
                    }
                    if (!ok) break;
                    (n == 2 ? vec2s : n == 3 ? vec3s : vec4s).insert({kv.first, vec});
                    break;
                }
                default: break;
            }
        }
    }
    bool has_string(const std::string& k) const { return strings.count(k) != 0; }
    bool has_vec3(const std::string& k) const { return vec3s.count(k) != 0; }
    int get_int(const std::string& k, int d = 0) const {
        auto it = ints.find(k);
        return it == ints.end() ? d : it->second;
    }
    bool get_bool(const std::string& k, bool d = false) const {
        auto it = bools.find(k);
        return it == bools.end() ? d : it->second;
    }
    float get_float(const std::string& k, float d = 0.0f) const {
        auto it = floats.find(k);
        return it == floats.end() ? d : it->second;
    }
    std::string get_string(const std::string& k, const std::string& d = "") const {
        auto it = strings.find(k);
        return it == strings.end() ? d : it->second;
    }
    Vec get_vec(const std::map<std::string, Vec>& m, const std::string& k, Vec d) const {
        auto it = m.find(k);
        return it == m.end() ? d : it->second;
    }
};

Vec vec(float a, float b = 0, float c = 0, float d = 0) {
    Vec r;
    r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d;
    return r;
}

const float kPi = 3.14159265358979323f;  // GoblinUtils.h:43
float radians(float deg) { return kPi * (deg / 180.0f); }  // GoblinUtils.h:132-134

// ---------------------------------------------------------------------------
// Quaternion helpers needed at load time only: "euler" orientations
// (GoblinUtils.cpp:78-90, GoblinQuaternion.cpp:8-14,124-149).
// ---------------------------------------------------------------------------
struct Quat {
    float w, x, y, z;
};

Quat quat_axis_angle(int axis, float angle) {
    float t = angle * 0.5f;
    float ax[3] = {0, 0, 0};
    ax[axis] = 1.0f;
    // normalize(axis): v / length, i.e. multiply by 1/len
    float inv = 1.0f / std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    float s = std::sin(t);
    Quat q;
    q.w = std::cos(t);
    q.x = ax[0] * inv * s;
    q.y = ax[1] * inv * s;
    q.z = ax[2] * inv * s;
    return q;
}

Quat quat_mul(const Quat& a, const Quat& b) {
    // Quaternion(w*rw - dot(v,rv), w*rv + rw*v + cross(v,rv))   GoblinQuaternion.h:45-48
    Quat r;
    r.w = a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z);
    float cx = a.y * b.z - a.z * b.y;
    float cy = a.z * b.x - a.x * b.z;
    float cz = a.x * b.y - a.y * b.x;
    r.x = a.w * b.x + b.w * a.x + cx;
    r.y = a.w * b.y + b.w * a.y + cy;
    r.z = a.w * b.z + b.w * a.z + cz;
    return r;
}

bool read_orientation(const Params& p, float out[4], std::string* err) {
    if (p.has_vec3("euler")) {
        Vec e = p.get_vec(p.vec3s, "euler", vec(0, 0, 0));
        std::string order = p.get_string("rotation_order", "xyz");
        Quat qx = quat_axis_angle(0, radians(e.v[0]));
        Quat qy = quat_axis_angle(1, radians(e.v[1]));
        Quat qz = quat_axis_angle(2, radians(e.v[2]));
        Quat r;
        if (order == "xzy") r = quat_mul(quat_mul(qy, qz), qx);
        else if (order == "yxz") r = quat_mul(quat_mul(qz, qx), qy);
        else if (order == "yzx") r = quat_mul(quat_mul(qx, qz), qy);
        else if (order == "zxy") r = quat_mul(quat_mul(qy, qx), qz);
        else if (order == "zyx") r = quat_mul(quat_mul(qx, qy), qz);
        else r = quat_mul(quat_mul(qz, qy), qx);  // "xyz" and the unrecognised fallback
        out[0] = r.w; out[1] = r.x; out[2] = r.y; out[3] = r.z;
    } else {
        Vec q = p.get_vec(p.vec4s, "orientation", vec(1, 0, 0, 0));
        for (int i = 0; i < 4; ++i) out[i] = q.v[i];
    }
    (void)err;
    return true;
}

void read_trs(const Params& p, gbl_trs* t) {
    Vec pos = p.get_vec(p.vec3s, "position", vec(0, 0, 0));
    Vec scl = p.get_vec(p.vec3s, "scale", vec(1, 1, 1));
    for (int i = 0; i < 3; ++i) {
        t->position[i] = pos.v[i];
        t->scale[i] = scl.v[i];
    }
    std::string err;
    read_orientation(p, t->orientation, &err);
}

// ---------------------------------------------------------------------------
// OBJ loader: v / vn / vt / f with triangles and quads, format fixed by the
// first face, python-style negative indices, (v,vn,vt) de-duplicated in face
// order (GoblinPolygonMesh.cpp:58-262).
// ---------------------------------------------------------------------------
struct MeshData {
    std::vector<float> pos, nrm, uv;
    std::vector<uint32_t> idx;
    bool has_normal = false, has_uv = false;
};

struct Corner {
    int v, n, t;
    bool operator<(const Corner& o) const {
        if (v != o.v) return v < o.v;
        if (n != o.n) return n < o.n;
        return t < o.t;
    }
};

std::string read_file(const std::string& path, bool* ok);

bool load_obj(const std::string& path, MeshData* mesh, std::string* err) {
    // The reference reads line by line through iostreams (GoblinPolygonMesh.cpp:58-205); this is the same grammar on a
    // single buffer with strtof / atoi (a 330 MB, 8 M-triangle file loads in ~1 s instead of ~6 s).
    bool ok = false;
    std::string text = read_file(path, &ok);
    if (!ok) {
        *err = "can't open obj file: " + path;
        return false;
    }
    std::vector<float> vs, ns, ts;
    std::vector<Corner> corners;  // 3 per face
    enum { V_ONLY, V_UV, V_N, V_UV_N } format = V_ONLY;
    bool first_face = true;
    int line_no = 0;
    const char* p = text.c_str();
    const char* const end = p + text.size();
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    auto parse_floats = [&](const char* q, const char* eol, int n, float* out) {
        for (int i = 0; i < n; ++i) {
            while (q < eol && is_space(*q)) ++q;
            if (q >= eol) return false;
            char* stop = nullptr;
            out[i] = strtof(q, &stop);
            if (stop == q || stop > eol) return false;
            q = stop;
        }
        return true;
    };
    while (p < end) {
        ++line_no;
        const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) eol = end;
        const char* q = p;
        while (q < eol && is_space(*q)) ++q;
        const char* tok = q;
        while (q < eol && !is_space(*q)) ++q;
        const size_t tl = static_cast<size_t>(q - tok);
        if ((tl == 1 && tok[0] == 'v') || (tl == 2 && tok[0] == 'v' && tok[1] == 'n')) {
            float f[3];
            if (!parse_floats(q, eol, 3, f)) {
                *err = path + ": syntax error on line " + std::to_string(line_no);
                return false;
            }
            std::vector<float>& dst = tl == 1 ? vs : ns;
            dst.push_back(f[0]); dst.push_back(f[1]); dst.push_back(f[2]);
        } else if (tl == 2 && tok[0] == 'v' && tok[1] == 't') {
            float f[2];
            if (!parse_floats(q, eol, 2, f)) {
                *err = path + ": uv syntax error on line " + std::to_string(line_no);
                return false;
            }
            ts.push_back(f[0]); ts.push_back(f[1]);
        } else if (tl == 1 && tok[0] == 'f') {
            const char* ft[5];
            size_t fl[5];
            size_t nf = 0;
            while (q < eol) {
                while (q < eol && is_space(*q)) ++q;
                if (q >= eol) break;
                const char* t0 = q;
                while (q < eol && !is_space(*q)) ++q;
                if (nf < 5) {
                    ft[nf] = t0;
                    fl[nf] = static_cast<size_t>(q - t0);
                }
                ++nf;
            }
            if (nf < 3 || nf > 4) {
                *err = path + ": incorrect face vertices number on line " + std::to_string(line_no);
                return false;
            }
            if (first_face) {
                first_face = false;
                const std::string f0(ft[0], fl[0]);
                size_t p1 = f0.find('/');
                if (f0.find("//") != std::string::npos) {
                    format = V_N;
                    mesh->has_normal = true;
                } else if (p1 == std::string::npos) {
                    format = V_ONLY;
                } else if (p1 == f0.rfind('/')) {
                    format = V_UV;
                    mesh->has_uv = true;
                } else {
                    format = V_UV_N;
                    mesh->has_normal = true;
                    mesh->has_uv = true;
                }
            }
            Corner c[4];
            for (size_t i = 0; i < nf; ++i) {
                char buf[96];   // a NUL-terminated copy: atoi / strcspn must not run into the next token
                size_t n = std::min(fl[i], sizeof(buf) - 1);
                memcpy(buf, ft[i], n);
                buf[n] = 0;
                const char* sft = buf;
                c[i].v = atoi(sft);
                c[i].n = 0;
                c[i].t = 0;
                if (format == V_UV || format == V_UV_N) {
                    sft += strcspn(sft, "/") + 1;
                    if (sft <= buf + n) c[i].t = atoi(sft);
                    if (format == V_UV_N) {
                        sft += strcspn(sft, "/") + 1;
                        if (sft <= buf + n) c[i].n = atoi(sft);
                    }
                } else if (format == V_N) {
                    sft += strcspn(sft, "/") + 2;
                    if (sft <= buf + n) c[i].n = atoi(sft);
                }
            }
            corners.push_back(c[0]); corners.push_back(c[1]); corners.push_back(c[2]);
            if (nf == 4) {
                corners.push_back(c[0]); corners.push_back(c[2]); corners.push_back(c[3]);
            }
        }
        p = eol < end ? eol + 1 : end;
    }
    int nv = static_cast<int>(vs.size() / 3), nn = static_cast<int>(ns.size() / 3),
        nt = static_cast<int>(ts.size() / 2);
    for (Corner& c : corners) {
        if (c.v < 0) c.v += nv + 1;
        if (c.n < 0) c.n += nn + 1;
        if (c.t < 0) c.t += nt + 1;
        --c.v; --c.n; --c.t;
        if (c.v < 0 || c.v >= nv || c.n < -1 || c.n >= nn || c.t < -1 || c.t >= nt) {
            *err = path + ": invalid index in face";
            return false;
        }
    }
    // (v, vn, vt) -> vertex id in first-seen order (the reference's std::map only ever looks up and inserts, so a
    // hash map numbers the vertices identically)
    struct CornerHash {
        size_t operator()(const Corner& c) const {
            uint64_t h = static_cast<uint32_t>(c.v) * 0x9E3779B97F4A7C15ull;
            h ^= (static_cast<uint64_t>(static_cast<uint32_t>(c.n)) + 0x7F4A7C15u + (h << 6) + (h >> 2));
            h ^= (static_cast<uint64_t>(static_cast<uint32_t>(c.t)) + 0x165667B1u + (h << 6) + (h >> 2));
            return static_cast<size_t>(h ^ (h >> 29));
        }
    };
    struct CornerEq {
        bool operator()(const Corner& a, const Corner& b) const { return a.v == b.v && a.n == b.n && a.t == b.t; }
    };
    std::unordered_map<Corner, uint32_t, CornerHash, CornerEq> seen;
    seen.reserve(corners.size() / 2);
    for (const Corner& c : corners) {
        auto ins = seen.insert({c, static_cast<uint32_t>(seen.size())});
        if (ins.second) {
            for (int k = 0; k < 3; ++k) mesh->pos.push_back(vs[3 * c.v + k]);
            for (int k = 0; k < 3; ++k) mesh->nrm.push_back(c.n < 0 ? 0.0f : ns[3 * c.n + k]);
            for (int k = 0; k < 2; ++k) mesh->uv.push_back(c.t < 0 ? 0.0f : ts[2 * c.t + k]);
        }
        mesh->idx.push_back(ins.first->second);
    }
    return true;
}

std::string read_file(const std::string& path, bool* ok) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) {
        *ok = false;
        return "";
    }
    std::ostringstream ss;
    ss << f.rdbuf();
    *ok = true;
    return ss.str();
}

}  // namespace

// ---------------------------------------------------------------------------
// The loaded scene: owns the arrays gbl_scene_desc points into.
// ---------------------------------------------------------------------------
struct gbl_host_scene {
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices;
    std::vector<gbl_mesh> meshes;
    std::vector<gbl_material> materials;
    std::vector<gbl_texture> textures;
    std::vector<gbl_image> images;       // MIP pyramids of image textures / image based lights (mipmap.cpp)
    std::vector<float> texels;
    std::vector<gbl_instance> instances;
    std::vector<gbl_light> lights;
    std::vector<float> density;          // heterogeneous volume: the .vol file's voxels
    std::string output_path, default_output_path = "goblin.exr";
    gbl_scene_desc desc;
};

namespace gbl_host_detail {
gbl_status load_image(const std::string& path, float** rgba, int32_t* w, int32_t* h);                        // exr_reader.cpp
gbl_image build_mipmap(std::vector<float>& pool, std::vector<float> level0, int w, int h, int channels);      // mipmap.cpp
}

namespace {

struct GeometryDecl {
    std::string type, file;
    float radius = 1.0f;
    int mesh_id = -1;
};
struct TextureDecl {
    std::string format, type;
    float color[3] = {0, 0, 0};
    float value = 0.5f;
    // checkerboard / scale (createColor/FloatCheckerboardTexture, createColor/FloatScaleTexture)
    std::string texture1, texture2, texture, scale_name, mapping;
    bool filter = false;
    float uv_scale[2] = {1.0f, 1.0f}, uv_offset[2] = {0.0f, 0.0f};
    gbl_trs to_tex;
    // image (getImageTextureParams, GoblinTexture.cpp:677-732)
    std::string file, image_filter, address, channel;
    float gamma = 1.0f, max_anisotropy = 10.0f;
    int order = 0;      // position in the "textures" list: a texture only sees the ones defined before it
    int id = -2;        // -2 unresolved, -1 constant, >= 0 index into gbl_host_scene::textures
};
struct ModelDecl {
    bool is_instance = false;
    std::string geometry, material, area_light;
};

class Loader {
public:
    Loader(const gbl_json::Value& root, const std::string& dir, gbl_host_scene* out)
        : root_(root), dir_(dir), s_(out) {}

    gbl_status run() {
        gbl_status st;
        if ((st = read_setting()) != GBL_OK) return st;
        if ((st = read_camera()) != GBL_OK) return st;
        if ((st = add_lens()) != GBL_OK) return st;
        if ((st = read_volume()) != GBL_OK) return st;
        read_geometries();
        read_textures();
        if ((st = read_primitives()) != GBL_OK) return st;
        if ((st = read_lights()) != GBL_OK) return st;
        finish();
        return GBL_OK;
    }

private:
    const gbl_json::Value& root_;
    std::string dir_;
    gbl_host_scene* s_;
    std::map<std::string, GeometryDecl> geometries_;
    std::map<std::string, TextureDecl> textures_;
    std::map<std::string, int> material_ids_;  // resolved materials by name
    std::map<std::string, ModelDecl> primitives_;
    std::map<std::string, int> area_lights_;   // name -> light index
    std::map<std::string, int32_t> image_cache_;
    std::string camera_type_;

    std::string resolve(const std::string& f) const {
        if (!f.empty() && (f[0] == '/' || (f.size() > 1 && f[1] == ':'))) return f;
        return dir_ + "/" + f;
    }

    const gbl_json::Value* list(const char* key) const {
        const gbl_json::Value* v = root_.find(key);
        return (v && v->kind == gbl_json::Value::Array) ? v : nullptr;
    }

    // createVolume (GoblinContextLoader.cpp:189-207) + createHomogeneousVolume (GoblinVolume.cpp:343-360)
    gbl_status read_volume() {
        gbl_volume& v = s_->desc.volume;
        memset(&v, 0, sizeof(v));
        const gbl_json::Value* node = root_.find("volume");
        if (!node) return GBL_OK;
        Params p(node);
        if (p.get_string("type") == "heterogeneous") return read_heterogeneous_volume(p, v);
        v.type = GBL_VOLUME_HOMOGENEOUS;   // "homogeneous" and the unknown-type fallback
        const Vec att = p.get_vec(p.vec3s, "attenuation", vec(0, 0, 0)), alb = p.get_vec(p.vec3s, "albedo", vec(0, 0, 0)),
                  emi = p.get_vec(p.vec3s, "emission", vec(0, 0, 0)), lo = p.get_vec(p.vec3s, "box_min", vec(0, 0, 0)),
                  hi = p.get_vec(p.vec3s, "box_max", vec(0, 0, 0));
        for (int i = 0; i < 3; ++i) {
            v.attenuation[i] = att.v[i];
            v.albedo[i] = alb.v[i];
            v.emission[i] = emi.v[i];
            v.box_min[i] = lo.v[i];
            v.box_max[i] = hi.v[i];
        }
        v.g = p.get_float("g", 0.0f);
        v.sample_num = p.get_int("sample_num", 5);
        read_trs(p, &v.to_world);
        return GBL_OK;
    }

    // createHeterogeneousVolume (GoblinVolume.cpp:362-384) + loadVolFile (:222-257): Mitsuba's .vol grid, float32 encoding
    // only -- any other encoding, an invalid header or a missing file falls back to the reference's one-cell grid of
    // density 1 on [-1, 1]^3.
    gbl_status read_heterogeneous_volume(Params& p, gbl_volume& v) {
        v.type = GBL_VOLUME_HETEROGENEOUS;
        int32_t nx = 1, ny = 1, nz = 1, nch = 1;
        float lo[3] = {-1.0f, -1.0f, -1.0f}, hi[3] = {1.0f, 1.0f, 1.0f};
        s_->density.assign(1, 1.0f);
        const std::string path = resolve(p.get_string("density_grid"));
        if (FILE* f = fopen(path.c_str(), "rb")) {
            struct Header {   // VolHeader, GoblinVolume.cpp:68-133 (48 bytes)
                char sig[4];
                int32_t encoding, nx, ny, nz, nch;
                float lo[3], hi[3];
            } h;
            static_assert(sizeof(Header) == 48, ".vol header");
            // a header whose dimensions overflow, exceed what the device indexes with 32-bit ints, or promise more data than the
            // file holds is not a grid: the reference's own fallback for an unreadable file (one cell of density 1) takes over
            if (fread(&h, sizeof(h), 1, f) == 1 && h.sig[0] == 'V' && h.sig[1] == 'O' && h.sig[2] == 'L' && (h.nch == 1 || h.nch == 3) && h.nx > 0 &&
                h.ny > 0 && h.nz > 0 && h.encoding == 1) {
                const uint64_t n64 = static_cast<uint64_t>(h.nx) * static_cast<uint64_t>(h.ny);
                const bool fits = n64 < (1ull << 31) && n64 * static_cast<uint64_t>(h.nz) < (1ull << 31) &&
                                  n64 * static_cast<uint64_t>(h.nz) * static_cast<uint64_t>(h.nch) < (1ull << 31);
                if (fits) {
                    const size_t n = static_cast<size_t>(n64) * h.nz * h.nch;
                    long here = ftell(f);
                    long end = here;
                    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
                        end = ftell(f);
                        fseek(f, here, SEEK_SET);
                    }
                    if (here >= 0 && end >= here && static_cast<uint64_t>(end - here) >= n * sizeof(float)) {
                        std::vector<float> data(n, 0.0f);
                        if (fread(data.data(), sizeof(float), n, f) == n) {
                            s_->density.swap(data);
                            nx = h.nx, ny = h.ny, nz = h.nz, nch = h.nch;
                            for (int i = 0; i < 3; ++i) lo[i] = h.lo[i], hi[i] = h.hi[i];
                        }
                    }
                }
            }
            fclose(f);
        }
        const Vec alb = p.get_vec(p.vec3s, "albedo", vec(0, 0, 0));
        for (int i = 0; i < 3; ++i) {
            v.albedo[i] = alb.v[i];
            v.box_min[i] = lo[i];
            v.box_max[i] = hi[i];
        }
        v.g = p.get_float("g", 0.0f);
        v.step_size = p.get_float("step_size", 0.1f);
        v.sample_num = p.get_int("sample_num", 5);
        v.grid[0] = nx, v.grid[1] = ny, v.grid[2] = nz;
        v.grid_channels = nch;
        v.density = s_->density.data();
        read_trs(p, &v.to_world);
        return GBL_OK;
    }

    // createRenderer (GoblinContextLoader.cpp:67-92) + createPathTracer / createAO
    gbl_status read_setting() {
        Params p(root_.find("render_setting"));
        std::string method = p.get_string("render_method", "path_tracing");
        gbl_render_setting& rs = s_->desc.setting;
        if (method == "ao") {
            rs.integrator = GBL_INTEGRATOR_AO;
        } else if (method == "whitted") {
            rs.integrator = GBL_INTEGRATOR_WHITTED;
        } else if (method == "light_tracing" || method == "bdpt" || method == "sppm") {
            return fail(GBL_ERR_UNSUPPORTED, "render_method \"" + method + "\" is outside the device path (path_tracing, whitted and ao only)");
        } else {
            rs.integrator = GBL_INTEGRATOR_PATH;  // "path_tracing" and the unknown-string fallback
        }
        rs.sample_per_pixel = p.get_int("sample_per_pixel", 1);
        rs.max_ray_depth = std::max(1, p.get_int("max_ray_depth", 5));
        rs.bssrdf_sample_num = p.get_int("bssrdf_sample_num", 4);
        rs.ao_sample_num = p.get_int("ao_sample_num", 25);
        rs.thread_num = p.get_int("thread_num", 0);
        return GBL_OK;
    }

    // createCamera / createFilm / createFilter (GoblinContextLoader.cpp:94-187)
    gbl_status read_camera() {
        const gbl_json::Value* cam = root_.find("camera");
        Params p(cam);
        std::string type = p.get_string("type");
        gbl_camera& c = s_->desc.camera;
        c.type = type == "orthographic" ? GBL_CAMERA_ORTHOGRAPHIC : GBL_CAMERA_PERSPECTIVE;   // unknown -> perspective
        c.film_width = p.get_float("film_width", 35.0f);
        camera_type_ = type;
        Vec pos = p.get_vec(p.vec3s, "position", vec(0, 0, 0));
        for (int i = 0; i < 3; ++i) c.position[i] = pos.v[i];
        std::string err;
        read_orientation(p, c.orientation, &err);
        c.fov_degrees = p.get_float("fov", 60.0f);
        c.near_plane = p.get_float("near_plane", 0.1f);
        c.far_plane = p.get_float("far_plane", 1000.0f);
        c.lens_radius = p.get_float("lens_radius", 0.0f);
        c.focal_distance = p.get_float("focal_distance", 1.0f);

        Params fp(cam ? cam->find("film") : nullptr);
        gbl_film& f = s_->desc.film;
        Vec res = fp.get_vec(fp.vec2s, "resolution", vec(512, 512));
        f.xres = static_cast<int>(res.v[0]);
        f.yres = static_cast<int>(res.v[1]);
        Vec crop = fp.get_vec(fp.vec4s, "crop", vec(0, 1, 0, 1));
        for (int i = 0; i < 4; ++i) f.crop[i] = crop.v[i];
        if (f.xres <= 0 || f.yres <= 0) return fail(GBL_ERR_INVALID, "film resolution must be positive");
        // createImageFilm (GoblinFilm.cpp:202-218); the default path comes from createFilm / ContextLoader::load
        // (GoblinContextLoader.cpp:127-129, 474-484): <scene file without extension>.exr
        f.tone_mapping = fp.get_bool("tone_mapping", false) ? 1u : 0u;
        f.bloom_radius = fp.get_float("bloom_radius", 0.0f);
        f.bloom_weight = fp.get_float("bloom_weight", 0.0f);
        s_->output_path = fp.get_string("file", s_->default_output_path);

        Params flt(cam ? cam->find("filter") : nullptr);
        std::string ft = flt.get_string("type");
        Vec w = flt.get_vec(flt.vec2s, "width", vec(1.0f, 1.0f));
        f.filter_width[0] = w.v[0];
        f.filter_width[1] = w.v[1];
        f.gaussian_falloff = flt.get_float("falloff", 2.0f);
        f.mitchell_b = flt.get_float("b", 2.0f);
        f.mitchell_c = flt.get_float("c", 2.0f);
        if (ft == "box") f.filter_type = GBL_FILTER_BOX;
        else if (ft == "triangle") f.filter_type = GBL_FILTER_TRIANGLE;
        else if (ft == "mitchell") f.filter_type = GBL_FILTER_MITCHELL;
        else f.filter_type = GBL_FILTER_GAUSSIAN;
        return GBL_OK;
    }

    // createCamera (GoblinContextLoader.cpp:146-176): a camera with "lens_radius" != 0 -- whatever its type --
    // puts a black-lambert Disk of that radius into the scene, instanced with the camera's own transform.
    // It is the first geometry, material and instance of the scene.
    gbl_status add_lens() {
        const gbl_json::Value* cam = root_.find("camera");
        Params p(cam);
        float lens_radius = p.get_float("lens_radius", 0.0f);
        if (lens_radius == 0.0f) return GBL_OK;
        GeometryDecl g;
        g.type = "disk";
        g.radius = lens_radius;
        const std::string gname = camera_type_ + "_lens_geom";
        geometries_.insert({gname, g});
        gbl_material black;
        memset(&black, 0, sizeof(black));
        black.type = GBL_MAT_LAMBERT;
        black.tex_color = black.tex_color2 = black.tex_exponent = black.masked_material = black.tex_color3 = black.tex_bump = black.tex_normal = -1;
        material_ids_[camera_type_ + "_lens_material"] = static_cast<int>(s_->materials.size());
        s_->materials.push_back(black);
        ModelDecl d;
        d.geometry = gname;
        d.material = camera_type_ + "_lens_material";
        primitives_.insert({camera_type_ + "_lens_model", d});
        gbl_instance inst;
        memset(&inst, 0, sizeof(inst));
        int mesh;
        gbl_status st = mesh_id(gname, &mesh);
        if (st != GBL_OK) return st;
        inst.mesh = mesh;
        inst.material = material_ids_[d.material];
        inst.area_light = -1;
        read_trs(p, &inst.to_world);   // createInstance(cameraParams): position / orientation / scale of the camera block
        s_->instances.push_back(inst);
        return GBL_OK;
    }

    void read_geometries() {
        const gbl_json::Value* l = list("geometries");
        if (!l) return;
        for (const gbl_json::Value& g : l->arr) {
            Params p(&g);
            GeometryDecl d;
            d.type = p.get_string("type");
            d.file = p.get_string("file");
            d.radius = p.get_float("radius", 1.0f);
            geometries_.insert({p.get_string("name"), d});
        }
    }

    void read_textures() {
        const gbl_json::Value* l = list("textures");
        if (!l) return;
        int order = 0;
        for (const gbl_json::Value& t : l->arr) {
            Params p(&t);
            TextureDecl d;
            d.type = p.get_string("type");
            d.format = p.get_string("format", "color");
            Vec c = p.get_vec(p.vec3s, "color", vec(0, 0, 0));
            for (int i = 0; i < 3; ++i) d.color[i] = c.v[i];
            d.value = p.get_float("float", 0.5f);
            d.texture1 = p.get_string("texture1");
            d.texture2 = p.get_string("texture2");
            d.texture = p.get_string("texture");
            d.scale_name = p.get_string("scale");
            d.mapping = p.get_string("mapping", "uv");
            d.filter = p.get_bool("filter", false);            // checkerboard: a bool; image: a string (ParamSet keeps types apart)
            d.file = p.get_string("file");
            d.image_filter = p.get_string("filter", "nearest");
            d.address = p.get_string("address", "repeat");
            d.channel = p.get_string("channel", "All");
            d.gamma = p.get_float("gamma", 1.0f);
            d.max_anisotropy = p.get_float("max_anisotropy", 10.0f);
            Vec sc = p.get_vec(p.vec2s, "scale", vec(1.0f, 1.0f));      // UVMapping (getTextureMapping, :599-615)
            Vec of = p.get_vec(p.vec2s, "offset", vec(0.0f, 0.0f));
            for (int i = 0; i < 2; ++i) {
                d.uv_scale[i] = sc.v[i];
                d.uv_offset[i] = of.v[i];
            }
            read_trs(p, &d.to_tex);                                     // SphericalMapping: getTransform(params)
            d.order = order++;
            // color and float textures live in separate maps in the reference
            textures_.insert({d.format + ":" + p.get_string("name"), d});
        }
    }

    // Resolve a texture by name into either a constant (returns -1, value in `constant`) or an entry of
    // gbl_host_scene::textures.  `before` is the list position of the texture asking (a checkerboard / scale
    // texture is created while the list is being read, so it only finds the ones defined earlier,
    // createTextures GoblinContextLoader.cpp:245-307); materials see the whole list.
    gbl_status texture_ref(bool is_float, const std::string& name, int before, int* id, float constant[3]) {
        auto it = textures_.find(std::string(is_float ? "float:" : "color:") + name);
        if (it == textures_.end() || it->second.order >= before) return fail(GBL_ERR_INVALID, "Texture " + name + " not defined!");
        TextureDecl& d = it->second;
        const std::string& t = d.type;
        if (t == "image") {
            if (d.id == -2) {
                gbl_texture g;
                memset(&g, 0, sizeof(g));
                g.type = GBL_TEX_IMAGE;
                g.is_float = is_float ? 1u : 0u;
                g.child[0] = g.child[1] = -1;
                set_mapping(d, &g);
                g.image_filter = d.image_filter == "bilinear" ? GBL_IMAGE_FILTER_BILINEAR : d.image_filter == "trilinear" ? GBL_IMAGE_FILTER_TRILINEAR
                                 : d.image_filter == "EWA" ? GBL_IMAGE_FILTER_EWA : GBL_IMAGE_FILTER_NONE;   // "nearest" and the unrecognised-filter fallback
                g.address = d.address == "clamp" ? GBL_ADDRESS_CLAMP : d.address == "border" ? GBL_ADDRESS_BORDER : GBL_ADDRESS_REPEAT;
                // createColorImageTexture does not forward max_anisotropy: colour images always get the default (:741-745)
                g.max_anisotropy = is_float ? d.max_anisotropy : 10.0f;
                int channel = 4;   // ChannelAll, also the unrecognised-channel fallback
                if (d.channel == "R") channel = 0;
                else if (d.channel == "G") channel = 1;
                else if (d.channel == "B") channel = 2;
                else if (d.channel == "A") channel = 3;
                gbl_status st = image_ref(resolve(d.file), is_float, d.gamma, channel, nullptr, &g.image);
                if (st != GBL_OK) return st;
                d.id = static_cast<int>(s_->textures.size());
                s_->textures.push_back(g);
            }
            *id = d.id;
            return GBL_OK;
        }
        if (t != "checkerboard" && t != "scale") {   // "constant" and the unknown-type fallback (:282-285, :300-303)
            *id = -1;
            if (is_float) constant[0] = constant[1] = constant[2] = d.value;
            else for (int i = 0; i < 3; ++i) constant[i] = d.color[i];
            return GBL_OK;
        }
        if (d.id == -2) {
            gbl_texture g;
            memset(&g, 0, sizeof(g));
            g.is_float = is_float ? 1u : 0u;
            g.image = -1;
            gbl_status st;
            float c0[3], c1[3];
            int i0, i1;
            if (t == "checkerboard") {
                g.type = GBL_TEX_CHECKERBOARD;
                if ((st = texture_ref(is_float, d.texture1, d.order, &i0, c0)) != GBL_OK) return st;
                if ((st = texture_ref(is_float, d.texture2, d.order, &i1, c1)) != GBL_OK) return st;
                set_mapping(d, &g);
                g.filter = d.filter ? 1u : 0u;
            } else {
                g.type = GBL_TEX_SCALE;
                if ((st = texture_ref(is_float, d.texture, d.order, &i0, c0)) != GBL_OK) return st;
                if ((st = texture_ref(true, d.scale_name, d.order, &i1, c1)) != GBL_OK) return st;
            }
            // constant children become constant texture entries so every child is an index
            g.child[0] = i0 >= 0 ? i0 : add_constant(is_float, c0);
            g.child[1] = i1 >= 0 ? i1 : add_constant(t == "scale" ? true : is_float, c1);
            d.id = static_cast<int>(s_->textures.size());
            s_->textures.push_back(g);
        }
        *id = d.id;
        return GBL_OK;
    }

    // getTextureMapping (GoblinTexture.cpp:599-615)
    static void set_mapping(const TextureDecl& d, gbl_texture* g) {
        if (d.mapping == "spherical") {
            g->mapping = GBL_MAP_SPHERICAL;
        } else {   // "uv" and the unknown-mapping fallback: UVMapping((1,1), (0,0)) (:609-613)
            g->mapping = GBL_MAP_UV;
            const bool known = d.mapping == "uv";
            for (int i = 0; i < 2; ++i) {
                g->uv_scale[i] = known ? d.uv_scale[i] : 1.0f;
                g->uv_offset[i] = known ? d.uv_offset[i] : 0.0f;
            }
        }
        g->to_tex = d.to_tex;
    }

    // ImageTexture<T>::getMIPMap (GoblinTexture.cpp:458-487) and ImageBasedLight's constructor (GoblinLight.cpp:475-487):
    // load the file, convert every texel (convertTexel<float / Color>, :489-523; or the light's colour filter), build
    // the pyramid.  Pyramids are cached by (file, gamma, channel, format) like ImageTexture::imageCache.  A file that
    // cannot be read is an error here (the reference substitutes a 1 x 1 magenta image and renders on).
    gbl_status image_ref(const std::string& path, bool is_float, float gamma, int channel, const float* light_filter, int32_t* out) {
        char key[64];
        snprintf(key, sizeof(key), "|%d|%d|%.9g|", is_float ? 1 : 0, channel, gamma);
        std::string k = path + key;
        if (light_filter) {
            snprintf(key, sizeof(key), "L%.9g,%.9g,%.9g", light_filter[0], light_filter[1], light_filter[2]);
            k += key;
        }
        auto it = image_cache_.find(k);
        if (it != image_cache_.end()) {
            *out = it->second;
            return GBL_OK;
        }
        float* rgba = nullptr;
        int32_t w = 0, h = 0;
        gbl_status st = gbl_host_detail::load_image(path, &rgba, &w, &h);
        if (st != GBL_OK) return st;
        const size_t n = static_cast<size_t>(w) * h;
        const int channels = is_float ? 1 : 4;
        std::vector<float> level0(n * channels);
        for (size_t i = 0; i < n; ++i) {
            const float* in = rgba + 4 * i;
            if (light_filter) {          // buffer[i] *= filter
                for (int c = 0; c < 3; ++c) level0[4 * i + c] = in[c] * light_filter[c];
                level0[4 * i + 3] = in[3];
            } else if (is_float) {
                const float v = channel == 4 ? 0.212671f * in[0] + 0.715160f * in[1] + 0.072169f * in[2] : in[channel];
                level0[i] = powf(v, gamma);
            } else {
                float c[4] = {in[0], in[1], in[2], in[3]};
                if (channel != 4) {      // Color(in.<channel>): grey, alpha 1
                    c[0] = c[1] = c[2] = in[channel];
                    c[3] = 1.0f;
                }
                for (int j = 0; j < 3; ++j) level0[4 * i + j] = gamma == 1.0f ? c[j] : powf(c[j], gamma);
                level0[4 * i + 3] = c[3];
            }
        }
        free(rgba);
        s_->images.push_back(gbl_host_detail::build_mipmap(s_->texels, std::move(level0), w, h, channels));
        *out = static_cast<int32_t>(s_->images.size()) - 1;
        image_cache_[k] = *out;
        return GBL_OK;
    }

    int add_constant(bool is_float, const float c[3]) {
        gbl_texture g;
        memset(&g, 0, sizeof(g));
        g.type = GBL_TEX_CONSTANT;
        g.is_float = is_float ? 1u : 0u;
        for (int i = 0; i < 3; ++i) g.value[i] = c[i];
        g.child[0] = g.child[1] = -1;
        g.image = -1;
        s_->textures.push_back(g);
        return static_cast<int>(s_->textures.size()) - 1;
    }

    // BSSRDF(Kd, diffuseMeanFreePath, eta, g) -> sigma_a, sigma_s' (BSSRDF::convertFromDiffuse / diffuseReflectance /
    // Fdr, GoblinMaterial.cpp:176-220, GoblinMaterial.h:94-105): 16 bisection steps on alpha' per channel.
    static float bssrdf_fdr(float eta) {
        if (eta < 1.0f) return -0.4399f + 0.7099f / eta - 0.3319f / (eta * eta) + 0.0636f / (eta * eta * eta);
        return -1.4399f / (eta * eta) + 0.7099f / eta + 0.6681f + 0.0636f * eta;
    }
    static void convert_from_diffuse(const float kd[3], const float mean_free_path[3], float eta, float absorb[3], float scatter_prime[3]) {
        const float fdr = bssrdf_fdr(eta);
        const float A = (1.0f + fdr) / (1.0f - fdr);
        for (int i = 0; i < 3; ++i) {
            const float sigma_tr = 1.0f / mean_free_path[i];
            float lo = 0.0f, hi = 1.0f;
            for (int j = 0; j < 16; ++j) {
                const float mid = 0.5f * (lo + hi);
                const float sq = sqrtf(3.0f * (1.0f - mid));
                const float rd = 0.5f * mid * (1.0f + expf(-(4.0f / 3.0f) * A * sq)) * expf(-sq);
                if (rd > kd[i]) hi = mid; else lo = mid;
            }
            const float alpha = 0.5f * (lo + hi);
            const float sigma_tp = sigma_tr / sqrtf(3.0f * (1.0f - alpha));
            scatter_prime[i] = alpha * sigma_tp;
            absorb[i] = sigma_tp - scatter_prime[i];
        }
    }

    gbl_status color_texture(const std::string& name, float out[3], int32_t* tex) {
        int id;
        gbl_status st = texture_ref(false, name, 1 << 30, &id, out);
        if (st != GBL_OK) return st;
        *tex = id;
        if (id >= 0) out[0] = out[1] = out[2] = 0.0f;
        return GBL_OK;
    }

    gbl_status float_texture(const std::string& name, float* out, int32_t* tex) {
        int id;
        float c[3] = {0, 0, 0};
        gbl_status st = texture_ref(true, name, 1 << 30, &id, c);
        if (st != GBL_OK) return st;
        *tex = id;
        *out = id >= 0 ? 0.0f : c[0];
        return GBL_OK;
    }

    // createMaterials (GoblinContextLoader.cpp:309-350) + the create*Material
    // factories (GoblinMaterial.cpp:825-877), resolved on first use.
    gbl_status material_id(const std::string& name, int* out) {
        auto hit = material_ids_.find(name);
        if (hit != material_ids_.end()) {
            *out = hit->second;
            return GBL_OK;
        }
        const gbl_json::Value* l = list("materials");
        const gbl_json::Value* decl = nullptr;
        if (l) {
            for (const gbl_json::Value& m : l->arr) {
                Params p(&m);
                if (p.get_string("name") == name) {
                    decl = &m;
                    break;  // first definition wins
                }
            }
        }
        if (!decl) return fail(GBL_ERR_INVALID, "Material " + name + " not defined!");
        Params p(decl);
        std::string type = p.get_string("type");
        gbl_material m;
        memset(&m, 0, sizeof(m));
        m.tex_color = m.tex_color2 = m.tex_exponent = m.masked_material = m.tex_color3 = m.tex_bump = m.tex_normal = -1;
        gbl_status st;
        if (type != "mask") {
            // getBumpShaders (GoblinMaterial.cpp:813-824): every factory but the mask's reads "bumpmap" (float texture) and
            // "normalmap" (colour texture); a constant texture perturbs nothing but is still looked up
            float unused[3];
            if (p.has_string("bumpmap")) {
                if ((st = texture_ref(true, p.get_string("bumpmap"), 1 << 30, &m.tex_bump, unused)) != GBL_OK) return st;
                if (m.tex_bump < 0) m.tex_bump = add_constant(true, unused);
            }
            if (p.has_string("normalmap")) {
                if ((st = texture_ref(false, p.get_string("normalmap"), 1 << 30, &m.tex_normal, unused)) != GBL_OK) return st;
                if (m.tex_normal < 0) m.tex_normal = add_constant(false, unused);
            }
        }
        if (type == "blinn") {
            m.type = GBL_MAT_BLINN;
            if ((st = color_texture(p.get_string("Kg"), m.color, &m.tex_color)) != GBL_OK) return st;
            if ((st = float_texture(p.get_string("exponent"), &m.exponent, &m.tex_exponent)) != GBL_OK) return st;
            m.index = p.get_float("index", 1.5f);
            m.k = p.get_float("k", -1.0f);
        } else if (type == "transparent") {
            m.type = GBL_MAT_TRANSPARENT;
            if ((st = color_texture(p.get_string("Kr"), m.color, &m.tex_color)) != GBL_OK) return st;
            if ((st = color_texture(p.get_string("Kt"), m.color2, &m.tex_color2)) != GBL_OK) return st;
            m.index = p.get_float("index", 1.5f);
        } else if (type == "mirror") {
            m.type = GBL_MAT_MIRROR;
            if ((st = color_texture(p.get_string("Kr"), m.color, &m.tex_color)) != GBL_OK) return st;
            m.index = p.get_float("index", 0.8f);
            m.k = p.get_float("k", 6.0f);
        } else if (type == "mask") {
            // createMaskMaterial (GoblinMaterial.cpp:929-950).  The wrapped material is looked up while the list is
            // being read, so it has to be defined earlier in "materials".
            m.type = GBL_MAT_MASK;
            m.exponent = 1.0f;
            m.color[0] = m.color[1] = m.color[2] = 1.0f;
            if (p.has_string("alpha") && (st = float_texture(p.get_string("alpha"), &m.exponent, &m.tex_exponent)) != GBL_OK) return st;
            if (p.has_string("transparent_color") &&
                (st = color_texture(p.get_string("transparent_color"), m.color, &m.tex_color)) != GBL_OK) return st;
            const std::string inner = p.get_string("material");
            bool earlier = false;
            for (const gbl_json::Value& mv : l->arr) {
                if (&mv == decl) break;
                if (Params(&mv).get_string("name") == inner) earlier = true;
            }
            if (!earlier) return fail(GBL_ERR_INVALID, "Material " + inner + " not defined!");
            int inner_id;
            if ((st = material_id(inner, &inner_id)) != GBL_OK) return st;
            if (s_->materials[inner_id].type == GBL_MAT_MASK || s_->materials[inner_id].type == GBL_MAT_SUBSURFACE)
                return fail(GBL_ERR_UNSUPPORTED, "a mask material wrapping a mask or a subsurface material is outside the device path");
            m.masked_material = inner_id;
        } else if (type == "subsurface") {
            // createSubsurfaceMaterial (GoblinMaterial.cpp:881-927)
            m.type = GBL_MAT_SUBSURFACE;
            m.index = p.get_float("index", 1.5f);
            m.k = p.get_float("g", 0.0f);
            m.color3[0] = m.color3[1] = m.color3[2] = 1.0f;
            if (p.has_string("Kr") && (st = color_texture(p.get_string("Kr"), m.color3, &m.tex_color3)) != GBL_OK) return st;
            if (p.has_vec3("Kd")) {
                const Vec kd = p.get_vec(p.vec3s, "Kd", vec(0, 0, 0));
                const Vec mfp = p.get_vec(p.vec3s, "mean_free_path", vec(1.0f, 1.0f, 1.0f));
                convert_from_diffuse(kd.v, mfp.v, m.index, m.color, m.color2);
            } else {
                const float marble_absorb[3] = {0.0021f, 0.0041f, 0.0071f}, marble_scatterp[3] = {2.19f, 2.62f, 3.00f};
                for (int i = 0; i < 3; ++i) {
                    m.color[i] = marble_absorb[i];
                    m.color2[i] = marble_scatterp[i];
                }
                if (p.has_string("absorb") && (st = color_texture(p.get_string("absorb"), m.color, &m.tex_color)) != GBL_OK) return st;
                if (p.has_string("scatter_prime") &&
                    (st = color_texture(p.get_string("scatter_prime"), m.color2, &m.tex_color2)) != GBL_OK) return st;
            }
        } else {  // "lambert" and the unknown-type fallback
            m.type = GBL_MAT_LAMBERT;
            if ((st = color_texture(p.get_string("Kd"), m.color, &m.tex_color)) != GBL_OK) return st;
        }
        *out = static_cast<int>(s_->materials.size());
        s_->materials.push_back(m);
        material_ids_[name] = *out;
        return GBL_OK;
    }

    gbl_status mesh_id(const std::string& geometry, int* out) {
        auto it = geometries_.find(geometry);
        if (it == geometries_.end()) return fail(GBL_ERR_INVALID, "Geometry " + geometry + " not defined!");
        GeometryDecl& g = it->second;
        if (g.mesh_id < 0 && g.type != "mesh") {
            // createGeometries (:226-234): "sphere", "disk", and a sphere for any unknown type
            gbl_mesh m;
            memset(&m, 0, sizeof(m));
            m.shape = g.type == "disk" ? GBL_SHAPE_DISK : GBL_SHAPE_SPHERE;
            m.radius = g.radius;
            m.vertex_offset = static_cast<uint32_t>(s_->positions.size() / 3);
            m.tri_offset = static_cast<uint32_t>(s_->indices.size() / 3);
            g.mesh_id = static_cast<int>(s_->meshes.size());
            s_->meshes.push_back(m);
        }
        if (g.mesh_id < 0) {
            MeshData md;
            std::string err;
            if (!load_obj(resolve(g.file), &md, &err)) return fail(GBL_ERR_IO, err);
            if (md.idx.empty()) return fail(GBL_ERR_INVALID, "mesh " + g.file + " has no faces");
            gbl_mesh m;
            memset(&m, 0, sizeof(m));
            m.shape = GBL_SHAPE_MESH;
            m.vertex_offset = static_cast<uint32_t>(s_->positions.size() / 3);
            m.vertex_count = static_cast<uint32_t>(md.pos.size() / 3);
            m.tri_offset = static_cast<uint32_t>(s_->indices.size() / 3);
            m.tri_count = static_cast<uint32_t>(md.idx.size() / 3);
            m.has_normal = md.has_normal;
            m.has_uv = md.has_uv;
            s_->positions.insert(s_->positions.end(), md.pos.begin(), md.pos.end());
            s_->normals.insert(s_->normals.end(), md.nrm.begin(), md.nrm.end());
            s_->uvs.insert(s_->uvs.end(), md.uv.begin(), md.uv.end());
            s_->indices.insert(s_->indices.end(), md.idx.begin(), md.idx.end());
            g.mesh_id = static_cast<int>(s_->meshes.size());
            s_->meshes.push_back(m);
        }
        *out = g.mesh_id;
        return GBL_OK;
    }

    // createPrimitives (GoblinContextLoader.cpp:352-387): models and instances
    // share one name map; only instances reach the scene (:381-383, :498).
    gbl_status read_primitives() {
        const gbl_json::Value* l = list("primitives");
        if (!l) return GBL_OK;
        for (const gbl_json::Value& pv : l->arr) {
            Params p(&pv);
            std::string type = p.get_string("type");
            std::string name = p.get_string("name");
            if (type == "instance") {
                gbl_status st = add_instance(p, -1);
                if (st != GBL_OK) return st;
                ModelDecl d;
                d.is_instance = true;
                primitives_.insert({name, d});
            } else {  // "model" and the unknown-type fallback
                ModelDecl d;
                d.geometry = p.get_string("geometry");
                d.material = p.get_string("material");
                d.area_light = p.get_string("area_light");
                primitives_.insert({name, d});
            }
        }
        return GBL_OK;
    }

    gbl_status add_instance(const Params& p, int area_light) {
        std::string model = p.get_string("model");
        auto it = primitives_.find(model);
        if (it == primitives_.end()) return fail(GBL_ERR_INVALID, "Primitive " + model + " not defined!");
        if (it->second.is_instance) return fail(GBL_ERR_UNSUPPORTED, "instance of an instance (\"" + model + "\") is outside the device path");
        const ModelDecl& m = it->second;
        // A user model's "area_light" can never resolve in the reference: models
        // are created before lights (GoblinContextLoader.cpp:494-495), so
        // SceneCache::getAreaLight falls back to its nullptr entry
        // (GoblinScene.cpp:219-226).  Mirror that: not emissive.
        if (!m.area_light.empty() && area_light < 0) {
            auto al = area_lights_.find(m.area_light);
            if (al != area_lights_.end()) area_light = al->second;
        }
        gbl_instance inst;
        memset(&inst, 0, sizeof(inst));
        int mesh, mat;
        gbl_status st;
        if ((st = mesh_id(m.geometry, &mesh)) != GBL_OK) return st;
        if ((st = material_id(m.material, &mat)) != GBL_OK) return st;
        inst.mesh = mesh;
        inst.material = mat;
        inst.area_light = area_light;
        read_trs(p, &inst.to_world);
        s_->instances.push_back(inst);
        return GBL_OK;
    }

    // createLights (GoblinContextLoader.cpp:389-445) + factories
    // (GoblinLight.cpp:630-676).
    gbl_status read_lights() {
        const gbl_json::Value* l = list("lights");
        if (!l) return GBL_OK;
        for (const gbl_json::Value& lv : l->arr) {
            Params p(&lv);
            std::string type = p.get_string("type");
            std::string name = p.get_string("name");
            gbl_light lt;
            memset(&lt, 0, sizeof(lt));
            lt.sample_num = 1;   // Light::getSamplesNum (GoblinLight.h:124); area lights read "sample_num"
            lt.to_world.orientation[0] = 1.0f;
            lt.to_world.scale[0] = lt.to_world.scale[1] = lt.to_world.scale[2] = 1.0f;
            lt.image = -1;
            if (type == "ibl") {
                lt.type = GBL_LIGHT_IBL;   // createImageBasedLight, GoblinLight.cpp:681-691
                Vec F = p.get_vec(p.vec3s, "filter", vec(0, 0, 0));
                for (int i = 0; i < 3; ++i) lt.color[i] = F.v[i];
                std::string err;
                read_orientation(p, lt.to_world.orientation, &err);
                lt.sample_num = static_cast<uint32_t>(std::max(0, p.get_int("sample_num", 1)));
                gbl_status st = image_ref(resolve(p.get_string("file")), false, 1.0f, 4, lt.color, &lt.image);
                if (st != GBL_OK) return st;
            } else if (type == "directional") {
                lt.type = GBL_LIGHT_DIRECTIONAL;   // createDirectionalLight, GoblinLight.cpp:640-646
                Vec R = p.get_vec(p.vec3s, "radiance", vec(0, 0, 0));
                Vec D = p.get_vec(p.vec3s, "direction", vec(0, 0, 0));
                for (int i = 0; i < 3; ++i) {
                    lt.color[i] = R.v[i];
                    lt.direction[i] = D.v[i];
                }
            } else if (type == "spot") {
                lt.type = GBL_LIGHT_SPOT;
                Vec I = p.get_vec(p.vec3s, "intensity", vec(0, 0, 0));
                Vec P = p.get_vec(p.vec3s, "position", vec(0, 0, 0));
                float dir[3];
                if (p.has_vec3("target")) {
                    Vec T = p.get_vec(p.vec3s, "target", vec(0, 0, 0));
                    float d[3] = {T.v[0] - P.v[0], T.v[1] - P.v[1], T.v[2] - P.v[2]};
                    float inv = 1.0f / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    for (int i = 0; i < 3; ++i) dir[i] = d[i] * inv;
                } else {
                    Vec D = p.get_vec(p.vec3s, "direction", vec(0, 0, 0));
                    for (int i = 0; i < 3; ++i) dir[i] = D.v[i];
                }
                for (int i = 0; i < 3; ++i) {
                    lt.color[i] = I.v[i];
                    lt.position[i] = P.v[i];
                    lt.direction[i] = dir[i];
                }
                lt.cos_theta_max = std::cos(radians(p.get_float("theta_max")));
                lt.cos_falloff_start = std::cos(radians(p.get_float("falloff_start")));
            } else if (type == "area") {
                lt.type = GBL_LIGHT_AREA;
                Vec R = p.get_vec(p.vec3s, "radiance", vec(0, 0, 0));
                for (int i = 0; i < 3; ++i) lt.color[i] = R.v[i];
                int mesh;
                gbl_status st = mesh_id(p.get_string("geometry"), &mesh);
                if (st != GBL_OK) return st;
                lt.mesh = mesh;
                read_trs(p, &lt.to_world);
                lt.sample_num = static_cast<uint32_t>(std::max(0, p.get_int("sample_num", 1)));   // createAreaLight, GoblinLight.cpp:675
            } else {  // "point" and the unknown-type fallback
                lt.type = GBL_LIGHT_POINT;
                Vec I = p.get_vec(p.vec3s, "intensity", vec(0, 0, 0));
                Vec P = p.get_vec(p.vec3s, "position", vec(0, 0, 0));
                for (int i = 0; i < 3; ++i) {
                    lt.color[i] = I.v[i];
                    lt.position[i] = P.v[i];
                }
            }
            int light_index = static_cast<int>(s_->lights.size());
            s_->lights.push_back(lt);
            if (type == "area") {
                // the loader auto-creates an emissive model (black lambert) and
                // instances it with the light's own transform (:419-441)
                area_lights_.insert({name, light_index});
                gbl_material black;
                memset(&black, 0, sizeof(black));
                black.type = GBL_MAT_LAMBERT;
                black.tex_color = black.tex_color2 = black.tex_exponent = black.masked_material = black.tex_color3 = black.tex_bump = black.tex_normal = -1;
                gbl_instance inst;
                memset(&inst, 0, sizeof(inst));
                inst.mesh = lt.mesh;
                inst.material = static_cast<uint32_t>(s_->materials.size());
                s_->materials.push_back(black);
                inst.area_light = light_index;
                inst.to_world = lt.to_world;
                s_->instances.push_back(inst);
            }
        }
        return GBL_OK;
    }

    void finish() {
        gbl_scene_desc& d = s_->desc;
        d.abi_version = GBL_ABI_VERSION;
        d.num_vertices = static_cast<uint32_t>(s_->positions.size() / 3);
        d.positions = s_->positions.data();
        d.normals = s_->normals.data();
        d.uvs = s_->uvs.data();
        d.num_triangles = static_cast<uint32_t>(s_->indices.size() / 3);
        d.indices = s_->indices.data();
        d.num_meshes = static_cast<uint32_t>(s_->meshes.size());
        d.meshes = s_->meshes.data();
        d.num_materials = static_cast<uint32_t>(s_->materials.size());
        d.materials = s_->materials.data();
        d.num_textures = static_cast<uint32_t>(s_->textures.size());
        d.textures = s_->textures.data();
        d.num_images = static_cast<uint32_t>(s_->images.size());
        d.images = s_->images.data();
        d.num_texels = s_->texels.size();
        d.texels = s_->texels.data();
        d.num_instances = static_cast<uint32_t>(s_->instances.size());
        d.instances = s_->instances.data();
        d.num_lights = static_cast<uint32_t>(s_->lights.size());
        d.lights = s_->lights.data();
    }
};

int round_to_square(int n) {
    int s = static_cast<int>(std::ceil(std::sqrt(static_cast<float>(n))));
    return s * s;
}

}  // namespace

extern "C" {

static gbl_status load_text(const char* json_text, const char* scene_dir, const std::string& default_output, gbl_host_scene** out);

static gbl_status gbl_host_load_string_impl(const char* json_text, const char* scene_dir, gbl_host_scene** out) {
    return load_text(json_text, scene_dir, "goblin.exr", out);   // createImageFilm's own default (GoblinFilm.cpp:212)
}
gbl_status gbl_host_load_string(const char* json_text, const char* scene_dir, gbl_host_scene** out) {
    return gbl_guard([&] { return gbl_host_load_string_impl(json_text, scene_dir, out); }, [&](const std::string& what) { g_last_error = what; });
}

static gbl_status load_text(const char* json_text, const char* scene_dir, const std::string& default_output, gbl_host_scene** out) {
    if (!json_text || !out) return fail(GBL_ERR_INVALID, "null argument");
    *out = nullptr;
    gbl_json::Value root;
    std::string err;
    gbl_json::Parser parser(json_text, strlen(json_text));
    if (!parser.parse(&root, &err)) return fail(GBL_ERR_IO, "json parse error: " + err);
    if (root.kind != gbl_json::Value::Object) return fail(GBL_ERR_IO, "scene json must be an object");
    gbl_host_scene* s = new gbl_host_scene();
    s->default_output_path = default_output;
    memset(&s->desc, 0, sizeof(s->desc));
    Loader loader(root, scene_dir ? scene_dir : ".", s);
    gbl_status st = loader.run();
    if (st != GBL_OK) {
        delete s;
        return st;
    }
    *out = s;
    return GBL_OK;
}

static gbl_status gbl_host_load_file_impl(const char* json_path, gbl_host_scene** out) {
    if (!json_path || !out) return fail(GBL_ERR_INVALID, "null argument");
    *out = nullptr;
    bool ok;
    std::string text = read_file(json_path, &ok);
    if (!ok) return fail(GBL_ERR_IO, std::string("error reading scene file: ") + json_path);
    std::string path(json_path), dir = ".";
    size_t cut = path.find_last_of('/');
    if (cut == std::string::npos) cut = path.find_last_of('\\');
    if (cut != std::string::npos) dir = path.substr(0, cut);
    // default output path (GoblinContextLoader.cpp:474-484)
    std::string def;
    size_t ext = path.find_last_of('.');
    if (ext != std::string::npos && ext < path.length() - 1 && path[ext + 1] != '/' && path[ext + 1] != '\\') def = path.substr(0, ext) + ".exr";
    else def = path + ".exr";
    return load_text(text.c_str(), dir.c_str(), def, out);
}
gbl_status gbl_host_load_file(const char* json_path, gbl_host_scene** out) {
    return gbl_guard([&] { return gbl_host_load_file_impl(json_path, out); }, [&](const std::string& what) { g_last_error = what; });
}

const char* gbl_host_output_path(const gbl_host_scene* scene) { return scene ? scene->output_path.c_str() : ""; }

const gbl_scene_desc* gbl_host_desc(const gbl_host_scene* scene) { return scene ? &scene->desc : nullptr; }

void gbl_host_free(gbl_host_scene* scene) { delete scene; }

const char* gbl_host_last_error(void) { return g_last_error.c_str(); }

void gbl_host_sample_window(const gbl_film* film, int32_t out[4]) {
    // Film ctor + Film::getSampleRange (GoblinFilm.cpp:92-112,131-138)
    int xs = static_cast<int>(std::ceil(film->xres * film->crop[0]));
    int xc = std::max(1, static_cast<int>(std::ceil(film->xres * film->crop[1])) - xs);
    int ys = static_cast<int>(std::ceil(film->yres * film->crop[2]));
    int yc = std::max(1, static_cast<int>(std::ceil(film->yres * film->crop[3])) - ys);
    out[0] = static_cast<int>(std::floor(xs + 0.5f - film->filter_width[0]));
    out[1] = static_cast<int>(std::floor(xs + 0.5f + xc + film->filter_width[0]));
    out[2] = static_cast<int>(std::floor(ys + 0.5f - film->filter_width[1]));
    out[3] = static_cast<int>(std::floor(ys + 0.5f + yc + film->filter_width[1]));
}

int32_t gbl_host_round_to_square(int32_t n) { return round_to_square(n); }

int32_t gbl_host_sample_dimension(const gbl_render_setting* rs) {
    if (rs->integrator == GBL_INTEGRATOR_AO) {
        // one 2D pattern of roundToSquare(roundToSquare(n)) points (GoblinAO.cpp:39-42)
        return 4 + 2 * round_to_square(round_to_square(rs->ao_sample_num));
    }
    // per bounce: light{1D,2D} + bsdf{1D,2D} + pick{1D} with n=1 -> 7 floats;
    // then the BSSRDF block: 4 x 1D + 2 x 2D patterns of n' points
    int depth = std::max(1, rs->max_ray_depth);
    int n1 = round_to_square(rs->bssrdf_sample_num);
    int n2 = round_to_square(n1);
    return 4 + 7 * depth + 4 * n1 + 2 * 2 * n2;
}

int32_t gbl_host_sample_dimension_scene(const gbl_scene_desc* desc, const gbl_render_setting* rs) {
    if (rs->integrator != GBL_INTEGRATOR_WHITTED) return gbl_host_sample_dimension(rs);
    // per light: LightSampleIndex + BSDFSampleIndex of getSamplesNum() points (1D + 2D each); one pick 1D; the BSSRDF block
    int dims = 4;
    for (uint32_t i = 0; i < desc->num_lights; ++i) {
        const int n1 = round_to_square(static_cast<int>(desc->lights[i].sample_num));
        const int n2 = round_to_square(n1);
        dims += 2 * n1 + 2 * 2 * n2;
    }
    const int b1 = round_to_square(rs->bssrdf_sample_num), b2 = round_to_square(b1);
    return dims + 1 + 4 * b1 + 2 * 2 * b2;
}

void gbl_host_film_normalize(const float* accum, int32_t xres, int32_t yres, float* rgb_out) {
    // Color / float multiplies by the reciprocal (GoblinColor.h:76-79)
    for (int64_t i = 0; i < static_cast<int64_t>(xres) * yres; ++i) {
        float inv = 1.0f / accum[4 * i + 3];
        rgb_out[3 * i + 0] = accum[4 * i + 0] * inv;
        rgb_out[3 * i + 1] = accum[4 * i + 1] * inv;
        rgb_out[3 * i + 2] = accum[4 * i + 2] * inv;
    }
}

static gbl_status gbl_host_write_pfm_impl(const char* path, const float* rgb, int32_t xres, int32_t yres) {
    FILE* f = fopen(path, "wb");
    if (!f) return fail(GBL_ERR_IO, std::string("can't open ") + path);
    fprintf(f, "PF\n%d %d\n-1.0\n", xres, yres);
    for (int y = yres - 1; y >= 0; --y) fwrite(rgb + 3 * static_cast<size_t>(y) * xres, sizeof(float), 3 * static_cast<size_t>(xres), f);
    fclose(f);
    return GBL_OK;
}
gbl_status gbl_host_write_pfm(const char* path, const float* rgb, int32_t xres, int32_t yres) {
    return gbl_guard([&] { return gbl_host_write_pfm_impl(path, rgb, xres, yres); }, [&](const std::string& what) { g_last_error = what; });
}

}  // extern "C"
