// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// CPU restatement ("oracle") of the reference's path-tracing hot path, written
// from the reference's published behaviour in its float operation order.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; nothing under goblin_amd/ links, imports or calls it.
//
// Parity pin: PINNED.  The reference has no tests or golden vectors of its own
// (SURVEY.md 8c), so this file is pinned against the real reference compiled
// from /root/reference/src by oracle/Makefile (oracle/_ref/ref_harness): the
// fixtures under tests/golden/ were captured from that build by
// tests/golden/make_golden.py, and tests/test_oracle_vs_reference.py checks this
// restatement against them (Film accumulators, per-sample (Sample -> Li) pairs,
// camera rays, filter table, light power, closest hits).
//
// Every function cites the reference file:line it restates
// (paths relative to /root/reference/src).
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off (x86-64 SSE2, no FMA contraction,
// so add/mul/div/sqrt round exactly as in the reference's own -O2 build).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>

#include "../include/goblin_hip.h"

namespace {

// ---------------------------------------------------------------------------
// Math kernel set (GoblinVector.h, GoblinColor.h, GoblinUtils.h)
// ---------------------------------------------------------------------------
const float PI = 3.14159265358979323f;     // GoblinUtils.h:43
const float TWO_PI = 6.28318530718f;       // :44
const float INV_PI = 0.31830988618379067154f;   // :45
const float INV_TWOPI = 0.15915494309189533577f;  // :46
const float INF = INFINITY;

struct V3 {
    float x, y, z;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return (&x)[i]; }
    float& operator[](int i) { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return a * s; }   // GoblinVector.h:233-235 (rhs * s)
// Vector3::operator/ multiplies by the reciprocal (GoblinVector.h:166-169)
inline V3 operator/(V3 a, float s) {
    float inv = 1.0f / s;
    return V3(a.x * inv, a.y * inv, a.z * inv);
}
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // :209-211
inline float absdot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline V3 cross(V3 a, V3 b) {   // :217-222
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V3 a) { return std::sqrt(sqlen(a)); }
inline V3 normalize(V3 a) { return a / length(a); }   // :229-231

// Color: rgba; binary ops keep the LEFT operand's alpha, += / *= leave alpha
// alone, == compares alpha too (GoblinColor.h:32-108).
struct Col {
    float r, g, b, a;
    Col() = default;
    explicit Col(float c) : r(c), g(c), b(c), a(1.0f) {}
    Col(float r_, float g_, float b_, float a_ = 1.0f) : r(r_), g(g_), b(b_), a(a_) {}
};
inline Col operator*(Col c, float s) { return Col(c.r * s, c.g * s, c.b * s, c.a); }
inline Col operator*(float s, Col c) { return c * s; }
inline Col operator*(Col a, Col b) { return Col(a.r * b.r, a.g * b.g, a.b * b.b, a.a); }
inline Col operator/(Col c, float s) {
    float inv = 1.0f / s;
    return Col(c.r * inv, c.g * inv, c.b * inv, c.a);
}
inline Col operator+(Col a, Col b) { return Col(a.r + b.r, a.g + b.g, a.b + b.b, a.a); }   // GoblinColor.h:32-34
inline Col& operator+=(Col& a, Col b) {
    a.r += b.r; a.g += b.g; a.b += b.b;
    return a;
}
inline Col& operator*=(Col& a, Col b) {
    a.r *= b.r; a.g *= b.g; a.b *= b.b;
    return a;
}
inline bool operator==(Col a, Col b) { return a.r == b.r && a.g == b.g && a.b == b.b && a.a == b.a; }
inline bool operator!=(Col a, Col b) { return !(a == b); }
inline float luminance(Col c) { return 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b; }
const Col BLACK(0.0f, 0.0f, 0.0f, 1.0f);

inline float clampf(float f, float lo, float hi) { return f < lo ? lo : (f > hi ? hi : f); }
inline int ceil_int(float f) { return static_cast<int>(std::ceil(f)); }
inline int floor_int(float f) { return static_cast<int>(std::floor(f)); }
inline int round_to_square(int n, int* root = nullptr) {   // GoblinUtils.h:124-130
    int s = ceil_int(std::sqrt(static_cast<float>(n)));
    if (root) *root = s;
    return s * s;
}
inline float radians(float deg) { return PI * (deg / 180.0f); }

// ---------------------------------------------------------------------------
// Matrix4 / Quaternion / Transform (GoblinMatrix.cpp, GoblinQuaternion.cpp,
// GoblinTransform.cpp)
// ---------------------------------------------------------------------------
struct M4 {
    float m[4][4];
};

M4 quat_to_matrix(const float q[4]) {   // Quaternion::toMatrix, GoblinQuaternion.cpp:62-81
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = 2.0f * x, y2 = 2.0f * y, z2 = 2.0f * z;
    float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, xw2 = x2 * w;
    float yy2 = y2 * y, yz2 = y2 * z, yw2 = y2 * w;
    float zz2 = z2 * z, zw2 = z2 * w;
    M4 r = {{{1 - yy2 - zz2, xy2 - zw2, xz2 + yw2, 0.0f},
             {xy2 + zw2, 1 - xx2 - zz2, yz2 - xw2, 0.0f},
             {xz2 - yw2, yz2 + xw2, 1 - xx2 - yy2, 0.0f},
             {0.0f, 0.0f, 0.0f, 1.0f}}};
    return r;
}

V3 quat_rotate(const float q[4], V3 p) {   // Quaternion::operator*(Vector3), GoblinQuaternion.cpp:86-92
    V3 v(q[1], q[2], q[3]);
    V3 uv = cross(v, p);
    V3 uuv = cross(v, uv);
    uv = uv * (2.0f * q[0]);
    uuv = uuv * 2.0f;
    return p + uv + uuv;
}

M4 mat_mul(const M4& a, const M4& b) {   // Matrix4::operator*, GoblinMatrix.cpp:305-314
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}

// inverse(Matrix4*, const Matrix4&), GoblinMatrix.cpp:419-486: adjugate from
// 2x2 sub-determinants of row pairs (2,3), (1,3), (1,2); returns false and
// leaves *inv partially written when |det| < 1e-5.
bool mat_inverse(M4* out, const M4& in) {
    const float(*a)[4] = in.m;
    float(*o)[4] = out->m;
    // sub[r][k]: 2x2 determinants of rows (r0,r1) over column pairs, in the order 23,13,12,03,02,01
    auto sub = [&](int r0, int r1, float s[6]) {
        s[0] = a[r0][2] * a[r1][3] - a[r0][3] * a[r1][2];
        s[1] = a[r0][1] * a[r1][3] - a[r0][3] * a[r1][1];
        s[2] = a[r0][1] * a[r1][2] - a[r0][2] * a[r1][1];
        s[3] = a[r0][0] * a[r1][3] - a[r0][3] * a[r1][0];
        s[4] = a[r0][0] * a[r1][2] - a[r0][2] * a[r1][0];
        s[5] = a[r0][0] * a[r1][1] - a[r0][1] * a[r1][0];
    };
    // cofactor column from a row `row` against sub-determinants s
    auto col = [&](int row, const float s[6], float c[4]) {
        c[0] = a[row][1] * s[0] - a[row][2] * s[1] + a[row][3] * s[2];
        c[1] = a[row][0] * s[0] - a[row][2] * s[3] + a[row][3] * s[4];
        c[2] = a[row][0] * s[1] - a[row][1] * s[3] + a[row][3] * s[5];
        c[3] = a[row][0] * s[2] - a[row][1] * s[4] + a[row][2] * s[5];
    };
    float s23[6], s13[6], s12[6], c[4];
    sub(2, 3, s23);
    col(1, s23, c);
    o[0][0] = +c[0]; o[1][0] = -c[1]; o[2][0] = +c[2]; o[3][0] = -c[3];
    float det = a[0][0] * o[0][0] + a[0][1] * o[1][0] + a[0][2] * o[2][0] + a[0][3] * o[3][0];
    if (std::fabs(det) < 1e-5f) return false;
    float inv_det = 1.0f / det;
    col(0, s23, c);
    o[0][1] = -c[0]; o[1][1] = +c[1]; o[2][1] = -c[2]; o[3][1] = +c[3];
    sub(1, 3, s13);
    col(0, s13, c);
    o[0][2] = +c[0]; o[1][2] = -c[1]; o[2][2] = +c[2]; o[3][2] = -c[3];
    sub(1, 2, s12);
    col(0, s12, c);
    o[0][3] = -c[0]; o[1][3] = +c[1]; o[2][3] = -c[2]; o[3][3] = +c[3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[i][j] *= inv_det;
    return true;
}

struct Xform {
    M4 M, inv;
    V3 scale;
    // Transform::update, GoblinTransform.cpp:182-193: M = R*S, translation
    // written into column 3, then the general inverse.
    void set(const float pos[3], const float q[4], const float s[3]) {
        M4 S = {{{s[0], 0, 0, 0}, {0, s[1], 0, 0}, {0, 0, s[2], 0}, {0, 0, 0, 1}}};
        M = mat_mul(quat_to_matrix(q), S);
        M.m[0][3] = pos[0]; M.m[1][3] = pos[1]; M.m[2][3] = pos[2];
        M4 id = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
        inv = id;
        mat_inverse(&inv, M);
        scale = V3(s[0], s[1], s[2]);
    }
    V3 on_point(V3 p) const {   // :97-103
        return V3(M.m[0][0] * p.x + M.m[0][1] * p.y + M.m[0][2] * p.z + M.m[0][3],
                  M.m[1][0] * p.x + M.m[1][1] * p.y + M.m[1][2] * p.z + M.m[1][3],
                  M.m[2][0] * p.x + M.m[2][1] * p.y + M.m[2][2] * p.z + M.m[2][3]);
    }
    V3 on_vector(V3 v) const {   // :113-119
        return V3(M.m[0][0] * v.x + M.m[0][1] * v.y + M.m[0][2] * v.z,
                  M.m[1][0] * v.x + M.m[1][1] * v.y + M.m[1][2] * v.z,
                  M.m[2][0] * v.x + M.m[2][1] * v.y + M.m[2][2] * v.z);
    }
    V3 on_normal(V3 n) const {   // (M^-1)^T n, :105-111
        return V3(inv.m[0][0] * n.x + inv.m[1][0] * n.y + inv.m[2][0] * n.z,
                  inv.m[0][1] * n.x + inv.m[1][1] * n.y + inv.m[2][1] * n.z,
                  inv.m[0][2] * n.x + inv.m[1][2] * n.y + inv.m[2][2] * n.z);
    }
    V3 invert_point(V3 p) const {   // :138-144
        return V3(inv.m[0][0] * p.x + inv.m[0][1] * p.y + inv.m[0][2] * p.z + inv.m[0][3],
                  inv.m[1][0] * p.x + inv.m[1][1] * p.y + inv.m[1][2] * p.z + inv.m[1][3],
                  inv.m[2][0] * p.x + inv.m[2][1] * p.y + inv.m[2][2] * p.z + inv.m[2][3]);
    }
    V3 invert_vector(V3 v) const {   // :154-160
        return V3(inv.m[0][0] * v.x + inv.m[0][1] * v.y + inv.m[0][2] * v.z,
                  inv.m[1][0] * v.x + inv.m[1][1] * v.y + inv.m[1][2] * v.z,
                  inv.m[2][0] * v.x + inv.m[2][1] * v.y + inv.m[2][2] * v.z);
    }
};

struct Box {
    V3 lo, hi;
    Box() : lo(INF, INF, INF), hi(-INF, -INF, -INF) {}   // GoblinBBox.h default
    void expand(V3 p) {
        lo = V3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
        hi = V3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
    }
    void expand(const Box& b) {
        lo = V3(std::min(lo.x, b.lo.x), std::min(lo.y, b.lo.y), std::min(lo.z, b.lo.z));
        hi = V3(std::max(hi.x, b.hi.x), std::max(hi.y, b.hi.y), std::max(hi.z, b.hi.z));
    }
    int longest_axis() const {   // GoblinBBox.cpp:78-87
        V3 d = hi - lo;
        if (d.x > d.y && d.x > d.z) return 0;
        if (d.y > d.z) return 1;
        return 2;
    }
};

struct Ray {
    V3 o, d;
    float mint, maxt;
};

// ---------------------------------------------------------------------------
// BVH exactly as the reference builds it (GoblinBVH.cpp:34-151): DFS-linear
// nodes, leaf iff one primitive (or coincident centroids), split on the longest
// centroid axis at the median with std::nth_element ("equal_count" is what every
// caller passes: GoblinScene.cpp:15, GoblinModel.cpp:24).
// ---------------------------------------------------------------------------
struct BvhNode {
    Box box;
    uint32_t offset;   // leaf: first primitive; interior: second child
    uint8_t nprims, axis;
};

struct BuildItem {   // BVHPrimitiveInfo, GoblinBVH.cpp:8-14
    Box box;
    int index;
    V3 center;
};

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prims;   // ordered primitive ids
    Box bounds;

    void build(const std::vector<Box>& boxes) {
        nodes.clear();
        prims.clear();
        bounds = Box();
        for (const Box& b : boxes) bounds.expand(b);
        if (boxes.empty()) return;
        std::vector<BuildItem> items;
        items.reserve(boxes.size());
        for (size_t i = 0; i < boxes.size(); ++i) {
            BuildItem it;
            it.box = boxes[i];
            it.index = static_cast<int>(i);
            it.center = 0.5f * (boxes[i].lo + boxes[i].hi);
            items.push_back(it);
        }
        nodes.reserve(2 * boxes.size());
        recurse(items, 0, static_cast<uint32_t>(items.size()));
    }

    uint32_t recurse(std::vector<BuildItem>& it, uint32_t start, uint32_t end) {
        uint32_t me = static_cast<uint32_t>(nodes.size());
        nodes.push_back(BvhNode());
        Box box;
        for (uint32_t i = start; i < end; ++i) box.expand(it[i].box);
        uint32_t n = end - start;
        auto make_leaf = [&]() {
            uint32_t first = static_cast<uint32_t>(prims.size());
            for (uint32_t i = start; i < end; ++i) prims.push_back(it[i].index);
            nodes[me].box = box;
            nodes[me].offset = first;
            nodes[me].nprims = static_cast<uint8_t>(n);
            nodes[me].axis = 0;
        };
        if (n == 1) {
            make_leaf();
            return me;
        }
        Box cb;
        for (uint32_t i = start; i < end; ++i) cb.expand(it[i].center);
        int dim = cb.longest_axis();
        if (cb.lo[dim] == cb.hi[dim]) {
            make_leaf();
            return me;
        }
        uint32_t mid = (start + end) / 2;
        std::nth_element(&it[start], &it[mid], &it[end - 1] + 1,
                         [dim](const BuildItem& a, const BuildItem& b) { return a.center[dim] < b.center[dim]; });
        recurse(it, start, mid);
        uint32_t second = recurse(it, mid, end);
        nodes[me].box = box;
        nodes[me].offset = second;
        nodes[me].axis = static_cast<uint8_t>(dim);
        nodes[me].nprims = 0;
        return me;
    }
};

// static intersect(bbox, ray, invDir, dirIsNeg), GoblinBVH.cpp:156-187
inline bool slab_test(const Box& b, const Ray& ray, V3 inv, const uint32_t neg[3]) {
    const V3* bb = &b.lo;   // bb[0]=lo, bb[1]=hi
    float tmin = (bb[neg[0]].x - ray.o.x) * inv.x;
    float tmax = (bb[1 - neg[0]].x - ray.o.x) * inv.x;
    float tymin = (bb[neg[1]].y - ray.o.y) * inv.y;
    float tymax = (bb[1 - neg[1]].y - ray.o.y) * inv.y;
    if (tymax < tmin || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (bb[neg[2]].z - ray.o.z) * inv.z;
    float tzmax = (bb[1 - neg[2]].z - ray.o.z) * inv.z;
    if (tzmax < tmin || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < ray.maxt) && (tmax > ray.mint);
}

struct Counters {
    uint64_t closest = 0, anyhit = 0, filtered = 0, nodes = 0, tris = 0;
    // debugging aid (orc_debug_rays): every scene query of a sample, 16 floats each --
    // {0 closest / 1 any, o(3), d(3), mint, maxt, result: hit t or -1 / 1 occluded or 0, shading normal(3), tangent(3)}
    std::vector<float>* log = nullptr;
    void add(const Counters& o) {
        closest += o.closest; anyhit += o.anyhit; filtered += o.filtered; nodes += o.nodes; tris += o.tris;
    }
};

// Generic stack traversal shared by intersect/occluded (GoblinBVH.cpp:189-280).
// `leaf(prim)` returns true to stop early (any-hit).
template <class Leaf>
inline void traverse(const Bvh& bvh, const Ray& ray, Counters* cnt, Leaf&& leaf) {
    if (bvh.nodes.empty()) return;
    V3 inv(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    uint32_t neg[3] = {ray.d.x < 0.0f, ray.d.y < 0.0f, ray.d.z < 0.0f};
    uint32_t node = 0, sp = 0, todo[64];
    while (true) {
        const BvhNode& nd = bvh.nodes[node];
        ++cnt->nodes;
        if (slab_test(nd.box, ray, inv, neg)) {
            if (nd.nprims > 0) {
                for (uint32_t i = 0; i < nd.nprims; ++i)
                    if (leaf(bvh.prims[nd.offset + i])) return;
                if (sp == 0) break;
                node = todo[--sp];
            } else if (neg[nd.axis]) {
                todo[sp++] = node + 1;
                node = nd.offset;
            } else {
                todo[sp++] = nd.offset;
                node = node + 1;
            }
        } else {
            if (sp == 0) break;
            node = todo[--sp];
        }
    }
}

// ---------------------------------------------------------------------------
// Prepared scene
// ---------------------------------------------------------------------------
struct Frag {   // Fragment, GoblinGeometry.h:14-127
    V3 p, n, dpdu, dpdv;
    float u, v;
    // set by Intersection::computeUVDifferential (GoblinPrimitive.cpp:32-97); zero without ray differentials
    V3 dpdx = V3(0, 0, 0), dpdy = V3(0, 0, 0);
    float dudx = 0.0f, dvdx = 0.0f, dudy = 0.0f, dvdy = 0.0f;
};

struct RayDiff {   // RayDifferential's auxiliary rays (GoblinRay.h:33-46)
    bool has = false;
    V3 dxo, dxd, dyo, dyd;
};

struct Mesh {
    const float *pos, *nrm, *uv;
    const uint32_t* idx;
    uint32_t ntris;
    bool has_n, has_uv;
    uint32_t shape = GBL_SHAPE_MESH;   // sphere / disk are intersectable geometries: no BLAS (GoblinModel.cpp:14)
    float radius = 1.0f;
    Bvh bvh;
    Box bounds;   // PolygonMesh::mBBox over de-duplicated vertices | Sphere/Disk::getObjectBound
    V3 P(uint32_t i) const { return V3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); }
    V3 N(uint32_t i) const { return V3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]); }
};

struct Instance {
    uint32_t mesh, material;
    int area_light;
    bool is_mask = false;   // its material's type carries BSDFnullptr: what isOpaque / notOpaque test (GoblinPathtracer.cpp:5-11)
    Xform xf;
};

// IntersectFilter of the scene queries: none, isOpaque (skip masks), notOpaque (masks only)
enum { FILTER_NONE = 0, FILTER_OPAQUE = 1, FILTER_MASK = 2 };

struct AreaGeo {   // GeometrySet, GoblinLight.cpp:289-343
    std::vector<float> area, cdf;
    float sum_area = 0.0f, integral = 0.0f, dx = 0.0f;
};

struct Light {
    uint32_t type;
    Col color;
    V3 pos;
    V3 spot_axis;   // mToWorld.onVector(UnitZ)
    float cos_max, cos_falloff;
    uint32_t mesh;
    Xform xf;
    AreaGeo geo;
    int image = -1;   // ImageBasedLight: mRadiance (index into orc_scene::images); its CDF2D is orc_scene::ibl_rows / ibl_marginal
};

// CDF1D, GoblinSampler.cpp:309-342
struct Cdf {
    std::vector<float> f, cdf;
    float integral = 0.0f, dx = 0.0f;
    void init(const std::vector<float>& fn) {
        f = fn;
        size_t n = f.size();
        dx = 1.0f / n;
        cdf.assign(n + 1, 0.0f);
        for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + (f[i - 1] * dx);
        integral = cdf[n];
        for (size_t i = 1; i < n + 1; ++i) cdf[i] /= integral;
    }
    int sample_discrete(float u, float* pdf) const {
        auto lb = std::lower_bound(cdf.begin(), cdf.end(), u);
        int off = std::max(0, static_cast<int>(lb - cdf.begin() - 1));
        if (pdf) *pdf = (f[off] / integral) * dx;
        return off;
    }
    float sample_continuous(float u, float* pdf, int* index) const {   // CDF1D::sampleContinuous, :344-356
        auto lb = std::lower_bound(cdf.begin(), cdf.end(), u);
        int off = std::max(0, static_cast<int>(lb - cdf.begin() - 1));
        float d = (u - cdf[off]) / (cdf[off + 1] - cdf[off]);
        if (pdf) *pdf = f[off] / integral;
        if (index) *index = off;
        return (static_cast<float>(off) + d) / f.size();
    }
};

}  // namespace

struct orc_scene {
    gbl_scene_desc desc;
    std::vector<Mesh> meshes;
    std::vector<Instance> instances;
    std::vector<gbl_material> materials;
    std::vector<gbl_texture> textures;
    std::vector<Xform> tex_xf;   // SphericalMapping::mToTex per texture
    std::vector<gbl_image> images;   // MIP pyramids (built by the scene front end exactly as MIPMap's constructor does)
    std::vector<float> texels;
    std::vector<float> ewa_lut;      // MIPMap<T>::EWALut
    std::vector<Cdf> ibl_marginal;               // per light (image based lights only): CDF2D::mMarginalDist ...
    std::vector<std::vector<Cdf>> ibl_rows;      // ... and mConditionalDist
    bool has_masks = false;
    std::vector<uint32_t> light_samples;   // Light::getSamplesNum per light (the Whitted renderer's quota)
    // HomogeneousVolumeRegion (GoblinVolume.h:72-112) / HeterogeneousVolumeRegion + VolumeGrid (GoblinVolume.cpp:135-341)
    struct Volume {
        bool on = false;
        bool hetero = false;
        Col attenuation, scatter, emission;
        Col albedo;
        float g = 0.0f;
        int sample_num = 0;
        float step = 0.0f;
        int nx = 0, ny = 0, nz = 0, nch = 0;
        V3 normalize;                 // VolumeGrid::mNormalizeTerm: 1 / (bbox extent)
        std::vector<float> density;
        Box box;
        Xform xf;
    } volume;
    std::vector<Light> lights;
    std::vector<Cdf> light_geo_cdf;   // per light (area only)
    Cdf light_power;
    std::vector<Col> light_power_rgb;
    Bvh tlas;
    // camera
    V3 cam_pos;
    float cam_q[4];
    float proj00, proj11;
    uint32_t cam_type = 0;
    float lens_radius = 0.0f, focal_distance = 1.0f, film_w = 0.0f, film_h = 0.0f;
    // film
    int xres, yres, xstart, ystart, xcount, ycount;
    float inv_xres, inv_yres;
    float filter_w[2];
    float filter_table[256];
    int window[4];
};

namespace {
// MIPMap lookups (defined with the textures below; the scene constructor's image based light needs them)
Col mip_level(const orc_scene* sc, const gbl_image& im, int level, float s, float t, uint32_t mode);
inline void level_dims(const gbl_image& im, int level, int* w, int* h, size_t* off);
}  // namespace

namespace {

// ---------------------------------------------------------------------------
// Reconstruction filters (GoblinFilter.cpp) and FilterTable (GoblinFilm.cpp:10-37)
// ---------------------------------------------------------------------------
struct FilterFn {
    uint32_t type;
    float wx, wy, alpha, expx, expy, b, c;
    float gaussian(float v, float base) const { return std::max(0.0f, expf(-alpha * v * v) - base); }
    float mitchell(float x) const {   // GoblinFilter.cpp:79-91
        x = std::fabs(2.0f * x);
        if (x > 1.0f)
            return ((-b - 6 * c) * x * x * x + (6 * b + 30 * c) * x * x + (-12 * b - 48 * c) * x + (8 * b + 24 * c)) / 6.0f;
        return ((12 - 9 * b - 6 * c) * x * x * x + (-18 + 12 * b + 6 * c) * x * x + (6 - 2 * b)) / 6.0f;
    }
    float eval(float x, float y) const {
        switch (type) {
            case GBL_FILTER_BOX: return 1.0f;
            case GBL_FILTER_TRIANGLE: return std::max(0.0f, wx - fabsf(x)) * std::max(0.0f, wy - fabsf(y));
            case GBL_FILTER_MITCHELL: return mitchell(x * (1.0f / wx)) * mitchell(y * (1.0f / wy));
            default: return gaussian(x, expx) * gaussian(y, expy);
        }
    }
    float normalize_term() const {
        switch (type) {
            case GBL_FILTER_BOX: return 4.0f * wx * wy;
            case GBL_FILTER_TRIANGLE: return wx * wx * wy * wy;
            case GBL_FILTER_MITCHELL:   // GoblinFilter.cpp:72-77
                return 4.0f * ((12 - 9 * b - 6 * c) / 4 + (-18 + 12 * b + 6 * c) / 3 + (6 - 2 * b) + 15 * (-b - 6 * b) / 4 +
                               7 * (6 * b + 30 * c) / 3 + 3 * (-12 * b - 48 * c) / 2 + (8 * b + 24 * c)) / 6.0f;
            default: {   // numeric 20x20 quadrature, GoblinFilter.cpp:48-64
                size_t step = 20;
                float dx = wx / static_cast<float>(step), dy = wy / static_cast<float>(step);
                float result = 0.0f;
                for (size_t i = 0; i < step; ++i)
                    for (size_t j = 0; j < step; ++j)
                        result += 4.0f * dx * dy * gaussian(i * dx, expx) * gaussian(j * dy, expy);
                return result;
            }
        }
    }
};

void build_filter_table(orc_scene* s) {
    const gbl_film& f = s->desc.film;
    FilterFn fn;
    fn.type = f.filter_type;
    fn.wx = f.filter_width[0];
    fn.wy = f.filter_width[1];
    fn.alpha = f.gaussian_falloff;
    fn.expx = expf(-fn.alpha * fn.wx * fn.wx);
    fn.expy = expf(-fn.alpha * fn.wy * fn.wy);
    fn.b = f.mitchell_b;
    fn.c = f.mitchell_c;
    float dx = fn.wx / 16, dy = fn.wy / 16;
    float norm = fn.normalize_term();
    size_t k = 0;
    for (int y = 0; y < 16; ++y) {
        float fy = y * dy;
        for (int x = 0; x < 16; ++x) {
            float fx = x * dx;
            s->filter_table[k++] = fn.eval(fx, fy) / norm;
        }
    }
    s->filter_w[0] = fn.wx;
    s->filter_w[1] = fn.wy;
}

inline float filter_lookup(const orc_scene* s, float x, float y) {   // FilterTable::evaluate, GoblinFilm.cpp:29-37
    int iy = std::min(floor_int(std::fabs(16 * y / s->filter_w[1])), 15);
    int ix = std::min(floor_int(std::fabs(16 * x / s->filter_w[0])), 15);
    return s->filter_table[iy * 16 + ix];
}

// ImageTile::addSample over the full-film tile (GoblinFilm.cpp:61-90,
// GoblinThreadLocalStorage.h:30-35).  film = xres*yres float4.
inline void add_sample(const orc_scene* s, float* film, float image_x, float image_y, Col L, uint64_t* splats) {
    if (L.r != L.r || L.g != L.g || L.b != L.b || L.a != L.a) return;
    float dx = image_x - 0.5f, dy = image_y - 0.5f;
    int x0 = ceil_int(dx - s->filter_w[0]), x1 = floor_int(dx + s->filter_w[0]);
    int y0 = ceil_int(dy - s->filter_w[1]), y1 = floor_int(dy + s->filter_w[1]);
    x0 = std::max(x0, s->xstart);
    x1 = std::min(x1, s->xstart + s->xcount - 1);
    y0 = std::max(y0, s->ystart);
    y1 = std::min(y1, s->ystart + s->ycount - 1);
    for (int y = y0; y <= y1; ++y) {
        for (int x = x0; x <= x1; ++x) {
            float w = filter_lookup(s, x - dx, y - dy);
            float* px = film + 4 * (static_cast<size_t>(y) * s->xres + x);
            Col wl = w * L;
            px[0] += wl.r;
            px[1] += wl.g;
            px[2] += wl.b;
            px[3] += w;
            if (splats) ++*splats;
        }
    }
}

// ---------------------------------------------------------------------------
// Scene preparation
// ---------------------------------------------------------------------------
void coordinate_axes(V3 a1, V3* a2, V3* a3) {   // coordinateAxises, GoblinUtils.cpp:58-69
    if (fabsf(a1.x) > fabsf(a1.y)) {
        float inv = 1.0f / sqrtf(a1.x * a1.x + a1.z * a1.z);
        *a2 = V3(-a1.z * inv, 0.0f, a1.x * inv);
    } else {
        float inv = 1.0f / sqrtf(a1.y * a1.y + a1.z * a1.z);
        *a2 = V3(0.0f, -a1.z * inv, a1.y * inv);
    }
    *a3 = cross(a1, *a2);
}

void quat_from_matrix3(const float R[3][3], float q_wxyz[4]) {   // Quaternion(Matrix3), GoblinQuaternion.cpp:20-52
    float q[4];
    float trace = R[0][0] + R[1][1] + R[2][2];
    if (trace > 0.0f) {
        float s = std::sqrt(trace + 1.0f);
        q[3] = s * 0.5f;
        float t = 0.5f / s;
        q[0] = (R[2][1] - R[1][2]) * t;
        q[1] = (R[0][2] - R[2][0]) * t;
        q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int next[3] = {1, 2, 0};
        int j = next[i], k = next[j];
        float s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0f);
        q[i] = s * 0.5f;
        float t = s != 0.0f ? 0.5f / s : s;
        q[3] = (R[k][j] - R[j][k]) * t;
        q[j] = (R[j][i] + R[i][j]) * t;
        q[k] = (R[k][i] + R[i][k]) * t;
    }
    q_wxyz[0] = q[3]; q_wxyz[1] = q[0]; q_wxyz[2] = q[1]; q_wxyz[3] = q[2];
}

float tri_area(const Mesh& m, uint32_t t) {   // Triangle::area, GoblinTriangle.cpp:179-189
    V3 p0 = m.P(m.idx[3 * t]), p1 = m.P(m.idx[3 * t + 1]), p2 = m.P(m.idx[3 * t + 2]);
    return 0.5f * length(cross(p1 - p0, p2 - p0));
}

Box transform_box(const Xform& xf, const Box& b) {   // Transform::onBBox, GoblinTransform.cpp:125-136
    Box r;
    r.expand(xf.on_point(b.lo));
    r.expand(xf.on_point(V3(b.hi.x, b.lo.y, b.lo.z)));
    r.expand(xf.on_point(V3(b.lo.x, b.hi.y, b.lo.z)));
    r.expand(xf.on_point(V3(b.lo.x, b.lo.y, b.hi.z)));
    r.expand(xf.on_point(V3(b.hi.x, b.hi.y, b.lo.z)));
    r.expand(xf.on_point(V3(b.hi.x, b.lo.y, b.hi.z)));
    r.expand(xf.on_point(V3(b.lo.x, b.hi.y, b.hi.z)));
    r.expand(xf.on_point(b.hi));
    return r;
}

void prepare(orc_scene* s) {
    const gbl_scene_desc& d = s->desc;
    // meshes + per-mesh BLAS over per-triangle AABBs (GoblinModel.cpp:10-26)
    s->meshes.resize(d.num_meshes);
    for (uint32_t mi = 0; mi < d.num_meshes; ++mi) {
        const gbl_mesh& gm = d.meshes[mi];
        Mesh& m = s->meshes[mi];
        m.pos = d.positions + 3 * static_cast<size_t>(gm.vertex_offset);
        m.nrm = d.normals + 3 * static_cast<size_t>(gm.vertex_offset);
        m.uv = d.uvs + 2 * static_cast<size_t>(gm.vertex_offset);
        m.idx = d.indices + 3 * static_cast<size_t>(gm.tri_offset);
        m.ntris = gm.tri_count;
        m.has_n = gm.has_normal != 0;
        m.has_uv = gm.has_uv != 0;
        m.shape = gm.shape;
        m.radius = gm.radius;
        m.bounds = Box();
        if (m.shape == GBL_SHAPE_SPHERE) {   // Sphere::getObjectBound, GoblinSphere.cpp:140-143
            m.ntris = 0;
            m.bounds.expand(V3(m.radius, m.radius, m.radius));
            m.bounds.expand(V3(-m.radius, -m.radius, -m.radius));
            continue;
        }
        if (m.shape == GBL_SHAPE_DISK) {     // Disk::getObjectBound, GoblinDisk.cpp:81-84
            m.ntris = 0;
            m.bounds.expand(V3(m.radius, m.radius, 0.0f));
            m.bounds.expand(V3(-m.radius, -m.radius, 0.0f));
            continue;
        }
        for (uint32_t v = 0; v < gm.vertex_count; ++v) m.bounds.expand(m.P(v));
        std::vector<Box> boxes(m.ntris);
        for (uint32_t t = 0; t < m.ntris; ++t) {   // Triangle::getObjectBound, GoblinTriangle.cpp:191-205
            boxes[t].expand(m.P(m.idx[3 * t]));
            boxes[t].expand(m.P(m.idx[3 * t + 1]));
            boxes[t].expand(m.P(m.idx[3 * t + 2]));
        }
        m.bvh.build(boxes);
    }
    s->materials.assign(d.materials, d.materials + d.num_materials);
    s->textures.assign(d.textures, d.textures + d.num_textures);
    s->images.assign(d.images, d.images + d.num_images);
    s->texels.assign(d.texels, d.texels + d.num_texels);
    s->ewa_lut.resize(128);   // MIPMap<T>::initEWALut, GoblinTexture.cpp:262-271
    for (size_t i = 0; i < 128; ++i) {
        float r2 = float(i) / float(128 - 1);
        s->ewa_lut[i] = expf(-2.0f * r2) - expf(-2.0f);
    }
    s->tex_xf.resize(d.num_textures);
    for (uint32_t i = 0; i < d.num_textures; ++i)
        s->tex_xf[i].set(d.textures[i].to_tex.position, d.textures[i].to_tex.orientation, d.textures[i].to_tex.scale);
    // instances + TLAS over instances only (GoblinScene.cpp:15, GoblinPrimitive.cpp:119-121)
    s->instances.resize(d.num_instances);
    std::vector<Box> iboxes(d.num_instances);
    for (uint32_t i = 0; i < d.num_instances; ++i) {
        const gbl_instance& gi = d.instances[i];
        Instance& in = s->instances[i];
        in.mesh = gi.mesh;
        in.material = gi.material;
        in.area_light = gi.area_light;
        // MaskMaterial ORs BSDFnullptr into its type; SubsurfaceMaterial's type is BSDFAll, which holds that bit too
        in.is_mask = d.materials[gi.material].type == GBL_MAT_MASK || d.materials[gi.material].type == GBL_MAT_SUBSURFACE;
        s->has_masks = s->has_masks || in.is_mask;
        in.xf.set(gi.to_world.position, gi.to_world.orientation, gi.to_world.scale);
        iboxes[i] = transform_box(in.xf, s->meshes[in.mesh].bounds);   // Model::getAABB = mesh bound
    }
    s->tlas.build(iboxes);
    // lights
    s->lights.resize(d.num_lights);
    if (d.volume.type == GBL_VOLUME_HOMOGENEOUS || d.volume.type == GBL_VOLUME_HETEROGENEOUS) {
        orc_scene::Volume& v = s->volume;
        v.on = true;
        v.hetero = d.volume.type == GBL_VOLUME_HETEROGENEOUS;
        v.albedo = Col(d.volume.albedo[0], d.volume.albedo[1], d.volume.albedo[2]);
        if (v.hetero) {
            v.step = d.volume.step_size;
            v.nx = d.volume.grid[0], v.ny = d.volume.grid[1], v.nz = d.volume.grid[2], v.nch = d.volume.grid_channels;
            v.density.assign(d.volume.density, d.volume.density + static_cast<size_t>(v.nx) * v.ny * v.nz * v.nch);
        }
        v.attenuation = Col(d.volume.attenuation[0], d.volume.attenuation[1], d.volume.attenuation[2]);
        v.scatter = v.attenuation * Col(d.volume.albedo[0], d.volume.albedo[1], d.volume.albedo[2]);   // mScatter(attenuation * albedo)
        v.emission = Col(d.volume.emission[0], d.volume.emission[1], d.volume.emission[2]);
        v.g = d.volume.g;
        v.sample_num = d.volume.sample_num;
        const float* a = d.volume.box_min;
        const float* b = d.volume.box_max;
        v.box.lo = V3(std::min(a[0], b[0]), std::min(a[1], b[1]), std::min(a[2], b[2]));   // BBox(p1, p2), GoblinBBox.h:20-23
        v.box.hi = V3(std::max(a[0], b[0]), std::max(a[1], b[1]), std::max(a[2], b[2]));
        v.xf.set(d.volume.to_world.position, d.volume.to_world.orientation, d.volume.to_world.scale);
        const V3 dim = v.box.hi - v.box.lo;
        v.normalize = V3(1.0f / dim.x, 1.0f / dim.y, 1.0f / dim.z);
    }
    s->light_samples.resize(d.num_lights);
    for (uint32_t i = 0; i < d.num_lights; ++i) s->light_samples[i] = d.lights[i].sample_num;
    std::vector<float> powers;
    s->light_power_rgb.clear();
    s->ibl_marginal.resize(d.num_lights);
    s->ibl_rows.resize(d.num_lights);
    for (uint32_t i = 0; i < d.num_lights; ++i) {
        const gbl_light& gl = d.lights[i];
        Light& l = s->lights[i];
        l.type = gl.type;
        l.color = Col(gl.color[0], gl.color[1], gl.color[2]);
        l.pos = V3(gl.position[0], gl.position[1], gl.position[2]);
        l.cos_max = gl.cos_theta_max;
        l.cos_falloff = gl.cos_falloff_start;
        l.mesh = gl.mesh;
        Col power(0.0f);
        if (gl.type == GBL_LIGHT_SPOT) {
            // SpotLight ctor + Light::setOrientation (GoblinLight.cpp:212-223,66-76):
            // direction -> basis -> Matrix3 -> Quaternion -> Transform -> column z
            V3 dir = normalize(V3(gl.direction[0], gl.direction[1], gl.direction[2]));
            V3 xa, ya;
            coordinate_axes(dir, &xa, &ya);
            float R[3][3] = {{xa.x, ya.x, dir.x}, {xa.y, ya.y, dir.y}, {xa.z, ya.z, dir.z}};
            float q[4];
            quat_from_matrix3(R, q);
            float one[3] = {1.0f, 1.0f, 1.0f};
            l.xf.set(gl.position, q, one);
            l.spot_axis = l.xf.on_vector(V3(0.0f, 0.0f, 1.0f));
            // SpotLight::power, GoblinLight.cpp:269-275
            power = l.color * TWO_PI * (1.0f - 0.5f * (l.cos_max + l.cos_falloff));
        } else if (gl.type == GBL_LIGHT_DIRECTIONAL) {
            // DirectionalLight ctor -> Light::setOrientation (GoblinLight.cpp:136-143,66-76); getDirection() is
            // mToWorld.onVector(UnitZ) (GoblinLight.h:199).  The ctor does NOT normalise D (the spot light's does).
            V3 dir = V3(gl.direction[0], gl.direction[1], gl.direction[2]);
            V3 xa, ya;
            coordinate_axes(dir, &xa, &ya);
            float R[3][3] = {{xa.x, ya.x, dir.x}, {xa.y, ya.y, dir.y}, {xa.z, ya.z, dir.z}};
            float q[4];
            quat_from_matrix3(R, q);
            float one[3] = {1.0f, 1.0f, 1.0f}, zero[3] = {0.0f, 0.0f, 0.0f};
            l.xf.set(zero, q, one);
            l.spot_axis = l.xf.on_vector(V3(0.0f, 0.0f, 1.0f));
            // DirectionalLight::power (:203-210) over Scene::getBoundingSphere = the TLAS bound (GoblinScene.cpp:63-65,
            // GoblinBBox.h:51-54: radius is the full diagonal)
            float radius = length(s->tlas.bounds.hi - s->tlas.bounds.lo);
            power = radius * radius * PI * l.color;
        } else if (gl.type == GBL_LIGHT_AREA) {
            l.xf.set(gl.to_world.position, gl.to_world.orientation, gl.to_world.scale);
            const Mesh& m = s->meshes[gl.mesh];
            if (m.shape != GBL_SHAPE_MESH) {   // GeometrySet over one intersectable geometry (:291-292)
                float a = m.shape == GBL_SHAPE_SPHERE ? 4.0f * PI * m.radius * m.radius : PI * m.radius * m.radius;
                l.geo.area.assign(1, a);
                l.geo.sum_area = 0.0f;
                l.geo.sum_area += a;
            } else {
                l.geo.area.resize(m.ntris);
                l.geo.sum_area = 0.0f;
                for (uint32_t t = 0; t < m.ntris; ++t) {
                    float a = tri_area(m, t);
                    l.geo.area[t] = a;
                    l.geo.sum_area += a;
                }
            }
            // AreaLight::power, GoblinLight.cpp:446-455
            float world_area = l.geo.sum_area * (l.xf.scale.x * l.xf.scale.y);
            power = l.color * PI * world_area;
        } else if (gl.type == GBL_LIGHT_IBL) {
            // ImageBasedLight's constructor (GoblinLight.cpp:464-508): the frame, the average radiance, the CDF2D
            l.image = gl.image;
            struct Q { float w, x, y, z; };
            auto qmul = [](const Q& a, const Q& b) {   // GoblinQuaternion.h:45-48
                float d3 = a.x * b.x + a.y * b.y + a.z * b.z;
                Q r;
                r.w = a.w * b.w - d3;
                r.x = a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y);
                r.y = a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z);
                r.z = a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x);
                return r;
            };
            auto qnorm = [](const Q& q) {   // normalize(Quaternion), GoblinQuaternion.cpp:94-100
                float inv = 1.0f / std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
                Q r = {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
                return r;
            };
            auto axis_angle = [](int axis, float angle) {   // Quaternion(unit axis, angle), GoblinQuaternion.cpp:9-15
                float t = angle * 0.5f, st = std::sin(t);
                Q r = {std::cos(t), axis == 0 ? 1.0f * st : 0.0f * st, axis == 1 ? 1.0f * st : 0.0f * st, 0.0f * st};
                return r;
            };
            Q q = {1.0f, 0.0f, 0.0f, 0.0f};
            q = qnorm(qmul(axis_angle(0, -0.5f * PI), q));   // mToWorld.rotateX(-0.5f * PI)
            q = qnorm(qmul(axis_angle(1, -0.5f * PI), q));   // mToWorld.rotateY(-0.5f * PI)
            Q given = {gl.to_world.orientation[0], gl.to_world.orientation[1], gl.to_world.orientation[2], gl.to_world.orientation[3]};
            q = qmul(given, q);                              // setOrientation(orientation * mToWorld.getOrientation())
            float qq[4] = {q.w, q.x, q.y, q.z}, one[3] = {1.0f, 1.0f, 1.0f}, zero[3] = {0.0f, 0.0f, 0.0f};
            l.xf.set(zero, qq, one);
            const gbl_image& im = s->images[gl.image];
            int max_level = static_cast<int>(im.levels) - 1;
            Col average = mip_level(s, im, max_level, 0.0f, 0.0f, GBL_ADDRESS_REPEAT);
            int dist_level = std::max(0, max_level - 8);
            int dw, dh;
            size_t off;
            level_dims(im, dist_level, &dw, &dh, &off);
            std::vector<float> row_integrals;
            s->ibl_rows[i].resize(dh);
            for (int r = 0; r < dh; ++r) {
                float sin_theta = std::sin(((float)r + 0.5f) / (float)dh * PI);
                std::vector<float> f(dw);
                for (int cc = 0; cc < dw; ++cc) {
                    const float* px = s->texels.data() + off + (static_cast<size_t>(r) * dw + cc) * 4;
                    f[cc] = luminance(Col(px[0], px[1], px[2])) * sin_theta;
                }
                s->ibl_rows[i][r].init(f);
                row_integrals.push_back(s->ibl_rows[i][r].integral);
            }
            s->ibl_marginal[i].init(row_integrals);
            // ImageBasedLight::power (:606-613)
            float radius = length(s->tlas.bounds.hi - s->tlas.bounds.lo);
            power = average * PI * (4.0f * PI * radius * radius);
        } else {
            power = 4.0f * PI * l.color;   // PointLight::power, GoblinLight.cpp:132-134
        }
        s->light_power_rgb.push_back(power);
        powers.push_back(luminance(power));
    }
    s->light_geo_cdf.resize(d.num_lights);
    for (uint32_t i = 0; i < d.num_lights; ++i)
        if (s->lights[i].type == GBL_LIGHT_AREA) s->light_geo_cdf[i].init(s->lights[i].geo.area);
    if (!powers.empty()) s->light_power.init(powers);
    // camera (GoblinCamera.cpp:11-19,83-95; matrixPerspectiveLHD3D GoblinMatrix.cpp:631-642)
    const gbl_camera& c = d.camera;
    s->cam_pos = V3(c.position[0], c.position[1], c.position[2]);
    memcpy(s->cam_q, c.orientation, sizeof(s->cam_q));
    float aspect = static_cast<float>(d.film.xres) / static_cast<float>(d.film.yres);
    float fov = radians(c.fov_degrees);
    float yscale = 1.0f / std::tan(fov / 2.0f);
    s->proj11 = yscale;
    s->proj00 = yscale / aspect;
    s->cam_type = c.type;
    s->lens_radius = c.lens_radius;
    s->focal_distance = c.focal_distance;
    s->film_w = c.film_width;            // OrthographicCamera ctor, GoblinCamera.cpp:288-296
    s->film_h = c.film_width / aspect;
    // film (GoblinFilm.cpp:92-112,131-138)
    const gbl_film& f = d.film;
    s->xres = f.xres;
    s->yres = f.yres;
    s->xstart = ceil_int(f.xres * f.crop[0]);
    s->xcount = std::max(1, ceil_int(f.xres * f.crop[1]) - s->xstart);
    s->ystart = ceil_int(f.yres * f.crop[2]);
    s->ycount = std::max(1, ceil_int(f.yres * f.crop[3]) - s->ystart);
    s->inv_xres = 1.0f / static_cast<float>(f.xres);
    s->inv_yres = 1.0f / static_cast<float>(f.yres);
    build_filter_table(s);
    s->window[0] = floor_int(s->xstart + 0.5f - s->filter_w[0]);
    s->window[1] = floor_int(s->xstart + 0.5f + s->xcount + s->filter_w[0]);
    s->window[2] = floor_int(s->ystart + 0.5f - s->filter_w[1]);
    s->window[3] = floor_int(s->ystart + 0.5f + s->ycount + s->filter_w[1]);
}

// ---------------------------------------------------------------------------
// Camera (PerspectiveCamera::generateRay pinhole branch, GoblinCamera.cpp:97-148)
// ---------------------------------------------------------------------------
// uniformSampleDisk, GoblinSampler.cpp:565-602
inline void uniform_sample_disk(float u1, float u2, float* ox, float* oy) {
    float r, theta;
    float x = 2.0f * u1 - 1.0f;
    float y = 2.0f * u2 - 1.0f;
    if (x + y > 0) {
        if (x > y) {
            r = x;
            theta = 0.25f * PI * (y / x);
        } else {
            r = y;
            theta = 0.25f * PI * (2.0f - x / y);
        }
    } else {
        if (x < y) {
            r = -x;
            theta = 0.25f * PI * (4.0f + y / x);
        } else {
            r = -y;
            if (y != 0.0f) theta = 0.25f * PI * (6.0f - x / y);
            else theta = 0.0f;
        }
    }
    *ox = r * std::cos(theta);
    *oy = r * std::sin(theta);
}

inline Ray camera_ray(const orc_scene* s, float image_x, float image_y, float lens_u1 = 0.0f, float lens_u2 = 0.0f,
                      RayDiff* rd = nullptr) {
    float xndc = +2.0f * image_x * s->inv_xres - 1.0f;
    float yndc = -2.0f * image_y * s->inv_yres + 1.0f;
    float dxndc = +2.0f * (image_x + 1.0f) * s->inv_xres - 1.0f;
    float dyndc = -2.0f * (image_y + 1.0f) * s->inv_yres + 1.0f;
    if (rd) rd->has = true;
    Ray r;
    if (s->cam_type == GBL_CAMERA_ORTHOGRAPHIC) {   // OrthographicCamera::generateRay, GoblinCamera.cpp:298-326
        float xv = 0.5f * s->film_w * xndc;
        float yv = 0.5f * s->film_h * yndc;
        r.o = s->cam_pos + quat_rotate(s->cam_q, V3(xv, yv, 0.0f));
        r.d = quat_rotate(s->cam_q, V3(0.0f, 0.0f, 1.0f));
        r.mint = 0.0f;
        r.maxt = INF;
        if (rd) {
            float dxv = 0.5f * s->film_w * dxndc;
            float dyv = 0.5f * s->film_h * dyndc;
            rd->dxo = s->cam_pos + quat_rotate(s->cam_q, V3(dxv, yv, 0.0f));
            rd->dyo = s->cam_pos + quat_rotate(s->cam_q, V3(xv, dyv, 0.0f));
            rd->dxd = rd->dyd = r.d;
        }
        return r;
    }
    float xv = xndc / s->proj00;
    float yv = yndc / s->proj11;
    V3 view(xv, yv, 1.0f);
    V3 dx_view(dxndc / s->proj00, yv, 1.0f), dy_view(xv, dyndc / s->proj11, 1.0f);
    if (s->lens_radius == 0.0f) {
        r.o = s->cam_pos;
        r.d = quat_rotate(s->cam_q, normalize(view));
        if (rd) {
            rd->dxo = rd->dyo = s->cam_pos;
            rd->dxd = quat_rotate(s->cam_q, normalize(dx_view));
            rd->dyd = quat_rotate(s->cam_q, normalize(dy_view));
        }
    } else {   // thin lens, :127-141
        if (rd) {
            float ft0 = s->focal_distance / view.z;
            float lx0, ly0;
            uniform_sample_disk(lens_u1, lens_u2, &lx0, &ly0);
            V3 vo(s->lens_radius * lx0, s->lens_radius * ly0, 0.0f);
            rd->dxo = rd->dyo = quat_rotate(s->cam_q, vo) + s->cam_pos;
            rd->dxd = quat_rotate(s->cam_q, normalize(dx_view * ft0 - vo));
            rd->dyd = quat_rotate(s->cam_q, normalize(dy_view * ft0 - vo));
        }
        float ft = s->focal_distance / view.z;
        V3 p_focus = view * ft;
        float lx, ly;
        uniform_sample_disk(lens_u1, lens_u2, &lx, &ly);
        V3 view_origin(s->lens_radius * lx, s->lens_radius * ly, 0.0f);
        r.o = quat_rotate(s->cam_q, view_origin) + s->cam_pos;
        r.d = quat_rotate(s->cam_q, normalize(p_focus - view_origin));
    }
    r.mint = 1e-3f;
    r.maxt = INF;
    return r;
}

// ---------------------------------------------------------------------------
// Triangle (GoblinTriangle.cpp:38-163)
// ---------------------------------------------------------------------------
inline bool tri_test(const Mesh& m, uint32_t t, const Ray& ray, float* t_out, float* b1_out, float* b2_out) {
    V3 p0 = m.P(m.idx[3 * t]), p1 = m.P(m.idx[3 * t + 1]), p2 = m.P(m.idx[3 * t + 2]);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 s1 = cross(ray.d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.0f) return false;
    float inv = 1.0f / divisor;
    const float eps = 1e-7f;
    V3 sv = ray.o - p0;
    float b1 = dot(sv, s1) * inv;
    if (b1 + eps < 0.0f || b1 - eps > 1.0f) return false;
    V3 s2 = cross(sv, e1);
    float b2 = dot(ray.d, s2) * inv;
    if (b2 + eps < 0.0f || b1 + b2 - eps > 1.0f) return false;
    float tt = dot(e2, s2) * inv;
    if (tt < ray.mint || tt > ray.maxt) return false;
    *t_out = tt;
    *b1_out = b1;
    *b2_out = b2;
    return true;
}

// Fragment outputs of Triangle::intersect (:77-123).  `prev` is the fragment
// the caller passed in: the degenerate-uv branch reads it before overwriting
// (:113-117).
inline void tri_fragment(const Mesh& m, uint32_t t, const Ray& ray, float tt, float b1, float b2, Frag* f) {
    uint32_t i0 = m.idx[3 * t], i1 = m.idx[3 * t + 1], i2 = m.idx[3 * t + 2];
    V3 p0 = m.P(i0), p1 = m.P(i1), p2 = m.P(i2);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    float b0 = 1.0f - b1 - b2;
    V3 position = ray.o + tt * ray.d;
    V3 normal;
    if (m.has_n) normal = normalize(b0 * m.N(i0) + b1 * m.N(i1) + b2 * m.N(i2));
    else normal = normalize(cross(e1, e2));
    float u0, v0, u1, v1, u2, v2;
    if (m.has_uv) {
        u0 = m.uv[2 * i0]; v0 = m.uv[2 * i0 + 1];
        u1 = m.uv[2 * i1]; v1 = m.uv[2 * i1 + 1];
        u2 = m.uv[2 * i2]; v2 = m.uv[2 * i2 + 1];
    } else {
        u0 = 0.0f; v0 = 0.0f; u1 = 1.0f; v1 = 0.0f; u2 = 0.0f; v2 = 1.0f;
    }
    // Vector2: b0*uv0 + b1*uv1 + b2*uv2
    float u = b0 * u0 + b1 * u1 + b2 * u2;
    float v = b0 * v0 + b1 * v1 + b2 * v2;
    float du1 = u1 - u0, dv1 = v1 - v0, du2 = u2 - u0, dv2 = v2 - v0;
    float det = du1 * dv2 - dv1 * du2;
    V3 dpdu, dpdv;
    if (det == 0.0f) {
        dpdu = normalize(e1 - dot(f->n, e1) * f->n);
        dpdv = cross(f->n, f->dpdv);
    } else {
        float inv_det = 1.0f / det;
        dpdu = inv_det * (dv2 * e1 - dv1 * e2);
        dpdv = inv_det * (-du2 * e1 + du1 * e2);
    }
    f->p = position;
    f->n = normal;
    f->u = u;
    f->v = v;
    f->dpdu = dpdu;
    f->dpdv = dpdv;
    // `*fragment = Fragment(position, normal, uv, dpdu, dpdv)`: the constructor zeroes the differentials (GoblinGeometry.cpp:7-12),
    // which Material::perturb's lookups read before computeUVDifferential sets them
    f->dpdx = f->dpdy = V3(0, 0, 0);
    f->dudx = f->dvdx = f->dudy = f->dvdy = 0.0f;
}

// quadratic, GoblinUtils.cpp:93-113
inline bool quadratic(float A, float B, float C, float* t1, float* t2) {
    float discriminant = B * B - 4.0f * A * C;
    if (discriminant < 0.0f) return false;
    float root = std::sqrt(discriminant);
    float q;
    if (B < 0) q = -0.5f * (B - root);
    else q = -0.5f * (B + root);
    *t1 = q / A;
    *t2 = C / q;
    if (*t1 > *t2) std::swap(*t1, *t2);
    return true;
}

// Sphere::intersect / occluded up to the accepted distance (GoblinSphere.cpp:12-31, 88-107)
inline bool sphere_test(float radius, const Ray& ray, float* t_out) {
    float A = sqlen(ray.d);
    float B = 2.0f * dot(ray.d, ray.o);
    float C = sqlen(ray.o) - radius * radius;
    float t_near, t_far;
    if (!quadratic(A, B, C, &t_near, &t_far)) return false;
    if (t_near > ray.maxt || t_far < ray.mint) return false;
    float t_hit = t_near;
    if (t_hit < ray.mint) {
        t_hit = t_far;
        if (t_hit > ray.maxt) return false;
    }
    *t_out = t_hit;
    return true;
}
// the Fragment Sphere::intersect fills (:32-86)
inline void sphere_fragment(float radius, const Ray& ray, float t_hit, Frag* f) {
    V3 p = ray.o + t_hit * ray.d;
    float phi = std::atan2(p.y, p.x);
    if (phi < 0.0f) phi += TWO_PI;
    float u = phi * INV_TWOPI;
    float theta = std::acos(p.z / radius);
    float v = theta * INV_PI;
    float inv_r = 1.0f / std::sqrt(p.x * p.x + p.y * p.y);
    float cos_phi = p.x * inv_r;
    float sin_phi = p.y * inv_r;
    f->p = p;
    f->n = normalize(p);
    f->u = u;
    f->v = v;
    f->dpdu = V3(-TWO_PI * p.y, TWO_PI * p.x, 0.0f);
    f->dpdv = PI * V3(p.z * cos_phi, p.z * sin_phi, -radius * std::sin(theta));
}
// Disk::intersect / occluded (GoblinDisk.cpp:12-31, 63-74).  The two differ at the rim: intersect rejects
// squareR > r^2, occluded accepts squareR <= r^2 -- the same set.
inline bool disk_test(float radius, const Ray& ray, float* t_out) {
    if (std::fabs(ray.d.z) < 1e-7f) return false;
    float t = -ray.o.z / ray.d.z;
    V3 p = ray.o + t * ray.d;
    if (t < ray.mint || t > ray.maxt) return false;
    float square_r = p.x * p.x + p.y * p.y;
    if (square_r > radius * radius) return false;
    *t_out = t;
    return true;
}
inline void disk_fragment(float radius, const Ray& ray, float t, Frag* f) {   // :33-60
    V3 p = ray.o + t * ray.d;
    float square_r = p.x * p.x + p.y * p.y;
    float r = std::sqrt(square_r);
    float phi = std::atan2(p.y, p.x);
    if (phi < 0.0f) phi += TWO_PI;
    f->p = p;
    f->n = V3(0.0f, 0.0f, 1.0f);
    f->u = phi * INV_TWOPI;
    f->v = r / radius;
    f->dpdu = V3(-TWO_PI * p.y, TWO_PI * p.x, 0.0f);
    f->dpdv = V3(radius * p.x / r, radius * p.y / r, 0.0f);
}
inline bool shape_test(const Mesh& m, const Ray& ray, float* t_out) {
    return m.shape == GBL_SHAPE_SPHERE ? sphere_test(m.radius, ray, t_out) : disk_test(m.radius, ray, t_out);
}
inline void shape_fragment(const Mesh& m, const Ray& ray, float t, Frag* f) {
    if (m.shape == GBL_SHAPE_SPHERE) sphere_fragment(m.radius, ray, t, f);
    else disk_fragment(m.radius, ray, t, f);
    f->dpdx = f->dpdy = V3(0, 0, 0);   // `*fragment = Fragment(...)`, as in tri_fragment
    f->dudx = f->dvdx = f->dudy = f->dvdy = 0.0f;
}

struct Hit {
    Frag frag;
    int instance = -1;
    float epsilon = 0.0f;
};

// Scene::intersect -> TLAS -> InstancedPrimitive::intersect -> Model BLAS ->
// Triangle::intersect (GoblinScene.cpp:75-83, GoblinPrimitive.cpp:103-112,
// GoblinModel.cpp:39-55).  ray.maxt shrinks in place.  Frag state persists
// across candidate hits exactly as the reference's single Intersection does.
bool scene_intersect_impl(const orc_scene* s, Ray& ray, Hit* hit, Counters* cnt, int filter);
Col tex_lookup(const orc_scene* sc, int id, const Frag& f);
// Material::perturb -> BumpShaders::evaluate (GoblinMaterial.cpp:221-283), run by Scene::intersect on the closest hit's
// fragment (GoblinScene.cpp:75-83) -- before any ray differential is attached to it, so the lookups see a zero footprint.
// MaskMaterial::perturb forwards to the wrapped material (GoblinMaterial.h:456-458).
inline void perturb_fragment(const orc_scene* s, int material, Frag* f) {
    const gbl_material* m = &s->materials[material];
    if (m->type == GBL_MAT_MASK) m = &s->materials[m->masked_material];
    if (m->tex_bump >= 0) {
        const V3 p = f->p, n = f->n;
        const float u = f->u, v = f->v;
        const float bump_d = tex_lookup(s, m->tex_bump, *f).r;
        const float du = 0.002f;
        Frag fdu = *f;
        fdu.p = p + du * f->dpdu;
        fdu.u = u + du;
        fdu.v = v + 0.0f;
        const float bump_ddu = tex_lookup(s, m->tex_bump, fdu).r;
        const V3 bump_dpdu = f->dpdu + (bump_ddu - bump_d) / du * n;
        const float dv = 0.002f;
        Frag fdv = *f;
        fdv.p = p + dv * f->dpdv;
        fdv.u = u + 0.0f;
        fdv.v = v + dv;
        const float bump_ddv = tex_lookup(s, m->tex_bump, fdv).r;
        const V3 bump_dpdv = f->dpdv + (bump_ddv - bump_d) / dv * n;
        V3 bump_n = normalize(cross(bump_dpdu, bump_dpdv));
        if (dot(bump_n, n) < 0.0f) bump_n = bump_n * -1.0f;
        f->n = bump_n;
        f->dpdu = bump_dpdu;
        f->dpdv = bump_dpdv;
    }
    if (m->tex_normal >= 0) {
        const Col c = tex_lookup(s, m->tex_normal, *f);
        V3 ns(c.r, c.g, c.b);
        ns = 2.0f * ns - V3(1.0f, 1.0f, 1.0f);
        // Fragment::getWorldToShade rows t, b, n; its transpose times nShade
        const V3 n = f->n;
        const V3 t = normalize(f->dpdu - n * dot(f->dpdu, n));
        const V3 b = cross(n, t);
        V3 nw(t.x * ns.x + b.x * ns.y + n.x * ns.z, t.y * ns.x + b.y * ns.y + n.y * ns.z, t.z * ns.x + b.z * ns.y + n.z * ns.z);
        nw = normalize(nw);
        if (dot(nw, f->n) < 0.0f) nw = nw * -1.0f;
        f->n = nw;
    }
}
bool scene_intersect(const orc_scene* s, Ray& ray, Hit* hit, Counters* cnt, int filter = FILTER_NONE) {
    const Ray in = ray;
    const bool any = scene_intersect_impl(s, ray, hit, cnt, filter);
    if (any) perturb_fragment(s, static_cast<int>(s->instances[hit->instance].material), &hit->frag);
    if (cnt->log) {
        V3 fn(0, 0, 0), ft(0, 0, 0);
        if (any) {   // Fragment::getWorldToShade's n and t rows
            fn = hit->frag.n;
            ft = normalize(hit->frag.dpdu - fn * dot(hit->frag.dpdu, fn));
        }
        const float rec[16] = {0.0f, in.o.x, in.o.y, in.o.z, in.d.x, in.d.y, in.d.z, in.mint, in.maxt, any ? ray.maxt : -1.0f,
                               fn.x, fn.y, fn.z, ft.x, ft.y, ft.z};
        cnt->log->insert(cnt->log->end(), rec, rec + 16);
    }
    return any;
}
bool scene_intersect_impl(const orc_scene* s, Ray& ray, Hit* hit, Counters* cnt, int filter) {
    ++cnt->closest;
    bool any = false;
    traverse(s->tlas, ray, cnt, [&](uint32_t inst_id) {
        const Instance& in = s->instances[inst_id];
        // Model::intersect: `if (f != nullptr && !f(this, ray)) return false` (GoblinModel.cpp:44-46, 30-32)
        if (filter != FILTER_NONE && (in.is_mask ? FILTER_MASK : FILTER_OPAQUE) != filter) return false;
        const Mesh& m = s->meshes[in.mesh];
        Ray r;   // Transform::invertRay: d is NOT renormalised, so t is shared
        r.o = in.xf.invert_point(ray.o);
        r.d = in.xf.invert_vector(ray.d);
        r.mint = ray.mint;
        r.maxt = ray.maxt;
        bool h = false;
        if (m.shape != GBL_SHAPE_MESH) {   // Model::intersect without a BVH (GoblinModel.cpp:46-54)
            ++cnt->tris;
            float tt;
            if (shape_test(m, r, &tt)) {
                r.maxt = tt;
                hit->epsilon = 1e-3f * tt;
                shape_fragment(m, r, tt, &hit->frag);
                hit->instance = static_cast<int>(inst_id);
                h = true;
            }
        } else
        traverse(m.bvh, r, cnt, [&](uint32_t tri) {
            ++cnt->tris;
            float tt, b1, b2;
            if (tri_test(m, tri, r, &tt, &b1, &b2)) {
                r.maxt = tt;
                hit->epsilon = 1e-3f * tt;
                tri_fragment(m, tri, r, tt, b1, b2, &hit->frag);
                hit->instance = static_cast<int>(inst_id);
                h = true;
            }
            return false;
        });
        if (h) {
            // Fragment::transform, GoblinGeometry.cpp:31-37
            Frag& f = hit->frag;
            f.p = in.xf.on_point(f.p);
            f.n = normalize(in.xf.on_normal(f.n));
            f.dpdu = in.xf.on_vector(f.dpdu);
            f.dpdv = in.xf.on_vector(f.dpdv);
            ray.maxt = r.maxt;
            any = true;
        }
        return false;
    });
    return any;
}

bool scene_occluded(const orc_scene* s, const Ray& ray, Counters* cnt, int filter = FILTER_NONE) {   // GoblinScene.cpp:85-87, GoblinBVH.cpp:189-232
    ++cnt->anyhit;
    bool occ = false;
    traverse(s->tlas, ray, cnt, [&](uint32_t inst_id) {
        const Instance& in = s->instances[inst_id];
        if (filter != FILTER_NONE && (in.is_mask ? FILTER_MASK : FILTER_OPAQUE) != filter) return false;
        const Mesh& m = s->meshes[in.mesh];
        Ray r;
        r.o = in.xf.invert_point(ray.o);
        r.d = in.xf.invert_vector(ray.d);
        r.mint = ray.mint;
        r.maxt = ray.maxt;
        if (m.shape != GBL_SHAPE_MESH) {
            ++cnt->tris;
            float tt;
            if (shape_test(m, r, &tt)) occ = true;
            return occ;
        }
        traverse(m.bvh, r, cnt, [&](uint32_t tri) {
            ++cnt->tris;
            float tt, b1, b2;
            if (tri_test(m, tri, r, &tt, &b1, &b2)) {
                occ = true;
                return true;
            }
            return false;
        });
        return occ;
    });
    if (cnt->log) {
        const float rec[16] = {1.0f, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, ray.mint, ray.maxt, occ ? 1.0f : 0.0f,
                               0, 0, 0, 0, 0, 0};
        cnt->log->insert(cnt->log->end(), rec, rec + 16);
    }
    return occ;
}

// A full traversal whose every leaf is rejected by the intersect filter before
// the geometry test (Model::intersect with notOpaque in a mask-free scene,
// GoblinModel.cpp:44-46): what evalAttenuation costs the reference
// (GoblinPathtracer.cpp:21-48).  Only run when ref_faithful != 0.
void scene_filtered_traversal(const orc_scene* s, const Ray& ray, Counters* cnt) {
    ++cnt->filtered;
    traverse(s->tlas, ray, cnt, [&](uint32_t inst_id) {
        const Instance& in = s->instances[inst_id];
        const Mesh& m = s->meshes[in.mesh];
        Ray r;
        r.o = in.xf.invert_point(ray.o);
        r.d = in.xf.invert_vector(ray.d);
        r.mint = ray.mint;
        r.maxt = ray.maxt;
        if (m.shape == GBL_SHAPE_MESH) traverse(m.bvh, r, cnt, [&](uint32_t) { return false; });
        return false;
    });
}

// Intersection::computeUVDifferential, GoblinPrimitive.cpp:32-97
inline void compute_uv_differential(Frag* f, const RayDiff* rd) {
    float dudx = 0.0f, dvdx = 0.0f, dudy = 0.0f, dvdy = 0.0f;
    f->dpdx = f->dpdy = V3(0, 0, 0);   // every intersection builds a fresh Fragment (GoblinGeometry.cpp:7-13)
    if (rd && rd->has) {
        V3 p = f->p, n = f->n;
        float minus_d = dot(p, n);
        float tdx = (minus_d - dot(rd->dxo, n)) / dot(rd->dxd, n);
        float tdy = (minus_d - dot(rd->dyo, n)) / dot(rd->dyd, n);
        if (!std::isnan(tdx) && !std::isnan(tdy)) {
            V3 pdx = rd->dxo + tdx * rd->dxd;
            V3 pdy = rd->dyo + tdy * rd->dyd;
            V3 dpdx = pdx - p, dpdy = pdy - p;
            f->dpdx = dpdx;
            f->dpdy = dpdy;
            int axis[2];
            if (std::fabs(n.x) > std::fabs(n.y) && std::fabs(n.x) > std::fabs(n.z)) {
                axis[0] = 1; axis[1] = 2;
            } else if (std::fabs(n.y) > std::fabs(n.z)) {
                axis[0] = 0; axis[1] = 2;
            } else {
                axis[0] = 0; axis[1] = 1;
            }
            float A[2][2] = {{f->dpdu[axis[0]], f->dpdv[axis[0]]}, {f->dpdu[axis[1]], f->dpdv[axis[1]]}};
            auto solve = [&](const float B[2], float* x, float* y) {   // solve2x2LinearSystem, GoblinUtils.h:151-163
                float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
                if (std::fabs(det) < 1e-10f) return false;
                *x = (+A[1][1] * B[0] - A[0][1] * B[1]) / det;
                *y = (-A[1][0] * B[0] + A[0][0] * B[1]) / det;
                if (std::isnan(*x) || std::isnan(*y)) return false;
                return true;
            };
            float Bx[2] = {dpdx[axis[0]], dpdx[axis[1]]};
            if (!solve(Bx, &dudx, &dvdx)) dudx = dvdx = 0.0f;
            float By[2] = {dpdy[axis[0]], dpdy[axis[1]]};
            if (!solve(By, &dudy, &dvdy)) dudy = dvdy = 0.0f;
        }
    }
    f->dudx = dudx; f->dvdx = dvdx; f->dudy = dudy; f->dvdy = dvdy;
}

// TextureMapping::map (UVMapping / SphericalMapping, GoblinTexture.cpp:296-347)
struct TexCoord {
    float s, t, dsdx, dtdx, dsdy, dtdy;
};
inline void point_to_st(const Xform& to_tex, V3 p, float* s, float* t) {
    V3 v = normalize(to_tex.on_point(p) - V3(0.0f, 0.0f, 0.0f));
    float theta = std::acos(std::min(std::max(v.z, -1.0f), 1.0f));   // sphericalTheta / clamp, GoblinUtils.h:142-149
    float phi = std::atan2(v.y, v.x);
    phi = phi < 0.0f ? phi + TWO_PI : phi;
    *s = phi * INV_TWOPI;
    *t = theta * INV_PI;
}
inline TexCoord tex_map(const orc_scene* sc, uint32_t id, const Frag& f) {
    const gbl_texture& g = sc->textures[id];
    TexCoord tc;
    if (g.mapping == GBL_MAP_SPHERICAL) {
        const Xform& xf = sc->tex_xf[id];
        point_to_st(xf, f.p, &tc.s, &tc.t);
        float sdx, tdx, sdy, tdy;
        point_to_st(xf, f.p + f.dpdx, &sdx, &tdx);
        point_to_st(xf, f.p + f.dpdy, &sdy, &tdy);
        float dsdx = sdx - tc.s;
        if (dsdx > 0.5f) dsdx -= 1.0f;
        else if (dsdx < -0.5f) dsdx += 1.0f;
        float dsdy = sdy - tc.s;
        if (dsdy > 0.5f) dsdy -= 1.0f;
        else if (dsdy < -0.5f) dsdy += 1.0f;
        tc.dsdx = dsdx;
        tc.dtdx = tdx - tc.t;
        tc.dsdy = dsdy;
        tc.dtdy = tdy - tc.t;
    } else {
        tc.s = g.uv_scale[0] * f.u + g.uv_offset[0];
        tc.t = g.uv_scale[1] * f.v + g.uv_offset[1];
        tc.dsdx = g.uv_scale[0] * f.dudx;
        tc.dtdx = g.uv_scale[1] * f.dvdx;
        tc.dsdy = g.uv_scale[0] * f.dudy;
        tc.dtdy = g.uv_scale[1] * f.dvdy;
    }
    return tc;
}
inline float integrate_checker(float x) {   // GoblinTexture.cpp:371-375
    float x_half = 0.5f * x;
    return std::floor(x_half) + 2.0f * std::max(x_half - std::floor(x_half) - 0.5f, 0.0f);
}
// The reference's own log2 (GoblinUtils.h:84-87; the Goblin namespace shadows libm's): logf times a float 1 / ln 2.
inline float goblin_log2(float n) {
    static const float inv_log2 = 1.0f / logf(2.0f);
    return logf(n) * inv_log2;
}
// ---- MIPMap<T> (GoblinTexture.cpp:40-291).  The levels come built (gbl_image); T = float is kept in .r of a Col.
inline void level_dims(const gbl_image& im, int level, int* w, int* h, size_t* off) {   // (declared above)
    size_t o = 0;
    for (int l = 0; l < level; ++l) o += static_cast<size_t>(std::max(1u, im.width >> l)) * std::max(1u, im.height >> l) * im.channels;
    *w = static_cast<int>(std::max(1u, im.width >> level));
    *h = static_cast<int>(std::max(1u, im.height >> level));
    *off = static_cast<size_t>(im.texel_offset) + o;
}
Col image_texel(const orc_scene* sc, const gbl_image& im, int level, int s, int t, uint32_t mode) {   // ImageBuffer<T>::texel, :10-38
    int w, h;
    size_t off;
    level_dims(im, level, &w, &h, &off);
    if (mode == GBL_ADDRESS_CLAMP) {
        s = std::min(std::max(s, 0), w - 1);
        t = std::min(std::max(s, 0), h - 1);   // sic: clamp(s, 0, height - 1), :15
    } else if (mode == GBL_ADDRESS_BORDER) {
        if (s < 0 || t < 0 || s >= w || t >= h) return Col(0.0f);
    } else {
        s = s % w;
        t = t % h;
        if (s < 0) s += w;
        if (t < 0) t += h;
    }
    const float* p = sc->texels.data() + off + (static_cast<size_t>(t) * w + s) * im.channels;
    if (im.channels == 1) return Col(p[0], p[0], p[0]);
    return Col(p[0], p[1], p[2]);
}
Col mip_level(const orc_scene* sc, const gbl_image& im, int level, float s, float t, uint32_t mode) {   // MIPMap<T>::lookup(level, ...), :276-291
    // (the reference clamps to [0, levels] and reads mPyramid[levels] out of bounds at the top: clamped to the last level here)
    level = std::min(std::max(level, 0), static_cast<int>(im.levels) - 1);
    int w, h;
    size_t off;
    level_dims(im, level, &w, &h, &off);
    float s_res = s * w - 0.5f, t_res = t * h - 0.5f;
    int s0 = floor_int(s_res), t0 = floor_int(t_res);
    float ds = s_res - (float)s0, dt = t_res - (float)t0;
    return (1.0f - ds) * (1.0f - dt) * image_texel(sc, im, level, s0, t0, mode) + (ds) * (1.0f - dt) * image_texel(sc, im, level, s0 + 1, t0, mode) +
           (1.0f - ds) * (dt)*image_texel(sc, im, level, s0, t0 + 1, mode) + (ds) * (dt)*image_texel(sc, im, level, s0 + 1, t0 + 1, mode);
}
Col mip_trilinear(const orc_scene* sc, const gbl_image& im, float s, float t, float width, uint32_t mode) {   // :112-127
    int levels = static_cast<int>(im.levels);
    float level = levels - 1 + goblin_log2(std::max(width, 1e-8f));
    int il = floor_int(level);
    if (il < 0) return mip_level(sc, im, 0, s, t, mode);
    if (il >= levels - 1) return mip_level(sc, im, levels - 1, s, t, mode);
    float delta = level - (float)il;
    return (1.0f - delta) * mip_level(sc, im, il, s, t, mode) + (delta)*mip_level(sc, im, il + 1, s, t, mode);
}
Col mip_ewa_level(const orc_scene* sc, const gbl_image& im, bool is_float, int level, float s, float t, float A, float B, float C, uint32_t mode) {   // :181-259
    int w, h;
    size_t off;
    level_dims(im, level, &w, &h, &off);
    float s_res = (float)w, t_res = (float)h;
    s = s * w - 0.5f;
    t = t * h - 0.5f;
    A = A / (s_res * s_res);
    B = B / (s_res * t_res);
    C = C / (t_res * t_res);
    float inv_det = 1.0f / (-B * B + 4.0f * A * C);
    float off_s = 2.0f * sqrtf(C * inv_det), off_t = 2.0f * sqrtf(A * inv_det);
    int s0 = static_cast<int>(ceilf(s - off_s)), s1 = floor_int(s + off_s), t0 = static_cast<int>(ceilf(t - off_t)), t1 = floor_int(t + off_t);
    float weight_sum = 0.0f;
    Col result(0.0f);
    for (int is = s0; is <= s1; ++is) {
        for (int it = t0; it <= t1; ++it) {
            float ss = is - s, tt = it - t;
            float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 <= 1.0f) {
                size_t li = std::min((size_t)floor_int(r2 * 128), (size_t)127);
                float weight = sc->ewa_lut[li];
                result += weight * image_texel(sc, im, level, is, it, mode);
                weight_sum += weight;
            }
        }
    }
    if (weight_sum > 0.0f) {
        if (is_float) return Col(result.r / weight_sum, result.g / weight_sum, result.b / weight_sum);   // float /= float
        return result / weight_sum;                                                                        // Color /= float: times 1 / s
    }
    return image_texel(sc, im, level, (int)s, (int)t, mode);
}
Col mip_lookup(const orc_scene* sc, const gbl_image& im, bool is_float, const TexCoord& tc, uint32_t filter, uint32_t mode, float max_aniso) {   // :78-97
    if (filter == GBL_IMAGE_FILTER_BILINEAR) {
        float width = std::max(std::max(std::fabs(tc.dsdx), std::fabs(tc.dtdx)), std::max(std::fabs(tc.dsdy), std::fabs(tc.dtdy)));
        float level = static_cast<int>(im.levels) - 1 + goblin_log2(std::max(width, 1e-8f));
        return mip_level(sc, im, floor_int(level + 0.5f), tc.s, tc.t, mode);
    }
    if (filter == GBL_IMAGE_FILTER_TRILINEAR) {
        float width = std::max(std::max(std::fabs(tc.dsdx), std::fabs(tc.dtdx)), std::max(std::fabs(tc.dsdy), std::fabs(tc.dtdy)));
        return mip_trilinear(sc, im, tc.s, tc.t, width, mode);
    }
    if (filter == GBL_IMAGE_FILTER_EWA) {   // lookupEWA, :129-179
        float ds0 = tc.dsdx, dt0 = tc.dtdx, ds1 = tc.dsdy, dt1 = tc.dtdy;
        float major = sqrtf(ds0 * ds0 + dt0 * dt0), minor = sqrtf(ds1 * ds1 + dt1 * dt1);
        if (major < minor) {
            std::swap(ds0, ds1);
            std::swap(dt0, dt1);
            std::swap(major, minor);
        }
        if (minor * max_aniso < major && minor > 0.0f) {
            float scale = major / (minor * max_aniso);
            minor *= scale;
            ds1 *= scale;
            dt1 *= scale;
        }
        float A = dt0 * dt0 + dt1 * dt1;
        float B = -2.0f * (ds0 * dt0 + ds1 * dt1);
        float C = ds0 * ds0 + ds1 * ds1;
        float F = A * C - 0.25f * B * B;
        if (minor == 0.0f || F <= 0.0f) return mip_trilinear(sc, im, tc.s, tc.t, minor, mode);
        float inv_f = 1.0f / F;
        A *= inv_f;
        B *= inv_f;
        C *= inv_f;
        int levels = static_cast<int>(im.levels);
        float level = levels - 1 + goblin_log2(minor);
        int il = floor_int(level);
        if (il < 0) return mip_level(sc, im, 0, tc.s, tc.t, mode);
        if (il >= levels - 1) return mip_level(sc, im, levels - 1, tc.s, tc.t, mode);
        float delta = level - (float)il;
        return (1.0f - delta) * mip_ewa_level(sc, im, is_float, il, tc.s, tc.t, A, B, C, mode) +
               (delta)*mip_ewa_level(sc, im, is_float, il + 1, tc.s, tc.t, A, B, C, mode);
    }
    return mip_level(sc, im, 0, tc.s, tc.t, mode);   // FilterNone: lookupNearest = the bilinear lookup of level 0 (:99-102)
}

// Texture<T>::lookup for T = Color (is_float == 0) and T = float (value in .r)
Col tex_lookup(const orc_scene* sc, int id, const Frag& f) {
    const gbl_texture& g = sc->textures[id];
    if (g.type == GBL_TEX_IMAGE) {   // ImageTexture<T>::lookup, :452-456
        TexCoord tc = tex_map(sc, id, f);
        return mip_lookup(sc, sc->images[g.image], g.is_float != 0, tc, g.image_filter, g.address, g.max_anisotropy);
    }
    if (g.type == GBL_TEX_CONSTANT) return g.is_float ? Col(g.value[0], g.value[0], g.value[0]) : Col(g.value[0], g.value[1], g.value[2]);
    if (g.type == GBL_TEX_SCALE) {   // mScale->lookup(f) * mTexture->lookup(f), :421-425
        float sc_v = tex_lookup(sc, g.child[1], f).r;
        Col t = tex_lookup(sc, g.child[0], f);
        if (g.is_float) {
            float v = sc_v * t.r;
            return Col(v, v, v);
        }
        return sc_v * t;
    }
    // CheckboardTexture<T>::lookup, :377-416
    TexCoord tc = tex_map(sc, id, f);
    float s = tc.s, t = tc.t;
    auto pick = [&]() { return (floor_int(s) + floor_int(t)) % 2 == 0 ? tex_lookup(sc, g.child[0], f) : tex_lookup(sc, g.child[1], f); };
    if (!g.filter) return pick();
    float ds = std::max(std::fabs(tc.dsdx), std::fabs(tc.dsdy));
    float dt = std::max(std::fabs(tc.dtdx), std::fabs(tc.dtdy));
    float s0 = s - ds, s1 = s + ds, t0 = t - dt, t1 = t + dt;
    if (floor_int(s0) == floor_int(s1) && floor_int(t0) == floor_int(t1)) return pick();
    float s_ratio = (integrate_checker(s1) - integrate_checker(s0)) / (2.0f * ds);
    float t_ratio = (integrate_checker(t1) - integrate_checker(t0)) / (2.0f * dt);
    float area2 = s_ratio + t_ratio - 2.0f * s_ratio * t_ratio;
    if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
    Col a = tex_lookup(sc, g.child[0], f), b = tex_lookup(sc, g.child[1], f);
    if (g.is_float) {
        float v = (1.0f - area2) * a.r + area2 * b.r;
        return Col(v, v, v);
    }
    return (1.0f - area2) * a + area2 * b;
}
// The material with every texture slot evaluated at this fragment: all lookups a bounce makes see the same
// Fragment, so they all return these values.
inline gbl_material resolve_material(const orc_scene* sc, const gbl_material& m, const Frag& f) {
    if (m.tex_color < 0 && m.tex_color2 < 0 && m.tex_exponent < 0 && m.tex_color3 < 0) return m;
    gbl_material r = m;
    if (m.tex_color3 >= 0) {
        Col c = tex_lookup(sc, m.tex_color3, f);
        r.color3[0] = c.r; r.color3[1] = c.g; r.color3[2] = c.b;
    }
    if (m.tex_color >= 0) {
        Col c = tex_lookup(sc, m.tex_color, f);
        r.color[0] = c.r; r.color[1] = c.g; r.color[2] = c.b;
    }
    if (m.tex_color2 >= 0) {
        Col c = tex_lookup(sc, m.tex_color2, f);
        r.color2[0] = c.r; r.color2[1] = c.g; r.color2[2] = c.b;
    }
    if (m.tex_exponent >= 0) r.exponent = tex_lookup(sc, m.tex_exponent, f).r;
    return r;
}

// A hit's material ready to evaluate.  For MaskMaterial (GoblinMaterial.cpp:747-811) `m` is the wrapped material,
// `alpha` / `tcolor` the mask's own two lookups.
struct ResolvedMat {
    gbl_material m;
    bool is_mask = false;
    float alpha = 1.0f;
    Col tcolor = Col(1.0f);
};
inline ResolvedMat resolve_hit_material(const orc_scene* sc, uint32_t material, const Frag& f) {
    ResolvedMat r;
    gbl_material outer = resolve_material(sc, sc->materials[material], f);
    if (outer.type != GBL_MAT_MASK) {
        r.m = outer;
        return r;
    }
    r.is_mask = true;
    r.alpha = outer.exponent;
    r.tcolor = Col(outer.color[0], outer.color[1], outer.color[2]);
    r.m = resolve_material(sc, sc->materials[outer.masked_material], f);
    return r;
}

// Fragment::getWorldToShade rows t, b, n (GoblinGeometry.cpp:17-29)
struct Frame {
    V3 t, b, n;
};
inline Frame shade_frame(const Frag& f) {
    Frame fr;
    fr.n = f.n;
    fr.t = normalize(f.dpdu - fr.n * dot(f.dpdu, fr.n));
    fr.b = cross(fr.n, fr.t);
    return fr;
}
// shadeToWorld * v with shadeToWorld = worldToShade^T (Matrix3 * Vector3 row dot products)
inline V3 shade_to_world(const Frame& fr, V3 v) {
    return V3(fr.t.x * v.x + fr.b.x * v.y + fr.n.x * v.z, fr.t.y * v.x + fr.b.y * v.y + fr.n.y * v.z,
              fr.t.z * v.x + fr.b.z * v.y + fr.n.z * v.z);
}

// ---------------------------------------------------------------------------
// Sampling warps (GoblinSampler.cpp:420-575)
// ---------------------------------------------------------------------------
inline V3 cosine_sample_hemisphere(float u1, float u2) {   // :549-557
    float sin_t = sqrtf(u1);
    float cos_t = sqrtf(std::max(0.0f, 1.0f - u1));
    float phi = TWO_PI * u2;
    float z = cos_t;
    float x = sin_t * std::cos(phi);
    float y = sin_t * std::sin(phi);
    return V3(x, y, z);
}
inline V3 uniform_sample_hemisphere(float u1, float u2) {   // :517-524
    float z = u1;
    float sin_t = sqrtf(std::max(0.0f, 1.0f - u1 * u1));
    float phi = TWO_PI * u2;
    float x = sin_t * std::cos(phi);
    float y = sin_t * std::sin(phi);
    return V3(x, y, z);
}
inline float power_heuristic(float na, float pa, float nb, float pb) {   // GoblinSampler.h:286-290
    float A = na * pa, B = nb * pb;
    return A * A / (A * A + B * B);
}

// ---------------------------------------------------------------------------
// Materials (GoblinMaterial.cpp)
// ---------------------------------------------------------------------------
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_NULL = 32, BSDF_ALL = 63 };

inline int material_type_bits(const gbl_material& m) {
    switch (m.type) {
        case GBL_MAT_BLINN: return BSDF_GLOSSY | BSDF_REFLECTION;
        case GBL_MAT_TRANSPARENT: return BSDF_SPECULAR | BSDF_REFLECTION | BSDF_TRANSMISSION;
        case GBL_MAT_MIRROR: return BSDF_SPECULAR | BSDF_REFLECTION;
        case GBL_MAT_SUBSURFACE: return BSDF_ALL;   // GoblinMaterial.h:393,401
        default: return BSDF_DIFFUSE | BSDF_REFLECTION;
    }
}
inline bool match_type(int type, int to_match) { return (type & to_match) == to_match; }
inline int sample_type(V3 wo, V3 wi, V3 n, int type) {   // Material::getSampleType, :285-294
    if (dot(n, wo) * dot(n, wi) > 0.0f) return type & ~BSDF_TRANSMISSION;
    return type & ~BSDF_REFLECTION;
}
inline bool same_hemisphere(V3 n, V3 wo, V3 wi) { return dot(wo, n) * dot(wi, n) > 0.0f; }

float fresnel_dielectric(float cosi, float etai, float etat) {   // :390-406
    cosi = clampf(cosi, -1.0f, 1.0f);
    float sint = (etai / etat) * std::sqrt(std::max(0.0f, 1.0f - cosi * cosi));
    if (sint >= 1.0f) return 1.0f;
    float cost = std::sqrt(std::max(0.0f, 1 - sint * sint));
    cosi = std::fabs(cosi);
    float r_parl = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float r_perp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
float fresnel_conductor(float cosi, float eta, float k) {   // :408-416
    float tmp = (eta * eta + k * k);
    float cosi2 = cosi * cosi;
    float r_parl2 = (tmp * cosi2 - 2.0f * eta * cosi + 1.0f) / (tmp * cosi2 + 2.0f * eta * cosi + 1.0f);
    float r_perp2 = (tmp - 2.0f * eta * cosi + cosi2) / (tmp + 2.0f * eta * cosi + cosi2);
    return (r_parl2 + r_perp2) * 0.5f;
}
float specular_reflect_dielectric(V3 n, V3 wo, V3* wi, float etai, float etat) {   // :306-327
    float cosi = dot(n, wo);
    float ei = etai, et = etat;
    if (!(cosi > 0.0f)) {
        std::swap(ei, et);
        n = -n;
        cosi = -cosi;
    }
    float f = fresnel_dielectric(cosi, ei, et);
    *wi = 2 * cosi * n - wo;
    float cosr = cosi;
    return f / cosr;
}
float specular_reflect_conductor(V3 n, V3 wo, V3* wi, float eta, float k) {   // :329-341
    float cosi = dot(n, wo);
    if (cosi <= 0.0f) return 0.0f;
    float f = fresnel_conductor(cosi, eta, k);
    *wi = 2 * cosi * n - wo;
    float cosr = cosi;
    return f / cosr;
}
float specular_refract(V3 n, V3 wo, V3* wi, float etao, float etai) {   // :343-387, radiance mode
    float coso = dot(n, wo);
    float et = etao, ei = etai;
    if (!(coso > 0.0f)) {
        std::swap(ei, et);
        n = -n;
        coso = -coso;
    }
    float f = fresnel_dielectric(coso, et, ei);
    if (f == 1.0f) return 0.0f;
    float eta = et / ei;
    *wi = normalize(n * (eta * coso - std::sqrt(std::max(0.0f, 1.0f - eta * eta * (1.0f - coso * coso)))) - eta * wo);
    return eta * eta * (1.0f - f) / absdot(*wi, n);
}

inline Col mat_color(const gbl_material& m) { return Col(m.color[0], m.color[1], m.color[2]); }
inline Col mat_color2(const gbl_material& m) { return Col(m.color2[0], m.color2[1], m.color2[2]); }

Col blinn_bsdf(const gbl_material& m, V3 n, V3 wo, V3 wi, int type) {   // :540-570
    type = sample_type(wo, wi, n, type);
    if (!match_type(type, BSDF_GLOSSY | BSDF_REFLECTION)) return BLACK;
    float cosi = absdot(n, wi), coso = absdot(n, wo);
    if (cosi == 0.0f || coso == 0.0f) return BLACK;
    V3 wh = normalize(wo + wi);
    float cosh = absdot(n, wh);
    float e = m.exponent;
    float D = (e + 2.0f) * INV_TWOPI * std::pow(cosh, e);
    float wo_wh = absdot(wo, wh);
    float G = std::min(1.0f, std::min(2.0f * cosh * coso / wo_wh, 2.0f * cosh * cosi / wo_wh));
    float F = m.k > 0.0f ? fresnel_conductor(wo_wh, m.index, m.k) : fresnel_dielectric(wo_wh, 1.0f, m.index);
    return mat_color(m) * D * G * F / (4.0f * cosi * coso);
}
float blinn_pdf(const gbl_material& m, V3 n, V3 wo, V3 wi) {   // :629-644
    if (!same_hemisphere(n, wo, wi)) return 0.0f;
    V3 wh = normalize(wo + wi);
    float cos_h = absdot(wh, n);
    float e = m.exponent;
    return (e + 1.0f) * std::pow(cos_h, e) / (TWO_PI * 4.0f * dot(wo, wh));
}

// material->bsdf(fragment, wo, wi) with the defaults BSDFAll / BSDFRadiance
Col mat_bsdf(const gbl_material& m, V3 n, V3 wo, V3 wi) {
    switch (m.type) {
        case GBL_MAT_LAMBERT: {   // :437-446
            Col f(BLACK);
            int type = sample_type(wo, wi, n, BSDF_ALL);
            if (match_type(type, BSDF_DIFFUSE | BSDF_REFLECTION)) f += mat_color(m) * INV_PI;
            return f;
        }
        case GBL_MAT_BLINN: return blinn_bsdf(m, n, wo, wi, BSDF_ALL);
        default: return BLACK;   // transparent / mirror: GoblinMaterial.h:306-309,346-350
    }
}
float mat_pdf(const gbl_material& m, V3 n, V3 wo, V3 wi) {
    switch (m.type) {
        case GBL_MAT_LAMBERT: return same_hemisphere(n, wo, wi) ? absdot(n, wi) * INV_PI : 0.0f;   // :472-480
        case GBL_MAT_BLINN: return blinn_pdf(m, n, wo, wi);
        default: return 0.0f;
    }
}

// material->sampleBSDF(fragment, wo, bs, &wi, &pdf, BSDFAll, &sampledType)
Col mat_sample(const gbl_material& m, const Frag& frag, V3 wo, float u_comp, float u1, float u2, V3* wi, float* pdf, int* sampled) {
    V3 n = frag.n;
    switch (m.type) {
        case GBL_MAT_LAMBERT: {   // :448-470
            V3 local = cosine_sample_hemisphere(u1, u2);
            if (dot(wo, n) < 0.0f) local = local * -1.0f;
            *wi = shade_to_world(shade_frame(frag), local);
            *pdf = mat_pdf(m, n, wo, *wi);
            *sampled = BSDF_DIFFUSE | BSDF_REFLECTION;
            return mat_color(m) * INV_PI;
        }
        case GBL_MAT_BLINN: {   // :594-623
            float e = m.exponent;
            float cos_t = std::pow(u1, 1.0f / (e + 1.0f));
            float sin_t = sqrtf(std::max(0.0f, 1.0f - cos_t * cos_t));
            float phi = u2 * TWO_PI;
            V3 wh_local(sin_t * std::cos(phi), sin_t * std::sin(phi), cos_t);
            if (dot(wo, n) < 0.0f) wh_local = wh_local * -1.0f;
            V3 wh = shade_to_world(shade_frame(frag), wh_local);
            *wi = -wo + 2.0f * dot(wo, wh) * wh;
            *pdf = blinn_pdf(m, n, wo, *wi);
            *sampled = BSDF_GLOSSY | BSDF_REFLECTION;
            return blinn_bsdf(m, n, wo, *wi, BSDF_GLOSSY | BSDF_REFLECTION);
        }
        case GBL_MAT_TRANSPARENT: {   // :647-706 with type == BSDFAll -> nMatch == 2
            V3 w_refl(0, 0, 0), w_refr(0, 0, 0);
            float reflect = specular_reflect_dielectric(n, wo, &w_refl, 1.0f, m.index);
            float refract = specular_refract(n, wo, &w_refr, 1.0f, m.index);
            float fresnel = reflect * absdot(w_refl, n);
            float chance = fresnel;
            if (u_comp < chance) {
                *wi = w_refl;
                *sampled = BSDF_SPECULAR | BSDF_REFLECTION;
                *pdf = chance;
                return mat_color(m) * reflect;
            }
            *wi = w_refr;
            *sampled = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1.0f - chance;
            return mat_color2(m) * refract;
        }
        case GBL_MAT_SUBSURFACE: {   // :728-745 with type == BSDFAll == getType(): the Fresnel mirror lobe, pdf 1
            *wi = V3(0, 0, 0);
            Col f = Col(m.color3[0], m.color3[1], m.color3[2]) * specular_reflect_dielectric(n, wo, wi, 1.0f, m.index);
            *pdf = 1.0f;
            *sampled = BSDF_ALL;
            return f;
        }
        default: {   // mirror, :709-726
            *wi = V3(0, 0, 0);
            Col f = mat_color(m) * specular_reflect_conductor(n, wo, wi, m.index, m.k);
            *pdf = 1.0f;
            *sampled = BSDF_SPECULAR | BSDF_REFLECTION;
            return f;
        }
    }
}

// ---------------------------------------------------------------------------
// Lights (GoblinLight.cpp)
// ---------------------------------------------------------------------------
// MaskMaterial::bsdf / pdf / sampleBSDF with type == BSDFAll (GoblinMaterial.cpp:747-811)
Col rmat_bsdf(const ResolvedMat& r, V3 n, V3 wo, V3 wi) {
    if (!r.is_mask) return mat_bsdf(r.m, n, wo, wi);
    return r.alpha * mat_bsdf(r.m, n, wo, wi);
}
float rmat_pdf(const ResolvedMat& r, V3 n, V3 wo, V3 wi) {
    if (!r.is_mask) return mat_pdf(r.m, n, wo, wi);
    return r.alpha * mat_pdf(r.m, n, wo, wi);
}
Col rmat_sample(const ResolvedMat& r, const Frag& frag, V3 wo, float u_comp, float u1, float u2, V3* wi, float* pdf, int* sampled) {
    if (!r.is_mask) return mat_sample(r.m, frag, wo, u_comp, u1, u2, wi, pdf, sampled);
    float masked_prob = r.alpha;
    if (u_comp < masked_prob) {
        Col result = r.alpha * mat_sample(r.m, frag, wo, u_comp, u1, u2, wi, pdf, sampled);
        *pdf *= masked_prob;
        return result;
    }
    Col result = (1.0f - r.alpha) * r.tcolor;
    *wi = -normalize(wo);
    *pdf = 1.0f - masked_prob;
    *sampled = BSDF_NULL;
    return result;
}

float spot_falloff(const Light& l, V3 w) {   // :277-287
    float cos_t = dot(w, l.spot_axis);
    if (cos_t < l.cos_max) return 0.0f;
    if (cos_t > l.cos_falloff) return 1.0f;
    float d = (cos_t - l.cos_max) / (l.cos_falloff - l.cos_max);
    return d * d * d * d;
}

// Geometry::pdf for one light triangle in light-local space (GoblinGeometry.cpp:44-62)
float tri_pdf(const Mesh& m, uint32_t t, float area, V3 p, V3 wi) {
    Ray ray;
    ray.o = p;
    ray.d = wi;
    ray.mint = 1e-3f;
    ray.maxt = INF;
    float tt, b1, b2;
    if (!tri_test(m, t, ray, &tt, &b1, &b2)) return 0.0f;
    Frag f;
    f.n = V3(0, 0, 0);
    f.dpdv = V3(0, 0, 0);
    tri_fragment(m, t, ray, tt, b1, b2, &f);
    float pdf = sqlen(p - f.p) / (area * absdot(-wi, f.n));
    if (std::isinf(pdf)) pdf = 0.0f;
    return pdf;
}
// Geometry::pdf for an analytic shape (GoblinGeometry.cpp:44-62)
float shape_area_pdf(const Mesh& m, float area, V3 p, V3 wi) {
    Ray ray;
    ray.o = p;
    ray.d = wi;
    ray.mint = 1e-3f;
    ray.maxt = INF;
    float tt;
    if (!shape_test(m, ray, &tt)) return 0.0f;
    Frag f;
    shape_fragment(m, ray, tt, &f);
    float pdf = sqlen(p - f.p) / (area * absdot(-wi, f.n));
    if (std::isinf(pdf)) pdf = 0.0f;
    return pdf;
}
// Sphere::pdf, GoblinSphere.cpp:126-138
float sphere_pdf(const Mesh& m, float area, V3 p, V3 wi) {
    float d2 = sqlen(p);
    float r2 = m.radius * m.radius;
    if (d2 - r2 < 1e-4f) return shape_area_pdf(m, area, p, wi);
    float sin_max2 = r2 / d2;
    float cos_max = std::sqrt(std::max(0.0f, 1.0f - sin_max2));
    return 1.0f / (TWO_PI * (1.0f - cos_max));   // uniformConePdf, GoblinSampler.h:168-170
}
// uniformSampleSphere, GoblinSampler.cpp:489-496
inline V3 uniform_sample_sphere(float u1, float u2) {
    float z = 1.0f - 2.0f * u1;
    float sin_t = sqrtf(std::max(0.0f, 1.0f - z * z));
    float phi = TWO_PI * u2;
    return V3(sin_t * std::cos(phi), sin_t * std::sin(phi), z);
}
// Sphere::sample(p, u1, u2, &n) (:115-124 uniform, :117-... cone) / Disk::sample (GoblinDisk.cpp:76-80)
V3 shape_sample(const Mesh& m, V3 p, float u1, float u2, V3* normal) {
    if (m.shape == GBL_SHAPE_DISK) {
        *normal = V3(0.0f, 0.0f, 1.0f);
        float x, y;
        uniform_sample_disk(u1, u2, &x, &y);
        return V3(m.radius * x, m.radius * y, 0.0f);
    }
    float r2 = m.radius * m.radius;
    float d2 = sqlen(p);
    if (d2 - r2 < 1e-4f) {
        *normal = uniform_sample_sphere(u1, u2);
        return m.radius * (*normal);
    }
    V3 z_axis = normalize(-p);
    V3 x_axis, y_axis;
    coordinate_axes(z_axis, &x_axis, &y_axis);
    float sin_max2 = r2 / d2;
    float cos_max = std::sqrt(std::max(0.0f, 1.0f - sin_max2));
    // uniformSampleCone(u1, u2, cosThetaMax, x, y, z), GoblinSampler.cpp:459-467
    float cos_t = 1.0f - u1 + u1 * cos_max;
    float sin_t = sqrtf(std::max(0.0f, 1.0f - cos_t * cos_t));
    float phi = TWO_PI * u2;
    Ray ray;
    ray.o = p;
    ray.d = x_axis * sin_t * std::cos(phi) + y_axis * sin_t * std::sin(phi) + z_axis * cos_t;
    ray.mint = 1e-3f;
    ray.maxt = INF;
    V3 p_hit;
    float tt;
    if (sphere_test(m.radius, ray, &tt)) {
        p_hit = ray.o + tt * ray.d;
    } else {
        p_hit = ray.o + (std::sqrt(d2) * cos_max) * ray.d;   // ray scratches over the sphere's surface
    }
    *normal = normalize(p_hit);
    return p_hit;
}
float geoset_pdf(const orc_scene* s, const Light& l, V3 p, V3 wi) {   // GeometrySet::pdf, :336-343
    const Mesh& m = s->meshes[l.mesh];
    float pdf = 0.0f;
    if (m.shape != GBL_SHAPE_MESH) {
        float a = l.geo.area[0];
        pdf += a * (m.shape == GBL_SHAPE_SPHERE ? sphere_pdf(m, a, p, wi) : shape_area_pdf(m, a, p, wi));
        pdf /= l.geo.sum_area;
        return pdf;
    }
    for (uint32_t t = 0; t < m.ntris; ++t) pdf += l.geo.area[t] * tri_pdf(m, t, l.geo.area[t], p, wi);
    pdf /= l.geo.sum_area;
    return pdf;
}

// light->sampleL(p, epsilon, ls, &wi, &pdf, &shadowRay)
Col light_sample(const orc_scene* s, int li, V3 p, float epsilon, float u_comp, float u1, float u2, V3* wi, float* pdf, Ray* shadow) {
    const Light& l = s->lights[li];
    if (l.type == GBL_LIGHT_AREA) {   // AreaLight::sampleL, :373-394
        const Mesh& m = s->meshes[l.mesh];
        V3 p_local = l.xf.invert_point(p);
        int tri = s->light_geo_cdf[li].sample_discrete(u_comp, nullptr);   // GeometrySet::sample, :313-323
        V3 ns_local, ps_local;
        if (m.shape != GBL_SHAPE_MESH) {
            ps_local = shape_sample(m, p_local, u1, u2, &ns_local);
        } else {
        // Triangle::sample, GoblinTriangle.cpp:165-177 ; uniformSampleTriangle GoblinSampler.cpp:420-424
        float root = sqrtf(u1);
        float b0 = 1.0f - root, b1 = root * u2;
        V3 p0 = m.P(m.idx[3 * tri]), p1 = m.P(m.idx[3 * tri + 1]), p2 = m.P(m.idx[3 * tri + 2]);
        ns_local = normalize(cross(p1 - p0, p2 - p0));
        ps_local = b0 * p0 + b1 * p1 + (1.0f - b0 - b1) * p2;
        }
        V3 wi_local = normalize(ps_local - p_local);
        *pdf = geoset_pdf(s, l, p_local, wi_local);
        V3 ps = l.xf.on_point(ps_local);
        V3 ns = normalize(l.xf.on_normal(ns_local));
        *wi = normalize(ps - p);
        shadow->o = p;
        shadow->d = *wi;
        shadow->mint = epsilon;
        shadow->maxt = length(ps - p) - epsilon;
        return dot(ns, -*wi) > 0.0f ? l.color : BLACK;   // AreaLight::L, :368-371
    }
    if (l.type == GBL_LIGHT_IBL) {   // ImageBasedLight::sampleL, :529-555
        float pdf_row, pdf_col;
        int row;
        float v = s->ibl_marginal[li].sample_continuous(u2, &pdf_row, &row);   // CDF2D::sampleContinuous, GoblinSampler.cpp:378-390
        float u = s->ibl_rows[li][row].sample_continuous(u1, &pdf_col, nullptr);
        float pdf_st = pdf_row * pdf_col;
        float theta = v * PI, phi = u * TWO_PI;
        float cos_theta = std::cos(theta), sin_theta = std::sin(theta), cos_phi = std::cos(phi), sin_phi = std::sin(phi);
        *wi = l.xf.on_vector(V3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
        *pdf = pdf_st / (TWO_PI * PI * sin_theta);   // (the sinTheta == 0 guard before it is overwritten)
        shadow->o = p;
        shadow->d = *wi;
        shadow->mint = epsilon;
        shadow->maxt = INF;
        return mip_level(s, s->images[l.image], 0, u, v, GBL_ADDRESS_REPEAT);
    }
    if (l.type == GBL_LIGHT_DIRECTIONAL) {   // DirectionalLight::sampleL, :145-154 (maxt stays the Ray default)
        *wi = -l.spot_axis;
        *pdf = 1.0f;
        shadow->o = p;
        shadow->d = *wi;
        shadow->mint = epsilon;
        shadow->maxt = INF;
        return l.color;
    }
    // PointLight / SpotLight::sampleL, :87-99 / :225-237
    V3 dir = l.pos - p;
    *wi = normalize(dir);
    *pdf = 1.0f;
    shadow->o = p;
    shadow->d = *wi;
    shadow->mint = epsilon;
    float d2 = sqlen(dir);
    shadow->maxt = std::sqrt(d2) - epsilon;
    if (l.type == GBL_LIGHT_SPOT) return spot_falloff(l, -(*wi)) * l.color / d2;
    return l.color / d2;
}
inline float spherical_theta(V3 v) { return std::acos(std::min(std::max(v.z, -1.0f), 1.0f)); }   // GoblinUtils.h:142-149
inline float spherical_phi(V3 v) {
    float phi = std::atan2(v.y, v.x);
    return phi < 0.0f ? phi + TWO_PI : phi;
}
float light_pdf(const orc_scene* s, int li, V3 p, V3 wi) {   // Light::pdf default 0 ; AreaLight::pdf :457-461
    const Light& l = s->lights[li];
    if (l.type == GBL_LIGHT_IBL) {   // ImageBasedLight::pdf, :615-628 ; CDF2D::pdf, GoblinSampler.cpp:392-405
        V3 w = l.xf.invert_vector(wi);
        float theta = spherical_theta(w);
        float sin_theta = std::sin(theta);
        if (sin_theta == 0.0f) return 0.0f;
        float phi = spherical_phi(w);
        float u = phi * INV_TWOPI, v = theta * INV_PI;
        const Cdf& mg = s->ibl_marginal[li];
        int rows = static_cast<int>(mg.f.size());
        int row = std::min(std::max(floor_int(rows * v), 0), rows - 1);
        const Cdf& rw = s->ibl_rows[li][row];
        int cols = static_cast<int>(rw.f.size());
        int col = std::min(std::max(floor_int(cols * u), 0), cols - 1);
        float integral = mg.integral * rw.integral;
        if (integral == 0.0f) return 0.0f;
        float pdf = mg.f[row] * rw.f[col] / integral;
        return pdf / (TWO_PI * PI * sin_theta);
    }
    if (l.type != GBL_LIGHT_AREA) return 0.0f;
    return geoset_pdf(s, l, l.xf.invert_point(p), l.xf.invert_vector(wi));
}
inline bool light_is_delta(const Light& l) { return l.type != GBL_LIGHT_AREA && l.type != GBL_LIGHT_IBL; }   // GoblinLight.h:116, :296, :346
// light->Le(ray): Black but for an image based light (GoblinLight.h:76, GoblinLight.cpp:520-527)
Col light_le_escaped(const orc_scene* s, int li, V3 dir) {
    const Light& l = s->lights[li];
    if (l.type != GBL_LIGHT_IBL) return BLACK;
    V3 w = l.xf.invert_vector(dir);
    float theta = spherical_theta(w), phi = spherical_phi(w);
    return mip_level(s, s->images[l.image], 0, phi * INV_TWOPI, theta * INV_PI, GBL_ADDRESS_REPEAT);
}
// Scene::evalEnvironmentLight, GoblinScene.cpp:89-95
Col environment_le(const orc_scene* s, V3 dir) {
    Col L(0.0f);
    for (size_t i = 0; i < s->lights.size(); ++i) L += light_le_escaped(s, static_cast<int>(i), dir);
    return L;
}

// Intersection::Le, GoblinPrimitive.cpp:8-14
Col hit_Le(const orc_scene* s, const Hit& h, V3 out_dir) {
    int al = s->instances[h.instance].area_light;
    if (al < 0) return BLACK;
    return dot(h.frag.n, out_dir) > 0.0f ? s->lights[al].color : BLACK;
}

// ---------------------------------------------------------------------------
// Sample record layout: {imageX, imageY, lensU1, lensU2, u1D[0].., u2D[0]..}
// (Sample::allocateQuota, GoblinSampler.cpp:35-58).
// ---------------------------------------------------------------------------
struct Quota {
    std::vector<uint32_t> n1, n2;
    std::vector<uint32_t> off1, off2;   // float offsets into the record (after the 4 header floats)
    uint32_t size = 0;
    // native sampler only: 1D / 2D patterns from these indices on get their strata shuffled per camera sample
    // (the path tracer's BSSRDF block, whose n > 1 patterns are consumed slot by slot together)
    uint32_t perm1_from = 0xffffffffu, perm2_from = 0xffffffffu;
    uint32_t one_d(uint32_t n) {   // SampleQuota::requestOneDQuota, :23-27
        n1.push_back(round_to_square(n));
        return static_cast<uint32_t>(n1.size() - 1);
    }
    uint32_t two_d(uint32_t n) {   // requestTwoDQuota, :29-33
        n2.push_back(round_to_square(round_to_square(n)));
        return static_cast<uint32_t>(n2.size() - 1);
    }
    void finish() {
        uint32_t o = 4;
        off1.clear();
        off2.clear();
        for (uint32_t n : n1) {
            off1.push_back(o);
            o += n;
        }
        for (uint32_t n : n2) {
            off2.push_back(o);
            o += 2 * n;
        }
        size = o - 4;
    }
    uint32_t dims() const { return size + 4; }
};

struct PtIndices {   // PathTracer::querySampleQuota, GoblinPathtracer.cpp:181-208
    std::vector<uint32_t> light1, light2, bsdf1, bsdf2, pick;
    // BSSRDFSampleIndex (GoblinLight.cpp:35-43): pattern indices and samplesNum
    uint32_t sss_ls1 = 0, sss_ls2 = 0, sss_pick = 0, sss_axis = 0, sss_disc = 0, sss_single = 0, sss_n = 0;
};

Quota pt_quota(const gbl_render_setting& rs, PtIndices* ix) {
    Quota q;
    int bounces = std::max(1, rs.max_ray_depth);
    for (int i = 0; i < bounces; ++i) {
        uint32_t l1 = q.one_d(1), l2 = q.two_d(1);   // LightSampleIndex, GoblinLight.cpp:11-20
        uint32_t b1 = q.one_d(1), b2 = q.two_d(1);   // BSDFSampleIndex, GoblinMaterial.cpp:15-24
        uint32_t pk = q.one_d(1);
        if (ix) {
            ix->light1.push_back(l1); ix->light2.push_back(l2);
            ix->bsdf1.push_back(b1); ix->bsdf2.push_back(b2);
            ix->pick.push_back(pk);
        }
    }
    q.perm1_from = static_cast<uint32_t>(q.n1.size());
    q.perm2_from = static_cast<uint32_t>(q.n2.size());
    // BSSRDFSampleIndex, GoblinLight.cpp:35-43
    int n = rs.bssrdf_sample_num;
    uint32_t s_ls1 = q.one_d(n), s_ls2 = q.two_d(n);   // lsIndex
    uint32_t s_pick = q.one_d(n);                      // pickLight
    uint32_t s_axis = q.one_d(n);                      // pickAxis
    uint32_t s_disc = q.two_d(n);                      // disc
    uint32_t s_single = q.one_d(n);                    // singleScatter
    if (ix) {
        ix->sss_ls1 = s_ls1; ix->sss_ls2 = s_ls2; ix->sss_pick = s_pick; ix->sss_axis = s_axis; ix->sss_disc = s_disc;
        ix->sss_single = s_single;
        ix->sss_n = std::min(q.n1[s_ls1], q.n2[s_ls2]);   // LightSampleIndex::samplesNum, GoblinLight.cpp:16
    }
    q.finish();
    return q;
}

// WhittedRenderer::querySampleQuota, GoblinWhitted.cpp:46-70: per light a LightSampleIndex and a BSDFSampleIndex of
// getSamplesNum() points, one pick-light 1D (requested, never read by Li), the BSSRDF block
Quota whitted_quota(const orc_scene* s, const gbl_render_setting& rs, PtIndices* ix) {
    Quota q;
    for (size_t i = 0; i < s->lights.size(); ++i) {
        uint32_t n = s->light_samples[i];
        uint32_t l1 = q.one_d(n), l2 = q.two_d(n);
        uint32_t b1 = q.one_d(n), b2 = q.two_d(n);
        if (ix) {
            ix->light1.push_back(l1); ix->light2.push_back(l2);
            ix->bsdf1.push_back(b1); ix->bsdf2.push_back(b2);
        }
    }
    uint32_t pk = q.one_d(1);
    if (ix) ix->pick.push_back(pk);
    q.perm1_from = 0;
    q.perm2_from = 0;
    int n = rs.bssrdf_sample_num;
    uint32_t s_ls1 = q.one_d(n), s_ls2 = q.two_d(n);
    uint32_t s_pick = q.one_d(n);
    uint32_t s_axis = q.one_d(n);
    uint32_t s_disc = q.two_d(n);
    uint32_t s_single = q.one_d(n);
    if (ix) {
        ix->sss_ls1 = s_ls1; ix->sss_ls2 = s_ls2; ix->sss_pick = s_pick; ix->sss_axis = s_axis; ix->sss_disc = s_disc;
        ix->sss_single = s_single;
        ix->sss_n = std::min(q.n1[s_ls1], q.n2[s_ls2]);
    }
    q.finish();
    return q;
}

Quota ao_quota(const gbl_render_setting& rs) {   // AORenderer::querySampleQuota, GoblinAO.cpp:39-42
    Quota q;
    q.two_d(rs.ao_sample_num);
    q.finish();
    return q;
}

// ---------------------------------------------------------------------------
// RNG (GoblinUtils.cpp:13-56): mt19937 seeded with the next value of the
// process-global, never-seeded libc rand(); floats via
// uniform_real_distribution<float>(0,1), uints via uniform_int_distribution.
// glibc's rand() is the TYPE_3 additive-feedback generator r[i] = r[i-3] +
// r[i-31] seeded with 1; restated here so the oracle does not depend on (or
// disturb) the process-global state.  tests/ check it against libc itself.
// ---------------------------------------------------------------------------
struct GlibcRand {
    int32_t r[34];
    std::vector<uint32_t> st;
    size_t k;
    explicit GlibcRand(uint32_t seed = 1) {
        r[0] = static_cast<int32_t>(seed);
        for (int i = 1; i < 31; ++i) {
            int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
            int64_t w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            r[i] = static_cast<int32_t>(w);
        }
        st.resize(344);
        for (int i = 0; i < 31; ++i) st[i] = static_cast<uint32_t>(r[i]);
        for (int i = 31; i < 34; ++i) st[i] = st[i - 31];
        for (int i = 34; i < 344; ++i) st[i] = st[i - 31] + st[i - 3];
        k = 344;
    }
    int next() {
        uint32_t v = st[k - 31] + st[k - 3];
        st.push_back(v);
        ++k;
        return static_cast<int>(v >> 1);
    }
};

struct Rng {
    std::mt19937 engine;
    std::uniform_real_distribution<float> real{0.0f, 1.0f};
    std::uniform_int_distribution<uint32_t> uint{0, std::numeric_limits<uint32_t>::max()};
    explicit Rng(uint32_t seed) : engine(seed) {}
    float f() { return real(engine); }
    uint32_t u() { return uint(engine); }
};

template <class T>
void shuffle(T* buf, uint32_t num, uint32_t dim, Rng* rng) {   // GoblinSampler.h:149-157
    for (uint32_t n = 0; n < num; ++n) {
        size_t other = rng->u() % num;
        for (uint32_t d = 0; d < dim; ++d) std::swap(buf[n * dim + d], buf[other * dim + d]);
    }
}

// Sampler (GoblinSampler.cpp:60-307) over one SampleRange, emitting flattened records.
struct Sampler {
    int x0, x1, y0, y1, cx, cy, spp, root;
    const Quota& q;
    Rng* rng;
    std::vector<float> buf;
    Sampler(int xs, int xe, int ys, int ye, int sample_per_pixel, const Quota& quota, Rng* r)
        : x0(xs), x1(xe), y0(ys), y1(ye), cx(xs), cy(ys), q(quota), rng(r) {
        spp = round_to_square(sample_per_pixel, &root);
    }
    void strat1(float* b, uint32_t n) {   // stratifiedUniform1D, :276-286
        float strata = 1.0f / static_cast<float>(n);
        float sub = strata / spp;
        for (uint32_t i = 0; i < n; ++i)
            for (int j = 0; j < spp; ++j) {
                float off = j + rng->f();
                b[i * spp + j] = i * strata + off * sub;
            }
    }
    void strat2(float* b, uint32_t n) {   // stratifiedUniform2D, :288-307
        int r = static_cast<int>(sqrtf(static_cast<float>(n)));
        float strata = 1.0f / r;
        float sub = strata / root;
        for (uint32_t k = 0; k < n; ++k) {
            int ux = k % r, uy = k / r;
            for (int p = 0; p < spp; ++p) {
                int px = p % root, py = p / root;
                float xo = px + rng->f();
                float yo = py + rng->f();
                int index = 2 * (k * spp + py * root + px);
                b[index] = ux * strata + xo * sub;
                b[index + 1] = uy * strata + yo * sub;
            }
        }
    }
    // requestSamples, :108-197.  out: spp records of q.dims() floats.
    int request(float* out) {
        if (cy == y1) return 0;
        if (buf.empty()) buf.resize(static_cast<size_t>(spp) * (4 + q.size));
        uint32_t dims = q.dims();
        float* image = buf.data();
        float* lens = image + 2 * spp;
        float* quota = image + 4 * spp;
        strat2(image, 1);
        strat2(lens, 1);
        float* cur = quota;
        for (uint32_t n : q.n1) {
            strat1(cur, n);
            cur += n * spp;
        }
        for (uint32_t n : q.n2) {
            strat2(cur, n);
            cur += 2 * n * spp;
        }
        shuffle(lens, spp, 2, rng);
        float* sh = quota;
        for (uint32_t n : q.n1)
            for (uint32_t j = 0; j < n; ++j) {
                shuffle(sh, spp, 1, rng);
                sh += spp;
            }
        for (uint32_t n : q.n2)
            for (uint32_t j = 0; j < n; ++j) {
                shuffle(sh, spp, 2, rng);
                sh += 2 * spp;
            }
        for (int i = 0; i < spp; ++i) {
            float* rec = out + static_cast<size_t>(i) * dims;
            rec[0] = cx + image[2 * i];
            rec[1] = cy + image[2 * i + 1];
            rec[2] = lens[2 * i];
            rec[3] = lens[2 * i + 1];
        }
        float* fill = quota;
        for (size_t i = 0; i < q.n1.size(); ++i)
            for (uint32_t j = 0; j < q.n1[i]; ++j) {
                for (int k = 0; k < spp; ++k) out[static_cast<size_t>(k) * dims + q.off1[i] + j] = fill[k];
                fill += spp;
            }
        for (size_t i = 0; i < q.n2.size(); ++i)
            for (uint32_t j = 0; j < q.n2[i]; ++j) {
                for (int k = 0; k < spp; ++k) {
                    float* rec = out + static_cast<size_t>(k) * dims + q.off2[i];
                    rec[2 * j] = fill[2 * k];
                    rec[2 * j + 1] = fill[2 * k + 1];
                }
                fill += 2 * spp;
            }
        for (int i = 0; i < spp; ++i) {
            float* rec = out + static_cast<size_t>(i) * dims;
            for (size_t j = 0; j < q.n1.size(); ++j) shuffle(rec + q.off1[j], q.n1[j], 1, rng);
            for (size_t j = 0; j < q.n2.size(); ++j) shuffle(rec + q.off2[j], q.n2[j], 2, rng);
        }
        if (++cx == x1) {
            cx = x0;
            cy++;
        }
        return spp;
    }
};

// ---------------------------------------------------------------------------
// Integrators
// ---------------------------------------------------------------------------
struct LiCtx {
    const orc_scene* s;
    const gbl_render_setting* rs;
    const Quota* q;
    const PtIndices* ix;
    Rng* rng;          // null in replay: the three discarded draws are skipped
    int ref_faithful;  // also run the reference's redundant traversals
    Counters cnt;
    uint64_t dims_used = 0;
    float primary_maxt = INF;   // the camera ray's maxt after Li: Ray::maxt is mutable and the first scene query clips it
    // Russian roulette, the device's build-side extension (the reference's loop is fixed length, GoblinPathtracer.cpp:76):
    // restated here only so that the device's kill draws and 1 / q factors can be checked sample by sample
    // (goblin_amd/csrc/kernels/render_kernels.h, wavefront.h).  Native sampler only: the draw is keyed by the pixel and sample.
    bool rr = false;
    uint32_t rr_pixel_key = 0, rr_k = 0;
};
inline uint32_t nat_mix(uint32_t a, uint32_t b);
inline float nat_u01(uint32_t h);
// From the third bounce on a path survives with probability q = min(0.95, max component of the throughput its sampled
// direction would carry on); a survivor's bsdf value f carries 1 / q into everything that depends on the extension ray (the
// BSDF-sampled light term, the environment term of an escaping ray, the next throughput), the vertex's light-sampled term is
// collected either way.  Returns false when the path is killed.
inline bool russian_roulette(const LiCtx* c, int bounce, Col throughput, Col* f, float cosw, float bsdf_pdf) {
    if (!c->rr || bounce < 2) return true;
    const Col tn = throughput * ((*f) * cosw / bsdf_pdf);
    const float q = fminf(0.95f, fmaxf(tn.r, fmaxf(tn.g, tn.b)));
    const float u = nat_u01(nat_mix(nat_mix(c->rr_pixel_key, 0xBADC0DEu + static_cast<uint32_t>(bounce)), c->rr_k));
    if (!(u < q)) return false;
    *f = (*f) * (1.0f / q);
    return true;
}

// BSDFSample(rng), GoblinMaterial.cpp:26-30 -- three draws whose values never
// matter in a mask-free scene but which advance the tile's stream.
inline void draw_bsdf_sample(LiCtx* c) {
    if (c->rng) {
        c->rng->f();
        c->rng->f();
        c->rng->f();
    }
}

// PathTracer::evalAttenuation, GoblinPathtracer.cpp:21-48: walks the BSDFnullptr (mask) surfaces along the ray;
// identity without them (then only ref_faithful pays for the empty filtered traversal).
inline Col eval_attenuation(LiCtx* c, const Ray& ray) {
    const orc_scene* s = c->s;
    if (!s->has_masks) {
        if (c->ref_faithful) scene_filtered_traversal(s, ray, &c->cnt);
        return Col(1.0f);
    }
    Col throughput(1.0f);
    float maxt = ray.maxt;
    Ray cur = ray;
    Hit hit;
    hit.frag.n = V3(0, 0, 0);
    hit.frag.dpdv = V3(0, 0, 0);
    while (true) {
        if (!scene_intersect(s, cur, &hit, &c->cnt, FILTER_MASK)) break;
        compute_uv_differential(&hit.frag, nullptr);
        ResolvedMat rm = resolve_hit_material(s, s->instances[hit.instance].material, hit.frag);
        // sampleBSDF(..., BSDFnullptr): sampleAlpha only -> (1 - alpha) * transparentColor (:784-791).  A subsurface
        // material does not match the request (matchType(BSDFnullptr, BSDFAll) fails, :733-736) and returns Black.
        if (rm.m.type == GBL_MAT_SUBSURFACE && !rm.is_mask) throughput *= BLACK;
        else throughput *= (1.0f - rm.alpha) * rm.tcolor;
        if (throughput == BLACK) break;
        cur.mint = cur.maxt + hit.epsilon;
        cur.maxt = maxt;
    }
    return throughput;
}


// ---------------------------------------------------------------------------
// BSSRDF (GoblinMaterial.cpp:32-220, GoblinMaterial.h:61-116) and
// Renderer::Lsubsurface (GoblinRenderer.cpp:128-296).  The textures of a
// subsurface material are looked up per fragment: color = absorb (sigma_a),
// color2 = scatterPrime (sigma_s'), index = eta, k = g.
// ---------------------------------------------------------------------------
inline Col operator-(Col a, Col b) { return Col(a.r - b.r, a.g - b.g, a.b - b.b, a.a); }
inline Col operator-(Col a) { return Col(-a.r, -a.g, -a.b, a.a); }
inline Col operator/(Col a, Col b) { return Col(a.r / b.r, a.g / b.g, a.b / b.b, a.a); }
inline Col sqrt_color(Col c) { return Col(std::sqrt(c.r), std::sqrt(c.g), std::sqrt(c.b)); }
inline Col exp_color(Col c) { return Col(std::exp(c.r), std::exp(c.g), std::exp(c.b)); }
inline Col clamp_color(Col c) { return Col(clampf(c.r, 0.0f, INF), clampf(c.g, 0.0f, INF), clampf(c.b, 0.0f, INF)); }

inline float bssrdf_fdr(float eta) {   // BSSRDF::Fdr, GoblinMaterial.h:94-105
    if (eta < 1.0f) return -0.4399f + 0.7099f / eta - 0.3319f / (eta * eta) + 0.0636f / (eta * eta * eta);
    return -1.4399f / (eta * eta) + 0.7099f / eta + 0.6681f + 0.0636f * eta;
}
inline float bssrdf_A(const gbl_material& m) {   // BSSRDF ctor, :35-36
    float fdr = bssrdf_fdr(m.index);
    return (1.0f + fdr) / (1.0f - fdr);
}
// `m` below is the material resolved at the fragment in question
inline Col bssrdf_scatter(const gbl_material& m) { return mat_color2(m) / (1.0f - m.k); }           // getScatter, GoblinMaterial.h:111-114
inline Col bssrdf_attenuation(const gbl_material& m) { return bssrdf_scatter(m) + mat_color(m); }   // getAttenuation, :107-109
inline Col bssrdf_sigma_tr(const gbl_material& m) {                                                 // getSigmaTr, GoblinMaterial.cpp:166-171
    Col sigma_a = mat_color(m), sigma_sp = mat_color2(m);
    Col sigma_tp = sigma_a + sigma_sp;
    return sqrt_color(3.0f * sigma_a * sigma_tp);
}
inline float phase_hg(V3 wi, V3 wo, float g) {   // GoblinVolume.h:126-134
    if (g < 1e-3) return 0.25f * INV_PI;
    float cos_theta = dot(wi, wo);
    return 0.25f * INV_PI * (1.0f - g * g) / powf(1.0f + g * g - 2.0f * g * cos_theta, 1.5f);
}
inline V3 refract_dir(V3 wo, V3 n, float etai, float etat) {   // Goblin::specularRefract(wo, n, etai, etat), GoblinMaterial.cpp:418-434
    float eta = etai / etat;
    float cosi = absdot(n, wo);
    return normalize(n * (eta * cosi - std::sqrt(std::max(0.0f, 1.0f - eta * eta * (1.0f - cosi * cosi)))) - eta * wo);
}
Col bssrdf_rd(const gbl_material& m, float A, float d2) {   // BSSRDF::Rd, :60-81
    Col sigma_a = mat_color(m), sigma_sp = mat_color2(m);
    Col sigma_tp = sigma_a + sigma_sp;
    Col sigma_tr = sqrt_color(3.0f * sigma_a * sigma_tp);
    Col one(1.0f);
    Col zr = one / sigma_tp;
    Col zv = zr * (1.0f + 4.0f / 3.0f * A);
    Col dr = sqrt_color(zr * zr + Col(d2));
    Col dv = sqrt_color(zv * zv + Col(d2));
    Col alpha_p = sigma_sp / sigma_tp;
    Col s_dr = sigma_tr * dr;
    Col s_dv = sigma_tr * dv;
    Col rd = 0.25f * INV_PI * alpha_p * ((zr * (one + s_dr) * exp_color(-s_dr) / (dr * dr * dr)) + (zv * (one + s_dv) * exp_color(-s_dv) / (dv * dv * dv)));
    return clamp_color(rd);
}
inline float gaussian_pdf_2d(float x, float y, float falloff, float rmax) {   // GoblinSampler.h:194-204
    return (INV_PI * falloff * std::exp(-falloff * (x * x + y * y))) / (1.0f - std::exp(-falloff * rmax * rmax));
}
inline float gaussian_pdf_proj(V3 center, V3 sample, V3 N, float falloff, float rmax) {   // GoblinSampler.cpp:645-657
    V3 d = sample - center;
    V3 projected = d - N * dot(d, N);
    return (INV_PI * falloff * std::exp(-falloff * sqlen(projected))) / (1.0f - std::exp(-falloff * rmax * rmax));
}
enum { SSS_U_AXIS = 0, SSS_V_AXIS = 1, SSS_N_AXIS = 2 };
float bssrdf_mis_weight(const Frag& fo, const Frag& fi, int axis, float pdf, float sigma_tr, float rmax) {   // BSSRDF::MISWeight, :83-127
    float weight = 0.0f;
    V3 pwo = fo.p, pwi = fi.p, ni = fi.n;
    if (axis == SSS_N_AXIS) {
        V3 u = normalize(fo.dpdu), v = normalize(fo.dpdv);
        float u_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, u, sigma_tr, rmax) * absdot(u, ni);
        float v_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, v, sigma_tr, rmax) * absdot(v, ni);
        float num = 4 * pdf * pdf;
        weight = num / (num + u_pdf * u_pdf + v_pdf * v_pdf);
    } else if (axis == SSS_U_AXIS) {
        V3 n = fo.n, v = normalize(fo.dpdv);
        float n_pdf = 0.5f * gaussian_pdf_proj(pwo, pwi, n, sigma_tr, rmax) * absdot(n, ni);
        float v_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, v, sigma_tr, rmax) * absdot(v, ni);
        float num = pdf * pdf;
        weight = num / (4 * n_pdf * n_pdf + num + v_pdf * v_pdf);
    } else {
        V3 n = fo.n, u = normalize(fo.dpdu);
        float n_pdf = 0.5f * gaussian_pdf_proj(pwo, pwi, n, sigma_tr, rmax) * absdot(n, ni);
        float u_pdf = 0.25f * gaussian_pdf_proj(pwo, pwi, u, sigma_tr, rmax) * absdot(u, ni);
        float num = pdf * pdf;
        weight = num / (4 * n_pdf * n_pdf + u_pdf * u_pdf + num);
    }
    return weight;
}
int bssrdf_sample_probe_ray(const Frag& frag, float u_axis, float u_disc0, float u_disc1, float sigma_tr, float rmax, Ray* probe,
                            float* pdf) {   // BSSRDF::sampleProbeRay, :129-164
    Frame fr = shade_frame(frag);
    V3 pwo = frag.p;
    // gaussianSample2D(u1, u2, falloff, Rmax), GoblinSampler.cpp:638-643
    float r = sqrtf(std::log(1.0f - u_disc0 * (1.0f - std::exp(-sigma_tr * rmax * rmax))) / -sigma_tr);
    float theta = TWO_PI * u_disc1;
    float sx = r * std::cos(theta), sy = r * std::sin(theta);
    float half_len = std::sqrt(rmax * rmax - (sx * sx + sy * sy));
    int axis;
    if (u_axis <= 0.5f) {
        probe->o = pwo + shade_to_world(fr, V3(sx, sy, -half_len));
        probe->d = frag.n;
        axis = SSS_N_AXIS;
        *pdf = 0.5f;
    } else if (u_axis <= 0.75f) {
        probe->o = pwo + shade_to_world(fr, V3(-half_len, sx, sy));
        probe->d = normalize(frag.dpdu);
        axis = SSS_U_AXIS;
        *pdf = 0.25f;
    } else {
        probe->o = pwo + shade_to_world(fr, V3(sy, -half_len, sx));
        probe->d = normalize(frag.dpdv);
        axis = SSS_V_AXIS;
        *pdf = 0.25f;
    }
    probe->mint = 0.0f;
    probe->maxt = 2.0f * half_len;
    *pdf *= gaussian_pdf_2d(sx, sy, sigma_tr, rmax);
    return axis;
}

struct SssSample {   // BSSRDFSample(sample, index, n), GoblinLight.cpp:53-61
    float ls_comp, ls_geo[2], pick_light, pick_axis, disc[2], single;
};
inline SssSample sss_sample(const LiCtx* c, const float* rec, uint32_t i) {
    const Quota& q = *c->q;
    const PtIndices& ix = *c->ix;
    SssSample s;
    s.ls_comp = rec[q.off1[ix.sss_ls1] + i];
    s.ls_geo[0] = rec[q.off2[ix.sss_ls2] + 2 * i];
    s.ls_geo[1] = rec[q.off2[ix.sss_ls2] + 2 * i + 1];
    s.pick_light = rec[q.off1[ix.sss_pick] + i];
    s.pick_axis = rec[q.off1[ix.sss_axis] + i];
    s.disc[0] = rec[q.off2[ix.sss_disc] + 2 * i];
    s.disc[1] = rec[q.off2[ix.sss_disc] + 2 * i + 1];
    s.single = rec[q.off1[ix.sss_single] + i];
    return s;
}

// Renderer::LbssrdfSingle, GoblinRenderer.cpp:128-204
Col l_bssrdf_single(LiCtx* c, const Frag& frag, uint32_t material, V3 wo, const float* rec) {
    const orc_scene* s = c->s;
    const gbl_material mo = resolve_material(s, s->materials[material], frag);
    V3 pwo = frag.p, no = frag.n;
    float coso = absdot(wo, frag.n);
    float eta = mo.index;
    float Ft = 1.0f - fresnel_dielectric(coso, 1.0f, eta);
    Col scatter = bssrdf_scatter(mo);
    Col sigma_t = bssrdf_attenuation(mo);
    float falloff = luminance(sigma_t);
    V3 wo_refract = refract_dir(wo, no, 1.0f, eta);
    Col Ls(0.0f);
    for (uint32_t i = 0; i < c->ix->sss_n; ++i) {
        const SssSample bs = sss_sample(c, rec, i);
        c->dims_used += 8;
        float d = -std::log(bs.single) / falloff;           // exponentialSample, GoblinSampler.h:219-221
        V3 p_sample = pwo + d * wo_refract;
        float sample_pdf = falloff * std::exp(-falloff * d);   // exponentialPdf, :223-225
        float pick_pdf;
        int light = s->light_power.sample_discrete(bs.pick_light, &pick_pdf);
        V3 wi;
        float light_pdf_v;
        Ray shadow;
        Col L = light_sample(s, light, p_sample, 1e-5f, bs.ls_comp, bs.ls_geo[0], bs.ls_geo[1], &wi, &light_pdf_v, &shadow);
        if (L == BLACK || light_pdf_v == 0.0f) continue;
        float maxt = shadow.maxt;
        Hit wh;
        wh.frag.n = V3(0, 0, 0);
        wh.frag.dpdv = V3(0, 0, 0);
        if (scene_intersect(s, shadow, &wh, &c->cnt)) {
            if (s->instances[wh.instance].material == material) {   // getBSSRDF() == bssrdf: the same material object
                const Frag& fwi = wh.frag;
                V3 pwi = fwi.p, ni = fwi.n;
                shadow.mint = shadow.maxt + wh.epsilon;
                shadow.maxt = maxt;
                if (!scene_occluded(s, shadow, &c->cnt)) {
                    const gbl_material mi = resolve_material(s, s->materials[material], fwi);
                    float ph = phase_hg(wi, wo_refract, mo.k);
                    float cosi = absdot(ni, wi);
                    float Fti = 1.0f - fresnel_dielectric(cosi, 1.0f, eta);
                    Col sigma_ti = bssrdf_attenuation(mi);
                    float G = absdot(ni, wo_refract) / cosi;
                    Col sigma_tc = sigma_t + G * sigma_ti;
                    float di = length(pwi - p_sample);
                    float et = 1.0f / eta;
                    float di_prime = di * absdot(wi, ni) / std::sqrt(1.0f - et * et * (1.0f - cosi * cosi));
                    Ls += (Ft * Fti * ph * scatter / sigma_tc) * exp_color(-di_prime * sigma_ti) * exp_color(-d * sigma_t) * L /
                          (light_pdf_v * pick_pdf * sample_pdf);
                }
            }
        }
    }
    Ls = Ls / static_cast<float>(c->ix->sss_n);
    return Ls;
}

// Renderer::LbssrdfDiffusion, GoblinRenderer.cpp:206-274
Col l_bssrdf_diffusion(LiCtx* c, const Frag& frag, uint32_t material, V3 wo, const float* rec) {
    const orc_scene* s = c->s;
    const gbl_material mo = resolve_material(s, s->materials[material], frag);
    const float A = bssrdf_A(mo);
    V3 pwo = frag.p;
    float coso = absdot(wo, frag.n);
    float eta = mo.index;
    float Ft = 1.0f - fresnel_dielectric(coso, 1.0f, eta);
    float sigma_tr = luminance(bssrdf_sigma_tr(mo));
    float skip_ratio = 0.01f;
    float rmax = std::sqrt(std::log(skip_ratio) / -sigma_tr);
    Col Lm(0.0f);
    for (uint32_t i = 0; i < c->ix->sss_n; ++i) {
        const SssSample bs = sss_sample(c, rec, i);
        Ray probe;
        float disc_pdf;
        int axis = bssrdf_sample_probe_ray(frag, bs.pick_axis, bs.disc[0], bs.disc[1], sigma_tr, rmax, &probe, &disc_pdf);
        Hit ph;
        ph.frag.n = V3(0, 0, 0);
        ph.frag.dpdv = V3(0, 0, 0);
        if (scene_intersect(s, probe, &ph, &c->cnt)) {
            if (s->instances[ph.instance].material == material) {
                const Frag& pf = ph.frag;
                V3 p_probe = pf.p;
                const gbl_material mp = resolve_material(s, s->materials[material], pf);
                Col Rd = bssrdf_rd(mp, A, sqlen(p_probe - pwo));
                float pick_pdf;
                int light = s->light_power.sample_discrete(bs.pick_light, &pick_pdf);
                V3 wi;
                float light_pdf_v;
                Ray shadow;
                V3 ni = pf.n;
                Col L = light_sample(s, light, p_probe, ph.epsilon, bs.ls_comp, bs.ls_geo[0], bs.ls_geo[1], &wi, &light_pdf_v, &shadow);
                if (L == BLACK || light_pdf_v == 0.0f || scene_occluded(s, shadow, &c->cnt)) continue;
                float cosi = absdot(ni, wi);
                Col irradiance = L * cosi / (light_pdf_v * pick_pdf);
                float Fti = 1.0f - fresnel_dielectric(cosi, 1.0f, eta);
                float pdf = disc_pdf * absdot(probe.d, ni);
                float w = bssrdf_mis_weight(frag, pf, axis, pdf, sigma_tr, rmax);
                Lm += (w * INV_PI * Ft * Fti * Rd * irradiance) / pdf;
            }
        }
    }
    Lm = Lm / static_cast<float>(c->ix->sss_n);
    return Lm;
}

// Renderer::Lsubsurface, GoblinRenderer.cpp:276-296
Col l_subsurface(LiCtx* c, const Hit& hit, V3 wo, const float* rec) {
    const orc_scene* s = c->s;
    const uint32_t material = s->instances[hit.instance].material;
    if (s->materials[material].type != GBL_MAT_SUBSURFACE || s->lights.empty()) return Col(0.0f);
    Col single = l_bssrdf_single(c, hit.frag, material, wo, rec);
    Col multi = l_bssrdf_diffusion(c, hit.frag, material, wo, rec);
    return single + multi;
}

// PathTracer::Li, GoblinPathtracer.cpp:50-179
Col path_li(LiCtx* c, const Ray& primary, const float* rec, const RayDiff* primary_diff) {
    const orc_scene* s = c->s;
    if (s->lights.empty()) return Col(0.0f);
    Col Li(0.0f);
    Ray ray = primary;
    Hit hit;
    hit.frag.n = V3(0, 0, 0);
    hit.frag.dpdv = V3(0, 0, 0);
    const bool primary_hit = scene_intersect(s, ray, &hit, &c->cnt);
    c->primary_maxt = ray.maxt;
    if (!primary_hit) {   // get image based lighting if the ray didn't hit anything (:61-65)
        Li += environment_le(s, ray.d);
        return Li;
    }
    Li += hit_Le(s, hit, -ray.d);
    Li += l_subsurface(c, hit, -ray.d, rec);   // :69 -- before computeUVDifferential, so its lookups see zero differentials
    Ray cur = ray;
    Col throughput(1.0f);
    float epsilon = hit.epsilon;
    bool first_bounce = true;   // :75; stays true while the path only punches through masks (`continue` skips :176)
    for (int bounce = 0; bounce < c->rs->max_ray_depth - 1; ++bounce) {
        float ls_comp = rec[c->q->off1[c->ix->light1[bounce]]];
        const float* ls_geo = rec + c->q->off2[c->ix->light2[bounce]];
        float bs_comp = rec[c->q->off1[c->ix->bsdf1[bounce]]];
        const float* bs_dir = rec + c->q->off2[c->ix->bsdf2[bounce]];
        float pick = rec[c->q->off1[c->ix->pick[bounce]]];
        c->dims_used += 7;
        float pick_pdf;
        int light = s->light_power.sample_discrete(pick, &pick_pdf);   // Scene::sampleLight, GoblinScene.cpp:97-104
        Col Ld(0.0f);
        compute_uv_differential(&hit.frag, bounce == 0 ? primary_diff : nullptr);   // :77; only the camera ray has differentials
        const ResolvedMat mat = resolve_hit_material(s, s->instances[hit.instance].material, hit.frag);
        const Frag& frag = hit.frag;
        V3 wo = -cur.d;
        V3 wi;
        V3 p = frag.p, n = frag.n;
        float light_pdf_v, bsdf_pdf;
        Ray shadow;
        Col L = light_sample(s, light, p, epsilon, ls_comp, ls_geo[0], ls_geo[1], &wi, &light_pdf_v, &shadow);
        if (L != BLACK && light_pdf_v > 0.0f) {
            Col f = rmat_bsdf(mat, n, wo, wi);
            if (f != BLACK && !scene_occluded(s, shadow, &c->cnt, FILTER_OPAQUE)) {
                draw_bsdf_sample(c);
                Col tr = eval_attenuation(c, shadow);
                if (light_is_delta(s->lights[light])) {
                    Ld += f * tr * L * absdot(n, wi) / light_pdf_v;
                } else {
                    bsdf_pdf = rmat_pdf(mat, n, wo, wi);
                    float lw = power_heuristic(1, light_pdf_v, 1, bsdf_pdf);
                    Ld += f * tr * L * absdot(n, wi) * lw / light_pdf_v;
                }
            }
        }
        int sampled = 0;
        Col f = rmat_sample(mat, frag, wo, bs_comp, bs_dir[0], bs_dir[1], &wi, &bsdf_pdf, &sampled);
        if (f != BLACK && bsdf_pdf > 0.0f) {
            if (sampled == BSDF_NULL) {
                // stepped on an index-matched (mask) BSDF: punch through, no direct light for this bounce (:122-136)
                throughput *= (f / bsdf_pdf);
                cur.o = p; cur.d = wi; cur.mint = epsilon; cur.maxt = INF;
                Hit nh;
                nh.frag = hit.frag;
                if (!scene_intersect(s, cur, &nh, &c->cnt)) {
                    // "primary ray need to evaluate image based lighting in this case" (:125-131)
                    if (first_bounce) Li += throughput * environment_le(s, cur.d);
                    break;
                }
                hit = nh;
                epsilon = nh.epsilon;
                continue;
            }
            float fw = 1.0f;
            if (!(sampled & BSDF_SPECULAR)) {
                light_pdf_v = light_pdf(s, light, p, wi);
                fw = power_heuristic(1, bsdf_pdf, 1, light_pdf_v);
            }
            if (!russian_roulette(c, bounce, throughput, &f, absdot(wi, n), bsdf_pdf)) {   // (extension; off in every parity mode)
                Li += throughput * Ld / pick_pdf;
                break;
            }
            Ray r;
            r.o = p; r.d = wi; r.mint = epsilon; r.maxt = INF;
            // The reference traces this "MIS ray" and then the identical
            // extension ray below.  Unless asked to be faithful to that cost we
            // trace it once and reuse the hit.
            if (s->has_masks) {
                // With masks the MIS query (isOpaque filter, :148) and the extension query (:170) differ.
                Hit lh;
                lh.frag = hit.frag;
                Ray rr = r;
                bool lhit = scene_intersect(s, rr, &lh, &c->cnt, FILTER_OPAQUE);
                draw_bsdf_sample(c);
                Col tr = eval_attenuation(c, rr);
                if (lhit && s->instances[lh.instance].area_light == light) {
                    Col Le = hit_Le(s, lh, -wi);
                    if (Le != BLACK) Ld += f * tr * Le * absdot(wi, n) * fw / bsdf_pdf;
                } else if (!lhit) {
                    Ld += f * tr * light_le_escaped(s, light, r.d) * fw / bsdf_pdf;   // the radiance contribution from IBL (:157-161)
                }
                Li += throughput * Ld / pick_pdf;
                throughput *= f * absdot(wi, n) / bsdf_pdf;
                cur = r;
                Hit nh;
                nh.frag = hit.frag;
                if (!scene_intersect(s, cur, &nh, &c->cnt)) break;
                hit = nh;
                epsilon = nh.epsilon;
                first_bounce = false;
                continue;
            }
            Hit lh;
            lh.frag = hit.frag;
            Ray rr = r;
            bool lhit = scene_intersect(s, rr, &lh, &c->cnt);
            if (lhit) {
                draw_bsdf_sample(c);
                Col tr = eval_attenuation(c, r);
                if (s->instances[lh.instance].area_light == light) {
                    Col Le = hit_Le(s, lh, -wi);
                    if (Le != BLACK) Ld += f * tr * Le * absdot(wi, n) * fw / bsdf_pdf;
                }
            } else {
                draw_bsdf_sample(c);
                Col tr = eval_attenuation(c, r);
                Ld += f * tr * light_le_escaped(s, light, r.d) * fw / bsdf_pdf;   // the radiance contribution from IBL (:157-161)
            }
            Li += throughput * Ld / pick_pdf;
            throughput *= f * absdot(wi, n) / bsdf_pdf;
            if (c->ref_faithful) {   // the second, identical closest-hit query (GoblinPathtracer.cpp:170)
                Hit again;
                again.frag = hit.frag;
                Ray r2 = r;
                scene_intersect(s, r2, &again, &c->cnt);
            }
            if (!lhit) break;
            cur = r;
            cur.maxt = rr.maxt;
            // The reference's second query starts from the previous bounce's
            // Intersection; only the degenerate-uv branch could tell the
            // difference, and it reads fields both queries overwrite alike.
            hit = lh;
            epsilon = lh.epsilon;
            first_bounce = false;
        } else {
            Li += throughput * Ld / pick_pdf;
            break;
        }
    }
    return Li;
}

// AORenderer::Li, GoblinAO.cpp:12-37
Col ao_li(LiCtx* c, const Ray& primary, const float* rec) {
    const orc_scene* s = c->s;
    Col Li = BLACK;
    Ray ray = primary;
    Hit hit;
    hit.frag.n = V3(0, 0, 0);
    hit.frag.dpdv = V3(0, 0, 0);
    const bool primary_hit = scene_intersect(s, ray, &hit, &c->cnt);
    c->primary_maxt = ray.maxt;
    if (primary_hit) {
        uint32_t n = static_cast<uint32_t>(round_to_square(c->rs->ao_sample_num));   // SampleIndex.sampleNum
        uint32_t occluded = 0;
        const float* u = rec + c->q->off2[0];
        Frame fr = shade_frame(hit.frag);
        for (uint32_t i = 0; i < n; ++i) {
            V3 dir = uniform_sample_hemisphere(u[2 * i], u[2 * i + 1]);
            V3 wdir = shade_to_world(fr, dir);
            Ray occ;
            occ.o = hit.frag.p; occ.d = wdir; occ.mint = hit.epsilon; occ.maxt = INF;
            if (scene_occluded(s, occ, &c->cnt)) ++occluded;
        }
        c->dims_used += 2 * n;
        Li = Col(static_cast<float>(n - occluded) / static_cast<float>(n));
    }
    return Li;
}

// ---------------------------------------------------------------------------
// WhittedRenderer (GoblinWhitted.cpp:13-44) over Renderer::multiSampleLd / estimateLd / specularReflect /
// specularRefract (GoblinRenderer.cpp:474-648).  Mask and subsurface materials are not restated for this integrator.
// ---------------------------------------------------------------------------
// materials whose sampleBSDF does not match a BSDFAll & ~BSDFSpecular request: the specular ones, and SubsurfaceMaterial whose
// type is BSDFAll and only matches that very request (matchType, GoblinMaterial.h:191-193; GoblinMaterial.cpp:732-736)
inline bool mat_is_specular(const gbl_material& m) {
    return m.type == GBL_MAT_TRANSPARENT || m.type == GBL_MAT_MIRROR || m.type == GBL_MAT_SUBSURFACE;
}

// Renderer::estimateLd with type = BSDFAll & ~BSDFSpecular (:502-567)
Col estimate_ld(LiCtx* c, V3 wo, float epsilon, const Hit& hit, const ResolvedMat& mat, int light, float ls_comp, const float* ls_geo,
                float bs_comp, const float* bs_dir) {
    const orc_scene* s = c->s;
    Col Ld(BLACK);
    const Frag& frag = hit.frag;
    V3 wi;
    V3 p = frag.p, n = frag.n;
    float light_pdf_v, bsdf_pdf;
    Ray shadow;
    Col L = light_sample(s, light, p, epsilon, ls_comp, ls_geo[0], ls_geo[1], &wi, &light_pdf_v, &shadow);
    if (L != BLACK && light_pdf_v > 0.0f) {
        Col f = rmat_bsdf(mat, n, wo, wi);
        if (f != BLACK && !scene_occluded(s, shadow, &c->cnt)) {
            if (light_is_delta(s->lights[light])) return f * L * absdot(n, wi) / light_pdf_v;
            bsdf_pdf = rmat_pdf(mat, n, wo, wi);
            float lw = power_heuristic(1, light_pdf_v, 1, bsdf_pdf);
            Ld += f * L * absdot(n, wi) * lw / light_pdf_v;
        }
    }
    // sampleBSDF(..., BSDFAll & ~BSDFSpecular): the specular materials do not match the request (pdf 0, Black);
    // Lambert and Blinn sample exactly as under BSDFAll
    // (a MaskMaterial forwards the request to the wrapped material when uComponent < alpha -- Black, pdf 0 for those -- and
    // answers it itself with the BSDFnullptr lobe otherwise, GoblinMaterial.cpp:757-784)
    if (mat_is_specular(mat.m) && !(mat.is_mask && !(bs_comp < mat.alpha))) return Ld;
    int sampled = 0;
    Col f = rmat_sample(mat, frag, wo, bs_comp, bs_dir[0], bs_dir[1], &wi, &bsdf_pdf, &sampled);
    if (f != BLACK && bsdf_pdf > 0.0f) {
        float fw = 1.0f;
        if (!(sampled & BSDF_SPECULAR)) {
            light_pdf_v = light_pdf(s, light, p, wi);
            if (light_pdf_v == 0.0f) return Ld;
            fw = power_heuristic(1, bsdf_pdf, 1, light_pdf_v);
        }
        Ray r;
        r.o = p; r.d = wi; r.mint = epsilon; r.maxt = INF;
        Hit lh;
        lh.frag = hit.frag;
        if (scene_intersect(s, r, &lh, &c->cnt)) {
            if (s->instances[lh.instance].area_light == light) {
                Col Le = hit_Le(s, lh, -wi);
                if (Le != BLACK) Ld += f * Le * absdot(wi, n) * fw / bsdf_pdf;
            }
        } else {
            Ld += f * light_le_escaped(s, light, r.d) * fw / bsdf_pdf;   // the radiance contribution from IBL (GoblinRenderer.cpp:558-561)
        }
    }
    return Ld;
}

// Renderer::multiSampleLd (:474-500)
Col multi_sample_ld(LiCtx* c, const Ray& ray, float epsilon, const Hit& hit, const ResolvedMat& mat, const float* rec) {
    const orc_scene* s = c->s;
    Col total(BLACK);
    for (size_t i = 0; i < s->lights.size(); ++i) {
        Col Ld(BLACK);
        uint32_t n_samples = std::min(c->q->n1[c->ix->light1[i]], c->q->n2[c->ix->light2[i]]);   // LightSampleIndex::samplesNum
        for (uint32_t n = 0; n < n_samples; ++n) {
            // LightSample ls(rng); BSDFSample bs(rng); -- six draws, then both are replaced by the Sample's values
            draw_bsdf_sample(c);
            draw_bsdf_sample(c);
            float ls_comp = rec[c->q->off1[c->ix->light1[i]] + n];
            const float* ls_geo = rec + c->q->off2[c->ix->light2[i]] + 2 * n;
            float bs_comp = rec[c->q->off1[c->ix->bsdf1[i]] + n];
            const float* bs_dir = rec + c->q->off2[c->ix->bsdf2[i]] + 2 * n;
            c->dims_used += 6;
            Ld += estimate_ld(c, -ray.d, epsilon, hit, mat, static_cast<int>(i), ls_comp, ls_geo, bs_comp, bs_dir);
        }
        Ld = Ld / static_cast<float>(n_samples);
        total += Ld;
    }
    return total;
}

Col whitted_li(LiCtx* c, const Ray& ray_in, const float* rec, const RayDiff* diff, int depth) {
    const orc_scene* s = c->s;
    Col Li(BLACK);
    Ray ray = ray_in;
    Hit hit;
    hit.frag.n = V3(0, 0, 0);
    hit.frag.dpdv = V3(0, 0, 0);
    const bool primary_hit = scene_intersect(s, ray, &hit, &c->cnt);
    if (depth == 0) c->primary_maxt = ray.maxt;
    if (!primary_hit) {   // get image based lighting if the ray didn't hit anything (GoblinWhitted.cpp:40-43)
        Li += environment_le(s, ray.d);
        return Li;
    }
    compute_uv_differential(&hit.frag, diff);
    Li += hit_Le(s, hit, -ray.d);
    // Lsubsurface with the fragment's differentials in place (GoblinWhitted.cpp:22-27: after computeUVDifferential, at every
    // level of the recursion, always from the camera sample's one BSSRDF block).  The material's Fresnel mirror lobe never
    // answers the two specular requests below: its type is BSDFAll
    Li += l_subsurface(c, hit, -ray.d, rec);
    const ResolvedMat mat = resolve_hit_material(s, s->instances[hit.instance].material, hit.frag);
    Li += multi_sample_ld(c, ray, hit.epsilon, hit, mat, rec);
    if (depth < c->rs->max_ray_depth) {
        const Frag& frag = hit.frag;
        V3 n = frag.n, p = frag.p, wo = -ray.d;
        // specularReflect (:598-622): sampleBSDF(..., BSDFSample(rng), BSDFSpecular | BSDFReflection)
        {
            Col L(BLACK);
            draw_bsdf_sample(c);
            V3 wi(0, 0, 0);
            float pdf = 0.0f;
            Col f(BLACK);
            if (mat.m.type == GBL_MAT_MIRROR) {
                f = mat_color(mat.m) * specular_reflect_conductor(n, wo, &wi, mat.m.index, mat.m.k);
                pdf = 1.0f;
            } else if (mat.m.type == GBL_MAT_TRANSPARENT) {   // nMatch == 1, the reflection lobe (:660-667)
                f = mat_color(mat.m) * specular_reflect_dielectric(n, wo, &wi, 1.0f, mat.m.index);
                pdf = 1.0f;
            }
            // MaskMaterial: the request has no BSDFnullptr bit, so it is the wrapped material's answer times alpha, at the
            // wrapped material's pdf (GoblinMaterial.cpp:781-783)
            if (mat.is_mask) f = mat.alpha * f;
            if (f != BLACK && absdot(wi, n) != 0.0f) {
                Ray child;
                child.o = p; child.d = wi; child.mint = hit.epsilon; child.maxt = INF;
                Col Lr = whitted_li(c, child, rec, nullptr, depth + 1);
                L += f * Lr * absdot(wi, n) / pdf;
            }
            Li += L;
        }
        // specularRefract (:624-648): BSDFSpecular | BSDFTransmission
        {
            Col L(BLACK);
            draw_bsdf_sample(c);
            V3 wi(0, 0, 0);
            float pdf = 0.0f;
            Col f(BLACK);
            if (mat.m.type == GBL_MAT_TRANSPARENT) {
                f = mat_color2(mat.m) * specular_refract(n, wo, &wi, 1.0f, mat.m.index);
                pdf = 1.0f;
            }
            // MaskMaterial: the request has no BSDFnullptr bit, so it is the wrapped material's answer times alpha, at the
            // wrapped material's pdf (GoblinMaterial.cpp:781-783)
            if (mat.is_mask) f = mat.alpha * f;
            if (f != BLACK && absdot(wi, n) != 0.0f) {
                Ray child;
                child.o = p; child.d = wi; child.mint = hit.epsilon; child.maxt = INF;
                Col Lr = whitted_li(c, child, rec, nullptr, depth + 1);
                L += f * Lr * absdot(wi, n) / pdf;
            }
            Li += L;
        }
    }
    return Li;
}

// ---------------------------------------------------------------------------
// The participating medium around the camera ray: RenderTask::run's tr * L + Lv (GoblinRenderer.cpp:40-47) with
// Renderer::transmittance / Renderer::Lv (:298-455, homogeneous branch) over HomogeneousVolumeRegion
// (GoblinVolume.cpp:12-36, GoblinVolume.h:72-112).  Its random numbers come straight from the tile's generator;
// without one (replay / native records) they are a hash of the sample's image position, the rule the device's
// non-stream modes share.
// ---------------------------------------------------------------------------
inline uint32_t nat_mix(uint32_t a, uint32_t b);
inline float nat_u01(uint32_t h);
struct VolRand {
    LiCtx* c;
    uint32_t key, i = 0;
    VolRand(LiCtx* ctx, const float* rec) : c(ctx) {
        uint32_t bx, by;
        memcpy(&bx, rec, 4);
        memcpy(&by, rec + 1, 4);
        key = nat_mix(bx, by);
    }
    float f() { return c->rng ? c->rng->f() : nat_u01(nat_mix(key, 0x766f6c00u + i++)); }
};
inline float vol_rand_next(VolRand& rnd);
inline bool box_intersect(const Box& b, V3 o, V3 d, float mint, float maxt, float* tmin, float* tmax) {   // BBox::intersect, GoblinBBox.cpp:57-77
    float t0 = mint, t1 = maxt;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z};
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / dd[i];
        float tn = (lo[i] - oo[i]) * inv, tf = (hi[i] - oo[i]) * inv;
        if (tn > tf) std::swap(tn, tf);
        t0 = (tn > t0) ? tn : t0;
        t1 = (tf < t1) ? tf : t1;
        if (t0 > t1) return false;
    }
    *tmin = t0;
    *tmax = t1;
    return true;
}
inline bool vol_intersect(const orc_scene* s, const Ray& r, float* tmin, float* tmax) {   // VolumeRegion::intersect, GoblinVolume.cpp:12-15
    const auto& v = s->volume;
    return box_intersect(v.box, v.xf.invert_point(r.o), v.xf.invert_vector(r.d), r.mint, r.maxt, tmin, tmax);
}
inline bool vol_contains(const orc_scene* s, V3 p) {   // BBox::contain(invertPoint(p))
    const auto& v = s->volume;
    V3 q = v.xf.invert_point(p);
    return v.box.lo.x <= q.x && q.x <= v.box.hi.x && v.box.lo.y <= q.y && q.y <= v.box.hi.y && v.box.lo.z <= q.z && q.z <= v.box.hi.z;
}
// VolumeGrid::getVoxel / eval (GoblinVolume.cpp:148-196): trilinear interpolation of the density at a point of the region's
// own space.  (As written there the index scale is  normalize * n - 0.5  -- the half-cell shift multiplies.)
inline Col grid_voxel(const orc_scene::Volume& v, int x, int y, int z) {
    if (x < 0 || x >= v.nx || y < 0 || y >= v.ny || z < 0 || z >= v.nz) return Col(0.0f);
    if (v.nch == 1) return Col(v.density[z * v.nx * v.ny + y * v.nx + x]);
    if (v.nch == 3) {
        int off = 3 * (z * v.nx * v.ny + y * v.nx + x);
        return Col(v.density[off], v.density[off + 1], v.density[off + 2]);
    }
    return Col(0.0f);
}
inline Col col_lerp(float t, Col a, Col b) { return (1.0f - t) * a + t * b; }   // lerp<T>, GoblinUtils.h:109-112
inline Col grid_eval(const orc_scene::Volume& v, V3 p_local) {
    V3 f = p_local - v.box.lo;
    f.x *= v.normalize.x * v.nx - 0.5f;
    f.y *= v.normalize.y * v.ny - 0.5f;
    f.z *= v.normalize.z * v.nz - 0.5f;
    int ix = floor_int(f.x), iy = floor_int(f.y), iz = floor_int(f.z);
    float dx = f.x - ix, dy = f.y - iy, dz = f.z - iz;
    Col d00 = col_lerp(dx, grid_voxel(v, ix, iy, iz), grid_voxel(v, ix + 1, iy, iz));
    Col d10 = col_lerp(dx, grid_voxel(v, ix, iy + 1, iz), grid_voxel(v, ix + 1, iy + 1, iz));
    Col d01 = col_lerp(dx, grid_voxel(v, ix, iy, iz + 1), grid_voxel(v, ix + 1, iy, iz + 1));
    Col d11 = col_lerp(dx, grid_voxel(v, ix, iy + 1, iz + 1), grid_voxel(v, ix + 1, iy + 1, iz + 1));
    Col d0 = col_lerp(dy, d00, d10);
    Col d1 = col_lerp(dy, d01, d11);
    return col_lerp(dz, d0, d1);
}
// HeterogeneousVolumeRegion::getAttenuation (:313-321)
inline Col hetero_attenuation(const orc_scene* s, V3 p) {
    const auto& v = s->volume;
    V3 q = v.xf.invert_point(p);
    bool inside = v.box.lo.x <= q.x && q.x <= v.box.hi.x && v.box.lo.y <= q.y && q.y <= v.box.hi.y && v.box.lo.z <= q.z && q.z <= v.box.hi.z;
    return inside ? grid_eval(v, q) : Col(0.0f);
}
struct VolRand;
inline float vol_rand_next(VolRand& rnd);
// HomogeneousVolumeRegion::transmittance (:25-36): Beer's law, no random number.  HeterogeneousVolumeRegion::transmittance
// (:323-341): a jittered ray march -- one random number -- and BLACK, not white, for a ray that misses the region.
inline Col vol_transmittance(const orc_scene* s, const Ray& r, VolRand& rnd) {
    float tmin, tmax;
    if (s->volume.hetero) {
        if (!vol_intersect(s, r, &tmin, &tmax)) return Col(0.0f);
        const float step = s->volume.step;
        float t = tmin;
        float jitter = vol_rand_next(rnd) * step;
        Col tau = jitter * hetero_attenuation(s, r.o + t * r.d);
        t += jitter;
        while (t + step < tmax) {
            tau += step * hetero_attenuation(s, r.o + t * r.d);
            t += step;
        }
        tau += (tmax - t) * hetero_attenuation(s, r.o + t * r.d);
        return Col(std::exp(-tau.r), std::exp(-tau.g), std::exp(-tau.b));
    }
    if (!vol_intersect(s, r, &tmin, &tmax)) return Col(1.0f);
    Col tau = length((r.o + tmax * r.d) - (r.o + tmin * r.d)) * s->volume.attenuation;
    return Col(std::exp(-tau.r), std::exp(-tau.g), std::exp(-tau.b));
}
// Light::samplePosition (GoblinLight.cpp:101-108, 161-175, 239-244, 396-409)
V3 light_sample_position(const orc_scene* s, int li, float u_comp, float u1, float u2) {
    const Light& l = s->lights[li];
    if (l.type == GBL_LIGHT_AREA) {
        const Mesh& m = s->meshes[l.mesh];
        V3 p_local;
        if (m.shape == GBL_SHAPE_SPHERE) {
            p_local = m.radius * uniform_sample_sphere(u1, u2);   // Sphere::sample(u1, u2, &n), GoblinSphere.cpp:103-106
        } else if (m.shape == GBL_SHAPE_DISK) {
            float x, y;
            uniform_sample_disk(u1, u2, &x, &y);                  // Disk::sample, GoblinDisk.cpp:78-82
            p_local = V3(m.radius * x, m.radius * y, 0.0f);
        } else {
            int tri = s->light_geo_cdf[li].sample_discrete(u_comp, nullptr);   // GeometrySet::sample(ls, &n), :326-334
            float root = sqrtf(u1);
            float b0 = 1.0f - root, b1 = root * u2;
            V3 p0 = m.P(m.idx[3 * tri]), p1 = m.P(m.idx[3 * tri + 1]), p2 = m.P(m.idx[3 * tri + 2]);
            p_local = b0 * p0 + b1 * p1 + (1.0f - b0 - b1) * p2;
        }
        return l.xf.on_point(p_local);
    }
    if (l.type == GBL_LIGHT_DIRECTIONAL) {
        // a disc of the scene's bounding sphere, pushed back along the light's direction
        V3 center = 0.5f * (s->tlas.bounds.lo + s->tlas.bounds.hi);
        float radius = length(s->tlas.bounds.hi - s->tlas.bounds.lo);
        V3 z = l.spot_axis, x, y;
        coordinate_axes(z, &x, &y);
        float dx, dy;
        uniform_sample_disk(u1, u2, &dx, &dy);
        V3 disk = center + radius * (dx * x + dy * y);
        return disk - z * radius;
    }
    if (l.type == GBL_LIGHT_IBL) {   // ImageBasedLight::samplePosition (:556-568): a point of the scene's bounding sphere
        V3 center = 0.5f * (s->tlas.bounds.lo + s->tlas.bounds.hi);
        float radius = length(s->tlas.bounds.hi - s->tlas.bounds.lo);
        return center + radius * uniform_sample_sphere(u1, u2);
    }
    return l.pos;
}
inline float vol_rand_next(VolRand& rnd) { return rnd.f(); }
// Renderer::Lv, heterogeneous branch (GoblinRenderer.cpp:397-445): march the camera ray in steps of step_size from a
// jittered start; at every sample point one light sample (pick + LightSample: 4 random numbers, one more for the shadow
// ray's own jittered transmittance when it is unoccluded).
Col volume_lv_hetero(LiCtx* c, const Ray& ray, float tmin, float tmax, VolRand& rnd) {
    const orc_scene* s = c->s;
    const auto& vol = s->volume;
    Col Lv(0.0f);
    const float step = vol.step;
    V3 p_prev = ray.o + tmin * ray.d;
    float t = tmin + step * rnd.f();
    V3 p = ray.o + t * ray.d;
    Col transmittance(1.0f);
    while (t <= tmax) {
        // HeterogeneousVolumeRegion::eval (:297-311)
        Col sigma_t = hetero_attenuation(s, p);
        Col sigma_s = sigma_t * vol.albedo;
        Col emission(0.0f);
        Col tau = sigma_t * length(p - p_prev);
        transmittance *= Col(std::exp(-tau.r), std::exp(-tau.g), std::exp(-tau.b));
        Lv += transmittance * emission;
        float pick = rnd.f();
        float pick_pdf = 0.0f;
        int light = s->lights.empty() ? -1 : s->light_power.sample_discrete(pick, &pick_pdf);
        if (light >= 0 && pick_pdf != 0.0f) {
            float u_comp = rnd.f(), u1 = rnd.f(), u2 = rnd.f();   // LightSample ls(rng)
            V3 wi;
            float light_pdf_v;
            Ray shadow;
            Col L = light_sample(s, light, p, 0.0f, u_comp, u1, u2, &wi, &light_pdf_v, &shadow);
            if (L != BLACK && light_pdf_v > 0.0f) {
                if (!scene_occluded(s, shadow, &c->cnt)) {
                    Col tr_light = vol_transmittance(s, shadow, rnd);
                    Col Ld = tr_light * L / (pick_pdf * light_pdf_v);
                    float phase = vol_contains(s, p) ? phase_hg(ray.d, wi, vol.g) : 0.0f;   // VolumeRegion::phase
                    Lv += transmittance * sigma_s * phase * Ld;
                }
            }
        }
        t += step;
        p_prev = p;
        p = ray.o + t * ray.d;
    }
    return step * Lv;
}
// Renderer::Lv, homogeneous branch (GoblinRenderer.cpp:298-391)
Col volume_lv(LiCtx* c, const Ray& ray, VolRand& rnd) {
    const orc_scene* s = c->s;
    const auto& vol = s->volume;
    float tmin, tmax;
    if (!vol.on || !vol_intersect(s, ray, &tmin, &tmax)) return BLACK;
    if ((tmax - tmin) < 1e-5f) return BLACK;
    if (vol.hetero) return volume_lv_hetero(c, ray, tmin, tmax, rnd);
    Col Lv(0.0f);
    const int n_samples = vol.sample_num;
    for (int i = 0; i < n_samples; ++i) {
        float pick = rnd.f();
        float pick_pdf = 0.0f;
        int light = s->lights.empty() ? -1 : s->light_power.sample_discrete(pick, &pick_pdf);
        if (light < 0 || pick_pdf == 0.0f) continue;
        float e_comp = rnd.f(), e_u1 = rnd.f(), e_u2 = rnd.f();   // LightSample lsEqui(rng)
        V3 p_light = light_sample_position(s, light, e_comp, e_u1, e_u2);
        float delta = dot(p_light - ray.o, ray.d);
        float a = tmin - delta, b = tmax - delta;
        float D = length(p_light - (ray.o + delta * ray.d));
        float theta_a = std::atan2(a, D), theta_b = std::atan2(b, D);
        float ue = rnd.f();
        float te = D * std::tan((1 - ue) * theta_a + ue * theta_b);            // equiAngularSample, GoblinSampler.h:276-279
        float pdf_te = D / ((theta_b - theta_a) * (D * D + te * te));          // equiAngularPdf
        V3 p_e = ray.o + (delta + te) * ray.d;
        bool in_e = vol_contains(s, p_e);
        Col sigma_te = in_e ? vol.attenuation : Col(0.0f), scatter_e = in_e ? vol.scatter : Col(0.0f);
        Col tr_e(std::exp(-sigma_te.r * (te - a)), std::exp(-sigma_te.g * (te - a)), std::exp(-sigma_te.b * (te - a)));
        {
            V3 wi;
            float light_pdf_v;
            Ray shadow;
            Col Le = light_sample(s, light, p_e, 0.0f, e_comp, e_u1, e_u2, &wi, &light_pdf_v, &shadow);
            if (Le != BLACK && light_pdf_v > 0.0f) {
                if (!scene_occluded(s, shadow, &c->cnt)) {
                    Col tr_light = vol_transmittance(s, shadow, rnd);
                    Col Ld = tr_light * Le / (pick_pdf * light_pdf_v);
                    float phase = in_e ? phase_hg(ray.d, wi, vol.g) : 0.0f;   // VolumeRegion::phase(p, wi, wo), GoblinVolume.cpp:17-23
                    float sig = luminance(sigma_te);
                    float pdf_td = sig / (std::exp(sig * (te - a)) - std::exp(sig * (te - b)));   // exponentialPdf(t, sigma, a, b)
                    float mis = power_heuristic(1, pdf_te, 1, pdf_td);
                    Lv += mis * tr_e * scatter_e * phase * Ld / pdf_te;
                }
            }
        }
        // distance sampling
        Col sigma_td = vol_contains(s, ray.o + (0.5f * (tmin + tmax)) * ray.d) ? vol.attenuation : Col(0.0f);
        float ud = rnd.f();
        float sig_d = luminance(sigma_td);
        float td = a - std::log(1.0f - ud * (1.0f - std::exp(sig_d * (a - b)))) / sig_d;   // exponentialSample(u, sigma, a, b)
        float pdf_td = sig_d / (std::exp(sig_d * (td - a)) - std::exp(sig_d * (td - b)));
        V3 p_d = ray.o + (delta + td) * ray.d;
        Col tr_d(std::exp(-sigma_td.r * (td - a)), std::exp(-sigma_td.g * (td - a)), std::exp(-sigma_td.b * (td - a)));
        bool in_d = vol_contains(s, p_d);
        Col scatter_d = in_d ? vol.scatter : Col(0.0f);
        float d_comp = rnd.f(), d_u1 = rnd.f(), d_u2 = rnd.f();   // LightSample lsDistance(rng)
        {
            V3 wi;
            float light_pdf_v;
            Ray shadow;
            Col Ldist = light_sample(s, light, p_d, 0.0f, d_comp, d_u1, d_u2, &wi, &light_pdf_v, &shadow);
            if (Ldist != BLACK && light_pdf_v > 0.0f) {
                if (!scene_occluded(s, shadow, &c->cnt)) {
                    Col tr_light = vol_transmittance(s, shadow, rnd);
                    Col Ld = tr_light * Ldist / (pick_pdf * light_pdf_v);
                    float phase = in_d ? phase_hg(ray.d, wi, vol.g) : 0.0f;
                    float pdf_te2 = D / ((theta_b - theta_a) * (D * D + td * td));
                    float mis = power_heuristic(1, pdf_td, 1, pdf_te2);
                    Lv += mis * tr_d * scatter_d * phase * Ld / pdf_td;
                }
            }
        }
    }
    return Lv / static_cast<float>(n_samples);
}
// what RenderTask::run adds to the tile for one camera sample: w * (tr * L + Lv), w = 1
inline Col task_sample(LiCtx* c, const float* rec, Col L) {
    Col tr(1.0f), Lv(BLACK);
    if (c->s->volume.on) {
        Ray ray = camera_ray(c->s, rec[0], rec[1], rec[2], rec[3], nullptr);
        ray.maxt = c->primary_maxt;
        VolRand rnd(c, rec);
        tr = vol_transmittance(c->s, ray, rnd);   // Renderer::transmittance: the homogeneous region draws nothing, the heterogeneous one its jitter
        Lv = volume_lv(c, ray, rnd);
    }
    Col TL = tr * L;
    Col out(TL.r + Lv.r, TL.g + Lv.g, TL.b + Lv.b, TL.a);
    return 1.0f * out;
}

inline Col eval_li(LiCtx* c, const float* rec) {
    RayDiff rd;
    Ray ray = camera_ray(c->s, rec[0], rec[1], rec[2], rec[3], &rd);
    c->primary_maxt = INF;
    c->dims_used += 2;
    if (c->rs->integrator == GBL_INTEGRATOR_WHITTED) return whitted_li(c, ray, rec, &rd, 0);
    return c->rs->integrator == GBL_INTEGRATOR_AO ? ao_li(c, ray, rec) : path_li(c, ray, rec, &rd);
}

// ---------------------------------------------------------------------------
// "Native" sampler: the counter-based law the device path uses, restated here
// bit-for-bit (integer hashing only) so GPU-native renders can be checked
// sample by sample.  Same stratification as Sampler::requestSamples: every
// pattern is jittered over (strata x per-pixel sub-strata) and the sub-stratum
// a camera sample receives is a per-(pixel, pattern, stratum) permutation of
// the sample index; image samples are not permuted (GoblinSampler.cpp:130-131).
// Definition shared with goblin_amd/csrc/kernels/sampler.hip.h (restated there,
// not included from here).
// ---------------------------------------------------------------------------
inline uint32_t nat_mix(uint32_t a, uint32_t b) {
    uint32_t h = (a ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h ^= b + 0x7F4A7C15u + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
inline float nat_u01(uint32_t h) { return static_cast<float>(h >> 8) * (1.0f / 16777216.0f); }
// keyed bijection on [0, n): cycle-walked 3-round xor/multiply/xorshift network on the next power of two
inline uint32_t nat_permute(uint32_t i, uint32_t n, uint32_t key) {
    if (n <= 1) return 0;
    uint32_t w = n - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    uint32_t k1 = nat_mix(key, 0x3C6EF372u) | 1u, k2 = nat_mix(key, 0xDAA66D2Bu) | 1u;
    do {
        i ^= key & w;
        i = (i * k1) & w;
        i ^= i >> 3;
        i ^= (key >> 11) & w;
        i = (i * k2) & w;
        i ^= i >> 5;
        i = (i * 0x2C1B3C6Du) & w;
        i ^= i >> 2;
    } while (i >= n);
    return i;
}

struct NativeSampler {
    uint32_t seed_key;
    int spp, root;
    NativeSampler(uint64_t seed, int sample_per_pixel) {
        spp = round_to_square(sample_per_pixel, &root);
        seed_key = nat_mix(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    }
    // pattern ids: 0 image, 1 lens, 2+i one-D pattern i, 0x10000+i two-D pattern i
    uint32_t key(uint32_t pixel, uint32_t pattern, uint32_t stratum) const {
        return nat_mix(nat_mix(nat_mix(seed_key, pixel), pattern), stratum);
    }
    float one_d(uint32_t pixel, uint32_t pattern, uint32_t n, uint32_t i, uint32_t k) const {
        uint32_t ky = key(pixel, 2 + pattern, i);
        uint32_t j = nat_permute(k, spp, ky);
        float strata = 1.0f / static_cast<float>(n);
        float sub = strata / spp;
        float off = j + nat_u01(nat_mix(ky, k));
        return i * strata + off * sub;
    }
    void two_d(uint32_t pixel, uint32_t pattern_id, uint32_t n, uint32_t i, uint32_t k, bool permute, float out[2]) const {
        uint32_t ky = key(pixel, pattern_id, i);
        uint32_t p = permute ? nat_permute(k, spp, ky) : k;
        int r = static_cast<int>(sqrtf(static_cast<float>(n)));
        float strata = 1.0f / r;
        float sub = strata / root;
        int ux = i % r, uy = i / r;
        int px = p % root, py = p / root;
        float xo = px + nat_u01(nat_mix(ky, 2 * k));
        float yo = py + nat_u01(nat_mix(ky, 2 * k + 1));
        out[0] = ux * strata + xo * sub;
        out[1] = uy * strata + yo * sub;
    }
    // Fill one record.  `pixel` = linear index in the FULL sample window, so a
    // sharded render draws the same numbers as a whole one.
    void fill(const Quota& q, uint32_t pixel, int px, int py, uint32_t k, float* rec) const {
        float im[2], ln[2];
        two_d(pixel, 0, 1, 0, k, false, im);
        two_d(pixel, 1, 1, 0, k, true, ln);
        rec[0] = px + im[0];
        rec[1] = py + im[1];
        rec[2] = ln[0];
        rec[3] = ln[1];
        // slot j of an n-strata pattern holds stratum j, or -- where the slots of several patterns are consumed together
        // (Quota::perm*_from) -- stratum perm(j) under a per-sample keyed bijection, the stand-in for the reference's
        // in-pattern shuffle (GoblinSampler.cpp:171-196); without it slot j of every pattern would share stratum j
        for (size_t i = 0; i < q.n1.size(); ++i)
            for (uint32_t j = 0; j < q.n1[i]; ++j) {
                uint32_t st = j;
                if (i >= q.perm1_from && q.n1[i] > 1)
                    st = nat_permute(j, q.n1[i], nat_mix(key(pixel, 2 + static_cast<uint32_t>(i), 0x5bd1e995u), k));
                rec[q.off1[i] + j] = one_d(pixel, static_cast<uint32_t>(i), q.n1[i], st, k);
            }
        for (size_t i = 0; i < q.n2.size(); ++i)
            for (uint32_t j = 0; j < q.n2[i]; ++j) {
                uint32_t st = j;
                if (i >= q.perm2_from && q.n2[i] > 1)
                    st = nat_permute(j, q.n2[i], nat_mix(key(pixel, 0x10000u + static_cast<uint32_t>(i), 0x5bd1e995u), k));
                two_d(pixel, 0x10000u + static_cast<uint32_t>(i), q.n2[i], st, k, true, rec + q.off2[i] + 2 * j);
            }
    }
};

Quota make_quota(const gbl_render_setting& rs, PtIndices* ix, const orc_scene* s = nullptr) {
    if (rs.integrator == GBL_INTEGRATOR_WHITTED && s) return whitted_quota(s, rs, ix);
    return rs.integrator == GBL_INTEGRATOR_AO ? ao_quota(rs) : pt_quota(rs, ix);
}

}  // namespace

// ===========================================================================
// C interface (used by tests/, smoke() and bench.py's cpu_baseline only)
// ===========================================================================
extern "C" {

typedef struct orc_counters {
    uint64_t paths, closest_queries, anyhit_queries, filtered_queries, nodes, tris, splats, dims;
} orc_counters;

orc_scene* orc_create(const gbl_scene_desc* desc) {
    if (!desc || desc->abi_version != GBL_ABI_VERSION) return nullptr;
    orc_scene* s = new orc_scene();
    s->desc = *desc;   // arrays stay owned by the caller and must outlive the scene
    prepare(s);
    return s;
}

void orc_destroy(orc_scene* s) { delete s; }

void orc_sample_window(const orc_scene* s, int32_t out[4]) { memcpy(out, s->window, sizeof(s->window)); }

int32_t orc_sample_dimension(const gbl_render_setting* rs) { return static_cast<int32_t>(make_quota(*rs, nullptr).dims()); }
int32_t orc_sample_dimension_scene(const orc_scene* s, const gbl_render_setting* rs) {
    return static_cast<int32_t>(make_quota(*rs, nullptr, s).dims());
}

// Float offsets of the path tracer's per-bounce sample slots inside a record:
// out[5*b + {0..4}] = {light component, light geometry(2), bsdf component, bsdf direction(2), pick light}
void orc_pt_offsets(const gbl_render_setting* rs, int32_t* out) {
    PtIndices ix;
    Quota q = pt_quota(*rs, &ix);
    for (size_t b = 0; b < ix.pick.size(); ++b) {
        out[5 * b + 0] = q.off1[ix.light1[b]];
        out[5 * b + 1] = q.off2[ix.light2[b]];
        out[5 * b + 2] = q.off1[ix.bsdf1[b]];
        out[5 * b + 3] = q.off2[ix.bsdf2[b]];
        out[5 * b + 4] = q.off1[ix.pick[b]];
    }
}

// First `n` values of glibc rand() after srand(1)
void orc_glibc_rand(int32_t* out, int32_t n) {
    GlibcRand g(1);
    for (int i = 0; i < n; ++i) out[i] = g.next();
}

// --- known-answer probes ---------------------------------------------------
void orc_filter_table(const orc_scene* s, float out[256]) { memcpy(out, s->filter_table, sizeof(s->filter_table)); }
void orc_camera_ray(const orc_scene* s, float image_x, float image_y, float out[8]) {
    Ray r = camera_ray(s, image_x, image_y);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z;
    out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
    out[6] = r.mint; out[7] = r.maxt;
}
int32_t orc_light_power(const orc_scene* s, float* out4_per_light) {
    for (size_t i = 0; i < s->lights.size(); ++i) {
        Col p = s->light_power_rgb[i];
        out4_per_light[4 * i] = p.r; out4_per_light[4 * i + 1] = p.g; out4_per_light[4 * i + 2] = p.b;
        out4_per_light[4 * i + 3] = luminance(p);
    }
    return static_cast<int32_t>(s->lights.size());
}
// closest hit along a ray: out = {t, eps, p(3), n(3), tangent(3), instance}
int32_t orc_intersect(const orc_scene* s, const float o[3], const float d[3], float mint, float maxt, float out[12]) {
    Ray r;
    r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.mint = mint; r.maxt = maxt < 0 ? INF : maxt;
    Hit h;
    h.frag.n = V3(0, 0, 0); h.frag.dpdv = V3(0, 0, 0);
    Counters c;
    if (!scene_intersect(s, r, &h, &c)) return 0;
    Frame fr = shade_frame(h.frag);
    out[0] = r.maxt; out[1] = h.epsilon;
    out[2] = h.frag.p.x; out[3] = h.frag.p.y; out[4] = h.frag.p.z;
    out[5] = h.frag.n.x; out[6] = h.frag.n.y; out[7] = h.frag.n.z;
    out[8] = fr.t.x; out[9] = fr.t.y; out[10] = fr.t.z;
    out[11] = static_cast<float>(h.instance);
    return 1;
}
int32_t orc_occluded(const orc_scene* s, const float o[3], const float d[3], float mint, float maxt) {
    Ray r;
    r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.mint = mint; r.maxt = maxt < 0 ? INF : maxt;
    Counters c;
    return scene_occluded(s, r, &c) ? 1 : 0;
}

// --- replay: Li for caller-supplied Sample records -------------------------
// samples: n records of orc_sample_dimension floats; li_out: n rgba.
int32_t orc_li_replay(const orc_scene* s, const gbl_render_setting* rs, const float* samples, int64_t n, float* li_out,
                      int32_t threads, orc_counters* counters) {
    PtIndices ix;
    Quota q = make_quota(*rs, &ix, s);
    uint32_t dims = q.dims();
    int nt = std::max(1, threads);
    std::vector<Counters> cnts(nt);
    std::vector<uint64_t> dims_used(nt, 0);
    std::vector<std::thread> pool;
    auto work = [&](int tid) {
        LiCtx c;
        c.s = s; c.rs = rs; c.q = &q; c.ix = &ix; c.rng = nullptr; c.ref_faithful = 0;
        for (int64_t i = tid; i < n; i += nt) {
            Col L = eval_li(&c, samples + i * dims);
            if (s->volume.on) L = task_sample(&c, samples + i * dims, L);
            li_out[4 * i] = L.r; li_out[4 * i + 1] = L.g; li_out[4 * i + 2] = L.b; li_out[4 * i + 3] = L.a;
        }
        cnts[tid] = c.cnt;
        dims_used[tid] = c.dims_used;
    };
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        counters->paths = static_cast<uint64_t>(n);
        for (int t = 0; t < nt; ++t) {
            counters->closest_queries += cnts[t].closest; counters->anyhit_queries += cnts[t].anyhit;
            counters->filtered_queries += cnts[t].filtered; counters->nodes += cnts[t].nodes; counters->tris += cnts[t].tris;
            counters->dims += dims_used[t];
        }
    }
    return 0;
}

// Debugging aid: every scene query the Li evaluation of ONE record issues (see Counters::log); returns the count.
int32_t orc_debug_rays(const orc_scene* s, const gbl_render_setting* rs, const float* rec, float* out, int32_t max_rays) {
    PtIndices ix;
    Quota q = make_quota(*rs, &ix, s);
    std::vector<float> log;
    LiCtx c;
    c.s = s; c.rs = rs; c.q = &q; c.ix = &ix; c.rng = nullptr; c.ref_faithful = 0;
    c.cnt.log = &log;
    eval_li(&c, rec);
    int32_t n = static_cast<int32_t>(log.size() / 16);
    memcpy(out, log.data(), sizeof(float) * 16 * std::min(n, max_rays));
    return n;
}

// Splat (sample, Li) pairs into a film in order: ImageTile::addSample.
int32_t orc_splat(const orc_scene* s, const float* samples, int32_t dims, const float* li, int64_t n, float* film_accum) {
    uint64_t splats = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* rec = samples + i * dims;
        add_sample(s, film_accum, rec[0], rec[1], Col(li[4 * i], li[4 * i + 1], li[4 * i + 2], li[4 * i + 3]), &splats);
    }
    return static_cast<int32_t>(splats > 0x7fffffff ? 0x7fffffff : splats);
}

// Native (counter-based) sample records for a sub-window, pixel-major.
int32_t orc_native_samples(const orc_scene* s, const gbl_render_setting* rs, uint64_t seed, const int32_t window[4], float* out) {
    Quota q = make_quota(*rs, nullptr, s);
    NativeSampler ns(seed, rs->sample_per_pixel);
    int fw = s->window[1] - s->window[0];
    uint32_t dims = q.dims();
    size_t rec = 0;
    for (int y = window[2]; y < window[3]; ++y)
        for (int x = window[0]; x < window[1]; ++x) {
            uint32_t pixel = static_cast<uint32_t>((y - s->window[2]) * fw + (x - s->window[0]));
            for (int k = 0; k < ns.spp; ++k) ns.fill(q, pixel, x, y, k, out + (rec++) * dims);
        }
    return 0;
}

// Li of the native sampler's records for a sub-window, pixel-major, with the device's Russian roulette extension on or off
// (rr = 0 gives what orc_native_samples + orc_li_replay give).
int32_t orc_li_native(const orc_scene* s, const gbl_render_setting* rs, uint64_t seed, const int32_t window[4], int32_t rr, float* li_out,
                      int32_t threads) {
    PtIndices ix;
    Quota q = make_quota(*rs, &ix, s);
    const uint32_t dims = q.dims();
    NativeSampler ns(seed, rs->sample_per_pixel);
    const int fw = s->window[1] - s->window[0], ww = window[1] - window[0];
    const int64_t npix = static_cast<int64_t>(ww) * (window[3] - window[2]);
    const int nt = std::max(1, threads);
    std::vector<std::thread> pool;
    auto work = [&](int tid) {
        LiCtx c;
        c.s = s; c.rs = rs; c.q = &q; c.ix = &ix; c.rng = nullptr; c.ref_faithful = 0;
        c.rr = rr != 0;
        std::vector<float> rec(dims);
        for (int64_t pi = tid; pi < npix; pi += nt) {
            const int x = window[0] + static_cast<int>(pi % ww), y = window[2] + static_cast<int>(pi / ww);
            const uint32_t pixel = static_cast<uint32_t>((y - s->window[2]) * fw + (x - s->window[0]));
            c.rr_pixel_key = nat_mix(ns.seed_key, pixel);
            for (int k = 0; k < ns.spp; ++k) {
                ns.fill(q, pixel, x, y, k, rec.data());
                c.rr_k = static_cast<uint32_t>(k);
                Col L = eval_li(&c, rec.data());
                if (s->volume.on) L = task_sample(&c, rec.data(), L);
                float* o = li_out + 4 * (pi * ns.spp + k);
                o[0] = L.r; o[1] = L.g; o[2] = L.b; o[3] = L.a;
            }
        }
    };
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
    return 0;
}

// --- the whole render loop as the reference runs it ------------------------
// Renderer::render + RenderTask::run (GoblinRenderer.cpp:99-126,29-52): 8x8
// tiles in row-major order, one mt19937 per tile seeded from consecutive
// rand() values, stratified sampler, Li, film splat.  threads == 1 reproduces
// the reference's thread_num=1 accumulation order exactly; threads > 1 gives
// every worker its own full film and sums them at the end (the reference's
// per-thread ImageTile + mergeTile).
//   samples_out / li_out: optional, all records in tile-then-pixel order.
//   sampler: 0 = reference stream (mt19937), 1 = native counter-based law.
int32_t orc_render(const orc_scene* s, const gbl_render_setting* rs, int32_t threads, int32_t ref_faithful, int32_t sampler,
                   uint64_t seed, float* film_accum, float* samples_out, float* li_out, double* seconds,
                   orc_counters* counters) {
    PtIndices ix;
    Quota q = make_quota(*rs, &ix, s);
    uint32_t dims = q.dims();
    int root;
    int spp = round_to_square(rs->sample_per_pixel, &root);
    struct Tile {
        int x0, x1, y0, y1;
        uint32_t seed;
        size_t first_record;
    };
    std::vector<Tile> tiles;
    GlibcRand libc(1);
    size_t nrec = 0;
    for (int y = s->window[2]; y < s->window[3]; y += 8)       // Renderer::getSampleRanges, :650-665
        for (int x = s->window[0]; x < s->window[1]; x += 8) {
            Tile t;
            t.x0 = x; t.x1 = std::min(x + 8, s->window[1]);
            t.y0 = y; t.y1 = std::min(y + 8, s->window[3]);
            t.seed = static_cast<uint32_t>(libc.next());        // RNGImp ctor, GoblinUtils.cpp:19-20
            t.first_record = nrec;
            nrec += static_cast<size_t>(t.x1 - t.x0) * (t.y1 - t.y0) * spp;
            tiles.push_back(t);
        }
    size_t film_floats = static_cast<size_t>(s->xres) * s->yres * 4;
    int nt = std::max(1, threads);
    std::vector<std::vector<float>> films(nt > 1 ? nt : 0);
    for (auto& f : films) f.assign(film_floats, 0.0f);
    std::vector<Counters> cnts(nt);
    std::vector<uint64_t> splats(nt, 0), dims_used(nt, 0);
    std::atomic<size_t> next_tile{0};
    NativeSampler native(seed, rs->sample_per_pixel);
    int fw = s->window[1] - s->window[0];
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int tid) {
        float* film = nt > 1 ? films[tid].data() : film_accum;
        std::vector<float> recs(static_cast<size_t>(spp) * dims);
        LiCtx c;
        c.s = s; c.rs = rs; c.q = &q; c.ix = &ix; c.ref_faithful = ref_faithful;
        while (true) {
            size_t ti = nt > 1 ? next_tile.fetch_add(1) : next_tile++;
            if (ti >= tiles.size()) break;
            const Tile& t = tiles[ti];
            Rng rng(t.seed);
            c.rng = sampler == 0 ? &rng : nullptr;
            Sampler smp(t.x0, t.x1, t.y0, t.y1, rs->sample_per_pixel, q, &rng);
            size_t rec_index = t.first_record;
            for (int py = t.y0; py < t.y1; ++py)
                for (int px = t.x0; px < t.x1; ++px) {
                    if (sampler == 0) {
                        smp.request(recs.data());
                    } else {
                        uint32_t pixel = static_cast<uint32_t>((py - s->window[2]) * fw + (px - s->window[0]));
                        for (int k = 0; k < spp; ++k) native.fill(q, pixel, px, py, k, recs.data() + static_cast<size_t>(k) * dims);
                    }
                    for (int k = 0; k < spp; ++k) {
                        const float* rec = recs.data() + static_cast<size_t>(k) * dims;
                        Col L = eval_li(&c, rec);
                        Col out = task_sample(&c, rec, L);   // RenderTask::run: w * (tr * L + Lv)
                        add_sample(s, film, rec[0], rec[1], out, &splats[tid]);
                        if (s->volume.on) L = out;           // with a medium the per-sample output is what the tile receives
                        if (samples_out) memcpy(samples_out + rec_index * dims, rec, dims * sizeof(float));
                        if (li_out) {
                            li_out[4 * rec_index] = L.r; li_out[4 * rec_index + 1] = L.g;
                            li_out[4 * rec_index + 2] = L.b; li_out[4 * rec_index + 3] = L.a;
                        }
                        ++rec_index;
                    }
                }
        }
        cnts[tid] = c.cnt;
        dims_used[tid] = c.dims_used;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
    if (nt > 1)   // Film::mergeTile, GoblinFilm.cpp:140-153
        for (int t = 0; t < nt; ++t)
            for (size_t i = 0; i < film_floats; ++i) film_accum[i] += films[t][i];
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        counters->paths = nrec;
        for (int t = 0; t < nt; ++t) {
            counters->closest_queries += cnts[t].closest; counters->anyhit_queries += cnts[t].anyhit;
            counters->filtered_queries += cnts[t].filtered; counters->nodes += cnts[t].nodes; counters->tris += cnts[t].tris;
            counters->splats += splats[t]; counters->dims += dims_used[t];
        }
    }
    return 0;
}

// MIPMap<T>::lookup of image `image` of the scene for n queries {s, t, dsdx, dtdx, dsdy, dtdy} -> n x {r, g, b, a}
// (the checker of tests/test_oracle_vs_reference.py's lookup-level comparison with the compiled reference)
int32_t orc_mip_lookup(const orc_scene* s, uint32_t image, int32_t is_float, const float* queries, uint64_t n, uint32_t filter, uint32_t mode,
                       float max_aniso, float* out) {
    if (!s || image >= s->images.size() || !queries || !out) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        TexCoord tc;
        tc.s = queries[6 * i], tc.t = queries[6 * i + 1];
        tc.dsdx = queries[6 * i + 2], tc.dtdx = queries[6 * i + 3], tc.dsdy = queries[6 * i + 4], tc.dtdy = queries[6 * i + 5];
        const Col c = mip_lookup(s, s->images[image], is_float != 0, tc, filter, mode, max_aniso);
        out[4 * i] = c.r, out[4 * i + 1] = c.g, out[4 * i + 2] = c.b, out[4 * i + 3] = c.a;
    }
    return 0;
}

int32_t orc_hardware_threads(void) { return static_cast<int32_t>(std::thread::hardware_concurrency()); }

}  // extern "C"
