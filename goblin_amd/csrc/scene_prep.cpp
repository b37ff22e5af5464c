// Host-side scene packing for the device integrator: transforms, the two-level
// BVH (our own binned-SAH builder, NOT the reference's median-split tree), the
// packed triangle / instance / material / light tables, camera and film
// constants.  Replaces the constructors the reference runs at load time:
//   Transform::update            GoblinTransform.cpp:182-193
//   Model / Scene BVH build      GoblinModel.cpp:10-26, GoblinScene.cpp:11-27, GoblinBVH.cpp:34-151
//   SpotLight / AreaLight ctors  GoblinLight.cpp:212-223, 345-361
//   PerspectiveCamera ctor       GoblinCamera.cpp:83-95
//   FilterTable ctor             GoblinFilm.cpp:10-27
// Radiance is BVH-independent except for exact t ties, so the tree is built for
// the GPU: both child boxes in the parent (64 B nodes), <= 4 triangles per leaf,
// SAH splits, depth-capped so the LDS traversal stack has a hard bound.
#include "scene_prep.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

const float kPi = 3.14159265358979323f;
const float kTwoPi = 6.28318530718f;

// ---------------------------------------------------------------- transforms
struct Mat4 {
    float v[4][4];
};

Mat4 identity() {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.v[i][j] = i == j ? 1.0f : 0.0f;
    return r;
}

Mat4 multiply(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a.v[i][0] * b.v[0][j];
            acc = acc + a.v[i][1] * b.v[1][j];
            acc = acc + a.v[i][2] * b.v[2][j];
            acc = acc + a.v[i][3] * b.v[3][j];
            r.v[i][j] = acc;
        }
    return r;
}

Mat4 rotation_of(const float q[4]) {   // q = w x y z
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float xx = tx * x, xy = tx * y, xz = tx * z, xw = tx * w;
    const float yy = ty * y, yz = ty * z, yw = ty * w;
    const float zz = tz * z, zw = tz * w;
    Mat4 r = identity();
    r.v[0][0] = 1 - yy - zz; r.v[0][1] = xy - zw;     r.v[0][2] = xz + yw;
    r.v[1][0] = xy + zw;     r.v[1][1] = 1 - xx - zz; r.v[1][2] = yz - xw;
    r.v[2][0] = xz - yw;     r.v[2][1] = yz + xw;     r.v[2][2] = 1 - xx - yy;
    return r;
}

// 2x2 minor of rows (r0, r1), columns (c0, c1)
inline float minor2(const Mat4& m, int r0, int r1, int c0, int c1) { return m.v[r0][c0] * m.v[r1][c1] - m.v[r0][c1] * m.v[r1][c0]; }

// General 4x4 inverse by cofactors, expanding each cofactor along the row that
// is NOT in the minor's row pair, in the same association as the reference so
// the float32 result is the same; fails (returns false) when |det| < 1e-5.
bool invert(const Mat4& m, Mat4* out) {
    const int cols[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
    float a[6], b[6], c[6];   // minors of row pairs (2,3), (1,3), (1,2)
    for (int k = 0; k < 6; ++k) {
        a[k] = minor2(m, 2, 3, cols[k][0], cols[k][1]);
        b[k] = minor2(m, 1, 3, cols[k][0], cols[k][1]);
        c[k] = minor2(m, 1, 2, cols[k][0], cols[k][1]);
    }
    auto cof = [&](int row, const float s[6], float r[4]) {
        r[0] = m.v[row][1] * s[0] - m.v[row][2] * s[1] + m.v[row][3] * s[2];
        r[1] = m.v[row][0] * s[0] - m.v[row][2] * s[3] + m.v[row][3] * s[4];
        r[2] = m.v[row][0] * s[1] - m.v[row][1] * s[3] + m.v[row][3] * s[5];
        r[3] = m.v[row][0] * s[2] - m.v[row][1] * s[4] + m.v[row][2] * s[5];
    };
    float k0[4], k1[4], k2[4], k3[4];
    cof(1, a, k0);
    cof(0, a, k1);
    cof(0, b, k2);
    cof(0, c, k3);
    const float sgn[4] = {1.0f, -1.0f, 1.0f, -1.0f};
    Mat4& o = *out;
    for (int i = 0; i < 4; ++i) o.v[i][0] = sgn[i] > 0 ? k0[i] : -k0[i];
    float det = m.v[0][0] * o.v[0][0] + m.v[0][1] * o.v[1][0] + m.v[0][2] * o.v[2][0] + m.v[0][3] * o.v[3][0];
    if (std::fabs(det) < 1e-5f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 4; ++i) {
        o.v[i][1] = sgn[i] > 0 ? -k1[i] : k1[i];
        o.v[i][2] = sgn[i] > 0 ? k2[i] : -k2[i];
        o.v[i][3] = sgn[i] > 0 ? -k3[i] : k3[i];
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o.v[i][j] *= inv_det;
    return true;
}

struct Trs {
    Mat4 m, inv;
    bool invertible;
};

Trs compose(const float pos[3], const float q[4], const float scale[3]) {
    Mat4 S = identity();
    S.v[0][0] = scale[0]; S.v[1][1] = scale[1]; S.v[2][2] = scale[2];
    Trs t;
    t.m = multiply(rotation_of(q), S);
    t.m.v[0][3] = pos[0]; t.m.v[1][3] = pos[1]; t.m.v[2][3] = pos[2];
    t.inv = identity();
    t.invertible = invert(t.m, &t.inv);
    return t;
}

void store3x4(const Mat4& m, float out[12]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = m.v[i][j];
}

void point_by(const Mat4& m, const float p[3], float out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = m.v[i][0] * p[0] + m.v[i][1] * p[1] + m.v[i][2] * p[2] + m.v[i][3];
}

// ---------------------------------------------------------------------- BVH
struct Aabb {
    float lo[3], hi[3];
    Aabb() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
    }
    void grow(const float p[3]) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Aabb& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float half_area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0f;
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};

struct Prim {
    Aabb box;
    float c[3];
    uint32_t id;
};

struct TmpNode {
    Aabb box;
    int left = -1, right = -1;   // interior
    uint32_t first = 0, count = 0;   // leaf when count > 0
};

int balanced_height(size_t n, int leaf) {
    int h = 1;
    size_t leaves = (n + leaf - 1) / leaf;
    while (leaves > 1) {
        leaves = (leaves + 1) / 2;
        ++h;
    }
    return h;
}

// Most entries a ray can have on its traversal stack while inside the subtree of `ref` (kernels/trace.h trav_interior: a node
// whose k children are all hit leaves k - 1 of them on the stack while the first is visited, and any child can be the first).
// The 3-per-level bound assumes four children at every level of the deepest path; SAH trees of real meshes need about two
// thirds of it, and the megakernel's LDS stacks are sized by this number (gbl_api.hip: three workgroups per CU or two).
// GBL_STACK_LEVEL_BOUND=1 (a test aid): count three siblings at every node, i.e. the per-level bound the stacks had before.
static bool stack_level_bound() {
    static const bool on = [] { const char* e = getenv("GBL_STACK_LEVEL_BOUND"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }();
    return on;
}
static int blas_stack_need(const std::vector<DevNode>& nodes, int32_t ref) {
    if (ref < 0 || ref >= static_cast<int32_t>(nodes.size())) return 0;   // a leaf (or an analytic shape's root)
    const DevNode& n = nodes[static_cast<size_t>(ref)];
    int k = 0, deepest = 0;
    for (int c = 0; c < 4; ++c) {
        if (n.child[c] == GBL_REF_NONE) continue;
        ++k;
        deepest = std::max(deepest, blas_stack_need(nodes, n.child[c]));
    }
    return k > 0 ? (stack_level_bound() ? 3 : k - 1) + deepest : 0;
}

static int tlas_stack_need(const std::vector<DevNode>& tlas, int32_t tlas_base, int32_t ref, const std::vector<DevInstance>& instances,
                           const std::vector<int>& mesh_need) {
    if (ref == GBL_REF_NONE) return 0;
    if (ref < 0) {   // an instance: its sentinel, then its BLAS (an analytic shape's "root" is a leaf)
        const uint32_t i = (~static_cast<uint32_t>(ref)) >> 2;
        if (i >= instances.size()) return 1;
        const DevInstance& di = instances[i];
        const bool mesh = di.shape == 0u && di.mesh >= 0 && static_cast<size_t>(di.mesh) < mesh_need.size();
        return 1 + (mesh ? mesh_need[static_cast<size_t>(di.mesh)] : 0);
    }
    const size_t local = static_cast<size_t>(ref - tlas_base);
    if (local >= tlas.size()) return 0;
    const DevNode& n = tlas[local];
    int k = 0, deepest = 0;
    for (int c = 0; c < 4; ++c) {
        if (n.child[c] == GBL_REF_NONE) continue;
        ++k;
        deepest = std::max(deepest, tlas_stack_need(tlas, tlas_base, n.child[c], instances, mesh_need));
    }
    return k > 0 ? (stack_level_bound() ? 3 : k - 1) + deepest : 0;
}

// SAH cost of visiting a node relative to one triangle test (tuning knob: GBL_SAH_CT)
inline float sah_traversal_cost() {
    static const float ct = [] {
        const char* e = getenv("GBL_SAH_CT");
        return e ? static_cast<float>(atof(e)) : 1.0f;
    }();
    return ct;
}

struct Builder {
    std::vector<Prim>& prims;
    std::vector<TmpNode> nodes;
    int max_leaf, depth_cap, height = 0;
    // parallel build: subranges of at most `defer_below` primitives are not descended into but recorded as tasks
    struct Task {
        int node;
        size_t start, end;
        int depth;
    };
    size_t defer_below = 0;
    std::vector<Task> tasks;
    Builder(std::vector<Prim>& p, int leaf, int cap) : prims(p), max_leaf(leaf), depth_cap(cap) {}

    // Same tree as build(0, n, 1), built on several host threads: the top of the tree sequentially, every subtree of
    // <= n/64 primitives as an independent task (the tasks own disjoint ranges of `prims`), merged in task order --
    // so the result does not depend on the number of threads.
    int build_parallel(size_t n) {
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (n < 65536 || hw < 2) return build(0, n, 1);
        defer_below = std::max<size_t>(4096, n / 64);
        int root = build(0, n, 1);
        defer_below = 0;
        std::vector<Builder> subs;
        subs.reserve(tasks.size());
        for (size_t i = 0; i < tasks.size(); ++i) subs.emplace_back(prims, max_leaf, depth_cap);
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (size_t i = next.fetch_add(1); i < tasks.size(); i = next.fetch_add(1)) subs[i].build(tasks[i].start, tasks[i].end, tasks[i].depth);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < std::min<size_t>(hw, tasks.size()); ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
        for (size_t i = 0; i < tasks.size(); ++i) {
            const std::vector<TmpNode>& sn = subs[i].nodes;   // sn[0] is the subtree's root
            const int offset = static_cast<int>(nodes.size()) - 1;
            auto remap = [&](TmpNode nd) {
                if (nd.count == 0) {
                    nd.left += offset;
                    nd.right += offset;
                }
                return nd;
            };
            nodes[tasks[i].node] = remap(sn[0]);
            for (size_t k = 1; k < sn.size(); ++k) nodes.push_back(remap(sn[k]));
            height = std::max(height, subs[i].height);
        }
        tasks.clear();
        return root;
    }

    int build(size_t start, size_t end, int depth) {
        int me = static_cast<int>(nodes.size());
        nodes.push_back(TmpNode());
        Aabb box, cbox;
        for (size_t i = start; i < end; ++i) {
            box.grow(prims[i].box);
            cbox.grow(prims[i].c);
        }
        nodes[me].box = box;
        height = std::max(height, depth);
        size_t n = end - start;
        if (defer_below && depth > 1 && n <= defer_below && n > static_cast<size_t>(max_leaf)) {
            tasks.push_back(Task{me, start, end, depth});
            return me;
        }
        int room = depth_cap - depth;   // levels available below this node
        // binned SAH over the centroid bounds, all three axes
        // (tuning knobs like GBL_SAH_CT: GBL_SAH_BINS, and GBL_MAX_LEAF below.  On bunny.json 8 / 16 / 32 bins trace alike --
        //  51.3 / 51.3 / 51.1 ms -- while 64 bins or 2-triangle leaves add a BLAS level and with it the LDS for a third
        //  workgroup per CU: 65.6 / 66.1 ms)
        static const int B = [] { const char* e = getenv("GBL_SAH_BINS"); return e ? std::max(4, std::min(64, atoi(e))) : 16; }();
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        float parent_area = box.half_area();
        if (n > 1 && parent_area > 0.0f) {
            for (int axis = 0; axis < 3; ++axis) {
                float lo = cbox.lo[axis], ext = cbox.hi[axis] - cbox.lo[axis];
                if (!(ext > 0.0f)) continue;
                Aabb bb[64];
                size_t bn[64] = {0};
                float scale = B / ext;
                for (size_t i = start; i < end; ++i) {
                    int b = std::min(B - 1, static_cast<int>((prims[i].c[axis] - lo) * scale));
                    bb[b].grow(prims[i].box);
                    bn[b]++;
                }
                Aabb right_acc[64];
                size_t right_n[64];
                Aabb acc;
                size_t cnt = 0;
                for (int b = B - 1; b >= 1; --b) {
                    acc.grow(bb[b]);
                    cnt += bn[b];
                    right_acc[b] = acc;
                    right_n[b] = cnt;
                }
                Aabb left;
                size_t ln = 0;
                for (int b = 0; b < B - 1; ++b) {
                    left.grow(bb[b]);
                    ln += bn[b];
                    size_t rn = right_n[b + 1];
                    if (ln == 0 || rn == 0) continue;
                    if (balanced_height(std::max(ln, rn), max_leaf) > room) continue;   // would break the depth cap
                    float cost = sah_traversal_cost() + (left.half_area() * ln + right_acc[b + 1].half_area() * rn) / parent_area;
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_bin = b;
                    }
                }
            }
        }
        bool can_leaf = n <= static_cast<size_t>(max_leaf);
        if (can_leaf && (best_axis < 0 || static_cast<float>(n) <= best_cost)) {
            nodes[me].first = static_cast<uint32_t>(start);
            nodes[me].count = static_cast<uint32_t>(n);
            return me;
        }
        size_t mid;
        if (best_axis >= 0) {
            float lo = cbox.lo[best_axis], scale = B / (cbox.hi[best_axis] - cbox.lo[best_axis]);
            int axis = best_axis, bin = best_bin;
            auto it = std::partition(prims.begin() + start, prims.begin() + end, [&](const Prim& p) {
                return std::min(B - 1, static_cast<int>((p.c[axis] - lo) * scale)) <= bin;
            });
            mid = static_cast<size_t>(it - prims.begin());
        } else {
            // object median on the widest centroid axis (also the depth-cap fallback)
            int axis = 0;
            float e[3] = {cbox.hi[0] - cbox.lo[0], cbox.hi[1] - cbox.lo[1], cbox.hi[2] - cbox.lo[2]};
            if (e[1] > e[axis]) axis = 1;
            if (e[2] > e[axis]) axis = 2;
            mid = (start + end) / 2;
            std::nth_element(prims.begin() + start, prims.begin() + mid, prims.begin() + end,
                             [axis](const Prim& a, const Prim& b) { return a.c[axis] < b.c[axis]; });
        }
        if (mid == start || mid == end) mid = (start + end) / 2;
        int l = build(start, mid, depth + 1);
        int r = build(mid, end, depth + 1);
        nodes[me].left = l;
        nodes[me].right = r;
        return me;
    }
};

// Collapse the binary tree into 4-wide nodes and quantise the child boxes to 8 bits on a
// per-node power-of-two grid.  Boxes are first nudged outwards by a few ulps and then rounded
// outwards to the grid, so the device slab test (fused multiply-add form) can never cull a
// triangle the exact test would keep.
inline float nudge_down(float v) { return v - std::fabs(v) * 4e-7f - 1e-30f; }
inline float nudge_up(float v) { return v + std::fabs(v) * 4e-7f + 1e-30f; }

struct Flat4 {
    const Builder& b;
    std::vector<DevNode>& out;
    int32_t base;         // index out[0] has in the device node array
    int depth = 0;        // deepest 4-wide level reached (root = 1)
    Flat4(const Builder& bb, std::vector<DevNode>& o, int32_t base_index = 0) : b(bb), out(o), base(base_index) {}

    template <class LeafRef>
    int32_t emit(int node, int level, LeafRef& leaf_ref) {
        const TmpNode& r = b.nodes[node];
        depth = std::max(depth, level);
        if (r.count > 0) return leaf_ref(r.first, r.count);
        // gather up to 4 children: keep splitting the interior child with the largest area
        int kids[4] = {r.left, r.right, -1, -1};
        int n = 2;
        while (n < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (int i = 0; i < n; ++i) {
                const TmpNode& c = b.nodes[kids[i]];
                if (c.count == 0 && c.box.half_area() > best_area) {
                    best_area = c.box.half_area();
                    best = i;
                }
            }
            if (best < 0) break;
            const TmpNode& c = b.nodes[kids[best]];
            kids[best] = c.left;
            kids[n++] = c.right;
        }
        int32_t me = base + static_cast<int32_t>(out.size());
        out.push_back(DevNode());
        int32_t refs[4];
        for (int i = 0; i < 4; ++i) refs[i] = i < n ? emit(kids[i], level + 1, leaf_ref) : static_cast<int32_t>(GBL_REF_NONE);
        DevNode nd;
        memset(&nd, 0, sizeof(nd));
        // node bounds over the (nudged) children
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
            for (int i = 0; i < n; ++i) {
                lo[a] = std::min(lo[a], nudge_down(b.nodes[kids[i]].box.lo[a]));
                hi[a] = std::max(hi[a], nudge_up(b.nodes[kids[i]].box.hi[a]));
            }
        }
        for (int a = 0; a < 3; ++a) {
            nd.o[a] = lo[a];
            // grid step 2^e with 255 * 2^e >= extent (plus headroom for the rounding of o + q * step)
            float extent = (hi[a] - lo[a]) * 1.0001f + 1e-30f;
            int e = 0;
            std::frexp(extent / 255.0f, &e);   // extent/255 = m * 2^e, m in [0.5, 1)  ->  2^e >= extent/255
            int biased = std::min(254, std::max(1, e + 127));
            float step = std::ldexp(1.0f, biased - 127);
            nd.scale[a] = step;
            uint32_t ql = 0, qh = 0;
            for (int i = 0; i < 4; ++i) {
                uint32_t l = 255, h = 0;   // unused slot: inverted box, never hit
                if (i < n) {
                    float cl = nudge_down(b.nodes[kids[i]].box.lo[a]), ch = nudge_up(b.nodes[kids[i]].box.hi[a]);
                    double fl = std::floor((static_cast<double>(cl) - lo[a]) / step);
                    double fh = std::ceil((static_cast<double>(ch) - lo[a]) / step);
                    // guard the float evaluation o + q * step on the device against rounding inwards
                    while (fl > 0 && lo[a] + static_cast<float>(fl) * step > cl) fl -= 1;
                    while (fh < 255 && lo[a] + static_cast<float>(fh) * step < ch) fh += 1;
                    l = static_cast<uint32_t>(std::min(255.0, std::max(0.0, fl)));
                    h = static_cast<uint32_t>(std::min(255.0, std::max(0.0, fh)));
                }
                ql |= l << (8 * i);
                qh |= h << (8 * i);
            }
            nd.qlo[a] = ql;
            nd.qhi[a] = qh;
        }
        for (int i = 0; i < 4; ++i) nd.child[i] = refs[i];
        out[me - base] = nd;
        return me;
    }
};

// -------------------------------------------------------------------- filter
struct Filter {
    uint32_t type;
    float wx, wy, alpha, ex, ey, b, c;
    float gauss(float v, float base) const { return std::max(0.0f, expf(-alpha * v * v) - base); }
    float mitchell1(float x) const {
        x = std::fabs(2.0f * x);
        if (x > 1.0f) return ((-b - 6 * c) * x * x * x + (6 * b + 30 * c) * x * x + (-12 * b - 48 * c) * x + (8 * b + 24 * c)) / 6.0f;
        return ((12 - 9 * b - 6 * c) * x * x * x + (-18 + 12 * b + 6 * c) * x * x + (6 - 2 * b)) / 6.0f;
    }
    float eval(float x, float y) const {
        if (type == GBL_FILTER_BOX) return 1.0f;
        if (type == GBL_FILTER_TRIANGLE) return std::max(0.0f, wx - fabsf(x)) * std::max(0.0f, wy - fabsf(y));
        if (type == GBL_FILTER_MITCHELL) return mitchell1(x * (1.0f / wx)) * mitchell1(y * (1.0f / wy));
        return gauss(x, ex) * gauss(y, ey);
    }
    float norm() const {
        if (type == GBL_FILTER_BOX) return 4.0f * wx * wy;
        if (type == GBL_FILTER_TRIANGLE) return wx * wx * wy * wy;
        if (type == GBL_FILTER_MITCHELL)
            return 4.0f * ((12 - 9 * b - 6 * c) / 4 + (-18 + 12 * b + 6 * c) / 3 + (6 - 2 * b) + 15 * (-b - 6 * b) / 4 +
                           7 * (6 * b + 30 * c) / 3 + 3 * (-12 * b - 48 * c) / 2 + (8 * b + 24 * c)) / 6.0f;
        const size_t steps = 20;   // the reference integrates the gaussian numerically (GoblinFilter.cpp:48-64)
        float dx = wx / static_cast<float>(steps), dy = wy / static_cast<float>(steps), sum = 0.0f;
        for (size_t i = 0; i < steps; ++i)
            for (size_t j = 0; j < steps; ++j) sum += 4.0f * dx * dy * gauss(i * dx, ex) * gauss(j * dy, ey);
        return sum;
    }
};

// Levels below a material slot: a constant is 0, a checkerboard / scale of constants is 1, ...  -1: bad index or cycle.
int texture_depth(const gbl_scene_desc* d, int32_t id, int guard) {
    if (id < 0 || static_cast<uint32_t>(id) >= d->num_textures || guard > 64) return -1;
    const gbl_texture& g = d->textures[id];
    if (g.type == GBL_TEX_CONSTANT) return 0;
    if (g.type == GBL_TEX_IMAGE) return 1;   // a leaf with a lookup of its own
    int a = texture_depth(d, g.child[0], guard + 1), b = texture_depth(d, g.child[1], guard + 1);
    if (a < 0 || b < 0) return -1;
    return 1 + std::max(a, b);
}

int ceil_i(float f) { return static_cast<int>(std::ceil(f)); }

}  // namespace

int scene_stack_entries(const std::vector<DevNode>& tlas, int32_t tlas_base, int32_t tlas_root, const std::vector<DevInstance>& instances,
                        const std::vector<int>& mesh_stack_need) {
    return 1 + tlas_stack_need(tlas, tlas_base, tlas_root, instances, mesh_stack_need) + 1;
}

// Instances (transforms in the reference's float order, world boxes by Transform::onBBox) and the TLAS over them.
// TLAS node k gets device index tlas_base + k; sb_lo/hi return the union of the instance boxes.
gbl_status build_tlas(const gbl_instance* inst, uint32_t n, const gbl_mesh* meshes, const gbl_material* materials, const float* mesh_lo,
                      const float* mesh_hi, const int32_t* mesh_root, int32_t tlas_base, std::vector<DevInstance>* out_inst,
                      std::vector<DevNode>* out_nodes, int32_t* tlas_root, int* tlas_depth, float sb_lo[3], float sb_hi[3], std::string* err,
                      std::vector<DevInstanceBound>* bounds_out) {
    const int kTlasCap = 22;
    if (bounds_out) bounds_out->clear();
    out_inst->resize(n);
    out_nodes->clear();
    std::vector<Prim> iprims(n);
    for (uint32_t i = 0; i < n; ++i) {
        const gbl_instance& gi = inst[i];
        Trs t = compose(gi.to_world.position, gi.to_world.orientation, gi.to_world.scale);
        if (!t.invertible) {
            *err = "instance " + std::to_string(i) + ": |det(toWorld)| < 1e-5, the reference cannot invert this transform "
                   "(GoblinMatrix.cpp:451); use a uniform scale >= 0.0216";
            return GBL_ERR_INVALID;
        }
        DevInstance& di = (*out_inst)[i];
        memset(&di, 0, sizeof(di));
        store3x4(t.m, di.m);
        store3x4(t.inv, di.inv);
        di.root = mesh_root[gi.mesh];
        di.material = static_cast<int32_t>(gi.material);
        di.area_light = gi.area_light;
        di.mesh = static_cast<int32_t>(gi.mesh);
        di.shape = meshes[gi.mesh].shape;
        di.radius = meshes[gi.mesh].radius;
        // MaskMaterial ORs BSDFnullptr into its type; SubsurfaceMaterial's type is BSDFAll, which holds the bit as well
        di.is_mask = (materials[gi.material].type == GBL_MAT_MASK || materials[gi.material].type == GBL_MAT_SUBSURFACE) ? 1u : 0u;
        // Transform::onBBox: the 8 corners of the mesh bound
        const float* lo = mesh_lo + 3 * gi.mesh;
        const float* hi = mesh_hi + 3 * gi.mesh;
        Prim& p = iprims[i];
        for (int c = 0; c < 8; ++c) {
            float corner[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
            float w[3];
            point_by(t.m, corner, w);
            p.box.grow(w);
        }
        for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        p.id = i;
        if (bounds_out) {
            DevInstanceBound wb;
            for (int k = 0; k < 3; ++k) {
                wb.lo[k] = p.box.lo[k];
                wb.hi[k] = p.box.hi[k];
            }
            bounds_out->push_back(wb);
        }
    }
    Aabb scene_bound;
    for (const Prim& p : iprims) {
        scene_bound.grow(p.box.lo);
        scene_bound.grow(p.box.hi);
    }
    for (int k = 0; k < 3; ++k) {
        sb_lo[k] = scene_bound.lo[k];
        sb_hi[k] = scene_bound.hi[k];
    }
    *tlas_depth = 0;
    *tlas_root = 0;
    if (n > 0) {
        Builder tb(iprims, 1, kTlasCap);
        int root = tb.build(0, iprims.size(), 1);
        Flat4 t4(tb, *out_nodes, tlas_base);
        auto inst_ref = [&](uint32_t first, uint32_t) { return ~static_cast<int32_t>(iprims[first].id << 2); };
        *tlas_root = t4.emit(root, 1, inst_ref);
        *tlas_depth = t4.depth;
    }
    return GBL_OK;
}

namespace {
int floor_i(float f) { return static_cast<int>(std::floor(f)); }

}  // namespace

// BVH::buildLinearBVH (GoblinBVH.cpp:81-151) with the EqualCount split, followed only as far as the SHAPE of the tree:
// per triangle the root-to-leaf path, the split axes along it and its place in a multi-triangle leaf.
namespace {
struct RefItem {   // BVHPrimitiveInfo (:8-14)
    float lo[3], hi[3], c[3];
    uint32_t id;
};
void reference_order_rec(std::vector<RefItem>& it, uint32_t start, uint32_t end, uint32_t depth, uint32_t path, uint64_t axes,
                         DevTriOrder* out) {
    auto leaf = [&]() {
        for (uint32_t i = start; i < end; ++i) {
            DevTriOrder& o = out[it[i].id];
            o.path = path;
            o.axes_lo = static_cast<uint32_t>(axes);
            o.axes_hi = static_cast<uint32_t>(axes >> 32);
            o.depth_rank = depth | ((i - start) << 8);
        }
    };
    if (end - start == 1 || depth >= 32) return leaf();
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = start; i < end; ++i)
        for (int k = 0; k < 3; ++k) {
            clo[k] = std::min(clo[k], it[i].c[k]);
            chi[k] = std::max(chi[k], it[i].c[k]);
        }
    const float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
    const int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);   // BBox::longestAxis, GoblinBBox.cpp:79-88
    if (clo[dim] == chi[dim]) return leaf();
    const uint32_t mid = (start + end) / 2;
    std::nth_element(&it[start], &it[mid], &it[end - 1] + 1, [dim](const RefItem& a, const RefItem& b) { return a.c[dim] < b.c[dim]; });
    axes |= static_cast<uint64_t>(dim) << (2 * depth);
    reference_order_rec(it, start, mid, depth + 1, path, axes, out);
    reference_order_rec(it, mid, end, depth + 1, path | (1u << depth), axes, out);
}
void reference_order(const float* P, const uint32_t* I, uint32_t n, DevTriOrder* out) {
    std::vector<RefItem> it(n);
    for (uint32_t t = 0; t < n; ++t) {
        RefItem& r = it[t];
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = INFINITY;
            r.hi[k] = -INFINITY;
        }
        for (int v = 0; v < 3; ++v) {   // Triangle::getObjectBound: the three vertices' bound
            const float* p = P + 3 * I[3 * t + v];
            for (int k = 0; k < 3; ++k) {
                r.lo[k] = std::min(r.lo[k], p[k]);
                r.hi[k] = std::max(r.hi[k], p[k]);
            }
        }
        for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (r.lo[k] + r.hi[k]);
        r.id = t;
    }
    reference_order_rec(it, 0, n, 0, 0u, 0ull, out);
}
}   // namespace

gbl_status pack_scene(const gbl_scene_desc* d, PackedScene* out, std::string* err, bool device_blas) {
    if (!d || d->abi_version != GBL_ABI_VERSION) {
        *err = "scene description has the wrong abi_version";
        return GBL_ERR_INVALID;
    }
    out->extended = (d->camera.lens_radius != 0.0f || d->camera.type != GBL_CAMERA_PERSPECTIVE) ? 1 : 0;
    if (d->camera.type > GBL_CAMERA_ORTHOGRAPHIC) {
        *err = "unknown camera type";
        return GBL_ERR_INVALID;
    }
    for (uint32_t i = 0; i < d->num_instances; ++i) {
        if (d->instances[i].mesh >= d->num_meshes || d->instances[i].material >= d->num_materials ||
            d->instances[i].area_light >= static_cast<int32_t>(d->num_lights)) {
            *err = "instance " + std::to_string(i) + " references a mesh/material/light out of range";
            return GBL_ERR_INVALID;
        }
    }
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        const gbl_mesh& m = d->meshes[i];
        if (m.shape == GBL_SHAPE_SPHERE || m.shape == GBL_SHAPE_DISK) {
            out->extended = 1;
            continue;
        }
        if (m.shape != GBL_SHAPE_MESH) {
            *err = "mesh " + std::to_string(i) + " has an unknown shape";
            return GBL_ERR_INVALID;
        }
        if (m.tri_count == 0 || static_cast<uint64_t>(m.vertex_offset) + m.vertex_count > d->num_vertices ||
            static_cast<uint64_t>(m.tri_offset) + m.tri_count > d->num_triangles) {
            *err = "mesh " + std::to_string(i) + " is empty or out of range";
            return GBL_ERR_INVALID;
        }
        for (uint32_t t = 0; t < 3 * m.tri_count; ++t)
            if (d->indices[3 * static_cast<size_t>(m.tri_offset) + t] >= m.vertex_count) {
                *err = "mesh " + std::to_string(i) + " has a vertex index out of range";
                return GBL_ERR_INVALID;
            }
        if (m.has_uv) {
            // A triangle whose uv determinant is 0 makes the reference build its tangent from
            // whatever Fragment the caller passed in (GoblinTriangle.cpp:113-117): not reproducible.
            const float* uv = d->uvs + 2 * static_cast<size_t>(m.vertex_offset);
            const uint32_t* I = d->indices + 3 * static_cast<size_t>(m.tri_offset);
            for (uint32_t t = 0; t < m.tri_count; ++t) {
                const float* a = uv + 2 * I[3 * t];
                const float* b = uv + 2 * I[3 * t + 1];
                const float* c = uv + 2 * I[3 * t + 2];
                float du1 = b[0] - a[0], dv1 = b[1] - a[1], du2 = c[0] - a[0], dv2 = c[1] - a[1];
                if (du1 * dv2 - dv1 * du2 == 0.0f) {
                    *err = "mesh " + std::to_string(i) + " triangle " + std::to_string(t) +
                           " has degenerate texture coordinates (stale-fragment branch of the reference)";
                    return GBL_ERR_UNSUPPORTED;
                }
            }
        }
    }

    // ---- vertex attributes used at shading time
    out->positions.assign(d->positions, d->positions + 3 * static_cast<size_t>(d->num_vertices));
    out->normals.assign(d->normals, d->normals + 3 * static_cast<size_t>(d->num_vertices));
    out->uvs.assign(d->uvs, d->uvs + 2 * static_cast<size_t>(d->num_vertices));

    // ---- the reference's visiting order of every mesh's triangles (DevTriOrder)
    out->tri_order.assign(d->num_triangles, DevTriOrder());
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi) {
        const gbl_mesh& gm = d->meshes[mi];
        if (gm.shape != GBL_SHAPE_MESH || gm.tri_count == 0) continue;
        reference_order(d->positions + 3 * static_cast<size_t>(gm.vertex_offset), d->indices + 3 * static_cast<size_t>(gm.tri_offset),
                        gm.tri_count, out->tri_order.data() + gm.tri_offset);
    }

    // ---- every triangle's own bound (the reference BLAS's leaf boxes, DevTriBound)
    out->tri_bounds.assign(d->num_triangles, DevTriBound());
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi) {
        const gbl_mesh& gm = d->meshes[mi];
        if (gm.shape != GBL_SHAPE_MESH) continue;
        const float* P = d->positions + 3 * static_cast<size_t>(gm.vertex_offset);
        const uint32_t* I = d->indices + 3 * static_cast<size_t>(gm.tri_offset);
        for (uint32_t t = 0; t < gm.tri_count; ++t) {
            DevTriBound& b = out->tri_bounds[gm.tri_offset + t];
            memset(&b, 0, sizeof(b));
            const float *a = P + 3 * I[3 * t], *bb = P + 3 * I[3 * t + 1], *c = P + 3 * I[3 * t + 2];
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = std::min(std::min(a[k], bb[k]), c[k]);
                b.hi[k] = std::max(std::max(a[k], bb[k]), c[k]);
            }
        }
    }

    // ---- one BLAS per mesh
    const int kBlasCap = 40;
    std::vector<int32_t> mesh_root(d->num_meshes);
    std::vector<Aabb> mesh_bounds(d->num_meshes);
    out->tris.clear();
    out->tri_shade.assign(d->num_triangles, DevTriShade());
    out->blas_max_depth = 0;
    out->mesh_stack_need.assign(d->num_meshes, 0);
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi) {
        const gbl_mesh& gm = d->meshes[mi];
        if (gm.shape != GBL_SHAPE_MESH) {
            // Sphere / Disk::getObjectBound (GoblinSphere.cpp:140-143, GoblinDisk.cpp:81-84); no BLAS: the
            // instance's root is a marker reference and the traversal tests the shape analytically
            const float r = gm.radius, z = gm.shape == GBL_SHAPE_SPHERE ? gm.radius : 0.0f;
            float hi[3] = {r, r, z}, lo[3] = {-r, -r, -z};
            mesh_bounds[mi].grow(hi);
            mesh_bounds[mi].grow(lo);
            mesh_root[mi] = gm.shape == GBL_SHAPE_SPHERE ? GBL_REF_SPHERE : GBL_REF_DISK;
            continue;
        }
        const float* P = d->positions + 3 * static_cast<size_t>(gm.vertex_offset);
        const uint32_t* I = d->indices + 3 * static_cast<size_t>(gm.tri_offset);
        for (uint32_t v = 0; v < gm.vertex_count; ++v) mesh_bounds[mi].grow(P + 3 * v);
        if (device_blas) {   // shading records only; the tree and the DevTri order come from kernels/lbvh.h
            for (uint32_t t = 0; t < gm.tri_count; ++t) {
                DevTriShade& s = out->tri_shade[gm.tri_offset + t];
                for (int k = 0; k < 3; ++k) s.v[k] = gm.vertex_offset + I[3 * t + k];
                s.flags = (gm.has_normal ? 1u : 0u) | (gm.has_uv ? 2u : 0u);
            }
            mesh_root[mi] = 0;
            continue;
        }
        std::vector<Prim> prims(gm.tri_count);
        for (uint32_t t = 0; t < gm.tri_count; ++t) {
            Prim& p = prims[t];
            for (int k = 0; k < 3; ++k) p.box.grow(P + 3 * I[3 * t + k]);
            for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
            p.id = t;
            DevTriShade& s = out->tri_shade[gm.tri_offset + t];
            for (int k = 0; k < 3; ++k) s.v[k] = gm.vertex_offset + I[3 * t + k];
            s.flags = (gm.has_normal ? 1u : 0u) | (gm.has_uv ? 2u : 0u);
        }
        const int leaf_knob = [] { const char* e = getenv("GBL_MAX_LEAF"); return e ? std::max(1, std::min(GBL_MAX_LEAF_TRIS, atoi(e))) : GBL_MAX_LEAF_TRIS; }();
        Builder b(prims, leaf_knob, kBlasCap);
        int root = b.build_parallel(prims.size());
        Flat4 f4(b, out->nodes);
        uint32_t tri_base = static_cast<uint32_t>(out->tris.size());
        for (const Prim& p : prims) {
            const float* p0 = P + 3 * I[3 * p.id];
            const float* p1 = P + 3 * I[3 * p.id + 1];
            const float* p2 = P + 3 * I[3 * p.id + 2];
            DevTri t;
            memset(&t, 0, sizeof(t));
            for (int k = 0; k < 3; ++k) {
                t.p0[k] = p0[k];
                t.e1[k] = p1[k] - p0[k];
                t.e2[k] = p2[k] - p0[k];
            }
            t.shade = gm.tri_offset + p.id;
            t.flags = (gm.has_normal ? 1u : 0u) | (gm.has_uv ? 2u : 0u);
            out->tris.push_back(t);
        }
        auto leaf_ref = [&](uint32_t first, uint32_t count) {
            return ~static_cast<int32_t>(((tri_base + first) << 2) | (count - 1));
        };
        mesh_root[mi] = f4.emit(root, 1, leaf_ref);
        out->blas_max_depth = std::max(out->blas_max_depth, f4.depth);
        out->mesh_stack_need[mi] = blas_stack_need(out->nodes, mesh_root[mi]);
    }
    out->blas_nodes = out->nodes.size();
    out->mesh_lo.resize(3 * d->num_meshes);
    out->mesh_hi.resize(3 * d->num_meshes);
    for (uint32_t mi = 0; mi < d->num_meshes; ++mi)
        for (int k = 0; k < 3; ++k) {
            out->mesh_lo[3 * mi + k] = mesh_bounds[mi].lo[k];
            out->mesh_hi[3 * mi + k] = mesh_bounds[mi].hi[k];
        }

    // ---- instances + TLAS
    out->mesh_root = mesh_root;
    out->tlas_base = static_cast<int32_t>(out->nodes.size());
    out->tlas_capacity = std::max<uint32_t>(1u, d->num_instances);
    std::vector<DevNode> tlas;
    float sb_lo[3], sb_hi[3];
    {
        gbl_status ts = build_tlas(d->instances, d->num_instances, d->meshes, d->materials, out->mesh_lo.data(), out->mesh_hi.data(),
                                   mesh_root.data(), out->tlas_base, &out->instances, &tlas, &out->tlas_root, &out->tlas_depth, sb_lo, sb_hi, err,
                                   &out->instance_bounds);
        if (ts != GBL_OK) return ts;
    }
    for (const DevInstance& di : out->instances)
        if (di.is_mask) out->has_masks = out->extended = 1;
    Aabb scene_bound;   // BVH::getAABB of the scene BVH: the union of the instance boxes (GoblinBVH.cpp:46-50)
    scene_bound.grow(sb_lo);
    scene_bound.grow(sb_hi);
    out->nodes.insert(out->nodes.end(), tlas.begin(), tlas.end());
    out->tlas_nodes = tlas.size();
    // room for any TLAS over the same instances (gbl_update_instances rebuilds it in place)
    out->nodes.resize(static_cast<size_t>(out->tlas_base) + out->tlas_capacity, DevNode());
    // (with device-built BLASes the caller fills mesh_stack_need from their depths and calls scene_stack_entries again)
    out->stack_entries = scene_stack_entries(tlas, out->tlas_base, out->tlas_root, out->instances, out->mesh_stack_need);

    // ---- materials
    out->materials.resize(d->num_materials);
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const gbl_material& m = d->materials[i];
        if (m.type > GBL_MAT_SUBSURFACE) {
            *err = "unknown material type";
            return GBL_ERR_INVALID;
        }
        if (m.type == GBL_MAT_MASK && (m.masked_material < 0 || static_cast<uint32_t>(m.masked_material) >= d->num_materials ||
                                       d->materials[m.masked_material].type == GBL_MAT_MASK ||
                                       d->materials[m.masked_material].type == GBL_MAT_SUBSURFACE)) {
            *err = "mask material " + std::to_string(i) + " must wrap a non-mask, non-subsurface material of the scene";
            return GBL_ERR_INVALID;
        }
        DevMaterial& dm = out->materials[i];
        memset(&dm, 0, sizeof(dm));
        dm.type = m.type;
        for (int k = 0; k < 3; ++k) {
            dm.color[k] = m.color[k];
            dm.color2[k] = m.color2[k];
        }
        dm.index = m.index;
        dm.k = m.k;
        dm.exponent = m.exponent;
        dm.tex_color = m.tex_color;
        dm.tex_color2 = m.tex_color2;
        dm.tex_exponent = m.tex_exponent;
        dm.has_tex = (m.tex_color >= 0 || m.tex_color2 >= 0 || m.tex_exponent >= 0) ? 1u : 0u;
        dm.masked = m.type == GBL_MAT_MASK ? m.masked_material : -1;
        dm.tex_color3 = -1;
        // Material::perturb -> BumpShaders::evaluate; MaskMaterial::perturb forwards to the wrapped material (GoblinMaterial.h:456-458)
        const gbl_material& bump_of = m.type == GBL_MAT_MASK ? d->materials[m.masked_material] : m;
        dm.tex_bump = bump_of.tex_bump;
        dm.tex_normal = bump_of.tex_normal;
        if (dm.tex_bump >= 0 || dm.tex_normal >= 0) dm.has_tex = 1u;
        if (m.type == GBL_MAT_SUBSURFACE) {
            for (int k = 0; k < 3; ++k) dm.color3[k] = m.color3[k];
            dm.tex_color3 = m.tex_color3;
            dm.has_tex = 1u;   // its fragments always carry dpdu / dpdv: BSSRDF::sampleProbeRay and MISWeight read them
            // BSSRDF ctor (GoblinMaterial.cpp:32-38): A from the diffuse Fresnel reflectance polynomial (GoblinMaterial.h:94-105)
            const float eta = m.index;
            const float fdr = eta < 1.0f ? -0.4399f + 0.7099f / eta - 0.3319f / (eta * eta) + 0.0636f / (eta * eta * eta)
                                         : -1.4399f / (eta * eta) + 0.7099f / eta + 0.6681f + 0.0636f * eta;
            dm.exponent = (1.0f + fdr) / (1.0f - fdr);
            out->has_bssrdf = out->extended = 1;
        }
        if (m.type == GBL_MAT_MASK) {
            const gbl_material& in = d->materials[m.masked_material];
            if (in.tex_color >= 0 || in.tex_color2 >= 0 || in.tex_exponent >= 0) dm.has_tex = 1u;
        }
        for (int32_t t : {m.tex_color, m.tex_color2, m.tex_exponent, m.type == GBL_MAT_SUBSURFACE ? m.tex_color3 : -1, m.tex_bump, m.tex_normal}) {
            if (t < 0) continue;
            out->extended = 1;
            int depth = texture_depth(d, t, 0);
            if (depth < 0) {
                *err = "material " + std::to_string(i) + " references a texture out of range (or a cyclic texture graph)";
                return GBL_ERR_INVALID;
            }
            if (depth > GBL_TEX_MAX_DEPTH) {
                *err = "material " + std::to_string(i) + ": texture graph deeper than " + std::to_string(GBL_TEX_MAX_DEPTH) +
                       " levels below the material slot is outside the device path";
                return GBL_ERR_UNSUPPORTED;
            }
        }
    }

    // ---- images: the pyramids arrive built (gbl_image); the device wants every level's offset
    out->images.resize(d->num_images);
    for (uint32_t i = 0; i < d->num_images; ++i) {
        const gbl_image& gi = d->images[i];
        DevImage& di = out->images[i];
        memset(&di, 0, sizeof(di));
        if (gi.width == 0 || gi.height == 0 || (gi.width & (gi.width - 1)) || (gi.height & (gi.height - 1)) || (gi.channels != 1 && gi.channels != 4) ||
            gi.levels == 0 || gi.levels > 18) {
            *err = "image " + std::to_string(i) + ": sides must be powers of two, channels 1 or 4, at most 18 levels";
            return GBL_ERR_INVALID;
        }
        di.width = gi.width;
        di.height = gi.height;
        di.levels = gi.levels;
        di.channels = gi.channels;
        di.offset = gi.texel_offset;
        uint64_t off = 0;
        for (uint32_t l = 0; l < gi.levels; ++l) {
            di.level_offset[l] = static_cast<uint32_t>(off);
            off += static_cast<uint64_t>(std::max(1u, gi.width >> l)) * std::max(1u, gi.height >> l) * gi.channels;
        }
        if (gi.texel_offset + off > d->num_texels || off >= (1ull << 32)) {
            *err = "image " + std::to_string(i) + ": texels out of range";
            return GBL_ERR_INVALID;
        }
    }
    // MIPMap<T>::initEWALut (GoblinTexture.cpp:262-271)
    out->ewa_lut.resize(128);
    for (int i = 0; i < 128; ++i) {
        const float r2 = static_cast<float>(i) / static_cast<float>(128 - 1);
        out->ewa_lut[i] = expf(-2.0f * r2) - expf(-2.0f);
    }

    // ---- textures
    out->textures.resize(d->num_textures);
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const gbl_texture& g = d->textures[i];
        if (g.type > GBL_TEX_IMAGE || g.mapping > GBL_MAP_SPHERICAL) {
            *err = "unknown texture or mapping type";
            return GBL_ERR_INVALID;
        }
        if (g.type == GBL_TEX_IMAGE && (g.image < 0 || static_cast<uint32_t>(g.image) >= d->num_images || g.image_filter > GBL_IMAGE_FILTER_EWA ||
                                        g.address > GBL_ADDRESS_BORDER || d->images[g.image].channels != (g.is_float ? 1u : 4u))) {
            *err = "image texture " + std::to_string(i) + ": bad image index, filter, address mode or channel count";
            return GBL_ERR_INVALID;
        }
        DevTexture& t = out->textures[i];
        memset(&t, 0, sizeof(t));
        t.type = g.type;
        t.is_float = g.is_float;
        for (int k = 0; k < 3; ++k) t.value[k] = g.is_float ? g.value[0] : g.value[k];
        t.child[0] = g.child[0];
        t.child[1] = g.child[1];
        t.mapping = g.mapping;
        t.filter = g.type == GBL_TEX_IMAGE ? g.image_filter : g.filter;
        t.image = g.type == GBL_TEX_IMAGE ? g.image : -1;
        t.address = g.address;
        t.max_aniso = g.max_anisotropy;
        for (int k = 0; k < 2; ++k) {
            t.uv_scale[k] = g.uv_scale[k];
            t.uv_offset[k] = g.uv_offset[k];
        }
        if ((g.type == GBL_TEX_CHECKERBOARD || g.type == GBL_TEX_IMAGE) && g.mapping == GBL_MAP_SPHERICAL) {
            Trs tt = compose(g.to_tex.position, g.to_tex.orientation, g.to_tex.scale);   // SphericalMapping::mToTex
            store3x4(tt.m, t.to_tex);
        }
    }

    // ---- lights, power distribution (Scene ctor, CDF1D::init)
    out->lights.resize(d->num_lights);
    out->light_tris.clear();
    std::vector<float> power(d->num_lights);
    for (uint32_t i = 0; i < d->num_lights; ++i) {
        const gbl_light& gl = d->lights[i];
        DevLight& dl = out->lights[i];
        memset(&dl, 0, sizeof(dl));
        dl.type = gl.type;
        for (int k = 0; k < 3; ++k) {
            dl.color[k] = gl.color[k];
            dl.pos[k] = gl.position[k];
        }
        dl.cos_max = gl.cos_theta_max;
        dl.cos_falloff = gl.cos_falloff_start;
        float pr, pg, pb;
        if (gl.type == GBL_LIGHT_SPOT || gl.type == GBL_LIGHT_DIRECTIONAL) {
            // Light::setOrientation (GoblinLight.cpp:66-76): direction -> basis -> quaternion -> matrix; the axis
            // the light works with is that matrix's third column.  The spot light's ctor normalises the direction,
            // the directional light's does not (:136-143).
            float dir[3] = {gl.direction[0], gl.direction[1], gl.direction[2]};
            if (gl.type == GBL_LIGHT_SPOT) {
                float inv_len = 1.0f / std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                for (int k = 0; k < 3; ++k) dir[k] *= inv_len;
            }
            float xa[3], ya[3];
            if (fabsf(dir[0]) > fabsf(dir[1])) {
                float il = 1.0f / sqrtf(dir[0] * dir[0] + dir[2] * dir[2]);
                xa[0] = -dir[2] * il; xa[1] = 0.0f; xa[2] = dir[0] * il;
            } else {
                float il = 1.0f / sqrtf(dir[1] * dir[1] + dir[2] * dir[2]);
                xa[0] = 0.0f; xa[1] = -dir[2] * il; xa[2] = dir[1] * il;
            }
            ya[0] = dir[1] * xa[2] - dir[2] * xa[1];
            ya[1] = dir[2] * xa[0] - dir[0] * xa[2];
            ya[2] = dir[0] * xa[1] - dir[1] * xa[0];
            float R[3][3] = {{xa[0], ya[0], dir[0]}, {xa[1], ya[1], dir[1]}, {xa[2], ya[2], dir[2]}};
            float qv[4];   // x y z w
            float trace = R[0][0] + R[1][1] + R[2][2];
            if (trace > 0.0f) {
                float s = std::sqrt(trace + 1.0f);
                qv[3] = s * 0.5f;
                float t = 0.5f / s;
                qv[0] = (R[2][1] - R[1][2]) * t;
                qv[1] = (R[0][2] - R[2][0]) * t;
                qv[2] = (R[1][0] - R[0][1]) * t;
            } else {
                int a = 0;
                if (R[1][1] > R[0][0]) a = 1;
                if (R[2][2] > R[a][a]) a = 2;
                int b2 = (a + 1) % 3, c2 = (b2 + 1) % 3;
                float s = std::sqrt(R[a][a] - R[b2][b2] - R[c2][c2] + 1.0f);
                qv[a] = s * 0.5f;
                float t = s != 0.0f ? 0.5f / s : s;
                qv[3] = (R[c2][b2] - R[b2][c2]) * t;
                qv[b2] = (R[b2][a] + R[a][b2]) * t;
                qv[c2] = (R[c2][a] + R[a][c2]) * t;
            }
            float q[4] = {qv[3], qv[0], qv[1], qv[2]};
            float one[3] = {1.0f, 1.0f, 1.0f}, zero[3] = {0.0f, 0.0f, 0.0f};
            Trs t = compose(gl.type == GBL_LIGHT_SPOT ? gl.position : zero, q, one);
            for (int k = 0; k < 3; ++k) dl.axis[k] = t.m.v[k][0] * 0.0f + t.m.v[k][1] * 0.0f + t.m.v[k][2] * 1.0f;
            if (gl.type == GBL_LIGHT_SPOT) {
                float solid = kTwoPi;
                float f = (1.0f - 0.5f * (dl.cos_max + dl.cos_falloff));
                pr = dl.color[0] * solid * f; pg = dl.color[1] * solid * f; pb = dl.color[2] * solid * f;
            } else {
                // DirectionalLight::power (:203-210): radius^2 * PI * radiance over Scene::getBoundingSphere, whose
                // radius is the full diagonal of the scene bound (GoblinBBox.h:51-54)
                out->extended = 1;
                float dx = scene_bound.hi[0] - scene_bound.lo[0], dy = scene_bound.hi[1] - scene_bound.lo[1],
                      dz = scene_bound.hi[2] - scene_bound.lo[2];
                float radius = std::sqrt(dx * dx + dy * dy + dz * dz);
                float a = radius * radius * kPi;
                pr = a * dl.color[0]; pg = a * dl.color[1]; pb = a * dl.color[2];
            }
        } else if (gl.type == GBL_LIGHT_AREA && gl.mesh >= d->num_meshes) {
            *err = "area light references a mesh out of range";
            return GBL_ERR_INVALID;
        } else if (gl.type == GBL_LIGHT_AREA && d->meshes[gl.mesh].shape != GBL_SHAPE_MESH) {
            // GeometrySet over one intersectable geometry (GoblinLight.cpp:289-303)
            out->extended = 1;
            Trs t = compose(gl.to_world.position, gl.to_world.orientation, gl.to_world.scale);
            store3x4(t.m, dl.m);
            store3x4(t.inv, dl.inv);
            const gbl_mesh& gm = d->meshes[gl.mesh];
            dl.shape = gm.shape;
            dl.radius = gm.radius;
            float a = gm.shape == GBL_SHAPE_SPHERE ? 4.0f * kPi * gm.radius * gm.radius : kPi * gm.radius * gm.radius;
            float sum = 0.0f;
            sum += a;
            dl.sum_area = sum;
            dl.tri_first = dl.tri_count = 0;
            float world_area = sum * (gl.to_world.scale[0] * gl.to_world.scale[1]);
            pr = dl.color[0] * kPi * world_area; pg = dl.color[1] * kPi * world_area; pb = dl.color[2] * kPi * world_area;
        } else if (gl.type == GBL_LIGHT_AREA) {
            if (gl.mesh >= d->num_meshes) {
                *err = "area light references a mesh out of range";
                return GBL_ERR_INVALID;
            }
            Trs t = compose(gl.to_world.position, gl.to_world.orientation, gl.to_world.scale);
            store3x4(t.m, dl.m);
            store3x4(t.inv, dl.inv);
            const gbl_mesh& gm = d->meshes[gl.mesh];
            const float* P = d->positions + 3 * static_cast<size_t>(gm.vertex_offset);
            const float* N = d->normals + 3 * static_cast<size_t>(gm.vertex_offset);
            const uint32_t* I = d->indices + 3 * static_cast<size_t>(gm.tri_offset);
            dl.tri_first = static_cast<uint32_t>(out->light_tris.size());
            dl.tri_count = gm.tri_count;
            std::vector<float> areas(gm.tri_count);
            float sum = 0.0f;
            for (uint32_t k = 0; k < gm.tri_count; ++k) {
                const float* p0 = P + 3 * I[3 * k];
                const float* p1 = P + 3 * I[3 * k + 1];
                const float* p2 = P + 3 * I[3 * k + 2];
                float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                float e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                areas[k] = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);
                sum += areas[k];
            }
            dl.sum_area = sum;
            // CDF1D over the triangle areas
            float dx = 1.0f / gm.tri_count;
            std::vector<float> cdf(gm.tri_count + 1, 0.0f);
            for (uint32_t k = 1; k <= gm.tri_count; ++k) cdf[k] = cdf[k - 1] + (areas[k - 1] * dx);
            float integral = cdf[gm.tri_count];
            for (uint32_t k = 1; k <= gm.tri_count; ++k) cdf[k] /= integral;
            for (uint32_t k = 0; k < gm.tri_count; ++k) {
                DevLightTri lt;
                memset(&lt, 0, sizeof(lt));
                for (int c = 0; c < 3; ++c) {
                    lt.p0[c] = P[3 * I[3 * k] + c];
                    lt.p1[c] = P[3 * I[3 * k + 1] + c];
                    lt.p2[c] = P[3 * I[3 * k + 2] + c];
                    lt.n0[c] = N[3 * I[3 * k] + c];
                    lt.n1[c] = N[3 * I[3 * k + 1] + c];
                    lt.n2[c] = N[3 * I[3 * k + 2] + c];
                }
                lt.area = areas[k];
                lt.cdf_lo = cdf[k];
                lt.cdf_hi = cdf[k + 1];
                lt.has_normal = gm.has_normal ? 1.0f : 0.0f;
                out->light_tris.push_back(lt);
            }
            float world_area = sum * (gl.to_world.scale[0] * gl.to_world.scale[1]);
            pr = dl.color[0] * kPi * world_area; pg = dl.color[1] * kPi * world_area; pb = dl.color[2] * kPi * world_area;
        } else if (gl.type == GBL_LIGHT_IBL) {
            // ImageBasedLight's constructor (GoblinLight.cpp:464-508)
            if (gl.image < 0 || static_cast<uint32_t>(gl.image) >= d->num_images || d->images[gl.image].channels != 4) {
                *err = "image based light " + std::to_string(i) + ": bad image index";
                return GBL_ERR_INVALID;
            }
            out->extended = 1;
            out->has_ibl = 1;
            dl.image = gl.image;
            // mToWorld.rotateX(-PI / 2); rotateY(-PI / 2); setOrientation(orientation * mToWorld.getOrientation())
            struct Q { float w, x, y, z; };
            auto qmul = [](const Q& a, const Q& b) {   // GoblinQuaternion.h:45-48
                const float d3 = a.x * b.x + a.y * b.y + a.z * b.z;
                Q r;
                r.w = a.w * b.w - d3;
                r.x = a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y);
                r.y = a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z);
                r.z = a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x);
                return r;
            };
            auto qnorm = [](const Q& q) {              // normalize(Quaternion), GoblinQuaternion.cpp:94-100
                const float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
                Q r = {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
                return r;
            };
            auto axis_angle = [](int axis, float angle) {   // Quaternion(axis, angle), GoblinQuaternion.cpp:9-15 (unit axes)
                const float t = angle * 0.5f, st = sinf(t);
                Q r = {cosf(t), axis == 0 ? 1.0f * st : 0.0f * st, axis == 1 ? 1.0f * st : 0.0f * st, 0.0f * st};
                return r;
            };
            Q q = {1.0f, 0.0f, 0.0f, 0.0f};
            q = qnorm(qmul(axis_angle(0, -0.5f * kPi), q));
            q = qnorm(qmul(axis_angle(1, -0.5f * kPi), q));
            const Q given = {gl.to_world.orientation[0], gl.to_world.orientation[1], gl.to_world.orientation[2], gl.to_world.orientation[3]};
            q = qmul(given, q);
            const float qq[4] = {q.w, q.x, q.y, q.z}, one[3] = {1.0f, 1.0f, 1.0f}, zero[3] = {0.0f, 0.0f, 0.0f};
            Trs t = compose(zero, qq, one);
            store3x4(t.m, dl.m);
            store3x4(t.inv, dl.inv);
            const DevImage& im = out->images[gl.image];
            auto texel = [&](uint32_t level, int s, int tt, int c) {   // AddressRepeat
                const int w = static_cast<int>(std::max(1u, im.width >> level)), h = static_cast<int>(std::max(1u, im.height >> level));
                s %= w; tt %= h;
                if (s < 0) s += w;
                if (tt < 0) tt += h;
                return d->texels[im.offset + im.level_offset[level] + (static_cast<size_t>(tt) * w + s) * 4 + c];
            };
            // mAverageRadiance = mRadiance->lookup(maxLevel, 0, 0): the bilinear lookup of the 1 x 1 level
            const uint32_t max_level = im.levels - 1;
            float avg[3];
            {
                const int w = static_cast<int>(std::max(1u, im.width >> max_level)), h = static_cast<int>(std::max(1u, im.height >> max_level));
                const float s_res = 0.0f * w - 0.5f, t_res = 0.0f * h - 0.5f;
                const int s0 = static_cast<int>(floorf(s_res)), t0 = static_cast<int>(floorf(t_res));
                const float ds = s_res - static_cast<float>(s0), dt = t_res - static_cast<float>(t0);
                for (int c = 0; c < 3; ++c)
                    avg[c] = (1.0f - ds) * (1.0f - dt) * texel(max_level, s0, t0, c) + (ds) * (1.0f - dt) * texel(max_level, s0 + 1, t0, c) +
                             (1.0f - ds) * (dt)*texel(max_level, s0, t0 + 1, c) + (ds) * (dt)*texel(max_level, s0 + 1, t0 + 1, c);
            }
            // the sampling distribution: luminance * sin(theta) of level max(0, maxLevel - 8), as a CDF2D
            const uint32_t dl_level = im.levels > 9 ? max_level - 8 : 0;
            const int dw = static_cast<int>(std::max(1u, im.width >> dl_level)), dh = static_cast<int>(std::max(1u, im.height >> dl_level));
            dl.dist_offset = static_cast<uint32_t>(out->ibl_dist.size());
            dl.dist_w = static_cast<uint32_t>(dw);
            dl.dist_h = static_cast<uint32_t>(dh);
            auto cdf1d = [](const std::vector<float>& f, std::vector<float>* out_block) {   // CDF1D::init; block = func[n], cdf[n + 1], integral
                const size_t n = f.size();
                const float dx = 1.0f / n;
                std::vector<float> cdf(n + 1, 0.0f);
                for (size_t k = 1; k < n + 1; ++k) cdf[k] = cdf[k - 1] + (f[k - 1] * dx);
                const float integral = cdf[n];
                for (size_t k = 1; k < n + 1; ++k) cdf[k] /= integral;
                out_block->insert(out_block->end(), f.begin(), f.end());
                out_block->insert(out_block->end(), cdf.begin(), cdf.end());
                out_block->push_back(integral);
                return integral;
            };
            std::vector<float> rows_block, row_integrals;
            for (int r = 0; r < dh; ++r) {
                const float sin_theta = sinf((static_cast<float>(r) + 0.5f) / static_cast<float>(dh) * kPi);
                std::vector<float> f(dw);
                for (int c = 0; c < dw; ++c)
                    f[c] = (0.212671f * texel(dl_level, c, r, 0) + 0.715160f * texel(dl_level, c, r, 1) + 0.072169f * texel(dl_level, c, r, 2)) * sin_theta;
                row_integrals.push_back(cdf1d(f, &rows_block));
            }
            std::vector<float> marginal_block;
            cdf1d(row_integrals, &marginal_block);
            out->ibl_dist.insert(out->ibl_dist.end(), marginal_block.begin(), marginal_block.end());
            out->ibl_dist.insert(out->ibl_dist.end(), rows_block.begin(), rows_block.end());
            // ImageBasedLight::power (:606-613): mAverageRadiance * PI * (4 PI r^2) over the scene's bounding sphere
            float dx = scene_bound.hi[0] - scene_bound.lo[0], dy = scene_bound.hi[1] - scene_bound.lo[1], dz = scene_bound.hi[2] - scene_bound.lo[2];
            float radius = std::sqrt(dx * dx + dy * dy + dz * dz);
            float a = 4.0f * kPi * radius * radius;
            pr = avg[0] * kPi * a; pg = avg[1] * kPi * a; pb = avg[2] * kPi * a;
        } else if (gl.type == GBL_LIGHT_POINT) {
            float s = 4.0f * kPi;
            pr = dl.color[0] * s; pg = dl.color[1] * s; pb = dl.color[2] * s;
        } else {
            *err = "unknown light type";
            return GBL_ERR_INVALID;
        }
        power[i] = 0.212671f * pr + 0.715160f * pg + 0.072169f * pb;
        // WhittedRenderer::querySampleQuota: LightSampleIndex(quota, getSamplesNum()) -> roundToSquare slots
        {
            const int root = static_cast<int>(std::ceil(std::sqrt(static_cast<float>(gl.sample_num))));
            dl.wh_n = static_cast<uint32_t>(root * root);
            dl.wh_prefix = static_cast<uint32_t>(out->wh_slots);
            out->wh_slots += static_cast<int32_t>(dl.wh_n);
        }
    }
    out->light_cdf.assign(d->num_lights + 1, 0.0f);
    out->light_pick_pdf.assign(std::max<uint32_t>(1, d->num_lights), 0.0f);
    if (d->num_lights > 0) {
        float dx = 1.0f / d->num_lights;
        for (uint32_t i = 1; i <= d->num_lights; ++i) out->light_cdf[i] = out->light_cdf[i - 1] + (power[i - 1] * dx);
        float integral = out->light_cdf[d->num_lights];
        for (uint32_t i = 1; i <= d->num_lights; ++i) out->light_cdf[i] /= integral;
        for (uint32_t i = 0; i < d->num_lights; ++i) out->light_pick_pdf[i] = (power[i] / integral) * dx;
    }

    // ---- participating medium
    memset(&out->volume, 0, sizeof(out->volume));
    if (d->volume.type == GBL_VOLUME_HOMOGENEOUS || d->volume.type == GBL_VOLUME_HETEROGENEOUS) {
        DevVolume& v = out->volume;
        v.on = 1u;
        if (d->volume.type == GBL_VOLUME_HETEROGENEOUS) {
            const gbl_volume& g = d->volume;
            if (g.grid[0] <= 0 || g.grid[1] <= 0 || g.grid[2] <= 0 || (g.grid_channels != 1 && g.grid_channels != 3) || g.density == nullptr) {
                *err = "heterogeneous volume: the density grid needs positive dimensions, 1 or 3 channels and its data";
                return GBL_ERR_INVALID;
            }
            // the march loops of kernels/medium.h advance t by step_size: zero, negative, NaN or a step below the float spacing
            // of t never terminates -- a spin on the reference's CPU, an unrecoverable hang on a GPU
            if (!(g.step_size > 0.0f) || !std::isfinite(g.step_size)) {
                *err = "heterogeneous volume: step_size must be a positive finite number";
                return GBL_ERR_INVALID;
            }
            // the device indexes the grid with 32-bit ints
            const uint64_t cells = static_cast<uint64_t>(g.grid[0]) * static_cast<uint64_t>(g.grid[1]);
            if (cells > (1ull << 31) || cells * static_cast<uint64_t>(g.grid[2]) > (1ull << 31) ||
                cells * static_cast<uint64_t>(g.grid[2]) * static_cast<uint64_t>(g.grid_channels) >= (1ull << 31)) {
                *err = "heterogeneous volume: the density grid holds 2^31 values or more";
                return GBL_ERR_INVALID;
            }
            v.hetero = 1u;
            v.step = g.step_size;
            v.nx = g.grid[0], v.ny = g.grid[1], v.nz = g.grid[2], v.nch = g.grid_channels;
            out->vol_density.assign(g.density, g.density + static_cast<size_t>(v.nx) * v.ny * v.nz * v.nch);
        }
        for (int k = 0; k < 3; ++k) {
            v.attenuation[k] = d->volume.attenuation[k];
            v.scatter[k] = d->volume.attenuation[k] * d->volume.albedo[k];
            v.emission[k] = d->volume.emission[k];
            v.lo[k] = std::min(d->volume.box_min[k], d->volume.box_max[k]);   // BBox(p1, p2), GoblinBBox.h:20-23
            v.hi[k] = std::max(d->volume.box_min[k], d->volume.box_max[k]);
            v.bound_center[k] = 0.5f * (scene_bound.lo[k] + scene_bound.hi[k]);
        }
        for (int k = 0; k < 3; ++k) {
            v.albedo[k] = d->volume.albedo[k];
            v.normalize[k] = 1.0f / (v.hi[k] - v.lo[k]);
        }
        v.g = d->volume.g;
        v.sample_num = d->volume.sample_num;
        Trs t = compose(d->volume.to_world.position, d->volume.to_world.orientation, d->volume.to_world.scale);
        store3x4(t.m, v.m);
        store3x4(t.inv, v.inv);
        if (v.hetero != 0u) {
            // a march is at most (the region's longest world-space diagonal) / step_size points long: bounded at the 10^6 the
            // stream sampler's draw budget assumes (gbl_api.hip medium_draws_per_sample) -- a step of 1e-12 is a hang, not a render
            double diag = 0.0;
            for (int sgn = 0; sgn < 4; ++sgn) {
                const double e[3] = {double(v.hi[0] - v.lo[0]), (sgn & 1 ? -1.0 : 1.0) * double(v.hi[1] - v.lo[1]), (sgn & 2 ? -1.0 : 1.0) * double(v.hi[2] - v.lo[2])};
                double w[3];
                for (int r = 0; r < 3; ++r) w[r] = v.m[4 * r] * e[0] + v.m[4 * r + 1] * e[1] + v.m[4 * r + 2] * e[2];
                diag = std::max(diag, std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
            }
            if (!(diag / double(v.step) <= 1.0e6)) {
                *err = "heterogeneous volume: step_size is too small for the region (more than 10^6 steps across it)";
                return GBL_ERR_INVALID;
            }
        }
        const float dx = scene_bound.hi[0] - scene_bound.lo[0], dy = scene_bound.hi[1] - scene_bound.lo[1], dz = scene_bound.hi[2] - scene_bound.lo[2];
        v.bound_radius = std::sqrt(dx * dx + dy * dy + dz * dz);
        out->extended = 1;
    } else if (d->volume.type != GBL_VOLUME_NONE) {
        *err = "unknown volume type";
        return GBL_ERR_INVALID;
    }

    // ---- camera
    const gbl_camera& c = d->camera;
    memset(&out->camera, 0, sizeof(out->camera));
    for (int k = 0; k < 3; ++k) out->camera.pos[k] = c.position[k];
    for (int k = 0; k < 4; ++k) out->camera.q[k] = c.orientation[k];
    float aspect = static_cast<float>(d->film.xres) / static_cast<float>(d->film.yres);
    float fov = kPi * (c.fov_degrees / 180.0f);
    float ys = 1.0f / std::tan(fov / 2.0f);
    out->camera.proj11 = ys;
    out->camera.proj00 = ys / aspect;
    out->camera.inv_xres = 1.0f / static_cast<float>(d->film.xres);
    out->camera.inv_yres = 1.0f / static_cast<float>(d->film.yres);
    out->camera.type = c.type;
    out->camera.lens_radius = c.lens_radius;
    out->camera.focal_distance = c.focal_distance;
    out->camera.film_w = c.film_width;              // OrthographicCamera ctor, GoblinCamera.cpp:288-296
    out->camera.film_h = c.film_width / aspect;

    // ---- film + filter table
    const gbl_film& f = d->film;
    if (f.xres <= 0 || f.yres <= 0) {
        *err = "film resolution must be positive";
        return GBL_ERR_INVALID;
    }
    DevFilm& df = out->film;
    memset(&df, 0, sizeof(df));
    df.xres = f.xres;
    df.yres = f.yres;
    df.xstart = ceil_i(f.xres * f.crop[0]);
    df.xcount = std::max(1, ceil_i(f.xres * f.crop[1]) - df.xstart);
    df.ystart = ceil_i(f.yres * f.crop[2]);
    df.ycount = std::max(1, ceil_i(f.yres * f.crop[3]) - df.ystart);
    df.wx = f.filter_width[0];
    df.wy = f.filter_width[1];
    df.window[0] = floor_i(df.xstart + 0.5f - df.wx);
    df.window[1] = floor_i(df.xstart + 0.5f + df.xcount + df.wx);
    df.window[2] = floor_i(df.ystart + 0.5f - df.wy);
    df.window[3] = floor_i(df.ystart + 0.5f + df.ycount + df.wy);
    df.halo = ceil_i(std::max(df.wx, df.wy) + 0.5f);
    if (!(df.wx > 0.0f) || !(df.wy > 0.0f) || df.halo > GBL_MAX_FILTER_HALO) {
        *err = "filter width must be in (0, " + std::to_string(GBL_MAX_FILTER_HALO - 0.5f) + "] pixels";
        return GBL_ERR_UNSUPPORTED;
    }
    Filter flt;
    flt.type = f.filter_type;
    flt.wx = df.wx; flt.wy = df.wy;
    flt.alpha = f.gaussian_falloff;
    flt.ex = expf(-flt.alpha * flt.wx * flt.wx);
    flt.ey = expf(-flt.alpha * flt.wy * flt.wy);
    flt.b = f.mitchell_b; flt.c = f.mitchell_c;
    float norm = flt.norm();
    float dxs = flt.wx / 16, dys = flt.wy / 16;
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) out->filter_table[16 * y + x] = flt.eval(x * dxs, y * dys) / norm;
    // the triangle bounds in the order of `tris` (what the kernels index with a hit's triangle); a device build gathers them itself
    out->tri_bounds_leaf.resize(out->tris.size());
    for (size_t i = 0; i < out->tris.size(); ++i) out->tri_bounds_leaf[i] = out->tri_bounds[out->tris[i].shade];
    // ---- hot prefix: the nodes every ray starts with, once more, at indices [0, hot_nodes) in breadth-first order from the
    // TLAS root through the instances' BLAS roots.  The lean kernels keep as many of them as their LDS has room for beside
    // the traversal stacks (trace.h HotNodes): a reference below hot_nodes is served from LDS, anything else from memory.  The
    // originals stay where they were (unreferenced from now on: at most GBL_HOT_NODES_MAX * 64 bytes); every interior reference
    // -- child slots, instance roots, mesh roots, the TLAS root -- is rewritten to `hot index` or `old index + hot_nodes`, and
    // tlas_base moves with the rest, so gbl_update_instances rebuilds the TLAS in the ordinary region and the BLAS part of the
    // prefix stays valid.  (Host-built trees only: the device builder writes its nodes on the device.)
    out->hot_nodes = 0;
    if (!device_blas && !out->nodes.empty() && out->instances.size() > 0) {
        uint32_t want = GBL_HOT_NODES_MAX;
        if (const char* e = getenv("GBL_HOT_NODES")) want = static_cast<uint32_t>(std::max(0, std::min(4096, atoi(e))));
        auto interior = [&](int32_t r) { return r >= 0 && static_cast<uint32_t>(r) < static_cast<uint32_t>(GBL_REF_NONE); };
        std::vector<int32_t> order;          // old indices in breadth-first order
        std::vector<int32_t> hot_of(out->nodes.size(), -1);
        auto visit = [&](int32_t r) {
            if (interior(r) && static_cast<size_t>(r) < out->nodes.size() && hot_of[r] < 0 && order.size() < want) {
                hot_of[r] = static_cast<int32_t>(order.size());
                order.push_back(r);
            }
        };
        visit(out->tlas_root);
        for (size_t head = 0; head < order.size() && order.size() < want; ++head) {
            const DevNode& nd = out->nodes[order[head]];
            for (int k = 0; k < 4; ++k) {
                const int32_t c = nd.child[k];
                if (interior(c)) {
                    visit(c);
                } else if (c < 0 && order[head] >= out->tlas_base) {   // a TLAS leaf: on into the instance's BLAS
                    const uint32_t inst = (~static_cast<uint32_t>(c)) >> 2;
                    if (inst < out->instances.size() && out->instances[inst].shape == 0u) visit(out->instances[inst].root);
                }
            }
        }
        if (interior(out->tlas_root) == false)   // a one-instance scene: the TLAS "root" is the instance's leaf reference itself
            for (const DevInstance& di : out->instances)
                if (di.shape == 0u) visit(di.root);
        for (size_t head = 0; head < order.size() && order.size() < want; ++head)
            for (int k = 0; k < 4; ++k) visit(out->nodes[order[head]].child[k]);
        const int32_t H = static_cast<int32_t>(order.size());
        if (H > 0) {
            auto remap = [&](int32_t r) { return !interior(r) ? r : (hot_of[r] >= 0 ? hot_of[r] : r + H); };
            std::vector<DevNode> moved;
            moved.reserve(out->nodes.size() + H);
            for (int32_t i = 0; i < H; ++i) moved.push_back(out->nodes[order[i]]);
            moved.insert(moved.end(), out->nodes.begin(), out->nodes.end());
            for (DevNode& nd : moved)
                for (int k = 0; k < 4; ++k) nd.child[k] = remap(nd.child[k]);
            out->nodes.swap(moved);
            for (DevInstance& di : out->instances)
                if (di.shape == 0u) di.root = remap(di.root);
            for (uint32_t mi = 0; mi < d->num_meshes; ++mi)
                if (d->meshes[mi].shape == GBL_SHAPE_MESH) out->mesh_root[mi] = remap(out->mesh_root[mi]);
            out->tlas_root = remap(out->tlas_root);
            out->tlas_base += H;
            out->hot_nodes = static_cast<uint32_t>(H);
        }
    }
    return GBL_OK;
}
