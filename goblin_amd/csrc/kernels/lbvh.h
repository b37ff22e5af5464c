// BLAS construction on the device: a linear BVH over Morton-sorted triangles
// (Lauterbach et al. 2009; hierarchy by Karras 2012, "Maximizing Parallelism in the Construction of BVHs,
// Octrees, and k-d Trees"), fitted bottom-up, then collapsed into the same 4-wide, 8-bit quantised DevNode
// layout the host SAH builder (scene_prep.cpp Flat4) emits -- so the traversal kernels do not know which
// builder ran.  The reference builds its BVH on the host by median splits (GoblinBVH.cpp:34-151); radiance
// does not depend on the tree (apart from exact-t ties), only the build and traversal cost do.
//
//   lbvh_keys        per triangle: bounds, 30-bit Morton code of the centroid, key = code << 32 | index
//   (hipcub)         radix sort of the 64-bit keys
//   lbvh_hierarchy   per internal node: Karras' range / split search on the sorted keys
//   lbvh_fit         leaves up: the second child to arrive merges the boxes (one atomic counter per node)
//   lbvh_collapse    level by level from the root: a node takes its two children, keeps opening the
//                    largest one until it has four, quantises their boxes and queues the interior ones
//   lbvh_tris        DevTri records in sorted order
#pragma once
#include <hip/hip_runtime.h>

#include "../device_scene.h"

struct LbvhBox {
    float lo[3], hi[3];
};

// Binary radix tree.  Node ids: internal nodes 0 .. n-2, leaf k is encoded as ~k.
struct LbvhTree {
    int32_t* left;       // [n-1]
    int32_t* right;      // [n-1]
    int32_t* parent;     // [2n-1]: internal i at i, leaf k at (n-1) + k
    uint32_t* first;     // [n-1] first sorted triangle of the subtree
    uint32_t* last;      // [n-1] last sorted triangle (inclusive)
    LbvhBox* box;        // [n-1] internal node boxes
    LbvhBox* leaf_box;   // [n]   sorted triangle boxes
    uint32_t* visits;    // [n-1] zeroed; lbvh_fit's arrival counters
};

__device__ __forceinline__ uint32_t lbvh_expand_bits(uint32_t v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void lbvh_keys(const float* pos, const uint32_t* idx, uint32_t n, LbvhBox mesh, unsigned long long* keys, LbvhBox* tri_box) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    LbvhBox b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = INFINITY;
        b.hi[a] = -INFINITY;
    }
    for (int k = 0; k < 3; ++k) {
        const float* p = pos + 3 * static_cast<size_t>(idx[3 * static_cast<size_t>(t) + k]);
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = fminf(b.lo[a], p[a]);
            b.hi[a] = fmaxf(b.hi[a], p[a]);
        }
    }
    tri_box[t] = b;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        float ext = mesh.hi[a] - mesh.lo[a];
        float c = 0.5f * (b.lo[a] + b.hi[a]);
        float u = ext > 0.0f ? (c - mesh.lo[a]) / ext : 0.0f;
        q[a] = static_cast<uint32_t>(fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f));
    }
    uint32_t code = (lbvh_expand_bits(q[0]) << 2) | (lbvh_expand_bits(q[1]) << 1) | lbvh_expand_bits(q[2]);
    keys[t] = (static_cast<unsigned long long>(code) << 32) | t;   // the index makes every key unique
}

__global__ void lbvh_gather_boxes(const unsigned long long* keys, const LbvhBox* tri_box, uint32_t n, LbvhBox* leaf_box) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) leaf_box[k] = tri_box[static_cast<uint32_t>(keys[k] & 0xffffffffull)];
}

// length of the common prefix of keys i and j, -1 outside the array (Karras' delta)
__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll(static_cast<long long>(keys[i] ^ keys[j]));
}

__global__ void lbvh_hierarchy(const unsigned long long* keys, int n, LbvhTree t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    // direction of the range and an upper bound of its length
    int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (lbvh_delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    int j = i + l * d;
    // split position: the highest differing bit inside the range
    int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int div = 2, step = (l + div - 1) / div;; div *= 2, step = (l + div - 1) / div) {
        if (lbvh_delta(keys, n, i, i + (s + step) * d) > dnode) s += step;
        if (step <= 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int left = lo == gamma ? ~gamma : gamma;
    int right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = left;
    t.right[i] = right;
    t.first[i] = static_cast<uint32_t>(lo);
    t.last[i] = static_cast<uint32_t>(hi);
    t.parent[left >= 0 ? left : (n - 1) + ~left] = i;
    t.parent[right >= 0 ? right : (n - 1) + ~right] = i;
    if (i == 0) t.parent[0] = -1;
}

__device__ __forceinline__ LbvhBox lbvh_node_box(const LbvhTree& t, int ref) { return ref >= 0 ? t.box[ref] : t.leaf_box[~ref]; }

__global__ void lbvh_fit(int n, LbvhTree t) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int node = t.parent[(n - 1) + k];
    while (node >= 0) {
        // the first child to arrive stops; the second one sees both subtrees finished
        if (atomicAdd(&t.visits[node], 1u) == 0u) return;
        __threadfence();
        LbvhBox a = lbvh_node_box(t, t.left[node]), b = lbvh_node_box(t, t.right[node]);
        LbvhBox m;
        for (int x = 0; x < 3; ++x) {
            m.lo[x] = fminf(a.lo[x], b.lo[x]);
            m.hi[x] = fmaxf(a.hi[x], b.hi[x]);
        }
        t.box[node] = m;
        __threadfence();
        node = t.parent[node];
    }
}

struct LbvhFrontier {
    int32_t node;   // binary node that becomes a 4-wide node
    int32_t slot;   // its index in the output array (relative to node_base)
};

__device__ __forceinline__ float lbvh_nudge_down(float v) { return v - fabsf(v) * 4e-7f - 1e-30f; }
__device__ __forceinline__ float lbvh_nudge_up(float v) { return v + fabsf(v) * 4e-7f + 1e-30f; }

// A subtree of at most GBL_MAX_LEAF_TRIS triangles is one leaf: its triangles are contiguous in sorted order.
__device__ __forceinline__ bool lbvh_is_leaf(const LbvhTree& t, int ref) { return ref < 0 || t.last[ref] - t.first[ref] + 1u <= GBL_MAX_LEAF_TRIS; }

__global__ void lbvh_collapse(LbvhTree t, const LbvhFrontier* in, uint32_t n_in, LbvhFrontier* out, uint32_t* n_out, uint32_t* n_nodes,
                              DevNode* nodes, int32_t node_base, uint32_t tri_base) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_in) return;
    const LbvhFrontier f = in[w];
    int kids[4] = {t.left[f.node], t.right[f.node], 0, 0};
    int n = 2;
    while (n < 4) {   // open the interior child with the largest surface area (scene_prep.cpp Flat4::emit)
        int best = -1;
        float best_area = -1.0f;
        for (int i = 0; i < n; ++i) {
            if (lbvh_is_leaf(t, kids[i])) continue;
            LbvhBox b = t.box[kids[i]];
            float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
            float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area) {
                best_area = area;
                best = i;
            }
        }
        if (best < 0) break;
        int c = kids[best];
        kids[best] = t.left[c];
        kids[n++] = t.right[c];
    }
    DevNode nd;
    LbvhBox kb[4];
    for (int i = 0; i < 4; ++i) {
        nd.child[i] = static_cast<int32_t>(GBL_REF_NONE);
        if (i >= n) continue;
        kb[i] = lbvh_node_box(t, kids[i]);
        if (lbvh_is_leaf(t, kids[i])) {
            uint32_t first = kids[i] < 0 ? static_cast<uint32_t>(~kids[i]) : t.first[kids[i]];
            uint32_t count = kids[i] < 0 ? 1u : t.last[kids[i]] - t.first[kids[i]] + 1u;
            nd.child[i] = ~static_cast<int32_t>(((tri_base + first) << 2) | (count - 1u));
        } else {
            uint32_t slot = atomicAdd(n_nodes, 1u);
            nd.child[i] = node_base + static_cast<int32_t>(slot);
            uint32_t q = atomicAdd(n_out, 1u);
            out[q].node = kids[i];
            out[q].slot = static_cast<int32_t>(slot);
        }
    }
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) {
            lo = fminf(lo, lbvh_nudge_down(kb[i].lo[a]));
            hi = fmaxf(hi, lbvh_nudge_up(kb[i].hi[a]));
        }
        nd.o[a] = lo;
        // grid step 2^e with 255 * 2^e >= extent (same rule as the host packer)
        float extent = (hi - lo) * 1.0001f + 1e-30f;
        int e = 0;
        frexpf(extent / 255.0f, &e);
        int biased = min(254, max(1, e + 127));
        float step = ldexpf(1.0f, biased - 127);
        nd.scale[a] = step;
        uint32_t ql = 0, qh = 0;
        for (int i = 0; i < 4; ++i) {
            uint32_t l = 255, h = 0;
            if (i < n) {
                float cl = lbvh_nudge_down(kb[i].lo[a]), ch = lbvh_nudge_up(kb[i].hi[a]);
                double fl = floor((static_cast<double>(cl) - lo) / step);
                double fh = ceil((static_cast<double>(ch) - lo) / step);
                while (fl > 0 && lo + static_cast<float>(fl) * step > cl) fl -= 1;
                while (fh < 255 && lo + static_cast<float>(fh) * step < ch) fh += 1;
                l = static_cast<uint32_t>(fmin(255.0, fmax(0.0, fl)));
                h = static_cast<uint32_t>(fmin(255.0, fmax(0.0, fh)));
            }
            ql |= l << (8 * i);
            qh |= h << (8 * i);
        }
        nd.qlo[a] = ql;
        nd.qhi[a] = qh;
    }
    nodes[f.slot] = nd;
}

__global__ void lbvh_tris(const float* pos, const uint32_t* idx, const unsigned long long* keys, uint32_t n, uint32_t shade_base, uint32_t flags, DevTri* out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t t = static_cast<uint32_t>(keys[k] & 0xffffffffull);
    const float* p0 = pos + 3 * static_cast<size_t>(idx[3 * static_cast<size_t>(t)]);
    const float* p1 = pos + 3 * static_cast<size_t>(idx[3 * static_cast<size_t>(t) + 1]);
    const float* p2 = pos + 3 * static_cast<size_t>(idx[3 * static_cast<size_t>(t) + 2]);
    DevTri d;
    for (int a = 0; a < 3; ++a) {
        d.p0[a] = p0[a];
        d.e1[a] = p1[a] - p0[a];
        d.e2[a] = p2[a] - p0[a];
    }
    d.shade = shade_base + t;
    d.flags = flags;
    d.pad1 = 0.0f;
    out[k] = d;
}
