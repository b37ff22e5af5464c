// Host-side packing of a gbl_scene_desc into the device layout of device_scene.h.
#pragma once
#include <string>
#include <vector>

#include "../../include/goblin_hip.h"
#include "device_scene.h"

struct PackedScene {
    std::vector<DevNode> nodes;
    std::vector<DevTri> tris;
    std::vector<DevTriShade> tri_shade;
    std::vector<DevTriBound> tri_bounds;        // per original triangle (DevTri::shade)
    std::vector<DevTriBound> tri_bounds_leaf;   // ... and in the order of `tris` (host-built trees)
    std::vector<DevInstanceBound> instance_bounds;
    std::vector<DevTriOrder> tri_order;
    std::vector<float> positions;         // 3 per vertex
    std::vector<float> normals, uvs;
    std::vector<DevInstance> instances;
    std::vector<DevMaterial> materials;
    std::vector<DevTexture> textures;
    std::vector<DevImage> images;
    std::vector<float> ewa_lut, ibl_dist, vol_density;
    int32_t has_ibl = 0;
    std::vector<DevLight> lights;
    std::vector<DevLightTri> light_tris;
    std::vector<float> light_cdf, light_pick_pdf;
    float filter_table[256];
    int32_t tlas_root = 0;
    int32_t stack_entries = 0;
    int32_t extended = 0;   // DevScene::extended
    int32_t has_masks = 0;  // DevScene::has_masks
    int32_t has_bssrdf = 0; // DevScene::has_bssrdf
    int32_t wh_slots = 0;   // DevScene::wh_slots
    DevVolume volume = {};
    uint64_t blas_nodes = 0, tlas_nodes = 0;
    int blas_max_depth = 0, tlas_depth = 0;
    std::vector<int> mesh_stack_need;      // per mesh: most entries a ray can have on its stack inside its BLAS (exact for host-built trees)
    std::vector<float> mesh_lo, mesh_hi;   // 3 per mesh: object bounds
    std::vector<int32_t> mesh_root;        // per mesh: BLAS root reference
    int32_t tlas_base = 0;                 // device index of TLAS node 0
    uint32_t tlas_capacity = 0;            // nodes reserved at tlas_base (any TLAS over the instances fits)
    uint32_t hot_nodes = 0;                // nodes[0 .. hot_nodes): the top of the two-level tree in breadth-first order (hot_prefix)
    DevCamera camera;
    DevFilm film;
};

// Returns GBL_OK or an error with *err set.
// device_blas: leave the triangle BLASes to the device builder (kernels/lbvh.h): `nodes` then holds the TLAS only
// (at indices 0..), `tris` stays empty, mesh instances get root = 0 (patched after the device build) and
// `mesh_lo/hi` carry the object bounds the Morton codes are scaled by.
gbl_status pack_scene(const gbl_scene_desc* desc, PackedScene* out, std::string* err, bool device_blas = false);

// The instance records and the TLAS over them (also used by gbl_update_instances to rebuild after transform edits).
gbl_status build_tlas(const gbl_instance* inst, uint32_t n, const gbl_mesh* meshes, const gbl_material* materials, const float* mesh_lo,
                      const float* mesh_hi, const int32_t* mesh_root, int32_t tlas_base, std::vector<DevInstance>* out_inst,
                      std::vector<DevNode>* out_nodes, int32_t* tlas_root, int* tlas_depth, float sb_lo[3], float sb_hi[3], std::string* err,
                      std::vector<DevInstanceBound>* bounds_out = nullptr);

// Traversal stack entries a scene needs (kernels/trace.h): exit marker + the worst root-to-leaf sum over the TLAS nodes
// (tlas[0..] are the nodes at absolute indices tlas_base + i) of (children - 1), + per instance its sentinel and what its
// mesh's BLAS needs, + one spare.
int scene_stack_entries(const std::vector<DevNode>& tlas, int32_t tlas_base, int32_t tlas_root, const std::vector<DevInstance>& instances,
                        const std::vector<int>& mesh_stack_need);
