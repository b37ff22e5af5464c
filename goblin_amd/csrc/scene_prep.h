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
    std::vector<float> normals, uvs;
    std::vector<DevInstance> instances;
    std::vector<DevMaterial> materials;
    std::vector<DevTexture> textures;
    std::vector<DevLight> lights;
    std::vector<DevLightTri> light_tris;
    std::vector<float> light_cdf, light_pick_pdf;
    float filter_table[256];
    int32_t tlas_root = 0;
    int32_t stack_entries = 0;
    int32_t extended = 0;   // DevScene::extended
    int32_t has_masks = 0;  // DevScene::has_masks
    uint64_t blas_nodes = 0, tlas_nodes = 0;
    int blas_max_depth = 0, tlas_depth = 0;
    DevCamera camera;
    DevFilm film;
};

// Returns GBL_OK or an error with *err set.
gbl_status pack_scene(const gbl_scene_desc* desc, PackedScene* out, std::string* err);
