// flattenSceneForHip: a loaded Goblin Scene -> gbl_scene_desc (include/goblin_hip.h), read from the reference's own objects the way
// INTEGRATION.md 1's table says.  TEST INFRASTRUCTURE (see GoblinHipPathtracer.h); compiled with -fno-access-control like
// oracle/ref_harness.cpp, because the reference keeps what a flattener needs private (Scene::mBVH's primitive list,
// GoblinScene.h:43-49; Model::mGeometry, GoblinModel.h:41-46; the materials' texture pointers) -- in the reference tree the
// maintainer would add accessors or a friend declaration instead.
//
// Covers the headline feature set: triangle meshes, instances, lambert / transparent / mirror / blinn materials over
// constant textures, point / spot / area lights, the perspective pinhole camera, every reconstruction filter.  Anything else
// throws std::runtime_error naming the feature (the shipped loader, libgoblin_host.so, covers the rest).
#ifndef GOBLIN_FLATTEN_SCENE_H
#define GOBLIN_FLATTEN_SCENE_H
#include <vector>

#include "GoblinScene.h"
#include "goblin_hip.h"

namespace Goblin {
// Owns the arrays the description points into.
struct FlatScene {
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices;
    std::vector<gbl_mesh> meshes;
    std::vector<gbl_material> materials;
    std::vector<gbl_instance> instances;
    std::vector<gbl_light> lights;
    gbl_scene_desc desc;
};
// setting: the "render_setting" block as createRenderer reads it (GoblinContextLoader.cpp:67-92)
void flattenSceneForHip(const ScenePtr& scene, const gbl_render_setting& setting, FlatScene* out);
}
#endif
