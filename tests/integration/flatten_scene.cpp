// See flatten_scene.h.  Every field cites where the reference keeps it.
#include "flatten_scene.h"

#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "GoblinCamera.h"
#include "GoblinFilm.h"
#include "GoblinFilter.h"
#include "GoblinLight.h"
#include "GoblinMaterial.h"
#include "GoblinModel.h"
#include "GoblinPolygonMesh.h"
#include "GoblinPrimitive.h"
#include "GoblinTexture.h"
#include "GoblinUtils.h"

namespace Goblin {
namespace {

[[noreturn]] void unsupported(const std::string& what) {
    throw std::runtime_error("flattenSceneForHip: " + what + " is outside this binding's subset (use libgoblin_host.so's loader)");
}

gbl_trs trs_of(const Transform& t) {   // getTransform's position / orientation / scale (GoblinUtils.cpp:92-100)
    gbl_trs o;
    const Vector3& p = t.getPosition();
    const Quaternion& q = t.getOrientation();
    const Vector3& s = t.getScale();
    o.position[0] = p.x; o.position[1] = p.y; o.position[2] = p.z;
    o.orientation[0] = q.w; o.orientation[1] = q.v.x; o.orientation[2] = q.v.y; o.orientation[3] = q.v.z;
    o.scale[0] = s.x; o.scale[1] = s.y; o.scale[2] = s.z;
    return o;
}

// a texture slot: the constant's value (createColorConstantTexture, GoblinTexture.cpp:617-625)
void constant_color(const ColorTexturePtr& tex, float out[3], const char* slot) {
    const ConstantTexture<Color>* c = dynamic_cast<const ConstantTexture<Color>*>(tex.get());
    if (!c) unsupported(std::string("a non-constant texture in slot ") + slot);
    out[0] = c->mValue.r; out[1] = c->mValue.g; out[2] = c->mValue.b;
}
float constant_float(const FloatTexturePtr& tex, const char* slot) {
    const ConstantTexture<float>* c = dynamic_cast<const ConstantTexture<float>*>(tex.get());
    if (!c) unsupported(std::string("a non-constant texture in slot ") + slot);
    return c->mValue;
}

struct Flattener {
    FlatScene* out;
    std::map<const Geometry*, uint32_t> mesh_of;
    std::map<const Material*, uint32_t> material_of;
    std::map<const Light*, int32_t> light_of;

    uint32_t mesh(const Geometry* g) {
        auto it = mesh_of.find(g);
        if (it != mesh_of.end()) return it->second;
        const PolygonMesh* pm = dynamic_cast<const PolygonMesh*>(g);
        if (!pm) unsupported("an analytic shape (sphere / disk)");
        gbl_mesh m;
        memset(&m, 0, sizeof(m));
        m.vertex_offset = static_cast<uint32_t>(out->positions.size() / 3);
        m.vertex_count = static_cast<uint32_t>(pm->mVertices.size());       // PolygonMesh after OBJ loading + de-duplication
        m.tri_offset = static_cast<uint32_t>(out->indices.size() / 3);
        m.tri_count = static_cast<uint32_t>(pm->mTriangles.size());
        m.has_normal = pm->hasNormal() ? 1u : 0u;
        m.has_uv = pm->hasTexCoord() ? 1u : 0u;
        m.shape = GBL_SHAPE_MESH;
        m.radius = 1.0f;
        for (uint32_t i = 0; i < m.vertex_count; ++i) {
            const Vertex* v = pm->getVertexPtr(i);
            out->positions.insert(out->positions.end(), {v->position.x, v->position.y, v->position.z});
            if (m.has_normal) out->normals.insert(out->normals.end(), {v->normal.x, v->normal.y, v->normal.z});
            else out->normals.insert(out->normals.end(), {0.0f, 0.0f, 0.0f});
            if (m.has_uv) out->uvs.insert(out->uvs.end(), {v->texC.x, v->texC.y});
            else out->uvs.insert(out->uvs.end(), {0.0f, 0.0f});
        }
        for (uint32_t i = 0; i < m.tri_count; ++i) {
            const TriangleIndex* t = pm->getFacePtr(i);
            out->indices.insert(out->indices.end(), {t->v[0], t->v[1], t->v[2]});
        }
        const uint32_t id = static_cast<uint32_t>(out->meshes.size());
        out->meshes.push_back(m);
        mesh_of[g] = id;
        return id;
    }

    uint32_t material(const MaterialPtr& mp) {
        auto it = material_of.find(mp.get());
        if (it != material_of.end()) return it->second;
        gbl_material m;
        memset(&m, 0, sizeof(m));
        m.tex_color = m.tex_color2 = m.tex_exponent = m.tex_color3 = m.tex_bump = m.tex_normal = -1;
        m.masked_material = -1;
        m.color3[0] = m.color3[1] = m.color3[2] = 1.0f;
        if (mp->mBumpShaders.bumpMap || mp->mBumpShaders.normalMap) unsupported("a bump or normal map");
        if (const LambertMaterial* l = dynamic_cast<const LambertMaterial*>(mp.get())) {              // GoblinMaterial.cpp:825-833
            m.type = GBL_MAT_LAMBERT;
            constant_color(l->mDiffuseFactor, m.color, "Kd");
            m.index = 1.5f;
        } else if (const TransparentMaterial* t = dynamic_cast<const TransparentMaterial*>(mp.get())) {   // :853-866
            m.type = GBL_MAT_TRANSPARENT;
            constant_color(t->mReflectFactor, m.color, "Kr");
            constant_color(t->mRefractFactor, m.color2, "Kt");
            m.index = t->mEtat;
        } else if (const MirrorMaterial* r = dynamic_cast<const MirrorMaterial*>(mp.get())) {              // :868-879
            m.type = GBL_MAT_MIRROR;
            constant_color(r->mReflectFactor, m.color, "Kr");
            m.index = r->mEta;
            m.k = r->mK;
        } else if (const BlinnMaterial* b = dynamic_cast<const BlinnMaterial*>(mp.get())) {                // :835-851
            m.type = GBL_MAT_BLINN;
            constant_color(b->mGlossyFactor, m.color, "Kg");
            m.exponent = constant_float(b->mExp, "exponent");
            m.index = b->mEta;
            m.k = b->mK;
        } else {
            unsupported("a mask or subsurface material");
        }
        const uint32_t id = static_cast<uint32_t>(out->materials.size());
        out->materials.push_back(m);
        material_of[mp.get()] = id;
        return id;
    }
};

}   // namespace

void flattenSceneForHip(const ScenePtr& scene, const gbl_render_setting& setting, FlatScene* out) {
    Flattener f;
    f.out = out;
    if (scene->getVolumeRegion() != nullptr) unsupported("a participating medium");

    // ---- lights, in SceneCache::getLights() order (Scene keeps that list: GoblinScene.cpp:11-27)
    const std::vector<Light*>& lights = scene->getLights();
    out->lights.resize(lights.size());
    for (size_t i = 0; i < lights.size(); ++i) {
        const Light* l = lights[i];
        f.light_of[l] = static_cast<int32_t>(i);
        gbl_light g;
        memset(&g, 0, sizeof(g));
        g.sample_num = 1;
        g.image = -1;
        g.to_world = trs_of(Transform());
        const Vector3& pos = l->mToWorld.getPosition();
        if (const SpotLight* s = dynamic_cast<const SpotLight*>(l)) {                       // GoblinLight.cpp:212-223
            g.type = GBL_LIGHT_SPOT;
            g.color[0] = s->mIntensity.r; g.color[1] = s->mIntensity.g; g.color[2] = s->mIntensity.b;
            g.position[0] = pos.x; g.position[1] = pos.y; g.position[2] = pos.z;
            const Vector3 d = l->getParams().getVector3("direction");                       // the normalised direction the ctor kept
            g.direction[0] = d.x; g.direction[1] = d.y; g.direction[2] = d.z;
            g.cos_theta_max = s->mCosThetaMax;
            g.cos_falloff_start = s->mCosFalloffStart;
        } else if (const PointLight* p = dynamic_cast<const PointLight*>(l)) {              // :78-86
            g.type = GBL_LIGHT_POINT;
            g.color[0] = p->mIntensity.r; g.color[1] = p->mIntensity.g; g.color[2] = p->mIntensity.b;
            g.position[0] = pos.x; g.position[1] = pos.y; g.position[2] = pos.z;
        } else if (const AreaLight* a = dynamic_cast<const AreaLight*>(l)) {                // :345-366; mesh filled in below
            g.type = GBL_LIGHT_AREA;
            g.color[0] = a->mLe.r; g.color[1] = a->mLe.g; g.color[2] = a->mLe.b;
            g.to_world = trs_of(l->mToWorld);
            g.sample_num = a->getSamplesNum();
        } else {
            unsupported("a directional or image based light");
        }
        out->lights[i] = g;
    }

    // ---- instances: what the scene BVH was built over (Scene::mBVH's primitive list; SceneCache::getInstances() is gone by now,
    // and only "instance" primitives are rendered, GoblinContextLoader.cpp:381-383)
    for (const Primitive* prim : scene->mBVH.mRefinedPrimitives) {
        const InstancedPrimitive* ip = dynamic_cast<const InstancedPrimitive*>(prim);
        if (!ip) unsupported("a scene-level primitive that is not an instance");
        const Model* model = dynamic_cast<const Model*>(ip->mPrimitive);
        if (!model) unsupported("an instance of an instance");
        if (model->isCameraLens()) unsupported("the thin lens's disk");
        gbl_instance gi;
        memset(&gi, 0, sizeof(gi));
        gi.mesh = f.mesh(model->mGeometry);
        gi.material = f.material(model->getMaterial());
        gi.area_light = -1;
        if (const AreaLight* al = model->getAreaLight()) {
            gi.area_light = f.light_of.at(al);
            out->lights[gi.area_light].mesh = gi.mesh;
        }
        gi.to_world = trs_of(ip->mToWorld);
        out->instances.push_back(gi);
    }

    // ---- camera, film, filter
    gbl_scene_desc& d = out->desc;
    memset(&d, 0, sizeof(d));
    d.abi_version = GBL_ABI_VERSION;
    const CameraPtr cam = scene->getCamera();
    const PerspectiveCamera* pc = dynamic_cast<const PerspectiveCamera*>(cam.get());
    if (!pc) unsupported("the orthographic camera");
    if (pc->mLensRadius != 0.0f) unsupported("the thin lens");
    const Vector3& cp = cam->getPosition();
    const Quaternion& cq = cam->getOrientation();
    d.camera.position[0] = cp.x; d.camera.position[1] = cp.y; d.camera.position[2] = cp.z;
    d.camera.orientation[0] = cq.w; d.camera.orientation[1] = cq.v.x; d.camera.orientation[2] = cq.v.y; d.camera.orientation[3] = cq.v.z;
    {   // createPerspectiveCamera stores radians(fov) (GoblinCamera.cpp:377-387); the C ABI takes the file's degrees and converts
        // the same way: find the degrees whose radians() is the stored value, bit for bit
        float deg = degrees(pc->mFOV);
        bool found = radians(deg) == pc->mFOV;
        for (int k = 1; k <= 4 && !found; ++k) {
            float lo = deg, hi = deg;
            for (int j = 0; j < k; ++j) { lo = std::nextafter(lo, -INFINITY); hi = std::nextafter(hi, INFINITY); }
            if (radians(lo) == pc->mFOV) { deg = lo; found = true; }
            else if (radians(hi) == pc->mFOV) { deg = hi; found = true; }
        }
        const float rounded = std::round(deg * 1000.0f) / 1000.0f;   // what a scene file would say
        if (radians(rounded) == pc->mFOV) deg = rounded;
        d.camera.fov_degrees = deg;
    }
    d.camera.near_plane = pc->mZNear;
    d.camera.far_plane = pc->mZFar;
    d.camera.lens_radius = 0.0f;
    d.camera.focal_distance = pc->mFocalDistance;
    d.camera.type = GBL_CAMERA_PERSPECTIVE;
    d.camera.film_width = 35.0f;

    const Film* film = cam->getFilm();
    d.film.xres = film->getXResolution();
    d.film.yres = film->getYResolution();
    memcpy(d.film.crop, film->mCrop, sizeof(d.film.crop));                                    // GoblinFilm.cpp:92-104
    d.film.tone_mapping = film->mToneMapping ? 1u : 0u;
    d.film.bloom_radius = film->mBloomRadius;
    d.film.bloom_weight = film->mBloomWeight;
    const Filter* flt = film->mFilter;
    d.film.filter_width[0] = flt->getXWidth();
    d.film.filter_width[1] = flt->getYWidth();
    d.film.gaussian_falloff = 2.0f;
    d.film.mitchell_b = d.film.mitchell_c = 1.0f / 3.0f;
    if (const GaussianFilter* gf = dynamic_cast<const GaussianFilter*>(flt)) {                // GoblinFilter.h:47-61
        d.film.filter_type = GBL_FILTER_GAUSSIAN;
        d.film.gaussian_falloff = gf->mAlpha;
    } else if (const MitchellFilter* mf = dynamic_cast<const MitchellFilter*>(flt)) {
        d.film.filter_type = GBL_FILTER_MITCHELL;
        d.film.mitchell_b = mf->mB;
        d.film.mitchell_c = mf->mC;
    } else if (dynamic_cast<const TriangleFilter*>(flt)) {
        d.film.filter_type = GBL_FILTER_TRIANGLE;
    } else {
        d.film.filter_type = GBL_FILTER_BOX;
    }

    d.setting = setting;
    d.num_vertices = static_cast<uint32_t>(out->positions.size() / 3);
    d.positions = out->positions.data();
    d.normals = out->normals.data();
    d.uvs = out->uvs.data();
    d.num_triangles = static_cast<uint32_t>(out->indices.size() / 3);
    d.indices = out->indices.data();
    d.num_meshes = static_cast<uint32_t>(out->meshes.size());
    d.meshes = out->meshes.data();
    d.num_materials = static_cast<uint32_t>(out->materials.size());
    d.materials = out->materials.data();
    d.num_instances = static_cast<uint32_t>(out->instances.size());
    d.instances = out->instances.data();
    d.num_lights = static_cast<uint32_t>(out->lights.size());
    d.lights = out->lights.data();
    d.volume.type = GBL_VOLUME_NONE;
}
}   // namespace Goblin
