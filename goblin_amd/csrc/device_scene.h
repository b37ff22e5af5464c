// Device-resident scene layout (HBM) shared by the host packer (scene_prep.cpp)
// and the kernels.  Everything is POD and sized/aligned for 16-byte vector loads.
//
//   nodes[]      64 B  4-wide BVH node, 8-bit quantised child boxes (one fetch = four
//                      box tests).  BLAS nodes of every mesh first, then the TLAS
//                      nodes; all child references are absolute.
//   tris[]       48 B  p0, e1, e2 in object space, in BLAS leaf order; e1/e2 are
//                      the same float32 differences the reference forms per test
//                      (GoblinTriangle.cpp:52-53) so Moller-Trumbore is bit-equal.
//   tri_shade[]  16 B  vertex ids for normal/uv interpolation at the final hit.
//   instances[] 128 B  3x4 toWorld, 3x4 inverse (Transform::update order), ids.
#pragma once
#include <stdint.h>

#define GBL_TILE 8              // sample tiles are 8x8 pixels, as Renderer::getSampleRanges (GoblinRenderer.cpp:654)
#define GBL_BLOCK 256           // threads per workgroup (4 waves of 64)
#define GBL_MAX_LEAF_TRIS 4
#define GBL_STACK_SENTINEL 0x7fffffff
#ifndef GBL_PT_WAVES
#define GBL_PT_WAVES 3
#endif
#ifndef GBL_EXT_WAVES
#define GBL_EXT_WAVES 2         // waves per SIMD the EXT instantiations of the persistent kernels are compiled for (256 registers)
#endif
#define GBL_MAX_FILTER_HALO 6   // LDS film tile is (8 + 2*halo)^2 pixels
#define GBL_STREAM_LDS_WORDS 1248   // GBL_SAMPLES_STREAM: two blocks of 624 mt19937 state words (kernels/stream.h)
#define GBL_WHITTED_MAX_DEPTH 12   // frames of the Whitted kernel's explicit recursion (kernels/whitted.h)

// child reference: >= 0 interior node index; < 0 leaf: ~ref = (first << 2) | (count - 1)
// (TLAS leaves: first = instance id, count = 1); GBL_REF_NONE marks an unused child slot.
//
// 4-wide BVH node with 8-bit quantised child boxes, 64 bytes = four 16-byte loads per
// traversal step for FOUR box tests.  A child's box on axis a is
//   [o_a + qlo[a][c] * scale_a,  o_a + qhi[a][c] * scale_a]
// rounded outwards, so the test is conservative: it can only add candidates, never lose
// a triangle the exact test would accept.
struct DevNode {
    float o[3];           // quantisation origin (the node's lower corner, nudged down)
    float scale[3];       // grid step per axis, a power of two
    uint32_t qlo[3];      // per axis: 4 child bytes, child c in byte c
    uint32_t qhi[3];
    int32_t child[4];     // unused slots: GBL_REF_NONE with qlo = 255, qhi = 0 (an empty interval for every ray)
};
#define GBL_REF_NONE 0x7ffffffd
#define GBL_WF_HOT_NODES 128      // nodes the wavefront trace kernels keep in LDS (measured: see DESIGN.md 9)
#define GBL_HOT_NODES_MAX 512   // most nodes of the breadth-first prefix (scene_prep.cpp hot prefix; 32 KB)
// BLAS "roots" of analytic shapes: leaf references whose first-triangle field is out of range
#define GBL_SHAPE_FIRST_SPHERE 0x1fffffffu
#define GBL_SHAPE_FIRST_DISK 0x1ffffffeu
#define GBL_REF_SPHERE (~static_cast<int32_t>(GBL_SHAPE_FIRST_SPHERE << 2))
#define GBL_REF_DISK (~static_cast<int32_t>(GBL_SHAPE_FIRST_DISK << 2))

struct DevTri {
    float p0[3];
    uint32_t shade;   // index into tri_shade / original triangle id within the scene
    float e1[3];
    uint32_t flags;   // DevTriShade::flags again (bit 0: vertex normals, bit 1: uvs): a plain mesh's hit needs no tri_shade fetch
    float e2[3];
    float pad1;
};

struct DevTriShade {
    uint32_t v[3];    // absolute vertex ids into normals / uvs
    uint32_t flags;   // bit0 has_normal, bit1 has_uv
};

// Where a triangle sits in the REFERENCE's own BLAS over its mesh (BVH::buildLinearBVH, GoblinBVH.cpp:81-151: median
// split by std::nth_element along the longest axis of the centroid bounds, one triangle per leaf).  The device does not
// traverse that tree, but its visiting order -- near child first by the ray's sign on the split axis -- decides which of
// two triangles wins when a ray meets their shared edge at exactly the same t (`t <= maxt`, the later test wins).
struct DevTriOrder {
    uint32_t path;       // bit l: the triangle is in the second child at depth l (bit 0 = the root's split)
    uint32_t axes_lo;    // 2 bits per depth: the split axis there (depths 0-15)
    uint32_t axes_hi;    // depths 16-31
    uint32_t depth_rank; // depth of its leaf | position inside a multi-triangle leaf << 8
};

struct DevInstance {
    float m[12];      // toWorld rows 0..2 (3x4)
    float inv[12];    // inverse rows 0..2 (3x4)
    int32_t root;     // BLAS root child reference
    int32_t material;
    int32_t area_light;
    int32_t mesh;
    uint32_t shape;     // gbl_shape: 0 triangles below `root`; sphere / disk are tested analytically (root = GBL_REF_SPHERE/DISK)
    float radius;
    uint32_t is_mask;   // the material is a MaskMaterial (its type carries BSDFnullptr): what isOpaque / notOpaque filter on
    int32_t pad;
};
// InstancedPrimitive's world bound (Transform::onBBox of the mesh bound): the reference's TLAS leaf box.  Kept apart
// from DevInstance (128 B, one cache line per traversal step); only the medium's self-occlusion rule reads it.
struct DevInstanceBound {
    float lo[3], hi[3];
};
// Triangle::getObjectBound (the three vertices' bound): the reference BLAS's leaf box, one triangle per leaf.  Indexed like
// `tris` (a hit's triangle number finds it in one load); read by the TIES builds only (ref_reached / trace_needs_redo, kernels/trace.h).
struct DevTriBound {
    float lo[3], hi[3];
    float pad[2];
};
// IntersectFilter of a scene query (GoblinPathtracer.cpp:5-11): none, isOpaque (skip masks), notOpaque (masks only)
#define GBL_FILTER_NONE 0
#define GBL_FILTER_OPAQUE 1
#define GBL_FILTER_MASK 2

struct DevMaterial {
    uint32_t type;
    float color[3];
    float color2[3];
    float index, k, exponent;
    int32_t tex_color, tex_color2, tex_exponent;   // -1: the constant above; else an index into DevScene::textures
    uint32_t has_tex;                               // any of the three >= 0 (for a mask: also the wrapped material's)
    int32_t masked;                                 // type 4 (mask): the wrapped material; alpha = exponent, transparent colour = color
    // type 5 (subsurface): color = sigma_a, color2 = sigma_s', color3 = Kr, index = eta, k = g, exponent = A = (1 + Fdr) / (1 - Fdr)
    float color3[3];
    int32_t tex_color3;
    int32_t tex_bump, tex_normal;                   // BumpShaders: float displacement / colour normal map, -1 none (a mask carries the wrapped material's)
    float pad;
};

// One node of a procedural texture graph (gbl_texture).  Graphs are at most GBL_TEX_MAX_DEPTH levels deep below a
// material slot (checked when the scene is packed), so the device evaluates them by bounded template recursion.
#define GBL_TEX_MAX_DEPTH 2
struct DevTexture {
    uint32_t type, is_float;
    float value[3];
    int32_t child[2];
    uint32_t mapping, filter;
    float uv_scale[2], uv_offset[2];
    float to_tex[12];   // spherical mapping: toTex 3x4
    int32_t image;      // type 3 (image): index into DevScene::images; `filter` is then the gbl_image_filter
    uint32_t address;   // gbl_address_mode
    float max_aniso;
    float pad[2];
};

// One MIPMap<T> (gbl_image): level l is max(1, width >> l) x max(1, height >> l) texels of `channels` floats at
// texels[offset + level_offset[l]]
struct DevImage {
    uint32_t width, height, levels, channels;
    uint64_t offset;
    uint32_t level_offset[18];
};

// one emitting triangle of an area light, light-local space (GeometrySet, GoblinLight.cpp:289-343)
struct DevLightTri {
    float p0[3], area;
    float p1[3], cdf_lo;   // normalised area CDF value at this triangle's start (CDF1D::mCDF[i])
    float p2[3], cdf_hi;
    float n0[3], has_normal;
    float n1[3], pad0;
    float n2[3], pad1;
};

struct DevLight {
    uint32_t type;
    float color[3];
    float pos[3];
    float cos_max;
    float axis[3];          // spot: toWorld.onVector(UnitZ)
    float cos_falloff;
    float m[12], inv[12];   // area: light toWorld / inverse
    uint32_t tri_first, tri_count;   // into light_tris
    float sum_area;
    uint32_t shape;         // area: gbl_shape of the emitting geometry
    float radius;           // area: sphere / disk radius
    uint32_t wh_n;          // Whitted quota: roundToSquare(Light::getSamplesNum()) slots of this light's patterns ...
    uint32_t wh_prefix;     // ... and the slots of the lights before it
    // type 4 (image based light): m / inv = the light's rotation; the radiance MIPMap; its CDF2D (kernels/image.h)
    int32_t image;
    uint32_t dist_offset, dist_w, dist_h;   // into DevScene::ibl_dist
    float pad;
};

// HomogeneousVolumeRegion (GoblinVolume.h:72-112) + what Light::samplePosition needs of the scene (its bounding sphere)
struct DevVolume {
    uint32_t on;
    float attenuation[3];
    float scatter[3];       // attenuation * albedo
    float emission[3];
    float g;
    int32_t sample_num;
    float lo[3], hi[3];
    float m[12], inv[12];
    float bound_center[3];  // Scene::getBoundingSphere: centre of the scene bound ...
    float bound_radius;     // ... and its full diagonal (GoblinBBox.h:51-54)
    // HeterogeneousVolumeRegion (GoblinVolume.cpp:283-341): sigma_t from DevScene::vol_density, ray marched
    uint32_t hetero;
    float step;             // ray marching step (world units)
    float albedo[3];
    int32_t nx, ny, nz, nch;
    float normalize[3];     // VolumeGrid::mNormalizeTerm: 1 / (box extent)
    float pad[3];
};

struct DevCamera {
    float pos[3];
    float proj00;
    float q[4];       // w x y z
    float proj11, inv_xres, inv_yres;
    uint32_t type;    // gbl_camera_type
    float lens_radius, focal_distance, film_w, film_h;   // thin lens; orthographic film size in world units
};

struct DevFilm {
    int32_t xres, yres;
    int32_t xstart, ystart, xcount, ycount;   // crop rect (Film ctor, GoblinFilm.cpp:101-104)
    float wx, wy;                              // filter half widths
    int32_t window[4];                         // full sample window x0 x1 y0 y1
    int32_t halo;                              // LDS tile halo in pixels
    int32_t pad[3];
};

struct DevScene {
    const DevNode* nodes;
    const DevTri* tris;
    const DevTriShade* tri_shade;
    const DevTriBound* tri_bounds;
    const DevTriOrder* tri_order;   // per original triangle id (DevTri::shade); null: ties fall to the device's own order
    const float* positions;         // 3 per vertex, as given: the reference's leaf boxes in a tie (trace.h tie_goes_to)
    const float* normals;    // 3 per vertex
    const float* uvs;        // 2 per vertex
    const DevInstance* instances;
    const DevInstanceBound* instance_bounds;
    const DevMaterial* materials;
    const DevTexture* textures;
    const DevLight* lights;
    const DevLightTri* light_tris;
    const float* light_cdf;       // num_lights + 1, normalised (CDF1D::mCDF)
    const float* light_pick_pdf;  // per light: (f[i] / integral) * dx
    const float* filter_table;    // 256 floats
    const DevImage* images;       // MIP pyramids (image textures, image based lights) ...
    const float* texels;          // ... and their texels
    const float* ewa_lut;         // MIPMap::EWALut, 128 floats
    const float* ibl_dist;        // the image based lights' CDF2Ds
    const float* vol_density;     // heterogeneous medium: data[((z * ny + y) * nx + x) * nch + c]
    int32_t tlas_root;            // child reference of the TLAS root
    int32_t num_instances;
    int32_t num_lights;
    int32_t stack_entries;        // traversal stack depth this scene needs
    int32_t extended;             // scene uses analytic shapes, a directional light or a non-pinhole camera: EXT kernels
    int32_t has_masks;            // some instance carries a mask material: filtered queries + attenuation walks (megakernel only)
    int32_t has_bssrdf;           // some material is a subsurface material: sss_kernel runs ahead of the path kernels
    int32_t wh_slots;             // Whitted quota: sum of the lights' wh_n
    int32_t has_ibl;              // some light is image based: rays that leave the scene collect Le (evalEnvironmentLight)
    uint32_t hot_nodes;           // nodes[0 .. hot_nodes) are the top of the two-level tree, breadth first (scene_prep.cpp)
    DevCamera camera;
    DevFilm film;
    DevVolume volume;
};

#define GBL_PRIM_MISS (-1)   // RenderArgs::prim_inst of a camera ray that left the scene
#define GBL_PRIM_TIED (-2)   // ... of one that met two triangles at exactly its closest distance: the path kernel traces it itself
struct RenderArgs {
    int32_t integrator;
    int32_t spp, root;          // roundToSquare(sample_per_pixel) and its root
    int32_t max_depth;
    int32_t ao_n;               // AO directions per camera sample
    int32_t dims;               // floats per Sample record
    int32_t off2_base;          // float offset of the first 2D pattern in a record
    int32_t window[4];          // sub-window being rendered x0 x1 y0 y1
    int32_t tiles_x, tiles_y;
    int32_t shard_index, shard_count;   // this launch owns tiles t with t % count == index
    int32_t local_tiles;        // number of tiles this launch owns
    int32_t chunks;             // each tile's spp split in `chunks` work items
    int32_t chunk_spp;
    uint32_t seed_key;
    uint32_t russian_roulette;
    int32_t bssrdf_n;           // BSSRDFSampleIndex::samplesNum: roundToSquare(bssrdf_sample_num)
    int32_t bssrdf_n2;          // size of its 2D patterns (rounded to a square once more)
    const float* sss;           // per-sample Lsubsurface of this render (float4, pixel-major like li_out), or null
    float* vol;                 // per camera sample {transmittance.rgb, -, Lv.rgb, -} of the medium (2 float4), or null
    // GBL_SAMPLES_STREAM (kernels/stream.h)
    const uint32_t* tile_seeds; // the reference's per-tile mt19937 seeds, row-major over the FULL sample window's tiles
    uint32_t* stream_scratch;   // stream_stride words per workgroup
    uint64_t stream_stride;
    uint32_t sss_off1, sss_off2;   // the BSSRDF block of a camera sample's record: first 1D / first 2D float (kernels/bssrdf.h)
    uint32_t sss_pat1, sss_pat2;   // ... and its first 1D / 2D pattern number under the native sampler
    uint32_t stream_tail_cap;   // words of a workgroup's scratch behind its records that hold a pixel's medium draws (>= one sample's)
    int32_t full_tiles_x;
    uint32_t stream_lperm_words;   // LDS words behind the traversal stacks' base the shuffles may use (>= the stacks' own)
    float* image_xy;            // per camera sample (indexed like li_out): the image position its record held, for the splat
    const float* replay;        // Sample records for the sub-window, pixel-major
    const float* prim_hit;      // primary pass (kernels/packet.h): per camera sample {t, b1, b2, as_float(tri)} of the camera ray's hit ...
    uint32_t* prim_items;       // ... per work item of the path kernel: nonzero when one of its camera rays did not simply miss (zeroed per call)
    const int32_t* prim_inst;   // ... and its instance, or GBL_PRIM_MISS / GBL_PRIM_TIED; null: the path kernel traces the camera rays itself
    uint32_t hot_count, hot_word;   // quad kernels: nodes[0 .. hot_count) also live in LDS, at word hot_word of the workgroup's block
    float* li_out;
    float* li_defer;            // when set, the render kernel stores per-sample radiance here (pixel-major)
                                // and a separate splat kernel filters it into the film afterwards
    float* film;                // xres*yres float4 accumulators
    uint32_t* work_counter;     // zeroed before each launch
    unsigned long long* stats;  // 8 counters
};
