/*
 * goblin_hip.h -- C ABI of the MI355X-native path-tracing integrator.
 *
 * This is the drop-in boundary behind Goblin's Renderer seam.  The reference
 * has no FFI layer: the seam is the C++ virtual `Renderer::render(ScenePtr)`
 * (/root/reference/src/GoblinRenderer.h:55-59, GoblinRenderContext.h:19-22)
 * selected by `createRenderer` (GoblinContextLoader.cpp:67-92).  A GPU renderer
 * replaces that one call: scene arrays in, Film accumulation buffer out.
 *
 * Everything here is plain C: POD structs, pointers + counts, int status
 * codes, no exceptions, caller-owned output buffers.  INTEGRATION.md shows the
 * `HipPathTracer : Renderer` stub a Goblin maintainer would add on top.
 *
 * Two shared libraries implement it:
 *   libgoblin_host.so  (g++,   no HIP dependency)  gbl_host_*  scene front end
 *   libgoblin_hip.so   (hipcc, gfx950)             gbl_*       device integrator
 */
#ifndef GOBLIN_HIP_H
#define GOBLIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBL_ABI_VERSION 14

typedef enum gbl_status {
    GBL_OK = 0,
    GBL_ERR_INVALID = 1,     /* bad argument / inconsistent description       */
    GBL_ERR_UNSUPPORTED = 2, /* scene uses a Goblin feature outside the path  */
    GBL_ERR_IO = 3,          /* file missing / parse error                    */
    GBL_ERR_DEVICE = 4,      /* HIP runtime error, no device                  */
    GBL_ERR_OOM = 5,
    GBL_ERR_INTERNAL = 6     /* a C++ exception stopped at the boundary (message in *_last_error) */
} gbl_status;

/* ------------------------------------------------------------------------- */
/* Scene description (host memory, owned by the caller for the duration of    */
/* gbl_create).  Mirrors what ContextLoader builds (GoblinContextLoader.cpp   */
/* :447-504) restricted to the hot path: triangle meshes, instances,          */
/* constant-texture materials, point/spot/area lights, perspective camera.   */
/* ------------------------------------------------------------------------- */

/* position / orientation(w,x,y,z) / scale, as parsed by getTransform
 * (GoblinUtils.cpp:78-100) and composed by Transform::update
 * (GoblinTransform.cpp:182-193). */
typedef struct gbl_trs {
    float position[3];
    float orientation[4]; /* quaternion w, x, y, z */
    float scale[3];
} gbl_trs;

typedef enum gbl_shape {
    GBL_SHAPE_MESH = 0,   /* PolygonMesh, refined into triangles (GoblinPolygonMesh.cpp) */
    GBL_SHAPE_SPHERE = 1, /* Sphere(radius), intersected analytically (GoblinSphere.cpp:12-150) */
    GBL_SHAPE_DISK = 2    /* Disk(radius) in the local z = 0 plane (GoblinDisk.cpp:12-91)     */
} gbl_shape;

/* One Geometry.  A PolygonMesh after OBJ loading and (v,vn,vt) de-duplication
 * (GoblinPolygonMesh.cpp:58-262) -- vertex attributes live in the shared
 * arrays of gbl_scene_desc, indices are mesh-local -- or an analytic shape
 * (createGeometries, GoblinContextLoader.cpp:210-243), for which only
 * `shape` and `radius` are read. */
typedef struct gbl_mesh {
    uint32_t vertex_offset; /* first vertex in positions/normals/uvs          */
    uint32_t vertex_count;
    uint32_t tri_offset;    /* first triangle in indices (3 uint32 each)      */
    uint32_t tri_count;
    uint32_t has_normal;    /* PolygonMesh::hasNormal()                       */
    uint32_t has_uv;        /* PolygonMesh::hasTexCoord()                     */
    uint32_t shape;         /* gbl_shape                                      */
    float radius;           /* sphere / disk ("radius", default 1.0)          */
} gbl_mesh;

typedef enum gbl_material_type {
    GBL_MAT_LAMBERT = 0,     /* GoblinMaterial.cpp:437-480 */
    GBL_MAT_BLINN = 1,       /* GoblinMaterial.cpp:540-644 */
    GBL_MAT_TRANSPARENT = 2, /* GoblinMaterial.cpp:647-706 */
    GBL_MAT_MIRROR = 3,      /* GoblinMaterial.cpp:709-726 */
    GBL_MAT_MASK = 4,        /* GoblinMaterial.cpp:747-811: alpha-masked wrapper, BSDFnullptr punch-through */
    GBL_MAT_SUBSURFACE = 5   /* GoblinMaterial.cpp:728-745 + BSSRDF :32-220; Renderer::Lsubsurface, GoblinRenderer.cpp:128-296 */
} gbl_material_type;

typedef enum gbl_texture_type {
    GBL_TEX_CONSTANT = 0,     /* ConstantTexture,   GoblinTexture.cpp:349-354, 617-625 */
    GBL_TEX_CHECKERBOARD = 1, /* CheckboardTexture, GoblinTexture.cpp:356-416, 627-647 */
    GBL_TEX_SCALE = 2,        /* ScaleTexture,      GoblinTexture.cpp:418-425, 649-665 */
    GBL_TEX_IMAGE = 3         /* ImageTexture over a MIPMap, GoblinTexture.cpp:428-470, 40-291, 667-745 */
} gbl_texture_type;

typedef enum gbl_image_filter {   /* "filter" of an image texture (GoblinTexture.cpp:687-700); MIPMap<T>::lookup :78-97 */
    GBL_IMAGE_FILTER_NONE = 0,      /* "nearest": bilinear at level 0 (lookupNearest, :99-102) */
    GBL_IMAGE_FILTER_BILINEAR = 1,  /* one level, rounded from the footprint width (:104-110) */
    GBL_IMAGE_FILTER_TRILINEAR = 2, /* two levels blended (:112-127) */
    GBL_IMAGE_FILTER_EWA = 3        /* elliptically weighted average (:129-258) */
} gbl_image_filter;

typedef enum gbl_address_mode {   /* ImageBuffer<T>::texel, GoblinTexture.cpp:10-38 */
    GBL_ADDRESS_REPEAT = 0,
    GBL_ADDRESS_CLAMP = 1,   /* as written there: t is clamped from s (:15) */
    GBL_ADDRESS_BORDER = 2
} gbl_address_mode;

/* One MIPMap<T> (GoblinTexture.cpp:40-68): the image a texture or an image based light reads, already converted
 * (ImageTexture::convertTexel: channel pick and gamma, :489-523; the light's colour filter, GoblinLight.cpp:483-485),
 * resized to powers of two and reduced level by level with resizeImage's gaussian (:531-597).  Texels are
 * `channels` floats each (1 for float textures, 4 rgba otherwise), level 0 first, each level row-major. */
typedef struct gbl_image {
    uint32_t width, height;   /* level 0 */
    uint32_t levels;          /* floor(max(log2 w, log2 h)) + 1; level l is max(1, w >> l) x max(1, h >> l) */
    uint32_t channels;
    uint64_t texel_offset;    /* in floats, into gbl_scene_desc.texels */
} gbl_image;

typedef enum gbl_mapping_type {
    GBL_MAP_UV = 0,       /* UVMapping,        GoblinTexture.cpp:293-304 */
    GBL_MAP_SPHERICAL = 1 /* SphericalMapping, GoblinTexture.cpp:306-347 */
} gbl_mapping_type;

/* One texture of the scene's "textures" list ("format": color | float). */
typedef struct gbl_texture {
    uint32_t type;       /* gbl_texture_type                                            */
    uint32_t is_float;   /* "format" == "float"                                         */
    float value[3];      /* constant: "color", or "float" in value[0]                   */
    int32_t child[2];    /* checkerboard: texture1, texture2 (same format);
                          * scale: "texture" (same format), "scale" (float format)      */
    uint32_t mapping;    /* checkerboard: gbl_mapping_type ("mapping", default uv)      */
    float uv_scale[2];   /* uv mapping "scale" (default 1, 1)                           */
    float uv_offset[2];  /* uv mapping "offset" (default 0, 0)                          */
    gbl_trs to_tex;      /* spherical mapping: getTransform(params)                     */
    uint32_t filter;     /* checkerboard "filter" (bool, default false)                 */
    /* image textures (getImageTextureParams, GoblinTexture.cpp:677-732); mapping / uv_* / to_tex as above */
    int32_t image;          /* index into gbl_scene_desc.images, -1 otherwise                 */
    uint32_t image_filter;  /* gbl_image_filter                                                */
    uint32_t address;       /* gbl_address_mode                                                */
    float max_anisotropy;   /* "max_anisotropy" (default 10; colour image textures always use the default, :741-745) */
} gbl_texture;

/* Materials.  Each texture slot is either a constant (tex_* == -1: the value is
 * in color / color2 / exponent, createColorConstantTexture GoblinTexture.cpp:
 * 622-625) or an index into gbl_scene_desc.textures.  Colours are rgb; alpha is
 * 1 as in Color(r,g,b). */
typedef struct gbl_material {
    uint32_t type;     /* gbl_material_type                                    */
    float color[3];    /* Lambert Kd | Blinn Kg | Transparent Kr | Mirror Kr   */
    float color2[3];   /* Transparent Kt                                       */
    float index;       /* eta: blinn/transparent default 1.5, mirror 0.8       */
    float k;           /* absorption: blinn conductor iff > 0; mirror 6.0      */
    float exponent;    /* blinn exponent (float texture) | Mask alpha          */
    int32_t tex_color, tex_color2, tex_exponent; /* -1: constant above          */
    /* Mask: color = "transparent_color" (default white), exponent = "alpha"
     * (default 1), masked_material = index of the wrapped material (which must
     * not be a mask itself); -1 for every other type. */
    int32_t masked_material;
    /* Subsurface (createSubsurfaceMaterial, GoblinMaterial.cpp:881-927): color = "absorb" sigma_a, color2 =
     * "scatter_prime" sigma_s', color3 = "Kr" (default white), index = eta (default 1.5), k = "g" (default 0).  The
     * "Kd" + "mean_free_path" form is converted to sigma_a / sigma_s' by the loader exactly as the reference's
     * constructor does (BSSRDF::convertFromDiffuse, :176-212).  The material's type is BSDFAll, which carries the
     * BSDFnullptr bit: the path tracer's isOpaque / notOpaque filters treat it like a mask (GoblinPathtracer.cpp:5-11,
     * GoblinMaterial.h:393). */
    float color3[3];
    int32_t tex_color3;
    /* BumpShaders (getBumpShaders, GoblinMaterial.cpp:813-824; evaluated by Material::perturb at every closest hit,
     * GoblinScene.cpp:75-83, GoblinMaterial.cpp:221-283): "bumpmap" = a FLOAT texture displacing the surface along its
     * normal (forward differences over du = dv = 0.002), "normalmap" = a COLOUR texture holding 2 n - 1 in the shading
     * frame.  -1: none.  A mask forwards to the wrapped material's. */
    int32_t tex_bump, tex_normal;
} gbl_material;

/* InstancedPrimitive over a Model(geometry, material[, areaLight])
 * (GoblinPrimitive.cpp:99-112, GoblinModel.cpp:10-26). */
typedef struct gbl_instance {
    uint32_t mesh;
    uint32_t material;
    int32_t area_light; /* index into lights, -1 if not emissive              */
    gbl_trs to_world;
} gbl_instance;

/* Values follow Light::Type (GoblinLight.h:62-68). */
typedef enum gbl_light_type {
    GBL_LIGHT_POINT = 0,       /* GoblinLight.cpp:78-134  */
    GBL_LIGHT_DIRECTIONAL = 1, /* GoblinLight.cpp:136-210 */
    GBL_LIGHT_SPOT = 2,        /* GoblinLight.cpp:212-287 */
    GBL_LIGHT_AREA = 3,  /* GoblinLight.cpp:345-461 */
    GBL_LIGHT_IBL = 4    /* ImageBasedLight, GoblinLight.cpp:464-629 */
} gbl_light_type;

typedef struct gbl_light {
    uint32_t type;           /* gbl_light_type                                 */
    float color[3];          /* intensity (point/spot) or radiance (directional/area) */
    float position[3];       /* point/spot                                     */
    float direction[3];      /* spot/directional: as given to the light's ctor */
    float cos_theta_max;     /* spot: cos(radians(theta_max))                  */
    float cos_falloff_start; /* spot                                           */
    uint32_t mesh;           /* area: emitting geometry                        */
    gbl_trs to_world;        /* area                                           */
    uint32_t sample_num;     /* area / ibl: "sample_num" (default 1) = Light::getSamplesNum, read by the Whitted
                              * renderer's per-light quota (GoblinLight.cpp:675, GoblinWhitted.cpp:60-66); 1 for the rest */
    int32_t image;           /* ibl: the radiance map's MIPMap (already multiplied by "filter"), index into images;
                              * to_world.orientation holds "orientation" as given (the constructor's rotateX / rotateY
                              * pre-rotation is applied by gbl_create, GoblinLight.cpp:470-474) */
} gbl_light;

typedef enum gbl_camera_type {
    GBL_CAMERA_PERSPECTIVE = 0, /* GoblinCamera.cpp:83-148, 377-387 */
    GBL_CAMERA_ORTHOGRAPHIC = 1 /* GoblinCamera.cpp:288-326, 390-398 */
} gbl_camera_type;

/* PerspectiveCamera (pinhole, or thin lens when lens_radius != 0: the loader
 * then also adds the lens Disk instance, GoblinContextLoader.cpp:146-176) or
 * OrthographicCamera. */
typedef struct gbl_camera {
    float position[3];
    float orientation[4]; /* w, x, y, z */
    float fov_degrees;
    float near_plane, far_plane;
    float lens_radius, focal_distance;
    uint32_t type;        /* gbl_camera_type */
    float film_width;     /* orthographic ("film_width", default 35) */
} gbl_camera;

typedef enum gbl_filter_type {
    GBL_FILTER_BOX = 0,
    GBL_FILTER_TRIANGLE = 1,
    GBL_FILTER_GAUSSIAN = 2,
    GBL_FILTER_MITCHELL = 3
} gbl_filter_type;

/* Film + reconstruction filter (GoblinFilm.cpp:92-112,202-218,
 * GoblinFilter.h:85-106). */
typedef struct gbl_film {
    int32_t xres, yres;
    float crop[4]; /* x0 x1 y0 y1 fractions */
    uint32_t filter_type;
    float filter_width[2];
    float gaussian_falloff;
    float mitchell_b, mitchell_c;
    /* Film::writeImage post-processing (GoblinFilm.cpp:164-198, 212-215); host side only */
    uint32_t tone_mapping;
    float bloom_radius, bloom_weight;
} gbl_film;

typedef enum gbl_integrator {
    GBL_INTEGRATOR_PATH = 0, /* PathTracer  (GoblinPathtracer.cpp) */
    GBL_INTEGRATOR_AO = 1,   /* AORenderer  (GoblinAO.cpp)         */
    /* WhittedRenderer (GoblinWhitted.cpp:13-44): emission + multiSampleLd over every light (GoblinRenderer.cpp:474-596)
     * + Lsubsurface at every level (:25-27) + the specular reflection / refraction tree (:598-648); mask materials
     * answer its requests as MaskMaterial does (GoblinMaterial.cpp:747-811).  max_ray_depth <= 12. */
    GBL_INTEGRATOR_WHITTED = 2
} gbl_integrator;

/* render_setting block (GoblinPathtracer.cpp:210-217, GoblinAO.cpp:44-49). */
typedef struct gbl_render_setting {
    uint32_t integrator;
    int32_t sample_per_pixel;
    int32_t max_ray_depth;
    int32_t bssrdf_sample_num;
    int32_t ao_sample_num;
    int32_t thread_num; /* CPU paths only */
} gbl_render_setting;

/* The scene's participating medium ("volume", GoblinContextLoader.cpp:189-207).  RenderTask::run adds it around the
 * integrator's radiance for the CAMERA ray only: sample = transmittance * Li + Lv (GoblinRenderer.cpp:40-47), the ray
 * clipped at the first surface.  Homogeneous regions only (createHomogeneousVolume, GoblinVolume.cpp:343-360);
 * "heterogeneous" (a density grid file) is GBL_ERR_UNSUPPORTED. */
typedef enum gbl_volume_type {
    GBL_VOLUME_NONE = 0,
    GBL_VOLUME_HOMOGENEOUS = 1,
    GBL_VOLUME_HETEROGENEOUS = 2   /* HeterogeneousVolumeRegion (GoblinVolume.h:106-122): sigma_t from a density grid, ray marched */
} gbl_volume_type;
typedef struct gbl_volume {
    uint32_t type;          /* gbl_volume_type                                              */
    float attenuation[3];   /* sigma_t                                                      */
    float albedo[3];        /* sigma_s = attenuation * albedo                               */
    float emission[3];
    float g;                /* Henyey-Greenstein asymmetry ("g", default 0)                 */
    int32_t sample_num;     /* light samples along the camera ray ("sample_num", default 5) */
    float box_min[3], box_max[3]; /* the region, in its own space (heterogeneous: the grid's bounding box) */
    gbl_trs to_world;
    /* GBL_VOLUME_HETEROGENEOUS (createHeterogeneousVolume, GoblinVolume.cpp:362-384): attenuation / emission unused */
    float step_size;        /* ray marching step in world units ("step_size", default 0.1)   */
    int32_t grid[3];        /* cells along x, y, z of the density grid (.vol file, float32)  */
    int32_t grid_channels;  /* 1 or 3                                                        */
    const float* density;   /* data[((z * ny + y) * nx + x) * channels + c]                  */
} gbl_volume;

typedef struct gbl_scene_desc {
    uint32_t abi_version; /* GBL_ABI_VERSION */

    uint32_t num_vertices;
    const float* positions; /* 3 per vertex */
    const float* normals;   /* 3 per vertex (zeros where the mesh has none)   */
    const float* uvs;       /* 2 per vertex (zeros where the mesh has none)   */
    uint32_t num_triangles;
    const uint32_t* indices; /* 3 per triangle, mesh-local */

    uint32_t num_meshes;
    const gbl_mesh* meshes;
    uint32_t num_materials;
    const gbl_material* materials;
    uint32_t num_textures;
    const gbl_texture* textures; /* only the non-constant ones materials reach */
    uint32_t num_images;
    const gbl_image* images;     /* MIP pyramids of image textures and image based lights */
    uint64_t num_texels;         /* floats */
    const float* texels;
    uint32_t num_instances;
    const gbl_instance* instances; /* in SceneCache::getInstances() order */
    uint32_t num_lights;
    const gbl_light* lights;       /* in SceneCache::getLights() order    */

    gbl_camera camera;
    gbl_film film;
    gbl_render_setting setting;
    gbl_volume volume;
} gbl_scene_desc;

/* ------------------------------------------------------------------------- */
/* libgoblin_host.so : scene front end (JSON + OBJ -> gbl_scene_desc).        */
/* Replaces ContextLoader::load (GoblinContextLoader.cpp:447-504) for the     */
/* subset above; same keys, defaults and int-vs-float strictness.             */
/* ------------------------------------------------------------------------- */

typedef struct gbl_host_scene gbl_host_scene;

/* Load a Goblin scene file.  Relative mesh paths resolve against the file's
 * directory (SceneCache::resolvePath, GoblinScene.cpp:236-243). */
gbl_status gbl_host_load_file(const char* json_path, gbl_host_scene** out);
/* Same, from JSON text; `scene_dir` plays the role of the file's directory. */
gbl_status gbl_host_load_string(const char* json_text, const char* scene_dir, gbl_host_scene** out);
/* The flattened description; valid until gbl_host_free. */
const gbl_scene_desc* gbl_host_desc(const gbl_host_scene* scene);
void gbl_host_free(gbl_host_scene* scene);
/* Message for the last failing gbl_host_* call on this thread. */
const char* gbl_host_last_error(void);

/* Film::getSampleRange (GoblinFilm.cpp:131-138): the film padded by the
 * filter radius.  out = {x0, x1, y0, y1}, half-open. */
void gbl_host_sample_window(const gbl_film* film, int32_t out[4]);
/* roundToSquare (GoblinUtils.h:124-130): the spp the sampler really takes. */
int32_t gbl_host_round_to_square(int32_t n);
/* Floats per Sample record when the quota depends on the scene (the Whitted renderer's per-light patterns,
 * WhittedRenderer::querySampleQuota, GoblinWhitted.cpp:46-70); equals gbl_host_sample_dimension otherwise. */
int32_t gbl_host_sample_dimension_scene(const gbl_scene_desc* desc, const gbl_render_setting* rs);
/* Floats per Sample for an integrator: 4 + quota (GoblinPathtracer.cpp:181-208,
 * GoblinAO.cpp:39-42, GoblinSampler.h:27-31). */
int32_t gbl_host_sample_dimension(const gbl_render_setting* setting);
/* Film::writeImage's normalise step (GoblinFilm.cpp:164-172): rgb/weight.
 * accum = W*H float4 {sum w*L rgb, sum w}; rgb_out = W*H*3 floats. */
void gbl_host_film_normalize(const float* accum, int32_t xres, int32_t yres, float* rgb_out);
/* Portable float map writer (build-side extra: lossless, for parity checks). */
gbl_status gbl_host_write_pfm(const char* path, const float* rgb, int32_t xres, int32_t yres);
/* Film "file" of the loaded scene, or the reference's default <scene>.exr
 * (GoblinContextLoader.cpp:127-129, 474-484).  Valid until gbl_host_free. */
const char* gbl_host_output_path(const gbl_host_scene* scene);
/* Goblin::bloom (GoblinImageIO.cpp:169-218), in place on W*H*3 floats. */
void gbl_host_bloom(float* rgb, int32_t xres, int32_t yres, float bloom_radius, float bloom_weight);
/* Goblin::toneMapping (GoblinImageIO.cpp:220-236), in place. */
void gbl_host_tone_map(float* rgb, int32_t xres, int32_t yres);
/* writeImagePPM (GoblinImageIO.cpp:101-127): ASCII P3, gamma 2.2. */
gbl_status gbl_host_write_ppm(const char* path, const float* rgb, int32_t xres, int32_t yres);
/* writeImageEXR (GoblinImageIO.cpp:35-98): HALF channels B, G, R with tinyexr's
 * float->half rounding; scanlines are stored uncompressed (tinyexr defaults to
 * ZIP), so the pixels are the reference's, the bytes are not. */
gbl_status gbl_host_write_exr(const char* path, const float* rgb, int32_t xres, int32_t yres);
/* Goblin::writeImage (GoblinImageIO.cpp:146-167): picks the format from the
 * extension (.exr, .ppm with optional tone mapping, none/unknown -> <path>.ppm);
 * .pfm is a build-side extra.  May tone-map rgb in place. */
gbl_status gbl_host_write_image(const char* path, float* rgb, int32_t xres, int32_t yres, int32_t tone_mapping);
/* Goblin::loadImage (GoblinImageIO.cpp:14-34, 128-144): an .exr file as xres*yres float4, row-major, top row first,
 * assembled as tinyexr's LoadEXR does (one channel is replicated into all four; otherwise R, G, B must exist and A
 * defaults to 1).  Single-part scanline files, NONE / RLE / ZIPS / ZIP, HALF / FLOAT channels; anything else is
 * GBL_ERR_UNSUPPORTED, another extension GBL_ERR_IO like the reference's nullptr.  Free with gbl_host_free_image. */
gbl_status gbl_host_read_image(const char* path, float** rgba_out, int32_t* xres_out, int32_t* yres_out);
void gbl_host_free_image(float* rgba);

/* ------------------------------------------------------------------------- */
/* libgoblin_hip.so : the device integrator.                                  */
/* ------------------------------------------------------------------------- */

typedef struct gbl_ctx gbl_ctx;

typedef enum gbl_sample_mode {
    /* Counter-based device sampler with the reference's stratification law
     * (jittered strata x per-pixel sub-strata, permuted across the pixel's
     * samples: GoblinSampler.cpp:108-197), keyed by (seed, pixel, dim, k). */
    GBL_SAMPLES_NATIVE = 0,
    /* Caller uploads Sample records (device pointer): per sample
     * gbl_host_sample_dimension floats laid out {imageX, imageY, lensU1,
     * lensU2, u1D[0][..], u1D[1][..], ..., u2D[0][..], ...} exactly as
     * Sampler::requestSamples fills them; pixel-major, S samples per pixel,
     * pixels in row-major order over the sample window given in params. */
    GBL_SAMPLES_REPLAY = 1,
    /* The reference's own stream, generated on the device: one mt19937 per 8x8
     * sample tile (RNGImp, GoblinUtils.cpp:13-56) seeded with the tile's value
     * of the never-seeded libc rand() in row-major tile order (RenderTask,
     * GoblinRenderer.cpp:29-52, 99-126), Sampler::requestSamples per pixel
     * (GoblinSampler.cpp:108-197) and the three discarded floats of every
     * BSDFSample(rng) in PathTracer::Li (GoblinPathtracer.cpp:103,150,159).
     * A participating medium's draws follow each sample's Li draws in the
     * tile's stream (RenderTask::run, GoblinRenderer.cpp:43-46): carried too.
     * The Film accumulators then equal the reference binary's (glibc /
     * libstdc++ build) up to float summation order, with nothing uploaded.
     * A tile is sequential by construction -- this is the bit-faithful mode,
     * NATIVE the fast one.  All three integrators, megakernel schedule only;
     * the window must be whole tiles of the full sample window. */
    GBL_SAMPLES_STREAM = 2
} gbl_sample_mode;

/* How the device schedules the same arithmetic (identical per-sample radiance):
 *  WAVEFRONT   path pool in HBM, compacted ray queues, extend / shade / shadow kernels
 *  MEGAKERNEL  one persistent kernel, path state in registers, in-wave regeneration */
/* AUTO (path tracer): WAVEFRONT when the scene's paths are long -- at least RAYS_PER_PATH scene queries per camera path, measured
 * once per context and max_ray_depth by a one-sample pilot inside the first such gbl_render call (which therefore synchronises)
 * -- and the call (this rank's tiles x spp) holds at least PATHS camera samples; MEGAKERNEL otherwise and for mask scenes, AO,
 * Whitted and GBL_SAMPLES_STREAM (gbl_stats.schedule reports what a call ran under) */
#define GBL_AUTO_WAVEFRONT_RAYS_PER_PATH 6.0f
#define GBL_AUTO_WAVEFRONT_PATHS (1ull << 22)
typedef enum gbl_schedule {
    GBL_SCHEDULE_AUTO = 0,
    GBL_SCHEDULE_MEGAKERNEL = 1,
    GBL_SCHEDULE_WAVEFRONT = 2
} gbl_schedule;

typedef struct gbl_render_params {
    uint32_t integrator;      /* gbl_integrator                               */
    int32_t sample_per_pixel; /* rounded up to a square like the reference    */
    int32_t max_ray_depth;    /* PathTracer: loop runs max_ray_depth-1 times  */
    int32_t ao_sample_num;
    int32_t bssrdf_sample_num; /* only sizes the replay record (dims)         */
    /* sub-window of the sample window to render, half-open pixel coords; the
     * multi-GPU shard unit.  {0,0,0,0} = the whole Film::getSampleRange. */
    int32_t window[4];        /* x0, x1, y0, y1 */
    /* Interleaved tile sharding inside the window: this call renders only the
     * 8x8-pixel sample tiles t (row-major over the window) with
     * t % tile_shard_count == tile_shard_index.  count <= 1 renders every tile.
     * One GPU per shard + gbl_film_allreduce reproduces the whole film. */
    int32_t tile_shard_index;
    int32_t tile_shard_count;
    uint32_t sample_mode;     /* gbl_sample_mode                              */
    uint64_t seed;            /* native mode                                  */
    const float* replay_samples; /* device pointer, replay mode               */
    /* optional device pointer: per-sample radiance rgba as Li() returns it,
     * same order as the samples (replay/native), or NULL. */
    float* li_out;
    uint32_t russian_roulette; /* build-side extension; MUST be 0 for parity
                                  (the reference loop is fixed length,
                                  GoblinPathtracer.cpp:76)                    */
    uint32_t collect_stats;    /* fill node/triangle counters (slower)        */
    uint32_t schedule;         /* gbl_schedule                                */
    uint32_t exact_ties;       /* native sampler, scenes of the headline feature set (the lean kernels): follow the reference's
                                  BVH where two triangles are accepted at exactly the same t, and its non-watertight box tests
                                  where a ray grazes one.  Every other render does anyway (replay, stream,
                                  instrumented, scenes that need the EXT kernels); here it costs ~16 % for the few rays per
                                  10^6 it decides, so it is off by default                                              */
    void* stream;              /* hipStream_t, NULL = default stream          */
} gbl_render_params;

typedef struct gbl_stats {
    uint64_t paths;          /* camera samples = Li evaluations               */
    uint64_t extension_rays; /* closest-hit scene queries (incl. primary)     */
    uint64_t shadow_rays;    /* any-hit scene queries                         */
    uint64_t nodes;          /* child boxes tested (32 B each)                */
    uint64_t tris;           /* triangles tested (48 B each)                  */
    uint64_t splats;         /* film pixel updates                            */
    uint64_t dims;           /* sample dimensions consumed (floats)           */
    double kernel_ms;        /* HIP-event time of the render kernel(s)        */
    uint32_t schedule;       /* the gbl_schedule the call ran under (what GBL_SCHEDULE_AUTO resolved to; megakernel for AO / Whitted) */
    uint32_t reserved;
} gbl_stats;

/* Build the two-level BVH on the host, pack and upload the scene once
 * (replaces Scene/BVH/Model construction: GoblinScene.cpp:11-27,
 * GoblinBVH.cpp:34-151, GoblinModel.cpp:10-26).  `device` is a HIP ordinal. */
gbl_status gbl_create(const gbl_scene_desc* desc, int device, gbl_ctx** out);
/* Same with flags (GBL_CREATE_*, declared below).  gbl_create is gbl_create_ex
 * with flags 0, or GBL_CREATE_DEVICE_BVH when GBL_BVH_BUILD=device is set. */
gbl_status gbl_create_ex(const gbl_scene_desc* desc, int device, uint32_t flags, gbl_ctx** out);

/* Run the integrator over params->window and ACCUMULATE into film_accum, a
 * device buffer of xres*yres float4 {sum w*L.rgb, sum w} (the per-thread
 * ImageTile of the reference, GoblinFilm.cpp:61-90).  Replaces
 * Renderer::render's task loop (GoblinRenderer.cpp:99-126, 29-52).
 * Asynchronous on params->stream unless stats != NULL (then it synchronises
 * to read counters and timing). */
gbl_status gbl_render(gbl_ctx* ctx, const gbl_render_params* params, float* film_accum, gbl_stats* stats);

/* Sum film_accum across the ranks of an RCCL communicator (ncclComm_t passed
 * as void*): replaces Film::mergeTile under the TLS mutex
 * (GoblinFilm.cpp:140-153, GoblinThreadLocalStorage.h:69-75). */
gbl_status gbl_film_allreduce(gbl_ctx* ctx, void* rccl_comm, float* film_accum, void* stream);

/* Device-side Film::writeImage normalise: rgb_out[W*H*3] = rgb / weight. */
gbl_status gbl_film_resolve(gbl_ctx* ctx, const float* film_accum, float* rgb_out, void* stream);

/* Device time of recent gbl_render calls, from HIP events recorded on the render stream
 * around the dominant kernel and around the whole call (no host synchronisation happens
 * inside gbl_render for this).  out[0] is the most recent call.  Blocks until those events
 * have completed.  Returns the number of entries filled (at most 64 calls are remembered). */
typedef struct gbl_timing {
    double main_kernel_ms; /* path_trace_kernel / ao_kernel (megakernel schedule) or all
                              wf_* kernels of the call (wavefront schedule)            */
    double total_ms;       /* main kernel(s) + film splat kernel                        */
} gbl_timing;
int gbl_get_timings(gbl_ctx* ctx, int n, gbl_timing* out);

/* Scene facts the caller needs for buffer sizing and reporting. */
/* gbl_create_ex flags */
#define GBL_CREATE_DEVICE_BVH 1u /* build the triangle BLASes on the GPU (Morton-sorted linear BVH) instead of the
                                  * host's binned-SAH build: faster to construct, slower to trace */

typedef struct gbl_info {
    int32_t xres, yres;
    int32_t window[4];      /* full sample window x0,x1,y0,y1 */
    uint64_t blas_nodes, tlas_nodes, triangles, instances;
    uint64_t scene_bytes;   /* device bytes held by the scene */
    uint64_t instanced_triangles; /* sum over instances of their mesh's triangle count */
    double build_ms;        /* host wall time of scene packing + BVH build + geometry upload in gbl_create */
    int32_t blas_depth, tlas_depth; /* 4-wide levels */
} gbl_info;
gbl_status gbl_get_info(const gbl_ctx* ctx, gbl_info* out);

/* Instance edits: give instances [first, first + count) new transforms and rebuild the TLAS over all instances in
 * place (BLASes untouched; the reference would rebuild its whole scene BVH, GoblinScene.cpp:11-27).  Synchronises
 * the device.  Instances that carry an area light, and scenes with a directional or image based light (whose power and
 * sampling sphere depend on the scene bound), are GBL_ERR_UNSUPPORTED: re-create the context for those. */
gbl_status gbl_update_instances(gbl_ctx* ctx, uint32_t first, uint32_t count, const gbl_trs* to_world);

/* Self-test hook: the device's sinf / cosf (glibc's algorithm restated, kernels/refmath.h) on n device floats. */
gbl_status gbl_selftest_sincos(gbl_ctx* ctx, const float* in, float* sin_out, float* cos_out, uint64_t n);
/* Self-test hook: out[i] = fn(a[i] [, b[i]]) on n device floats, fn one of the libm functions the reference's sampling code
 * passes through (exp / log / log2 / pow / atan / atan2 / tan / acos on floats: glibc's algorithms restated, kernels/refmath.h).
 * b may be NULL for the one-argument functions. */
typedef enum gbl_libm_fn {
    GBL_LIBM_EXPF = 0, GBL_LIBM_LOGF = 1, GBL_LIBM_LOG2F = 2, GBL_LIBM_POWF = 3,
    GBL_LIBM_ATANF = 4, GBL_LIBM_ATAN2F = 5, GBL_LIBM_TANF = 6, GBL_LIBM_ACOSF = 7
} gbl_libm_fn;
gbl_status gbl_selftest_libm(gbl_ctx* ctx, int fn, const float* a, const float* b, float* out, uint64_t n);
/* Self-test hook: out[4 i ..] = {sqrtf(a), a / b, 1 / a, normalize(a, b, 0.5).x} on n device floats -- the basic
 * operations must round like the host's for per-sample radiance to be bit-identical with the reference. */
gbl_status gbl_selftest_arith(gbl_ctx* ctx, const float* a, const float* b, float* out, uint64_t n);
/* Self-test hook: scene queries for n device rays {kind (0 Scene::intersect, 1 Scene::occluded), o(3), d(3), mint, maxt}
 * -> out n x 8 floats {hit t or -1 | 1 occluded or 0, hit instance or -1, shading normal(3), tangent(3)}. */
gbl_status gbl_selftest_trace(gbl_ctx* ctx, const float* rays, float* out, uint32_t n);

/* Self-test hook: VALU issue-rate microbenchmark, no memory traffic.  Every CU runs one workgroup of 4 * waves_per_simd
 * waves (waves_per_simd 1..4 resident on each SIMD), every wave `iters` x 64 instructions of kind `op` over 16 independent
 * register chains.  The launch that is read follows two seconds of the same launch back to back (the chip settles its clock
 * under a load over seconds).  out[8] = {launch ms (HIP events), wave-instructions of the launch, s_memtime ticks per wave,
 * ticks per instruction of a wave, s_memrealtime (100 MHz) ticks per wave over the same span, shader clock in GHz = the ratio
 * of the two x 0.1, the longest and the shortest span of a wave in 100 MHz ticks}.  The waves of a SIMD start together but the
 * older one is served first, so they finish one after the other: a SIMD's issue rate follows from the LONGEST span (or the
 * launch time), not from the average one.  The peak the traversal kernels' issue rate is read against comes from this measurement. */
typedef enum gbl_valu_op {
    GBL_VALU_FMA_F32 = 0, GBL_VALU_PK_FMA_F32 = 1, GBL_VALU_ADD_F32 = 2, GBL_VALU_MAX3_F32 = 3, GBL_VALU_CVT_UBYTE = 4,
    GBL_VALU_PERM_B32 = 5, GBL_VALU_MOV_DPP = 6, GBL_VALU_CNDMASK = 7, GBL_VALU_AND_B32 = 8, GBL_VALU_RCP_F32 = 9,
    GBL_VALU_MED3_F32 = 10, GBL_VALU_CMP_F32 = 11, GBL_VALU_MUL_F32 = 12, GBL_VALU_FMAC_F32 = 13, GBL_VALU_MAX_F32 = 14, GBL_VALU_MOV_B32 = 15,
    GBL_VALU_ADD_U32 = 16, GBL_VALU_LSHL_B32 = 17, GBL_VALU_ADD_F32_E64 = 18, GBL_VALU_FMA_F32_2SRC = 19, GBL_VALU_MUL_LO_U32 = 20,
    GBL_VALU_MUL_U32_U24 = 21, GBL_VALU_MUL_HI_U32 = 22, GBL_VALU_XOR_B32 = 23, GBL_VALU_LSHR_B32 = 24, GBL_VALU_CNDMASK_SGPR = 25, GBL_VALU_OP_COUNT = 26
} gbl_valu_op;
gbl_status gbl_selftest_valu_issue(gbl_ctx* ctx, int op, int waves_per_simd, uint32_t iters, double* out);

void gbl_destroy(gbl_ctx* ctx);
/* Message for the last failing call on ctx (or on creation when ctx == NULL). */
const char* gbl_last_error(const gbl_ctx* ctx);
int gbl_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GOBLIN_HIP_H */
