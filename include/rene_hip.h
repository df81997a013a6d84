/*
 * rene_hip.h -- C ABI of the MI355X-native render path for hatoo/rene.
 *
 * The reference (Rust + Vulkan-RT) has no FFI seam of its own: the render path is inlined in
 * `main()` (rene/src/main.rs:209-1687).  The natural seam, and the one this library implements, is
 *
 *     flat `Scene` tables in  (rene/src/scene.rs:36-49, consumed by
 *                              SceneBuffers::new, rene/src/main.rs:2910-3336)
 *     3 accumulation layers out (RGBA32F array image, rene/src/main.rs:383-408,
 *                              read back at rene/src/main.rs:1453-1623)
 *
 * Every struct below is a plain-old-data restatement of one reference table; the comment on each
 * names the reference type it replaces.  No C++ / torch / HIP types appear in any signature, so a
 * Rust host binds this with `extern "C"` + `#[repr(C)]` (see INTEGRATION.md for the stub).
 *
 * Conventions: all matrices are column-major f32 (glam `Mat4::to_cols_array`, `Affine3A` as
 * x_axis,y_axis,z_axis,translation); all indices are u32; every function returns 0 on success and
 * a negative `rene_status` otherwise, with a thread-local message behind `rene_last_error()`.
 * Nothing in this library aborts or throws across the boundary (the reference `unwrap()`s every
 * Vulkan call, rene/src/main.rs passim).
 */
#ifndef RENE_HIP_H
#define RENE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RENE_ABI_VERSION 5u

typedef enum rene_status {
  RENE_OK = 0,
  RENE_ERR_INVALID_ARGUMENT = -1,
  RENE_ERR_INVALID_SCENE = -2,
  RENE_ERR_DEVICE = -3,          /* a HIP call failed; message carries hipGetErrorString */
  RENE_ERR_UNSUPPORTED = -4,     /* e.g. GIF / TIFF / WebP textures, tiled or DWA-compressed EXR files */
  RENE_ERR_OUT_OF_MEMORY = -5,
  RENE_ERR_IO = -6,
  RENE_ERR_PARSE = -7
} rene_status;

/* ---- enumerations; numeric values equal the reference's #[repr(u32)] discriminants ---------- */

/* ShaderOffset, rene/src/main.rs:41-45 */
enum { RENE_SHAPE_TRIANGLE = 0, RENE_SHAPE_SPHERE = 1 };
/* MaterialType, rene-shader/src/material.rs:52-63 */
enum {
  RENE_MATERIAL_NONE = 0, RENE_MATERIAL_MATTE = 1, RENE_MATERIAL_GLASS = 2,
  RENE_MATERIAL_SUBSTRATE = 3, RENE_MATERIAL_METAL = 4, RENE_MATERIAL_MIRROR = 5,
  RENE_MATERIAL_UBER = 6, RENE_MATERIAL_PLASTIC = 7
};
/* TextureType, rene-shader/src/texture.rs:22-29 */
enum { RENE_TEXTURE_SOLID = 0, RENE_TEXTURE_CHECKERBOARD = 1, RENE_TEXTURE_IMAGEMAP = 2,
       RENE_TEXTURE_SCALE = 3 };
/* AreaLightType, rene-shader/src/area_light.rs:9-13 */
enum { RENE_AREA_LIGHT_NULL = 0, RENE_AREA_LIGHT_DIFFUSE = 1 };
/* LightType, rene-shader/src/light.rs:19-21 */
enum { RENE_LIGHT_DISTANT = 0 };
/* Integrator, rene/src/scene/intermediate_scene.rs:186-190 */
enum { RENE_INTEGRATOR_PATH = 0, RENE_INTEGRATOR_VOLPATH = 1 };
/* accumulation layers, rene-shader/src/lib.rs:165-172, 210, 226, 230-231 */
enum { RENE_LAYER_RADIANCE = 0, RENE_LAYER_NORMAL = 1, RENE_LAYER_ALBEDO = 2, RENE_LAYER_COUNT = 3 };

/* ---- scene tables ---------------------------------------------------------------------------- */

/* Vertex, rene-shader/src/lib.rs:883-890 (position, normal, uv); packed to 32 B here (the
 * reference's Vec3A padding carries no information). */
typedef struct rene_vertex {
  float position[3];
  float normal[3];
  float uv[2];
} rene_vertex;

/* TriangleMesh, rene/src/scene/intermediate_scene.rs:149-153.  Indices are mesh-local; the
 * library rebases them the way SceneBuffers::new does (rene/src/main.rs:2943-2963). */
typedef struct rene_mesh {
  const rene_vertex* vertices;
  const uint32_t* indices;
  uint32_t n_vertices;
  uint32_t n_indices; /* multiple of 3 */
} rene_mesh;

/* TlasInstance, rene/src/scene.rs:25-34.  `matrix` is the Affine3A object-to-world transform
 * (for spheres already multiplied by scale(radius), rene/src/scene.rs:418-423). */
typedef struct rene_instance {
  uint32_t shape;           /* RENE_SHAPE_* */
  int32_t mesh_index;       /* blas_index; -1 for spheres */
  uint32_t material_index;
  uint32_t area_light_index;
  uint32_t interior_medium_index;
  uint32_t exterior_medium_index;
  float matrix[12];         /* x_axis, y_axis, z_axis, translation */
} rene_instance;

/* EnumMaterial, rene-shader/src/material.rs:43-70.  Field use per type follows the reference's
 * `new_data` constructors (material.rs:107-115, 138-152, 228-242, 319-326, 353-360, 495-517,
 * 642-657). */
typedef struct rene_material {
  uint32_t type;
  uint32_t u0[4];
  uint32_t u1[4];
  float v0[4];
} rene_material;

/* EnumTexture, rene-shader/src/texture.rs:14-36; field use per texture.rs:62-96. */
typedef struct rene_texture {
  uint32_t type;
  uint32_t u0[4];
  float v0[4];
} rene_texture;

/* EnumAreaLight, rene-shader/src/area_light.rs:21-33 */
typedef struct rene_area_light {
  uint32_t type;
  float v0[4]; /* L.rgb, 0 */
} rene_area_light;

/* EnumLight, rene-shader/src/light.rs:8-28.  Distant: v0 = normalize(from - to), v1 = L
 * (light.rs:43-50). */
typedef struct rene_light {
  uint32_t type;
  float v0[4];
  float v1[4];
} rene_light;

/* EnumMedium, rene-shader/src/medium.rs:47-72.  Homogeneous: v0 = sigma_a.rgb, g; v1 = sigma_s.rgb, 0
 * (medium.rs:78-84).  Only the volumetric integrator reads media. */
enum { RENE_MEDIUM_VACUUM = 0, RENE_MEDIUM_HOMOGENEOUS = 1 };
typedef struct rene_medium {
  uint32_t type;
  float v0[4];
  float v1[4];
} rene_medium;

/* Image, rene/src/scene/image.rs:1-19: linear RGBA32F, row 0 first. */
typedef struct rene_image {
  const float* rgba;
  uint32_t width;
  uint32_t height;
} rene_image;

/* Uniform, rene-shader/src/lib.rs:90-102.  `projection_inv` is PerspectiveCamera.projection, i.e.
 * Mat4::perspective_lh(fov, aspect, 0.01, 1000).inverse() (rene/src/scene.rs:163-164);
 * lights_len / emit_object_len / emit_primitives are derived by the library
 * (rene/src/scene.rs:166, rene/src/main.rs:3278-3280). */
typedef struct rene_uniform {
  float camera_to_world[16];
  float background_matrix[16];
  float background_color[4];
  float projection_inv[16];
  uint32_t background_texture;
} rene_uniform;

/* Scene, rene/src/scene.rs:36-49 (+ Film, intermediate_scene.rs:155-160).  All pointers are
 * borrowed for the duration of the call they are passed to. */
typedef struct rene_scene_desc {
  uint32_t struct_size;   /* sizeof(rene_scene_desc), ABI guard */
  uint32_t integrator;    /* RENE_INTEGRATOR_* */
  uint32_t xresolution;
  uint32_t yresolution;
  rene_uniform uniform;
  uint32_t n_instances;
  uint32_t n_meshes;
  uint32_t n_materials;
  uint32_t n_textures;
  uint32_t n_area_lights;
  uint32_t n_lights;
  uint32_t n_images;
  const rene_instance* instances;
  const rene_mesh* meshes;
  const rene_material* materials;   /* [0] is the None sentinel, rene/src/scene.rs:109 */
  const rene_texture* textures;     /* [0] is solid white, rene/src/scene.rs:113-116 */
  const rene_area_light* area_lights; /* [0] is the Null sentinel, rene/src/scene.rs:110 */
  const rene_light* lights;
  const rene_image* images;
  const rene_medium* mediums;       /* [0] is the Vacuum sentinel, rene/src/scene.rs:111; may be NULL when n_mediums == 0 */
  uint32_t n_mediums;
  uint32_t reserved;
} rene_scene_desc;

/* ---- render options (additive; the reference hard-codes these, rene/src/main.rs:77-81, 1301) - */

#define RENE_DEFAULT_SEED 0x52454E45u /* "RENE" */
#define RENE_TILE_SIZE 32u

enum {
  RENE_FLAG_COUNTERS = 1u << 0, /* also count BVH node visits / primitive tests (slower kernel) */
  RENE_FLAG_NO_AOV = 1u << 1,   /* skip layers 1-2 (first-hit normal/albedo) */
  RENE_FLAG_FORCE_BVH = 1u << 2, /* traverse the BVH even when the scene qualifies for the small-scene item loop */
  RENE_FLAG_SINGLE_LEVEL = 1u << 3, /* one work item per pixel per launch (no long/short split; same results, for A/B tests) */
  RENE_FLAG_NO_RESTART = 1u << 4, /* BVH scenes: use the plain while-while kernel instead of the traversal-restart one (A/B tests) */
  RENE_FLAG_DYNAMIC_FIRST = 1u << 5, /* accepted and ignored: every work batch comes from the atomic counter (a statically owned
                                        first batch made a launch depend on all of its waves being resident) */
  RENE_FLAG_WAVEFRONT = 1u << 6, /* BVH scenes: the stage-separated wavefront integrator (wavefront.inc) instead of the traversal-restart megakernel */
  RENE_FLAG_FP16_PAYLOAD = 1u << 8, /* with RENE_FLAG_WAVEFRONT: a path slot keeps its ray direction and throughput as fp16 (BASELINE config 5's
                                       "fp16 ray payload"): 100 instead of 116 bytes per slot and round trip; origin, distances, sums and
                                       random streams stay fp32 / u32.  The image then differs from the fp32 payload's within the
                                       tolerance tests/test_gpu_scenes.py states; ignored by the default integrators, whose ray
                                       payload never leaves the registers */
  RENE_FLAG_OVERLAP = 1u << 7 /* accepted and ignored since ABI v4.  (Until v3 consecutive rene_render launches alternated between
                                 two streams so that one started while the previous drained its longest paths.  One launch now
                                 renders any number of frames in short work items, so a job is ONE rene_render call with no tail
                                 between launches to hide, and no launch ever waits for another: the occasional stall of the
                                 two-stream scheme -- docs/history.md section 4g -- has nothing left to come from.) */
  ,
  RENE_FLAG_FRAME_GROUPS = 1u << 9 /* accepted and ignored since ABI v5: what it asked for is how every context renders.  A pixel's frames are
                                      EIGHT independent chains -- global frame f belongs to chain f % 8 (under RENE_SHARD_FRAMES: (f / shard_count) % 8),
                                      each chain summed in frame order into an image of its own, across rene_render calls -- and the image a call hands
                                      out (rene_download, rene_framebuffer, rene_reduce, rene_gather_tiles, the caller's opts.framebuffer after rene_sync)
                                      is ((c0 + c1) + c2) + ... + c7.  The reference adds every frame onto the last (rene/src/main.rs:1315-1397), which
                                      on a persistent kernel makes a pixel's frames one sequential chain: a job could not end before its most expensive
                                      pixel had been through all its frames (rene's teapot scene at 8192 spp ended 13 % after its median wave) and a
                                      tile shard had fewer chains than the chip has lanes.  The rule is on the frame NUMBER, so the image is bit-identical
                                      however a job is cut into calls, launches, work items and tile shards, as before; it differs from the strict
                                      frame order only in the rounding of the regrouped fp32 sums (max 4e-5 of the image's maximum at 1024 - 8192 spp). */
};
enum { RENE_SHARD_TILES = 0, RENE_SHARD_FRAMES = 1 };

typedef struct rene_opts {
  uint32_t struct_size;  /* sizeof(rene_opts) */
  uint32_t seed;         /* master seed; frame k uses the k-th next_u32() of PCG32si::new(seed) */
  int32_t device;        /* HIP device ordinal */
  uint32_t flags;        /* RENE_FLAG_* */
  uint32_t shard_mode;   /* RENE_SHARD_* */
  uint32_t shard_rank;   /* this context renders tiles (or frames) with index % shard_count == shard_rank */
  uint32_t shard_count;  /* 0 or 1 = unsharded */
  uint32_t reserved;
  void* framebuffer;     /* optional caller-owned DEVICE buffer of 3*yres*xres*4 floats, else NULL: where the image is handed out.  The
                            library writes it whenever launches are waited for (rene_sync, rene_download, rene_get_stats,
                            rene_framebuffer, rene_reduce, rene_gather_tiles): r, g, b = the sums of the frames rendered so far
                            (the eight frame chains added, see RENE_FLAG_FRAME_GROUPS), the fourth float 0; zero it through rene_reset */
  void* stream;          /* optional hipStream_t to launch on, else NULL (library-owned stream) */
} rene_opts;

/* Device counters; a "ray" is one traversal query (SURVEY section 8 d). */
typedef struct rene_stats {
  uint64_t rays_closest;  /* rene-shader/src/lib.rs:195-207 */
  uint64_t rays_shadow;   /* lib.rs:245-258 */
  uint64_t rays_emitter;  /* lib.rs:301-314 */
  uint64_t paths;         /* raygen invocations */
  uint64_t bounces;       /* loop iterations that shaded a hit (path-state round trips) */
  uint64_t hits;          /* closest hits shaded */
  uint64_t adds;          /* add_image calls, lib.rs:165-172 */
  uint64_t node_visits;   /* only with RENE_FLAG_COUNTERS */
  uint64_t prim_tests;    /* only with RENE_FLAG_COUNTERS */
  uint64_t frames;        /* frames rendered so far (per context) */
  uint64_t launches;      /* kernel launches so far (a launch that had to be replayed after dropped work items counts again) */
  double kernel_ms;       /* sum of HIP-event durations of those launches */
  double last_launch_ms;
  double sclk_mhz;        /* engine clock while those launches ran, measured by the kernels (shader-clock ticks per tick of the
                             constant 100 MHz clock over the lifetime of one wave per launch); 0 before the first launch (ABI v4) */
} rene_stats;

/* One closest-hit record (what Vulkan traversal hands the hit shaders: t, instance, primitive,
 * barycentrics; rene-shader/src/lib.rs:892-905). */
typedef struct rene_hit {
  float t;               /* < 0: miss */
  float u, v;
  uint32_t instance;
  uint32_t primitive;
} rene_hit;

/* What rene_create would build for a scene; filled by rene_scene_pack_info without touching the
 * GPU (host-side validation + flattening + BVH build only). */
typedef struct rene_pack_info {
  uint32_t n_instances;
  uint32_t n_triangles;      /* world-space triangles after instancing is flattened */
  uint32_t n_spheres;
  uint32_t n_nodes_main, n_slots_main, depth_main;
  uint32_t n_nodes_emit, n_slots_emit, depth_emit;
  uint32_t features;         /* kernel specialisation bits: 1 spheres, 2 general BSDFs, 4 textures, 8 distant lights, 16 background,
                                32 multi-lobe materials, 64 wave-coherent item loop (no BVH), 128 volpath; for single-lobe general
                                scenes also what is absent: 256 no Glass / Mirror, 512 no Substrate, 1024 no Metal; 2048 no emit objects */
  uint32_t emit_object_len;  /* rene/src/main.rs:3279 */
  uint32_t lights_len;       /* rene/src/scene.rs:166 */
  uint64_t device_bytes;     /* HBM the scene tables will occupy (framebuffer excluded) */
  uint32_t n_items_main;     /* items the small-scene loop visits per ray (0: the scene uses the BVH), main / emitter-only */
  uint32_t n_items_emit;
} rene_pack_info;

typedef struct rene_ctx rene_ctx;

/* ---- render path ----------------------------------------------------------------------------- */

/* Replaces SceneBuffers::new + pipeline/SBT/descriptor setup (rene/src/main.rs:513-1199): flatten,
 * build the BVHs, upload, clear the accumulation image (main.rs:1229-1237). */
int rene_create(const rene_scene_desc* scene, const rene_opts* opts, rene_ctx** out);

/* Replaces the trace loop (rene/src/main.rs:1315-1397): render frames
 * [first_frame, first_frame + n_frames) and add them into the accumulation layers.  Asynchronous
 * on the context's stream; ordered with later calls on the same context.  ONE persistent launch per call (requests beyond
 * 65 536 frames are cut): a whole job is best rendered by one call -- every call ends on the longest paths of its last work
 * items -- and the image does not depend on how a job is cut into calls (bit-identical). */
int rene_render(rene_ctx* ctx, uint32_t first_frame, uint32_t n_frames);

/* Waits for everything queued on the context (queue_wait_idle, main.rs:1389). */
int rene_sync(rene_ctx* ctx);

/* Replaces layer readback + f32_4_to_3 (main.rs:1453-1619): copies layer `layer` as tightly packed
 * RGB (channels == 3) or RGBA (channels == 4) f32 rows, top row first, un-averaged sums. */
int rene_download(rene_ctx* ctx, int layer, int channels, float* dst, size_t dst_floats);

/* Zero the accumulation layers and the counters (main.rs:1229-1237). */
int rene_reset(rene_ctx* ctx);

/* Optional, before rendering: picks how long the work items of a launch of `n_frames` frames are (no reference counterpart;
 * rene dispatches one frame at a time, main.rs:1355-1372).  Few, long items cost the least bookkeeping; short ones balance
 * scenes whose pixels differ widely in cost and end the launch on a short tail.  Renders three launches of `n_frames` frames
 * per candidate (one item per pixel and launch, then items of 256, 128, ... 16 frames), keeps the fastest, then resets the
 * context like rene_reset.  Untuned contexts cut a pixel's frames into sixteen items per launch (of at least 64 frames) with the
 * small-scene kernels and 32 (of at least 16, the last ones halving) with the BVH kernels, over its eight frame chains.  The choice
 * changes no bit of any image -- a chain's frames are added in the same order however they are cut. */
int rene_tune(rene_ctx* ctx, uint32_t n_frames);

/* Device address of the accumulation image [3][yres][xres][4] f32 (for callers that run their own exchange, e.g.
 * torch.distributed; rene_reduce / rene_gather_tiles below do it inside the library).  Waits for the launches issued so far: the
 * image is the frame chains added together, which happens then. */
int rene_framebuffer(rene_ctx* ctx, void** device_ptr, size_t* n_floats);

int rene_get_stats(rene_ctx* ctx, rene_stats* out);

/* Batch closest-hit queries against the main (which == 0) or emitter-only (which == 1) structure;
 * host pointers; 0 <= tmin <= tmax.  Exposes the traversal the Vulkan driver hides. */
int rene_trace(rene_ctx* ctx, int which, size_t n, const float* origins, const float* directions,
               float tmin, float tmax, rene_hit* out);

/* The J1 gate (probes, no reference counterpart): what does traversal alone cost when its rays come from a queue?
 * rene_ray_dump renders frames [first_frame, first_frame + n_frames) on a RENE_FLAG_COUNTERS context of a deep-BVH path-integrator scene and
 * records every traversal query the launch issues -- 8 floats per ray: o.xyz, tmax, d.xyz, bits(pixel | depth << 21 | any-hit << 27 | emitter
 * structure << 28 | (frame & 7) << 29) -- up to `capacity` rays into host memory; *n_issued = the queries issued (may exceed capacity).  The frames are accumulated
 * like any rendered frames.  rene_trace_queue runs a traversal-only persistent pass over n rays in queue order (origin + tmax as 4 floats; the
 * direction as three halves + flags in the fourth half (fp16 != 0, 8 bytes per ray) or three floats + flags (16 bytes); flag bit 0 any-hit,
 * bit 1 emitter-only structure): free lanes are refilled from the queue as soon as `refill_min` of a wave's 64 are free; `repeats` timed
 * launches, *ms = the fastest; hits4 (optional): t (-1 = miss), u, v, bits(slot) per ray; steps5 (optional): wave-steps and lane-steps of the
 * node and leaf steps + iterations. */
int rene_ray_dump(rene_ctx* ctx, uint32_t first_frame, uint32_t n_frames, size_t capacity, float* rays8, uint64_t* n_issued);
int rene_trace_queue(rene_ctx* ctx, size_t n, const float* o_tmax4, const void* d_flags, int fp16, uint32_t refill_min, uint32_t leaf_min,
                     uint32_t blocks_per_cu, uint32_t repeats, float* hits4, float* ms, uint64_t* steps5);

/* Per-function probe of the device BSDF code (EnumMaterial::compute_bsdf + Bsdf::{f, pdf, sample_f},
 * rene-shader/src/material.rs:739-769, reflection.rs:286-342): for each of the n items builds the
 * lobes of `material_index` at (normal, uv) and writes 12 floats: f(wo,wi).rgb, pdf(wo,wi),
 * sample.wi.xyz, sample.f.rgb, sample.pdf (one sample_f(wo) from PCG32si::new(seed)), lobe count.
 * Host pointers; world-space directions. */
int rene_bsdf_eval(rene_ctx* ctx, uint32_t material_index, size_t n, const float* normals3,
                   const float* uvs2, const float* wo3, const float* wi3, const uint32_t* seeds,
                   float* out12);

/* Per-function probe of the device medium code (EnumMedium::{tr, phase, sample, sample_p},
 * rene-shader/src/medium.rs:104-158) for n items with ray origin 0.  Writes 16 floats per item:
 * tr(rd, t_max).rgb, phase(wo, wi), sample.sampled (0/1), sample.position.xyz, sample.tr.rgb,
 * sample_p(wo).xyz (drawn after `sample` from the same PCG32si::new(seed)), the bits of the stream's
 * next u32, 0.  Host pointers.  RENE_ERR_INVALID_ARGUMENT unless the scene's integrator is volpath. */
int rene_medium_eval(rene_ctx* ctx, uint32_t medium_index, size_t n, const float* rd3,
                     const float* t_max, const float* wo3, const float* wi3, const uint32_t* seeds,
                     float* out16);

/* Per-function probe of the emitter-pdf query (rene-shader/src/lib.rs:301-318: trace `direction` from `origin`
 * against the emitter-only structure, tmin 0.001, tmax 1e5, then main_miss_pdf / triangle_closest_hit_pdf /
 * sphere_closest_hit_pdf, lib.rs:959-1066): out[i] = pdf_l of ray i (0 on a miss).  Host pointers. */
int rene_emitter_pdf(rene_ctx* ctx, size_t n, const float* origins, const float* directions, float* out);

/* Probe of the device random stream (PCG32si, rene-shader/src/rand.rs:4-52): out[k] = the k-th next_u32() of
 * PCG32si::new(seed) computed by one lane of `device`.  Integer-exact known answers: tests/golden/pcg32si_kat.json. */
int rene_pcg_probe(int device, uint32_t seed, uint32_t n, uint32_t* out);

/* ---- multi-GPU exchange step inside the boundary: RCCL over xGMI ---------------------------------
 * The reference renders on one GPU; this build shards a job over the GPUs of a node (one context per GPU; tiles or
 * frame blocks, rene_opts.shard_*) and needs exactly one exchange at the end of a job -- the sum of the per-GPU
 * accumulation images (SURVEY section 8 e).  These entry points keep it on the device: ncclReduce / ncclSend /
 * ncclRecv on the context's own stream, ordered after its launches.  RCCL (librccl.so) is loaded when the first of
 * them is called; RENE_ERR_UNSUPPORTED if it is absent.
 *   one process per GPU:  rank 0 calls rene_comm_unique_id and hands the 128 bytes to the other ranks by whatever means
 *                         the host has (a file, a socket, torch.distributed's store); every rank then calls rene_comm_init;
 *   one process, n GPUs:  rene_comm_init_all over its n contexts (ncclCommInitAll); calls on different contexts of one
 *                         communicator must then be made from different host threads or inside rene_comm_group_begin /
 *                         rene_comm_group_end (ncclGroupStart / ncclGroupEnd).
 * After an exchange the root's image holds the job's sums; every context of the communicator needs rene_reset before
 * it renders again (the records' version words have been summed or overwritten). */
#define RENE_COMM_ID_BYTES 128
int rene_comm_unique_id(uint8_t id[RENE_COMM_ID_BYTES]);
int rene_comm_init(rene_ctx* ctx, int n_ranks, int rank, const uint8_t id[RENE_COMM_ID_BYTES]);
int rene_comm_init_all(rene_ctx** ctxs, int n);
int rene_comm_group_begin(void);
int rene_comm_group_end(void);
/* Frame-sharded jobs (RENE_SHARD_FRAMES, or any cut in which several ranks add to the same pixel): sum of the
 * [3][yres][xres][4] f32 images onto rank `root`, in place (ncclReduce, ncclSum).  Differs from a one-GPU render only
 * in fp32 summation order. */
int rene_reduce(rene_ctx* ctx, int root);
/* Tile-sharded jobs (RENE_SHARD_TILES: every pixel has exactly one owner): every rank sends ONLY the 32x32 tiles it
 * owns to `root` (1 / n_ranks of the image per rank), which places them; the root's image is then bit-identical to a
 * one-GPU render.  All contexts of the communicator must have been created with shard_count == n_ranks and
 * shard_rank == their rank. */
int rene_gather_tiles(rene_ctx* ctx, int root);

void rene_destroy(rene_ctx* ctx);

/* Validate + flatten + build on the host only (no HIP call): same checks and status codes as
 * rene_create. */
int rene_scene_pack_info(const rene_scene_desc* scene, rene_pack_info* out);

const char* rene_last_error(void);
uint32_t rene_abi_version(void);

/* ---- output transform (rene/src/main.rs:1758-1810); pure host functions ----------------------- */

/* average (main.rs:1758-1764) then to_rgb8 (main.rs:1785-1792) */
void rene_to_rgb8(const float* sums, size_t n_floats, uint32_t n_samples, uint8_t* out);
/* to_aov / to_aov_normal (main.rs:1794-1810) after average */
void rene_to_aov8(const float* sums, size_t n_floats, uint32_t n_samples, int is_normal, uint8_t* out);
/* The build-defined seed schedule (SURVEY section 8 d): out[k] = frame seed of frame first+k. */
void rene_frame_seeds(uint32_t master_seed, uint32_t first_frame, uint32_t n, uint32_t* out);

/* ---- caller side: pbrt-v3 loader (pbrt-parser/src/lib.rs, rene/src/scene.rs) ------------------ */

typedef struct rene_scene rene_scene;

/* expand_include + parse_pbrt + Scene::create (rene/src/main.rs:107-205). */
int rene_scene_load_pbrt(const char* path, rene_scene** out);
/* same, from memory; base_dir resolves Include / plymesh / imagemap paths */
int rene_scene_parse_pbrt(const char* text, const char* base_dir, rene_scene** out);
const rene_scene_desc* rene_scene_get_desc(const rene_scene* scene);
/* Film filename (intermediate_scene.rs:155-170) */
const char* rene_scene_film_filename(const rene_scene* scene);
void rene_scene_free(rene_scene* scene);

#ifdef __cplusplus
}
#endif
#endif /* RENE_HIP_H */
