// device_scene.h -- HBM layout of a flattened scene, shared by the host packer and the kernels.
//
// The reference keeps object-space vertex/index buffers plus Vulkan TLAS/BLAS objects
// (rene/src/main.rs:2910-3336).  CDNA4 has no RT hardware, so the layout here is designed for a
// software traversal with 16-byte vector loads:
//   * instancing is flattened: every triangle is stored once per instance in WORLD space
//     (288 GB of HBM makes duplication cheap; one-level traversal has no ray re-transform);
//   * a BVH4 node is one 64-byte record (four 8-bit quantised child boxes + four links) = one cache line;
//   * a primitive's intersection record is 48 bytes (3 x float4), its shading record 64 bytes,
//     both stored in BVH leaf order so a leaf's primitives are contiguous.
#pragma once
#include <stdint.h>

namespace rene {

// ---- BVH4 node with 8-bit child boxes: 4 x float4 = one 64-byte line for four children ----------------------
// (a traversal step costs four divergent 16-byte loads whatever it fetches -- measured: the texture
// addresser is the busiest unit of the BVH kernels -- so a line should carry as many children as it can)
//   q0 = origin.x origin.y origin.z step.z        step_a = 2^e_a, the smallest power of two with 255 steps >= the extent;
//                                                  corner = origin + byte * step
//   q1 = bits(child0) bits(child1) bits(child2) bits(child3)
//   q2 = bits(lo.x of children 0..3, one byte each) bits(lo.y ...) bits(lo.z ...) bits(hi.x ...)
//   q3 = bits(hi.y ...) bits(hi.z ...) step.x step.y
// lo is rounded down and hi up, so the decoded box contains the exact one; an absent child has lo = 255 > hi = 0.
// child word: bit31 = 0 -> inner node index
//             bit31 = 1 -> leaf: bit30 = sphere leaf, bits29..26 = count-1, bits25..0 = first slot
struct Node {
  float q[16];
};
constexpr uint32_t TOP_NODES_MAX = 341;  // 1 + 4 + 16 + 64 + 256: the top levels the builder stores first, breadth-first (21.8 KB)
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t SPHERE_BIT = 0x40000000u;
constexpr uint32_t LEAF_COUNT_SHIFT = 26;
constexpr uint32_t LEAF_FIRST_MASK = (1u << 26) - 1u;
constexpr uint32_t MAX_LEAF_PRIMS = 16;

// ---- primitive intersection record: 3 x float4 (Moeller-Trumbore operands) -----------------------
//   q0 = p0.x p0.y p0.z e1.x
//   q1 = e1.y e1.z e2.x e2.y
//   q2 = e2.z bits(instance) bits(primitive id within its mesh) bits(sphere index or ~0)
// For a sphere slot q0/q1 are unused and q2.w indexes the sphere table.
struct PrimIsect {
  float q[12];
};

// ---- primitive shading record: 4 x float4 ------------------------------------------------------------
//   q0 = n0.xyz uv0.x   q1 = n1.xyz uv0.y   q2 = n2.xyz uv1.x   q3 = uv1.y uv2.x uv2.y 0
// n_i = world_to_object^T * vertex normal (not normalised; linear, so interpolation commutes with
// the transform, rene-shader/src/lib.rs:931-949).  If all three vertex normals are exactly zero the
// transformed geometric normal cross(v1-v0, v2-v0) is stored three times (lib.rs:931-932).
struct PrimShade {
  float q[16];
};

// ---- emitter-pdf record, one per slot of the emitter-only structure -------------------------------
//   triangle: n.xyz (normalised world face normal), area      (lib.rs:1004-1036)
struct EmitPdf {
  float q[4];
};

// ---- sphere table: object_to_world and world_to_object as 3x4 column vectors ----------------------
struct Sphere {
  float o2w[12];
  float w2o[12];
};

// ---- per-instance data (IndexData, lib.rs:108-118, plus what the hit shaders look up) -------------
struct Inst {
  uint32_t material;
  uint32_t area_light;
  float primitive_count;   // as f32, lib.rs:1043
  uint32_t material_type;  // copy of materials[material].type
  float kd[4];             // Matte + Solid texture shortcut: albedo; .w = 1 if the shortcut is valid
  float emit[4];           // area light radiance; .w = 1 if area light is non-null
  // A single-lobe general material (Glass, Substrate, Metal, Mirror) whose textures are all Solid, resolved at upload like
  // the Matte shortcut: compute_bsdf then needs neither the material record nor its texture records (two dependent
  // rounds of loads per bounce).  res_type = RENE_MATERIAL_* or 0 (not resolved: the general path runs), or
  // INST_RES_MATTE_CHECKER: Matte over a checkerboard of two Solid textures (res_ru / res_rv = uscale / vscale, c0 / c1 = the
  // colours of the "same parity" / "other parity" squares).
  uint32_t res_type;
  uint32_t res_remap;      // Substrate / Metal: remap_roughness
  float res_ru, res_rv;    // Substrate / Metal: uroughness.x, vroughness.x as stored (the remap stays on the device)
  float res_c0[4];         // Substrate Kd | Metal eta | Mirror Kr | Glass ir 0 0
  float res_c1[4];         // Substrate Ks | Metal k
};

constexpr uint32_t INST_RES_MATTE_CHECKER = 0x100u;

// ---- emit objects (EnumSurfaceSample, surface_sample.rs:20-33) --------------------------------------
struct EmitObject {
  uint32_t type;        // 0 triangle, 1 sphere
  uint32_t first_tri;   // into emit_tris (world-space, ORIGINAL mesh order: `next_u32 % count`)
  uint32_t prim_count;
  uint32_t pad;
  float matrix[12];     // spheres: object_to_world
};
// world-space emitter triangle for sampling: p0.xyz p1.x | p1.yz p2.xy | p2.z 0 0 0
struct EmitTri {
  float q[12];
};

struct Material {  // rene_material padded to 64 B
  uint32_t type;
  uint32_t u0[4];
  uint32_t u1[4];
  float v0[4];
  uint32_t pad[3];
};
struct Texture {  // rene_texture padded to 48 B
  uint32_t type;
  uint32_t u0[4];
  float v0[4];
  uint32_t pad[3];
};
struct Light {  // rene_light (distant): dir.xyz, 0, L.rgb, 0
  float dir[4];
  float L[4];
};
struct ImageRef {
  uint64_t offset;  // float offset into the image pool
  uint32_t width, height;
};
// ---- participating media (volpath integrator only) ---------------------------------------------------
struct Medium {  // EnumMedium, medium.rs:60-74: sigma_a.rgb g | sigma_s.rgb bits(type)
  float sa_g[4];
  float ss_t[4];
};
struct InstMedium {  // IndexData::{interior,exterior}_medium_index, lib.rs:108-118; kept out of Inst so
  uint32_t interior; // that the path integrator's per-hit record stays 48 bytes
  uint32_t exterior;
};

// ---- small-scene item list (<= SMALL_MAX_ITEMS primitives per structure) ----------------------------
// Scenes this small are intersected by a wave-coherent loop over every item instead of a BVH: all 64
// lanes test the same item, so the record comes through the scalar cache into SGPRs and there is no
// traversal divergence at all (a BVH over 36 triangles ran at ~30 % SIMD efficiency).  Two triangles
// of one instance that form a parallelogram are merged into one item (half the tests).
// An item is the surface X(s,r) = O + s a + r b, stored in the form the loop evaluates fastest -- the
// plane first, then the hit point's coordinates in the reciprocal basis of (a, b):
//   q[0..3]  = n = a x b (not normalised), n . O        ray parameter t = (n.O - n.o) / (n.d)
//   q[4..11] = u'.x v'.x  u'.y v'.y  u'.z v'.z  -O.u' -O.v'   with u' = (b x n)/|n|^2, v' = (n x a)/|n|^2,
//              so that for P = o + t d:  s = P.u' - O.u',  r = P.v' - O.v'   (interleaved for v_pk_fma_f32)
//   q[12]    = kind as a float (SMALL_KIND_*): 1 triangle (s + r <= 1), 0 parallelogram (s, r <= 1), 2 sphere;
//              for the first two it is the coefficient c of the inside test  1 - s - c r >= 0, 1 - r - c s >= 0
//   q[13]    = bits(slot of the triangle covering s + r <= 1)   (sphere: its slot)
//   q[14]    = bits(slot of the triangle covering s + r  > 1)
//   q[15]    = bits(perm1 | perm2 << 8): for each triangle, which of the three generic corner weights is
//              its u (bits 1..0) and its v (bits 3..2); corners are (O, O+a, O+b) with weights
//              (1-s-r, s, r) for the first triangle and (O+a+b, O+a, O+b) with (s+r-1, 1-r, 1-s) for the second
// A degenerate item (n = 0) is stored as all zeros: n.d = 0 rejects every ray.
struct SmallItem {
  float q[16];
};
constexpr uint32_t SMALL_MAX_ITEMS = 64;
constexpr float SMALL_KIND_QUAD = 0.0f, SMALL_KIND_TRIANGLE = 1.0f, SMALL_KIND_SPHERE = 2.0f, SMALL_KIND_BOX = 3.0f,
                SMALL_KIND_BALL = 4.0f;  // a sphere whose transform is translation + uniform scale: q[0..2] = centre, q[3] = radius^2
// A box item (six parallelograms of one instance bounding a parallelepiped O + s a + r b + k c, s, r, k in [0, 1]):
//   q[0..3], q[4..7], q[8..11] = a' -O.a' | b' -O.b' | c' -O.c'   (reciprocal basis: a'.a = 1, a'.b = a'.c = 0, ...):
//   the coordinate of a point p along a is p.a' - O.a', in [0, 1] inside the box
//   q[12] = SMALL_KIND_BOX, q[13] = bits(index of its auxiliary record in the same array), q[14] = bits(open face:
//   0 none, 1 the face at coordinate 0 of the third axis, 2 the one at coordinate 1 -- a five-sided box)
// auxiliary record, two words per face f = 2 * axis + side (side 0: coordinate 0, side 1: coordinate 1):
//   q[2f]   = bits(slot1 | slot2 << 8 | perms << 16)          the two triangles of the face, as in a parallelogram item
//   q[2f+1] = bits(sel_s | flip_s << 2 | sel_r << 3 | flip_r << 5)   the face's own (s, r) from the two in-face box
//             coordinates (ascending axis order): s = flip_s ? 1 - coord[sel_s] : coord[sel_s], likewise r

// ---- small scenes: the tables the shading steps read, resident in LDS --------------------------------------------
// A FEAT_SMALL scene carries an *LDS image* (SceneView::small_image, at most SMALL_LDS_MAX_BYTES): every workgroup of a
// render kernel copies it into its LDS once, when it starts (the launch is persistent), and hit shading, emitter
// sampling and the emitter pdf read their records from there -- a per-lane ds_read_b128 returns in ~64 cycles where the
// same gather through L1 / L2 took 200-500, sixteen dependent round trips per bounce (rocprofv3: 39 % of the wave
// cycles parked on s_waitcnt).  Records are fattened so that one index reaches everything a step needs:
//   byte 0                   main structure's items + auxiliary records (copy of Accel::items), 64 B each:
//                            the (item, s, r) -> (slot, u, v) mapping after the item loop
//   small_off[SMALL_OFF_EMIT_ITEMS]  the emitter structure's items
//   small_off[SMALL_OFF_HIT]   per main slot, 208 B: PrimIsect (48) | PrimShade (64) | the slot's Inst (96)
//   small_off[SMALL_OFF_EMIT]  per emitter slot, 80 B: PrimIsect (48) | EmitPdf (16) | primitive_count 0 0 0
//   small_off[SMALL_OFF_EOBJ]  EmitObject[] (64 B each)
//   small_off[SMALL_OFF_ETRI]  EmitTri[] (48 B each)
//   small_off[SMALL_OFF_SPHERES]  Sphere[] (96 B each): a hit on a sphere, or the pdf of a sphere emitter, reads its matrices
// A scene whose image would not fit is not FEAT_SMALL (it renders through the BVH kernels).
constexpr uint32_t SMALL_LDS_MAX_BYTES = 22u * 1024u;  // six workgroups per CU keep image + seed tables in 160 KB of LDS
enum : uint32_t { SMALL_OFF_EMIT_ITEMS = 0, SMALL_OFF_HIT, SMALL_OFF_EMIT, SMALL_OFF_EOBJ, SMALL_OFF_ETRI, SMALL_OFF_SPHERES, SMALL_OFF_COUNT };
constexpr uint32_t SMALL_HIT_FLOATS = 52, SMALL_EMIT_FLOATS = 20;

// one traversable structure
struct Accel {
  const Node* nodes;
  const PrimIsect* isect;
  const SmallItem* items;  // small-scene path; NULL if the structure is too large for it
  uint32_t n_nodes;
  uint32_t n_slots;
  uint32_t n_items;
  uint32_t n_top;          // nodes [0, n_top) are the tree's top levels, breadth-first (TOP_NODES_MAX)
};

// Uniform, rene-shader/src/lib.rs:90-102: kept in memory rather than in the kernel arguments -- 52
// floats pinned in SGPRs for the whole persistent loop made the compiler spill SGPRs into VGPR lanes
// (v_writelane / v_readlane in the hot loop); they are needed once per path only
struct Uniforms {
  float c2w[16];
  float proj_inv[16];
  float bg_matrix[16];
  float bg_color[4];
  // the background texture resolved at upload (main_miss would otherwise walk texture record -> image record -> texels):
  // bg_kind 0 = go through the texture tables, 1 = a Solid texture (bg_solid), 2 = an ImageMap (bg_image_*)
  uint32_t bg_kind, bg_image_width, bg_image_height, bg_pad;
  unsigned long long bg_image_offset;  // float offset into the image pool
  unsigned long long bg_pad2;
  float bg_solid[4];
};

// everything a kernel needs, passed by value (lives in SGPRs / kernarg)
struct SceneView {
  Accel main;
  Accel emit;
  const PrimShade* shade;     // main structure, slot order
  const EmitPdf* emit_pdf;    // emitter structure, slot order
  const Sphere* spheres;
  const Inst* insts;
  const EmitObject* emit_objects;
  const EmitTri* emit_tris;
  const Material* materials;
  const Texture* textures;
  const Light* lights;
  const ImageRef* images;
  const float* image_pool;
  const struct Uniforms* uni;  // camera / background block (read on demand through the scalar cache)
  const Medium* mediums;          // volpath only; [0] is the vacuum
  const InstMedium* inst_medium;  // volpath only; per instance
  uint32_t bg_texture;
  uint32_t lights_len;
  uint32_t emit_object_len;
  uint32_t width, height;
  uint32_t small_bytes;                  // FEAT_SMALL: size of the LDS image (a multiple of 16), else 0
  const float* small_image;              // FEAT_SMALL: the LDS image in HBM (layout above)
  uint32_t small_off[SMALL_OFF_COUNT];   // byte offsets of its tables
  uint32_t lds_insts;                    // traversal-restart kernels: instances (all of them) + the distant lights kept in LDS behind
                                         // the stack, or 0 (they stay in global memory); set by the launcher
};

// scene feature bits -> kernel specialisation
enum : uint32_t {
  FEAT_SPHERES = 1u << 0,      // any sphere instance
  FEAT_GENERAL_BSDF = 1u << 1, // any material other than None / Matte
  FEAT_TEXTURES = 1u << 2,     // any non-solid texture reachable (checkerboard / scale / imagemap)
  FEAT_LIGHTS = 1u << 3,       // distant lights
  FEAT_BACKGROUND = 1u << 4,   // non-black background
  FEAT_MULTI_LOBE = 1u << 5,   // Plastic / Uber (more than one lobe)
  FEAT_SMALL = 1u << 6,        // both structures fit the wave-coherent item loop (no BVH traversal)
  FEAT_VOLPATH = 1u << 7,      // Integrator "volpath": media, None-material boundaries, depth 80, no roulette
  // what a general-BSDF scene does NOT contain (set only together with FEAT_GENERAL_BSDF, never with FEAT_MULTI_LOBE):
  // a kernel instantiated with these bits leaves the lobe kinds out, which is worth an occupancy step
  FEAT_NO_SPECULAR = 1u << 8,    // no Glass, no Mirror (FresnelSpecular / SpecularReflection / SpecularTransmission lobes)
  FEAT_NO_BLEND = 1u << 9,       // no Substrate (FresnelBlend lobe)
  FEAT_NO_MICROFACET = 1u << 10, // no Metal (MicrofacetReflection lobe)
  FEAT_ABSENT_MASK = FEAT_NO_SPECULAR | FEAT_NO_BLEND | FEAT_NO_MICROFACET,
  FEAT_NO_EMITTERS = 1u << 11,   // no emit objects (area lights): the emitter mixture of lib.rs:274-324 never runs (the traversal-restart
                                 // kernels of the two large bench scenes are instantiated without it)
};

// A pixel record's version: epoch << VERSION_LEVEL_BITS | work items of the pixel committed in that launch.  Ten bits of
// levels let ONE launch render a whole job in short items (8192 frames in 32-frame items = 256 levels); 22 bits of
// epoch are four million launches between two clears of the version words (rene_hip.cpp).
constexpr uint32_t VERSION_LEVEL_BITS = 10;
constexpr uint32_t MAX_LEVELS = (1u << VERSION_LEVEL_BITS) - 1u;
constexpr uint32_t MAX_EPOCH = (1u << (32u - VERSION_LEVEL_BITS)) - 1u;
constexpr uint32_t MAX_LAUNCH_FRAMES = 65536u;  // frames of one launch (its seed tables: SEED_TAB_*); rene_render cuts longer requests

// Frame chains (round 4).  The reference adds every frame of a pixel onto the last (rene/src/main.rs:1315-1397: one dispatch per frame), which on a
// persistent kernel makes a pixel's frames ONE sequential chain: a job cannot end before its most expensive pixel has been through all its frames
// (rene's teapot scene at 8192 spp: the last wave ended 13 % after the median one), and a tile shard of an image has fewer chains than the chip has
// lanes.  Here a pixel's frames are CHAINS = 8 independent chains: frame f belongs to chain (f / frame_stride) mod 8, chain g sums its frames in
// frame order into image g of `RenderParams::framebuffer` ([CHAINS][3][H][W][4]), across launches; the image handed out is
// ((((((c0 + c1) + c2) + c3) + c4) + c5) + c6) + c7 (resolve_chains_kernel), computed when a call hands it out.  A fixed rule on the frame NUMBER: the image
// does not depend on how a job is cut into calls, launches, work items or tile shards (bit-identical, tests), and differs from the strict frame
// order only in the rounding of the regrouped fp32 sums (max 4e-5 of the image's maximum at 1024 - 8192 spp; T1 against the oracle is untouched).
constexpr uint32_t CHAINS_LOG2 = 3;
constexpr uint32_t CHAINS = 1u << CHAINS_LOG2;
// (a lane carries its item's pixel and chain in one register: x | y << 14 | chain << 28)
constexpr uint32_t MAX_RESOLUTION = 16384u;

// The launch's frame seeds in LDS.  RenderParams::seed_tab = word offset of the tables in the workgroup's LDS | shift << 24.
//   shift == 0 (launches of at most SEED_TAB_DIRECT_MAX frames): T[i] = the seed of launch frame i -- one ds_read per path start;
//   shift >= 5: two levels -- the generator state of launch frame i is T1[i >> shift] pushed on by j = i & (2^shift - 1) frame
//   steps, T2[j] = (mul, add) being the affine map of j frame steps (a frame step = frame_stride steps of PCG32si's LCG);
//   T2 (2 * 2^shift words) first, then T1: three ds_reads, a multiply-add and the output permutation per path start,
//   1.3 KB for a launch of 8192 frames.
// SEED_TAB_NONE: no room in LDS, every path start composes the jump itself (frame_seed, ~150 integer instructions).
constexpr uint32_t SEED_TAB_NONE = 0xffffffffu;
constexpr uint32_t SEED_TAB_DIRECT_MAX = 1024u;
inline uint32_t seed_tab_shift(uint32_t n_frames) {
  if (n_frames <= SEED_TAB_DIRECT_MAX) return 0;
  uint32_t shift = 5;
  while (((n_frames + (1u << shift) - 1u) >> shift) > 256u) ++shift;
  return shift;
}
inline uint32_t seed_tab_words(uint32_t n_frames) {
  const uint32_t shift = seed_tab_shift(n_frames);
  if (shift == 0) return n_frames;
  return 2u * (1u << shift) + ((n_frames + (1u << shift) - 1u) >> shift);
}

struct RenderParams {
  float* framebuffer;      // [CHAINS][3][H][W][4]: per chain r, g, b sums + the record's version (device_code.inc, fb_store)
  uint32_t seed_state0;    // state of PCG32si::new(master seed): the seed of global frame g is the stream's g-th output
  uint32_t first_frame;    // global number of the launch's frame 0 ...
  uint32_t frame_stride;   // ... and of the step to its next one (RENE_SHARD_FRAMES deals frames round-robin; else 1)
  uint32_t stack_entries;  // traversal-restart kernels: the LDS stack's depth (what follows it in LDS starts at stack_entries * BLOCK words)
  uint32_t* work_counter;  // next work id
  unsigned long long* counters;  // 9 x u64
  uint32_t n_frames;
  uint32_t n_work;         // work ids of one level: owned tiles * 1024 * CHAINS (id within the level = pixel slot * CHAINS + chain)
  uint32_t shard_rank, shard_count;  // tile sharding (shard_count == 1: all tiles)
  uint32_t tiles_x, n_tiles;
  uint32_t flags;
  uint32_t n_levels;       // every pixel's frames are cut into n_levels work items (ordered hand-off): n_uniform items of level_step
                           // frames, then the rest in halving items (device_code.inc, item_frames)
  uint32_t epoch;          // launch number (1 .. MAX_EPOCH); a pixel record's version is epoch << VERSION_LEVEL_BITS | items committed
  uint32_t* item_done;     // [CHAINS][H][W] the same versions for the traversal-restart kernels, whose records are 12 bytes
  uint32_t ready_min;      // traversal-restart kernel: lanes waiting before the logic step runs
  uint32_t leaf_min;       // traversal-restart kernel: lanes at a leaf before the leaf step runs
  uint32_t level_step;     // frames per uniform work item: item (level < n_uniform, pixel) renders frames [level * step, (level + 1) * step)
  uint32_t n_uniform;      // uniform levels; the remaining n_levels - n_uniform levels halve what is left (n_uniform == n_levels: none)
  uint32_t seed_tab;       // the launch's seed tables in LDS (SEED_TAB_*), set by the launcher
  uint32_t static_waves;   // waves whose first batch is assigned statically (<= co-resident waves)
  uint32_t work_batch;     // work ids a wave takes per global atomic: 128 when items are plentiful, fewer
                           // (down to 16) when a launch has too few items to give every wave a full batch
  uint32_t prev_final;     // version the context's previous launch left on every pixel record (0: a zeroed image):
                           // what a pixel's first item continues from (launches are serial: it is there)
  float inv_n_work, inv_tiles_x;  // 1.0f / n_work, 1.0f / tiles_x (udiv_small in the work-item bookkeeping)
  uint32_t chain_phase;    // frame chains: the launch's frame i (global frame first_frame + i * frame_stride) belongs to chain (chain_phase + i) % CHAINS,
                           // chain g into image g of `framebuffer` ([CHAINS][3][H][W][4]) and block g of `item_done` ([CHAINS][H][W])
  uint32_t group_frames;   // the most frames of the launch any chain has, ceil(n_frames / CHAINS): what item_frames cuts (an item's range is clipped
                           // to its own chain's count, which may be one less)
  unsigned long long* wave_times;  // RENE_DEBUG: [waves][2] start / end of every wave on the 100 MHz clock, else null
  // rene_ray_dump (probe; the counting instantiation of the traversal-restart kernel only): every query the launch issues, 8 floats each --
  // o.xyz, tmax, d.xyz, bits(pixel | depth << 21 | any << 27 | emitter structure << 28 | (launch frame & 7) << 29) -- from float 8 on; word 0 = rays dumped so far (an atomic
  // counter: rays beyond ray_dump_cap are counted, not stored); else null
  float* ray_dump;
  uint32_t ray_dump_cap;
};


}  // namespace rene
