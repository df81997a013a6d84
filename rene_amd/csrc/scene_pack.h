// scene_pack.h -- host side of the upload step: flatten rene's Scene tables into the HBM layout of
// device_scene.h and build the two BVHs.  Replaces SceneBuffers::new (rene/src/main.rs:2910-3336)
// and the driver's acceleration-structure builds (main.rs:2437-2908).
#pragma once
#include <string>
#include <vector>

#include "../../include/rene_hip.h"
#include "device_scene.h"

namespace rene {

struct BuiltAccel {
  std::vector<Node> nodes;
  std::vector<PrimIsect> isect;  // slot order
  std::vector<SmallItem> items;  // small-scene item list (empty if the structure is too large): n_loop items the
                                 // wave-coherent loop visits, then one auxiliary record per box item
  uint32_t n_loop = 0;
  uint32_t n_top = 0;            // nodes [0, n_top): the tree's top levels in breadth-first order (whole levels, <= TOP_NODES_MAX)
  uint32_t depth = 0;            // max stack depth a traversal can need
};

struct PackedScene {
  BuiltAccel main, emit;
  std::vector<PrimShade> shade;   // main slot order
  std::vector<EmitPdf> emit_pdf;  // emit slot order
  std::vector<Sphere> spheres;
  std::vector<Inst> insts;
  std::vector<EmitObject> emit_objects;
  std::vector<EmitTri> emit_tris;
  std::vector<Material> materials;
  std::vector<Texture> textures;
  std::vector<Light> lights;
  std::vector<Medium> mediums;          // volpath only
  std::vector<InstMedium> inst_medium;  // volpath only
  std::vector<ImageRef> images;
  std::vector<float> image_pool;
  rene_uniform uniform{};
  uint32_t width = 0, height = 0;
  uint32_t features = 0;
  uint32_t n_triangles = 0;  // world-space triangles after flattening
  std::vector<float> small_image;              // FEAT_SMALL: the LDS image (device_scene.h)
  uint32_t small_off[SMALL_OFF_COUNT] = {};
};

// returns a rene_status; fills err on failure
int pack_scene(const rene_scene_desc* d, PackedScene& out, std::string& err);

}  // namespace rene
