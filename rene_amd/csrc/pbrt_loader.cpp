// pbrt_loader.cpp -- caller side of the render path: pbrt-v3 subset parser + Scene::create.
// (placeholder until the loader lands; the symbols exist so that the ABI is complete)
#include "../../include/rene_hip.h"

extern "C" {
int rene_scene_load_pbrt(const char*, rene_scene** out) { if (out) *out = nullptr; return RENE_ERR_UNSUPPORTED; }
int rene_scene_parse_pbrt(const char*, const char*, rene_scene** out) { if (out) *out = nullptr; return RENE_ERR_UNSUPPORTED; }
const rene_scene_desc* rene_scene_get_desc(const rene_scene*) { return nullptr; }
const char* rene_scene_film_filename(const rene_scene*) { return nullptr; }
void rene_scene_free(rene_scene*) {}
}
