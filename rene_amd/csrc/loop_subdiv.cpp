// loop_subdiv.cpp -- Shape "loopsubdiv" (host only): uniform Loop subdivision of a triangle mesh.
//
// Reference: rene/src/scene/subdivision.rs:25-76 hands the mesh to OpenSubdiv (far::TopologyRefiner,
// Scheme::Loop, default options, refine_uniform(level), PrimvarRefiner::interpolate per level on the
// positions only), takes the last level's face-vertex list, zeroes normals and uvs and regenerates the
// normals from the faces (subdivision.rs:7-23).  The OpenSubdiv binding (opensubdiv-petite @ 72b0ea9e)
// is an un-vendored git dependency, absent from the checkout, so this file restates the *published* Loop
// scheme with OpenSubdiv 3.x's conventions as documented in its sources (sdc/loopScheme.h masks,
// vtr/triRefinement.cpp child ordering).  Parity with the reference's meshes is unpinned (no reference test
// or fixture holds a subdivided mesh); tests/test_subdiv.py checks this against an independent numpy
// restatement and closed forms.
//
// Conventions restated:
//   * level topology: edges are numbered in first-seen order walking faces in order, edge k of face
//     (a, b, c) being (a,b), (b,c), (c,a);
//   * child vertices: one per parent vertex first (same index), then one per parent edge (nV + edge);
//   * child faces of (a, b, c) with edge children E0, E1, E2, four consecutive:
//       (a', E0, E2), (E0, b', E1), (E2, E1, c'), (E1, E2, E0);
//   * masks: interior edge 3/8, 3/8 (ends) + 1/8, 1/8 (opposite vertices); boundary edge 1/2, 1/2;
//     smooth vertex of valence n: neighbours w = (5/8 - beta^2) / n with beta = 3/8 + cos(2 pi / n) / 4
//     (1/16 for n = 6), itself 1 - n w; a vertex on the boundary (two boundary edges) 3/4 + 1/8 + 1/8 along
//     the boundary; a vertex where more than two boundary edges meet, or none of a non-manifold fan, stays.
//     Weights are applied smallest first (opposite / neighbour weights, then the vertex itself), in fp32.
//   * an edge with more than two faces is refused (non-manifold input).
#include <cmath>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rene_hip.h"

namespace rene {

namespace {

struct P3 {
  float x, y, z;
};
inline void axpy(P3& d, float w, const P3& s) {
  d.x += w * s.x;
  d.y += w * s.y;
  d.z += w * s.z;
}

struct Level {
  std::vector<P3> pos;
  std::vector<uint32_t> idx;  // 3 per face
};

struct Edge {
  uint32_t v0, v1;
  uint32_t opp[2];  // vertex opposite the edge in each incident face
  uint32_t n_faces;
};

// returns false on a non-manifold edge
bool refine(const Level& in, Level& out) {
  const size_t nv = in.pos.size(), nf = in.idx.size() / 3;
  std::vector<Edge> edges;
  edges.reserve(nf * 3 / 2 + 8);
  std::vector<uint32_t> face_edges(nf * 3);
  std::unordered_map<uint64_t, uint32_t> edge_of;
  edge_of.reserve(nf * 2);
  for (size_t f = 0; f < nf; ++f) {
    const uint32_t* v = &in.idx[3 * f];
    for (int k = 0; k < 3; ++k) {
      uint32_t a = v[k], b = v[(k + 1) % 3], c = v[(k + 2) % 3];
      uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = edge_of.find(key);
      uint32_t e;
      if (it == edge_of.end()) {
        e = (uint32_t)edges.size();
        edge_of.emplace(key, e);
        edges.push_back(Edge{a, b, {c, c}, 1});
      } else {
        e = it->second;
        if (edges[e].n_faces >= 2) return false;
        edges[e].opp[edges[e].n_faces++] = c;
      }
      face_edges[3 * f + k] = e;
    }
  }
  const size_t ne = edges.size();

  // vertex neighbourhoods: interior neighbours in edge order; the (up to two) boundary neighbours apart
  std::vector<uint32_t> valence(nv, 0), n_boundary(nv, 0);
  std::vector<uint32_t> bnd(nv * 2, 0);
  for (const Edge& e : edges) {
    for (int s = 0; s < 2; ++s) {
      uint32_t v = s ? e.v1 : e.v0, o = s ? e.v0 : e.v1;
      ++valence[v];
      if (e.n_faces < 2) {
        if (n_boundary[v] < 2) bnd[2 * v + n_boundary[v]] = o;
        ++n_boundary[v];
      }
    }
  }
  std::vector<uint32_t> first(nv + 1, 0);
  for (size_t v = 0; v < nv; ++v) first[v + 1] = first[v] + valence[v];
  std::vector<uint32_t> nbr(first[nv]), fill(nv, 0);
  for (const Edge& e : edges) {
    nbr[first[e.v0] + fill[e.v0]++] = e.v1;
    nbr[first[e.v1] + fill[e.v1]++] = e.v0;
  }

  out.pos.assign(nv + ne, P3{0, 0, 0});
  for (size_t v = 0; v < nv; ++v) {
    P3 d{0, 0, 0};
    const uint32_t n = valence[v];
    if (n == 0 || n_boundary[v] > 2 || n_boundary[v] == 1) {
      d = in.pos[v];  // isolated, or a corner of boundaries: interpolated
    } else if (n_boundary[v] == 2) {
      axpy(d, 0.125f, in.pos[bnd[2 * v]]);
      axpy(d, 0.125f, in.pos[bnd[2 * v + 1]]);
      axpy(d, 0.75f, in.pos[v]);
    } else {
      float ew = 0.0625f, vw = 0.625f;
      if (n != 6) {
        double inv = 1.0 / (double)n;
        double beta = 0.25 * std::cos(M_PI * 2.0 * inv) + 0.375;
        ew = (float)((0.625 - beta * beta) * inv);
        vw = (float)(1.0 - (double)ew * (double)n);
      }
      for (uint32_t k = 0; k < n; ++k) axpy(d, ew, in.pos[nbr[first[v] + k]]);
      axpy(d, vw, in.pos[v]);
    }
    out.pos[v] = d;
  }
  for (size_t e = 0; e < ne; ++e) {
    const Edge& E = edges[e];
    P3 d{0, 0, 0};
    if (E.n_faces == 2) {
      axpy(d, 0.125f, in.pos[E.opp[0]]);
      axpy(d, 0.125f, in.pos[E.opp[1]]);
      axpy(d, 0.375f, in.pos[E.v0]);
      axpy(d, 0.375f, in.pos[E.v1]);
    } else {
      axpy(d, 0.5f, in.pos[E.v0]);
      axpy(d, 0.5f, in.pos[E.v1]);
    }
    out.pos[nv + e] = d;
  }
  out.idx.resize(nf * 12);
  for (size_t f = 0; f < nf; ++f) {
    const uint32_t a = in.idx[3 * f], b = in.idx[3 * f + 1], c = in.idx[3 * f + 2];
    const uint32_t e0 = (uint32_t)nv + face_edges[3 * f], e1 = (uint32_t)nv + face_edges[3 * f + 1], e2 = (uint32_t)nv + face_edges[3 * f + 2];
    const uint32_t child[12] = {a, e0, e2, e0, b, e1, e2, e1, c, e1, e2, e0};
    for (int k = 0; k < 12; ++k) out.idx[12 * f + k] = child[k];
  }
  return true;
}

}  // namespace

// subdivision.rs:25-76.  Returns an empty string, or the reason the mesh was refused.
std::string loop_subdivide(std::vector<rene_vertex>& verts, std::vector<uint32_t>& idx, unsigned levels) {
  Level cur;
  cur.pos.resize(verts.size());
  for (size_t i = 0; i < verts.size(); ++i) cur.pos[i] = P3{verts[i].position[0], verts[i].position[1], verts[i].position[2]};
  cur.idx = idx;
  for (unsigned l = 0; l < levels; ++l) {
    // 4^levels faces: keep the result inside the 32-bit primitive index space
    if (cur.idx.size() / 3 > (size_t)0x3fffffff / 4) return "loopsubdiv: too many faces after subdivision";
    Level next;
    if (!refine(cur, next)) return "loopsubdiv: an edge is shared by more than two faces";
    cur = std::move(next);
  }
  // generate_normal, subdivision.rs:7-23: area-weighted face normals summed per vertex, then normalised
  // (a vertex no face touches keeps 0 / 0 = NaN exactly as the reference's normalize() would produce)
  std::vector<P3> nrm(cur.pos.size(), P3{0, 0, 0});
  for (size_t f = 0; f + 2 < cur.idx.size(); f += 3) {
    const P3 &a = cur.pos[cur.idx[f]], &b = cur.pos[cur.idx[f + 1]], &c = cur.pos[cur.idx[f + 2]];
    const float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z;
    const float vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
    const P3 p{uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx};
    for (int k = 0; k < 3; ++k) {
      P3& n = nrm[cur.idx[f + k]];
      n.x += p.x;
      n.y += p.y;
      n.z += p.z;
    }
  }
  verts.assign(cur.pos.size(), rene_vertex{});
  for (size_t i = 0; i < cur.pos.size(); ++i) {
    const P3& n = nrm[i];
    const float inv = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    rene_vertex v{};
    v.position[0] = cur.pos[i].x;
    v.position[1] = cur.pos[i].y;
    v.position[2] = cur.pos[i].z;
    v.normal[0] = n.x * inv;
    v.normal[1] = n.y * inv;
    v.normal[2] = n.z * inv;
    verts[i] = v;
  }
  idx = std::move(cur.idx);
  return std::string();
}

}  // namespace rene
