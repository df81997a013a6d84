// scene_pack.cpp -- see scene_pack.h.  Pure host C++ (no HIP calls).
#include "scene_pack.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rene {
namespace {

struct V3 {
  float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::numeric_limits<float>::infinity();
      hi[a] = -std::numeric_limits<float>::infinity();
    }
  }
  void grow(V3 p) {
    lo[0] = std::min(lo[0], p.x); hi[0] = std::max(hi[0], p.x);
    lo[1] = std::min(lo[1], p.y); hi[1] = std::max(hi[1], p.y);
    lo[2] = std::min(lo[2], p.z); hi[2] = std::max(hi[2], p.z);
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
  }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

// 3x4 affine as column vectors (glam Affine3A)
struct Aff {
  V3 x, y, z, w;
};
inline Aff aff_from(const float* m) {
  return {{m[0], m[1], m[2]}, {m[3], m[4], m[5]}, {m[6], m[7], m[8]}, {m[9], m[10], m[11]}};
}
inline V3 xform_point(const Aff& m, V3 p) {
  return {p.x * m.x.x + p.y * m.y.x + p.z * m.z.x + m.w.x, p.x * m.x.y + p.y * m.y.y + p.z * m.z.y + m.w.y,
          p.x * m.x.z + p.y * m.y.z + p.z * m.z.z + m.w.z};
}
bool aff_inverse(const Aff& m, Aff& out) {
  double a[3][3] = {{m.x.x, m.y.x, m.z.x}, {m.x.y, m.y.y, m.z.y}, {m.x.z, m.y.z, m.z.z}};
  double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
  double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
  double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
  double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
  if (det == 0.0 || !std::isfinite(det)) return false;
  double id = 1.0 / det;
  double inv[3][3];
  inv[0][0] = c00 * id;
  inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
  inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
  inv[1][0] = c01 * id;
  inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
  inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
  inv[2][0] = c02 * id;
  inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
  inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
  double t[3] = {m.w.x, m.w.y, m.w.z}, it[3];
  for (int r = 0; r < 3; ++r) it[r] = -(inv[r][0] * t[0] + inv[r][1] * t[1] + inv[r][2] * t[2]);
  out.x = {(float)inv[0][0], (float)inv[1][0], (float)inv[2][0]};
  out.y = {(float)inv[0][1], (float)inv[1][1], (float)inv[2][1]};
  out.z = {(float)inv[0][2], (float)inv[1][2], (float)inv[2][2]};
  out.w = {(float)it[0], (float)it[1], (float)it[2]};
  return true;
}
// normal transform: (w2o.x . n, w2o.y . n, w2o.z . n)  -- rene-shader/src/lib.rs:944-948
inline V3 xform_normal(const Aff& w2o, V3 n) { return {dot(w2o.x, n), dot(w2o.y, n), dot(w2o.z, n)}; }

inline float bits_to_float(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// a primitive before BVH ordering
struct Prim {
  Box box;
  PrimIsect isect;
  PrimShade shade;
  EmitPdf pdf;
  bool sphere;
};

void pad_box(Box& b) {
  // rays are tested against boxes with a few ulp of slack; make flat boxes (axis-aligned quads)
  // robust against the rounding of (plane - origin) * inv_dir
  for (int a = 0; a < 3; ++a) {
    float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
    float pad = 2e-6f * m + 1e-7f;
    b.lo[a] -= pad;
    b.hi[a] += pad;
  }
}

// Binned-SAH BVH2 over `prims`, emitting the 64-byte GPU nodes.  Spheres get leaves of their own.
struct Builder {
  const std::vector<Prim>& prims;
  std::vector<uint32_t> order;
  std::vector<float> cen;
  std::vector<Node> nodes;
  uint32_t max_leaf;
  uint32_t max_depth = 0;

  Builder(const std::vector<Prim>& p, uint32_t ml) : prims(p), max_leaf(ml) {}

  static void set_child(Node& n, int which, const Box& b, uint32_t word) {
    float* q = n.q;
    if (which == 0) {
      q[0] = b.lo[0]; q[1] = b.lo[1]; q[2] = b.lo[2]; q[3] = b.hi[0]; q[4] = b.hi[1]; q[5] = b.hi[2];
      q[12] = bits_to_float(word);
    } else {
      q[6] = b.lo[0]; q[7] = b.lo[1]; q[8] = b.lo[2]; q[9] = b.hi[0]; q[10] = b.hi[1]; q[11] = b.hi[2];
      q[13] = bits_to_float(word);
    }
  }
  static Box empty_box() {
    Box b;  // lo = +inf, hi = -inf never passes a slab test
    b.reset();
    return b;
  }

  bool leaf_ok(uint32_t first, uint32_t count) const {
    if (count == 1) return true;
    if (count > max_leaf) return false;
    for (uint32_t i = first; i < first + count; ++i)
      if (prims[order[i]].sphere) return false;
    return true;
  }
  uint32_t leaf_word(uint32_t first, uint32_t count) const {
    uint32_t w = LEAF_BIT | ((count - 1) << LEAF_COUNT_SHIFT) | first;
    if (prims[order[first]].sphere) w |= SPHERE_BIT;
    return w;
  }

  // splits [first, first+count) and returns mid
  uint32_t split(uint32_t first, uint32_t count) {
    Box cb;
    cb.reset();
    for (uint32_t i = first; i < first + count; ++i) {
      const float* c = &cen[3 * order[i]];
      cb.grow(V3{c[0], c[1], c[2]});
    }
    constexpr int NB = 32;
    int best_axis = -1, best_split = -1;
    float best_cost = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; ++a) {
      float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.0f)) continue;
      Box bb[NB];
      uint32_t bc[NB] = {0};
      for (auto& b : bb) b.reset();
      float scale = (float)NB / ext;
      for (uint32_t i = first; i < first + count; ++i) {
        int bi = std::min(NB - 1, (int)((cen[3 * order[i] + a] - cb.lo[a]) * scale));
        bb[bi].grow(prims[order[i]].box);
        bc[bi]++;
      }
      float ra[NB];
      uint32_t rc[NB];
      Box acc;
      acc.reset();
      uint32_t cnt = 0;
      for (int i = NB - 1; i > 0; --i) {
        if (bc[i]) acc.grow(bb[i]);
        cnt += bc[i];
        ra[i] = cnt ? acc.half_area() : 0.f;
        rc[i] = cnt;
      }
      acc.reset();
      cnt = 0;
      for (int i = 0; i < NB - 1; ++i) {
        if (bc[i]) acc.grow(bb[i]);
        cnt += bc[i];
        if (!cnt || !rc[i + 1]) continue;
        float cost = acc.half_area() * (float)cnt + ra[i + 1] * (float)rc[i + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = i; }
      }
    }
    uint32_t mid = first + count / 2;
    if (best_axis >= 0) {
      float ext = cb.hi[best_axis] - cb.lo[best_axis];
      float scale = (float)NB / ext;
      float lo = cb.lo[best_axis];
      auto* b = order.data() + first;
      auto* m = std::partition(b, b + count, [&](uint32_t p) {
        int bi = std::min(NB - 1, (int)((cen[3 * p + best_axis] - lo) * scale));
        return bi <= best_split;
      });
      uint32_t mm = (uint32_t)(m - order.data());
      if (mm != first && mm != first + count) mid = mm;
    }
    return mid;
  }

  Box range_box(uint32_t first, uint32_t count) const {
    Box b;
    b.reset();
    for (uint32_t i = first; i < first + count; ++i) b.grow(prims[order[i]].box);
    return b;
  }

  void build() {
    uint32_t n = (uint32_t)prims.size();
    order.resize(n);
    cen.resize((size_t)n * 3);
    for (uint32_t i = 0; i < n; ++i) {
      order[i] = i;
      for (int a = 0; a < 3; ++a) cen[3 * i + a] = 0.5f * (prims[i].box.lo[a] + prims[i].box.hi[a]);
    }
    nodes.clear();
    nodes.emplace_back();
    std::memset(&nodes[0], 0, sizeof(Node));
    if (n == 0) {
      set_child(nodes[0], 0, empty_box(), LEAF_BIT);
      set_child(nodes[0], 1, empty_box(), LEAF_BIT);
      max_depth = 1;
      return;
    }
    if (leaf_ok(0, n)) {  // whole scene in one leaf: root = {leaf, nothing}
      set_child(nodes[0], 0, range_box(0, n), leaf_word(0, n));
      set_child(nodes[0], 1, empty_box(), LEAF_BIT);
      max_depth = 1;
      return;
    }
    struct Item { uint32_t node, first, count, depth; };
    std::vector<Item> stack{{0u, 0u, n, 1u}};
    while (!stack.empty()) {
      Item it = stack.back();
      stack.pop_back();
      max_depth = std::max(max_depth, it.depth);
      uint32_t mid = split(it.first, it.count);
      uint32_t cf[2] = {it.first, mid}, cc[2] = {mid - it.first, it.first + it.count - mid};
      for (int c = 0; c < 2; ++c) {
        Box b = range_box(cf[c], cc[c]);
        if (leaf_ok(cf[c], cc[c])) {
          set_child(nodes[it.node], c, b, leaf_word(cf[c], cc[c]));
        } else {
          uint32_t idx = (uint32_t)nodes.size();
          nodes.emplace_back();
          std::memset(&nodes.back(), 0, sizeof(Node));
          set_child(nodes[it.node], c, b, idx);
          stack.push_back({idx, cf[c], cc[c], it.depth + 1});
        }
      }
    }
  }
};

inline uint32_t float_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

// ---- BVH2 -> BVH4 with 8-bit child boxes (device_scene.h, Node) ------------------------------------------
// Collapse: a node adopts the children of its inner children -- shallowest first (both BVH2 children before any
// grandchild, which halves the tree depth and with it the traversal stack), largest surface area among
// equals -- until it has four children or only leaves.  Quantisation: child box corners on a 256-step grid anchored at the node's
// origin with a power-of-two step per axis; lo rounded down, hi rounded up, so a decoded box always
// contains the exact one.  Returns the worst-case traversal stack depth in `stack_need`.
struct Collapse4 {
  const std::vector<Node>& n2;
  std::vector<Node> out;
  struct Child { Box box; uint32_t word; int level = 1; };

  explicit Collapse4(const std::vector<Node>& nodes) : n2(nodes) {}
  static bool empty(const Box& b) { return !(b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]); }
  void children_of(uint32_t idx, std::vector<Child>& v) const {
    const float* q = n2[idx].q;
    Child a{{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}}, float_bits(q[12]), 1};
    Child b{{{q[6], q[7], q[8]}, {q[9], q[10], q[11]}}, float_bits(q[13]), 1};
    if (!empty(a.box)) v.push_back(a);
    if (!empty(b.box)) v.push_back(b);
  }
  static void encode(Node& n, const std::vector<Child>& ch) {
    std::memset(&n, 0, sizeof(n));
    uint32_t words[4] = {LEAF_BIT, LEAF_BIT, LEAF_BIT, LEAF_BIT};
    uint8_t qlo[3][4], qhi[3][4];
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 4; ++c) { qlo[a][c] = 255; qhi[a][c] = 0; }  // lo > hi: never hit
    float scale[3] = {0, 0, 0};  // 2^e per axis, as floats (the traversal multiplies by them; decoding exponent bytes cost six instructions a node)
    float origin[3] = {0, 0, 0};
    if (!ch.empty()) {
      for (int a = 0; a < 3; ++a) {
        double lo = ch[0].box.lo[a], hi = ch[0].box.hi[a];
        for (const Child& c : ch) { lo = std::min<double>(lo, c.box.lo[a]); hi = std::max<double>(hi, c.box.hi[a]); }
        origin[a] = (float)lo;  // a child lo, exactly representable
        double ext = hi - lo;
        int e = -126;
        if (ext > 0.0) {
          int ex;
          std::frexp(ext / 255.0, &ex);  // ext / 255 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= ext / 255
          e = std::max(-126, std::min(127, ex));
        }
        const double step = std::ldexp(1.0, e);
        scale[a] = (float)step;  // 2^e, e in [-126, 127]: a normal float, exact
        for (size_t c = 0; c < ch.size(); ++c) {
          double l = std::floor((ch[c].box.lo[a] - lo) / step), h = std::ceil((ch[c].box.hi[a] - lo) / step);
          qlo[a][c] = (uint8_t)std::max(0.0, std::min(255.0, l));
          qhi[a][c] = (uint8_t)std::max(0.0, std::min(255.0, h));
        }
      }
      for (size_t c = 0; c < ch.size(); ++c) words[c] = ch[c].word;
    }
    n.q[0] = origin[0]; n.q[1] = origin[1]; n.q[2] = origin[2];
    n.q[3] = scale[2];
    n.q[14] = scale[0];
    n.q[15] = scale[1];
    for (int c = 0; c < 4; ++c) n.q[4 + c] = bits_to_float(words[c]);
    for (int a = 0; a < 3; ++a) {
      n.q[8 + a] = bits_to_float(qlo[a][0] | (qlo[a][1] << 8) | (qlo[a][2] << 16) | ((uint32_t)qlo[a][3] << 24));
      n.q[11 + a] = bits_to_float(qhi[a][0] | (qhi[a][1] << 8) | (qhi[a][2] << 16) | ((uint32_t)qhi[a][3] << 24));
    }
  }
  uint32_t run() {  // returns the stack entries a traversal can need
    out.clear();
    out.emplace_back();
    struct Item { uint32_t src, dst; };
    std::vector<Item> work{{0u, 0u}};
    std::vector<std::vector<uint32_t>> kids(1);
    std::vector<uint32_t> nchild(1, 0);
    while (!work.empty()) {
      Item it = work.back();
      work.pop_back();
      std::vector<Child> ch;
      children_of(it.src, ch);
      for (;;) {
        if (ch.size() >= 4) break;
        int pick = -1;
        float best = -1.0f;
        int level = 1 << 30;
        for (size_t c = 0; c < ch.size(); ++c)
          if (!(ch[c].word & LEAF_BIT) &&
              (ch[c].level < level || (ch[c].level == level && ch[c].box.half_area() > best))) {
            level = ch[c].level;
            best = ch[c].box.half_area();
            pick = (int)c;
          }
        if (pick < 0) break;
        std::vector<Child> sub;
        children_of(ch[pick].word, sub);
        for (Child& s : sub) s.level = ch[pick].level + 1;
        if (ch.size() - 1 + sub.size() > 4) break;
        ch.erase(ch.begin() + pick);
        ch.insert(ch.end(), sub.begin(), sub.end());
      }
      for (Child& c : ch)
        if (!(c.word & LEAF_BIT)) {
          uint32_t idx = (uint32_t)out.size();
          out.emplace_back();
          kids.emplace_back();
          nchild.push_back(0);
          work.push_back({c.word, idx});
          kids[it.dst].push_back(idx);
          c.word = idx;
        }
      nchild[it.dst] = (uint32_t)ch.size();
      encode(out[it.dst], ch);
    }
    // children are always created after their parent: one reverse sweep computes the stack bound
    std::vector<uint32_t> need(out.size(), 0);
    for (size_t i = out.size(); i-- > 0;) {
      uint32_t m = 0;
      for (uint32_t k : kids[i]) m = std::max(m, need[k]);
      need[i] = (nchild[i] ? nchild[i] - 1 : 0) + m;
    }
    return need[0] + 1;
  }
};

// Small-scene item list over the slots of a built structure: merge triangle pairs of one instance
// that share an edge X-Y and whose third vertices satisfy Z1 + Z2 = X + Y (a parallelogram).
void build_small_items(BuiltAccel& acc) {
  acc.items.clear();
  const size_t n = acc.isect.size();
  if (n == 0 || n > 2 * SMALL_MAX_ITEMS) return;
  struct Tri { V3 v[3]; uint32_t inst, prim; bool sphere; };
  std::vector<Tri> t(n);
  for (size_t s = 0; s < n; ++s) {
    const float* q = acc.isect[s].q;
    t[s].sphere = float_bits(q[11]) != 0xffffffffu;
    t[s].inst = float_bits(q[9]);
    t[s].prim = float_bits(q[10]);
    t[s].v[0] = V3{q[0], q[1], q[2]};
    t[s].v[1] = V3{q[0] + q[3], q[1] + q[4], q[2] + q[5]};
    t[s].v[2] = V3{q[0] + q[6], q[1] + q[7], q[2] + q[8]};
  }
  auto close = [](V3 a, V3 b, float tol) {
    return std::fabs(a.x - b.x) <= tol && std::fabs(a.y - b.y) <= tol && std::fabs(a.z - b.z) <= tol;
  };
  std::vector<char> used(n, 0);
  // plane + reciprocal-basis form of X(s,r) = O + s a + r b (device_scene.h), evaluated in double
  auto surface = [](SmallItem& it, V3 O, V3 a, V3 b) {
    const double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z, ox = O.x, oy = O.y, oz = O.z;
    const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const double n2 = nx * nx + ny * ny + nz * nz;
    if (!(n2 > 0.0) || !std::isfinite(n2)) return;  // degenerate: stays all zero, never hit
    const double ux = (by * nz - bz * ny) / n2, uy = (bz * nx - bx * nz) / n2, uz = (bx * ny - by * nx) / n2;
    const double vx = (ny * az - nz * ay) / n2, vy = (nz * ax - nx * az) / n2, vz = (nx * ay - ny * ax) / n2;
    it.q[0] = (float)nx; it.q[1] = (float)ny; it.q[2] = (float)nz; it.q[3] = (float)(nx * ox + ny * oy + nz * oz);
    it.q[4] = (float)ux; it.q[5] = (float)vx; it.q[6] = (float)uy; it.q[7] = (float)vy;
    it.q[8] = (float)uz; it.q[9] = (float)vz;
    it.q[10] = (float)-(ux * ox + uy * oy + uz * oz);
    it.q[11] = (float)-(vx * ox + vy * oy + vz * oz);
  };
  struct Geo { V3 O, a, b; uint32_t inst; bool quad; };
  std::vector<Geo> geo;  // the surface of every item, for the parallelepiped merge below
  auto emit_tri = [&](size_t s) {
    SmallItem it;
    std::memset(&it, 0, sizeof(it));
    const float* q = acc.isect[s].q;
    geo.push_back(Geo{V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}, V3{q[6], q[7], q[8]}, t[s].inst, false});
    if (!t[s].sphere) surface(it, V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}, V3{q[6], q[7], q[8]});  // O = p0, a = e1, b = e2
    it.q[12] = t[s].sphere ? SMALL_KIND_SPHERE : SMALL_KIND_TRIANGLE;
    if (t[s].sphere && q[4] == 1.0f) {  // centre, radius^2 (pack_scene)
      it.q[0] = q[0]; it.q[1] = q[1]; it.q[2] = q[2]; it.q[3] = q[3];
      it.q[12] = SMALL_KIND_BALL;
    }
    it.q[13] = bits_to_float((uint32_t)s);
    it.q[14] = bits_to_float((uint32_t)s);
    it.q[15] = bits_to_float((1u | (2u << 2)) * 0x101u);  // u = weight of O+a, v = weight of O+b
    acc.items.push_back(it);
  };
  for (size_t i = 0; i < n; ++i) {
    if (used[i]) continue;
    used[i] = 1;
    if (t[i].sphere) { emit_tri(i); continue; }
    bool merged = false;
    for (size_t j = i + 1; j < n && !merged; ++j) {
      if (used[j] || t[j].sphere || t[j].inst != t[i].inst) continue;
      // scale-aware tolerance
      float scale = 0.f;
      for (int k = 0; k < 3; ++k)
        scale = std::max(scale, std::max(std::fabs(t[i].v[k].x), std::max(std::fabs(t[i].v[k].y), std::fabs(t[i].v[k].z))));
      float tol = 4e-6f * std::max(scale, 1e-3f);
      // find the shared edge: vertices of i matching vertices of j
      int mi[3] = {-1, -1, -1};
      int shared = 0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          if (mi[a] < 0 && close(t[i].v[a], t[j].v[b], tol)) { mi[a] = b; shared++; break; }
      if (shared != 2) continue;
      int zi = mi[0] < 0 ? 0 : (mi[1] < 0 ? 1 : 2);       // third vertex of i
      int xi = (zi + 1) % 3, yi = (zi + 2) % 3;          // shared edge in i's winding
      int zj = 3 - mi[xi] - mi[yi];                        // third vertex of j
      V3 Z1 = t[i].v[zi], X = t[i].v[xi], Y = t[i].v[yi], Z2 = t[j].v[zj];
      V3 lhs{Z1.x + Z2.x, Z1.y + Z2.y, Z1.z + Z2.z}, rhs{X.x + Y.x, X.y + Y.y, X.z + Y.z};
      if (!close(lhs, rhs, tol)) continue;
      SmallItem it;
      std::memset(&it, 0, sizeof(it));
      surface(it, Z1, sub(X, Z1), sub(Y, Z1));
      geo.push_back(Geo{Z1, sub(X, Z1), sub(Y, Z1), t[i].inst, true});
      it.q[12] = SMALL_KIND_QUAD;
      it.q[13] = bits_to_float((uint32_t)i);
      it.q[14] = bits_to_float((uint32_t)j);
      // generic corner index (0 = O / far corner, 1 = X, 2 = Y) of each triangle's v1 and v2
      auto corner_i = [&](int v) { return v == zi ? 0u : (v == xi ? 1u : 2u); };
      auto corner_j = [&](int v) { return v == zj ? 0u : (v == mi[xi] ? 1u : 2u); };
      it.q[15] = bits_to_float((corner_i(1) | (corner_i(2) << 2)) | ((corner_j(1) | (corner_j(2) << 2)) << 8));
      acc.items.push_back(it);
      used[j] = 1;
      merged = true;
    }
    if (!merged) emit_tri(i);
  }
  // ---- parallelograms that are faces of one parallelepiped become ONE box item: three slab pairs in the box's own
  // (skew) coordinates instead of one plane test per face.  All six faces (Cornell's two blocks, veach-mis's four
  // plates), or five with the sixth open (Cornell's room: floor, ceiling, back and side walls -- five instances).
  // A line meets a convex box in its entry and exit faces, which are exactly the parallelograms the separate tests
  // would report; an open face reports nothing.  Cornell: 18 items -> 4; veach-mis: 29 -> 9.
  {
    auto add = [](V3 p, V3 q) { return V3{p.x + q.x, p.y + q.y, p.z + q.z}; };
    auto corner = [&](V3 O, V3 a, V3 b, V3 c, int m) {
      V3 p = O;
      if (m & 1) p = add(p, a);
      if (m & 2) p = add(p, b);
      if (m & 4) p = add(p, c);
      return p;
    };
    const size_t n_items = acc.items.size();
    float scale = 0.f;
    for (size_t j = 0; j < n_items; ++j)
      if (geo[j].quad)
        for (V3 p : {geo[j].O, add(geo[j].O, geo[j].a), add(geo[j].O, geo[j].b)})
          scale = std::max(scale, std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
    const float tol = 8e-6f * std::max(scale, 1e-3f);
    struct Face { int item = -1; uint32_t map = 0; };
    // which free parallelograms are faces of the box (O, e[0], e[1], e[2]); returns how many of the six were found
    std::vector<char> gone(n_items, 0);
    auto faces_of = [&](V3 O, const V3* e, Face* face) {
      int found_faces = 0;
      for (int fi = 0; fi < 6; ++fi) face[fi] = Face();
      for (size_t f = 0; f < n_items; ++f) {
        if (gone[f] || !geo[f].quad) continue;
        int cc[3][3];
        V3 pts[3] = {geo[f].O, add(geo[f].O, geo[f].a), add(geo[f].O, geo[f].b)};
        bool all = true;
        for (int v = 0; v < 3 && all; ++v) {
          bool found = false;
          for (int m = 0; m < 8 && !found; ++m)
            if (close(pts[v], corner(O, e[0], e[1], e[2], m), tol)) {
              cc[v][0] = m & 1; cc[v][1] = (m >> 1) & 1; cc[v][2] = m >> 2;
              found = true;
            }
          all = found;
        }
        if (!all) continue;
        // the face's axis: the coordinate all three corners share; its s / r run along the other two
        int axis = -1, s_ax = -1, r_ax = -1;
        for (int ax = 0; ax < 3; ++ax) {
          if (cc[0][ax] == cc[1][ax] && cc[0][ax] == cc[2][ax]) axis = ax;
          if (cc[0][ax] != cc[1][ax]) s_ax = ax;
          if (cc[0][ax] != cc[2][ax]) r_ax = ax;
        }
        if (axis < 0 || s_ax < 0 || r_ax < 0 || s_ax == r_ax || s_ax == axis || r_ax == axis) continue;
        int fi = axis * 2 + cc[0][axis];
        if (face[fi].item >= 0) continue;  // a duplicate face stays a separate item
        face[fi].item = (int)f;
        // in-face coordinates are the two box coordinates != axis in ascending order: index 0 or 1
        auto in_face = [&](int ax) { return ax > axis ? ax - 1 : ax; };
        face[fi].map = (uint32_t)in_face(s_ax) | ((uint32_t)(cc[0][s_ax] ? 1 : 0) << 2) |
                       ((uint32_t)in_face(r_ax) << 3) | ((uint32_t)(cc[0][r_ax] ? 1 : 0) << 5);
        found_faces++;
      }
      return found_faces;
    };
    struct Box6 { size_t first; SmallItem item; SmallItem aux; };
    std::vector<Box6> boxes;
    for (size_t i0 = 0; i0 < n_items; ++i0) {
      if (gone[i0] || !geo[i0].quad) continue;
      const Geo& q0 = geo[i0];
      bool made = false;
      // the third edge: a corner of another parallelogram, so that both are faces of O + {0,1}a + {0,1}b + {0,1}c
      for (size_t j = 0; j < n_items && !made; ++j) {
        if (j == i0 || gone[j] || !geo[j].quad) continue;
        for (int cj = 0; cj < 4 && !made; ++cj) {
          V3 e[3] = {q0.a, q0.b, sub(corner(geo[j].O, geo[j].a, geo[j].b, V3{0, 0, 0}, cj), q0.O)};
          Face face[6];
          int nf = faces_of(q0.O, e, face);
          if (nf < 5) continue;
          V3 O = q0.O;
          uint32_t open_flag = 0;
          if (nf == 5) {  // put the open face on the third axis, the one whose two parameters the loop still holds
            int miss = 0;
            for (int fi = 0; fi < 6; ++fi)
              if (face[fi].item < 0) miss = fi;
            const int m_ax = miss >> 1;
            if (m_ax != 2) {
              V3 t = e[m_ax];
              e[m_ax] = e[2];
              e[2] = t;
              if (faces_of(O, e, face) != 5) continue;
              for (int fi = 0; fi < 6; ++fi)
                if (face[fi].item < 0) miss = fi;
            }
            if ((miss >> 1) != 2) continue;
            open_flag = 1u + (uint32_t)(miss & 1);
          }
          // reciprocal basis of (a, b, c): a' = (b x c) / det, b' = (c x a) / det, c' = (a x b) / det
          const double A[3] = {e[0].x, e[0].y, e[0].z}, B[3] = {e[1].x, e[1].y, e[1].z}, Cc[3] = {e[2].x, e[2].y, e[2].z};
          auto crossd = [](const double* u, const double* v, double* o) {
            o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0];
          };
          double bc[3], ca[3], ab[3];
          crossd(B, Cc, bc); crossd(Cc, A, ca); crossd(A, B, ab);
          const double det = A[0] * bc[0] + A[1] * bc[1] + A[2] * bc[2];
          if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) continue;
          Box6 bx;
          bx.first = i0;
          std::memset(&bx.item, 0, sizeof(SmallItem));
          std::memset(&bx.aux, 0, sizeof(SmallItem));
          const double* rec[3] = {bc, ca, ab};
          for (int ax = 0; ax < 3; ++ax) {
            double r0 = rec[ax][0] / det, r1 = rec[ax][1] / det, r2 = rec[ax][2] / det;
            bx.item.q[4 * ax + 0] = (float)r0; bx.item.q[4 * ax + 1] = (float)r1; bx.item.q[4 * ax + 2] = (float)r2;
            bx.item.q[4 * ax + 3] = (float)-(r0 * O.x + r1 * O.y + r2 * O.z);
          }
          bx.item.q[12] = SMALL_KIND_BOX;
          bx.item.q[14] = bits_to_float(open_flag);
          for (int fi = 0; fi < 6; ++fi) {
            if (face[fi].item < 0) continue;
            const SmallItem& src = acc.items[face[fi].item];
            uint32_t s1 = float_bits(src.q[13]), s2 = float_bits(src.q[14]), perm = float_bits(src.q[15]);
            bx.aux.q[2 * fi] = bits_to_float((s1 & 0xffu) | ((s2 & 0xffu) << 8) | ((perm & 0xffffu) << 16));
            bx.aux.q[2 * fi + 1] = bits_to_float(face[fi].map);
            gone[face[fi].item] = 1;
          }
          boxes.push_back(bx);
          made = true;
        }
      }
    }
    if (!boxes.empty()) {
      std::vector<SmallItem> items;
      std::sort(boxes.begin(), boxes.end(), [](const Box6& x, const Box6& y) { return x.first < y.first; });
      for (size_t i = 0; i < n_items; ++i) {
        for (const Box6& bx : boxes)
          if (bx.first == i) items.push_back(bx.item);
        if (!gone[i]) items.push_back(acc.items[i]);
      }
      // aux records follow the loop items, in the order of their boxes; a box item names its own by absolute index
      const size_t n_loop = items.size();
      size_t nb = 0;
      for (SmallItem& it : items)
        if (it.q[12] == SMALL_KIND_BOX) it.q[13] = bits_to_float((uint32_t)(n_loop + nb++));
      for (const Box6& bx : boxes) items.push_back(bx.aux);
      acc.items = std::move(items);
      acc.n_loop = (uint32_t)n_loop;
    } else {
      acc.n_loop = (uint32_t)acc.items.size();
    }
  }
  if (acc.n_loop > SMALL_MAX_ITEMS || acc.isect.size() > 255) { acc.items.clear(); acc.n_loop = 0; }
}

void finish_accel(const std::vector<Prim>& prims, uint32_t max_leaf, BuiltAccel& out,
                  std::vector<uint32_t>& order) {
  Builder b(prims, max_leaf);
  b.build();
  Collapse4 wide(b.nodes);
  out.depth = wide.run();
  out.nodes = std::move(wide.out);
  // The tree's top levels first, in breadth-first order (every ray starts there): nodes [0, n_top) are whole levels, at
  // most TOP_NODES_MAX of them; the rest keep their depth-first order.  Inner child words are renumbered to match.
  {
    const size_t n = out.nodes.size();
    std::vector<uint32_t> fresh(n, 0xffffffffu), bfs;
    std::vector<uint32_t> level{0u};
    auto inner_children = [&](uint32_t i, std::vector<uint32_t>& to) {
      for (int c = 0; c < 4; ++c) {
        const uint32_t w = float_bits(out.nodes[i].q[4 + c]);
        if (!(w & LEAF_BIT)) to.push_back(w);
      }
    };
    while (!level.empty() && bfs.size() + level.size() <= TOP_NODES_MAX) {
      std::vector<uint32_t> next;
      for (uint32_t i : level) {
        fresh[i] = (uint32_t)bfs.size();
        bfs.push_back(i);
        inner_children(i, next);
      }
      level.swap(next);
    }
    out.n_top = (uint32_t)bfs.size();
    uint32_t k = out.n_top;
    for (size_t i = 0; i < n; ++i)
      if (fresh[i] == 0xffffffffu) fresh[i] = k++;
    std::vector<Node> moved(n);
    for (size_t i = 0; i < n; ++i) {
      Node nd = out.nodes[i];
      for (int c = 0; c < 4; ++c) {
        const uint32_t w = float_bits(nd.q[4 + c]);
        if (!(w & LEAF_BIT)) nd.q[4 + c] = bits_to_float(fresh[w]);
      }
      moved[fresh[i]] = nd;
    }
    out.nodes.swap(moved);
  }
  out.isect.resize(prims.size());
  for (size_t s = 0; s < prims.size(); ++s) out.isect[s] = prims[b.order[s]].isect;
  order = std::move(b.order);
  build_small_items(out);
}

}  // namespace

int pack_scene(const rene_scene_desc* d, PackedScene& out, std::string& err) {
  if (!d) { err = "scene is NULL"; return RENE_ERR_INVALID_ARGUMENT; }
  if (d->struct_size != sizeof(rene_scene_desc)) {
    err = "rene_scene_desc.struct_size mismatch (ABI skew)";
    return RENE_ERR_INVALID_ARGUMENT;
  }
  if (d->integrator != RENE_INTEGRATOR_PATH && d->integrator != RENE_INTEGRATOR_VOLPATH) {
    err = "unknown integrator";
    return RENE_ERR_INVALID_SCENE;
  }
  if (d->n_mediums && !d->mediums) { err = "mediums is NULL"; return RENE_ERR_INVALID_ARGUMENT; }
  if (d->xresolution < 2 || d->yresolution < 2) {  // lib.rs:178-179 divides by W-1, H-1
    err = "film resolution must be at least 2x2";
    return RENE_ERR_INVALID_SCENE;
  }
  if (!d->n_materials || !d->n_area_lights || !d->n_textures) {
    err = "materials / area_lights / textures need their index-0 sentinels (rene/src/scene.rs:109-116)";
    return RENE_ERR_INVALID_SCENE;
  }
  out = PackedScene();
  out.width = d->xresolution;
  out.height = d->yresolution;
  out.uniform = d->uniform;
  if (d->uniform.background_texture >= d->n_textures) { err = "background_texture out of range"; return RENE_ERR_INVALID_SCENE; }

  // ---- tables ----
  out.materials.resize(d->n_materials);
  uint32_t lobe_kinds = 0;  // 1 specular (Glass / Mirror), 2 FresnelBlend (Substrate), 4 microfacet (Metal)
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    const rene_material& m = d->materials[i];
    Material& o = out.materials[i];
    std::memset(&o, 0, sizeof(o));
    o.type = m.type;
    std::memcpy(o.u0, m.u0, 16);
    std::memcpy(o.u1, m.u1, 16);
    std::memcpy(o.v0, m.v0, 16);
    if (m.type > RENE_MATERIAL_PLASTIC) { err = "unknown material type"; return RENE_ERR_INVALID_SCENE; }
    // texture indices that the material will dereference
    uint32_t refs[7];
    int nrefs = 0;
    switch (m.type) {
      case RENE_MATERIAL_MATTE: case RENE_MATERIAL_MIRROR: refs[nrefs++] = m.u0[0]; break;
      case RENE_MATERIAL_SUBSTRATE: case RENE_MATERIAL_METAL:
        for (int k = 0; k < 4; ++k) refs[nrefs++] = m.u0[k];
        break;
      case RENE_MATERIAL_UBER:
        for (int k = 0; k < 4; ++k) refs[nrefs++] = m.u0[k];
        refs[nrefs++] = m.u1[0]; refs[nrefs++] = m.u1[2]; refs[nrefs++] = m.u1[3];
        break;
      case RENE_MATERIAL_PLASTIC: refs[nrefs++] = m.u0[0]; refs[nrefs++] = m.u0[1]; refs[nrefs++] = m.u0[3]; break;
      default: break;
    }
    for (int k = 0; k < nrefs; ++k)
      if (refs[k] >= d->n_textures) { err = "material references a texture out of range"; return RENE_ERR_INVALID_SCENE; }
    if (m.type != RENE_MATERIAL_NONE && m.type != RENE_MATERIAL_MATTE) out.features |= FEAT_GENERAL_BSDF;
    if (m.type == RENE_MATERIAL_UBER || m.type == RENE_MATERIAL_PLASTIC) out.features |= FEAT_MULTI_LOBE;
    if (m.type == RENE_MATERIAL_GLASS || m.type == RENE_MATERIAL_MIRROR) lobe_kinds |= 1u;
    if (m.type == RENE_MATERIAL_SUBSTRATE) lobe_kinds |= 2u;
    if (m.type == RENE_MATERIAL_METAL) lobe_kinds |= 4u;
  }
  if ((out.features & FEAT_GENERAL_BSDF) && !(out.features & FEAT_MULTI_LOBE)) {  // what the single-lobe scene leaves out
    if (!(lobe_kinds & 1u)) out.features |= FEAT_NO_SPECULAR;
    if (!(lobe_kinds & 2u)) out.features |= FEAT_NO_BLEND;
    if (!(lobe_kinds & 4u)) out.features |= FEAT_NO_MICROFACET;
  }
  out.textures.resize(d->n_textures);
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const rene_texture& t = d->textures[i];
    Texture& o = out.textures[i];
    std::memset(&o, 0, sizeof(o));
    o.type = t.type;
    std::memcpy(o.u0, t.u0, 16);
    std::memcpy(o.v0, t.v0, 16);
    if (t.type > RENE_TEXTURE_SCALE) { err = "unknown texture type"; return RENE_ERR_INVALID_SCENE; }
    if (t.type != RENE_TEXTURE_SOLID) out.features |= FEAT_TEXTURES;
    if ((t.type == RENE_TEXTURE_CHECKERBOARD || t.type == RENE_TEXTURE_SCALE) &&
        (t.u0[0] >= d->n_textures || t.u0[1] >= d->n_textures)) {
      err = "texture references a texture out of range";
      return RENE_ERR_INVALID_SCENE;
    }
    if (t.type == RENE_TEXTURE_IMAGEMAP && t.u0[0] >= d->n_images) { err = "imagemap references an image out of range"; return RENE_ERR_INVALID_SCENE; }
  }
  for (uint32_t i = 0; i < d->n_images; ++i) {
    const rene_image& im = d->images[i];
    if (!im.rgba || !im.width || !im.height) { err = "empty image"; return RENE_ERR_INVALID_SCENE; }
    ImageRef r{(uint64_t)out.image_pool.size(), im.width, im.height};
    out.image_pool.insert(out.image_pool.end(), im.rgba, im.rgba + (size_t)4 * im.width * im.height);
    out.images.push_back(r);
  }
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    if (d->lights[i].type != RENE_LIGHT_DISTANT) { err = "unknown light type"; return RENE_ERR_INVALID_SCENE; }
    Light l;
    std::memcpy(l.dir, d->lights[i].v0, 16);
    std::memcpy(l.L, d->lights[i].v1, 16);
    out.lights.push_back(l);
  }
  if (d->n_lights) out.features |= FEAT_LIGHTS;
  if (d->uniform.background_color[0] != 0.f || d->uniform.background_color[1] != 0.f ||
      d->uniform.background_color[2] != 0.f)
    out.features |= FEAT_BACKGROUND;

  // ---- media (scene.rs:111, 405-416).  An absent table means "just the vacuum". ----
  const uint32_t n_mediums = d->n_mediums ? d->n_mediums : 1u;
  if (d->integrator == RENE_INTEGRATOR_VOLPATH) {
    out.features |= FEAT_VOLPATH;
    out.mediums.resize(n_mediums);
    std::memset(out.mediums.data(), 0, n_mediums * sizeof(Medium));
    for (uint32_t i = 0; i < d->n_mediums; ++i) {
      const rene_medium& m = d->mediums[i];
      if (m.type > RENE_MEDIUM_HOMOGENEOUS) { err = "unknown medium type"; return RENE_ERR_INVALID_SCENE; }
      for (int k = 0; k < 4; ++k)
        if (!std::isfinite(m.v0[k]) || !std::isfinite(m.v1[k])) { err = "non-finite medium coefficient"; return RENE_ERR_INVALID_SCENE; }
      std::memcpy(out.mediums[i].sa_g, m.v0, 16);
      std::memcpy(out.mediums[i].ss_t, m.v1, 12);
      std::memcpy(&out.mediums[i].ss_t[3], &m.type, 4);
    }
  }

  // ---- flatten instances to world space ----
  std::vector<Prim> prims, eprims;
  for (uint32_t ii = 0; ii < d->n_instances; ++ii) {
    const rene_instance& in = d->instances[ii];
    if (in.material_index >= d->n_materials || in.area_light_index >= d->n_area_lights) {
      err = "instance references a material / area light out of range";
      return RENE_ERR_INVALID_SCENE;
    }
    if (in.interior_medium_index >= n_mediums || in.exterior_medium_index >= n_mediums) {
      err = "instance references a medium out of range";
      return RENE_ERR_INVALID_SCENE;
    }
    if (out.features & FEAT_VOLPATH) out.inst_medium.push_back(InstMedium{in.interior_medium_index, in.exterior_medium_index});
    for (int k = 0; k < 12; ++k)
      if (!std::isfinite(in.matrix[k])) { err = "non-finite instance matrix"; return RENE_ERR_INVALID_SCENE; }
    Aff o2w = aff_from(in.matrix), w2o;
    if (!aff_inverse(o2w, w2o)) { err = "singular instance matrix"; return RENE_ERR_INVALID_SCENE; }
    const rene_material& mat = d->materials[in.material_index];
    const rene_area_light& al = d->area_lights[in.area_light_index];
    bool emitter = al.type != RENE_AREA_LIGHT_NULL;
    Inst inst;
    std::memset(&inst, 0, sizeof(inst));
    inst.material = in.material_index;
    inst.area_light = in.area_light_index;
    inst.material_type = mat.type;
    if (mat.type == RENE_MATERIAL_MATTE && d->textures[mat.u0[0]].type == RENE_TEXTURE_SOLID) {
      const float* c = d->textures[mat.u0[0]].v0;
      inst.kd[0] = c[0]; inst.kd[1] = c[1]; inst.kd[2] = c[2]; inst.kd[3] = 1.0f;
    }
    if (emitter) { inst.emit[0] = al.v0[0]; inst.emit[1] = al.v0[1]; inst.emit[2] = al.v0[2]; inst.emit[3] = 1.0f; }
    if (!std::getenv("RENE_NO_RESOLVE")) {  // single-lobe general materials over Solid textures only, resolved here (device_scene.h, Inst::res_*); the knob is for A/B tests
      auto solid = [&](uint32_t t) { return d->textures[t].type == RENE_TEXTURE_SOLID; };
      auto rgb = [&](float* dst, uint32_t t) { std::memcpy(dst, d->textures[t].v0, 12); };
      switch (mat.type) {
        case RENE_MATERIAL_GLASS:  // material.rs:342-350
          inst.res_type = mat.type;
          inst.res_c0[0] = mat.v0[0];
          break;
        case RENE_MATERIAL_SUBSTRATE:  // material.rs:188-216: Kd, Ks, uroughness, vroughness
        case RENE_MATERIAL_METAL:      // material.rs:279-307: eta, k, uroughness, vroughness
          if (solid(mat.u0[0]) && solid(mat.u0[1]) && solid(mat.u0[2]) && solid(mat.u0[3])) {
            inst.res_type = mat.type;
            inst.res_remap = mat.u1[0];
            rgb(inst.res_c0, mat.u0[0]);
            rgb(inst.res_c1, mat.u0[1]);
            inst.res_ru = d->textures[mat.u0[2]].v0[0];
            inst.res_rv = d->textures[mat.u0[3]].v0[0];
          }
          break;
        case RENE_MATERIAL_MATTE: {  // over a checkerboard of two solid textures (texture.rs:97-118): the teapot scene's floor
          const rene_texture& t = d->textures[mat.u0[0]];
          if (t.type == RENE_TEXTURE_CHECKERBOARD && solid(t.u0[0]) && solid(t.u0[1])) {
            inst.res_type = INST_RES_MATTE_CHECKER;
            inst.res_ru = t.v0[0];
            inst.res_rv = t.v0[1];
            rgb(inst.res_c0, t.u0[0]);
            rgb(inst.res_c1, t.u0[1]);
          }
          break;
        }
        case RENE_MATERIAL_MIRROR:  // material.rs:363-373
          if (solid(mat.u0[0])) {
            inst.res_type = mat.type;
            rgb(inst.res_c0, mat.u0[0]);
          }
          break;
        default: break;
      }
    }

    if (in.shape == RENE_SHAPE_SPHERE) {
      out.features |= FEAT_SPHERES;
      inst.primitive_count = 1.0f;  // main.rs:3071-3074
      Sphere s;
      std::memcpy(s.o2w, in.matrix, 48);
      const float w[12] = {w2o.x.x, w2o.x.y, w2o.x.z, w2o.y.x, w2o.y.y, w2o.y.z,
                           w2o.z.x, w2o.z.y, w2o.z.z, w2o.w.x, w2o.w.y, w2o.w.z};
      std::memcpy(s.w2o, w, 48);
      uint32_t sidx = (uint32_t)out.spheres.size();
      out.spheres.push_back(s);
      Prim p;
      std::memset(&p, 0, sizeof(p));
      p.sphere = true;
      p.box.reset();
      for (int c = 0; c < 8; ++c)  // unit AABB BLAS, main.rs:2444-2451
        p.box.grow(xform_point(o2w, V3{c & 1 ? 1.f : -1.f, c & 2 ? 1.f : -1.f, c & 4 ? 1.f : -1.f}));
      pad_box(p.box);
      p.isect.q[9] = bits_to_float(ii);
      p.isect.q[10] = bits_to_float(0u);
      p.isect.q[11] = bits_to_float(sidx);
      {  // a translated, uniformly scaled unit sphere (every sphere the pbrt loader makes): centre and radius^2 for
         // the item loop's short form of sphere_intersection; q[4] = 1 marks it
        const float* m = in.matrix;
        const float r = m[0];
        const float tolr = 1e-6f * std::fabs(r);
        if (r > 0.0f && std::fabs(m[4] - r) <= tolr && std::fabs(m[8] - r) <= tolr && std::fabs(m[1]) <= tolr &&
            std::fabs(m[2]) <= tolr && std::fabs(m[3]) <= tolr && std::fabs(m[5]) <= tolr && std::fabs(m[6]) <= tolr &&
            std::fabs(m[7]) <= tolr) {
          p.isect.q[0] = m[9]; p.isect.q[1] = m[10]; p.isect.q[2] = m[11];
          p.isect.q[3] = r * r;
          p.isect.q[4] = 1.0f;
        }
      }
      prims.push_back(p);
      if (emitter) {
        eprims.push_back(p);
        EmitObject eo;
        std::memset(&eo, 0, sizeof(eo));
        eo.type = 1;
        eo.prim_count = 1;
        std::memcpy(eo.matrix, in.matrix, 48);
        out.emit_objects.push_back(eo);
      }
    } else if (in.shape == RENE_SHAPE_TRIANGLE) {
      if (in.mesh_index < 0 || (uint32_t)in.mesh_index >= d->n_meshes) { err = "instance mesh_index out of range"; return RENE_ERR_INVALID_SCENE; }
      const rene_mesh& me = d->meshes[in.mesh_index];
      if (me.n_indices % 3) { err = "mesh index count is not a multiple of 3"; return RENE_ERR_INVALID_SCENE; }
      uint32_t ntri = me.n_indices / 3;
      inst.primitive_count = (float)ntri;
      EmitObject eo;
      if (emitter) {
        std::memset(&eo, 0, sizeof(eo));
        eo.type = 0;
        eo.first_tri = (uint32_t)out.emit_tris.size();
        eo.prim_count = ntri;
        std::memcpy(eo.matrix, in.matrix, 48);
        if (ntri == 0) { err = "emitter mesh without triangles"; return RENE_ERR_INVALID_SCENE; }  // `% 0`, surface_sample.rs:75
        out.emit_objects.push_back(eo);
      }
      for (uint32_t t = 0; t < ntri; ++t) {
        const rene_vertex* v[3];
        for (int k = 0; k < 3; ++k) {
          uint32_t idx = me.indices[3 * t + k];
          if (idx >= me.n_vertices) { err = "mesh index out of range"; return RENE_ERR_INVALID_SCENE; }
          v[k] = &me.vertices[idx];
        }
        V3 po[3], pw[3], nw[3];
        bool all_zero = true;
        for (int k = 0; k < 3; ++k) {
          po[k] = V3{v[k]->position[0], v[k]->position[1], v[k]->position[2]};
          pw[k] = xform_point(o2w, po[k]);
          if (v[k]->normal[0] != 0.f || v[k]->normal[1] != 0.f || v[k]->normal[2] != 0.f) all_zero = false;
        }
        V3 ng_obj = cross(sub(po[1], po[0]), sub(po[2], po[0]));
        V3 ng_w = xform_normal(w2o, ng_obj);
        for (int k = 0; k < 3; ++k)
          nw[k] = all_zero ? ng_w : xform_normal(w2o, V3{v[k]->normal[0], v[k]->normal[1], v[k]->normal[2]});
        Prim p;
        std::memset(&p, 0, sizeof(p));
        p.sphere = false;
        p.box.reset();
        for (int k = 0; k < 3; ++k) p.box.grow(pw[k]);
        pad_box(p.box);
        V3 e1 = sub(pw[1], pw[0]), e2 = sub(pw[2], pw[0]);
        float* q = p.isect.q;
        q[0] = pw[0].x; q[1] = pw[0].y; q[2] = pw[0].z; q[3] = e1.x; q[4] = e1.y; q[5] = e1.z;
        q[6] = e2.x; q[7] = e2.y; q[8] = e2.z;
        q[9] = bits_to_float(ii); q[10] = bits_to_float(t); q[11] = bits_to_float(0xffffffffu);
        float* s = p.shade.q;
        s[0] = nw[0].x; s[1] = nw[0].y; s[2] = nw[0].z; s[3] = v[0]->uv[0];
        s[4] = nw[1].x; s[5] = nw[1].y; s[6] = nw[1].z; s[7] = v[0]->uv[1];
        s[8] = nw[2].x; s[9] = nw[2].y; s[10] = nw[2].z; s[11] = v[1]->uv[0];
        s[12] = v[1]->uv[1]; s[13] = v[2]->uv[0]; s[14] = v[2]->uv[1]; s[15] = 0.f;
        prims.push_back(p);
        if (emitter) {
          // triangle_closest_hit_pdf operands, lib.rs:1004-1036
          float len = std::sqrt(dot(ng_w, ng_w));
          V3 c = cross(e1, e2);
          p.pdf.q[0] = ng_w.x / len; p.pdf.q[1] = ng_w.y / len; p.pdf.q[2] = ng_w.z / len;
          p.pdf.q[3] = 0.5f * std::sqrt(dot(c, c));
          eprims.push_back(p);
          EmitTri et;
          std::memset(&et, 0, sizeof(et));
          et.q[0] = pw[0].x; et.q[1] = pw[0].y; et.q[2] = pw[0].z; et.q[3] = pw[1].x; et.q[4] = pw[1].y;
          et.q[5] = pw[1].z; et.q[6] = pw[2].x; et.q[7] = pw[2].y; et.q[8] = pw[2].z;
          out.emit_tris.push_back(et);
        }
      }
      out.n_triangles += ntri;
    } else {
      err = "unknown instance shape";
      return RENE_ERR_INVALID_SCENE;
    }
    out.insts.push_back(inst);
  }
  if (prims.size() > LEAF_FIRST_MASK) { err = "too many primitives for the 26-bit leaf index"; return RENE_ERR_UNSUPPORTED; }
  if (out.emit_objects.empty()) out.features |= FEAT_NO_EMITTERS;

  std::vector<uint32_t> order;
  uint32_t max_leaf = 2;  // measured with the BVH4 (MI355X, Grays/s, dragon- / teapot-class): 1: 4.42 / 4.23, 2: 4.82 / 4.63, 3: 4.56 / 4.47, 4: 4.35 / 4.32, 8: 3.66 / 3.86
  if (const char* e = std::getenv("RENE_MAX_LEAF")) max_leaf = (uint32_t)std::max(1, std::min(16, std::atoi(e)));  // tuning knob
  finish_accel(prims, max_leaf, out.main, order);
  out.shade.resize(prims.size());
  for (size_t s = 0; s < prims.size(); ++s) out.shade[s] = prims[order[s]].shade;
  finish_accel(eprims, max_leaf, out.emit, order);
  out.emit_pdf.resize(eprims.size());
  for (size_t s = 0; s < eprims.size(); ++s) out.emit_pdf[s] = eprims[order[s]].pdf;
  if (!out.main.items.empty() && (eprims.empty() || !out.emit.items.empty())) {
    // the LDS image of the small-scene kernels (device_scene.h): items for the hit mapping, one fat record per slot
    std::vector<float>& img = out.small_image;
    auto append = [&](const float* p, size_t n) { img.insert(img.end(), p, p + n); };
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    for (const SmallItem& it : out.main.items) append(it.q, 16);
    out.small_off[SMALL_OFF_EMIT_ITEMS] = (uint32_t)(img.size() * 4);
    for (const SmallItem& it : out.emit.items) append(it.q, 16);
    out.small_off[SMALL_OFF_HIT] = (uint32_t)(img.size() * 4);
    for (size_t s = 0; s < out.main.isect.size(); ++s) {
      append(out.main.isect[s].q, 12);
      append(out.shade[s].q, 16);
      const Inst& in = out.insts[bits(out.main.isect[s].q[9])];
      static_assert(sizeof(Inst) == 96 && SMALL_HIT_FLOATS == 12 + 16 + 24, "Inst is six float4");
      append(reinterpret_cast<const float*>(&in), 24);
    }
    out.small_off[SMALL_OFF_EMIT] = (uint32_t)(img.size() * 4);
    for (size_t s = 0; s < out.emit.isect.size(); ++s) {
      append(out.emit.isect[s].q, 12);
      append(out.emit_pdf[s].q, 4);
      const float tail[4] = {out.insts[bits(out.emit.isect[s].q[9])].primitive_count, 0.f, 0.f, 0.f};
      append(tail, 4);
    }
    out.small_off[SMALL_OFF_EOBJ] = (uint32_t)(img.size() * 4);
    static_assert(sizeof(EmitObject) == 64 && sizeof(EmitTri) == 48, "LDS image record sizes");
    for (const EmitObject& e : out.emit_objects) append(reinterpret_cast<const float*>(&e), 16);
    out.small_off[SMALL_OFF_ETRI] = (uint32_t)(img.size() * 4);
    for (const EmitTri& e : out.emit_tris) append(e.q, 12);
    out.small_off[SMALL_OFF_SPHERES] = (uint32_t)(img.size() * 4);
    static_assert(sizeof(Sphere) == 96, "LDS image record sizes");
    for (const Sphere& sp : out.spheres) append(reinterpret_cast<const float*>(&sp), 24);
    if (img.size() * 4 <= SMALL_LDS_MAX_BYTES) out.features |= FEAT_SMALL;
    else img.clear();  // too many slots for the LDS-resident tables: the BVH kernels render it
  }
  return RENE_OK;
}

}  // namespace rene
