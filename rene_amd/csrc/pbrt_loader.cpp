// pbrt_loader.cpp -- caller side of the render path: rene's pbrt-v3 subset parser and Scene::create,
// restated in C++ (host only) behind rene_scene_load_pbrt / rene_scene_parse_pbrt.
//
// Grammar:   pbrt-parser/src/lib.rs:114-577  (chumsky combinators -> a hand-written recursive descent)
// Includes:  pbrt-parser/src/include.rs:36-84
// Typing:    rene/src/scene/intermediate_scene.rs:263-1107 (arguments, defaults, shapes, PLY, PFM)
// Assembly:  rene/src/scene.rs:100-460 (Scene::create, append_world)
//
// Deliberately reproduced quirks: TransformBegin/End is an *attribute*
// scope (lib.rs:561-566); ReverseOrientation is a no-op (scene.rs:266-268); ObjectInstance composes
// object.matrix * CTM (scene.rs:296); "mirror" reads Kd (intermediate_scene.rs:516-521); unknown
// integrators select volpath (intermediate_scene.rs:1069-1072); Sampler / PixelFilter and all
// Integrator parameters are ignored (scene.rs:120-128).
// Spectral colours ("blackbody L" [T scale], "spectrum Kd" "file.spd"; intermediate_scene.rs:272-285, spectrum.rs:1468-1521)
// are converted with the CIE 1931 2-degree matching functions at 1 nm (cie1931.inc: the standard's table, which the
// reference tabulates too) by from_sampled restated in f32, statement for statement: a `.spd` colour equals the
// reference's.  Blackbody RGB comes from the crate `blackbody` 0.0.0 upstream, whose source is not in the checkout: here it
// is pbrt-v3's own definition (the peak-normalised Planck spectrum through the same from_sampled) -- that one stays UNPINNED.
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rene_hip.h"

namespace rene {
void set_last_error(const std::string& msg);  // rene_hip.cpp: the thread-local behind rene_last_error()
std::string loop_subdivide(std::vector<rene_vertex>& verts, std::vector<uint32_t>& idx, unsigned levels);  // loop_subdiv.cpp
// image_io.cpp: rows top first; LDR decoders give RGBA8, the EXR one linear f32
bool decode_tga(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err);
bool decode_bmp(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err);
bool decode_jpeg(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err);
bool decode_exr(const std::string& data, uint32_t& w, uint32_t& h, std::vector<float>& rgba, std::string& err);
}

namespace {

struct LoadError {
  int code;
  std::string msg;
};
[[noreturn]] void fail(int code, const std::string& m) { throw LoadError{code, m}; }

// ---- glam subset (column-major Mat4 in double, rounded to f32 at the table boundary) ------------------
struct M4 {
  double m[16];  // column-major: m[c*4 + r]
  static M4 identity() {
    M4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
  }
  double& at(int r, int c) { return m[c * 4 + r]; }
  double at(int r, int c) const { return m[c * 4 + r]; }
};
M4 mul(const M4& a, const M4& b) {
  M4 o{};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a.at(r, k) * b.at(k, c);
      o.at(r, c) = s;
    }
  return o;
}
// every matrix is rounded to f32 after each operation, as glam's f32 Mat4 would hold it
M4 round32(M4 a) {
  for (double& v : a.m) v = (double)(float)v;
  return a;
}
M4 inverse(const M4& a) {
  // Gauss-Jordan in double (glam::Mat4::inverse is a cofactor expansion in f32; unpinned)
  double w[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      w[r][c] = a.at(r, c);
      w[r][c + 4] = r == c ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (std::fabs(w[r][col]) > std::fabs(w[piv][col])) piv = r;
    if (w[piv][col] == 0.0) fail(RENE_ERR_INVALID_SCENE, "singular transform");
    if (piv != col)
      for (int c = 0; c < 8; ++c) std::swap(w[piv][c], w[col][c]);
    double d = w[col][col];
    for (int c = 0; c < 8; ++c) w[col][c] /= d;
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        double f = w[r][col];
        if (f != 0.0)
          for (int c = 0; c < 8; ++c) w[r][c] -= f * w[col][c];
      }
  }
  M4 o{};
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) o.at(r, c) = w[r][c + 4];
  return round32(o);
}
M4 from_translation(const float* t) {
  M4 r = M4::identity();
  r.at(0, 3) = t[0];
  r.at(1, 3) = t[1];
  r.at(2, 3) = t[2];
  return r;
}
M4 from_scale(const float* s) {
  M4 r = M4::identity();
  r.at(0, 0) = s[0];
  r.at(1, 1) = s[1];
  r.at(2, 2) = s[2];
  return r;
}
M4 from_axis_angle(const float* axis_in, float angle_deg) {  // intermediate_scene.rs:1035-1038
  double l = std::sqrt((double)axis_in[0] * axis_in[0] + (double)axis_in[1] * axis_in[1] + (double)axis_in[2] * axis_in[2]);
  double x = (float)(axis_in[0] / l), y = (float)(axis_in[1] / l), z = (float)(axis_in[2] / l);
  float angle = angle_deg * 3.14159265358979323846f / 180.0f;  // deg_to_radian, intermediate_scene.rs:612-614
  double s = (float)std::sin((double)angle), c = (float)std::cos((double)angle), omc = 1.0 - c;
  M4 r = M4::identity();
  r.at(0, 0) = x * x * omc + c; r.at(1, 0) = x * y * omc + z * s; r.at(2, 0) = x * z * omc - y * s;
  r.at(0, 1) = x * y * omc - z * s; r.at(1, 1) = y * y * omc + c; r.at(2, 1) = y * z * omc + x * s;
  r.at(0, 2) = x * z * omc + y * s; r.at(1, 2) = y * z * omc - x * s; r.at(2, 2) = z * z * omc + c;
  return round32(r);
}
M4 look_at_lh(const float* eye, const float* center, const float* up) {  // intermediate_scene.rs:1049-1053
  double d[3] = {(double)center[0] - eye[0], (double)center[1] - eye[1], (double)center[2] - eye[2]};
  double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double f[3] = {d[0] / dl, d[1] / dl, d[2] / dl};
  double s[3] = {up[1] * f[2] - up[2] * f[1], up[2] * f[0] - up[0] * f[2], up[0] * f[1] - up[1] * f[0]};
  double sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (double& v : s) v /= sl;
  double u[3] = {f[1] * s[2] - f[2] * s[1], f[2] * s[0] - f[0] * s[2], f[0] * s[1] - f[1] * s[0]};
  M4 r = M4::identity();
  for (int c = 0; c < 3; ++c) {
    r.at(0, c) = s[c];
    r.at(1, c) = u[c];
    r.at(2, c) = f[c];
  }
  r.at(0, 3) = -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]);
  r.at(1, 3) = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
  r.at(2, 3) = -(f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]);
  return round32(r);
}
M4 perspective_lh(double fov, double aspect, double zn, double zf) {  // scene.rs:163-164
  double s = (float)std::sin(0.5 * fov), c = (float)std::cos(0.5 * fov);
  double h = c / s, w = h / aspect, r = zf / (zf - zn);
  M4 m{};
  m.at(0, 0) = w;
  m.at(1, 1) = h;
  m.at(2, 2) = r;
  m.at(3, 2) = 1.0;
  m.at(2, 3) = -r * zn;
  return round32(m);
}
void to_f32(const M4& a, float* out) {
  for (int i = 0; i < 16; ++i) out[i] = (float)a.m[i];
}
void affine12(const M4& a, float* out) {  // Affine3A::from_mat4: x_axis, y_axis, z_axis, translation
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) out[c * 3 + r] = (float)a.at(r, c);
}

// ---- AST (pbrt-parser/src/lib.rs:6-112) ---------------------------------------------------------------------
enum class VType { Float, Bool, Integer, Rgb, BlackBody, Point, Normal, String, Texture, Spectrum };
struct Value {
  VType type = VType::Float;
  std::vector<float> f;          // Float / Rgb / Point / Normal / BlackBody (flattened)
  std::vector<int> i;            // Integer
  std::vector<bool> b;           // Bool
  std::vector<std::string> s;    // String / Texture / Spectrum
};
struct Argument {
  std::string name;
  Value value;
};
struct Object {
  std::string kind;  // directive: Camera, Film, Shape, Material, ...
  std::string t;     // type string
  std::vector<Argument> args;
  const Value* get(const std::string& n) const {  // Object::get_value, lib.rs:105-111 (first match)
    for (const auto& a : args)
      if (a.name == n) return &a.value;
    return nullptr;
  }
};
struct World;
using Worlds = std::vector<World>;
struct World {
  enum Kind { Obj, Attribute, ObjectBeginEnd, ObjectInstance, Transform, ConcatTransform, Translate, Scale, Rotate,
              CoordSysTransform, Texture, NamedMaterial, MediumInterface, ReverseOrientation } kind = Obj;
  Object obj;                 // Obj / Texture (t = class, name/value_type below)
  std::string name, name2;    // Texture: name, value_type; ObjectBeginEnd / ObjectInstance / NamedMaterial / CoordSys
  float v[16] = {0};          // matrices / vectors / angle in v[0] + axis in v[1..3]
  std::shared_ptr<Worlds> children;
};
struct SceneStmt {
  enum Kind { Transform, ConcatTransform, LookAt, Rotate, Scale, Translate, SceneObject, WorldBlock } kind = Transform;
  float v[16] = {0};
  Object obj;
  Worlds world;
};

// ---- lexer / parser -------------------------------------------------------------------------------------------
struct Parser {
  const std::string& src;
  size_t pos = 0;
  explicit Parser(const std::string& s) : src(s) {}

  [[noreturn]] void error(const std::string& what) {
    size_t line = 1, col = 1;
    for (size_t i = 0; i < pos && i < src.size(); ++i) {
      if (src[i] == '\n') { line++; col = 1; } else col++;
    }
    std::ostringstream o;
    o << "parse error at line " << line << ", column " << col << ": " << what;
    if (pos < src.size()) o << " (found '" << src.substr(pos, 12) << "')"; else o << " (end of input)";
    fail(RENE_ERR_PARSE, o.str());
  }
  bool eof() const { return pos >= src.size(); }
  void sp() {  // lib.rs:120-129: whitespace and `#` comments
    for (;;) {
      while (pos < src.size() && std::isspace((unsigned char)src[pos])) pos++;
      if (pos < src.size() && src[pos] == '#') {
        while (pos < src.size() && src[pos] != '\n') pos++;
        continue;
      }
      break;
    }
  }
  bool peek_word(const char* w) const {
    size_t n = std::strlen(w);
    return src.compare(pos, n, w) == 0;
  }
  bool accept_word(const char* w) {  // chumsky `just(w)`: a prefix match, no word-boundary check
    if (peek_word(w)) {
      pos += std::strlen(w);
      return true;
    }
    return false;
  }
  void expect(char c) {
    if (pos >= src.size() || src[pos] != c) error(std::string("expected '") + c + "'");
    pos++;
  }
  float number_float() {  // lib.rs:131-148
    size_t b = pos;
    if (pos < src.size() && src[pos] == '-') pos++;
    size_t digits = 0;
    while (pos < src.size() && std::isdigit((unsigned char)src[pos])) { pos++; digits++; }
    if (pos < src.size() && src[pos] == '.' && pos + 1 < src.size() && std::isdigit((unsigned char)src[pos + 1])) {
      pos++;
      while (pos < src.size() && std::isdigit((unsigned char)src[pos])) { pos++; digits++; }
    }
    if (!digits) {
      pos = b;
      error("expected a float");
    }
    if (pos < src.size() && (src[pos] == 'e' || src[pos] == 'E')) {
      size_t save = pos;
      pos++;
      if (pos < src.size() && (src[pos] == '+' || src[pos] == '-')) pos++;
      if (pos < src.size() && std::isdigit((unsigned char)src[pos])) {
        while (pos < src.size() && std::isdigit((unsigned char)src[pos])) pos++;
      } else {
        pos = save;
      }
    }
    return std::strtof(src.substr(b, pos - b).c_str(), nullptr);  // Rust f32::from_str: correctly rounded
  }
  int number_int() {  // lib.rs:150-158
    size_t b = pos;
    if (pos < src.size() && src[pos] == '-') pos++;
    size_t d = 0;
    while (pos < src.size() && std::isdigit((unsigned char)src[pos])) { pos++; d++; }
    if (!d) {
      pos = b;
      error("expected an integer");
    }
    return (int)std::strtol(src.substr(b, pos - b).c_str(), nullptr, 10);
  }
  std::string string_lit() {  // lib.rs:160-179
    expect('"');
    std::string out;
    for (;;) {
      if (pos >= src.size()) error("unterminated string");
      char c = src[pos++];
      if (c == '"') break;
      if (c == '\\') {
        if (pos >= src.size()) error("unterminated escape");
        char e = src[pos++];
        switch (e) {
          case '\\': out += '\\'; break;
          case '/': out += '/'; break;
          case '"': out += '"'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'n': out += '\n'; break;
          case 'r': out += '\r'; break;
          case 't': out += '\t'; break;
          default: pos--; error("unknown escape");
        }
      } else {
        out += c;
      }
    }
    return out;
  }
  void floats_n(float* out, int n) {  // parse_vec3 / parse_vec4, lib.rs:188-207
    for (int k = 0; k < n; ++k) {
      out[k] = number_float();
      sp();
    }
  }
  std::vector<float> bracket_floats() {  // bracket(float()), lib.rs:269-276
    std::vector<float> v;
    expect('[');
    sp();
    while (pos < src.size() && src[pos] != ']') {
      v.push_back(number_float());
      sp();
    }
    expect(']');
    return v;
  }
  void matrix16(float* out) {  // parse_transform, lib.rs:209-222
    expect('[');
    sp();
    floats_n(out, 16);
    expect(']');
  }

  Argument argument() {  // lib.rs:292-430
    Argument a;
    size_t start = pos;
    expect('"');
    static const struct { const char* kw; VType t; } kinds[] = {
        {"float", VType::Float}, {"bool", VType::Bool}, {"integer", VType::Integer}, {"string", VType::String},
        {"point", VType::Point}, {"normal", VType::Normal}, {"texture", VType::Texture},
        {"blackbody", VType::BlackBody}, {"rgb", VType::Rgb}, {"color", VType::Rgb}, {"spectrum", VType::Spectrum}};
    bool found = false;
    for (const auto& k : kinds)
      if (accept_word(k.kw)) {
        a.value.type = k.t;
        found = true;
        break;
      }
    if (!found) {
      pos = start;
      error("expected an argument type");
    }
    while (pos < src.size() && std::isspace((unsigned char)src[pos])) pos++;
    size_t nb = pos;
    while (pos < src.size() && (std::isalnum((unsigned char)src[pos]) || src[pos] == '_')) pos++;
    if (nb == pos) error("expected an argument name");
    a.name = src.substr(nb, pos - nb);
    expect('"');
    sp();
    Value& v = a.value;
    switch (v.type) {
      case VType::Float:
        if (pos < src.size() && src[pos] == '[') v.f = bracket_floats(); else v.f.push_back(number_float());
        break;
      case VType::Rgb:
        v.f = bracket_floats();
        if (v.f.size() != 3) error("length of rgb must be 3");
        break;
      case VType::BlackBody:
        v.f = bracket_floats();
        if (v.f.size() % 2) error("length of blackbody value must be a multiple of 2");
        break;
      case VType::Point: case VType::Normal:
        v.f = bracket_floats();
        if (v.f.size() % 3) error("length of point/normal value must be a multiple of 3");
        break;
      case VType::Integer:
        if (pos < src.size() && src[pos] == '[') {
          expect('[');
          sp();
          while (pos < src.size() && src[pos] != ']') {
            v.i.push_back(number_int());
            sp();
          }
          expect(']');
        } else {
          v.i.push_back(number_int());
        }
        break;
      case VType::Bool: {
        auto one = [&]() {
          std::string s = string_lit();
          if (s == "true") v.b.push_back(true);
          else if (s == "false") v.b.push_back(false);
          else error("expected \"true\" or \"false\"");
        };
        if (pos < src.size() && src[pos] == '[') {
          expect('[');
          sp();
          while (pos < src.size() && src[pos] != ']') {
            one();
            sp();
          }
          expect(']');
        } else {
          one();
        }
        break;
      }
      case VType::String: case VType::Texture:
        if (pos < src.size() && src[pos] == '[') {
          expect('[');
          sp();
          while (pos < src.size() && src[pos] != ']') {
            v.s.push_back(string_lit());
            sp();
          }
          expect(']');
        } else {
          v.s.push_back(string_lit());
        }
        break;
      case VType::Spectrum: v.s.push_back(string_lit()); break;
    }
    return a;
  }
  void arguments(Object& o) {
    for (;;) {
      sp();
      if (pos < src.size() && src[pos] == '"') o.args.push_back(argument()); else break;
    }
  }
  Object typed_object(const std::string& kind) {  // parse_scene_object / parse_world_object
    Object o;
    o.kind = kind;
    sp();
    o.t = string_lit();
    arguments(o);
    return o;
  }

  Worlds worlds(const char* terminator) {  // parse_worlds, lib.rs:532-577
    Worlds out;
    for (;;) {
      sp();
      if (eof()) {
        if (terminator) error(std::string("expected ") + terminator);
        break;
      }
      if (terminator && peek_word(terminator)) break;
      World w;
      // keyword order follows the reference's `choice` (longer spellings first where prefixes collide)
      if (accept_word("Texture")) {
        w.kind = World::Texture;
        sp(); w.name = string_lit();
        sp(); w.name2 = string_lit();
        sp(); w.obj.t = string_lit();
        w.obj.kind = "Texture";
        arguments(w.obj);
      } else if (accept_word("NamedMaterial")) {
        w.kind = World::NamedMaterial;
        sp(); w.name = string_lit();
      } else if (accept_word("LightSource")) { w.obj = typed_object("LightSource");
      } else if (accept_word("AreaLightSource")) { w.obj = typed_object("AreaLightSource");
      } else if (accept_word("MakeNamedMaterial")) { w.obj = typed_object("MakeNamedMaterial");
      } else if (accept_word("MakeNamedMedium")) { w.obj = typed_object("MakeNamedMedium");
      } else if (accept_word("Material")) { w.obj = typed_object("Material");
      } else if (accept_word("Shape")) { w.obj = typed_object("Shape");
      } else if (accept_word("ObjectInstance")) {
        w.kind = World::ObjectInstance;
        sp(); w.name = string_lit();
      } else if (accept_word("TransformBegin")) {  // Q10: an *Attribute* scope, lib.rs:561-566
        w.kind = World::Attribute;
        w.children = std::make_shared<Worlds>(worlds("TransformEnd"));
        accept_word("TransformEnd");
      } else if (accept_word("Transform")) {
        w.kind = World::Transform;
        sp(); matrix16(w.v);
      } else if (accept_word("ConcatTransform")) {
        w.kind = World::ConcatTransform;
        sp(); matrix16(w.v);
      } else if (accept_word("Translate")) {
        w.kind = World::Translate;
        sp(); floats_n(w.v, 3);
      } else if (accept_word("Scale")) {
        w.kind = World::Scale;
        sp(); floats_n(w.v, 3);
      } else if (accept_word("Rotate")) {
        w.kind = World::Rotate;
        sp(); floats_n(w.v, 4);
      } else if (accept_word("CoordSysTransform")) {
        w.kind = World::CoordSysTransform;
        sp(); w.name = string_lit();
      } else if (accept_word("MediumInterface")) {
        w.kind = World::MediumInterface;
        sp(); w.name = string_lit();
        sp(); w.name2 = string_lit();
      } else if (accept_word("ReverseOrientation")) {
        w.kind = World::ReverseOrientation;
      } else if (accept_word("AttributeBegin")) {
        w.kind = World::Attribute;
        w.children = std::make_shared<Worlds>(worlds("AttributeEnd"));
        accept_word("AttributeEnd");
      } else if (accept_word("ObjectBegin")) {
        w.kind = World::ObjectBeginEnd;
        sp(); w.name = string_lit();
        w.children = std::make_shared<Worlds>(worlds("ObjectEnd"));
        accept_word("ObjectEnd");
      } else {
        error("unknown world statement");
      }
      out.push_back(std::move(w));
    }
    return out;
  }

  std::vector<SceneStmt> scene() {  // parse_pbrt, lib.rs:440-464
    std::vector<SceneStmt> out;
    for (;;) {
      sp();
      if (eof()) break;
      SceneStmt s;
      if (accept_word("LookAt")) {
        s.kind = SceneStmt::LookAt;
        sp(); floats_n(s.v, 9);
      } else if (accept_word("Rotate")) {
        s.kind = SceneStmt::Rotate;
        sp(); floats_n(s.v, 4);
      } else if (accept_word("Scale")) {
        s.kind = SceneStmt::Scale;
        sp(); floats_n(s.v, 3);
      } else if (accept_word("Translate")) {
        s.kind = SceneStmt::Translate;
        sp(); floats_n(s.v, 3);
      } else if (accept_word("ConcatTransform")) {
        s.kind = SceneStmt::ConcatTransform;
        sp(); matrix16(s.v);
      } else if (accept_word("Transform")) {
        s.kind = SceneStmt::Transform;
        sp(); matrix16(s.v);
      } else if (accept_word("Camera")) { s.kind = SceneStmt::SceneObject; s.obj = typed_object("Camera");
      } else if (accept_word("Sampler")) { s.kind = SceneStmt::SceneObject; s.obj = typed_object("Sampler");
      } else if (accept_word("Integrator")) { s.kind = SceneStmt::SceneObject; s.obj = typed_object("Integrator");
      } else if (accept_word("PixelFilter")) { s.kind = SceneStmt::SceneObject; s.obj = typed_object("PixelFilter");
      } else if (accept_word("Film")) { s.kind = SceneStmt::SceneObject; s.obj = typed_object("Film");
      } else if (accept_word("WorldBegin")) {
        s.kind = SceneStmt::WorldBlock;
        s.world = worlds("WorldEnd");
        accept_word("WorldEnd");
      } else {
        error("unknown scene statement");
      }
      out.push_back(std::move(s));
    }
    return out;
  }
};

// ---- include expansion, include.rs:36-84 -----------------------------------------------------------------------
std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail(RENE_ERR_IO, "cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
std::string join_path(const std::string& dir, const std::string& file) {
  if (!file.empty() && file[0] == '/') return file;
  if (dir.empty()) return file;
  return dir.back() == '/' ? dir + file : dir + "/" + file;
}
std::string expand_include(const std::string& input, const std::string& dir, int depth = 0) {
  if (depth > 32) fail(RENE_ERR_PARSE, "Include nesting too deep");
  std::string result;
  size_t rest = 0;
  for (;;) {
    size_t mid = input.find("Include", rest);
    if (mid == std::string::npos) {
      result.append(input, rest, std::string::npos);
      return result;
    }
    result.append(input, rest, mid - rest);
    size_t p = mid + 7;
    // sp: spaces and comments (include.rs:15-29)
    for (;;) {
      while (p < input.size() && std::strchr(" \t\r\n", input[p])) p++;
      if (p < input.size() && input[p] == '#') {
        while (p < input.size() && input[p] != '\n') p++;
        continue;
      }
      break;
    }
    if (p < input.size() && input[p] == '"') {
      size_t q = p + 1;
      std::string path;
      bool ok = false;
      while (q < input.size()) {
        if (input[q] == '\\' && q + 1 < input.size()) { path += input[q + 1]; q += 2; continue; }
        if (input[q] == '"') { ok = true; break; }
        path += input[q++];
      }
      if (ok) {
        std::string buf = read_file(join_path(dir, path));
        result += expand_include(buf, dir, depth + 1);
        rest = q + 1;
        continue;
      }
    }
    result += "Include";  // not an include statement: keep the text (include.rs:66-70)
    rest = mid + 7;
  }
}

// ---- typed access with the reference's defaults (intermediate_scene.rs:263-610) -------------------------------
struct TexOrColor {
  bool is_tex = false;
  float c[3] = {0, 0, 0};
  std::string name;
};
[[noreturn]] void unsupported(const std::string& what) { fail(RENE_ERR_UNSUPPORTED, what + " is not supported by this loader"); }

bool get_float(const Object& o, const char* n, float& out) {
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type != VType::Float) fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  if (v->f.size() != 1) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
  out = v->f[0];
  return true;
}
bool get_int(const Object& o, const char* n, int& out) {
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type != VType::Integer) fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  if (v->i.size() != 1) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
  out = v->i[0];
  return true;
}
bool get_bool(const Object& o, const char* n, bool& out) {
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type != VType::Bool) fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  if (v->b.size() != 1) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
  out = v->b[0];
  return true;
}
bool get_str(const Object& o, const char* n, std::string& out) {
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type != VType::String) fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  if (v->s.size() != 1) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
  out = v->s[0];
  return true;
}
// ---- spectral colours -> RGB ---------------------------------------------------------------------------------------
thread_local std::string g_spd_base_dir;  // directory `spectrum` file names are relative to (set by load_impl)

// CIE 1931 2-degree matching functions at 360 .. 830 nm in 1 nm steps: the standard's own table (cie1931.inc; the one the
// reference tabulates, spectrum.rs:5-1467 -- tests/test_spectrum_reference.py holds the two against each other)
#include "cie1931.inc"
// from_sampled, spectrum.rs:1487-1506, statement for statement in f32: xyz += value(lambda_i) * matching_i, then
// scale = (lambda_last - lambda_first) / (CIE_Y_INTEGRAL * N), then XYZ -> linear sRGB
template <class F>
void spectrum_to_rgb(F value, float rgb[3]) {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  for (int i = 0; i < kCieSamples; ++i) {
    const float val = value(360.0f + (float)i);
    x += val * kCieX[i];
    y += val * kCieY[i];
    z += val * kCieZ[i];
  }
  const float scale = (830.0f - 360.0f) / (kCieYIntegral * (float)kCieSamples);
  x *= scale;
  y *= scale;
  z *= scale;
  rgb[0] = 3.240479f * x - 1.537150f * y - 0.498535f * z;
  rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
  rgb[2] = 0.055648f * x - 0.204043f * y + 1.057311f * z;
}
// "blackbody" [T scale ...]: sum of scale * RGB(peak-normalised Planck spectrum at T) (pbrt-v3's BlackbodyNormalized)
void blackbody_rgb(const std::vector<float>& pairs, float out[3]) {
  out[0] = out[1] = out[2] = 0.0f;
  for (size_t k = 0; k + 1 < pairs.size(); k += 2) {
    const double T = pairs[k], scale = pairs[k + 1];
    if (!(T > 0.0)) fail(RENE_ERR_INVALID_SCENE, "blackbody temperature must be positive");
    auto planck = [T](double l_nm) {
      const double c = 299792458.0, h = 6.62606957e-34, kb = 1.3806488e-23, l = l_nm * 1e-9;
      const double l5 = l * l * l * l * l;
      return 2.0 * h * c * c / (l5 * (std::exp(h * c / (l * kb * T)) - 1.0));
    };
    const double peak = planck(2.8977721e-3 / T * 1e9);  // Wien's displacement law
    float rgb[3];
    spectrum_to_rgb([&](float l) { return (float)(planck((double)l) / peak); }, rgb);
    for (int a = 0; a < 3; ++a) out[a] += (float)scale * rgb[a];
  }
}
// "spectrum" "file.spd": lines of `lambda value` (parse_spd, spectrum.rs:1508-1521), piecewise-linear in between and
// constant outside (interpolate, spectrum.rs:1468-1485)
void spd_file_rgb(const std::string& file, float out[3]) {
  const std::string path = join_path(g_spd_base_dir, file);
  const std::string text = read_file(path);
  std::vector<std::pair<float, float>> sp;
  const char* p = text.c_str();
  for (;;) {
    char* e = nullptr;
    const float l = std::strtof(p, &e);
    if (e == p) break;
    p = e;
    const float v = std::strtof(p, &e);
    if (e == p) fail(RENE_ERR_PARSE, "spectrum file " + path + ": a wavelength without a value");
    p = e;
    sp.push_back({l, v});
  }
  if (sp.size() < 2) fail(RENE_ERR_PARSE, "spectrum file " + path + ": fewer than two samples");
  std::sort(sp.begin(), sp.end());
  // interpolate, spectrum.rs:1468-1485, index for index: a wavelength that is not a sample is looked up at the binary
  // search's INSERTION point i and blended between samples i and i + 1 -- the segment after the one it lies in (t < 0: a
  // linear extrapolation backwards).  Kept (parity is with the reference's arithmetic); where i + 1 runs off the table
  // the reference panics, here the scene is refused.
  bool off_table = false;
  spectrum_to_rgb([&](float l) {
    if (l < sp.front().first) return sp.front().second;
    if (l > sp.back().first) return sp.back().second;
    size_t i = (size_t)(std::lower_bound(sp.begin(), sp.end(), l, [](const std::pair<float, float>& a, float b) { return a.first < b; }) - sp.begin());
    if (i + 1 >= sp.size()) {
      if (sp[i].first == l) return sp[i].second;  // an exact hit on the last sample: t = 0 never reads sample i + 1 ... in exact arithmetic
      off_table = true;
      return 0.0f;
    }
    const float t = (l - sp[i].first) / (sp[i + 1].first - sp[i].first);
    return (1.0f - t) * sp[i].second + t * sp[i + 1].second;
  }, out);
  if (off_table) fail(RENE_ERR_INVALID_SCENE, "spectrum file " + path + ": a CIE wavelength falls into its last segment (the reference indexes past the table there)");
}

bool get_rgb(const Object& o, const char* n, float* out) {  // intermediate_scene.rs:264-289
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type == VType::Rgb) {
    std::copy(v->f.begin(), v->f.begin() + 3, out);
    return true;
  }
  if (v->type == VType::BlackBody) {
    blackbody_rgb(v->f, out);
    return true;
  }
  if (v->type == VType::Spectrum) {
    if (v->s.empty()) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
    spd_file_rgb(v->s[0], out);
    return true;
  }
  fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
}
bool get_point(const Object& o, const char* n, float* out) {
  const Value* v = o.get(n);
  if (!v) return false;
  if (v->type != VType::Point) fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  if (v->f.size() != 3) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
  std::copy(v->f.begin(), v->f.end(), out);
  return true;
}
bool get_tex_or_color(const Object& o, const char* n, TexOrColor& out) {  // intermediate_scene.rs:291-324
  const Value* v = o.get(n);
  if (!v) return false;
  switch (v->type) {
    case VType::Float:
      if (v->f.size() != 1) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
      out.is_tex = false;
      out.c[0] = out.c[1] = out.c[2] = v->f[0];
      return true;
    case VType::Rgb:
      out.is_tex = false;
      std::copy(v->f.begin(), v->f.begin() + 3, out.c);
      return true;
    case VType::Texture:
      out.is_tex = true;
      out.name = v->s.empty() ? std::string() : v->s[0];
      return true;
    case VType::BlackBody:
      out.is_tex = false;
      blackbody_rgb(v->f, out.c);
      return true;
    case VType::Spectrum:
      if (v->s.empty()) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
      out.is_tex = false;
      spd_file_rgb(v->s[0], out.c);
      return true;
    default: fail(RENE_ERR_INVALID_SCENE, std::string("unmatched type on ") + n);
  }
}
TexOrColor color3(float r, float g, float b) {
  TexOrColor t;
  t.c[0] = r; t.c[1] = g; t.c[2] = b;
  return t;
}
TexOrColor tex_default(const Object& o, const char* n, float r, float g, float b) {
  TexOrColor t;
  if (get_tex_or_color(o, n, t)) return t;
  return color3(r, g, b);
}

}  // namespace

// =================================================================================================
// rene_scene: owns every table the desc points at
// =================================================================================================
struct rene_scene {
  rene_scene_desc desc{};
  std::string film_filename = "out.png";  // Film default, intermediate_scene.rs:162-170
  std::vector<rene_instance> instances;
  std::vector<std::vector<rene_vertex>> mesh_vertices;
  std::vector<std::vector<uint32_t>> mesh_indices;
  std::vector<rene_mesh> meshes;
  std::vector<rene_material> materials;
  std::vector<rene_texture> textures;
  std::vector<rene_area_light> area_lights;
  std::vector<rene_light> lights;
  std::vector<rene_medium> mediums;
  std::vector<std::vector<float>> image_data;
  std::vector<rene_image> images;
};

namespace {

struct WorldState {  // scene.rs:66-78
  uint32_t material = 0;
  uint32_t area_light = 0;
  uint32_t medium_interior = 0, medium_exterior = 0;  // current_medium_index, None == (0, 0)
  M4 ctm = M4::identity();
  std::map<std::string, uint32_t> textures, materials, mediums;
  std::map<std::string, std::vector<rene_instance>> objects;
  std::map<std::string, M4> coord_system;
};

struct Builder {
  rene_scene& sc;
  std::string base_dir;
  M4 world_to_camera = M4::identity();
  float fov = 0.5f * 3.14159265358979323846f;  // "90 degree", scene.rs:106-107
  uint32_t xres = 640, yres = 480;
  uint32_t integrator = RENE_INTEGRATOR_PATH;
  float bg_color[4] = {0, 0, 0, 0};
  uint32_t bg_texture = 0;
  M4 bg_matrix = M4::identity();

  Builder(rene_scene& s, std::string dir) : sc(s), base_dir(std::move(dir)) {
    // index-0 sentinels, scene.rs:109-116
    rene_material none{};
    none.type = RENE_MATERIAL_NONE;
    sc.materials.push_back(none);
    rene_area_light nul{};
    nul.type = RENE_AREA_LIGHT_NULL;
    sc.area_lights.push_back(nul);
    rene_medium vacuum{};
    vacuum.type = RENE_MEDIUM_VACUUM;
    sc.mediums.push_back(vacuum);
    solid(1.0f, 1.0f, 1.0f);
  }

  uint32_t solid(float r, float g, float b) {  // EnumTexture::new_solid
    rene_texture t{};
    t.type = RENE_TEXTURE_SOLID;
    t.v0[0] = r; t.v0[1] = g; t.v0[2] = b;
    sc.textures.push_back(t);
    return (uint32_t)sc.textures.size() - 1;
  }
  uint32_t texture(const TexOrColor& t, const WorldState& st) {  // Scene::texture, scene.rs:81-98
    if (!t.is_tex) return solid(t.c[0], t.c[1], t.c[2]);
    auto it = st.textures.find(t.name);
    if (it == st.textures.end()) fail(RENE_ERR_INVALID_SCENE, "Not Found Texture: " + t.name);
    return it->second;
  }

  // ---- LDR images: load_image's fall-through branch (intermediate_scene.rs:657-675) decodes with the
  // `image` crate (0.24.1, absent from /root/reference) and stores, per pixel of DynamicImage::pixels()
  // (RGBA8), inverse_gamma_correct(c / 255) for r, g, b (intermediate_scene.rs:616-622) and a / 255.
  // Restated here for PNG (RFC 2083: zlib stream + the five scanline filters, Adam7 interlacing, 1 to 16 bits per
  // sample, tRNS); grey -> r = g = b, missing alpha -> 255, like the crate's to-RGBA8 conversion.
  static float inverse_gamma_correct(float v) {
    return v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
  }
  uint32_t load_png(const std::string& path, const std::string& file) {
    std::string data = read_file(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto bad = [&](const char* why) { fail(RENE_ERR_IO, std::string("PNG decode error (") + why + "): " + file); };
    if (data.size() < 8 || std::memcmp(data.data(), sig, 8) != 0) bad("signature");
    auto be32 = [&](size_t p) {
      return ((uint32_t)(unsigned char)data[p] << 24) | ((uint32_t)(unsigned char)data[p + 1] << 16) |
             ((uint32_t)(unsigned char)data[p + 2] << 8) | (uint32_t)(unsigned char)data[p + 3];
    };
    uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::string idat;
    std::vector<unsigned char> plte, trns;
    bool end = false;
    for (size_t p = 8; p + 12 <= data.size() && !end;) {
      uint32_t len = be32(p);
      std::string type = data.substr(p + 4, 4);
      if (p + 12 + (size_t)len > data.size()) bad("truncated chunk");
      const char* body = data.data() + p + 8;
      if (type == "IHDR") {
        if (len != 13) bad("IHDR");
        w = be32(p + 8); h = be32(p + 12);
        depth = (unsigned char)body[8]; ctype = (unsigned char)body[9]; interlace = (unsigned char)body[12];
      } else if (type == "PLTE") plte.assign(body, body + len);
      else if (type == "tRNS") trns.assign(body, body + len);
      else if (type == "IDAT") idat.append(body, len);
      else if (type == "IEND") end = true;
      p += 12 + (size_t)len;
    }
    if (!w || !h || idat.empty()) bad("no image data");
    uint32_t ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) bad("colour type");
    const bool depth_ok = depth == 8 || (depth == 16 && ctype != 3) || ((depth == 1 || depth == 2 || depth == 4) && (ctype == 0 || ctype == 3));
    if (!depth_ok) bad("bit depth");
    if (interlace > 1) bad("interlace method");
    if ((uint64_t)w * h > (1ull << 28)) bad("too large");
    // one pass, or the seven of Adam7 (RFC 2083 section 2.6): first pixel and step of each reduced image
    static const uint32_t ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Pass { uint32_t x0, y0, dx, dy, pw, ph; size_t stride; };
    std::vector<Pass> passes;
    size_t raw_bytes = 0;
    for (int k = 0; k < (interlace ? 7 : 1); ++k) {
      Pass q{interlace ? ax0[k] : 0u, interlace ? ay0[k] : 0u, interlace ? adx[k] : 1u, interlace ? ady[k] : 1u, 0, 0, 0};
      q.pw = w > q.x0 ? (w - q.x0 + q.dx - 1) / q.dx : 0;
      q.ph = h > q.y0 ? (h - q.y0 + q.dy - 1) / q.dy : 0;
      if (!q.pw || !q.ph) continue;
      q.stride = ((size_t)q.pw * ch * depth + 7) / 8;  // bytes per scanline of the pass
      raw_bytes += (q.stride + 1) * q.ph;
      passes.push_back(q);
    }
    const size_t bpp = std::max<size_t>(1, (size_t)ch * depth / 8);  // filter distance, RFC 2083 section 6.2
    std::vector<unsigned char> raw(raw_bytes);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, reinterpret_cast<const Bytef*>(idat.data()), (uLong)idat.size()) != Z_OK ||
        out_len != raw.size())
      bad("zlib stream");
    // samples at their original depth, one uint16 per channel, in image order
    std::vector<uint16_t> smp((size_t)w * h * ch);
    size_t at = 0;
    std::vector<unsigned char> line, prev;
    for (const Pass& q : passes) {
      line.assign(q.stride, 0);
      prev.assign(q.stride, 0);
      for (uint32_t py = 0; py < q.ph; ++py) {  // un-filter, RFC 2083 section 6
        const unsigned char* in = raw.data() + at;
        at += q.stride + 1;
        const unsigned filter = in[0];
        if (filter > 4) bad("filter type");
        for (size_t i = 0; i < q.stride; ++i) {
          int a = i >= bpp ? line[i - bpp] : 0, b = py ? prev[i] : 0, c = (py && i >= bpp) ? prev[i - bpp] : 0, pred = 0;
          switch (filter) {
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) / 2; break;
            case 4: {
              int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
              pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
              break;
            }
            default: break;
          }
          line[i] = (unsigned char)(in[1 + i] + pred);
        }
        const uint32_t y = q.y0 + py * q.dy;
        for (uint32_t pxi = 0; pxi < q.pw; ++pxi) {
          uint16_t* o = &smp[((size_t)y * w + q.x0 + (size_t)pxi * q.dx) * ch];
          for (uint32_t c2 = 0; c2 < ch; ++c2) {
            const size_t k = (size_t)pxi * ch + c2;
            if (depth == 8) o[c2] = line[k];
            else if (depth == 16) o[c2] = (uint16_t)((line[2 * k] << 8) | line[2 * k + 1]);
            else {  // 1 / 2 / 4-bit samples, most significant first
              const size_t bit = k * depth;
              o[c2] = (line[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1u);
            }
          }
        }
        prev.swap(line);
      }
    }
    // to RGBA8 like the crate's DynamicImage::to_rgba8: grey levels scale to 0..255, 16-bit samples round to
    // (v + 128) / 257, a tRNS colour key (grey / RGB images) makes the matching pixels transparent
    const unsigned maxv = (1u << depth) - 1u;
    auto to8 = [&](unsigned v) -> unsigned { return depth == 16 ? (v + 128u) / 257u : depth == 8 ? v : v * 255u / maxv; };
    auto be16 = [&](size_t k) -> unsigned { return k + 1 < trns.size() ? (unsigned)((trns[k] << 8) | trns[k + 1]) : 0x10000u; };
    std::vector<float> rgba((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
      const uint16_t* s = &smp[i * ch];
      unsigned r, g, b, a = 255;
      if (ctype == 3) {
        if ((size_t)s[0] * 3 + 2 >= plte.size()) bad("palette index");
        r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
        if (s[0] < trns.size()) a = trns[s[0]];
      } else if (ch <= 2) {
        r = g = b = to8(s[0]);
        if (ch == 2) a = to8(s[1]);
        else if (trns.size() >= 2 && s[0] == be16(0)) a = 0;
      } else {
        r = to8(s[0]); g = to8(s[1]); b = to8(s[2]);
        if (ch == 4) a = to8(s[3]);
        else if (trns.size() >= 6 && s[0] == be16(0) && s[1] == be16(2) && s[2] == be16(4)) a = 0;
      }
      rgba[i * 4 + 0] = inverse_gamma_correct((float)r / 255.0f);
      rgba[i * 4 + 1] = inverse_gamma_correct((float)g / 255.0f);
      rgba[i * 4 + 2] = inverse_gamma_correct((float)b / 255.0f);
      rgba[i * 4 + 3] = (float)a / 255.0f;
    }
    return push_image(std::move(rgba), w, h);
  }
  uint32_t push_image(std::vector<float>&& rgba, uint32_t w, uint32_t h) {
    sc.image_data.push_back(std::move(rgba));
    rene_image im{};
    im.rgba = sc.image_data.back().data();
    im.width = w;
    im.height = h;
    sc.images.push_back(im);
    return (uint32_t)sc.images.size() - 1;
  }

  // ---- HDR images: PFM (pfm_parser.rs:10-61; load_image intermediate_scene.rs:631-677) ----
  uint32_t load_image(const std::string& file) {
    std::string path = join_path(base_dir, file);
    size_t dot = path.rfind('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "png") return load_png(path, file);
    if (ext == "tga" || ext == "bmp" || ext == "jpg" || ext == "jpeg") {  // the `image` crate branch, intermediate_scene.rs:657-675
      std::string bytes = read_file(path), why;
      uint32_t w = 0, h = 0;
      std::vector<unsigned char> px;
      const bool jpg = ext == "jpg" || ext == "jpeg";
      if (!(ext == "tga" ? rene::decode_tga(bytes, w, h, px, why) : jpg ? rene::decode_jpeg(bytes, w, h, px, why) : rene::decode_bmp(bytes, w, h, px, why)))
        fail(RENE_ERR_IO, (ext == "tga" ? "TGA" : jpg ? "JPEG" : "BMP") + std::string(" decode error (") + why + "): " + file);
      std::vector<float> rgba((size_t)w * h * 4);
      for (size_t i = 0; i < (size_t)w * h; ++i) {
        for (int k = 0; k < 3; ++k) rgba[i * 4 + k] = inverse_gamma_correct((float)px[i * 4 + k] / 255.0f);
        rgba[i * 4 + 3] = (float)px[i * 4 + 3] / 255.0f;
      }
      return push_image(std::move(rgba), w, h);
    }
    if (ext == "exr") {  // exr::prelude::read_first_rgba_layer_from_file, intermediate_scene.rs:641-656
      std::string bytes = read_file(path), why;
      uint32_t w = 0, h = 0;
      std::vector<float> rgba;
      if (!rene::decode_exr(bytes, w, h, rgba, why)) fail(RENE_ERR_IO, "EXR decode error (" + why + "): " + file);
      return push_image(std::move(rgba), w, h);
    }
    if (ext != "pfm") unsupported("image format ." + ext + " (" + file + ")");
    std::string data = read_file(path);
    // header: "PF\n<w> <h>\n<scale>\n" then rows bottom-to-top; negative scale = little endian
    size_t p = 0;
    auto token = [&]() {
      while (p < data.size() && std::isspace((unsigned char)data[p])) p++;
      size_t b = p;
      while (p < data.size() && !std::isspace((unsigned char)data[p])) p++;
      return data.substr(b, p - b);
    };
    std::string magic = token();
    if (magic != "PF") fail(RENE_ERR_IO, "PFM decode error: " + file);
    uint32_t w = (uint32_t)std::atoi(token().c_str()), h = (uint32_t)std::atoi(token().c_str());
    float scale = std::strtof(token().c_str(), nullptr);
    p++;  // single whitespace after the scale
    if (!w || !h || data.size() < p + (size_t)w * h * 12) fail(RENE_ERR_IO, "PFM decode error: " + file);
    bool little = scale < 0.0f;
    std::vector<float> rgba((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y)
      for (uint32_t x = 0; x < w; ++x)
        for (int c = 0; c < 3; ++c) {
          unsigned char b[4];
          std::memcpy(b, &data[p + (((size_t)y * w + x) * 3 + c) * 4], 4);
          if (!little) std::reverse(b, b + 4);
          float v;
          std::memcpy(&v, b, 4);
          // PFM stores the bottom row first; Image rows are top first (pfm_parser.rs:41-55)
          rgba[(((size_t)(h - 1 - y)) * w + x) * 4 + c] = v;
        }
    for (size_t i = 0; i < (size_t)w * h; ++i) rgba[i * 4 + 3] = 1.0f;
    sc.image_data.push_back(std::move(rgba));
    rene_image im{};
    im.rgba = sc.image_data.back().data();
    im.width = w;
    im.height = h;
    sc.images.push_back(im);
    return (uint32_t)sc.images.size() - 1;
  }

  // ---- materials: Object::get_material (intermediate_scene.rs:422-594) + Scene::material (scene.rs:170-257)
  void roughness_pair(const Object& o, float dflt, TexOrColor& ru, TexOrColor& rv) {
    TexOrColor r;
    if (get_tex_or_color(o, "roughness", r)) {
      ru = rv = r;
      return;
    }
    TexOrColor a, b;
    if (get_tex_or_color(o, "uroughness", a) && get_tex_or_color(o, "vroughness", b)) {
      ru = a;
      rv = b;
      return;
    }
    ru = rv = color3(dflt, dflt, dflt);
  }
  rene_material material(const Object& o, const std::string& type, const WorldState& st) {
    rene_material m{};
    bool remap = true;
    if (type == "none" || type == "") {
      m.type = RENE_MATERIAL_NONE;
    } else if (type == "matte") {
      m.type = RENE_MATERIAL_MATTE;
      m.u0[0] = texture(tex_default(o, "Kd", 0.5f, 0.5f, 0.5f), st);
    } else if (type == "glass") {
      m.type = RENE_MATERIAL_GLASS;
      float idx = 1.5f;
      get_float(o, "index", idx);
      m.v0[0] = idx;
    } else if (type == "substrate") {
      m.type = RENE_MATERIAL_SUBSTRATE;
      TexOrColor kd = tex_default(o, "Kd", 0.5f, 0.5f, 0.5f), ks = tex_default(o, "Ks", 0.5f, 0.5f, 0.5f), ru, rv;
      roughness_pair(o, 0.0f, ru, rv);
      get_bool(o, "remaproughness", remap);
      m.u0[0] = texture(kd, st);
      m.u0[1] = texture(ks, st);
      m.u0[2] = texture(ru, st);
      m.u0[3] = texture(rv, st);
      m.u1[0] = remap ? 1 : 0;
    } else if (type == "metal") {
      m.type = RENE_MATERIAL_METAL;
      TexOrColor eta = tex_default(o, "eta", 0.19999069f, 0.9220846f, 1.0998759f);
      TexOrColor k = tex_default(o, "k", 3.9046354f, 2.4476333f, 2.1376526f), ru, rv;
      roughness_pair(o, 0.01f, ru, rv);
      get_bool(o, "remaproughness", remap);
      m.u0[0] = texture(eta, st);
      m.u0[1] = texture(k, st);
      m.u0[2] = texture(ru, st);
      m.u0[3] = texture(rv, st);
      m.u1[0] = remap ? 1 : 0;
    } else if (type == "mirror") {
      m.type = RENE_MATERIAL_MIRROR;
      m.u0[0] = texture(tex_default(o, "Kd", 0.9f, 0.9f, 0.9f), st);  // sic: Kd, intermediate_scene.rs:516-521
    } else if (type == "uber") {
      m.type = RENE_MATERIAL_UBER;
      TexOrColor kd = tex_default(o, "Kd", 0.25f, 0.25f, 0.25f), ks = tex_default(o, "Ks", 0.25f, 0.25f, 0.25f);
      TexOrColor kr = tex_default(o, "Kr", 0, 0, 0), kt = tex_default(o, "Kt", 0, 0, 0), ru, rv;
      roughness_pair(o, 0.1f, ru, rv);
      float eta = 1.5f;
      get_float(o, "eta", eta);
      TexOrColor op = tex_default(o, "opacity", 1, 1, 1);
      get_bool(o, "remaproughness", remap);
      // texture creation order, scene.rs:232-241
      uint32_t i_kd = texture(kd, st), i_ks = texture(ks, st), i_kr = texture(kr, st), i_kt = texture(kt, st);
      uint32_t i_ru = texture(ru, st), i_rv = texture(rv, st), i_op = texture(op, st);
      m.u0[0] = i_kd; m.u0[1] = i_ks; m.u0[2] = i_kr; m.u0[3] = i_kt;
      m.u1[0] = i_op; m.u1[1] = remap ? 1 : 0; m.u1[2] = i_ru; m.u1[3] = i_rv;
      m.v0[0] = eta;
    } else if (type == "plastic") {
      m.type = RENE_MATERIAL_PLASTIC;
      TexOrColor kd = tex_default(o, "Kd", 0.25f, 0.25f, 0.25f), ks = tex_default(o, "Ks", 0.25f, 0.25f, 0.25f);
      TexOrColor r = tex_default(o, "roughness", 0.1f, 0.1f, 0.1f);
      get_bool(o, "remaproughness", remap);
      m.u0[0] = texture(kd, st);
      m.u0[1] = texture(ks, st);
      m.u0[3] = texture(r, st);
      m.u0[2] = remap ? 1 : 0;  // stored in u0.z, read from u1.z (Q8), material.rs:650-676
    } else {
      fail(RENE_ERR_INVALID_SCENE, "Invalid Material type " + type);
    }
    return m;
  }

  // ---- PLY (load_ply, intermediate_scene.rs:679-752): ascii and binary_little_endian ----
  void load_ply(const std::string& file, std::vector<rene_vertex>& verts, std::vector<uint32_t>& idx) {
    std::string data = read_file(join_path(base_dir, file));
    size_t p = 0;
    auto line = [&]() {
      size_t e = data.find('\n', p);
      if (e == std::string::npos) e = data.size();
      std::string l = data.substr(p, e - p);
      p = std::min(data.size(), e + 1);
      if (!l.empty() && l.back() == '\r') l.pop_back();
      return l;
    };
    if (line() != "ply") fail(RENE_ERR_IO, "Ply error: " + file);
    enum Fmt { Ascii, LE, BE } fmt = Ascii;
    struct Prop { std::string name, type, count_type; bool list = false; };
    struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
    std::vector<Elem> elems;
    for (;;) {
      if (p >= data.size()) fail(RENE_ERR_IO, "Ply error: unterminated header in " + file);
      std::istringstream ls(line());
      std::string kw;
      ls >> kw;
      if (kw == "format") {
        std::string f;
        ls >> f;
        fmt = f == "ascii" ? Ascii : (f == "binary_little_endian" ? LE : BE);
      } else if (kw == "element") {
        Elem e;
        ls >> e.name >> e.count;
        elems.push_back(e);
      } else if (kw == "property") {
        if (elems.empty()) fail(RENE_ERR_IO, "Ply error: property before element");
        Prop pr;
        std::string t;
        ls >> t;
        if (t == "list") {
          pr.list = true;
          ls >> pr.count_type >> pr.type >> pr.name;
        } else {
          pr.type = t;
          ls >> pr.name;
        }
        elems.back().props.push_back(pr);
      } else if (kw == "end_header") {
        break;
      }
    }
    if (fmt == BE) unsupported("binary_big_endian PLY");
    auto type_size = [](const std::string& t) -> size_t {
      if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
      if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
      if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
      if (t == "double" || t == "float64") return 8;
      fail(RENE_ERR_IO, "Ply error: unknown type " + t);
    };
    std::istringstream ascii;
    if (fmt == Ascii) ascii.str(data.substr(p));
    auto read_num = [&](const std::string& t) -> double {
      if (fmt == Ascii) {
        double v;
        if (!(ascii >> v)) fail(RENE_ERR_IO, "Ply error: truncated data");
        return v;
      }
      size_t n = type_size(t);
      if (p + n > data.size()) fail(RENE_ERR_IO, "Ply error: truncated data");
      const char* s = &data[p];
      p += n;
      if (t == "float" || t == "float32") { float v; std::memcpy(&v, s, 4); return v; }
      if (t == "double" || t == "float64") { double v; std::memcpy(&v, s, 8); return v; }
      if (t == "uchar" || t == "uint8") { return (unsigned char)s[0]; }
      if (t == "char" || t == "int8") { return (signed char)s[0]; }
      if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, s, 2); return v; }
      if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, s, 2); return v; }
      if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, s, 4); return v; }
      int32_t v;
      std::memcpy(&v, s, 4);
      return v;
    };
    bool have_vertex = false, have_face = false;
    for (const Elem& e : elems) {
      if (e.name == "vertex") {
        have_vertex = true;
        verts.resize(e.count);
        for (size_t i = 0; i < e.count; ++i) {
          rene_vertex v{};
          bool hx = false, hy = false, hz = false, hn[3] = {false, false, false}, hu = false, hv = false;
          float n[3] = {0, 0, 0}, uv[2] = {0, 0};
          for (const Prop& pr : e.props) {
            if (pr.list) {
              size_t c = (size_t)read_num(pr.count_type);
              for (size_t k = 0; k < c; ++k) read_num(pr.type);
              continue;
            }
            float val = (float)read_num(pr.type);
            if (pr.name == "x") { v.position[0] = val; hx = true; }
            else if (pr.name == "y") { v.position[1] = val; hy = true; }
            else if (pr.name == "z") { v.position[2] = val; hz = true; }
            else if (pr.name == "nx") { n[0] = val; hn[0] = true; }
            else if (pr.name == "ny") { n[1] = val; hn[1] = true; }
            else if (pr.name == "nz") { n[2] = val; hn[2] = true; }
            else if (pr.name == "u") { uv[0] = val; hu = true; }
            else if (pr.name == "v") { uv[1] = val; hv = true; }
          }
          if (!hx || !hy || !hz) fail(RENE_ERR_IO, "Ply error: vertex without x/y/z");
          if (hn[0] && hn[1] && hn[2]) std::copy(n, n + 3, v.normal);
          if (hu && hv) std::copy(uv, uv + 2, v.uv);
          verts[i] = v;
        }
      } else if (e.name == "face") {
        have_face = true;
        for (size_t i = 0; i < e.count; ++i) {
          std::vector<uint32_t> face;
          for (const Prop& pr : e.props) {
            if (pr.list) {
              size_t c = (size_t)read_num(pr.count_type);
              std::vector<uint32_t> tmp(c);
              for (size_t k = 0; k < c; ++k) tmp[k] = (uint32_t)(int64_t)read_num(pr.type);
              if (pr.name == "vertex_indices") face = tmp;
            } else {
              read_num(pr.type);
            }
          }
          for (uint32_t ix : face)
            if (ix >= verts.size()) fail(RENE_ERR_IO, "Ply error: face index out of range");
          if (face.size() == 3) {
            idx.insert(idx.end(), face.begin(), face.end());
          } else if (face.size() == 4) {  // quad split, intermediate_scene.rs:741-744
            const uint32_t q[6] = {face[0], face[1], face[2], face[0], face[2], face[3]};
            idx.insert(idx.end(), q, q + 6);
          } else {
            fail(RENE_ERR_IO, "Ply error: unsupported face length");
          }
        }
      } else {
        for (size_t i = 0; i < e.count; ++i)
          for (const Prop& pr : e.props) {
            if (pr.list) {
              size_t c = (size_t)read_num(pr.count_type);
              for (size_t k = 0; k < c; ++k) read_num(pr.type);
            } else {
              read_num(pr.type);
            }
          }
      }
    }
    if (!have_vertex || !have_face) fail(RENE_ERR_IO, "Ply error: missing vertex or face element");
  }

  uint32_t push_mesh(std::vector<rene_vertex>&& v, std::vector<uint32_t>&& i) {
    sc.mesh_vertices.push_back(std::move(v));
    sc.mesh_indices.push_back(std::move(i));
    return (uint32_t)sc.mesh_vertices.size() - 1;
  }

  // ---- append_world, scene.rs:259-460 ----
  void append_world(WorldState& st, const Worlds& ws) {
    for (const World& w : ws) {
      switch (w.kind) {
        case World::ReverseOrientation: break;  // "not yet implemented", scene.rs:266-268
        case World::Attribute: {
          WorldState tmp = st;
          append_world(tmp, *w.children);
          st.objects = tmp.objects;  // scene.rs:269-273
          break;
        }
        case World::ObjectBeginEnd: {  // scene.rs:279-288
          size_t cur = sc.instances.size();
          append_world(st, *w.children);
          std::vector<rene_instance> objs(sc.instances.begin() + cur, sc.instances.end());
          sc.instances.resize(cur);
          st.objects[w.name] = objs;
          break;
        }
        case World::ObjectInstance: {  // scene.rs:289-299
          auto it = st.objects.find(w.name);
          if (it == st.objects.end()) fail(RENE_ERR_INVALID_SCENE, "Not Object: " + w.name);
          for (rene_instance inst : it->second) {
            // tlas.matrix * Affine3A::from_mat4(current_matrix)   (object matrix on the LEFT, Q10)
            M4 om = M4::identity();
            for (int c = 0; c < 4; ++c)
              for (int r = 0; r < 3; ++r) om.at(r, c) = inst.matrix[c * 3 + r];
            M4 ctm = st.ctm;
            ctm.at(3, 0) = ctm.at(3, 1) = ctm.at(3, 2) = 0.0;  // from_mat4 drops the last row
            ctm.at(3, 3) = 1.0;
            affine12(round32(mul(om, ctm)), inst.matrix);
            sc.instances.push_back(inst);
          }
          break;
        }
        case World::ConcatTransform: {
          M4 m;
          for (int i = 0; i < 16; ++i) m.m[i] = w.v[i];
          st.ctm = round32(mul(st.ctm, m));
          break;
        }
        case World::Translate: st.ctm = round32(mul(st.ctm, from_translation(w.v))); break;
        case World::Scale: st.ctm = round32(mul(st.ctm, from_scale(w.v))); break;
        case World::Rotate: st.ctm = round32(mul(st.ctm, from_axis_angle(w.v + 1, w.v[0]))); break;
        case World::Transform:
          for (int i = 0; i < 16; ++i) st.ctm.m[i] = w.v[i];
          break;
        case World::NamedMaterial: {
          auto it = st.materials.find(w.name);
          if (it == st.materials.end()) fail(RENE_ERR_INVALID_SCENE, "Unknown Material " + w.name);
          st.material = it->second;
          break;
        }
        case World::CoordSysTransform: {
          auto it = st.coord_system.find(w.name);
          if (it == st.coord_system.end()) fail(RENE_ERR_INVALID_SCENE, "Not Found Coord system: " + w.name);
          st.ctm = it->second;
          break;
        }
        case World::MediumInterface: {  // scene.rs:320-341; "" names the vacuum at index 0
          auto find = [&](const std::string& n) -> uint32_t {
            if (n.empty()) return 0u;
            auto it = st.mediums.find(n);
            if (it == st.mediums.end()) fail(RENE_ERR_INVALID_SCENE, "Unknown Medium " + n);
            return it->second;
          };
          st.medium_interior = find(w.name);
          st.medium_exterior = find(w.name2);
          break;
        }
        case World::Texture: {  // scene.rs:342-365; intermediate_scene.rs:769-833
          const Object& o = w.obj;
          rene_texture t{};
          if (o.t == "constant") {
            float v = 1.0f, rgb[3] = {1, 1, 1};
            const Value* val = o.get("value");
            if (val && val->type == VType::Float && val->f.size() == 1) { v = val->f[0]; rgb[0] = rgb[1] = rgb[2] = v; }
            else if (val && val->type == VType::Rgb) std::copy(val->f.begin(), val->f.begin() + 3, rgb);
            t.type = RENE_TEXTURE_SOLID;
            std::copy(rgb, rgb + 3, t.v0);
          } else if (o.t == "scale") {
            uint32_t a = texture(tex_default(o, "tex1", 1, 1, 1), st);
            uint32_t b = texture(tex_default(o, "tex2", 1, 1, 1), st);
            t.type = RENE_TEXTURE_SCALE;
            t.u0[0] = a;
            t.u0[1] = b;
          } else if (o.t == "checkerboard") {
            uint32_t a = texture(tex_default(o, "tex1", 0, 0, 0), st);
            uint32_t b = texture(tex_default(o, "tex2", 1, 1, 1), st);
            float us = 2.0f, vs = 2.0f;
            get_float(o, "uscale", us);
            get_float(o, "vscale", vs);
            t.type = RENE_TEXTURE_CHECKERBOARD;
            t.u0[0] = a;
            t.u0[1] = b;
            t.v0[0] = us;
            t.v0[1] = vs;
          } else if (o.t == "imagemap") {
            std::string fn;
            if (!get_str(o, "filename", fn)) fail(RENE_ERR_INVALID_SCENE, "Argument not found filename");
            t.type = RENE_TEXTURE_IMAGEMAP;
            t.u0[0] = load_image(fn);
          } else {
            fail(RENE_ERR_INVALID_SCENE, "Invalid Texture type " + o.t);
          }
          sc.textures.push_back(t);
          st.textures[w.name] = (uint32_t)sc.textures.size() - 1;
          break;
        }
        case World::Obj: world_object(st, w.obj); break;
      }
    }
  }

  void world_object(WorldState& st, const Object& o) {
    if (o.kind == "LightSource") {  // intermediate_scene.rs:835-869; scene.rs:367-388
      if (o.t == "infinite") {
        float L[3] = {1, 1, 1};
        get_rgb(o, "L", L);
        std::copy(L, L + 3, bg_color);
        bg_color[3] = 0.0f;
        std::string map;
        if (get_str(o, "mapname", map)) {
          uint32_t img = load_image(map);
          rene_texture t{};
          t.type = RENE_TEXTURE_IMAGEMAP;
          t.u0[0] = img;
          sc.textures.push_back(t);
          bg_matrix = inverse(st.ctm);
          bg_texture = (uint32_t)sc.textures.size() - 1;
        }
      } else if (o.t == "distant") {
        float from[3] = {0, 0, 0}, to[3] = {0, 0, 1}, L[3] = {1, 1, 1};
        get_point(o, "from", from);
        get_point(o, "to", to);
        get_rgb(o, "L", L);
        float d[3] = {from[0] - to[0], from[1] - to[1], from[2] - to[2]};  // light.rs:43-50
        float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        rene_light l{};
        l.type = RENE_LIGHT_DISTANT;
        for (int k = 0; k < 3; ++k) {
          l.v0[k] = d[k] / len;
          l.v1[k] = L[k];
        }
        sc.lights.push_back(l);
      } else {
        fail(RENE_ERR_INVALID_SCENE, "Invalid LightSource type " + o.t);
      }
    } else if (o.kind == "AreaLightSource") {  // intermediate_scene.rs:870-878
      if (o.t != "diffuse" && o.t != "area") fail(RENE_ERR_INVALID_SCENE, "Invalid AreaLightSource type " + o.t);
      float L[3];
      if (!get_rgb(o, "L", L)) fail(RENE_ERR_INVALID_SCENE, "Argument not found L");
      rene_area_light a{};
      a.type = RENE_AREA_LIGHT_DIFFUSE;
      std::copy(L, L + 3, a.v0);
      st.area_light = (uint32_t)sc.area_lights.size();
      sc.area_lights.push_back(a);
    } else if (o.kind == "Material") {
      rene_material m = material(o, o.t, st);
      st.material = (uint32_t)sc.materials.size();
      sc.materials.push_back(m);
    } else if (o.kind == "MakeNamedMaterial") {  // intermediate_scene.rs:882-892; scene.rs:394-399
      std::string type;
      if (!get_str(o, "type", type)) fail(RENE_ERR_INVALID_SCENE, "Argument not found type");
      rene_material m = material(o, type, st);
      st.materials[o.t] = (uint32_t)sc.materials.size();
      st.material = (uint32_t)sc.materials.size();
      sc.materials.push_back(m);
    } else if (o.kind == "MakeNamedMedium") {  // intermediate_scene.rs:893-914; scene.rs:405-416
      // every named medium is homogeneous whatever its "type" says; defaults are the reference's
      rene_medium m{};
      m.type = RENE_MEDIUM_HOMOGENEOUS;
      float sigma_a[3] = {0.0011f, 0.0024f, 0.014f}, sigma_s[3] = {2.55f, 3.21f, 3.77f}, g = 0.0f;
      get_rgb(o, "sigma_a", sigma_a);
      get_rgb(o, "sigma_s", sigma_s);
      get_float(o, "g", g);
      std::copy(sigma_a, sigma_a + 3, m.v0);
      m.v0[3] = g;
      std::copy(sigma_s, sigma_s + 3, m.v1);
      st.mediums[o.t] = (uint32_t)sc.mediums.size();
      sc.mediums.push_back(m);
    } else if (o.kind == "Shape") {
      rene_instance inst{};
      inst.material_index = st.material;
      inst.area_light_index = st.area_light;
      inst.interior_medium_index = st.medium_interior;
      inst.exterior_medium_index = st.medium_exterior;
      if (o.t == "sphere") {  // scene.rs:418-436
        float radius = 1.0f;
        get_float(o, "radius", radius);
        const float s3[3] = {radius, radius, radius};
        inst.shape = RENE_SHAPE_SPHERE;
        inst.mesh_index = -1;
        affine12(round32(mul(st.ctm, from_scale(s3))), inst.matrix);
      } else if (o.t == "trianglemesh" || o.t == "loopsubdiv") {  // intermediate_scene.rs:922-996
        const Value* iv = o.get("indices");
        const Value* pv = o.get("P");
        if (!iv) fail(RENE_ERR_INVALID_SCENE, "Argument not found indices");
        if (!pv) fail(RENE_ERR_INVALID_SCENE, "Argument not found P");
        if (iv->type != VType::Integer) fail(RENE_ERR_INVALID_SCENE, "unmatched type on indices");
        if (pv->type != VType::Point) fail(RENE_ERR_INVALID_SCENE, "unmatched type on P");
        const Value* nv = o.get("N");
        if (nv && nv->type != VType::Normal) fail(RENE_ERR_INVALID_SCENE, "unmatched type on N");
        const Value* uvv = o.get("st");
        if (!uvv) uvv = o.get("uv");
        if (uvv && uvv->type != VType::Float) fail(RENE_ERR_INVALID_SCENE, "unmatched type on st");
        if (iv->i.size() % 3) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
        size_t nvert = pv->f.size() / 3;
        if (nv && nv->f.size() != pv->f.size()) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
        if (uvv && uvv->f.size() < 2 * nvert) fail(RENE_ERR_INVALID_SCENE, "unmatched value length");
        std::vector<rene_vertex> verts(nvert);
        for (size_t i = 0; i < nvert; ++i) {
          rene_vertex v{};
          std::copy(&pv->f[3 * i], &pv->f[3 * i] + 3, v.position);
          if (nv) std::copy(&nv->f[3 * i], &nv->f[3 * i] + 3, v.normal);
          if (uvv) { v.uv[0] = uvv->f[2 * i]; v.uv[1] = uvv->f[2 * i + 1]; }
          verts[i] = v;
        }
        std::vector<uint32_t> idx(iv->i.size());
        for (size_t i = 0; i < idx.size(); ++i) {
          idx[i] = (uint32_t)iv->i[i];
          if (idx[i] >= nvert) fail(RENE_ERR_INVALID_SCENE, "trianglemesh index out of range");
        }
        if (o.t == "loopsubdiv") {  // intermediate_scene.rs:985-990, subdivision.rs:25-76
          int nlevels = 0;
          if (!get_int(o, "nlevels", nlevels)) fail(RENE_ERR_INVALID_SCENE, "Argument not found nlevels");
          if (nlevels < 0 || nlevels > 10) fail(RENE_ERR_INVALID_SCENE, "loopsubdiv nlevels out of range (0..10)");
          std::string why = rene::loop_subdivide(verts, idx, (unsigned)nlevels);
          if (!why.empty()) fail(RENE_ERR_INVALID_SCENE, why);
        }
        inst.shape = RENE_SHAPE_TRIANGLE;
        inst.mesh_index = (int32_t)push_mesh(std::move(verts), std::move(idx));
        affine12(st.ctm, inst.matrix);
      } else if (o.t == "plymesh") {  // intermediate_scene.rs:997-1012
        std::string fn;
        if (!get_str(o, "filename", fn)) fail(RENE_ERR_INVALID_SCENE, "Argument not found filename");
        std::vector<rene_vertex> verts;
        std::vector<uint32_t> idx;
        load_ply(fn, verts, idx);
        inst.shape = RENE_SHAPE_TRIANGLE;
        inst.mesh_index = (int32_t)push_mesh(std::move(verts), std::move(idx));
        affine12(st.ctm, inst.matrix);
      } else {
        fail(RENE_ERR_INVALID_SCENE, "Invalid Shape type " + o.t);
      }
      sc.instances.push_back(inst);
    }
  }

  // ---- Scene::create, scene.rs:100-168; IntermediateScene::from_scene 1043-1107 ----
  void run(const std::vector<SceneStmt>& stmts) {
    for (const SceneStmt& s : stmts) {
      switch (s.kind) {
        case SceneStmt::LookAt: world_to_camera = round32(mul(world_to_camera, look_at_lh(s.v, s.v + 3, s.v + 6))); break;
        case SceneStmt::Translate: world_to_camera = round32(mul(world_to_camera, from_translation(s.v))); break;
        case SceneStmt::Rotate: world_to_camera = round32(mul(world_to_camera, from_axis_angle(s.v + 1, s.v[0]))); break;
        case SceneStmt::Scale: world_to_camera = round32(mul(world_to_camera, from_scale(s.v))); break;
        case SceneStmt::ConcatTransform: {
          M4 m;
          for (int i = 0; i < 16; ++i) m.m[i] = s.v[i];
          world_to_camera = round32(mul(world_to_camera, m));
          break;
        }
        case SceneStmt::Transform:
          for (int i = 0; i < 16; ++i) world_to_camera.m[i] = s.v[i];
          break;
        case SceneStmt::SceneObject: {
          const Object& o = s.obj;
          if (o.kind == "Camera") {
            if (o.t != "perspective") fail(RENE_ERR_INVALID_SCENE, "Invalid Camera type " + o.t);
            float f = 90.0f;
            get_float(o, "fov", f);
            fov = f * 3.14159265358979323846f / 180.0f;
          } else if (o.kind == "Film") {
            if (o.t != "image") fail(RENE_ERR_INVALID_SCENE, "Invalid Film type " + o.t);
            std::string fn = "out.png";
            int xr = 640, yr = 480;
            get_str(o, "filename", fn);
            get_int(o, "xresolution", xr);
            get_int(o, "yresolution", yr);
            sc.film_filename = fn;
            xres = (uint32_t)xr;
            yres = (uint32_t)yr;
          } else if (o.kind == "Integrator") {
            integrator = o.t == "path" ? RENE_INTEGRATOR_PATH : RENE_INTEGRATOR_VOLPATH;  // Q7
          }  // Sampler, PixelFilter: ignored (scene.rs:120-128)
          break;
        }
        case SceneStmt::WorldBlock: {
          WorldState st;
          st.coord_system["camera"] = world_to_camera;
          append_world(st, s.world);
          break;
        }
      }
    }
  }

  void finish() {
    double aspect = (double)(float)((float)xres / (float)yres);
    double f = fov;
    if (yres > xres) f = (float)(std::atan(std::tan(f * 0.5) / (double)xres * (double)yres) * 2.0);  // scene.rs:156-162
    rene_scene_desc& d = sc.desc;
    d.struct_size = sizeof(rene_scene_desc);
    d.integrator = integrator;
    d.xresolution = xres;
    d.yresolution = yres;
    to_f32(inverse(perspective_lh(f, aspect, 0.01f, 1000.0)), d.uniform.projection_inv);
    to_f32(inverse(world_to_camera), d.uniform.camera_to_world);
    to_f32(bg_matrix, d.uniform.background_matrix);
    std::copy(bg_color, bg_color + 4, d.uniform.background_color);
    d.uniform.background_texture = bg_texture;
    sc.meshes.resize(sc.mesh_vertices.size());
    for (size_t i = 0; i < sc.meshes.size(); ++i) {
      sc.meshes[i].vertices = sc.mesh_vertices[i].data();
      sc.meshes[i].indices = sc.mesh_indices[i].data();
      sc.meshes[i].n_vertices = (uint32_t)sc.mesh_vertices[i].size();
      sc.meshes[i].n_indices = (uint32_t)sc.mesh_indices[i].size();
    }
    for (size_t i = 0; i < sc.images.size(); ++i) sc.images[i].rgba = sc.image_data[i].data();
    d.n_instances = (uint32_t)sc.instances.size();
    d.instances = sc.instances.data();
    d.n_meshes = (uint32_t)sc.meshes.size();
    d.meshes = sc.meshes.data();
    d.n_materials = (uint32_t)sc.materials.size();
    d.materials = sc.materials.data();
    d.n_textures = (uint32_t)sc.textures.size();
    d.textures = sc.textures.data();
    d.n_area_lights = (uint32_t)sc.area_lights.size();
    d.area_lights = sc.area_lights.data();
    d.n_lights = (uint32_t)sc.lights.size();
    d.lights = sc.lights.data();
    d.n_images = (uint32_t)sc.images.size();
    d.images = sc.images.data();
    d.n_mediums = (uint32_t)sc.mediums.size();
    d.mediums = sc.mediums.data();
  }
};

int load_impl(const std::string& text_in, const std::string& base_dir, rene_scene** out) {
  if (!out) {
    rene::set_last_error("NULL out pointer");
    return RENE_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  try {
    std::string text = expand_include(text_in, base_dir);
    Parser p(text);
    std::vector<SceneStmt> stmts = p.scene();
    std::unique_ptr<rene_scene> sc(new rene_scene());
    g_spd_base_dir = base_dir;
    Builder b(*sc, base_dir);
    b.run(stmts);
    b.finish();
    *out = sc.release();
    return RENE_OK;
  } catch (const LoadError& e) {
    rene::set_last_error(e.msg);
    return e.code;
  } catch (const std::exception& e) {
    rene::set_last_error(e.what());
    return RENE_ERR_INVALID_SCENE;
  }
}

}  // namespace

extern "C" {

int rene_scene_parse_pbrt(const char* text, const char* base_dir, rene_scene** out) {
  if (!text) {
    rene::set_last_error("NULL text");
    return RENE_ERR_INVALID_ARGUMENT;
  }
  return load_impl(text, base_dir ? base_dir : "", out);
}

int rene_scene_load_pbrt(const char* path, rene_scene** out) {  // rene/src/main.rs:107-205
  if (!path) {
    rene::set_last_error("NULL path");
    return RENE_ERR_INVALID_ARGUMENT;
  }
  std::string p(path), dir;
  size_t slash = p.rfind('/');
  if (slash != std::string::npos) dir = p.substr(0, slash);
  std::string text;
  try {
    text = read_file(p);
  } catch (const LoadError& e) {
    rene::set_last_error(e.msg);
    if (out) *out = nullptr;
    return e.code;
  }
  return load_impl(text, dir, out);
}

const rene_scene_desc* rene_scene_get_desc(const rene_scene* s) { return s ? &s->desc : nullptr; }
const char* rene_scene_film_filename(const rene_scene* s) { return s ? s->film_filename.c_str() : nullptr; }
void rene_scene_free(rene_scene* s) { delete s; }

}  // extern "C"
