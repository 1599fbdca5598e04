// rthost_config.cpp — host-side mirror of the reference's config / OBJ / MTL ingest and scene assembly:
//   loadConfig            /root/reference/app/Config.hs:127-141 (aeson Generic decoding, schema :29-125)
//   parseSceneObjects     app/Rendering/SceneParser.hs:26-80
//   parseObj/parseFace    app/Parser/Object.hs:16-88
//   parseMaterial         app/Parser/Material.hs:11-57
// including the quirks SURVEY.md §5.6 lists: JSON material ids from 1, MTL ids from 0 in name order,
// left-biased unions, 3-gons emitted as (v0, v2, v1), n-gons fanned un-flipped, config triangles twice.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rthost.h"

namespace rth { int fail(int code, const char *fmt, ...); }
using rth::fail;

namespace {

// ------------------------------------------------------------------ a small JSON reader
struct JValue {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;
  const JValue *get(const char *key) const {
    if (kind != Obj) return nullptr;
    const JValue *found = nullptr;
    for (auto &kv : obj) if (kv.first == key) found = &kv.second;   // aeson keeps the last duplicate
    return found;
  }
};

struct JParser {
  const char *p, *end;
  std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool fail_(const std::string &m) { if (err.empty()) err = m; return false; }
  bool value(JValue &v) {
    ws();
    if (p >= end) return fail_("unexpected end of input");
    char c = *p;
    if (c == '{') return object(v);
    if (c == '[') return array(v);
    if (c == '"') { v.kind = JValue::Str; return string(v.str); }
    if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; v.kind = JValue::Bool; v.b = true; return true; }
    if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; v.kind = JValue::Bool; v.b = false; return true; }
    if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; v.kind = JValue::Null; return true; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      // the JSON number grammar (what aeson accepts): -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?  - scanned first,
      // so that strtod's extras (hex floats, "inf", "nan", a leading '+') are rejected as the reference rejects them
      const char *q = p;
      auto digits = [&]() { const char *q0 = q; while (q < end && *q >= '0' && *q <= '9') ++q; return q > q0; };
      if (*q == '-') ++q;
      if (q < end && *q == '0') ++q; else if (!digits()) return fail_("bad number");
      if (q < end && *q == '.') { ++q; if (!digits()) return fail_("bad number"); }
      if (q < end && (*q == 'e' || *q == 'E')) { ++q; if (q < end && (*q == '+' || *q == '-')) ++q; if (!digits()) return fail_("bad number"); }
      char *e = nullptr;
      std::string tmp(p, (size_t)(q - p));
      double d = strtod(tmp.c_str(), &e);
      if (e != tmp.c_str() + tmp.size()) return fail_("bad number");
      p = q;
      v.kind = JValue::Num; v.num = d;
      return true;
    }
    return fail_(std::string("unexpected character '") + c + "'");
  }
  bool string(std::string &s) {
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return fail_("bad escape");
        switch (*p) {
          case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
          case 'b': s += '\b'; break; case 'f': s += '\f'; break; case '/': s += '/'; break;
          case '\\': s += '\\'; break; case '"': s += '"'; break;
          case 'u': {
            if (end - p < 5) return fail_("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) s += (char)cp;
            else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
            else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: return fail_("bad escape");
        }
        ++p;
      } else s += *p++;
    }
    if (p >= end) return fail_("unterminated string");
    ++p;
    return true;
  }
  bool array(JValue &v) {
    v.kind = JValue::Arr; ++p; ws();
    if (p < end && *p == ']') { ++p; return true; }
    for (;;) {
      v.arr.emplace_back();
      if (!value(v.arr.back())) return false;
      ws();
      if (p < end && *p == ',') { ++p; continue; }
      if (p < end && *p == ']') { ++p; return true; }
      return fail_("expected ',' or ']'");
    }
  }
  bool object(JValue &v) {
    v.kind = JValue::Obj; ++p; ws();
    if (p < end && *p == '}') { ++p; return true; }
    for (;;) {
      ws();
      if (p >= end || *p != '"') return fail_("expected object key");
      std::string k;
      if (!string(k)) return false;
      ws();
      if (p >= end || *p != ':') return fail_("expected ':'");
      ++p;
      v.obj.emplace_back(k, JValue());
      if (!value(v.obj.back().second)) return false;
      ws();
      if (p < end && *p == ',') { ++p; continue; }
      if (p < end && *p == '}') { ++p; return true; }
      return fail_("expected ',' or '}'");
    }
  }
};

// ------------------------------------------------------------------ decoding helpers (aeson Generic semantics)
struct Dec {
  std::string err;
  bool bad(const std::string &m) { if (err.empty()) err = m; return false; }
  bool vec3(const JValue *v, const char *what, double out[3]) {        // FromJSON Vec3, Core/Vec3.hs:39-48
    if (!v || v->kind != JValue::Arr) return bad(std::string("expected Vec3 array for ") + what);
    if (v->arr.size() < 3) return bad("Failed parsing Vec3 from JSON config. Vec3 must contain exactly three valid numbers.");
    for (int k = 0; k < 3; ++k) {
      if (v->arr[k].kind != JValue::Num) return bad("Failed parsing Vec3 from JSON config. Vec3 must contain exactly three valid numbers.");
      out[k] = v->arr[k].num;
    }
    return true;
  }
  bool num(const JValue *v, const char *what, double &out) {
    if (!v || v->kind != JValue::Num) return bad(std::string("key \"") + what + "\" not found or not a number");
    out = v->num; return true;
  }
  bool integer(const JValue *v, const char *what, int &out) {
    double d;
    if (!num(v, what, d)) return false;
    if (d != std::floor(d) || std::fabs(d) > 2147483647.0) return bad(std::string("key \"") + what + "\" is not an Int");
    out = (int)d; return true;
  }
  bool boolean(const JValue *v, const char *what, bool &out) {
    if (!v || v->kind != JValue::Bool) return bad(std::string("key \"") + what + "\" not found or not a Bool");
    out = v->b; return true;
  }
  bool present(const JValue *v) { return v && v->kind != JValue::Null; }   // Maybe field: absent or null => Nothing
};

struct Mat {          // Rendering/Material.hs:13-21
  double diffuse[3] = {1, 1, 1};
  bool hasEmission = false; double emission[3] = {0, 0, 0};
  bool hasShininess = false; double shininess = 0;
  bool hasIor = false; double ior = 0;
  bool hasTransmission = false; double transmission = 0;
};

bool decode_material(Dec &d, const JValue *v, Mat &m) {   // FromJSON Material, Material.hs:37-46
  if (!v || v->kind != JValue::Obj) return d.bad("expected Material object");
  if (d.present(v->get("diffuseColor")) && !d.vec3(v->get("diffuseColor"), "diffuseColor", m.diffuse)) return false;
  if (d.present(v->get("emissionColor"))) { m.hasEmission = true; if (!d.vec3(v->get("emissionColor"), "emissionColor", m.emission)) return false; }
  if (d.present(v->get("specularColor"))) { double tmp[3]; if (!d.vec3(v->get("specularColor"), "specularColor", tmp)) return false; }
  if (d.present(v->get("shininess"))) { m.hasShininess = true; if (!d.num(v->get("shininess"), "shininess", m.shininess)) return false; }
  if (d.present(v->get("ior"))) { m.hasIor = true; if (!d.num(v->get("ior"), "ior", m.ior)) return false; }
  if (d.present(v->get("dissolve"))) { double tmp; if (!d.num(v->get("dissolve"), "dissolve", tmp)) return false; }
  if (d.present(v->get("transmission"))) { m.hasTransmission = true; if (!d.num(v->get("transmission"), "transmission", m.transmission)) return false; }
  return true;
}

rt_material to_abi(int id, const Mat &m) {
  rt_material r;
  std::memset(&r, 0, sizeof r);
  r.id = id;
  for (int k = 0; k < 3; ++k) { r.diffuse[k] = m.diffuse[k]; r.emission[k] = m.emission[k]; }
  if (m.hasEmission) r.flags |= RT_MAT_HAS_EMISSION;
  if (m.hasShininess) r.flags |= RT_MAT_HAS_SHININESS;
  if (m.hasIor) r.flags |= RT_MAT_HAS_IOR;
  if (m.hasTransmission) r.flags |= RT_MAT_HAS_TRANSMISSION;
  r.shininess = m.shininess; r.ior = m.ior; r.transmission = m.transmission;
  return r;
}

// Haskell `words`: split on isSpace
std::vector<std::string> words(const std::string &l) {
  std::vector<std::string> w;
  size_t i = 0;
  while (i < l.size()) {
    while (i < l.size() && isspace((unsigned char)l[i])) ++i;
    size_t j = i;
    while (j < l.size() && !isspace((unsigned char)l[j])) ++j;
    if (j > i) w.emplace_back(l, i, j - i);
    i = j;
  }
  return w;
}
// Haskell `lines`
std::vector<std::string> lines_of(const std::string &s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find('\n', i);
    if (j == std::string::npos) { out.emplace_back(s, i); break; }
    out.emplace_back(s, i, j - i);
    i = j + 1;
  }
  return out;
}
bool read_file(const std::string &path, std::string &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss; ss << f.rdbuf(); out = ss.str();
  return true;
}
// allocation-free tokens for the OBJ reader: pointer ranges into the file text (`words`: split at any white space)
struct Tok { const char *b, *e; };
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
inline void split_words(const char *b, const char *e, std::vector<Tok> &out) {
  out.clear();
  while (b < e) {
    while (b < e && is_ws(*b)) ++b;
    if (b >= e) break;
    const char *s = b;
    while (b < e && !is_ws(*b)) ++b;
    out.push_back(Tok{s, b});
  }
}
inline bool tok_is(const Tok &t, const char *lit) { const size_t n = strlen(lit); return (size_t)(t.e - t.b) == n && memcmp(t.b, lit, n) == 0; }
inline bool tok_double(const Tok &t, double &out) {          // same acceptance as read_double below
  const size_t n = (size_t)(t.e - t.b);
  char buf[64];
  if (n == 0) return false;
  if (n < sizeof buf) { memcpy(buf, t.b, n); buf[n] = 0; char *e = nullptr; out = strtod(buf, &e); return e == buf + n; }
  const std::string s(t.b, n); char *e = nullptr; out = strtod(s.c_str(), &e); return e != s.c_str() && *e == 0;
}
inline bool tok_long(const Tok &t, long &out) {
  const size_t n = (size_t)(t.e - t.b);
  char buf[32];
  if (n == 0 || n >= sizeof buf) return false;
  memcpy(buf, t.b, n); buf[n] = 0;
  char *e = nullptr; out = strtol(buf, &e, 10);
  return e == buf + n;
}
bool read_double(const std::string &s, double &out) {
  char *e = nullptr;
  out = strtod(s.c_str(), &e);
  return e != s.c_str() && *e == 0;
}

// parseMaterial, Parser/Material.hs:11-50.  Returns name -> Mat (ordered by name: M.toList order).
bool parse_mtl(const std::string &content, std::map<std::string, Mat> &out, std::string &err) {
  bool have = false; std::string name; Mat cur; bool hasIllum = false; int illum = 0;
  auto finalize = [&]() {                                              // finalizeMaterial, :43-50
    Mat m = cur;
    bool t = hasIllum && m.hasIor && (illum == 7 || illum == 9) && m.ior > 1.0;
    m.hasTransmission = t; m.transmission = t ? 1.0 : 0.0;
    out[name] = m;                                                     // M.insert: later newmtl of the same name overwrites
  };
  for (auto &l : lines_of(content)) {
    auto w = words(l);
    if (w.empty()) continue;
    auto need = [&](size_t n) { return w.size() >= n; };
    auto rd = [&](size_t i, double &d) { if (!read_double(w[i], d)) { err = "Prelude.read: no parse (" + w[i] + ")"; return false; } return true; };
    if (w[0] == "newmtl" && need(2)) { if (have) finalize(); have = true; name = w[1]; cur = Mat(); hasIllum = false; }
    else if (!have) continue;                                          // updateCurrent on Nothing, :41
    else if (w[0] == "Kd" && need(4)) { if (!rd(1, cur.diffuse[0]) || !rd(2, cur.diffuse[1]) || !rd(3, cur.diffuse[2])) return false; }
    else if (w[0] == "Ks" && need(4)) { double t; if (!rd(1, t) || !rd(2, t) || !rd(3, t)) return false; }
    else if (w[0] == "Ke" && need(4)) { cur.hasEmission = true; if (!rd(1, cur.emission[0]) || !rd(2, cur.emission[1]) || !rd(3, cur.emission[2])) return false; }
    else if (w[0] == "Ns" && need(2)) { cur.hasShininess = true; if (!rd(1, cur.shininess)) return false; }
    else if (w[0] == "Ni" && need(2)) { cur.hasIor = true; if (!rd(1, cur.ior)) return false; }
    else if (w[0] == "d" && need(2)) { double t; if (!rd(1, t)) return false; }
    else if (w[0] == "illum" && need(2)) {   // `read :: Int` (Parser/Material.hs): optional '-', decimal digits, nothing else ("2.0" is no parse)
      const std::string &tk = w[1];
      size_t k = (tk.size() > 1 && tk[0] == '-') ? 1 : 0;
      bool ok = tk.size() > k;
      for (size_t q = k; q < tk.size(); ++q) ok = ok && tk[q] >= '0' && tk[q] <= '9';
      if (!ok) { err = "Prelude.read: no parse (illum)"; return false; }
      hasIllum = true; illum = atoi(tk.c_str());
    }
  }
  if (have) finalize();
  return true;
}

}  // namespace

struct rth_config {
  rth_config_view view{};
  std::vector<int32_t> kind, material;
  std::vector<double> data;
  std::vector<rt_material> materials;
  std::vector<rt_light> lights;
  std::string sky;
};

extern "C" {

int rth_config_load(const char *json_path, rth_config **out) {
  if (!json_path || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  std::string text;
  if (!read_file(json_path, text)) return fail(RT_ERR_INVALID, "Error: Config file not found -> %s", json_path);   // Config.hs:132
  JParser jp{text.data(), text.data() + text.size(), {}};
  JValue root;
  if (!jp.value(root)) return fail(RT_ERR_INVALID, "Error parsing config file -> %s\nJSON Error: %s", json_path, jp.err.c_str());
  jp.ws();
  if (jp.p != jp.end) return fail(RT_ERR_INVALID, "Error parsing config file -> %s\nJSON Error: trailing garbage", json_path);
  Dec d;
  auto cfg = std::make_unique<rth_config>();
  rth_config_view &v = cfg->view;
  auto bad = [&]() { return fail(RT_ERR_INVALID, "Error parsing config file -> %s\nJSON Error: %s", json_path, d.err.c_str()); };
  if (root.kind != JValue::Obj) { d.bad("expected Config object"); return bad(); }

  // ---- image (Config.hs:29-37), raytracer (:58-64): every field required
  const JValue *img = root.get("image"), *rtv = root.get("raytracer"), *cam = root.get("camera"),
               *bg = root.get("background"), *sc = root.get("scene");
  if (!img || img->kind != JValue::Obj) { d.bad("key \"image\" not found"); return bad(); }
  if (!bg || bg->kind != JValue::Obj) { d.bad("key \"background\" not found"); return bad(); }
  if (!cam || cam->kind != JValue::Obj) { d.bad("key \"camera\" not found"); return bad(); }
  if (!rtv || rtv->kind != JValue::Obj) { d.bad("key \"raytracer\" not found"); return bad(); }
  if (!sc || sc->kind != JValue::Obj) { d.bad("key \"scene\" not found"); return bad(); }
  int W, H, spp, maxb, depth, leaf; bool aa, useBVH; double gamma, exposure;
  if (!d.integer(img->get("width"), "width", W) || !d.integer(img->get("height"), "height", H) ||
      !d.integer(img->get("samplesPerPixel"), "samplesPerPixel", spp) || !d.boolean(img->get("antialiasing"), "antialiasing", aa) ||
      !d.num(img->get("gamma"), "gamma", gamma) || !d.num(img->get("exposure"), "exposure", exposure)) return bad();
  if (!d.integer(rtv->get("maxBounces"), "maxBounces", maxb) || !d.boolean(rtv->get("useBVH"), "useBVH", useBVH) ||
      !d.integer(rtv->get("bvhMaxDepth"), "bvhMaxDepth", depth) || !d.integer(rtv->get("leafThreshold"), "leafThreshold", leaf)) return bad();
  std::memset(&v.params, 0, sizeof v.params);
  v.params.width = W; v.params.height = H; v.params.samples_per_pixel = spp; v.params.antialiasing = aa;
  v.params.max_bounces = maxb; v.params.exposure = exposure; v.params.gamma = gamma;
  v.use_bvh = useBVH; v.bvh_max_depth = depth; v.leaf_threshold = leaf;

  // ---- background (Config.hs:40-43): {"tag": "Gradient"|"SolidColor", ...}
  const JValue *tag = bg->get("tag");
  if (!tag || tag->kind != JValue::Str) { d.bad("background: key \"tag\" not found"); return bad(); }
  if (tag->str == "Gradient") {
    v.bg_kind = RT_BG_GRADIENT;
    if (!d.vec3(bg->get("color1"), "color1", v.bg_color1) || !d.vec3(bg->get("color2"), "color2", v.bg_color2)) return bad();
  } else if (tag->str == "SolidColor") {
    v.bg_kind = RT_BG_SOLID;
    if (!d.vec3(bg->get("color"), "color", v.bg_color1)) return bad();
    std::memcpy(v.bg_color2, v.bg_color1, sizeof v.bg_color1);
  } else { d.bad("background: unknown tag " + tag->str); return bad(); }

  // ---- camera (Config.hs:46-54) -> defaultCamera with aspect = W/H (ImageGenerator.hs:50-58)
  double lf[3], la[3], up[3], vfov, aperture, focus;
  if (!d.vec3(cam->get("lookFrom"), "lookFrom", lf) || !d.vec3(cam->get("lookAt"), "lookAt", la) || !d.vec3(cam->get("vUp"), "vUp", up) ||
      !d.num(cam->get("vfov"), "vfov", vfov) || !d.num(cam->get("aperture"), "aperture", aperture) || !d.num(cam->get("focusDist"), "focusDist", focus)) return bad();
  rth_default_camera(lf, la, up, vfov, (double)W / (double)H, aperture, focus, &v.camera);

  // ---- Phase 0: JSON materials, ids from 1 (Parser/Material.hs:52-57)
  std::map<std::string, int> jsonNameToId, mtlNameToId, nameToId;
  std::map<int, Mat> jsonIdToMat, mtlIdToMat, idToMat;
  if (d.present(sc->get("materials"))) {
    const JValue *ms = sc->get("materials");
    if (ms->kind != JValue::Arr) { d.bad("materials: expected a list of [name, material] pairs"); return bad(); }
    int id = 1;
    for (auto &pair : ms->arr) {
      if (pair.kind != JValue::Arr || pair.arr.size() != 2 || pair.arr[0].kind != JValue::Str) { d.bad("materials: expected [name, material]"); return bad(); }
      Mat m;
      if (!decode_material(d, &pair.arr[1], m)) return bad();
      jsonNameToId[pair.arr[0].str] = id;       // M.fromList: a later duplicate name overwrites
      jsonIdToMat[id] = m;
      ++id;
    }
  }
  // ---- objFiles entries (Config.hs:80-86)
  struct ObjEntry { std::string path; double pos[3]; };
  std::vector<ObjEntry> entries;
  if (d.present(sc->get("objFiles"))) {
    const JValue *of = sc->get("objFiles");
    if (of->kind != JValue::Arr) { d.bad("objFiles: expected a list"); return bad(); }
    for (auto &e : of->arr) {
      if (e.kind != JValue::Obj) { d.bad("objFiles: expected objects"); return bad(); }
      ObjEntry oe;
      const JValue *pth = e.get("path");
      if (!pth || pth->kind != JValue::Str) { d.bad("objFiles: key \"path\" not found"); return bad(); }
      oe.path = pth->str;
      if (!d.vec3(e.get("objposition"), "objposition", oe.pos)) return bad();
      if (d.present(e.get("overrideColor"))) { double t[3]; if (!d.vec3(e.get("overrideColor"), "overrideColor", t)) return bad(); }
      if (d.present(e.get("overrideMaterial"))) { Mat t; if (!decode_material(d, e.get("overrideMaterial"), t)) return bad(); }   // parsed, never used (SceneParser.hs:56-62)
      entries.push_back(oe);
    }
  }
  // ---- Phase 1: MTL materials per OBJ file (SceneParser.hs:33-40, 83-95); M.unions is left-biased
  std::vector<std::string> objTexts(entries.size());
  for (size_t i = 0; i < entries.size(); ++i) {
    if (!read_file(entries[i].path, objTexts[i])) return fail(RT_ERR_INVALID, "%s: openFile: does not exist (No such file or directory)", entries[i].path.c_str());
    std::string mtlLine;
    {                                                                          // first line that starts with "mtllib"
      const std::string &t = objTexts[i];
      for (size_t b0 = 0; b0 < t.size();) {
        size_t e0 = t.find('\n', b0);
        if (e0 == std::string::npos) e0 = t.size();
        if (e0 - b0 >= 6 && t.compare(b0, 6, "mtllib") == 0) { mtlLine = t.substr(b0, e0 - b0); break; }
        b0 = e0 + 1;
      }
    }
    auto w = words(mtlLine);
    std::string mtlText;
    if (w.size() == 2 && w[0] == "mtllib") {
      std::string base = entries[i].path;
      size_t slash = base.find_last_of('/');
      base = slash == std::string::npos ? std::string(".") : base.substr(0, slash);
      read_file(base + "/" + w[1], mtlText);                                  // missing file => ""
    }
    std::map<std::string, Mat> raw;
    std::string perr;
    if (!parse_mtl(mtlText, raw, perr)) return fail(RT_ERR_INVALID, "%s", perr.c_str());
    int id = 0;                                                                // zip [0..] (M.toList rawMap), Material.hs:14-16
    for (auto &kv : raw) {
      mtlNameToId.emplace(kv.first, id);                                       // emplace keeps the first: left-biased union
      mtlIdToMat.emplace(id, kv.second);
      ++id;
    }
  }
  // ---- Phase 2: merge, JSON shadows MTL (SceneParser.hs:43-44)
  nameToId = jsonNameToId; for (auto &kv : mtlNameToId) nameToId.emplace(kv.first, kv.second);
  idToMat = jsonIdToMat; for (auto &kv : mtlIdToMat) idToMat.emplace(kv.first, kv.second);
  auto mat_of = [&](const JValue &name) -> int {                               // fromMaybe 0 (mname >>= lookup), :108
    if (name.kind != JValue::Str) return 0;
    auto it = nameToId.find(name.str);
    return it == nameToId.end() ? 0 : it->second;
  };

  // ---- Phase 3: config objects (Config.hs:73-77): {"tag": ..., "contents": [...]}
  struct CfgObj { int kind; double d[9]; int mat; };
  std::vector<CfgObj> hittables;
  if (d.present(sc->get("objects"))) {
    const JValue *objs = sc->get("objects");
    if (objs->kind != JValue::Arr) { d.bad("objects: expected a list"); return bad(); }
    for (auto &o : objs->arr) {
      const JValue *t = o.get("tag"), *c = o.get("contents");
      if (!t || t->kind != JValue::Str || !c || c->kind != JValue::Arr) { d.bad("objects: expected {tag, contents}"); return bad(); }
      CfgObj co; std::memset(&co, 0, sizeof co);
      double col[3];
      if (t->str == "SphereObj") {
        if (c->arr.size() != 4) { d.bad("SphereObj: expected 4 fields"); return bad(); }
        co.kind = RT_PRIM_SPHERE;
        if (!d.vec3(&c->arr[0], "center", co.d) || !d.num(&c->arr[1], "radius", co.d[3]) || !d.vec3(&c->arr[2], "color", col)) return bad();
        co.mat = mat_of(c->arr[3]);
      } else if (t->str == "PlaneObj") {
        if (c->arr.size() != 4) { d.bad("PlaneObj: expected 4 fields"); return bad(); }
        co.kind = RT_PRIM_PLANE;
        if (!d.vec3(&c->arr[0], "point", co.d) || !d.vec3(&c->arr[1], "normal", co.d + 3) || !d.vec3(&c->arr[2], "color", col)) return bad();
        co.mat = mat_of(c->arr[3]);
      } else if (t->str == "TriangleObj") {
        if (c->arr.size() != 5) { d.bad("TriangleObj: expected 5 fields"); return bad(); }
        co.kind = RT_PRIM_TRIANGLE;
        if (!d.vec3(&c->arr[0], "v0", co.d) || !d.vec3(&c->arr[1], "v1", co.d + 3) || !d.vec3(&c->arr[2], "v2", co.d + 6) || !d.vec3(&c->arr[3], "color", col)) return bad();
        co.mat = mat_of(c->arr[4]);
      } else { d.bad("objects: unknown tag " + t->str); return bad(); }
      hittables.push_back(co);
    }
  }
  auto push_obj = [&](int kind, const double *dd, int mat) {
    cfg->kind.push_back(kind); cfg->material.push_back(mat);
    cfg->data.insert(cfg->data.end(), dd, dd + 9);
  };
  // configTriangles first (SceneParser.hs:70-72)
  int nTri = 0;
  for (auto &h : hittables) if (h.kind == RT_PRIM_TRIANGLE) { push_obj(h.kind, h.d, h.mat); ++nTri; }
  // ---- Phase 4: OBJ triangles (Parser/Object.hs:16-88).  Files are independent: each is parsed by its own thread into
  //      its own arrays (no per-line strings: tokens are pointer ranges into the file text), then appended in file order,
  //      so the object list and the first error reported are those of the sequential reference.
  struct ObjOut { std::vector<double> data; std::vector<int32_t> mat; std::string err; bool empty = false; };
  std::vector<ObjOut> parsed(entries.size());
  auto parse_obj_file = [&](size_t i) {
    ObjOut &o = parsed[i];
    const std::string &content = objTexts[i];
    if (content.empty()) { o.empty = true; return; }                             // Object.hs:79-82
    const char *tb = content.data(), *te = tb + content.size();
    auto next_line = [&](const char *&p, const char *&lb, const char *&le) {   // `lines`: split at \n, the terminator dropped
      if (p >= te) return false;
      lb = p;
      const char *nl = (const char *)memchr(p, '\n', (size_t)(te - p));
      le = nl ? nl : te;
      p = nl ? nl + 1 : te;
      return true;
    };
    std::vector<Tok> w;
    std::vector<double> verts;
    std::vector<char> bad_vert;
    const char *p = tb, *lb, *le;
    while (next_line(p, lb, le)) {
      if (le - lb < 2 || lb[0] != 'v' || lb[1] != ' ') continue;                 // "v " `isPrefixOf` l, :19
      split_words(lb, le, w);
      double x = 0, y = 0, z = 0;
      // `verts` is a lazy list in the reference: a malformed "v" line (:39-40) only raises when a face references it (:52-54)
      const bool ok = w.size() == 4 && tok_double(w[1], x) && tok_double(w[2], y) && tok_double(w[3], z);
      bad_vert.push_back(ok ? 0 : 1);
      verts.push_back(x); verts.push_back(y); verts.push_back(z);
    }
    const long nv = (long)verts.size() / 3;
    bool haveMat = false; int curMatId = 0;
    std::vector<long> fv;
    p = tb;
    while (next_line(p, lb, le)) {
      split_words(lb, le, w);
      if (w.size() >= 2 && tok_is(w[0], "usemtl")) {                             // :27-28
        haveMat = true;
        auto it = nameToId.find(std::string(w[1].b, w[1].e));
        curMatId = it == nameToId.end() ? 0 : it->second;
        continue;
      }
      if (w.empty() || !tok_is(w[0], "f")) continue;
      const int matId = haveMat ? curMatId : 0;                                  // :30
      if (w.size() < 4) { o.err = "Invalid face format in .obj file"; return; }  // :67
      fv.clear();
      for (size_t k = 1; k < w.size(); ++k) {
        const char *sl = (const char *)memchr(w[k].b, '/', (size_t)(w[k].e - w[k].b));   // takeWhile (/= '/'), :47
        const Tok tok{w[k].b, sl ? sl : w[k].e};
        long ix;
        if (!tok_long(tok, ix)) { o.err = "Prelude.read: no parse (face index '" + std::string(tok.b, tok.e) + "')"; return; }
        if (ix < 1 || ix > nv) { o.err = "Prelude.!!: index out of range (face index " + std::to_string(ix) + ", " + std::to_string(nv) + " vertices)"; return; }
        if (bad_vert[ix - 1]) { o.err = "Invalid vertex format in .obj file"; return; }                                // :40, forced here
        fv.push_back(ix - 1);
      }
      auto emit = [&](long a, long b, long c) {
        double t[9];
        for (int k = 0; k < 3; ++k) {                                            // offsetTriangle, :85-86
          t[k] = verts[3 * a + k] + entries[i].pos[k];
          t[3 + k] = verts[3 * b + k] + entries[i].pos[k];
          t[6 + k] = verts[3 * c + k] + entries[i].pos[k];
        }
        o.data.insert(o.data.end(), t, t + 9); o.mat.push_back(matId);
      };
      if (fv.size() == 3) emit(fv[0], fv[2], fv[1]);                             // :49-57 (v0, v2, v1)
      else for (size_t k = 1; k + 1 < fv.size(); ++k) emit(fv[0], fv[k], fv[k + 1]);   // :59-66 fan, un-flipped
    }
  };
  {
    const size_t nt = std::min<size_t>(entries.size(), std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    std::atomic<size_t> next_file(0);
    std::vector<std::thread> pool;
    auto work = [&]() { for (size_t i; (i = next_file.fetch_add(1)) < entries.size();) parse_obj_file(i); };
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
  }
  for (size_t i = 0; i < entries.size(); ++i) {
    ObjOut &o = parsed[i];
    if (o.empty) { fprintf(stderr, "Error: OBJ file is empty or could not be read.\n"); continue; }
    if (!o.err.empty()) return fail(RT_ERR_INVALID, "%s", o.err.c_str());
    cfg->data.insert(cfg->data.end(), o.data.begin(), o.data.end());
    cfg->kind.insert(cfg->kind.end(), o.mat.size(), RT_PRIM_TRIANGLE);
    cfg->material.insert(cfg->material.end(), o.mat.begin(), o.mat.end());
    nTri += (int)o.mat.size();
    ObjOut().data.swap(o.data);
  }
  // all config objects again (SceneParser.hs:73)
  for (auto &h : hittables) push_obj(h.kind, h.d, h.mat);

  for (auto &kv : idToMat) cfg->materials.push_back(to_abi(kv.first, kv.second));

  // ---- lights (Config.hs:67-70)
  if (d.present(sc->get("lights"))) {
    const JValue *ls = sc->get("lights");
    if (ls->kind != JValue::Arr) { d.bad("lights: expected a list"); return bad(); }
    for (auto &l : ls->arr) {
      const JValue *t = l.get("tag");
      if (!t || t->kind != JValue::Str) { d.bad("lights: key \"tag\" not found"); return bad(); }
      rt_light L; std::memset(&L, 0, sizeof L);
      if (t->str == "PointLight") { L.kind = RT_LIGHT_POINT; if (!d.vec3(l.get("position"), "position", L.vec)) return bad(); }
      else if (t->str == "DirectionalLight") { L.kind = RT_LIGHT_DIRECTIONAL; if (!d.vec3(l.get("direction"), "direction", L.vec)) return bad(); }
      else { d.bad("lights: unknown tag " + t->str); return bad(); }
      if (!d.vec3(l.get("intensity"), "intensity", L.intensity)) return bad();
      cfg->lights.push_back(L);
    }
  }
  if (d.present(sc->get("skyTexture"))) {
    const JValue *s = sc->get("skyTexture");
    if (s->kind != JValue::Str) { d.bad("skyTexture: expected a string"); return bad(); }
    cfg->sky = s->str;
  }

  v.n_objs = (int32_t)cfg->kind.size();
  v.obj_kind = cfg->kind.data(); v.obj_data = cfg->data.data(); v.obj_material = cfg->material.data();
  v.n_triangles = nTri;
  v.n_materials = (int32_t)cfg->materials.size(); v.materials = cfg->materials.data();
  v.n_lights = (int32_t)cfg->lights.size(); v.lights = cfg->lights.data();
  v.sky_path = cfg->sky.empty() ? nullptr : cfg->sky.c_str();
  *out = cfg.release();
  return RT_OK;
}

void rth_config_destroy(rth_config *cfg) { delete cfg; }

int rth_config_get(const rth_config *cfg, rth_config_view *view) {
  if (!cfg || !view) return fail(RT_ERR_INVALID, "null argument");
  *view = cfg->view;
  return RT_OK;
}

}  // extern "C"
