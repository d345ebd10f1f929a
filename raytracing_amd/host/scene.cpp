// scene.cpp — scene reader, flattening and renderFrame of the C++ front-end.
// See scene.hpp for the reference files each part restates.
#include "scene.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <regex>
#include <sstream>
#include <stdexcept>

#include "json.hpp"

namespace rt {

ht_camera Camera::to_c() const {
  ht_camera c{};
  c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
  c.target[0] = target.x; c.target[1] = target.y; c.target[2] = target.z;
  c.vup[0] = vup.x; c.vup[1] = vup.y; c.vup[2] = vup.z;
  c.vfov = vfov;
  c.aspect_ratio = aspect_ratio;
  c.aperture = aperture;
  c.focus_distance = focus_distance;
  c.background[0] = background.x; c.background[1] = background.y; c.background[2] = background.z;
  return c;
}

// ---- files -----------------------------------------------------------------------
static std::string basename_of(const std::string& p) {
  size_t k = p.find_last_of('/');
  return k == std::string::npos ? p : p.substr(k + 1);
}

// jvm FileLoader.scala:8-12: absolute paths as they are, everything else is looked
// up by its LAST path component in the scene file's directory.
std::string FileLoader::file(const std::string& path) const {
  if (!path.empty() && path[0] == '/') return path;
  return scene_dir + "/" + basename_of(path);
}

static std::vector<uint8_t> read_bytes(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::string FileLoader::loadFile(const std::string& path) const {
  std::vector<uint8_t> b = read_bytes(file(path));
  return std::string(b.begin(), b.end());
}

Image FileLoader::loadImage(const std::string& path) const {
  std::vector<uint8_t> b = read_bytes(file(path));
  if (b.size() >= 8 && b[0] == 0x89 && b[1] == 'P' && b[2] == 'N' && b[3] == 'G') return decodePNG(b);
  if (b.size() >= 3 && b[0] == 0xFF && b[1] == 0xD8) return decodeJPEG(b);
  throw std::runtime_error("unsupported image format: " + path);
}

// ---- ASCII STL -------------------------------------------------------------------
STL fromSTL(const std::string& source, MaterialPtr material) {
  // the reference's pattern verbatim (BVH.scala:36-43): the '.' is not escaped
  static const std::string num = "(-?\\d+.\\d+)";
  static const std::regex re("^facet normal " + num + " " + num + " " + num + "\nouter loop\nvertex " + num +
                                 " " + num + " " + num + "\nvertex " + num + " " + num + " " + num +
                                 "\nvertex " + num + " " + num + " " + num + "\nendloop\nendfacet",
                             std::regex::ECMAScript | std::regex::multiline);
  auto to_double = [](const std::string& s) {
    size_t used = 0;
    double v = std::stod(s, &used);
    if (used != s.size()) throw std::invalid_argument("not a number: " + s);  // String.toDouble throws too
    return v;
  };
  STL out;
  for (std::sregex_iterator it(source.begin(), source.end(), re), end; it != end; ++it) {
    const std::smatch& m = *it;
    Triangle t;
    t.a = {to_double(m[4]), to_double(m[5]), to_double(m[6])};
    t.b = {to_double(m[7]), to_double(m[8]), to_double(m[9])};
    t.c = {to_double(m[10]), to_double(m[11]), to_double(m[12])};
    t.material = material;
    out.triangles.push_back(t);
  }
  return out;
}

// ---- JSON schema (JsonFormats.scala) -------------------------------------------------
static Vec3 read_vec3(const Json& j) { return {j.at(0).number(), j.at(1).number(), j.at(2).number()}; }

static void perlin_tables(uint64_t seed, Perlin& p) {
  // Texture.scala:38-42 draws these from an unseeded RNG; here from a splitmix64
  // stream, the same one raytracing_amd/scene.py uses, so both hosts get equal tables
  uint64_t state = seed;
  auto next = [&state]() {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto uniform = [&next]() { return -1.0 + 2.0 * ((double)(next() >> 11) * 0x1p-53); };
  p.ranvec.resize(256 * 3);
  for (int i = 0; i < 256; ++i) {
    double x = uniform(), y = uniform(), z = uniform();
    double l = std::sqrt(x * x + y * y + z * z);
    p.ranvec[i * 3] = x / l; p.ranvec[i * 3 + 1] = y / l; p.ranvec[i * 3 + 2] = z / l;
  }
  p.perm.resize(3 * 256);
  for (int k = 0; k < 3; ++k) {
    for (int i = 0; i < 256; ++i) p.perm[k * 256 + i] = i;
    for (int i = 255; i > 0; --i) {
      int j = (int)(next() % (uint64_t)(i + 1));
      std::swap(p.perm[k * 256 + i], p.perm[k * 256 + j]);
    }
  }
}

static TexturePtr read_texture(const Json& j, const FileLoader& fl, uint64_t& perlin_seed) {
  const std::string& t = j.at("type").string();  // JsonFormats.scala:19-29
  auto tex = std::make_shared<Texture>();
  if (t == "SolidColor") {
    tex->v = SolidColor{read_vec3(j.at("color"))};
  } else if (t == "Checker") {
    Checker c;
    c.size = j.at("size").number();
    c.t1 = read_texture(j.at("t1"), fl, perlin_seed);
    c.t2 = read_texture(j.at("t2"), fl, perlin_seed);
    tex->v = c;
  } else if (t == "Perlin") {
    Perlin p;
    p.scale = j.at("scale").number();
    perlin_tables(perlin_seed++, p);
    tex->v = p;
  } else if (t == "Image") {
    tex->v = fl.loadImage(j.at("file").string());
  } else {
    throw std::runtime_error("unknown texture type " + t);  // scala.MatchError in the reference
  }
  return tex;
}

static MaterialPtr read_material(const Json& j, const FileLoader& fl, uint64_t& perlin_seed) {
  const std::string& t = j.at("type").string();  // JsonFormats.scala:35-45
  TexturePtr tex = read_texture(j.at("texture"), fl, perlin_seed);
  auto m = std::make_shared<Material>();
  if (t == "Diffuse") m->v = Diffuse{tex};
  else if (t == "Glass") m->v = Glass{tex, j.at("ior").number()};
  else if (t == "Glossy") m->v = Glossy{tex, j.at("roughness").number()};
  else if (t == "Emission") m->v = Emission{tex, j.at("strength").number()};
  else throw std::runtime_error("unknown material type " + t);
  return m;
}

static void dump_world(const Scene& s);

std::optional<Scene> loadScene(const std::string& path, bool dump, uint64_t perlin_seed) {
  try {
    FileLoader fl;
    size_t k = path.find_last_of('/');
    fl.scene_dir = k == std::string::npos ? "." : path.substr(0, k);
    std::vector<uint8_t> bytes = read_bytes(path);
    Json json = Json::parse(std::string(bytes.begin(), bytes.end()));
    std::vector<std::pair<std::string, MaterialPtr>> materials;  // main.scala:57
    for (auto& kv : json.at("materials").obj) materials.emplace_back(kv.first, read_material(kv.second, fl, perlin_seed));
    auto material = [&](const std::string& name) -> MaterialPtr {
      for (auto& kv : materials) if (kv.first == name) return kv.second;
      throw std::runtime_error("key not found: " + name);  // materials(material), a NoSuchElementException
    };
    Scene s;
    const Json& c = json.at("camera");  // Json.format[Camera], JsonFormats.scala:10
    s.camera.origin = read_vec3(c.at("origin"));
    s.camera.target = read_vec3(c.at("target"));
    s.camera.vup = read_vec3(c.at("vup"));
    s.camera.vfov = c.at("vfov").number();
    s.camera.aspect_ratio = c.at("aspect_ratio").number();
    s.camera.aperture = c.at("aperture").number();
    s.camera.focus_distance = c.at("focus_distance").number();
    s.camera.background = read_vec3(c.at("background"));
    for (const Json& o : json.at("world").arr) {  // JsonFormats.scala:87-99
      const std::string& t = o.at("type").string();
      MaterialPtr m = material(o.at("material").string());
      if (t == "STL") {
        s.world.push_back(fromSTL(fl.loadFile(o.at("file").string()), m));
      } else if (t == "Triangle") {
        const Json& v = o.at("vertices");
        s.world.push_back(Triangle{read_vec3(v.at(0)), read_vec3(v.at(1)), read_vec3(v.at(2)), m});
      } else if (t == "Sphere") {
        s.world.push_back(Sphere{read_vec3(o.at("center")), o.at("radius").number(), m});
      } else if (t == "XYRect" || t == "XZRect" || t == "YZRect") {
        const Json& sd = o.at("sides");
        double a0 = sd.at(0).number(), a1 = sd.at(1).number(), b0 = sd.at(2).number(), b1 = sd.at(3).number();
        if (t == "XYRect") s.world.push_back(XYRect{a0, a1, b0, b1, o.at("z").number(), m});
        else if (t == "XZRect") s.world.push_back(XZRect{a0, a1, b0, b1, o.at("y").number(), m});
        else s.world.push_back(YZRect{a0, a1, b0, b1, o.at("x").number(), m});
      } else {
        throw std::runtime_error("unknown hittable type " + t);
      }
    }
    if (s.world.empty()) throw std::runtime_error("empty world");  // BVH(Seq()) fails in the reference as well
    if (dump) dump_world(s);
    return s;
  } catch (const std::exception& e) {  // main.scala:66-70: print, return None
    std::cerr << "loadScene: " << e.what() << std::endl;
    return std::nullopt;
  }
}

// --dump (main.scala:64 pretty-prints the world)
static void dump_texture(const Texture& t, int ind) {
  std::string pad((size_t)ind, ' ');
  if (auto* s = std::get_if<SolidColor>(&t.v)) std::printf("%sSolidColor(%g, %g, %g)\n", pad.c_str(), s->color.x, s->color.y, s->color.z);
  else if (auto* c = std::get_if<Checker>(&t.v)) {
    std::printf("%sChecker(size = %g)\n", pad.c_str(), c->size);
    dump_texture(*c->t1, ind + 2);
    dump_texture(*c->t2, ind + 2);
  } else if (auto* p = std::get_if<Perlin>(&t.v)) std::printf("%sPerlin(scale = %g)\n", pad.c_str(), p->scale);
  else if (auto* i = std::get_if<Image>(&t.v)) std::printf("%sImage(%d x %d)\n", pad.c_str(), i->width, i->height);
}
static void dump_material(const Material& m, int ind) {
  std::string pad((size_t)ind, ' ');
  if (auto* d = std::get_if<Diffuse>(&m.v)) { std::printf("%sDiffuse\n", pad.c_str()); dump_texture(*d->texture, ind + 2); }
  else if (auto* g = std::get_if<Glossy>(&m.v)) { std::printf("%sGlossy(roughness = %g)\n", pad.c_str(), g->roughness); dump_texture(*g->texture, ind + 2); }
  else if (auto* g2 = std::get_if<Glass>(&m.v)) { std::printf("%sGlass(ior = %g)\n", pad.c_str(), g2->ior); dump_texture(*g2->texture, ind + 2); }
  else if (auto* e = std::get_if<Emission>(&m.v)) { std::printf("%sEmission(strength = %g)\n", pad.c_str(), e->strength); dump_texture(*e->texture, ind + 2); }
}
static void dump_world(const Scene& s) {
  std::printf("World(%zu entries)\n", s.world.size());
  for (const Hittable& h : s.world) {
    if (auto* sp = std::get_if<Sphere>(&h)) {
      std::printf("  Sphere(center = (%g, %g, %g), radius = %g)\n", sp->center.x, sp->center.y, sp->center.z, sp->radius);
      dump_material(*sp->material, 4);
    } else if (auto* r = std::get_if<XYRect>(&h)) {
      std::printf("  XYRect(%g, %g, %g, %g, z = %g)\n", r->x0, r->x1, r->y0, r->y1, r->z);
      dump_material(*r->material, 4);
    } else if (auto* r2 = std::get_if<XZRect>(&h)) {
      std::printf("  XZRect(%g, %g, %g, %g, y = %g)\n", r2->x0, r2->x1, r2->z0, r2->z1, r2->y);
      dump_material(*r2->material, 4);
    } else if (auto* r3 = std::get_if<YZRect>(&h)) {
      std::printf("  YZRect(%g, %g, %g, %g, x = %g)\n", r3->y0, r3->y1, r3->z0, r3->z1, r3->x);
      dump_material(*r3->material, 4);
    } else if (auto* t = std::get_if<Triangle>(&h)) {
      std::printf("  Triangle((%g, %g, %g), (%g, %g, %g), (%g, %g, %g))\n", t->a.x, t->a.y, t->a.z, t->b.x, t->b.y,
                  t->b.z, t->c.x, t->c.y, t->c.z);
      dump_material(*t->material, 4);
    } else if (auto* st = std::get_if<STL>(&h)) {
      std::printf("  STL(%zu triangles)\n", st->triangles.size());
      if (!st->triangles.empty()) dump_material(*st->triangles[0].material, 4);
    }
  }
}

// ---- flattening ------------------------------------------------------------------------
int FlatScene::texture(const TexturePtr& t) {
  for (size_t i = 0; i < tex_seen_.size(); ++i) if (tex_seen_[i] == t.get()) return (int)i;
  int idx = (int)tex_type_.size();
  tex_seen_.push_back(t.get());
  tex_type_.push_back(0);
  tex_data_.insert(tex_data_.end(), HT_TEX_STRIDE, 0.0);
  tex_child_.push_back(-1);
  tex_child_.push_back(-1);
  if (auto* s = std::get_if<SolidColor>(&t->v)) {
    tex_type_[idx] = HT_TEX_SOLID;
    tex_data_[idx * HT_TEX_STRIDE] = s->color.x;
    tex_data_[idx * HT_TEX_STRIDE + 1] = s->color.y;
    tex_data_[idx * HT_TEX_STRIDE + 2] = s->color.z;
  } else if (auto* c = std::get_if<Checker>(&t->v)) {
    tex_type_[idx] = HT_TEX_CHECKER;
    tex_data_[idx * HT_TEX_STRIDE] = c->size;
    int a = texture(c->t1), b = texture(c->t2);
    tex_child_[idx * 2] = a;
    tex_child_[idx * 2 + 1] = b;
  } else if (auto* p = std::get_if<Perlin>(&t->v)) {
    tex_type_[idx] = HT_TEX_PERLIN;
    tex_data_[idx * HT_TEX_STRIDE] = p->scale;
    tex_child_[idx * 2] = (int)(ranvec_.size() / 768);
    ranvec_.insert(ranvec_.end(), p->ranvec.begin(), p->ranvec.end());
    perm_.insert(perm_.end(), p->perm.begin(), p->perm.end());
  } else if (auto* im = std::get_if<Image>(&t->v)) {
    tex_type_[idx] = HT_TEX_IMAGE;
    tex_child_[idx * 2] = (int)image_w_.size();
    image_w_.push_back(im->width);
    image_h_.push_back(im->height);
    image_off_.push_back((int64_t)texels_.size());
    texels_.insert(texels_.end(), im->pixels.begin(), im->pixels.end());
  }
  return idx;
}

int FlatScene::material(const MaterialPtr& m) {
  for (size_t i = 0; i < mat_seen_.size(); ++i) if (mat_seen_[i] == m.get()) return (int)i;
  mat_seen_.push_back(m.get());
  int kind = 0;
  double param = 0;
  TexturePtr tex;
  if (auto* d = std::get_if<Diffuse>(&m->v)) { kind = HT_MAT_DIFFUSE; tex = d->texture; }
  else if (auto* g = std::get_if<Glossy>(&m->v)) { kind = HT_MAT_GLOSSY; param = g->roughness; tex = g->texture; }
  else if (auto* g2 = std::get_if<Glass>(&m->v)) { kind = HT_MAT_GLASS; param = g2->ior; tex = g2->texture; }
  else if (auto* e = std::get_if<Emission>(&m->v)) { kind = HT_MAT_EMISSION; param = e->strength; tex = e->texture; }
  int t = texture(tex);
  mat_type_.push_back(kind);
  mat_texture_.push_back(t);
  mat_param_.push_back(param);
  return (int)mat_type_.size() - 1;
}

void FlatScene::add(int kind, std::initializer_list<double> data, const MaterialPtr& m, int group) {
  prim_type_.push_back(kind);
  prim_material_.push_back(material(m));
  prim_group_.push_back(group);
  size_t base = prim_data_.size();
  prim_data_.resize(base + HT_PRIM_STRIDE, 0.0);
  size_t i = 0;
  for (double v : data) prim_data_[base + i++] = v;
}

FlatScene::FlatScene(const std::vector<Hittable>& world) {
  int g = 0;
  for (const Hittable& h : world) {
    if (auto* s = std::get_if<Sphere>(&h)) add(HT_PRIM_SPHERE, {s->center.x, s->center.y, s->center.z, s->radius}, s->material, g);
    else if (auto* r = std::get_if<XYRect>(&h)) add(HT_PRIM_XYRECT, {r->x0, r->x1, r->y0, r->y1, r->z}, r->material, g);
    else if (auto* r2 = std::get_if<XZRect>(&h)) add(HT_PRIM_XZRECT, {r2->x0, r2->x1, r2->z0, r2->z1, r2->y}, r2->material, g);
    else if (auto* r3 = std::get_if<YZRect>(&h)) add(HT_PRIM_YZRECT, {r3->y0, r3->y1, r3->z0, r3->z1, r3->x}, r3->material, g);
    else if (auto* t = std::get_if<Triangle>(&h))
      add(HT_PRIM_TRIANGLE, {t->a.x, t->a.y, t->a.z, t->b.x, t->b.y, t->b.z, t->c.x, t->c.y, t->c.z}, t->material, g);
    else if (auto* st = std::get_if<STL>(&h))
      for (const Triangle& t2 : st->triangles)
        add(HT_PRIM_TRIANGLE, {t2.a.x, t2.a.y, t2.a.z, t2.b.x, t2.b.y, t2.b.z, t2.c.x, t2.c.y, t2.c.z}, t2.material, g);
    ++g;
  }
  if (texels_.empty()) texels_.push_back(0);
  desc_.n_prims = (int32_t)prim_type_.size();
  desc_.prim_type = prim_type_.data();
  desc_.prim_material = prim_material_.data();
  desc_.prim_data = prim_data_.data();
  desc_.prim_group = prim_group_.data();
  desc_.n_materials = (int32_t)mat_type_.size();
  desc_.mat_type = mat_type_.data();
  desc_.mat_texture = mat_texture_.data();
  desc_.mat_param = mat_param_.data();
  desc_.n_textures = (int32_t)tex_type_.size();
  desc_.tex_type = tex_type_.data();
  desc_.tex_data = tex_data_.data();
  desc_.tex_child = tex_child_.data();
  desc_.n_images = (int32_t)image_w_.size();
  desc_.image_width = image_w_.data();
  desc_.image_height = image_h_.data();
  desc_.image_offset = image_off_.data();
  desc_.image_texels = texels_.data();
  desc_.n_perlin = (int32_t)(ranvec_.size() / 768);
  desc_.perlin_ranvec = ranvec_.data();
  desc_.perlin_perm = perm_.data();
}

// ---- renderFrame -----------------------------------------------------------------------------
void renderFrame(const Camera& camera, const std::vector<Hittable>& world, BufferedImage& img, int samples,
                 const std::function<void(int)>& update, const std::function<void(double)>& finish,
                 const std::function<bool()>& shouldBreak, uint64_t seed, const std::vector<int>& devices) {
  const auto start = std::chrono::steady_clock::now();
  FlatScene flat(world);
  ht_scene* scene = nullptr;
  ht_frame* frame = nullptr;
  auto check = [&](int rc) {
    if (rc != HT_OK) {
      std::string msg = ht_last_error();
      if (frame) ht_frame_destroy(frame);
      if (scene) ht_scene_destroy(scene);
      throw std::runtime_error("libhiptrace: " + msg);
    }
  };
  if (devices.size() > 1) {
    // samples of the frame shared between several devices, one scene handle each
    std::vector<ht_scene*> scenes;
    auto drop = [&]() { for (ht_scene* s : scenes) ht_scene_destroy(s); };
    ht_render_params p{};
    p.width = img.getWidth();
    p.height = img.getHeight();
    p.samples = samples;
    p.max_depth = 50;
    p.jitter = 1;
    p.precision = HT_PRECISION_FP32;
    p.seed = seed;
    p.device = -1;
    const ht_camera cam = camera.to_c();
    int rc = HT_OK;
    for (int d : devices) {
      ht_scene* s = nullptr;
      rc = ht_scene_create(&flat.desc(), d, &s);
      if (rc != HT_OK) break;
      scenes.push_back(s);
    }
    if (rc == HT_OK && !shouldBreak())
      rc = ht_render_sharded(scenes.data(), (int32_t)scenes.size(), &cam, &p, nullptr, img.rgb.data(), nullptr);
    if (rc != HT_OK) {
      std::string msg = ht_last_error();
      drop();
      throw std::runtime_error("libhiptrace: " + msg);
    }
    drop();
    update(img.getHeight() - 1);
    finish((double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() / 1000.0);
    return;
  }
  check(ht_scene_create(&flat.desc(), devices.empty() ? -1 : devices[0], &scene));
  check(ht_frame_create(scene, img.getWidth(), img.getHeight(), &frame));
  const ht_camera cam = camera.to_c();
  const int batch = 64;  // shouldBreak granularity (the reference polls once per row, main.scala:80)
  for (int done = 0; done < samples; done += batch) {
    if (shouldBreak()) break;
    ht_render_params p{};
    p.width = img.getWidth();
    p.height = img.getHeight();
    p.samples = std::min(batch, samples - done);
    p.sample_offset = done;
    p.max_depth = 50;
    p.jitter = 1;
    p.precision = HT_PRECISION_FP32;
    p.seed = seed;
    p.device = -1;
    check(ht_frame_accumulate(frame, &cam, &p, nullptr));
    check(ht_frame_read(frame, nullptr, img.rgb.data()));  // the picture so far, for a caller that repaints
    // the frame advances everywhere at once: report the row the reference would
    // have finished after the same share of the work (update(j), main.scala:95)
    update(std::max(0, (int)((int64_t)img.getHeight() * (done + p.samples) / samples) - 1));
  }
  ht_frame_destroy(frame);
  ht_scene_destroy(scene);
  // whole milliseconds, as System.currentTimeMillis gives them (main.scala:77, 97-98)
  finish((double)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() / 1000.0);
}

}  // namespace rt
