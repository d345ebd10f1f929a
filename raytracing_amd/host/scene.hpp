// scene.hpp — host-side scene model of the C++ front-end: the reference's case
// classes as plain structs, its scene reader and its renderFrame, on top of the C
// ABI (include/hiptrace.h).  Nothing in here traces rays.
//
// Reference (Ivoah/RayTracing): shared/src/main/scala/{Camera,Hittable,Material,
// Texture}.scala for the classes, shared/src/main/scala/JsonFormats.scala:5-99 for
// the file schema, jvm/src/main/scala/main.scala:32-99 for loadScene / renderFrame,
// jvm/src/main/scala/FileLoader.scala:8-21 for the file rules.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/hiptrace.h"

namespace rt {

struct Vec3 { double x = 0, y = 0, z = 0; };

// ---- textures (Texture.scala:12-107) ---------------------------------------
struct Texture;
using TexturePtr = std::shared_ptr<const Texture>;
struct SolidColor { Vec3 color; };
struct Checker { double size; TexturePtr t1, t2; };
struct Perlin {
  double scale;
  std::vector<double> ranvec;   // 256 * 3 unit vectors   (Texture.scala:38)
  std::vector<int32_t> perm;    // perm_x, perm_y, perm_z (Texture.scala:40-42)
};
struct Image {
  int width, height;
  std::vector<uint32_t> pixels;  // 0x00RRGGBB, row 0 = top (what BufferedImage.getRGB returns)
};
struct Texture { std::variant<SolidColor, Checker, Perlin, Image> v; };

// ---- materials (Material.scala:9-55) ----------------------------------------
struct Diffuse { TexturePtr texture; };
struct Glossy { TexturePtr texture; double roughness; };
struct Glass { TexturePtr texture; double ior; };
struct Emission { TexturePtr texture; double strength; };
struct Material { std::variant<Diffuse, Glossy, Glass, Emission> v; };
using MaterialPtr = std::shared_ptr<const Material>;

// ---- primitives (Hittable.scala:25-115) ---------------------------------------
struct Sphere { Vec3 center; double radius; MaterialPtr material; };
struct XYRect { double x0, x1, y0, y1, z; MaterialPtr material; };
struct XZRect { double x0, x1, z0, z1, y; MaterialPtr material; };
struct YZRect { double y0, y1, z0, z1, x; MaterialPtr material; };
struct Triangle { Vec3 a, b, c; MaterialPtr material; };
struct STL { std::vector<Triangle> triangles; };  // BVH.fromSTL's facets in file order
using Hittable = std::variant<Sphere, XYRect, XZRect, YZRect, Triangle, STL>;

// ---- camera (Camera.scala:3-4) ----------------------------------------------------
struct Camera {
  Vec3 origin, target, vup;
  double vfov, aspect_ratio, aperture, focus_distance;
  Vec3 background;
  ht_camera to_c() const;
};

// ---- file rules (jvm FileLoader.scala:8-21) -----------------------------------------
struct FileLoader {
  std::string scene_dir;  // System.setProperty("user.dir", ...) of main.scala:34
  std::string file(const std::string& path) const;   // absolute, or scene_dir / basename
  Image loadImage(const std::string& path) const;    // PNG and baseline JPEG
  std::string loadFile(const std::string& path) const;
};

// BVH.fromSTL (BVH.scala:35-55): ASCII facets by the reference's own pattern.
STL fromSTL(const std::string& source, MaterialPtr material);

struct Scene {
  Camera camera;
  std::vector<Hittable> world;
};

// loadScene (main.scala:32-71) for the JSON format.  Like the reference it never
// throws: any error is reported on stderr and the result is empty.
std::optional<Scene> loadScene(const std::string& path, bool dump = false, uint64_t perlin_seed = 1);

// The world as the arrays of ht_scene_desc (keeps its own storage alive).
class FlatScene {
 public:
  explicit FlatScene(const std::vector<Hittable>& world);
  const ht_scene_desc& desc() const { return desc_; }
  int n_prims() const { return (int)prim_type_.size(); }

 private:
  int material(const MaterialPtr& m);
  int texture(const TexturePtr& t);
  void add(int kind, std::initializer_list<double> data, const MaterialPtr& m, int group);
  std::vector<int32_t> prim_type_, prim_material_, prim_group_;
  std::vector<double> prim_data_;
  std::vector<int32_t> mat_type_, mat_texture_;
  std::vector<double> mat_param_;
  std::vector<int32_t> tex_type_, tex_child_;
  std::vector<double> tex_data_;
  std::vector<int32_t> image_w_, image_h_;
  std::vector<int64_t> image_off_;
  std::vector<uint32_t> texels_;
  std::vector<double> ranvec_;
  std::vector<int32_t> perm_;
  std::vector<const Material*> mat_seen_;
  std::vector<const Texture*> tex_seen_;
  ht_scene_desc desc_{};
};

// The picture renderFrame fills: TYPE_INT_RGB BufferedImage layout, 0x00RRGGBB,
// row 0 = top (main.scala:93, 109).
struct BufferedImage {
  int width, height;
  std::vector<uint32_t> rgb;
  BufferedImage(int w, int h) : width(w), height(h), rgb((size_t)w * h, 0) {}
  int getWidth() const { return width; }
  int getHeight() const { return height; }
};

// renderFrame (main.scala:73-99).  The device renders every row at once, in
// batches of samples: after each batch `img` holds the picture so far and `update`
// gets the index of the row the reference would have finished after the same share
// of the work (height-1 at the end); `finish` gets the seconds taken; `shouldBreak`
// is polled between batches.  `devices`: HIP device ordinals to share the samples of
// the frame between (ht_render_sharded; one whole-frame call, so `update` comes once);
// empty or one entry: the progressive single-device path.  Throws std::runtime_error with the library's message
// if the device path fails (there is no CPU path to fall back to).
void renderFrame(const Camera& camera, const std::vector<Hittable>& world, BufferedImage& img, int samples,
                 const std::function<void(int)>& update = [](int) {},
                 const std::function<void(double)>& finish = [](double) {},
                 const std::function<bool()>& shouldBreak = [] { return false; }, uint64_t seed = 0,
                 const std::vector<int>& devices = {});

// image codecs of the front-end
Image decodePNG(const std::vector<uint8_t>& bytes);
Image decodeJPEG(const std::vector<uint8_t>& bytes);
std::vector<uint8_t> encodePNG(const BufferedImage& img);

}  // namespace rt
