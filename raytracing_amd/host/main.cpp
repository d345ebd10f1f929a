// main.cpp — command-line front-end: the reference's `cmd` mode
// (jvm/src/main/scala/main.scala:20-30 options, 106-119 cmd, 101-104 formatDuration)
// on top of libhiptrace.
//
//   hiptrace-render -o out.png [--width 400] [--height 225] [--samples 32] [--dump] scene.json
//
// Without -o the reference opens its Swing window; this program has no window
// and says so.  Two developer modes serve the host tests and render nothing:
//   --tool-decode <image> <out.bin>     width, height (int32 LE) then 0x00RRGGBB words
//   --tool-flatten <scene.json> <out.json>   the arrays handed to ht_scene_create
//   --tool-recode <image> <out.png>     decode, then write with the PNG encoder
//   --tool-duration <seconds>           formatDuration of the argument
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "scene.hpp"

namespace {

struct Options {
  std::string filename, scene;
  bool has_filename = false, has_scene = false, dump = false;
  int width = 400, height = 225, samples = 32;
  uint64_t seed = 0;
  std::vector<int> devices;  // --devices 0,1,..: HIP devices that share the frame's samples
};

[[noreturn]] void usage_error(const std::string& msg) {
  std::fprintf(stderr, "[hiptrace-render] Error: %s\n", msg.c_str());
  std::exit(1);
}

int to_int(const std::string& name, const std::string& v) {
  try {
    size_t used = 0;
    long x = std::stol(v, &used);
    if (used != v.size()) throw std::invalid_argument(v);
    return (int)x;
  } catch (const std::exception&) {
    usage_error("Bad arguments for option '" + name + "': '" + v + "' - bad Int value");
  }
}

Options parse(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], val;
    bool has_val = false;
    if (a.rfind("--", 0) == 0) {
      size_t eq = a.find('=');
      if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
    }
    auto value = [&]() -> std::string {
      if (has_val) return val;
      if (i + 1 >= argc) usage_error("Bad arguments for option '" + a + "': '' - wrong arguments format");
      return argv[++i];
    };
    if (a == "-o" || a == "--filename") { o.filename = value(); o.has_filename = true; }
    else if (a == "-w" || a == "--width") o.width = to_int("width", value());
    else if (a == "-h" || a == "--height") o.height = to_int("height", value());
    else if (a == "-s" || a == "--samples") o.samples = to_int("samples", value());
    else if (a == "-d" || a == "--dump") o.dump = true;
    else if (a == "--seed") o.seed = std::stoull(value());
    else if (a == "--devices") {
      std::string list = value();
      size_t at = 0;
      while (at <= list.size()) {
        size_t comma = list.find(',', at);
        if (comma == std::string::npos) comma = list.size();
        o.devices.push_back(to_int("devices", list.substr(at, comma - at)));
        at = comma + 1;
      }
    }
    else if (a == "--help") {
      std::printf("  -d, --dump\n  -o, --filename  <arg>\n  -h, --height  <arg>\n  -s, --samples  <arg>\n"
                  "  -w, --width  <arg>\n      --seed  <arg>\n      --devices  <d0,d1,..>\n      --help   Show help message\n\n"
                  " trailing arguments:\n  scene (not required)\n");
      std::exit(0);
    } else if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) usage_error("Unknown option '" + a.substr(a.find_first_not_of('-')) + "'");
    else if (!o.has_scene) { o.scene = a; o.has_scene = true; }
    else usage_error("Excess arguments provided: '" + a + "'");
  }
  // dependsOnAll(filename, List(scene)), main.scala:28
  if (o.has_filename && !o.has_scene) usage_error("When specifying 'filename', at least one of the following options must be provided: scene");
  return o;
}

// java.util.Formatter's %.<d>f: shortest decimal that round-trips the double,
// then HALF_UP to d places (printf would round the exact binary value to even)
std::string java_fixed(double v, int decimals) {
  if (!std::isfinite(v)) return std::isnan(v) ? "NaN" : (v < 0 ? "-Infinity" : "Infinity");
  char buf[64];
  int prec = 1;
  for (; prec <= 17; ++prec) {
    std::snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
    if (std::strtod(buf, nullptr) == v) break;
  }
  // buf = [-]d.ddddde[+-]xx
  std::string s = buf;
  bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  size_t e = s.find('e');
  int exp10 = std::atoi(s.c_str() + e + 1);
  std::string digits;
  for (size_t i = 0; i < e; ++i) if (s[i] != '.') digits += s[i];
  // value = 0.digits * 10^(exp10+1)
  int point = exp10 + 1;  // digits before the decimal point (may be <= 0)
  std::string ip, fp;
  if (point <= 0) { ip = "0"; fp = std::string((size_t)(-point), '0') + digits; }
  else if ((size_t)point >= digits.size()) { ip = digits + std::string((size_t)point - digits.size(), '0'); }
  else { ip = digits.substr(0, (size_t)point); fp = digits.substr((size_t)point); }
  bool up = fp.size() > (size_t)decimals && fp[(size_t)decimals] >= '5';
  fp.resize((size_t)decimals, '0');
  if (up) {
    std::string all = ip + fp;
    int k = (int)all.size() - 1;
    while (k >= 0 && all[(size_t)k] == '9') all[(size_t)k--] = '0';
    if (k >= 0) ++all[(size_t)k]; else all.insert(all.begin(), '1');
    ip = all.substr(0, all.size() - (size_t)decimals);
    fp = all.substr(all.size() - (size_t)decimals);
  }
  bool zero = ip.find_first_not_of('0') == std::string::npos && fp.find_first_not_of('0') == std::string::npos;
  std::string out = (neg && !zero ? "-" : "") + ip;
  if (decimals > 0) out += "." + fp;
  return out;
}

std::string zero_pad(std::string s, size_t width) {
  while (s.size() < width) s.insert(s.begin(), '0');
  return s;
}

}  // namespace

// formatDuration (main.scala:101-104), rounding quirks included: minutes are the
// ROUNDED value of (t mod 3600)/60, so 90 s reads "02m:30.00s"
std::string formatDuration(double t) {
  std::string m = zero_pad(java_fixed(std::fmod(t, 3600.0) / 60.0, 0), 2) + "m:" + zero_pad(java_fixed(std::fmod(t, 60.0), 2), 5) + "s";
  if (t >= 3600) return zero_pad(java_fixed(t / 3600.0, 0), 2) + "h:" + m;
  return m;
}

static int tool_decode(const char* in, const char* out) {
  std::ifstream f(in, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", in); return 1; }
  std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  rt::Image im = (b.size() > 1 && b[0] == 0x89) ? rt::decodePNG(b) : rt::decodeJPEG(b);
  std::ofstream o(out, std::ios::binary);
  int32_t wh[2] = {im.width, im.height};
  o.write((const char*)wh, sizeof wh);
  o.write((const char*)im.pixels.data(), (std::streamsize)(im.pixels.size() * 4));
  return o ? 0 : 1;
}

static int tool_recode(const char* in, const char* out) {
  std::ifstream f(in, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", in); return 1; }
  std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  rt::Image im = (b.size() > 1 && b[0] == 0x89) ? rt::decodePNG(b) : rt::decodeJPEG(b);
  rt::BufferedImage img(im.width, im.height);
  img.rgb = im.pixels;
  std::vector<uint8_t> png = rt::encodePNG(img);
  std::ofstream o(out, std::ios::binary);
  o.write((const char*)png.data(), (std::streamsize)png.size());
  return o ? 0 : 1;
}

template <typename T>
static void put_array(std::ostream& o, const char* name, const T* p, size_t n, bool last = false) {
  o << "\"" << name << "\": [";
  char buf[40];
  for (size_t i = 0; i < n; ++i) {
    if constexpr (std::is_floating_point<T>::value) std::snprintf(buf, sizeof buf, "%.17g", (double)p[i]);
    else std::snprintf(buf, sizeof buf, "%lld", (long long)p[i]);
    o << (i ? "," : "") << buf;
  }
  o << "]" << (last ? "\n" : ",\n");
}

static int tool_flatten(const char* scene_path, const char* out) {
  auto s = rt::loadScene(scene_path);
  if (!s) return 1;
  rt::FlatScene flat(s->world);
  const ht_scene_desc& d = flat.desc();
  ht_camera c = s->camera.to_c();
  std::ofstream o(out);
  o << "{\n";
  put_array(o, "camera", (const double*)&c, sizeof c / sizeof(double));
  put_array(o, "prim_type", d.prim_type, (size_t)d.n_prims);
  put_array(o, "prim_material", d.prim_material, (size_t)d.n_prims);
  put_array(o, "prim_group", d.prim_group, (size_t)d.n_prims);
  put_array(o, "prim_data", d.prim_data, (size_t)d.n_prims * HT_PRIM_STRIDE);
  put_array(o, "mat_type", d.mat_type, (size_t)d.n_materials);
  put_array(o, "mat_texture", d.mat_texture, (size_t)d.n_materials);
  put_array(o, "mat_param", d.mat_param, (size_t)d.n_materials);
  put_array(o, "tex_type", d.tex_type, (size_t)d.n_textures);
  put_array(o, "tex_data", d.tex_data, (size_t)d.n_textures * HT_TEX_STRIDE);
  put_array(o, "tex_child", d.tex_child, (size_t)d.n_textures * 2);
  put_array(o, "image_width", d.image_width, (size_t)d.n_images);
  put_array(o, "image_height", d.image_height, (size_t)d.n_images);
  put_array(o, "image_offset", d.image_offset, (size_t)d.n_images);
  uint64_t texel_sum = 0;
  size_t n_texels = 0;
  for (int i = 0; i < d.n_images; ++i) n_texels = std::max(n_texels, (size_t)d.image_offset[i] + (size_t)d.image_width[i] * d.image_height[i]);
  for (size_t i = 0; i < n_texels; ++i) texel_sum = texel_sum * 1000003u + d.image_texels[i];
  o << "\"n_texels\": " << n_texels << ",\n\"texel_hash\": " << texel_sum << ",\n";
  put_array(o, "perlin_ranvec", d.perlin_ranvec, (size_t)d.n_perlin * 768);
  put_array(o, "perlin_perm", d.perlin_perm, (size_t)d.n_perlin * 768, true);
  o << "}\n";
  return o ? 0 : 1;
}

int main(int argc, char** argv) {
  try {
    if (argc == 4 && !std::strcmp(argv[1], "--tool-decode")) return tool_decode(argv[2], argv[3]);
    if (argc == 4 && !std::strcmp(argv[1], "--tool-flatten")) return tool_flatten(argv[2], argv[3]);
    if (argc == 4 && !std::strcmp(argv[1], "--tool-recode")) return tool_recode(argv[2], argv[3]);
    if (argc == 3 && !std::strcmp(argv[1], "--tool-duration")) { std::printf("%s\n", formatDuration(std::atof(argv[2])).c_str()); return 0; }
    Options o = parse(argc, argv);
    if (!o.has_filename) {
      std::fprintf(stderr, "hiptrace-render has no window: give -o <file.png> and a scene (the reference's GUI mode is not part of this program)\n");
      return 2;
    }
    auto scene = rt::loadScene(o.scene, o.dump);  // main.scala:107
    if (!scene) {
      std::printf("Error loading scene %s\n", o.scene.c_str());
      return 1;
    }
    rt::BufferedImage img(o.width, o.height);
    rt::renderFrame(
        scene->camera, scene->world, img, o.samples,
        [&](int line) { std::printf("\rRendered line [%d/%d]", line + 1, o.height); std::fflush(stdout); },
        [&](double t) { std::printf("\nTime: %s\n", formatDuration(t).c_str()); },
        [] { return false; }, o.seed, o.devices);
    std::vector<uint8_t> png = rt::encodePNG(img);
    std::ofstream f(o.filename, std::ios::binary);
    f.write((const char*)png.data(), (std::streamsize)png.size());
    if (!f) { std::fprintf(stderr, "cannot write %s\n", o.filename.c_str()); return 1; }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
