// codecs.cpp — the image formats the front-end touches: PNG in and out, baseline
// JPEG in.  The reference leaves these to javax.imageio (jvm FileLoader.scala:14,
// main.scala:114); a JVM-less host needs its own.
//
// The JPEG decoder follows the arithmetic of the IJG library that ImageIO wraps
// (accurate-integer inverse DCT with 13-bit constants, triangle-filter chroma
// upsampling, 16-bit fixed-point YCbCr conversion), so texels come out equal to
// what BufferedImage.getRGB hands the reference.
#include <zlib.h>

#include <cstring>
#include <stdexcept>

#include "scene.hpp"

namespace rt {

// A texture file is untrusted input: sizes are capped before anything is allocated from them.
static const uint32_t kMaxImageSide = 65535;
static const uint64_t kMaxImagePixels = 1ull << 28;

// =============================== PNG ============================================
static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Image decodePNG(const std::vector<uint8_t>& bytes) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (bytes.size() < 8 || std::memcmp(bytes.data(), sig, 8) != 0) throw std::runtime_error("PNG: bad signature");
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> palette, idat;
  size_t pos = 8;
  bool end = false;
  while (!end) {
    if (pos + 12 > bytes.size()) throw std::runtime_error("PNG: truncated");
    uint32_t len = be32(&bytes[pos]);
    const uint8_t* type = &bytes[pos + 4];
    if (pos + 12 + (size_t)len > bytes.size()) throw std::runtime_error("PNG: truncated chunk");
    const uint8_t* data = &bytes[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) throw std::runtime_error("PNG: short IHDR");
      width = be32(data);
      height = be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
    } else if (!std::memcmp(type, "PLTE", 4)) {
      palette.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || width == 0 || height == 0) throw std::runtime_error("PNG: no IHDR");
  if (width > kMaxImageSide || height > kMaxImageSide || (uint64_t)width * height > kMaxImagePixels)
    throw std::runtime_error("PNG: picture too large");
  if (interlace) throw std::runtime_error("PNG: interlaced files are not supported");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("PNG: bad colour type");
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
    throw std::runtime_error("PNG: bad bit depth");
  if (ctype == 3 && depth == 16) throw std::runtime_error("PNG: bad bit depth");
  const size_t bpp = (size_t)std::max(1, channels * depth / 8);
  const size_t stride = ((size_t)width * channels * depth + 7) / 8;
  std::vector<uint8_t> raw((stride + 1) * (size_t)height);
  uLongf out_len = (uLongf)raw.size();
  int rc = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
  if (rc != Z_OK || out_len != raw.size()) throw std::runtime_error("PNG: inflate failed");
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  Image img;
  img.width = (int)width;
  img.height = (int)height;
  img.pixels.resize((size_t)width * height);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t* row = &raw[(stride + 1) * y];
    int filter = row[0];
    for (size_t i = 0; i < stride; ++i) {
      int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = row[1 + i];
      switch (filter) {
        case 0: break;
        case 1: x += a; break;
        case 2: x += b; break;
        case 3: x += (a + b) >> 1; break;
        case 4: x += paeth(a, b, c); break;
        default: throw std::runtime_error("PNG: bad filter");
      }
      cur[i] = (uint8_t)x;
    }
    for (uint32_t xx = 0; xx < width; ++xx) {
      auto sample = [&](int ch) -> int {  // 8-bit value of channel ch
        if (depth == 8) return cur[(size_t)xx * channels + ch];
        if (depth == 16) return cur[((size_t)xx * channels + ch) * 2];
        size_t bit = (size_t)xx * depth;
        int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
      };
      int r, g, b;
      if (ctype == 3) {
        size_t k = (size_t)sample(0) * 3;
        if (k + 3 > palette.size()) throw std::runtime_error("PNG: palette index out of range");
        r = palette[k]; g = palette[k + 1]; b = palette[k + 2];
      } else if (ctype == 0 || ctype == 4) {
        r = g = b = sample(0);
      } else {
        r = sample(0); g = sample(1); b = sample(2);
      }
      img.pixels[(size_t)y * width + xx] = (uint32_t)r << 16 | (uint32_t)g << 8 | (uint32_t)b;
    }
    prev.swap(cur);
  }
  return img;
}

static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put_be32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
}

// 8-bit RGB, what ImageIO.write(img, "png", ...) produces for TYPE_INT_RGB
std::vector<uint8_t> encodePNG(const BufferedImage& img) {
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)img.width);
  put_be32(ihdr, (uint32_t)img.height);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
  put_chunk(out, "IHDR", ihdr);
  const size_t stride = (size_t)img.width * 3 + 1;
  std::vector<uint8_t> raw(stride * img.height);
  for (int y = 0; y < img.height; ++y) {
    uint8_t* row = &raw[stride * y];
    row[0] = 0;
    for (int x = 0; x < img.width; ++x) {
      uint32_t p = img.rgb[(size_t)y * img.width + x];
      row[1 + x * 3] = (uint8_t)(p >> 16);
      row[2 + x * 3] = (uint8_t)(p >> 8);
      row[3 + x * 3] = (uint8_t)p;
    }
  }
  uLongf bound = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(bound);
  if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("PNG: deflate failed");
  z.resize(bound);
  put_chunk(out, "IDAT", z);
  put_chunk(out, "IEND", {});
  return out;
}

// =============================== JPEG ==========================================
namespace {

struct Huff {
  // canonical code tables: for each length the first code, and where its symbols start
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  uint8_t look_len[512];  // 9-bit prefix lookup: code length or 0
  uint8_t look_sym[512];
  bool present = false;
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocks_w = 0, blocks_h = 0;  // in blocks, padded to whole MCUs
  int width = 0, height = 0;       // true downsampled size
  int pred = 0;
  std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  void fill() {
    while (n <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;
          } else {
            hit_marker = true;  // leave the marker for the segment parser; feed zeros
            b = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  int peek(int k) {
    if (n < k) fill();
    return (int)(acc >> (32 - k));
  }
  void skip(int k) { acc <<= k; n -= k; }
  int get(int k) {
    if (k == 0) return 0;
    int v = peek(k);
    skip(k);
    return v;
  }
  void reset() { acc = 0; n = 0; hit_marker = false; }
};

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void build_huff(Huff& h, const uint8_t* bits /*[1..16]*/, const uint8_t* vals, int nvals) {
  if (nvals < 0 || nvals > 256) throw std::runtime_error("JPEG: bad DHT");
  std::memcpy(h.vals, vals, (size_t)nvals);
  int code = 0, k = 0;
  std::memset(h.look_len, 0, sizeof h.look_len);
  for (int len = 1; len <= 16; ++len) {
    h.valptr[len] = k;
    h.mincode[len] = code;
    // the code lengths must form a prefix code: no more codes of a length than that length has left
    if (code + (int)bits[len] > (1 << len) || k + (int)bits[len] > nvals)
      throw std::runtime_error("JPEG: DHT code lengths are not a prefix code");
    for (int i = 0; i < bits[len]; ++i, ++k, ++code) {
      if (len <= 9) {
        int first = code << (9 - len);
        for (int j = 0; j < (1 << (9 - len)); ++j) {
          h.look_len[first + j] = (uint8_t)len;
          h.look_sym[first + j] = vals[k];
        }
      }
    }
    h.maxcode[len] = bits[len] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7FFFFFFF;
  h.present = true;
}

int decode_symbol(BitReader& br, const Huff& h) {
  int look = br.peek(9);
  if (h.look_len[look]) {
    br.skip(h.look_len[look]);
    return h.look_sym[look];
  }
  int code = br.peek(16);
  for (int len = 10; len <= 16; ++len) {
    int c = code >> (16 - len);
    if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) {
      br.skip(len);
      return h.vals[h.valptr[len] + c - h.mincode[len]];
    }
  }
  throw std::runtime_error("JPEG: bad Huffman code");
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// accurate integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants,
// two extra bits kept between the passes)
void idct_islow(const int* coef /*dequantised, natural order*/, uint8_t* out, int out_stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                 F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* in = coef + c;
    long z2 = in[16], z3 = in[48];
    long z1 = (z2 + z3) * F0_541;
    long t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
    z2 = in[0]; z3 = in[32];
    long t0 = (z2 + z3) * (1L << CB), t1 = (z2 - z3) * (1L << CB);
    long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = in[56]; t1 = in[40]; t2 = in[24]; t3 = in[8];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    long z4 = t1 + t3, z5 = (z3 + z4) * F1_175;
    t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    long* w = ws + c;
    w[0] = descale(t10 + t3, CB - P1);  w[56] = descale(t10 - t3, CB - P1);
    w[8] = descale(t11 + t2, CB - P1);  w[48] = descale(t11 - t2, CB - P1);
    w[16] = descale(t12 + t1, CB - P1); w[40] = descale(t12 - t1, CB - P1);
    w[24] = descale(t13 + t0, CB - P1); w[32] = descale(t13 - t0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const long* w = ws + r * 8;
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0_541;
    long t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
    long t0 = (w[0] + w[4]) * (1L << CB), t1 = (w[0] - w[4]) * (1L << CB);
    long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    long z4 = t1 + t3, z5 = (z3 + z4) * F1_175;
    t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    uint8_t* o = out + r * out_stride;
    constexpr int S = CB + P1 + 3;
    o[0] = clamp8((int)descale(t10 + t3, S) + 128); o[7] = clamp8((int)descale(t10 - t3, S) + 128);
    o[1] = clamp8((int)descale(t11 + t2, S) + 128); o[6] = clamp8((int)descale(t11 - t2, S) + 128);
    o[2] = clamp8((int)descale(t12 + t1, S) + 128); o[5] = clamp8((int)descale(t12 - t1, S) + 128);
    o[3] = clamp8((int)descale(t13 + t0, S) + 128); o[4] = clamp8((int)descale(t13 - t0, S) + 128);
  }
}

// chroma upsampling to full resolution: triangle filter for 2:1 horizontal and
// 2:1 x 2:1, pixel replication for every other ratio
std::vector<uint8_t> upsample(const Component& c, int hmax, int vmax, int W, int H) {
  const int stride = c.blocks_w * 8;
  std::vector<uint8_t> out((size_t)W * H);
  const int hr = hmax / c.h, vr = vmax / c.v;
  const int cw = c.width, ch = c.height;
  auto src = [&](int y) { return &c.plane[(size_t)std::min(std::max(y, 0), ch - 1) * stride]; };
  if (hr == 1 && vr == 1) {
    for (int y = 0; y < H; ++y) std::memcpy(&out[(size_t)y * W], src(y), (size_t)W);
  } else if (hr == 2 && vr == 1) {
    std::vector<uint8_t> line((size_t)cw * 2);
    for (int y = 0; y < H; ++y) {
      const uint8_t* in = src(y);
      if (cw == 1) {
        line[0] = line[1] = in[0];
      } else {
        line[0] = in[0];
        line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < cw - 1; ++i) {
          line[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
          line[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
        }
        line[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
        line[2 * cw - 1] = in[cw - 1];
      }
      std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
    }
  } else if (hr == 2 && vr == 2) {
    std::vector<uint8_t> line((size_t)cw * 2);
    for (int y = 0; y < H; ++y) {
      const uint8_t* in0 = src(y >> 1);                        // nearer row, weight 3
      const uint8_t* in1 = src((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1);  // farther row, weight 1
      auto colsum = [&](int i) { return in0[i] * 3 + in1[i]; };
      if (cw == 1) {
        line[0] = (uint8_t)((colsum(0) * 4 + 8) >> 4);
        line[1] = (uint8_t)((colsum(0) * 4 + 7) >> 4);
      } else {
        int last = colsum(0), cur = colsum(0), next = colsum(1);
        line[0] = (uint8_t)((cur * 4 + 8) >> 4);
        line[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        for (int i = 1; i < cw - 1; ++i) {
          last = cur; cur = next; next = colsum(i + 1);
          line[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4);
          line[2 * i + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        }
        last = cur; cur = next;
        line[2 * cw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4);
        line[2 * cw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
      }
      std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
    }
  } else {
    for (int y = 0; y < H; ++y) {
      const uint8_t* in = src(y / vr);
      for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = in[std::min(x / hr, cw - 1)];
    }
  }
  return out;
}

}  // namespace

Image decodeJPEG(const std::vector<uint8_t>& bytes) {
  const uint8_t* p = bytes.data();
  const uint8_t* end = p + bytes.size();
  if (bytes.size() < 4 || p[0] != 0xFF || p[1] != 0xD8) throw std::runtime_error("JPEG: no SOI");
  p += 2;
  uint16_t quant[4][64] = {};
  Huff dc[4], ac[4];
  std::vector<Component> comps;
  int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
  bool adobe = false;
  int adobe_transform = -1;
  bool done = false, decoded = false;
  while (!done) {
    while (p < end && *p != 0xFF) ++p;  // resynchronise
    while (p < end && *p == 0xFF) ++p;
    if (p >= end) break;
    int marker = *p++;
    if (marker == 0xD9) break;
    if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
    if (p + 2 > end) throw std::runtime_error("JPEG: truncated");
    int len = p[0] << 8 | p[1];
    if (len < 2 || p + len > end) throw std::runtime_error("JPEG: bad segment length");
    const uint8_t* seg = p + 2;
    const uint8_t* seg_end = p + len;
    p += len;
    switch (marker) {
      case 0xDB:  // DQT
        while (seg < seg_end) {
          int pq = seg[0] >> 4, tq = seg[0] & 15;
          ++seg;
          if (tq > 3 || seg + (pq ? 128 : 64) > seg_end) throw std::runtime_error("JPEG: bad DQT");
          for (int i = 0; i < 64; ++i) {
            quant[tq][kZigzag[i]] = pq ? (uint16_t)(seg[0] << 8 | seg[1]) : seg[0];
            seg += pq ? 2 : 1;
          }
        }
        break;
      case 0xC4:  // DHT
        while (seg < seg_end) {
          int tc = seg[0] >> 4, th = seg[0] & 15;
          if (tc > 1 || th > 3 || seg + 17 > seg_end) throw std::runtime_error("JPEG: bad DHT");
          uint8_t bits[17] = {0};
          int n = 0;
          for (int i = 1; i <= 16; ++i) { bits[i] = seg[i]; n += seg[i]; }
          if (n > 256 || seg + 17 + n > seg_end) throw std::runtime_error("JPEG: bad DHT");
          build_huff(tc ? ac[th] : dc[th], bits, seg + 17, n);
          seg += 17 + n;
        }
        break;
      case 0xC0:
      case 0xC1: {  // baseline / extended sequential, Huffman
        if (len < 8) throw std::runtime_error("JPEG: bad SOF");
        if (seg[0] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
        H = seg[1] << 8 | seg[2];
        W = seg[3] << 8 | seg[4];
        int n = seg[5];
        if (W <= 0 || H <= 0 || (n != 1 && n != 3) || len < 8 + 3 * n) throw std::runtime_error("JPEG: bad SOF");
        comps.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
          comps[i].id = seg[6 + 3 * i];
          comps[i].h = seg[7 + 3 * i] >> 4;
          comps[i].v = seg[7 + 3 * i] & 15;
          comps[i].tq = seg[8 + 3 * i];
          if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3)
            throw std::runtime_error("JPEG: bad component");
          hmax = std::max(hmax, comps[i].h);
          vmax = std::max(vmax, comps[i].v);
        }
        // every component must divide the largest sampling factors (the upsampler replicates or
        // filters by whole ratios), and the picture must be of a size worth allocating
        for (int i = 0; i < n; ++i)
          if (hmax % comps[i].h != 0 || vmax % comps[i].v != 0)
            throw std::runtime_error("JPEG: fractional sampling ratios are not supported");
        if ((uint64_t)W * (uint64_t)H > kMaxImagePixels) throw std::runtime_error("JPEG: picture too large");
        break;
      }
      case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
        throw std::runtime_error("JPEG: only baseline / sequential Huffman files are supported");
      case 0xDD:
        if (len >= 4) restart_interval = seg[0] << 8 | seg[1];
        break;
      case 0xEE:
        if (len >= 14 && !std::memcmp(seg, "Adobe", 5)) { adobe = true; adobe_transform = seg[11]; }
        break;
      case 0xDA: {  // SOS: one interleaved scan of all components
        if (comps.empty()) throw std::runtime_error("JPEG: SOS before SOF");
        if (len < 3) throw std::runtime_error("JPEG: bad SOS");
        int ns = seg[0];
        if (ns != (int)comps.size()) throw std::runtime_error("JPEG: non-interleaved scans are not supported");
        if (len < 2 + 1 + 2 * ns + 3) throw std::runtime_error("JPEG: bad SOS");
        for (int i = 0; i < ns; ++i) {
          int cid = seg[1 + 2 * i];
          Component* c = nullptr;
          for (auto& cc : comps) if (cc.id == cid) c = &cc;
          if (!c) throw std::runtime_error("JPEG: unknown scan component");
          c->td = seg[2 + 2 * i] >> 4;
          c->ta = seg[2 + 2 * i] & 15;
          if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present) throw std::runtime_error("JPEG: missing Huffman table");
        }
        const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
        const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
        for (auto& c : comps) {
          if (comps.size() == 1) { c.h = c.v = 1; }
          c.blocks_w = mcus_x * c.h;
          c.blocks_h = mcus_y * c.v;
          c.width = (W * c.h + hmax - 1) / hmax;
          c.height = (H * c.v + vmax - 1) / vmax;
          c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
          c.pred = 0;
        }
        if (comps.size() == 1) { hmax = vmax = 1; }
        BitReader br{p, end};
        int until_restart = restart_interval;
        int coef[64];
        for (int my = 0; my < mcus_y; ++my)
          for (int mx = 0; mx < mcus_x; ++mx) {
            if (restart_interval && until_restart == 0) {
              // drop the remaining bits, step over the RSTn marker
              const uint8_t* q = br.p;
              while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
              if (q + 1 >= end) throw std::runtime_error("JPEG: missing restart marker");
              br.p = q + 2;
              br.reset();
              for (auto& c : comps) c.pred = 0;
              until_restart = restart_interval;
            }
            for (auto& c : comps)
              for (int by = 0; by < c.v; ++by)
                for (int bx = 0; bx < c.h; ++bx) {
                  std::memset(coef, 0, sizeof coef);
                  int s = decode_symbol(br, dc[c.td]);
                  if (s > 11) throw std::runtime_error("JPEG: bad DC size");
                  if (s) c.pred += extend(br.get(s), s);
                  const uint16_t* q = quant[c.tq];
                  coef[0] = c.pred * q[0];
                  for (int k = 1; k < 64;) {
                    int rs = decode_symbol(br, ac[c.ta]);
                    int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                      if (r != 15) break;
                      k += 16;
                      continue;
                    }
                    k += r;
                    if (k > 63) throw std::runtime_error("JPEG: coefficient index out of range");
                    int z = kZigzag[k];
                    coef[z] = extend(br.get(sz), sz) * q[z];
                    ++k;
                  }
                  int x0 = (mx * c.h + bx) * 8, y0 = (my * c.v + by) * 8;
                  idct_islow(coef, &c.plane[(size_t)y0 * c.blocks_w * 8 + x0], c.blocks_w * 8);
                }
            if (restart_interval) --until_restart;
          }
        p = br.p;
        decoded = true;
        done = true;  // a sequential file has one scan
        break;
      }
      default:
        break;  // APPn, COM, ...
    }
  }
  if (!decoded) throw std::runtime_error("JPEG: no image data");
  Image img;
  img.width = W;
  img.height = H;
  img.pixels.resize((size_t)W * H);
  if (comps.size() == 1) {
    const Component& c = comps[0];
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        uint32_t g = c.plane[(size_t)y * c.blocks_w * 8 + x];
        img.pixels[(size_t)y * W + x] = g << 16 | g << 8 | g;
      }
    return img;
  }
  std::vector<uint8_t> Y = upsample(comps[0], hmax, vmax, W, H), Cb = upsample(comps[1], hmax, vmax, W, H),
                       Cr = upsample(comps[2], hmax, vmax, W, H);
  // stored as RGB already: Adobe transform 0, or component ids 'R','G','B' without a JFIF/Adobe marker
  bool rgb = (adobe && adobe_transform == 0) ||
             (!adobe && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
  int cr_r[256], cb_b[256];
  long cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    long x = i - 128;
    cr_r[i] = (int)((91881L * x + 32768) >> 16);   // 1.40200
    cb_b[i] = (int)((116130L * x + 32768) >> 16);  // 1.77200
    cr_g[i] = -46802L * x;                         // 0.71414
    cb_g[i] = -22554L * x + 32768;                 // 0.34414
  }
  for (size_t i = 0; i < (size_t)W * H; ++i) {
    int r, g, b;
    if (rgb) {
      r = Y[i]; g = Cb[i]; b = Cr[i];
    } else {
      int y = Y[i], cb = Cb[i], cr = Cr[i];
      r = clamp8(y + cr_r[cr]);
      g = clamp8(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
      b = clamp8(y + cb_b[cb]);
    }
    img.pixels[i] = (uint32_t)r << 16 | (uint32_t)g << 8 | (uint32_t)b;
  }
  return img;
}

}  // namespace rt
