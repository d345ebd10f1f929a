// work_queue.h — geometry of the render kernels' work queue, planned by the host per launch and
// read by the waves at every unit fetch.  Plain C++ (no device types): the CPU-only checker
// (build_check.cpp) runs the same planning and numbering code that the library and the kernels use.
//
// A launch renders `samples` samples per pixel of its own 8x8 tiles (tile sharding: every
// tile_stride-th tile of the frame from tile_first on).  A work unit is (step, tile): the first
// `big` steps of `chunk` samples (s_big samples in all), the other steps of `chunk_tail` samples (the
// units still running when the queue runs dry decide how long the launch drags on, so the last
// `chunk` samples are cut finer).  On-chip store: one queue, unit = step * tiles + own tile number.
// HBM store: the own tiles in `bands` bands of consecutive tiles (rows of the picture), a queue per
// band, unit = step * band_tiles + tile number within the band.
#pragma once
#include <stdint.h>

namespace ht {

constexpr uint32_t QUEUE_BANDS = 8;       // one band per XCD
constexpr uint32_t QUEUE_BAND_WORDS = 8;  // eight queue heads, 32 bytes apart, fill a 256-byte slot

// Division of any 32-bit n by a divisor known when the launch is planned (Granlund & Montgomery
// 1994): five scalar instructions instead of the forty of a division at every unit fetch.
struct FastDiv {
  uint32_t m, sh;  // sh = sh1 | sh2 << 8
};
constexpr FastDiv make_fastdiv(uint32_t d) {  // d >= 1
  uint32_t l = 0;
  while (l < 32 && ((uint64_t)1 << l) < d) ++l;  // ceil(log2 d)
  const uint64_t m = (((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1;
  return FastDiv{(uint32_t)m, (l ? 1u : 0u) | ((l ? l - 1 : 0u) << 8)};
}
constexpr uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t sh) {
  const uint32_t t = (uint32_t)(((uint64_t)m * n) >> 32);
  return (t + ((n - t) >> (sh & 255u))) >> (sh >> 8);
}

struct QueueGeom {
  uint32_t tiles_x;       // tile columns of the frame
  uint32_t tiles;         // the launch's own tiles
  uint32_t big, s_big;    // steps of `chunk` samples, and the samples they cover
  uint32_t steps;         // all steps
  uint32_t units;         // steps * tiles
  FastDiv div_tiles_x, div_tiles;
  uint32_t bands;         // 1, or QUEUE_BANDS
  uint32_t band_start[QUEUE_BANDS], band_tiles[QUEUE_BANDS], band_units[QUEUE_BANDS];
  FastDiv band_div[QUEUE_BANDS];
};

constexpr uint32_t own_tiles(uint32_t n_tiles, uint32_t first, uint32_t stride) {
  return first < n_tiles ? (n_tiles - first + stride - 1u) / stride : 0u;
}

// banded: the HBM store's queue per XCD; needs as many first units per band as waves take by
// their number in the launch (grid % QUEUE_BANDS == 0) and bands worth having
inline QueueGeom plan_queue(int32_t W, int32_t H, uint32_t tile_first, uint32_t tile_stride, uint32_t samples,
                            uint32_t chunk, uint32_t chunk_tail, bool banded) {
  QueueGeom q{};
  const uint32_t tiles_y = (uint32_t)(H + 7) / 8u;
  if (chunk < 1u) chunk = 1u;
  if (chunk_tail < 1u) chunk_tail = 1u;
  if (tile_stride < 1u) tile_stride = 1u;
  q.tiles_x = (uint32_t)(W + 7) / 8u;
  q.tiles = own_tiles(q.tiles_x * tiles_y, tile_first, tile_stride);
  q.s_big = samples > chunk ? samples - chunk : 0u;
  q.big = (q.s_big + chunk - 1u) / chunk;
  q.steps = q.big + (samples - q.s_big + chunk_tail - 1u) / chunk_tail;
  q.units = q.tiles * q.steps;
  q.div_tiles_x = make_fastdiv(q.tiles_x ? q.tiles_x : 1u);
  q.div_tiles = make_fastdiv(q.tiles ? q.tiles : 1u);
  q.bands = banded && q.tiles >= 8u * QUEUE_BANDS ? QUEUE_BANDS : 1u;
  for (uint32_t b = 0; b < QUEUE_BANDS; ++b) {
    const uint32_t k0 = b < q.bands ? (uint32_t)((uint64_t)q.tiles * b / q.bands) : q.tiles;
    const uint32_t k1 = b < q.bands ? (uint32_t)((uint64_t)q.tiles * (b + 1u) / q.bands) : q.tiles;
    q.band_start[b] = k0;
    q.band_tiles[b] = k1 - k0;
    q.band_units[b] = (k1 - k0) * q.steps;
    q.band_div[b] = make_fastdiv(k1 > k0 ? k1 - k0 : 1u);
  }
  return q;
}

// samples [first, first + count) of step ck
constexpr void step_samples(uint32_t ck, uint32_t big, uint32_t s_big, uint32_t chunk, uint32_t chunk_tail,
                            uint32_t samples, uint32_t& first, uint32_t& count) {
  if (ck < big) {
    first = ck * chunk;
    count = chunk < s_big - first ? chunk : s_big - first;
  } else {
    first = s_big + (ck - big) * chunk_tail;
    count = chunk_tail < samples - first ? chunk_tail : samples - first;
  }
}

}  // namespace ht
