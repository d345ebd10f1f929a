// build_check.cpp — CPU-only checker of the scene builder (host_scene.cpp): random worlds in,
// structural invariants of the hierarchy and of the scene image out.  Built with
// AddressSanitizer + UBSan by `make build_check`; no device code, no HIP runtime.
//
//   build_check <n_spheres> <n_triangles> <n_rects> <seed> [force_hbm [external]]
//   build_check queue <seed>    the work queue of a launch (work_queue.h): the planning the library does
//                               and the numbering the kernels read a unit's tile and samples from
// external: 1 = the hierarchy comes from a stand-in for the device builder (median splits),
// 2 = from one that loses a primitive, 3 = from one that is a chain as deep as the scene;
// 2 and 3 must be refused and the host's own hierarchy used.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "host_scene.h"
#include "work_queue.h"

using namespace ht;

namespace {

struct Box3 {
  double lo[3], hi[3];
};

int g_fail = 0;
#define CHECK(cond, ...)                        \
  do {                                          \
    if (!(cond)) {                              \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);             \
      fprintf(stderr, "\n");                    \
      g_fail++;                                 \
    }                                           \
  } while (0)

Box3 prim_box(const HostScene& hs, int p) {
  const double* d = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
  Box3 b;
  for (int a = 0; a < 3; ++a) { b.lo[a] = 1e300; b.hi[a] = -1e300; }
  auto grow = [&](const double* q) {
    for (int a = 0; a < 3; ++a) { b.lo[a] = fmin(b.lo[a], q[a]); b.hi[a] = fmax(b.hi[a], q[a]); }
  };
  switch (hs.prim_type[p]) {
    case HT_PRIM_SPHERE: {
      double r = fabs(d[3]);
      double lo[3] = {d[0] - r, d[1] - r, d[2] - r}, hi[3] = {d[0] + r, d[1] + r, d[2] + r};
      grow(lo); grow(hi);
      break;
    }
    case HT_PRIM_XYRECT: { double lo[3] = {fmin(d[0], d[1]), fmin(d[2], d[3]), d[4]}, hi[3] = {fmax(d[0], d[1]), fmax(d[2], d[3]), d[4]}; grow(lo); grow(hi); break; }
    case HT_PRIM_XZRECT: { double lo[3] = {fmin(d[0], d[1]), d[4], fmin(d[2], d[3])}, hi[3] = {fmax(d[0], d[1]), d[4], fmax(d[2], d[3])}; grow(lo); grow(hi); break; }
    case HT_PRIM_YZRECT: { double lo[3] = {d[4], fmin(d[0], d[1]), fmin(d[2], d[3])}, hi[3] = {d[4], fmax(d[0], d[1]), fmax(d[2], d[3])}; grow(lo); grow(hi); break; }
    default: grow(d); grow(d + 3); grow(d + 6);
  }
  return b;
}

// primitives a leaf reference stands for
void leaf_prims(const HostScene& hs, uint32_t ref, std::vector<int>& out) {
  const uint32_t kind = (ref >> WREF_KIND_SHIFT) & 7u, i = ref & WREF_INDEX_MASK;
  switch (kind) {
    case KIND_SPHERE: case KIND_BIG_SPHERE:
      CHECK(i < hs.sph.size(), "sphere index %u of %zu", i, hs.sph.size());
      if (i < hs.sph.size()) {
        CHECK((hs.sph_big[i] != 0) == (kind == KIND_BIG_SPHERE), "sphere %u: kind %u against big flag %d", i, kind, hs.sph_big[i]);
        out.push_back(hs.sph[i]);
      }
      break;
    case KIND_SPH_PAIR:
      CHECK(i + 1 < hs.sph.size(), "sphere pair %u of %zu", i, hs.sph.size());
      if (i + 1 < hs.sph.size()) {
        CHECK(!hs.sph_big[i] && !hs.sph_big[i + 1], "a big sphere inside a pair at %u", i);
        out.push_back(hs.sph[i]);
        out.push_back(hs.sph[i + 1]);
      }
      break;
    case KIND_TRIANGLE:
      CHECK(i < hs.tri.size(), "triangle index %u of %zu", i, hs.tri.size());
      if (i < hs.tri.size()) out.push_back(hs.tri[i]);
      break;
    case KIND_TRI_PAIR:
      CHECK(i + 1 < hs.tri.size(), "triangle pair %u of %zu", i, hs.tri.size());
      if (i + 1 < hs.tri.size()) { out.push_back(hs.tri[i]); out.push_back(hs.tri[i + 1]); }
      break;
    default:
      CHECK(i < hs.rect.size(), "rect index %u of %zu", i, hs.rect.size());
      if (i < hs.rect.size()) {
        const int p = hs.rect[i];
        const int want = kind == KIND_XYRECT ? HT_PRIM_XYRECT : (kind == KIND_XZRECT ? HT_PRIM_XZRECT : HT_PRIM_YZRECT);
        CHECK(hs.prim_type[p] == want, "rect %u: kind %u against primitive type %d", i, kind, hs.prim_type[p]);
        out.push_back(p);
      }
  }
}

struct Walk {
  const HostScene& hs;
  std::vector<int> seen;   // times every primitive was reached
  std::vector<int> visited;  // nodes
  int max_level = 0, pairs = 0;
  // returns the box of everything below `ref`
  Box3 below(uint32_t ref, int level) {
    Box3 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = 1e300; b.hi[a] = -1e300; }
    if (ref == WREF_EMPTY) return b;
    if (ref & WREF_LEAF) {
      std::vector<int> ps;
      leaf_prims(hs, ref, ps);
      if (ps.size() == 2) pairs++;
      for (int p : ps) {
        seen[p]++;
        Box3 q = prim_box(hs, p);
        for (int a = 0; a < 3; ++a) { b.lo[a] = fmin(b.lo[a], q.lo[a]); b.hi[a] = fmax(b.hi[a], q.hi[a]); }
      }
      return b;
    }
    CHECK(ref < hs.nodes.size(), "node index %u of %zu", ref, hs.nodes.size());
    if (ref >= hs.nodes.size()) return b;
    CHECK(visited[ref] == 0, "node %u reached twice", ref);
    visited[ref]++;
    if (level + 1 > max_level) max_level = level + 1;
    const BvhNode& n = hs.nodes[ref];
    for (int c = 0; c < 2; ++c) {
      Box3 q = below(n.child[c], level + 1);
      for (int a = 0; a < 3; ++a) {
        if (q.lo[a] <= q.hi[a]) {
          CHECK(n.lo[c][a] <= q.lo[a] && n.hi[c][a] >= q.hi[a], "node %u child %d axis %d: box [%g, %g] does not hold [%g, %g]",
                ref, c, a, n.lo[c][a], n.hi[c][a], q.lo[a], q.hi[a]);
          b.lo[a] = fmin(b.lo[a], q.lo[a]);
          b.hi[a] = fmax(b.hi[a], q.hi[a]);
        }
      }
    }
    return b;
  }
};

// stand-ins for the device builder
struct Standin {
  int mode;
  uint32_t n;
  const float* boxes;
  ExternalNode* nodes;
  uint32_t next;
  std::vector<uint32_t> order;
  void box_of(uint32_t b, uint32_t e, float* lo, float* hi) {
    for (int a = 0; a < 3; ++a) { lo[a] = 3e38f; hi[a] = -3e38f; }
    for (uint32_t k = b; k < e; ++k)
      for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], boxes[(size_t)order[k] * 6 + a]);
        hi[a] = fmaxf(hi[a], boxes[(size_t)order[k] * 6 + 3 + a]);
      }
  }
  uint32_t build(uint32_t b, uint32_t e, int level) {  // e - b >= 2
    const uint32_t idx = next++;
    uint32_t mid = mode == 3 ? b + 1 : b + (e - b) / 2;
    const int axis = level % 3;
    if (mode != 3)
      std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) {
        return boxes[(size_t)x * 6 + axis] + boxes[(size_t)x * 6 + 3 + axis] < boxes[(size_t)y * 6 + axis] + boxes[(size_t)y * 6 + 3 + axis];
      });
    const uint32_t r[2][2] = {{b, mid}, {mid, e}};
    for (int c = 0; c < 2; ++c) {
      box_of(r[c][0], r[c][1], nodes[idx].lo[c], nodes[idx].hi[c]);
      nodes[idx].child[c] = r[c][1] - r[c][0] == 1 ? (EXTERNAL_LEAF | order[r[c][0]]) : build(r[c][0], r[c][1], level + 1);
    }
    return idx;
  }
};
bool standin_builder(void* ctx, uint32_t n, const float* boxes, const float*, const float*, ExternalNode* nodes) {
  Standin s{*(int*)ctx, n, boxes, nodes, 0, {}};
  s.order.resize(n);
  for (uint32_t i = 0; i < n; ++i) s.order[i] = i;
  s.build(0, n, 0);
  if (s.mode == 2) {  // the last leaf reference names the first primitive again
    for (uint32_t i = n - 1; i-- > 0;)
      for (int c = 1; c >= 0; --c)
        if (nodes[i].child[c] & EXTERNAL_LEAF) { nodes[i].child[c] = EXTERNAL_LEAF | s.order[0]; return true; }
  }
  return true;
}

// The work queue of launches of many shapes: fastdiv against the division it replaces, and every
// (own tile, sample) of the launch handed out exactly once whichever way the units are drawn from
// the one queue (on-chip store) or from the bands' queues (HBM store), as the kernels' unit fetch
// numbers them.
int check_queue(uint64_t seed) {
  std::mt19937_64 rng(seed);
  auto rnd = [&](uint32_t lo, uint32_t hi) { return lo + (uint32_t)(rng() % ((uint64_t)hi - lo + 1)); };
  // fastdiv: small and large divisors, dividends around the multiples and at the ends of the range
  for (int it = 0; it < 200000; ++it) {
    const uint32_t d = it < 5000 ? (uint32_t)it + 1u : (it & 1 ? rnd(1, 1u << 20) : rnd(1, 0xFFFFFFFFu));
    const FastDiv f = make_fastdiv(d);
    const uint32_t mult = rnd(0, 0xFFFFFFFFu / d) * d;
    const uint32_t ns[] = {0u, 1u, d - 1u, d, d + 1u, mult, mult - 1u, mult + 1u, 0xFFFFFFFFu, 0x7FFFFFFFu, rnd(0, 0xFFFFFFFFu)};
    for (uint32_t n : ns) CHECK(fastdiv(n, f.m, f.sh) == n / d, "fastdiv(%u, %u) = %u", n, d, fastdiv(n, f.m, f.sh));
  }
  int launches = 0;
  for (int it = 0; it < 300; ++it) {
    const int32_t W = (int32_t)rnd(2, it % 7 == 0 ? 700 : 200), H = (int32_t)rnd(2, it % 5 == 0 ? 400 : 120);
    const uint32_t stride = it % 3 == 0 ? rnd(1, 9) : 1u, first = rnd(0, stride - 1u);
    const uint32_t samples = rnd(1, it % 4 == 0 ? 300 : 40), chunk = 1u << rnd(0, 4), chunk_tail = std::min(chunk, 1u << rnd(0, 3));
    for (int banded = 0; banded < 2; ++banded) {
      const QueueGeom q = plan_queue(W, H, first, stride, samples, chunk, chunk_tail, banded != 0);
      const uint32_t tiles_x = (uint32_t)(W + 7) / 8u, tiles_y = (uint32_t)(H + 7) / 8u, n_tiles = tiles_x * tiles_y;
      std::vector<uint32_t> seen((size_t)n_tiles * samples, 0u);
      auto take = [&](uint32_t ck, uint32_t k) {  // what the kernel does with a unit
        uint32_t f0 = 0, cnt = 0;
        step_samples(ck, q.big, q.s_big, chunk, chunk_tail, samples, f0, cnt);
        const uint32_t tile = first + k * stride;
        const uint32_t ty = fastdiv(tile, q.div_tiles_x.m, q.div_tiles_x.sh), tx = tile - ty * q.tiles_x;
        CHECK(tile < n_tiles && ty < tiles_y && tx < tiles_x && ty * tiles_x + tx == tile, "tile %u of %u", tile, n_tiles);
        CHECK(f0 + cnt <= samples, "samples [%u, %u) of %u", f0, f0 + cnt, samples);
        if (tile < n_tiles)
          for (uint32_t s_ = f0; s_ < f0 + cnt && s_ < samples; ++s_) seen[(size_t)tile * samples + s_]++;
      };
      if (!banded) {
        CHECK(q.bands == 1u, "one queue");
        for (uint32_t unit = 0; unit < q.units; ++unit) {
          const uint32_t ck = fastdiv(unit, q.div_tiles.m, q.div_tiles.sh);
          take(ck, unit - ck * q.tiles);
        }
      } else {
        CHECK(q.bands == (q.tiles >= 8u * QUEUE_BANDS ? QUEUE_BANDS : 1u), "bands");
        uint32_t sum = 0;
        for (uint32_t b = 0; b < q.bands; ++b) {
          CHECK(q.band_start[b] == sum && q.band_units[b] == q.band_tiles[b] * q.steps, "band %u", b);
          sum += q.band_tiles[b];
          for (uint32_t u = 0; u < q.band_units[b]; ++u) {
            const uint32_t ck = fastdiv(u, q.band_div[b].m, q.band_div[b].sh);
            take(ck, q.band_start[b] + (u - ck * q.band_tiles[b]));
          }
        }
        CHECK(sum == q.tiles, "the bands hold %u of %u tiles", sum, q.tiles);
      }
      size_t bad = 0;
      for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t want = (t >= first && (t - first) % stride == 0u) ? 1u : 0u;
        for (uint32_t s_ = 0; s_ < samples; ++s_) bad += seen[(size_t)t * samples + s_] != want;
      }
      CHECK(bad == 0, "%zu (tile, sample) pairs not rendered exactly once: %dx%d first %u stride %u samples %u chunk %u/%u banded %d",
            bad, W, H, first, stride, samples, chunk, chunk_tail, banded);
      ++launches;
    }
  }
  if (g_fail) return 1;
  printf("ok queue launches=%d\n", launches);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "queue")) return check_queue((uint64_t)atoll(argv[2]));
  if (argc < 5) { fprintf(stderr, "usage: build_check n_spheres n_triangles n_rects seed [force_hbm]\n"); return 2; }
  const int ns = atoi(argv[1]), nt = atoi(argv[2]), nr = atoi(argv[3]);
  std::mt19937_64 rng((uint64_t)atoll(argv[4]));
  const bool force_hbm = argc > 5 && atoi(argv[5]) != 0;
  std::uniform_real_distribution<double> U(-1.0, 1.0);

  std::vector<int32_t> type, mat;
  std::vector<double> data;
  auto add = [&](int t, std::initializer_list<double> v) {
    type.push_back(t);
    mat.push_back((int32_t)(type.size() % 3));
    size_t o = data.size();
    data.resize(o + HT_PRIM_STRIDE, 0.0);
    size_t k = 0;
    for (double x : v) data[o + k++] = x;
  };
  const double spread = 4.0 + cbrt((double)(ns + nt + nr));
  for (int i = 0; i < ns; ++i) {
    if (i == ns / 2) { add(HT_PRIM_SPHERE, {0.0, -1000.0 - spread, 0.0, 1000.0}); continue; }  // a ground sphere among the rest
    add(HT_PRIM_SPHERE, {U(rng) * spread, U(rng) * spread, U(rng) * spread, 0.05 + 0.3 * fabs(U(rng))});
  }
  for (int i = 0; i < nt; ++i) {
    double a[3] = {U(rng) * spread, U(rng) * spread, U(rng) * spread};
    add(HT_PRIM_TRIANGLE, {a[0], a[1], a[2], a[0] + U(rng), a[1] + U(rng), a[2] + U(rng), a[0] + U(rng), a[1] + U(rng), a[2] + U(rng)});
  }
  for (int i = 0; i < nr; ++i) {
    double c[3] = {U(rng) * spread, U(rng) * spread, U(rng) * spread};
    add(HT_PRIM_XYRECT + i % 3, {c[0] - 0.3, c[0] + 0.3, c[1] - 0.2, c[1] + 0.2, c[2]});
  }
  // shuffle the world order: kinds interleaved as a scene file would have them
  const int n = (int)type.size();
  for (int i = n - 1; i > 0; --i) {
    int j = (int)(rng() % (uint64_t)(i + 1));
    std::swap(type[i], type[j]);
    std::swap(mat[i], mat[j]);
    for (int k = 0; k < HT_PRIM_STRIDE; ++k) std::swap(data[(size_t)i * HT_PRIM_STRIDE + k], data[(size_t)j * HT_PRIM_STRIDE + k]);
  }
  std::vector<int32_t> group(n);
  for (int i = 0; i < n; ++i) group[i] = i;
  int32_t mat_type[3] = {HT_MAT_DIFFUSE, HT_MAT_GLASS, HT_MAT_EMISSION}, mat_tex[3] = {0, 0, 0};
  double mat_param[3] = {0.0, 1.5, 2.0};
  int32_t tex_type[1] = {HT_TEX_SOLID}, tex_child[2] = {0, 0};
  double tex_data[HT_TEX_STRIDE] = {0.5, 0.5, 0.5};

  ht_scene_desc d;
  memset(&d, 0, sizeof d);
  d.n_prims = n;
  d.prim_type = type.data();
  d.prim_material = mat.data();
  d.prim_data = data.data();
  d.prim_group = group.data();
  d.n_materials = 3;
  d.mat_type = mat_type;
  d.mat_texture = mat_tex;
  d.mat_param = mat_param;
  d.n_textures = 1;
  d.tex_type = tex_type;
  d.tex_data = tex_data;
  d.tex_child = tex_child;

  HostScene hs;
  std::string err;
  BuildOptions opt;
  opt.force_hbm = force_hbm;
  int ext_mode = argc > 6 ? atoi(argv[6]) : 0;
  if (ext_mode) {
    opt.external_bvh = standin_builder;
    opt.external_ctx = &ext_mode;
    opt.external_min_prims = 2;
  }
  int rc = build_host_scene(&d, hs, err, opt);
  if (rc != HT_OK) { fprintf(stderr, "build failed: %d %s\n", rc, err.c_str()); return 1; }

  // the per-kind arrays are a partition of the primitives, and prim_ref finds each one again
  CHECK((int)(hs.sph.size() + hs.rect.size() + hs.tri.size()) == n, "per-kind arrays hold %zu primitives of %d", hs.sph.size() + hs.rect.size() + hs.tri.size(), n);
  CHECK(hs.sph_big.size() == hs.sph.size(), "big flags");
  bool small_seen = false;
  for (size_t i = 0; i < hs.sph.size(); ++i) {
    if (!hs.sph_big[i]) small_seen = true;
    else CHECK(!small_seen, "big sphere %zu after an ordinary one", i);
  }
  for (int p = 0; p < n; ++p) {
    std::vector<int> ps;
    const uint32_t kind = (hs.prim_ref[p] >> WREF_KIND_SHIFT) & 7u;
    CHECK(kind != KIND_TRI_PAIR && kind != KIND_SPH_PAIR, "primitive %d is referenced as a pair", p);
    leaf_prims(hs, hs.prim_ref[p], ps);
    CHECK(ps.size() == 1 && ps[0] == p, "prim_ref of %d leads to %d", p, ps.empty() ? -1 : ps[0]);
  }
  // every primitive is reached once: through the hoist list or through the hierarchy
  Walk w{hs, std::vector<int>(n, 0), std::vector<int>(hs.nodes.size(), 0)};
  for (uint32_t r : hs.hoist) {
    std::vector<int> ps;
    leaf_prims(hs, r, ps);
    CHECK(ps.size() == 1, "a pair in the hoist list");
    for (int p : ps) w.seen[p]++;
  }
  const bool twin = hs.nodes.size() == 1 && hs.nodes[0].child[0] == hs.nodes[0].child[1];  // BVH(only, only)
  if (!hs.nodes.empty()) {
    Box3 all = w.below(0u, 0);
    for (int a = 0; a < 3; ++a)
      CHECK(hs.root_lo[a] <= all.lo[a] && hs.root_hi[a] >= all.hi[a], "root box axis %d", a);
  }
  for (int p = 0; p < n; ++p) CHECK(w.seen[p] == ((twin && w.seen[p] == 2) ? 2 : 1), "primitive %d reached %d times", p, w.seen[p]);
  for (size_t i = 0; i < hs.nodes.size(); ++i) CHECK(w.visited[i] == 1, "node %zu reached %d times", i, w.visited[i]);
  CHECK(w.max_level == hs.depth, "depth %d, walked %d", hs.depth, w.max_level);
  // the scene images
  for (const SceneLayout* L : {&hs.layout32, &hs.layout64}) {
    const std::vector<uint8_t>& blob = L == &hs.layout32 ? hs.blob32 : hs.blob64;
    CHECK(L->total_bytes == blob.size() && blob.size() % 16 == 0, "image size");
    CHECK(L->n_nodes == hs.nodes.size() && L->depth == (uint32_t)hs.depth, "layout counts");
    CHECK(!force_hbm || L->wide, "forced HBM store not taken");
    if (L->wide) {
      // the HBM store's image holds references in 24 bits (wide_ref24): bits 31-28 and 19-0 of the
      // builder's form only, and never the empty reference
      for (size_t i = 0; i < hs.nodes.size(); ++i)
        for (int c = 0; c < 2; ++c) {
          const uint32_t r = hs.nodes[i].child[c];
          const uint32_t packed = wide_ref24(r);
          CHECK(((packed & 0xFFFFFu) | ((packed & 0xF00000u) << 8)) == r && r != WREF_EMPTY, "node %zu child %d: %08x does not fit 24 bits", i, c, r);
        }
    }
    if (!L->wide) {
      CHECK(hs.depth <= STACK_MAX_DEPTH && hs.sph.size() <= MAX_PER_KIND && hs.tri.size() <= MAX_PER_KIND, "on-chip limits");
      const uint32_t stride = L == &hs.layout32 ? 64u : 112u, co = L == &hs.layout32 ? 48u : 96u;
      for (size_t i = 0; i < hs.nodes.size(); ++i) {
        uint32_t cc;
        memcpy(&cc, blob.data() + L->nodes + i * stride + co, 4);
        CHECK((cc & 0xFFFFu) == narrow_ref(hs.nodes[i].child[0]) && (cc >> 16) == narrow_ref(hs.nodes[i].child[1]), "node %zu: children in the image", i);
        CHECK((cc & 0xFFFFu) != REF_EMPTY && (cc >> 16) != REF_EMPTY, "node %zu: a child reads as the empty reference", i);
      }
    }
  }
  if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
  printf("ok prims=%d nodes=%zu depth=%d pairs=%d hoisted=%zu wide32=%u wide64=%u image32=%u external=%d\n", n, hs.nodes.size(),
         hs.depth, w.pairs, hs.hoist.size(), hs.layout32.wide, hs.layout64.wide, hs.layout32.total_bytes, hs.external_bvh_used ? 1 : 0);
  return 0;
}
