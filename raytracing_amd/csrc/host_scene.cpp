// host_scene.cpp — scene validation, the library's own BVH (binned SAH, one
// primitive per leaf, children's boxes stored in the parent) and the flattening
// into the scene image (scene_layout.h), for the on-chip store or the HBM store.
//
// The reference builds a random-axis median-split hierarchy (BVH.scala:18-33);
// any correct hierarchy returns the same closest hit, so this one is built for
// traversal cost instead.  Unlike AABB.hit (AABB.scala:5-19) the boxes here are
// conservative: flat boxes are accepted and every box is rounded outward.
#include "host_scene.h"

#include <float.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <limits>
#include <stdexcept>

namespace ht {

namespace {

struct Box {
  double lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::numeric_limits<double>::infinity();
      hi[a] = -std::numeric_limits<double>::infinity();
    }
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], b.lo[a]);
      hi[a] = std::max(hi[a], b.hi[a]);
    }
  }
  void grow_point(const double* p) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
    }
  }
  double half_area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Item {
  Box box;
  double c[3];
  uint32_t ref;
};

// bounding boxes as the reference defines them (Hittable.scala:51, 65, 79, 93, 114)
Box prim_box(int kind, const double* d) {
  Box b;
  b.reset();
  switch (kind) {
    case HT_PRIM_SPHERE: {
      double r = fabs(d[3]);
      for (int a = 0; a < 3; ++a) { b.lo[a] = d[a] - r; b.hi[a] = d[a] + r; }
      break;
    }
    case HT_PRIM_XYRECT:
      b.lo[0] = d[0]; b.hi[0] = d[1]; b.lo[1] = d[2]; b.hi[1] = d[3];
      b.lo[2] = d[4] - 0.0001; b.hi[2] = d[4] + 0.0001;
      break;
    case HT_PRIM_XZRECT:
      b.lo[0] = d[0]; b.hi[0] = d[1]; b.lo[2] = d[2]; b.hi[2] = d[3];
      b.lo[1] = d[4] - 0.0001; b.hi[1] = d[4] + 0.0001;
      break;
    case HT_PRIM_YZRECT:
      b.lo[1] = d[0]; b.hi[1] = d[1]; b.lo[2] = d[2]; b.hi[2] = d[3];
      b.lo[0] = d[4] - 0.0001; b.hi[0] = d[4] + 0.0001;
      break;
    default:
      b.grow_point(d); b.grow_point(d + 3); b.grow_point(d + 6);
  }
  // a rect given with swapped sides still gets a proper box
  for (int a = 0; a < 3; ++a)
    if (b.lo[a] > b.hi[a]) std::swap(b.lo[a], b.hi[a]);
  return b;
}

// A subtree of a large hierarchy that is built on its own, into its own node array, so that
// several can be built at once; the parent's child slot holds JOB_REF | index until the parts
// are put together.  Which ranges become jobs depends on the scene alone, never on the
// number of threads, so the hierarchy (and its numbering) is the same on every machine.
constexpr uint32_t JOB_REF = 0x7F000000u;
constexpr int JOB_MIN_SCENE = 65536;  // scenes below this are built in one piece
constexpr int JOB_MAX_ITEMS = 16384;  // a range of at most this many primitives becomes a job
struct BuildJob {
  int begin, end, levels_left, level;
};

struct Builder {
  BvhNode* nodes;        // room for one node less than the builder has primitives
  std::vector<Item>& items;
  uint32_t n_nodes = 0;
  std::vector<BuildJob>* jobs = nullptr;  // non-null: ranges of at most JOB_MAX_ITEMS are deferred
  std::vector<Item> scratch;
  int max_level = 0;
  // leaves of two triangles / two ordinary spheres: (kind-local index of the first, of the
  // second); a pair's reference carries its position in the list of its kind until the
  // primitives are renumbered
  std::vector<std::pair<uint32_t, uint32_t>> pairs, sph_pairs;

  static constexpr int NBINS = 16;

  // returns the reference of the subtree over items[begin, end)
  uint32_t build(int begin, int end, int levels_left, int level, Box& out_box) {
    const int n = end - begin;
    if (n == 1) {
      out_box = items[begin].box;
      return items[begin].ref;
    }
    // two triangles (or two ordinary spheres) below an inner node become one leaf: the node
    // that would separate them costs a visit of its own, more than testing the second
    // primitive outright
    if (n == 2 && level > 0) {
      const uint32_t k0 = (items[begin].ref >> WREF_KIND_SHIFT) & 7u, k1 = (items[begin + 1].ref >> WREF_KIND_SHIFT) & 7u;
      if (k0 == k1 && (k0 == KIND_TRIANGLE || k0 == KIND_SPHERE)) {
        out_box = items[begin].box;
        out_box.grow(items[begin + 1].box);
        auto& list = k0 == KIND_TRIANGLE ? pairs : sph_pairs;
        list.emplace_back(items[begin].ref & WREF_INDEX_MASK, items[begin + 1].ref & WREF_INDEX_MASK);
        return WREF_LEAF | ((k0 == KIND_TRIANGLE ? KIND_TRI_PAIR : KIND_SPH_PAIR) << WREF_KIND_SHIFT) |
               (uint32_t)(list.size() - 1);
      }
    }
    if (jobs && n <= JOB_MAX_ITEMS) {
      out_box = items[begin].box;
      for (int i = begin + 1; i < end; ++i) out_box.grow(items[i].box);
      jobs->push_back(BuildJob{begin, end, levels_left, level});
      return JOB_REF | (uint32_t)(jobs->size() - 1);
    }
    const uint32_t idx = n_nodes++;
    nodes[idx] = BvhNode();
    max_level = std::max(max_level, level + 1);

    Box cb;
    cb.reset();
    for (int i = begin; i < end; ++i) cb.grow_point(items[i].c);

    int best_axis = -1, best_split = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    for (int axis = 0; axis < 3; ++axis) {
      const double ext = cb.hi[axis] - cb.lo[axis];
      if (!(ext > 0)) continue;
      Box bins[NBINS];
      int cnt[NBINS] = {0};
      for (auto& b : bins) b.reset();
      const double scale = NBINS / ext;
      for (int i = begin; i < end; ++i) {
        int k = std::min(NBINS - 1, std::max(0, (int)((items[i].c[axis] - cb.lo[axis]) * scale)));
        bins[k].grow(items[i].box);
        cnt[k]++;
      }
      double right_area[NBINS];
      int right_cnt[NBINS];
      Box acc;
      acc.reset();
      int c = 0;
      for (int k = NBINS - 1; k > 0; --k) {
        acc.grow(bins[k]);
        c += cnt[k];
        right_area[k] = acc.half_area();
        right_cnt[k] = c;
      }
      acc.reset();
      c = 0;
      for (int k = 0; k < NBINS - 1; ++k) {
        acc.grow(bins[k]);
        c += cnt[k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        double cost = acc.half_area() * c + right_area[k + 1] * right_cnt[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
      }
    }

    int mid = -1;
    if (best_axis >= 0) {
      const double ext = cb.hi[best_axis] - cb.lo[best_axis];
      const double scale = NBINS / ext;
      const double lo = cb.lo[best_axis];
      const int ax = best_axis, sp = best_split;
      // a stable partition through the builder's own scratch array (std::stable_partition
      // takes a temporary buffer from the allocator on every call; with several builders
      // at work those calls serialise on the process's memory map)
      scratch.clear();
      int j = begin;
      for (int i = begin; i < end; ++i) {
        const int k = std::min(NBINS - 1, std::max(0, (int)((items[i].c[ax] - lo) * scale)));
        if (k <= sp) items[j++] = items[i];
        else scratch.push_back(items[i]);
      }
      std::copy(scratch.begin(), scratch.end(), items.begin() + j);
      mid = j;
    }
    // the traversal stack is bounded: a side must still fit in the levels left
    const int cap = (levels_left - 1 >= 30) ? (1 << 30) : (1 << std::max(0, levels_left - 1));
    if (mid <= begin || mid >= end || (mid - begin) > cap || (end - mid) > cap) {
      int axis = 0;
      for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
      std::stable_sort(items.begin() + begin, items.begin() + end,
                       [axis](const Item& x, const Item& y) { return x.c[axis] < y.c[axis]; });
      mid = begin + n / 2;
    }
    Box lb, rb;
    uint32_t l = build(begin, mid, levels_left - 1, level + 1, lb);
    uint32_t r = build(mid, end, levels_left - 1, level + 1, rb);
    BvhNode& nd = nodes[idx];
    for (int a = 0; a < 3; ++a) {
      nd.lo[0][a] = lb.lo[a]; nd.hi[0][a] = lb.hi[a];
      nd.lo[1][a] = rb.lo[a]; nd.hi[1][a] = rb.hi[a];
    }
    nd.child[0] = l;
    nd.child[1] = r;
    out_box = lb;
    out_box.grow(rb);
    return idx;
  }
};

// ---- hierarchy optimiser -------------------------------------------------------
// Improves a finished hierarchy by removing subtrees and putting them back where they cost
// least (Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume
// hierarchies", 2013).  The cost is the surface-area heuristic's: the sum of the areas of all
// inner nodes' boxes, proportional to the expected number of node visits of a random ray.  A
// top-down builder fixes every split before it has seen what is below it; the low-polygon meshes
// of the reference (long thin triangles, a few hundred of them) leave it 10-20 % to find.
// Leaves (single primitives and the leaves of two) are moved whole.  The depth of the result
// never exceeds `max_depth`: every level is two bytes of traversal stack per lane.
struct Optimiser {
  struct TNode {
    Box box;
    int parent = -1, c[2] = {-1, -1};  // c[0] < 0: a leaf
    uint32_t ref = 0;                  // leaf reference
    int h = 0;                         // levels of inner nodes at and below this node
  };
  std::vector<TNode> t;
  int root = 0, max_depth = 0;

  int add(const std::vector<BvhNode>& nodes, uint32_t ref, const Box& box, int parent) {
    const int idx = (int)t.size();
    t.emplace_back();
    t[idx].box = box;
    t[idx].parent = parent;
    if (ref & WREF_LEAF) {
      t[idx].ref = ref;
      return idx;
    }
    const BvhNode& nd = nodes[ref];
    int h = 0;
    for (int k = 0; k < 2; ++k) {
      Box cb;
      for (int a = 0; a < 3; ++a) { cb.lo[a] = nd.lo[k][a]; cb.hi[a] = nd.hi[k][a]; }
      const int ci = add(nodes, nd.child[k], cb, idx);
      t[idx].c[k] = ci;
      h = std::max(h, t[ci].h);
    }
    t[idx].h = h + 1;
    return idx;
  }
  void refit_from(int i) {
    for (; i >= 0; i = t[i].parent) {
      TNode& n = t[i];
      n.box = t[n.c[0]].box;
      n.box.grow(t[n.c[1]].box);
      n.h = 1 + std::max(t[n.c[0]].h, t[n.c[1]].h);
    }
  }
  double cost() const {
    double c = 0;
    for (const TNode& n : t)
      if (n.c[0] >= 0) c += n.box.half_area();
    return c;
  }
  static double joined_area(const Box& a, const Box& b) {
    Box u = a;
    u.grow(b);
    return u.half_area();
  }
  // takes n out (its parent p goes with it, the sibling moves up) and puts it back at the
  // cheapest place; p is reused as the new parent
  void reinsert(int n) {
    const int p = t[n].parent;
    if (p < 0 || p == root) return;
    const int g = t[p].parent, s = t[p].c[0] == n ? t[p].c[1] : t[p].c[0];
    t[g].c[t[g].c[0] == p ? 0 : 1] = s;
    t[s].parent = g;
    refit_from(g);
    // best-first search over the places x: the new parent of (x, n) takes x's place
    struct Cand { double induced; int node, level; };
    auto later = [](const Cand& a, const Cand& b) { return a.induced > b.induced; };
    std::vector<Cand> heap;
    heap.push_back(Cand{0.0, root, 0});
    const double area_n = t[n].box.half_area();
    double best_cost = std::numeric_limits<double>::infinity();
    int best = -1;
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), later);
      const Cand c = heap.back();
      heap.pop_back();
      if (c.induced + area_n >= best_cost) break;  // nothing that is left can do better
      const TNode& x = t[c.node];
      const double direct = joined_area(x.box, t[n].box), total = c.induced + direct;
      if (total < best_cost && c.level + 1 + std::max(x.h, t[n].h) <= max_depth) {
        best_cost = total;
        best = c.node;
      }
      if (x.c[0] >= 0) {
        const double below = total - x.box.half_area();  // what n adds to x's own box
        if (below + area_n < best_cost && c.level + 2 + t[n].h <= max_depth)
          for (int k = 0; k < 2; ++k) {
            heap.push_back(Cand{below, x.c[k], c.level + 1});
            std::push_heap(heap.begin(), heap.end(), later);
          }
      }
    }
    if (best < 0) best = s;  // cannot happen (the old place satisfies the depth bound); stay safe
    const int gx = t[best].parent;
    t[p].c[0] = best;
    t[p].c[1] = n;
    t[p].parent = gx;
    t[best].parent = p;
    t[n].parent = p;
    if (gx >= 0) t[gx].c[t[gx].c[0] == best ? 0 : 1] = p;
    else root = p;
    refit_from(p);
  }
  uint32_t write(int i, std::vector<BvhNode>& out) const {
    if (t[i].c[0] < 0) return t[i].ref;
    const uint32_t idx = (uint32_t)out.size();
    out.emplace_back();
    for (int k = 0; k < 2; ++k) {
      const uint32_t r = write(t[i].c[k], out);
      BvhNode& nd = out[idx];
      nd.child[k] = r;
      const Box& b = t[t[i].c[k]].box;
      for (int a = 0; a < 3; ++a) { nd.lo[k][a] = b.lo[a]; nd.hi[k][a] = b.hi[a]; }
    }
    return idx;
  }
};

// Runs the optimiser over `nodes` (node 0 = root) in place; `depth` is the hierarchy's depth and
// the bound the result keeps.  Returns the cost before and after in `before` / `after`.
void optimise_hierarchy(std::vector<BvhNode>& nodes, int& depth, int passes, double& before, double& after) {
  before = after = 0;
  if (nodes.size() < 3 || passes <= 0) return;
  Optimiser o;
  o.t.reserve(nodes.size() * 2 + 1);
  o.max_depth = depth;
  Box rb;
  rb.reset();
  for (int k = 0; k < 2; ++k)
    for (int a = 0; a < 3; ++a) {
      rb.lo[a] = std::min(rb.lo[a], nodes[0].lo[k][a]);
      rb.hi[a] = std::max(rb.hi[a], nodes[0].hi[k][a]);
    }
  o.root = o.add(nodes, 0u, rb, -1);
  before = after = o.cost();
  std::vector<int> order;
  for (int pass = 0; pass < passes; ++pass) {
    // large boxes first: they are the ones that sit in the wrong place most expensively
    order.clear();
    for (int i = 0; i < (int)o.t.size(); ++i)
      if (i != o.root) order.push_back(i);
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return o.t[x].box.half_area() > o.t[y].box.half_area(); });
    for (int n : order) o.reinsert(n);
    const double now = o.cost();
    const bool done = now > after * 0.998;
    after = now;
    if (done) break;
  }
  std::vector<BvhNode> out;
  out.reserve(nodes.size());
  o.write(o.root, out);
  nodes.swap(out);
  depth = o.t[o.root].h;
}

// ---- blob writer -------------------------------------------------------------
template <typename R> R round_down(double d);
template <typename R> R round_up(double d);
template <> double round_down<double>(double d) { return nextafter(d, -INFINITY); }
template <> double round_up<double>(double d) { return nextafter(d, INFINITY); }
template <> float round_down<float>(double d) {
  float f = (float)d;
  if ((double)f > d) f = nextafterf(f, -INFINITY);
  return nextafterf(f, -INFINITY);
}
template <> float round_up<float>(double d) {
  float f = (float)d;
  if ((double)f < d) f = nextafterf(f, INFINITY);
  return nextafterf(f, INFINITY);
}

// ---- IEEE half floats, rounded toward a chosen side --------------------------------
// Halves are ordered like the signed integer key = +-(bits & 0x7FFF); a binary search
// over the keys finds the nearest half on the wanted side of d.
double half_value(int key) {
  // every non-negative half, tabulated once (31 745 doubles)
  static const std::vector<double> table = [] {
    std::vector<double> t(0x7C01);
    for (int m = 0; m <= 0x7C00; ++m) {
      const int e = m >> 10, f = m & 0x3FF;
      t[m] = e == 0 ? ldexp((double)f, -24) : (e == 31 ? INFINITY : ldexp((double)(f | 0x400), e - 25));
    }
    return t;
  }();
  return key < 0 ? -table[-key] : table[key];
}
double half_value_of(uint16_t bits) { return half_value((bits & 0x8000) ? -(int)(bits & 0x7FFF) : (int)bits); }
uint16_t half_bits(int key) { return key < 0 ? (uint16_t)(0x8000 | -key) : (uint16_t)key; }
// A half at or just inside |d| toward zero, from the binary exponent; the two functions
// below step from it to the exact neighbour (one or two comparisons instead of a search
// over all 63 489 keys: the image of a million-triangle mesh rounds fourteen million numbers).
int half_key_toward_zero(double d) {
  const double a = fabs(d);
  int key;
  if (!(a >= ldexp(1.0, -24))) {
    key = 0;
  } else if (a >= 65504.0) {
    key = a == INFINITY ? 0x7C00 : 0x7BFF;
  } else {
    int ex;
    const double fr = frexp(a, &ex);  // a = fr * 2^ex, fr in [0.5, 1)
    const int e = ex - 1;
    if (e < -14) key = (int)floor(ldexp(a, 24));  // subnormal halves: multiples of 2^-24
    else key = ((e + 15) << 10) | (int)floor((2.0 * fr - 1.0) * 1024.0);
  }
  return d < 0 ? -key : key;
}
uint16_t half_round_down(double d) {  // largest half <= d
  int key = half_key_toward_zero(d);
  while (key > -0x7C00 && half_value(key) > d) key--;
  while (key < 0x7C00 && half_value(key + 1) <= d) key++;
  return half_bits(key);
}
uint16_t half_round_up(double d) {    // smallest half >= d
  int key = half_key_toward_zero(d);
  while (key < 0x7C00 && half_value(key) < d) key++;
  while (key > -0x7C00 && half_value(key - 1) >= d) key--;
  return half_bits(key);
}

// fn(begin, end) over [0, n) on several threads when the range is worth it; every call
// writes its own part of the image
template <typename F>
void parallel_ranges(uint32_t n, F fn) {
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 65536u || hw == 1) { fn(0u, n); return; }
  std::vector<std::thread> pool;
  std::atomic<bool> failed{false};  // an exception must not leave a thread: it is thrown again here
  const uint32_t per = (n + hw - 1) / hw;
  for (unsigned t = 0; t < hw; ++t) {
    const uint32_t b = std::min(n, t * per), e = std::min(n, b + per);
    if (b >= e) continue;
    auto part = [=, &failed] { try { fn(b, e); } catch (...) { failed = true; } };
    try { pool.emplace_back(part); } catch (...) { part(); }  // no thread to be had: here, then
  }
  for (auto& th : pool) th.join();
  if (failed) throw std::runtime_error("writing the scene image failed");
}

struct Blob {
  std::vector<uint8_t>& b;
  uint32_t reserve(size_t bytes) {
    size_t off = (b.size() + 15) & ~(size_t)15;
    b.resize(off + bytes, 0);
    return (uint32_t)off;
  }
  template <typename T> void put(uint32_t off, T v) { memcpy(b.data() + off, &v, sizeof(T)); }
};

// a u32 stored in the low 32 bits of an R-sized slot
template <typename R> void put_bits(Blob& bl, uint32_t off, uint32_t v) {
  if (sizeof(R) == 8) bl.put<uint64_t>(off, (uint64_t)v);
  else bl.put<uint32_t>(off, v);
}

template <typename R>
void write_blob(const HostScene& hs, SceneLayout& L, std::vector<uint8_t>& out, bool wide) {
  out.clear();
  Blob bl{out};
  const uint32_t RS = sizeof(R);
  memset(&L, 0, sizeof L);
  L.wide = wide ? 1u : 0u;
  auto ref_out = [wide](uint32_t r) { return wide ? wide_ref24(r) : narrow_ref(r); };
  L.n_nodes = (uint32_t)hs.nodes.size();
  L.n_sph = (uint32_t)hs.sph.size();
  L.n_rect = (uint32_t)hs.rect.size();
  L.n_tri = (uint32_t)hs.tri.size();
  L.n_mat = (uint32_t)hs.mat_type.size();
  L.n_tex = (uint32_t)hs.tex_type.size();
  L.n_perlin = (uint32_t)hs.n_perlin;
  L.n_images = (uint32_t)hs.image_width.size();
  L.depth = (uint32_t)hs.depth;

  const uint32_t ns = node_stride<R>(wide);
  L.nodes = bl.reserve((size_t)L.n_nodes * ns);
  parallel_ranges(L.n_nodes, [&](uint32_t i0, uint32_t i1) {
  for (uint32_t i = i0; i < i1; ++i) {
    const BvhNode& n = hs.nodes[i];
    uint32_t o = L.nodes + i * ns;
    uint32_t co;  // offset of the child references
    // per axis four numbers, one pair for each child: fp32 kernels read (centre 0, centre 1,
    // half-extent 0, half-extent 1), fp64 kernels (lo 0, lo 1, hi 0, hi 1)
    if (sizeof(R) == 4) {
      for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 3; ++a) {
          const double lo = n.lo[c][a], hi = n.hi[c][a];
          double m = 0.5 * (lo + hi), h = -1.0;  // an empty box is never entered: c + |e| > c - |e|
          if (lo <= hi) {
            if (wide) {
              m = half_value_of(half_round_down(std::min(std::max(m, -65504.0), 65504.0)));
            } else {
              m = (double)(float)std::min(std::max(m, -(double)FLT_MAX), (double)FLT_MAX);
            }
            // the stored centre is exact to the kernel; the half-extent covers the box from
            // there, plus the rounding of (centre - o)/d -/+ half/|d| in fp32
            h = std::max(hi - m, m - lo) * (1.0 + ldexp(1.0, -20));
          }
          if (wide) {
            bl.put<uint16_t>(o + (a * 4 + c) * 2, half_round_down(m));
            bl.put<uint16_t>(o + (a * 4 + 2 + c) * 2, h < 0.0 ? half_round_down(h) : half_round_up(h));
          } else {
            bl.put<float>(o + (a * 4 + c) * 4, (float)m);
            bl.put<float>(o + (a * 4 + 2 + c) * 4, h < 0.0 ? -1.0f : round_up<float>(h));
          }
        }
      co = o + (wide ? 24 : 48);
    } else {
      for (int c = 0; c < 2; ++c) {
        for (int a = 0; a < 3; ++a) bl.put<R>(o + (a * 4 + c) * RS, round_down<R>(n.lo[c][a]));
        for (int a = 0; a < 3; ++a) bl.put<R>(o + (a * 4 + 2 + c) * RS, round_up<R>(n.hi[c][a]));
      }
      co = o + 12 * RS;
    }
    if (wide) {
      bl.put<uint32_t>(co, wide_ref24(n.child[0]));
      bl.put<uint32_t>(co + 4, wide_ref24(n.child[1]));
    } else {
      bl.put<uint32_t>(co, narrow_ref(n.child[0]) | (narrow_ref(n.child[1]) << 16));
    }
  }
  });

  L.sph = bl.reserve((size_t)L.n_sph * 4 * RS);
  uint32_t n_big = 0;  // the big spheres come first in the sphere array; only they have a K
  while (n_big < L.n_sph && hs.sph_big[n_big]) n_big++;
  L.sph_k = bl.reserve((size_t)n_big * RS);
  L.meta_sph = bl.reserve((size_t)L.n_sph * 4);
  for (uint32_t i = 0; i < L.n_sph; ++i) {
    int p = hs.sph[i];
    const double* d = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
    for (int a = 0; a < 4; ++a) bl.put<R>(L.sph + (i * 4 + a) * RS, (R)d[a]);
    double K = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] - d[3] * d[3];
    if (i < n_big) bl.put<R>(L.sph_k + i * RS, (R)K);
    bl.put<uint32_t>(L.meta_sph + i * 4, ((uint32_t)hs.prim_material[p] << META_MAT_SHIFT) | (uint32_t)p);
  }

  L.rect = bl.reserve((size_t)L.n_rect * 8 * RS);
  L.meta_rect = bl.reserve((size_t)L.n_rect * 4);
  for (uint32_t i = 0; i < L.n_rect; ++i) {
    int p = hs.rect[i];
    const double* d = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
    for (int a = 0; a < 5; ++a) bl.put<R>(L.rect + (i * 8 + a) * RS, (R)d[a]);
    bl.put<uint32_t>(L.meta_rect + i * 4, ((uint32_t)hs.prim_material[p] << META_MAT_SHIFT) | (uint32_t)p);
  }

  L.tri = bl.reserve((size_t)L.n_tri * 12 * RS);
  L.meta_tri = bl.reserve((size_t)L.n_tri * 4);
  parallel_ranges(L.n_tri, [&](uint32_t i0, uint32_t i1) {
  for (uint32_t i = i0; i < i1; ++i) {
    int p = hs.tri[i];
    const double* d = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
    // edges and normal in fp64 (Hittable.scala:98-100), rounded once
    double e1[3], e2[3], n[3];
    for (int a = 0; a < 3; ++a) { e1[a] = d[3 + a] - d[a]; e2[a] = d[6 + a] - d[a]; }
    n[0] = e1[1] * e2[2] - e1[2] * e2[1];
    n[1] = e1[2] * e2[0] - e1[0] * e2[2];
    n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    uint32_t o = L.tri + i * 12 * RS;
    // fp32 kernels take u and v from the hit point: the two middle vectors are the dual basis
    // U = (e2 x n)/|n|^2, V = (n x e1)/|n|^2 there, the edges themselves for fp64
    double s1[3] = {e1[0], e1[1], e1[2]}, s2[3] = {e2[0], e2[1], e2[2]};
    if (sizeof(R) == 4) {
      const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
      const double inv = nn > 0.0 ? 1.0 / nn : 0.0;  // a degenerate triangle is never hit (det = 0)
      s1[0] = (e2[1] * n[2] - e2[2] * n[1]) * inv;
      s1[1] = (e2[2] * n[0] - e2[0] * n[2]) * inv;
      s1[2] = (e2[0] * n[1] - e2[1] * n[0]) * inv;
      s2[0] = (n[1] * e1[2] - n[2] * e1[1]) * inv;
      s2[1] = (n[2] * e1[0] - n[0] * e1[2]) * inv;
      s2[2] = (n[0] * e1[1] - n[1] * e1[0]) * inv;
    }
    for (int a = 0; a < 3; ++a) {
      bl.put<R>(o + a * RS, (R)d[a]);
      bl.put<R>(o + (3 + a) * RS, (R)s1[a]);
      bl.put<R>(o + (6 + a) * RS, (R)s2[a]);
      bl.put<R>(o + (9 + a) * RS, (R)n[a]);
    }
    bl.put<uint32_t>(L.meta_tri + i * 4, ((uint32_t)hs.prim_material[p] << META_MAT_SHIFT) | (uint32_t)p);
  }
  });

  L.mat = bl.reserve((size_t)L.n_mat * 2 * RS);
  for (uint32_t i = 0; i < L.n_mat; ++i) {
    put_bits<R>(bl, L.mat + i * 2 * RS, (uint32_t)hs.mat_type[i] | ((uint32_t)hs.mat_texture[i] << 4));
    bl.put<R>(L.mat + (i * 2 + 1) * RS, (R)hs.mat_param[i]);
  }

  L.tex = bl.reserve((size_t)L.n_tex * 4 * RS);
  for (uint32_t i = 0; i < L.n_tex; ++i) {
    const double* d = &hs.tex_data[(size_t)i * HT_TEX_STRIDE];
    for (int a = 0; a < 3; ++a) bl.put<R>(L.tex + (i * 4 + a) * RS, (R)d[a]);
    uint32_t kind = (uint32_t)hs.tex_type[i];
    uint32_t c0 = 0, c1 = 0;
    if (kind == HT_TEX_CHECKER) { c0 = (uint32_t)hs.tex_child[i * 2]; c1 = (uint32_t)hs.tex_child[i * 2 + 1]; }
    else if (kind != HT_TEX_SOLID) { c0 = (uint32_t)hs.tex_child[i * 2]; }
    put_bits<R>(bl, L.tex + (i * 4 + 3) * RS, kind | (c0 << 2) | (c1 << 17));
    if (kind == HT_TEX_IMAGE) L.has_uv_textures = 1;
  }

  L.perlin_stride = 256u * 4u * RS + 768u;
  L.perlin = bl.reserve((size_t)L.n_perlin * L.perlin_stride);
  for (uint32_t t = 0; t < L.n_perlin; ++t) {
    uint32_t o = L.perlin + t * L.perlin_stride;
    for (int i = 0; i < 256; ++i)
      for (int a = 0; a < 3; ++a)
        bl.put<R>(o + (i * 4 + a) * RS, (R)hs.perlin_ranvec[((size_t)t * 256 + i) * 3 + a]);
    for (int i = 0; i < 768; ++i)
      bl.put<uint8_t>(o + 256u * 4u * RS + i, (uint8_t)hs.perlin_perm[(size_t)t * 768 + i]);
  }

  L.images = bl.reserve((size_t)L.n_images * 16);
  for (uint32_t i = 0; i < L.n_images; ++i) {
    bl.put<uint32_t>(L.images + i * 16, (uint32_t)hs.image_width[i]);
    bl.put<uint32_t>(L.images + i * 16 + 4, (uint32_t)hs.image_height[i]);
    bl.put<uint32_t>(L.images + i * 16 + 8, (uint32_t)((uint64_t)hs.image_offset[i] & 0xFFFFFFFFu));
    bl.put<uint32_t>(L.images + i * 16 + 12, (uint32_t)((uint64_t)hs.image_offset[i] >> 32));
  }
  L.solid_textures_only = 1;
  for (int t : hs.tex_type) if (t != HT_TEX_SOLID) L.solid_textures_only = 0;
  L.scatter_diffuse_only = 1;
  for (int m : hs.mat_type) if (m != HT_MAT_DIFFUSE && m != HT_MAT_EMISSION) L.scatter_diffuse_only = 0;
  // kinds of primitive the hierarchy holds: what the leaf step has to be able to test
  {
    std::vector<uint8_t> hoisted_ref;
    auto in_hoist = [&](uint32_t r) { return std::find(hs.hoist.begin(), hs.hoist.end(), r) != hs.hoist.end(); };
    for (size_t pi = 0; pi < hs.prim_ref.size() && !hs.nodes.empty(); ++pi) {
      const uint32_t r = hs.prim_ref[pi];
      if (in_hoist(r)) continue;
      const uint32_t kind = (r >> WREF_KIND_SHIFT) & 7u;
      if (kind == KIND_SPHERE) L.tree_kinds |= TREE_SPHERES;
      else if (kind == KIND_BIG_SPHERE) L.tree_kinds |= TREE_SPHERES | TREE_MIXED;
      else if (kind == KIND_TRIANGLE) L.tree_kinds |= TREE_TRIANGLES;
      else L.tree_kinds |= TREE_RECTS;
    }
  }
  L.n_hoist = (uint32_t)hs.hoist.size();
  L.hoist = bl.reserve((size_t)L.n_hoist * 4);
  for (uint32_t i = 0; i < L.n_hoist; ++i) bl.put<uint32_t>(L.hoist + i * 4, ref_out(hs.hoist[i]));
  L.root_box = bl.reserve(8 * RS);
  for (int a = 0; a < 3; ++a) {
    bl.put<R>(L.root_box + a * RS, round_down<R>(hs.root_lo[a]));
    bl.put<R>(L.root_box + (3 + a) * RS, round_up<R>(hs.root_hi[a]));
  }
  L.total_bytes = bl.reserve(0);
  out.resize(L.total_bytes);
}

// Checker textures may share children (a DAG) but must not reach themselves.  Three-colour
// depth-first search with an explicit stack: every texture is expanded once, so a chain of
// checkers that name one child twice costs n steps, not 2^n.
bool textures_acyclic(const HostScene& hs) {
  const int n = (int)hs.tex_type.size();
  std::vector<uint8_t> colour((size_t)n, 0);  // 0 unseen, 1 on the path, 2 finished
  std::vector<std::pair<int, int>> path;      // (texture, next child to visit)
  for (int root = 0; root < n; ++root) {
    if (colour[(size_t)root]) continue;
    path.push_back({root, 0});
    colour[(size_t)root] = 1;
    while (!path.empty()) {
      const int t = path.back().first;
      const int c = path.back().second;
      if (hs.tex_type[(size_t)t] != HT_TEX_CHECKER || c == 2) {
        colour[(size_t)t] = 2;
        path.pop_back();
        continue;
      }
      path.back().second++;
      const int ch = hs.tex_child[(size_t)t * 2 + (size_t)c];
      if (colour[(size_t)ch] == 1) return false;  // back edge
      if (colour[(size_t)ch] == 0) {
        colour[(size_t)ch] = 1;
        path.push_back({ch, 0});
      }
    }
  }
  return true;
}

}  // namespace

// A hierarchy that came from outside -> the host's node array, depth first, with leaves of
// two where both children of a node below the root are single triangles or single ordinary
// spheres.  Only the TOPOLOGY of the input is used, and all of it is checked (it was computed
// on another device): every inner node and every primitive reached exactly once, indices in
// range, depth within `max_levels`.  The child boxes are not taken from the input: every node's
// are worked out here, bottom-up, as the union of the boxes of the primitives below it, so a
// wrong or stale box from the device can never cull geometry.  Returns false if anything is
// off; `nodes` and the pair lists are then garbage.
struct ExternalConverter {
  const std::vector<ExternalNode>& in;
  const std::vector<Item>& items;
  std::vector<BvhNode>& nodes;
  std::vector<std::pair<uint32_t, uint32_t>>&tri_pairs, &sph_pairs;
  std::vector<uint8_t> node_seen, item_seen;
  int max_levels, max_level = 0;
  bool ok = true;

  uint32_t leaf(uint32_t r, Box& box) {
    const uint32_t i = r & ~EXTERNAL_LEAF;
    if (i >= items.size() || item_seen[i]) { ok = false; return WREF_EMPTY; }
    item_seen[i] = 1;
    box = items[i].box;
    return items[i].ref;
  }
  // the reference that stands for subtree g, and the box of everything below it
  uint32_t convert(uint32_t g, int level, Box& box) {
    box.reset();
    if (!ok) return WREF_EMPTY;
    if (g >= in.size() || node_seen[g] || level >= max_levels) { ok = false; return WREF_EMPTY; }
    node_seen[g] = 1;
    const ExternalNode& e = in[g];
    Box cb[2];
    cb[0].reset();
    cb[1].reset();
    if (level > 0 && (e.child[0] & EXTERNAL_LEAF) && (e.child[1] & EXTERNAL_LEAF)) {
      const uint32_t a = leaf(e.child[0], cb[0]), b = leaf(e.child[1], cb[1]);
      if (!ok) return WREF_EMPTY;
      box = cb[0];
      box.grow(cb[1]);
      const uint32_t ka = (a >> WREF_KIND_SHIFT) & 7u, kb = (b >> WREF_KIND_SHIFT) & 7u;
      if (ka == kb && (ka == KIND_TRIANGLE || ka == KIND_SPHERE)) {
        auto& list = ka == KIND_TRIANGLE ? tri_pairs : sph_pairs;
        list.emplace_back(a & WREF_INDEX_MASK, b & WREF_INDEX_MASK);
        return WREF_LEAF | ((ka == KIND_TRIANGLE ? KIND_TRI_PAIR : KIND_SPH_PAIR) << WREF_KIND_SHIFT) | (uint32_t)(list.size() - 1);
      }
      const uint32_t idx = (uint32_t)nodes.size();
      nodes.emplace_back();
      max_level = std::max(max_level, level + 1);
      fill(idx, cb, a, b);
      return idx;
    }
    const uint32_t idx = (uint32_t)nodes.size();
    nodes.emplace_back();
    max_level = std::max(max_level, level + 1);
    uint32_t c[2];
    for (int k = 0; k < 2; ++k)
      c[k] = (e.child[k] & EXTERNAL_LEAF) ? leaf(e.child[k], cb[k]) : convert(e.child[k], level + 1, cb[k]);
    if (!ok) return WREF_EMPTY;
    fill(idx, cb, c[0], c[1]);
    box = cb[0];
    box.grow(cb[1]);
    return idx;
  }
  void fill(uint32_t idx, const Box cb[2], uint32_t c0, uint32_t c1) {
    BvhNode& nd = nodes[idx];
    for (int k = 0; k < 2; ++k)
      for (int a = 0; a < 3; ++a) {
        nd.lo[k][a] = cb[k].lo[a];
        nd.hi[k][a] = cb[k].hi[a];
        if (!(cb[k].lo[a] <= cb[k].hi[a])) ok = false;  // NaN, or an empty subtree
      }
    nd.child[0] = c0;
    nd.child[1] = c1;
  }
};

// LDS a 256-thread workgroup needs beside the image: stacks + tile sums (kernels.hip)
static uint32_t lds_beside_image(uint32_t depth) {
  return ((256u * 2u * (depth + 2u) + 15u) & ~15u) + (256u / 64u) * 64u * 3u * 8u;
}

int build_host_scene(const ht_scene_desc* d, HostScene& hs, std::string& err, const BuildOptions& opt) {
  if (!d) { err = "scene description is NULL"; return HT_E_INVALID; }
  if (d->n_prims <= 0 || !d->prim_type || !d->prim_material || !d->prim_data) {
    err = "scene has no primitives (the reference cannot build a BVH over an empty world either)";
    return HT_E_INVALID;
  }
  if (d->n_materials <= 0 || !d->mat_type || !d->mat_texture || !d->mat_param) {
    err = "scene has no materials"; return HT_E_INVALID;
  }
  if (d->n_textures <= 0 || !d->tex_type || !d->tex_data || !d->tex_child) {
    err = "scene has no textures"; return HT_E_INVALID;
  }
  if ((uint32_t)d->n_prims > MAX_PRIMS) { err = "too many primitives"; return HT_E_TOO_LARGE; }
  if ((uint32_t)d->n_materials > MAX_MATERIALS) { err = "too many materials"; return HT_E_TOO_LARGE; }
  if ((uint32_t)d->n_textures > MAX_TEXTURES) { err = "too many textures"; return HT_E_TOO_LARGE; }
  if (d->n_images < 0 || d->n_perlin < 0) { err = "negative table count"; return HT_E_INVALID; }
  // a texture record names its Perlin table or image in 15 bits
  if ((uint32_t)d->n_images > MAX_TEXTURES) { err = "too many images"; return HT_E_TOO_LARGE; }
  if ((uint32_t)d->n_perlin > MAX_TEXTURES) { err = "too many perlin tables"; return HT_E_TOO_LARGE; }

  hs.prim_type.assign(d->prim_type, d->prim_type + d->n_prims);
  hs.prim_material.assign(d->prim_material, d->prim_material + d->n_prims);
  hs.prim_data.assign(d->prim_data, d->prim_data + (size_t)d->n_prims * HT_PRIM_STRIDE);
  hs.mat_type.assign(d->mat_type, d->mat_type + d->n_materials);
  hs.mat_texture.assign(d->mat_texture, d->mat_texture + d->n_materials);
  hs.mat_param.assign(d->mat_param, d->mat_param + d->n_materials);
  hs.tex_type.assign(d->tex_type, d->tex_type + d->n_textures);
  hs.tex_data.assign(d->tex_data, d->tex_data + (size_t)d->n_textures * HT_TEX_STRIDE);
  hs.tex_child.assign(d->tex_child, d->tex_child + (size_t)d->n_textures * 2);
  if (d->n_images > 0) {
    if (!d->image_width || !d->image_height || !d->image_offset || !d->image_texels) {
      err = "image tables missing"; return HT_E_INVALID;
    }
    hs.image_width.assign(d->image_width, d->image_width + d->n_images);
    hs.image_height.assign(d->image_height, d->image_height + d->n_images);
    hs.image_offset.assign(d->image_offset, d->image_offset + d->n_images);
    int64_t total = 0;
    for (int k = 0; k < d->n_images; ++k) {
      if (hs.image_width[k] <= 0 || hs.image_height[k] <= 0 || hs.image_offset[k] < 0) {
        err = "bad image dimensions"; return HT_E_INVALID;
      }
      total = std::max(total, hs.image_offset[k] + (int64_t)hs.image_width[k] * hs.image_height[k]);
    }
    hs.image_texels.assign(d->image_texels, d->image_texels + total);
  }
  hs.n_perlin = d->n_perlin;
  if (d->n_perlin > 0) {
    if (!d->perlin_ranvec || !d->perlin_perm) { err = "perlin tables missing"; return HT_E_INVALID; }
    hs.perlin_ranvec.assign(d->perlin_ranvec, d->perlin_ranvec + (size_t)d->n_perlin * 768);
    hs.perlin_perm.assign(d->perlin_perm, d->perlin_perm + (size_t)d->n_perlin * 768);
    for (int32_t v : hs.perlin_perm)
      if (v < 0 || v > 255) { err = "perlin permutation entry out of range"; return HT_E_INVALID; }
  }

  for (int i = 0; i < d->n_textures; ++i) {
    int k = hs.tex_type[i];
    if (k < HT_TEX_SOLID || k > HT_TEX_IMAGE) { err = "unknown texture type"; return HT_E_INVALID; }
    if (k == HT_TEX_CHECKER) {
      for (int c = 0; c < 2; ++c) {
        int ch = hs.tex_child[i * 2 + c];
        if (ch < 0 || ch >= d->n_textures) { err = "checker child out of range"; return HT_E_INVALID; }
      }
    } else if (k == HT_TEX_PERLIN) {
      int t = hs.tex_child[i * 2];
      if (t < 0 || t >= d->n_perlin) { err = "perlin table index out of range"; return HT_E_INVALID; }
    } else if (k == HT_TEX_IMAGE) {
      int t = hs.tex_child[i * 2];
      if (t < 0 || t >= d->n_images) { err = "image index out of range"; return HT_E_INVALID; }
    }
  }
  if (!textures_acyclic(hs)) { err = "checker textures form a cycle"; return HT_E_INVALID; }
  for (int i = 0; i < d->n_materials; ++i) {
    if (hs.mat_type[i] < HT_MAT_DIFFUSE || hs.mat_type[i] > HT_MAT_EMISSION) {
      err = "unknown material type"; return HT_E_INVALID;
    }
    if (hs.mat_texture[i] < 0 || hs.mat_texture[i] >= d->n_textures) {
      err = "material texture out of range"; return HT_E_INVALID;
    }
  }

  // ---- per-kind arrays and leaf references ---------------------------------
  const int n = d->n_prims;
  std::vector<Box> boxes(n);
  for (int p = 0; p < n; ++p) {
    int k = hs.prim_type[p];
    if (k < HT_PRIM_SPHERE || k > HT_PRIM_TRIANGLE) { err = "unknown primitive type"; return HT_E_INVALID; }
    if (hs.prim_material[p] < 0 || hs.prim_material[p] >= d->n_materials) {
      err = "primitive material out of range"; return HT_E_INVALID;
    }
    const double* pd = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
    int used = (k == HT_PRIM_SPHERE) ? 4 : (k == HT_PRIM_TRIANGLE ? 9 : 5);
    for (int a = 0; a < used; ++a)
      if (!std::isfinite(pd[a])) { err = "non-finite primitive coordinate"; return HT_E_INVALID; }
    if (k == HT_PRIM_SPHERE && pd[3] == 0.0) { err = "sphere of radius 0"; return HT_E_INVALID; }
    boxes[p] = prim_box(k, pd);
  }
  hs.prim_ref.assign(n, 0);
  for (int p = 0; p < n; ++p) {
    int k = hs.prim_type[p];
    if (k == HT_PRIM_SPHERE) {
      hs.sph.push_back(p);
    } else if (k == HT_PRIM_TRIANGLE) {
      hs.prim_ref[p] = WREF_LEAF | (KIND_TRIANGLE << WREF_KIND_SHIFT) | (uint32_t)hs.tri.size();
      hs.tri.push_back(p);
    } else {
      uint32_t kind = k == HT_PRIM_XYRECT ? KIND_XYRECT : (k == HT_PRIM_XZRECT ? KIND_XZRECT : KIND_YZRECT);
      hs.prim_ref[p] = WREF_LEAF | (kind << WREF_KIND_SHIFT) | (uint32_t)hs.rect.size();
      hs.rect.push_back(p);
    }
  }
  const bool fits_refs16 =
      hs.sph.size() <= MAX_PER_KIND && hs.rect.size() <= MAX_PER_KIND && hs.tri.size() <= MAX_PER_KIND;
  // A sphere that dwarfs everything else (a "ground" sphere) is intersected in
  // the expanded form |o|^2 - 2 o.c + (|c|^2 - r^2): its error scales with the
  // distance of the ray origin from the WORLD origin, not with r^2.
  hs.sph_big.assign(hs.sph.size(), 0);
  // reach of every primitive from the world origin: the largest and the runner-up
  // give "everything else in the scene" for any one primitive in O(1)
  double reach1 = 0, reach2 = 0;
  int reach1_at = -1;
  for (int q = 0; q < n; ++q) {
    const Box& b = boxes[q];
    double m = 0;
    for (int a = 0; a < 3; ++a) { double v = std::max(fabs(b.lo[a]), fabs(b.hi[a])); m += v * v; }
    m = sqrt(m);
    if (m > reach1) { reach2 = reach1; reach1 = m; reach1_at = q; }
    else if (m > reach2) reach2 = m;
  }
  for (size_t i = 0; i < hs.sph.size(); ++i) {
    int p = hs.sph[i];
    const double* pd = &hs.prim_data[(size_t)p * HT_PRIM_STRIDE];
    double r = fabs(pd[3]);
    const double E = (p == reach1_at) ? reach2 : reach1;  // radius about the world origin of everything else
    double cl = sqrt(pd[0] * pd[0] + pd[1] * pd[1] + pd[2] * pd[2]);
    double K = fabs(cl * cl - r * r);
    bool big = 4.0 * (E * E + 2.0 * E * cl + K) < r * r;
    hs.sph_big[i] = big ? 1 : 0;
    hs.prim_ref[p] = WREF_LEAF | ((big ? KIND_BIG_SPHERE : KIND_SPHERE) << WREF_KIND_SHIFT) | (uint32_t)i;
  }

  // ---- hoisted primitives + hierarchy ---------------------------------------------
  // A primitive whose box covers most of the scene (a ground sphere, a room wall)
  // gains nothing from a hierarchy: every ray reaches it anyway.  Such primitives,
  // and ALL primitives of a tiny scene, are tested linearly by every ray at ray
  // start, where all lanes of a wave run the same code on the same primitive; the
  // BVH holds the rest.
  constexpr int MAX_HOIST = 8;
  std::vector<int> in_bvh;
  hs.hoist.clear();
  if (n <= MAX_HOIST) {
    for (int p = 0; p < n; ++p) hs.hoist.push_back(hs.prim_ref[p]);
  } else {
    Box all;
    all.reset();
    for (int p = 0; p < n; ++p) all.grow(boxes[p]);
    // the scene box without the very large primitives would be the honest yardstick,
    // but the median primitive size does as well and needs no iteration
    std::vector<double> areas(n);
    for (int p = 0; p < n; ++p) areas[p] = boxes[p].half_area();
    std::vector<double> sorted = areas;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double median = sorted[n / 2];
    std::vector<int> order(n);
    for (int p = 0; p < n; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return areas[a] > areas[b]; });
    std::vector<uint8_t> hoisted(n, 0);
    for (int k = 0; k < n && (int)hs.hoist.size() < MAX_HOIST; ++k) {
      int p = order[k];
      if (areas[p] >= 0.25 * all.half_area() && areas[p] > 1e4 * median) {
        hoisted[p] = 1;
      } else {
        break;
      }
    }
    for (int p = 0; p < n; ++p) {  // keep primitive order inside the hoist list
      if (hoisted[p]) hs.hoist.push_back(hs.prim_ref[p]);
      else in_bvh.push_back(p);
    }
  }
  const int nb = (int)in_bvh.size();
  std::vector<Item> items(nb);
  Box root;
  root.reset();
  for (int k = 0; k < nb; ++k) {
    int p = in_bvh[k];
    items[k].box = boxes[p];
    for (int a = 0; a < 3; ++a) items[k].c[a] = 0.5 * (boxes[p].lo[a] + boxes[p].hi[a]);
    items[k].ref = hs.prim_ref[p];
    root.grow(boxes[p]);
  }
  hs.nodes.clear();
  hs.depth = 0;
  hs.external_bvh_used = false;
  std::vector<std::pair<uint32_t, uint32_t>> tri_pairs, sph_pairs;  // leaves of two, from the builder
  if (nb == 1) {
    // like the reference's BVH(only, only) (BVH.scala:30): the same leaf twice
    BvhNode nd;
    for (int c = 0; c < 2; ++c)
      for (int a = 0; a < 3; ++a) { nd.lo[c][a] = items[0].box.lo[a]; nd.hi[c][a] = items[0].box.hi[a]; }
    nd.child[0] = nd.child[1] = items[0].ref;
    hs.nodes.push_back(nd);
    hs.depth = 1;
  } else if (nb > 1) {
    Box rb;
    // a hierarchy that can live on chip gets the on-chip depth bound; a larger one
    // gets the deeper bound of the HBM store
    const bool small = fits_refs16 && (uint32_t)(nb - 1) <= MAX_NODES && !opt.force_hbm;
    // HBM store: every level costs 4 bytes of LDS per thread, and with 18 levels (20
    // stack slots) two 768-thread workgroups still fit a CU (24 waves; measured +11 % at
    // 100 k triangles over the 23 levels the SAH tree would take, same number of box tests)
    int wide_levels = 1;
    while ((1 << std::min(wide_levels - 1, 30)) < nb && wide_levels < WIDE_STACK_MAX_DEPTH) wide_levels++;
    wide_levels = std::min(WIDE_STACK_MAX_DEPTH, std::max(18, wide_levels));
    const int wide_levels_cap = WIDE_STACK_MAX_DEPTH;  // what the HBM store's traversal stack can be sized for
    // A large scene: the top of the hierarchy here, the subtrees of up to JOB_MAX_ITEMS
    // primitives on several threads, each into its own arrays, appended in job order.
    const int max_levels = small ? STACK_MAX_DEPTH : wide_levels_cap;
    if (opt.external_bvh && opt.external_min_prims > 0 && nb >= std::max(2, opt.external_min_prims)) {
      // the other builder works on float boxes rounded outward: still boxes around the primitives
      std::vector<float> fb((size_t)nb * 6);
      float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (int k = 0; k < nb; ++k)
        for (int a = 0; a < 3; ++a) {
          fb[(size_t)k * 6 + a] = round_down<float>(items[k].box.lo[a]);
          fb[(size_t)k * 6 + 3 + a] = round_up<float>(items[k].box.hi[a]);
          const float c = (float)items[k].c[a];
          clo[a] = std::min(clo[a], c);
          chi[a] = std::max(chi[a], c);
        }
      std::vector<ExternalNode> ext((size_t)nb - 1);
      if (opt.external_bvh(opt.external_ctx, (uint32_t)nb, fb.data(), clo, chi, ext.data())) {
        ExternalConverter cv{ext, items, hs.nodes, tri_pairs, sph_pairs, std::vector<uint8_t>(ext.size(), 0),
                             std::vector<uint8_t>((size_t)nb, 0), max_levels};
        hs.nodes.reserve(ext.size());
        Box whole;
        const uint32_t r = cv.convert(0u, 0, whole);
        bool all = cv.ok && r == 0u;
        for (size_t k = 0; all && k < cv.node_seen.size(); ++k) all = cv.node_seen[k] != 0;
        for (size_t k = 0; all && k < cv.item_seen.size(); ++k) all = cv.item_seen[k] != 0;
        if (all) {
          hs.depth = cv.max_level;
          hs.external_bvh_used = true;
        } else {  // not a hierarchy over these primitives, or too deep for the traversal stack
          hs.nodes.clear();
          tri_pairs.clear();
          sph_pairs.clear();
        }
      }
    }
    if (!hs.external_bvh_used) {
    std::vector<BuildJob> jobs;
    hs.nodes.resize((size_t)nb);  // a hierarchy over nb primitives has at most nb - 1 nodes
    Builder top{hs.nodes.data(), items};
    if (nb >= JOB_MIN_SCENE) top.jobs = &jobs;
    top.build(0, nb, small ? STACK_MAX_DEPTH : wide_levels, 0, rb);
    hs.depth = top.max_level;
    tri_pairs.swap(top.pairs);
    sph_pairs.swap(top.sph_pairs);
    uint32_t n_nodes = top.n_nodes;
    if (!jobs.empty()) {
      // every job writes its nodes into a slice of the one array (no allocation inside the
      // threads: large allocations serialise on the process's memory map); the slices are
      // closed up afterwards
      struct Part {
        uint32_t slice = 0, n_nodes = 0, root = 0;
        int max_level = 0;
        std::vector<std::pair<uint32_t, uint32_t>> pairs, sph_pairs;
      };
      std::vector<Part> parts(jobs.size());
      uint32_t slice = top.n_nodes;
      for (size_t j = 0; j < jobs.size(); ++j) {
        parts[j].slice = slice;
        slice += (uint32_t)(jobs[j].end - jobs[j].begin) - 1u;
      }
      std::atomic<size_t> next{0};
      std::atomic<bool> failed{false};  // an exception (out of memory) must not leave a thread
      auto worker = [&] {
        try {
        std::vector<Item> scratch;  // kept from job to job
        for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
          Builder b{hs.nodes.data() + parts[j].slice, items};  // the ranges of different jobs do not overlap
          b.scratch.swap(scratch);
          Box box;
          const uint32_t root = b.build(jobs[j].begin, jobs[j].end, jobs[j].levels_left, jobs[j].level, box);
          Part done;
          done.slice = parts[j].slice;
          done.n_nodes = b.n_nodes;
          done.root = root;
          done.max_level = b.max_level;
          done.pairs.swap(b.pairs);
          done.sph_pairs.swap(b.sph_pairs);
          parts[j] = std::move(done);
          scratch.swap(b.scratch);
        }
        } catch (...) { failed = true; }
      };
      const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < hw; ++t) {
        try { pool.emplace_back(worker); } catch (...) { break; }  // fewer threads, same jobs
      }
      worker();
      for (auto& th : pool) th.join();
      if (failed) throw std::runtime_error("building the hierarchy failed");
      std::vector<uint32_t> job_root(jobs.size());
      for (size_t j = 0; j < jobs.size(); ++j) {
        const uint32_t node0 = n_nodes, tri0 = (uint32_t)tri_pairs.size(), sph0 = (uint32_t)sph_pairs.size();
        auto moved = [&](uint32_t r) -> uint32_t {  // a reference of the part in the whole hierarchy
          if (!(r & WREF_LEAF)) return r + node0;
          const uint32_t kind = (r >> WREF_KIND_SHIFT) & 7u;
          if (kind == KIND_TRI_PAIR) return r + tri0;
          if (kind == KIND_SPH_PAIR) return r + sph0;
          return r;
        };
        for (uint32_t k = 0; k < parts[j].n_nodes; ++k) {  // node0 + k <= slice + k: moving down, in order
          BvhNode nd = hs.nodes[parts[j].slice + k];
          nd.child[0] = moved(nd.child[0]);
          nd.child[1] = moved(nd.child[1]);
          hs.nodes[node0 + k] = nd;
        }
        n_nodes += parts[j].n_nodes;
        tri_pairs.insert(tri_pairs.end(), parts[j].pairs.begin(), parts[j].pairs.end());
        sph_pairs.insert(sph_pairs.end(), parts[j].sph_pairs.begin(), parts[j].sph_pairs.end());
        job_root[j] = moved(parts[j].root);
        hs.depth = std::max(hs.depth, parts[j].max_level);
      }
      for (uint32_t i = 0; i < n_nodes; ++i)
        for (int c = 0; c < 2; ++c) {
          uint32_t& r = hs.nodes[i].child[c];
          if (!(r & WREF_LEAF) && r >= JOB_REF) r = job_root[r - JOB_REF];
        }
    }
    hs.nodes.resize(n_nodes);
    }
  }
  if (opt.optimise_passes > 0 && hs.nodes.size() <= (size_t)opt.optimise_max_nodes)
    optimise_hierarchy(hs.nodes, hs.depth, opt.optimise_passes, hs.sah_before, hs.sah_after);
  {
    // the two primitives of a pair must be neighbours in the array of their kind: flagged
    // primitives first (the big spheres, never part of a pair), then the pairs in the order
    // the builder made them, then every other one in its old order
    auto regroup = [&](const std::vector<std::pair<uint32_t, uint32_t>>& pairs, std::vector<int32_t>& prims,
                       uint32_t single_kind, uint32_t pair_kind, std::vector<uint8_t>* flags) {
      if (pairs.empty() && !flags) return;
      const size_t np = prims.size();
      std::vector<uint32_t> new_index(np, 0xFFFFFFFFu);
      std::vector<int32_t> new_prims;
      new_prims.reserve(np);
      for (size_t old = 0; flags && old < np; ++old)
        if ((*flags)[old]) {
          new_index[old] = (uint32_t)new_prims.size();
          new_prims.push_back(prims[old]);
        }
      std::vector<uint32_t> pair_first(pairs.size());
      for (size_t k = 0; k < pairs.size(); ++k) {
        pair_first[k] = (uint32_t)new_prims.size();
        for (uint32_t old : {pairs[k].first, pairs[k].second}) {
          new_index[old] = (uint32_t)new_prims.size();
          new_prims.push_back(prims[old]);
        }
      }
      for (size_t old = 0; old < np; ++old)
        if (new_index[old] == 0xFFFFFFFFu) {
          new_index[old] = (uint32_t)new_prims.size();
          new_prims.push_back(prims[old]);
        }
      prims.swap(new_prims);
      if (flags) {
        std::vector<uint8_t> nf(np);
        for (size_t old = 0; old < np; ++old) nf[new_index[old]] = (*flags)[old];
        flags->swap(nf);
      }
      auto renumber = [&](uint32_t r) -> uint32_t {
        if (!(r & WREF_LEAF) || r == WREF_EMPTY) return r;
        const uint32_t kind = (r >> WREF_KIND_SHIFT) & 7u, i = r & WREF_INDEX_MASK;
        // a big sphere shares the sphere array with the ordinary ones
        if (kind == single_kind || (single_kind == KIND_SPHERE && kind == KIND_BIG_SPHERE))
          return (r & ~WREF_INDEX_MASK) | new_index[i];
        if (kind == pair_kind) return (r & ~WREF_INDEX_MASK) | pair_first[i];
        return r;
      };
      for (auto& nd : hs.nodes) { nd.child[0] = renumber(nd.child[0]); nd.child[1] = renumber(nd.child[1]); }
      for (auto& r : hs.hoist) r = renumber(r);
      for (auto& r : hs.prim_ref) r = renumber(r);
    };
    regroup(tri_pairs, hs.tri, KIND_TRIANGLE, KIND_TRI_PAIR, nullptr);
    regroup(sph_pairs, hs.sph, KIND_SPHERE, KIND_SPH_PAIR, &hs.sph_big);
  }
  for (int a = 0; a < 3; ++a) {
    hs.root_lo[a] = nb ? root.lo[a] : 0.0;
    hs.root_hi[a] = nb ? root.hi[a] : 0.0;
  }
  if (hs.sph.size() > WIDE_MAX_PER_KIND || hs.rect.size() > WIDE_MAX_PER_KIND || hs.tri.size() > WIDE_MAX_PER_KIND) {
    err = "too many primitives of one kind";
    return HT_E_TOO_LARGE;
  }
  const bool chip_ok = fits_refs16 && hs.nodes.size() <= MAX_NODES && hs.depth <= STACK_MAX_DEPTH && !opt.force_hbm;
  const uint32_t beside = lds_beside_image((uint32_t)hs.depth);
  auto image32 = [&] {
    write_blob<float>(hs, hs.layout32, hs.blob32, !chip_ok);
    if (chip_ok && (uint64_t)hs.layout32.total_bytes + beside > opt.lds_bytes)
      write_blob<float>(hs, hs.layout32, hs.blob32, true);
  };
  auto image64 = [&] {
    write_blob<double>(hs, hs.layout64, hs.blob64, !chip_ok);
    if (chip_ok && (uint64_t)hs.layout64.total_bytes + beside > opt.lds_bytes)
      write_blob<double>(hs, hs.layout64, hs.blob64, true);
  };
  if (n >= JOB_MIN_SCENE) {  // the two images of a large scene side by side
    std::atomic<bool> failed{false};
    std::thread other([&] { try { image64(); } catch (...) { failed = true; } });
    try { image32(); } catch (...) { failed = true; }
    other.join();
    if (failed) throw std::runtime_error("writing the scene images failed");
  } else {
    image32();
    image64();
  }
  if (hs.blob32.size() > 0xFFFFFFF0ull || hs.blob64.size() > 0xFFFFFFF0ull) {
    err = "scene image exceeds 4 GiB";
    return HT_E_TOO_LARGE;
  }
  return HT_OK;
}

// Camera.scala:5-17
void camera_basis(const ht_camera& in, CameraBasis& c) {
  auto sub = [](const double* a, const double* b, double* o) { for (int i = 0; i < 3; ++i) o[i] = a[i] - b[i]; };
  auto len = [](const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
  auto unit = [&](double* a) { double l = len(a); for (int i = 0; i < 3; ++i) a[i] /= l; };
  auto cross = [](const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const double viewport_height = 2.0 * tan(in.vfov * 0.017453292519943295 / 2);
  const double viewport_width = in.aspect_ratio * viewport_height;
  sub(in.origin, in.target, c.w);
  unit(c.w);
  cross(in.vup, c.w, c.u);
  unit(c.u);
  cross(c.w, c.u, c.v);
  double to[3];
  sub(in.target, in.origin, to);
  const double focus = in.focus_distance < 0 ? len(to) : in.focus_distance;
  for (int i = 0; i < 3; ++i) {
    c.horizontal[i] = (focus * viewport_width) * c.u[i];
    c.vertical[i] = (focus * viewport_height) * c.v[i];
    c.llc[i] = in.origin[i] - c.horizontal[i] / 2 - c.vertical[i] / 2 - focus * c.w[i];
    c.origin[i] = in.origin[i];
    c.background[i] = in.background[i];
  }
  c.lens_radius = in.aperture / 2;
}

}  // namespace ht
