// mvk_grid.cpp -- host-side construction of the broadphase grid over the world
// boxes (no device code in this file).
//
// The collision role MoveIt/FCL's dynamic AABB tree plays in the reference
// (PlanningScene::checkCollision, call site motion_planning.cpp:156-161) is a
// uniform grid here: a tree walk is a per-lane pointer chase, a grid is one
// index computation and one contiguous list read per query.
//
// Query the kernels make: "every box within `radius` of point c" (c = centre of
// a segment's bounding sphere).  The cell holding c lists every box whose
// distance to ANY point of the cell is <= radius, so a single cell lookup
// answers the query conservatively; the exact tests that follow decide.  Points
// outside the grid volume are farther than `radius` from every box by
// construction (the volume is the boxes' bound inflated by radius).
//
// Layout: CSR.  start[c] .. start[c+1] index GROUPS of kGridGroup ids in
// items[]; every cell's list is padded to whole groups with kGridNone so that a
// lane fetches kGridGroup candidates with one aligned 8-byte load.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "mvk_internal.h"

namespace mvk {

namespace {

struct CellRange { int lo[3], hi[3]; };

// lower bound on the distance between the cube (centre cc, half side hh) and the
// oriented box: the larger of (a) the gap between the cube and the box's world
// AABB and (b) the gap measured in the box frame, where the cube is replaced by
// its own bounding box along the box axes.  Both never exceed the true distance.
double cube_box_lower_bound(const DevObs& o, const double ext[3], const double cc[3], double hh) {
  double a2 = 0.0, b2 = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double ga = std::fabs(cc[k] - o.c[k]) - ext[k] - hh;
    if (ga > 0.0) a2 += ga * ga;
  }
  const double d[3] = {cc[0] - o.c[0], cc[1] - o.c[1], cc[2] - o.c[2]};
  for (int i = 0; i < 3; ++i) {
    const double t = o.R[i] * d[0] + o.R[3 + i] * d[1] + o.R[6 + i] * d[2];
    const double e = hh * (std::fabs(o.R[i]) + std::fabs(o.R[3 + i]) + std::fabs(o.R[6 + i]));
    const double gb = std::fabs(t) - o.h[i] - e;
    if (gb > 0.0) b2 += gb * gb;
  }
  return std::sqrt(std::max(a2, b2));
}

}  // namespace

namespace {
// fn(t, lo, hi) for T contiguous ranges of [0, n), range t on its own thread (range 0 on the caller's); a thread that
// cannot be started leaves its range to the caller
template <class F>
void for_ranges(size_t n, int T, F&& fn) {
  std::vector<std::thread> pool;
  std::vector<char> started(static_cast<size_t>(T), 0);
  for (int t = 1; t < T; ++t) {
    try {
      pool.emplace_back([&fn, n, T, t] { fn(t, n * static_cast<size_t>(t) / T, n * static_cast<size_t>(t + 1) / T); });
      started[t] = 1;
    } catch (...) {
      break;
    }
  }
  fn(0, size_t{0}, n / static_cast<size_t>(T));
  for (int t = 1; t < T; ++t)
    if (!started[t]) fn(t, n * static_cast<size_t>(t) / T, n * static_cast<size_t>(t + 1) / T);
  for (auto& th : pool) th.join();
}
}  // namespace

// host threads of the builds started from the calling thread (mvk_capi.cpp sets it per thread before a build; the
// result does not depend on it)
static thread_local int g_build_threads = 1;
void grid_build_threads(int threads) { g_build_threads = threads < 1 ? 1 : threads; }

int build_grid(const DevObs* obs, size_t n, double radius, GridHost& g, bool by_bound, double pad, double cell_frac) {
  const int threads = g_build_threads;
  g.start.clear();
  g.items.clear();
  g.glb.clear();
  g.radius = radius;
  for (int k = 0; k < 3; ++k) { g.origin[k] = 0.0; g.n[k] = 1; }
  g.h = 1.0;
  if (n == 0) {
    g.start.assign(2, 0u);
    return MVK_OK;
  }
  if (n >= kGridNone) return fail(MVK_ERR_CAPACITY, "at most %d obstacles (16-bit candidate ids)", kGridNone - 1);
  // slack for the rounding of the device's cell-index computation (fp64, |coordinates| < 1e6)
  const double R = radius * (1.0 + 1e-9) + 1e-7;
  std::vector<double> ext(3 * n);  // half sizes of each box's world AABB
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      const double e = std::fabs(obs[i].R[3 * k]) * obs[i].h[0] + std::fabs(obs[i].R[3 * k + 1]) * obs[i].h[1] +
                       std::fabs(obs[i].R[3 * k + 2]) * obs[i].h[2];
      ext[3 * i + k] = e;
      lo[k] = std::min(lo[k], obs[i].c[k] - e - R);
      hi[k] = std::max(hi[k], obs[i].c[k] + e + R);
    }
  const double vol = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
  double h = std::max(cell_frac * R, std::cbrt(vol / static_cast<double>(kGridMaxCells)));
  for (int attempt = 0; attempt < 40; ++attempt, h *= 1.25) {
    uint64_t ncell = 1;
    int dim[3];
    for (int k = 0; k < 3; ++k) {
      dim[k] = static_cast<int>(std::ceil((hi[k] - lo[k]) / h));
      if (dim[k] < 1) dim[k] = 1;
      ncell *= static_cast<uint64_t>(dim[k]);
    }
    if (ncell > kGridMaxCells) continue;
    const double hh = 0.5 * h;
    auto range_of = [&](size_t i) {
      CellRange r;
      for (int k = 0; k < 3; ++k) {
        r.lo[k] = std::max(0, static_cast<int>(std::floor((obs[i].c[k] - ext[3 * i + k] - R - lo[k]) / h)) - 1);
        r.hi[k] = std::min(dim[k] - 1, static_cast<int>(std::floor((obs[i].c[k] + ext[3 * i + k] + R - lo[k]) / h)) + 1);
      }
      return r;
    };
    auto for_cells = [&](size_t i, auto&& fn) {
      const CellRange r = range_of(i);
      for (int z = r.lo[2]; z <= r.hi[2]; ++z)
        for (int y = r.lo[1]; y <= r.hi[1]; ++y)
          for (int x = r.lo[0]; x <= r.hi[0]; ++x) {
            const double cc[3] = {lo[0] + (x + 0.5) * h, lo[1] + (y + 0.5) * h, lo[2] + (z + 0.5) * h};
            const double lb = cube_box_lower_bound(obs[i], &ext[3 * i], cc, hh);
            if (lb <= R) fn((static_cast<uint64_t>(z) * dim[1] + y) * dim[0] + x, lb);
          }
    };
    std::vector<uint32_t> count(ncell, 0u);
    // the (cell, bound) hits of the counting pass are kept and replayed by the filling pass (the bound against a cell is
    // the costly part of either); beyond kHitKeep hits the filling pass computes them again.  The counting pass -- a
    // bound per (box, cell in the box's reach), most of the build -- runs over contiguous box ranges on `threads`
    // threads, each with its own hit list; the lists are replayed in range order, which is box order: the same grid.
    struct Hit { uint32_t cell; float lb; };
    constexpr size_t kHitKeep = size_t{1} << 24;
    const int T = static_cast<int>(std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(std::max(threads, 1)), size_t{16}, n / 64 + 1})));
    std::vector<std::vector<Hit>> hits_t(static_cast<size_t>(T));
    std::vector<uint32_t> hits_of_box(n, 0u);
    std::atomic<size_t> n_hits{0};
    for_ranges(n, T, [&](int t, size_t b0, size_t b1) {
      std::vector<Hit>& mine = hits_t[static_cast<size_t>(t)];
      for (size_t i = b0; i < b1 && n_hits.load(std::memory_order_relaxed) <= kHitKeep; ++i) {
        const size_t before = mine.size();
        for_cells(i, [&](uint64_t c, double lb) { mine.push_back(Hit{static_cast<uint32_t>(c), static_cast<float>(lb)}); });
        hits_of_box[i] = static_cast<uint32_t>(mine.size() - before);
        n_hits.fetch_add(mine.size() - before, std::memory_order_relaxed);
      }
    });
    const bool kept = n_hits.load() <= kHitKeep;
    if (kept) {
      for (const auto& list : hits_t)
        for (const Hit& e : list) ++count[e.cell];
    } else {  // (a scene beyond the kept hits: count without keeping, the filling pass computes the bounds again)
      hits_t.clear();
      for (size_t i = 0; i < n; ++i) for_cells(i, [&](uint64_t c, double) { ++count[c]; });
    }
    uint64_t groups = 0;
    for (uint64_t c = 0; c < ncell; ++c) groups += (count[c] + kGridGroup - 1) / kGridGroup;
    if (groups * kGridGroup > kGridMaxItems) continue;
    g.start.assign(ncell + 1, 0u);
    uint32_t at = 0;
    for (uint64_t c = 0; c < ncell; ++c) {
      g.start[c] = at;
      at += (count[c] + kGridGroup - 1) / kGridGroup;
    }
    g.start[ncell] = at;
    g.items.assign(static_cast<size_t>(at) * kGridGroup, static_cast<uint16_t>(kGridNone));
    std::fill(count.begin(), count.end(), 0u);
    std::vector<float> lbs;
    if (by_bound) lbs.assign(g.items.size(), INFINITY);
    auto place = [&](size_t i, uint64_t c, double lb) {
      const size_t at_item = static_cast<size_t>(g.start[c]) * kGridGroup + count[c]++;
      g.items[at_item] = static_cast<uint16_t>(i);
      // (the cube test above took the cell with hh exact; the device may file a point within 1e-7 of a face under
      // the neighbouring cell: the bound gives that much away, and the caller's pad; the float of a kept hit may sit
      // half an ulp above the double it came from: the relative 1e-6 covers that as well)
      if (by_bound) lbs[at_item] = static_cast<float>(std::max(0.0, lb * (1.0 - 1e-6) - pad - 1e-6));
    };
    // boxes in ascending id order inside every cell: the hit lists are in box order (a list per box range, a box's hits
    // in for_cells' order), and the box of a hit is the one whose range the replay has reached
    if (kept) {
      for (int t = 0; t < T; ++t) {
        const std::vector<Hit>& list = hits_t[static_cast<size_t>(t)];
        size_t e = 0;
        for (size_t i = n * static_cast<size_t>(t) / T; i < n * static_cast<size_t>(t + 1) / T; ++i)
          for (uint32_t k = 0; k < hits_of_box[i]; ++k, ++e) place(i, list[e].cell, static_cast<double>(list[e].lb));
      }
    } else {
      for (size_t i = 0; i < n; ++i) for_cells(i, [&](uint64_t c, double lb) { place(i, c, lb); });
    }
    if (by_bound) {  // ... then by bound, ascending (ties: by id), the padding last
      g.glb.assign(at, 0.0f);
      constexpr uint64_t kChunk = 512;  // cells per ticket: the lists are uneven across the volume
      std::atomic<uint64_t> ticket{0};
      for_ranges(static_cast<size_t>(T), T, [&](int, size_t, size_t) {
        std::vector<std::pair<float, uint16_t>> cell;
        for (uint64_t c0 = ticket.fetch_add(kChunk); c0 < ncell; c0 = ticket.fetch_add(kChunk))
        for (uint64_t c = c0; c < std::min<uint64_t>(ncell, c0 + kChunk); ++c) {
          const size_t b0 = static_cast<size_t>(g.start[c]) * kGridGroup;
          cell.clear();
          for (uint32_t e = 0; e < count[c]; ++e) cell.emplace_back(lbs[b0 + e], g.items[b0 + e]);
          std::sort(cell.begin(), cell.end());
          for (uint32_t e = 0; e < count[c]; ++e) {
            g.items[b0 + e] = cell[e].second;
            if (e % kGridGroup == 0) g.glb[g.start[c] + e / kGridGroup] = std::nextafterf(cell[e].first, -INFINITY) < 0.0f ? 0.0f : std::nextafterf(cell[e].first, -INFINITY);
          }
        }
      });
    }
    for (int k = 0; k < 3; ++k) { g.origin[k] = lo[k]; g.n[k] = dim[k]; }
    g.h = h;
    return MVK_OK;
  }
  return fail(MVK_ERR_CAPACITY, "broadphase grid does not fit its budget (scene too large or too dense)");
}

// Far grid: nearest-candidate lists that need no radius.  With l(x, b) the distance from point x to box b and
// lmin(x) its minimum over the boxes, the cell holding x lists every box with l(x, b) <= lmin(x) + reach: whatever
// primitive of bounding radius <= reach is centred at x, the boxes that can realise its minimum distance (lower
// bound l - reach, upper bound lmin) are all in the list, however far away they are.  l is 1-Lipschitz in x, so
// with m the cell's centre and rc its half diagonal the list {b : l(m, b) <= lmin(m) + 2 rc + reach} covers every
// point of the cell.  The volume is the boxes' bound inflated by `margin`; points outside it are not answered
// (grid_query returns 0 there, the kernels fall back to the sweep over all boxes).
int build_far_grid(const DevObs* obs, size_t n, double reach, double margin, double pad, GridHost& g) {
  g.start.clear();
  g.items.clear();
  g.radius = INFINITY;
  for (int k = 0; k < 3; ++k) { g.origin[k] = 0.0; g.n[k] = 1; }
  g.h = 1.0;
  if (n == 0 || n >= kGridNone) return fail(MVK_ERR_CAPACITY, "far grid: no boxes, or more than the 16-bit ids hold");
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], obs[i].c[k] - obs[i].bound - margin);
      hi[k] = std::max(hi[k], obs[i].c[k] + obs[i].bound + margin);
    }
  // cells: as many as keep (cells x boxes) cheap tests affordable at scene set-up
  const double want = std::min(24576.0, std::max(4096.0, 6.0e7 / static_cast<double>(n)));
  const double vol = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
  const double h = std::max(std::cbrt(vol / want), 1e-3);
  int dim[3];
  uint64_t ncell = 1;
  for (int k = 0; k < 3; ++k) {
    dim[k] = std::max(1, static_cast<int>(std::ceil((hi[k] - lo[k]) / h)));
    ncell *= static_cast<uint64_t>(dim[k]);
  }
  if (ncell > (1u << 20)) return fail(MVK_ERR_CAPACITY, "far grid: volume too elongated for its cell budget");
  // (slack for the rounding of the device's cell-index computation, as build_grid)
  const double rc = 0.5 * h * std::sqrt(3.0) * (1.0 + 1e-9) + 1e-7;
  const double add = 2.0 * rc + reach + pad;
  std::vector<std::vector<uint16_t>> lists(ncell);
  // box centres and bounding radii as flat arrays: the cheap passes below stream over them
  std::vector<double> cx(n), cy(n), cz(n), bd(n);
  for (size_t b = 0; b < n; ++b) { cx[b] = obs[b].c[0]; cy[b] = obs[b].c[1]; cz[b] = obs[b].c[2]; bd[b] = obs[b].bound; }
  std::atomic<uint64_t> next{0};
  auto work = [&]() {
    std::vector<std::pair<uint16_t, double>> near;
    std::vector<double> d2(n);
    for (;;) {
      const uint64_t c = next.fetch_add(1);
      if (c >= ncell) break;
      const int x = static_cast<int>(c % dim[0]), y = static_cast<int>((c / dim[0]) % dim[1]), z = static_cast<int>(c / (static_cast<uint64_t>(dim[0]) * dim[1]));
      const double m[3] = {lo[0] + (x + 0.5) * h, lo[1] + (y + 0.5) * h, lo[2] + (z + 0.5) * h};
      // cheap pass: a box's centre lies in it, so |m - centre| bounds l(m, b) from above, and minus the box's
      // bounding radius from below
      double up2 = INFINITY;
      for (size_t b = 0; b < n; ++b) {
        const double dx = m[0] - cx[b], dy = m[1] - cy[b], dz = m[2] - cz[b];
        const double v = dx * dx + dy * dy + dz * dz;
        d2[b] = v;
        up2 = std::min(up2, v);
      }
      const double T = std::sqrt(up2) + add;
      near.clear();
      double lmin = INFINITY;
      for (size_t b = 0; b < n; ++b) {
        const double lim = T + bd[b];
        if (d2[b] > lim * lim) continue;  // |m - centre| - bound > (upper bound of lmin) + add
        const DevObs& o = obs[b];
        const double d[3] = {m[0] - o.c[0], m[1] - o.c[1], m[2] - o.c[2]};
        double s2 = 0.0;
        for (int a = 0; a < 3; ++a) {
          const double t = std::fabs(o.R[a] * d[0] + o.R[3 + a] * d[1] + o.R[6 + a] * d[2]) - o.h[a];
          if (t > 0.0) s2 += t * t;
        }
        const double l = std::sqrt(s2);
        lmin = std::min(lmin, l);
        near.emplace_back(static_cast<uint16_t>(b), l);
      }
      std::vector<uint16_t>& out = lists[c];
      size_t keep = 0;
      for (const auto& e : near) keep += e.second <= lmin + add;
      out.reserve(keep);
      for (const auto& e : near)
        if (e.second <= lmin + add) out.push_back(e.first);  // ascending ids
    }
  };
  {
    // (one thread when the caller asked for one -- MVK_SCENE_THREADS=1, the host diagnostics; else up to 8: this build
    // is cells x boxes of cheap tests, the longest of the three that run side by side)
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = g_build_threads <= 1 ? 1u : std::max(1u, std::min(8u, hw ? hw / 2 : 1u));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
  }
  uint64_t groups = 0;
  for (uint64_t c = 0; c < ncell; ++c) groups += (lists[c].size() + kGridGroup - 1) / kGridGroup;
  // (group indices travel in 21 bits of the distance kernels' ring items)
  if (groups >= (1u << 21) || groups * kGridGroup > kGridMaxItems) return fail(MVK_ERR_CAPACITY, "far grid: lists beyond the budget");
  g.start.assign(ncell + 1, 0u);
  uint32_t at = 0;
  for (uint64_t c = 0; c < ncell; ++c) {
    g.start[c] = at;
    at += static_cast<uint32_t>((lists[c].size() + kGridGroup - 1) / kGridGroup);
  }
  g.start[ncell] = at;
  g.items.assign(static_cast<size_t>(at) * kGridGroup, static_cast<uint16_t>(kGridNone));
  for (uint64_t c = 0; c < ncell; ++c) std::copy(lists[c].begin(), lists[c].end(), g.items.begin() + static_cast<size_t>(g.start[c]) * kGridGroup);
  for (int k = 0; k < 3; ++k) { g.origin[k] = lo[k]; g.n[k] = dim[k]; }
  g.h = h;
  return MVK_OK;
}

int grid_query(const GridHost& g, const double c[3], const uint16_t** ids) {
  *ids = nullptr;
  if (g.items.empty()) return 0;
  int ix[3];
  const double inv_h = 1.0 / g.h;
  for (int k = 0; k < 3; ++k) {
    const double f = (c[k] - g.origin[k]) * inv_h;
    if (!(f >= 0.0) || !(f < static_cast<double>(g.n[k]))) return 0;
    ix[k] = static_cast<int>(f);
  }
  const uint64_t cell = (static_cast<uint64_t>(ix[2]) * g.n[1] + ix[1]) * g.n[0] + ix[0];
  *ids = g.items.data() + static_cast<size_t>(g.start[cell]) * kGridGroup;
  return static_cast<int>(g.start[cell + 1] - g.start[cell]) * kGridGroup;
}

}  // namespace mvk
