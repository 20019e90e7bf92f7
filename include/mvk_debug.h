/* mvk_debug.h -- diagnostic entry points of libmvk.so.  Not part of the drop-in boundary (include/mvk.h): no
 * reference interface stands behind them; tests and profiles/ use them to look inside the kernels.  They are
 * exported by the release library so the GPU tests can run against the shipped binary; signatures may change. */
#ifndef MVK_DEBUG_H
#define MVK_DEBUG_H
#include "mvk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One pass of the fp32 screen stage (MVK_PRECISION_F32_REFINED) that also writes, per state, what its stability
 * classification worked with: d_lo = smallest (slack - error bound) over the tests it could bound, d_hi = smallest
 * (slack + error bound), d_cls = 1 certainly stable / 0 certainly unstable / -1 uncertain (refined in fp64) /
 * -2 non-finite input.  tests/test_gpu_f32.py. */
int mvk_debug_f32_bounds(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                         double* d_lo, double* d_hi, double* d_cls, void* stream);

/* Number of host calls on this ctx whose one-launch (fused) form gave up at its grid barrier and were repeated
 * with two launches (mvk_capi.cpp: fused_wait).  The environment variable MVK_FUSED_SPIN_LIMIT, read at mvk_create,
 * bounds the barrier's polls (0 = give up at once unless every block has already arrived):
 * tests/test_gpu_fused_giveup.py.  reset != 0 clears the count after reading it. */
long long mvk_debug_fused_giveups(mvk_ctx* ctx, int reset);

/* The distance grids' form of mvk_broadphase_query (mvk_grid.cpp build_grid, by_bound): per point the ids of its
 * cell's list in list order -- sorted by the box-to-cell lower bound -- and, per entry, the bound of its group of four
 * (less `pad`): a lower bound of the distance from any point of the cell to that entry and to every later one, which
 * is what lets the kernels' walks stop early.  counts[i] = entries of point i (may exceed cap; ids/bounds hold the
 * first cap).  Host only.  tests/test_grid_bounds.py. */
int mvk_debug_grid_bounds(const mvk_box* boxes, size_t n_boxes, double radius, double pad, const double* points,
                          size_t n_points, int32_t* counts, int32_t* ids, float* bounds, int cap);

/* mvk_pipe_create looks whether its lanes' streams run side by side (a one-wave spin kernel on two of them: through one
 * hardware queue the pair takes twice as long) and replaces the stream of a lane that shares a queue with another:
 * the number of streams it replaced for this pipe (csrc/mvk_pipe.cpp; MVK_PIPE_NO_PROBE=1 skips the look).  -1 for a
 * null pipe.  mvk_debug_pipe_lanes_overlap runs the same look again on lanes a and b: 1 side by side, 0 one after the
 * other, -1 the probe failed or a bad argument.  tests/test_gpu_pipe.py. */
int mvk_debug_pipe_restreamed(const mvk_pipe* pipe);
int mvk_debug_pipe_lanes_overlap(mvk_pipe* pipe, int lane_a, int lane_b);

#ifdef __cplusplus
}
#endif
#endif
