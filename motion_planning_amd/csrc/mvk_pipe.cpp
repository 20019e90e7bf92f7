// mvk_pipe.cpp -- independent batches over several contexts and streams (include/mvk.h, "pipelined batches").
//
// Calls of one mvk_ctx are ordered: the flag path's exact stage runs behind its screen stage and before the ctx's
// next call, because a caller may read the mask of call k as soon as call k has finished on its stream.  The exact
// stage is a latency-bound tail (one chain of dependent gathers per block, a twelfth of the chip busy); behind the
// NEXT batch's screen stage it would be free.  Batches are independent, so the way to that overlap is not a change of
// the per-call semantics but more than one context: a pipe owns `depth` contexts, each with a stream of its own, and
// hands consecutive batches to them round-robin.  Built on the public C ABI alone (mvk_create, mvk_is_valid_mask_dev,
// ...) plus HIP streams and events: whatever holds for a call on a ctx holds for a batch in a pipe.
//
// No counterpart in the reference (one state per isStateValid call, motion_planning.cpp:76-164); it serves the batched
// entry point the north star adds.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>

#include "mvk_internal.h"
#include "../../include/mvk_debug.h"

namespace {
constexpr int kPipeMaxDepth = 8;
}

struct mvk_pipe {
  int device = 0, depth = 0;
  uint64_t next = 0;  // batches submitted so far: batch k runs on lane k % depth
  mvk_ctx* ctx[kPipeMaxDepth] = {};
  hipStream_t stream[kPipeMaxDepth] = {};
  hipEvent_t done[kPipeMaxDepth] = {};   // recorded behind the lane's latest batch
  hipEvent_t ready = nullptr;            // recorded on a producer stream, waited for by a lane
  int restreamed = 0;                    // streams replaced at creation because they shared a hardware queue with another lane
};

using mvk::fail;
namespace mvk {
int last_code();                              // mvk_capi.cpp: the code of this thread's latest failure (mvk_create returns a pointer)
int launch_spin(void* stream, unsigned ticks);  // mvk_probe.hip: one wave busy for `ticks` of the 100 MHz wall clock
}

namespace {
// Do kernels on streams a and b run side by side?  One wave spins for 30 us on a alone, then one on each: through
// different hardware queues the second pair takes as long as the first kernel, through one queue twice as long.
// (-1: the probe itself failed; the caller keeps what it has)
int lanes_overlap(hipStream_t a, hipStream_t b) {
  hipEvent_t e[3] = {nullptr, nullptr, nullptr};
  int verdict = -1;
  constexpr unsigned kTicks = 3000;
  do {
    bool made = true;
    for (auto& x : e) made = made && hipEventCreate(&x) == hipSuccess;
    if (!made) break;
    if (mvk::launch_spin(a, 100) != 0 || mvk::launch_spin(b, 100) != 0) break;  // (code resident, queues awake)
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) break;
    float t_one = 0.0f, t_a = 0.0f, t_b = 0.0f;
    if (hipEventRecord(e[0], a) != hipSuccess || mvk::launch_spin(a, kTicks) != 0 || hipEventRecord(e[1], a) != hipSuccess) break;
    if (hipStreamSynchronize(a) != hipSuccess || hipEventElapsedTime(&t_one, e[0], e[1]) != hipSuccess) break;
    if (hipEventRecord(e[0], a) != hipSuccess || mvk::launch_spin(a, kTicks) != 0 || mvk::launch_spin(b, kTicks) != 0 ||
        hipEventRecord(e[1], a) != hipSuccess || hipEventRecord(e[2], b) != hipSuccess)
      break;
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) break;
    if (hipEventElapsedTime(&t_a, e[0], e[1]) != hipSuccess || hipEventElapsedTime(&t_b, e[0], e[2]) != hipSuccess) break;
    verdict = (t_a > t_b ? t_a : t_b) < 1.6f * t_one ? 1 : 0;
  } while (false);
  (void)hipGetLastError();
  for (auto& x : e)
    if (x) (void)hipEventDestroy(x);
  return verdict;
}
}  // namespace

extern "C" int mvk_pipe_create(const mvk_model* model, const mvk_box* boxes, size_t n_boxes, const mvk_params* params, int device,
                               int depth, mvk_pipe** out) {
  if (!out) return fail(MVK_ERR_INVALID_ARG, "mvk_pipe_create: null out");
  *out = nullptr;
  if (depth < 1 || depth > kPipeMaxDepth) return fail(MVK_ERR_INVALID_ARG, "mvk_pipe_create: depth must be 1..%d", kPipeMaxDepth);
  mvk_pipe* p = new (std::nothrow) mvk_pipe;
  if (!p) return fail(MVK_ERR_CAPACITY, "mvk_pipe_create: out of host memory");
  p->device = device;
  // The lanes are only worth having if their streams reach the GPU through DIFFERENT hardware queues (HIP has a handful
  // per process and hands them to streams as they are created; after a few create / destroy cycles two fresh streams can
  // end up behind one queue, and the lanes then run one after the other: measured, 0.120 -> 0.144 ms per batch).  The
  // runtime keeps a separate set of queues per stream priority, so the lanes take different priorities, from the lowest
  // up (depth 2: low and normal); which lane has which does not matter, the batches alternate.
  int prio_least = 0, prio_greatest = 0;
  for (int i = 0; i < depth; ++i) {
    p->ctx[i] = mvk_create(model, boxes, n_boxes, params, device);  // (sets the device; its own error text on failure)
    if (!p->ctx[i]) {  // (the text and the code are mvk_create's: no device, a bad model, ...)
      const int rc = mvk::last_code();
      mvk_pipe_destroy(p);
      return rc != MVK_OK ? rc : MVK_ERR_INVALID_ARG;
    }
    p->depth = i + 1;
    if (i == 0 && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = prio_greatest = 0;
    const int levels = prio_least - prio_greatest + 1;  // (numerically, the greatest priority is the smallest value)
    const int prio = prio_least - (levels > 1 ? i % levels : 0);  // from the lowest up: never above the caller's own streams
    if (hipStreamCreateWithPriority(&p->stream[i], hipStreamNonBlocking, prio) != hipSuccess ||
        hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming) != hipSuccess) {
      mvk_pipe_destroy(p);
      return fail(MVK_ERR_HIP, "mvk_pipe_create: stream / event creation failed");
    }
  }
  if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess) {
    mvk_pipe_destroy(p);
    return fail(MVK_ERR_HIP, "mvk_pipe_create: event creation failed");
  }
  // The priorities keep the lanes apart in a process of its own; beside other users of the runtime's queues (a
  // communicator's streams, a framework's) two lanes were still seen behind one queue now and then -- one run in four of
  // the exchange rehearsal ran at the speed of one ctx.  So the pipe looks: every lane against lane 0 with the spin
  // probe, and a lane that does not run beside it gets a fresh stream (the runtime hands the queues out in turn), a few
  // times at most.  MVK_PIPE_NO_PROBE=1 skips it (diagnostics).
  const char* no_probe = std::getenv("MVK_PIPE_NO_PROBE");
  if (depth > 1 && !(no_probe && no_probe[0] == '1')) {
    for (int i = 1; i < depth; ++i) {
      for (int attempt = 0; attempt < 6; ++attempt) {
        bool apart = true;
        for (int j = 0; j < i && apart; ++j) apart = lanes_overlap(p->stream[j], p->stream[i]) != 0;  // (-1: cannot tell, keep)
        if (apart) break;
        ++p->restreamed;
        hipStream_t fresh = nullptr;
        int prio = 0;
        (void)hipStreamGetPriority(p->stream[i], &prio);
        if (hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); break; }
        // (the old stream is destroyed after the new one exists, so that the runtime does not hand the same queue back)
        (void)hipStreamDestroy(p->stream[i]);
        p->stream[i] = fresh;
      }
    }
  }
  *out = p;
  return MVK_OK;
}

extern "C" void mvk_pipe_destroy(mvk_pipe* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  for (int i = 0; i < kPipeMaxDepth; ++i) {
    if (p->stream[i]) (void)hipStreamSynchronize(p->stream[i]);
    if (p->ctx[i]) mvk_destroy(p->ctx[i]);
    if (p->done[i]) (void)hipEventDestroy(p->done[i]);
    if (p->stream[i]) (void)hipStreamDestroy(p->stream[i]);
  }
  if (p->ready) (void)hipEventDestroy(p->ready);
  delete p;
}

extern "C" int mvk_debug_pipe_restreamed(const mvk_pipe* p) { return p ? p->restreamed : -1; }
extern "C" int mvk_debug_pipe_lanes_overlap(mvk_pipe* p, int a, int b) {
  if (!p || a < 0 || b < 0 || a >= p->depth || b >= p->depth || a == b) return -1;
  if (hipSetDevice(p->device) != hipSuccess) return -1;
  return lanes_overlap(p->stream[a], p->stream[b]);
}
extern "C" int mvk_pipe_depth(const mvk_pipe* p) { return p ? p->depth : 0; }
extern "C" void* mvk_pipe_stream(mvk_pipe* p, int lane) { return (p && lane >= 0 && lane < p->depth) ? static_cast<void*>(p->stream[lane]) : nullptr; }
extern "C" mvk_ctx* mvk_pipe_ctx(mvk_pipe* p, int lane) { return (p && lane >= 0 && lane < p->depth) ? p->ctx[lane] : nullptr; }

extern "C" int mvk_pipe_set_scene(mvk_pipe* p, const mvk_box* boxes, size_t n_boxes) {
  if (!p) return fail(MVK_ERR_INVALID_ARG, "null pipe");
  for (int i = 0; i < p->depth; ++i) {
    if (hipStreamSynchronize(p->stream[i]) != hipSuccess) return fail(MVK_ERR_HIP, "mvk_pipe_set_scene: a lane's stream failed");
    if (int rc = mvk_set_scene(p->ctx[i], boxes, n_boxes)) return rc;
  }
  return MVK_OK;
}

namespace {
// the lane of the next batch, its stream made to wait for the producer of the inputs
int begin_batch(mvk_pipe* p, void* producer_stream, int& lane) {
  if (!p) return fail(MVK_ERR_INVALID_ARG, "null pipe");
  if (hipSetDevice(p->device) != hipSuccess) return fail(MVK_ERR_HIP, "hipSetDevice failed");
  lane = static_cast<int>(p->next % static_cast<uint64_t>(p->depth));
  if (producer_stream != MVK_PIPE_NO_PRODUCER) {
    if (hipEventRecord(p->ready, static_cast<hipStream_t>(producer_stream)) != hipSuccess ||
        hipStreamWaitEvent(p->stream[lane], p->ready, 0) != hipSuccess)
      return fail(MVK_ERR_HIP, "mvk_pipe: could not order the batch behind its producer stream");
  }
  return MVK_OK;
}
int end_batch(mvk_pipe* p, int lane, int rc, int* ticket) {
  if (rc) return rc;
  if (hipEventRecord(p->done[lane], p->stream[lane]) != hipSuccess) return fail(MVK_ERR_HIP, "mvk_pipe: event record failed");
  ++p->next;
  if (ticket) *ticket = lane;
  return MVK_OK;
}
}  // namespace

extern "C" int mvk_pipe_is_valid_mask(mvk_pipe* p, const double* d_soa, size_t n, size_t ld, int layout, uint64_t* d_mask,
                                      void* producer_stream, int* ticket) {
  int lane = 0;
  if (int rc = begin_batch(p, producer_stream, lane)) return rc;
  return end_batch(p, lane, mvk_is_valid_mask_dev(p->ctx[lane], d_soa, n, ld, layout, d_mask, p->stream[lane]), ticket);
}

extern "C" int mvk_pipe_sweep_mask(mvk_pipe* p, const double* root_pose7, uint64_t first_index, size_t n, uint64_t seed, uint64_t* d_mask,
                                   int* ticket) {
  int lane = 0;
  if (int rc = begin_batch(p, MVK_PIPE_NO_PRODUCER, lane)) return rc;
  return end_batch(p, lane, mvk_sweep_mask_dev(p->ctx[lane], root_pose7, first_index, n, seed, d_mask, p->stream[lane]), ticket);
}

extern "C" int mvk_pipe_eval_batch(mvk_pipe* p, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags, double* d_min_dist,
                                   double* d_margin, uint8_t* d_band, void* producer_stream, int* ticket) {
  int lane = 0;
  if (int rc = begin_batch(p, producer_stream, lane)) return rc;
  return end_batch(p, lane, mvk_eval_batch_dev(p->ctx[lane], d_soa, n, ld, layout, d_flags, d_min_dist, d_margin, d_band, nullptr, p->stream[lane]),
                   ticket);
}

extern "C" int mvk_pipe_wait(mvk_pipe* p, int ticket, void* consumer_stream) {
  if (!p || ticket < 0 || ticket >= p->depth) return fail(MVK_ERR_INVALID_ARG, "mvk_pipe_wait: bad pipe or ticket");
  if (hipSetDevice(p->device) != hipSuccess) return fail(MVK_ERR_HIP, "hipSetDevice failed");
  const hipError_t e = consumer_stream == MVK_PIPE_HOST ? hipEventSynchronize(p->done[ticket])
                                                        : hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), p->done[ticket], 0);
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "mvk_pipe_wait: %s", hipGetErrorString(e));
  return MVK_OK;
}

extern "C" int mvk_pipe_drain(mvk_pipe* p) {
  if (!p) return fail(MVK_ERR_INVALID_ARG, "null pipe");
  if (hipSetDevice(p->device) != hipSuccess) return fail(MVK_ERR_HIP, "hipSetDevice failed");
  for (int i = 0; i < p->depth; ++i)
    if (hipStreamSynchronize(p->stream[i]) != hipSuccess) return fail(MVK_ERR_HIP, "mvk_pipe_drain: a lane's stream failed");
  return MVK_OK;
}
