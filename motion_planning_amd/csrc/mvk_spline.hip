// mvk_spline.hip -- batched evaluation of the path B-spline on the GPU.
// One lane per query time: BsplineRos::evaluate (bspline_ros/src/bspline_ros.cpp:88-110) for a
// whole sampling grid at once (the navigation node samples position and velocity at its control
// rate; a 100 s path at 40 Hz is 4000 evaluations of a 6+J dimensional degree-5 curve).
// Memory bound and tiny: the knots and control points (a few KB) are read through the scalar
// and vector caches, outputs are written SoA so that a wave stores 512 contiguous bytes per row.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "mvk_internal.h"

struct mvk_spline;

namespace mvk {

const std::vector<double>& spline_ctrl(const mvk_spline* s, int k);
const std::vector<double>& spline_knots(const mvk_spline* s, int k);
void spline_shape(const mvk_spline* s, int* degree, int* dim, int* n_ctrl, double* t0, double* t1);
void** spline_device_slot(mvk_spline* s, int** device);

namespace {

constexpr int kMaxDegree = 7;
constexpr int kMaxDim = 6 + MVK_MAX_JOINT;

struct SplineDev {
  const double* ctrl;   // [n][dim]
  const double* knots;  // [n + p + 1]
  int p, n, dim;
  double t0, span, scale;
};

__global__ void __launch_bounds__(256)
spline_eval_kernel(const SplineDev S, const double* __restrict__ t, uint64_t nt, double* __restrict__ out, uint64_t ld) {
  const uint64_t q = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= nt) return;
  double x = t[q];
  x = fmin(fmax(x, S.t0), S.t0 + S.span);
  x = (x - S.t0) / S.span;
  int k = S.p;  // knot span by bisection: u[k] <= x < u[k+1], last span for x == 1
  {
    int lo = S.p, hi = S.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (x >= S.knots[mid]) lo = mid; else hi = mid - 1;
    }
    k = lo;
  }
  double a[kMaxDegree + 1][kMaxDegree + 1];  // de Boor weights: the same for every dimension
#pragma unroll
  for (int r = 1; r <= kMaxDegree; ++r)
#pragma unroll
    for (int j = kMaxDegree; j >= 1; --j)
      if (r <= S.p && j >= r && j <= S.p) {
        const double lo = S.knots[j + k - S.p], den = S.knots[j + 1 + k - r] - lo;
        a[r][j] = den > 0.0 ? (x - lo) / den : 0.0;
      }
  for (int d = 0; d < S.dim; ++d) {
    double w[kMaxDegree + 1];
#pragma unroll
    for (int j = 0; j <= kMaxDegree; ++j) w[j] = j <= S.p ? S.ctrl[static_cast<uint64_t>(k - S.p + j) * S.dim + d] : 0.0;
#pragma unroll
    for (int r = 1; r <= kMaxDegree; ++r)
#pragma unroll
      for (int j = kMaxDegree; j >= 1; --j)
        if (r <= S.p && j >= r && j <= S.p) w[j] = (1.0 - a[r][j]) * w[j - 1] + a[r][j] * w[j];
    double v = 0.0;
#pragma unroll
    for (int j = 0; j <= kMaxDegree; ++j) v = (j == S.p) ? w[j] : v;
    out[static_cast<uint64_t>(d) * ld + q] = v * S.scale;
  }
}

// number of set bits among the first n_bits of a packed validity mask, by one block (the masks are
// n/8 bytes: a 64 M-sample sweep is 8 MiB)
__global__ void __launch_bounds__(1024)
mask_count_kernel(const unsigned long long* __restrict__ mask, uint64_t n_bits, unsigned long long* __restrict__ count) {
  __shared__ unsigned long long part[16];
  const uint64_t words = (n_bits + 63) / 64;
  unsigned long long acc = 0;
  for (uint64_t w = threadIdx.x; w < words; w += 1024) {
    unsigned long long v = mask[w];
    if (w == words - 1 && (n_bits & 63)) v &= (1ull << (n_bits & 63)) - 1ull;
    acc += static_cast<unsigned long long>(__popcll(v));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int k = 0; k < 16; ++k) tot += part[k];
    *count = tot;
  }
}

struct DeviceCopy {
  std::vector<double*> ctrl, knots;
};

}  // namespace

void spline_device_release(mvk_spline* s) {
  int* dev;
  void** slot = spline_device_slot(s, &dev);
  if (!*slot) return;
  DeviceCopy* dc = static_cast<DeviceCopy*>(*slot);
  (void)hipSetDevice(*dev);
  for (double* p : dc->ctrl) (void)hipFree(p);
  for (double* p : dc->knots) (void)hipFree(p);
  delete dc;
  *slot = nullptr;
}

}  // namespace mvk

using namespace mvk;

// device pointers: d_t [nt] query times, d_out [dim][ld] (SoA).  The spline's control points are
// uploaded to `device` on the first call and stay resident until mvk_spline_destroy.
extern "C" int mvk_spline_eval_dev(mvk_spline* s, int device, const double* d_t, size_t nt, int derive, double* d_out,
                                   size_t ld, void* stream) {
  if (!s) return fail(MVK_ERR_INVALID_ARG, "null spline");
  if (nt == 0) return MVK_OK;
  if (!d_t || !d_out || ld < nt) return fail(MVK_ERR_INVALID_ARG, "mvk_spline_eval_dev: bad arrays (null, or ld < nt)");
  int degree, dim, n_ctrl;
  double t0, t1;
  spline_shape(s, &degree, &dim, &n_ctrl, &t0, &t1);
  if (derive < 0 || derive >= degree) return fail(MVK_ERR_INVALID_ARG, "derivative %d exceeds the %d kept", derive, degree - 1);
  if (degree > kMaxDegree || dim > kMaxDim) return fail(MVK_ERR_CAPACITY, "spline degree %d / dimension %d beyond the kernel's %d / %d", degree, dim, kMaxDegree, kMaxDim);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return fail(MVK_ERR_NO_DEVICE, "mvk_spline_eval_dev: no such HIP device");
  if (hipSetDevice(device) != hipSuccess) return fail(MVK_ERR_HIP, "hipSetDevice failed");
  int* dev_of;
  void** slot = spline_device_slot(s, &dev_of);
  if (*slot && *dev_of != device) spline_device_release(s);
  if (!*slot) {
    DeviceCopy* dc = new DeviceCopy();
    for (int k = 0; k < degree; ++k) {
      const std::vector<double>& c = spline_ctrl(s, k);
      const std::vector<double>& u = spline_knots(s, k);
      double *pc = nullptr, *pu = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&pc), c.size() * sizeof(double)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&pu), u.size() * sizeof(double)) != hipSuccess ||
          hipMemcpy(pc, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(pu, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        if (pc) (void)hipFree(pc);
        if (pu) (void)hipFree(pu);
        for (double* p : dc->ctrl) (void)hipFree(p);
        for (double* p : dc->knots) (void)hipFree(p);
        delete dc;
        return fail(MVK_ERR_HIP, "mvk_spline_eval_dev: upload failed");
      }
      dc->ctrl.push_back(pc);
      dc->knots.push_back(pu);
    }
    *slot = dc;
    *dev_of = device;
  }
  const DeviceCopy* dc = static_cast<const DeviceCopy*>(*slot);
  SplineDev S;
  S.ctrl = dc->ctrl[derive];
  S.knots = dc->knots[derive];
  S.p = degree - derive;
  S.n = n_ctrl - derive;
  S.dim = dim;
  S.t0 = t0;
  S.span = t1 - t0;
  S.scale = 1.0 / std::pow(t1 - t0, derive);
  const uint64_t blocks = (nt + 255) / 256;
  if (blocks > 0x7fffffffULL) return fail(MVK_ERR_CAPACITY, "too many query times");
  hipLaunchKernelGGL(spline_eval_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), S,
                     d_t, static_cast<uint64_t>(nt), d_out, static_cast<uint64_t>(ld));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "spline kernel launch failed: %s", hipGetErrorString(e));
  return MVK_OK;
}

// valid count of a packed mask (the statistic that is all-reduced across GPUs next to the mask gather)
extern "C" int mvk_mask_count_dev(const uint64_t* d_mask, size_t n_bits, uint64_t* d_count, void* stream) {
  if (!d_mask || !d_count) return fail(MVK_ERR_INVALID_ARG, "null argument");
  hipLaunchKernelGGL(mask_count_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long*>(d_mask), static_cast<uint64_t>(n_bits),
                     reinterpret_cast<unsigned long long*>(d_count));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "mask count kernel launch failed: %s", hipGetErrorString(e));
  return MVK_OK;
}
