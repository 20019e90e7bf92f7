// mvk_probe.hip -- a one-wave kernel that holds a stream busy for a given time, and nothing else.  mvk_pipe.cpp uses
// it at mvk_pipe_create to find out whether two of its streams reach the GPU through different hardware queues (kernels
// side by side) or through the same one (one after the other): see lanes_overlap there.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mvk {
namespace {
__global__ void __launch_bounds__(64) spin_kernel(unsigned ticks, unsigned* sink) {
  // the constant-rate wall clock (100 MHz); bounded by an iteration count as well, so the wave ends whatever the clock does
  const unsigned long long t0 = wall_clock64();
  unsigned spins = 0;
  const unsigned cap = ticks * 64u + 1024u;
  while (wall_clock64() - t0 < ticks && spins < cap) ++spins;
  if (sink && spins == 0xFFFFFFFFu) *sink = spins;  // (never true: keeps the loop observable)
}
}  // namespace

int launch_spin(void* stream, unsigned ticks) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), ticks, static_cast<unsigned*>(nullptr));
  return static_cast<int>(hipGetLastError());
}
}  // namespace mvk
