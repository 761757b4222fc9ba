// libsfmi: version + small host utilities of the C ABI (include/sfmi.h).
#include "sfmi_common.h"

// One wavefront that waits `ticks` of the constant 100 MHz wall clock: the probe `shapeformer_amd/gpt.py:_chain_streams` uses to
// find HIP streams that really run concurrently (streams mapped to one hardware queue serialise; the runtime hands out only a few
// hardware queues and assigns them by first use, so which streams share one depends on what the process did before).
__global__ void stream_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {
int sfmi_version(void) { return 100; }
int sfmi_stream_spin(long long ticks, void* stream) {
  if (ticks < 0 || ticks > 100000000LL) return SFMI_EINVAL;   // <= 1 s
  hipLaunchKernelGGL(stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
}
