// libsfmi: version + small host utilities of the C ABI (include/sfmi.h).
#include "sfmi_common.h"

extern "C" {
int sfmi_version(void) { return 101; }

// A HIP stream whose kernels may only run on a subset of the compute units: `every` / `of` selects CUs i with
// (i % of) < every out of `total_cus` (e.g. 3 of every 16 -> 48 of 256 CUs, spread over the XCDs).  Used to run the
// MFMA-bound decode stage of batch i beside the latency-sensitive decode chains of batch i+1 without flooding every CU
// with long-running convolution workgroups.  The caller owns the stream (sfmi_stream_destroy).
int sfmi_stream_create_cu_subset(int every, int of, int total_cus, void** stream_out) {
  if (!stream_out || every <= 0 || of <= 0 || every > of || total_cus <= 0 || total_cus > 1024) return SFMI_EINVAL;
  uint32_t mask[32] = {0};
  for (int i = 0; i < total_cus; ++i)
    if ((i % of) < every) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((total_cus + 31) / 32), mask);
  if (e != hipSuccess) return (int)e;
  *stream_out = (void*)st;
  return SFMI_OK;
}
int sfmi_stream_destroy(void* stream) { return stream ? (int)hipStreamDestroy((hipStream_t)stream) : SFMI_EINVAL; }
}
