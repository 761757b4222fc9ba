// libsfmi: version + small host utilities of the C ABI (include/sfmi.h).
#include "sfmi_common.h"
#include <string>

SfmiTune g_sfmi_tune = {0, 4, 16, 0, 512, 1, 0, 0, 2, 512, 0, 0, 0, 1, 0};
static int g_sfmi_tune_generation = 0;

// One wavefront that waits `ticks` of the constant 100 MHz wall clock: the probe `shapeformer_amd/gpt.py:_chain_streams` uses to
// find HIP streams that really run concurrently (streams mapped to one hardware queue serialise; the runtime hands out only a few
// hardware queues and assigns them by first use, so which streams share one depends on what the process did before).
__global__ void stream_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Where did each workgroup of a launch run?  out[wg] = {XCC_ID register, HW_ID register} (gfx9: HW_ID bits 8-11 CU, 12 SH, 13-15 SE).
__global__ void hwid_probe_kernel(unsigned* out, long long ticks) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
  }
  const long long t0 = wall_clock64();      // stay resident for a while so that the grid spreads over every CU the stream may use
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {
int sfmi_version(void) { return 100; }
int sfmi_tune_set(const char* name, int value) {
  if (!name) return SFMI_EINVAL;
  const std::string n(name);
  SfmiTune& t = g_sfmi_tune;
  if (n == "attn_blocks" && value >= 0) t.attn_blocks = value;
  else if (n == "attn_unroll" && (value == 2 || value == 4 || value == 8 || value == 16)) t.attn_unroll = value;
  else if (n == "attn_waves" && (value == 4 || value == 8 || value == 16)) t.attn_waves = value;
  else if (n == "attn_lds_pad" && value >= 0 && value <= 140 * 1024) t.attn_lds_pad = value;
  else if (n == "sdf_blocks" && value >= 1 && value <= 512) t.sdf_blocks = value;
  else if (n == "dgemm_nt2" && value >= 0 && value <= 2) t.dgemm_nt2 = value;
  else if (n == "dgemm_nw" && (value == 0 || value == 4 || value == 8 || value == 16)) t.dgemm_nw = value;
  else if (n == "dgemm_un" && value >= 0 && value <= 8) t.dgemm_un = value;
  else if (n == "conv_xreuse" && value >= 0 && value <= 3) t.conv_xreuse = value;
  else if (n == "enc_fused" && (value == 0 || value == 1)) t.enc_fused = value;
  else if (n == "dgemm_prio" && value >= 0 && value <= 3) t.dgemm_prio = value;
  else if (n == "sk_grid" && value >= 256 && value <= 1024 && value % 256 == 0) t.sk_grid = value;
  else if (n == "sk_tile" && value >= 0 && value <= 2) t.sk_tile = value;
  else if (n == "sk_loop" && value >= 0 && value <= 1) t.sk_loop = value;
  else if (n == "sk_stagger" && value >= 0 && value <= 64) t.sk_stagger = value;
  else return SFMI_EINVAL;
  ++g_sfmi_tune_generation;
  return SFMI_OK;
}
int sfmi_tune_get(const char* name) {
  if (!name) return -1;
  const std::string n(name);
  const SfmiTune& t = g_sfmi_tune;
  if (n == "attn_blocks") return t.attn_blocks;
  if (n == "attn_unroll") return t.attn_unroll;
  if (n == "attn_waves") return t.attn_waves;
  if (n == "attn_lds_pad") return t.attn_lds_pad;
  if (n == "sdf_blocks") return t.sdf_blocks;
  if (n == "dgemm_nt2") return t.dgemm_nt2;
  if (n == "dgemm_nw") return t.dgemm_nw;
  if (n == "dgemm_un") return t.dgemm_un;
  if (n == "conv_xreuse") return t.conv_xreuse;
  if (n == "enc_fused") return t.enc_fused;
  if (n == "dgemm_prio") return t.dgemm_prio;
  if (n == "sk_grid") return t.sk_grid;
  if (n == "sk_tile") return t.sk_tile;
  if (n == "sk_loop") return t.sk_loop;
  if (n == "sk_stagger") return t.sk_stagger;
  return -1;
}
int sfmi_tune_generation(void) { return g_sfmi_tune_generation; }
// A HIP stream whose kernels may only be placed on the compute units whose bit is set in `mask` (hipExtStreamCreateWithCUMask; bit i of
// word i / 32).  On gfx950 consecutive bits go round the 8 XCDs, so the first n bits are n / 8 CUs of every XCD.  Measurement plumbing of
// tools/ar_sweep.py (profiles/r06_overlap.md: at how many CUs does the decode attention's KV stream stop scaling?); the product path
// creates no masked stream.
int sfmi_stream_create_cumask(const unsigned* mask, int words, void** stream_out) {
  if (!mask || words <= 0 || !stream_out) return SFMI_EINVAL;
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) return (int)e;
  *stream_out = (void*)st;
  return SFMI_OK;
}
// [measurement plumbing] out (2 * blocks) u32: {XCC_ID, HW_ID} of each of `blocks` workgroups of `threads` threads that stay resident `ticks` x 10 ns
int sfmi_hwid_probe(unsigned* out, int blocks, int threads, long long ticks, void* stream) {
  if (!out || blocks <= 0 || threads <= 0 || threads > 1024 || ticks < 0 || ticks > 10000000LL) return SFMI_EINVAL;
  hipLaunchKernelGGL(hwid_probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, ticks);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_stream_destroy(void* stream) {
  if (!stream) return SFMI_EINVAL;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? SFMI_OK : (int)e;
}
int sfmi_stream_spin(long long ticks, void* stream) {
  if (ticks < 0 || ticks > 100000000LL) return SFMI_EINVAL;   // <= 1 s
  hipLaunchKernelGGL(stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
}
