// libsfmi: version + small host utilities of the C ABI (include/sfmi.h).
#include "sfmi_common.h"
#include <string>

SfmiTune g_sfmi_tune = {0, 4, 16, 0, 512, 1, 0, 0, 2, 512, 0, 0, 0, 1};
static int g_sfmi_tune_generation = 0;

// One wavefront that waits `ticks` of the constant 100 MHz wall clock: the probe `shapeformer_amd/gpt.py:_chain_streams` uses to
// find HIP streams that really run concurrently (streams mapped to one hardware queue serialise; the runtime hands out only a few
// hardware queues and assigns them by first use, so which streams share one depends on what the process did before).
__global__ void stream_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
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
  if (n == "sk_grid") return t.sk_grid;
  if (n == "sk_tile") return t.sk_tile;
  if (n == "sk_loop") return t.sk_loop;
  if (n == "sk_stagger") return t.sk_stagger;
  return -1;
}
int sfmi_tune_generation(void) { return g_sfmi_tune_generation; }
int sfmi_stream_spin(long long ticks, void* stream) {
  if (ticks < 0 || ticks > 100000000LL) return SFMI_EINVAL;   // <= 1 s
  hipLaunchKernelGGL(stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
}
