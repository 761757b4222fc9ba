// Integer token packing for gfx950: whole-batch mode ("empty" code), dense (B,R,R,R) code grid <-> sparse
// (pos,val) token rows.  Bit-exact restatement of shapeformer/models/common.py:20-23,84-189 and
// vqdif.py:50-58 (mode outside the occupancy mask) as order-preserving scan/compact kernels instead of
// nonzero/unique_consecutive/cumsum/index_put chains.  Rows are RAGGED on device: row b holds len[b]
// tokens (the last one is the end-token pair) inside a fixed (B,Lpad,2) int32 buffer padded with end
// tokens, so no host sync is needed to size tensors (the reference syncs to build (B,Lmax+1,2)).
#include "sfmi_common.h"

// rows > 1: one histogram per row of n/rows elements (per-shape mode, the reference's batch-1 inference)
// Wave-aggregated: the lanes of a wavefront that hold the same (row, value) send ONE atomic with their count.  The mode of a code
// grid is the "empty" code of 97 % of its cells - with one atomic per element they all hit one address and the launch took 1.3 ms
// for 131 072 elements (9 % of BASELINE config 2's batch); the counts are integers, so the result is the same bit for bit.
__global__ void hist_kernel(const int* __restrict__ idx, int* __restrict__ hist, long long n, int K, long long per_row) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int bin = -1;                                  // rows * K < 2^31 (host-checked)
  if (i < n) {
    const int v = idx[i];
    if (v >= 0 && v < K) bin = (int)(i / per_row) * K + v;
  }
  bool todo = bin >= 0;
  unsigned long long m;
  while ((m = __ballot(todo)) != 0ull) {
    const int leader = __ffsll((long long)m) - 1;
    const int lb = __shfl(bin, leader, 64);
    const bool mine = todo && bin == lb;
    const unsigned long long mm = __ballot(mine);
    if (lane == leader) atomicAdd(&hist[lb], __popcll(mm));
    if (mine) todo = false;
  }
}

// most frequent value, smallest on ties (torch.unique+argmax / torch.mode semantics)
__global__ __launch_bounds__(256) void mode_select_kernel(const int* __restrict__ hist_all, int K, int* __restrict__ mode_all) {
  __shared__ int sc[256], sv[256];
  const int* hist = hist_all + (long long)blockIdx.x * K;
  int* mode = mode_all + blockIdx.x;
  int bc = -1, bv = 0;
  for (int v = threadIdx.x; v < K; v += 256) {
    int c = hist[v];
    if (c > bc) { bc = c; bv = v; }  // ascending v per thread -> keeps smallest v on ties
  }
  sc[threadIdx.x] = bc; sv[threadIdx.x] = bv;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      int oc = sc[threadIdx.x + s], ov = sv[threadIdx.x + s];
      if (oc > sc[threadIdx.x] || (oc == sc[threadIdx.x] && ov < sv[threadIdx.x])) { sc[threadIdx.x] = oc; sv[threadIdx.x] = ov; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *mode = sv[0];
}

// vqdif.py:54-57: quant_ind = mode everywhere, raw index inside the occupancy mask
__global__ void apply_mask_kernel(const int* __restrict__ idx, const unsigned char* __restrict__ mask,
                                  const int* __restrict__ mode, int* __restrict__ out, long long n, long long per_row) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mask[i] ? idx[i] : mode[i / per_row];
}

// common.py:151-168 + 84-123: row b = all cells != mode in ascending pos, then the end-token pair.
__global__ __launch_bounds__(256) void dense2sparse_kernel(const int* __restrict__ q, const int* __restrict__ mode_p,
                                                           int* __restrict__ tokens /*(B,Lpad,2)*/, int* __restrict__ len,
                                                           int ncell, int Lpad, int max_length, int end0, int end1,
                                                           int mode_stride) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mode = mode_p[b * mode_stride];
  const int keep_max = max_length - 1;  // room for the forced end token (common.py:118-122)
  int* row = tokens + (long long)b * Lpad * 2;
  for (int i = tid; i < Lpad; i += 256) { row[2 * i] = end0; row[2 * i + 1] = end1; }
  if (tid == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < ncell; c0 += 256) {
    const int pos = c0 + tid;
    const int v = pos < ncell ? q[(long long)b * ncell + pos] : mode;
    const bool f = v != mode;
    const unsigned long long bal = __ballot(f);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int r = off + pre;
    if (f && r < keep_max && r < Lpad) { row[2 * r] = pos; row[2 * r + 1] = v; }
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) {
    int cnt = base;
    int l = cnt + 1;
    if (l > max_length) l = max_length;
    if (l > Lpad) l = Lpad;
    len[b] = l;
  }
}

// common.py:126-140 + 171-189: drop rows where ANY... no: keep rows where BOTH elements differ from their
// end token; dense filled with *empty, dense[b][pos] = val.
__global__ void sparse2dense_kernel(const int* __restrict__ tokens, const int* __restrict__ start, const int* __restrict__ len,
                                    const int* __restrict__ empty_p, int* __restrict__ dense, int ncell, int Lpad, int end0,
                                    int end1, int empty_stride) {
  const int b = blockIdx.x;
  const int empty = empty_p[b * empty_stride];
  for (int i = threadIdx.x; i < ncell; i += blockDim.x) dense[(long long)b * ncell + i] = empty;
  __syncthreads();
  const int r0 = start ? start[b] : 0;
  const int n = len ? min(len[b], Lpad) : Lpad;
  for (int r = r0 + threadIdx.x; r < n; r += blockDim.x) {
    const int pos = tokens[((long long)b * Lpad + r) * 2], val = tokens[((long long)b * Lpad + r) * 2 + 1];
    if (pos != end0 && val != end1 && pos >= 0 && pos < ncell) dense[(long long)b * ncell + pos] = val;
  }
}

// AR_N.get_extra_indices (representers.py:188-196) + get_next_cond (:432-442): condition tokens carry their own position,
// generated tokens the first condition position > their own (searchsorted right=True over the ascending condition row, whose
// last entry is the end token), end tokens map to the end token.  One thread per (b, t); pos tensors (B,Lc) / (B,Lz) int32.
__global__ void ar_n_extra_kernel(const int* __restrict__ c_pos, const int* __restrict__ z_pos, int* __restrict__ extra, int B, int Lc,
                                  int Lz, int end0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Lt = Lc + Lz;
  if (i >= (long long)B * Lt) return;
  const int b = (int)(i / Lt), t = (int)(i - (long long)b * Lt);
  const int* cs = c_pos + (long long)b * Lc;
  int e;
  if (t < Lc) e = cs[t];
  else {
    const int z = z_pos[(long long)b * Lz + (t - Lc)];
    if (z == end0) e = end0;
    else {
      int lo = 0, hi = Lc;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (cs[mid] > z) hi = mid; else lo = mid + 1; }
      e = cs[lo < Lc ? lo : Lc - 1];
    }
  }
  extra[i] = e;
}

extern "C" {

// replaces pth_get_mode / torch.mode (common.py:20-23,155). hist: K ints of workspace.
// rows == 1: whole-tensor mode (vqdif.py:53 / common.py:155); rows > 1: one mode per row of n/rows elements.
// hist: rows*K ints of workspace; mode_out: rows ints.
int sfmi_mode_i32(const int* idx, long long n, int K, int rows, int* hist, int* mode_out, void* stream) {
  if (!idx || !hist || !mode_out || n <= 0 || K <= 0 || rows <= 0 || n % rows || (long long)rows * K >= (1ll << 31)) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(hist, 0, (size_t)K * rows * 4, st);
  hipLaunchKernelGGL(hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, hist, n, K, n / rows);
  hipLaunchKernelGGL(mode_select_kernel, dim3(rows), dim3(256), 0, st, hist, K, mode_out);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces vqdif.py:54-57
int sfmi_apply_mask_i32(const int* idx, const unsigned char* mask, const int* mode, int* out, long long n, int mode_rows,
                        void* stream) {
  if (!idx || !mask || !mode || !out || n <= 0 || mode_rows <= 0 || n % mode_rows) return SFMI_EINVAL;
  hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, mask, mode, out, n,
                     n / mode_rows);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces batch_dense2sparse/unpack_sparse (common.py:84-123,151-168), ragged rows (see header)
int sfmi_dense2sparse_i32(const int* q, const int* mode, int mode_per_row, int* tokens, int* len, int B, int ncell, int Lpad,
                          int max_length, int end0, int end1, void* stream) {
  if (!q || !mode || !tokens || !len || B <= 0 || ncell <= 0 || Lpad <= 0 || max_length <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(dense2sparse_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, q, mode, tokens, len, ncell, Lpad,
                     max_length, end0, end1, mode_per_row ? 1 : 0);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces pack_sparse + batch_sparse2dense (common.py:126-140,171-189); len may be NULL (use all Lpad rows)
// rows r in [start[b] (or 0), len[b] (or Lpad)) of row b are scattered; empty: scalar or per-row.
int sfmi_sparse2dense_i32(const int* tokens, const int* start, const int* len, const int* empty, int empty_per_row, int* dense,
                          int B, int ncell, int Lpad, int end0, int end1, void* stream) {
  if (!tokens || !empty || !dense || B <= 0 || ncell <= 0 || Lpad <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(sparse2dense_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tokens, start, len, empty, dense, ncell, Lpad,
                     end0, end1, empty_per_row ? 1 : 0);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces AR_N.get_extra_indices / get_next_cond (representers.py:188-196, 432-442): extra (B, Lc+Lz) int32
int sfmi_ar_n_extra_i32(const int* c_pos, const int* z_pos, int* extra, int B, int Lc, int Lz, int end0, void* stream) {
  if (!c_pos || !extra || B <= 0 || Lc <= 0 || Lz < 0 || (Lz > 0 && !z_pos)) return SFMI_EINVAL;
  const long long n = (long long)B * (Lc + Lz);
  hipLaunchKernelGGL(ar_n_extra_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c_pos, z_pos, extra, B, Lc, Lz,
                     end0);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
