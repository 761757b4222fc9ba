/* libsfmi — C ABI of the MI355X-native ShapeFormer hot path (gfx950, HIP).
 *
 * Drop-in boundary.  The reference (QhelDIV/ShapeFormer) is pure Python over PyTorch
 * operators; it has no FFI of its own, so each entry point below names the reference function / third-party
 * operator call site it replaces (file:line under shapeformer/ in the reference tree).  A reference-side
 * ctypes binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - the caller owns every buffer; nothing is allocated or freed here; scratch sizes come from *_bytes/_floats;
 *   - all pointers are DEVICE pointers unless the function is marked [host];
 *   - `stream` is a hipStream_t (passed as void*); launches are asynchronous on it, no internal sync;
 *   - returns 0 on success, a negative SFMI_E* code for bad arguments, or a positive hipError_t;
 *   - no global mutable state (apart from the launch-shape knobs of sfmi_tune_set and immutable, once-initialised tables: the rocBLAS function pointers and one
 *     rocblas_handle per host thread, csrc/blas.hip); re-entrant for distinct streams and buffers;
 *   - layouts: feature grids are channels-last (B,D,H,W,C) f32; token buffers are int32; decode activations of the
 *     transformer are "fragment-packed" [ceil(M/16)][N/16][64][4] (see sfmi_decode_gemm_f32).
 */
#ifndef SFMI_H
#define SFMI_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SFMI_OK 0
#define SFMI_EINVAL (-1)
#define SFMI_ELDS (-4)    /* hipFuncSetAttribute refused the dynamic-LDS size a kernel needs (csrc/sgemm.hip: 73.7 KB) */
#define SFMI_ENOBLAS (-3) /* rocBLAS could not be bound (csrc/blas.hip); callers fall back to the tile kernels */

int sfmi_version(void);
/* one wavefront busy for `ticks` of the 100 MHz wall clock (<= 1 s) on `stream`: the stream-concurrency probe of the interleaved
 * decode chains (no reference counterpart: the reference runs one chain on one stream, shapeformer.py:85-132) */
int sfmi_stream_spin(long long ticks, void* stream);
/* [host] a HIP stream restricted to the compute units whose bit is set in mask[0 .. words) (bit i of word i / 32; consecutive bits go round
 * the 8 XCDs of gfx950) / its release.  Measurement plumbing for the CU-partition experiments (profiles/r06_overlap.md); no reference
 * counterpart and no use on the product path. */
int sfmi_stream_create_cumask(const unsigned* mask, int words, void** stream_out);
int sfmi_stream_destroy(void* stream);
/* [measurement plumbing] out[2 wg] = {XCC_ID, HW_ID} hardware registers of each of `blocks` workgroups (kept resident `ticks` x 10 ns): which
 * compute units a (masked) stream really uses */
int sfmi_hwid_probe(unsigned* out, int blocks, int threads, long long ticks, void* stream);
/* [host] launch-shape tuning knobs of the decode step (performance only - no knob changes a result bit unless its comment in
 * csrc/gpt.hip says so); read at launch time, so a captured hipGraph keeps the values it was captured with.  No reference
 * counterpart.  sfmi_tune_get returns -1 for an unknown name. */
int sfmi_tune_set(const char* name, int value);
int sfmi_tune_get(const char* name);
/* [host] number of successful sfmi_tune_set calls so far: callers that cache captured hipGraphs key them on it */
int sfmi_tune_generation(void);

/* ---- VQDIF encoder, per-point path: enc.py:95-140 (LocalPoolPointnet.forward up to scatter_mean), layers.py:39-48,
 *      vqdif/common.py:260-321, torch_scatter.scatter_max / scatter_mean call sites enc.py:70-74,103-110 ---------- */
size_t sfmi_enc_pack_floats(void);
/* [host] fc_pos (64,3)+(64); 5 blocks {fc_0 (32,64)+(32), fc_1 (32,32)+(32), shortcut (32,64)} stacked; fc_c (32,32)+(32) */
int sfmi_enc_pack_weights(const float* fc_pos_w, const float* fc_pos_b, const float* fc0_w, const float* fc0_b,
                          const float* fc1_w, const float* fc1_b, const float* sc_w, const float* fc_c_w,
                          const float* fc_c_b, float* out);
size_t sfmi_enc_workspace_bytes(int B, int T);
/* cloud (B,T,3) in [-1,1] -> per-cell mean grid (B,64,64,64,32) + latent occupancy mask (B,R,R,R) u8 [z][y][x] */
/* same, with per-point taps for stage-wise parity tests: stage1 (B,T,32) = blocks[1] output, stage4c (B,T,64) = [blocks[4] | fc_c] */
int sfmi_encode_points_tap_f32(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask, int* cell_out, void* workspace,
                               int B, int T, int R, float* tap_stage1, float* tap_stage4c, void* stream);
int sfmi_encode_points_f32(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask, int* cell_out,
                           void* workspace, int B, int T, int R, void* stream);
/* the same with the FIRST Downsampler convolution fused in (enc.py:66-93 generate_grid_features: scatter_mean -> Downsampler,
 * updown.py:101-118 first Conv3d(32 -> 64, k2, s2, no bias) (+ReLU)): cloud -> y (B,32,32,32,64) channels-last straight from the
 * per-cell sums; the dense 64^3 x 32 mean grid is never written or read.  w_down0 = that convolution's weights as packed by
 * sfmi_conv_pack_weight ([8 taps][64][32]). */
int sfmi_encode_points_down_f32(const float* cloud, const float* wpack, const float* w_down0, float* y, unsigned char* mask,
                                int* cell_out, void* workspace, int B, int T, int R, int relu, void* stream);

/* ---- Conv3d / GroupNorm / pooling: updown.py:79-132 (Downsampler, Upsampler), unet3d.py:79-144,195-293,449-474 -- */
int sfmi_conv_pack_weight(const float* w, int Cout, int Cin, int KS, float* out); /* [host] (Cout,Cin,k,k,k)->[tap][Cout][Cin] */
/* nn.Conv3d with fused input GroupNorm-apply (in_scale/in_shift (B,Cin) or NULL), nearest-x2 input upsample (up=1),
 * bias, activation (relu: 0 none, 1 ReLU, 2 GELU-erf).  x (B,Di,Hi,Wi,Cin) -> y (B,Do,Ho,Wo,Cout). */
int sfmi_conv3d_cl_f32(const float* x, const float* wT, const float* in_scale, const float* in_shift, const float* bias,
                       float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int KS, int stride, int pad, int up,
                       int relu, void* stream);
/* Conv3d(k3,p1) of a nearest-x2-upsampled grid as 8 parity-wise 2^3 convolutions of the low-resolution grid with pre-summed
 * weights (8/27 of the FLOPs; updown.py:119-132 Upsample + conv): [host] packer + launcher.  x (B,Di,Hi,Wi,Cin) -> y (B,2Di,..,Cout) */
int sfmi_conv_pack_weight_subpixel(const float* w, int Cout, int Cin, float* out);   /* out: 64*Cout*Cin floats */
int sfmi_conv3d_up2_cl_f32(const float* x, const float* wsub, const float* in_scale, const float* in_shift, const float* bias,
                           float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int relu, void* stream);
/* the two convolutions with the GroupNorm statistics of THEIR OUTPUT taken in the epilogue (updown.py:119-132: Conv, ReLU, GroupNorm - the
 * statistics pass over y disappears): partial (B, *splits, Cout, 2) f64 {sum, sum of squares} per shape and tile, consumed by
 * sfmi_groupnorm_coeffs_partial_f32 (V = voxels of y per shape).  Return SFMI_EINVAL BEFORE launching anything when the geometry has no
 * statistics-capable instance (64- or 32-channel stride-1 x-reuse forms): the caller then uses the plain entry + sfmi_groupnorm_coeffs_f32.
 * y is bit-identical to the plain entries'. */
int sfmi_conv3d_cl_stats_f32(const float* x, const float* wT, const float* in_scale, const float* in_shift, const float* bias, float* y, int B,
                             int Di, int Hi, int Wi, int Cin, int Cout, int KS, int stride, int pad, int up, int relu, double* partial, int* splits,
                             void* stream);
int sfmi_conv3d_up2_cl_stats_f32(const float* x, const float* wsub, const float* in_scale, const float* in_shift, const float* bias,
                                 float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int relu, double* partial, int* splits, void* stream);
int sfmi_groupnorm_coeffs_partial_f32(const double* partial, const float* gamma, const float* beta, float* scale, float* shift, int B, int V, int C,
                                      int S, int groups, float eps, void* stream);
int sfmi_gn_splits(int V);
/* nn.GroupNorm statistics -> scale/shift (B,C) with GN(x) == x*scale+shift; partial: B*sfmi_gn_splits(V)*C*2 doubles */
int sfmi_groupnorm_coeffs_f32(const float* x, const float* gamma, const float* beta, float* scale, float* shift,
                              double* partial, int B, int V, int C, int groups, float eps, void* stream);
int sfmi_affine_cl_f32(const float* x, const float* scale, const float* shift, float* y, int B, long long V, int C, void* stream);
int sfmi_maxpool2_cl_f32(const float* x, float* y, int B, int Do, int Ho, int Wo, int C, void* stream); /* unet3d.py:218-237 */
int sfmi_upcat_cl_f32(const float* skip, const float* low, float* y, int B, int D, int H, int W, int Cs, int Cu, void* stream); /* unet3d.py:268-293 */

/* ---- Vector quantiser: quantizer.py:19-30 (get_code), :47-51 (distances + argmax) ----------------------------- */
size_t sfmi_vq_pack_floats(int K, int D);
int sfmi_vq_pack_codebook(const float* W, int K, int D, float* out); /* [host] embedding.weight (K,D) */
int sfmi_vq_argmin_f32(const float* x, const float* packed, int* idx_out, float* dmin_out, long long N, int K, int D, void* stream);
int sfmi_vq_gather_f32(const float* W, const int* idx, float* out, long long N, int D, void* stream);

/* ---- Sparse (pos,code) tokens: models/common.py:20-23 (mode), :84-189 (dense<->sparse), vqdif.py:50-58 ---------- */
int sfmi_mode_i32(const int* idx, long long n, int K, int rows, int* hist, int* mode_out, void* stream);
int sfmi_apply_mask_i32(const int* idx, const unsigned char* mask, const int* mode, int* out, long long n, int mode_rows, void* stream);
int sfmi_dense2sparse_i32(const int* q, const int* mode, int mode_per_row, int* tokens, int* len, int B, int ncell, int Lpad,
                          int max_length, int end0, int end1, void* stream);
int sfmi_sparse2dense_i32(const int* tokens, const int* start, const int* len, const int* empty, int empty_per_row, int* dense,
                          int B, int ncell, int Lpad, int end0, int end1, void* stream);
/* AR_N.get_extra_indices + get_next_cond (representers.py:188-196, 432-442): c_pos (B,Lc), z_pos (B,Lz) -> extra (B,Lc+Lz) */
int sfmi_ar_n_extra_i32(const int* c_pos, const int* z_pos, int* extra, int B, int Lc, int Lz, int end0, void* stream);

/* ---- CondTupleGPT: transformer/mingpt.py:46-111 (Block), :256-310 (embeddings, two-stage tuple head),
 *      shapeformer.py:54-123 (sample_indices), representers.py:120-155,188-196,432-442, models/common.py:260-299 ---- */
/* prefill: rows (b,t), t < P, either as a (B,P) rectangle (rowoff NULL; t >= nval[b] is padding) or PACKED back to back
 * (rowoff (B+1) exclusive offsets, M = rowoff[B] rows: no work on padding).  Plain GEMM y[remap(m)] = act(x W^T + bias) + resid */
int sfmi_gemm_f32(const float* x, const float* W, const float* bias, const float* resid, float* y, long long M, int N, int K,
                  int act, long long out_group, long long out_group_stride, void* stream);
/* plain large GEMMs through rocBLAS, bound lazily with dlopen (csrc/blas.hip): row-major C = alpha op(A) op(B) + beta C;
 * transX != 0: the stored matrix is the transpose.  sfmi_gemm_blas_f32 == sfmi_gemm_f32 without the row remap. */
int sfmi_blas_available(void);
int sfmi_sgemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                   float beta, float* C, int ldc, void* stream);
int sfmi_gemm_blas_f32(const float* x, const float* W, const float* bias, const float* resid, float* y, int M, int N, int K, int act,
                       void* stream);
/* get_embeddings (mingpt.py:256-286) + AR_N extra index (representers.py:188-196,432-442) + ln1 of the first block, one
 * row per (b,t<P) (P == 0: one row per sequence at t = len[b]-1); extra_out receives the extra index used (for the backward) */
int sfmi_gpt_embed_f32(const float* E0, const float* E1, const float* Ex, const float* pos_emb, const float* cond_pos_emb,
                       const int* seq, const int* len, const int* Lc, const int* nval, const int* extra, int* extra_out,
                       float* resid_out, float* xn, const float* gamma, const float* beta, int B, int P, int D, int Lmax,
                       int end0, const int* rowoff, int M_packed, void* stream);
/* residual add + LayerNorm of Block.forward (mingpt.py:107-111): x = resid + sum_s part[s] + bias (+ tok_embs[0][next pos] at
 * the stage boundary, mingpt.py:294); resid_out = x, xn = LN(x) */
int sfmi_gpt_rowprep_f32(const float* resid_in, const float* part, const float* bias, const float* Eadd, const int* seq,
                         const int* len, const int* Lc, const int* nval, float* resid_out, float* xn, const float* gamma,
                         const float* beta, int S, int M, int P, int D, int Lmax, const int* rowoff, int B, void* stream);
/* CausalSelfAttention.forward over the rows of a prefix (mingpt.py:73-91) on f32 MFMA; also writes the (B,H,Lmax,64) KV caches */
int sfmi_gpt_attn_prefill_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                              int Lmax, const int* rowoff, float attn_drop_p, unsigned attn_drop_seed /* training: mingpt.py:85 */, void* stream);
/* the same launch; lse != NULL also receives the (B,H,P) row log-sum-exps of the scaled scores (the training forward: the backward
 * pass, sfmi_attn_bwd_lse_f32, starts from them instead of recomputing Q K^T) */
int sfmi_gpt_attn_prefill_lse_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                  int Lmax, const int* rowoff, float attn_drop_p, unsigned attn_drop_seed, float* lse, void* stream);
/* plain f32 GEMM on the matrix cores (csrc/sgemm.hip): row-major C (M,N;ldc) = op(A) op(B) (+C) (+bias[n]) -> act -> (+resid).
 * transA == 0: A stored (M,K;lda), else (K,M;lda); transB != 0: B stored (N,K;ldb) (nn.Linear weight), else (K,N;ldb).
 * Replaces the sgemm behind nn.Linear and its autograd (mingpt.py:46-111) in the prefill and the training step.
 * ws (optional, >= splits*M*N floats): split-K scratch for outputs with too few tiles (weight gradients), summed in slice order */
int sfmi_sgemm_mfma_splits(int M, int N, int K);
int sfmi_sgemm_mfma_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                        int ldc, int accumulate, const float* bias, int act, const float* resid, float* ws, long long ws_floats,
                        float drop_p, unsigned drop_seed, void* stream); /* drop_p > 0: nn.Dropout on the product (mingpt.py:90,105), counter-hash mask */
/* the same product, work-balanced ("stream-K", csrc/sgemm_sk.hip), for the SMALL GEMMs of the training step at the YAML's batch
 * size (M = B*L ~ 500 rows: 32-128 output tiles for 512 workgroup slots): K-chunks of tiles are dealt out evenly to 512 workgroups,
 * tiles cut between workgroups are finished in the same launch (write-through slabs + tickets, slices added in k order:
 * deterministic).  act: 0 none, 1 ReLU, 2 GELU(erf) (C2 != NULL also receives the pre-activation), 3 multiply by GELU'(aux[m][n]).
 * slab (>= sfmi_sgemm_sk_slab_floats() floats) / cnt (>= sfmi_sgemm_sk_cnt_ints(M, N) ints, zeroed ONCE): caller-owned scratch,
 * one pair per stream that runs these launches concurrently.  Replaces nn.Linear and its autograd, mingpt.py:46-111 */
int sfmi_sgemm_sk_tile(int M, int N, int K);          /* [host] 2 = 128 x 128 workgroup tiles, 1 = 64 x 64 */
long long sfmi_sgemm_sk_slab_floats(void);            /* [host] */
long long sfmi_sgemm_sk_cnt_ints(int M, int N);       /* [host] */
int sfmi_sgemm_sk_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, float* C2,
                      int ldc, int accumulate, const float* bias, int act, const float* aux, const float* resid, float drop_p,
                      unsigned drop_seed, float* slab, long long slab_floats, int* cnt, long long cnt_ints, void* stream);
int sfmi_ce_rows_f32(const float* logits, const int* target, float* loss, long long M, int V, int ld, void* stream); /* shapeformer.py:132-140 */
/* decode step (M = B <= 192 rows per launch: workgroups of up to 6 row tiles, more rows = row groups in grid.z; larger batches run as
 * several chains; packed x/out/resid hold sfmi_decode_gemm_padded_rows(M) rows) */
int sfmi_decode_gemm_padded_rows(int M);
size_t sfmi_skinny16_pack_floats(int N, int K);
int sfmi_skinny16_pack_weight(const float* W, int N, int K, float* out); /* [host] (N,K) -> [N/16][K/16][64][4] */
/* the same on the device, with the LayerNorm in front of the Linear folded in (mingpt.py:103-111: LN(x) W^T + b = rstd (x W'^T - mean c1) + c2,
 * W' = W diag(gamma), c1 = rowsum(W'), c2 = W beta + b; float64 row sums); gamma = beta = c1 = c2 = NULL: plain pack */
int sfmi_ln_fold_pack_f32(const float* W, const float* gamma, const float* beta, const float* bias, float* Wp, float* c1, float* c2, int N,
                          int K, void* stream);
size_t sfmi_decode_gemm_slab_floats(int M, int N, int S);
int sfmi_decode_gemm_f32(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid, float* out,
                         int M, int N, int K, int ldo, int ln, int act, int out_packed, int S, float* slab, int* cnt, void* stream);
/* the same with in-situ launch timing (bench.py `roofline`; no reference counterpart): prof = 3 device u64 {earliest start (armed as
 * ~0), sum of launch durations, launches} in ticks of the 100 MHz wall clock, one sink per chain; pblk = 1 zeroed device int per chain.
 * The launch's last workgroup adds (its end - earliest workgroup start).  prof == NULL: identical to sfmi_decode_gemm_f32. */
int sfmi_decode_gemm_prof_f32(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid, float* out,
                              int M, int N, int K, int ldo, int ln, int act, int out_packed, int S, float* slab, int* cnt, int* pblk,
                              unsigned long long* prof, void* stream);
/* embedding of the token at t = len[b]-1 into the fragment-packed residual buffer (input of the first decode step) */
int sfmi_gpt_embed_packed_f32(const float* E0, const float* E1, const float* Ex, const float* pos_emb, const float* cond_pos_emb,
                              const int* seq, const int* len, const int* Lc, float* resid, int B, int D, int Lmax, int end0,
                              void* stream);
/* CausalSelfAttention.forward for ONE new position per row against the KV cache (appends the new K,V row first).
 * shared_len (optional, device int): positions < shared_len[0] are read from ROW 0's cache by every row - the `sample_n` copies
 * of one condition (shapeformer.py:222-260) keep the condition's keys / values once */
int sfmi_gpt_attn_decode_f32(const float* qkv_packed, const float* unused, float* Kc, float* Vc, const int* len, float* y_packed,
                             int S, int B, int D, int H, int Lmax, const int* shared_len, void* stream);
/* the same behind the attention turnstile of the interleaved decode chains (no reference counterpart: the reference runs one
 * chain): sem = 3 device ints {next ticket, finished launches, gate time-outs} shared by all chains, zeroed by the caller while
 * nothing is in flight; blk = 1 zeroed device int per chain; at most `lanes` gated launches stream their KV cache at a time, in
 * ticket order.  Scheduling only - results are those of sfmi_gpt_attn_decode_f32.  sem == NULL: no turnstile.
 * prof (optional, needs blk): in-situ launch timing sink of this chain, as in sfmi_decode_gemm_prof_f32. */
int sfmi_gpt_attn_decode_gated_f32(const float* qkv_packed, float* Kc, float* Vc, const int* len, float* y_packed, int B, int D, int H,
                                   int Lmax, const int* shared_len, int* sem, int* blk, int lanes, unsigned long long* prof,
                                   void* stream);
/* one tuple element of one sampling step per row: sampling_masker (representers.py:120-155) + filter_sampling_logits /
 * sample_logits (models/common.py:260-299: temperature, top-k with ties, top-p) + inverse-CDF draw from counter-hash uniforms
 * indexed (step, tuple, row_offset + b) + best_in_first greedy row + log-prob + optional masked-logit history; writes the
 * token into seq, and the next GEMM input (tok_embs add / next position's embedding) into `resid` */
int sfmi_gpt_sample_f32(const float* part, int* seq, int* len, const int* Lc, float* logp, float* hist, const int* force,
                        float* resid, const float* E0, const float* E1, const float* Ex, const float* pos_emb, int D, int S,
                        int B, int V, int ldv, int Lmax, int tuple_i, int end0, int end1, int top_k, float top_p,
                        float temperature, int greedy_row0, int mask_invalid, int mask_completion, int max_steps,
                        unsigned seed, const unsigned* seed_dev /* optional device-resident seed (overrides `seed`) */, int advance,
                        int row_offset, int rows_total,
                        int step_offset /* tokens generated before this run (non-empty z_indices, shapeformer.py:60-70): step j = len - Lc - step_offset */,
                        void* stream);
/* ShapeRepresenter.sampling_masker alone (representers.py:120-155): logits (B,ldv) -> masked copy out (B,V); no draw, seq / len
 * are read only.  Row b holds len[b] complete tokens; for tuple_i == 1 the position just drawn sits at seq[b][len[b]][0]. */
int sfmi_gpt_mask_logits_f32(const float* logits, const int* seq, const int* len, const int* Lc, float* out, int B, int V, int ldv,
                             int Lmax, int tuple_i, int end0, int end1, int mask_invalid, int mask_completion, void* stream);
int sfmi_set_len_i32(int* len, const int* src, int B, int delta, void* stream);

/* ---- Training step of the transformer (csrc/train.hip): shapeformer.py:26-46,132-207 (forward/loss/AdamW groups),
 *      backward of mingpt.py:46-111; GEMM-shaped gradients reuse sfmi_gemm_f32 on transposed operands ------------------- */
int sfmi_transpose_f32(const float* in, float* out, int R, int C, int ldin, int Rpad, void* stream);
int sfmi_colsum_f32(const float* x, float* out, int M, int N, int ld, int accumulate, void* stream);          /* bias gradients */
int sfmi_colsum_slices(int M, int N);
/* two-stage form for tall inputs; scratch: sfmi_colsum_slices(M,N)*N floats */
int sfmi_colsum_ws_f32(const float* x, float* out, int M, int N, int ld, int accumulate, float* scratch, void* stream);
size_t sfmi_layernorm_bwd_scratch_floats(int M, int D);                                                        /* size of `stats` below */
int sfmi_gelu_f32(const float* x, float* y, long long n, void* stream);                                       /* mingpt.py:103 */
int sfmi_gelu_bwd_f32(const float* dy, const float* x, float* dx, long long n, void* stream);
int sfmi_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* dgamma,
                           float* dbeta, float* stats, int M, int D, void* stream);
/* the row part alone (dx and the (M,2) row statistics); the parameter sums then join a block's sfmi_col_reduce_f32 launch */
int sfmi_layernorm_bwd_rows_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats, int M,
                                int D, void* stream);
/* the same; dx2 != NULL also receives nn.Dropout(dx) with the counter-hash mask of `drop_seed` (the backward of a forward dropout
 * on this tensor, mingpt.py:90,105, fused instead of a separate sfmi_dropout_f32 launch) */
int sfmi_layernorm_bwd_rows_drop_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats,
                                     float* dx2, float drop_p, unsigned drop_seed, int M, int D, void* stream);
/* up to 8 column reductions over M rows in ONE launch: kind 0 = bias gradient of a Linear layer (column sums of dY, mingpt.py:46-111),
 * kind 1 = LayerNorm parameter gradients (dgamma -> out, dbeta -> out2).  Replaces 8 colsum + 4 LayerNorm-parameter launches per
 * transformer block of the backward pass.  part / cnt: scratch for tall inputs (cnt zeroed once). */
int sfmi_col_reduce_slices(int M);                                   /* [host] */
long long sfmi_col_reduce_part_floats(int M, int total_cols);        /* [host] */
int sfmi_col_reduce_f32(int njobs, const int* kind, const float* const* a, const float* const* x, const float* const* stats, float* const* out,
                        float* const* out2, const int* N, const int* ld, int M, int accumulate, float* part, long long part_floats, int* cnt,
                        long long cnt_ints, void* stream);
int sfmi_ce_fwd_bwd_f32(const float* logits, const int* target, float* loss_rows, float* dlogits, int M, int V, int ld, int L,
                        int t0, float scale, void* stream);                                                    /* shapeformer.py:132-140 */
int sfmi_attn_bwd_f32(const float* qkv, const float* y, const float* dy, float* lse /*2*B*H*L floats scratch*/, float* dqkv,
                      int B, int L, int D, int H, float attn_drop_p, unsigned attn_drop_seed /* the forward's mask */, void* stream);
/* the same gradients from the forward's row log-sum-exps (sfmi_gpt_attn_prefill_lse_f32): a row-sum launch (delta = sum_d dO O, (B,H,L)
 * scratch) + ONE launch that runs the dQ blocks and the dK/dV blocks side by side (2 x the workgroups: fills the chip at batch 1) */
/* training forward of the same attention for launches too small to fill the chip with 64-row tiles (B * H * ceil(L / 64) <= 128; batch 1
 * of the YAML): 32-row tiles x two key-block groups per workgroup, merged online-softmax states; writes y (B*L, D) and the (B,H,L) row
 * log-sum-exps.  SFMI_EINVAL for larger launches (use sfmi_gpt_attn_prefill_lse_f32) */
int sfmi_attn_train_fwd_small_f32(const float* qkv, float* y, float* lse, int B, int L, int D, int H, float attn_drop_p,
                                  unsigned attn_drop_seed, void* stream);
int sfmi_attn_bwd_lse_f32(const float* qkv, const float* y, const float* dy, const float* lse, float* delta /*B*H*L floats scratch*/,
                          float* dqkv, int B, int L, int D, int H, float attn_drop_p, unsigned attn_drop_seed, void* stream);
/* nn.Dropout(p) forward == backward on a flat tensor: y = x * mask / (1-p), mask_i = hash(seed, i) >= p (mingpt.py:90,105,218,292) */
int sfmi_dropout_f32(const float* x, float* y, long long n, float p, unsigned seed, void* stream);                                                                           /* mingpt.py:73-91 */
int sfmi_embed_scatter_f32(const float* dx, const int* idx, long long* acc, long long M, int D, void* stream);
int sfmi_fixed_to_float_f32(const long long* acc, float* out, long long n, int accumulate, void* stream);
int sfmi_add_f32(const float* a, const float* b, float* out, long long n, void* stream);
int sfmi_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, void* stream);                                                /* shapeformer.py:198-206 */
/* the same update over a table of tensors in one launch (device tables; per-tensor weight decay = the two AdamW groups) */
int sfmi_adamw_multi_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                         const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                         float eps, int step, void* stream);
/* optimizer-sharded data parallelism (north_star: reduce-scatter -> update of the rank's 1/N shard -> all-gather; the reference
 * delegates this to Lightning's DDP, trainer.py:22,93): the same update over chunk tables that cover only this rank's shard, the
 * updated parameters ALSO written to pflat[flat index] (may alias g: the gradient buffer becomes the all-gather send buffer);
 * sfmi_unflatten_multi_f32 then copies the all-gathered flat buffer back into the parameter tensors */
int sfmi_adamw_multi_shard_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                               const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                               float eps, int step, float* pflat, void* stream);
int sfmi_unflatten_multi_f32(float* const* p, const long long* foff, const int* ctensor, const long long* coff, const int* clen,
                             int nchunks, const float* flat, void* stream);

/* ---- the training step captured in a hipGraph (no reference counterpart: Lightning enqueues every step from the host).  A captured launch
 *      keeps its arguments, so what changes from step to step is read from DEVICE memory at run time: the dropout seed of a site
 *      (mingpt.py:62-63,85,90,105,292) through `drop_seed_dev` (NULL: the by-value seed, as in the plain entry points), AdamW's bias
 *      corrections and learning rate through `bc_dev` = {1 - beta1^t, 1 - beta2^t, lr} (sfmi_adamw_bias_corrections forms the first two exactly
 *      as the by-value path).
 *      Same kernels, same arithmetic: a replayed step is bit-identical to the eager one. */
int sfmi_sgemm_sk_sd_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, float* C2,
                         int ldc, int accumulate, const float* bias, int act, const float* aux, const float* resid, float drop_p,
                         unsigned drop_seed, const unsigned* drop_seed_dev, float* slab, long long slab_floats, int* cnt, long long cnt_ints,
                         void* stream);
int sfmi_gpt_attn_prefill_lse_sd_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                     int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, float* lse,
                                     void* stream);
int sfmi_attn_train_fwd_small_sd_f32(const float* qkv, float* y, float* lse, int B, int L, int D, int H, float drop_p, unsigned drop_seed,
                                     const unsigned* drop_seed_dev, void* stream);
int sfmi_attn_bwd_lse_sd_f32(const float* qkv, const float* y, const float* dy, const float* lse, float* delta, float* dqkv, int B, int L,
                             int D, int H, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, void* stream);
int sfmi_layernorm_bwd_rows_drop_sd_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats,
                                        float* dx2, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, int M, int D, void* stream);
int sfmi_dropout_sd_f32(const float* x, float* y, long long n, float p, unsigned seed, const unsigned* seed_dev, void* stream);
int sfmi_adamw_bias_corrections(float beta1, float beta2, int step, float* out2); /* [host] */
int sfmi_adamw_multi_shard_bc_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                                  const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                                  float eps, int step, const float* bc_dev, float* pflat, void* stream);

/* ---- Implicit decoder SDF/occupancy query: dec.py:62-100 (grid_sample + 5-block conditioned MLP), layers.py:39-48 - */
size_t sfmi_sdf_pack_floats(void);
int sfmi_sdf_pack_weights(const float* fc_p_w, const float* fc_p_b, const float* fc_c_w, const float* fc_c_b,
                          const float* fc0_w, const float* fc0_b, const float* fc1_w, const float* fc1_b,
                          const float* fc_out_w, const float* fc_out_b, float* out); /* [host] */
int sfmi_sdf_query_f32(const float* xyz, const float* grid_cl, const float* wpack, float* out, int B, long long N, int G,
                       int apply_sigmoid, void* stream);
/* structured Q^3 lattice of nputil.makeGrid 'ij' (xgutils/nputil.py:618-654) from a Q-entry f32 axis table */
int sfmi_sdf_query_grid_f32(const float* axis, int Q, const float* grid_cl, const float* wpack, float* out, int B, int G,
                            int apply_sigmoid, void* stream);
/* planes x0 <= ix < x1 of that lattice (ix = the slowest index): out (B, (x1 - x0) Q^2); the slabs concatenate to the whole-lattice result bit
 * for bit (the z-slab split of one shape over the ranks; shapeformer.py:382-391, dec.py:62-100) */
int sfmi_sdf_query_grid_slab_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* wpack, float* out, int B, int G,
                                 int apply_sigmoid, void* stream);
/* the same on a decoder grid whose last GroupNorm (updown.py:119-132) has not been applied: its per-(shape, channel) affine aff_scale / aff_shift
 * (B,32) is applied to the interpolated features inside the kernel (the trilinear weights of the 'border' gather sum to one); both NULL: final grid */
int sfmi_sdf_query_grid_aff_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* aff_scale, const float* aff_shift,
                                const float* wpack, float* out, int B, int G, int apply_sigmoid, void* stream);
/* nputil.sigmoid over stored logits (vqdif.py:262, shapeformer.py:388): y = 1 / (1 + exp(-x)), the fused epilogue's expression; may alias */
int sfmi_sigmoid_f32(const float* x, float* y, long long n, void* stream);

/* ---- Training step of the VQDIF autoencoder (csrc/train_vqdif.hip): vqdif.py:78-137 (forward, VQLoss,
 *      Adam), quantizer.py:68-89 (EMA codebook, straight-through), backward of enc.py:66-140, updown.py:79-132,
 *      unet3d.py:79-293,449-474, dec.py:62-100.  Input gradients of convolutions / linears reuse sfmi_conv3d_cl_f32 /
 *      sfmi_gemm_f32 on tap-flipped / transposed weights. ------------------------------------------------------------- */
int sfmi_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, void* stream);
int sfmi_lincomb_f32(float a, const float* x, float b, const float* y, float* out, long long n, void* stream);   /* a x + b y (y may be NULL) */
/* out = A[b,c] u + Bc[b,c] v + Cc[b,c] on (B,V,C): GroupNorm backward is affine in (dy, x) per (sample, channel) */
int sfmi_affine2_cl_f32(const float* u, const float* v, const float* A, const float* Bc, const float* Cc, float* out, int B,
                        long long V, int C, void* stream);
/* partial (B,S,C,2) doubles: per slice sums of dy and dy*x per (sample, channel) */
int sfmi_chan_dot_stats_f32(const float* dy, const float* x, double* partial, int B, long long V, int C, int S, void* stream);
int sfmi_upsample2_cl_f32(const float* x, float* y, int B, int D, int H, int W, int C, void* stream);            /* nearest x2 */
int sfmi_sumpool2_cl_f32(const float* dy, float* dx, int B, int Do, int Ho, int Wo, int Ctot, int c0, int Cs, void* stream);
int sfmi_maxpool2_bwd_cl_f32(const float* x, const float* y, const float* dy, float* dx, int B, int Do, int Ho, int Wo, int C,
                             void* stream);
int sfmi_cells_f32(const float* cloud, int* cell, float* p_half, int B, int T, int G, void* stream);             /* common.py:260-321 */
/* torch_scatter.scatter_max + gather (enc.py:95-112): keys (B,ncell,C) int32 pre-filled with 0x80 bytes */
int sfmi_cell_max_f32(const float* net, const int* cell, int* keys, float* out, int B, int T, long long ncell, int C, int ldo, int co,
                      void* stream);
int sfmi_cell_scatter_add_f32(const float* src, const int* cell, long long* acc, int* count, int B, int T, long long ncell, int C,
                              int lds, int cs, void* stream);
int sfmi_cell_max_bwd_f32(const float* net, const int* keys, const long long* acc, const int* cell, float* dnet, int B, int T,
                          long long ncell, int C, int ldd, int accumulate, void* stream);
int sfmi_cell_mean_f32(const long long* acc, const int* count, float* grid, int B, long long ncell, int C, void* stream);  /* enc.py:66-75 */
int sfmi_cell_mean_bwd_f32(const float* dgrid, const int* count, const int* cell, float* dc, int B, int T, long long ncell, int C,
                           void* stream);
int sfmi_trilinear_cl_f32(const float* xyz, const float* grid, float* out, int B, long long N, int G, int C, void* stream);   /* dec.py:62-68 */
int sfmi_trilinear_bwd_cl_f32(const float* xyz, const float* dout, long long* acc, int B, long long N, int G, int C, void* stream);
int sfmi_bce_logits_f32(const float* logits, const float* label, float* loss_rows, float* dlogits, long long n, float scale,
                        void* stream);                                                                             /* vqdif.py:151-167 */
int sfmi_vq_stats_f32(const float* x, const int* idx, long long* sums, int* counts, long long rows, int D, void* stream);
int sfmi_vq_ema_update_f32(float* N, float* z_avg, float* emb, const float* counts, const float* sums, int K, int D, float gamma,
                           float eps, void* stream);                                                               /* quantizer.py:68-86 */
/* part (nsplit, KS^3, Cout, Cin): per-slice sums of dY[r][co] * X[shifted r][ci]; reduce with sfmi_colsum_f32 */
int sfmi_conv3d_wgrad_f32(const float* dy, const float* x, float* part, int B, int Di, int Hi, int Wi, int Cin, int Cout, int KS,
                          int stride, int pad, int ldy, int ldx, int nsplit, void* stream);

/* ---- Iso-surface extraction: xgutils/geoutil.py:175-233 array2mesh -> mcubes.marching_cubes
 *      (PyMCubes, third party) at thresh .5 over the decoded occupancy grid; call sites shapeformer.py:355-356 (vis_ind),
 *      xgutils/vis/npfvis.py:88-98 (plot_3d_recon).  Two passes so the caller can size the outputs. ------------------- */
size_t sfmi_mc_workspace_bytes(int B, int Q);
/* occ (B,Q,Q,Q) f32; offsets (device, 2*(B+1) ints): exclusive vertex offsets [B+1] then triangle offsets [B+1] */
int sfmi_mc_count_f32(const float* occ, float iso, int B, int Q, void* workspace, int* offsets, void* stream);
/* verts (V,3) f32 = index/(Q-1)*(hi-lo)+lo (array2mesh's mapping onto the bbox of coords); faces (T,3) int32, local per shape */
int sfmi_mc_emit_f32(const float* occ, float iso, int B, int Q, const void* workspace, const int* offsets, float lo0, float lo1,
                     float lo2, float hi0, float hi1, float hi2, float* verts, int* faces, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SFMI_H */
