/*
 * inferix_hip.h — C-ABI of libinferix_hip.so: the MI355X (gfx950) kernels of the
 * semi-autoregressive block-diffusion denoising step (causal Wan DiT) behind
 * Inferix's generator / attention / KV-cache interfaces.
 *
 * The reference (alibaba-damo-academy/Inferix) has no FFI layer: its hot path is
 * Python calling un-vendored CUDA libraries.  Each entry point below names the
 * reference Python call site(s) it replaces (paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + sizes, no torch types; caller owns every buffer
 *     (PyTorch caching allocator or the KV-cache manager); no hidden allocation.
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *     synchronises.
 *   - returns 0 on success, a negative IFX_E* code otherwise; never throws.
 *     ifx_last_error() returns a thread-local human-readable message.
 *   - bf16 = IEEE bfloat16 stored as uint16_t; activations are token-major
 *     [rows, channels] with the channel dimension contiguous.
 *   - rounding points follow the reference's bf16 module boundaries (documented
 *     per function) so results match its PyTorch path to bf16 rounding noise.
 */
#ifndef INFERIX_HIP_H
#define INFERIX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFX_OK 0
#define IFX_EINVAL (-1)   /* bad argument (null pointer, unsupported shape) */
#define IFX_ELAUNCH (-2)  /* HIP launch error */
#define IFX_EUNSUP (-3)   /* configuration not built into this library */

typedef uint16_t ifx_bf16;

/* library identity ------------------------------------------------------- */
/* The MINOR number is the ABI generation: it changes whenever an argument struct gains a field or an entry point changes its
 * signature (0.2: ifx_kv_view.seg_split / seg_delta, ifx_rope_grid.q_scale; 0.3: ifx_magi_head_prep_desc.rope_half, gemm_small_split; 0.4: ifx_gemm_q8_quant_out, ifx_layernorm_quant_static; 0.6: ifx_get_option, ifx_device_error, bounded device waits; 0.7: ifx_conv3d_desc.in_planar, IFX_NORM_OUT_PLANAR, conv_variant, attn_debug_counters).  Callers built against another minor
 * must not call in: zero-initialise every struct (new fields default to 0 = off) and compare IFX_ABI_MINOR with
 * (ifx_version() >> 8) & 255 at load time, as inferix_amd/_hip.py does. */
#define IFX_ABI_MINOR 7
int ifx_version(void);                 /* (major<<16)|(minor<<8)|patch */
const char* ifx_last_error(void);      /* thread-local, never NULL */
const char* ifx_arch(void);            /* "gfx950" */
/* Kernel-selection override for benchmarking and tests (default 0 = choose by shape):
 *   "gemm_variant": 1 register-staged 128x128; LDS-DMA tiles 2 = 256x128x64, 3 = 128x128x64 (two workgroups per CU),
 *                   4 = 64x64, 5 = 256x256x64, 6 = 128x64, 7 = 256x128x32 (two per CU), 8 = 128x128x32 (eight waves),
 *                   9 / 10 = warp-specialised 256x128 / 128x128, 11 = 256x256x32 four stages,
 *                   12 / 13 / 14 = 64x64 (four K-groups), 64x64 (two), 128x128 (two): split-K between the wave groups of one
 *                   workgroup, what 0 = auto picks for launches of at most one workgroup per CU; K / 64 must divide by the groups;
 *                   15 = 128x64 two stages (three per CU), 16 / 17 = 64x128 two / three stages,
 *                   18 = four-wave 256x256 on the LDS-DMA ring (experiment), 19 = four-wave 256x256x64 register-staged software
 *                   pipeline (ifx_gemm_w4.hip), 20 = auto including its split-K form when a workspace is given,
 *                   21 = 256x192x64 (what 0 = auto picks where 192-wide columns fill the rounds better: the QKV projection),
 *                   22 / 23 / 24 = the persistent ping-pong tiles of ifx_gemm_pp.hip: 256 / 192 / 128 tokens x 256 channels (what 0 =
 *                   auto picks for launches of at least 2048 rows); 22 splits K over two workgroups per tile where N <= 2048 and
 *                   K >= 4096 when the caller gives a workspace (ifx_gemm_bf16_ws), 25 = 22 without that split,
 *                   26 = stream-K on the 128-token ping-pong tile (needs the ifx_gemm_workspace_bytes workspace; an experiment that
 *                   lost to the tiles above at every size tried, kept for the lab: profiles/r3_gemm_pp.md),
 *                   27 / 28 / 29 = the 128-token ping-pong tile with K split over 2 / 4 / 8 workgroups per tile (each part >= 1 dumps
 *                   its fp32 tile image and raises its own flag, part 0 adds them in part order; needs the ifx_gemm_workspace_bytes
 *                   workspace and K / 64 divisible by the parts).  Under gemm_small_split 0 = auto takes the 4-way form for long-K
 *                   launches whose 128 x 256 tiles x 4 make one round of 129 .. 256 work items (a 4-way rank's FFN down-projection).
 *                   ifx_gemm_q8 reads the same option: 1 / 2 = register-staged 128x128 / 64-byte-row LDS-DMA tiles, 3 = the LDS-DMA
 *                   tiles, never the ping-pong tile, 22 / 23 / 24 = the ping-pong tile with 256 / 192 / 128 tokens (FP8 and INT8)
 *   "gemm_small_split": 1 lets the auto choice split K between the wave groups of one workgroup for launches of at most one workgroup
 *                   per CU (a sequence-parallel rank's 585 .. 2340 rows).  PER HOST THREAD (set, read and applied on the calling
 *                   thread only: launches another thread enqueues meanwhile keep their own value).  Off by default: those tiles sum K in a different order and
 *                   which launches get them depends on the row count, while the default auto choice keeps a row's bits independent
 *                   of the number of rows in the launch (its only K split, two workgroups per tile for N <= 2048 and K >= 4096
 *                   through ifx_gemm_bf16_ws, is a function of N and K).
 *   "attn_variant": 1 four-wave kernel, 2 eight-wave ping-pong schedule, 3 twelve-wave three-phase schedule,
 *                   4 free-running schedule, 5 software-pipelined schedule, 6 software-pipelined in four-wave workgroups, two per CU,
 *                   7 software-pipelined and unrolled four times over constant LDS slots (what 0 = auto picks for large launches)
 * Results are identical across variants up to fp32 summation order.  Returns IFX_EINVAL for unknown keys. */
/*   "conv_variant":     0 = auto (round 6: the persistent ping-pong kernel for 3x3 spatial kernels with cout % 96 == 0), 1 = the lock-step
 *                       kernel of round 1 everywhere (A/B: tools/bench_conv.py; the two produce identical bits).
 *   "attn_debug_counters": tests only — 1 allocates and zeroes a device word that the ping-pong attention kernels increment once per
 *                       (wave, key tile) that takes the rescale branch of the lazy row maximum; read it with
 *                       ifx_get_option("attn_rescale_count") (synchronises the device); 0 turns the counting off. */
int ifx_set_option(const char* key, int32_t value);
/* The value an option has NOW in the library (set through ifx_set_option by anyone in the process, or its environment default):
 * what a scoped override has to put back (a Python-side mirror misses options set through another binding).
 * Further keys of both calls:
 *   "spin_timeout_ms": budget of every DEVICE-side wait (the split-K / stream-K hand-off of ifx_gemm_bf16_ws / ifx_gemm_q8_ws), 1 ..
 *                      600000, default 2000 (env IFX_SPIN_TIMEOUT_MS).  A wait that runs out raises the device error word, the launch
 *                      completes with invalid results instead of hanging the GPU.
 *   "spin_fault":      tests only — 1 keeps the producers' flags down so that every such wait runs into its budget. */
int ifx_get_option(const char* key, int32_t* value);
/* Device error word: 0, or (kind << 24) | detail of the first wait a kernel gave up since the word was last cleared (kind 1 = split-K /
 * stream-K consumer, detail = flag index).  Pinned host memory: reading it needs no device synchronisation, but a launch only shows up
 * here once it has run — check after the stream has been synchronised.  `clear` != 0 resets the word.  ifx_last_error() reports (and
 * clears) a raised word ahead of the thread's last host-side message. */
int32_t ifx_device_error(int32_t clear);

/* ------------------------------------------------------------------------
 * Paged KV cache view (one request, one layer).
 * Replaces the tensors handed around by inferix/kvcache_manager/kvcache_manager.py:145-220
 * (`get` / `set`) and the slice arithmetic of
 * inferix/models/self_forcing/causal_model.py:277-304.
 *
 * Physical layout is the reference manager's own: K and V each
 * [num_slots][kv_heads][head_dim] bf16, token-major (manager tensor
 * (2, num_blocks, block_size, kv_heads, head_dim), kvcache_manager.py:222-244).
 * Logical token t lives in physical slot
 *     page_table ? page_table[t / page_size] * page_size + t % page_size : t     (or the two-segment map below)
 * so sink+rolling eviction (causal_model.py:282-300) is a page-table rotation
 * instead of a data move whenever the evicted span is page-aligned.
 * ---------------------------------------------------------------------- */
typedef struct {
  ifx_bf16* k;                 /* device */
  ifx_bf16* v;                 /* device */
  const int32_t* page_table;   /* device, may be NULL (identity) */
  int32_t page_size;           /* tokens per page (ignored when page_table == NULL) */
  int32_t num_slots;           /* capacity in tokens */
  int32_t kv_heads;
  int32_t head_dim;            /* 128 */
  /* Two-segment map for readers (the attention entry points), used when page_table == NULL and seg_split > 0: logical token
   * t < seg_split lives in slot t, t >= seg_split in slot t + seg_delta.  MAGI's cache rule leaves the rows of a forward that the
   * rule does not store outside the cache proper (inferix/kvcache_manager/model/magi_kv_cache_manager.py:76-187): here they sit in
   * a scratch tail of the same allocation and this map splices them behind the prefix without a per-token table.  0 = off. */
  int32_t seg_split;
  int32_t seg_delta;
} ifx_kv_view;

/* ------------------------------------------------------------------------
 * Block-causal flash-attention forward over the cached prefix, KV read in place.
 *   out[r,h,:] = softmax_j(q[r,h,:]·K[j,h,:] * scale) · V[j,h,:],  j in [kv_start, kv_len)
 * (logical cache tokens; kv_start > 0 is the split-KV / sequence-parallel case whose partial results
 * are combined with ifx_lse_merge)
 * No mask: block causality is realised by what is in the cache
 * (causal_model.py:307-315).  Replaces `attention()` / `flash_attention()`
 * (inferix/models/attention/flash_attention.py:42-200) and the registry backends'
 * (out, lse) contract (inferix/models/attention/backends.py:36-76).
 *   q, out : [q_rows, heads, 128] bf16 (row stride = heads*128); heads may be a multiple of kv->kv_heads
 *            (grouped-query attention, MAGI: query head h reads kv head h / (heads/kv_heads),
 *            inferix/models/magi/dit/dit_module.py:975-1018)
 *   lse    : optional [heads, q_rows] fp32 (natural log), NULL to skip
 *   scale  : softmax scale (1/sqrt(128) when <= 0)
 * bf16 in, fp32 softmax/accumulate, P rounded to bf16 for the PV product, bf16 out.
 * ---------------------------------------------------------------------- */
int ifx_attn_fwd_paged(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv,
                       int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                       void* stream);

/* Split-KV form of ifx_attn_fwd_paged for launches with few query rows (the sequence-parallel shards of
 * causal_model.py:939-942: N/P = 585 rows at P = 8, i.e. 36 (tile, head) pairs for 256 CUs).  The key range is cut
 * into `num_splits` chunks that run as independent workgroups; fp32 partial results go to `workspace` and are
 * combined with the ifx_lse_merge algebra (distributed.py:30-48) before the single bf16 rounding.  Result: as
 * ifx_attn_fwd_paged.  ifx_attn_split_plan returns the chunk count that fills the chip for a shape (1 = do not
 * split) and the workspace size that count needs; no allocation happens inside the library. */
int32_t ifx_attn_split_plan(int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len,
                            int64_t* workspace_bytes);
int ifx_attn_fwd_paged_split(const ifx_bf16* q, ifx_bf16* out, float* lse, const ifx_kv_view* kv,
                             int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale,
                             int32_t num_splits, void* workspace, int64_t workspace_bytes, void* stream);

/* ifx_attn_fwd_paged with explicit row strides of q and out (elements, >= heads * 128): the query rows may be a column block of
 * a wider matrix and the result may be written straight into a column block of the next GEMM's input — MAGI concatenates the
 * self-attention and cross-attention outputs in front of linear_proj (inferix/models/magi/dit/dit_module.py:1281-1295); with the
 * two attention launches writing columns [0, Q) and [Q, 2Q) of one [rows, 2Q] buffer the concatenation never exists as a copy.
 * num_splits / workspace as ifx_attn_fwd_paged_split (1 / NULL = no split). */
int ifx_attn_fwd_paged_ld(const ifx_bf16* q, int32_t ldq, ifx_bf16* out, int32_t ldo, float* lse, const ifx_kv_view* kv,
                          int32_t q_rows, int32_t heads, int32_t kv_start, int32_t kv_len, float scale, int32_t num_splits,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Attention over keys [0, kv_len) in which the LAST key stands for `last_key_multiplicity` identical (key, value) rows:
 *   out = softmax over the kv_len - 1 + multiplicity implied keys — computed by adding ln(multiplicity) to the last key's score.
 * Cross-attention of the block (inferix/models/wan_base/model.py:66-100) runs over text_len = 512 context rows of which all but
 * the prompt's own tokens are the zero-padding (causal_model.py:948-953): those rows are identical after `text_embedding`, so are
 * their K and V rows in every layer, and 512 keys become n_prompt + 1 — one 64-key tile instead of eight.  Exact algebra; the
 * implied keys' probabilities are rounded to bf16 once (as one product) instead of `multiplicity` times.  kv_len <= 1024. */
int ifx_attn_fwd_dedup(const ifx_bf16* q, ifx_bf16* out, const ifx_kv_view* kv, int32_t q_rows, int32_t heads, int32_t kv_len,
                       int32_t last_key_multiplicity, float scale, void* stream);

/* Several (query range, key range) pairs of one cache in ONE launch — MAGI's core_attention (inferix/models/magi/dit/dit_module.py:
 * 972-1015): per denoising range i, queries [q_ranges[i][0], q_ranges[i][1]) attend keys [k_ranges[i][0], k_ranges[i][1]), no mask
 * inside a range.  One rank of cp = 8 has 3 query heads: a single range is 144 workgroups for 256 CUs, four ranges together fill
 * the chip without splitting keys.  q_ranges / k_ranges are HOST arrays [n_ranges][2]; 1 <= n_ranges <= 8; ranges are tiled
 * separately (a query tile never straddles two ranges) and scheduled longest key range first.  Strides as ifx_attn_fwd_paged_ld
 * (0 = dense). */
int ifx_attn_fwd_ranges(const ifx_bf16* q, int32_t ldq, ifx_bf16* out, int32_t ldo, const ifx_kv_view* kv, int32_t q_rows,
                        int32_t heads, int32_t n_ranges, const int32_t* q_ranges, const int32_t* k_ranges, float scale,
                        void* stream);

/* The same split-KV algebra in separate launches that share one workspace: under sequence parallelism the cached
 * prefix is attended while the collective that delivers the new block's keys is still in flight, then the new keys,
 * then ONE merge.  ifx_attn_fwd_partial writes the fp32 partials of up to `num_splits` key chunks of
 * [kv_start, kv_len) into slots [slot_base, slot_base + *slots_used) of a workspace laid out for `slot_cap` slots
 * (slot_cap * q_rows * heads * 129 floats); ifx_attn_merge_partials combines the first `slots_used` slots into
 * out (bf16, one rounding) and the optional lse. */
int ifx_attn_fwd_partial(const ifx_bf16* q, const ifx_kv_view* kv, int32_t q_rows, int32_t heads, int32_t kv_start,
                         int32_t kv_len, float scale, int32_t num_splits, void* workspace, int64_t workspace_bytes,
                         int32_t slot_base, int32_t slot_cap, int32_t* slots_used, void* stream);
int ifx_attn_merge_partials(const void* workspace, int32_t slot_cap, int32_t slots_used, ifx_bf16* out, float* lse,
                            int32_t q_rows, int32_t heads, void* stream);

/* Merge two partial attention results over disjoint key sets (split-KV / context
 * parallel).  Replaces update_out_and_lse_pass_q
 * (inferix/models/attention/distributed.py:30-48).
 *   out_a/out_b [rows, heads, 128] bf16, lse_a/lse_b [heads, rows] fp32;
 *   result written to out_a / lse_a. */
int ifx_lse_merge(ifx_bf16* out_a, float* lse_a, const ifx_bf16* out_b, const float* lse_b,
                  int32_t rows, int32_t heads, void* stream);

/* ------------------------------------------------------------------------
 * Fused QK-RMSNorm + 3-axis RoPE + KV append.
 * Replaces, per layer: WanRMSNorm on q and k over ALL channels
 * (inferix/models/wan_base/components.py:107-126, applied causal_model.py:172-173),
 * causal_rope_apply[_chunked] (causal_model.py:33-100) and the cache write
 * causal_model.py:303-304 (+ the manager round trip :416-441).
 *
 *   qkv      : [rows, 3*dim] bf16 — fused q|k|v projection output (bias included)
 *   q_out    : [rows, dim] bf16   — RMSNorm(q)*wq then RoPE
 *   K slot(local_start + r) <- RoPE(RMSNorm(k)*wk),  V slot <- v   (via kv page table)
 *   rope     : position tables + token grid (NULL = no rotation: cross-attn q)
 *   kv       : NULL = no append (cross-attention query path)
 * Row r of this rank is token (frame r / hw_local, hw index hw_offset + r % hw_local):
 * temporal position start_frame + frame, height (hw / width), width (hw % width) — the
 * single-GPU case is hw_offset = 0, hw_local = height*width; the context-parallel case
 * is the reference's per-frame hw slice (causal_model.py:64-100,939-942).
 * Pairs are adjacent channels (2i, 2i+1); per head the head_dim/2 complex pairs split
 * [c-2*(c/3) temporal | c/3 height | c/3 width] (causal_model.py:34-37).
 * Rounding: y = bf16(x * rsqrt(mean(x^2)+eps)) ; y = bf16(y * w) ; rotation in fp64 ; bf16.
 * ---------------------------------------------------------------------- */
typedef struct {
  const double* freqs;   /* device [max_pos, head_dim/2, 2] = (cos, sin): the reference's
                            complex128 `self.freqs` (causal_model.py:636-641) viewed as real */
  int32_t max_pos;       /* 1024 */
  int32_t start_frame;   /* current_start // frame_seqlen (causal_model.py:255-256) */
  int32_t height, width; /* patch grid of one frame */
  int32_t hw_offset;     /* rank * hw_local */
  int32_t hw_local;      /* height*width / world_size */
  float q_scale;         /* ifx_rmsnorm_rope_kv_append only: the rotated q is multiplied by this in fp32 before its single rounding
                            to bf16 (0 = 1 = the reference's q).  With q_scale = softmax_scale * log2(e) and the attention entry
                            points called with scale = ln 2 the attention is the same function of the unrounded q — q·k·scale —
                            and the self-attention kernel takes its exponent fast path (scores are exp2 arguments; no scale-FMA
                            per score).  K / V and the cache are untouched. */
  int32_t flags;         /* ABI 0.6, ifx_rmsnorm_rope_kv_append only: bit 0 = the V rows of this block are ALREADY in their cache slots
                            (written by the projection, ifx_epilogue.y2): normalise / rotate / store q and K only */
} ifx_rope_grid;

int ifx_rmsnorm_rope_kv_append(const ifx_bf16* qkv, int32_t qkv_row_stride, ifx_bf16* q_out,
                               const ifx_bf16* wq, const ifx_bf16* wk, const ifx_rope_grid* rope,
                               const ifx_kv_view* kv, int32_t local_start, int32_t rows,
                               int32_t dim, float eps, void* stream);

/* RMSNorm only, in place or out of place, [rows, dim] (cross-attention q / text K). */
int ifx_rmsnorm(const ifx_bf16* x, int32_t x_row_stride, ifx_bf16* y, int32_t y_row_stride,
                const ifx_bf16* w, int32_t rows, int32_t dim, float eps, void* stream);

/* ------------------------------------------------------------------------
 * LayerNorm (+ AdaLN modulation).  Replaces WanLayerNorm
 * (components.py:129-142) and the per-frame modulation
 * `norm(x).unflatten(1,(F,fs)) * (1 + e[scale]) + e[shift]`
 * (causal_model.py:433,451-452,514).
 *   mode IFX_LN_PLAIN    : y = bf16(LN(x))
 *   mode IFX_LN_AFFINE   : y = bf16(LN(x) * gamma + beta)                 (norm3)
 *   mode IFX_LN_MODULATE : y = bf16(bf16(bf16(LN(x)) * bf16(1 + s)) + t), s/t = rows of `mod`
 *   mod : [groups, mod_slots, dim] bf16 = (modulation + e0) of this layer;
 *         row r uses group r / rows_per_group; shift_slot/scale_slot select the chunk.
 * ---------------------------------------------------------------------- */
enum { IFX_LN_PLAIN = 0, IFX_LN_AFFINE = 1, IFX_LN_MODULATE = 2 };
int ifx_layernorm(const ifx_bf16* x, ifx_bf16* y, int32_t rows, int32_t dim, float eps, int32_t mode,
                  const ifx_bf16* gamma, const ifx_bf16* beta, const ifx_bf16* mod,
                  int32_t mod_slots, int32_t shift_slot, int32_t scale_slot, int32_t rows_per_group,
                  void* stream);

/* ------------------------------------------------------------------------
 * bf16 linear layer  y = epilogue(x @ W^T + bias)  on MFMA, fp32 accumulate.
 * Replaces nn.Linear calls of the block (causal_model.py:125-128,171-175,332-333,377-379,
 * wan_base/model.py:66-100) together with the elementwise ops that follow them.
 *   x [M, K] bf16 (row stride ldx), W [N, K] bf16 (nn.Linear layout), bias [N] bf16 or NULL,
 *   y [M, N] bf16 (row stride ldy).
 *   IFX_EPI_BIAS      : y = bf16(acc + b)
 *   IFX_EPI_GELU_TANH : y = bf16(gelu_tanh(bf16(acc + b)))                   (ffn.0 + GELU)
 *   IFX_EPI_RESIDUAL  : y = bf16(res + bf16(acc + b))                         (cross-attn o)
 *   IFX_EPI_GATE_RES  : y = bf16(res + bf16(bf16(acc + b) * gate[r / rows_per_group]))
 *                       gate = rows of `mod` [groups, mod_slots, N] at gate_slot   (:444,455-456)
 *   IFX_EPI_GELU_ERF  : y = bf16(gelu_erf(bf16(acc + b)))   exact GELU, MAGI CustomMLP fc1 (magi/dit/dit_module.py:545-557)
 * ---------------------------------------------------------------------- */
enum { IFX_EPI_BIAS = 0, IFX_EPI_GELU_TANH = 1, IFX_EPI_RESIDUAL = 2, IFX_EPI_GATE_RES = 3, IFX_EPI_GELU_ERF = 4 };
typedef struct {
  int32_t epilogue;
  const ifx_bf16* residual;  /* [M, N], row stride ld_res */
  int32_t ld_res;
  const ifx_bf16* mod;       /* [groups, mod_slots, N] */
  int32_t mod_slots;
  int32_t gate_slot;
  int32_t rows_per_group;
  /* ABI 0.6 — second destination (zero-initialised = off): output columns [split_col, N) go to y2 [M, N - split_col] (row stride
   * ldy2, 16-byte aligned, ldy2 % 8 == 0) instead of y.  The block's fused q|k|v projection writes its V columns straight into the
   * rows of the KV cache that way (a contiguous cache: y2 = v + local_start * dim) and ifx_rmsnorm_rope_kv_append is told not to copy
   * them (ifx_rope_grid.flags).  split_col must be a multiple of 256; served by the persistent ping-pong tiles only (launches of
   * >= 2048 rows, bf16, IFX_EPI_BIAS): any other launch with y2 set returns IFX_EINVAL rather than dropping the columns. */
  ifx_bf16* y2;
  int32_t ldy2;
  int32_t split_col;
} ifx_epilogue;
int ifx_gemm_bf16(const ifx_bf16* x, int32_t ldx, const ifx_bf16* w, const ifx_bf16* bias,
                  ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K,
                  const ifx_epilogue* epi, void* stream);

/* ifx_gemm_bf16 with a caller-provided scratch buffer, which lets the library pick tiles that split K between workgroups (the
 * four-wave 256 x 256 tile with two K halves for long-K launches whose tiles would fill less than half of the chip: the FFN
 * down-projection of the block, 4680 x 1536 x 8960).  ifx_gemm_workspace_bytes returns what a shape wants (0 = none: the call is
 * then identical to ifx_gemm_bf16).  The FIRST 4096 bytes of the workspace must be zero on entry (zero it once after allocation);
 * every call leaves them zero.  Launches that share a workspace must be ordered on one stream.  Results are deterministic
 * (a + b == b + a for the two partial sums). */
int64_t ifx_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K);
int ifx_gemm_bf16_ws(const ifx_bf16* x, int32_t ldx, const ifx_bf16* w, const ifx_bf16* bias, ifx_bf16* y, int32_t ldy,
                     int32_t M, int32_t N, int32_t K, const ifx_epilogue* epi, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------------
 * Dynamic 8-bit linear layers: per-token activation x per-channel weight, FP8 (OCP e4m3fn) or INT8.
 * Replaces the DAX `quantize_dynamic` linears wired in
 * example/quantization/run_self_forcing_quantized.py:19-23,47-64 (every nn.Linear under generator.model
 * except text_embedding / proj_out / head).  DAX is un-vendored and unpinned: the arithmetic is defined HERE
 * (restated in oracle/quant_oracle.py); parity with DAX itself is unpinned.
 *
 *   ifx_quant_per_token : s[m] = max_k|x[m,k]| / QMAX (1.0 for a zero row), QMAX = 448 | 127;
 *                         q[m,k] = cast(clamp(x[m,k] / s[m], +-QMAX)), e4m3: RNE, int8: rint.
 *                         x [rows, K] bf16 (row stride ldx), q [rows, K] bytes (row stride ldq), scale [rows] fp32.
 *   ifx_gemm_q8         : y = epilogue(bf16(acc * (x_scale[m] * w_scale[n]) + bias[n])), acc = sum_k xq*wq in
 *                         fp32 (fp8 MFMA) or exact int32 (int8 MFMA); wq [N, K] bytes, w_scale [N] fp32 from the
 *                         same rule applied per output channel.  Epilogues as ifx_gemm_bf16.  K % 128 == 0.
 * ---------------------------------------------------------------------- */
enum { IFX_Q_FP8_E4M3 = 0, IFX_Q_INT8 = 1 };
int ifx_quant_per_token(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, float* scale, int32_t rows,
                        int32_t K, int32_t format, void* stream);
int ifx_gemm_q8(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                const ifx_bf16* bias, ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t format,
                const ifx_epilogue* epi, void* stream);
/* ifx_gemm_q8 with a caller-provided scratch buffer (the contract of ifx_gemm_bf16_ws: first 4096 bytes zero on entry and on exit,
 * one stream per workspace): long-K, narrow-N launches — the block's FFN down-projection, 4680 x 1536 x 8960: 114 tiles of 256 x 256
 * for 256 CUs — then run the 256-token ping-pong tile with K split between two workgroups.  Which shapes split is a function of N
 * and K only (a row's bits do not depend on the launch's row count); e4m3: first half + second half in fp32, int8: the two exact
 * int32 sums are added as integers — the same bits as the unsplit launch.  ifx_gemm_q8_workspace_bytes: what a shape wants (0 = none).
 * e4m3 BIT STABILITY: "a row's bits do not depend on the launch" holds when every launch of a shape goes through THIS entry
 * point with the workspace it asks for — what inferix_amd does.  A launch of a split shape runs UNSPLIT (first + second K half summed
 * in one fp32 chain instead of two: e4m3 results may differ in a few elements per million, int8 results never) when: the caller uses
 * plain ifx_gemm_q8 or passes no / too small a workspace; the gate epilogue has rows_per_group < 128; ifx_gemm_q8_quant_out (static
 * output quantiser) is used; bias / residual / gate / scale pointers or ldx miss the tile's alignment (8 / 16 bytes, ldx % 16); the
 * launch has more than 1024 tiles of 256 x 256 (M > ~43 000 at N = 1536); or gemm_variant forces another tile. */
int64_t ifx_gemm_q8_workspace_bytes(int32_t M, int32_t N, int32_t K);
int ifx_gemm_q8_ws(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale, const ifx_bf16* bias,
                   ifx_bf16* y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t format, const ifx_epilogue* epi,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* ifx_layernorm_quant: ifx_layernorm followed by ifx_quant_per_token of its bf16 result, in one pass over the row — what the
 * quantised qkv / cross-attention q / ffn.0 linears of a block see (the reference's DAX wrapper quantises the input of every
 * nn.Linear, i.e. the norm's bf16 output: run_self_forcing_quantized.py:47-64, causal_model.py:419-428,470-476).  Same bytes and
 * scales as the two calls, bit for bit; the bf16 row is not written.  Arguments as ifx_layernorm + (q, ldq, scale, format). */
int ifx_layernorm_quant(const ifx_bf16* x, void* q, int32_t ldq, float* scale, int32_t rows, int32_t dim, float eps,
                        int32_t mode, const ifx_bf16* gamma, const ifx_bf16* beta, const ifx_bf16* mod, int32_t mod_slots,
                        int32_t shift_slot, int32_t scale_slot, int32_t rows_per_group, int32_t format, void* stream);

/* ifx_gemm_q8_quant_out: ifx_gemm_q8 with a GELU epilogue (IFX_EPI_GELU_TANH / IFX_EPI_GELU_ERF) whose bf16 result goes straight through
 * ifx_quant_static's rule instead of to memory: yq[m, n] = e4m3(div_clamp_to(gelu(bf16(acc * scales + bias)), out_divisor[n])), bytes at
 * yq + m * ldyq + n (ldyq in BYTES, % 8).  It is fc1 -> fc2 of MAGI's fp8_quant MLP (CustomMLP, dit_module.py:493-560: the input of the
 * PerChannelQuantizedFp8Linear fc2 is fc1's activation divided by smooth_scale): same bytes as ifx_gemm_q8 + ifx_quant_static, bit for
 * bit, without the [rows, ffn] bf16 round trip.  N % 8 == 0.
 * ifx_layernorm_quant_static: ifx_layernorm (plain / affine) followed by n_out ifx_quant_static passes of its bf16 row, each with its own
 * per-input-channel divisor vector (divisors [n_out][dim] fp32): output j lands at byte column j * dim of q (row stride ldq >= n_out *
 * dim bytes).  MAGI's q / qx / k / v linears quantise the same normalised row with their own input_scale vectors (:448-462). */
int ifx_gemm_q8_quant_out(const void* xq, int32_t ldx, const float* x_scale, const void* wq, const float* w_scale,
                          const ifx_bf16* bias, void* yq, int32_t ldyq, int32_t M, int32_t N, int32_t K, int32_t format,
                          const ifx_epilogue* epi, const float* out_divisor, int32_t via_bf16, void* stream);
int ifx_layernorm_quant_static(const ifx_bf16* x, void* q, int32_t ldq, const float* divisors, int32_t n_out, int32_t rows,
                               int32_t dim, float eps, int32_t mode, const ifx_bf16* gamma, const ifx_bf16* beta,
                               int32_t via_bf16, void* stream);

/* Static-scale and per-tensor quantisers (the other qconfig families north_star names, and MAGI's own FP8 linears).
 *   ifx_quant_static     : q[m,k] = cast(clamp(x[m,k] / divisor[k or 0], +-QMAX)).  With via_bf16 = 1 the clamped quotient is
 *                          rounded to bf16 before the e4m3 cast — exactly `div_clamp_to`, the one quantisation routine in the
 *                          reference tree (inferix/models/magi/dit/dit_module.py:367-387), used by PerTensorQuantizedFp8Linear
 *                          (divisor = input_scale [in_features], :448-462) and PerChannelQuantizedFp8Linear (divisor =
 *                          smooth_scale [1, in_features], :480-490).  divisor_len = K (per input channel) or 1.  row_scale:
 *                          optional [rows] fp32, filled with divisor[0] when divisor_len == 1 (feeds ifx_gemm_q8's x_scale).
 *                          The matmul itself is flashinfer's bmm_fp8 upstream (un-vendored): ifx_gemm_q8 with x_scale[m] =
 *                          input_scale and w_scale[n] = weight_scale computes bf16(acc * (input_scale * weight_scale)).
 *   ifx_quant_per_tensor : dynamic per-TENSOR scale s = max|x| / QMAX (1.0 for an all-zero tensor), q = cast(clamp(x / s)),
 *                          row_scale[m] = s for every row.  Two kernels on the stream (abs-max reduction into the 4-byte
 *                          amax_workspace, which the call zeroes with hipMemsetAsync, then the quantiser); no host sync. */
int ifx_quant_static(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, const float* divisor, int32_t divisor_len,
                     float* row_scale, int32_t rows, int32_t K, int32_t format, int32_t via_bf16, void* stream);
int ifx_quant_per_tensor(const ifx_bf16* x, int32_t ldx, void* q, int32_t ldq, float* row_scale, void* amax_workspace,
                         int32_t rows, int32_t K, int32_t format, void* stream);

/* ------------------------------------------------------------------------
 * KV cache maintenance.  ifx_kv_roll: the reference's eviction shift
 * cache[sink : sink+rolled] <- cache[sink+evicted : sink+evicted+rolled]
 * (causal_model.py:287-292) as a physical move, for spans that are not page aligned
 * (page-aligned spans are handled by rotating the page table on the host). */
int ifx_kv_roll(const ifx_kv_view* kv, int32_t sink_tokens, int32_t evicted, int32_t rolled,
                ifx_bf16* scratch, void* stream);

/* Sequence-parallel cache write (the "cache write post-a2a" of CoreAttention.forward,
 * inferix/models/attention/distributed.py:183-208, for the replicated-cache design of DESIGN.md §6): the
 * all-gathered K/V rows of the new block, rank-major
 *   gathered [world][2 (K,V)][frames*hw_local][kv_heads*128] bf16,
 * go to the cache slots of logical tokens local_start + f*frame_tokens + r*hw_local + i, i.e. the single-GPU
 * (frame, hw) order of causal_model.py:939-942 / 1008-1022, through the page table when there is one. */
int ifx_kv_scatter_shards(const ifx_bf16* gathered, int32_t world, int32_t frames, int32_t hw_local,
                          int32_t frame_tokens, int32_t local_start, const ifx_kv_view* kv, void* stream);

/* Sequence-parallel exchange WITHOUT a collective: every rank stores its post-RoPE K and raw V rows of the new block
 * straight into the cache slots of every peer's replicated KV cache over xGMI.  Stands where the reference has the Ulysses
 * all-to-all + ring p2p of inferix/models/attention/distributed.py:183-208,610-706 (and where this library's first design had one
 * all-gather + ifx_kv_scatter_shards per layer).
 *   ifx_peer_alloc / _free     device memory other processes can map (fine_grained != 0: visible to running kernels — flag blocks)
 *   ifx_peer_export / _open / _close   64-byte HIP IPC handle of the allocation `ptr` lies in + the offset of `ptr` in it  <->  the
 *                              allocation's base mapped into this process (add the offset)
 *   ifx_rmsnorm_rope_kv_push   rows [rows, >= 2*dim] = (k | v) of the K/V-only projection: K <- RoPE(RMSNorm(k) * wk), V raw, row r
 *                              (f = r / slot_hw_local, i = r % slot_hw_local) stored at the slot of logical token
 *                              local_start + f*frame_tokens + slot_hw_offset + i of EVERY destination (same page table / segment map
 *                              everywhere: `geometry` gives it, its k/v pointers are ignored).  One destination = the rank's own cache
 *                              or staging buffer: the K/V-only form of ifx_rmsnorm_rope_kv_append.
 *   ifx_peer_signal            flags[p][index] <- value in every peer's flag block, stream-ordered, released at system scope
 *   ifx_peer_wait              stream-ordered wait until flags[i] >= value for i < count (this rank's own block); after timeout_ms
 *                              the kernel gives up and stores 1 + i into *status (host checks it at a synchronisation point). */
#define IFX_MAX_PEERS 8
#define IFX_PEER_HANDLE_BYTES 64
typedef struct {
  int32_t count;
  ifx_bf16* k[IFX_MAX_PEERS];
  ifx_bf16* v[IFX_MAX_PEERS];
} ifx_peer_caches;
typedef struct {
  int32_t count;
  int32_t* flags[IFX_MAX_PEERS];
} ifx_peer_flags;
int ifx_peer_alloc(int64_t bytes, int32_t fine_grained, void** ptr);
int ifx_peer_free(void* ptr);
int ifx_peer_export(const void* ptr, uint8_t* handle, int64_t* offset);
int ifx_peer_open(const uint8_t* handle, void** ptr);
int ifx_peer_close(void* ptr);
int ifx_rmsnorm_rope_kv_push(const ifx_bf16* kv_rows, int32_t kv_row_stride, const ifx_bf16* wk, const ifx_rope_grid* rope,
                             const ifx_peer_caches* peers, const ifx_kv_view* geometry, int32_t local_start, int32_t frame_tokens,
                             int32_t slot_hw_local, int32_t slot_hw_offset, int32_t rows, int32_t dim, float eps, void* stream);
int ifx_peer_signal(const ifx_peer_flags* peers, int32_t index, int32_t value, void* stream);
int ifx_peer_wait(const int32_t* flags, int32_t count, int32_t value, int32_t timeout_ms, int32_t* status, void* stream);

/* ----------------------------------------------------------------------
 * VAE decoder: channels-last causal 3-D convolution.
 * Replaces CausalConv3d.forward + the feature-cache concatenation around it
 * (inferix/models/wan_base/vae.py:26-34, 207-216), the nearest-2x Upsample +
 * Conv2d of Resample (vae.py:58-64, 83-90, 139-141) and the residual add of
 * ResidualBlock.forward (vae.py:219).
 *   x          frames [slot][hs][ws][cin] bf16; logical input frame f (0 .. t_out+kt-2: for kt = 3 the two
 *              history frames, then the new ones) lives at x + in_slots[f] * in_frame_stride; in_slots[f] < 0
 *              = an all-zero frame (the causal padding in front of the stream)
 *   in_planar  1: the input frames are [cin/32][hs][ws][32] instead (32-channel planes: the halo patch of one
 *              channel chunk is then rows of contiguous 64-byte pixels — LDS-DMA moves 64-byte rows that lie
 *              192+ bytes apart at a third of the rate of contiguous ones, round 6; the frame rings in front of
 *              the 3x3x3 convs use this layout, written by ifx_rmsnorm_cl with IFX_NORM_OUT_PLANAR)
 *   upsample   1: the 3x3 taps read the nearest-2x upsampled frame (output 2hs x 2ws), never materialised
 *   w          [kt*ks*ks][cin/32][cout][32] bf16: tap-major, 32-channel-chunk-major repack of the torch
 *              [cout][cin][kt][ks][ks] weight (a DMA piece of 16 output channels x 32 input channels is contiguous)
 *   y          output frame t at y + out_slots[t] * out_frame_stride, [ho][wo][cout]
 *   residual   NULL or contiguous [t_out][ho][wo][cout]: y = bf16(bf16(conv + bias) + residual)
 *   zero_page  >= 64 bytes of zeros in device memory
 * in_slots / out_slots are HOST arrays (copied into the launch).
 * ---------------------------------------------------------------------- */
typedef struct {
  const ifx_bf16* x;
  int64_t in_frame_stride;      /* elements */
  const int32_t* in_slots;
  int32_t hs, ws, cin, upsample;
  const ifx_bf16* w;
  const ifx_bf16* bias;         /* [cout] or NULL */
  int32_t kt, ks;               /* temporal 1|3, spatial 1|3 */
  ifx_bf16* y;
  int64_t out_frame_stride;     /* elements */
  const int32_t* out_slots;
  int32_t cout, t_out;
  const ifx_bf16* residual;
  const void* zero_page;
  int32_t in_planar;            /* ABI minor 7; 0 = channels-last input frames */
} ifx_conv3d_desc;
int ifx_conv3d_cl(const ifx_conv3d_desc* desc, void* stream);

/* Per-pixel channel RMS norm (+ SiLU) of channels-last frames x [frames][frame_pixels][channels], with the bf16 op
 * chain of RMS_norm.forward (vae.py:52-55) and nn.SiLU; frame f is written to y + out_slots[f] * out_frame_stride
 * (the ring buffer that holds the next conv's input and its two-frame history).  out_slots is a HOST array.
 * flags: IFX_NORM_SILU (1) applies SiLU; IFX_NORM_OUT_PLANAR (2; channels % 32 == 0) writes the frame as
 * [channels/32][frame_pixels][32] (ifx_conv3d_desc.in_planar) instead of channels-last. */
#define IFX_NORM_SILU 1
#define IFX_NORM_OUT_PLANAR 2
int ifx_rmsnorm_cl(const ifx_bf16* x, const ifx_bf16* gamma, ifx_bf16* y, int64_t out_frame_stride,
                   const int32_t* out_slots, int32_t frames, int32_t frame_pixels, int32_t channels,
                   int32_t flags, void* stream);

/* probs = softmax(scores * scale) row-wise, bf16 [rows][ld] (single-head attention of the VAE middle block,
 * vae.py:250-254, between the two ifx_gemm_bf16 launches that form QK^T and PV). */
int ifx_softmax_rows(const ifx_bf16* scores, ifx_bf16* probs, int32_t rows, int32_t cols, int32_t ld, float scale,
                     void* stream);

/* ----------------------------------------------------------------------
 * umT5 text encoder.
 * ifx_t5_attention: bidirectional self-attention of T5Attention.forward between its q/k/v and o linears
 * (inferix/models/wan_base/text_encoder/t5.py:96-117), head_dim 64, NO 1/sqrt(d) scaling:
 *   s = bf16(bf16(q.k) + bias[h][key - query]) ; keys >= seq_lens[b] get finfo(bf16).min ; p = bf16(softmax_fp32(s)) ;
 *   out = bf16(p v).
 *   q/k/v/out   rows b*seq_len_padded + t, channels h*64 + c, given row strides (q, k, v may be the column blocks of one
 *               fused qkv GEMM output)
 *   rel_bias    [heads][2*seq_len_padded - 1] bf16: the T5RelativeEmbedding value (t5.py:235-266) of relative offset
 *               (key - query) at index offset + seq_len_padded - 1; the [1, heads, L, L] tensor is never built
 *   seq_lens    [batch] int32 in device memory
 * seq_len_padded in [32, 512], multiple of 32 (K and V^T of one head are staged in LDS).
 * ---------------------------------------------------------------------- */
int ifx_t5_attention(const ifx_bf16* q, int32_t ldq, const ifx_bf16* k, int32_t ldk, const ifx_bf16* v, int32_t ldv,
                     ifx_bf16* out, int32_t ldo, const ifx_bf16* rel_bias, const int32_t* seq_lens, int32_t batch,
                     int32_t seq_len_padded, int32_t heads, void* stream);

/* h[rows][ffn] = fc1 * GELU(gate) with the tanh GELU of t5.py:50-52 evaluated op by op in bf16 (T5FeedForward.forward,
 * t5.py:138-139); gate_fc1 = [rows][2*ffn], columns [0, ffn) = gate pre-activation, [ffn, 2 ffn) = fc1 (one fused GEMM). */
int ifx_t5_gated_gelu(const ifx_bf16* gate_fc1, ifx_bf16* h, int32_t rows, int32_t ffn, void* stream);

/* ----------------------------------------------------------------------
 * MAGI transformer layer (BASELINE config 5), the row kernels between its GEMMs and attention calls.
 *
 * ifx_magi_head_prep: everything FullyParallelAttention.get_q / get_k / get_v / get_xqkv do to the projection outputs
 * (inferix/models/magi/dit/dit_module.py:902-970), one pass over the fused projection row:
 *   layout 0: in = [rows, q_heads q | q_heads qx | kv_heads k | kv_heads v] x 128 (ONE GEMM over the concatenated
 *             linear_qkv.{q,qx,k,v} weights)
 *               q  : fp32 LayerNorm(128; qn_w (+1), qn_b) -> non-interleaved rotary (flash-attn apply_rotary_emb: with the head
 *                    split x1 | x2, out = x1*cos - x2*sin | x1*sin + x2*cos, fp32) -> bf16 -> q_out[r, h*128]
 *               k  : the same with kn_w / kn_b -> k_out[dest(r), h]
 *               v  : copy -> v_out[dest(r), h]
 *               qx : bf16 LayerNorm(xn_w (+1 in bf16), xn_b), one rounding -> qx_out[r, h*128]      (q_layernorm_xattn)
 *   layout 1: in = [rows, kv_heads x (k | v)] x 128 (linear_kv_xattn output viewed [y, hn, 2*hd], :959-968)
 *               k  : bf16 LayerNorm(xn_w, xn_b) (k_layernorm_xattn) -> k_out[dest(r), h];  v : copy -> v_out[dest(r), h]
 *   rope      [rows, 128] fp32 = (sin[64] | cos[64]) per token, the reference's rotary_pos_emb rows (:1097)
 *   k / v destination: element offset dest(r) * ld_kv + h * kv_head_stride with
 *               dest(r) = r < split ? row0 + r : row1 + (r - split)
 *             i.e. the rows MagiKVCacheManager's rule stores go to their final cache slots and the rest to the scratch tail
 *             (inferix/kvcache_manager/model/magi_kv_cache_manager.py:76-187); or (row0 = 0, split = rows) a staging buffer
 *             of the Ulysses all-to-all.
 * ---------------------------------------------------------------------- */
typedef struct {
  const ifx_bf16* in;
  int32_t ld_in, rows, layout, q_heads, kv_heads, head_dim;
  const float* rope;
  const float *qn_w, *qn_b, *kn_w, *kn_b;     /* fp32 [128] */
  const ifx_bf16 *xn_w, *xn_b;                /* bf16 [128] */
  float eps;
  int32_t layernorm_1p;                       /* apply_layernorm_1p: norm weight + 1 */
  ifx_bf16* q_out;
  int32_t ld_q;
  ifx_bf16* qx_out;
  int32_t ld_qx;
  ifx_bf16* k_out;
  ifx_bf16* v_out;
  int32_t ld_kv, kv_head_stride;
  int32_t row0, split, row1;
  float q_scale;                              /* q (self-attention queries only) is multiplied by this in fp32 before its one rounding to bf16;
                                                 0 = 1.  softmax_scale * log2(e) here + attention scale = ln 2 = the same attention on the
                                                 exponent fast path (see ifx_rope_grid.q_scale) */
  int32_t rope_half;                          /* rotary pairs per head: rope rows are [2 * rope_half] fp32 = (sin | cos), channel e < rope_half
                                                 pairs with e + rope_half and the channels behind 2 * rope_half pass through.  0 = 64 (the whole
                                                 head).  MAGI's table covers 96 of the 128 head channels (3 axes x head_dim / 8 bands,
                                                 dit_module.py:673-720): rope_half = 48.  Multiple of 8. */
  int32_t q_group;                            /* ABI 0.6, 0 = off: self-attention q head h is written at
                                                 q_out + (h / q_group) * q_group_stride + r * ld_q + (h % q_group) * 128 — the send order of
                                                 the head -> rank all-to-all ([(cp seq), heads / cp, 128]: q_group = heads / cp, ld_q = q_group *
                                                 128, q_group_stride = rows * ld_q), so that no layout copy stands between this kernel and the
                                                 collective (context_parallel.py:317-335's permute + contiguous) */
  int64_t q_group_stride;
} ifx_magi_head_prep_desc;
int ifx_magi_head_prep(const ifx_magi_head_prep_desc* desc, void* stream);

/* bias_modulate_add (dit_module.py:295-313): the Triton `range_mod_kernel_fwd` (:204-292) + post-norm + residual in one pass:
 *   y[r] = bf16( LayerNorm_fp32( float(x[r]) * float(gate[condition_map[r]]) ; norm_w (+1), norm_b ) + float(residual[r]) )
 *   gate: rows of the softcapped AdaModulateLayer output, row stride ld_gate (pass gate + hidden for the MLP half);
 *   norm_w / norm_b fp32 [dim] (self_attn_post_norm / mlp_post_norm are fp32 modules, dit_model.py:620-637). */
int ifx_magi_gate_norm_residual(const ifx_bf16* x, int32_t ldx, const ifx_bf16* residual, int32_t ld_res,
                                const int32_t* condition_map, const ifx_bf16* gate, int32_t ld_gate, const float* norm_w,
                                const float* norm_b, int32_t layernorm_1p, ifx_bf16* y, int32_t ldy, int32_t rows, int32_t dim,
                                float eps, void* stream);

/* Elementwise bf16 -> bf16 with fp32 math and one rounding: IFX_ACT_SILU (AdaModulateLayer.act, :196-198),
 * IFX_ACT_TANH (softcap with cap 1, :363-364,:1300-1303). */
enum { IFX_ACT_SILU = 0, IFX_ACT_TANH = 1 };
int ifx_act_rows(const ifx_bf16* x, ifx_bf16* y, int64_t n, int32_t mode, void* stream);
/* MagiKVCacheManager's store after the head -> rank all-to-all (inferix/models/magi/dit/dit_module.py:905-952: the K | V rows of the
 * message go into the cache, the first `split` rows to the run that is kept, the rest to the scratch run behind it):
 *   kv [rows, heads, 2 * 128] contiguous; row r -> cache row (r < split ? row0 + r : row1 + r - split) of k_cache / v_cache
 *   [slots, heads, 128] (head_dim 128).  One launch instead of four strided copies per layer. */
int ifx_kv_split_rows(const ifx_bf16* kv, ifx_bf16* k_cache, ifx_bf16* v_cache, int32_t rows, int32_t heads, int32_t row0,
                      int32_t split, int32_t row1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INFERIX_HIP_H */
