/* libgfv - C ABI of the MI355X-native (gfx950) Gen-FVGN hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a hipStream_t (as void*), allocates
 * nothing, is asynchronous on the given stream and returns 0 (GFV_OK) or a negative error code.  State the library keeps:
 * the product form (a process-wide default, gfv_set_f16split, with an optional per-thread override,
 * gfv_set_f16split_thread) and the model's hidden size (thread-local, gfv_set_hidden_size - like the HIP runtime's current
 * device); a chain launch may carry its own of both in its argument struct,
 * the device status word (gfv_status_flags), the optional profiling records (gfv_profile_*) and the mesh plans a caller
 * creates and destroys (gfv_plan_create / gfv_plan_destroy).  All floating point data is fp32, all index data int32 (plans are narrowed from the reference's
 * int64 once per mesh batch).
 *
 * What each entry replaces in the reference (paths relative to /root/reference/src) is cited per function; the
 * reference has no native code of its own: these replace the third-party CUDA kernels it reaches through
 * torch_scatter / ATen / cuBLAS / Inductor.
 */
#ifndef GFV_H_
#define GFV_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): gfv_rowtile_args_t grew (rc_Wh, rc_bias; round 3 had already turned its former pad fields hidden / flags /
 * product_form into live inputs and appended fin_stats .. dw_in_ld without a bump), gfv_set_f16split became a process-wide
 * default with gfv_set_f16split_thread beside it.  A binding checks gfv_abi_version() AND gfv_struct_size() at load. */
/* 3 (round 6): buffer-size contracts that moved in round 5 without a bump are now part of the version - (1) ln_partial of a
 * GFV_IN_LNBWD / GFV_FIN_LNBWD launch and of gfv_trans_mlp_bwd must hold gfv_rowtile_ln_rows(M) = ceil(M / 32) rows (ABI 2 said
 * gfv_rowtile_tiles(M) = ceil(M / 64): the small-tile backward families fill one row per 32 rows); (2) a fused dw_partial launch
 * writes blocks 0 .. gfv_rowtile_dw_partials_m(M) - 1 ONLY (ABI 2 callers reduced gfv_rowtile_dw_partials() blocks: the blocks
 * beyond that count are NOT written and hold whatever the buffer held); (3) the status word has an asynchronous mirror
 * (gfv_status_mirror / gfv_status_publish) and gfv_adam_step_dev publishes it; Adam is one launch (state[16], gfv_adam_state_init;
 * gfv_adam_tick_dev / gfv_adam_update_dev removed); (4) new entry points
 * gfv_prep_stats / gfv_prep_apply, gfv_fvm_fwd_fused / gfv_fvm_bwd_fused, gfv_slice_token_attention_fwd / _bwd. */
#define GFV_ABI_VERSION 3
int gfv_abi_version(void);
/* sizeof of the argument structs as the library was compiled (which: 0 gfv_seg_t, 1 gfv_layer_t, 2 gfv_rowtile_args_t,
 * 3 gfv_wimg_desc_t, 4 gfv_dw_tile_t, 5 gfv_reduce_piece_t, 6 gfv_plan_desc_t, 7 gfv_trans_mlp_t, 8 gfv_trans_mlp_bwd_t,
 * 9 gfv_fvm_mesh_t): lets a
 * binding check its own layout */
int gfv_struct_size(int32_t which);

/* ------------------------------------------------------------------------------------------------------------
 * Segmented (CSR) gather-reduce:  out[r,:] = scale[r] * sum_{k in [rowptr[r],rowptr[r+1])} src[col[k],:]
 * Atomics-free wavefront segmented reduce.  Replaces torch_scatter.scatter_add / scatter_mean and the fused
 * "gather then scatter" pairs at FVMmodel/Models/FVGN/blocks.py:35-51,92-99 and the index_put_ backward of
 * the gathers at blocks.py:101-102.  src is [n_src, F] row-major (ld = F), F in {4,8,...,256} multiple of 4,
 * or any F through the scalar path.  scale (per destination row) and src_scale (per source row, applied to
 * each gathered row) may be NULL.  `accumulate` != 0 adds into out.
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_seg_gather_sum(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                       const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate, void* stream);

/* same; nnz_hint = rowptr[n_rows] if the caller knows it (only used for the profiler's algorithmic-byte count) */
int gfv_seg_gather_sum_nnz(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                           const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate,
                           int64_t nnz_hint, void* stream);

/* same; n_src_hint = rows of src (the distinct source rows: what SURVEY.md 8(d) prices a gather by); profiler only */
int gfv_seg_gather_sum_ex(const float* src, const int32_t* rowptr, const int32_t* col, const float* scale,
                          const float* src_scale, float* out, int32_t n_rows, int32_t F, int32_t accumulate,
                          int64_t nnz_hint, int64_t n_src_hint, void* stream);

/* Round 6.  The same reduce over the 64-column HALVES of 128-wide rows (half-row c = row c >> 1, columns 64 (c & 1) ...: the view
 * blocks.py:35-42 scatters to the two end nodes of an edge), with LayerNorm applied on the way in:
 *   out[r, :] = sum_k LN(y)[half-row col[k]],   LN(y)[e, c] = (y[e, c] - mean_e) * rstd_e * gamma[c] + beta[c],
 * y [*,128] = the pre-LayerNorm rows a chain launch saved (gfv_rowtile_args_t.fin_presave), stats [*,2] = its (mean, 1 / std)
 * (fin_stats).  The expression is the chain launch's own, so the sums equal - bit for bit - those over its LayerNorm output,
 * which the EdgeBlock forward then need not write a second time without the residual (512 B per edge row).  out [n_rows, 64];
 * y, gamma, beta, out 16-byte aligned.  The two hints: as gfv_seg_gather_sum_ex (profiler only). */
int gfv_seg_gather_sum_ln(const float* y, const float* stats, const float* gamma, const float* beta, const int32_t* rowptr,
                          const int32_t* col, float* out, int32_t n_rows, int64_t nnz_hint, int64_t n_src_hint, void* stream);

/* out[e, 0:F] = a[s[e], :], out[e, F:2F] = a[r[e], :]  (+ base[e,:] if base != NULL).  Adjoint of the
 * chunked edge->node scatter at blocks.py:34-42. */
int gfv_gather_pair(const float* a, const int32_t* s, const int32_t* r, const float* base, float* out,
                    int32_t n_edges, int32_t F, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused GEMM chain: up to three Linear layers applied to a tile of rows, with gather / concat / segmented-sum / LayerNorm
 * prologues and bias / GELU / LayerNorm / residual epilogues; forward and the dX chain of the backward are the same entry
 * with different element ops.  One entry point, several kernel families behind it (gfv_rowtile_last_path tells which):
 * the register-resident row-owner chain (a wave owns 16 rows; split-fp16 products from per-step weight images, or fp32
 * MFMA), the column-owner persistent backward with fused weight gradients (dw_partial) and the lean kernel for single-layer
 * launches.  ACCEPTED SHAPES (anything else returns GFV_ERR_ARG and launches nothing - the generic LDS kernel that used to take
 * the rest was retired with ABI 2; tests/test_kernels_gpu.py holds the negative test):
 *   plain:   every segment width a multiple of 32 and every row stride (seg ld, out_ld, res_ld) a multiple of 4; K and ldw of
 *            every layer multiples of 4; inner layers 128 wide, the last layer a multiple of 64 (exactly 128 with a LayerNorm
 *            epilogue).  Only this class takes LayerNorm backward (GFV_IN_LNBWD / GFV_FIN_LNBWD), segmented-sum segments
 *            (csr_rowptr) and per-segment saves.
 *   ragged:  any segment width / row stride, any first-layer K, a last layer of any width <= 128 (or a multiple of 64 above);
 *            inner layers 128 wide with K = 128; NO LayerNorm backward; a last layer that is not a multiple of 16 wide takes
 *            neither GFV_OP_MUL_DGELU nor a LayerNorm epilogue nor out_nores; GFV_IN_LN and gadd need a 128-wide seg[0] with a
 *            row stride that is a multiple of 4.
 * Replaces nn.Linear/GELU/LayerNorm inside build_mlp (FVMmodel/Models/FVGN/EPD.py:10-63), the concat + MLP of
 * EdgeBlock/NodeBlock (blocks.py:54,101-111) and the Linear layers of the Transolver block
 * (FVMmodel/Models/GraphTransolver/GraphTransolver.py:54-59,95,98-128,163-169).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
  const float* ptr;   /* [rows, ld] */
  const int32_t* idx; /* optional row gather index [M]; NULL = identity */
  int32_t width;      /* valid columns (<= 128); zero padded to a multiple of 16 inside the kernel */
  int32_t ld;         /* row stride in floats */
  /* optional segmented-sum segment (register-resident chain, no LayerNorm backward, widths multiples of 32):
   *   row m = csr_scale[m] * sum_{k in [csr_rowptr[m], csr_rowptr[m+1])} ptr[idx[k] * ld + 0:width]
   * (idx is then the CSR column list [nnz]; csr_scale may be NULL) - the GnBlock's neighbour aggregation
   * (blocks.py:25-51,84-99: scatter_add / scatter_mean over the two-way edge list) and the per-side scatter of the factored
   * EdgeBlock's adjoint as the prologue of the launch that consumes them, entries added in CSR order. */
  const int32_t* csr_rowptr;
  const float* csr_scale;
  float* save;        /* optional [M, 128]: the assembled rows of this segment (columns >= width are written as zeros) */
} gfv_seg_t;

enum {                /* per-layer element op applied to the accumulator */
  GFV_OP_NONE = 0,      /* v = acc + bias                                   */
  GFV_OP_BIAS_GELU = 1, /* v = acc + bias; save v; next input = gelu(v)     */
  GFV_OP_MUL_DGELU = 2  /* v = acc * gelu'(aux); save v; next input = v     */
};
enum { GFV_IN_NONE = 0, GFV_IN_GELU = 1, GFV_IN_LN = 2, GFV_IN_LNBWD = 3 };
/* gfv_rowtile_args_t.flags: the split-fp16 form has two kernel families - the row-owner chain (a wave owns 16 rows, weights
 * stream through LDS) and the column-owner persistent chain (a wave owns 16 columns, weights stay in registers for the whole
 * launch; 3-layer MLP launches).  By default big launches of a covered shape take the second. */
enum { GFV_CHAIN_ROW_OWNER = 1,     /* never the column-owner family */
       GFV_CHAIN_COLUMN_OWNER = 2   /* the column-owner family whatever the row count (if the shape is covered) */ };
enum { GFV_FIN_PLAIN = 0, GFV_FIN_LN = 1, GFV_FIN_LNBWD = 2 };

typedef struct {
  const float* W;     /* [N, K] row-major: nn.Linear weight (forward) or its transpose (dX chain); NULL only with Wh: a
                       * virtual layer whose 128-row passes come from different weight blocks, image = their images back to back */
  const float* bias;  /* [N] or NULL */
  int32_t K, N;       /* K = input width, N = output width (128 for inner layers; last layer: <= 384) */
  int32_t op;
  int32_t ldw;        /* row stride of W in floats; 0 = K (a column block of a wider weight: W1[:, 256:384]) */
  float* save;        /* optional [M, N] */
  const float* aux;   /* [M, N] for GFV_OP_MUL_DGELU */
  const void* Wh;     /* optional: split-fp16 image of W (gfv_weight_images); when every layer of a launch has one
                       * (and args.wmax is set) the products run on the f16 MFMA pipe, see below */
  const float* bias2; /* optional: bias of output columns >= 128 (a last layer whose two 128-row passes come from
                       * different Linear layers); NULL = bias[128..] */
} gfv_layer_t;

typedef struct {
  int32_t M;
  int32_t nseg;
  gfv_seg_t seg[3];       /* input row m = concat_i seg[i][idx_i[m], 0:width_i] */
  const float* in_add;    /* optional: seg[0] row += in_add row (same idx / ld as seg[0]) */
  int32_t in_op;
  int32_t nlayers;
  const float* in_gamma;  /* GFV_IN_LN / GFV_IN_LNBWD */
  const float* in_beta;
  const float* in_aux;    /* GFV_IN_LNBWD: the pre-LayerNorm values y [M,128] */
  const float* gadd;      /* optional [*,64]: input row m += [gadd[gadd_s[m]] | gadd[gadd_r[m]]] (before in_op) */
  const int32_t* gadd_s;
  const int32_t* gadd_r;
  float* in_save;         /* optional [M,128]: result of the prologue */
  float* ln_partial;      /* GFV_IN_LNBWD / GFV_FIN_LNBWD: [n_tiles, 2, 128] per-tile (dgamma, dbeta) partial sums; provide
                           * gfv_rowtile_ln_rows(M) rows, sum the first gfv_rowtile_last_ln_rows() of them (below) */
  gfv_layer_t layer[3];
  int32_t fin_op;
  int32_t hidden;         /* hidden_size of the model this launch belongs to (LayerNorm width; gfv_set_hidden_size); 0 = the calling
                           * thread's setting.  (Inside the library the launcher always fills it in for the kernel.) */
  const float* fin_gamma;
  const float* fin_beta;
  const float* fin_aux;   /* GFV_FIN_LNBWD: the LayerNorm input rows [M,128] */
  float* fin_presave;     /* GFV_FIN_LN: pre-LayerNorm values [M,128] */
  const float* res[3];    /* optional addend per 128-wide output chunk */
  int32_t res_ld[3];
  int32_t out_ld[3];
  float* out[3];          /* output pointer per 128-wide chunk of the last layer */
  float* out_nores;       /* optional [M,128]: chunk 0 result BEFORE the residual addend (EdgeBlock output e') */
  /* optional gathered addend to the FIRST layer's pre-activation (needs nlayers >= 2):
   *   z1[m, c] += padd[padd_s[m], c] + padd[padd_r[m], 128 + c],  c < 128,  padd = [*, padd_ld >= 256].
   * EdgeBlock first layer factored through the nodes: W1 [x_s | x_r | e] = (W1a x)[s] + (W1b x)[r] + W1c e
   * (blocks.py:54 concat + EPD.py:21 Linear), the two node-level products are computed once per node. */
  const float* padd;
  const int32_t* padd_s;
  const int32_t* padd_r;
  int32_t padd_ld;
  int32_t flags;          /* GFV_CHAIN_*: which kernel family may take the launch (0: the library decides by shape and size) */
  const float* wmax;      /* device scalar max|W| the layers' Wh images were built with (gfv_weight_images) */
  /* optional, written by the split-fp16 form only (gfv_rowtile_last_path() >= 5): per group of 16 consecutive rows the
   * exact power of two s with s * max|v| in [2^13, 2^14) of the gradient rows this launch leaves for the weight-gradient
   * kernel - slot 0: the prologue result (in_save), slot l + 1: layer l's saved GFV_OP_MUL_DGELU product, slot nlayers
   * (nlayers <= 2): output chunk 0.  Layout gscale[slot * gscale_ld + row / 16], gscale_ld >= ceil(M / 16).
   * gfv_dw_tile_t.gscale takes a slot: the slab scale of the gradient rows then needs no extra pass over them. */
  float* gscale;
  int32_t gscale_ld;
  int32_t product_form;   /* 0 = gfv_f16split_enabled() of the calling thread; 1 fp32 MFMA, 2 split-fp16, 3 / 4 reduced precision (fp16 / bf16) */
  /* LayerNorm statistics of the rows, [M, 2] = (mean, 1 / sqrt(var + eps)): written by a GFV_FIN_LN launch when fin_stats is
   * given, read by a GFV_IN_LNBWD launch when in_stats is given (the column-owner backward needs them: its LayerNorm backward
   * is spread over eight waves and takes the row statistics as they were in the forward instead of recomputing them) */
  float* fin_stats;
  const float* in_stats;
  /* Weight gradients fused into a dX chain (column-owner family only; gfv_rowtile_fuses_dw() tells whether a launch takes
   * it).  A backward chain [W3^T (x gelu'(z2)), W2^T (x gelu'(z1)), W1^T] holds, tile by tile, exactly the operands of the
   * weight gradients of the forward's third and second Linear:  dW3 = g3^T gelu(z2),  dW2 = gz2^T gelu(z1)  (g3 = the
   * prologue result, gz2 = layer 0's product, z2 / z1 = layer[0].aux / layer[1].aux), their bias gradients (column sums of
   * g3 / gz2) and the LayerNorm's (dgamma, dbeta).  With dw_partial set the launch accumulates them per workgroup - no float
   * atomics - and leaves  dw_partial[wg * dw_partial_stride + ...] = [dW3 (128 x 128, row n, column k) | db3 (128) | dW2 |
   * db2 | dgamma | dbeta]  (GFV_DW_FUSED_FLOATS floats) for wg < gfv_rowtile_dw_partials_m(M); sum them with gfv_reduce_multi.
   * BLOCKS AT OR BEYOND gfv_rowtile_dw_partials_m(M) ARE NOT WRITTEN (ABI 3; until round 5 a launch zero-filled all
   * gfv_rowtile_dw_partials() blocks): reduce exactly gfv_rowtile_dw_partials_m(M) of them.
   * layer[0].save / in_save / ln_partial may then be NULL (nothing else reads g3 / gz2). */
  float* dw_partial;
  int64_t dw_partial_stride;
  /* RESERVED (ABI 3): rounds 3 - 5 took the input rows of the forward's first Linear here and fused that weight gradient as well
   * (GFV_DW_FUSED_FLOATS_IN floats per block) - time-neutral at its best (profiles/r05_ab_dw1_trailing.txt), removed with its
   * kernel instantiations.  Must be NULL / 0: a launch with dw_in set does not run fused (gfv_rowtile_fuses_dw says 0). */
  const float* dw_in;
  int32_t dw_in_ld;
  int32_t reserved2_;
  /* optional with dw_partial (ABI 2): RECOMPUTE instead of re-read.  rc_Wh[0] / rc_Wh[1] = split-fp16 images (built with the
   * same wmax) of the FORWARD's second and third Linear ([128, 128] each, un-transposed), rc_bias[0] / rc_bias[1] their biases
   * (or NULL).  The launch then rebuilds z2 = W2 gelu(z1) + b2 and the LayerNorm input y = W3 gelu(z2) + b3 tile by tile from
   * z1 (layer[1].aux) on the matrix cores; layer[0].aux (z2) and in_aux (y) are NOT read and may be NULL - the forward launch
   * need not save them (gfv_layer_t.save of its second layer, fin_presave).  in_stats is still read (the rows' forward
   * statistics). */
  const void* rc_Wh[2];
  const float* rc_bias[2];
} gfv_rowtile_args_t;
enum { GFV_DW_FUSED_FLOATS = 2 * 128 * 128 + 4 * 128 };
int gfv_rowtile_dw_partials(void);                              /* the most workgroups (= partial blocks) a fused launch runs */
int gfv_rowtile_dw_partials_m(int32_t M);                       /* ... a fused launch over M rows runs: blocks 0 .. this - 1 are written */
int gfv_rowtile_fuses_dw(const gfv_rowtile_args_t* args);       /* 1: gfv_rowtile_chain would run this launch with fused weight gradients */

int gfv_rowtile_tiles(int32_t M); /* number of 64-row tiles = rows of ln_partial every family but the small-tile backward fills */
/* rows of ln_partial to PROVIDE for a launch over M rows (one per 32 rows), and how many of them the calling thread's last
 * gfv_rowtile_chain launch filled: ceil(M / 32) when the column-owner small-tile backward took it (csrc/cbwd.hip; last_path & 128),
 * gfv_rowtile_tiles(M) otherwise.  The rows beyond that count are not written. */
int gfv_rowtile_ln_rows(int32_t M);
int gfv_rowtile_last_ln_rows(void);
/* Dispatch limits (round 6): which kernel family takes a launch of M rows is decided by ONE table (csrc/gfv_limits.h lists the
 * entries: GFV_CBWD_MAX_M, GFV_CFWD_TG2_MAX_M, GFV_CFWDP_MIN_M ...).  A limit takes the value of the environment variable of its
 * name ONCE, at first use (or its built-in default); gfv_set_limit moves it afterwards (value < 0: back to environment / default) -
 * process-wide, for tests and A/B tools.  gfv_limit_name(which) = the variable's name, NULL past the last entry. */
int gfv_get_limit(int32_t which);
int gfv_set_limit(int32_t which, int32_t value);
const char* gfv_limit_name(int32_t which);
int gfv_rowtile_chain(const gfv_rowtile_args_t* args, void* stream);
/* which kernel the calling thread's last gfv_rowtile_chain launch took: 1 register-resident chain, 2 its ragged-shape
 * instantiation (0 was the generic LDS row-tile kernel, retired with ABI 2); + 4 when the products ran as split-fp16; + 8 when the column-owner persistent
 * family took the launch, + 16 when it ran with fused weight gradients, + 32 when a single-layer launch ran on the lean
 * one-Linear kernel (csrc/lin1.hip: the layer's image staged in LDS once per 128-row workgroup; same products, same
 * results to rounding), + 64 when a short 3-layer LayerNorm forward ran on the column-owner small-tile kernel (csrc/cfwd.hip:
 * a wave owns 32 output columns of a 32- / 64-row tile; hidden activations split behind a fixed scale as in the column-owner
 * backward), + 128 when a short dX chain behind a LayerNorm backward (in_stats given, no fused weight gradients) ran on the
 * column-owner small-tile backward (csrc/cbwd.hip: 32- / 64-row tiles on 8 waves, the scales of the persistent backward)
 * (tests assert the path they mean) */
int gfv_rowtile_last_path(void);

/* fp32 products on the f16 MFMA pipe (v_mfma_f32_16x16x32_f16, 16x the f32 MFMA rate): every fp32 operand is split
 * into two fp16 parts, x = hi + lo (22 mantissa bits after an exact power-of-two scaling: per input row for the
 * activations, one global scale 2^s from max|W| for the weights), and hi*hi + hi*lo + lo*hi is accumulated in fp32 -
 * 3 MFMAs instead of 8, error <= that of the f32 MFMA (products are exact, the dropped lo*lo term is 2^-22 relative).
 * The weights are split once per step into an image in MFMA-fragment order:
 *   image[pass = n/128][T = k/32][nt = (n%128)/16][part][lane = 16 g + i][e]  (fp16),
 *   element = part(2^s W[128 pass + 16 nt + i][32 T + 16 (e>>2) + 4 g + (e&3)]),  zero outside [N, K]
 * (16 KB per (pass, T): exactly the LDS slice the chain kernel streams).  The activations are split in registers. */
typedef struct {
  const float* W; /* [N, K], row stride ldw */
  void* img;      /* gfv_weight_image_bytes(N, K) bytes */
  int32_t ldw, N, K, reserved;
} gfv_wimg_desc_t;
size_t gfv_weight_image_bytes(int32_t N, int32_t K);
/* wmax[0] = max |W| over all described blocks (device scalar, overwritten) */
/* the product form of the launches that do not name their own: a PROCESS-WIDE default (initial value: environment
 * GFV_F16SPLIT, default 1; gfv_set_f16split) - it must reach launches issued from other threads than the one that chose it:
 * PyTorch runs the backward of an autograd node on its device worker thread - and an optional override of the CALLING THREAD
 * (gfv_set_f16split_thread(0 / 1 / 2 / 3); -1 removes it) for hosts that drive two models in different forms from two threads.
 * gfv_set_f16split also removes the calling thread's override.  gfv_f16split_enabled() = what a launch from this thread gets.
 * 0 = fp32 MFMA everywhere (chain launches ignore their images, weight gradients take the fp32 kernel);
 * 1 = split-fp16 products (fp32 accuracy);
 * 2 = reduced precision: one fp16 x fp16 product per term with fp32 accumulation - the high parts of the same operands
 *     (11 significand bits: results agree with the fp32 forms to ~1e-3; the counterpart of the reference's autocast runs,
 *     BASELINE configs 3 / 5 - never the form the parity claims or bench.py's `value` are made on);
 * 3 = the same single product with bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulation; 8 significand bits: ~1e-2 of
 *     the fp32 forms) - BASELINE config 3's "bf16 MLP GEMMs on MFMA" to the letter.  Same fragment layouts and scales; the
 *     weight images hold bf16 high parts and no low parts, so they must be BUILT in this form: gfv_weight_images reads the
 *     calling thread's form, and images built in forms 0 - 2 are not valid for form 3 (nor the other way round). */
int gfv_f16split_enabled(void);
int gfv_set_f16split(int32_t on);
int gfv_set_f16split_thread(int32_t on);
/* hidden_size of the model the following launches belong to (the reference's --hidden_size, utils/get_param.py:69; default
 * 128; multiples of 16 in [16, 128]).  Every kernel works on 128-column latent rows; a narrower model runs zero-padded to
 * 128 columns (the host side pads its parameters: FVMmodel/padding.py) and differs in two places only - LayerNorm takes its
 * statistics over the h real columns, and the slice attention scales by (h / 8) ** -0.5.  Thread-local host state read by
 * the launchers: set it before the launches of a model (gfv.engine.Engine does, on every forward and backward); a chain
 * launch may name its own (gfv_rowtile_args_t.hidden). */
int gfv_hidden_size(void);
int gfv_set_hidden_size(int32_t h);
int gfv_weight_absmax(const gfv_wimg_desc_t* descs_dev, int32_t n_desc, float* wmax, void* stream);
/* build every image; max_frags = max over descs of gfv_weight_image_bytes / 32 */
int gfv_weight_images(const gfv_wimg_desc_t* descs_dev, int32_t n_desc, int64_t max_frags, const float* wmax,
                      void* stream);
/* The form tag of a set of images: -1 = `wmax` is not the scale of images this library built, 0 = fp16 (hi, lo) parts (built in
 * forms 0 / 1 / 2), 1 = bf16 high parts (built in form 3).  Kept per `wmax` address (host state, set by gfv_weight_images);
 * gfv_rowtile_chain and gfv_trans_mlp_fwd / _bwd return GFV_ERR_ARG for a launch whose product form is of the other class than
 * the images it names through args.wmax - images carry nothing in their bytes that would tell. */
int gfv_weight_images_form(const float* wmax);

/* Weight gradient of a Linear layer: dW[n,k] = sum_m G[m,n] * A[m,k], db[n] = sum_m G[m,n] (two-stage,
 * deterministic).  A is assembled like the forward input (segments, gather, optional GELU of a saved
 * pre-activation).  partial: workspace [gfv_dw_chunks(M), N(=128), Kpad] (+ N floats per chunk for db).
 * Replaces the autograd mm/addmm wgrad of nn.Linear (pre_train_Adam.py:188). */
int gfv_dw_chunks(int32_t M);
int gfv_linear_dw(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                  const float* in_add, int32_t a_gelu, int32_t M, float* dW, float* db, float* workspace,
                  int32_t accumulate, void* stream);
/* same as _ex, with the per-16-row scales of the G rows handed over (gfv_dw_tile_t.gscale; NULL = none) */
int gfv_linear_dw_gs(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                     const float* in_add, int32_t a_op, const float* a_gamma, const float* a_beta, int32_t M,
                     float* dW, const float* gscale, float* db, float* workspace, int32_t accumulate, void* stream);
size_t gfv_linear_dw_workspace_floats(int32_t M, int32_t n_out, int32_t K);
/* same, with the input-row op spelled out: a_op 0 none, 1 GELU, 2 LayerNorm(a_gamma, a_beta) (single 128-wide segment) */
int gfv_linear_dw_ex(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                     const float* in_add, int32_t a_op, const float* a_gamma, const float* a_beta, int32_t M,
                     float* dW, int32_t reserved, float* db, float* workspace, int32_t accumulate, void* stream);

/* All weight gradients of one fused MLP (or any set of <= 6 output tiles) in ONE launch + ONE reduction.
 * Tile t: dW_t[n,k] = sum_m G_t[m,n] * op(A_t[idx_t[m], k]) for n < n_out, k < width, written at
 * block[out_off + n*ld_out + k]; if db_off >= 0 also block[db_off + n] = sum_m G_t[m,n].  `grad_block` is the
 * contiguous gradient storage of the parameter block ([W1|b1|W2|b2|W3|b3] in state_dict order, each tensor padded
 * to a multiple of 4 floats), block_floats its length.  workspace: gfv_dw_multi_workspace_floats(...) floats,
 * zero-initialised once by the caller (padding slots are never written). */
typedef struct {
  const float* G;       /* [M, ldg] upstream gradient rows */
  const float* A;       /* input rows [*, ld] */
  const int32_t* idx;   /* optional gather index for A rows */
  const float* in_add;  /* optional second addend for A (same idx / ld) */
  const float* a_gamma; /* a_op == 2 */
  const float* a_beta;
  int32_t ldg, n_out, width, ld;
  int32_t a_op;         /* 0 none, 1 GELU, 2 LayerNorm(a_gamma, a_beta) (width 128); | GFV_DW_COLSCALE (with 0 only) */
  int32_t ld_out;       /* row stride (K of the weight) inside the block */
  int64_t out_off;      /* float offset of dW_t[0, 0] inside the block */
  int64_t db_off;       /* float offset of the bias gradient, or -1 */
  const float* gscale;  /* optional (split-fp16 form): per-16-row power-of-two scales of the G rows as the chain launch
                         * that produced them wrote them (gfv_rowtile_args_t.gscale); NULL: one pass over G finds the maximum */
} gfv_dw_tile_t;
/* Range of the split-fp16 weight-gradient form.  The gradient rows G carry ONE power of two per slab of rows.  The
 * activations A carry one power of two per COLUMN and slab when the tile asks for it (a_op = 0 | GFV_DW_COLSCALE: a pass
 * over the slab's A rows finds the column maxima - encoder inputs hold geometric columns at mesh-spacing scale next to O(1)
 * features); otherwise they are split unscaled (latent rows, GELU / LayerNorm outputs: O(1)), and a value beyond the fp16
 * range raises GFV_FLAG_DW_RANGE in the device status word instead of being clamped. */
enum { GFV_DW_COLSCALE = 8 };
enum { GFV_FLAG_DW_RANGE = 1,
       /* column-owner chain family: a hidden activation beyond 2^11 (they are split after a fixed scale, see
        * csrc/colchain_kernel.h; GFV_COLCHAIN=0 keeps every launch on the row-scaled family) */
       GFV_FLAG_CHAIN_RANGE = 2 };
/* device status word: OR of GFV_FLAG_* raised by kernels since the last call; reads (synchronising) and clears it */
int gfv_status_flags(int32_t* flags_out);
/* The same word WITHOUT a synchronisation (round 6): *host_word = a pinned, device-mapped int32 the library owns (allocated at the
 * first call, one per process).  gfv_status_publish enqueues a one-thread kernel on `stream` that copies a non-zero status word
 * into it (the device word stays raised until gfv_status_flags clears it); gfv_adam_step_dev does the same from inside the Adam
 * launch once the mirror exists.  A host loop reads *host_word after the NEXT step was issued (the previous step's
 * kernels have long finished: no wait), and on a non-zero value calls gfv_status_flags (clears the device word), zeroes
 * *host_word and raises.  gfv.trainer.TrainStep.step and FVMmodel.importer.NNmodel.forward do exactly that (FloatingPointError). */
int gfv_status_mirror(int32_t** host_word);
int gfv_status_publish(void* stream);
int gfv_dw_slabs(int32_t M, int32_t ntiles, int32_t* rows_per_slab);
size_t gfv_dw_multi_workspace_floats(int32_t M, int32_t ntiles, int64_t block_floats);
int gfv_dw_multi(const gfv_dw_tile_t* tiles, int32_t ntiles, int32_t M, int64_t block_floats, float* workspace,
                 float* grad_block, int32_t accumulate, void* stream);

/* out[j] (+)= sum_c partial[c, j]  (c < n_chunks, j < n) */
int gfv_reduce_partials(const float* partial, int32_t n_chunks, int32_t n, float* out, int32_t accumulate,
                        void* stream);

/* 2-D form: out[r * ld_out + c] = sum_chunk partial[chunk * chunk_stride + r * cols + c] (r < rows, c < cols; cols, ld_out,
 * chunk_stride multiples of 4, 16-byte aligned): a [rows, cols] piece of the slab workspace of gfv_dw_multi reduced straight
 * into a column block of a wider gradient matrix.  gfv_dw_multi with grad_block == NULL leaves the reduction to the caller. */
int gfv_reduce_partials_2d(const float* partial, int32_t n_chunks, int64_t chunk_stride, int32_t rows, int32_t cols,
                           int32_t ld_out, float* out, void* stream);

/* Several such reductions in ONE launch (the parameter gradients of one MLP: slab partials of gfv_dw_multi with
 * grad_block == NULL, per-tile LayerNorm partials, small per-block partials), each straight into its place:
 *   out[r * ld_out + c] = sum_{chunk < n_chunks} partial[chunk * chunk_stride + r * ld_in + c]   (r < rows, c < cols)
 * cols, ld_in, ld_out, chunk_stride multiples of 4 floats, pointers 16-byte aligned; n_pieces <= 12.  Fixed summation
 * order (16 interleaved chunk lanes, then an ordered fold): deterministic, no atomics. */
typedef struct {
  const float* partial;
  float* out;
  int64_t chunk_stride; /* floats between consecutive chunks */
  int32_t n_chunks, rows, cols, ld_in, ld_out, reserved;
} gfv_reduce_piece_t;
int gfv_reduce_multi(const gfv_reduce_piece_t* pieces, int32_t n_pieces, void* stream);

/* Segmented form: out[b, 0:n] = sum of the chunk rows seg_ptr[b] .. seg_ptr[b+1]-1 of partial [n_chunks, n] (n % 4 == 0,
 * 16-byte aligned).  Pre-reduces the per-chunk slice tokens of each graph (GraphTransolver.py:64-73 global_add_pool). */
int gfv_reduce_partials_seg(const float* partial, const int32_t* seg_ptr, int32_t n_seg, int32_t n, float* out,
                            void* stream);

/* Batched form: one launch for all weight transposes of a step; `descs` [n] lives in DEVICE memory. */
typedef struct {
  const float* in; /* [rows, ld_in], columns [0, cols) are transposed */
  float* out;      /* [cols, ld_out] */
  int32_t rows, cols, ld_in, ld_out; /* ld_out: row stride of out, 0 = rows */
} gfv_transpose_desc_t;
int gfv_transpose_batch(const gfv_transpose_desc_t* descs, int32_t n, int32_t max_rows, int32_t max_cols, void* stream);

/* Batch assembly of the device-resident state pool (replaces the per-step host batching + H2D copy of
 * Load_mesh/Graph_loader.py:405-480,830-1006): descriptor i copies n_words 32-bit words src -> dst, kind 0 verbatim
 * (floats, ids), kind 1 adding `add` to every (int32) word (index offsets of the block-diagonal batch), kind 2 filling
 * `add` (graph id).  `descs` [n_desc] lives in DEVICE memory; one launch for the whole batch. */
typedef struct {
  const void* src;
  void* dst;
  int64_t n_words;
  int32_t kind;
  int32_t add;
} gfv_concat_desc_t;
int gfv_concat_offsets(const gfv_concat_desc_t* descs, int32_t n_desc, int32_t blocks_per_desc, void* stream);

/* out [cols, rows] = in^T for a [rows, cols] fp32 matrix with row stride ld_in (weights for the dX chain). */
int gfv_transpose(const float* in, int32_t ld_in, float* out, int32_t rows, int32_t cols, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * Transolver physics attention (H=8 heads, D=16, G=32 slices).  Replaces
 * FVMmodel/Models/GraphTransolver/GraphTransolver.py:61-95 (softmax slice weights, scatter_add slice tokens over
 * `batch`, 32x32 attention per (graph, head), de-slice) without materialising [N,8,32,16].
 * Layouts: xmid/fx/out_x [N,128] = [N,8,16]; w, gw [N,8,32]; tokens [B,8,32,16]; norm [B,8,32]; attn [B,8,32,32].
 * Node chunks: contiguous node ranges inside one graph (chunk_beg/chunk_end [n_chunks]; gchunk_ptr [B+1]).
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_slice_softmax_fwd(const float* xmid, const float* Ws, const float* bs, const float* temp, float* w, int32_t N,
                          void* stream);
int gfv_slice_softmax_bwd_blocks(int32_t N); /* rows of `partial` ([blocks][552] = dWs 512 | dbs 32 | dT 8) */
int gfv_slice_softmax_bwd(const float* xmid, const float* Ws, const float* bs, const float* temp, const float* w,
                          const float* gw, float* gxmid, float* partial, int32_t N, void* stream);
/* partial[chunk][h*32+g][0:16] = sum_n w[n,h,g] a[n,h,:], [16] = sum_n w[n,h,g] */
int gfv_slice_token_partial(const float* w, const float* a, const int32_t* chunk_beg, const int32_t* chunk_end,
                            int32_t n_chunks, float* partial, void* stream);
int gfv_slice_attention_fwd(const float* partial, const int32_t* gchunk_ptr, int32_t B, const float* Wq, const float* Wk,
                            const float* Wv, float* token, float* norm, float* attn, float* out_token, void* stream);
/* dW_partial [B*8][3][16][16] (q,k,v), reduce with gfv_reduce_partials */
int gfv_slice_attention_bwd(const float* gpartial, const int32_t* gchunk_ptr, int32_t B, const float* Wq, const float* Wk,
                            const float* Wv, const float* token, const float* norm, const float* attn, float* g_raw,
                            float* g_norm, float* dW_partial, void* stream);
/* gfv_slice_softmax_fwd followed by gfv_slice_token_partial(w, a) in one pass (w is written out as well): the forward of
 * GraphTransolver.py:64-73 up to the per-chunk slice tokens, on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) */
int gfv_slice_softmax_token(const float* xmid, const float* Ws, const float* bs, const float* temp, const float* a,
                            const int32_t* chunk_beg, const int32_t* chunk_end, int32_t n_chunks, float* w, float* partial,
                            void* stream);
/* out[n,h,:] (+)= sum_g w[n,h,g] T[batch[n],h,g,:].  accumulate: bit 0 = add into out; bit 2 (value 4) = the batch is ONE
 * graph (spares the pass over workgroups that straddle two graphs) */
int gfv_deslice(const float* w, const float* T, const int32_t* batch, float* out, int32_t N, int32_t accumulate,
                void* stream);
/* gw[n,h,g] (+)= sum_c a[n,h,c] T[batch[n],h,g,c] + add[batch[n],h,g] */
int gfv_slice_gw(const float* a, const float* T, const float* add, const int32_t* batch, float* gw, int32_t N,
                 int32_t accumulate, void* stream);
/* Everything behind the attention adjoint of a Transolver block in one pass over the nodes (GraphTransolver.py:64-92,
 * backward): gw = gfv_slice_gw(g_out_x, out_token) + gfv_slice_gw(fx_mid, g_raw, g_norm) stays in registers, g_fx_mid =
 * gfv_deslice(w, g_raw), (g_x_mid, partial) = gfv_slice_softmax_bwd(.., gw) - the same terms in the same order as those
 * four launches (equal to rounding).  partial: gfv_slice_softmax_bwd_blocks(N) x 552 floats.  n_graphs: graphs in the batch
 * (1: every workgroup of 32 nodes lies in one graph and runs on the matrix cores, v_mfma_f32_16x16x4_f32 - exact fp32
 * products; otherwise the workgroups that straddle two graphs are taken by a second, scalar launch). */
int gfv_slice_post_bwd(const float* xmid, const float* Ws, const float* bs, const float* temp, const float* w,
                       const float* g_out_x, const float* out_token, const float* fx_mid, const float* g_raw,
                       const float* g_norm, const int32_t* batch, float* g_x_mid, float* g_fx_mid, float* partial,
                       int32_t N, int32_t n_graphs, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Finite-volume discretisation (conserved form).  Replaces FVMmodel/FVdiscretization/FVscheme.py:618-724,50-274,
 * FVgrad.py:235-367 (precomputed-moments branch), FVInterpolation.py:36-185,218-265 and the clamp / Dirichlet /
 * integrator mixing of FVMmodel/importer.py:187-201,223-231.  Padded layouts: phi [N,8] = (u,v,p,uh,vh,uo,vo,0),
 * grad [N,16] = d(phi_c)/d(x,y) at 2c+a, Ff [E,16] = (phi_f[0:5], grad_f[c][a] at 5+2c+a), cres [C,4] =
 * (div, Rx, Ry, sum |lp|^2), losses [B,4] = (cont, mom_x, mom_y, press).  mode: 0 explicit, 1 implicit, 2 imex.
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_phi_fwd(const float* dec, const float* y, const int32_t* node_type, const float* uv_old, float* phi, int32_t N,
                int32_t mode, void* stream);
int gfv_phi_bwd(const float* gphi, const float* dec, const int32_t* node_type, float* gdec, int32_t N, int32_t mode,
                void* stream);
/* rowptr/outn/Bp: directed stencil in CSR order of the receiving node, Bp [S,5] permuted moment vectors;
 * An [N,25] the moment matrix A as the reference stores it (graph_node_x.A_node_to_node, fp32), rn [N,5] = its row norms
 * + 1e-8 (FVgrad.py:335).  The kernel forms A / rn, accumulates the right-hand side and runs the pivoted LU + solves in
 * DOUBLE: cond(A_n) reaches 4e5 on the reference's boundary-layer meshes, where an fp32 elimination (the reference's own
 * included) returns rounding noise of size cond * 2^-24; the exact solution of the fp32 data sits in the middle of it. */
int gfv_wlsq_fwd(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp, const float* An,
                 const float* rn, float* grad, int32_t N, void* stream);
/* adjoint; rowptr_o/inn/Bo: the same stencil in CSR order of the SENDING node, sumB [N,5]; grhs_ws [N,8,5];
 * adds into gphi */
int gfv_wlsq_bwd(const float* ggrad, const float* An, const float* rn, const int32_t* rowptr_o, const int32_t* inn,
                 const float* Bo, const float* sumB, float* grhs_ws, float* gphi, int32_t N, void* stream);
/* stand-alone node_based_WLSQ: all five 2nd-order derivative entries, full5 / g5 laid out [N,8,5]; 7 channels */
int gfv_wlsq_fwd_full(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp, const float* An,
                      const float* rn, float* grad, float* full5, int32_t N, void* stream);
int gfv_wlsq_bwd_full(const float* g5, const float* An, const float* rn, const int32_t* rowptr_o, const int32_t* inn,
                      const float* Bo, const float* sumB, float* grhs_ws, float* gphi, int32_t N, void* stream);
/* any reconstruction order (FVorder.py:23-72; row f4): terms = Taylor terms M = 2 (1st) / 5 (2nd) / 9 (3rd) / 14 (4th);
 * Bp / Bo [S,M], An [N,M*M], rn / sumB [N,M], grhs_ws [N,8,M].  full / gfull (optional) [N,8,M]: all M derivative entries
 * (7 channels) out of the forward / into the adjoint; the adjoint takes exactly one of ggrad [N,16] and gfull. */
int gfv_wlsq_fwd_ex(const float* phi, const int32_t* rowptr, const int32_t* outn, const float* Bp, const float* An,
                    const float* rn, float* grad, float* full, int32_t N, int32_t terms, void* stream);
int gfv_wlsq_bwd_ex(const float* ggrad, const float* gfull, const float* An, const float* rn, const int32_t* rowptr_o,
                    const int32_t* inn, const float* Bo, const float* sumB, float* grhs_ws, float* gphi, int32_t N,
                    int32_t terms, void* stream);
int gfv_face_fwd(const float* phi, const float* grad, const int32_t* es, const int32_t* er, const float* pos,
                 const float* fpos, const int32_t* ftype, const float* y, float* Ff, int32_t E, void* stream);
int gfv_cell_fwd(const float* phi, const float* grad, const float* Ff, const float* pos, const int32_t* crow,
                 const int32_t* kface, const int32_t* knode, const float* kS, const int32_t* ftype, const float* centroid,
                 const float* area, const int32_t* cbatch, const float* theta, const float* dt, const float* uvp_dim,
                 const float* sigma, float* phic, float* cres, float* uvp_cell, int32_t C, void* stream);
/* Same with the residuals of the NON-conserved form (FVscheme.py:276-511 as the reference calls it, hessian None):
 * continuity, convection and pressure terms from the cell means of the node gradients (saved to gradc [C,16]). */
int gfv_cell_fwd_ex(const float* phi, const float* grad, const float* Ff, const float* pos, const int32_t* crow,
                    const int32_t* kface, const int32_t* knode, const float* kS, const int32_t* ftype,
                    const float* centroid, const float* area, const int32_t* cbatch, const float* theta, const float* dt,
                    const float* uvp_dim, const float* sigma, float* phic, float* cres, float* uvp_cell, int32_t C,
                    int32_t non_conserved, float* gradc, void* stream);
int gfv_graph_loss(const float* cres, const int32_t* gcell_ptr, const float* theta, const float* sigma, float* sums,
                   float* losses, int32_t B, void* stream);
int gfv_cell_to_node(const float* phic, const int32_t* nrow, const int32_t* ncell, const float* pos, const float* centroid,
                     const int32_t* node_type, const float* y, const int32_t* nbatch, const float* uvp_dim,
                     const float* sigma, const float* phi, int32_t smooth, float* out, int32_t N, void* stream);
/* adjoint of face_fwd + cell_fwd + graph_loss: gloss [B,4] -> gphi [N,8], ggrad [N,16] (overwritten) */
int gfv_fvm_bwd(const float* cres, const float* sums, const float* gloss, const float* Ff, const int32_t* cbatch,
                const float* theta, const float* sigma, const float* dt, const int32_t* frow, const int32_t* fk,
                const int32_t* kcell, const float* kS, const int32_t* ftype, const int32_t* nfrow, const int32_t* nfcol2,
                const int32_t* nrow, const int32_t* ncell, const int32_t* crow, const float* pos, const float* fpos,
                const float* centroid, const float* area, float* gc_ws, float* gFf_ws, float* gphi, float* ggrad,
                int32_t N, int32_t E, int32_t C, void* stream);
int gfv_fvm_bwd_ex(const float* cres, const float* sums, const float* gloss, const float* Ff, const int32_t* cbatch,
                   const float* theta, const float* sigma, const float* dt, const int32_t* frow, const int32_t* fk,
                   const int32_t* kcell, const float* kS, const int32_t* ftype, const int32_t* nfrow, const int32_t* nfcol2,
                   const int32_t* nrow, const int32_t* ncell, const int32_t* crow, const float* pos, const float* fpos,
                   const float* centroid, const float* area, float* gc_ws, float* gFf_ws, float* gphi, float* ggrad,
                   int32_t N, int32_t E, int32_t C, int32_t non_conserved, const float* gradc, const float* phic,
                   void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input normalisation / edge features / optimiser.  Replace FVMmodel/importer.py:54-93,114-130,166-178,
 * utils/normalization.py:32-85, torch.optim.Adam (pre_train_Adam.py:79,191) and the loss of
 * pre_train_Adam.py:177-184.
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_graph_norm_stats(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats, void* stream);
/* the same over 64 workgroups per graph (a single workgroup reads at the bandwidth of one CU): sums of x and x^2 in double per
 * workgroup, folded in a fixed order by a second small launch; workspace: gfv_graph_norm_workspace_bytes(B) bytes */
size_t gfv_graph_norm_workspace_bytes(int32_t B);
int gfv_graph_norm_stats_ws(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats, void* workspace,
                            void* stream);
int gfv_normalizer_blocks(int32_t N);
int gfv_normalizer_update(const float* x, int32_t ldx, int32_t N, int32_t accumulate, float* acc_count, float* num_acc,
                          float* acc_sum, float* acc_sq, float* partial_ws, float* mean_std, void* stream);
int gfv_node_prep(float* x, int32_t ldx, const int32_t* batch, const float* stats, const float* uvp_dim,
                  const float* mean_std, int32_t norm_global, float* uv_old, int32_t N, void* stream);
int gfv_edge_attr(const float* x, int32_t ldx, const float* pos, const int32_t* es, const int32_t* er, float* out16,
                  float* out15, int32_t E, void* stream);
/* Round 6 - the input preparation as TWO launches (importer.py:114-130,166-178: was graph_norm_stats_ws + normalizer_update +
 * node_prep + edge_attr, and a state-restore copy in front of them in the solve loop, solve_with_grad_GPU.py:143).
 * gfv_prep_stats: stats[B,6] as gfv_graph_norm_stats_ws forms them (same partial sums, same fold order - by the workgroup of
 *   the graph that arrives last); workspace: gfv_prep_workspace_bytes(B) bytes, 8-byte aligned, its first 4096 bytes (the
 *   arrival counters of up to 1024 graphs: B <= 1024) ZERO before the first launch - every launch leaves them at zero, so one
 *   workspace serves batches of any size one after the other; one workspace per stream that may run this concurrently.
 *   x_raw != NULL: also copies the rows x[:, 0:12] -> x_raw [N,12] (a caller that normalises x in place needs the raw rows
 *   for the edge features).  mean_std != NULL: also derives the Normalizer's (mean[9], std[9]) from acc_count / acc_sum /
 *   acc_sq WITHOUT accumulating (= gfv_normalizer_update(accumulate = 0)); an accumulating step calls gfv_normalizer_update first.
 * gfv_prep_apply: node i < N: uv_old[i] = x_raw[i, 0:2] / uvp_dim[batch[i]], x_out[i] = the normalised row (gfv_node_prep's
 *   arithmetic); edge e < E: out16 / out15 = gfv_edge_attr's relative features of the NORMALISED end-node rows, formed from the
 *   raw rows with the same expressions (bit-identical to gfv_node_prep followed by gfv_edge_attr).  x_raw and x_out are
 *   [N,12] contiguous and must not alias. */
size_t gfv_prep_workspace_bytes(int32_t B);
int gfv_prep_stats(const float* x, int32_t ldx, const int32_t* gnode_ptr, int32_t B, float* stats, void* workspace, float* x_raw,
                   const float* acc_count, const float* acc_sum, const float* acc_sq, float* mean_std, void* stream);
int gfv_prep_apply(const float* x_raw, float* x_out, const int32_t* batch, const float* stats, const float* uvp_dim,
                   const float* mean_std, int32_t norm_global, float* uv_old, int32_t N, const float* pos, const int32_t* es,
                   const int32_t* er, float* out16, float* out15, int32_t E, void* stream);
/* Fused Adam on the flat buffers (torch.optim.Adam defaults; pre_train_Adam.py:115,189-191).  Step counter and
 * hyper-parameters are DEVICE resident so that a captured hipGraph follows learning-rate changes:
 *   state[16]: [0] t = completed steps; [1..2] 1 - beta1^(t+1) as a (hi, lo) float pair and [3] sqrt(1 - beta2^(t+1)): the bias
 *              corrections of the NEXT step, formed in double as torch's host code does; [4] arrival counter (int32, 0 between
 *              launches); [5] 1 - beta1, [6] 1 - beta2 rounded from double (torch: `value = 1 - beta2`); [8..11] the running
 *              powers beta^(t+1) and [12..15] the betas as (hi, lo) pairs.  gfv_adam_state_init writes all of it for a given t and
 *              the betas in DOUBLE (once, and again after a checkpoint load or a change of the betas); every gfv_adam_step_dev
 *              launch then applies it with the lr of hyper[0] as it is at that moment and - its last workgroup to finish -
 *              advances t and the powers by one multiplication (ABI 3: ONE launch per step; ABI 2 had state[4] and a tick
 *              launch in front, gfv_adam_tick_dev / gfv_adam_update_dev: removed)
 *   hyper[8] = {lr, beta1, beta2, eps, grad_scale (1 / world size), w_cont, w_mom, w_press}  (beta1 / beta2 here: the fp32 factors
 *              of the moment decay, as torch's fp32 kernels take them)
 * The same launch publishes the status word once gfv_status_mirror has been called. */
int gfv_adam_state_init(float* state, double beta1, double beta2, float steps_done, void* stream);
int gfv_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, const float* hyper,
                      void* stream);
int gfv_train_loss(const float* losses, int32_t B, float w_cont, float w_mom, float w_press, float* loss, float* gloss,
                   void* stream);
/* same, weights read from the device: hyper[5..7] = {w_cont, w_mom, w_press} of the buffer gfv_adam_step_dev takes */
int gfv_train_loss_dev(const float* losses, int32_t B, const float* hyper, float* loss, float* gloss, void* stream);

/* Round 6 - the finite-volume section with fewer launches (FVscheme.py:145-262 and its adjoint; the stand-alone entry points
 * above stay what the operator modules use and what these are tested against, bit for bit).  The mesh tables of a batch in one
 * struct (all device pointers, the tables of gfv_plan_create / gfv/plan.py): */
typedef struct {
  int32_t N, E, C, B;
  int32_t terms;        /* Taylor terms of the WLSQ order: 2 / 5 / 9 / 14 */
  int32_t mode;         /* integrator: 0 explicit, 1 implicit, 2 imex (gfv_phi_fwd) */
  int32_t smooth;       /* gfv_cell_to_node's `smooth` bits */
  int32_t reserved;
  const int32_t* node_type; const int32_t* batch; const int32_t* ftype; const int32_t* cbatch; const int32_t* gcell_ptr;
  const float* pos; const float* fpos; const float* y; const float* centroid; const float* area;
  const float* theta; const float* sigma; const float* uvp_dim; const float* dt;
  const int32_t* crow; const int32_t* kcell; const float* kS;            /* cell <- (face, node) incidences */
  const int32_t* frow; const int32_t* fk;                                /* face <- incidences */
  const int32_t* nfrow; const int32_t* nfcol2;                           /* node <- faces (2 * face + side) */
  const int32_t* nrow; const int32_t* ncell;                             /* node <- incidences */
  const float* An; const float* rn;                                      /* WLSQ moment matrices as stored + row norms */
  const int32_t* xo_rowptr; const int32_t* xo_in; const float* xo_B; const float* sumB;   /* stencil in sender order */
} gfv_fvm_mesh_t;
/* forward tail: gfv_graph_loss + gfv_train_loss_dev (hyper != NULL; its last-arriving workgroup) + gfv_cell_to_node (uvp_node !=
 * NULL) as ONE launch.  counter: one int32 of the caller's, zero before the first launch (the launch leaves it zero). */
int gfv_fvm_fwd_tail(const gfv_fvm_mesh_t* mesh, const float* cres, const float* phic, const float* phi, float* sums, float* losses,
                     float* uvp_node, const float* hyper, float* loss, float* gloss, int32_t* counter, void* stream);
/* backward of the conserved form from the loss gradients to the gradient of the decoder output, THREE launches (was six:
 * gfv_fvm_bwd_ex's cell / face / node kernels, gfv_wlsq_bwd_ex's solve / gather, gfv_phi_bwd): per face (the cell factors formed on
 * the fly); per (node, channel) the node adjoint + its transposed WLSQ solve; per (node, channel) the stencil gather + the
 * decoder-output adjoint.  Workspaces: gFf_ws [E,16], gphi_ws [N,8], grhs_ws [N,8,terms].  Same sums in the same order as the six. */
int gfv_fvm_bwd_fused(const gfv_fvm_mesh_t* mesh, const float* cres, const float* sums, const float* gloss, const float* Ff,
                      const float* dec, float* gFf_ws, float* gphi_ws, float* grhs_ws, float* gdec, void* stream);

/* k-hop reconstruction stencil of a mesh on the device (per-mesh preprocessing; parse_to_h5.py:228-254,
 * Load_mesh.py:421-521): the unordered node pairs (i < j) with j within k edges of i, as np.unique(axis=1) orders them
 * (by i, then j).  face0 / face1 [F]: the end nodes of every face (int64, as the mesh files hold them; duplicates allowed).
 *   gfv_khop_count builds the CSR adjacency and counts; the number of pairs is then ws[4 (N + 1) - 1] (device memory:
 *   read it back, allocate out0 / out1 [pairs] int64), ws[4 (N + 1)] != 0 says a neighbourhood exceeded 512 nodes (error);
 *   gfv_khop_fill writes the pairs.  ws: gfv_khop_workspace_ints(N, F) int32, untouched between the two calls. */
size_t gfv_khop_workspace_ints(int32_t N, int32_t F);
int gfv_khop_count(const int64_t* face0, const int64_t* face1, int32_t F, int32_t N, int32_t k, int32_t* ws, void* stream);
int gfv_khop_fill(int32_t N, int32_t k, const int32_t* ws, int64_t* out0, int64_t* out1, void* stream);

/* WLSQ moment matrices of a mesh on the device (per-mesh preprocessing; Load_mesh.py:247-272 ->
 * FVgrad.py:183-232 -> FVorder.py:7-86), float64: for node i and its directed stencil entries k in CSR order
 * (rowptr [N+1], outn [S] = the other node of entry k, entry [S] = the entry's position in the caller's edge order):
 *   d = pos[outn[k]] - pos[i], t = the `terms` (2 / 5 / 9 / 14) Taylor monomials of d, w = 1 / |d|,
 *   A[i] (+)= w t t^T  [N, terms, terms],   B[entry[k]] = w t  [S, terms]. */
int gfv_wlsq_moments(const double* pos, const int32_t* rowptr, const int32_t* outn, const int32_t* entry, double* A, double* B,
                     int32_t N, int32_t terms, void* stream);

/* Stand-alone 2nd-order interpolation (FVInterpolation.py `Interplot`: node_to_cell_2nd_order :36-109,
 * node_to_face_2nd_order :111-185, cell_to_node_2nd_order :218-265), any channel count C:
 *   out[r, c] = sum_{k in row r} w_k (phi[col[k], c] + (tgtpos[r] - srcpos[col[k]]) . grad[col[k], c, 0:2]) / W_r
 * mode 0: w = 1, W = max(entries, 1) (mean / average); mode 1: w = 1 / |tgtpos[r] - srcpos[col[k]]|, W = sum w
 * (inverse-distance weights).  grad may be NULL.  wsum [R] receives W (the adjoint needs it).  The adjoint takes the
 * transposed incidence: for source s the forward rows trow[s] .. trow[s+1]-1 of tidx that read it; ggrad may be NULL. */
int gfv_interp2_fwd(const float* phi, const float* grad, const float* srcpos, const float* tgtpos, const int32_t* rowptr,
                    const int32_t* col, int32_t mode, float* out, float* wsum, int32_t R, int32_t C, void* stream);
int gfv_interp2_bwd(const float* gout, const float* wsum, const float* srcpos, const float* tgtpos, const int32_t* trow,
                    const int32_t* tidx, int32_t mode, float* gphi, float* ggrad, int32_t S, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optional per-launch HIP-event timing of the main kernels on their own stream (bench.py roofline leg).
 * kind: 1 row-tile chain, 2 weight gradient, 3 segmented reduce.  out = {launches, total ms, algorithmic flops,
 * algorithmic bytes} summed since the last reset.  Off by default.
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_profile_enable(int on);
int gfv_profile_collect(int kind, double* out);
int gfv_profile_reset(void);
/* kinds: 1 row-tile chain (LDS form), 2 weight gradient, 3 segmented reduce, 4 Transolver slice kernels, 5 finite-volume
 * kernels, 6 input preparation / loss / Adam, 7-10 register-resident chain instantiations, 11 deterministic second-stage
 * reductions, 12 per-step weight images and transposed copies, 13 chain launches with segmented-sum segments, 14 / 15 the
 * column-owner backward (fused weight gradients) / forward family, 16 single-layer launches on the lean kernel.  Sizes an entry point cannot see from its arguments
 * (directed stencil entries S, (cell, face) incidences Sigma) are given here so the finite-volume kernels can be priced. */
int gfv_profile_set_sizes(double stencil_entries, double incidences);

/* ------------------------------------------------------------------------------------------------------------
 * Mesh plan handle: every index table the hot-path kernels walk, built on the device from the
 * reference's own int64 index tensors of one (batched) mesh - what gfv/plan.py builds with torch ops, for callers that
 * have no Python.  The handle owns its tables (device memory) until gfv_plan_destroy; gfv_plan_create synchronises the
 * stream once (it validates every index against its range: GFV_ERR_ARG when one is outside) - it is per-batch set-up,
 * not part of the per-step path.  Stable sorts: inside every CSR row the entries keep the order of the reference's
 * index tensors (the summation order of torch_scatter / index_add).
 *
 * desc (all pointers device memory, int64 as the reference stores them):
 *   edge_index   [2, n_faces]          graph_node.edge_index = face|face_node            (blocks.py:24-31)
 *   cells_node / cells_face / cells_index [n_incidences]  graph_node.face / graph_edge.face / graph_cell.face
 *                                                                                         (FVscheme.py:60-120)
 *   face_node_x  [2, n_stencil_pairs], support_edge [2, n_support_pairs]                  (FVgrad.py:264-271)
 * tables (int32 unless noted; n = rows):
 *   ES, ER [E]                       senders / receivers
 *   N_ROWPTR [N+1], N_COL_NODE, N_COL_EDGE2 [2E], INV_DEG [N] (float)   two-way adjacency by receiving node; the other
 *                                    end of every entry, its slot in the [E, 2] edge message, 1 / max(degree, 1)
 *   S_ROWPTR [N+1], S_COL [E]; R_ROWPTR [N+1], R_COL [E]                edges by sender / by receiver
 *   X_ROWPTR [N+1], X_OUT [S], X_ORDER [S]     directed stencil [fx, fx.flip(0), support] by RECEIVING node: sending
 *                                    node of every entry and its position in the directed list (the caller permutes the
 *                                    moment vectors B by it);  XO_ROWPTR, XO_IN, XO_ORDER: the same by SENDING node
 *   CROW [C+1], K_ORDER, KFACE, KNODE, KCELL [Sigma]   incidences by cell; FROW [E+1], FK [Sigma] by face (entries are
 *                                    positions in the by-cell order); NROW [N+1], NCELL [Sigma] by node
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct gfv_plan gfv_plan_t;
typedef struct {
  int64_t n_nodes, n_faces, n_cells, n_incidences, n_stencil_pairs, n_support_pairs;
  const int64_t* edge_index;
  const int64_t* cells_node;
  const int64_t* cells_face;
  const int64_t* cells_index;
  const int64_t* face_node_x;
  const int64_t* support_edge;
} gfv_plan_desc_t;
enum {
  GFV_PLAN_ES = 0, GFV_PLAN_ER, GFV_PLAN_N_ROWPTR, GFV_PLAN_N_COL_NODE, GFV_PLAN_N_COL_EDGE2, GFV_PLAN_INV_DEG,
  GFV_PLAN_S_ROWPTR, GFV_PLAN_S_COL, GFV_PLAN_R_ROWPTR, GFV_PLAN_R_COL,
  GFV_PLAN_X_ROWPTR, GFV_PLAN_X_OUT, GFV_PLAN_X_ORDER, GFV_PLAN_XO_ROWPTR, GFV_PLAN_XO_IN, GFV_PLAN_XO_ORDER,
  GFV_PLAN_CROW, GFV_PLAN_K_ORDER, GFV_PLAN_KFACE, GFV_PLAN_KNODE, GFV_PLAN_KCELL, GFV_PLAN_FROW, GFV_PLAN_FK,
  GFV_PLAN_NROW, GFV_PLAN_NCELL, GFV_PLAN_TABLE_COUNT
};
int gfv_plan_create(const gfv_plan_desc_t* desc, gfv_plan_t** plan, void* stream);
int gfv_plan_destroy(gfv_plan_t* plan);                 /* NULL is accepted */
/* device pointer and element count of one table (valid until gfv_plan_destroy) */
int gfv_plan_table(const gfv_plan_t* plan, int32_t which, const void** ptr, int64_t* count);
/* sizes5 = {N, E, C, Sigma, S = 2 n_stencil_pairs + n_support_pairs} */
int gfv_plan_sizes(const gfv_plan_t* plan, int64_t* sizes5);

/* ------------------------------------------------------------------------------------------------------------
 * The row-local Linear chains of a Transolver block, one launch each (round 5; split-fp16 product forms only: GFV_ERR_ARG under
 * gfv_set_f16split(0), where the caller issues the three gfv_rowtile_chain launches instead).  All arrays row-major fp32 with
 * row stride = width, 16-byte aligned; img_* = split-fp16 images (gfv_weight_images, built with `wmax`) of the named weights.
 *   forward:   fx1 = out_x W_out^T + b_out + fx_in;   z = LayerNorm(fx1; gamma, beta) W_pre^T + b_pre   [M,256];
 *              out = gelu(z) W_post^T + b_post + fx1          (GraphTransolver.py:93-95,163-169)
 *   backward:  g = g_out (+ g_add) (-> g_sum);  g_z = (g W_post) * gelu'(z);  g_fx1 = LayerNorm-backward(g_z W_pre; fx1, gamma) + g;
 *              g_out_x = g_fx1 W_out;  ln_partial[tile, 0:128 | 128:256] = per-tile (dgamma, dbeta) sums - gfv_trans_mlp_ln_rows(M)
 *              rows (one per 64 rows; one per 32 where the small-tile form of csrc/ctrans.hip takes the launch: provide
 *              gfv_rowtile_ln_rows(M)); sum them with gfv_reduce_multi;  gscale[row / 16] = the 16-row-group scales of g
 *              (gfv_dw_tile_t.gscale).
 *   The backward takes the images of the TRANSPOSED weights: img_post_t of W_post^T [256,128], img_pre_t of W_pre^T [128,256],
 *   img_out_t of W_out^T [128,128].  The weight gradients stay with gfv_dw_multi (they read g / z, g_z / fx1, g_fx1 / out_x).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
  const float* x;        /* out_x [M,128] */
  const float* res;      /* fx_in [M,128] */
  const void* img_out;   /* to_out.0.weight [128,128] */
  const void* img_pre;   /* mlp.linear_pre.0.weight [256,128] */
  const void* img_post;  /* mlp.linear_post.weight [128,256] */
  const float* b_out;    /* [128] or NULL */
  const float* b_pre;    /* [256] or NULL */
  const float* b_post;   /* [128] or NULL */
  const float* gamma;    /* ln_2 [128] */
  const float* beta;
  const float* wmax;
  float* fx1;            /* [M,128] out (saved for the backward) */
  float* z;              /* [M,256] out (saved) */
  float* out;            /* [M,128] */
  int32_t M, reserved;
} gfv_trans_mlp_t;
typedef struct {
  const float* g;        /* [M,128] */
  const float* g_add;    /* optional [M,128] */
  float* g_sum;          /* optional [M,128] */
  const float* z;        /* [M,256] */
  const float* fx1;      /* [M,128] */
  const void* img_post_t;
  const void* img_pre_t;
  const void* img_out_t;
  const float* gamma;
  const float* wmax;
  float* g_z;            /* [M,256] out */
  float* g_fx1;          /* [M,128] out */
  float* g_out_x;        /* [M,128] out */
  float* ln_partial;     /* optional [gfv_rowtile_ln_rows(M), 256] (PROVIDE one row per 32 rows; gfv_trans_mlp_ln_rows(M) of them are filled) */
  float* gscale;         /* optional [ceil(M / 128) * 8] */
  int32_t M, reserved;
} gfv_trans_mlp_bwd_t;
int gfv_trans_mlp_fwd(const gfv_trans_mlp_t* args, void* stream);
int gfv_trans_mlp_bwd(const gfv_trans_mlp_bwd_t* args, void* stream);
int gfv_trans_mlp_ln_rows(int32_t M);   /* rows of ln_partial a gfv_trans_mlp_bwd launch over M rows fills (the calling process's switches) */

/* ------------------------------------------------------------------------------------------------------------
 * Native command list (round 5).  Between gfv_record_begin and gfv_record_end every kernel launch the CALLING THREAD issues
 * through this library is executed as usual AND noted - kernel, grid, block, dynamic LDS, stream, arguments by value.
 * gfv_record_replay(handle, first, last) issues the noted launches [first, last) again, on the streams they were recorded on,
 * with no host work but the launches themselves.  What makes a replay valid is the caller's business: the device buffers the
 * recorded arguments point to must still be the step's buffers (gfv/cmdlist.py records inside a private torch memory pool), and
 * host-side decisions (shapes, kernel families, product form) are frozen at record time.  gfv_stream_wait(waiter, waited) is
 * the stream-to-stream edge of such a list: `waiter` waits for everything submitted to `waited` so far (recorded like a launch).
 * gfv_record_count: launches noted so far by the calling thread's open recording (-1: none open).  Replaces, for the replayed
 * step, the per-launch Python / ctypes / argument-check path (the reference pays the same per-op host cost in PyTorch eager:
 * pre_train_Adam.py:158-191).
 * ---------------------------------------------------------------------------------------------------------- */
int gfv_record_begin(void);
int gfv_record_count(void);
int64_t gfv_record_end(void);                 /* -> handle (> 0), 0 if no recording was open */
int gfv_record_length(int64_t handle);
int gfv_record_replay(int64_t handle, int32_t first, int32_t last);
int gfv_record_free(int64_t handle);
int gfv_stream_wait(void* waiter_stream, void* waited_stream);
/* Reorders a recorded list (round 6): every run of commands of other streams than `main_stream` moves behind up to k of the main
 * stream's launches that follow it - never across a command in which the main stream waits, and by at most a third of the distance
 * to it.  A later position only adds ordering (the run's leading wait then covers more of the main stream).  For steps whose main
 * stream runs dry while the host issues a side-stream burst (small meshes).  Returns the number of runs moved, or GFV_ERR_ARG. */
int gfv_record_delay_side(int64_t handle, void* main_stream, int32_t k);

#ifdef __cplusplus
}
#endif
#endif /* GFV_H_ */
