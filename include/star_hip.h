/* star_hip.h -- C ABI of libstar_hip.so, the gfx950 (MI355X) kernels behind the STAR completion path.
 *
 * The reference (coperception/star) is pure Python/PyTorch and has no FFI; what each entry point
 * REPLACES is therefore a PyTorch op sequence in the reference, cited per function as
 * file:line under /root/reference.  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ ;
 *   - the library never allocates or frees device memory and keeps no pointer after return:
 *     inputs, outputs and workspaces are caller-allocated (PyTorch tensors in the host package);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and nothing synchronises;
 *   - return value: 0 = ok, negative = error; star_last_error() returns a thread-local message;
 *   - no exception crosses the ABI; entry points are re-entrant (no global mutable state);
 *   - `dt` selects the activation storage type: STAR_BF16 (bf16 storage, fp32 accumulation),
 *     STAR_F32 (fp32 storage and fp32 MFMA, the reference's numerics) or, for the forward kernels,
 *     STAR_BF16X3 (bf16 hi/lo pairs, three bf16 MFMA products per multiply: float32-grade results at bf16-pipe speed).
 */
#ifndef STAR_HIP_H
#define STAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAR_BF16 0
#define STAR_F32 1
#define STAR_BF16X3 2   /* inference-grade exactness on the bf16 matrix pipe: every activation and weight is a PAIR of bf16 values
                           hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 significant bits) stored side by side -- a row of a [R, C]
                           matrix is [hi (C) | lo (C)], 2 C bf16 values -- and every product runs as hi*hi + lo*hi + hi*lo with
                           float32 accumulation.  Accepted by star_layernorm_fwd, star_gemm and star_attn_fwd (forward only). */

int star_version(void);
const char* star_last_error(void);

/* ---- K10: inter-agent pose warp + union fusion -------------------------------------------------
 * Replaces late_fusion / ego_late_fusion (star/models/multiagent_mae.py:870-944), which loops in
 * Python over scenes/agents calling feature_transformation (star/models/mae_base.py:589-614: two
 * affine_grid + two grid_sample(mode='nearest')) and fusion('union') (mae_base.py:624-633).
 *   pred      uint8 [A*B, C, H, W] agent-major (row a*B+b), the per-agent 0/1 occupancy decisions
 *   ego       optional float32 maps used for the ego slot instead of pred (inference: the agent's own
 *             input); element strides (in floats) ego_sn, ego_sc, ego_sh, ego_sw; NULL in training
 *   theta     float32 [B, A, A, 8]: for (b, ego i, neighbour j) the six rotation-theta entries that
 *             matter (r00 r01 r10 r11) followed by (x_trans, y_trans) and two pads, formed on the host
 *             in the source dtype of trans_matrices and then cast (mae_base.py:595-606)
 *   num_agent int32/int64 [B, A] on the device (only [b,0] is read, as multiagent_mae.py:887)
 *   out       float32 [A*B, C, H, W]
 * Bit-exact with the reference CPU path (half-to-even rounding, zero padding, k-ordered fma chain). */
int star_warp_fuse(const uint8_t* pred, const float* ego, int64_t ego_sn, int64_t ego_sc, int64_t ego_sh,
                   int64_t ego_sw, const float* theta, const void* num_agent, int num_agent_is64,
                   float* out, void* bits_ws, int B, int A, int C, int H, int W, void* stream);
/* Optional fast path for {0,1} maps (the arg-max decisions) with C <= 16: pass a workspace of star_warp_fuse_ws()
 * bytes as bits_ws; the channels of a pixel are packed into one uint16 and gathered together.  Same result, bit for
 * bit, as bits_ws = NULL (which accepts any uint8 values). */
int64_t star_warp_fuse_ws(int B, int A, int H, int W);

/* ---- K5: mask index build ------------------------------------------------------------------------
 * Replaces ids_shuffle = argsort(noise); ids_restore = argsort(ids_shuffle)
 * (multiagent_mae.py:811-814, mae_base.py:417-421).  noise float32 [R, L] (L <= 4096); stable.
 * Outputs int64 [R, L] each (either may be NULL). */
int star_mask_sort(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, int R, int L, void* stream);

/* Row maps for the sub-sampling gather / temporal un-mixing, built on the device from the ids.
 * Replaces the gather/cat/slice index arithmetic of amortized_complement_masking/_unmasking
 * (multiagent_mae.py:816-867) and random_masking / amortized_random_unmasking
 * (mae_base.py:423-431, multiagent_mae.py:761-796).
 *  complement: keep = ceil(L/T).
 *    src_row [BA*T*keep] : row of the [BA*T*L] token matrix that feeds compressed row r, or -1 (pad)
 *    dst_row [BA*T*keep] : row of the [BA*L] decoder input that compressed row r lands on, or -1
 *                          (pad; with `single`, rows of timestamps > 0 are dropped too)
 *    mask    float32 [BA, T, L] (0 = kept) or NULL
 *    fill    int32 [BA*L] : 1 where the decoder input row is NOT covered (gets mask_token), else 0
 *  random: keep = int(L*(1-ratio)); ids_* are [BA*T, L].
 *    dst_row picks, for every token, the first timestamp that kept it (multiagent_mae.py:791-796). */
int star_build_rowmaps(const int64_t* ids_shuffle, const int64_t* ids_restore, int32_t* src_row,
                       int32_t* dst_row, float* mask, int32_t* fill, int BA, int T, int L, int keep,
                       int random_mode, int single, void* stream);

/* ---- K2: LayerNorm ---------------------------------------------------------------------------------
 * Replaces nn.LayerNorm(eps=1e-6) inside timm Block and self.norm / self.decoder_norm
 * (mae_base.py:279-282,308-312; multiagent_mae.py:976-978,1044-1046).
 * x float32 [M, D] (the residual stream), y [M, D] in `dt`, mean/rstd float32 [M]. */
int star_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int M, int D, float eps, int dt, void* stream);
/* dx = LN'(dy); g_out = (g_in ? g_in : 0) + dx in float32, plus an optional copy g_out_lp in `dt`.
 * dgamma/dbeta are written (not accumulated); ws float32 [star_layernorm_bwd_ws(D)]. */
int64_t star_layernorm_bwd_ws(int D);
int star_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* g_in, float* g_out, void* g_out_lp, float* dgamma,
                       float* dbeta, float* ws, int M, int D, int dt, void* stream);
/* Deferred parameter gradients: star_layernorm_bwd with dgamma = dbeta = NULL leaves its partial rows in `ws`
 * (star_layernorm_fold_rows(M) rows of 2*D floats); star_layernorm_fold_batch folds the rows of many such calls in ONE launch
 * (bit-identical to the immediate fold).  `jobs` is a device array.  Same op as above (autograd backward of nn.LayerNorm): a
 * training step's 22 LayerNorm parameter gradients feed nothing in the backward chain, and 22 five-microsecond launches leave it. */
typedef struct StarLnFoldJob {
    const float* ws;
    float* dgamma;
    float* dbeta;
    int32_t nb;      /* star_layernorm_fold_rows(M) of the call that filled ws */
    int32_t D;
} StarLnFoldJob;
/* max_D must be the LARGEST D over the jobs (it sizes the grid: a smaller value leaves columns of the wider jobs unwritten);
 * a job's dgamma or dbeta may be NULL (that half is skipped). */
int star_layernorm_fold_rows(int M);
int star_layernorm_fold_batch(const StarLnFoldJob* jobs, int njobs, int max_D, void* stream);

/* Allocation-free variant of star_layernorm_bwd.  It has NO counterpart in the reference: it exists for callers that
 * cannot allocate device memory or bring a workspace (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  Same kernels and results, bit for bit; the partial dgamma/dbeta rows live in a 6 MiB
 * file-scope device array of the library (one copy per GPU, held until the process ends) instead of `ws`.
 * Limit: D <= 1024 (a larger D is rejected with an error; use star_layernorm_bwd).  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race.  The FIRST call on each device looks the array's
 * address up (hipGetSymbolAddress, which may load the code object) and is therefore not capture-safe: make one call per device
 * outside stream capture before using it under a graph. */
int star_layernorm_bwd_static(const void* dy, const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* g_in, float* g_out, void* g_out_lp, float* dgamma,
                              float* dbeta, int M, int D, int dt, void* stream);

/* ---- K3: Linear layers -------------------------------------------------------------------------------
 * Replaces every nn.Linear on the path (timm Attention.qkv/proj, Mlp.fc1/fc2, compressor,
 * decompressor, decoder_embed, decoder_pred_occ/free; mae_base.py:279-317) and, with the patch
 * gather of star_patchify, the Conv2d patch embedding (mae_base.py:269).
 * C[m, n] = epilogue( sum_k A[arow(m), k] * W[n, k] ), A [*, K] and W [N, K] in `dt`, K % 8 == 0. */
typedef struct StarGemm {
    const void* A;          /* [rows, K] dt */
    const void* W;          /* [N, K] dt */
    const int32_t* a_rows;  /* optional [M]: A row per output row, -1 = zero row */
    const int32_t* c_rows;  /* optional [M]: destination row per output row, -1 = drop */
    const float* bias;      /* optional [N] */
    const float* res;       /* optional float32 [res_mod or M, N]: added after the activation */
    int32_t res_mod;        /* >0: residual row = destination row % res_mod */
    const float* res2;      /* optional float32 [res2_div-indexed rows, N]: row = (m / res2_div) % res2_mod */
    int32_t res2_div, res2_mod;
    const void* aux;        /* optional dt [M, N] used by ACT_DGELU / ACT_DRELU (pre-activation / output) / ACT_MULAUX */
    int32_t act;            /* STAR_ACT_* */
    void* pre;              /* optional dt [M, N]: value before the activation (or its derivative: PRE_IS_GRAD) */
    void* out_lp;           /* optional dt [M(dest), N] */
    float* out_f32;         /* optional float32 [M(dest), N] */
    int32_t M, N, K;
    int32_t dt;
    int32_t flags;          /* STAR_GEMM_* */
    int32_t conv;           /* 0, or STAR_GEMM_CONV3X3(log2w, log2h): a 3 x 3 convolution (padding 1) over channels-last images as ONE
                               product -- A [M, K / 9] holds the pixels of H x W images (powers of two, M a multiple of H W), W is
                               [N, 9 (K / 9)] with tap t = 3 (dy + 1) + (dx + 1) in columns t K/9 .., and k-block t of output row m reads
                               the A row of pixel (y + dy, x + dx) of the same image, zeros outside.  bf16, K / 9 a multiple of 32,
                               bias-only epilogue, no a_rows (nn.Conv2d of the ConvPred heads, mae_base.py:16-105; the data gradient is
                               the same product over the output gradient with the taps reversed and the weights transposed) */
} StarGemm;
#define STAR_GEMM_CONV3X3(log2w, log2h) ((int32_t)(0x40000000 | (log2w) | ((log2h) << 5)))
#define STAR_GEMM_RES_BY_AROW 1 /* residual row = A row % res_mod, and rows with a_rows[m] < 0 are written as zeros
                                   (encode_partial: the patch embedding of a zero-padded slot is a zero row) */
#define STAR_GEMM_PRE_IS_GRAD 2 /* with ACT_GELU / ACT_RELU: `pre` receives act'(pre-activation), the factor the backward
                                   pass multiplies by (ACT_MULAUX), instead of the pre-activation itself */
#define STAR_GEMM_A_PLAIN 4     /* dt = STAR_BF16X3 only: A is a plain bf16 matrix [*, K] whose values are exact (the 0/1 occupancy
                                   patches): two products A * W_hi + A * W_lo instead of three */
#define STAR_ACT_NONE 0
#define STAR_ACT_GELU 1   /* exact erf GELU (timm Mlp act_layer=nn.GELU) */
#define STAR_ACT_RELU 2   /* compressor ReLU (mae_base.py:284-287) */
#define STAR_ACT_DGELU 3  /* backward: multiply by gelu'(aux) */
#define STAR_ACT_DRELU 4  /* backward: multiply by (aux > 0) */
#define STAR_ACT_MULAUX 5 /* backward: multiply by aux (a derivative stored by the forward, see STAR_GEMM_PRE_IS_GRAD) */
int star_gemm(const StarGemm* g, void* stream);

/* Weight gradient: dW[n, k] = sum_m dY[m, n] * X[xrow(m), k]  (float32, written), optional
 * db[n] = sum_m dY[m, n].  dY [M, N], X [*, K] in `dt`.  ws float32 [star_gemm_wgrad_ws(M,N,K)]. */
int64_t star_gemm_wgrad_ws(int M, int N, int K);
int star_gemm_wgrad(const void* dY, const void* X, const int32_t* x_rows, float* dW, float* db, float* ws,
                    int M, int N, int K, int dt, void* stream);
/* Several weight gradients in ONE launch (the four Linears of a timm Block, mae_base.py:279-281,308-310, whose autograd
 * backward the reference runs as four separate addmm's): job j computes dW_j [N_j, K_j] = dY_j^T X_j and, when db_j is given,
 * db_j = column sums of dY_j, exactly as star_gemm_wgrad does for one product (bf16 operands, N and K multiples of 8, no row
 * gather, 16-byte aligned pointers, at most 2 GiB per operand).  A job with `tap` != 0 is one tap of a 3 x 3 convolution's weight
 * gradient over channels-last images (mae_base.py:16-105, the ConvPred heads: nn.Conv2d(k=3, padding=1) backward): its rows are the
 * pixels of H x W images (powers of two, M a multiple of H W), and row m of X is taken from the pixel (y + dy, x + dx) of the same
 * image, zeros outside -- the gather is arithmetic inside the kernel, no row table (STAR_WGRAD_TAP packs dy, dx, log2 W, log2 H).
 * Every job is cut into row splits and 128 x 128, 384 x 192 or 256 x 256 output tiles by ITS OWN shape only -- so its result has the same bits whatever it shares a launch with -- few enough that a block's
 * four products fill the chip together; the split partial tiles are folded inside the kernel in a fixed order (bitwise
 * reproducible results, no atomics on data).  Jobs of the two tile shapes go out as two launches.  ws: float32
 * [star_gemm_wgrad_group_ws(...)] (-1: the jobs are not eligible, see star_last_error); its first 4 KiB are ticket counters that
 * MUST be zero on entry and are zero again when the call has run, so one zero-initialised buffer serves every call -- calls
 * sharing a workspace must be ordered on one stream. */
#define STAR_WGRAD_GROUP_MAX 8
typedef struct StarWgradJob {
    const void* dY;   /* [M, N] bf16 */
    const void* X;    /* [M, K] bf16 */
    float* dW;        /* [N, K] float32, written */
    float* db;        /* optional [N] float32, written */
    int32_t M, N, K;
    int32_t tap;      /* 0: X rows as they lie; else STAR_WGRAD_TAP(dy, dx, log2w, log2h) */
} StarWgradJob;
#define STAR_WGRAD_TAP(dy, dx, log2w, log2h) ((int32_t)(0x40000000 | ((dy) + 1) | (((dx) + 1) << 2) | ((log2w) << 4) | ((log2h) << 9)))
int64_t star_gemm_wgrad_group_ws(const StarWgradJob* jobs, int njobs, int dt);
int star_gemm_wgrad_group(const StarWgradJob* jobs, int njobs, float* ws, int64_t ws_floats, int dt, void* stream);

/* ---- K3 + K2: the Linear that closes a residual branch and the LayerNorm that reads its result, one launch -------------
 * Replaces, in timm 0.3.2 Block.forward (ctor mae_base.py:279-281,308-310), `x = x + proj(...)` / `x = x + fc2(...)` together
 * with the LayerNorm applied to that x next (norm2, the next block's norm1, or self.norm / self.decoder_norm,
 * mae_base.py:282,312):  x_out[M, N] (float32) = res + A W^T + bias;  h[M, N] (bf16) = LayerNorm_eps(x_out) * gamma + beta;
 * mean / rstd [M] optional.  A [M, K], W [N, K] bf16; N = 256 or 384 (a tile spans whole rows), K % 32 == 0.  x_out may be res.
 * star_gemm_res_ln_supported(N, K, dt) != 0 tells whether a kernel exists; other shapes -> error (callers launch star_gemm and
 * star_layernorm_fwd instead).  A tile spans whole rows, so small M leaves CUs idle: star_gemm_res_ln_prefers says whether the one
 * launch is expected to be the faster path at this M (measured on MI355X). */
int star_gemm_res_ln_supported(int N, int K, int dt);
int star_gemm_res_ln_prefers(int M, int N, int K, int dt);   /* supported AND expected to beat the two launches at this M */
int star_gemm_res_ln(const void* A, const void* W, const float* bias, const float* res, float* x_out, const float* gamma,
                     const float* beta, float eps, void* h, float* mean, float* rstd, int M, int N, int K, int dt,
                     void* stream);

/* ---- K4: attention -----------------------------------------------------------------------------------
 * Replaces timm Attention.forward between qkv and proj: softmax(q k^T hd^-0.5) v, no mask, no
 * dropout.  qkv [NB, L, 3, H, hd] in `dt` (the qkv Linear output as it lies in memory), out
 * [NB, L, H*hd] in `dt`, lse float32 [NB, H, L] (log2-domain row log-sum-exp, kept for backward).
 * hd in {16, 32, 64}; any L >= 1. */
int star_attn_fwd(const void* qkv, void* out, float* lse, int NB, int L, int H, int hd, int dt, void* stream);
/* dqkv [NB, L, 3, H, hd] from dout [NB, L, H*hd]; delta float32 [2, NB, H, L] is workspace. */
int star_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                  void* dqkv, int NB, int L, int H, int hd, int dt, void* stream);

/* ---- K1: voxel -> patch gather ------------------------------------------------------------------------
 * Replaces the im2col half of Conv2d(13, D, k=p, s=p) + flatten(2).transpose(1,2) (timm PatchEmbed;
 * mae_base.py:269, multiagent_mae.py:960-962).  Frame f = ba*T + t is imgs1[ba] for t = 0 and
 * imgs_next[ba, t-1] otherwise; both are float32 with element strides (the reference hands over
 * NHWC memory viewed as NCHW, CoModule.py:93-96).  out [BA*T*(H/p)*(W/p), C*p*p] in `dt`, patch vector
 * order (c, py, px) = the Conv2d weight's own flattening. */
int star_patchify(const float* img1, const int64_t* s1 /*[4] n,c,h,w*/, const float* imgn,
                  const int64_t* sn /*[5] n,t,c,h,w*/, void* out, int BA, int T, int C, int H, int W, int p,
                  int dt, void* stream);

/* ---- K9: occupancy loss + decision ----------------------------------------------------------------------
 * Replaces CrossEntropyLoss over the class dim + argmax(softmax) (multiagent_mae.py:1076-1079,1099)
 * and the unpatchify/stack in front of them (mae_base.py:392-406, multiagent_mae.py:1051-1057).
 * logits float32 [BA*L, 2*V] in patch layout, V = C*p*p, column = cls*V + (c*p + py)*p + px
 * (cls 0 = free, 1 = occupied; the host permutes the head weights' rows into this order).
 * target float32 image with element strides.  Outputs (each optional):
 *   ind64 int64 [BA,C,H,W], ind8 uint8 [BA,C,H,W], pred float32 [BA,2,C,H,W] (the reference's layout),
 *   dlogits dt [BA*L, 2*V] = (softmax - onehot) / (BA*C*H*W),
 *   loss_partial float32 [star_ce_nblocks(...)] per-block sums; star_ce_finish folds them into
 *   loss[0] = mean CE in fixed order. */
int star_ce_nblocks(int BA, int C, int H, int W);
int star_ce_argmax(const float* logits, const float* target, const int64_t* st /*[4]*/, int64_t* ind64,
                   uint8_t* ind8, float* pred, void* dlogits, float* loss_partial, int BA, int C, int H,
                   int W, int p, int dt, void* stream);
int star_ce_finish(const float* loss_partial, int nblocks, int64_t count, float* loss, void* stream);

/* ---- K7 helper: rows the temporal un-mixing does not cover get mask_token (+temp embed of t=0) + pos.
 * x float32 [BA*L, D]; fill int32 [BA*L] from star_build_rowmaps. */
int star_fill_rows(float* x, const int32_t* fill, const float* mask_token, const float* temp0,
                   const float* pos, int BA, int L, int D, void* stream);
/* Q2 (multiagent_mae.py:797 -> :1037): the random-mode decoder input is the [BA, L, D] token matrix
 * transposed to [BA, D, L] and re-read as [BA, L, D]; then + pos.  src/dst float32. */
int star_transpose_rows(const float* src, float* dst, const float* pos, int BA, int L, int D, void* stream);
int star_transpose_rows_bwd(const float* gdst, float* gsrc, int BA, int L, int D, void* stream);

/* Row gather / scatter of float32 matrices (the residual stream and its gradient).
 *   gather : dst[m, :] = rows[m] >= 0 ? src[rows[m], :] : 0, written as float32 and/or `dt`
 *   scatter: dst[rows[m], :] = src[m, :] for rows[m] >= 0 (dst must be zeroed by the caller when the
 *            map is not onto); both for the gather/cat/slice steps of multiagent_mae.py:816-867. */
int star_gather_rows(const float* src, const int32_t* rows, float* dst_f32, void* dst_lp, int M, int D, int dt, void* stream);
int star_scatter_rows(const void* src, int src_dt, const int32_t* rows, float* dst, int M, int D, void* stream);
/* Grouped column sums: out[g, :] = sum over rows r with (r / div) % mod == g and (sel == NULL or
 * sel[r] matches want) of src[r, :].  src in `dt` (STAR_F32 reads float32).  Used for the gradients of
 * decoder_temp_embed, mask_token and the patch-embedding bias.  D % 4 == 0; ws float32 [64 * mod * D]. */
int star_colsum_groups(const void* src, int dt, const int32_t* sel, int sel_nonneg, int R, int D, int div, int mod, float* ws,
                       float* out, void* stream);
/* dst[c, r] = cast(src[r, c]): the transposed low-precision copy of a weight used by the dX GEMMs. */
int star_cast_transpose(const float* src, void* dst, int R, int C, int dt, void* stream);

/* ---- parameter plumbing -----------------------------------------------------------------------------------
 * cast float32 -> dt for many tensors in one launch is done by star_cast; star_colsum gives bias grads. */
int star_cast(const float* src, void* dst, int64_t n, int dt, void* stream);
int star_cast_back(const void* src, float* dst, int64_t n, int dt, void* stream);
/* The per-step refresh of every low-precision weight copy in one launch.  jobs_dev: device array, sorted by
 * tile_start (running count of 64x64 tiles, first = 0); total_tiles = sum over jobs of ceil(rows/64)*ceil(cols/64). */
typedef struct StarCastJob {
    const float* src;        /* float32 [*, cols]; row r of the job is src row (src_rows ? src_rows[r] : r) */
    void* dst;               /* dt (float32 when dst_f32 != 0): element (r, c) goes to dst[dst_off + r * dst_ld + c], or to
                                dst[dst_off + c * dst_ld + r] when transpose != 0 */
    int32_t rows, cols;
    int32_t transpose;
    int32_t tile_start;
    const int32_t* src_rows; /* optional [rows]: row gather (the fused prediction-head weight takes the rows of two Linear
                                weights in class-major order) */
    int32_t dst_ld;          /* leading dimension of dst; 0 = dense (cols, or rows when transposed) */
    int32_t dst_off;         /* element offset into dst: several jobs can fill one destination */
    int32_t dst_f32;         /* != 0: this job's destination is float32 whatever dt says (bias vectors) */
    int32_t split;           /* != 0 (bf16, not transposed): write bf16 PAIRS for STAR_BF16X3 -- hi at (r, c), lo = bf16(v - hi) at
                                (r, cols + c); dst_ld must be >= 2 * cols */
} StarCastJob;
int star_cast_batch(const StarCastJob* jobs_dev, int njobs, int total_tiles, int dt, void* stream);
/* AdamW (decoupled weight decay) over one flat float32 buffer, torch.optim.AdamW semantics
 * (train_completion.py:125): p -= lr*wd*p; m,v updates; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
 * wd_mask uint8 per element (1 = decay) or NULL; gscale optional device scalar multiplied into g. */
int star_adamw(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, const float* gscale,
               int64_t n, float lr, float beta1, float beta2, float eps, float wd, int step, void* stream);
/* sum of squares of a flat float32 buffer -> out[0] (deterministic two-stage); ws float32 [1024]. */
int star_sumsq(const float* x, int64_t n, float* ws, float* out, void* stream);

/* ---- K11: occupancy IoU confusion counts ----------------------------------------------------------------
 * Replaces Metrics.add_batch (star/utils/metrics.py:146-190): threshold >= 0.5, MaxPool3d((2,2,2),
 * stride 2, padding (1,0,0)) twice, 2x2 confusion per scale.  pred/target float32 [N, Z, H, W]
 * contiguous; conf int64 [3, 2, 2] (rows = prediction) is ADDED to (zero it first). */
int star_iou_confusion(const float* pred, const float* target, unsigned long long* conf, int N, int Z, int H,
                       int W, void* stream);

/* ---- in front of the path: sparse voxel indices -> dense BEV frames ------------------------------------------
 * Replaces the dataset's per-frame numpy densification (star/datasets/MultiTempDet.py:134-143, 201-209,
 * 244-247; MultiTempSeg.py likewise): zeros(dims, bool); v[idx[:,0], idx[:,1], idx[:,2]] = 1; rot90(v, 3);
 * astype(float32) -- and the dense host->device copy behind it.
 *   idx       int32 (or int64: idx_is64) [n_points, 3] = (x, y, z) of every occupied voxel, frames back to back
 *   offsets   int64 [F + 1] on the device: frame f owns points offsets[f] .. offsets[f+1]-1 (empty frames allowed)
 *   h_offsets optional HOST copy of offsets: lets the call work through the frames in chunks that stay inside the
 *             256 MiB Infinity Cache (fill, then scatter into the still-resident lines); NULL = one chunk
 *   out       float32 [F, Y, X, Z] (the reference's channel-last frame; voxel (x,y,z) -> out[y, X-1-x, z]) or,
 *             with channels_first != 0, [F, Z, Y, X]; written entirely (zero fill + 1.0f per voxel)
 *   err       optional int32 device word: set to 1 if any index is outside [-dim, dim) (numpy raises IndexError
 *             there; indices in [-dim, -1] wrap as in numpy); the caller zeroes it. */
int star_voxel_scatter(const void* idx, int idx_is64, const int64_t* offsets, const int64_t* h_offsets, int64_t n_points,
                       float* out, int F, int X, int Y, int Z, int channels_first, int32_t* err, void* stream);

/* ---- VQ-STAR quantiser (the `--com vqstar` variant of the path) ---------------------------------------------
 * Replaces STARVectorQuantizer.forward (star/models/mae_base.py:122-173) between compressor and decompressor
 * (star/models/VQSTAR.py:432,456) and its autograd.
 * star_vq_assign: x float32 [M, E] (compressor output), codebook float32 [K, E] ->
 *   idx int64 [M] = argmin_k (|x|^2 + |e_k|^2 - 2 x.e_k) in float32, first minimum;
 *   q_f32 [M, E] / q_lp [M, E] in `dt` (each optional) = codebook[idx];
 *   part float32 [star_vq_nblocks(M)] per-workgroup sums of (q - x)^2 (optional);
 *   counts int32 [K] histogram of idx (optional; zeroed by the call).  E in {8,16,32,64,128}.
 * star_vq_finish: out3 = {loss = mse + commitment*mse (ema != 0: commitment*mse alone, mae_base.py:244-245),
 *   perplexity = exp(-sum p log(p+1e-10)), mse}; loss_inout (optional) += loss, in stream order (the model's total
 *   loss, VQSTAR.py:443).
 * star_vq_bwd: dlat `dt` [M, E] = (dq + commitment * 2 (x - q)/(M E)) * [x > 0]  (straight-through estimator
 *   mae_base.py:166, then the compressor's ReLU); dcodebook float32 [K, E] = 2/(M E) * sum_{idx[m]=k} (q - x[m]),
 *   summed in a fixed order (deterministic), WRITTEN for every code; needs ws float32 [star_vq_bwd_ws(M,E,K)].
 *   Either output may be NULL.
 * EMA codebook, STARVectorQuantizerEMA.forward in training mode (star/models/mae_base.py:228-241; `--decay` > 0):
 *   star_vq_code_sums: dw float32 [K, E] = sum_{idx[m]=k} x[m, :] in a fixed order (encodings^T x); ws as star_vq_bwd;
 *   star_vq_ema_update, in place: cluster_size [K] = cluster_size*decay + (1-decay)*counts, Laplace-smoothed with
 *     epsilon; ema_w [K, E] = ema_w*decay + (1-decay)*dw; codebook = ema_w / cluster_size[:, None].  Between the two
 *     calls a data-parallel job sums counts and dw over its ranks.
 * star_onehot_rows: the `encodings` matrix the reference returns from inference (mae_base.py:154-155),
 *   float32 [M, K], K % 4 == 0. */
int star_vq_nblocks(int M);
int star_vq_assign(const float* x, const float* codebook, int64_t* idx, float* q_f32, void* q_lp, float* part,
                   int32_t* counts, int M, int E, int K, int dt, void* stream);
int star_vq_finish(const float* part, int nblocks, const int32_t* counts, int M, int E, int K, float commitment,
                   int ema, float* out3, float* loss_inout, void* stream);
int64_t star_vq_bwd_ws(int M, int E, int K);
int star_vq_bwd(const void* dq, const float* x, const int64_t* idx, const float* codebook, void* dlat,
                float* dcodebook, float* ws, int M, int E, int K, float commitment, int dt, void* stream);
int star_vq_code_sums(const float* x, const int64_t* idx, float* dw, float* ws, int M, int E, int K, void* stream);
int star_vq_ema_update(const int32_t* counts, const float* dw, float* codebook, float* ema_w, float* cluster_size,
                       int E, int K, double decay, double epsilon, void* stream);
int star_onehot_rows(const int64_t* idx, float* out, int M, int K, void* stream);

/* ---- next row f-1: the joint (latent-level fusion) variant, AmortizedFusionMMAEViT ---------------------------------
 * Everything channels-last: token rows [N*H*W, C] are the NHWC feature map; the 3x3 convolutions of the conv head run as
 * nine row-gathered star_gemm taps (a_rows = the shifted-pixel table, -1 = zero padding) accumulating through `res`.
 * star_feat_warp_fuse replaces forward_fusion's loop (star/models/multiagent_mae.py:497-551): for every scene b and ego
 *   slot i < num_agent[b,0], out = threshold(feat_i + sum_{j != i} warp_{j->i}(feat_j)) with feature_transformation's two
 *   nearest resamples at the G x G token grid (star/models/mae_base.py:589-614) and fusion("sum")'s threshold
 *   (mae_base.py:624-633: >= 0.5 -> 1, < 0.5 -> 0); slots >= num_agent keep feat.  `pos` [G*G, D] (decoder_pos_embed, added in
 *   forward_decoder :585) is added to every output row when given.  passthrough uint8 [A*B]: 1 where the slot kept its
 *   own features (the only rows a gradient passes through: the thresholded rows are constants).
 *   feat/out float32 [A*B, G*G, D], agent-major. */
int star_feat_warp_fuse(const float* feat, const float* theta, const void* num_agent, int num_agent_is64, const float* pos,
                        float* out, uint8_t* passthrough, int B, int A, int G, int D, void* stream);
/* F.interpolate(scale_factor=2, mode='bilinear') of ConvPred (mae_base.py:40,44,48), channels-last [N,H,W,C] -> [N,2H,2W,C]
 * in `dt`, and its adjoint on float32 gradients (gather form, deterministic).  C % 4 == 0. */
int star_upsample2x_fwd(const void* x, void* y, int N, int H, int W, int C, int dt, void* stream);
int star_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* nn.BatchNorm2d of ConvPred (mae_base.py:27-35) over rows [R = N*H*W, ld] (ld % 4 == 0; pad columns carry zeros):
 * star_bn_stats: per-column mean, biased variance (two passes) and rstd = 1/sqrt(var + eps); ws float32 [star_bn_ws(ld)].
 * star_bn_apply: y = (x - mean) * rstd * gamma + beta, optional ReLU (F.relu(self.bnK(...))), y in `dt`.
 * star_bn_bwd:   backward of that pair; dy float32, dx in `dt`; dgamma / dbeta written; training = batch statistics. */
int64_t star_bn_ws(int ld);
int star_bn_stats(const float* x, int64_t R, int ld, float eps, float* mean, float* var, float* rstd, float* ws, void* stream);
int star_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, int64_t R,
                  int ld, void* y, int dt, void* stream);
int star_bn_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                int relu, int training, int64_t R, int ld, void* dx, int dt, float* dgamma, float* dbeta, float* ws, void* stream);
/* CrossEntropyLoss over stack((free, occ), 1) + argmax(softmax) of the joint model (multiagent_mae.py:598-605,655-681) on
 * the two heads' channels-last outputs [N*H*W, ld] (C <= ld valid).  target: float image with element strides st[4] =
 * (n, c, h, w).  ind int64 [N,C,H,W]; pred float32 [N,2,C,H,W] (the tensor the reference returns); dfree / docc float32
 * [N*H*W, ld] = (softmax - onehot) / (N*C*H*W); loss_partial float32 [star_ce_nhwc_blocks()] -> star_ce_finish. */
int star_ce_nhwc_blocks(int N, int H, int W);
int star_ce_nhwc(const float* lfree, const float* locc, const float* target, const int64_t* st, int64_t* ind, float* pred, float* dfree,
                 float* docc, float* loss_partial, int N, int C, int H, int W, int ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif
