// K3: every Linear layer of the path as one MFMA GEMM family for gfx950.
//
//   star_gemm        C[m,n] = epilogue( sum_k A[arow(m),k] * W[n,k] )      ("NT": both K-contiguous)
//   star_gemm_wgrad  dW[n,k] = sum_m dY[m,n] * X[xrow(m),k]                ("TN": reduction over rows)
//
// Reference ops replaced: nn.Linear in timm Attention/Mlp and the compressor / decompressor /
// decoder_embed / prediction heads (star/models/mae_base.py:279-317), and their autograd backward.
//
// Design (bf16 storage, fp32 accumulate):
//   * 256 threads = 4 waves (2x2); v_mfma_f32_16x16x32_bf16; the WEIGHT tile is the MFMA A operand
//     and the activation tile the B operand, so each lane ends up with 4 consecutive output columns
//     of one row: the epilogue (bias, GELU/ReLU, residuals, row scatter) works on 16-byte vectors.
//   * BK = 64; operand tiles go global -> registers -> LDS (double-buffered LDS, one barrier per
//     k-step, next tile's global loads in flight during the MFMAs).
//   * LDS rows are 128 B; 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7): conflict-free
//     for both the ds_write_b128 fill and the ds_read_b128 fragment reads (guide section 2 banking).
//   * The TN kernel stages [m][n] / [m][k] tiles as they lie in memory and reads fragments with
//     ds_read_b64_tr_b16 (hardware transpose), slots XOR-swizzled by (row&3)|((row>>3&1)<<2).
//   * block id -> tile with an XCD-aware bijective remap: the n-tiles of one m-tile run on one XCD
//     and share its L2 copy of the activation tile.
//   * Split-M in the TN kernel writes fp32 slabs reduced in a fixed order (deterministic gradients).
// The fp32 path (STAR_F32) uses v_mfma_f32_16x16x4_f32 on the same skeleton with BK = 32.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define GT 256  // threads

#include "gelu.h"

// ------------------------------------------------------------------------------------------------------
// element traits: how many elements a 16-byte chunk holds and how a k-step is fed to the matrix core
template <typename T> struct GemmT;
template <> struct GemmT<bf16_t> {
    static constexpr int EPC = 8;    // elements per 16-byte chunk
    static constexpr int BK = 64;    // k-tile
};
template <> struct GemmT<float> {
    static constexpr int EPC = 4;
    static constexpr int BK = 32;
};

// LDS tile: rows of 128 B = 8 chunks of 16 B, chunk swizzled by ((row>>1)&7)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int TM, int TN>
__global__ __launch_bounds__(GT) void gemm_nt_kernel(const StarGemm g) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int EPC = GemmT<T>::EPC, BK = GemmT<T>::BK;
    constexpr int A_IT = BM * 8 / GT, W_IT = (BN * 8 + GT - 1) / GT;  // 16-byte chunks per thread per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = (BM + BN) * 128;  // bytes per pipeline stage: A tile then W tile

    const int M = g.M, N = g.N, K = g.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m_blk = (tile / tiles_n) * BM, n_blk = (tile % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int lr = lane & 15, lg = lane >> 4;

    const T* A = (const T*)g.A;
    const T* W = (const T*)g.W;

    // global -> register staging: thread owns chunk (tid & 7) of rows (tid >> 3) + 32*i
    const int ld_chunk = tid & 7, ld_row = tid >> 3;
    const T* a_ptr[A_IT];
    const T* w_ptr[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m_blk + ld_row + 32 * i;
        long long src = -1;
        if (m < M) src = g.a_rows ? (long long)g.a_rows[m] : (long long)m;
        a_ptr[i] = src >= 0 ? A + (size_t)src * K + ld_chunk * EPC : nullptr;
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int n = n_blk + ld_row + 32 * i;
        w_ptr[i] = (n < N && ld_row + 32 * i < BN) ? W + (size_t)n * K + ld_chunk * EPC : nullptr;
    }
    uint4 ra[A_IT], rw[W_IT];
    auto load_tile = [&](int k0) {
        const bool kin = k0 + ld_chunk * EPC < K;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            ra[i] = (a_ptr[i] && kin) ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            rw[i] = (w_ptr[i] && kin) ? *reinterpret_cast<const uint4*>(w_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(smem + buf * STAGE + lds_off(ld_row + 32 * i, ld_chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            if (ld_row + 32 * i < BN) *reinterpret_cast<uint4*>(smem + buf * STAGE + BM * 128 + lds_off(ld_row + 32 * i, ld_chunk)) = rw[i];
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const unsigned char* pa = smem + buf * STAGE;
        const unsigned char* pw = pa + BM * 128;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // two 32-deep MFMA steps per 64-deep tile
                short8_t fw[TN], fx[TM];
#pragma unroll
                for (int a = 0; a < TN; ++a)
                    fw[a] = *reinterpret_cast<const short8_t*>(pw + lds_off(wn * TN * 16 + a * 16 + lr, ks * 4 + lg));
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    fx[b] = *reinterpret_cast<const short8_t*>(pa + lds_off(wm * TM * 16 + b * 16 + lr, ks * 4 + lg));
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[a]),
                                                                            __builtin_bit_cast(bf16x8_t, fx[b]), acc[a][b], 0, 0, 0);
            }
        } else {
            // fp32: v_mfma_f32_16x16x4_f32 takes one value per lane (k = lane>>4).  A 16-byte chunk
            // holds 4 consecutive k; lane group lg reads chunk (2*ks*... ) so that the four MFMAs of a
            // chunk-quad use k = 4*lg + j on both operands (any consistent k permutation is valid).
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // 2 x 16 k per 32-deep tile
                f32x4_t fw[TN], fx[TM];
#pragma unroll
                for (int a = 0; a < TN; ++a)
                    fw[a] = *reinterpret_cast<const f32x4_t*>(pw + lds_off(wn * TN * 16 + a * 16 + lr, ks * 4 + lg));
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    fx[b] = *reinterpret_cast<const f32x4_t*>(pa + lds_off(wm * TM * 16 + b * 16 + lr, ks * 4 + lg));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < TN; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[a][j], fx[b][j], acc[a][b], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n0..n0+3], m = .. + lr, n0 = .. + lg*4
    T* out_lp = (T*)g.out_lp;
    T* pre = (T*)g.pre;
    const T* aux = (const T*)g.aux;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m_blk + wm * TM * 16 + b * 16 + lr;
        if (m >= M) continue;
        const long long d = g.c_rows ? (long long)g.c_rows[m] : (long long)m;
        if (d < 0) continue;
        const bool by_arow = (g.flags & STAR_GEMM_RES_BY_AROW) != 0;
        const long long arow = (by_arow && g.a_rows) ? (long long)g.a_rows[m] : d;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n0 = n_blk + wn * TN * 16 + a * 16 + lg * 4;
            if (n0 >= N) continue;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (by_arow && arow < 0) {  // zero-padded slot: stays a zero row
                const float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (out_lp) st4(out_lp + (size_t)d * N + n0, z);
                if (g.out_f32) st4(g.out_f32 + (size_t)d * N + n0, z);
                continue;
            }
            if (g.bias) {
                float t[4];
                ld4(g.bias + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
            const bool pre_grad = (g.flags & STAR_GEMM_PRE_IS_GRAD) != 0;
            if (pre && !pre_grad) st4(pre + (size_t)d * N + n0, v);
            if (g.act == STAR_ACT_GELU) {
                float dv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) gelu_both(v[e], v[e], dv[e]);
                if (pre && pre_grad) st4(pre + (size_t)d * N + n0, dv);
            } else if (g.act == STAR_ACT_RELU) {
                if (pre && pre_grad) {
                    const float dv[4] = {v[0] > 0.f ? 1.f : 0.f, v[1] > 0.f ? 1.f : 0.f, v[2] > 0.f ? 1.f : 0.f, v[3] > 0.f ? 1.f : 0.f};
                    st4(pre + (size_t)d * N + n0, dv);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // NaN stays NaN, as torch.relu (a max would drop it)
            } else if (g.act == STAR_ACT_DGELU) {
                float t[4];
                ld4(aux + (size_t)m * N + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(t[e]);
            } else if (g.act == STAR_ACT_MULAUX) {
                float t[4];
                ld4(aux + (size_t)m * N + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= t[e];
            } else if (g.act == STAR_ACT_DRELU) {
                float t[4];
                ld4(aux + (size_t)m * N + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = t[e] > 0.f ? v[e] : 0.f;
            }
            if (g.res) {
                const long long rb = by_arow ? arow : d;
                const long long rr = g.res_mod > 0 ? rb % g.res_mod : rb;
                float t[4];
                ld4(g.res + (size_t)rr * N + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
            if (g.res2) {
                const int rr = (m / g.res2_div) % g.res2_mod;
                float t[4];
                ld4(g.res2 + (size_t)rr * N + n0, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += t[e];
            }
            if (out_lp) st4(out_lp + (size_t)d * N + n0, v);
            if (g.out_f32) st4(g.out_f32 + (size_t)d * N + n0, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// bf16 fast path: the same 128x128 tile, but operands stream global -> LDS directly (buffer_load ... lds,
// no staging registers) through an NST-deep ring with NST-1 stages in flight behind a COUNTED vmcnt, one
// raw s_barrier per 32-deep k-step; out-of-range rows / gathered "-1" rows are zero-filled by the buffer
// descriptor's range check.  With K = 256..1536 a tile has only 8..48 k-steps, so what matters is that
// HBM/L2 latency is covered by loads issued three steps earlier (and by the second/third block resident on
// the CU), not by one-step register prefetch.  LDS rows are 64 B; 16-byte chunk c of row r sits at
// c ^ S[(r>>2)&3], S = {0,2,3,1}: conflict-free for the 16x16x32 fragment reads (ds_read_b128 lane groups).
// The swizzle is applied on the SOURCE address (the LDS side of an LDS-DMA is lane-linear).
// Epilogue: accumulators are staged through the (now idle) ring as fp32 [64][132] twice, so that bias /
// activation / residual / row-scatter run on whole rows with 16-byte coalesced global accesses.
#define NTF_BK 32
#ifndef NT_DIAG
#define NT_DIAG 0       // diagnostic builds (tools/diag_nt.sh; results WRONG, timing only): 1 no LDS-DMA in the loop, 2 no MFMAs, 4 no result stores
#endif
__device__ __forceinline__ int ntf_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// EPI specialises the epilogue at compile time (the fully general one costs ~10k instructions and spills):
#define EPI_GENERIC 0   // everything decided at run time
#define EPI_LINEAR 1    // bias only (any row gather / scatter); no activation, residual, aux, pre
#define EPI_RES 2       // bias + one float32 residual table; no activation
#define EPI_GELU_GRAD 3 // bias + GELU; `pre`, when given, receives gelu' (STAR_GEMM_PRE_IS_GRAD)
#define EPI_MULAUX 4    // times aux; no bias, no residual
// Tile geometry: WM x WN waves, each wave RB 16-row blocks (rows) x 64 columns.
//   <2,2,4>: 128 x 128, 4 waves (3 blocks per CU)   -- narrow outputs (N <= 256) and small problems
//   <4,2,4>: 256 x 128, 8 waves (2 blocks per CU)   -- N = 384: a 128-wide W panel per tile, 25 % fewer operand bytes per flop
//   <2,4,8>: 256 x 256, 8 waves (1 block per CU)    -- wide outputs: half the operand bytes per flop
//   <2,2,2>:  64 x 128, 4 waves                      -- problems with too few 128-row tiles for the chip
// What bounds the 128 x 128 tile at K = 256..384 is the rate at which a CU takes operand bytes in through LDS-DMA
// (measured: 12 waves spend 55 % of their cycles issue-stalled, MFMA pipe 24 % busy): every tile re-reads a full
// 98 KB W panel and A panel for 12.6 MFLOP.  Bigger tiles cut that ratio; the per-wave code is the same.
// Column groups (CG): which 16 output columns of its wave's 64-column block a lane owns.  Accumulator block a, register j of
// lane group lg is LDS row a*16 + lg*4 + j of the W image; the image holds W row perm(row), so that a lane's four
// 4-column runs land where a 16-byte store per lane makes every ROW's four lanes write one contiguous 64-byte segment:
//   CG = 8 (a bf16 output exists): run a -> columns (a>>1)*32 + lg*8 + (a&1)*4 : lanes lg = 0..3 x 16 B of bf16 = 32 columns
//   CG = 4 (float32 outputs only): run a -> columns a*16 + lg*4                 : lanes lg = 0..3 x 16 B of f32  = 16 columns
// (16 contiguous columns per lane -- the earlier layout -- made each store instruction write 16-byte pieces 32 or 64
// bytes apart: four times the write requests for the same bytes.)
template <int CG> __device__ __forceinline__ int ntf_run_off(int lg, int a) {
    return CG == 8 ? ((a >> 1) * 32 + lg * 8 + (a & 1) * 4) : (a * 16 + lg * 4);
}
// SPL = 1 (dt = STAR_BF16X3): operands are bf16 PAIRS hi + lo (a row of A / W is [hi (K) | lo (K)], an out_lp row [hi (N) | lo (N)])
// and every k-step runs three MFMA products hi*hi + lo*hi + hi*lo on the bf16 pipe -- float32-grade products (2^-17 per operand,
// the lo*lo term of 2^-18 dropped) at 3 MFMAs for 2x the operand bytes.  A stage then holds the hi images followed by the lo images.
// CONV = 1 (g.conv): an implicit 3 x 3 convolution -- k-block t of an output pixel reads the A row of the pixel shifted by tap t (or
// zeros outside the image), so the nine taps accumulate in the registers instead of nine launches through a float32 output.
template <int NST, int EPI, int WM, int WN, int RB, int CG, int SPL = 0, int CONV = 0>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(SPL ? 2 : (WM * WN == 4 ? 3 : (RB == 8 ? 2 : 4)))))
void gemm_nt_fast_kernel(const StarGemm g) {
    static_assert(CG == 4 || CG == 8, "column groups of 4 (float32 outputs) or 8 (bf16 outputs)");
    constexpr int NW = WM * WN, BM = WM * RB * 16, BN = WN * 64, HALF = (BM + BN) * 64, STAGE = HALF * (SPL ? 2 : 1), PF = NST - 1;
    constexpr int NA = BM / 16 / NW, NWP = BN / 16 / NW, NDMA = (NA + NWP) * (SPL ? 2 : 1);   // LDS-DMA pieces (16 rows x 64 B) per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = g.M, N = g.N, K = g.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- LDS-DMA addressing: a wave instruction fills 16 rows x 64 B = 1 KiB; this wave owns rows wave*32..+31
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.W), 0, 0x7fffffff, 0x00020000);
    unsigned offA[2], offW[2];   // NA, NWP <= 2 (fixed bounds: a dependent bound here makes hipcc drop the kernel on the host side)
    unsigned offAl[2], offWl[2]; // SPL: the lo halves (K elements further along the row; out of range for a plain-bf16 A operand)
    int colA[2], colW[2];        // SPL: element column of the lane's 16-byte chunk inside a k-step (K % 32 != 0: the last step is cut)
    static_assert(NA <= 2 && NWP <= 2, "LDS-DMA offset tables hold two pieces");
    const bool a_plain = SPL && (g.flags & STAR_GEMM_A_PLAIN) != 0;
    const int conv_kt = CONV ? K / 9 : K;                 // CONV: channels per tap
    const int conv_kpt = conv_kt / NTF_BK;                // k-steps per tap
    const int conv_lw = g.conv & 31, conv_lh = (g.conv >> 5) & 31;
    int cm[2], cpx[2], cpy[2];                            // CONV: output pixel of the lane's row (-1 past M), its x and y
    const long long lda = CONV ? conv_kt : (SPL && !a_plain ? 2LL * K : K), ldw = SPL ? 2LL * K : K;      // row pitch in elements
    auto dma_setup = [&](int m_blk, int n_blk) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = (wave * NA + j) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ntf_swz(r);
            const int m = m_blk + r;
            if constexpr (CONV != 0) {
                cm[j] = m < M ? m : -1;
                cpx[j] = m & ((1 << conv_lw) - 1);
                cpy[j] = (m >> conv_lw) & ((1 << conv_lh) - 1);
                offA[j] = c * 16;
                continue;
            }
            long long src = -1;
            if (m < M) src = g.a_rows ? (long long)g.a_rows[m] : (long long)m;
            offA[j] = src >= 0 ? (unsigned)(src * lda * 2 + c * 16) : 0x7fffffffu;
            offAl[j] = (src >= 0 && !a_plain) ? offA[j] + (unsigned)(K * 2) : 0x7fffffffu;
            colA[j] = c * 8;
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int r = (wave * NWP + j) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ntf_swz(r);
            // LDS row r = a*16 + lg*4 + j (within its 64-row block) holds W row = the output column of accumulator
            // (a, lg, j): see ntf_run_off
            const int n = n_blk + (r & ~63) + ntf_run_off<CG>((r >> 2) & 3, (r >> 4) & 3) + (r & 3);
            offW[j] = n < N ? (unsigned)((long long)n * ldw * 2 + c * 16) : 0x7fffffffu;
            offWl[j] = n < N ? offW[j] + (unsigned)(K * 2) : 0x7fffffffu;
            colW[j] = c * 8;
        }
    };
    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NST) * STAGE;
        const unsigned kb = kt * (NTF_BK * 2);
        if constexpr (CONV != 0) {
            const int t = kt / conv_kpt, kk = kt - t * conv_kpt;
            const int ty = (t * 11) >> 5;                     // t / 3 for t = 0..8
            const int dy = ty - 1, dx = t - 3 * ty - 1;
            const int shift = dy * (1 << conv_lw) + dx;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const bool ok = cm[j] >= 0 && (unsigned)(cpx[j] + dx) < (1u << conv_lw) && (unsigned)(cpy[j] + dy) < (1u << conv_lh);
                const unsigned o = ok ? (unsigned)((long long)(cm[j] + shift) * conv_kt * 2) + offA[j] + (unsigned)(kk * (NTF_BK * 2)) : 0x7fffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * NA + j) * 16 * 64), 16, o, 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * NA + j) * 16 * 64), 16,
                                                     offA[j] + kb, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + BM * 64 + (wave * NWP + j) * 16 * 64), 16,
                                                     offW[j] + kb, 0, 0, 0);
        if constexpr (SPL != 0) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + HALF + (wave * NA + j) * 16 * 64), 16,
                                                         offAl[j] + kb, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NWP; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + HALF + BM * 64 + (wave * NWP + j) * 16 * 64), 16,
                                                         offWl[j] + kb, 0, 0, 0);
        }
    };
    // SPL with K % 32 != 0 (K % 8 == 0): the last k-step's chunks past column K read as zeros (an out-of-range offset) instead of
    // running into the neighbouring half of the row
    auto issue_tail = [&](int kt) {
        unsigned char* st = smem + (kt % NST) * STAGE;
        const unsigned kb = kt * (NTF_BK * 2);
        const int k0 = kt * NTF_BK;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * NA + j) * 16 * 64), 16,
                                                     k0 + colA[j] < K ? offA[j] + kb : 0x7fffffffu, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + BM * 64 + (wave * NWP + j) * 16 * 64), 16,
                                                     k0 + colW[j] < K ? offW[j] + kb : 0x7fffffffu, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + HALF + (wave * NA + j) * 16 * 64), 16,
                                                     k0 + colA[j] < K ? offAl[j] + kb : 0x7fffffffu, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + HALF + BM * 64 + (wave * NWP + j) * 16 * 64), 16,
                                                     k0 + colW[j] < K ? offWl[j] + kb : 0x7fffffffu, 0, 0, 0);
    };
    auto issue_any = [&](int kt, int nk_) {
        if (SPL != 0 && kt == nk_ - 1 && (K % NTF_BK) != 0) issue_tail(kt); else issue(kt);
    };
    // fragment addresses inside a stage (constant over k-steps and tiles)
    int fw[4], fx[RB];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int rw_ = wn * 64 + a * 16 + lr;
        fw[a] = BM * 64 + rw_ * 64 + ((lg ^ ntf_swz(rw_)) << 4);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int rx_ = wm * (RB * 16) + b * 16 + lr;
        fx[b] = rx_ * 64 + ((lg ^ ntf_swz(rx_)) << 4);
    }
    // epilogue constants
    bf16_t* out_lp = (bf16_t*)g.out_lp;
    bf16_t* pre = (bf16_t*)g.pre;
    const bf16_t* aux = (const bf16_t*)g.aux;
    constexpr bool GEN = EPI == EPI_GENERIC;
    const bool by_arow = (EPI == EPI_GENERIC || EPI == EPI_RES) && (g.flags & STAR_GEMM_RES_BY_AROW) != 0;
    const bool pre_grad = GEN ? (g.flags & STAR_GEMM_PRE_IS_GRAD) != 0 : true;
    const bool has_pre = GEN ? pre != nullptr : (EPI == EPI_GELU_GRAD && pre != nullptr);   // inference: GELU without its derivative
    const bool has_bias = EPI == EPI_MULAUX ? false : g.bias != nullptr;
    const bool has_res = GEN ? g.res != nullptr : EPI == EPI_RES;
    const bool has_res2 = GEN && g.res2 != nullptr;
    const int act = GEN ? g.act : (EPI == EPI_GELU_GRAD ? STAR_ACT_GELU : (EPI == EPI_MULAUX ? STAR_ACT_MULAUX : STAR_ACT_NONE));
    const bool use_aux = act == STAR_ACT_DGELU || act == STAR_ACT_MULAUX || act == STAR_ACT_DRELU;
    const int nk = SPL ? (K + NTF_BK - 1) / NTF_BK : K / NTF_BK;

    // ---- persistent over tiles (grid = resident blocks): the ring of the NEXT tile is filled while this tile's
    // epilogue runs, and the epilogue's stores drain under the next main loop.  Tiles t, t + grid, ... of one block
    // stay on its XCD (grid % 8 == 0), and xcd_remap gives each XCD a contiguous range (A panel reuse in its L2).
    int t = blockIdx.x;
    int m_blk, n_blk;
    {
        const int tile = xcd_remap(t, ntiles);
        m_blk = (tile / tiles_n) * BM;
        n_blk = (tile % tiles_n) * BN;
    }
    dma_setup(m_blk, n_blk);
    for (int s0 = 0; s0 < PF && s0 < nk; ++s0) issue_any(s0, nk);
    bool first = true;
    while (true) {
        // ---- epilogue set-up, done BEFORE the main loop so that its global reads (row indices, the first two row
        // groups of the residual / aux operand) are in flight under the MFMA work.  Lane (lr, lg) owns rows
        // wm*RB*16 + b*16 + lr (b = 0..RB-1) and four 4-column runs (ntf_run_off); the four lanes of a row cover a contiguous
        // 64-byte segment per store instruction.  The output may alias the residual (x += ...), so the compiler would keep
        // every row group's reads behind the previous group's stores: reads are issued by hand, two groups ahead.
        const int cm_blk = m_blk;
        int roff[4];      // column of run a (clamped to 0 when the run lies past N: reads stay in bounds, stores are skipped)
        bool rok[4];      // N % 16 == 0 on this path: a run (and a 16-byte pair of bf16 runs) is all in or all out
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = n_blk + wn * 64 + ntf_run_off<CG>(lg, a);
            rok[a] = col < N;
            roff[a] = rok[a] ? col : 0;
        }
        int drow[RB], arow[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int m = cm_blk + wm * (RB * 16) + b * 16 + lr;
            drow[b] = -1;
            arow[b] = -1;
            if (m < M) {
                drow[b] = g.c_rows ? g.c_rows[m] : m;
                arow[b] = (by_arow && g.a_rows) ? g.a_rows[m] : drow[b];
            }
        }
        uint4 pf[2][4];   // raw residual rows (4 runs x 16 B of float32), or -- without a residual -- aux rows (bf16: 16 B per run pair / 8 B per run)
        auto prefetch = [&](int b, uint4 (&dst)[4]) {   // unconditional: rows that will not be stored read a clamped address
            if (has_res) {
                const long long rb = max(by_arow ? arow[b] : drow[b], 0);
                const float* src = g.res + (size_t)(g.res_mod > 0 ? rb % g.res_mod : rb) * N;
#pragma unroll
                for (int a = 0; a < 4; ++a) dst[a] = *reinterpret_cast<const uint4*>(src + roff[a]);
            } else if (use_aux) {
                const int m = min(cm_blk + wm * (RB * 16) + b * 16 + lr, M - 1);
                const bf16_t* src = aux + (size_t)m * N;
                if constexpr (CG == 8) {
                    dst[0] = *reinterpret_cast<const uint4*>(src + roff[0]);
                    dst[1] = *reinterpret_cast<const uint4*>(src + roff[2]);
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const uint2 t = *reinterpret_cast<const uint2*>(src + roff[a]);
                        dst[a] = make_uint4(t.x, t.y, 0u, 0u);
                    }
                }
            }
        };
        // 16 values of a lane <-> its four runs
        auto ld_runs = [&](const float* row, float (&v)[16]) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 t = *reinterpret_cast<const float4*>(row + roff[a]);
                v[4 * a] = t.x; v[4 * a + 1] = t.y; v[4 * a + 2] = t.z; v[4 * a + 3] = t.w;
            }
        };
        auto ld_runs_lp = [&](const bf16_t* row, float (&v)[16]) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint2 t = *reinterpret_cast<const uint2*>(row + roff[a]);
                v[4 * a] = __uint_as_float(t.x << 16); v[4 * a + 1] = __uint_as_float(t.x & 0xffff0000u);
                v[4 * a + 2] = __uint_as_float(t.y << 16); v[4 * a + 3] = __uint_as_float(t.y & 0xffff0000u);
            }
        };
        auto st_runs = [&](float* row, const float (&v)[16]) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (rok[a]) *reinterpret_cast<float4*>(row + roff[a]) = make_float4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]);
        };
        auto st_runs_lp = [&](bf16_t* row, const float (&v)[16]) {
            if constexpr (SPL != 0) {       // hi at column n, lo = bf16(v - hi) at column N + n
                float lo[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) lo[e] = v[e] - bf2f(f2bf(v[e]));
                if constexpr (CG == 8) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (rok[2 * h]) {
                            *reinterpret_cast<uint4*>(row + roff[2 * h]) = make_uint4(pack_bf16(v[8 * h], v[8 * h + 1]), pack_bf16(v[8 * h + 2], v[8 * h + 3]),
                                                                                      pack_bf16(v[8 * h + 4], v[8 * h + 5]), pack_bf16(v[8 * h + 6], v[8 * h + 7]));
                            *reinterpret_cast<uint4*>(row + N + roff[2 * h]) = make_uint4(pack_bf16(lo[8 * h], lo[8 * h + 1]), pack_bf16(lo[8 * h + 2], lo[8 * h + 3]),
                                                                                          pack_bf16(lo[8 * h + 4], lo[8 * h + 5]), pack_bf16(lo[8 * h + 6], lo[8 * h + 7]));
                        }
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        if (rok[a]) {
                            *reinterpret_cast<uint2*>(row + roff[a]) = make_uint2(pack_bf16(v[4 * a], v[4 * a + 1]), pack_bf16(v[4 * a + 2], v[4 * a + 3]));
                            *reinterpret_cast<uint2*>(row + N + roff[a]) = make_uint2(pack_bf16(lo[4 * a], lo[4 * a + 1]), pack_bf16(lo[4 * a + 2], lo[4 * a + 3]));
                        }
                }
            } else if constexpr (CG == 8) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (rok[2 * h])
                        *reinterpret_cast<uint4*>(row + roff[2 * h]) = make_uint4(pack_bf16(v[8 * h], v[8 * h + 1]), pack_bf16(v[8 * h + 2], v[8 * h + 3]),
                                                                                  pack_bf16(v[8 * h + 4], v[8 * h + 5]), pack_bf16(v[8 * h + 6], v[8 * h + 7]));
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (rok[a]) *reinterpret_cast<uint2*>(row + roff[a]) = make_uint2(pack_bf16(v[4 * a], v[4 * a + 1]), pack_bf16(v[4 * a + 2], v[4 * a + 3]));
            }
        };
        if (has_res || use_aux) {
            prefetch(0, pf[0]);
            prefetch(1, pf[1]);
        }

        f32x4_t acc[4][RB];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt) {
            // stages allowed to stay in flight behind stage kt; a later tile's first step also retires the previous
            // epilogue's stores and this tile's prefetches, whose number is not known: plain vmcnt(0)
            const int rem = (kt == 0 && !first) ? 0 : min(nk - 1 - kt, PF - 1);
            static_assert(3 * NDMA <= 63, "counted vmcnt");
            if (rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!(NT_DIAG & 1) && kt + PF < nk) issue_any(kt + PF, nk);
            const unsigned char* st = smem + (kt % NST) * STAGE;
            short8_t w8[4], x8[RB];
#pragma unroll
            for (int a = 0; a < 4; ++a) w8[a] = *reinterpret_cast<const short8_t*>(st + fw[a]);
#pragma unroll
            for (int b = 0; b < RB; ++b) x8[b] = *reinterpret_cast<const short8_t*>(st + fx[b]);
            if constexpr (SPL != 0) {
                short8_t wl[4], xl[RB];
#pragma unroll
                for (int a = 0; a < 4; ++a) wl[a] = *reinterpret_cast<const short8_t*>(st + HALF + fw[a]);
#pragma unroll
                for (int b = 0; b < RB; ++b) xl[b] = *reinterpret_cast<const short8_t*>(st + HALF + fx[b]);
                // the two small products first, the hi * hi product last (the order float32 would round least in)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < RB; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wl[a]), __builtin_bit_cast(bf16x8_t, x8[b]),
                                                                            acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w8[a]), __builtin_bit_cast(bf16x8_t, xl[b]),
                                                                            acc[a][b], 0, 0, 0);
                    }
            }
            if (NT_DIAG & 2) {
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(w8[a]), "v"(x8[a % RB]));
            } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w8[a]), __builtin_bit_cast(bf16x8_t, x8[b]),
                                                                        acc[a][b], 0, 0, 0);
            }
        }

        // ---- hand the ring to the next tile before the epilogue
        const int tn = t + (int)gridDim.x;
        const bool more = tn < ntiles;
        if (more) {
            __builtin_amdgcn_s_barrier();   // every wave has consumed the last stages
            asm volatile("" ::: "memory");
            const int tile = xcd_remap(tn, ntiles);
            m_blk = (tile / tiles_n) * BM;
            n_blk = (tile % tiles_n) * BN;
            dma_setup(m_blk, n_blk);
            for (int s0 = 0; s0 < PF && s0 < nk; ++s0) issue_any(s0, nk);
        }

        // ---- epilogue straight from the accumulators (its stores drain under the next tile's main loop)
        {
        float bias16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bias16[e] = 0.f;
        if (has_bias) ld_runs(g.bias, bias16);
        // arithmetic and stores, in place in the accumulator registers
#pragma unroll
        for (int bb = 0; bb < RB; ++bb)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][bb][j] += bias16[a * 4 + j];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int m = cm_blk + wm * (RB * 16) + b * 16 + lr;
            const long long d = drow[b];
            float cur[16];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cur[4 * i] = __uint_as_float(pf[b & 1][i].x); cur[4 * i + 1] = __uint_as_float(pf[b & 1][i].y);
                    cur[4 * i + 2] = __uint_as_float(pf[b & 1][i].z); cur[4 * i + 3] = __uint_as_float(pf[b & 1][i].w);
                }
            } else if constexpr (CG == 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned w[4] = {pf[b & 1][i].x, pf[b & 1][i].y, pf[b & 1][i].z, pf[b & 1][i].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cur[8 * i + 2 * k] = __uint_as_float(w[k] << 16);
                        cur[8 * i + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cur[4 * i] = __uint_as_float(pf[b & 1][i].x << 16); cur[4 * i + 1] = __uint_as_float(pf[b & 1][i].x & 0xffff0000u);
                    cur[4 * i + 2] = __uint_as_float(pf[b & 1][i].y << 16); cur[4 * i + 3] = __uint_as_float(pf[b & 1][i].y & 0xffff0000u);
                }
            }
            if (b + 2 < RB && (has_res || use_aux)) prefetch(b + 2, pf[b & 1]);
            if (d < 0) continue;
            float v[16];
            if (by_arow && arow[b] < 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = 0.f;
                if (out_lp) st_runs_lp(out_lp + (size_t)d * (SPL ? 2 * N : N), v);
                if (g.out_f32) st_runs(g.out_f32 + (size_t)d * N, v);
                continue;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[a * 4 + j] = acc[a][b][j];
            if (has_pre && !pre_grad) st_runs_lp(pre + (size_t)d * N, v);
            if (act == STAR_ACT_GELU) {
                if (has_pre && pre_grad) {
                    float dv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) gelu_both(v[e], v[e], dv[e]);
                    st_runs_lp(pre + (size_t)d * N, dv);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = gelu_f(v[e]);
                }
            } else if (act == STAR_ACT_RELU) {
                if (has_pre && pre_grad) {
                    float dv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) dv[e] = v[e] > 0.f ? 1.f : 0.f;
                    st_runs_lp(pre + (size_t)d * N, dv);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // NaN stays NaN, as torch.relu (a max would drop it)
            } else if (use_aux) {
                if (has_res) {   // both a residual and an activation gradient: the aux row was not prefetched
                    float t[16];
                    ld_runs_lp(aux + (size_t)m * N, t);
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        v[e] = act == STAR_ACT_DGELU ? v[e] * dgelu_f(t[e]) : (act == STAR_ACT_MULAUX ? v[e] * t[e] : (t[e] > 0.f ? v[e] : 0.f));
                } else if (act == STAR_ACT_DGELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] *= dgelu_f(cur[e]);
                } else if (act == STAR_ACT_MULAUX) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] *= cur[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = cur[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += cur[e];
            }
            if (has_res2) {
                const int rr = (m / g.res2_div) % g.res2_mod;
                float t2[16];
                ld_runs(g.res2 + (size_t)rr * N, t2);
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t2[e];
            }
            if ((NT_DIAG & 4) && v[0] != 12345.678f) continue;       // diagnostic: no result stores
            if (out_lp) st_runs_lp(out_lp + (size_t)d * (SPL ? 2 * N : N), v);
            if (g.out_f32) st_runs(g.out_f32 + (size_t)d * N, v);
        }
        }
        if (!more) break;
        t = tn;
        first = false;
    }
}

template <int NST, int EPI, int WM, int WN, int RB, int CG>
static void launch_nt_fast_cg(const StarGemm* g, hipStream_t s) {
    constexpr int BM = WM * RB * 16, BN = WN * 64;
    const int tiles = cdiv(g->M, BM) * cdiv(g->N, BN);
    const size_t lds = (size_t)NST * (BM + BN) * 64;   // the ring only: the epilogue does not stage through LDS
    static unsigned long long attr_done = 0;       // per device ordinal
    if (lds > 48 * 1024 && star_once_per_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast_kernel<NST, EPI, WM, WN, RB, CG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    // The kernel can walk several tiles per block (grid < tiles), prefilling the next tile's ring under the epilogue.
    // Measured on MI355X this is no faster than one tile per block, so every tile gets its own block and the hardware
    // dispatcher balances the tail.
    int grid = tiles;
    static const char* penv = getenv("STAR_AMD_GEMM_PERSIST");      // 1: resident workgroups walk the tiles (A/B)
    if (penv && penv[0] == '1') {
        int occ = (int)(160 * 1024 / lds);
        const int by_waves = (WM * WN == 4 ? 3 : (RB == 8 ? 1 : 2));
        if (occ > by_waves) occ = by_waves;
        if (occ < 1) occ = 1;
        const int resident = star_num_cus() * occ;
        if (tiles > resident) grid = resident;
    }
    hipLaunchKernelGGL((gemm_nt_fast_kernel<NST, EPI, WM, WN, RB, CG>), dim3(grid), dim3(WM * WN * 64), lds, s, *g);
}

template <int NST, int EPI, int WM, int WN, int RB>
static void launch_nt_fast_geo(const StarGemm* g, hipStream_t s) {
    // column groups by output type: any bf16 output -> 8 columns per 16-byte store, float32 only -> 4
    if (g->out_lp || g->pre) launch_nt_fast_cg<NST, EPI, WM, WN, RB, 8>(g, s);
    else launch_nt_fast_cg<NST, EPI, WM, WN, RB, 4>(g, s);
}

// Geometry by epilogue and shape, from measurements at M = 40 960 on MI355X (profiles/r01_d_gemm_geometry.log):
//   plain Linear (bias only, bf16 or f32 out): the 256 x 128 tile wins by 10-20 % (no spills at 128 registers, 16 waves per CU);
//   fc1 + GELU at N >= 1536: the 256 x 256 tile wins by 12 %;  residual / aux epilogues: the 128 x 128 tile (the larger tiles
//   spill or halve the occupancy); problems whose 128 x 128 tiles cover at most half the CUs: 64 x 128 tiles (M = 5120, N = 384: fc2 + residual
//   24.9 -> 18.1 us, the fc1 input gradient 19.9 -> 15.5; at M = 10 240 the same shapes lose 5-15 %).  STAR_AMD_GEMM_GEO=0..3 forces one geometry.
template <int EPI>
static int nt_geometry(const StarGemm* g) {
    const char* env = getenv("STAR_AMD_GEMM_GEO");      // read per call: a test walks the geometries
    if (env && env[0] >= '0' && env[0] <= '3') return env[0] - '0';
    if ((long long)cdiv(g->M, 128) * cdiv(g->N, 128) * 2 <= star_num_cus()) return 3;   // few tiles: halve them (one-scene inference)
    if (EPI == EPI_LINEAR && (long long)cdiv(g->M, 256) * cdiv(g->N, 128) >= star_num_cus()) return 1;
    if (EPI == EPI_GELU_GRAD && g->N >= 1536 && (long long)cdiv(g->M, 256) * cdiv(g->N, 256) >= 2 * star_num_cus()) return 2;
    return 0;
}

template <int NST, int EPI>
static void launch_nt_fast_epi(const StarGemm* g, hipStream_t s) {
    switch (EPI == EPI_GENERIC ? 0 : nt_geometry<EPI>(g)) {   // the run-time epilogue needs too many registers for the larger tiles
        case 2: launch_nt_fast_geo<NST, EPI, 2, 4, 8>(g, s); break;
        case 1: launch_nt_fast_geo<NST, EPI, 4, 2, 4>(g, s); break;
        case 3: launch_nt_fast_geo<NST, EPI, 2, 2, 2>(g, s); break;
        default: launch_nt_fast_geo<NST, EPI, 2, 2, 4>(g, s); break;
    }
}
// ------------------------------------------------------------------------------------------------------
// W-in-registers NT GEMM for K = 32 NK <= 384 (plain Linear, bf16 output).  What bounds the tiled kernel above at small K
// is operand traffic through LDS: per MFMA it moves 512 B of fragment reads plus 256 B of LDS-DMA writes, and re-fetches
// the W panel for every tile.  Here a workgroup owns ONE 64 WN-column panel for its whole life: each consumer wave keeps
// its 64 columns of W (all of K) in registers (4 x NK fragments = 16 NK VGPRs), so only A passes through LDS -- 256 B of
// reads + 64 B of DMA per MFMA -- and L2 sees W once per workgroup.  One wave per SIMD (~330 VGPRs): WN consumer waves
// plus a PRODUCER wave that does nothing but LDS-DMA (its vmcnt sees no stores, so the ring stays counted across tiles
// and the consumers' epilogue stores never have to be drained).  One s_barrier per 32-deep stage; consumers read stage
// s + 1 into a second fragment set before the MFMAs of stage s; a ring slot is refilled two barriers after its reads.
// Measured variants that did not help (qkv shape, 47 us): 128-byte stage rows so that every LDS-DMA lane group fetches a
// whole line (48 us); a 15-stage ring, vmcnt up to 60 (61 us at 64-row tiles); 64-row tiles (52 us).  With the MFMAs
// compiled out the kernel still takes 36 us (one producer wave lands a 1 KiB piece every ~94 cycles = 11 B/clk/CU), with
// the stores compiled out 37 us: the remaining lever is hiding the epilogue of tile t in the MFMA gaps of tile t + 1.
template <int NK, int WN, int RB, int EPI>
__global__ __launch_bounds__((WN + 1) * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_nt_wreg_kernel(const StarGemm g, int row_tiles) {
    static_assert(EPI == EPI_LINEAR || EPI == EPI_MULAUX, "epilogues of the W-in-registers kernel");
    constexpr int BM = RB * 16, NP = RB, NST = 8, PF = NST - 2, STAGE = BM * 64, K = NK * 32;   // a deeper ring (vmcnt up to 60) measured slower
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = g.N;
    const int panels = N / (WN * 64);
    const int q = xcd_remap(blockIdx.x, gridDim.x);
    const int panel = q % panels, j0 = q / panels, J = (int)gridDim.x / panels;
    const int my_tiles = j0 < row_tiles ? (row_tiles - j0 + J - 1) / J : 0;
    const int total = my_tiles * NK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == WN) {
        // ---- producer: stage s = (tile s / NK, k-step s % NK) = 64 rows x 64 B, four 16-row pieces
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, 0x7fffffff, 0x00020000);
        unsigned off[8];
        static_assert(NP <= 8, "piece table");
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int r = j * 16 + (lane >> 2);
            off[j] = (unsigned)(r * K * 2 + (((lane & 3) ^ ntf_swz(r)) << 4));
        }
        auto issue = [&](int st_) {
            const int tl = st_ / NK, kt = st_ - tl * NK;
            const unsigned base = (unsigned)((j0 + tl * J) * BM) * (unsigned)(K * 2) + (unsigned)(kt * 64);
            unsigned char* dst = smem + (st_ % NST) * STAGE;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + j * 16 * 64), 16, off[j] + base, 0, 0, 0);
        };
        for (int s0 = 0; s0 < PF && s0 < total; ++s0) issue(s0);
        for (int st = 0; st < total; ++st) {
            const int rem = min(total - 1 - st, PF - 1);
            switch (rem * NP) {   // the asm needs immediates: every multiple of 4 up to 60 (vmcnt holds 63)
#define WREG_VM(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
                WREG_VM(60) WREG_VM(56) WREG_VM(52) WREG_VM(48) WREG_VM(44) WREG_VM(40) WREG_VM(36) WREG_VM(32) WREG_VM(28) WREG_VM(24)
                WREG_VM(20) WREG_VM(16) WREG_VM(12) WREG_VM(8) WREG_VM(4)
#undef WREG_VM
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + PF < total) issue(st + PF);
        }
        return;
    }
    // ---- consumers
    const int lr = lane & 15, lg = lane >> 4;
    const int n_blk = panel * (WN * 64) + wave * 64;
    const bf16_t* W = (const bf16_t*)g.W;
    short8_t wreg[4][NK];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n_blk + ntf_run_off<8>((lr >> 2) & 3, a) + (lr & 3);
        const bf16_t* src = W + (size_t)n * K + lg * 8;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) wreg[a][kt] = *reinterpret_cast<const short8_t*>(src + kt * 32);
    }
    int fx[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int rx_ = b * 16 + lr;
        fx[b] = rx_ * 64 + ((lg ^ ntf_swz(rx_)) << 4);
    }
    int roff[4];
    float bias16[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) roff[a] = n_blk + ntf_run_off<8>(lg, a);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) t = *reinterpret_cast<const float4*>(g.bias + roff[a]);
        bias16[4 * a] = t.x; bias16[4 * a + 1] = t.y; bias16[4 * a + 2] = t.z; bias16[4 * a + 3] = t.w;
    }
    bf16_t* out_lp = (bf16_t*)g.out_lp;
    const bf16_t* aux = (const bf16_t*)g.aux;     // EPI_MULAUX: the factor rows
    int st = 0;
    for (int tile = 0; tile < my_tiles; ++tile) {
        const int m_blk = (j0 + tile * J) * BM;
        // EPI_MULAUX: the tile's factor rows are requested before its MFMA chain (consumer waves have no other loads in flight: the
        // LDS-DMA ring belongs to the producer wave), 8 values per (row group, column half)
        uint4 pa[EPI == EPI_MULAUX ? RB : 1][2];
        if constexpr (EPI == EPI_MULAUX) {
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const bf16_t* arow = aux + (size_t)(m_blk + b * 16 + lr) * N;
                pa[b][0] = *reinterpret_cast<const uint4*>(arow + roff[0]);
                pa[b][1] = *reinterpret_cast<const uint4*>(arow + roff[2]);
            }
        }
        f32x4_t acc[4][RB];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        short8_t x8[2][RB];
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const unsigned char* sp = smem + (st % NST) * STAGE;
#pragma unroll
            for (int b = 0; b < RB; ++b) x8[0][b] = *reinterpret_cast<const short8_t*>(sp + fx[b]);
        }
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 1 < NK) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned char* sp = smem + ((st + kt + 1) % NST) * STAGE;
#pragma unroll
                for (int b = 0; b < RB; ++b) x8[(kt + 1) & 1][b] = *reinterpret_cast<const short8_t*>(sp + fx[b]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[a][kt]),
                                                                        __builtin_bit_cast(bf16x8_t, x8[kt & 1][b]), acc[a][b], 0, 0, 0);
        }
        st += NK;
        // epilogue: bias, bf16, two 16-byte stores per row (a row's four lanes write one contiguous 64-byte segment each)
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            bf16_t* row = out_lp + (size_t)(m_blk + b * 16 + lr) * N;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[2 * h + (e >> 2)][b][e & 3] + bias16[8 * h + e];
                if constexpr (EPI == EPI_MULAUX) {
                    const unsigned w[4] = {pa[b][h].x, pa[b][h].y, pa[b][h].z, pa[b][h].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] *= __uint_as_float(w[k] << 16);
                        v[2 * k + 1] *= __uint_as_float(w[k] & 0xffff0000u);
                    }
                }
                *reinterpret_cast<uint4*>(row + roff[2 * h]) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            }
        }
    }
}

// eligibility + launch of the W-in-registers kernel; returns false when the shape / epilogue is not covered.
// Measured alone on MI355X at M = 40 960, K = 384 against the 256 x 128 tile: N = 1152 47.5 vs 53.8 us, N = 1536 62.5 vs
// 74.2 us, N = 384 23.2 vs 22.0 us -> used from N = 1152 up.  STAR_AMD_GEMM_WREG=0 turns it off, =1 takes every eligible shape.
// Round 2: the x-aux input gradient of fc2 (K = 384, N = 1536: `du = (g W2) * gelu'`) runs here too (EPI_MULAUX: the factor rows of a tile
// are requested before its MFMA chain): 104 -> 84 us alone, 12.00 -> 11.90 ms per step.  The fc1 + GELU forward does NOT: with one
// wave per SIMD the GELU arithmetic is serial behind the MFMAs (115 us against 103 for the 256 x 256 tile).
template <int EPI>
static bool launch_nt_wreg(const StarGemm* g, hipStream_t s) {
    static const char* env = getenv("STAR_AMD_GEMM_WREG");
    if (env && env[0] == '0') return false;
    if (EPI != EPI_LINEAR) {
        static const char* env2 = getenv("STAR_AMD_GEMM_WREG_EPI");     // 0: the x-aux epilogue stays on the tiled kernel (A/B)
        if (env2 && env2[0] == '0') return false;
    }
    constexpr int RB = 8, BM = RB * 16, LDS = 8 * BM * 64;
    if (g->K != 384 || g->N % 192 != 0 || g->M % BM != 0 || g->a_rows || g->c_rows || g->out_f32 || !g->out_lp) return false;
    if (EPI == EPI_MULAUX && !g->aux) return false;
    if (!(env && env[0] == '1') && g->N < 1152) return false;
    if ((long long)g->M * g->K * 2 >= 0x7fffffffLL) return false;
    const int panels = g->N / 192;
    int l = panels;
    while (l % 8 != 0) l += panels;            // lcm(8, panels): whole panels per workgroup, whole workgroup sets per XCD
    const int cus = star_num_cus();
    if (l > cus) return false;
    const int grid = cus / l * l;
    static unsigned long long attr = 0;            // per device ordinal
    if (star_once_per_device(attr))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_wreg_kernel<12, 3, RB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL((gemm_nt_wreg_kernel<12, 3, RB, EPI>), dim3(grid), dim3(256), (size_t)LDS, s, *g, g->M / BM);
    return true;
}

// dt = STAR_BF16X3: the 256 x 128 tile (8 waves, two per SIMD: 48 MFMAs per wave and k-step against 6 LDS-DMA pieces) for every
// epilogue -- the per-wave accumulator block is the 128 x 128 tile's, so the epilogues keep their register budgets; a stage is
// 48 KB (hi and lo images), the ring 144 KB: one workgroup per CU.
template <int EPI, int CG>
static void launch_nt_split_cg(const StarGemm* g, hipStream_t s) {
    constexpr int NST = 3, WM = 4, WN = 2, RB = 4, BM = WM * RB * 16, BN = WN * 64;
    constexpr int LDS = NST * 2 * (BM + BN) * 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast_kernel<NST, EPI, WM, WN, RB, CG, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL((gemm_nt_fast_kernel<NST, EPI, WM, WN, RB, CG, 1>), dim3(cdiv(g->M, BM) * cdiv(g->N, BN)), dim3(WM * WN * 64), LDS, s, *g);
}
template <int EPI>
static void launch_nt_split_epi(const StarGemm* g, hipStream_t s) {
    if (g->out_lp) launch_nt_split_cg<EPI, 8>(g, s); else launch_nt_split_cg<EPI, 4>(g, s);
}
static int launch_nt_split(const StarGemm* g, hipStream_t s) {
    const bool plain = !g->res2 && g->act == STAR_ACT_NONE;
    if (plain && !g->res) launch_nt_split_epi<EPI_LINEAR>(g, s);
    else if (plain) launch_nt_split_epi<EPI_RES>(g, s);
    else if (g->act == STAR_ACT_GELU && !g->res && !g->res2) launch_nt_split_epi<EPI_GELU_GRAD>(g, s);
    else launch_nt_split_epi<EPI_GENERIC>(g, s);
    return 0;
}

template <int NST>
static int launch_nt_fast(const StarGemm* g, hipStream_t s) {
    const bool plain = !g->res2 && !g->aux && !g->pre && g->act == STAR_ACT_NONE;
    if (plain && !g->res && star_gemm_rows_try(g, s) == 0) return 0;
    if (plain && !g->res && launch_nt_wreg<EPI_LINEAR>(g, s)) return 0;
    if (plain && !g->res) launch_nt_fast_epi<NST, EPI_LINEAR>(g, s);
    else if (plain) launch_nt_fast_epi<NST, EPI_RES>(g, s);
    else if (g->act == STAR_ACT_GELU && (!g->pre || (g->flags & STAR_GEMM_PRE_IS_GRAD)) && !g->aux && !g->res && !g->res2) launch_nt_fast_epi<NST, EPI_GELU_GRAD>(g, s);
    else if (g->act == STAR_ACT_MULAUX && !g->bias && !g->res && !g->res2 && !g->pre) {
        if (!launch_nt_wreg<EPI_MULAUX>(g, s)) launch_nt_fast_epi<NST, EPI_MULAUX>(g, s);
    }
    else launch_nt_fast_epi<NST, EPI_GENERIC>(g, s);
    return 0;
}

// implicit 3 x 3 convolution: the 256 x 128 tile when it fills the chip, else 128 x 128; column groups by output type
template <int WM, int WN, int RB, int CG>
static void launch_nt_conv_geo(const StarGemm* g, hipStream_t s) {
    constexpr int NST = 3, BM = WM * RB * 16, BN = WN * 64;
    const int tiles = cdiv(g->M, BM) * cdiv(g->N, BN);
    const size_t lds = (size_t)NST * (BM + BN) * 64;
    static unsigned long long attr_done = 0;
    if (lds > 48 * 1024 && star_once_per_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast_kernel<NST, EPI_LINEAR, WM, WN, RB, CG, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL((gemm_nt_fast_kernel<NST, EPI_LINEAR, WM, WN, RB, CG, 0, 1>), dim3(tiles), dim3(WM * WN * 64), lds, s, *g);
}
static void launch_nt_conv(const StarGemm* g, hipStream_t s) {
    const bool big = (long long)cdiv(g->M, 256) * cdiv(g->N, 128) >= star_num_cus();
    if (g->out_lp) { if (big) launch_nt_conv_geo<4, 2, 4, 8>(g, s); else launch_nt_conv_geo<2, 2, 4, 8>(g, s); }
    else { if (big) launch_nt_conv_geo<4, 2, 4, 4>(g, s); else launch_nt_conv_geo<2, 2, 4, 4>(g, s); }
}

template <typename T, int TM, int TN>
static int launch_nt(const StarGemm* g, hipStream_t s) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    const int tiles = cdiv(g->M, BM) * cdiv(g->N, BN);
    const size_t lds = (size_t)2 * (BM + BN) * 128;
    hipLaunchKernelGGL((gemm_nt_kernel<T, TM, TN>), dim3(tiles), dim3(GT), lds, s, *g);
    return 0;
}

extern "C" int star_gemm(const StarGemm* g, void* stream) {
    STAR_CHECK(g && g->A && g->W, "star_gemm: null operand");
    STAR_CHECK(g->M >= 0 && g->N > 0 && g->K > 0, "star_gemm: bad shape M=%d N=%d K=%d", g->M, g->N, g->K);
    STAR_CHECK(g->dt == STAR_BF16 || g->dt == STAR_F32 || g->dt == STAR_BF16X3, "star_gemm: unknown dtype %d", g->dt);
    if (g->dt == STAR_BF16X3) {
        STAR_CHECK(g->K % 8 == 0 && g->N % 16 == 0, "star_gemm(bf16x3): K=%d must be a multiple of 8 and N=%d of 16", g->K, g->N);
        STAR_CHECK(!g->pre && !g->aux && (g->act == STAR_ACT_NONE || g->act == STAR_ACT_GELU || g->act == STAR_ACT_RELU),
                   "star_gemm(bf16x3): forward epilogues only (no pre-activation output, no activation gradients)");
        STAR_CHECK(g->out_lp || g->out_f32, "star_gemm: no output");
        STAR_CHECK(!g->res2 || (g->res2_div > 0 && g->res2_mod > 0), "star_gemm: res2 needs div/mod");
        STAR_CHECK(((reinterpret_cast<uintptr_t>(g->A) | reinterpret_cast<uintptr_t>(g->W)) & 15) == 0, "star_gemm(bf16x3): operands must be 16-byte aligned");
        STAR_CHECK((long long)g->N * g->K * 4 < 0x7fffffffLL, "star_gemm(bf16x3): weight over 2 GiB");
        if (g->M == 0) return 0;
        launch_nt_split(g, (hipStream_t)stream);
        STAR_LAUNCH_CHECK("star_gemm(bf16x3)");
        return 0;
    }
    const int epc = g->dt == STAR_BF16 ? 8 : 4;
    STAR_CHECK(g->K % epc == 0, "star_gemm: K=%d must be a multiple of %d", g->K, epc);
    STAR_CHECK(g->N % 4 == 0, "star_gemm: N=%d must be a multiple of 4", g->N);
    STAR_CHECK(g->out_lp || g->out_f32 || g->pre, "star_gemm: no output");
    STAR_CHECK(!(g->act == STAR_ACT_DGELU || g->act == STAR_ACT_DRELU || g->act == STAR_ACT_MULAUX) || g->aux, "star_gemm: activation gradient needs aux");
    STAR_CHECK(!g->res2 || (g->res2_div > 0 && g->res2_mod > 0), "star_gemm: res2 needs div/mod");
    if (g->M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (g->conv != 0) {
        const int lw = g->conv & 31, lh = (g->conv >> 5) & 31;
        STAR_CHECK((g->conv & 0x40000000) && lw + lh <= 30 && g->M % (1 << (lw + lh)) == 0, "star_gemm(conv): bad conv word 0x%x for M=%d (STAR_GEMM_CONV3X3)", g->conv, g->M);
        STAR_CHECK(g->dt == STAR_BF16 && g->K % (9 * NTF_BK) == 0 && g->N % 16 == 0, "star_gemm(conv): bf16, K=%d a multiple of 288, N=%d of 16", g->K, g->N);
        STAR_CHECK(!g->a_rows && !g->c_rows && !g->res && !g->res2 && !g->aux && !g->pre && g->act == STAR_ACT_NONE && g->flags == 0,
                   "star_gemm(conv): bias-only epilogue, no row tables");
        STAR_CHECK((long long)g->M * (g->K / 9) * 2 < 0x7fffffffLL && (long long)g->N * g->K * 2 < 0x7fffffffLL, "star_gemm(conv): operand over 2 GiB");
        STAR_CHECK(((reinterpret_cast<uintptr_t>(g->A) | reinterpret_cast<uintptr_t>(g->W)) & 15) == 0, "star_gemm(conv): operands must be 16-byte aligned");
        launch_nt_conv(g, s);
        STAR_LAUNCH_CHECK("star_gemm(conv)");
        return 0;
    }
    const bool narrow = g->N <= 32;
    const bool fast_ok = g->dt == STAR_BF16 && !narrow && g->K % NTF_BK == 0 && g->N % 16 == 0 && (long long)g->N * g->K * 2 < 0x7fffffffLL &&
                         ((reinterpret_cast<uintptr_t>(g->A) | reinterpret_cast<uintptr_t>(g->W)) & 15) == 0;
    if (fast_ok) {
        launch_nt_fast<3>(g, s);
    } else if (g->dt == STAR_BF16) {
        if (narrow) launch_nt<bf16_t, 4, 1>(g, s); else launch_nt<bf16_t, 4, 4>(g, s);
    } else {
        if (narrow) launch_nt<float, 4, 1>(g, s); else launch_nt<float, 4, 4>(g, s);
    }
    STAR_LAUNCH_CHECK("star_gemm");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// TN: dW[n, k] = sum_m dY[m, n] * X[xrow(m), k].  Output tile 128(k) x 128(n); the MFMA A operand is
// X^T (rows = k), the B operand dY (cols = n): each lane ends with 4 consecutive k of one n.
// LDS tiles are [64 rows of m][128 columns] bf16 (256-B rows), 32-byte slots XOR-swizzled.
#define WG_BKM 64
#define WG_TILE 128

__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int tn_off(int row, int slot) { return row * 256 + ((slot ^ tn_swz(row)) << 5); }

__device__ __forceinline__ short4_t ds_read_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p));
}

__global__ __launch_bounds__(GT) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X,
                                                          const int32_t* __restrict__ x_rows, float* __restrict__ out,
                                                          float* __restrict__ bias_part, int M, int N, int K, int m_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TILE_BYTES = WG_BKM * 256;

    const int tiles_n = (N + WG_TILE - 1) / WG_TILE;
    const int n_blk = (blockIdx.x % tiles_n) * WG_TILE, k_blk = (blockIdx.x / tiles_n) * WG_TILE;
    const int split = blockIdx.y;
    const int m_begin = split * m_per_split;
    const int m_end = min(M, m_begin + m_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wn = wave >> 1;  // wave tile: 64 k x 64 n

    // staging: 64 rows x 16 chunks (16 B) per operand tile = 1024 chunks, 4 per thread
    const int ld_chunk = tid & 15, ld_row = tid >> 4;  // rows ld_row + 16*i
    uint4 ry[4], rx[4];
    // bias gradient (column sums of dY) rides along in the k-tile-0 blocks: the dY chunks are in registers anyway
    const bool do_bias = bias_part != nullptr && k_blk == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool yin = n_blk + ld_chunk * 8 < N, xin = k_blk + ld_chunk * 8 < K;
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ld_row + 16 * i;
            ry[i] = make_uint4(0, 0, 0, 0);
            rx[i] = make_uint4(0, 0, 0, 0);
            if (m < m_end) {
                if (yin) ry[i] = *reinterpret_cast<const uint4*>(dY + (size_t)m * N + n_blk + ld_chunk * 8);
                if (xin) {
                    const long long xr = x_rows ? (long long)x_rows[m] : (long long)m;
                    if (xr >= 0) rx[i] = *reinterpret_cast<const uint4*>(X + (size_t)xr * K + k_blk + ld_chunk * 8);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(w[e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 16 * i;
            const int off = tn_off(row, ld_chunk >> 1) + ((ld_chunk & 1) << 4);
            *reinterpret_cast<uint4*>(smem + buf * 2 * TILE_BYTES + off) = ry[i];
            *reinterpret_cast<uint4*>(smem + buf * 2 * TILE_BYTES + TILE_BYTES + off) = rx[i];
        }
    };

    f32x4_t acc[4][4];  // [k tile][n tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: 16-lane group lg covers m rows 8*lg..8*lg+7 of a 32-row step;
    // lane 4q+p of the group supplies row q, columns 4p..4p+3 of a 16-column slot.
    const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;

    const int nsteps = (m_end - m_begin + WG_BKM - 1) / WG_BKM;
    if (nsteps > 0) {
        load_tile(m_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) load_tile(m_begin + (st + 1) * WG_BKM);
        const unsigned char* pY = smem + buf * 2 * TILE_BYTES;
        const unsigned char* pX = pY + TILE_BYTES;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {  // two 32-row MFMA steps per 64-row tile
            short8_t fx[4], fy[4];
            const int r0 = ms * 32 + lg * 8 + lq;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int slot = wk * 4 + a;
                const short4_t lo = ds_read_tr16(pX + tn_off(r0, slot) + lp * 8);
                const short4_t hi = ds_read_tr16(pX + tn_off(r0 + 4, slot) + lp * 8);
                fx[a] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int slot = wn * 4 + b;
                const short4_t lo = ds_read_tr16(pY + tn_off(r0, slot) + lp * 8);
                const short4_t hi = ds_read_tr16(pY + tn_off(r0 + 4, slot) + lp * 8);
                fy[b] = (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fx[a]),
                                                                        __builtin_bit_cast(bf16x8_t, fy[b]), acc[a][b], 0, 0, 0);
        }
        if (st + 1 < nsteps) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (do_bias) {  // fold the 16 row lanes of every column chunk through LDS (all tiles are consumed by now)
        float* sb = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[(ld_row * 16 + ld_chunk) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128 && n_blk + tid < N) {
            float a = 0.f;
            for (int r = 0; r < 16; ++r) a += sb[(r * 16 + (tid >> 3)) * 8 + (tid & 7)];
            bias_part[(size_t)split * N + n_blk + tid] = a;
        }
    }
    // D[i = k][j = n]: lane holds n = .. + (lane & 15), k = .. + (lane >> 4) * 4 + e
    float* o = out + (size_t)split * N * K;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n_blk + wn * 64 + b * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int k0 = k_blk + wk * 64 + a * 16 + lg * 4;
            if (k0 >= K) continue;
            *reinterpret_cast<float4*>(o + (size_t)n * K + k0) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
    }
}

// bf16 TN fast path: the same tile and fragment scheme, operands streamed by LDS-DMA through an NST-deep
// ring (32 rows of m per stage) behind a counted vmcnt, exactly like gemm_nt_fast_kernel.  Rows past the end
// of the matrices are zero-filled by the buffer range check (splits are multiples of 32 rows, so a stage
// never straddles two splits).  The bias gradient costs no memory traffic at all: in the k-tile-0 blocks
// one extra MFMA per n-tile multiplies the dY fragment with an all-ones A operand (row sums on the matrix core).
// Main loop of one 128 x 128 output tile over rows [m_begin, m_end): accumulators acc[a][b] (k block a, n block b of the wave) and,
// with do_bias_tile, the column sums of dY in accb[b][0] (waves with wk == 0).
template <int NST, int NWV>
__device__ __forceinline__ void tn_fast_mainloop(unsigned char* smem, const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, int M, int N,
                                                 int K, int n_blk, int k_blk, int m_begin, int m_end, bool do_bias_tile,
                                                 f32x4_t (&acc)[4][16 / NWV], f32x4_t (&accb)[16 / NWV]) {
    // NWV waves: 2 over k x NWV/2 over n; a wave owns 64 (k) x 128/(NWV/2) (n) of the 128 x 128 tile
    constexpr int TB = 32 * 256, STAGE = 2 * TB, PF = NST - 1, NJ = 8 / NWV, NB = 16 / NWV, RPW = 32 / NWV;
    static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;

    __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dY), 0, (unsigned)((long long)M * N * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (unsigned)((long long)M * K * 2), 0x00020000);
    unsigned offY[2], offX[2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = wave * RPW + j * 4 + (lane >> 4);       // row inside the 32-row stage
        const int pc = lane & 15;
        const int col = (((pc >> 1) ^ tn_swz(r)) << 4) + ((pc & 1) << 3);
        offY[j] = n_blk + col < N ? (unsigned)(((long long)(m_begin + r) * N + n_blk + col) * 2) : 0x80000000u;
        offX[j] = k_blk + col < K ? (unsigned)(((long long)(m_begin + r) * K + k_blk + col) * 2) : 0x80000000u;
    }
    const unsigned stepY = 32u * N * 2, stepX = 32u * K * 2;
    auto issue = [&](int st_) {
        unsigned char* sb = smem + (st_ % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (__attribute__((address_space(3))) void*)(sb + (wave * RPW + j * 4) * 256), 16,
                                                     offY[j] + st_ * stepY, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(sb + TB + (wave * RPW + j * 4) * 256), 16,
                                                     offX[j] + st_ * stepX, 0, 0, 0);
        }
    };
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        accb[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = do_bias_tile && wk == 0;
    const short one = (short)0x3F80;
    const short8_t ones = (short8_t){one, one, one, one, one, one, one, one};
    const int r0 = lg * 8 + lq;
    // rows r0 and r0+4 share the swizzle bits (row&3) only partly: the +4 variants are computed, not offset
    int ax[4], axh[4], ay[NB], ayh[NB];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ax[a] = TB + tn_off(r0, wk * 4 + a) + lp * 8;
        axh[a] = TB + tn_off(r0 + 4, wk * 4 + a) + lp * 8;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        ay[b] = tn_off(r0, wn * NB + b) + lp * 8;
        ayh[b] = tn_off(r0 + 4, wn * NB + b) + lp * 8;
    }
    const unsigned lds_base = lds_u32(smem);
    const int nsteps = (m_end - m_begin + 31) / 32;
    for (int s0 = 0; s0 < PF && s0 < nsteps; ++s0) issue(s0);
    for (int st = 0; st < nsteps; ++st) {
        const int rem = min(nsteps - 1 - st, PF - 1);
        if (rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NJ) : "memory");      // 2 NJ LDS-DMA loads per wave and stage
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NJ) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + PF < nsteps) issue(st + PF);
        // transposed reads as inline asm (common.h): the builtin would make the compiler drain the DMA ring (vmcnt(0))
        const unsigned sb = lds_base + (st % NST) * STAGE;
        // fragment reads in the order the MFMAs consume them (dY blocks first, then one X block per MFMA row), each MFMA row behind a
        // COUNTED lgkmcnt: ds_read_b64_tr_b16 at two waves per SIMD reaches a fraction of the LDS rate when every stage drains to
        // lgkmcnt(0) before its first MFMA (measured: no gain by itself at 128 x 128 -- the stage is bound by its LDS-DMA issue and barrier, see tn_wide_mainloop)
        short4_t xl[4], xh[4], yl[NB], yh[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            yl[b] = tr16<0>(sb + ay[b]);
            yh[b] = tr16<0>(sb + ayh[b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            xl[a] = tr16<0>(sb + ax[a]);
            xh[a] = tr16<0>(sb + axh[a]);
        }
        short8_t fy[NB];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a == 0) {
                if constexpr (NB == 4) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(yl[0]), "+v"(yl[1]), "+v"(yl[2]), "+v"(yl[3]), "+v"(yh[0]), "+v"(yh[1]), "+v"(yh[2]), "+v"(yh[3]), "+v"(xl[0]), "+v"(xh[0]));
                else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(yl[0]), "+v"(yl[1]), "+v"(yh[0]), "+v"(yh[1]), "+v"(xl[0]), "+v"(xh[0]));
#pragma unroll
                for (int b = 0; b < NB; ++b) fy[b] = (short8_t){yl[b][0], yl[b][1], yl[b][2], yl[b][3], yh[b][0], yh[b][1], yh[b][2], yh[b][3]};
            } else if (a == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xl[1]), "+v"(xh[1]));
            else if (a == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xl[2]), "+v"(xh[2]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xl[3]), "+v"(xh[3]));
            const short8_t fxa = (short8_t){xl[a][0], xl[a][1], xl[a][2], xl[a][3], xh[a][0], xh[a][1], xh[a][2], xh[a][3]};
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fxa), __builtin_bit_cast(bf16x8_t, fy[b]),
                                                                    acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // keeps the next row's wait behind this row's MFMAs (asm waits are free to move otherwise)
        }
        if (do_bias) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, fy[b]),
                                                                  accb[b], 0, 0, 0);
        }
    }
}

template <int NST, int NWV>
__global__ __launch_bounds__(NWV * 64) void gemm_tn_fast_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X,
                                                          float* __restrict__ out, float* __restrict__ bias_part, int M, int N, int K,
                                                          int m_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NB = 16 / NWV;
    // XCD-aware order: all output tiles of one split read the same rows of dY and X, so a split's tiles run on ONE XCD
    // (one L2 fetch per row block instead of one per XCD; measured 361 -> ~1x the operand bytes per launch)
    const int tiles_n = (N + WG_TILE - 1) / WG_TILE, tiles = tiles_n * ((K + WG_TILE - 1) / WG_TILE);
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = work % tiles, split = work / tiles;
    const int n_blk = (tile % tiles_n) * WG_TILE, k_blk = (tile / tiles_n) * WG_TILE;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wk = wave & 1, wn = wave >> 1, lg = lane >> 4;
    f32x4_t acc[4][NB], accb[NB];
    const bool do_bias_tile = bias_part != nullptr && k_blk == 0;
    tn_fast_mainloop<NST, NWV>(smem, dY, X, M, N, K, n_blk, k_blk, m_begin, m_end, do_bias_tile, acc, accb);
    if (do_bias_tile && wk == 0 && lg == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = n_blk + wn * (NB * 16) + b * 16 + (lane & 15);
            if (n < N) bias_part[(size_t)split * N + n] = accb[b][0];
        }
    }
    float* o = out + (size_t)split * N * K;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n_blk + wn * (NB * 16) + b * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int k0 = k_blk + wk * 64 + a * 16 + lg * 4;
            if (k0 >= K) continue;
            *reinterpret_cast<float4*>(o + (size_t)n * K + k0) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Grouped weight gradients: up to STAR_WGRAD_GROUP_MAX products dW_j = dY_j^T X_j in ONE launch (a transformer block's four
// Linears are 108 output tiles of 128 x 128 at D = 384: with a common split-M of 2-3 they fill the chip, where each product
// launched alone needs 8-28 splits and as many float32 slabs).  The split-M partial tiles are folded IN the kernel: every
// workgroup publishes its tile (write-through stores, agent-scope release), takes a ticket on the tile's counter, and the
// workgroup that draws the last ticket adds the S partial tiles in split order 0..S-1 -- a fixed order, whoever arrives last,
// so the gradients stay bitwise reproducible without atomics on the data -- and writes dW / db.  The counters return to zero.
// Work order: split-major, then job, then the tile walk with the SHORTER tile dimension fastest; xcd_remap gives every XCD a
// contiguous run of it, i.e. a near-rectangular set of tiles of one split, whose dY / X row blocks its L2 fetches once.
#define TN_GROUP_TABLE 768
#define TN_GROUP_EMPTY 0xffffffffu
struct TnGroupArgs {
    const bf16_t* dY[STAR_WGRAD_GROUP_MAX];
    const bf16_t* X[STAR_WGRAD_GROUP_MAX];
    float* dW[STAR_WGRAD_GROUP_MAX];
    float* db[STAR_WGRAD_GROUP_MAX];
    int M[STAR_WGRAD_GROUP_MAX], N[STAR_WGRAD_GROUP_MAX], K[STAR_WGRAD_GROUP_MAX], mper[STAR_WGRAD_GROUP_MAX], S[STAR_WGRAD_GROUP_MAX];
    int tiles_n[STAR_WGRAD_GROUP_MAX], tiles_k[STAR_WGRAD_GROUP_MAX], bias0[STAR_WGRAD_GROUP_MAX], tap[STAR_WGRAD_GROUP_MAX];
    int tile0[STAR_WGRAD_GROUP_MAX + 1];     // first tile (= ticket counter) of job j
    int work0[STAR_WGRAD_GROUP_MAX + 1];     // first work item (= partial-tile slot) of job j: tiles x splits, job-major, split, tile
    int njobs, bias_ld;
    int table_n;       // > 0: launch index -> work item through `table` (host-planned XCD placement), else the plain job-major order
    unsigned table[TN_GROUP_TABLE];      // job << 24 | split << 16 | tile, or TN_GROUP_EMPTY (the workgroup exits at once)
    float* slabs;      // [work item][TK x TN] as the accumulators lie
    float* bpart;      // [split][bias_ld]
    int* counters;     // [tiles], zero between launches
};
#define TN_GROUP_COUNTERS 1024   // counter words at the head of the workspace (4 KiB)

// write-through stores (sc0 sc1: nothing stays dirty in the XCD's L2) as buffer-store builtins, NOT inline asm: a 16-byte VMEM store
// followed by a VALU write of its data registers needs two wait states that the compiler only inserts for stores it can see (an asm
// form lost one accumulator of wave 0 exactly that way)
#define STAR_CPOL_SC0_SC1 17   // cache-policy bits of the buffer builtins on gfx940+: 1 = sc0, 2 = nt, 16 = sc1
__device__ __forceinline__ void st_wt16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs, byte_off, 0, STAR_CPOL_SC0_SC1);
}
__device__ __forceinline__ void st_wt4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, byte_off, 0, STAR_CPOL_SC0_SC1);
}

// Main loop of the grouped kernel: 8 waves as WKW (k) x 8/WKW (n); a wave owns AK x BN blocks of 16 x 16, so the tile is
// TK = 16 AK WKW by TN = 16 BN (8/WKW).  What bounds this kernel is the rate at which a CU takes operand bytes in (LDS-DMA, ~60 GB/s per
// CU measured with the MFMAs and fragment reads compiled out: tools/diag_wgroup.sh), so the tile is as large as the accumulators
// allow: 384 x 192 (encoder widths) moves 0.75 of the bytes per flop of 384 x 128, 256 x 256 (decoder widths) half those of 128 x 128.
// A stage (32 rows of m) is a set of 64-column PANELS [32 rows][128 B], TN/64 of dY then TK/64 of X; 32-byte slots are XOR-swizzled by
// row bits 1 and 3, which makes the 32 lanes of a ds_read_b64_tr_b16 group (rows r0..r0+3 and r0+8..r0+11, four 8-byte pieces each)
// hit 64 different banks.  An LDS-DMA piece is 8 rows x 128 B; wave w fills rows 8 (w & 3).. of the panels of its parity
// (w >> 2), so the early half (waves 0-3) takes ceil(NP / 2) pieces per stage and the late half floor(NP / 2).
// Rows r0 and r0 + 4 of a fragment share their swizzle, so the second read of a fragment is the first plus offset:512.
#ifndef TN_DIAG
#define TN_DIAG 0       // diagnostic builds (tools/diag_wgroup.sh): 1 no LDS-DMA in the loop, 2 no MFMAs, 4 no fragment reads, 8 no barriers,
#endif                  // 16 LDS-DMA source columns unswizzled, 32 the halves take alternate panels instead of contiguous runs
__device__ __forceinline__ int tp_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
__device__ __forceinline__ int tp_off(int row, int slot) { return row * 128 + ((slot ^ tp_swz(row)) << 5); }

template <int NST, int WKW, int AK, int BN>
__device__ __forceinline__ void tn_wide_mainloop(unsigned char* smem, const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, int M, int N,
                                                 int K, int n_blk, int k_blk, int m_begin, int m_end, bool do_bias_tile, int tap,
                                                 f32x4_t (&acc)[AK][BN], f32x4_t& accb) {
    constexpr int WNW = 8 / WKW, TK = 16 * AK * WKW, TN = 16 * BN * WNW, PY = TN / 64, PX = TK / 64, NP = PY + PX;
    constexpr int PANEL = 32 * 128, STAGE = NP * PANEL, NQ0 = (NP + 1) / 2, NQ1 = NP / 2;
    static_assert(TK % 64 == 0 && TN % 64 == 0, "tile sides are multiples of the 64-column panel");
    static_assert(NST >= 3 && NQ0 <= 31 && BN <= 16, "ring depth, counted waits, one bias accumulator");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WKW, wn = wave / WKW;
    const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
    const bool late = wave >= 4;

    __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dY), 0, (unsigned)((long long)M * N * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (unsigned)((long long)M * K * 2), 0x00020000);
    // piece q of this wave: panel q (waves 0-3) or NQ0 + q (waves 4-7).  Only the offsets of panel 0 of each operand are kept (an
    // array of per-piece offsets indexed through the two halves' code paths ends up in scratch memory, and its reloads wait on vmcnt,
    // i.e. on every LDS-DMA piece in flight)
    const int dr = (wave & 3) * 8 + (lane >> 3);
    const int dpc = lane & 7;
    const int dcol = (TN_DIAG & 16) ? dpc * 8 : (((dpc >> 1) ^ tp_swz(dr)) << 4) + ((dpc & 1) << 3);
    const int ncol = n_blk + dcol, kcol = k_blk + dcol;
    const unsigned offY = (unsigned)(((long long)(m_begin + dr) * N + ncol) * 2);
    const unsigned offX = (unsigned)(((long long)(m_begin + dr) * K + kcol) * 2);
    const unsigned stepY = 32u * N * 2, stepX = 32u * K * 2;
    // tap mode (one tap of a 3 x 3 convolution over channels-last images): the X row of pixel m is pixel (y + dy, x + dx), zeros outside
    const bool tapped = tap != 0;
    const int tdy = (tap & 3) - 1, tdx = ((tap >> 2) & 3) - 1, lw = (tap >> 4) & 31, lh = (tap >> 9) & 31;
    const int tshift = tdy * (1 << lw) + tdx;
    auto issue_half = [&](auto H, int st_) {
        constexpr int h = decltype(H)::value;
        unsigned char* sb = smem + (st_ % NST) * STAGE + (wave & 3) * 1024;
        const unsigned sy = st_ * stepY;
        unsigned xo = offX + st_ * stepX;
        if (tapped) {
            const int m = m_begin + st_ * 32 + dr;
            const unsigned px = (unsigned)((m & ((1 << lw) - 1)) + tdx), py = (unsigned)(((m >> lw) & ((1 << lh) - 1)) + tdy);
            const bool ok = m < M && px < (1u << lw) && py < (1u << lh);
            xo = ok ? (unsigned)(((long long)(m + tshift) * K + kcol) * 2) : 0x80000000u;
        }
#pragma unroll
        for (int q = 0; q < NQ0; ++q) {
            const int pnl = (TN_DIAG & 32) ? 2 * q + h : q + h * NQ0;
            if (pnl >= NP) continue;
            if (pnl < PY) {
                const unsigned o = ncol + pnl * 64 < N ? offY + (unsigned)(pnl * 128) + sy : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (__attribute__((address_space(3))) void*)(sb + pnl * PANEL), 16, o, 0, 0, 0);
            } else {
                const unsigned o = (kcol + (pnl - PY) * 64 < K && xo < 0x80000000u) ? xo + (unsigned)((pnl - PY) * 128) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(sb + pnl * PANEL), 16, o, 0, 0, 0);
            }
        }
    };
    auto issue = [&](int st_) {
        if (late) issue_half(std::integral_constant<int, 1>{}, st_);
        else issue_half(std::integral_constant<int, 0>{}, st_);
    };
    // wait until at most k stages of this wave's pieces are in flight
    constexpr int PFD = NST - 1;        // stage s + PFD is issued while stage s is read (into the slot of stage s - 1)
    static_assert(PFD >= 2 && PFD <= 3 && 2 * NQ0 <= 63, "counted waits");
    auto wait_stages = [&](int k) {
        if (late) {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ1) : "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ0) : "memory");
            else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ0) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    accb = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < BN; ++b)
#pragma unroll
        for (int a = 0; a < AK; ++a) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = do_bias_tile && wk == 0;
    const int r0 = lg * 8 + lq;
    int ax[AK], ay[BN];
#pragma unroll
    for (int a = 0; a < AK; ++a) {
        const int cb = wk * AK + a;
        ax[a] = (PY + (cb >> 2)) * PANEL + tp_off(r0, cb & 3) + lp * 8;
    }
#pragma unroll
    for (int b = 0; b < BN; ++b) {
        const int cb = wn * BN + b;
        ay[b] = (cb >> 2) * PANEL + tp_off(r0, cb & 3) + lp * 8;
    }
    const unsigned lds_base = lds_u32(smem);
    const int nsteps = (m_end - m_begin + 31) / 32;
    // PING-PONG: the two waves of a SIMD (w and w + 4) run half a step apart.  A step is two intervals between workgroup barriers;
    // in each, one half of the waves runs its MFMAs while the other half issues its LDS-DMA pieces for two steps ahead and reads its
    // fragments (an LDS-DMA piece costs its issuing wave 60-190 cycles; in step, both waves of a SIMD paid that together and then
    // queued on the matrix pipe together).
    //   interval 2s    : waves 0-3 load(s)      waves 4-7 compute(s-1)
    //   interval 2s + 1: waves 0-3 compute(s)   waves 4-7 load(s)
    // load(s) = issue stage s + PFD, read the fragments of stage s, wait for them (so that no read is in flight when the slot is refilled
    // two intervals later); stage s + 1 has landed (counted vmcnt) before the barrier that opens interval 2s + 2.  Three ring slots.
    for (int s0 = 0; s0 < PFD && s0 < nsteps; ++s0) issue(s0);
    wait_stages(min(PFD, nsteps) - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    for (int st = 0; st < nsteps; ++st) {
        // ---- load interval
        if (!(TN_DIAG & 1) && st + PFD < nsteps) issue(st + PFD);
        const unsigned sb = lds_base + (st % NST) * STAGE;
        short4_t xl[AK], xh[AK], yl[BN], yh[BN];
        if (TN_DIAG & 4) {
#pragma unroll
            for (int b = 0; b < BN; ++b) asm volatile("" : "=v"(yl[b]), "=v"(yh[b]));
#pragma unroll
            for (int a = 0; a < AK; ++a) asm volatile("" : "=v"(xl[a]), "=v"(xh[a]));
        } else {
#pragma unroll
            for (int b = 0; b < BN; ++b) {
                yl[b] = tr16<0>(sb + ay[b]);
                yh[b] = tr16<512>(sb + ay[b]);
            }
#pragma unroll
            for (int a = 0; a < AK; ++a) {
                xl[a] = tr16<0>(sb + ax[a]);
                xh[a] = tr16<512>(sb + ax[a]);
            }
        }
        // stages st + 1 .. min(st + PFD, nsteps - 1) are in flight; st + 1 must have landed before the early half reads it
        if (late) wait_stages(min(st + PFD, nsteps - 1) - (st + 1));       // ... one barrier from here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < BN; ++b) asm volatile("" : "+v"(yl[b]), "+v"(yh[b]));
#pragma unroll
        for (int a = 0; a < AK; ++a) asm volatile("" : "+v"(xl[a]), "+v"(xh[a]));
        if (!(TN_DIAG & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute interval
        short8_t fy[BN];
#pragma unroll
        for (int b = 0; b < BN; ++b) fy[b] = (short8_t){yl[b][0], yl[b][1], yl[b][2], yl[b][3], yh[b][0], yh[b][1], yh[b][2], yh[b][3]};
        if (TN_DIAG & 2) {
#pragma unroll
            for (int a = 0; a < AK; ++a) asm volatile("" :: "v"(xl[a]), "v"(xh[a]), "v"(fy[a % BN]));
        } else {
#pragma unroll
            for (int a = 0; a < AK; ++a) {
                const short8_t fxa = (short8_t){xl[a][0], xl[a][1], xl[a][2], xl[a][3], xh[a][0], xh[a][1], xh[a][2], xh[a][3]};
#pragma unroll
                for (int b = 0; b < BN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fxa), __builtin_bit_cast(bf16x8_t, fy[b]),
                                                                        acc[a][b], 0, 0, 0);
            }
        }
        if (do_bias && !(TN_DIAG & 2)) {
            // column sums of dY: the A operand of block b is ones in ROW b only, so the one accumulator collects block b's sums in its row b
            // (lane group b >> 2, register b & 3)
#pragma unroll
            for (int b = 0; b < BN; ++b) {
                const short sel = (lane & 15) == b ? (short)0x3F80 : (short)0;
                const short8_t rowb = (short8_t){sel, sel, sel, sel, sel, sel, sel, sel};
                accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, rowb), __builtin_bit_cast(bf16x8_t, fy[b]), accb, 0, 0, 0);
            }
        }
        if (!late) wait_stages(min(st + PFD, nsteps - 1) - (st + 1));      // the early half reaches the barrier that opens stage st + 1's reads from here
        __builtin_amdgcn_sched_barrier(0);
        if (!(TN_DIAG & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (!late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
}

template <int NST, int WKW, int AK, int BN>
__global__ __launch_bounds__(512) void gemm_tn_group_kernel(const TnGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_last;
    constexpr int WNW = 8 / WKW, TK = 16 * AK * WKW, TN = 16 * BN * WNW;
    int j = 0, split, t;
    if (g.table_n > 0) {
        // host-planned placement: the tiles of one (job, split) read the same rows of dY and X, so the host deals whole (job, split)
        // groups to the XCDs (workgroups b and b + 8 share one) -- each operand row block is fetched into ONE L2
        const unsigned e = g.table[blockIdx.x];
        if (e == TN_GROUP_EMPTY) return;
        j = (int)(e >> 24);
        split = (int)((e >> 16) & 0xff);
        t = (int)(e & 0xffff);
    } else {
        const int work = xcd_remap(blockIdx.x, gridDim.x);
        while (j + 1 < g.njobs && work >= g.work0[j + 1]) ++j;
        const int ntile = g.tiles_n[j] * g.tiles_k[j];
        split = (work - g.work0[j]) / ntile;
        t = (work - g.work0[j]) - split * ntile;
    }
    const int tn = g.tiles_n[j], tk = g.tiles_k[j];
    int nt, kt;
    if (tk <= tn) { kt = t % tk; nt = t / tk; } else { nt = t % tn; kt = t / tn; }
    const int M = g.M[j], N = g.N[j], K = g.K[j];
    const int n_blk = nt * TN, k_blk = kt * TK;
    const int m_begin = split * g.mper[j], m_end = min(M, m_begin + g.mper[j]);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wk = wave % WKW, wn = wave / WKW, lg = lane >> 4;
    float* const db = g.db[j];
    const bool do_bias_tile = db != nullptr && k_blk == 0;
    f32x4_t acc[AK][BN], accb;
    tn_wide_mainloop<NST, WKW, AK, BN>(smem, g.dY[j], g.X[j], M, N, K, n_blk, k_blk, m_begin, m_end, do_bias_tile, g.tap[j], acc, accb);
    // bias: lane group lg of the wk == 0 waves holds the column sums of blocks 4 lg .. 4 lg + 3 (register j = block & 3), column lane & 15
    const bool bias_lane = do_bias_tile && wk == 0 && lg * 4 < BN;
    const int bias_n0 = n_blk + wn * (BN * 16) + lg * 64 + (lane & 15);     // + 16 j
    const int S = g.S[j];
    const int ci = g.tile0[j] + t;
    if (S > 1) {
        float* slab = g.slabs + ((size_t)g.work0[j] + (size_t)t * S + split) * (TK * TN);
        __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(slab, 0, TK * TN * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < AK; ++a)
#pragma unroll
            for (int b = 0; b < BN; ++b) st_wt16(rsS, (unsigned)((((wave * AK + a) * BN + b) * 64 + lane) << 4), acc[a][b]);
        if (bias_lane) {
            __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(g.bpart + (size_t)split * g.bias_ld + g.bias0[j], 0, (unsigned)(N * 4), 0x00020000);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                if (lg * 4 + jb < BN) st_wt4(rsB, (unsigned)((bias_n0 + 16 * jb) << 2), accb[jb]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int old = __hip_atomic_fetch_add(g.counters + ci, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == S - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.counters + ci, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        // the last ticket: fold the S partial tiles in split order, all of them read back (this workgroup's own one included).  One
        // block row of the tile at a time -- sum over the splits, store -- so that the fold needs BN accumulators, not AK x BN
    }
    float* o = g.dW[j];
    auto store_row = [&](int a, const f32x4_t (&v)[BN]) {
        const int k0 = k_blk + wk * (AK * 16) + a * 16 + lg * 4;
        if (k0 >= K) return;
#pragma unroll
        for (int b = 0; b < BN; ++b) {
            const int n = n_blk + wn * (BN * 16) + b * 16 + (lane & 15);
            if (n < N) *reinterpret_cast<float4*>(o + (size_t)n * K + k0) = make_float4(v[b][0], v[b][1], v[b][2], v[b][3]);
        }
    };
    if (S > 1) {
        const float* slab0 = g.slabs + ((size_t)g.work0[j] + (size_t)t * S) * (TK * TN);
#pragma unroll 1
        for (int a = 0; a < AK; ++a) {
            // eight partial tiles' worth of this row in flight at a time (the loads are HBM-latency-bound: one round trip per chunk)
            constexpr int SC = 8;
            f32x4_t sum[BN];
            for (int s0 = 0; s0 < S; s0 += SC) {
                f32x4_t v[SC][BN];
#pragma unroll
                for (int c = 0; c < SC; ++c) {
                    const float* sl = slab0 + (size_t)min(s0 + c, S - 1) * (TK * TN);
#pragma unroll
                    for (int b = 0; b < BN; ++b)
                        v[c][b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(sl + ((((wave * AK + a) * BN + b) * 64 + lane) << 2)));
                }
#pragma unroll
                for (int c = 0; c < SC; ++c) {
                    if (s0 + c >= S) break;
#pragma unroll
                    for (int b = 0; b < BN; ++b) sum[b] = (s0 + c) == 0 ? v[c][b] : sum[b] + v[c][b];
                }
            }
            store_row(a, sum);
        }
        if (bias_lane) {
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    if (lg * 4 + jb >= BN || bias_n0 + 16 * jb >= N) continue;
                    const float vb = __builtin_nontemporal_load(g.bpart + (size_t)s * g.bias_ld + g.bias0[j] + bias_n0 + 16 * jb);
                    accb[jb] = s == 0 ? vb : accb[jb] + vb;
                }
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < AK; ++a) store_row(a, acc[a]);
    }
    if (bias_lane) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int n = bias_n0 + 16 * jb;
            if (lg * 4 + jb < BN && n < N) db[n] = accb[jb];
        }
    }
}

// fp32 TN path: simple LDS-tiled kernel on v_mfma_f32_16x16x4_f32; one value per lane per operand.
// Tile 64(k) x 64(n), 4 waves each 32x32, reduction step 16 rows.
__global__ __launch_bounds__(GT) void gemm_tn_f32_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                         const int32_t* __restrict__ x_rows, float* __restrict__ out,
                                                         int M, int N, int K, int m_per_split) {
    __shared__ float sY[16][64 + 4], sX[16][64 + 4];
    const int tiles_n = (N + 63) / 64;
    const int n_blk = (blockIdx.x % tiles_n) * 64, k_blk = (blockIdx.x / tiles_n) * 64;
    const int split = blockIdx.y;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wn = wave >> 1;
    const int ld_row = tid >> 4, ld_c4 = (tid & 15) << 2;  // 16 rows x 16 float4
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int m0 = m_begin; m0 < m_end; m0 += 16) {
        const int m = m0 + ld_row;
        float4 vy = make_float4(0, 0, 0, 0), vx = make_float4(0, 0, 0, 0);
        if (m < m_end) {
            if (n_blk + ld_c4 < N) vy = *reinterpret_cast<const float4*>(dY + (size_t)m * N + n_blk + ld_c4);
            if (k_blk + ld_c4 < K) {
                const long long xr = x_rows ? (long long)x_rows[m] : (long long)m;
                if (xr >= 0) vx = *reinterpret_cast<const float4*>(X + (size_t)xr * K + k_blk + ld_c4);
            }
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&sY[ld_row][ld_c4]) = vy;
        *reinterpret_cast<float4*>(&sX[ld_row][ld_c4]) = vx;
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {  // 4 rows of m per MFMA: lane group lg takes row s4*4 + lg
            const int r = s4 * 4 + (lane >> 4);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(sX[r][wk * 32 + a * 16 + (lane & 15)],
                                                                     sY[r][wn * 32 + b * 16 + (lane & 15)], acc[a][b], 0, 0, 0);
        }
    }
    float* o = out + (size_t)split * N * K;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n_blk + wn * 32 + b * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int k0 = k_blk + wk * 32 + a * 16 + (lane >> 4) * 4;
            if (k0 >= K) continue;
            *reinterpret_cast<float4*>(o + (size_t)n * K + k0) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
    }
}

// one launch folds the split-M slabs into dW and the per-split bias partials into db, in a fixed order
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int64_t n4, int S,
                                    const float* __restrict__ bpart, float* __restrict__ db, int N, int SB) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dW != nullptr && S > 1) {
        for (int64_t i = t0; i < n4; i += stride) {
            // eight slabs requested at once (one memory latency per eight), folded in slab order: same sum as a plain loop
            const float4* src = reinterpret_cast<const float4*>(slabs) + i;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < S; s += 8) {
                float4 b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = s + u < S ? src[(int64_t)(s + u) * n4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
            }
            reinterpret_cast<float4*>(dW)[i] = a;
        }
    }
    if (db != nullptr) {
        for (int64_t c = t0; c < N; c += stride) {
            float a = 0.f;
            for (int s = 0; s < SB; ++s) a += bpart[(int64_t)s * N + c];
            db[c] = a;
        }
    }
}

// column sums of dY (bias gradient): block = 64 columns x 4 row lanes, grid.y row splits, then reduce
template <typename T>
__global__ __launch_bounds__(GT) void colsum_kernel(const T* __restrict__ dY, float* __restrict__ part, int M, int N, int rows_per) {
    __shared__ float s[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float a = 0.f;
    if (c < N)
        for (int m = m0 + rl; m < m1; m += 4) a += Act<T>::ld(dY + (size_t)m * N + c);
    s[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}
static void wgrad_plan(int M, int N, int K, int dt, int* S, int* m_per, int* SC, int* rows_per) {
    const int tile = dt == STAR_BF16 ? WG_TILE : 64;
    const int step = dt == STAR_BF16 ? WG_BKM : 16;
    const int tiles = cdiv(N, tile) * cdiv(K, tile);
    // about one block per CU (256 CUs): alone on the chip twice as many blocks run a weight gradient ~5 % faster, but inside
    // the training step -- where it shares the CUs with the main backward chain -- the smaller launch is the better neighbour
    // (12.97 -> 12.89 ms/step; 150 % of the old target: 13.14) and halves the slab traffic (S x N x K x 8 bytes)
    // (re-measured with the 8-wave kernel of round 2: 75 % of this target 11.99-12.04 ms, 100 % 11.95-11.99, 150 % 12.06-12.09, 50 % 12.36)
    const int target = tiles <= 9 ? 128 : (tiles <= 16 ? 192 : 256);
    int s = cdiv(target, tiles);
    const int max_s = cdiv(M, 8 * step);       // at least 8 reduction steps per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    int mp = cdiv(cdiv(M, s), step) * step;
    s = cdiv(M, mp);
    *S = s; *m_per = mp;
    int sc = cdiv(M, 512);
    if (sc > 64) sc = 64;
    if (sc < 1) sc = 1;
    *SC = sc; *rows_per = cdiv(M, sc);
}

extern "C" int64_t star_gemm_wgrad_ws(int M, int N, int K) {
    int S, mp, SC, rp, S2, mp2;
    wgrad_plan(M, N, K, STAR_BF16, &S, &mp, &SC, &rp);
    wgrad_plan(M, N, K, STAR_F32, &S2, &mp2, &SC, &rp);
    if (S2 > S) S = S2;
    return (int64_t)S * N * K + (int64_t)(SC > S ? SC : S) * N + 16;
}

extern "C" int star_gemm_wgrad(const void* dY, const void* X, const int32_t* x_rows, float* dW, float* db, float* ws,
                               int M, int N, int K, int dt, void* stream) {
    STAR_CHECK(dY && X && ws && (dW || db), "star_gemm_wgrad: null pointer");
    STAR_CHECK(M > 0 && N > 0 && K > 0, "star_gemm_wgrad: bad shape M=%d N=%d K=%d", M, N, K);
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32, "star_gemm_wgrad: unknown dtype %d", dt);
    const int epc = dt == STAR_BF16 ? 8 : 4;
    STAR_CHECK(N % epc == 0 && K % epc == 0, "star_gemm_wgrad: N=%d and K=%d must be multiples of %d", N, K, epc);
    hipStream_t s = (hipStream_t)stream;
    int S, mp, SC, rp;
    wgrad_plan(M, N, K, dt, &S, &mp, &SC, &rp);
    float* part = ws + (size_t)S * N * K;
    int SB = 0;  // number of bias partial rows
    if (dW) {
        float* target = S == 1 ? dW : ws;
        if (dt == STAR_BF16) {
            dim3 grid(cdiv(N, WG_TILE) * cdiv(K, WG_TILE), S);
            const bool fast = x_rows == nullptr && (long long)M * N * 2 < 0x7fffffffLL && (long long)M * K * 2 < 0x7fffffffLL &&
                              ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0;
            static const char* wenv = getenv("STAR_AMD_WGRAD_WAVES");
            // 8 waves (2 x 4, a wave owns 64 x 32 of the tile, ~2 waves per SIMD hide each other's LDS / barrier latency): alone 10-20 %
            // faster than 4 waves (fc1 110 -> 97.5 us, proj 59.5 -> 48.8, decoder fc1 71.5 -> 58); inside the step the launches get
            // 9 % shorter and the step time stays (11.66 ms both ways).  STAR_AMD_WGRAD_WAVES=4 selects the 4-wave kernel (A/B).
            if (fast && !(wenv && wenv[0] == '4'))
                hipLaunchKernelGGL((gemm_tn_fast_kernel<3, 8>), dim3(grid.x * grid.y), dim3(512), 3 * 2 * 32 * 256, s, (const bf16_t*)dY, (const bf16_t*)X, target,
                                   db ? part : (float*)nullptr, M, N, K, mp);
            else if (fast)
                hipLaunchKernelGGL((gemm_tn_fast_kernel<3, 4>), dim3(grid.x * grid.y), dim3(GT), 3 * 2 * 32 * 256, s, (const bf16_t*)dY, (const bf16_t*)X, target,
                                   db ? part : (float*)nullptr, M, N, K, mp);
            else
                hipLaunchKernelGGL(gemm_tn_bf16_kernel, grid, dim3(GT), 4 * WG_BKM * 256, s, (const bf16_t*)dY, (const bf16_t*)X, x_rows, target,
                                   db ? part : (float*)nullptr, M, N, K, mp);
            SB = db ? S : 0;
        } else {
            dim3 grid(cdiv(N, 64) * cdiv(K, 64), S);
            hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, dim3(GT), 0, s, (const float*)dY, (const float*)X, x_rows, target, M, N, K, mp);
        }
        STAR_LAUNCH_CHECK("star_gemm_wgrad");
    }
    if (db && SB == 0) {  // fp32 path, or bias only: stand-alone column sums
        if (dt == STAR_BF16)
            hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(cdiv(N, 64), SC), dim3(GT), 0, s, (const bf16_t*)dY, part, M, N, rp);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(N, 64), SC), dim3(GT), 0, s, (const float*)dY, part, M, N, rp);
        STAR_LAUNCH_CHECK("star_gemm_wgrad(colsum)");
        SB = SC;
    }
    if ((dW && S > 1) || db) {
        const int64_t n4 = (int64_t)N * K / 4;
        int gb = cdiv(dW && S > 1 ? n4 : N, 256);
        if (gb > 2048) gb = 2048;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gb), dim3(256), 0, s, ws, (dW && S > 1) ? dW : (float*)nullptr, n4, S, part, db, N, SB);
        STAR_LAUNCH_CHECK("star_gemm_wgrad(reduce)");
    }
    return 0;
}

// ---- grouped weight gradients: plan + launch -------------------------------------------------------------------
// tile shapes of the grouped kernel: (waves along k, k blocks, n blocks per wave) -> TK x TN
struct TnShape { int wkw, ak, bn, tk, tn; };
static const TnShape TN_SHAPES[] = {
    {2, 4, 2, 128, 128},     // 0: any shape
    {4, 6, 6, 384, 192},     // 1: K a multiple of 384 (encoder widths 384 / 1536): 144 accumulator registers per lane
    {2, 8, 4, 256, 256},     // 2: K a multiple of 256 (decoder widths 256 / 1024): 128
};

// Tile shape and split count are functions of the JOB alone (never of what it shares a launch with), so a gradient's bits do not
// depend on how the caller groups its products.
static int wgrad_job_shape(const StarWgradJob& q) {
    static const char* env = getenv("STAR_AMD_WGRAD_TILE");     // 0..2 forces a tile shape where it divides K (A/B)
    int want = -1;
    if (env && env[0] >= '0' && env[0] <= '2') want = env[0] - '0';
    const bool ok1 = q.K % 384 == 0 && q.N >= 192, ok2 = q.K % 256 == 0 && q.N >= 256;
    if (want == 0 || (want == 1 && ok1) || (want == 2 && ok2)) return want;
    return ok1 ? 1 : (ok2 ? 2 : 0);
}
static int wgrad_job_splits(const StarWgradJob& q, int sh) {
    // Row splits per job, by tile shape (STAR_AMD_WGRAD_SPLITS="s0,s1,s2" overrides, A/B), whatever the job's size -- equal splits
    // give every workgroup of a launch the same number of rows, so a block's products finish together.  What the step pays for is
    // CU-time (a workgroup owns its CU: 8 waves of ~220 registers), so the split is the smallest that still gives a launch a few
    // workgroups per XCD: an encoder block is 24 tiles of 384 x 192 (x 8 = 192 workgroups), a decoder block 12 of 256 x 256 (x 8 = 96);
    // the partial tiles the last workgroup of a tile adds up cost S x 4 bytes per weight, ~5 % of the operand bytes at S = 8.
    static int want[3] = {-1, -1, -1};
    if (want[0] < 0) {
        int w[3] = {5, 8, 8};
        if (const char* e = getenv("STAR_AMD_WGRAD_SPLITS")) {
            int v[3] = {0, 0, 0};
            const int n = sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
            for (int i = 0; i < 3; ++i) { const int x = i < n ? v[i] : (n >= 1 ? v[0] : 0); if (x > 0) w[i] = x; }
        }
        want[2] = w[2]; want[1] = w[1]; want[0] = w[0];
    }
    int S = want[sh];
    // long products (a convolution's pixels): ~256 reduction stages per split up to 32 splits, ~512 beyond (up to 64)
    if (q.M / 8192 > S) S = q.M / 8192 < 32 ? q.M / 8192 : 32;
    if (q.M / 16384 > S) S = q.M / 16384;
    const int max_s = q.M / (8 * 32);                          // at least 8 reduction stages per split
    if (S > max_s) S = max_s;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return S;
}

// plan of the launch that takes the jobs of shape `sh` (jobs of the other shape are skipped); returns the number of jobs taken
static int wgrad_group_plan(const StarWgradJob* jobs, int njobs, int sh, TnGroupArgs* a, int64_t* ws_floats) {
    const TnShape& T = TN_SHAPES[sh];
    int n = 0, tiles = 0, work = 0, bias_ld = 0, max_S = 1;
    for (int j = 0; j < njobs; ++j) {
        const StarWgradJob& q = jobs[j];
        if (wgrad_job_shape(q) != sh) continue;
        a->dY[n] = (const bf16_t*)q.dY; a->X[n] = (const bf16_t*)q.X; a->dW[n] = q.dW; a->db[n] = q.db;
        a->M[n] = q.M; a->N[n] = q.N; a->K[n] = q.K; a->tap[n] = q.tap;
        a->tiles_n[n] = cdiv(q.N, T.tn); a->tiles_k[n] = cdiv(q.K, T.tk);
        const int S = wgrad_job_splits(q, sh);
        a->S[n] = S;
        a->mper[n] = cdiv(cdiv(q.M, S), 32) * 32;      // the last split may come out short or empty (a zero partial tile)
        a->tile0[n] = tiles;
        a->work0[n] = work;
        a->bias0[n] = bias_ld;
        tiles += a->tiles_n[n] * a->tiles_k[n];
        work += a->tiles_n[n] * a->tiles_k[n] * S;
        bias_ld += cdiv(q.N, 64) * 64;
        if (S > max_S) max_S = S;
        ++n;
    }
    for (int j = n; j <= STAR_WGRAD_GROUP_MAX; ++j) { a->tile0[j] = tiles; a->work0[j] = work; }
    a->njobs = n; a->bias_ld = bias_ld;
    *ws_floats = n == 0 ? 0 : (int64_t)max_S * bias_ld + (int64_t)work * T.tk * T.tn;
    // XCD placement: whole (job, split) groups, largest first, each to the XCD with the least tiles so far (LPT); slot i of XCD x is
    // launch index 8 i + x.  Falls back to the plain order when the padded grid does not fit the table.
    a->table_n = 0;
    static const char* tenv = getenv("STAR_AMD_WGRAD_PLACE");       // 0: plain job-major order through xcd_remap (A/B)
    if (n > 0 && !(tenv && tenv[0] == '0')) {
        struct Grp { int j, s, sz; };
        Grp grp[STAR_WGRAD_GROUP_MAX * 64];
        int ng = 0;
        for (int j = 0; j < n; ++j)
            for (int s2 = 0; s2 < a->S[j]; ++s2) grp[ng++] = {j, s2, a->tiles_n[j] * a->tiles_k[j]};
        for (int i = 1; i < ng; ++i) {          // insertion sort by size, descending (stable: equal sizes keep job / split order)
            const Grp key = grp[i];
            int q = i - 1;
            while (q >= 0 && grp[q].sz < key.sz) { grp[q + 1] = grp[q]; --q; }
            grp[q + 1] = key;
        }
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, owner[STAR_WGRAD_GROUP_MAX * 64];
        for (int i = 0; i < ng; ++i) {
            int best = 0;
            for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
            owner[i] = best;
            load[best] += grp[i].sz;
        }
        int longest = 0;
        for (int x = 0; x < 8; ++x) longest = load[x] > longest ? load[x] : longest;
        if (longest * 8 <= TN_GROUP_TABLE) {
            for (int i = 0; i < longest * 8; ++i) a->table[i] = TN_GROUP_EMPTY;
            int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < ng; ++i)
                for (int t = 0; t < grp[i].sz; ++t)
                    a->table[8 * fill[owner[i]]++ + owner[i]] = ((unsigned)grp[i].j << 24) | ((unsigned)grp[i].s << 16) | (unsigned)t;
            a->table_n = longest * 8;
        }
    }
    return tiles;
}

static int wgrad_group_check(const StarWgradJob* jobs, int njobs, int dt) {
    STAR_CHECK(jobs && njobs >= 1 && njobs <= STAR_WGRAD_GROUP_MAX, "star_gemm_wgrad_group: 1..%d jobs", STAR_WGRAD_GROUP_MAX);
    STAR_CHECK(dt == STAR_BF16, "star_gemm_wgrad_group: bf16 operands only (float32 products go through star_gemm_wgrad)");
    for (int j = 0; j < njobs; ++j) {
        const StarWgradJob& q = jobs[j];
        STAR_CHECK(q.dY && q.X && q.dW, "star_gemm_wgrad_group: job %d: null pointer", j);
        STAR_CHECK(q.M > 0 && q.N > 0 && q.K > 0 && q.N % 8 == 0 && q.K % 8 == 0, "star_gemm_wgrad_group: job %d: bad shape M=%d N=%d K=%d (N, K multiples of 8)", j, q.M,
                   q.N, q.K);
        STAR_CHECK((long long)q.M * q.N * 2 < 0x7fffffffLL && (long long)q.M * q.K * 2 < 0x7fffffffLL, "star_gemm_wgrad_group: job %d: operand over 2 GiB", j);
        if (q.tap != 0) {
            const int dy1 = q.tap & 3, dx1 = (q.tap >> 2) & 3, lw = (q.tap >> 4) & 31, lh = (q.tap >> 9) & 31;
            STAR_CHECK((q.tap & 0x40000000) && dy1 <= 2 && dx1 <= 2 && lw + lh <= 30 && q.M % (1 << (lw + lh)) == 0,
                       "star_gemm_wgrad_group: job %d: bad tap word 0x%x for M=%d (STAR_WGRAD_TAP; images of 2^%d x 2^%d pixels)", j, q.tap, q.M, lw, lh);
        }
        STAR_CHECK(((reinterpret_cast<uintptr_t>(q.dY) | reinterpret_cast<uintptr_t>(q.X) | reinterpret_cast<uintptr_t>(q.dW)) & 15) == 0,
                   "star_gemm_wgrad_group: job %d: operands must be 16-byte aligned", j);
    }
    return 0;
}

extern "C" int64_t star_gemm_wgrad_group_ws(const StarWgradJob* jobs, int njobs, int dt) {
    if (wgrad_group_check(jobs, njobs, dt) != 0) return -1;
    int64_t total = TN_GROUP_COUNTERS;
    for (int sh = 0; sh < 3; ++sh) {
        TnGroupArgs a;
        int64_t n = 0;
        const int tiles = wgrad_group_plan(jobs, njobs, sh, &a, &n);
        if (tiles > TN_GROUP_COUNTERS) { star_set_error(-1, "star_gemm_wgrad_group: %d output tiles, at most %d per call", tiles, TN_GROUP_COUNTERS); return -1; }
        total += (n + 3) / 4 * 4;
    }
    return total;
}

template <int NST, int WKW, int AK, int BN>
static void launch_tn_group(const TnGroupArgs& a, int work, hipStream_t s) {
    constexpr int LDS = NST * (16 * AK * WKW + 16 * BN * (8 / WKW)) * 64;       // 32 rows x 2 B per column and stage
    if (LDS > 48 * 1024)    // per launch: the attribute belongs to the current device, and the call is cheap
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_group_kernel<NST, WKW, AK, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL((gemm_tn_group_kernel<NST, WKW, AK, BN>), dim3(work), dim3(512), LDS, s, a);
}

extern "C" int star_gemm_wgrad_group(const StarWgradJob* jobs, int njobs, float* ws, int64_t ws_floats, int dt, void* stream) {
    if (int rc = wgrad_group_check(jobs, njobs, dt)) return rc;
    const int64_t need = star_gemm_wgrad_group_ws(jobs, njobs, dt);
    if (need < 0) return -1;
    STAR_CHECK(ws && ws_floats >= need, "star_gemm_wgrad_group: workspace of %lld floats, %lld needed", (long long)ws_floats, (long long)need);
    STAR_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "star_gemm_wgrad_group: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    static const char* nst_env = getenv("STAR_AMD_WGRAD_NST");     // ring depth (A/B): 4 stages (default: an encoder block alone 227 us
    const bool deep = !(nst_env && nst_env[0] == '3');              // against 233, a decoder block 161 against 175) or 3
    float* next = ws + TN_GROUP_COUNTERS;
    // one launch per tile shape present (a block's four products share one); the launches run one after the other on the stream
    // and use the ticket counters in turn, their partial tiles side by side
    for (int sh = 0; sh < 3; ++sh) {
        TnGroupArgs a;
        int64_t n = 0;
        wgrad_group_plan(jobs, njobs, sh, &a, &n);
        if (a.njobs == 0) continue;
        int max_S = 1;
        for (int j = 0; j < a.njobs; ++j) max_S = a.S[j] > max_S ? a.S[j] : max_S;
        a.counters = reinterpret_cast<int*>(ws);
        a.bpart = next;
        a.slabs = a.bpart + (size_t)max_S * a.bias_ld;
        next += (n + 3) / 4 * 4;
        const int work = a.table_n > 0 ? a.table_n : a.work0[a.njobs];
        switch (sh) {
            case 1: if (deep) launch_tn_group<4, 4, 6, 6>(a, work, s); else launch_tn_group<3, 4, 6, 6>(a, work, s); break;
            case 2: if (deep) launch_tn_group<4, 2, 8, 4>(a, work, s); else launch_tn_group<3, 2, 8, 4>(a, work, s); break;
            default: if (deep) launch_tn_group<4, 2, 4, 2>(a, work, s); else launch_tn_group<3, 2, 4, 2>(a, work, s); break;
        }
        STAR_LAUNCH_CHECK("star_gemm_wgrad_group");
    }
    return 0;
}
