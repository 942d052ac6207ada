// K3 + K2 in one launch: the Linear that closes a residual branch together with the LayerNorm that reads its result.
//   x_out[m, :] = res[m, :] + A[m, :] W^T + bias          (float32 residual stream)
//   h[m, :]     = LayerNorm(x_out[m, :]) * gamma + beta    (bf16, the next Linear's operand), mean[m], rstd[m]
// Reference ops replaced: timm Block.forward's `x = x + proj(...)` / `x = x + fc2(...)` followed by the next `norm2(x)` /
// `norm1(x)` / `self.norm(x)` (star/models/mae_base.py:279-282,308-312 build the blocks).
//
// Why one kernel: with N = D = 256..384 both Linears are HBM-bound on their epilogue (residual in, x_out out), and the
// LayerNorm behind them is a second pass that reads x_out again (63 MB at M = 40 960, D = 384) for 31 MB of output.  A tile
// that spans the WHOLE row (BN = N) has the finished rows in its accumulators, so the statistics and the normalised row cost
// no memory traffic beyond the bf16 store, and a launch goes.
//
// Tile: BM x N, waves WM x WN, each wave RB 16-row blocks x 64 columns on v_mfma_f32_16x16x32_bf16 (W fragment = A operand,
// activation fragment = B operand: a lane ends with 4-column runs of one row).  Operands stream by LDS-DMA through an NST-deep
// ring of 32-deep stages behind a counted vmcnt, one s_barrier per stage -- the scheme of gemm_nt_fast_kernel (gemm.hip); the
// stage's (BM + N) / 16 sixteen-row pieces are dealt to the waves round-robin.  Row statistics: lane partial -> the row's four
// lanes (DPP-free xor shuffles) -> the row's WN waves through a small LDS table, twice (mean, then the centred squares: the
// same two-pass form as ln_fwd_kernel).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 gl_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short gl_short8_t;
typedef __attribute__((ext_vector_type(4))) float gl_f32x4_t;

// LDS rows are 64 B; 16-byte chunk c of row r sits at c ^ S[(r>>2)&3], S = {0,2,3,1} (conflict-free 16x16x32 fragment reads)
__device__ __forceinline__ int gl_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }
// accumulator block a (0..3) of lane group lg holds output columns run_off .. +3 of the wave's 64-column block: a row's four
// lanes write one contiguous 64-byte segment per 16-byte bf16 store, and two adjacent 16-byte pieces per float32 store pair
__device__ __forceinline__ int gl_run_off(int lg, int a) { return (a >> 1) * 32 + lg * 8 + (a & 1) * 4; }

// The main loop: acc[a][b] += A-tile x W-panel over all of K.  Waves WM x WN; a wave owns RB 16-row blocks and
// CB 16-column blocks (CB * 16 adjacent columns).  `acc` comes in initialised (the bias).  Leaves the ring possibly still being
// read by slower waves: a caller that reuses it puts a barrier first.
template <int NST, int WM, int WN, int RB, int CB>
__device__ __forceinline__ void gl_mainloop(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, int M, int K, int m_blk,
                                            unsigned char* smem, gl_f32x4_t (&acc)[CB][RB]) {
    // pieces are dealt round-robin (piece p to wave p % NW): when they do not divide evenly the first NP % NW waves carry one more
    constexpr int NW = WM * WN, BM = WM * RB * 16, CW = CB * 16, N = WN * CW, NP = (BM + N) / 16, NDMA = (NP + NW - 1) / NW, NFULL = NP % NW,
                  STAGE = (BM + N) * 64, PF = NST - 1;
    static_assert(NDMA <= 6, "offset table");
    static_assert(PF >= 1 && PF <= 3 && PF * NDMA <= 63, "counted vmcnt");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lg = lane >> 4;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, 0x7fffffff, 0x00020000);
    const bool short_wave = NFULL != 0 && wave >= NFULL;      // this wave has NDMA - 1 pieces per stage
    unsigned off[6];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int p = j * NW + wave;               // piece = stage rows p*16 .. +15 (A rows first, then the W image)
        if (p >= NP) { off[j] = 0x7fffffffu; continue; }
        const int r = p * 16 + (lane >> 2);
        const int c = (lane & 3) ^ gl_swz(r);
        if (p < BM / 16) {
            const int m = m_blk + r;
            off[j] = m < M ? (unsigned)((long long)m * K * 2 + c * 16) : 0x7fffffffu;   // rows past M: zero-filled by the range check
        } else {
            const int rw = r - BM;                 // W image row a*16 + lg*4 + j of a wave's block holds the column of accumulator (a, lg, j)
            const int blk = rw / CW, rr = rw - blk * CW;
            const int n = blk * CW + gl_run_off((rr >> 2) & 3, rr >> 4) + (rr & 3);
            off[j] = (unsigned)((long long)n * K * 2 + c * 16);
        }
    }
    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NST) * STAGE;
        const unsigned kb = kt * 64;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int p = j * NW + wave;
            if (p >= NP) continue;                  // (only the last j can be short: wave-uniform)
            auto dst = (__attribute__((address_space(3))) void*)(st + p * 1024);
            if (p < BM / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, off[j] + kb, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, off[j] + kb, 0, 0, 0);
        }
    };
    int fw[CB], fx[RB];
#pragma unroll
    for (int a = 0; a < CB; ++a) {
        const int rw_ = wn * CW + a * 16 + lr;
        fw[a] = BM * 64 + rw_ * 64 + ((lg ^ gl_swz(rw_)) << 4);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int rx_ = wm * (RB * 16) + b * 16 + lr;
        fx[b] = rx_ * 64 + ((lg ^ gl_swz(rx_)) << 4);
    }
    const int nk = K / 32;
    for (int s0 = 0; s0 < PF && s0 < nk; ++s0) issue(s0);
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = min(nk - 1 - kt, PF - 1);   // stages allowed to stay in flight behind stage kt
        if (short_wave) {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NDMA - 1)) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + PF < nk) issue(kt + PF);
        const unsigned char* st = smem + (kt % NST) * STAGE;
        gl_short8_t w8[CB], x8[RB];
#pragma unroll
        for (int a = 0; a < CB; ++a) w8[a] = *reinterpret_cast<const gl_short8_t*>(st + fw[a]);
#pragma unroll
        for (int b = 0; b < RB; ++b) x8[b] = *reinterpret_cast<const gl_short8_t*>(st + fx[b]);
#pragma unroll
        for (int a = 0; a < CB; ++a)
#pragma unroll
            for (int b = 0; b < RB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gl_bf16x8_t, w8[a]), __builtin_bit_cast(gl_bf16x8_t, x8[b]),
                                                                    acc[a][b], 0, 0, 0);
    }
}

template <int NST, int WM, int WN, int RB>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(WM * WN == 8 ? (RB == 5 ? 2 : 4) : 3)))
void gemm_res_ln_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
                        const float* res, float* xout, const float* __restrict__ gamma, const float* __restrict__ beta,
                        bf16_t* __restrict__ h, float* __restrict__ mean, float* __restrict__ rstd, int M, int K, float eps) {
    constexpr int BM = WM * RB * 16, N = WN * 64, STAGE = (BM + N) * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_sum = reinterpret_cast<float*>(smem + NST * STAGE);   // [BM][WN] row sums per column wave
    float* s_sq = s_sum + BM * WN;                                  // [BM][WN] centred squares
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lg = lane >> 4;
    const int m_blk = blockIdx.x * BM;
    // the accumulators start from the bias (a lane's four runs of its wave's 64 columns): nothing of it is left for the epilogue
    int coff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) coff[a] = wn * 64 + gl_run_off(lg, a);
    gl_f32x4_t acc[4][RB];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const gl_f32x4_t bv = *reinterpret_cast<const gl_f32x4_t*>(bias + coff[a]);
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[a][b] = bv;
    }
    gl_mainloop<NST, WM, WN, RB, 4>(A, W, M, K, m_blk, smem, acc);

    // ---- epilogue.  Lane (lr, lg) owns rows wm*RB*16 + b*16 + lr (b = 0..RB-1) and four 4-column runs of its wave's 64 columns.
    auto ld_runs = [&](const float* row, float (&v)[16]) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 t = *reinterpret_cast<const float4*>(row + coff[a]);
            v[4 * a] = t.x; v[4 * a + 1] = t.y; v[4 * a + 2] = t.z; v[4 * a + 3] = t.w;
        }
    };
    const int row0 = wm * (RB * 16) + lr;     // tile row of row group 0
    // the residual rows are read by hand two groups ahead: xout may alias res, so the compiler would keep each group's reads behind
    // the previous group's stores
    uint4 pf[2][4];
    auto prefetch = [&](int b, uint4 (&dst)[4]) {   // unconditional: rows past M read a clamped address and are not stored
        const float* src = res + (size_t)min(m_blk + row0 + b * 16, M - 1) * N;
#pragma unroll
        for (int a = 0; a < 4; ++a) dst[a] = *reinterpret_cast<const uint4*>(src + coff[a]);
    };
    prefetch(0, pf[0]);
    if (RB > 1) prefetch(1, pf[1]);
    float s1[RB];
    // pass 1: the finished row values (kept in the accumulators), their float32 store and the lane's share of the row sums
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        float cur[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cur[4 * i] = __uint_as_float(pf[b & 1][i].x); cur[4 * i + 1] = __uint_as_float(pf[b & 1][i].y);
            cur[4 * i + 2] = __uint_as_float(pf[b & 1][i].z); cur[4 * i + 3] = __uint_as_float(pf[b & 1][i].w);
        }
        if (b + 2 < RB) prefetch(b + 2, pf[b & 1]);
        const int m = m_blk + row0 + b * 16;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[a][b][j] += cur[a * 4 + j];
            }
            s += (acc[a][b][0] + acc[a][b][1]) + (acc[a][b][2] + acc[a][b][3]);
        }
        if (m < M) {
            float* dst = xout + (size_t)m * N;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                *reinterpret_cast<float4*>(dst + coff[a]) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        s1[b] = s;
    }
    if (lg == 0) {
#pragma unroll
        for (int b = 0; b < RB; ++b) s_sum[(row0 + b * 16) * WN + wn] = s1[b];
    }
    __syncthreads();
    float mu[RB], rs[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) s += s_sum[(row0 + b * 16) * WN + w];
        mu[b] = s / (float)N;
        float q = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = acc[a][b][j] - mu[b];
                q += d * d;
            }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lg == 0) s_sq[(row0 + b * 16) * WN + wn] = q;
    }
    __syncthreads();
    float g16[16], t16[16];
    ld_runs(gamma, g16);
    ld_runs(beta, t16);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) q += s_sq[(row0 + b * 16) * WN + w];
        rs[b] = rsqrtf(q / (float)N + eps);
        const int m = m_blk + row0 + b * 16;
        if (m >= M) continue;
        if (wn == 0 && lg == 0) {
            if (mean) mean[m] = mu[b];
            if (rstd) rstd[m] = rs[b];
        }
        float o[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[a * 4 + j] = (acc[a][b][j] - mu[b]) * rs[b] * g16[a * 4 + j] + t16[a * 4 + j];
        bf16_t* dst = h + (size_t)m * N;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            *reinterpret_cast<uint4*>(dst + coff[2 * hh]) = make_uint4(pack_bf16(o[8 * hh], o[8 * hh + 1]), pack_bf16(o[8 * hh + 2], o[8 * hh + 3]),
                                                                       pack_bf16(o[8 * hh + 4], o[8 * hh + 5]), pack_bf16(o[8 * hh + 6], o[8 * hh + 7]));
    }
}

template <int NST, int WM, int WN, int RB>
static int launch_res_ln(const void* A, const void* W, const float* bias, const float* res, float* xout, const float* gamma,
                         const float* beta, float eps, void* h, float* mean, float* rstd, int M, int K, hipStream_t s) {
    constexpr int BM = WM * RB * 16, N = WN * 64;
    const size_t lds = (size_t)NST * (BM + N) * 64 + (size_t)2 * BM * WN * sizeof(float);
    static unsigned long long attr_done = 0;       // per device ordinal: the attribute belongs to the current device
    if (lds > 48 * 1024 && star_once_per_device(attr_done)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_res_ln_kernel<NST, WM, WN, RB>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return star_set_error(-2, "star_gemm_res_ln: LDS attribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((gemm_res_ln_kernel<NST, WM, WN, RB>), dim3(cdiv(M, BM)), dim3(WM * WN * 64), lds, s, (const bf16_t*)A,
                       (const bf16_t*)W, bias, res, xout, gamma, beta, (bf16_t*)h, mean, rstd, M, K, eps);
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// The same whole-row tile for a PLAIN Linear with a narrow bf16 result (out = A W^T (+ bias)): the decoder's input-gradient GEMMs
// (N = 256) come out as 320 tiles of 256 x 128 on the tiled kernel's 512 slots (62 %); 160-row whole-row tiles are 256, one per CU.
// Reached through star_gemm (no entry point of its own): star_gemm_rows_try returns 0 when it has launched.
template <int NST, int WM, int WN, int RB>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(WM * WN == 8 ? 2 : 3)))
void gemm_rows_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias, bf16_t* __restrict__ out,
                      int M, int K) {
    constexpr int BM = WM * RB * 16, N = WN * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lg = lane >> 4;
    const int m_blk = blockIdx.x * BM;
    int coff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) coff[a] = wn * 64 + gl_run_off(lg, a);
    gl_f32x4_t acc[4][RB];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const gl_f32x4_t bv = bias ? *reinterpret_cast<const gl_f32x4_t*>(bias + coff[a]) : (gl_f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[a][b] = bv;
    }
    gl_mainloop<NST, WM, WN, RB, 4>(A, W, M, K, m_blk, smem, acc);
    const int row0 = wm * (RB * 16) + lr;
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int m = m_blk + row0 + b * 16;
        if (m >= M) continue;
        bf16_t* dst = out + (size_t)m * N;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            *reinterpret_cast<uint4*>(dst + coff[2 * hh]) =
                make_uint4(pack_bf16(acc[2 * hh][b][0], acc[2 * hh][b][1]), pack_bf16(acc[2 * hh][b][2], acc[2 * hh][b][3]),
                           pack_bf16(acc[2 * hh + 1][b][0], acc[2 * hh + 1][b][1]), pack_bf16(acc[2 * hh + 1][b][2], acc[2 * hh + 1][b][3]));
    }
}

template <int NST, int WM, int WN, int RB>
static int launch_rows(const StarGemm* g, hipStream_t s) {
    constexpr int BM = WM * RB * 16, N = WN * 64;
    const size_t lds = (size_t)NST * (BM + N) * 64;
    static unsigned long long attr_done = 0;
    if (lds > 48 * 1024 && star_once_per_device(attr_done)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<NST, WM, WN, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 1;
    }
    hipLaunchKernelGGL((gemm_rows_kernel<NST, WM, WN, RB>), dim3(cdiv(g->M, BM)), dim3(WM * WN * 64), lds, s, (const bf16_t*)g->A, (const bf16_t*)g->W, g->bias,
                       (bf16_t*)g->out_lp, g->M, g->K);
    return 0;
}

// 0 = launched; nonzero = not this kernel's case (the caller goes on to the tiled kernel).  STAR_AMD_GEMM_ROWS=0 disables (A/B).
int star_gemm_rows_try(const StarGemm* g, hipStream_t s) {
    static const char* env = getenv("STAR_AMD_GEMM_ROWS");
    const int mode = env ? atoi(env) : 2;           // 2: N = 256 and 384 (default); 1: N = 256 only; 0: off
    if (mode == 0) return 1;
    if (g->dt != STAR_BF16 || !g->out_lp || g->out_f32 || g->pre || g->aux || g->res || g->res2 || g->a_rows || g->c_rows || g->act != STAR_ACT_NONE ||
        g->flags != 0 || g->conv != 0)
        return 1;
    if (!(g->N == 256 || (g->N == 384 && mode >= 2)) || g->K % 32 != 0 || g->K < 64) return 1;
    if ((long long)g->M * g->K * 2 >= 0x7fffffffLL) return 1;
    if (((reinterpret_cast<uintptr_t>(g->A) | reinterpret_cast<uintptr_t>(g->W) | reinterpret_cast<uintptr_t>(g->out_lp) | reinterpret_cast<uintptr_t>(g->bias)) & 15) != 0)
        return 1;
    const int cus = star_num_cus();
    const int t160 = cdiv(g->M, 160);
    // only where the 160-row tiles fill the chip about evenly: at least 3/4 of the CUs in the last round
    if (t160 < cus * 3 / 4 || (t160 % cus != 0 && t160 % cus < cus * 3 / 4)) return 1;
    return g->N == 256 ? launch_rows<3, 2, 4, 5>(g, s) : launch_rows<3, 2, 6, 5>(g, s);
}

extern "C" int star_gemm_res_ln_supported(int N, int K, int dt) {
    return dt == STAR_BF16 && (N == 256 || N == 384) && K >= 32 && K % 32 == 0;
}

// Geometry by shape, from measurements on MI355X (tools/bench_ops.py gemm_ln at M = 5120 .. 40 960).  A tile
// spans whole rows, so M alone sets the number of workgroups: -1 = the two-launch path is faster (too few tiles for the chip).
//   N = 384: 192-row tiles (12 waves) once they cover 3/4 of the CUs; below that 96-row tiles (6 waves, two per CU) while K <= 512
//            (they re-read the W panel twice as often: at K = 1536 they lose to star_gemm + star_layernorm_fwd);
//   N = 256: 128-row tiles (8 waves) from half the CUs up, 64-row tiles (4 waves) below.
//   160-row tiles (geometry 4) where they need fewer rounds x rows than the standard tile: a tile spans the row, so M alone sets the
//   workgroup count, and M = 40 960 is 214 tiles of 192 rows (a sixth of the CUs idle) or 320 of 128 (two per CU on a quarter of
//   them) but exactly 256 of 160.
static int res_ln_geometry(int M, int N, int K) {
    const int cus = star_num_cus();
    const long long r160 = (long long)cdiv(cdiv(M, 160), cus) * 160;
    if (N == 384) {
        if (cdiv(M, 192) * 4 >= cus * 3) return r160 < (long long)cdiv(cdiv(M, 192), cus) * 192 ? 4 : 0;
        return K <= 512 ? 1 : -1;
    }
    if (cdiv(M, 128) * 2 < cus) return 1;
    return r160 < (long long)cdiv(cdiv(M, 128), 2 * cus) * 2 * 128 ? 4 : 0;
}

extern "C" int star_gemm_res_ln_prefers(int M, int N, int K, int dt) {
    return star_gemm_res_ln_supported(N, K, dt) && M > 0 && (long long)M * K * 2 < 0x7fffffffLL && res_ln_geometry(M, N, K) >= 0;
}

extern "C" int star_gemm_res_ln(const void* A, const void* W, const float* bias, const float* res, float* x_out,
                                const float* gamma, const float* beta, float eps, void* h, float* mean, float* rstd, int M,
                                int N, int K, int dt, void* stream) {
    STAR_CHECK(A && W && bias && res && x_out && gamma && beta && h, "star_gemm_res_ln: null pointer");
    STAR_CHECK(star_gemm_res_ln_supported(N, K, dt), "star_gemm_res_ln: no kernel for N=%d K=%d dt=%d (bf16, N = 256 or 384, K %% 32 == 0)", N, K, dt);
    STAR_CHECK(M >= 0 && (long long)M * K * 2 < 0x7fffffffLL, "star_gemm_res_ln: M=%d K=%d: the A operand must stay below 2 GiB", M, K);
    STAR_CHECK(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                 reinterpret_cast<uintptr_t>(h)) & 15) == 0, "star_gemm_res_ln: every operand must be 16-byte aligned");
    if (M == 0) return 0;
    const char* env = getenv("STAR_AMD_GEMM_LN_GEO");   // read per call (tests walk the geometries in one process)
    int geo = env ? atoi(env) : res_ln_geometry(M, N, K);
    if (geo < 0) geo = N == 384 ? 1 : 0;                 // called although the two-launch path is faster: still correct
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (N == 384) {
        if (geo == 1) rc = launch_res_ln<2, 1, 6, 6>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else if (geo == 2) rc = launch_res_ln<3, 1, 6, 6>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else if (geo == 3) rc = launch_res_ln<4, 2, 6, 6>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else if (geo == 4) rc = launch_res_ln<3, 2, 6, 5>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else rc = launch_res_ln<3, 2, 6, 6>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
    } else {
        if (geo == 1) rc = launch_res_ln<3, 1, 4, 4>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else if (geo == 4) rc = launch_res_ln<3, 2, 4, 5>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
        else rc = launch_res_ln<3, 2, 4, 4>(A, W, bias, res, x_out, gamma, beta, eps, h, mean, rstd, M, K, s);
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_gemm_res_ln");
    return 0;
}

