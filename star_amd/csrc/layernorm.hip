// K2: LayerNorm (eps = 1e-6) forward / backward on the float32 residual stream.
// Reference: nn.LayerNorm inside timm Block + self.norm / self.decoder_norm
// (star/models/mae_base.py:279-282,308-312).  HBM-bound: one wave per row, the row lives in registers,
// 16-byte loads, wavefront reductions; the normalised row leaves in the activation type so that the
// following MFMA GEMM reads half the bytes.
#include <atomic>

#include "common.h"

#define LN_THREADS 256
#define LN_MAXV 8          // float4 per lane: D <= 64*4*8 = 2048
#define LN_BWD_BLOCKS 768   // partial dgamma/dbeta rows = workgroups: three per CU (A/B inside the step: 768 beats 512, 1024 and 2048 by 0.3-0.5 %; final round-2 build: 768 11.89-11.91 ms, 1024 11.89-11.92, 512 12.02-12.06)
#define LN_BWD_THREADS 512  // 8 waves: 1024 blocks put 8 waves on every SIMD, each with one row's loads in flight

// SPLIT (dt = STAR_BF16X3): y rows are [hi (D) | lo (D)] bf16 pairs, hi = bf16(o), lo = bf16(o - hi)
template <typename T, int NV, bool SPLIT = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                            int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;  // float4 per row
    const float* xr = x + (size_t)row * D;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            ld4(xr + c * 4, v[i]);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[i][k] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    T* yr = y + (size_t)row * (SPLIT ? 2 * D : D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            float g[4], b[4], o[4];
            ld4(gamma + c * 4, g);
            ld4(beta + c * 4, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mu) * rs * g[k] + b[k];
            st4(yr + c * 4, o);
            if constexpr (SPLIT) {
                float lo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) lo[k] = o[k] - bf2f(f2bf(o[k]));
                st4(yr + D + c * 4, lo);
            }
        }
    }
}

// Backward.  Each wave walks rows with a grid stride and keeps its dgamma/dbeta partials in registers;
// the block folds its waves through LDS and writes one partial row; a second kernel sums the partial
// rows in a fixed order (deterministic).  All three inputs of a row (x, dy and the incoming residual gradient) are
// requested before the first reduction, so a row costs ONE memory latency (g_in may alias g_out: same row, read first).
template <typename T, int NV>
__global__ __launch_bounds__(LN_BWD_THREADS) void ln_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* g_in,
                                                            float* g_out, T* __restrict__ g_lp,
                                                            float* __restrict__ ws, int M, int D) {
    extern __shared__ float s_part[];  // [waves][2][D]
    constexpr int NW = LN_BWD_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    float pg[NV][4], pb[NV][4], gm[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) pg[i][k] = pb[i][k] = 0.f;
        if (c < nv) ld4(gamma + c * 4, gm[i]);
    }
    for (int row = blockIdx.x * NW + wave; row < M; row += gridDim.x * NW) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NV][4], dg[NV][4], gi[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) gi[i][k] = 0.f;
            if (g_in && c < nv) ld4(g_in + (size_t)row * D + c * 4, gi[i]);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                float xv[4], dv[4];
                ld4(x + (size_t)row * D + c * 4, xv);
                ld4(dy + (size_t)row * D + c * 4, dv);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xh[i][k] = (xv[k] - mu) * rs;
                    dg[i][k] = dv[k] * gm[i][k];
                    s1 += dg[i][k];
                    s2 += dg[i][k] * xh[i][k];
                    pg[i][k] += dv[k] * xh[i][k];
                    pb[i][k] += dv[k];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = rs * (dg[i][k] - m1 - xh[i][k] * m2);
                if (g_in) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] += gi[i][k];
                }
                if (g_out) st4(g_out + (size_t)row * D + c * 4, o);
                if (g_lp) st4(g_lp + (size_t)row * D + c * 4, o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            st4(s_part + (size_t)(wave * 2 + 0) * D + c * 4, pg[i]);
            st4(s_part + (size_t)(wave * 2 + 1) * D + c * 4, pb[i]);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += LN_BWD_THREADS) {
        const int which = e / D, c = e - which * D;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += s_part[(size_t)(w * 2 + which) * D + c];
        ws[(size_t)blockIdx.x * 2 * D + e] = a;
    }
}

// second stage: block = 16 columns x 64 row lanes over the partial rows (48 blocks at D = 384 instead of 12: the kernel is
// a latency chain per block), four independent accumulators per lane, fixed order -> deterministic
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int nblocks, int D) {
    __shared__ float s[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < 2 * D) {
        int b = rl;
        for (; b + 192 < nblocks; b += 256) {
            a0 += ws[(size_t)b * 2 * D + e];
            a1 += ws[(size_t)(b + 64) * 2 * D + e];
            a2 += ws[(size_t)(b + 128) * 2 * D + e];
            a3 += ws[(size_t)(b + 192) * 2 * D + e];
        }
        for (; b < nblocks; b += 64) a0 += ws[(size_t)b * 2 * D + e];
    }
    s[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && e < 2 * D) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) a += s[r][cl];
        if (e < D) {
            if (dgamma) dgamma[e] = a;
        } else if (dbeta) {
            dbeta[e - D] = a;
        }
    }
}

extern "C" int star_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int M, int D, float eps, int dt, void* stream) {
    STAR_CHECK(x && gamma && beta && y, "star_layernorm_fwd: null pointer");
    STAR_CHECK(M >= 0 && D >= 4 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "star_layernorm_fwd: D=%d must be a multiple of 4, <= %d", D, 64 * 4 * LN_MAXV);
    if (M == 0) return 0;
    dim3 grid(cdiv(M, LN_THREADS / 64));
#define LN_FWD(NV)                                                                                                  \
    do {                                                                                                            \
        if (dt == STAR_BF16X3)                                                                                      \
            hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV, true>), grid, dim3(LN_THREADS), 0, (hipStream_t)stream, x, gamma, \
                               beta, (bf16_t*)y, mean, rstd, M, D, eps);                                            \
        else if (dt == STAR_BF16)                                                                                   \
            hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV>), grid, dim3(LN_THREADS), 0, (hipStream_t)stream, x, gamma, \
                               beta, (bf16_t*)y, mean, rstd, M, D, eps);                                            \
        else                                                                                                        \
            hipLaunchKernelGGL((ln_fwd_kernel<float, NV>), grid, dim3(LN_THREADS), 0, (hipStream_t)stream, x, gamma,  \
                               beta, (float*)y, mean, rstd, M, D, eps);                                             \
    } while (0)
    const int nvec = (D / 4 + 63) / 64;
    if (nvec <= 1) LN_FWD(1); else if (nvec == 2) LN_FWD(2); else if (nvec == 3) LN_FWD(3); else if (nvec == 4) LN_FWD(4); else LN_FWD(LN_MAXV);
    STAR_LAUNCH_CHECK("star_layernorm_fwd");
    return 0;
}

extern "C" int64_t star_layernorm_bwd_ws(int D) { return (int64_t)LN_BWD_BLOCKS * 2 * D; }

static int ln_bwd_launch(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* g_in, float* g_out, void* g_out_lp, float* dgamma, float* dbeta, float* ws, int M, int D,
                         int dt, void* stream) {
    STAR_CHECK(dy && x && gamma && mean && rstd && ws, "star_layernorm_bwd: null pointer");
    STAR_CHECK(M > 0 && D >= 4 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "star_layernorm_bwd: bad M=%d D=%d", M, D);
    int nb = cdiv(M, LN_BWD_THREADS / 64);
    if (nb > LN_BWD_BLOCKS) nb = LN_BWD_BLOCKS;
    const size_t lds = (size_t)(LN_BWD_THREADS / 64) * 2 * D * sizeof(float);
#define LN_BWD(NV)                                                                                                   \
    do {                                                                                                             \
        if (lds > 64 * 1024) {   /* D > 1024: more dynamic LDS than the default limit */                              \
            const hipError_t ea = dt == STAR_BF16                                                                    \
                ? hipFuncSetAttribute((const void*)ln_bwd_kernel<bf16_t, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) \
                : hipFuncSetAttribute((const void*)ln_bwd_kernel<float, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (ea != hipSuccess) return star_set_error(-2, "star_layernorm_bwd: LDS attribute: %s", hipGetErrorString(ea)); \
        }                                                                                                            \
        if (dt == STAR_BF16)                                                                                         \
            hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NV>), dim3(nb), dim3(LN_BWD_THREADS), lds, (hipStream_t)stream,      \
                               (const bf16_t*)dy, x, gamma, mean, rstd, g_in, g_out, (bf16_t*)g_out_lp, ws, M, D);   \
        else                                                                                                         \
            hipLaunchKernelGGL((ln_bwd_kernel<float, NV>), dim3(nb), dim3(LN_BWD_THREADS), lds, (hipStream_t)stream,       \
                               (const float*)dy, x, gamma, mean, rstd, g_in, g_out, (float*)g_out_lp, ws, M, D);     \
    } while (0)
    const int nvec = (D / 4 + 63) / 64;
    if (nvec <= 1) LN_BWD(1); else if (nvec == 2) LN_BWD(2); else if (nvec == 3) LN_BWD(3); else if (nvec == 4) LN_BWD(4); else LN_BWD(LN_MAXV);
    STAR_LAUNCH_CHECK("star_layernorm_bwd");
    if (dgamma == nullptr && dbeta == nullptr) return 0;   // deferred: the partial rows stay in ws for star_layernorm_fold_batch
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * D, 16)), dim3(1024), 0, (hipStream_t)stream, ws, dgamma, dbeta, nb, D);
    STAR_LAUNCH_CHECK("star_layernorm_bwd(reduce)");
    return 0;
}

// The folds of MANY LayerNorm backward calls in one launch (grid.y = job): a training step has 22 LayerNorms whose dgamma / dbeta
// nothing in the backward chain waits for; each fold is a 5 us launch on the critical path when issued beside its LayerNorm.
__global__ __launch_bounds__(1024) void ln_fold_batch_kernel(const StarLnFoldJob* __restrict__ jobs) {
    const StarLnFoldJob j = jobs[blockIdx.y];
    const int D = j.D, nblocks = j.nb;
    if ((int)blockIdx.x * 16 >= 2 * D) return;
    const float* __restrict__ ws = j.ws;
    __shared__ float s[64][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < 2 * D) {      // the summation order of ln_bwd_reduce_kernel: the batched fold is bit-identical to the immediate one
        int b = rl;
        for (; b + 192 < nblocks; b += 256) {
            a0 += ws[(size_t)b * 2 * D + e];
            a1 += ws[(size_t)(b + 64) * 2 * D + e];
            a2 += ws[(size_t)(b + 128) * 2 * D + e];
            a3 += ws[(size_t)(b + 192) * 2 * D + e];
        }
        for (; b < nblocks; b += 64) a0 += ws[(size_t)b * 2 * D + e];
    }
    s[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && e < 2 * D) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) a += s[r][cl];
        if (e < D) {
            if (j.dgamma) j.dgamma[e] = a;
        } else if (j.dbeta) {
            j.dbeta[e - D] = a;
        }
    }
}

extern "C" int star_layernorm_fold_rows(int M) {
    int nb = cdiv(M, LN_BWD_THREADS / 64);
    return nb > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : nb;
}

extern "C" int star_layernorm_fold_batch(const StarLnFoldJob* jobs_dev, int njobs, int max_D, void* stream) {
    STAR_CHECK(jobs_dev && njobs > 0 && njobs <= 65535 && max_D >= 4, "star_layernorm_fold_batch: bad arguments (njobs=%d max_D=%d)", njobs, max_D);
    hipLaunchKernelGGL(ln_fold_batch_kernel, dim3(cdiv(2 * max_D, 16), njobs), dim3(1024), 0, (hipStream_t)stream, jobs_dev);
    STAR_LAUNCH_CHECK("star_layernorm_fold_batch");
    return 0;
}

extern "C" int star_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, const float* g_in, float* g_out, void* g_out_lp, float* dgamma,
                                  float* dbeta, float* ws, int M, int D, int dt, void* stream) {
    return ln_bwd_launch(dy, x, gamma, mean, rstd, g_in, g_out, g_out_lp, dgamma, dbeta, ws, M, D, dt, stream);
}

// ---- the allocation-free variant -------------------------------------------------------------------------------
// Same kernels as star_layernorm_bwd; the partial dgamma/dbeta rows live in a file-scope device array instead of a
// caller-provided workspace, for callers that cannot allocate (HIP graph capture, tight serving loops).  One copy of the
// array per GPU, at its own address: the address is looked up once per device and kept.  Nothing in it has to start at
// zero (the first kernel writes every partial row the second one reads), so no clearing; calls that overlap on one GPU
// (two streams) race on it.
#define LN_STATIC_MAX_D 1024
#define LN_STATIC_MAX_DEV 64
static __device__ __align__(256) float scratch[(size_t)LN_BWD_BLOCKS * 2 * LN_STATIC_MAX_D];   // 6 MiB
static std::atomic<float*> g_scratch_ptr[LN_STATIC_MAX_DEV];      // racing first calls store the same address

extern "C" int star_layernorm_bwd_static(const void* dy, const float* x, const float* gamma, const float* mean,
                                         const float* rstd, const float* g_in, float* g_out, void* g_out_lp,
                                         float* dgamma, float* dbeta, int M, int D, int dt, void* stream) {
    STAR_CHECK(D <= LN_STATIC_MAX_D, "star_layernorm_bwd_static: D=%d exceeds the built-in scratch (D <= %d); use star_layernorm_bwd with a workspace", D, LN_STATIC_MAX_D);
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return star_set_error(-2, "star_layernorm_bwd_static: hipGetDevice: %s", hipGetErrorString(e));
    STAR_CHECK(dev >= 0 && dev < LN_STATIC_MAX_DEV, "star_layernorm_bwd_static: device %d out of range", dev);
    float* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {   // once per device, with that device current: every GPU holds its own copy of the array
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr)
            return star_set_error(-2, "star_layernorm_bwd_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    return ln_bwd_launch(dy, x, gamma, mean, rstd, g_in, g_out, g_out_lp, dgamma, dbeta, ptr, M, D, dt, stream);
}
