// VQ-STAR quantiser: the vector quantiser that sits between compressor and decompressor in the
// `--com vqstar` variant of the path.  Replaces STARVectorQuantizer.forward (star/models/mae_base.py:122-173):
//   distances = |x|^2 + |e|^2 - 2 x e^T        [M, K] float32, never materialised here
//   idx       = argmin_k distances              (first minimum)
//   quantized = one_hot(idx) @ codebook         = codebook[idx]
//   loss      = mse(q, x) + commitment * mse(q, x)      (q_latent + cc * e_latent; equal values, different gradients)
//   perplexity= exp(-sum p log(p + 1e-10)),  p = histogram(idx) / M
// and its autograd: straight-through dx = dq + cc * 2 (x - q) / (M E), codebook gradient 2 (q - x) / (M E) summed
// per code (the encodings^T matmul of the reference).
//
// star_vq_assign: the distance contraction runs on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact float32
// products and sums, so the argmin sees float32 distances as the reference does -- bf16 operands would move
// near-ties).  A 256-thread workgroup owns 64 rows (a wave: one 16-row tile, its x fragment in registers for the whole
// launch; 2 560 waves at M = 40 960 spread over the 1 024 SIMDs better than fewer, fatter waves) and streams the
// codebook through a double-buffered LDS image in 64-code chunks ([64][E+2] floats: the +2 makes the 16 codes x
// 2 k-slices of a fragment read hit 32 different banks).  Each lane keeps the running (distance, index) minimum of
// its 4 rows x 1 column residue; a 16-lane reduction ends the launch, then the 16 lanes of a row copy the winning
// code (float32 + the decompressor's operand type), accumulate the squared error and bump the code's histogram
// count.  MFMA-bound: 2 M K E flop at the fp32 matrix rate (157 TFLOP/s): 21.5 GFLOP = 137 us at M = 40 960,
// K = 8192, E = 32.
// Deterministic: squared-error partials per workgroup are folded in a fixed order; the histogram is integer.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float vq_f32x4;

#define VQ_THREADS 256
#define VQ_ROWS 64      // rows per workgroup: one 16-row tile per wave
#define VQ_CH 64        // codes per LDS chunk
#define VQ_MAXE 128

template <typename T, int E4>   // E4 = E / 4
__global__ __launch_bounds__(VQ_THREADS) void vq_assign_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                               int64_t* __restrict__ idx_out, float* __restrict__ q_f32, T* __restrict__ q_lp,
                                                               float* __restrict__ part, int* __restrict__ counts, int M, int K) {
    constexpr int E = E4 * 4, S = E + 2;
    extern __shared__ float vq_lds[];          // 2 x ([VQ_CH][S] codes + [VQ_CH] |e|^2), [4] wave partials, [VQ_ROWS] winners
    float* wsum = vq_lds + 2 * (VQ_CH * S + VQ_CH);
    int* win = reinterpret_cast<int*>(wsum + 4);   // [VQ_ROWS] winning code per row of the workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * VQ_ROWS + wave * 16;

    // x fragment: A[m = lr][k = lg] of k-slice kk; rows past M read the last row (never stored)
    float xa[E4];
    float rx2[4];                                // |x|^2 of the rows this lane's accumulator registers belong to
    {
        const int r = min(row0 + lr, M - 1);
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < E4; ++kk) {
            xa[kk] = x[(size_t)r * E + kk * 4 + lg];
            s = fmaf(xa[kk], xa[kk], s);
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);              // |x|^2 of row lr, in every lane with that lr
#pragma unroll
        for (int j = 0; j < 4; ++j) rx2[j] = __shfl(s, lg * 4 + j, 64);
    }
    float bd[4];
    int bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bd[j] = INFINITY;
        bi[j] = 0x7fffffff;
    }

    // codebook chunks, double-buffered: the next chunk's global loads are issued before this chunk's MFMAs and land in
    // the other LDS buffer after them; one barrier per chunk.  |e|^2 is folded while staging (E4 adjacent lanes per code).
    constexpr int NL = (VQ_CH * E4 + VQ_THREADS - 1) / VQ_THREADS;
    constexpr int BUF = VQ_CH * S + VQ_CH;       // floats per buffer: codes, then |e|^2
    float4 stg[NL];
    bool cbbad = false;                          // a staged code with non-finite |e|^2
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int t = tid + i * VQ_THREADS, c = t / E4, q = t - c * E4;
            stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < VQ_CH * E4 && c0 + c < K) stg[i] = *reinterpret_cast<const float4*>(cb + (size_t)(c0 + c) * E + q * 4);
        }
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int t = tid + i * VQ_THREADS, c = t / E4, q = t - c * E4;
            const float4 v = stg[i];
            float sq = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
#pragma unroll
            for (int o = 1; o < E4; o <<= 1) sq += __shfl_xor(sq, o, 64);   // E4 is a power of two: the E4 lanes of a code are aligned
            if (t < VQ_CH * E4) {
                float* d = buf + c * S + q * 4;      // S is even: 8-byte aligned pairs
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
                if (q == 0) buf[VQ_CH * S + c] = sq;
                cbbad |= !(sq < INFINITY);
            }
        }
    };
    // torch.argmin takes the FIRST NaN distance as the minimum.  A distance can only be NaN when |x|^2 or |e|^2 is not
    // finite, so the canonicalising compare (NaN -> -inf, two more VALU ops per distance: +8 %) runs only for workgroups that
    // hold such a row, or -- a second scan -- once a non-finite code has been staged.  Both decisions are workgroup-uniform.
    auto scan = [&](auto careful) {
        fetch(0);
        stash(vq_lds);
        __syncthreads();
        int cur = 0;
        for (int c0 = 0; c0 < K; c0 += VQ_CH) {
            const bool more = c0 + VQ_CH < K;
            if (more) fetch(c0 + VQ_CH);
            const float* buf = vq_lds + cur * BUF;
            const float* e2 = buf + VQ_CH * S;
    #pragma unroll
            for (int ct = 0; ct < VQ_CH / 16; ct += 2) {   // two column tiles at a time: two independent accumulator chains
                vq_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const float* bsrc = buf + (ct * 16 + lr) * S + lg;
    #pragma unroll
                for (int kk = 0; kk < E4; ++kk) {
                    // B[k = lg][n = lr] = code (ct*16 + lr), element kk*4 + lg
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk], bsrc[kk * 4], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk], bsrc[16 * S + kk * 4], acc[1], 0, 0, 0);
                }
    #pragma unroll
                for (int h = 0; h < 2; ++h) {            // ascending columns per lane: strict < keeps the first minimum
                    const int col = c0 + (ct + h) * 16 + lr;
                    const float ce2 = e2[(ct + h) * 16 + lr];
                    if (col < K) {
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // the reference's evaluation order: (|x|^2 + |e|^2) - 2 * dot, one rounding each
                            float d = __fsub_rn(__fadd_rn(rx2[j], ce2), __fmul_rn(2.0f, acc[h][j]));
                            if constexpr (decltype(careful)::value) d = (d != d) ? -INFINITY : d;
                            if (d < bd[j]) {
                                bd[j] = d;
                                bi[j] = col;
                            }
                        }
                    }
                }
            }
            if (more) stash(vq_lds + (cur ^ 1) * BUF);
            __syncthreads();
            cur ^= 1;
        }
    };
    bool rowbad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) rowbad |= !(rx2[j] < INFINITY);
    if (__syncthreads_or(rowbad)) {
        scan(std::true_type{});
    } else {
        scan(std::false_type{});
        if (__syncthreads_or(cbbad)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bd[j] = INFINITY;
                bi[j] = 0x7fffffff;
            }
            scan(std::true_type{});
        }
    }
    // minimum over the 16 column residues (lanes of one lg group); ties -> the smaller index
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float od = __shfl_xor(bd[j], o, 64);
            const int oi = __shfl_xor(bi[j], o, 64);
            if (od < bd[j] || (od == bd[j] && oi < bi[j])) {
                bd[j] = od;
                bi[j] = oi;
            }
        }
        if (bi[j] == 0x7fffffff) bi[j] = 0;          // every distance +inf: argmin's first minimum is code 0
    }
    // outputs: the 16 lanes of a group copy the winning code of each of the group's 4 rows
    float err = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = row0 + lg * 4 + j;
        if (r >= M) continue;
        const int k = bi[j];
        for (int e = lr; e < E; e += 16) {
            const float q = cb[(size_t)k * E + e];
            const float d = q - x[(size_t)r * E + e];
            err = fmaf(d, d, err);
            if (q_f32) q_f32[(size_t)r * E + e] = q;
            if (q_lp) Act<T>::st(q_lp + (size_t)r * E + e, q);
        }
        if (lr == 0) {
            idx_out[r] = k;
            win[wave * 16 + lg * 4 + j] = k;
        }
    }
    err = wave_sum(err);
    if (lane == 0) wsum[wave] = err;
    __syncthreads();
    if (tid == 0 && part) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    // histogram: one atomic per DISTINCT code of the workgroup's rows (a collapsed codebook would otherwise send every
    // row's increment to the same address)
    if (counts && tid < VQ_ROWS && blockIdx.x * VQ_ROWS + tid < M) {
        const int k = win[tid];
        int n = 0;
        bool first = true;
        const int rows_here = min(VQ_ROWS, M - blockIdx.x * VQ_ROWS);
        for (int t = 0; t < rows_here; ++t) {
            const bool same = win[t] == k;
            n += same;
            first = first && !(same && t < tid);
        }
        if (first) atomicAdd(&counts[k], n);
    }
}

// out3[0] = vq loss, out3[1] = perplexity, out3[2] = mse; *loss_inout += vq loss (the model's total loss).
// ema: the EMA quantiser has no codebook term in its loss (mae_base.py:244-245): loss = cc * mse.
__global__ __launch_bounds__(256) void vq_finish_kernel(const float* __restrict__ part, int nblk, const int* __restrict__ counts, int M, int E, int K,
                                                        float cc, int ema, float* __restrict__ out3, float* __restrict__ loss_inout) {
    __shared__ float s_a[256], s_b[256];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += part[i];
    for (int k = threadIdx.x; k < K; k += 256) {
        const float p = (float)counts[k] / (float)M;
        b += p * logf(p + 1e-10f);
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s_a[threadIdx.x] += s_a[threadIdx.x + o];
            s_b[threadIdx.x] += s_b[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = s_a[0] / ((float)M * (float)E);
        const float loss = ema ? cc * mse : mse + cc * mse;
        out3[0] = loss;
        out3[1] = expf(-s_b[0]);
        out3[2] = mse;
        if (loss_inout) loss_inout[0] += loss;
    }
}

// dlat = (dq + cc * 2 (x - q) / (M E)) * [x > 0]   (straight-through estimator, then the compressor's ReLU)
template <typename T>
__global__ __launch_bounds__(256) void vq_bwd_rows_kernel(const T* __restrict__ dq, const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          const float* __restrict__ cb, T* __restrict__ dlat, int64_t n, int E, float scale) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t m = t / E;
    const int e = (int)(t - m * E);
    const float xv = x[t];
    const float q = cb[(size_t)idx[m] * E + e];
    const float g = Act<T>::ld(dq + t) + scale * (xv - q);
    Act<T>::st(dlat + t, xv > 0.f ? g : 0.f);
}

// Codebook gradient, deterministic and balanced.  part[c, k, :] = sum over the rows m of row chunk c with idx[m] == k, in
// row order, of (cb[k, :] - x[m, :]); then dcb[k, :] = 2 / (M E) * sum_c part[c, k, :] in chunk order.
// A wave owns G codes x one chunk of rows and scans the chunk's indices (L2-resident); matches are taken in row order
// from the ballot, four at a time so that four x-row loads are in flight (a collapsed codebook sends every row of the
// chunk to ONE wave: the chunking bounds that chain to `rows_per` rows, the prefetch hides its load latency).
// SUMX: accumulate x[m, :] alone -- the per-code row sums of the EMA codebook update (encodings^T x, mae_base.py:236).
#define VQB_PF 4
template <bool SUMX>
__global__ __launch_bounds__(256) void vq_bwd_codes_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, const float* __restrict__ cb,
                                                           float* __restrict__ part, int M, int E, int K, int G, int rows_per) {
    extern __shared__ float vq_acc[];            // [4 waves][G][E]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 4 + wave) * G;
    const int chunk = blockIdx.y;
    float* acc = vq_acc + (size_t)wave * G * E;
    for (int t = lane; t < G * E; t += 64) acc[t] = 0.f;
    if (c0 >= K) return;
    const int m_lo = chunk * rows_per, m_hi = min(M, m_lo + rows_per);
    for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
        const int m = m0 + lane;
        const long long v = m < m_hi ? (long long)idx[m] : -1;
        const int g = (int)(v - c0);
        unsigned long long mask = __ballot(v >= c0 && g < G && v < K);
        while (mask) {
            int tt[VQB_PF], gg[VQB_PF];
#pragma unroll
            for (int u = 0; u < VQB_PF; ++u) {
                tt[u] = mask ? __ffsll((long long)mask) - 1 : -1;
                mask &= mask - 1;            // 0 stays 0
                gg[u] = __shfl(g, tt[u] < 0 ? 0 : tt[u], 64);
            }
            for (int e = lane; e < E; e += 64) {
                float xv[VQB_PF];
#pragma unroll
                for (int u = 0; u < VQB_PF; ++u) xv[u] = tt[u] >= 0 ? x[(size_t)(m0 + tt[u]) * E + e] : 0.f;
#pragma unroll
                for (int u = 0; u < VQB_PF; ++u)
                    if (tt[u] >= 0) acc[gg[u] * E + e] += SUMX ? xv[u] : cb[(size_t)(c0 + gg[u]) * E + e] - xv[u];
            }
        }
    }
    float* dst = part + (size_t)chunk * K * E;
    for (int t = lane; t < G * E; t += 64) {
        const int g = t / E;
        if (c0 + g < K) dst[(size_t)(c0 + g) * E + (t - g * E)] = acc[t];
    }
}

__global__ __launch_bounds__(256) void vq_bwd_codes_reduce_kernel(const float* __restrict__ part, float* __restrict__ dcb, int64_t n, int C, float scale) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += part[(size_t)c * n + t];
    dcb[t] = scale * s;
}

__global__ __launch_bounds__(256) void onehot_rows_kernel(const int64_t* __restrict__ idx, float* __restrict__ out, int64_t n4, int K4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one float4
    if (t >= n4) return;
    const int64_t m = t / K4;
    const int c = (int)(t - m * K4) * 4;
    const long long k = idx[m];
    reinterpret_cast<float4*>(out)[t] = make_float4(k == c ? 1.f : 0.f, k == c + 1 ? 1.f : 0.f, k == c + 2 ? 1.f : 0.f, k == c + 3 ? 1.f : 0.f);
}

extern "C" int star_vq_nblocks(int M) { return cdiv(M, VQ_ROWS); }

template <typename T>
static int launch_assign(const float* x, const float* cb, int64_t* idx, float* q_f32, void* q_lp, float* part, int* counts, int M, int E, int K,
                         hipStream_t s) {
    const size_t lds = ((size_t)2 * (VQ_CH * (E + 2) + VQ_CH) + 4 + VQ_ROWS) * sizeof(float);
    const dim3 grid(cdiv(M, VQ_ROWS));
#define VQ_CASE(e4)                                                                                                                        \
    case e4:                                                                                                                               \
        hipLaunchKernelGGL((vq_assign_kernel<T, e4>), grid, dim3(VQ_THREADS), lds, s, x, cb, idx, q_f32, (T*)q_lp, part, counts, M, K);   \
        break;
    switch (E / 4) {
        VQ_CASE(2) VQ_CASE(4) VQ_CASE(8) VQ_CASE(16) VQ_CASE(32)
        default: return star_set_error(-1, "star_vq_assign: embedding dim %d not in {8, 16, 32, 64, 128}", E);
    }
#undef VQ_CASE
    return 0;
}

extern "C" int star_vq_assign(const float* x, const float* codebook, int64_t* idx, float* q_f32, void* q_lp, float* part, int32_t* counts,
                              int M, int E, int K, int dt, void* stream) {
    STAR_CHECK(x && codebook && idx, "star_vq_assign: null pointer");
    STAR_CHECK(M > 0 && K > 0 && E > 0 && E % 4 == 0 && E <= VQ_MAXE, "star_vq_assign: bad shape M=%d K=%d E=%d", M, K, E);
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32, "star_vq_assign: unknown dtype %d", dt);
    STAR_CHECK(((reinterpret_cast<uintptr_t>(codebook) | reinterpret_cast<uintptr_t>(x)) & 15) == 0, "star_vq_assign: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (counts) {
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)K * sizeof(int32_t), s);
        if (e != hipSuccess) return star_set_error(-2, "star_vq_assign: memset: %s", hipGetErrorString(e));
    }
    const int rc = dt == STAR_BF16 ? launch_assign<bf16_t>(x, codebook, idx, q_f32, q_lp, part, counts, M, E, K, s)
                                   : launch_assign<float>(x, codebook, idx, q_f32, q_lp, part, counts, M, E, K, s);
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_vq_assign");
    return 0;
}

extern "C" int star_vq_finish(const float* part, int nblocks, const int32_t* counts, int M, int E, int K, float commitment, int ema, float* out3,
                              float* loss_inout, void* stream) {
    STAR_CHECK(part && counts && out3, "star_vq_finish: null pointer");
    STAR_CHECK(nblocks > 0 && M > 0 && E > 0 && K > 0, "star_vq_finish: bad shape");
    hipLaunchKernelGGL(vq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblocks, counts, M, E, K, commitment, ema, out3, loss_inout);
    STAR_LAUNCH_CHECK("star_vq_finish");
    return 0;
}

static void vq_bwd_plan(int M, int* C, int* rows_per) {
    int rp = 640;                                   // rows per chunk: ten 64-row windows
    if (cdiv(M, rp) > 64) rp = cdiv(cdiv(M, 64), 64) * 64;
    *rows_per = rp;
    *C = cdiv(M, rp);
}

extern "C" int64_t star_vq_bwd_ws(int M, int E, int K) {
    int C, rp;
    vq_bwd_plan(M, &C, &rp);
    return (int64_t)C * K * E;
}

extern "C" int star_vq_bwd(const void* dq, const float* x, const int64_t* idx, const float* codebook, void* dlat, float* dcodebook, float* ws,
                           int M, int E, int K, float commitment, int dt, void* stream) {
    STAR_CHECK(x && idx && codebook, "star_vq_bwd: null pointer");
    STAR_CHECK(M > 0 && K > 0 && E > 0, "star_vq_bwd: bad shape M=%d K=%d E=%d", M, K, E);
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32, "star_vq_bwd: unknown dtype %d", dt);
    hipStream_t s = (hipStream_t)stream;
    const float inv = 2.0f / ((float)M * (float)E);
    if (dlat) {
        STAR_CHECK(dq, "star_vq_bwd: dlat needs dq");
        const int64_t n = (int64_t)M * E;
        if (dt == STAR_BF16)
            hipLaunchKernelGGL(vq_bwd_rows_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const bf16_t*)dq, x, idx, codebook, (bf16_t*)dlat, n, E,
                               commitment * inv);
        else
            hipLaunchKernelGGL(vq_bwd_rows_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float*)dq, x, idx, codebook, (float*)dlat, n, E,
                               commitment * inv);
    }
    if (dcodebook) {
        STAR_CHECK(ws, "star_vq_bwd: dcodebook needs a workspace of star_vq_bwd_ws(M, E, K) floats");
        int C, rp;
        vq_bwd_plan(M, &C, &rp);
        const int G = K >= 1024 ? 8 : 2;
        const size_t lds = (size_t)4 * G * E * sizeof(float);
        hipLaunchKernelGGL(vq_bwd_codes_kernel<false>, dim3(cdiv(K, 4 * G), C), dim3(256), lds, s, x, idx, codebook, ws, M, E, K, G, rp);
        const int64_t n = (int64_t)K * E;
        hipLaunchKernelGGL(vq_bwd_codes_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, ws, dcodebook, n, C, inv);
    }
    STAR_LAUNCH_CHECK("star_vq_bwd");
    return 0;
}

// EMA codebook update in place (star/models/mae_base.py:228-241), one block (K E is a few 100k elements):
//   cs = cs * decay + (1 - decay) * counts;  n = sum(cs);  cs = (cs + eps) / (n + K eps) * n
//   ema_w = ema_w * decay + (1 - decay) * dw;  codebook = ema_w / cs[:, None]
__global__ __launch_bounds__(1024) void vq_ema_update_kernel(const int* __restrict__ counts, const float* __restrict__ dw, float* __restrict__ cb,
                                                             float* __restrict__ ema_w, float* __restrict__ cs, int E, int K, float decay,
                                                             float one_minus, float eps, float keps) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float v = __fadd_rn(__fmul_rn(cs[k], decay), __fmul_rn(one_minus, (float)counts[k]));   // torch: two products, one sum
        cs[k] = v;
        s += v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float n = red[0];
    for (int k = threadIdx.x; k < K; k += 1024) cs[k] = __fmul_rn(__fdiv_rn(cs[k] + eps, n + keps), n);   // same thread wrote cs[k] above
    __syncthreads();
    const int total = K * E;
    for (int t = threadIdx.x; t < total; t += 1024) {
        const float w = __fadd_rn(__fmul_rn(ema_w[t], decay), __fmul_rn(one_minus, dw[t]));
        ema_w[t] = w;
        cb[t] = w / cs[t / E];
    }
}

extern "C" int star_vq_code_sums(const float* x, const int64_t* idx, float* dw, float* ws, int M, int E, int K, void* stream) {
    STAR_CHECK(x && idx && dw && ws, "star_vq_code_sums: null pointer (ws: star_vq_bwd_ws(M, E, K) floats)");
    STAR_CHECK(M > 0 && K > 0 && E > 0, "star_vq_code_sums: bad shape M=%d K=%d E=%d", M, K, E);
    hipStream_t s = (hipStream_t)stream;
    int C, rp;
    vq_bwd_plan(M, &C, &rp);
    const int G = K >= 1024 ? 8 : 2;
    const size_t lds = (size_t)4 * G * E * sizeof(float);
    hipLaunchKernelGGL(vq_bwd_codes_kernel<true>, dim3(cdiv(K, 4 * G), C), dim3(256), lds, s, x, idx, (const float*)nullptr, ws, M, E, K, G, rp);
    const int64_t n = (int64_t)K * E;
    hipLaunchKernelGGL(vq_bwd_codes_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, ws, dw, n, C, 1.0f);
    STAR_LAUNCH_CHECK("star_vq_code_sums");
    return 0;
}

extern "C" int star_vq_ema_update(const int32_t* counts, const float* dw, float* codebook, float* ema_w, float* cluster_size, int E, int K,
                                  double decay, double epsilon, void* stream) {
    STAR_CHECK(counts && dw && codebook && ema_w && cluster_size, "star_vq_ema_update: null pointer");
    STAR_CHECK(K > 0 && E > 0 && (int64_t)K * E < (1ll << 31), "star_vq_ema_update: bad shape K=%d E=%d", K, E);
    STAR_CHECK(decay > 0.0 && decay < 1.0, "star_vq_ema_update: decay %g not in (0, 1)", decay);
    hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, dw, codebook, ema_w, cluster_size, E, K,
                       (float)decay, (float)(1.0 - decay), (float)epsilon, (float)(K * epsilon));
    STAR_LAUNCH_CHECK("star_vq_ema_update");
    return 0;
}

extern "C" int star_onehot_rows(const int64_t* idx, float* out, int M, int K, void* stream) {
    STAR_CHECK(idx && out, "star_onehot_rows: null pointer");
    STAR_CHECK(M > 0 && K > 0 && K % 4 == 0, "star_onehot_rows: K=%d must be a positive multiple of 4", K);
    const int64_t n4 = (int64_t)M * (K / 4);
    hipLaunchKernelGGL(onehot_rows_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, idx, out, n4, K / 4);
    STAR_LAUNCH_CHECK("star_onehot_rows");
    return 0;
}
