// Error reporting, dtype casts, the fused AdamW step, grad-norm, and the small row-movement kernels of
// the temporal un-mixing (K7) that the GEMM row maps do not cover.
#include "common.h"

static thread_local char g_err[512] = "";

int star_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int star_num_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

extern "C" const char* star_last_error(void) { return g_err; }
extern "C" int star_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float v[4];
        ld4(src + i * 4, v);
        st4(dst + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
        Act<T>::st(dst + i, src[i]);
    }
}
template <typename T>
__global__ void cast_back_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float v[4];
        ld4(src + i * 4, v);
        st4(dst + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
        dst[i] = Act<T>::ld(src + i);
    }
}

static inline int stream_grid(int64_t items, int threads) {
    int64_t g = (items + threads - 1) / threads;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int star_cast(const float* src, void* dst, int64_t n, int dt, void* stream) {
    STAR_CHECK(src && dst && n >= 0, "star_cast: bad arguments");
    if (n == 0) return 0;
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4, n);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, n / 4, n);
    STAR_LAUNCH_CHECK("star_cast");
    return 0;
}

// All weight copies of a training step in ONE launch: job j converts float32 [rows, cols] to dt, as is or transposed
// ([cols, rows]), in 64 x 64 tiles; tile_start is the running tile count (jobs sorted, last entry = total).
template <typename T, typename D>
__device__ __forceinline__ void cast_tile(const StarCastJob& job, D* __restrict__ dst, float (&tile)[64][65], int r0, int c0) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ld = job.dst_ld > 0 ? job.dst_ld : (job.transpose ? job.rows : job.cols);
    dst += job.dst_off;
    if (!job.transpose) {
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + i, c = c0 + tx;
            if (r < job.rows && c < job.cols) {
                const int sr = job.src_rows ? job.src_rows[r] : r;
                const float v = job.src[(size_t)sr * job.cols + c];
                Act<D>::st(dst + (size_t)r * ld + c, v);
                if constexpr (sizeof(D) == 2) {      // bf16 pair: the remainder goes `cols` elements further along the row
                    if (job.split) Act<D>::st(dst + (size_t)r * ld + job.cols + c, v - bf2f(f2bf(v)));
                }
            }
        }
        return;
    }
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < job.rows && c < job.cols) {
            const int sr = job.src_rows ? job.src_rows[r] : r;
            v = job.src[(size_t)sr * job.cols + c];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < job.rows && c < job.cols) Act<D>::st(dst + (size_t)c * ld + r, tile[tx][i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_batch_kernel(const StarCastJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile_start) ++j;
    const StarCastJob job = jobs[j];
    const int tc = (job.cols + 63) >> 6;
    const int t = blockIdx.x - job.tile_start, r0 = (t / tc) << 6, c0 = (t % tc) << 6;
    if (job.dst_f32) cast_tile<T, float>(job, (float*)job.dst, tile, r0, c0);      // block-uniform branch
    else cast_tile<T, T>(job, (T*)job.dst, tile, r0, c0);
}

extern "C" int star_cast_batch(const StarCastJob* jobs_dev, int njobs, int total_tiles, int dt, void* stream) {
    STAR_CHECK(jobs_dev && njobs > 0 && total_tiles > 0, "star_cast_batch: bad arguments");
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32, "star_cast_batch: unknown dtype %d", dt);
    if (dt == STAR_BF16) hipLaunchKernelGGL(cast_batch_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    else hipLaunchKernelGGL(cast_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    STAR_LAUNCH_CHECK("star_cast_batch");
    return 0;
}

extern "C" int star_cast_back(const void* src, float* dst, int64_t n, int dt, void* stream) {
    STAR_CHECK(src && dst && n >= 0, "star_cast_back: bad arguments");
    if (n == 0) return 0;
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(cast_back_kernel<bf16_t>, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n / 4, n);
    else
        hipLaunchKernelGGL(cast_back_kernel<float>, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, n / 4, n);
    STAR_LAUNCH_CHECK("star_cast_back");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// torch.optim.AdamW (non-amsgrad, non-maximize) element-wise, same operation order as the reference's
// single-tensor path: decay, moment updates, bias corrections, step.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             const uint8_t* __restrict__ wd_mask, const float* __restrict__ gscale, int64_t n, float lr,
                             float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
    const float gs = gscale ? gscale[0] : 1.0f;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i];
        const float gi = g[i] * gs;
        if (wd != 0.f && (!wd_mask || wd_mask[i])) pi = pi * (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);  // lerp, as torch does
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

extern "C" int star_adamw(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, const float* gscale,
                          int64_t n, float lr, float beta1, float beta2, float eps, float wd, int step, void* stream) {
    STAR_CHECK(p && g && m && v && n >= 0 && step >= 1, "star_adamw: bad arguments");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, gscale, n,
                       lr, beta1, beta2, eps, wd, (float)bc1, (float)sqrt(bc2));
    STAR_LAUNCH_CHECK("star_adamw");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
    __shared__ float s[4];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(1024) void sumsq_finish_kernel(const float* __restrict__ ws, int nb, float* __restrict__ out) {
    __shared__ double s[1024];
    s[threadIdx.x] = threadIdx.x < nb ? (double)ws[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)s[0];
}

extern "C" int star_sumsq(const float* x, int64_t n, float* ws, float* out, void* stream) {
    STAR_CHECK(x && ws && out && n >= 0, "star_sumsq: bad arguments");
    int nb = stream_grid(n, 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, ws);
    STAR_LAUNCH_CHECK("star_sumsq");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ws, nb, out);
    STAR_LAUNCH_CHECK("star_sumsq(finish)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// decoder-input rows that no timestamp covers: mask_token (+ temp embed of t = 0) (+ pos)
__global__ void fill_rows_kernel(float* __restrict__ x, const int32_t* __restrict__ fill, const float* __restrict__ mask_token,
                                 const float* __restrict__ temp0, const float* __restrict__ pos, int rows, int L, int D) {
    const int dq = D >> 2;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)rows * dq) return;
    const int row = (int)(item / dq), c = (int)(item % dq) << 2;
    if (!fill[row]) return;
    float v[4], t[4];
    ld4(mask_token + c, v);
    if (temp0) {
        ld4(temp0 + c, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += t[k];
    }
    if (pos) {
        ld4(pos + (size_t)(row % L) * D + c, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += t[k];
    }
    st4(x + (size_t)row * D + c, v);
}

extern "C" int star_fill_rows(float* x, const int32_t* fill, const float* mask_token, const float* temp0, const float* pos,
                              int BA, int L, int D, void* stream) {
    STAR_CHECK(x && fill && mask_token && D % 4 == 0, "star_fill_rows: bad arguments");
    const int64_t items = (int64_t)BA * L * (D / 4);
    if (items == 0) return 0;
    hipLaunchKernelGGL(fill_rows_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, x, fill, mask_token, temp0, pos, BA * L, L, D);
    STAR_LAUNCH_CHECK("star_fill_rows");
    return 0;
}

// Q2: dst viewed flat per agent-scene is the transpose [D][L] of src [L][D]; pos is added in dst's
// (reinterpreted) [L][D] indexing.  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void transpose_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             const float* __restrict__ pos, int L, int D, int bwd) {
    __shared__ float tile[32][33];
    const int ba = blockIdx.z;
    const int l0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const size_t base = (size_t)ba * L * D;
    if (!bwd) {
        for (int r = ty; r < 32; r += 8)
            if (l0 + r < L && d0 + tx < D) tile[r][tx] = src[base + (size_t)(l0 + r) * D + d0 + tx];
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (d0 + r < D && l0 + tx < L) {
                const size_t flat = (size_t)(d0 + r) * L + l0 + tx;
                dst[base + flat] = tile[tx][r] + (pos ? pos[flat] : 0.f);
            }
    } else {  // gradient: gsrc[l][d] = gdst_flat[d*L + l]
        for (int r = ty; r < 32; r += 8)
            if (d0 + r < D && l0 + tx < L) tile[r][tx] = src[base + (size_t)(d0 + r) * L + l0 + tx];
        __syncthreads();
        for (int r = ty; r < 32; r += 8)
            if (l0 + r < L && d0 + tx < D) dst[base + (size_t)(l0 + r) * D + d0 + tx] = tile[tx][r];
    }
}

extern "C" int star_transpose_rows(const float* src, float* dst, const float* pos, int BA, int L, int D, void* stream) {
    STAR_CHECK(src && dst && BA > 0 && L > 0 && D > 0 && BA <= 65535, "star_transpose_rows: bad arguments");
    hipLaunchKernelGGL(transpose_rows_kernel, dim3(cdiv(L, 32), cdiv(D, 32), BA), dim3(256), 0, (hipStream_t)stream, src, dst, pos, L, D, 0);
    STAR_LAUNCH_CHECK("star_transpose_rows");
    return 0;
}
extern "C" int star_transpose_rows_bwd(const float* gdst, float* gsrc, int BA, int L, int D, void* stream) {
    STAR_CHECK(gdst && gsrc && BA > 0 && L > 0 && D > 0 && BA <= 65535, "star_transpose_rows_bwd: bad arguments");
    hipLaunchKernelGGL(transpose_rows_kernel, dim3(cdiv(L, 32), cdiv(D, 32), BA), dim3(256), 0, (hipStream_t)stream, gdst, gsrc, (const float*)nullptr, L, D, 1);
    STAR_LAUNCH_CHECK("star_transpose_rows_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// K11: three-scale occupancy confusion counts (star/utils/metrics.py:146-190).
// MaxPool3d((2,2,2), stride 2, padding (1,0,0)) on [Z,H,W]: output z' covers input z in {2z'-1, 2z'}
// (out-of-range rows are -inf padding), so Z=13 -> 7 -> 4.  One thread per scale-4 cell gathers its
// 4x4x4 (depth: up to 4 valid) block and derives all three scales from it.
__global__ __launch_bounds__(256) void iou_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  unsigned long long* __restrict__ conf, int N, int Z, int H, int W) {
    __shared__ unsigned int s_cnt[12];
    if (threadIdx.x < 12) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int Z2 = Z / 2 + 1, Z4 = Z2 / 2 + 1, H4 = H / 4, W4 = W / 4;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned cnt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) cnt[i] = 0;
    if (item < (int64_t)N * Z4 * H4 * W4) {
        const int x4 = (int)(item % W4);
        const int y4 = (int)((item / W4) % H4);
        const int z4 = (int)((item / ((int64_t)W4 * H4)) % Z4);
        const int n = (int)(item / ((int64_t)W4 * H4 * Z4));
        const size_t base = (size_t)n * Z * H * W;
        float p4 = -INFINITY, t4 = -INFINITY;
        bool any4 = false;
        for (int dz2 = 0; dz2 < 2; ++dz2) {
            const int z2 = 2 * z4 - 1 + dz2;
            if (z2 < 0 || z2 >= Z2) continue;
            for (int dy2 = 0; dy2 < 2; ++dy2)
                for (int dx2 = 0; dx2 < 2; ++dx2) {
                    const int y2 = 2 * y4 + dy2, x2 = 2 * x4 + dx2;
                    float p2 = -INFINITY, t2 = -INFINITY;
                    for (int dz = 0; dz < 2; ++dz) {
                        const int z = 2 * z2 - 1 + dz;
                        if (z < 0 || z >= Z) continue;
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx) {
                                const size_t o = base + ((size_t)z * H + 2 * y2 + dy) * W + 2 * x2 + dx;
                                const float pv = pred[o], tv = target[o];
                                p2 = fmaxf(p2, pv);
                                t2 = fmaxf(t2, tv);
                            }
                    }
                    p4 = fmaxf(p4, p2);
                    t4 = fmaxf(t4, t2);
                    any4 = true;
                    // scale 2 cell (z2,y2,x2) belongs to exactly one scale-4 cell: z4 = (z2+1)/2
                    cnt[4 + ((p2 >= 0.5f) ? 2 : 0) + ((long long)t2 != 0 ? 1 : 0)] += 1;
                }
        }
        if (any4) cnt[8 + ((p4 >= 0.5f) ? 2 : 0) + ((long long)t4 != 0 ? 1 : 0)] += 1;
    }
#pragma unroll
    for (int i = 4; i < 12; ++i)
        if (cnt[i]) atomicAdd(&s_cnt[i], cnt[i]);
    __syncthreads();
    if (threadIdx.x >= 4 && threadIdx.x < 12 && s_cnt[threadIdx.x]) atomicAdd(&conf[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// scale 1: plain streaming count
__global__ __launch_bounds__(256) void iou1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                   unsigned long long* __restrict__ conf, int64_t n) {
    __shared__ unsigned int s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned cnt[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        cnt[((pred[i] >= 0.5f) ? 2 : 0) + ((long long)target[i] != 0 ? 1 : 0)] += 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned c = cnt[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[k], c);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&conf[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

extern "C" int star_iou_confusion(const float* pred, const float* target, unsigned long long* conf, int N, int Z, int H,
                                  int W, void* stream) {
    STAR_CHECK(pred && target && conf, "star_iou_confusion: null pointer");
    STAR_CHECK(N > 0 && Z > 0 && H % 4 == 0 && W % 4 == 0, "star_iou_confusion: H and W must be multiples of 4");
    const int64_t n = (int64_t)N * Z * H * W;
    hipLaunchKernelGGL(iou1_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, conf, n);
    STAR_LAUNCH_CHECK("star_iou_confusion(scale1)");
    const int Z2 = Z / 2 + 1, Z4 = Z2 / 2 + 1;
    const int64_t cells = (int64_t)N * Z4 * (H / 4) * (W / 4);
    hipLaunchKernelGGL(iou_kernel, dim3(cdiv(cells, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, conf, N, Z, H, W);
    STAR_LAUNCH_CHECK("star_iou_confusion(scale2/4)");
    return 0;
}
