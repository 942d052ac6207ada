// Row gather / scatter of the float32 residual stream, grouped column sums and the transposed weight
// cast: the data-movement pieces of the sub-sampling mask and the temporal un-mixing
// (star/models/multiagent_mae.py:816-867) that are not folded into a GEMM's row maps.
// HBM-bound: 16-byte accesses, consecutive lanes on consecutive columns of one row.
#include "common.h"

template <typename T>
__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows, float* __restrict__ dst_f32,
                                   T* __restrict__ dst_lp, int M, int D) {
    const int dq = D >> 2;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)M * dq) return;
    const int m = (int)(item / dq), c = (int)(item % dq) << 2;
    const int r = rows[m];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r >= 0) ld4(src + (size_t)r * D + c, v);
    if (dst_f32) st4(dst_f32 + (size_t)m * D + c, v);
    if (dst_lp) st4(dst_lp + (size_t)m * D + c, v);
}

extern "C" int star_gather_rows(const float* src, const int32_t* rows, float* dst_f32, void* dst_lp, int M, int D, int dt, void* stream) {
    STAR_CHECK(src && rows && (dst_f32 || dst_lp) && D % 4 == 0, "star_gather_rows: bad arguments");
    const int64_t items = (int64_t)M * (D / 4);
    if (items == 0) return 0;
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, src, rows, dst_f32, (bf16_t*)dst_lp, M, D);
    else
        hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, src, rows, dst_f32, (float*)dst_lp, M, D);
    STAR_LAUNCH_CHECK("star_gather_rows");
    return 0;
}

template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ rows, float* __restrict__ dst, int M, int D) {
    const int dq = D >> 2;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)M * dq) return;
    const int m = (int)(item / dq), c = (int)(item % dq) << 2;
    const int r = rows[m];
    if (r < 0) return;
    float v[4];
    ld4(src + (size_t)m * D + c, v);
    st4(dst + (size_t)r * D + c, v);
}

extern "C" int star_scatter_rows(const void* src, int src_dt, const int32_t* rows, float* dst, int M, int D, void* stream) {
    STAR_CHECK(src && rows && dst && D % 4 == 0, "star_scatter_rows: bad arguments");
    const int64_t items = (int64_t)M * (D / 4);
    if (items == 0) return 0;
    if (src_dt == STAR_BF16)
        hipLaunchKernelGGL(scatter_rows_kernel<bf16_t>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, rows, dst, M, D);
    else
        hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, rows, dst, M, D);
    STAR_LAUNCH_CHECK("star_scatter_rows");
    return 0;
}

// block = 256 columns (4 per lane) x 16 row lanes (waves); grid (column chunks, groups, row splits); every wave keeps eight
// row loads in flight; fixed-order folds (waves through LDS, then the second stage) -> deterministic
template <typename T>
__global__ __launch_bounds__(1024) void colsum_groups_kernel(const T* __restrict__ src, const int32_t* __restrict__ sel, int sel_nonneg,
                                                             float* __restrict__ part, int R, int D, int div, int mod, int rows_per) {
    __shared__ float s[16][256];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int grp = blockIdx.y;
    const int r0 = blockIdx.z * rows_per, r1 = min(R, r0 + rows_per);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + rl; r < r1; r += 16 * 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = r + u * 16;
            bool ok = c < D && rr < r1 && (rr / div) % mod == grp;
            if (ok && sel) {
                const int sv = sel[rr];
                ok = sel_nonneg ? sv >= 0 : sv != 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = 0.f;
            if (ok) ld4(src + (size_t)rr * D + c, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += v[u][k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[rl][lane * 4 + k] = a[k];
    __syncthreads();
    const int t = threadIdx.x, col = blockIdx.x * 256 + t;
    if (t < 256 && col < D) {
        float f = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) f += s[w][t];
        part[((size_t)blockIdx.z * mod + grp) * D + col] = f;
    }
}
__global__ void colsum_groups_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int S) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(size_t)s * n + e];
    out[e] = a;
}

extern "C" int star_colsum_groups(const void* src, int dt, const int32_t* sel, int sel_nonneg, int R, int D, int div, int mod, float* ws,
                                  float* out, void* stream) {
    STAR_CHECK(src && ws && out && R > 0 && D > 0 && D % 4 == 0 && div > 0 && mod > 0 && mod <= 65535, "star_colsum_groups: bad arguments (D %% 4 == 0)");
    int S = cdiv(R, 256);
    if (S > 64) S = 64;
    const int rows_per = cdiv(R, S);
    dim3 grid(cdiv(D, 256), mod, S);
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(colsum_groups_kernel<bf16_t>, grid, dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)src, sel, sel_nonneg, ws, R, D, div, mod, rows_per);
    else
        hipLaunchKernelGGL(colsum_groups_kernel<float>, grid, dim3(1024), 0, (hipStream_t)stream, (const float*)src, sel, sel_nonneg, ws, R, D, div, mod, rows_per);
    STAR_LAUNCH_CHECK("star_colsum_groups");
    hipLaunchKernelGGL(colsum_groups_reduce_kernel, dim3(cdiv((int64_t)mod * D, 256)), dim3(256), 0, (hipStream_t)stream, ws, out, mod * D, S);
    STAR_LAUNCH_CHECK("star_colsum_groups(reduce)");
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < R && c0 + tx < C) tile[r][tx] = src[(size_t)(r0 + r) * C + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < C && r0 + tx < R) Act<T>::st(dst + (size_t)(c0 + c) * R + r0 + tx, tile[tx][c]);
}

extern "C" int star_cast_transpose(const float* src, void* dst, int R, int C, int dt, void* stream) {
    STAR_CHECK(src && dst && R > 0 && C > 0, "star_cast_transpose: bad arguments");
    dim3 grid(cdiv(C, 32), cdiv(R, 32));
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C);
    else
        hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, C);
    STAR_LAUNCH_CHECK("star_cast_transpose");
    return 0;
}
