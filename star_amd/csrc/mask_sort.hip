// K5: mask index build (stable argsort of the noise rows and its inverse) and the row maps that turn
// the reference's gather / cat / slice / where index arithmetic into plain int32 row tables.
// Reference: star/models/multiagent_mae.py:743-867, star/models/mae_base.py:408-433.
#include "common.h"

#define MS_THREADS 256

__device__ __forceinline__ unsigned ms_orderable(float f) {
    if (f == 0.0f) f = 0.0f;  // -0 and +0 compare equal in a sort
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One block per row.  Bitonic network over (value, position) pairs packed in 64 bits: the position
// in the low word makes every key unique, so the result is the stable ascending order.
__global__ __launch_bounds__(MS_THREADS) void mask_sort_kernel(const float* __restrict__ noise,
                                                               int64_t* __restrict__ ids_shuffle,
                                                               int64_t* __restrict__ ids_restore, int L, int P) {
    extern __shared__ unsigned long long s_key[];
    const int r = blockIdx.x;
    const float* row = noise + (size_t)r * L;
    for (int i = threadIdx.x; i < P; i += MS_THREADS)
        s_key[i] = i < L ? (((unsigned long long)ms_orderable(row[i]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += MS_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = s_key[lo], b = s_key[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < L; i += MS_THREADS) {
        const int src = (int)(unsigned)(s_key[i] & 0xffffffffu);
        if (ids_shuffle) ids_shuffle[(size_t)r * L + i] = src;
        if (ids_restore) ids_restore[(size_t)r * L + src] = i;
    }
}

extern "C" int star_mask_sort(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, int R, int L, void* stream) {
    STAR_CHECK(noise && (ids_shuffle || ids_restore), "star_mask_sort: null pointer");
    STAR_CHECK(R >= 0 && L >= 1 && L <= 4096, "star_mask_sort: need 1 <= L <= 4096 (got %d)", L);
    if (R == 0) return 0;
    int P = 2;
    while (P < L) P <<= 1;
    hipLaunchKernelGGL(mask_sort_kernel, dim3(R), dim3(MS_THREADS), (size_t)P * 8, (hipStream_t)stream, noise,
                       ids_shuffle, ids_restore, L, P);
    STAR_LAUNCH_CHECK("star_mask_sort");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// compressed-row maps: one thread per compressed row (ba, t, j)
__global__ void rowmap_rows_kernel(const int64_t* __restrict__ ids_shuffle, const int64_t* __restrict__ ids_restore,
                                   int32_t* __restrict__ src_row, int32_t* __restrict__ dst_row, int BA, int T,
                                   int L, int keep, int random_mode, int single) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (int64_t)BA * T * keep) return;
    const int j = (int)(r % keep);
    const int f = (int)(r / keep);  // frame = ba*T + t
    const int t = f % T, ba = f / T;
    int src = -1, dst = -1;
    if (!random_mode) {
        const int pos = t * keep + j;  // position in the shared shuffled order
        if (pos < L) {
            const int tok = (int)ids_shuffle[(size_t)ba * L + pos];
            src = f * L + tok;
            if (!(single && t > 0)) dst = ba * L + tok;
        }
    } else {
        const int tok = (int)ids_shuffle[(size_t)f * L + j];
        src = f * L + tok;
        bool used = !(single && t > 0);
        for (int tp = 0; tp < t && used; ++tp)  // an earlier timestamp that kept the token wins
            if (ids_restore[(size_t)(ba * T + tp) * L + tok] < keep) used = false;
        if (used) dst = ba * L + tok;
    }
    if (src_row) src_row[r] = src;
    if (dst_row) dst_row[r] = dst;
}

// per-token products: mask (0 = kept) and the rows that no timestamp covers
__global__ void rowmap_tokens_kernel(const int64_t* __restrict__ ids_restore, float* __restrict__ mask,
                                     int32_t* __restrict__ fill, int BA, int T, int L, int keep, int random_mode,
                                     int single) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)BA * L) return;
    const int tok = (int)(e % L), ba = (int)(e / L);
    bool covered = false;
    for (int t = 0; t < T; ++t) {
        bool kept;
        if (!random_mode) {
            const int pos = (int)ids_restore[(size_t)ba * L + tok];
            const int hi = (t + 1) * keep < L ? (t + 1) * keep : L;
            kept = pos >= t * keep && pos < hi;
        } else {
            kept = ids_restore[(size_t)(ba * T + t) * L + tok] < keep;
        }
        if (mask) mask[(size_t)(ba * T + t) * L + tok] = kept ? 0.0f : 1.0f;
        if (kept && !(single && t > 0)) covered = true;
    }
    if (fill) fill[e] = covered ? 0 : 1;
}

extern "C" int star_build_rowmaps(const int64_t* ids_shuffle, const int64_t* ids_restore, int32_t* src_row,
                                  int32_t* dst_row, float* mask, int32_t* fill, int BA, int T, int L, int keep,
                                  int random_mode, int single, void* stream) {
    STAR_CHECK(ids_shuffle && ids_restore, "star_build_rowmaps: null ids");
    STAR_CHECK(BA > 0 && T > 0 && L > 0 && keep > 0 && keep <= L, "star_build_rowmaps: bad sizes BA=%d T=%d L=%d keep=%d", BA, T, L, keep);
    STAR_CHECK((int64_t)BA * T * L < 2147483647LL, "star_build_rowmaps: row index overflows int32");
    const int64_t rows = (int64_t)BA * T * keep;
    hipLaunchKernelGGL(rowmap_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, ids_shuffle,
                       ids_restore, src_row, dst_row, BA, T, L, keep, random_mode, single);
    STAR_LAUNCH_CHECK("star_build_rowmaps(rows)");
    if (mask || fill) {
        hipLaunchKernelGGL(rowmap_tokens_kernel, dim3(cdiv((int64_t)BA * L, 256)), dim3(256), 0, (hipStream_t)stream,
                           ids_restore, mask, fill, BA, T, L, keep, random_mode, single);
        STAR_LAUNCH_CHECK("star_build_rowmaps(tokens)");
    }
    return 0;
}
