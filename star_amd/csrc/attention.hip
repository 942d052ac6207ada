// K4: softmax(q k^T hd^-1/2) v for the small head dims of STAR (hd = 16, 32, 64; L = 1024 typ.).
// Reference: timm 0.3.2 Attention.forward between qkv and proj (star/models/mae_base.py:279-281,308-310).
//
// bf16 path (flash style, never materialises the L x L scores):
//   * v_mfma_f32_32x32x16_bf16 with SWAPPED operands: S^T = K Q^T puts a query on each lane, so the
//     row max / row sum are in-lane reductions plus ONE exchange with lane^32; the exponent is
//     exp2(fma(s, scale*log2e, -m*scale*log2e)) -- the softmax scale costs no extra instruction.
//   * P^T leaves the accumulator as the B operand of O^T = V^T P^T with no lane movement (the k order
//     inside an MFMA step is permuted identically on the V side, which is read from LDS with the
//     hardware-transposing ds_read_b64_tr_b16).
//   * K/V tiles of 64 keys go global -> registers -> LDS, double-buffered, one barrier per tile; the
//     row-read image is XOR-swizzled per 16-byte chunk (conflict-free ds_read_b128), the transposed-read
//     image is plain (hd <= 32) or swizzled per 64-byte half (hd = 64).
//   * With hd this small the loop is bound by the exp/VALU rate, not by the matrix pipe (128 MFMA flop
//     per exponential at hd = 32), so the VALU work per score is kept to fma, exp2, max, add, cvt.
//   * backward = two kernels with the same skeleton: dQ with the query on the lane, dK/dV with the
//     key on the lane (their accumulators feed the next MFMA directly); P is recomputed from the
//     stored log2-domain LSE; no atomics -> bitwise reproducible gradients.
//   * outputs are transposed through LDS so that every global store is a full 16-byte row chunk.
// fp32 path (STAR_F32, the reference's numerics): one query (or key) per lane on the vector ALU with
// K/V tiles broadcast from LDS; exact fp32 accumulation, used for tight parity checks.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define AT 256
#define LOG2E 1.4426950408889634f

template <int HD> struct ACfg {
    static constexpr int CH = HD / 8;                 // 16-byte chunks per row
    static constexpr int RB = HD * 2;                 // row bytes
    static constexpr int KS = HD / 16;                // MFMA k-steps over the head dim
    static constexpr int DB = HD >= 32 ? HD / 32 : 1; // 32-row blocks of d in the transposed products
    static constexpr int SH = CH == 2 ? 3 : (CH == 4 ? 2 : 1);
    static constexpr int TILE = 64 * RB;              // bytes of a 64-row tile
    static constexpr int NLD = (64 * CH + AT - 1) / AT;
};

template <int HD> __device__ __forceinline__ int row_off(int row, int chunk) {
    return row * ACfg<HD>::RB + ((chunk ^ ((row >> ACfg<HD>::SH) & (ACfg<HD>::CH - 1))) << 4);
}
template <int HD> __device__ __forceinline__ int tr_off(int row, int colbyte) {
    if constexpr (HD == 64) return row * 128 + ((((colbyte >> 6) ^ ((row >> 1) & 1)) << 6) | (colbyte & 63));
    return row * ACfg<HD>::RB + colbyte;
}

template <int HD> struct TileRegs { uint4 v[ACfg<HD>::NLD]; };

// rows [row0, row0+64) of a [*, HD] bf16 matrix whose rows are `rs` elements apart; rows >= nrows -> 0
template <int HD>
__device__ __forceinline__ void load_tile(TileRegs<HD>& t, const bf16_t* __restrict__ base, int64_t rs, int row0, int nrows, int tid) {
    constexpr int CH = ACfg<HD>::CH;
#pragma unroll
    for (int i = 0; i < ACfg<HD>::NLD; ++i) {
        const int id = tid + i * AT;
        t.v[i] = make_uint4(0, 0, 0, 0);
        if (id < 64 * CH) {
            const int r = row0 + id / CH;
            if (r < nrows) t.v[i] = *reinterpret_cast<const uint4*>(base + (int64_t)r * rs + (id % CH) * 8);
        }
    }
}
template <int HD> __device__ __forceinline__ void store_row(unsigned char* lds, const TileRegs<HD>& t, int tid) {
    constexpr int CH = ACfg<HD>::CH;
#pragma unroll
    for (int i = 0; i < ACfg<HD>::NLD; ++i) {
        const int id = tid + i * AT;
        if (id < 64 * CH) *reinterpret_cast<uint4*>(lds + row_off<HD>(id / CH, id % CH)) = t.v[i];
    }
}
template <int HD> __device__ __forceinline__ void store_tr(unsigned char* lds, const TileRegs<HD>& t, int tid) {
    constexpr int CH = ACfg<HD>::CH;
#pragma unroll
    for (int i = 0; i < ACfg<HD>::NLD; ++i) {
        const int id = tid + i * AT;
        if (id < 64 * CH) *reinterpret_cast<uint4*>(lds + tr_off<HD>(id / CH, (id % CH) * 16)) = t.v[i];
    }
}

template <int HD> __device__ __forceinline__ short8_t row_frag(const unsigned char* lds, int row, int chunk) {
    return *reinterpret_cast<const short8_t*>(lds + row_off<HD>(row, chunk));
}

__device__ __forceinline__ short4_t at_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p));
}

// A-operand fragment of a TRANSPOSED tile product: MFMA row = column (d) of the [rows][HD] LDS tile,
// MFMA k = tile row, in the k order in which a 32x32 accumulator hands over its registers 8s..8s+7:
// element j of lane half h2 <-> tile row blk + 16 s + 8 (j>>2) + 4 h2 + (j&3).
template <int HD> __device__ __forceinline__ short8_t tr_frag(const unsigned char* lds, int blk, int s, int db, int lane) {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3, h2 = g >> 1;
    const int dh = HD >= 32 ? (g & 1) : 0;
    const int row = blk + 16 * s + 4 * h2 + q4;
    const int colbyte = (db * 32 + dh * 16 + 4 * p) * 2;
    const short4_t lo = at_tr16(lds + tr_off<HD>(row, colbyte));
    const short4_t hi = at_tr16(lds + tr_off<HD>(row + 8, colbyte));
    return (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8_t cvt8(const f32x16_t& a, int s) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
    return r;
}

__device__ __forceinline__ f32x16_t mfma32(short8_t a, short8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma32(short8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), b, c, 0, 0, 0);
}

// accumulator block(s) [d][32 lanes' rows] -> dest[row][d] as bf16, 16-byte row chunks.
// wl: wave-private LDS region of 32*RB bytes.  Contains a block barrier: call uniformly.
template <int HD>
__device__ __forceinline__ void store_T(const f32x16_t (&acc)[ACfg<HD>::DB], float mul, unsigned char* wl, bf16_t* __restrict__ gbase,
                                        int64_t rs, int nvalid, int lane) {
    constexpr int CH = ACfg<HD>::CH, RB = ACfg<HD>::RB;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int db = 0; db < ACfg<HD>::DB; ++db)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int d0 = db * 32 + 8 * t + 4 * h2;
            if (d0 < HD) {
                uint2 w;
                w.x = (unsigned)f2bf(acc[db][4 * t] * mul) | ((unsigned)f2bf(acc[db][4 * t + 1] * mul) << 16);
                w.y = (unsigned)f2bf(acc[db][4 * t + 2] * mul) | ((unsigned)f2bf(acc[db][4 * t + 3] * mul) << 16);
                *reinterpret_cast<uint2*>(wl + r * RB + d0 * 2) = w;
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (32 * CH + 63) / 64; ++i) {
        const int id = lane + 64 * i;
        if (id < 32 * CH) {
            const int row = id / CH, c = id % CH;
            if (row < nvalid) *reinterpret_cast<uint4*>(gbase + (int64_t)row * rs + c * 8) = *reinterpret_cast<const uint4*>(wl + row * RB + c * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(AT) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, int L, int H, float c) {
    using C = ACfg<HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int h = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h2 = lane >> 5;
    const int64_t rs = (int64_t)3 * H * HD;
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const bf16_t* kb_ = qb + H * HD;
    const bf16_t* vb_ = qb + 2 * H * HD;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int q = q0 + r;

    short8_t qf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        qf[ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (q < L) qf[ks] = *reinterpret_cast<const short8_t*>(qb + (int64_t)q * rs + ks * 16 + 8 * h2);
    }
    f32x16_t o[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int nt = (L + 63) / 64;
    TileRegs<HD> kr, vr;
    load_tile<HD>(kr, kb_, rs, 0, L, tid);
    load_tile<HD>(vr, vb_, rs, 0, L, tid);
    store_row<HD>(smem, kr, tid);
    store_tr<HD>(smem + C::TILE, vr, tid);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const unsigned char* kl = smem + (it & 1) * 2 * C::TILE;
        const unsigned char* vl = kl + C::TILE;
        if (it + 1 < nt) {
            load_tile<HD>(kr, kb_, rs, (it + 1) * 64, L, tid);
            load_tile<HD>(vr, vb_, rs, (it + 1) * 64, L, tid);
        }
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) s[kb] = mfma32(row_frag<HD>(kl, kb * 32 + r, ks * 2 + h2), qf[ks], s[kb]);
        }
        if ((it + 1) * 64 > L) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = it * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                    if (key >= L) s[kb][e] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kb][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
        const float mc = mn * c;
        m = mn;
        l *= alpha;
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][e], c, -mc));
                s[kb][e] = pv;
                l += pv;
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = cvt8(s[kb], s2);
#pragma unroll
                for (int db = 0; db < C::DB; ++db) o[db] = mfma32(tr_frag<HD>(vl, kb * 32, s2, db, lane), pf, o[db]);
            }
        if (it + 1 < nt) {
            unsigned char* nl = smem + ((it + 1) & 1) * 2 * C::TILE;
            store_row<HD>(nl, kr, tid);
            store_tr<HD>(nl + C::TILE, vr, tid);
        }
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    if (h2 == 0 && q < L) lse[((int64_t)nb * H + h) * L + q] = m * c + log2f(l);
    store_T<HD>(o, 1.0f / l, smem + wave * 32 * C::RB, out + ((int64_t)nb * L + q0) * H * HD + h * HD, (int64_t)H * HD, L - q0, lane);
}

// delta[nb][h][l] = sum_d dO * O   (one thread per (row, head))
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, float* __restrict__ delta, int NB, int L, int H, int hd) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)NB * L * H) return;
    const int h = (int)(item % H);
    const int64_t row = item / H;
    const T* a = out + row * H * hd + h * hd;
    const T* b = dout + row * H * hd + h * hd;
    float s = 0.f;
    for (int d = 0; d < hd; d += 4) {
        float x[4], y[4];
        ld4(a + d, x);
        ld4(b + d, y);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    const int nb = (int)(row / L), l = (int)(row % L);
    delta[((int64_t)nb * H + h) * L + l] = s;
}

// dQ: query on the lane (same skeleton as the forward)
template <int HD>
__global__ __launch_bounds__(AT) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dqkv, int L, int H, float c, float scale) {
    using C = ACfg<HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][K row | K tr | V row]
    const int h = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h2 = lane >> 5;
    const int64_t rs = (int64_t)3 * H * HD;
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const bf16_t* kb_ = qb + H * HD;
    const bf16_t* vb_ = qb + 2 * H * HD;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int q = q0 + r;
    short8_t qf[C::KS], dof[C::KS];
    float ls = INFINITY, dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        qf[ks] = dof[ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (q < L) {
            qf[ks] = *reinterpret_cast<const short8_t*>(qb + (int64_t)q * rs + ks * 16 + 8 * h2);
            dof[ks] = *reinterpret_cast<const short8_t*>(dout + ((int64_t)nb * L + q) * H * HD + h * HD + ks * 16 + 8 * h2);
        }
    }
    if (q < L) {
        ls = lse[((int64_t)nb * H + h) * L + q];
        dl = delta[((int64_t)nb * H + h) * L + q];
    }
    f32x16_t dq[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[db][e] = 0.f;

    constexpr int STAGE = 3 * C::TILE;
    const int nt = (L + 63) / 64;
    TileRegs<HD> kr, vr;
    load_tile<HD>(kr, kb_, rs, 0, L, tid);
    load_tile<HD>(vr, vb_, rs, 0, L, tid);
    store_row<HD>(smem, kr, tid);
    store_tr<HD>(smem + C::TILE, kr, tid);
    store_row<HD>(smem + 2 * C::TILE, vr, tid);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const unsigned char* kl = smem + (it & 1) * STAGE;
        const unsigned char* kt = kl + C::TILE;
        const unsigned char* vl = kl + 2 * C::TILE;
        if (it + 1 < nt) {
            load_tile<HD>(kr, kb_, rs, (it + 1) * 64, L, tid);
            load_tile<HD>(vr, vb_, rs, (it + 1) * 64, L, tid);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma32(row_frag<HD>(kl, kb * 32 + r, ks * 2 + h2), qf[ks], s);
                dp = mfma32(row_frag<HD>(vl, kb * 32 + r, ks * 2 + h2), dof[ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = it * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -ls));
                if (key >= L) pv = 0.f;
                s[e] = pv * (dp[e] - dl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t dsf = cvt8(s, s2);
#pragma unroll
                for (int db = 0; db < C::DB; ++db) dq[db] = mfma32(tr_frag<HD>(kt, kb * 32, s2, db, lane), dsf, dq[db]);
            }
        }
        if (it + 1 < nt) {
            unsigned char* nl = smem + ((it + 1) & 1) * STAGE;
            store_row<HD>(nl, kr, tid);
            store_tr<HD>(nl + C::TILE, kr, tid);
            store_row<HD>(nl + 2 * C::TILE, vr, tid);
        }
        __syncthreads();
    }
    store_T<HD>(dq, scale, smem + wave * 32 * C::RB, dqkv + ((int64_t)nb * L + q0) * rs + h * HD, rs, L - q0, lane);
}

// dK, dV: key on the lane; sweeps the queries in tiles of 64
template <int HD>
__global__ __launch_bounds__(AT) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, int L, int H, float c, float scale) {
    using C = ACfg<HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][Q row | Q tr | dO row | dO tr | lse[64] | delta[64]]
    const int h = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h2 = lane >> 5;
    const int64_t rs = (int64_t)3 * H * HD, ro = (int64_t)H * HD;
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const bf16_t* kb_ = qb + H * HD;
    const bf16_t* vb_ = qb + 2 * H * HD;
    const bf16_t* dob = dout + (int64_t)nb * L * ro + h * HD;
    const float* lsb = lse + ((int64_t)nb * H + h) * L;
    const float* dlb = delta + ((int64_t)nb * H + h) * L;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int key = k0 + r;
    short8_t kf[C::KS], vf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        kf[ks] = vf[ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (key < L) {
            kf[ks] = *reinterpret_cast<const short8_t*>(kb_ + (int64_t)key * rs + ks * 16 + 8 * h2);
            vf[ks] = *reinterpret_cast<const short8_t*>(vb_ + (int64_t)key * rs + ks * 16 + 8 * h2);
        }
    }
    f32x16_t dk[C::DB], dv[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dk[db][e] = dv[db][e] = 0.f;

    constexpr int STAGE = 4 * C::TILE + 512;
    const int nt = (L + 63) / 64;
    TileRegs<HD> qr, dr;
    float lsr = INFINITY, dlr = 0.f;
    auto fetch = [&](int t0) {
        load_tile<HD>(qr, qb, rs, t0, L, tid);
        load_tile<HD>(dr, dob, ro, t0, L, tid);
        lsr = INFINITY;
        dlr = 0.f;
        if (tid < 64 && t0 + tid < L) {
            lsr = lsb[t0 + tid];
            dlr = dlb[t0 + tid];
        }
    };
    auto stash = [&](unsigned char* st) {
        store_row<HD>(st, qr, tid);
        store_tr<HD>(st + C::TILE, qr, tid);
        store_row<HD>(st + 2 * C::TILE, dr, tid);
        store_tr<HD>(st + 3 * C::TILE, dr, tid);
        if (tid < 64) {
            reinterpret_cast<float*>(st + 4 * C::TILE)[tid] = lsr;
            reinterpret_cast<float*>(st + 4 * C::TILE + 256)[tid] = dlr;
        }
    };
    fetch(0);
    stash(smem);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const unsigned char* st = smem + (it & 1) * STAGE;
        const unsigned char* ql = st;
        const unsigned char* qt = st + C::TILE;
        const unsigned char* dol = st + 2 * C::TILE;
        const unsigned char* dot = st + 3 * C::TILE;
        const float* lsl = reinterpret_cast<const float*>(st + 4 * C::TILE);
        const float* dll = lsl + 64;
        if (it + 1 < nt) fetch((it + 1) * 64);
#pragma unroll
        for (int qblk = 0; qblk < 2; ++qblk) {
            f32x16_t s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma32(row_frag<HD>(ql, qblk * 32 + r, ks * 2 + h2), kf[ks], s);
                dp = mfma32(row_frag<HD>(dol, qblk * 32 + r, ks * 2 + h2), vf[ks], dp);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 l4 = *reinterpret_cast<const float4*>(lsl + qblk * 32 + 8 * t + 4 * h2);
                const float4 d4 = *reinterpret_cast<const float4*>(dll + qblk * 32 + 8 * t + 4 * h2);
                const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * t + e], c, -la[e]));
                    s[4 * t + e] = pv;
                    dp[4 * t + e] = pv * (dp[4 * t + e] - da[e]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = cvt8(s, s2), dsf = cvt8(dp, s2);
#pragma unroll
                for (int db = 0; db < C::DB; ++db) {
                    dv[db] = mfma32(tr_frag<HD>(dot, qblk * 32, s2, db, lane), pf, dv[db]);
                    dk[db] = mfma32(tr_frag<HD>(qt, qblk * 32, s2, db, lane), dsf, dk[db]);
                }
            }
        }
        if (it + 1 < nt) stash(smem + ((it + 1) & 1) * STAGE);
        __syncthreads();
    }
    bf16_t* dkb = dqkv + ((int64_t)nb * L + k0) * rs + H * HD + h * HD;
    store_T<HD>(dk, scale, smem + wave * 32 * C::RB, dkb, rs, L - k0, lane);
    store_T<HD>(dv, 1.0f, smem + (4 + wave) * 32 * C::RB, dkb + H * HD, rs, L - k0, lane);
}

// ------------------------------------------------------------------------------------------------------
// fp32 path: vector-ALU flash attention, one query per lane; K/V tiles of 32 rows broadcast from LDS.
#define FT 128
template <int HD>
__global__ __launch_bounds__(FT) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                          int L, int H, float c) {
    __shared__ float sk[32][HD], sv[32][HD];
    const int h = blockIdx.y, nb = blockIdx.z, tid = threadIdx.x;
    const int64_t rs = (int64_t)3 * H * HD;
    const float* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const int q = blockIdx.x * FT + tid;
    float qv[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qv[d] = q < L ? qb[(int64_t)q * rs + d] : 0.f;
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < L; k0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * HD; e += FT) {
            const int kr = e / HD, d = e % HD;
            const bool ok = k0 + kr < L;
            sk[kr][d] = ok ? qb[(int64_t)(k0 + kr) * rs + H * HD + d] : 0.f;
            sv[kr][d] = ok ? qb[(int64_t)(k0 + kr) * rs + 2 * H * HD + d] : 0.f;
        }
        __syncthreads();
        const int nk = min(32, L - k0);
        float s[32];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = __builtin_fmaf(qv[d], sk[j][d], a);
            s[j] = j < nk ? a : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f((m - mn) * c);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float pv = exp2f((s[j] - mn) * c);
            l += pv;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = __builtin_fmaf(pv, sv[j][d], o[d]);
        }
    }
    if (q < L) {
        const float inv = 1.0f / l;
        float* op = out + ((int64_t)nb * L + q) * H * HD + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = o[d] * inv;
        lse[((int64_t)nb * H + h) * L + q] = m * c + log2f(l);
    }
}

// role 0: lane = query, computes dQ sweeping keys; role 1: lane = key, computes dK and dV sweeping queries
template <int HD, int ROLE>
__global__ __launch_bounds__(FT) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dqkv, int L, int H, float c, float scale) {
    __shared__ float sa[32][HD], sb[32][HD], sl[32], sd[32];
    const int h = blockIdx.y, nb = blockIdx.z, tid = threadIdx.x;
    const int64_t rs = (int64_t)3 * H * HD, ro = (int64_t)H * HD;
    const float* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const float* dob = dout + (int64_t)nb * L * ro + h * HD;
    const float* lsb = lse + ((int64_t)nb * H + h) * L;
    const float* dlb = delta + ((int64_t)nb * H + h) * L;
    const int me = blockIdx.x * FT + tid;
    float a[HD], b[HD], g1[HD], g2[HD];
    float ls = INFINITY, dl = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        g1[d] = g2[d] = 0.f;
        if (ROLE == 0) {  // q, dO of my query
            a[d] = me < L ? qb[(int64_t)me * rs + d] : 0.f;
            b[d] = me < L ? dob[(int64_t)me * ro + d] : 0.f;
        } else {          // k, v of my key
            a[d] = me < L ? qb[(int64_t)me * rs + H * HD + d] : 0.f;
            b[d] = me < L ? qb[(int64_t)me * rs + 2 * H * HD + d] : 0.f;
        }
    }
    if (ROLE == 0 && me < L) {
        ls = lsb[me];
        dl = dlb[me];
    }
    for (int t0 = 0; t0 < L; t0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * HD; e += FT) {
            const int rr = e / HD, d = e % HD;
            const bool ok = t0 + rr < L;
            if (ROLE == 0) {  // K, V rows
                sa[rr][d] = ok ? qb[(int64_t)(t0 + rr) * rs + H * HD + d] : 0.f;
                sb[rr][d] = ok ? qb[(int64_t)(t0 + rr) * rs + 2 * H * HD + d] : 0.f;
            } else {          // Q, dO rows
                sa[rr][d] = ok ? qb[(int64_t)(t0 + rr) * rs + d] : 0.f;
                sb[rr][d] = ok ? dob[(int64_t)(t0 + rr) * ro + d] : 0.f;
            }
        }
        if (ROLE == 1 && tid < 32) {
            sl[tid] = t0 + tid < L ? lsb[t0 + tid] : INFINITY;
            sd[tid] = t0 + tid < L ? dlb[t0 + tid] : 0.f;
        }
        __syncthreads();
        const int nr = min(32, L - t0);
        for (int j = 0; j < nr; ++j) {
            float s = 0.f, dp = 0.f;
            if (ROLE == 0) {  // s = q.k_j ; dp = dO.v_j
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s = __builtin_fmaf(a[d], sa[j][d], s);
                    dp = __builtin_fmaf(b[d], sb[j][d], dp);
                }
                const float pv = exp2f(s * c - ls);
                const float ds = pv * (dp - dl);
#pragma unroll
                for (int d = 0; d < HD; ++d) g1[d] = __builtin_fmaf(ds, sa[j][d], g1[d]);
            } else {          // s = q_j.k ; dp = dO_j.v
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s = __builtin_fmaf(sa[j][d], a[d], s);
                    dp = __builtin_fmaf(sb[j][d], b[d], dp);
                }
                const float pv = exp2f(s * c - sl[j]);
                const float ds = pv * (dp - sd[j]);
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    g1[d] = __builtin_fmaf(ds, sa[j][d], g1[d]);   // dK += dS q
                    g2[d] = __builtin_fmaf(pv, sb[j][d], g2[d]);   // dV += P dO
                }
            }
        }
    }
    if (me < L) {
        float* dst = dqkv + ((int64_t)nb * L + me) * rs + h * HD;
        if (ROLE == 0) {
#pragma unroll
            for (int d = 0; d < HD; ++d) dst[d] = g1[d] * scale;
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dst[H * HD + d] = g1[d] * scale;
                dst[2 * H * HD + d] = g2[d];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
template <typename K> static int set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return star_set_error(-2, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return 0;
}

template <int HD>
static int fwd_bf16(const void* qkv, void* out, float* lse, int NB, int L, int H, float c, hipStream_t s) {
    const size_t lds = 4 * ACfg<HD>::TILE;
    if (int rc = set_lds(attn_fwd_bf16_kernel<HD>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_bf16_kernel<HD>, dim3(cdiv(L, 128), H, NB), dim3(AT), lds, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, c);
    return 0;
}
template <int HD>
static int bwd_bf16(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int NB, int L, int H, float c,
                    float scale, hipStream_t s) {
    const size_t lds_q = 6 * ACfg<HD>::TILE;
    const size_t lds_kv = 2 * (4 * ACfg<HD>::TILE + 512);
    if (int rc = set_lds(attn_bwd_dq_bf16_kernel<HD>, lds_q)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_bf16_kernel<HD>, lds_kv)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<HD>, dim3(cdiv(L, 128), H, NB), dim3(AT), lds_q, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       delta, (bf16_t*)dqkv, L, H, c, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<HD>, dim3(cdiv(L, 128), H, NB), dim3(AT), lds_kv, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       delta, (bf16_t*)dqkv, L, H, c, scale);
    return 0;
}
template <int HD>
static void fwd_f32(const void* qkv, void* out, float* lse, int NB, int L, int H, float c, hipStream_t s) {
    hipLaunchKernelGGL(attn_fwd_f32_kernel<HD>, dim3(cdiv(L, FT), H, NB), dim3(FT), 0, s, (const float*)qkv, (float*)out, lse, L, H, c);
}
template <int HD>
static void bwd_f32(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int NB, int L, int H, float c,
                    float scale, hipStream_t s) {
    hipLaunchKernelGGL((attn_bwd_f32_kernel<HD, 0>), dim3(cdiv(L, FT), H, NB), dim3(FT), 0, s, (const float*)qkv, (const float*)dout, lse, delta,
                       (float*)dqkv, L, H, c, scale);
    hipLaunchKernelGGL((attn_bwd_f32_kernel<HD, 1>), dim3(cdiv(L, FT), H, NB), dim3(FT), 0, s, (const float*)qkv, (const float*)dout, lse, delta,
                       (float*)dqkv, L, H, c, scale);
}

static int attn_check(const char* who, int NB, int L, int H, int hd, int dt) {
    STAR_CHECK(NB > 0 && L > 0 && H > 0, "%s: bad shape NB=%d L=%d H=%d", who, NB, L, H);
    STAR_CHECK(hd == 16 || hd == 32 || hd == 64, "%s: head dim %d not in {16,32,64}", who, hd);
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32, "%s: unknown dtype %d", who, dt);
    STAR_CHECK(H <= 65535 && NB <= 65535, "%s: H, NB too large for the grid", who);
    return 0;
}

extern "C" int star_attn_fwd(const void* qkv, void* out, float* lse, int NB, int L, int H, int hd, int dt, void* stream) {
    STAR_CHECK(qkv && out && lse, "star_attn_fwd: null pointer");
    if (int rc = attn_check("star_attn_fwd", NB, L, H, hd, dt)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float c = (float)((double)LOG2E / sqrt((double)hd));
    int rc = 0;
    if (dt == STAR_BF16) {
        if (hd == 16) rc = fwd_bf16<16>(qkv, out, lse, NB, L, H, c, s);
        else if (hd == 32) rc = fwd_bf16<32>(qkv, out, lse, NB, L, H, c, s);
        else rc = fwd_bf16<64>(qkv, out, lse, NB, L, H, c, s);
    } else {
        if (hd == 16) fwd_f32<16>(qkv, out, lse, NB, L, H, c, s);
        else if (hd == 32) fwd_f32<32>(qkv, out, lse, NB, L, H, c, s);
        else fwd_f32<64>(qkv, out, lse, NB, L, H, c, s);
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_attn_fwd");
    return 0;
}

extern "C" int star_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                             int NB, int L, int H, int hd, int dt, void* stream) {
    STAR_CHECK(qkv && out && dout && lse && delta && dqkv, "star_attn_bwd: null pointer");
    if (int rc = attn_check("star_attn_bwd", NB, L, H, hd, dt)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float scale = (float)(1.0 / sqrt((double)hd));
    const float c = (float)((double)LOG2E / sqrt((double)hd));
    const int64_t items = (int64_t)NB * L * H;
    int rc = 0;
    if (dt == STAR_BF16) {
        hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3(cdiv(items, 256)), dim3(256), 0, s, (const bf16_t*)out, (const bf16_t*)dout, delta, NB, L, H, hd);
        STAR_LAUNCH_CHECK("star_attn_bwd(delta)");
        if (hd == 16) rc = bwd_bf16<16>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else if (hd == 32) rc = bwd_bf16<32>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else rc = bwd_bf16<64>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(cdiv(items, 256)), dim3(256), 0, s, (const float*)out, (const float*)dout, delta, NB, L, H, hd);
        STAR_LAUNCH_CHECK("star_attn_bwd(delta)");
        if (hd == 16) bwd_f32<16>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else if (hd == 32) bwd_f32<32>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else bwd_f32<64>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_attn_bwd");
    return 0;
}
