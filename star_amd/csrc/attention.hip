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
    static constexpr int RPP = 1024 / RB;             // rows per 1-KiB LDS-DMA piece
    static constexpr int NP = 64 / RPP;               // pieces per 64-row image
};

template <int HD> __device__ __forceinline__ int row_off(int row, int chunk) {
    return row * ACfg<HD>::RB + ((chunk ^ ((row >> ACfg<HD>::SH) & (ACfg<HD>::CH - 1))) << 4);
}
template <int HD> __device__ __forceinline__ int tr_off(int row, int colbyte) {
    return row * ACfg<HD>::RB + ((((colbyte >> 4) ^ ((row >> ACfg<HD>::SH) & (ACfg<HD>::CH - 1))) << 4) | (colbyte & 15));
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
// K/V (or Q/dO) tiles stream global -> LDS with buffer_load ... lds: no staging registers, and -- the point --
// no compiler-placed vmcnt(0) in the loop.  One wave instruction fills 1 KiB = RPP rows of one 64-row image;
// a lane's 16 bytes land lane-linearly, so the image's chunk swizzle is applied to the SOURCE address.  Rows
// past the end of the sequence are zero-filled by the descriptor's range check.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
template <int HD> __device__ __forceinline__ unsigned dma_src(int piece, int lane, int64_t row_bytes) {
    using C = ACfg<HD>;
    const int row = piece * C::RPP + lane / C::CH, pc = lane % C::CH;
    return (unsigned)(row * row_bytes + ((pc ^ ((row >> C::SH) & (C::CH - 1))) << 4));
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N <= 63, "wait_vm");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// a global load that must be complete before the DMA ring starts counting: consuming the registers here keeps the
// compiler from re-inserting its own (conservative, full) vmcnt wait inside the loop
__device__ __forceinline__ void pin(short8_t& v) {
    typedef __attribute__((ext_vector_type(4))) int i4;
    i4 t = __builtin_bit_cast(i4, v);
    asm volatile("" : "+v"(t));
    v = __builtin_bit_cast(short8_t, t);
}
// single-instruction 3-way maximum (the compiler's fmaxf puts a canonicalising v_max in front of every MFMA output)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float xor32_max(float x) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    const unsigned u = __float_as_uint(x);
    const u2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// the four 16-row k-slices of a 64-row image (rows +0, +16, +32, +48), low and high 8-row halves
template <int RB> __device__ __forceinline__ void tr_issue(unsigned a_lo, unsigned a_hi, short4_t (&lo)[4], short4_t (&hi)[4]) {
    lo[0] = tr16<0>(a_lo);        hi[0] = tr16<0>(a_hi);
    lo[1] = tr16<16 * RB>(a_lo);  hi[1] = tr16<16 * RB>(a_hi);
    lo[2] = tr16<32 * RB>(a_lo);  hi[2] = tr16<32 * RB>(a_hi);
    lo[3] = tr16<48 * RB>(a_lo);  hi[3] = tr16<48 * RB>(a_hi);
}
// two 16-row k-slices (32 rows) only
template <int RB> __device__ __forceinline__ void tr_issue_half(unsigned a_lo, unsigned a_hi, short4_t (&lo)[2], short4_t (&hi)[2]) {
    lo[0] = tr16<0>(a_lo);        hi[0] = tr16<0>(a_hi);
    lo[1] = tr16<16 * RB>(a_lo);  hi[1] = tr16<16 * RB>(a_hi);
}
__device__ __forceinline__ void tr_wait_half(short4_t (&a)[2], short4_t (&b)[2], short4_t (&c)[2], short4_t (&d)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]), "+v"(d[0]), "+v"(d[1]));
}
__device__ __forceinline__ void tr_wait(short4_t (&lo)[4], short4_t (&hi)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int OFF> __device__ __forceinline__ f32x4_t lds_f4(unsigned addr) {
    f32x4_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <int OFF> __device__ __forceinline__ f32x2_t lds_f2(unsigned addr) {
    f32x2_t r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ void lds_w4(unsigned addr, f32x4_t v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF));
}

__device__ __forceinline__ short8_t join8(short4_t lo, short4_t hi) { return (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

#define ONES8 ((short8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})
#define RESCALE_TH 8.0f   // log2 units: the running reference maximum is only raised when a row exceeds it by 2^8

// Forward.  Stage (ring of 3) = K image | V image | [hd 16: 2 KiB of bf16 ones].  The softmax denominator comes
// out of the matrix pipe: a ones row block appended to V^T (hd 16: the idle MFMA rows 16..31 read the ones area;
// hd >= 32: four in-register add chains, cheaper than an extra MFMA chain on the busy matrix pipe).
// SPL (dt = STAR_BF16X3): q, k, v and the output are bf16 PAIRS (a row is [hi | lo], the lo plane 3 H hd -- output: H hd --
// elements further along); S = k_hi q_hi + k_lo q_hi + k_hi q_lo, the probabilities are split the same way (p_hi = bf16(p),
// p_lo = bf16(p - p_hi)) and O = v_hi p_hi + v_lo p_hi + v_hi p_lo: three MFMAs per product, float32-grade results.  A stage then
// carries the lo images of K and V behind the hi ones, and the row sums stay in registers (exact float32) for every head dim.
template <int HD, int QB, bool SPL = false>
__global__ __launch_bounds__(AT) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, int L, int H, float c) {
    using C = ACfg<HD>;
    constexpr int NI = 2 * C::NP / 4 * (SPL ? 2 : 1);           // DMA instructions per wave per tile
    constexpr int LO = 2 * C::TILE + (HD == 16 ? 2048 : 0);     // SPL: offset of the lo images inside a stage
    constexpr int STAGE = LO * (SPL ? 2 : 1);
    constexpr int QW = 32 * QB, QBLK = 4 * QW;                  // queries per wave / per block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware work order: the row blocks of one (sequence, head) stream the same K/V (Q/dO) tiles, so they
    // run back to back on ONE XCD and share its L2 instead of fetching the tiles into all eight
    const int nrb = (L + QBLK - 1) / QBLK;
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int rblk = work % nrb, h = (work / nrb) % H, nb = work / (nrb * H);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t rs = (int64_t)3 * H * HD * (SPL ? 2 : 1);
    const int64_t lo_el = (int64_t)3 * H * HD;        // SPL: the lo plane of a qkv row
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const int q0 = rblk * QBLK + wave * QW;   // the wave's QB query blocks of 32 rows start here

    // QB query blocks per wave share every K fragment and V^T fragment read from LDS, every tile DMA and barrier, and
    // give the in-order wave independent MFMA / softmax chains to interleave
    short8_t qf[QB][C::KS], qfl[SPL ? QB : 1][C::KS];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int q = q0 + qi * 32 + r;
            qf[qi][ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
            if (q < L) qf[qi][ks] = *reinterpret_cast<const short8_t*>(qb + (int64_t)q * rs + ks * 16 + 8 * h2);
            if constexpr (SPL) {
                qfl[qi][ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
                if (q < L) qfl[qi][ks] = *reinterpret_cast<const short8_t*>(qb + (int64_t)q * rs + lo_el + ks * 16 + 8 * h2);
            }
        }
    if constexpr (HD == 16) {
        for (int i = tid; i < 3 * 128; i += AT) *reinterpret_cast<uint4*>(smem + (i >> 7) * STAGE + 2 * C::TILE + (i & 127) * 16) =
            make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
        __syncthreads();
    }
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            pin(qf[qi][ks]);
            if constexpr (SPL) pin(qfl[qi][ks]);
        }

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qb), 0, (unsigned)((L * rs - h * HD) * 2), 0x00020000);
    unsigned src[NI];
    int dst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        // pieces 0 .. 2 NP - 1: the K then the V image; SPL: pieces 2 NP .. 4 NP - 1 the same of the lo plane
        const int p4 = wave * NI + j, islo = SPL && p4 >= 2 * C::NP, p = islo ? p4 - 2 * C::NP : p4;
        const int isv = p >= C::NP, pp = isv ? p - C::NP : p;
        src[j] = dma_src<HD>(pp, lane, rs * 2) + (unsigned)((1 + isv) * H * HD * 2) + (islo ? (unsigned)(lo_el * 2) : 0u);
        dst[j] = islo * LO + isv * C::TILE + pp * 1024;
    }
    const unsigned tile_step = (unsigned)(64 * rs * 2);
    auto issue = [&](int t, int stage) {
#pragma unroll
        for (int j = 0; j < NI; ++j) dma16(rsrc, smem + stage * STAGE + dst[j], src[j] + (unsigned)t * tile_step);
    };

    f32x16_t o[QB][C::DB];
    float lsum[QB][4];        // hd >= 32: row sums as four independent add chains (hd 16: out of the matrix pipe)
    float m[QB];              // reference maximum the exponentials are taken against (raw score units)
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        m[qi] = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) lsum[qi][u] = 0.f;
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qi][db][e] = 0.f;
    }

    // per-lane fragment offsets inside a stage
    int koff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) koff[ks] = row_off<HD>(r, ks * 2 + h2);
    const int g = lane >> 4, i16 = lane & 15;
    const bool ones_lane = HD == 16 && !SPL && (g & 1);
    int voff[2][C::DB];       // [row | row+8][d block]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
            const int row = 4 * (g >> 1) + (i16 >> 2) + 8 * u;
            const int colbyte = (db * 32 + (HD >= 32 ? (g & 1) : 0) * 16 + 4 * (i16 & 3)) * 2;
            voff[u][db] = ones_lane ? 2 * C::TILE + (i16 & 3) * 8 + u * 32 : C::TILE + tr_off<HD>(row, colbyte);
        }

    const unsigned lds_base = lds_u32(smem);
    const int nt = (L + 63) / 64;
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) wait_vm<NI>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 2 < nt) issue(it + 2, stage >= 1 ? stage - 1 : 2);
        const unsigned char* st = smem + stage * STAGE;
        const unsigned sbase = lds_base + stage * STAGE;
        stage = stage == 2 ? 0 : stage + 1;

        f32x16_t s[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[qi][kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const short8_t kf = *reinterpret_cast<const short8_t*>(st + koff[ks] + kb * 32 * C::RB);
                if constexpr (SPL) {
                    const short8_t kfl = *reinterpret_cast<const short8_t*>(st + LO + koff[ks] + kb * 32 * C::RB);
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) {
                        s[qi][kb] = mfma32(kfl, qf[qi][ks], s[qi][kb]);
                        s[qi][kb] = mfma32(kf, qfl[qi][ks], s[qi][kb]);
                    }
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) s[qi][kb] = mfma32(kf, qf[qi][ks], s[qi][kb]);
            }
        }
        // V^T fragments for the whole tile: issued now, waited on just before the PV products
        short4_t vlo[C::DB][4], vhi[C::DB][4];
        short4_t wlo[SPL ? C::DB : 1][4], whi[SPL ? C::DB : 1][4];      // SPL: the same fragments of the lo image of V
#pragma unroll
        for (int db = 0; db < C::DB; ++db) tr_issue<C::RB>(sbase + voff[0][db], sbase + voff[1][db], vlo[db], vhi[db]);
        if constexpr (SPL) {
#pragma unroll
            for (int db = 0; db < C::DB; ++db) tr_issue<C::RB>(sbase + LO + voff[0][db], sbase + LO + voff[1][db], wlo[db], whi[db]);
        }
        if ((it + 1) * 64 > L) {   // ragged last tile only (uniform branch)
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = it * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                        if (key >= L) s[qi][kb][e] = -INFINITY;
                    }
        }
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float mx = max3(s[qi][0][0], s[qi][0][1], s[qi][1][0]);
            float mx2 = max3(s[qi][1][1], s[qi][0][2], s[qi][0][3]);
#pragma unroll
            for (int e = 4; e < 16; e += 4) {
                mx = max3(mx, s[qi][0][e], s[qi][0][e + 1]);
                mx2 = max3(mx2, s[qi][0][e + 2], s[qi][0][e + 3]);
            }
#pragma unroll
            for (int e = 2; e < 16; e += 4) {
                mx = max3(mx, s[qi][1][e], s[qi][1][e + 1]);
                if (e + 3 < 16) mx2 = max3(mx2, s[qi][1][e + 2], s[qi][1][e + 3]);
            }
            mx = xor32_max(max3(mx, mx2, s[qi][1][15]));
            if (__builtin_amdgcn_ballot_w64((mx - m[qi]) * c > RESCALE_TH) != 0) {   // rare after the first tiles (uniform branch)
                const float mn = fmaxf(m[qi], mx);
                const float alpha = __builtin_amdgcn_exp2f((m[qi] - mn) * c);
                m[qi] = mn;
#pragma unroll
                for (int db = 0; db < C::DB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[qi][db][e] *= alpha;
                if constexpr (HD >= 32 || SPL) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) lsum[qi][u] *= alpha;
                }
            }
            const float mc = m[qi] * c;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[qi][kb][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qi][kb][e], c, -mc));
                    if constexpr (HD >= 32 || SPL) lsum[qi][e & 3] += s[qi][kb][e];
                }
        }
        tr_wait(vlo[0], vhi[0]);
        if constexpr (C::DB == 2) tr_wait(vlo[1], vhi[1]);
        if constexpr (SPL) {
            tr_wait(wlo[0], whi[0]);
            if constexpr (C::DB == 2) tr_wait(wlo[1], whi[1]);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8_t ph[QB], pl[SPL ? QB : 1];
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    ph[qi] = cvt8(s[qi][kb], s2);
                    if constexpr (SPL) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pl[qi][j] = (__bf16)(s[qi][kb][8 * s2 + j] - (float)ph[qi][j]);
                    }
                }
#pragma unroll
                for (int db = 0; db < C::DB; ++db) {
                    const short8_t vf = join8(vlo[db][kb * 2 + s2], vhi[db][kb * 2 + s2]);
                    if constexpr (SPL) {
                        const short8_t vfl = join8(wlo[db][kb * 2 + s2], whi[db][kb * 2 + s2]);
#pragma unroll
                        for (int qi = 0; qi < QB; ++qi) {
                            o[qi][db] = mfma32(vfl, ph[qi], o[qi][db]);
                            o[qi][db] = mfma32(vf, pl[qi], o[qi][db]);
                        }
                    }
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) o[qi][db] = mfma32(vf, ph[qi], o[qi][db]);
                }
            }
    }
    __syncthreads();   // every wave is done with the ring before it becomes the store staging area
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const int qq = q0 + qi * 32;
        float l = o[qi][0][8];   // hd 16: the ones rows of V^T hold the row sums (both lane halves already combined by the MFMA)
        if constexpr (HD >= 32 || SPL) {
            l = (lsum[qi][0] + lsum[qi][1]) + (lsum[qi][2] + lsum[qi][3]);
            l += __shfl_xor(l, 32, 64);
        }
        if (h2 == 0 && qq + r < L) lse[((int64_t)nb * H + h) * L + qq + r] = m[qi] * c + log2f(l);
        if (qi > 0) __syncthreads();   // the staging rows of the previous block have been read
        if constexpr (SPL) {
            const float inv = 1.0f / l;
            f32x16_t ol[C::DB];
#pragma unroll
            for (int db = 0; db < C::DB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = o[qi][db][e] * inv;
                    o[qi][db][e] = v;
                    ol[db][e] = v - bf2f(f2bf(v));
                }
            bf16_t* orow = out + ((int64_t)nb * L + qq) * 2 * H * HD + h * HD;
            store_T<HD>(o[qi], 1.0f, smem + wave * 32 * C::RB, orow, (int64_t)2 * H * HD, L - qq, lane);
            __syncthreads();
            store_T<HD>(ol, 1.0f, smem + wave * 32 * C::RB, orow + H * HD, (int64_t)2 * H * HD, L - qq, lane);
        } else {
            store_T<HD>(o[qi], 1.0f / l, smem + wave * 32 * C::RB, out + ((int64_t)nb * L + qq) * H * HD + h * HD, (int64_t)H * HD, L - qq, lane);
        }
    }
}

// ws[0][nb][h][l] = sum_d dO * O (bf16 path: negated); bf16 path also ws[1][nb][h][l] = -lse / c, so that the
// dK/dV kernel can start its S and dP accumulators from the row constants (one thread per (row, head))
template <typename T, bool NEG>
__global__ void attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, const float* __restrict__ lse,
                                  float* __restrict__ ws, int NB, int L, int H, int hd, float inv_c) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)NB * L * H;
    if (item >= total) return;
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
    const int64_t o = ((int64_t)nb * H + h) * L + l;
    if (NEG) {
        ws[o] = -s;
        ws[total + o] = -lse[o] * inv_c;
    } else {
        ws[o] = s;
    }
}


// sum_j o[j] * d[j] over 8 bf16 pairs as one fma chain in element order (the dQ kernel folds the row dots into its prologue):
// half0 = chain over elements 16 ks + 0..7 (ks ascending), half1 = chain over 16 ks + 8..15, dot = half0 + half1
__device__ __forceinline__ float attn_dot8(uint4 o, uint4 d, float acc) {
    const unsigned x[4] = {o.x, o.y, o.z, o.w}, y[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc = __builtin_fmaf(__uint_as_float(x[k] << 16), __uint_as_float(y[k] << 16), acc);
        acc = __builtin_fmaf(__uint_as_float(x[k] & 0xffff0000u), __uint_as_float(y[k] & 0xffff0000u), acc);
    }
    return acc;
}

template <int HD>
__global__ __launch_bounds__(AT) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ ndelta_out, float* __restrict__ nlsc_out,
                                                              bf16_t* __restrict__ dqkv, int L, int H, float c, float scale, float inv_c) {
    using C = ACfg<HD>;
    constexpr int NI = 2 * C::NP / 4;
    constexpr int STAGE = 2 * C::TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware work order: the L/128 row blocks of one (sequence, head) stream the same K/V (Q/dO) tiles, so they
    // run back to back on ONE XCD and share its L2 instead of fetching the tiles into all eight
    const int nrb = (L + 127) >> 7;
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int rblk = work % nrb, h = (work / nrb) % H, nb = work / (nrb * H);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t rs = (int64_t)3 * H * HD;
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const int q0 = rblk * 128 + wave * 32;
    const int q = q0 + r;
    // The row dots delta = sum_d dO . O are computed HERE (each lane half over its 8 of every 16 head dims, the halves exchanged
    // once) instead of by a table pass in front of the backward; the lanes of half 0 leave -delta and -lse / c in the tables the
    // dK/dV kernel (launched behind this one) starts its accumulators from.
    short8_t qf[C::KS], dof[C::KS];
    float ls = INFINITY, part = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        qf[ks] = dof[ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (q < L) {
            const int64_t ro = ((int64_t)nb * L + q) * H * HD + h * HD + ks * 16 + 8 * h2;
            qf[ks] = *reinterpret_cast<const short8_t*>(qb + (int64_t)q * rs + ks * 16 + 8 * h2);
            dof[ks] = *reinterpret_cast<const short8_t*>(dout + ro);
            part = attn_dot8(*reinterpret_cast<const uint4*>(out + ro), __builtin_bit_cast(uint4, dof[ks]), part);
        }
    }
    if (q < L) ls = lse[((int64_t)nb * H + h) * L + q];
    float ndl;
    {
        typedef __attribute__((ext_vector_type(2))) unsigned u2;
        const unsigned u = __float_as_uint(part);
        const u2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        // sw = (lanes 0-31: [own, partner's], lanes 32-63: [partner's, own]) -> half0 + half1 on every lane
        ndl = -(__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    }
    if (h2 == 0 && q < L) {
        ndelta_out[((int64_t)nb * H + h) * L + q] = ndl;
        nlsc_out[((int64_t)nb * H + h) * L + q] = -ls * inv_c;
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        pin(qf[ks]);
        pin(dof[ks]);
    }
    asm volatile("" : "+v"(ls), "+v"(ndl));
    f32x16_t ndl16;
#pragma unroll
    for (int e = 0; e < 16; ++e) ndl16[e] = ndl;
    asm volatile("" : "+v"(ndl16));   // materialised once: not re-splatted per block

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qb), 0, (unsigned)((L * rs - h * HD) * 2), 0x00020000);
    unsigned src[NI];
    int dst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int p = wave * NI + j, isv = p >= C::NP, pp = isv ? p - C::NP : p;
        src[j] = dma_src<HD>(pp, lane, rs * 2) + (unsigned)((1 + isv) * H * HD * 2);
        dst[j] = isv * C::TILE + pp * 1024;
    }
    const unsigned tile_step = (unsigned)(64 * rs * 2);
    auto issue = [&](int t, int stage) {
#pragma unroll
        for (int j = 0; j < NI; ++j) dma16(rsrc, smem + stage * STAGE + dst[j], src[j] + (unsigned)t * tile_step);
    };

    f32x16_t dq[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[db][e] = 0.f;
    int koff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) koff[ks] = row_off<HD>(r, ks * 2 + h2);
    const int g = lane >> 4, i16 = lane & 15;
    int toff[2][C::DB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
            toff[u][db] = tr_off<HD>(4 * (g >> 1) + (i16 >> 2) + 8 * u, (db * 32 + (HD >= 32 ? (g & 1) : 0) * 16 + 4 * (i16 & 3)) * 2);

    const unsigned lds_base = lds_u32(smem);
    const int nt = (L + 63) / 64;
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) wait_vm<NI>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 2 < nt) issue(it + 2, stage >= 1 ? stage - 1 : 2);
        const unsigned char* kl = smem + stage * STAGE;
        const unsigned char* vl = kl + C::TILE;
        const unsigned sbase = lds_base + stage * STAGE;
        stage = stage == 2 ? 0 : stage + 1;

        short4_t klo[C::DB][4], khi[C::DB][4];
#pragma unroll
        for (int db = 0; db < C::DB; ++db) tr_issue<C::RB>(sbase + toff[0][db], sbase + toff[1][db], klo[db], khi[db]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // dP starts from the lane's -delta (a 16-register splat kept for the whole kernel, handed to the first MFMA as its
            // C operand): dS = p (dP - delta) costs no add per score
            f32x16_t s, dp = ndl16;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma32(*reinterpret_cast<const short8_t*>(kl + koff[ks] + kb * 32 * C::RB), qf[ks], s);
                dp = mfma32(*reinterpret_cast<const short8_t*>(vl + koff[ks] + kb * 32 * C::RB), dof[ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -ls)) * dp[e];
            if (kb == 0) {
                tr_wait(klo[0], khi[0]);
                if constexpr (C::DB == 2) tr_wait(klo[1], khi[1]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t dsf = cvt8(s, s2);
#pragma unroll
                for (int db = 0; db < C::DB; ++db) dq[db] = mfma32(join8(klo[db][kb * 2 + s2], khi[db][kb * 2 + s2]), dsf, dq[db]);
            }
        }
    }
    __syncthreads();
    store_T<HD>(dq, scale, smem + wave * 32 * C::RB, dqkv + ((int64_t)nb * L + q0) * rs + h * HD, rs, L - q0, lane);
}

// dK, dV: key on the lane; sweeps the queries in tiles of 64.  Stage = Q image | dO image | -lse/c [64] | -delta [64];
// the S and dP accumulators START from the row constants, so p = exp2(c S') and dS = p dP' need no subtraction.
// Query rows past L are zero (Q, dO and both constants): they add nothing to dK, dV.
template <int HD>
__global__ __launch_bounds__(AT) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ nlsc, const float* __restrict__ ndelta,
                                                               bf16_t* __restrict__ dqkv, int L, int H, float c, float scale) {
    using C = ACfg<HD>;
    constexpr int NI = 2 * C::NP / 4;
    constexpr int STAGE = 2 * C::TILE + 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware work order: the L/128 row blocks of one (sequence, head) stream the same K/V (Q/dO) tiles, so they
    // run back to back on ONE XCD and share its L2 instead of fetching the tiles into all eight
    const int nrb = (L + 127) >> 7;
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int rblk = work % nrb, h = (work / nrb) % H, nb = work / (nrb * H);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t rs = (int64_t)3 * H * HD, ro = (int64_t)H * HD;
    const bf16_t* qb = qkv + (int64_t)nb * L * rs + h * HD;
    const bf16_t* dob = dout + (int64_t)nb * L * ro + h * HD;
    const int k0 = rblk * 128 + wave * 32;
    const int key = k0 + r;
    short8_t kf[C::KS], vf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        kf[ks] = vf[ks] = (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (key < L) {
            kf[ks] = *reinterpret_cast<const short8_t*>(qb + H * HD + (int64_t)key * rs + ks * 16 + 8 * h2);
            vf[ks] = *reinterpret_cast<const short8_t*>(qb + 2 * H * HD + (int64_t)key * rs + ks * 16 + 8 * h2);
        }
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        pin(kf[ks]);
        pin(vf[ks]);
    }

    __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qb), 0, (unsigned)((L * rs - h * HD) * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dob), 0, (unsigned)((L * ro - h * HD) * 2), 0x00020000);
    const float* cst = ((wave & 1) ? ndelta : nlsc) + ((int64_t)nb * H + h) * L;
    __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cst), 0, (unsigned)(L * 4), 0x00020000);
    unsigned src[NI];
    int dst[NI];
    bool isd[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int p = j * 4 + wave;   // hd >= 32: compile-time split between the two descriptors
        isd[j] = p >= C::NP;
        const int pp = isd[j] ? p - C::NP : p;
        src[j] = dma_src<HD>(pp, lane, (isd[j] ? ro : rs) * 2);
        dst[j] = isd[j] * C::TILE + pp * 1024;
    }
    const unsigned step_q = (unsigned)(64 * rs * 2), step_d = (unsigned)(64 * ro * 2);
    const int cdst = 2 * C::TILE + (wave & 1) * 256;
    auto issue = [&](int t, int stage) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (isd[j]) dma16(rsd, sb + dst[j], src[j] + (unsigned)t * step_d);
            else dma16(rsq, sb + dst[j], src[j] + (unsigned)t * step_q);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (lds_ptr_t)(sb + cdst), 4, (unsigned)((t * 64 + lane) * 4), 0, 0, 0);
    };

    f32x16_t dk[C::DB], dv[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dk[db][e] = dv[db][e] = 0.f;
    int koff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) koff[ks] = row_off<HD>(r, ks * 2 + h2);
    const int g = lane >> 4, i16 = lane & 15;
    int toff[2][C::DB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
            toff[u][db] = tr_off<HD>(4 * (g >> 1) + (i16 >> 2) + 8 * u, (db * 32 + (HD >= 32 ? (g & 1) : 0) * 16 + 4 * (i16 & 3)) * 2);

    const unsigned lds_base = lds_u32(smem);
    const int nt = (L + 63) / 64;
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) wait_vm<NI + 1>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 2 < nt) issue(it + 2, stage >= 1 ? stage - 1 : 2);
        const unsigned char* ql = smem + stage * STAGE;
        const unsigned char* dol = ql + C::TILE;
        const unsigned sbase = lds_base + stage * STAGE;
        stage = stage == 2 ? 0 : stage + 1;

#pragma unroll
        for (int qblk = 0; qblk < 2; ++qblk) {
            // transposed Q / dO fragments of this 32-query half: issued now, waited on before the dV / dK products
            short4_t qlo[C::DB][2], qhi[C::DB][2], dlo[C::DB][2], dhi[C::DB][2];
#pragma unroll
            for (int db = 0; db < C::DB; ++db) {
                const unsigned hb = sbase + qblk * 32 * C::RB;
                tr_issue_half<C::RB>(hb + toff[0][db], hb + toff[1][db], qlo[db], qhi[db]);
                tr_issue_half<C::RB>(hb + C::TILE + toff[0][db], hb + C::TILE + toff[1][db], dlo[db], dhi[db]);
            }
            // row constants: asm reads (a compiler-visible read of a region filled by LDS-DMA makes hipcc drain the whole
            // ring with vmcnt(0) in every iteration); they land in the accumulators' registers, no VALU copy
            f32x16_t s, dp;
            {
                const unsigned ca = sbase + 2 * C::TILE + (qblk * 32 + 4 * h2) * 4;
                const f32x4_t l0 = lds_f4<0>(ca), l1 = lds_f4<32>(ca), l2 = lds_f4<64>(ca), l3 = lds_f4<96>(ca);
                const f32x4_t d0 = lds_f4<256>(ca), d1 = lds_f4<288>(ca), d2 = lds_f4<320>(ca), d3 = lds_f4<352>(ca);
                s = __builtin_shufflevector(__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7),
                                            __builtin_shufflevector(l2, l3, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                dp = __builtin_shufflevector(__builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7),
                                             __builtin_shufflevector(d2, d3, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(dp));
            }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma32(*reinterpret_cast<const short8_t*>(ql + koff[ks] + qblk * 32 * C::RB), kf[ks], s);
                dp = mfma32(*reinterpret_cast<const short8_t*>(dol + koff[ks] + qblk * 32 * C::RB), vf[ks], dp);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __builtin_amdgcn_exp2f(s[e] * c);
                dp[e] *= s[e];
            }
#pragma unroll
            for (int db = 0; db < C::DB; ++db) tr_wait_half(qlo[db], qhi[db], dlo[db], dhi[db]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = cvt8(s, s2), dsf = cvt8(dp, s2);
#pragma unroll
                for (int db = 0; db < C::DB; ++db) {
                    dv[db] = mfma32(join8(dlo[db][s2], dhi[db][s2]), pf, dv[db]);
                    dk[db] = mfma32(join8(qlo[db][s2], qhi[db][s2]), dsf, dk[db]);
                }
            }
        }
    }
    __syncthreads();
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
    const size_t lds = 3 * (2 * ACfg<HD>::TILE + (HD == 16 ? 2048 : 0));
    // QB = 2 (two query blocks per wave, K/V fragments shared) measured slower on MI355X: 166 vs 96 VGPRs costs two of
    // the five waves per SIMD, and occupancy hides more latency here than the extra in-wave parallelism does
    if (int rc = set_lds(attn_fwd_bf16_kernel<HD, 1>, lds)) return rc;
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, 1>), dim3(cdiv(L, 128) * H * NB), dim3(AT), lds, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, c);
    return 0;
}
template <int HD>
static int fwd_split(const void* qkv, void* out, float* lse, int NB, int L, int H, float c, hipStream_t s) {
    const size_t lds = 3 * 2 * (2 * ACfg<HD>::TILE + (HD == 16 ? 2048 : 0));
    if (int rc = set_lds(attn_fwd_bf16_kernel<HD, 1, true>, lds)) return rc;
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<HD, 1, true>), dim3(cdiv(L, 128) * H * NB), dim3(AT), lds, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, c);
    return 0;
}
template <int HD>
static int bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, const float* ws, void* dqkv, int NB, int L, int H,
                    float c, float scale, hipStream_t s) {
    const float* delta = ws;                           // negated
    const float* nlsc = ws + (int64_t)NB * H * L;
    const size_t lds_q = 3 * 2 * ACfg<HD>::TILE;
    const size_t lds_kv = 3 * (2 * ACfg<HD>::TILE + 512);
    if (int rc = set_lds(attn_bwd_dq_bf16_kernel<HD>, lds_q)) return rc;
    if (int rc = set_lds(attn_bwd_dkv_bf16_kernel<HD>, lds_kv)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<HD>, dim3(cdiv(L, 128) * H * NB), dim3(AT), lds_q, s, (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, const_cast<float*>(delta), const_cast<float*>(nlsc), (bf16_t*)dqkv, L, H, c, scale, 1.0f / c);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<HD>, dim3(cdiv(L, 128) * H * NB), dim3(AT), lds_kv, s, (const bf16_t*)qkv, (const bf16_t*)dout, nlsc,
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
    STAR_CHECK(dt == STAR_BF16 || dt == STAR_F32 || dt == STAR_BF16X3, "%s: unknown dtype %d", who, dt);
    STAR_CHECK(H <= 65535 && NB <= 65535 && (int64_t)cdiv(L, 128) * H * NB < (1ll << 31), "%s: H, NB, L too large for the grid", who);
    return 0;
}

extern "C" int star_attn_fwd(const void* qkv, void* out, float* lse, int NB, int L, int H, int hd, int dt, void* stream) {
    STAR_CHECK(qkv && out && lse, "star_attn_fwd: null pointer");
    if (int rc = attn_check("star_attn_fwd", NB, L, H, hd, dt)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float c = (float)((double)LOG2E / sqrt((double)hd));
    int rc = 0;
    if (dt == STAR_BF16X3) {
        STAR_CHECK((long long)L * 6 * H * hd * 2 < 0x7fffffffLL, "star_attn_fwd(bf16x3): sequence over 2 GiB");
        if (hd == 16) rc = fwd_split<16>(qkv, out, lse, NB, L, H, c, s);
        else if (hd == 32) rc = fwd_split<32>(qkv, out, lse, NB, L, H, c, s);
        else rc = fwd_split<64>(qkv, out, lse, NB, L, H, c, s);
    } else if (dt == STAR_BF16) {
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
    STAR_CHECK(dt != STAR_BF16X3, "star_attn_bwd: STAR_BF16X3 is a forward-only storage type");
    hipStream_t s = (hipStream_t)stream;
    const float scale = (float)(1.0 / sqrt((double)hd));
    const float c = (float)((double)LOG2E / sqrt((double)hd));
    const int64_t items = (int64_t)NB * L * H;
    int rc = 0;
    if (dt == STAR_BF16) {
        // two kernels: the dQ kernel computes the row dots in its prologue and leaves the tables for the dK/dV kernel behind it
        if (hd == 16) rc = bwd_bf16<16>(qkv, out, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else if (hd == 32) rc = bwd_bf16<32>(qkv, out, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else rc = bwd_bf16<64>(qkv, out, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
    } else {
        hipLaunchKernelGGL((attn_delta_kernel<float, false>), dim3(cdiv(items, 256)), dim3(256), 0, s, (const float*)out, (const float*)dout, lse, delta, NB, L, H, hd, 1.0f / c);
        STAR_LAUNCH_CHECK("star_attn_bwd(delta)");
        if (hd == 16) bwd_f32<16>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else if (hd == 32) bwd_f32<32>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
        else bwd_f32<64>(qkv, dout, lse, delta, dqkv, NB, L, H, c, scale, s);
    }
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_attn_bwd");
    return 0;
}
