// K3': the transformer block's MLP forward as ONE kernel -- x_out = x_res + fc2(gelu(fc1(h))) (timm 0.3.2 Mlp inside Block,
// reference star/models/mae_base.py:279-281,308-310: fc1 -> GELU(erf) -> fc2, plus the block's second residual add).
//
// The hidden activations never make the HBM round trip between the two GEMMs.  A wave owns 32 rows (tokens) for its whole
// life and walks the hidden dimension in chunks of 32:
//   * GEMM 1 is computed SWAPPED, S^T[hid][row] = W1[hid][:] . h^T[:][row] (v_mfma_f32_32x32x16_bf16, A = W1 rows from LDS,
//     B = the wave's h rows, resident in registers for all of K = D), so a row sits on a lane and the 32 hidden values
//     of the chunk in the accumulator's registers;
//   * bias + GELU (and, in training, gelu') run on the accumulator registers; the bf16 result leaves the accumulator as the
//     B operand of GEMM 2, out^T[d][row] += W2[d][hid] . V^T[hid][row], with no lane movement.  The k order in which an
//     accumulator hands over its registers is permuted (element j of lane half h = hidden 16 s + 8 (j >> 2) + 4 h + (j & 3)),
//     so the W2 chunk is laid into LDS with its columns permuted the same way (8-byte quads 0,2 | 1,3 of every 16);
//   * the output accumulators (D x 32 rows per wave: 192 registers at D = 384) stay in registers until the end, then take
//     bias 2 and the fp32 residual and leave through an LDS transpose as 128-byte row segments.
// The four waves of a workgroup share the weight chunks: [32 hid][D] of W1 and [D][32 hid] of W2 per step (48 KB at
// D = 384, 131 MFMA flop per byte of LDS fill, like a 256 x 256 GEMM tile), double-buffered in LDS, fetched a step ahead
// through registers (one wave per SIMD: the registers are there, and a compiler-visible pipeline schedules better than
// an LDS-DMA ring of asm reads).  Training also stores gelu(u) and gelu'(u) ([M, HID] bf16 each: the backward's operands,
// exactly what the two-GEMM path stores) through a wave-private LDS transpose as 64-byte row segments.
//
// STATUS: correct (tests/test_ops_gpu.py) but NOT faster than the two GEMM launches it replaces, so the engine does not use it by
// default (STAR_AMD_MLP_FUSED=1 switches): D = 384, M = 40 960: 249 us (inference form) against 178 us, 355 against 223 with the
// training stores.  One wave per SIMD (about 500 registers of state) has nothing to hide its own latencies behind: of 6200 cycles per
// hidden chunk 1536 are MFMA, the register-staged weight refresh costs 1450, the GELU arithmetic 750 despite the hand-pinned
// interleave, the chunk barrier 450 (builds with each part compiled out), and the bare MFMA + fragment path still runs at 2.3x its
// MFMA time.  See DESIGN.md section 6.
#include "common.h"
#include "gelu.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mlp_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short mlp_short8_t;
typedef __attribute__((ext_vector_type(16))) float mlp_f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned mlp_u32x4_t;   // plain vector registers (HIP's uint4 struct keeps arrays of it on the stack)

#define MLP_T 256       // 4 waves x 32 rows
#define MLP_ROWS 128
#define MLP_SROW 80     // staging row bytes of a 32 x 32 bf16 block (64 + 16: 16-byte aligned rows)
#define MLP_OROW 144    // staging row bytes of a 32 x 32 f32 block (128 + 16)

__device__ __forceinline__ mlp_f32x16_t mlp_mfma(mlp_short8_t a, mlp_short8_t b, mlp_f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mlp_bf16x8_t, a), __builtin_bit_cast(mlp_bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ mlp_f32x16_t mlp_mfma(mlp_short8_t a, mlp_bf16x8_t b, mlp_f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mlp_bf16x8_t, a), b, c, 0, 0, 0);
}

// Operand fragments are read by asm in program order, one group of MFMA steps ahead of their use: with one wave per SIMD nothing
// else hides an LDS latency, and left to itself hipcc reads each fragment right in front of its MFMA and waits for it (measured:
// 9700 cycles per hidden chunk against 3100 of work).  lgkmcnt counts LDS operations in order, so "all but the youngest G" retires
// the older group whatever the compiler issues in between.
template <int OFF> __device__ __forceinline__ mlp_short8_t mlp_lds_r8(unsigned addr) {
    mlp_short8_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int G, int KEEP> __device__ __forceinline__ void mlp_lgkm_wait(mlp_short8_t (&f)[G]) {
    static_assert(G == 4 || G == 6, "group size");
    if constexpr (G == 6) {
        if constexpr (KEEP == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    } else {
        if constexpr (KEEP == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    }
}
template <int I> struct mlp_ic { static constexpr int value = I; };
template <int N, int I = 0, typename F> __device__ __forceinline__ void mlp_for(F&& f) {
    if constexpr (I < N) {
        f(mlp_ic<I>{});
        mlp_for<N, I + 1>(f);
    }
}

template <int D, bool TRAIN>
__global__ __launch_bounds__(MLP_T) void mlp_fwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ w1, const float* __restrict__ b1,
                                                        const bf16_t* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* xres, float* xout, bf16_t* __restrict__ v_out,
                                                        bf16_t* __restrict__ g_out, int M, int HID) {
    constexpr int KS1 = D / 16, DB = D / 32;
    constexpr int CPR = D / 8;                      // 16-byte chunks per W1 row
    constexpr int RB1 = 2 * D, W1B = 32 * RB1, W2B = D * 64, WBUF = 2 * W1B + 3 * W2B;
    constexpr int NP = 4 * D / MLP_T;               // 16-byte pieces per thread and operand chunk (W1: 32 x D/8, W2: D x 4)
    static_assert(D % 64 == 0 && (4 * D) % (2 * MLP_T) == 0, "mlp_fwd_kernel: D");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* b1s = reinterpret_cast<float*>(smem + WBUF);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h2 = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stg = smem + WBUF + HID * 4 + wave * (2 * 32 * MLP_SROW);
    const int row0 = blockIdx.x * MLP_ROWS + wave * 32;
    const int row = row0 + r;
    const int nchunks = HID / 32;

    // image offsets: W1 [32 hid][D]: chunk ^ (row & 15); W2 [D][32 hid]: chunk ^ ((row >> 2) & 3) -- conflict-free ds_read_b128
    auto off1 = [](int rw, int c) { return rw * RB1 + ((c ^ (rw & 15)) << 4); };
    auto off2 = [](int rw, int c) { return rw * 64 + ((c ^ ((rw >> 2) & 3)) << 4); };

    // W2 comes in its plain [D][HID] copy: a thread takes whole 16-hidden groups (two 16-byte loads) and writes them as the two
    // 16-byte pieces (quads 0,2 | quads 1,3) the permuted k order asks for -- no permuted weight copy on the host
    mlp_u32x4_t pre[2 * NP];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + j * MLP_T;
            pre[j] = *reinterpret_cast<const mlp_u32x4_t*>(w1 + ((size_t)chunk * 32 + i / CPR) * D + (i % CPR) * 8);
        }
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) {
            const int gi = tid + j * MLP_T;                       // group: row gi >> 1, 16-hidden half gi & 1
            const bf16_t* src = w2 + (size_t)(gi >> 1) * HID + chunk * 32 + (gi & 1) * 16;
            pre[NP + 2 * j] = *reinterpret_cast<const mlp_u32x4_t*>(src);
            pre[NP + 2 * j + 1] = *reinterpret_cast<const mlp_u32x4_t*>(src + 8);
        }
    };
    auto stash = [&](int chunk) {          // W1 -> buffer chunk & 1, W2 -> buffer chunk % 3
        unsigned char* sb1 = smem + (chunk & 1) * W1B;
        unsigned char* sb2 = smem + 2 * W1B + (chunk % 3) * W2B;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + j * MLP_T;
            *reinterpret_cast<mlp_u32x4_t*>(sb1 + off1(i / CPR, i % CPR)) = pre[j];
        }
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) {
            const int gi = tid + j * MLP_T;
            const mlp_u32x4_t a = pre[NP + 2 * j], b = pre[NP + 2 * j + 1];
            *reinterpret_cast<mlp_u32x4_t*>(sb2 + off2(gi >> 1, (gi & 1) * 2)) = (mlp_u32x4_t){a[0], a[1], b[0], b[1]};
            *reinterpret_cast<mlp_u32x4_t*>(sb2 + off2(gi >> 1, (gi & 1) * 2 + 1)) = (mlp_u32x4_t){a[2], a[3], b[2], b[3]};
        }
    };

    fetch(0);
    for (int i = tid; i < HID; i += MLP_T) b1s[i] = b1[i];
    mlp_short8_t xf[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        xf[ks] = (mlp_short8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (row < M) xf[ks] = *reinterpret_cast<const mlp_short8_t*>(h + (size_t)row * D + ks * 16 + 8 * h2);
    }
    mlp_f32x16_t acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[db][e] = 0.f;
    stash(0);

    constexpr int G1 = KS1 / 4, G2 = DB / 2;     // 4 groups of fragment reads per GEMM and chunk
    const unsigned lds_base = lds_u32(smem);
    unsigned a1[8], a2[2];                        // fragment addresses inside a buffer; the rest is an immediate offset
#pragma unroll
    for (int m = 0; m < 8; ++m) a1[m] = r * RB1 + (((2 * m + h2) ^ (r & 15)) << 4);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) a2[s2] = r * 64 + (((s2 * 2 + h2) ^ ((r >> 2) & 3)) << 4);
    mlp_short8_t f2a[G2], f2b[G2];                // GEMM 2 fragment groups (step i: k half i / DB, d block i % DB)
    mlp_f32x16_t s0, s1;
    mlp_bf16x8_t vb[2];                           // gelu(u) of the chunk whose GEMM 2 is pending

    // GEMM 1 of chunk c: two accumulator chains (even / odd k-steps), the first one starts from the bias; fragments one group
    // ahead.  `then2` (a W2 buffer address or 0): the first fragment group of the pending GEMM 2 is requested behind the last
    // GEMM 1 group.
    auto gemm1 = [&](int c, unsigned then2, auto with2) {
        const unsigned sbu = lds_base + (c & 1) * W1B;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 bv = *reinterpret_cast<const float4*>(b1s + c * 32 + 8 * t + 4 * h2);
            s0[4 * t] = bv.x; s0[4 * t + 1] = bv.y; s0[4 * t + 2] = bv.z; s0[4 * t + 3] = bv.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s1[e] = 0.f;
        mlp_short8_t fa[G1], fb[G1];
        mlp_for<G1>([&](auto i) { fa[i.value] = mlp_lds_r8<(i.value >> 3) * 256>(sbu + a1[i.value & 7]); });
        mlp_for<4>([&](auto gq) {
            constexpr int g0 = gq.value * G1;
            mlp_short8_t (&cur)[G1] = (gq.value & 1) ? fb : fa;
            mlp_short8_t (&nxt)[G1] = (gq.value & 1) ? fa : fb;
            if constexpr (gq.value < 3) {
                mlp_for<G1>([&](auto i) { constexpr int ks = g0 + G1 + i.value; nxt[i.value] = mlp_lds_r8<(ks >> 3) * 256>(sbu + a1[ks & 7]); });
                mlp_lgkm_wait<G1, G1>(cur);
            } else if constexpr (decltype(with2)::value) {
                mlp_for<G2>([&](auto i) { f2a[i.value] = mlp_lds_r8<(i.value % DB) * 2048>(then2 + a2[i.value / DB]); });
                mlp_lgkm_wait<G1, G2>(cur);
            } else {
                mlp_lgkm_wait<G1, 0>(cur);
            }
            mlp_for<G1>([&](auto i) {
                constexpr int ks = g0 + i.value;
                if constexpr ((ks & 1) == 0) s0 = mlp_mfma(cur[i.value], xf[ks], s0);
                else s1 = mlp_mfma(cur[i.value], xf[ks], s1);
            });
        });
    };
    // GEMM 2 of the pending chunk (its W2 buffer at `sb2`, gelu(u) in `vbp`); `primed`: the first fragment group is already requested
    auto gemm2 = [&](unsigned sb2, const mlp_bf16x8_t (&vbp)[2], auto primed) {
        if constexpr (!decltype(primed)::value)
            mlp_for<G2>([&](auto i) { f2a[i.value] = mlp_lds_r8<(i.value % DB) * 2048>(sb2 + a2[i.value / DB]); });
        mlp_for<4>([&](auto gq) {
            constexpr int g0 = gq.value * G2;
            mlp_short8_t (&cur)[G2] = (gq.value & 1) ? f2b : f2a;
            mlp_short8_t (&nxt)[G2] = (gq.value & 1) ? f2a : f2b;
            if constexpr (gq.value < 3) {
                mlp_for<G2>([&](auto i) { constexpr int st = g0 + G2 + i.value; nxt[i.value] = mlp_lds_r8<(st % DB) * 2048>(sb2 + a2[st / DB]); });
                mlp_lgkm_wait<G2, G2>(cur);
            } else {
                mlp_lgkm_wait<G2, 0>(cur);
            }
            mlp_for<G2>([&](auto i) {
                constexpr int st = g0 + i.value;
                acc[st % DB] = mlp_mfma(cur[i.value], vbp[st / DB], acc[st % DB]);
            });
        });
    };
    // bias + GELU on the accumulators of chunk c -> vbn (bf16, GEMM 2's B operand); training: gelu and gelu' to memory
    auto gelu_finish = [&](int c, const unsigned (&vp)[8], const unsigned (&gp)[8], mlp_bf16x8_t (&vbn)[2]) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) vbn[s2] = __builtin_bit_cast(mlp_bf16x8_t, (u4_t){vp[4 * s2], vp[4 * s2 + 1], vp[4 * s2 + 2], vp[4 * s2 + 3]});
        if constexpr (TRAIN) {
            // registers 4t..4t+3 = hidden 8t + 4 h2 ..+3 of this lane's row -> [row][hid] staging -> 64-byte row segments
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                *reinterpret_cast<uint2*>(stg + r * MLP_SROW + (8 * t + 4 * h2) * 2) = make_uint2(vp[2 * t], vp[2 * t + 1]);
                *reinterpret_cast<uint2*>(stg + 32 * MLP_SROW + r * MLP_SROW + (8 * t + 4 * h2) * 2) = make_uint2(gp[2 * t], gp[2 * t + 1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int id = lane + 64 * j, rr = id >> 2, cc = id & 3;
                const uint4 a = *reinterpret_cast<const uint4*>(stg + rr * MLP_SROW + cc * 16);
                const uint4 b = *reinterpret_cast<const uint4*>(stg + 32 * MLP_SROW + rr * MLP_SROW + cc * 16);
                if (row0 + rr < M) {
                    *reinterpret_cast<uint4*>(v_out + (size_t)(row0 + rr) * HID + c * 32 + cc * 8) = a;
                    *reinterpret_cast<uint4*>(g_out + (size_t)(row0 + rr) * HID + c * 32 + cc * 8) = b;
                }
            }
        }
    };
    auto gelu = [&](int c, mlp_bf16x8_t (&vbn)[2]) {
        unsigned vp[8], gp[8];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            float va, vc, ga = 0.f, gc = 0.f;
            if constexpr (TRAIN) {
                gelu_both(s0[e] + s1[e], va, ga);
                gelu_both(s0[e + 1] + s1[e + 1], vc, gc);
            } else {
                va = gelu_f(s0[e] + s1[e]);
                vc = gelu_f(s0[e + 1] + s1[e + 1]);
            }
            vp[e >> 1] = pack_bf16(va, vc);
            gp[e >> 1] = pack_bf16(ga, gc);
        }
        gelu_finish(c, vp, gp, vbn);
    };
    // GEMM 2 of the pending chunk and the GELU of chunk c, interleaved BY HAND: one or two MFMAs, then one element's GELU
    // arithmetic (~19 vector instructions, 76+ issue cycles: enough to cover them).  An in-order wave overlaps the two pipes only
    // if its instruction stream alternates, and hipcc clusters the MFMAs if left alone: instruction selection orders the block
    // before sched_barrier / sched_group_barrier are looked at, so the order is pinned by data dependences on empty volatile asm
    // statements instead (an MFMA's fragment and an element's input / results pass through them in program order).
    auto gemm2_gelu = [&](unsigned sb2, const mlp_bf16x8_t (&vbp)[2], int c, mlp_bf16x8_t (&vbn)[2]) {
        unsigned vp[8], gp[8];
        float vl = 0.f, gl = 0.f;                  // the even element of the pair in flight
        mlp_for<16>([&](auto ei) {
            constexpr int e = ei.value;
            constexpr int st0 = e * (2 * DB) / 16, st1 = (e + 1) * (2 * DB) / 16;      // this element's share of the 2 DB steps
            mlp_for<st1 - st0>([&](auto k) {
                constexpr int st = st0 + k.value;
                if constexpr (st % G2 == 0) {
                    constexpr int gq = st / G2;
                    mlp_short8_t (&cur)[G2] = (gq & 1) ? f2b : f2a;
                    mlp_short8_t (&nxt)[G2] = (gq & 1) ? f2a : f2b;
                    if constexpr (gq < 3) {
                        mlp_for<G2>([&](auto i) { constexpr int sn = st + G2 + i.value; nxt[i.value] = mlp_lds_r8<(sn % DB) * 2048>(sb2 + a2[sn / DB]); });
                        mlp_lgkm_wait<G2, G2>(cur);
                    } else {
                        mlp_lgkm_wait<G2, 0>(cur);
                    }
                }
                mlp_short8_t (&cur)[G2] = ((st / G2) & 1) ? f2b : f2a;
                asm volatile("" : "+v"(cur[st % G2]));          // not before this point (volatile asm statements keep their order) ...
                acc[st % DB] = mlp_mfma(cur[st % G2], vbp[st / DB], acc[st % DB]);
            });
            float u = s0[e] + s1[e], ve, ge = 0.f;
            asm volatile("" : "+v"(u));                          // ... and this element's arithmetic between these two
            if constexpr (TRAIN) {
                gelu_both(u, ve, ge);
                asm volatile("" : "+v"(ve), "+v"(ge));
            } else {
                ve = gelu_f(u);
                asm volatile("" : "+v"(ve));
            }
            if constexpr (e & 1) {
                vp[e >> 1] = pack_bf16(vl, ve);
                gp[e >> 1] = pack_bf16(gl, ge);
            } else {
                vl = ve;
                gl = ge;
            }
        });
        gelu_finish(c, vp, gp, vbn);
    };

    // Software pipeline over the hidden chunks: GEMM 1 of chunk c, then GEMM 2 of chunk c - 1 TOGETHER with the GELU arithmetic of
    // chunk c (independent work for the matrix pipe and the vector ALU of the one wave on this SIMD).  W1 chunks rotate through two
    // LDS buffers, W2 chunks through three (chunk c - 1's W2 is still being read while chunk c + 1's is written).
    __syncthreads();
    if (nchunks > 1) fetch(1);
    gemm1(0, 0u, mlp_ic<0>{});
    gelu(0, vb);
    if (nchunks > 1) stash(1);
    int w2prev = 0;                                // W2 buffer of chunk c - 1
    for (int c = 1; c < nchunks; ++c) {
        __syncthreads();                           // chunk c is in LDS; everyone is done with the buffers chunk c + 1 goes to
        if (c + 1 < nchunks) fetch(c + 1);
        const unsigned sb2 = lds_base + 2 * W1B + w2prev * W2B;
        gemm1(c, sb2, mlp_ic<1>{});
        mlp_bf16x8_t vbn[2];
        gemm2_gelu(sb2, vb, c, vbn);
        vb[0] = vbn[0];
        vb[1] = vbn[1];
        w2prev = w2prev == 2 ? 0 : w2prev + 1;
        if (c + 1 < nchunks) stash(c + 1);
    }
    gemm2(lds_base + 2 * W1B + w2prev * W2B, vb, mlp_ic<0>{});
    __syncthreads();                           // the weight buffers become the output staging area
    unsigned char* og = smem + wave * (32 * MLP_OROW);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<float4*>(og + r * MLP_OROW + (8 * t + 4 * h2) * 4) =
                make_float4(acc[db][4 * t], acc[db][4 * t + 1], acc[db][4 * t + 2], acc[db][4 * t + 3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int id = lane + 64 * j, rr = id >> 3, cc = id & 7;
            float4 o = *reinterpret_cast<const float4*>(og + rr * MLP_OROW + cc * 16);
            if (row0 + rr < M) {
                const int d0 = db * 32 + cc * 4;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + d0);
                const float4 xr = *reinterpret_cast<const float4*>(xres + (size_t)(row0 + rr) * D + d0);
                o.x += bb.x + xr.x; o.y += bb.y + xr.y; o.z += bb.z + xr.z; o.w += bb.w + xr.w;
                *reinterpret_cast<float4*>(xout + (size_t)(row0 + rr) * D + d0) = o;
            }
        }
    }
}

template <int D, bool TRAIN>
static int mlp_launch(const void* h, const void* w1, const float* b1, const void* w2, const float* b2, const float* xres, float* xout,
                      void* v_out, void* g_out, int M, int HID, hipStream_t s) {
    const size_t lds = 2 * (size_t)(32 * 2 * D) + 3 * (size_t)(D * 64) + (size_t)HID * 4 + 4 * 2 * 32 * MLP_SROW;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel<D, TRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return star_set_error(-2, "star_mlp_fwd: hipFuncSetAttribute(%zu bytes of LDS): %s", lds, hipGetErrorString(e));
        done = true;
    }
    hipLaunchKernelGGL((mlp_fwd_kernel<D, TRAIN>), dim3((M + MLP_ROWS - 1) / MLP_ROWS), dim3(MLP_T), lds, s, (const bf16_t*)h, (const bf16_t*)w1, b1,
                       (const bf16_t*)w2, b2, xres, xout, (bf16_t*)v_out, (bf16_t*)g_out, M, HID);
    return 0;
}

// 1 if star_mlp_fwd has a kernel for this shape (bf16; D = 256 or 384; HID a multiple of 32 whose bias fits the LDS budget)
extern "C" int star_mlp_fwd_supported(int D, int HID, int dt) {
    return dt == STAR_BF16 && (D == 256 || D == 384) && HID > 0 && HID % 32 == 0 && HID <= 2048;
}

extern "C" int star_mlp_fwd(const void* h, const void* w1, const float* b1, const void* w2, const float* b2, const float* xres,
                            float* xout, void* v_out, void* g_out, int M, int D, int HID, int dt, void* stream) {
    STAR_CHECK(h && w1 && b1 && w2 && b2 && xres && xout, "star_mlp_fwd: null pointer");
    STAR_CHECK((v_out == nullptr) == (g_out == nullptr), "star_mlp_fwd: v_out and g_out go together (training) or are both null (inference)");
    STAR_CHECK(M > 0 && star_mlp_fwd_supported(D, HID, dt), "star_mlp_fwd: unsupported shape M=%d D=%d HID=%d dt=%d (star_mlp_fwd_supported)", M, D,
               HID, dt);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (D == 384) rc = v_out ? mlp_launch<384, true>(h, w1, b1, w2, b2, xres, xout, v_out, g_out, M, HID, s)
                             : mlp_launch<384, false>(h, w1, b1, w2, b2, xres, xout, v_out, g_out, M, HID, s);
    else rc = v_out ? mlp_launch<256, true>(h, w1, b1, w2, b2, xres, xout, v_out, g_out, M, HID, s)
                    : mlp_launch<256, false>(h, w1, b1, w2, b2, xres, xout, v_out, g_out, M, HID, s);
    if (rc) return rc;
    STAR_LAUNCH_CHECK("star_mlp_fwd");
    return 0;
}
