// K1: voxel -> patch gather (the im2col half of the Conv2d patch embedding) and
// K9: 2-class occupancy cross-entropy fused with the arg-max decision, the un-patchify and the loss
//     gradient.  Reference: timm PatchEmbed via star/models/mae_base.py:269;
//     star/models/multiagent_mae.py:1049-1057,1076-1079,1099; star/models/mae_base.py:392-406.
// Both are HBM-bound streaming kernels: 4 elements per lane, coalesced on the patch-matrix side.
#include "common.h"

#define PC_THREADS 256

// out[(f*L + token), k] with k ordered (c, py, px) [korder 0] or (py, px, c) [korder 1]
template <typename T>
__global__ __launch_bounds__(PC_THREADS) void patchify_kernel(const float* __restrict__ img1, int64_t s1n, int64_t s1c,
                                                              int64_t s1h, int64_t s1w, const float* __restrict__ imgn,
                                                              int64_t snn, int64_t snt, int64_t snc, int64_t snh,
                                                              int64_t snw, T* __restrict__ out, int BA, int Tn, int C,
                                                              int H, int W, int p, int korder, int vec) {
    const int gw = W / p, gh = H / p, L = gw * gh, K = C * p * p, kq = K >> 2;
    const int64_t item = (int64_t)blockIdx.x * PC_THREADS + threadIdx.x;
    if (item >= (int64_t)BA * Tn * L * kq) return;
    const int k0 = (int)(item % kq) << 2;
    const int64_t row = item / kq;
    const int tok = (int)(row % L);
    const int f = (int)(row / L);
    const int ba = f / Tn, t = f - ba * Tn;
    const int ty = tok / gw, tx = tok - ty * gw;
    const float* base;
    int64_t sc, sh, sw;
    if (t == 0) {
        base = img1 + (int64_t)ba * s1n; sc = s1c; sh = s1h; sw = s1w;
    } else {
        base = imgn + (int64_t)ba * snn + (int64_t)(t - 1) * snt; sc = snc; sh = snh; sw = snw;
    }
    float v[4];
    if (vec == 1) {   // channel-first image with unit pixel stride: the four values are one aligned 16-byte run of an image row
        const int c = k0 / (p * p), r = k0 - c * p * p, py = r / p, px = r - py * p;
        ld4(base + (int64_t)c * sc + (int64_t)(ty * p + py) * sh + (tx * p + px), v);
        st4(out + row * K + k0, v);
        return;
    }
    if (vec == 2) {   // channels-last image (unit channel stride, pixel stride C) in the (py,px,c) order: a patch row is p*C consecutive floats
        const int py = k0 / (p * C), r = k0 - py * p * C;
        ld4(base + (int64_t)(ty * p + py) * sh + (int64_t)tx * p * C + r, v);
        st4(out + row * K + k0, v);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        int c, py, px;
        if (korder == 0) {
            c = k / (p * p);
            const int r = k - c * p * p;
            py = r / p; px = r - py * p;
        } else {
            py = k / (p * C);
            const int r = k - py * p * C;
            px = r / C; c = r - px * C;
        }
        v[e] = base[(int64_t)c * sc + (int64_t)(ty * p + py) * sh + (int64_t)(tx * p + px) * sw];
    }
    st4(out + row * K + k0, v);
}

extern "C" int star_patchify(const float* img1, const int64_t* s1, const float* imgn, const int64_t* sn, void* out,
                             int BA, int T, int C, int H, int W, int p, int dt, void* stream) {
    STAR_CHECK(img1 && s1 && out, "star_patchify: null pointer");
    STAR_CHECK(T == 1 || (imgn && sn), "star_patchify: T=%d needs the next-frame tensor", T);
    STAR_CHECK(p > 0 && H % p == 0 && W % p == 0 && (C * p * p) % 4 == 0, "star_patchify: bad H=%d W=%d p=%d C=%d", H, W, p, C);
    const int64_t items = (int64_t)BA * T * (H / p) * (W / p) * (C * p * p / 4);
    if (items == 0) return 0;
    const int korder = dt >> 8;  // bit 8 of dt selects the (py,px,c) ordering
    const int adt = dt & 0xff;
    int64_t z5[5] = {0, 0, 0, 0, 0};
    if (!sn) sn = z5;
    // 16-byte image reads need unit pixel stride, the (c,py,px) order, and every other stride / base a multiple of 4 floats
    auto al4 = [](int64_t v) { return (v & 3) == 0; };
    int vec = korder == 0 && (p & 3) == 0 && s1[3] == 1 && al4(s1[0]) && al4(s1[1]) && al4(s1[2]) && (reinterpret_cast<uintptr_t>(img1) & 15) == 0;
    if (vec && T > 1)
        vec = sn[4] == 1 && al4(sn[0]) && al4(sn[1]) && al4(sn[2]) && al4(sn[3]) && (reinterpret_cast<uintptr_t>(imgn) & 15) == 0;
    if (!vec && korder == 1 && (p * C) % 4 == 0 && s1[1] == 1 && s1[3] == C && al4(s1[0]) && al4(s1[2]) && (reinterpret_cast<uintptr_t>(img1) & 15) == 0) {
        vec = 2;
        if (T > 1 && !(sn[2] == 1 && sn[4] == C && al4(sn[0]) && al4(sn[1]) && al4(sn[3]) && (reinterpret_cast<uintptr_t>(imgn) & 15) == 0)) vec = 0;
    }
    if (adt == STAR_BF16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(cdiv(items, PC_THREADS)), dim3(PC_THREADS), 0, (hipStream_t)stream,
                           img1, s1[0], s1[1], s1[2], s1[3], imgn, sn[0], sn[1], sn[2], sn[3], sn[4], (bf16_t*)out, BA, T, C, H, W, p, korder, vec);
    else
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(cdiv(items, PC_THREADS)), dim3(PC_THREADS), 0, (hipStream_t)stream,
                           img1, s1[0], s1[1], s1[2], s1[3], imgn, sn[0], sn[1], sn[2], sn[3], sn[4], (float*)out, BA, T, C, H, W, p, korder, vec);
    STAR_LAUNCH_CHECK("star_patchify");
    return 0;
}

// -----------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PC_THREADS) void ce_argmax_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                               int64_t tsn, int64_t tsc, int64_t tsh, int64_t tsw,
                                                               int64_t* __restrict__ ind64, uint8_t* __restrict__ ind8,
                                                               float* __restrict__ pred, T* __restrict__ dlogits,
                                                               float* __restrict__ loss_partial, int BA, int C, int H, int W,
                                                               int p, float inv_count) {
    __shared__ float s_red[PC_THREADS / 64];
    const int gw = W / p, gh = H / p, L = gw * gh, V = C * p * p, vq = V >> 2;
    const int64_t item = (int64_t)blockIdx.x * PC_THREADS + threadIdx.x;
    float lsum = 0.f;
    if (item < (int64_t)BA * L * vq) {
        const int v0 = (int)(item % vq) << 2;
        const int64_t row = item / vq;
        const int tok = (int)(row % L), n = (int)(row / L);
        const int c = v0 / (p * p), r = v0 - c * p * p, py = r / p, px = r - py * p;
        const int y = (tok / gw) * p + py, x = (tok % gw) * p + px;
        float f[4], o[4];
        ld4(logits + row * 2 * V + v0, f);
        ld4(logits + row * 2 * V + V + v0, o);
        const float* tp = target + (int64_t)n * tsn + (int64_t)c * tsc + (int64_t)y * tsh + (int64_t)x * tsw;
        float df[4], dO[4];
        unsigned char dec[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tc = ((long long)tp[(int64_t)e * tsw]) != 0;  // target.long()
            const float d = o[e] - f[e];
            // log(exp(f) + exp(o)) - chosen  ==  softplus(+-d)
            const float z = tc ? -d : d;
            const float en = expf(-fabsf(d));            // one exponential serves the loss and the probability (|z| == |d|)
            lsum += fmaxf(z, 0.f) + log1pf(en);
            const float p1 = (d >= 0.f ? 1.0f : en) / (1.0f + en);  // softmax prob of "occupied" = sigmoid(d)
            dO[e] = (p1 - (float)tc) * inv_count;
            df[e] = -dO[e];
            dec[e] = o[e] > f[e] ? 1 : 0;  // argmax(softmax): ties -> class 0
        }
        if (dlogits) {
            st4(dlogits + row * 2 * V + v0, df);
            st4(dlogits + row * 2 * V + V + v0, dO);
        }
        const size_t HW = (size_t)H * W;
        const size_t pix = ((size_t)n * C + c) * HW + (size_t)y * W + x;
        if (ind64) {
            longlong4 w;
            w.x = dec[0]; w.y = dec[1]; w.z = dec[2]; w.w = dec[3];
            *reinterpret_cast<longlong4*>(ind64 + pix) = w;
        }
        if (ind8) *reinterpret_cast<uchar4*>(ind8 + pix) = make_uchar4(dec[0], dec[1], dec[2], dec[3]);
        if (pred) {
            const size_t pf = ((size_t)n * 2 * C + c) * HW + (size_t)y * W + x;
            *reinterpret_cast<float4*>(pred + pf) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(pred + pf + (size_t)C * HW) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (loss_partial) {
        lsum = wave_sum(lsum);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) loss_partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// Tiled variant: a block owns G consecutive tokens of one token row, i.e. for every (channel, patch row) a run of G*p
// contiguous pixels.  The per-thread kernel above reads the target and writes the decisions in p-pixel pieces (8 pixels:
// 8-byte uint8 stores, 32-byte target reads); here the target run is read as it lies in memory (either layout), turned into
// byte flags in LDS [c][py][G*p], the logits are walked token-major (coalesced, as before) with the flags read from / the
// decisions written to LDS, and ind8 / ind64 leave in G*p-pixel runs.
template <typename T>
__global__ __launch_bounds__(PC_THREADS) void ce_argmax_tiled_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                                     int64_t tsn, int64_t tsc, int64_t tsh, int64_t tsw,
                                                                     int64_t* __restrict__ ind64, uint8_t* __restrict__ ind8,
                                                                     float* __restrict__ pred, T* __restrict__ dlogits,
                                                                     float* __restrict__ loss_partial, int BA, int C, int H, int W,
                                                                     int p, int G, float inv_count) {
    extern __shared__ unsigned char ce_flags[];      // [C][p][G*p + 4]: bit 0 = target class, bit 1 = decision
    __shared__ float s_red[PC_THREADS / 64];
    const int gw = W / p, gh = H / p, L = gw * gh, V = C * p * p, vq = V >> 2;
    const int groups_per_row = gw / G;
    const int blk = blockIdx.x;
    const int n = blk / (gh * groups_per_row);
    const int rem = blk - n * (gh * groups_per_row);
    const int ty = rem / groups_per_row, tx0 = (rem - ty * groups_per_row) * G;
    const int run = G * p;                           // contiguous pixels per (c, py)
    const int runp = run + 4;                        // LDS row pitch: consecutive (c, py) rows start one bank apart
    const int nvox = C * p * run;
    const float* tbase = target + (int64_t)n * tsn + (int64_t)(ty * p) * tsh + (int64_t)(tx0 * p) * tsw;
    // phase 1: target -> flags, in the order the target lies in memory
    if (tsc == 1) {          // channels-last: (py, x, c) with c fastest
        for (int i = threadIdx.x; i < nvox; i += PC_THREADS) {
            const int py = i / (run * C), r = i - py * run * C, x = r / C, c = r - x * C;
            const float v = tbase[(int64_t)py * tsh + (int64_t)x * tsw + c];
            ce_flags[(c * p + py) * runp + x] = ((long long)v) != 0 ? 1 : 0;
        }
    } else {                 // channel-first (or anything else): (c, py, x) with x fastest
        for (int i = threadIdx.x; i < nvox; i += PC_THREADS) {
            const int c = i / (p * run), r = i - c * p * run, py = r / run, x = r - py * run;
            const float v = tbase[(int64_t)c * tsc + (int64_t)py * tsh + (int64_t)x * tsw];
            ce_flags[(c * p + py) * runp + x] = ((long long)v) != 0 ? 1 : 0;
        }
    }
    __syncthreads();
    // phase 2: token-major walk over the logits
    float lsum = 0.f;
    const size_t HW = (size_t)H * W;
    for (int item = threadIdx.x; item < G * vq; item += PC_THREADS) {
        const int g = item / vq, v0 = (item - g * vq) << 2;
        const int64_t row = (int64_t)n * L + (int64_t)ty * gw + tx0 + g;
        const int c = v0 / (p * p), r = v0 - c * p * p, py = r / p, px = r - py * p;
        float f[4], o[4];
        ld4(logits + row * 2 * V + v0, f);
        ld4(logits + row * 2 * V + V + v0, o);
        unsigned char* fl = ce_flags + (c * p + py) * runp + g * p + px;      // 4-byte aligned: p % 4 == 0
        const unsigned tcs = *reinterpret_cast<const unsigned*>(fl);
        float df[4], dO[4];
        unsigned out4 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tc = (tcs >> (8 * e)) & 1;
            const float d = o[e] - f[e];
            const float z = tc ? -d : d;
            const float en = expf(-fabsf(d));
            lsum += fmaxf(z, 0.f) + log1pf(en);
            const float p1 = (d >= 0.f ? 1.0f : en) / (1.0f + en);
            dO[e] = (p1 - (float)tc) * inv_count;
            df[e] = -dO[e];
            out4 |= (unsigned)(o[e] > f[e] ? 1 : 0) << (8 * e);
        }
        *reinterpret_cast<unsigned*>(fl) = out4;                                 // decisions replace the flags
        if (dlogits) {
            st4(dlogits + row * 2 * V + v0, df);
            st4(dlogits + row * 2 * V + V + v0, dO);
        }
        if (pred) {
            const int y = ty * p + py, x = (tx0 + g) * p + px;
            const size_t pf = ((size_t)n * 2 * C + c) * HW + (size_t)y * W + x;
            *reinterpret_cast<float4*>(pred + pf) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(pred + pf + (size_t)C * HW) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    // phase 3: decisions out in runs of G*p pixels
    const int runq = run >> 2;
    for (int i = threadIdx.x; i < C * p * runq; i += PC_THREADS) {
        const int cp = i / runq, xq = (i - cp * runq) << 2;
        const int c = cp / p, py = cp - c * p;
        const unsigned d4 = *reinterpret_cast<const unsigned*>(ce_flags + cp * runp + xq);
        const size_t pix = ((size_t)n * C + c) * HW + (size_t)(ty * p + py) * W + (size_t)tx0 * p + xq;
        if (ind8) *reinterpret_cast<unsigned*>(ind8 + pix) = d4;
        if (ind64) {
            longlong4 w;
            w.x = d4 & 1; w.y = (d4 >> 8) & 1; w.z = (d4 >> 16) & 1; w.w = (d4 >> 24) & 1;
            *reinterpret_cast<longlong4*>(ind64 + pix) = w;
        }
    }
    if (loss_partial) {
        lsum = wave_sum(lsum);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) loss_partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// tokens per block of the tiled kernel: the largest power of two <= 8 that divides the token row and keeps the flags in 32 KiB
static int ce_group(int C, int W, int p) {
    const int gw = W / p;
    int G = 8;
    while (G > 1 && (gw % G != 0 || (int64_t)G * C * p * p > 32 * 1024)) G >>= 1;
    return (gw % G == 0 && (int64_t)G * C * p * p <= 48 * 1024) ? G : 0;
}
static bool ce_tiled() {
    static const char* env = getenv("STAR_AMD_CE_TILED");     // "0": the per-thread kernel (A/B)
    return !(env && env[0] == '0');
}

__global__ __launch_bounds__(1024) void ce_finish_kernel(const float* __restrict__ part, int n, double inv_count,
                                                         float* __restrict__ loss) {
    __shared__ double s[1024];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) a += (double)part[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(s[0] * inv_count);
}

extern "C" int star_ce_nblocks(int BA, int C, int H, int W) { return cdiv((int64_t)BA * C * H * W / 4, PC_THREADS); }   // upper bound over both kernels

extern "C" int star_ce_argmax(const float* logits, const float* target, const int64_t* st, int64_t* ind64, uint8_t* ind8,
                              float* pred, void* dlogits, float* loss_partial, int BA, int C, int H, int W, int p, int dt,
                              void* stream) {
    STAR_CHECK(logits && target && st, "star_ce_argmax: null pointer");
    STAR_CHECK(p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0, "star_ce_argmax: need p %% 4 == 0 and H,W multiples of p (p=%d H=%d W=%d)", p, H, W);
    const int64_t count = (int64_t)BA * C * H * W;
    if (count == 0) return 0;
    const float inv = (float)(1.0 / (double)count);
    const int G = ce_tiled() ? ce_group(C, W, p) : 0;
    const int nall = star_ce_nblocks(BA, C, H, W);
    const int nblk = G > 0 ? BA * (H / p) * ((W / p) / G) : 0;
    if (G > 0 && nblk <= nall) {
        // fewer blocks than star_ce_nblocks(): the unused partial sums must read as zero
        if (loss_partial && nall > nblk) {
            hipError_t e = hipMemsetAsync(loss_partial + nblk, 0, (size_t)(nall - nblk) * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return star_set_error(-2, "star_ce_argmax: memset: %s", hipGetErrorString(e));
        }
        const size_t lds = (size_t)C * p * (G * p + 4);
        if (dt == STAR_BF16)
            hipLaunchKernelGGL(ce_argmax_tiled_kernel<bf16_t>, dim3(nblk), dim3(PC_THREADS), lds, (hipStream_t)stream, logits, target, st[0], st[1], st[2],
                               st[3], ind64, ind8, pred, (bf16_t*)dlogits, loss_partial, BA, C, H, W, p, G, inv);
        else
            hipLaunchKernelGGL(ce_argmax_tiled_kernel<float>, dim3(nblk), dim3(PC_THREADS), lds, (hipStream_t)stream, logits, target, st[0], st[1], st[2],
                               st[3], ind64, ind8, pred, (float*)dlogits, loss_partial, BA, C, H, W, p, G, inv);
        STAR_LAUNCH_CHECK("star_ce_argmax(tiled)");
        return 0;
    }
    dim3 grid(star_ce_nblocks(BA, C, H, W));
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(ce_argmax_kernel<bf16_t>, grid, dim3(PC_THREADS), 0, (hipStream_t)stream, logits, target, st[0], st[1], st[2], st[3],
                           ind64, ind8, pred, (bf16_t*)dlogits, loss_partial, BA, C, H, W, p, inv);
    else
        hipLaunchKernelGGL(ce_argmax_kernel<float>, grid, dim3(PC_THREADS), 0, (hipStream_t)stream, logits, target, st[0], st[1], st[2], st[3],
                           ind64, ind8, pred, (float*)dlogits, loss_partial, BA, C, H, W, p, inv);
    STAR_LAUNCH_CHECK("star_ce_argmax");
    return 0;
}

extern "C" int star_ce_finish(const float* loss_partial, int nblocks, int64_t count, float* loss, void* stream) {
    STAR_CHECK(loss_partial && loss && nblocks > 0 && count > 0, "star_ce_finish: bad arguments");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_partial, nblocks, 1.0 / (double)count, loss);
    STAR_LAUNCH_CHECK("star_ce_finish");
    return 0;
}
