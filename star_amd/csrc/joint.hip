// Kernels of the joint (latent-level fusion) STAR variant -- SURVEY 8(f) rank 1.
// Reference: AmortizedFusionMMAEViT (star/models/multiagent_mae.py:347-681) and its conv prediction head ConvPred /
// ConvPred16 (star/models/mae_base.py:16-105): bilinear x2 -> 3x3 conv -> BatchNorm (-> ReLU), three or four times.
// Everything here is channels-LAST: the decoder's token rows [N*H*W, C] ARE the NHWC feature map, so the 3x3 convolutions
// run as nine row-gathered star_gemm taps and these kernels only move bytes (HBM-bound: 16-byte accesses, one pass each).
//   star_feat_warp_fuse   forward_fusion (multiagent_mae.py:497-551): pose warp of the neighbours' float feature maps into the
//                         ego frame (the index arithmetic of warp_fuse.hip at the token grid), sum, fusion("sum") threshold
//                         (mae_base.py:624-633), + decoder_pos_embed
//   star_upsample2x_*     F.interpolate(scale_factor=2, mode='bilinear'), align_corners=False, and its adjoint (gather form)
//   star_bn_*             nn.BatchNorm2d in training mode (batch statistics, two-pass variance) or eval mode, with the ReLU
//                         that follows it fused in both directions; deterministic two-stage column reductions
//   star_ce_nhwc          CrossEntropyLoss over the stacked (free, occ) logits + argmax (multiagent_mae.py:655-681)
#include "common.h"

#define JT 256

__device__ __forceinline__ float jw_base(int i, int n) {
    float step = __fdiv_rn(2.0f, (float)(n - 1));
    float lo = __fmaf_rn(step, (float)i, -1.0f);
    float hi = __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
    float ls = (i < n / 2) ? lo : hi;
    return __fdiv_rn(__fmul_rn(ls, (float)(n - 1)), (float)n);
}
__device__ __forceinline__ float jw_pix(float g, float size) {
    float t = __fadd_rn(g, 1.0f);
    t = __fmul_rn(t, size);
    t = __fsub_rn(t, 1.0f);
    t = __fmul_rn(t, 0.5f);
    return rintf(t);
}

// one block per output token (x, y) of ego slot i of scene b; threads over channels (4 each)
__global__ __launch_bounds__(128) void feat_warp_fuse_kernel(const float* __restrict__ feat, const float* __restrict__ theta,
                                                             const void* __restrict__ num_agent, int na64, const float* __restrict__ pos,
                                                             float* __restrict__ out, uint8_t* __restrict__ passthrough, int B, int A,
                                                             int G, int D) {
    const int tok = blockIdx.x, i = blockIdx.y, b = blockIdx.z;
    const int y = tok / G, x = tok - y * G;
    long long nraw = na64 ? ((const long long*)num_agent)[(size_t)b * A] : (long long)((const int*)num_agent)[(size_t)b * A];
    const int n = nraw < 0 ? 0 : (nraw > A ? A : (int)nraw);
    const size_t L = (size_t)G * G;
    const size_t own = ((size_t)(i * B + b) * L + tok) * D;
    const bool fused = i < n;
    if (tok == 0 && threadIdx.x == 0) passthrough[i * B + b] = fused ? 0 : 1;
    int src_tok[8];   // per neighbour: source token or -1 (A <= 8 checked on the host)
    if (fused) {
        const float fG = (float)G;
        const float bxx = jw_base(x, G), byy = jw_base(y, G);
        for (int j = 0; j < n; ++j) {
            src_tok[j] = -1;
            if (j == i) continue;
            const float* th = theta + ((size_t)(b * A + i) * A + j) * 8;
            const float px = jw_pix(__fadd_rn(bxx, th[4]), fG), py = jw_pix(__fadd_rn(byy, th[5]), fG);
            if (px >= 0.0f && px < fG && py >= 0.0f && py < fG) {
                const float gx0 = jw_base((int)px, G), gy0 = jw_base((int)py, G);
                const float gx = __fmaf_rn(gy0, th[1], __fmul_rn(gx0, th[0]));
                const float gy = __fmaf_rn(gy0, th[3], __fmul_rn(gx0, th[2]));
                const float sx = jw_pix(gx, fG), sy = jw_pix(gy, fG);
                if (sx >= 0.0f && sx < fG && sy >= 0.0f && sy < fG) src_tok[j] = (int)sy * G + (int)sx;
            }
        }
    }
    for (int c = threadIdx.x * 4; c < D; c += 128 * 4) {
        float v[4];
        ld4(feat + own + c, v);
        if (fused) {
            for (int j = 0; j < n; ++j) {        // torch.sum(torch.stack([ego, warped neighbours in agent order]))
                if (j == i) continue;
                if (src_tok[j] >= 0) {
                    float w[4];
                    ld4(feat + ((size_t)(j * B + b) * L + src_tok[j]) * D + c, w);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += w[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += 0.0f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] >= 0.5f ? 1.0f : (v[k] < 0.5f ? 0.0f : v[k]);   // NaN stays (neither assignment hits it)
        }
        if (pos != nullptr) {
            float p4[4];
            ld4(pos + (size_t)tok * D + c, p4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += p4[k];
        }
        st4(out + own + c, v);
    }
}

extern "C" int star_feat_warp_fuse(const float* feat, const float* theta, const void* num_agent, int num_agent_is64, const float* pos,
                                   float* out, uint8_t* passthrough, int B, int A, int G, int D, void* stream) {
    STAR_CHECK(feat && theta && num_agent && out && passthrough, "star_feat_warp_fuse: null pointer");
    STAR_CHECK(B > 0 && A > 0 && A <= 8 && G >= 2 && D >= 4 && D % 4 == 0, "star_feat_warp_fuse: bad shape B=%d A=%d G=%d D=%d (A <= 8, D %% 4 == 0)", B, A, G, D);
    hipLaunchKernelGGL(feat_warp_fuse_kernel, dim3(G * G, A, B), dim3(128), 0, (hipStream_t)stream, feat, theta, num_agent, num_agent_is64, pos, out,
                       passthrough, B, A, G, D);
    STAR_LAUNCH_CHECK("star_feat_warp_fuse");
    return 0;
}

// ---- bilinear x2 ------------------------------------------------------------------------------------------------------
// PyTorch (align_corners=False, scale 1/2): src = max(0.5*(dst + 0.5) - 0.5, 0); i0 = floor(src); i1 = i0 + (i0 < n-1); l1 = src - i0.
__device__ __forceinline__ void up_src(int d, int n, int& i0, int& i1, float& l0, float& l1) {
    float s = 0.5f * ((float)d + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

template <typename T>
__global__ __launch_bounds__(JT) void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)N * 2 * H * 2 * W * cv;
    for (int64_t id = (int64_t)blockIdx.x * JT + threadIdx.x; id < total; id += (int64_t)gridDim.x * JT) {
        const int c = (int)(id % cv) << 2;
        int64_t p = id / cv;
        const int X = (int)(p % (2 * W));
        p /= 2 * W;
        const int Y = (int)(p % (2 * H)), n = (int)(p / (2 * H));
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        up_src(Y, H, y0, y1, ly0, ly1);
        up_src(X, W, x0, x1, lx0, lx1);
        const T* b = x + (int64_t)n * H * W * C + c;
        float v00[4], v01[4], v10[4], v11[4], o[4];
        ld4(b + ((int64_t)y0 * W + x0) * C, v00);
        ld4(b + ((int64_t)y0 * W + x1) * C, v01);
        ld4(b + ((int64_t)y1 * W + x0) * C, v10);
        ld4(b + ((int64_t)y1 * W + x1) * C, v11);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ly0 * (lx0 * v00[k] + lx1 * v01[k]) + ly1 * (lx0 * v10[k] + lx1 * v11[k]);
        st4(y + (((int64_t)n * 2 * H + Y) * 2 * W + X) * C + c, o);
    }
}

// adjoint, gather form (deterministic): input pixel (yy, xx) collects from the output rows 2yy-1 .. 2yy+2 that reference it
__device__ __forceinline__ float up_weight(int d, int n, int i) {   // weight output index d gives input index i
    int i0, i1;
    float l0, l1;
    up_src(d, n, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}
__global__ __launch_bounds__(JT) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t id = (int64_t)blockIdx.x * JT + threadIdx.x; id < total; id += (int64_t)gridDim.x * JT) {
        const int c = (int)(id % cv) << 2;
        int64_t p = id / cv;
        const int xx = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H), n = (int)(p / H);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int Y = max(0, 2 * yy - 1); Y <= min(2 * H - 1, 2 * yy + 2); ++Y) {
            const float wy = up_weight(Y, H, yy);
            if (wy == 0.f) continue;
            for (int X = max(0, 2 * xx - 1); X <= min(2 * W - 1, 2 * xx + 2); ++X) {
                const float w = wy * up_weight(X, W, xx);
                if (w == 0.f) continue;
                float g[4];
                ld4(dy + (((int64_t)n * 2 * H + Y) * 2 * W + X) * C + c, g);
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += w * g[k];
            }
        }
        st4(dx + (((int64_t)n * H + yy) * W + xx) * C + c, a);
    }
}

static int stream_blocks(int64_t items) {
    int64_t b = (items + JT - 1) / JT;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int star_upsample2x_fwd(const void* x, void* y, int N, int H, int W, int C, int dt, void* stream) {
    STAR_CHECK(x && y && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "star_upsample2x_fwd: bad arguments (C %% 4 == 0)");
    const int nb = stream_blocks((int64_t)N * 4 * H * W * (C / 4));
    if (dt == STAR_BF16) hipLaunchKernelGGL(upsample2x_fwd_kernel<bf16_t>, dim3(nb), dim3(JT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C);
    else hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(nb), dim3(JT), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C);
    STAR_LAUNCH_CHECK("star_upsample2x_fwd");
    return 0;
}
extern "C" int star_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    STAR_CHECK(dy && dx && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "star_upsample2x_bwd: bad arguments (C %% 4 == 0)");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_blocks((int64_t)N * H * W * (C / 4))), dim3(JT), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
    STAR_LAUNCH_CHECK("star_upsample2x_bwd");
    return 0;
}

// ---- BatchNorm over rows [R, ld] (C <= ld valid channels, ld % 4 == 0) ---------------------------------------------
// Column reductions in two deterministic stages: block partials [nblocks][2][ld], then a fixed-order fold.
//   MODE 0: sum x                    MODE 1: sum (x - mean)^2
//   MODE 2: sum dy', sum dy' * xhat  with dy' = dy masked by the ReLU that followed the normalisation (when relu != 0)
#define BN_MAX_BLOCKS 512
template <int MODE>
__global__ __launch_bounds__(JT) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, int64_t R, int ld, float* __restrict__ part) {
    extern __shared__ float s_red[];   // [rl][2][ld]
    const int cv = ld >> 2, rls = JT / cv;               // column vectors, row lanes
    const int c4 = (threadIdx.x % cv) << 2, rl = threadIdx.x / cv;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
    float m[4] = {0.f, 0.f, 0.f, 0.f}, rs[4], g[4], bt[4];
    if (MODE >= 1) ld4(mean + c4, m);
    if (MODE == 2) {
        ld4(rstd + c4, rs);
        ld4(gamma + c4, g);
        ld4(beta + c4, bt);
    }
    if (rl < rls) {
        const int64_t rows_per = (R + gridDim.x - 1) / gridDim.x;
        const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(R, r0 + rows_per);
        for (int64_t r = r0 + rl; r < r1; r += rls) {
            float v[4];
            ld4(x + r * ld + c4, v);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += v[k];
            } else if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += (v[k] - m[k]) * (v[k] - m[k]);
            } else {
                float d[4];
                ld4(dy + r * ld + c4, d);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (v[k] - m[k]) * rs[k];
                    const float dd = (relu && !(xh * g[k] + bt[k] > 0.f)) ? 0.f : d[k];
                    a[k] += dd;
                    b2[k] += dd * xh;
                }
            }
        }
        st4(s_red + ((size_t)rl * 2 + 0) * ld + c4, a);
        st4(s_red + ((size_t)rl * 2 + 1) * ld + c4, b2);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * ld; e += JT) {
        float t = 0.f;
        for (int r = 0; r < rls; ++r) t += s_red[(size_t)r * 2 * ld + e];
        part[(size_t)blockIdx.x * 2 * ld + e] = t;
    }
}
// MODE 0 -> out0 = sum / R (mean); MODE 1 -> out0 = sum / R (biased variance), out1 = rstd; MODE 2 -> out0 = sum dy', out1 = sum dy' xhat
__global__ __launch_bounds__(JT) void bn_fold_kernel(const float* __restrict__ part, int nb, int ld, int mode, float invR, float eps,
                                                     float* __restrict__ out0, float* __restrict__ out1) {
    const int e = blockIdx.x * JT + threadIdx.x;
    if (e >= 2 * ld) return;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += (double)part[(size_t)b * 2 * ld + e];
    const int which = e / ld, c = e - which * ld;
    if (mode == 2) {
        (which == 0 ? out0 : out1)[c] = (float)t;
    } else if (which == 0) {
        const float v = (float)(t * (double)invR);
        out0[c] = v;
        if (mode == 1 && out1) out1[c] = 1.0f / sqrtf(v + eps);
    }
}

template <typename T>
__global__ __launch_bounds__(JT) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int64_t R,
                                                      int ld, T* __restrict__ y) {
    const int cv = ld >> 2;
    const int64_t total = R * cv;
    for (int64_t id = (int64_t)blockIdx.x * JT + threadIdx.x; id < total; id += (int64_t)gridDim.x * JT) {
        const int c4 = (int)(id % cv) << 2;
        float v[4], m[4], rs[4], g[4], b[4];
        ld4(x + (id / cv) * ld + c4, v);
        ld4(mean + c4, m);
        ld4(rstd + c4, rs);
        ld4(gamma + c4, g);
        ld4(beta + c4, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (v[k] - m[k]) * rs[k] * g[k] + b[k];
            if (relu) v[k] = v[k] < 0.f ? 0.f : v[k];   // NaN stays NaN, as torch.relu
        }
        st4(y + (id / cv) * ld + c4, v);
    }
}
// dx = gamma rstd (dy' - s1/R - xhat s2/R)   (training)   |   gamma rstd dy'   (eval: invR == 0)
template <typename T>
__global__ __launch_bounds__(JT) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ s1,
                                                          const float* __restrict__ s2, int relu, float invR, int64_t R, int ld,
                                                          T* __restrict__ dx) {
    const int cv = ld >> 2;
    const int64_t total = R * cv;
    for (int64_t id = (int64_t)blockIdx.x * JT + threadIdx.x; id < total; id += (int64_t)gridDim.x * JT) {
        const int c4 = (int)(id % cv) << 2;
        float v[4], d[4], m[4], rs[4], g[4], b[4], a1[4], a2[4], o[4];
        ld4(x + (id / cv) * ld + c4, v);
        ld4(dy + (id / cv) * ld + c4, d);
        ld4(mean + c4, m);
        ld4(rstd + c4, rs);
        ld4(gamma + c4, g);
        ld4(beta + c4, b);
        ld4(s1 + c4, a1);
        ld4(s2 + c4, a2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (v[k] - m[k]) * rs[k];
            const float dd = (relu && !(xh * g[k] + b[k] > 0.f)) ? 0.f : d[k];
            o[k] = g[k] * rs[k] * (dd - invR * a1[k] - xh * invR * a2[k]);
        }
        st4(dx + (id / cv) * ld + c4, o);
    }
}

extern "C" int64_t star_bn_ws(int ld) { return (int64_t)BN_MAX_BLOCKS * 2 * ld; }

static int bn_blocks(int64_t R) {
    int64_t nb = (R + 255) / 256;
    return (int)(nb > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : (nb < 1 ? 1 : nb));
}
#define BN_CHECK(name) STAR_CHECK(R > 0 && ld >= 4 && ld % 4 == 0 && ld <= 1024, name ": bad shape R=%lld ld=%d (ld %% 4 == 0, <= 1024)", (long long)R, ld)

/* batch statistics of x [R, ld] per column: mean, biased variance (two-pass), rstd = 1/sqrt(var + eps) */
extern "C" int star_bn_stats(const float* x, int64_t R, int ld, float eps, float* mean, float* var, float* rstd, float* ws, void* stream) {
    STAR_CHECK(x && mean && var && rstd && ws, "star_bn_stats: null pointer");
    BN_CHECK("star_bn_stats");
    hipStream_t s = (hipStream_t)stream;
    const int nb = bn_blocks(R);
    const size_t lds = (size_t)(JT / (ld / 4) > 0 ? JT / (ld / 4) : 1) * 2 * ld * sizeof(float);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nb), dim3(JT), lds, s, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, R, ld, ws);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(2 * ld, JT)), dim3(JT), 0, s, ws, nb, ld, 0, 1.0f / (float)R, eps, mean, (float*)nullptr);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nb), dim3(JT), lds, s, x, (const float*)nullptr, mean, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, R, ld, ws);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(2 * ld, JT)), dim3(JT), 0, s, ws, nb, ld, 1, 1.0f / (float)R, eps, var, rstd);
    STAR_LAUNCH_CHECK("star_bn_stats");
    return 0;
}
/* y = ((x - mean) * rstd * gamma + beta), optional ReLU; y in dt */
extern "C" int star_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu, int64_t R,
                             int ld, void* y, int dt, void* stream) {
    STAR_CHECK(x && mean && rstd && gamma && beta && y, "star_bn_apply: null pointer");
    BN_CHECK("star_bn_apply");
    const int nb = stream_blocks(R * (ld / 4));
    if (dt == STAR_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(nb), dim3(JT), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, relu, R, ld, (bf16_t*)y);
    else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(nb), dim3(JT), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, relu, R, ld, (float*)y);
    STAR_LAUNCH_CHECK("star_bn_apply");
    return 0;
}
/* backward of (BatchNorm [+ ReLU]): dgamma, dbeta [ld] written; dx [R, ld] in dt.  training != 0: batch statistics were used. */
extern "C" int star_bn_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int relu, int training, int64_t R, int ld, void* dx, int dt, float* dgamma, float* dbeta, float* ws, void* stream) {
    STAR_CHECK(x && dy && mean && rstd && gamma && beta && dx && dgamma && dbeta && ws, "star_bn_bwd: null pointer");
    BN_CHECK("star_bn_bwd");
    hipStream_t s = (hipStream_t)stream;
    const int nb = bn_blocks(R);
    const size_t lds = (size_t)(JT / (ld / 4) > 0 ? JT / (ld / 4) : 1) * 2 * ld * sizeof(float);
    hipLaunchKernelGGL(bn_reduce_kernel<2>, dim3(nb), dim3(JT), lds, s, x, dy, mean, rstd, gamma, beta, relu, R, ld, ws);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(2 * ld, JT)), dim3(JT), 0, s, ws, nb, ld, 2, 0.f, 0.f, dbeta, dgamma);
    const float invR = training ? 1.0f / (float)R : 0.0f;
    const int nb2 = stream_blocks(R * (ld / 4));
    if (dt == STAR_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(nb2), dim3(JT), 0, s, x, dy, mean, rstd, gamma, beta, dbeta, dgamma, relu, invR, R, ld, (bf16_t*)dx);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nb2), dim3(JT), 0, s, x, dy, mean, rstd, gamma, beta, dbeta, dgamma, relu, invR, R, ld, (float*)dx);
    STAR_LAUNCH_CHECK("star_bn_bwd");
    return 0;
}

// ---- two-class cross entropy over channels-last head outputs -------------------------------------------------------------
// logits of the two heads [R = N*H*W, ld] (C <= ld valid); target float image with element strides (class = (long)target);
// a thread = one pixel.  loss partial per block, decisions int64 [N, C, H, W], optional pred [N, 2, C, H, W], dlogits f32.
__global__ __launch_bounds__(JT) void ce_nhwc_kernel(const float* __restrict__ lfree, const float* __restrict__ locc, const float* __restrict__ target,
                                                     int64_t tn, int64_t tc, int64_t th, int64_t tw, int64_t* __restrict__ ind,
                                                     float* __restrict__ pred, float* __restrict__ dfree, float* __restrict__ docc,
                                                     float* __restrict__ part, int N, int C, int H, int W, int ld, float inv_count) {
    __shared__ float s_l[JT / 64];
    const int64_t R = (int64_t)N * H * W;
    float loss = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * JT + threadIdx.x; r < R; r += (int64_t)gridDim.x * JT) {
        const int x = (int)(r % W);
        const int y = (int)((r / W) % H), n = (int)(r / ((int64_t)W * H));
        for (int c0 = 0; c0 < ld; c0 += 4) {
            float f[4], o[4], df[4] = {0.f, 0.f, 0.f, 0.f}, dgo[4] = {0.f, 0.f, 0.f, 0.f};
            ld4(lfree + r * ld + c0, f);
            ld4(locc + r * ld + c0, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                if (c < C) {
                    const long long cls = (long long)target[(int64_t)n * tn + (int64_t)c * tc + (int64_t)y * th + (int64_t)x * tw];
                    const float mx = fmaxf(f[k], o[k]);
                    const float ef = expf(f[k] - mx), eo = expf(o[k] - mx);
                    const float lse = mx + logf(ef + eo);
                    loss += lse - (cls == 1 ? o[k] : f[k]);
                    const float pf = ef / (ef + eo), po = eo / (ef + eo);
                    df[k] = (pf - (cls == 0 ? 1.f : 0.f)) * inv_count;
                    dgo[k] = (po - (cls == 1 ? 1.f : 0.f)) * inv_count;
                    const int64_t oi = (((int64_t)n * C + c) * H + y) * W + x;
                    if (ind) ind[oi] = o[k] > f[k] ? 1 : 0;        // argmax(softmax): ties go to class 0 (free)
                    if (pred) {
                        pred[((((int64_t)n * 2 + 0) * C + c) * H + y) * W + x] = f[k];
                        pred[((((int64_t)n * 2 + 1) * C + c) * H + y) * W + x] = o[k];
                    }
                }
            }
            if (dfree) st4(dfree + r * ld + c0, df);
            if (docc) st4(docc + r * ld + c0, dgo);
        }
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) s_l[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0 && part) part[blockIdx.x] = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
}

extern "C" int star_ce_nhwc_blocks(int N, int H, int W) {
    const int64_t b = ((int64_t)N * H * W + JT - 1) / JT;
    return (int)(b > 2048 ? 2048 : b);
}
extern "C" int star_ce_nhwc(const float* lfree, const float* locc, const float* target, const int64_t* st, int64_t* ind, float* pred, float* dfree,
                            float* docc, float* loss_partial, int N, int C, int H, int W, int ld, void* stream) {
    STAR_CHECK(lfree && locc && target && st, "star_ce_nhwc: null pointer");
    STAR_CHECK(N > 0 && C > 0 && C <= ld && ld % 4 == 0 && H > 0 && W > 0, "star_ce_nhwc: bad shape");
    hipLaunchKernelGGL(ce_nhwc_kernel, dim3(star_ce_nhwc_blocks(N, H, W)), dim3(JT), 0, (hipStream_t)stream, lfree, locc, target, st[0], st[1], st[2],
                       st[3], ind, pred, dfree, docc, loss_partial, N, C, H, W, ld, 1.0f / ((float)N * C * H * W));
    STAR_LAUNCH_CHECK("star_ce_nhwc");
    return 0;
}
