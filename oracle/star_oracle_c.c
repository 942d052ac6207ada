/* TEST INFRASTRUCTURE ONLY -- CPU restatement in plain C of the integer / byte-exact parts of the STAR hot path.
 * Never linked into or called by the product (star_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.  Parity: PINNED by tests/test_oracle_golden.py against tests/golden/warp_full256.npz
 * and the model goldens (outputs of the reference's own files run in the build container).
 *
 * Functions and the reference lines they follow:
 *   oc_late_fusion    late_fusion / ego_late_fusion  star/models/multiagent_mae.py:870-944
 *                     feature_transformation         star/models/mae_base.py:589-614   (two chained nearest grid_samples)
 *                     fusion("union")                star/models/mae_base.py:624-633
 *   oc_argsort_pair   ids_shuffle / ids_restore      star/models/multiagent_mae.py:811-814, mae_base.py:417-421 (stable)
 *   oc_confusion      get_occupancy_IoU confusion    star/utils/metrics.py:14-82 (binary maps, one scale)
 *   oc_voxel_dense    dataset densification          star/datasets/MultiTempDet.py:134-143 (zeros, fancy assignment, rot90(k=3))
 * Float rules of the warp (every operation rounded once, no contraction: build with -ffp-contract=off):
 *   linspace(-1,1,n) in ATen's two-sided fma form, x (n-1) / n;  [x y 1] theta^T as the k-ordered chain
 *   rn(x*t0), fma(y,t1,.), fma(1,t2,.);  un-normalise ((g+1)*size-1)/2;  rintf (half to even);  zero padding. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float base_grid(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    const float lo = fmaf(step, (float)i, -1.0f);
    const float hi = fmaf(-step, (float)(n - 1 - i), 1.0f);
    const float ls = (i < n / 2) ? lo : hi;
    return (ls * (float)(n - 1)) / (float)n;
}

static float unnorm(float g, int size) { return rintf((((g + 1.0f) * (float)size) - 1.0f) / 2.0f); }

/* one nearest grid_sample of a [C,H,W] map under theta (2x3, float32): out[c][y][x] = in[c][sy][sx] or 0 */
static void sample(const float* in, float* out, const float th[6], int C, int H, int W) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float bx = base_grid(x, W), by = base_grid(y, H);
            float gx = bx * th[0];
            gx = fmaf(by, th[1], gx);
            gx = fmaf(1.0f, th[2], gx);
            float gy = bx * th[3];
            gy = fmaf(by, th[4], gy);
            gy = fmaf(1.0f, th[5], gy);
            const float fx = unnorm(gx, W), fy = unnorm(gy, H);
            const int ok = fx >= 0.0f && fx < (float)W && fy >= 0.0f && fy < (float)H;
            for (int c = 0; c < C; ++c)
                out[((size_t)c * H + y) * W + x] = ok ? in[((size_t)c * H + (int)fy) * W + (int)fx] : 0.0f;
        }
}

/* maps, ego (or NULL), out: float32 [A*B, C, H, W] agent-major (row a*B + b); trans: [B, A, A, 4, 4] float64 or
 * float32 (trans_is_f64); num_agent int64 [B, A] (column 0 is used, as in the reference). */
int oc_late_fusion(const float* maps, const float* ego, const void* trans, int trans_is_f64, const int64_t* num_agent,
                   float* out, int B, int A, int C, int H, int W) {
    const size_t frame = (size_t)C * H * W;
    float* rot = (float*)malloc(frame * sizeof(float));
    float* wrp = (float*)malloc(frame * sizeof(float));
    float* acc = (float*)malloc(frame * sizeof(float));
    if (!rot || !wrp || !acc) return -1;
    memcpy(out, maps, (size_t)A * B * frame * sizeof(float));
    for (int b = 0; b < B; ++b) {
        int64_t n = num_agent[(size_t)b * A];
        if (n > A) n = A;
        for (int i = 0; i < n; ++i) {
            memcpy(acc, (ego ? ego : maps) + ((size_t)i * B + b) * frame, frame * sizeof(float));
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                float th_r[6], th_t[6] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f};
                const size_t m = (((size_t)b * A + i) * A + j) * 16;
                if (trans_is_f64) {   /* theta entries are formed in the SOURCE dtype, then cast (mae_base.py:593-606) */
                    const double* t = (const double*)trans + m;
                    th_r[0] = (float)t[0]; th_r[1] = (float)t[1]; th_r[2] = 0.0f;
                    th_r[3] = (float)t[4]; th_r[4] = (float)t[5]; th_r[5] = 0.0f;
                    th_t[2] = (float)((4 * t[3]) / 128);
                    th_t[5] = (float)(-(4 * t[7]) / 128);
                } else {
                    const float* t = (const float*)trans + m;
                    th_r[0] = t[0]; th_r[1] = t[1]; th_r[2] = 0.0f;
                    th_r[3] = t[4]; th_r[4] = t[5]; th_r[5] = 0.0f;
                    th_t[2] = (4 * t[3]) / 128;
                    th_t[5] = -(4 * t[7]) / 128;
                }
                sample(maps + ((size_t)j * B + b) * frame, rot, th_r, C, H, W);
                sample(rot, wrp, th_t, C, H, W);
                for (size_t e = 0; e < frame; ++e) acc[e] += wrp[e];
            }
            float* o = out + ((size_t)i * B + b) * frame;
            for (size_t e = 0; e < frame; ++e) o[e] = acc[e] > 0.0f ? 1.0f : acc[e];
        }
    }
    free(rot); free(wrp); free(acc);
    return 0;
}

/* stable ascending argsort of each row of noise [R, L] and its inverse permutation */
typedef struct { float v; int64_t i; } oc_key;
static int oc_cmp(const void* a, const void* b) {
    const oc_key* x = (const oc_key*)a; const oc_key* y = (const oc_key*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return x->i < y->i ? -1 : (x->i > y->i);
}
int oc_argsort_pair(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, int R, int L) {
    oc_key* k = (oc_key*)malloc((size_t)L * sizeof(oc_key));
    if (!k) return -1;
    for (int r = 0; r < R; ++r) {
        for (int i = 0; i < L; ++i) { k[i].v = noise[(size_t)r * L + i]; k[i].i = i; }
        qsort(k, (size_t)L, sizeof(oc_key), oc_cmp);
        for (int i = 0; i < L; ++i) {
            ids_shuffle[(size_t)r * L + i] = k[i].i;
            ids_restore[(size_t)r * L + k[i].i] = i;
        }
    }
    free(k);
    return 0;
}

/* 2x2 confusion counts conf[target][pred] of two {0,1} float maps with n elements */
int oc_confusion(const float* pred, const float* target, int64_t n, int64_t conf[4]) {
    conf[0] = conf[1] = conf[2] = conf[3] = 0;
    for (int64_t e = 0; e < n; ++e) conf[(target[e] != 0.0f) * 2 + (pred[e] != 0.0f)] += 1;
    return 0;
}

/* frames of sparse voxel indices -> dense float32 [F, Y, X, Z] as the dataset builds each frame: v[x][y][z] = 1 for
 * every index row (negative indices wrap once, as numpy's), then rot90(k=3) over axes (0,1): out[i][j] = v[X-1-j][i].
 * idx int64 [n,3], offsets int64 [F+1].  Returns -2 on an index numpy would reject. */
int oc_voxel_dense(const int64_t* idx, const int64_t* offsets, float* out, int F, int X, int Y, int Z) {
    const size_t frame = (size_t)X * Y * Z;
    unsigned char* v = (unsigned char*)malloc(frame);
    if (!v) return -1;
    for (int f = 0; f < F; ++f) {
        memset(v, 0, frame);
        for (int64_t p = offsets[f]; p < offsets[f + 1]; ++p) {
            int64_t x = idx[p * 3], y = idx[p * 3 + 1], z = idx[p * 3 + 2];
            if (x < 0) x += X;
            if (y < 0) y += Y;
            if (z < 0) z += Z;
            if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) { free(v); return -2; }
            v[((size_t)x * Y + y) * Z + z] = 1;
        }
        float* o = out + (size_t)f * frame;
        for (int i = 0; i < Y; ++i)
            for (int j = 0; j < X; ++j)
                for (int z = 0; z < Z; ++z) o[((size_t)i * X + j) * Z + z] = (float)v[((size_t)(X - 1 - j) * Y + i) * Z + z];
    }
    free(v);
    return 0;
}
