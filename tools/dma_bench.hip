// LDS-DMA piece-shape microbenchmark (measurement tool, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 tools/dma_bench.hip -o tools/dma_bench && tools/dma_bench
// Every workgroup sweeps a region shared by the workgroups of its XCD (so the bytes come from that XCD's L2 after the first pass)
// with buffer_load_dwordx4 ... lds pieces of 1 KiB whose 64 lanes cover ROWS rows x (1024 / ROWS) contiguous bytes, rows `pitch`
// bytes apart.  Reports GB/s per CU by piece shape, waves per workgroup and pieces kept in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int Q> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory"); }

// ROWS rows per piece; Q pieces in flight per wave
template <int ROWS, int Q>
__global__ __launch_bounds__(1024) void dma_kernel(const unsigned char* __restrict__ src, long long region_bytes, int pitch, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RB = 1024 / ROWS;            // bytes per row of a piece
    constexpr int LPR = RB / 16;               // lanes per row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7;
    const unsigned char* base = src + (long long)xcd * region_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (unsigned)region_bytes, 0x00020000);
    const int colblocks = pitch / RB;                       // pieces side by side in a row block
    const int rowblocks = (int)(region_bytes / ((long long)pitch * ROWS));
    const int total = colblocks * rowblocks;                // pieces in the region
    const unsigned lane_off = (unsigned)((lane / LPR) * pitch + (lane % LPR) * 16);
    unsigned char* my = smem + wave * (Q * 1024);
    // workgroups start at different places of the region, waves take consecutive pieces
    int p = (int)(((long long)(blockIdx.x >> 3) * 7919 * nw + wave) % total);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int rb = p / colblocks, cb = p - rb * colblocks;
            const unsigned off = (unsigned)rb * (unsigned)(pitch * ROWS) + (unsigned)cb * RB + lane_off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my + q * 1024), 16, off, 0, 0, 0);
            p += nw;
            if (p >= total) p -= total;
            if (it > 0 || q == Q - 1) wait_vm<Q - 1>();
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0 && smem[blockIdx.x & 1023] == 123) sink[0] = 1;
}

template <int ROWS, int Q>
static void run(const unsigned char* src, long long region, int pitch, int nw, int* sink) {
    const int cus = 256, iters = 2048 / Q;
    const size_t lds = 1024 * 64 + 1024;      // > 64 KiB: one workgroup per CU... two fit in 160 KiB, so ask for 96 KiB
    const size_t lds_bytes = 96 * 1024;
    (void)lds;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<ROWS, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_kernel<ROWS, Q>), dim3(cus), dim3(nw * 64), lds_bytes, 0, src, region, pitch, iters, sink);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((dma_kernel<ROWS, Q>), dim3(cus), dim3(nw * 64), lds_bytes, 0, src, region, pitch, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)reps * cus * nw * iters * Q * 1024.0;
    printf("rows %2d x %4d B  pitch %5d  waves %2d  in flight %d : %7.1f GB/s per CU  %6.2f TB/s chip  (%.1f us)\n", ROWS, 1024 / ROWS, pitch, nw, Q,
           bytes / (ms * 1e-3) / 1e9 / cus, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / reps);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long region = 2LL << 20;       // per XCD: sits in its 4 MiB L2
    unsigned char* src;
    int* sink;
    CK(hipMalloc(&src, region * 8));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, region * 8));
    CK(hipMemset(sink, 0, 4));
    const int pitches[] = {3072, 1024};
    for (int pi = 0; pi < 2; ++pi) {
        const int pitch = pitches[pi];
        for (int nw : {4, 8, 16}) {
            run<16, 4>(src, region, pitch, nw, sink);
            run<8, 4>(src, region, pitch, nw, sink);
            run<4, 4>(src, region, pitch, nw, sink);
            run<2, 4>(src, region, pitch, nw, sink);
            run<1, 4>(src, region, pitch, nw, sink);
        }
        run<16, 8>(src, region, pitch, 8, sink);
        run<8, 8>(src, region, pitch, 8, sink);
        run<4, 8>(src, region, pitch, 8, sink);
        run<1, 8>(src, region, pitch, 8, sink);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
