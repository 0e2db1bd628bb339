// How long does ONE wave spend issuing a global -> LDS DMA piece (global_load_lds_dwordx4, 1 KiB) as a function of the gather shape?  8 waves per CU (the GEMM's
// population), each issues 8 pieces back to back (no waits between them), s_memtime around the burst; L2-resident source.  Shapes: 16 rows x 64 B (a 32-deep bf16
// K tile of a K-contiguous operand), 8 rows x 128 B (whole lines), 2 rows x 512 B (an index-contiguous operand image).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_issue_probe.hip -o build/dma_issue && build/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

template <int RB>   // bytes per row per piece: 64, 128, 512
__global__ __launch_bounds__(512, 1) void probe(const uint8_t* __restrict__ src, int64_t ld, int reps, long long* out) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    LDS uint8_t* smem = (LDS uint8_t*)smem_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = RB / 16, RPI = 1024 / RB, NP = 8;
    const uint8_t* base = src + (int64_t)(blockIdx.x % 8) * (1 << 20);
    int64_t off[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) off[q] = (int64_t)((q * 8 + wave) * RPI + lane / LPR) * ld + (lane % LPR) * 16;
    long long total = 0;
    for (int r = 0; r < reps; ++r) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int q = 0; q < NP; ++q)
            __builtin_amdgcn_global_load_lds((const GLB void*)(base + off[q] + (int64_t)(r & 7) * RB), (LDS void*)(smem + (q * 8 + wave) * 1024), 16, 0, 0);
        const long long t1 = __builtin_readcyclecounter();
        total += t1 - t0;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (lane == 0) out[blockIdx.x * 8 + wave] = total;
}

int main() {
    uint8_t* src; long long* out;
    hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20); hipMalloc(&out, 256 * 8 * 8);
    const int reps = 200;
    long long h[256 * 8];
    for (int pass = 0; pass < 2; ++pass)
        for (int rb : {64, 128, 512}) {
            const int64_t ld = rb == 512 ? 3072 : 3072;
            if (rb == 64) { hipFuncSetAttribute((const void*)probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<64>, dim3(256), dim3(512), 64 * 1024, 0, src, ld, reps, out); }
            if (rb == 128) { hipFuncSetAttribute((const void*)probe<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<128>, dim3(256), dim3(512), 64 * 1024, 0, src, ld, reps, out); }
            if (rb == 512) { hipFuncSetAttribute((const void*)probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<512>, dim3(256), dim3(512), 64 * 1024, 0, src, ld, reps, out); }
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256 * 8; ++i) s += (double)h[i];
            printf("%3d B per row (%2d rows per piece): %.1f s_memtime ticks per piece per wave (8 pieces back to back, 8 waves per CU)\n", rb, 1024 / rb, s / (256.0 * 8 * reps * 8));
        }
    return 0;
}
