// Does the global -> LDS DMA path of a CU care whether a wave instruction (64 lanes x 16 B) gathers 16 rows x 64 B (half cache lines: what a 32-deep bf16 K tile
// asks for) or 8 rows x 128 B (whole lines: a 64-deep K tile)?  One workgroup of 8 waves per CU streams a 512-row x K operand panel (the rows 3 072 bytes
// apart, as FastPitch's conv1 A operand) into a ring of LDS with 96 KB in flight, nothing else in the loop.  Prints GB/s per CU and B/clk at 2.1 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_pattern_probe.hip -o build/dma_probe && build/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

template <int KB, int PAIRED = 0>   // KB: bytes of K per row per instruction: 64 (half lines) or 128 (whole lines).  PAIRED (KB = 64): consecutive instructions
                                    // of a wave ask for the two halves of the SAME lines (does the L1 merge them?)
__global__ __launch_bounds__(512, 1) void probe(const uint8_t* __restrict__ src, int64_t ld_bytes, int ktiles, int64_t panel_stride, int* sink) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    LDS uint8_t* smem = (LDS uint8_t*)smem_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPI = 1024 / KB;                 // rows per instruction: 16 or 8
    constexpr int LPR = KB / 16;                   // lanes per row: 4 or 8
    constexpr int TILE = 32 * 1024;                // bytes per "tile" = 512 rows x 64 B ; with KB = 128 a tile is 256 rows x 128 B (same bytes, same flops-equivalent)
    constexpr int NI = TILE / 1024 / 8;            // instructions per wave per tile = 4
    const uint8_t* base = src + (int64_t)(blockIdx.x % 8) * panel_stride;     // one panel per XCD (workgroup i runs on XCD i % 8): after the first touch every fetch is an L2 hit
    int64_t off[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int row = PAIRED ? ((q >> 1) * 8 + wave) * RPI + lane / LPR + (q >> 1) * 0 : (q * 8 + wave) * RPI + lane / LPR;
        off[q] = (int64_t)row * ld_bytes + (lane % LPR) * 16 + (PAIRED ? (q & 1) * 64 : 0);
    }
    constexpr int NS = 4;
    const int kper = 3072 / (PAIRED ? 2 * KB : KB);                     // wrap inside the 3 072-byte rows: the panel is re-read (L2-resident) for `ktiles` tiles
    auto issue = [&](int t) {
        LDS uint8_t* st = smem + (t % NS) * TILE;
#pragma unroll
        for (int q = 0; q < NI; ++q)
            __builtin_amdgcn_global_load_lds((const GLB void*)(base + off[q] + (int64_t)(t % kper) * (PAIRED ? 2 * KB : KB) + (PAIRED ? (int64_t)(t & 0) : 0)), (LDS void*)(st + (q * 8 + wave) * 1024), 16, 0, 0);
    };
    issue(0); issue(1); issue(2);
    for (int t = 0; t < ktiles; ++t) {
        if (t + 3 < ktiles) { issue(t + 3); __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * NI)); }
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
    }
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(LDS int*)smem;
}

int main() {
    const int64_t ld = 3072;                       // bytes between rows
    const int ktiles = 960;                         // 48 x 64 B = 3072 B of K per row (KB = 64) ; 24 x 128 B for KB = 128 moves the same panel
    const int nwg = 256;
    const int64_t panel = 512 * ld;                // each workgroup its own 512 rows
    uint8_t* src; int* sink;
    hipMalloc(&src, nwg * panel + (1 << 20)); hipMemset(src, 1, nwg * panel + (1 << 20)); hipMalloc(&sink, nwg * 4);
    hipFuncSetAttribute((const void*)probe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)probe<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)probe<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass)
        for (int kb : {64, 128, 65}) {
            const int kt = kb == 128 ? ktiles / 2 : ktiles;
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (kb == 65) hipLaunchKernelGGL((probe<64, 1>), dim3(nwg), dim3(512), 128 * 1024, 0, src, ld, kt, panel, sink);
                else if (kb == 64) hipLaunchKernelGGL(probe<64>, dim3(nwg), dim3(512), 128 * 1024, 0, src, ld, kt, panel, sink);
                else hipLaunchKernelGGL(probe<128>, dim3(nwg), dim3(512), 128 * 1024, 0, src, ld, kt, panel, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)kt * 32 * 1024;      // per workgroup
            // KB = 128: a tile is 256 rows x 128 B — per workgroup half the rows, the same bytes
            if (kb == 65) printf("half lines, the two halves of a line in consecutive instructions of the wave:   ");
            printf("%3d B per row per piece (%2d rows x %3d B per wave instruction): %7.1f us  %6.1f GB/s per CU  %5.1f B/clk at 2.1 GHz  (%.2f TB/s chip)\n", kb & ~1, 1024 / (kb & ~1), kb & ~1, best * 1e3,
                   bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.1e9, bytes * nwg / (best * 1e-3) / 1e12);
        }
    return 0;
}
