// How fast does ONE wave per SIMD issue independent 16x16x32 bf16 MFMAs, against two waves per SIMD?  (hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_issue_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC, bool AGPR>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    long long t0 = wall_clock64();
    long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
    long long c1 = __builtin_readcyclecounter();
    long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
template <int NACC, bool AGPR>
void run(const char* name, int threads) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, AGPR><<<256, threads>>>(out, 10, cyc);
    hipEventRecord(e0);
    probe<NACC, AGPR><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NACC;
    const double tf = 256.0 * (threads / 64) * mf * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-34s %d waves/SIMD: %.1f shader cycles per MFMA per wave, %.2f ms, %.0f TFLOP/s\n", name, threads / 256, (double)h[0] / mf, ms, tf);
}
int main() {
    run<16, false>("16 accumulators (VGPR, builtin)", 256);
    run<16, false>("16 accumulators (VGPR, builtin)", 512);
    run<32, false>("32 accumulators (builtin)", 256);
    run<32, false>("32 accumulators (builtin)", 512);
    run<64, true>("64 accumulators (AGPR, inline asm)", 256);
    run<32, true>("32 accumulators (AGPR, inline asm)", 256);
    run<32, true>("32 accumulators (AGPR, inline asm)", 512);
    return 0;
}
