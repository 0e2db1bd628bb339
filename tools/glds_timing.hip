// Phase timeline of the direct-to-LDS kernels' workgroups (prologue / K loop / epilogue), measured with wall_clock64 (100 MHz).
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DXVA_GLDS_TIMING -Ixva-trainer_amd/csrc -Iinclude tools/glds_timing.hip xva-trainer_amd/csrc/core.hip -o build/glds_timing
#include "gemm_glds.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace xva_glds;

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static void* dev_bf16(size_t n, float scale, bool zero = false) {
    std::vector<uint16_t> h(n);
    uint32_t s = 12345u + (uint32_t)n;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = zero ? 0 : f2b(scale * ((int)(s >> 8 & 0xffff) - 32768) / 32768.f); }
    void* d; hipMalloc(&d, n * 2); hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d;
}

template <class F>
static void run(const char* name, long nblocks, F launch) {
    unsigned long long* dt; hipMalloc(&dt, nblocks * 64); hipMemset(dt, 0, nblocks * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_glds_timing), &dt, sizeof(dt));
    for (int it = 0; it < 3; ++it) launch();
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> t(nblocks * 8);
    hipMemcpy(t.data(), dt, nblocks * 64, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (long i = 0; i < nblocks; ++i) { t0 = std::min(t0, t[i * 8]); t1 = std::max(t1, t[i * 8 + 3]); }
    double ph[3] = {0, 0, 0};
    std::vector<double> starts(nblocks), durs(nblocks);
    for (long i = 0; i < nblocks; ++i) {
        for (int k = 0; k < 3; ++k) ph[k] += (double)(t[i * 8 + k + 1] - t[i * 8 + k]) * 0.01;
        starts[i] = (double)(t[i * 8] - t0) * 0.01; durs[i] = (double)(t[i * 8 + 3] - t[i * 8]) * 0.01;
    }
    std::sort(starts.begin(), starts.end());
    std::vector<double> ds = durs; std::sort(ds.begin(), ds.end());
    printf("%-34s blocks %5ld  event %7.1f us  span %7.1f us | mean prologue %6.2f  loop %6.2f  epilogue %6.2f us | block dur p10 %5.1f p50 %5.1f p90 %5.1f\n",
           name, nblocks, ms * 1e3, (double)(t1 - t0) * 0.01, ph[0] / nblocks, ph[1] / nblocks, ph[2] / nblocks, ds[nblocks / 10], ds[nblocks / 2], ds[nblocks * 9 / 10]);
    printf("    start-time quantiles (us):");
    for (int q = 0; q <= 10; ++q) printf(" %6.1f", starts[std::min(nblocks - 1, nblocks * q / 10)]);
    printf("\n");
    hipFree(dt);
}

static void fastpitch_shapes() {
    const int R = 32 * 862;
    void* x = dev_bf16((size_t)(R + 4) * 384, 1.f); void* h = dev_bf16((size_t)(R + 4) * 1536, 1.f);
    void* W1 = dev_bf16((size_t)1536 * 1152, 0.05f); void* W2 = dev_bf16((size_t)384 * 4608, 0.05f);
    void* o1 = dev_bf16((size_t)R * 1536, 0.f, true); void* o2 = dev_bf16((size_t)R * 384, 0.f, true);
    std::vector<float> hb(1536, 0.1f); float* bias; hipMalloc(&bias, 1536 * 4); hipMemcpy(bias, hb.data(), 1536 * 4, hipMemcpyHostToDevice);
    auto base = [&]() { xva_gemm_params p; memset(&p, 0, sizeof(p)); p.batch = 1; p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f; p.splitk = 1; p.compute = 1;
                        p.mask_mul = 1; p.a_dtype = p.b_dtype = p.c_dtype = XVA_BF16; return p; };
    {   // conv1 forward: [R x 1152] (overlapping rows of x) x W1^T -> [R x 1536], bias + ReLU
        xva_gemm_params p = base(); p.layout = XVA_GEMM_NT; p.A = x; p.B = W1; p.C = o1; p.M = R; p.N = 1536; p.K = 1152; p.lda = 384; p.ldb = 1152; p.ldc = 1536;
        p.bias = bias; p.act = XVA_ACT_RELU;
        run("FastPitch conv1 fwd 256x256", (long)((R + 255) / 256) * 6, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(p, 2, 0); });
        run("FastPitch conv1 fwd 256x256 staggered", (long)((R + 255) / 256) * 6, [&] { launch_tile8<XVA_GEMM_NT>(p, 2, 0); });
        if (XVA_GLDS_ABLATE) {   // ablation builds: the full grid and one round only
            xva_gemm_params q1 = p; q1.M = 256 * 42;
            run("  staggered, 252 workgroups", 252, [&] { launch_tile8<XVA_GEMM_NT>(q1, 2, 0); });
            run("  same, 252 workgroups", 252, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(q1, 2, 0); });
            return;
        }
        run("FastPitch conv1 fwd 128x128", (long)((R + 127) / 128) * 12, [&] { launch_tile<XVA_GEMM_NT, 128, 128, 64, 64>(p, 2, 0); });
        xva_gemm_params q = p; q.M = 256 * 10;      // 60 workgroups: a quarter of the CUs busy
        run("  same, 60 workgroups only", 60, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(q, 2, 0); });
        q.M = 256 * 42;                             // 252 workgroups: one round
        run("  same, 252 workgroups", 252, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(q, 2, 0); });
        q = p; q.c_dtype = XVA_F32; q.M = 256 * 42; q.C = h;   // fp32 output (twice the bytes), 252 workgroups
        run("  252 workgroups, fp32 C", 252, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(q, 2, 0); });
    }
    {   // conv2 forward: [R x 4608] x W2^T -> [R x 384]
        xva_gemm_params p = base(); p.layout = XVA_GEMM_NT; p.A = h; p.B = W2; p.C = o2; p.M = R; p.N = 384; p.K = 4608; p.lda = 1536; p.ldb = 4608; p.ldc = 384;
        p.bias = bias;
        run("FastPitch conv2 fwd 256x256", (long)((R + 255) / 256) * 2, [&] { launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(p, 2, 0); });
        run("FastPitch conv2 fwd 128x128", (long)((R + 127) / 128) * 3, [&] { launch_tile<XVA_GEMM_NT, 128, 128, 64, 64>(p, 2, 0); });
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 0) { fastpitch_shapes(); return 0; }
    const int nseq = argc > 1 ? atoi(argv[1]) : 64;
    const int PAD = 32;
    for (int C : {128, 64, 32}) {
        const int T = C == 128 ? 2048 : (C == 64 ? 4096 : 8192), Hp = T + 2 * PAD;
        const long rows = (long)nseq * Hp;
        void* xs = dev_bf16((rows + 2 * PAD + 64) * C, 1.f);
        void* R = dev_bf16(rows * C, 1.f);
        void* y = dev_bf16(rows * C, 0.f, true);
        std::vector<float> hb(C, 0.1f); float* bias; hipMalloc(&bias, C * 4); hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice);
        for (int k : {3, 11}) {
            const int d = 1, P_ = d * (k - 1) / 2;
            void* W = dev_bf16((size_t)C * k * C, 0.05f);
            xva_gemm_params p; memset(&p, 0, sizeof(p));
            p.A = (const uint16_t*)xs + (long)(PAD - P_) * C; p.B = W; p.C = y;
            p.M = (int)rows; p.N = C; p.K = k * C; p.lda = C; p.ldb = k * C; p.ldc = C; p.batch = 1; p.batch2 = 1;
            p.a_seglen = C; p.a_segadj = (long)d * C - C; p.alpha = 1.f; p.beta = 1.f; p.bias = bias;
            p.R = R; p.ldr = C; p.r_dtype = XVA_BF16; p.mask_mode = XVA_MASK_PAD; p.Tp = Hp; p.mask_pad = PAD; p.mask_len = T; p.mask_mul = 1;
            p.splitk = 1; p.compute = 1; p.layout = XVA_GEMM_NT; p.a_dtype = p.b_dtype = p.c_dtype = XVA_BF16;
            char name[64];
            const long nb = (rows + 127) / 128;
            for (int variant = 0; variant < 3; ++variant) {
                xva_gemm_params q = p;
                if (variant == 1) { q.R = nullptr; q.mask_mode = XVA_MASK_NONE; q.bias = nullptr; }
                if (variant == 2) { q.a_lrelu = 1; q.a_slope = 0.1f; }
                snprintf(name, sizeof(name), "conv_res C=%d k=%d %s", C, k, variant == 0 ? "bias+R+mask" : (variant == 1 ? "bare epilogue" : "a_lrelu"));
                if (C == 128) run(name, nb, [&] { launch_conv_res<XVA_GEMM_NT, 128, 128, 64, 64>(q, 2, d, 1, q.lda, 0); });
                else if (C == 64) run(name, nb, [&] { launch_conv_res<XVA_GEMM_NT, 64, 64, 32, 64>(q, 2, d, 1, q.lda, 0); });
                else run(name, nb, [&] { launch_conv_res<XVA_GEMM_NT, 32, 32, 32, 32>(q, 2, d, 1, q.lda, 0); });
            }
            hipFree(W);
        }
        hipFree(xs); hipFree(R); hipFree(y); hipFree(bias);
    }
    return 0;
}
