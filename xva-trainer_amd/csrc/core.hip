// core.hip — error reporting and library identification for libxvahip.so.
#include "xva_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void xva_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xva_last_error(void) { return g_err; }
extern "C" int xva_abi_version(void) { return 1; }
extern "C" const char* xva_target_arch(void) { return "gfx950"; }

// ---- optional per-launch GEMM timing (bench.py roofline leg) -------------------------------------------
// When enabled, xva_gemm brackets every launch with a pair of HIP events on the launch stream and records the
// algorithmic FLOPs (2*M*N*K*batch).  xva_prof_collect synchronises the events and returns totals.  Off by default:
// the timed throughput region never runs with it on.
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; double flops, bytes; int variant; int M, N, K, batch, splitk, bn, tag; };
bool g_prof_on = false;
int g_prof_tag = 0;
std::vector<ProfRec> g_prof;
}
extern "C" void xva_prof_enable(int on) { g_prof_on = on != 0; }
// free-form section tag the engines attach to the launches that follow (phase / network / layer: see tools/hg_gemm_profile.py)
void xva_prof_tag(int tag) { g_prof_tag = tag; }
bool xva_prof_is_on() { return g_prof_on; }
void xva_prof_shape(int M, int N, int K, int batch, int splitk, int bn, double bytes) {
    if (g_prof.empty()) return;
    ProfRec& r = g_prof.back(); r.M = M; r.N = N; r.K = K; r.batch = batch; r.splitk = splitk; r.bn = bn; r.bytes = bytes;
}
void xva_prof_begin(hipStream_t st, double flops, int variant) {
    ProfRec r; r.flops = flops; r.bytes = 0.0; r.variant = variant; r.M = r.N = r.K = r.batch = r.splitk = r.bn = 0; r.tag = g_prof_tag;
    hipEventCreate(&r.a); hipEventCreate(&r.b);
    hipEventRecord(r.a, st);
    g_prof.push_back(r);
}
void xva_prof_end(hipStream_t st) { hipEventRecord(g_prof.back().b, st); }
void xva_prof_cancel() { if (!g_prof.empty()) { hipEventDestroy(g_prof.back().a); hipEventDestroy(g_prof.back().b); g_prof.pop_back(); } }
// out[0] = launches, out[1] = total ms, out[2] = total flops ; per-variant (layout*2+compute) in out[3 + 3*v ..]
extern "C" int xva_prof_collect(double* out, int cap) {
    for (int i = 0; i < cap; ++i) out[i] = 0.0;
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        out[0] += 1; out[1] += ms; out[2] += r.flops;
        int o = 3 + 3 * r.variant;
        if (o + 2 < cap) { out[o] += 1; out[o + 1] += ms; out[o + 2] += r.flops; }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    int n = (int)g_prof.size();
    g_prof.clear();
    return n;
}

// ---- raw event / stream helpers for the data-parallel overlap (host code holds them as opaque pointers) ----
extern "C" void* xva_event_create(void) {
    hipEvent_t e = nullptr;
    // (timing-enabled events were tried for the data-parallel bucket events: no difference — a wait issued late resolves when the recording lane has
    // drained whatever the flags; the fix is WHEN the wait is issued: xva_fp_set_bucket_callback)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { xva_set_error("hipEventCreate failed"); return nullptr; }
    return (void*)e;
}
extern "C" int xva_event_destroy(void* e) { return e && hipEventDestroy((hipEvent_t)e) == hipSuccess ? XVA_OK : XVA_ERR_HIP; }
extern "C" int xva_event_record(void* e, void* stream) {
    if (hipEventRecord((hipEvent_t)e, (hipStream_t)stream) != hipSuccess) { xva_set_error("hipEventRecord failed"); return XVA_ERR_HIP; }
    return XVA_OK;
}
extern "C" int xva_stream_wait_event(void* stream, void* e) {
    if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0) != hipSuccess) { xva_set_error("hipStreamWaitEvent failed"); return XVA_ERR_HIP; }
    return XVA_OK;
}

// Dump one CSV line per recorded GEMM launch (variant = layout*3 + mode; bn = tile tag: BN * 1000 + BM for the direct-to-LDS
// kernel, BN for the general one; mbytes = ALGORITHMIC bytes: every distinct operand / result element once) and clear the records.
extern "C" int xva_prof_dump(const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return XVA_ERR_ARG;
    fprintf(f, "variant,M,N,K,batch,splitk,bn,ms,gflop,mbytes,tag\n");
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.5f,%.4f,%.4f,%d\n", r.variant, r.M, r.N, r.K, r.batch, r.splitk, r.bn, ms, r.flops * 1e-9, r.bytes * 1e-6, r.tag);
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    fclose(f);
    int n = (int)g_prof.size();
    g_prof.clear();
    return n;
}
