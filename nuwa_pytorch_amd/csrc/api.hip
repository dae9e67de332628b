// libamdnuwa misc entry points: ABI version, error strings, and an opt-in HIP-event launch timer
// (used by bench.py to measure the dominant kernel's average launch duration inside the timed region).
#include "common.h"
#include "../../include/amdnuwa.h"
#include <vector>
#include <mutex>

int g_amdnuwa_tuning[32] = {0};

extern "C" int amdnuwa_abi_version(void) { return 19; }

// fp16 saturation monitor: one counter word per translation unit with saturating fp16 stores (common.h: AMDNUWA_SAT_ACCESSOR)
extern "C" unsigned amdnuwa_sat_elementwise(int), amdnuwa_sat_gemm(int), amdnuwa_sat_sparse3dna(int), amdnuwa_sat_xattn(int), amdnuwa_sat_xattn2(int), amdnuwa_sat_xattn6(int);
extern "C" unsigned long long amdnuwa_f16_sat_count(int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    return (unsigned long long)amdnuwa_sat_elementwise(reset) + amdnuwa_sat_gemm(reset) + amdnuwa_sat_sparse3dna(reset) + amdnuwa_sat_xattn(reset) +
           amdnuwa_sat_xattn2(reset) + amdnuwa_sat_xattn6(reset);
}

extern "C" int amdnuwa_set_tuning(int key, int value) {
    if (key < 0 || key >= 32) return AMDNUWA_ERR_ARG;
    g_amdnuwa_tuning[key] = value;
    return AMDNUWA_OK;
}
extern "C" int amdnuwa_get_tuning(int key) { return (key < 0 || key >= 32) ? 0 : g_amdnuwa_tuning[key]; }

extern "C" const char* amdnuwa_error_string(int code) {
    if (code == 0) return "ok";
    if (code == AMDNUWA_ERR_ARG) return "amdnuwa: invalid argument";
    if (code == AMDNUWA_ERR_UNSUPPORTED) return "amdnuwa: unsupported shape/configuration for the gfx950 kernels";
    if (code == AMDNUWA_ERR_WORKSPACE) return "amdnuwa: workspace missing or too small";
    if (code == AMDNUWA_ERR_COMM) return "amdnuwa: RCCL error (amdnuwa_comm_last_error() has the text)";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "amdnuwa: unknown error";
}

// ---- launch timer -------------------------------------------------------------------------------
// amdnuwa_timer_begin/_end bracket ONE launch on `stream` with a pair of hipEvents (only while the
// timer is armed; otherwise they are no-ops costing one branch).  amdnuwa_timer_collect() synchronises
// and returns the summed elapsed ms and the number of bracketed launches.
namespace {
struct Pair { hipEvent_t a, b; };
std::vector<Pair> g_pairs;
std::vector<Pair> g_pool;
std::mutex g_mu;
bool g_armed = false;
}  // namespace

extern "C" void amdnuwa_timer_arm(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_armed = on != 0;
}

extern "C" int amdnuwa_timer_begin(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_armed) return 0;
    Pair p;
    if (!g_pool.empty()) { p = g_pool.back(); g_pool.pop_back(); }
    else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return (int)hipGetLastError();
    }
    g_pairs.push_back(p);
    return (int)hipEventRecord(p.a, stream);
}

extern "C" int amdnuwa_timer_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_armed || g_pairs.empty()) return 0;
    return (int)hipEventRecord(g_pairs.back().b, stream);
}

// per-launch durations in bracketing order (ms[i] of the i-th amdnuwa_timer_begin since the last collect); returns like _collect and
// leaves the pairs collected.  cap < number of pairs: AMDNUWA_ERR_ARG, nothing consumed.
extern "C" int amdnuwa_timer_collect_each(double* ms, long long cap, long long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!ms || cap < (long long)g_pairs.size()) { if (launches) *launches = (long long)g_pairs.size(); return AMDNUWA_ERR_ARG; }
    long long n = 0;
    for (auto& p : g_pairs) {
        if (hipEventSynchronize(p.b) != hipSuccess) return (int)hipGetLastError();
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) != hipSuccess) return (int)hipGetLastError();
        ms[n++] = t;
        g_pool.push_back(p);
    }
    g_pairs.clear();
    if (launches) *launches = n;
    return 0;
}

extern "C" int amdnuwa_timer_collect(double* total_ms, long long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    double tot = 0.0;
    long long n = 0;
    for (auto& p : g_pairs) {
        if (hipEventSynchronize(p.b) != hipSuccess) return (int)hipGetLastError();
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) return (int)hipGetLastError();
        tot += ms;
        ++n;
        g_pool.push_back(p);
    }
    g_pairs.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}
