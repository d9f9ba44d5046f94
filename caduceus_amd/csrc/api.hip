// Library identification, status strings and the opt-in HIP-event kernel timer behind cad_prof_*.
#include <mutex>
#include <vector>

#include "cad_common.h"

// A build with -D tuning defines says so (CAD_VARIANT = the define list, caduceus_amd/_build.py); a build whose kernels were cut down for
// TIMING experiments (-DSC_WHATIF=<bits> / -DSC_TIMING / -DGS_WHATIF=<bits>: wrong results by construction) carries the marker the loader refuses
// (caduceus_amd/_lib.py: only with CADUCEUS_AMD_ALLOW_TIMING_BUILD=1, which bench.py's floor worker and the A/B tools set).
#if (defined(SC_WHATIF) && SC_WHATIF != 0) || defined(SC_TIMING) || (defined(GS_WHATIF) && GS_WHATIF != 0)
#define CAD_TIMING_TAG " TIMING-BUILD (wrong results by construction; never the product)"
#else
#define CAD_TIMING_TAG ""
#endif
#ifdef CAD_VARIANT
#define CAD_VARIANT_TAG " variant[" CAD_VARIANT "]"
#else
#define CAD_VARIANT_TAG ""
#endif
extern "C" const char* cad_version(void) {
#ifdef CAD_EMU
    return "caduceus_amd 0.1.0 (host emulator build - tests only)";
#else
#ifdef CAD_SRC_HASH
    return "caduceus_amd 0.1.0 (hip gfx950) src " CAD_SRC_HASH CAD_VARIANT_TAG CAD_TIMING_TAG;
#else
    return "caduceus_amd 0.1.0 (hip gfx950)" CAD_VARIANT_TAG CAD_TIMING_TAG;
#endif
#endif
}

extern "C" int cad_is_device_build(void) { return CAD_DEVICE_BUILD; }

extern "C" const char* cad_status_string(int s) {
    switch (s) {
        case CAD_OK: return "ok";
        case CAD_ERR_BAD_ARG: return "bad argument (null pointer, shape or alignment)";
        case CAD_ERR_UNSUPPORTED: return "unsupported configuration (dtype / size outside the compiled range)";
        case CAD_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

int cad_after_launch() { return hipGetLastError() == hipSuccess ? CAD_OK : CAD_ERR_LAUNCH; }

int cad_cu_count() {
#ifdef CAD_EMU
    return 256;
#else
    static int cus[CAD_MAX_DEVICES] = {0};  // (benign race: every thread writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CAD_MAX_DEVICES) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
#endif
}

namespace {
std::mutex g_mu;
unsigned g_mask = 0;  // bit k: kind k is timed
#ifndef CAD_EMU
struct Rec {
    int kind, dev;
    hipEvent_t a, b;
};
std::vector<Rec> g_recs;
// events of earlier measurements, reused: no hipEventCreate on the launch path after the first pass.  One pool PER DEVICE: an event
// belongs to the device it was created on and cannot be recorded on another device's stream (multi-GPU single process)
std::vector<hipEvent_t> g_free[CAD_MAX_DEVICES];
int cur_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CAD_MAX_DEVICES) return -1;
    return d;
}
bool take_event(int dev, hipEvent_t* e) {
    if (dev < 0) return false;
    if (!g_free[dev].empty()) {
        *e = g_free[dev].back();
        g_free[dev].pop_back();
        return true;
    }
    return hipEventCreate(e) == hipSuccess;
}
#else
int64_t g_counts[CAD_PROF_KINDS];
#endif
}  // namespace

CadProfScope::CadProfScope(int k, void* s) : kind(k), stream(s), slot(-1) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!((g_mask >> k) & 1u)) return;
#ifndef CAD_EMU
    Rec r;
    r.kind = k;
    r.dev = cur_device();
    if (!take_event(r.dev, &r.a)) return;
    if (!take_event(r.dev, &r.b)) {
        g_free[r.dev].push_back(r.a);
        return;
    }
    (void)hipEventRecord(r.a, (hipStream_t)s);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
#else
    g_counts[k]++;
#endif
}

CadProfScope::~CadProfScope() {
#ifndef CAD_EMU
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot >= 0 && slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, (hipStream_t)stream);
#endif
}

extern "C" int cad_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = on ? ~0u : 0u;
    return CAD_OK;
}

extern "C" int cad_prof_enable_kinds(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
    return CAD_OK;
}

extern "C" int cad_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
#ifndef CAD_EMU
    for (auto& r : g_recs) {
        // an event still pending (reset without a preceding read / synchronise) is destroyed, not recycled: a recycled pending
        // event would make the next measurement it is used for wait on, or report, the old record
        if (hipEventQuery(r.b) == hipSuccess) {
            g_free[r.dev].push_back(r.a);
            g_free[r.dev].push_back(r.b);
        } else {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
    }
    g_recs.clear();
#else
    for (int i = 0; i < CAD_PROF_KINDS; ++i) g_counts[i] = 0;
#endif
    return CAD_OK;
}

extern "C" int cad_prof_read(int kind, double* total_ms, int64_t* launches) {
    if (kind < 0 || kind >= CAD_PROF_KINDS || !total_ms || !launches) return CAD_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    double t = 0;
    int64_t n = 0;
#ifndef CAD_EMU
    for (auto& r : g_recs) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return CAD_ERR_LAUNCH;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return CAD_ERR_LAUNCH;
        t += ms;
        n++;
    }
#else
    n = g_counts[kind];
#endif
    *total_ms = t;
    *launches = n;
    return CAD_OK;
}
