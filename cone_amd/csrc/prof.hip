// Opt-in per-launch timing (hipEvents on the launch stream) for the kernels bench.py prices against
// their roofline.  Off by default: a disabled ProfScope is two branches.  Not thread-safe: enable it
// from the one thread that drives the stream being measured.
#include <vector>

#include "common.h"

namespace cone {

struct ProfRec { hipEvent_t e0, e1; int kind; int64_t a, b, c; int m_slot; int64_t a_off; };

static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_used = 0;
static int* g_host_m = nullptr;  // pinned landing slots for device-side row counts
static int g_host_m_cap = 0, g_host_m_used = 0;

bool prof_enabled() { return g_on; }

static hipEvent_t take_event() {
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}

// A scope opened inside another one (a launcher that splits its rows over two kernels opens one for the whole job)
// records nothing: the outer record covers both launches.
static int g_depth = 0;

ProfScope::ProfScope(int kind, int64_t a, int64_t b, int64_t c, const int* a_dev, hipStream_t s, int64_t a_off) : idx_(-1), s_(s), counted_(false) {
    if (!g_on) return;
    counted_ = true;
    if (g_depth++ > 0) return;
    ProfRec r{take_event(), take_event(), kind, a, b, c, -1, a_off};
    if (!r.e0 || !r.e1) return;
    if (a_dev && g_host_m && g_host_m_used < g_host_m_cap) {
        r.m_slot = g_host_m_used++;
        (void)hipMemcpyAsync(g_host_m + r.m_slot, a_dev, sizeof(int), hipMemcpyDeviceToHost, s);
    }
    (void)hipEventRecord(r.e0, s);
    idx_ = (int)g_recs.size();
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (counted_) --g_depth;
    if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].e1, s_);
}

}  // namespace cone

extern "C" int cone_prof_enable(int on) {
    using namespace cone;
    g_recs.clear();
    g_pool_used = 0;
    g_host_m_used = 0;
    g_depth = 0;
    if (on && !g_host_m) {
        g_host_m_cap = 1 << 16;
        CONE_CHECK_HIP(hipHostMalloc((void**)&g_host_m, sizeof(int) * g_host_m_cap, hipHostMallocDefault));
    }
    g_on = on != 0;
    return 0;
}

extern "C" int64_t cone_prof_collect(double* out, int64_t max_rec) {
    using namespace cone;
    int64_t n = 0;
    for (auto& r : g_recs) {
        if (n >= max_rec) break;
        if (hipEventSynchronize(r.e1) != hipSuccess) { set_error("prof_collect: event sync failed"); return CONE_E_HIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { set_error("prof_collect: elapsed failed"); return CONE_E_HIP; }
        int64_t a = r.a;
        if (r.m_slot >= 0) { int64_t md = (int64_t)g_host_m[r.m_slot] - r.a_off; md = md < 0 ? 0 : md; a = md < a ? md : a; }
        double* o = out + n * 5;
        o[0] = r.kind; o[1] = (double)a; o[2] = (double)r.b; o[3] = (double)r.c; o[4] = ms;
        ++n;
    }
    return n;
}
