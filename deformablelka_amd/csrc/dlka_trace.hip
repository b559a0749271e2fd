// Launch trace (include/dlka.h: dlka_trace_*): per-kernel durations of whatever sequence of library calls the caller runs between
// dlka_trace_start and dlka_trace_stop, taken with HIP events on the stream each kernel is launched on.  bench.py uses it to report the
// roofline of the kernels the timed step REALLY launches (same entry points, arguments and predecessor state), not of stand-alone
// operator calls.  Not thread-safe, not for hipGraph capture; off unless started.
#include <cxxabi.h>

#include <string>
#include <vector>

#include "dlka_common.h"

namespace dlka {

int g_trace_on = 0;

namespace {
struct TraceRec { hipEvent_t ev; const void *fn; };
std::vector<TraceRec> g_rec;   // g_rec[0] = the start marker (fn == nullptr)
size_t g_used = 0;
bool g_overflow = false;

bool next_event(const void *fn, hipStream_t st)
{
    if (g_used >= g_rec.size()) { g_overflow = true; return false; }
    TraceRec &r = g_rec[g_used];
    if (hipEventRecord(r.ev, st) != hipSuccess) { g_overflow = true; return false; }
    r.fn = fn;
    ++g_used;
    return true;
}
}  // namespace

void trace_after_launch(const void *kernel_fn, hipStream_t st) { (void)next_event(kernel_fn, st); }

}  // namespace dlka

using namespace dlka;

#if !defined(HIPEMU)
extern "C" {

int dlka_trace_start(int max_events, void *stream)
{
    if (max_events <= 0 || max_events > (1 << 20)) return DLKA_ERR_SHAPE;
    if (g_trace_on) return DLKA_ERR_UNSUPPORTED;
    while ((int)g_rec.size() < max_events + 1) {
        TraceRec r;
        r.fn = nullptr;
        if (hipEventCreate(&r.ev) != hipSuccess) return DLKA_ERR_LAUNCH;
        g_rec.push_back(r);
    }
    g_used = 0;
    g_overflow = false;
    if (!next_event(nullptr, (hipStream_t)stream)) return DLKA_ERR_LAUNCH;
    g_trace_on = 1;
    return DLKA_OK;
}

int dlka_trace_mark(void *stream)
{
    if (!g_trace_on) return DLKA_ERR_UNSUPPORTED;
    return next_event(nullptr, (hipStream_t)stream) ? DLKA_OK : DLKA_ERR_WORKSPACE;
}

int dlka_trace_stop(void)
{
    if (!g_trace_on) return DLKA_ERR_UNSUPPORTED;
    g_trace_on = 0;
    if (g_used && hipEventSynchronize(g_rec[g_used - 1].ev) != hipSuccess) return DLKA_ERR_LAUNCH;
    return g_overflow ? DLKA_ERR_WORKSPACE : DLKA_OK;
}

int dlka_trace_count(void) { return g_used ? (int)g_used - 1 : 0; }

int dlka_trace_get(int i, char *name, size_t name_cap, float *ms)
{
    if (g_trace_on || i < 0 || (size_t)i + 1 >= g_used) return DLKA_ERR_SHAPE;
    const TraceRec &r = g_rec[(size_t)i + 1];
    if (ms && hipEventElapsedTime(ms, g_rec[(size_t)i].ev, r.ev) != hipSuccess) return DLKA_ERR_LAUNCH;
    if (name && name_cap) {
        std::string s = "(mark)";
        if (r.fn) {
            const char *mangled = hipKernelNameRefByPtr(r.fn, nullptr);
            s = mangled ? mangled : "(unknown kernel)";
            int status = 0;
            char *dm = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
            if (dm && status == 0) s = dm;
            free(dm);
        }
        const size_t n = s.size() < name_cap - 1 ? s.size() : name_cap - 1;
        memcpy(name, s.data(), n);
        name[n] = 0;
    }
    return DLKA_OK;
}

}  // extern "C"
#else
// host emulator build (tests/emu): no device timeline exists; the entry points report "unsupported"
extern "C" {
int dlka_trace_start(int, void *) { return DLKA_ERR_UNSUPPORTED; }
int dlka_trace_mark(void *) { return DLKA_ERR_UNSUPPORTED; }
int dlka_trace_stop(void) { return DLKA_ERR_UNSUPPORTED; }
int dlka_trace_count(void) { return 0; }
int dlka_trace_get(int, char *, size_t, float *) { return DLKA_ERR_UNSUPPORTED; }
}
#endif
