// sc_timing.hip -- per-entry-point timing with hipEvents recorded by the library itself on the stream each call is
// launched on (sc_last_timing).  Off by default; when on, every compute entry point brackets
// its launches with two events.  sc_last_timing() waits for the recorded events, reports (name, milliseconds) in call
// order and forgets them.  Events are pooled; nothing is allocated on the timed path after the first few calls.
#include <string.h>
#include <mutex>
#include <vector>
#include "sc_common.h"

namespace {
struct Rec { const char* name; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

ScTimed::ScTimed(const char* name, void* stream) : slot(-1), st(stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    Rec r{name, take_event(), take_event()};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}

ScTimed::~ScTimed() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, (hipStream_t)st);
}

extern "C" int sc_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    return SC_OK;
}

extern "C" int sc_last_timing(sc_timing* out, int max_entries, int* n_entries) {
    SC_REQUIRE(n_entries != nullptr && (out != nullptr || max_entries == 0), "NULL argument");
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
            (void)hipGetLastError();
            ms = -1.f;
        }
        if (n < max_entries) {
            strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
            out[n].name[sizeof(out[n].name) - 1] = 0;
            out[n].ms = ms;
            ++n;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    *n_entries = n;
    return SC_OK;
}
