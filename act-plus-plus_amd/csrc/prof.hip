// Per-launch HIP-event timing of every kernel class, on the stream the kernel is launched on.
// bench.py enables it around the timed region to obtain the dominant kernel's average launch duration and
// its algorithmic FLOPs/bytes (roofline.achieved); rocprofv3 --kernel-trace --stats must agree with it.
#include "common.h"

#include <map>
#include <sstream>
#include <vector>

namespace {
struct Entry { int cls; hipEvent_t e0, e1; };
struct Cls { std::string name; double flops = 0, bytes = 0, ms = 0; long count = 0; };
bool g_on = false;
std::vector<Cls> g_cls;
std::map<std::string, int> g_idx;
std::vector<Entry> g_entries;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
int g_open = -1;

hipEvent_t get_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

bool prof_enabled() { return g_on; }

void prof_begin(const char* name, double flops, double bytes, hipStream_t st) {
    if (!g_on) return;
    auto it = g_idx.find(name);
    int ci;
    if (it == g_idx.end()) {
        ci = (int)g_cls.size();
        g_idx[name] = ci;
        Cls c; c.name = name;
        g_cls.push_back(c);
    } else ci = it->second;
    g_cls[ci].flops += flops;
    g_cls[ci].bytes += bytes;
    g_cls[ci].count += 1;
    Entry e;
    e.cls = ci;
    e.e0 = get_event();
    e.e1 = get_event();
    if (!e.e0 || !e.e1) return;
    (void)hipEventRecord(e.e0, st);
    g_entries.push_back(e);
    g_open = (int)g_entries.size() - 1;
}

void prof_end(hipStream_t st) {
    if (!g_on || g_open < 0) return;
    (void)hipEventRecord(g_entries[g_open].e1, st);
    g_open = -1;
}

extern "C" int actmi_profile_enable(int on) {
    g_on = on != 0;
    return 0;
}

extern "C" int actmi_profile_reset(void) {
    g_entries.clear();
    g_cls.clear();
    g_idx.clear();
    g_pool_next = 0;
    g_open = -1;
    return 0;
}

// JSON: [{"name":..., "count":n, "ms":total, "flops":total, "bytes":total}, ...]; synchronises the recorded events.
extern "C" int actmi_profile_report(char* buf, int buflen) {
    for (auto& e : g_entries) {
        if (hipEventSynchronize(e.e1) != hipSuccess) return ACTMI_E_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) return ACTMI_E_LAUNCH;
        g_cls[e.cls].ms += ms;
    }
    g_entries.clear();
    g_pool_next = 0;
    std::ostringstream os;
    os << "[";
    for (size_t i = 0; i < g_cls.size(); ++i) {
        const Cls& c = g_cls[i];
        if (i) os << ",";
        os << "{\"name\":\"" << c.name << "\",\"count\":" << c.count << ",\"ms\":" << c.ms << ",\"flops\":" << c.flops
           << ",\"bytes\":" << c.bytes << "}";
    }
    os << "]";
    const std::string s = os.str();
    if ((int)s.size() + 1 > buflen) return ACTMI_E_INVALID;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
