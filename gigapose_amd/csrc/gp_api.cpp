// Error reporting + version for the C-ABI (include/gigapose_hip.h).
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "gp_common.h"

static thread_local char g_err[512] = "";
// Guard-rail words: one device int32 per HIP device, owned by the caller (gp_set_status_buffer registers the word of the CURRENT device).
// A launch takes the word of the device it is issued on: no process-wide pointer to re-point, nothing shared between GPUs or threads
// (the table is written at registration only; round 5 held ONE pointer the Python side switched per call).
constexpr int kMaxDevices = 64;
static int* g_status[kMaxDevices] = {};
int* gp_status_buffer()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return g_status[dev];
}

void gp_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- optional per-kernel-family timing with HIP events on the launch stream (bench.py roofline leg)
namespace {
struct Rec { hipEvent_t a, b; int kind; double work; };
bool g_prof = false;
int g_prof_only = -1, g_prof_stride = 1, g_prof_seen = 0;  // sampled mode: only launches of one kind, one in `stride`
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
const char* kKindNames[GP_PROF_KINDS] = {"gemm_kmajor", "match_tiles", "attention", "layernorm", "conv", "other", "gemm_split",
                                        "match_split"};
}  // namespace

GpProfScope::GpProfScope(int kind, double work, hipStream_t st) : idx_(-1), st_(st)
{
    if (!g_prof) return;
    if (g_prof_only >= 0 && (kind != g_prof_only || (g_prof_seen++ % g_prof_stride) != 0)) return;
    Rec r{get_event(), get_event(), kind, work};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    idx_ = (int)g_recs.size() - 1;
}
GpProfScope::~GpProfScope()
{
    if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, st_);
}

extern "C" {
void gp_prof_begin(int kind, int stride)   // kind < 0: every launch; else every `stride`-th launch of that family
{
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_prof_only = kind < 0 ? -1 : kind;
    g_prof_stride = stride > 0 ? stride : 1;
    g_prof_seen = 0;
    g_prof = true;
}
int gp_prof_end(int max_kinds, double* ms, double* work, long long* launches)
{
    g_prof = false;
    for (int k = 0; k < max_kinds; ++k) { ms[k] = 0; work[k] = 0; launches[k] = 0; }
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        if (r.kind < max_kinds) { ms[r.kind] += t; work[r.kind] += r.work; launches[r.kind] += 1; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return GP_PROF_KINDS;
}
const char* gp_prof_kind_name(int kind) { return (kind >= 0 && kind < GP_PROF_KINDS) ? kKindNames[kind] : ""; }
const char* gp_last_error(void) { return g_err; }
int gp_set_status_buffer(int* device_word)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        gp_set_error("gp_set_status_buffer: no current HIP device (or its index is beyond %d)", kMaxDevices);
        return GP_EINVAL;
    }
    g_status[dev] = device_word;
    return GP_OK;
}
int gp_abi_version(void) { return 2; }   // 2: round 6 (45 product entry points, per-device status words, probe library split off)
}
