// Error string, version and the in-library event profiler used by bench.py's roofline block.
#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void nast_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nast_last_error(void) { return g_err; }
extern "C" int nast_version(void) { return 100; }

// ---- profiler: one (start, stop) event pair per armed launch -------------------------------
namespace {
struct Prof {
  int kind = 0;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<hipEvent_t> pool;
  double flops = 0;
  int64_t launches = 0;
} g_prof;

hipEvent_t get_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

void nast_prof_before(int kind, hipStream_t s) {
  if (g_prof.kind != kind) return;
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  g_prof.ev.push_back(e);
}

void nast_prof_after(int kind, hipStream_t s, double flops) {
  if (g_prof.kind != kind) return;
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  g_prof.ev.push_back(e);
  g_prof.flops += flops;
  g_prof.launches += 1;
}

extern "C" int nast_prof_begin(int kind) {
  g_prof.kind = kind;
  g_prof.flops = 0;
  g_prof.launches = 0;
  for (hipEvent_t e : g_prof.ev) g_prof.pool.push_back(e);
  g_prof.ev.clear();
  return NAST_OK;
}

extern "C" int nast_prof_end(double* total_ms, double* total_flops, int64_t* launches) {
  g_prof.kind = 0;
  double ms = 0;
  for (size_t i = 0; i + 1 < g_prof.ev.size(); i += 2) {
    hipEventSynchronize(g_prof.ev[i + 1]);
    float t = 0;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) == hipSuccess) ms += t;
  }
  for (hipEvent_t e : g_prof.ev) g_prof.pool.push_back(e);
  g_prof.ev.clear();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = g_prof.flops;
  if (launches) *launches = g_prof.launches;
  return NAST_OK;
}
