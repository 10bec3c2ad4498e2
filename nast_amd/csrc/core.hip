// Error string, version and the in-library event profiler used by bench.py's roofline block.
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
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

// ---- profiler: one (start, stop) event pair per armed launch, per kernel family -------------
namespace {
constexpr int NKIND = 7;       // 1 = tiled GEMM (work = FLOP), 2 = vocabulary row kernels (algorithmic bytes), 3 = alignment DP (bytes),
                               // 4 = small-M GEMM of the greedy decode (FLOP), 5 = log-sum-exp alignment extension (bytes)
struct Fam {
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<std::string> sig; // per pair: problem signature (only kept when NAST_PROF_SHAPES is set)
  std::vector<double> wk;
  double work = 0;
  int64_t launches = 0;
  double ms = 0;
};
struct Prof {
  unsigned mask = 0;
  Fam fam[NKIND];
  std::vector<hipEvent_t> pool;
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
inline bool armed(int kind) { return kind > 0 && kind < NKIND && (g_prof.mask & (1u << kind)); }
}  // namespace

void nast_prof_before(int kind, hipStream_t s) {
  if (!armed(kind)) return;
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  g_prof.fam[kind].ev.push_back(e);
}

bool nast_prof_shapes(int kind) {
  static const bool on = getenv("NAST_PROF_SHAPES") != nullptr;
  return on && armed(kind);
}

void nast_prof_after(int kind, hipStream_t s, double work, const char* sig) {
  if (!armed(kind)) return;
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  Fam& f = g_prof.fam[kind];
  f.ev.push_back(e);
  f.work += work;
  f.launches += 1;
  if (nast_prof_shapes(kind)) {
    f.sig.emplace_back(sig ? sig : "");
    f.wk.push_back(work);
  }
}

extern "C" int nast_prof_begin(int kind) {
  g_prof.mask = kind == 0 ? 0x7Eu : (1u << kind);   // 0 = every family
  for (int k = 0; k < NKIND; ++k) {
    Fam& f = g_prof.fam[k];
    for (hipEvent_t e : f.ev) g_prof.pool.push_back(e);
    f.ev.clear();
    f.sig.clear(); f.wk.clear();
    f.work = 0; f.launches = 0; f.ms = 0;
  }
  return NAST_OK;
}

static void prof_collect() {
  g_prof.mask = 0;
  for (int k = 0; k < NKIND; ++k) {
    Fam& f = g_prof.fam[k];
    struct Agg { double ms = 0, work = 0; int n = 0; };
    std::map<std::string, Agg> by_sig;
    for (size_t i = 0; i + 1 < f.ev.size(); i += 2) {
      hipEventSynchronize(f.ev[i + 1]);
      float t = 0;
      if (hipEventElapsedTime(&t, f.ev[i], f.ev[i + 1]) == hipSuccess) f.ms += t;
      if (i / 2 < f.sig.size()) {
        Agg& a = by_sig[f.sig[i / 2]];
        a.ms += t; a.work += f.wk[i / 2]; a.n += 1;
      }
    }
    if (!by_sig.empty()) {   // tuning aid (NAST_PROF_SHAPES=1): time per problem signature, largest first
      std::vector<std::pair<std::string, Agg>> v(by_sig.begin(), by_sig.end());
      std::sort(v.begin(), v.end(), [](auto& x, auto& y) { return x.second.ms > y.second.ms; });
      fprintf(stderr, "[nast_prof] family %d: %.3f ms over %lld launches\n", k, f.ms, (long long)f.launches);
      for (auto& e : v)
        fprintf(stderr, "[nast_prof]  %8.3f ms  n=%5d  avg %8.2f us  %7.2f Twork/s  %s\n", e.second.ms, e.second.n,
                e.second.ms * 1e3 / e.second.n, e.second.work / (e.second.ms * 1e-3) / 1e12, e.first.c_str());
    }
    for (hipEvent_t e : f.ev) g_prof.pool.push_back(e);
    f.ev.clear();
  }
}

extern "C" int nast_prof_end(double* total_ms, double* total_flops, int64_t* launches) {
  prof_collect();
  if (total_ms) *total_ms = g_prof.fam[1].ms;
  if (total_flops) *total_flops = g_prof.fam[1].work;
  if (launches) *launches = g_prof.fam[1].launches;
  return NAST_OK;
}

extern "C" int nast_prof_get(int kind, double* total_ms, double* total_work, int64_t* launches) {
  if (kind <= 0 || kind >= NKIND) return NAST_EINVAL;
  if (total_ms) *total_ms = g_prof.fam[kind].ms;
  if (total_work) *total_work = g_prof.fam[kind].work;
  if (launches) *launches = g_prof.fam[kind].launches;
  return NAST_OK;
}
