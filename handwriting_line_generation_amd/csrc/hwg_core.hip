// Error reporting, version and device probe for libhwg_hip.so.
#include "hwg_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void hwg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hwg_last_error(void) { return g_err; }
extern "C" int hwg_abi_version(void) { return 6; }

#include <atomic>
#include <vector>
#include <utility>
static std::atomic<unsigned> g_tuning_epoch{1};
unsigned hwg_tuning_epoch() { return g_tuning_epoch.load(std::memory_order_relaxed); }
extern "C" int hwg_tuning_reload(void) { g_tuning_epoch.fetch_add(1); return HWG_OK; }
#include <mutex>
#include <stdlib.h>
static void tune_str(char* dst, size_t cap, const char* name) {
  const char* e = getenv(name);
  snprintf(dst, cap, "%s", e ? e : "");
}
static int tune_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
const HwgTune& hwg_tune() {
  static std::mutex mu;
  static HwgTune* cur = nullptr;                      // snapshots are immutable once published; superseded ones are leaked (reloads are rare)
  thread_local const HwgTune* mine = nullptr;
  const unsigned ep = hwg_tuning_epoch();
  if (mine && mine->epoch == ep) return *mine;
  std::lock_guard<std::mutex> lock(mu);
  if (!cur || cur->epoch != ep) {
    HwgTune* t = new HwgTune;
    t->epoch = ep;
    t->wino = tune_int("HWG_WINO", 1);
    t->wino_wgrad = tune_int("HWG_WINO_WGRAD", 1);
    t->wgrad_narrow = tune_int("HWG_WGRAD_NARROW", 1);
    t->w64_nodma = tune_int("HWG_W64_NODMA", 0);
    t->wino_order = tune_int("HWG_WINO_ORDER", 1);
    t->conv_merge = tune_int("HWG_CONV_MERGE", 1);
    t->wino_s2 = tune_int("HWG_WINO_S2", 1);
    t->wino_wgrad_split = tune_int("HWG_WINO_WGRAD_SPLIT", 0);
    t->wwg_debug = tune_int("HWG_WWG_DEBUG", 0);
    t->conv_pf = tune_int("HWG_CONV_PF", 3);
    t->conv_lds = tune_int("HWG_CONV_LDS", 1);
    t->to1_lanes = tune_int("HWG_TO1_LANES", 1);
    t->wgrad_reduce_rows = tune_int("HWG_WGRAD_REDUCE_ROWS", 1);
    t->split_inkernel = tune_int("HWG_SPLIT_INKERNEL", 0);
    t->norm_fused = tune_int("HWG_NORM_FUSED", 0);
    t->wgrad_c1 = tune_int("HWG_WGRAD_C1", 0);
    t->c1_rows = tune_int("HWG_C1_ROWS", 1);
    t->wgrad_c1_rows = tune_int("HWG_WGRAD_C1_ROWS", 1);
    t->conv_wk = tune_int("HWG_CONV_WK", 2);
    t->c1_mfma = tune_int("HWG_C1_MFMA", 1);
    t->conv_dbg = tune_int("HWG_CONV_DBG", 0);
    tune_str(t->wino_force, sizeof(t->wino_force), "HWG_WINO_FORCE");
    tune_str(t->wino_bal, sizeof(t->wino_bal), "HWG_WINO_BAL");
    tune_str(t->conv_force, sizeof(t->conv_force), "HWG_CONV_FORCE");
    tune_str(t->wgrad_force, sizeof(t->wgrad_force), "HWG_WGRAD_FORCE");
    tune_str(t->wino_cost6, sizeof(t->wino_cost6), "HWG_WINO_COST6");
    cur = t;
  }
  mine = cur;
  return *mine;
}
int* hwg_split_counters(hipStream_t st) {
  static std::mutex mu;
  static std::vector<std::pair<hipStream_t, int*>> bufs;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& b : bufs)
    if (b.first == st) return b.second;
  int* p = nullptr;
  if (hipMalloc(&p, sizeof(int) * HWG_SPLIT_COUNTERS) != hipSuccess || hipMemsetAsync(p, 0, sizeof(int) * HWG_SPLIT_COUNTERS, st) != hipSuccess) {
    hwg_set_error("split counters: allocation of %zu bytes failed", sizeof(int) * (size_t)HWG_SPLIT_COUNTERS);
    return nullptr;
  }
  bufs.emplace_back(st, p);
  return p;
}
// ---- stream fork / join (weight gradients on a side stream): side waits for main's queue as it stands / main waits for side's ----------
// One library call each (hipEventRecord + hipStreamWaitEvent on a ring of reusable events) instead of creating, recording and waiting on
// a framework event object per weight gradient.
namespace {
thread_local hipEvent_t g_fork_ev[64];
thread_local int g_fork_n = 0, g_fork_i = 0;
int stream_dep(hipStream_t from, hipStream_t to, const char* what) {
  if (g_fork_n < 64) {
    if (hipEventCreateWithFlags(&g_fork_ev[g_fork_n], hipEventDisableTiming) != hipSuccess) { hwg_set_error("%s: hipEventCreate failed", what); return HWG_ERR_LAUNCH; }
    ++g_fork_n;
  }
  hipEvent_t ev = g_fork_ev[g_fork_i];
  g_fork_i = (g_fork_i + 1) % g_fork_n;
  if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) { hwg_set_error("%s: event record / wait failed", what); return HWG_ERR_LAUNCH; }
  return HWG_OK;
}
}  // namespace
extern "C" int hwg_stream_fork(void* main_stream, void* side_stream) { return stream_dep((hipStream_t)main_stream, (hipStream_t)side_stream, "stream_fork"); }
extern "C" int hwg_stream_join(void* side_stream, void* main_stream) { return stream_dep((hipStream_t)side_stream, (hipStream_t)main_stream, "stream_join"); }

static int g_last_plan[3] = {-1, -1, -1};   // process-wide on purpose: backward passes launch from the autograd engine's thread
void hwg_note_plan(int engine, int cfg, int nsplit) { g_last_plan[0] = engine; g_last_plan[1] = cfg; g_last_plan[2] = nsplit; }
extern "C" int hwg_last_plan(int* engine_cfg_nsplit) {
  HWG_REQUIRE(engine_cfg_nsplit, "last_plan: null pointer");
  for (int i = 0; i < 3; ++i) engine_cfg_nsplit[i] = g_last_plan[i];
  return HWG_OK;
}
extern "C" int hwg_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Launch profiler: HIP event pairs recorded immediately around the matrix-core kernel launches, on the stream they are launched
// on (bench.py's roofline measurement). Recording them here instead of from Python keeps host time between "event" and "launch"
// out of the measured duration. Off by default; never synchronises except in hwg_prof_stop().
#include <atomic>
namespace {
struct ProfRec { hipEvent_t e0, e1; int kind, tag, launched; double work; int parent; };   // parent >= 0: a share of that record's time
ProfRec* g_prof = nullptr;
int g_prof_cap = 0;
std::atomic<int> g_prof_n{0};
std::atomic<int> g_prof_on{0};
thread_local int g_prof_tag = -1;
}  // namespace

namespace {
thread_local int g_prof_cur = -1;       // record opened by this thread and not closed yet
thread_local int g_prof_cur_launches = 0;
}
int hwg_prof_open(int kind, double work, hipStream_t st) {
  (void)st;
  if (!g_prof_on.load(std::memory_order_relaxed)) return -1;
  const int i = g_prof_n.fetch_add(1);
  if (i >= g_prof_cap) return -1;
  g_prof[i].kind = kind; g_prof[i].tag = g_prof_tag; g_prof[i].work = work; g_prof[i].launched = 0; g_prof[i].parent = -1;
  g_prof_cur = i; g_prof_cur_launches = 0;
  return i;
}
void hwg_prof_close(int i, hipStream_t st) {
  (void)st;
  if (i >= 0 && g_prof_cur == i) { g_prof[i].launched = g_prof_cur_launches; g_prof_cur = -1; }
}
int hwg_prof_current_tag() { return g_prof_tag; }
// a launch that serves several layers (the table-driven weight-gradient reduce): its time is reported as one record per layer, the
// parent's duration shared out in proportion to `work` (the parent itself is not reported)
void hwg_prof_add_child(int parent, int kind, int tag, double work) {
  if (parent < 0 || !g_prof_on.load(std::memory_order_relaxed)) return;
  const int i = g_prof_n.fetch_add(1);
  if (i >= g_prof_cap) return;
  g_prof[i].kind = kind; g_prof[i].tag = tag; g_prof[i].work = work; g_prof[i].launched = 0; g_prof[i].parent = parent;
}
HwgProfEv hwg_prof_launch_events() {
  HwgProfEv ev = {nullptr, nullptr};
  if (g_prof_cur < 0 || !g_prof) return ev;
  if (g_prof_cur_launches++ == 0) ev.e0 = g_prof[g_prof_cur].e0;
  ev.e1 = g_prof[g_prof_cur].e1;
  return ev;
}

extern "C" int hwg_prof_start(int max_records) {
  HWG_REQUIRE(max_records > 0 && !g_prof_on.load(), "prof_start: bad capacity or already running");
  g_prof = new ProfRec[max_records];
  for (int i = 0; i < max_records; ++i) {
    if (hipEventCreate(&g_prof[i].e0) != hipSuccess || hipEventCreate(&g_prof[i].e1) != hipSuccess) {
      hwg_set_error("prof_start: hipEventCreate failed");
      return HWG_ERR_LAUNCH;
    }
  }
  g_prof_cap = max_records;
  g_prof_n.store(0);
  g_prof_on.store(1);
  return HWG_OK;
}
extern "C" int hwg_prof_enable(int on) {
  if (!g_prof) return HWG_OK;   // no profile open: nothing to switch
  g_prof_on.store(on ? 1 : 0);
  return HWG_OK;
}
extern "C" int hwg_prof_tag(int tag) { g_prof_tag = tag; return HWG_OK; }
extern "C" int hwg_prof_stop(int* kinds, int* tags, double* work, float* ms, int capacity) {
  if (!g_prof) return 0;
  g_prof_on.store(0);
  int n = g_prof_n.load();
  if (n > g_prof_cap) n = g_prof_cap;
  int out = 0;
  double* child_work = new double[n > 0 ? n : 1]();     // per parent: total work of its children
  for (int i = 0; i < n; ++i)
    if (g_prof[i].parent >= 0 && g_prof[i].parent < n) child_work[g_prof[i].parent] += g_prof[i].work;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    const int src = g_prof[i].parent >= 0 ? g_prof[i].parent : i;
    if (src >= n || g_prof[src].launched <= 0) continue;      // bracket opened, nothing launched (an error path)
    if (g_prof[i].parent < 0 && child_work[i] > 0.0) continue; // reported through its children
    (void)hipEventSynchronize(g_prof[src].e1);
    if (hipEventElapsedTime(&t, g_prof[src].e0, g_prof[src].e1) != hipSuccess) continue;
    if (g_prof[i].parent >= 0) t = (float)(t * g_prof[i].work / child_work[src]);
    if (out < capacity && kinds && tags && work && ms) {
      kinds[out] = g_prof[i].kind; tags[out] = g_prof[i].tag; work[out] = g_prof[i].work; ms[out] = t;
    }
    ++out;
  }
  delete[] child_work;
  for (int i = 0; i < g_prof_cap; ++i) { (void)hipEventDestroy(g_prof[i].e0); (void)hipEventDestroy(g_prof[i].e1); }
  delete[] g_prof;
  g_prof = nullptr; g_prof_cap = 0; g_prof_n.store(0);
  return out < capacity ? out : capacity;
}
