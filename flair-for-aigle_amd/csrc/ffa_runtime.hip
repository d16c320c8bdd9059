// Error reporting and library identification for libflairhip.
// Contract (include/flairhip.h): every entry point returns 0 on success, a negative FFA_ERR_* code
// for argument / support errors, or a positive hipError_t for launch failures; nothing throws across
// the C boundary; the text of the last failure on this thread is available from ffa_last_error().
#include "ffa_common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_last_error[512] = "";

void ffa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

int ffa_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ffa_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return FFA_OK;
}

extern "C" const char* ffa_last_error(void) { return g_last_error; }

extern "C" int ffa_version(void) { return 100; }

extern "C" const char* ffa_target_arch(void) { return "gfx950"; }

// ---- kernel timing session (measurement only; bench.py's roofline object) ----------------------------------------
// Between ffa_ktime_begin(n) and ffa_ktime_end() the instrumented launches of the process (the 3x3 MFMA families: ring16,
// wgrad64) go through hipExtLaunchKernelGGL with a start / stop event pair of their own: the events are written by the
// command processor around the kernel itself, so the elapsed time is the kernel's duration as rocprofv3 reports it --
// a hipEventRecord pair around the launch call also brackets the dispatch gap (~4-5 us per launch, 10 % of a 43 us
// kernel).  Not for use inside a stream capture.  Outside a session nothing changes (plain hipLaunchKernelGGL).
#include <atomic>
#include <vector>
namespace {
struct KtimeSlot {
  hipEvent_t start, stop;
  int tag;
};
// process-wide, not per thread: autograd runs the backward launches (dgrad, wgrad) on its own device thread
std::vector<KtimeSlot> g_kt;
std::atomic<int> g_kt_used{-1};  // -1: no session
}  // namespace

bool ffa_ktime_next(int tag, hipEvent_t* start, hipEvent_t* stop) {
  if (g_kt_used.load(std::memory_order_relaxed) < 0) return false;
  const int i = g_kt_used.fetch_add(1);
  if (i < 0 || i >= (int)g_kt.size()) return false;  // (a launch racing with ffa_ktime_end simply goes untimed)
  KtimeSlot& s = g_kt[i];
  s.tag = tag;
  *start = s.start;
  *stop = s.stop;
  return true;
}

extern "C" int ffa_ktime_begin(int max_launches) {
  FFA_REQUIRE(max_launches > 0 && max_launches <= (1 << 16), "ktime_begin: 1 .. 65536 launches");
  FFA_REQUIRE(g_kt_used.load() < 0, "ktime_begin: a session is already open");
  g_kt.resize(max_launches);
  for (int i = 0; i < max_launches; ++i) {
    hipError_t e = hipEventCreate(&g_kt[i].start);
    if (e == hipSuccess) e = hipEventCreate(&g_kt[i].stop);
    if (e != hipSuccess) {
      ffa_set_error("ktime_begin: hipEventCreate: %s", hipGetErrorString(e));
      return (int)e;
    }
    g_kt[i].tag = 0;
  }
  g_kt_used.store(0);
  return FFA_OK;
}

// waits for the timed launches, writes up to `cap` (milliseconds, tag) pairs in launch order, closes the session;
// returns the number of timed launches (or a negative error)
extern "C" int ffa_ktime_end(float* ms, int* tags, int cap) {
  FFA_REQUIRE(g_kt_used.load() >= 0, "ktime_end: no session");
  int n = g_kt_used.exchange(-1);  // closes the session: later launches are plain again
  if (n > (int)g_kt.size()) n = (int)g_kt.size();
  int rc = n;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(g_kt[i].stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, g_kt[i].start, g_kt[i].stop);
    if (e != hipSuccess && rc >= 0) {
      ffa_set_error("ktime_end: %s", hipGetErrorString(e));
      rc = -(int)e;
    }
    if (i < cap && ms && tags) {
      ms[i] = t;
      tags[i] = g_kt[i].tag;
    }
  }
  for (auto& s : g_kt) {
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  g_kt.clear();
  return rc;
}
