// error plumbing + version of libcruller_hip.so
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/crl.h"

static thread_local char g_err[512] = "";

extern "C" void crl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* crl_last_error(void) { return g_err; }

// declared in common.h: a kernel's dynamic-LDS limit is a per-device attribute -- set once per (kernel, device), thread-safe
#include <mutex>
#include <utility>
int crl_enable_lds(const void* kernel, int bytes, const char* who) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { crl_set_error("%s: no current device", who); return -2; }
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& d : done)
    if (d.first == kernel && d.second == dev) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { crl_set_error("%s: cannot enable %d bytes of LDS: %s", who, bytes, hipGetErrorString(e)); return -2; }
  done.emplace_back(kernel, dev);
  return 0;
}
extern "C" int crl_version(void) { return 1; }

// ---- live per-kernel timing (include/crl.h: crl_prof_begin / crl_prof_end) -------------------------------------------
// HIP events recorded on the launch stream right before and after every launch of the instrumented kernels, from a pool
// created up front; nothing is synchronised until crl_prof_end.
struct ProfRec { int id; double work; hipEvent_t e0, e1; };
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
int g_crl_prof_on = 0;

extern "C" void crl_prof_mark(int id, int phase, void* stream, double work) {
  if (!g_crl_prof_on) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return;   // events recorded into a graph cannot be timed
  if (phase == 0) {
    if (g_prof_used >= g_prof.size()) return;          // pool exhausted: later launches go untimed
    ProfRec& r = g_prof[g_prof_used];
    r.id = id; r.work = work;
    (void)hipEventRecord(r.e0, (hipStream_t)stream);
  } else {
    if (g_prof_used >= g_prof.size() || g_prof[g_prof_used].id != id) return;
    (void)hipEventRecord(g_prof[g_prof_used].e1, (hipStream_t)stream);
    ++g_prof_used;
  }
}

extern "C" int crl_prof_begin(int capacity) {
  if (g_crl_prof_on) { crl_set_error("crl_prof_begin: already profiling"); return -1; }
  if (capacity < 1 || capacity > (1 << 20)) { crl_set_error("crl_prof_begin: bad capacity %d", capacity); return -1; }
  g_prof.assign((size_t)capacity, ProfRec{0, 0.0, nullptr, nullptr});
  for (auto& r : g_prof)
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { crl_set_error("crl_prof_begin: hipEventCreate failed"); return -2; }
  g_prof_used = 0;
  g_crl_prof_on = 1;
  return 0;
}

extern "C" int crl_prof_end(int n_ids, int* launches, double* ms, double* work) {
  if (!g_crl_prof_on) { crl_set_error("crl_prof_end: not profiling"); return -1; }
  g_crl_prof_on = 0;
  for (int i = 0; i < n_ids; ++i) { launches[i] = 0; ms[i] = 0.0; work[i] = 0.0; }
  int rc = 0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    ProfRec& r = g_prof[i];
    float t = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { rc = -2; continue; }
    if (r.id >= 0 && r.id < n_ids) { launches[r.id] += 1; ms[r.id] += t; work[r.id] += r.work; }
  }
  for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  if (rc) crl_set_error("crl_prof_end: an event could not be read");
  return rc;
}
