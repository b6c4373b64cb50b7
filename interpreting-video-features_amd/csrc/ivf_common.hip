// Error reporting and version for libivf_hip.so.
#include <cstring>

#include "ivf_common.h"

namespace ivf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ivf

extern "C" const char* ivf_last_error(void) { return ivf::g_err; }
extern "C" int ivf_version(void) { return 301; }   // (round 3: variant table of 70 LDS-halo tiles, 4 arithmetic modes)

// ---------------------------------------------------------------- launch profiler
// Optional HIP-event timing of the implicit-GEMM convolution launches, per tile
// variant, on the stream they are launched on.  Sampled every `every`-th search
// iteration so the event records do not perturb the timed region.
#include <vector>
namespace ivf {
struct Prof {
  bool enabled = false, active = false;
  int every = 1;
  size_t cap = 0, used = 0;
  std::vector<hipEvent_t> ev;          // 2 per sampled launch
  std::vector<int> variant;
  std::vector<int> site;
  std::vector<double> flops;
  double next_flops = 0.0;
  int next_site = -1;
};
static Prof g_prof;
static char g_prof_names[IVF_PROFILE_CLASSES][96];
void prof_name(int variant, const char* fmt, ...) {
  if (variant < 0 || variant >= IVF_PROFILE_CLASSES) return;   // (last call wins: a class id is reused across arithmetic modes)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_prof_names[variant], sizeof(g_prof_names[variant]), fmt, ap);
  va_end(ap);
}
void prof_set_iteration(int it) { g_prof.active = g_prof.enabled && it >= 0 && (it % g_prof.every == 0); }
void prof_set_flops(double f) { g_prof.next_flops = f; }
void prof_set_site(int site) { g_prof.next_site = site; }
bool prof_begin(hipStream_t s, int variant) {
  Prof& p = g_prof;
  if (!p.active || p.used >= p.cap) return false;
  if (p.ev.size() < 2 * (p.used + 1)) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    p.ev.push_back(a);
    p.ev.push_back(b);
  }
  p.variant.resize(p.used + 1);
  p.site.resize(p.used + 1);
  p.flops.resize(p.used + 1);
  p.variant[p.used] = variant;
  p.site[p.used] = p.next_site;
  p.flops[p.used] = p.next_flops;
  (void)hipEventRecord(p.ev[2 * p.used], s);
  return true;
}
void prof_end(hipStream_t s) {
  Prof& p = g_prof;
  (void)hipEventRecord(p.ev[2 * p.used + 1], s);
  p.used++;
}
}  // namespace ivf

extern "C" int ivf_profile_enable(int every, int max_launches) {
  IVF_CHECK_ARG(every >= 1 && max_launches > 0, "profile_enable: bad args");
  ivf::g_prof.enabled = true;
  ivf::g_prof.active = false;
  ivf::g_prof.every = every;
  ivf::g_prof.cap = (size_t)max_launches;
  ivf::g_prof.used = 0;
  return IVF_OK;
}

extern "C" int ivf_profile_disable(void) {
  ivf::g_prof.enabled = false;
  ivf::g_prof.active = false;
  return IVF_OK;
}

// Kernel (template instance) behind a profiler class id; "" until that class has launched.
extern "C" const char* ivf_profile_class_name(int cls) {
  return (cls >= 0 && cls < IVF_PROFILE_CLASSES) ? ivf::g_prof_names[cls] : "";
}

// The same sample summed per launch SITE (a convolution of the plan in one direction: site = 2 * op index
// + direction, see ivf_i3d_site_name): kernel_ms[s], launches[s], flops[s], variant[s] for s < max_sites.
// Does not reset the sample; call before ivf_profile_collect.
extern "C" int ivf_profile_collect_sites(double* kernel_ms, long long* launches, double* flops, int* variant,
                                         int max_sites) {
  IVF_CHECK_ARG(kernel_ms && launches && flops && variant && max_sites > 0, "profile_collect_sites: bad args");
  ivf::Prof& p = ivf::g_prof;
  for (int v = 0; v < max_sites; ++v) { kernel_ms[v] = 0.0; launches[v] = 0; flops[v] = 0.0; variant[v] = -1; }
  for (size_t i = 0; i < p.used; ++i) {
    const int st = p.site[i];
    if (st < 0 || st >= max_sites) continue;
    IVF_CHECK_HIP(hipEventSynchronize(p.ev[2 * i + 1]));
    float ms = 0.f;
    IVF_CHECK_HIP(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
    kernel_ms[st] += ms;
    launches[st] += 1;
    flops[st] += p.flops[i];
    variant[st] = p.variant[i];
  }
  return IVF_OK;
}

// Sums per kernel variant id v in [0,IVF_PROFILE_CLASSES): kernel_ms[v], launches[v], flops[v];
// resets the sample.  Synchronises on the recorded events (call outside the timed region).
extern "C" int ivf_profile_collect(double* kernel_ms, long long* launches, double* flops) {
  IVF_CHECK_ARG(kernel_ms && launches && flops, "profile_collect: null pointer");
  ivf::Prof& p = ivf::g_prof;
  for (int v = 0; v < IVF_PROFILE_CLASSES; ++v) { kernel_ms[v] = 0.0; launches[v] = 0; flops[v] = 0.0; }
  for (size_t i = 0; i < p.used; ++i) {
    IVF_CHECK_HIP(hipEventSynchronize(p.ev[2 * i + 1]));
    float ms = 0.f;
    IVF_CHECK_HIP(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
    int v = p.variant[i];
    if (v < 0 || v >= IVF_PROFILE_CLASSES) continue;
    kernel_ms[v] += ms;
    launches[v] += 1;
    flops[v] += p.flops[i];
  }
  p.used = 0;
  return IVF_OK;
}
