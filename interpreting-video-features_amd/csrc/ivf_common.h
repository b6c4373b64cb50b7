// Shared host-side helpers for libivf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/ivf_hip.h"

namespace ivf {

void set_error(const char* fmt, ...);

// launch profiler (ivf_common.hip)
void prof_set_iteration(int it);
void prof_set_flops(double algorithmic_flops);
void prof_set_site(int site);   // launch site of the next sampled launches (2 * op index + direction), -1 none
bool prof_begin(hipStream_t s, int variant);
void prof_end(hipStream_t s);
void prof_name(int variant, const char* fmt, ...);   // kernel name of a profiler class (first call wins)

#define IVF_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::ivf::set_error(__VA_ARGS__);             \
      return IVF_ERR_BAD_ARG;                    \
    }                                            \
  } while (0)

#define IVF_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      ::ivf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                       __FILE__, __LINE__);                                        \
      return IVF_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define IVF_CHECK_LAUNCH() IVF_CHECK_HIP(hipGetLastError())

#define IVF_PROPAGATE(expr)      \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != IVF_OK) return rc_; \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `done` is the call site's static flag array
struct LdsAttrOnce { bool done[32] = {}; };
static inline int raise_lds_limit(const void* kernel, int bytes, LdsAttrOnce& once) {
  int dev = 0;
  IVF_CHECK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 32 && once.done[dev]) return IVF_OK;
  IVF_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev >= 0 && dev < 32) once.done[dev] = true;
  return IVF_OK;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// TF-'same' front padding along one dim (reference I3D_doubled.py:9-13, 29-34).
static inline void same_pad(int n, int k, int s, int* front, int* back) {
  int p = (n % s == 0) ? (k - s) : (k - (n % s));
  if (p < 0) p = 0;
  *front = p / 2;
  *back = p - p / 2;
}
static inline int same_out(int n, int k, int s) {
  int f, b;
  same_pad(n, k, s, &f, &b);
  return (n + f + b - k) / s + 1;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): hand every XCD a contiguous range of
// tiles, so that tiles sharing input rows share an L2 (bijective on [0, nwg)).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  // Blocks are dealt round-robin over 8 XCDs; give each XCD a contiguous range of
  // tiles so blocks sharing an A panel share an L2 (bijective form).
  int q = nwg >> 3, r = nwg & 7;
  int xcd = id & 7, pos = id >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}

}  // namespace ivf
