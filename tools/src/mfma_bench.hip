// Bare MFMA loop rates on MI355X (dev tool): 32x32x16 vs 16x16x32 bf16, dependent chains of 3
// per accumulator as in the split-bf16 kernels.  hipcc -O3 --offload-arch=gfx950 mfma_bench.hip -o mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TILES>
__global__ void k32(float* out, int iters) {
  f32x16 acc[TILES];
  for (int t = 0; t < TILES; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < TILES; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TILES>
__global__ void k16(float* out, int iters) {
  f32x4 acc[TILES];
  for (int t = 0; t < TILES; ++t)
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < TILES; ++t)
    for (int r = 0; r < 4; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, int threads, double flop_per_mfma, int tiles, float* out) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * (threads / 64) * iters * tiles * 3 * flop_per_mfma;
  printf("%-28s %4d threads/WG (%d waves/SIMD)  %7.1f TFLOP/s  (%.2f ms)\n", name, threads, threads / 256, fl / ms / 1e9, ms);
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run("32x32x16, 3 tiles/wave", k32<3>, 1024, 2.0 * 32 * 32 * 16, 3, out);
  run("32x32x16, 3 tiles/wave", k32<3>, 512, 2.0 * 32 * 32 * 16, 3, out);
  run("32x32x16, 2 tiles/wave", k32<2>, 512, 2.0 * 32 * 32 * 16, 2, out);
  run("32x32x16, 3 tiles/wave", k32<3>, 256, 2.0 * 32 * 32 * 16, 3, out);
  run("16x16x32, 12 tiles/wave", k16<12>, 1024, 2.0 * 16 * 16 * 32, 12, out);
  run("16x16x32, 12 tiles/wave", k16<12>, 512, 2.0 * 16 * 16 * 32, 12, out);
  run("16x16x32, 8 tiles/wave", k16<8>, 512, 2.0 * 16 * 16 * 32, 8, out);
  run("16x16x32, 12 tiles/wave", k16<12>, 256, 2.0 * 16 * 16 * 32, 12, out);
  return 0;
}
