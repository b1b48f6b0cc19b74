// corun_probe.hip -- dev experiment (not part of libumx.so): which instruction class of a wave-per-row kernel is disturbed when other
// PROCESSES use the GPU at the same time?  Four tiny kernels on fixed inputs, each launched `iters` times and compared with its own
// first result:  copy (loads + stores only) | fma (per-lane arithmetic, no cross-lane traffic) | dpp (wave sums by DPP + readlane) |
// bperm (wave sums by ds_bpermute) | divs (IEEE division + sqrt sequences).
// usage: corun_probe [iters] [rows]      (start it beside build/overlap_bench verify ... and a second copy of itself)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "umx_common.h"
using namespace umx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int W = 1152;

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, long nt) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nt) return;
  float2 v[9], w[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    v[r] = *reinterpret_cast<const float2*>(x + row * W + r * 128 + 2 * lane);
    w[r] = *reinterpret_cast<const float2*>(g + row * W + r * 128 + 2 * lane);
  }
  float s = 1.0f;
  if (MODE == 1) {            // per-lane arithmetic only
#pragma unroll
    for (int r = 0; r < 9; ++r) s = fmaf(v[r].x, w[r].x, fmaf(v[r].y, w[r].y, s));
  } else if (MODE == 2 || MODE == 3) {   // wave sums
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) { a += v[r].x * v[r].x + v[r].y * v[r].y; b += v[r].x * w[r].x + v[r].y * w[r].y; }
    a = MODE == 2 ? wave_sum_dpp(a) : wave_sum_shfl(a);
    b = MODE == 2 ? wave_sum_dpp(b) : wave_sum_shfl(b);
    s = a + b;
  } else if (MODE == 4) {     // division / sqrt sequences, per lane
    float a = 1.0f;
#pragma unroll
    for (int r = 0; r < 9; ++r) a += v[r].x * v[r].x / 3.0f + v[r].y * v[r].y / 9.0f;
    s = 1.0f / sqrtf(a);
  }
#pragma unroll
  for (int r = 0; r < 9; ++r) *reinterpret_cast<float2*>(out + row * W + r * 128 + 2 * lane) = make_float2(w[r].x * s + v[r].x, w[r].y * s + v[r].y);
}

__global__ void k_diff(const unsigned int* a, const unsigned int* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10000;
  const long nn = argc > 2 ? atol(argv[2]) : 20000;
  const size_t n = (size_t)nn * W;
  std::vector<float> h(n);
  float *x, *g, *out, *ref; unsigned long long* cnt;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&cnt, 8));
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (rand() % 20001) / 10000.0f - 1.0f;
  CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < n; ++i) h[i] = 1e-3f * ((rand() % 20001) / 10000.0f - 1.0f);
  CK(hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const dim3 grid((unsigned)((nn + 3) / 4)), block(256);
  const char* names[5] = {"copy ", "fma  ", "dpp  ", "bperm", "divs "};
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 5; ++mode) {
      auto launch = [&](float* dst) {
        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, grid, block, 0, s, x, g, dst, nn);
        else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, grid, block, 0, s, x, g, dst, nn);
        else if (mode == 2) hipLaunchKernelGGL(k_probe<2>, grid, block, 0, s, x, g, dst, nn);
        else if (mode == 3) hipLaunchKernelGGL(k_probe<3>, grid, block, 0, s, x, g, dst, nn);
        else hipLaunchKernelGGL(k_probe<4>, grid, block, 0, s, x, g, dst, nn);
      };
      launch(ref); CK(hipStreamSynchronize(s));
      int bad = 0; unsigned long long words = 0;
      for (int it = 0; it < iters; ++it) {
        CK(hipMemsetAsync(cnt, 0, 8, s));
        launch(out);
        hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, s, reinterpret_cast<const unsigned int*>(out), reinterpret_cast<const unsigned int*>(ref), n, cnt);
        unsigned long long c = 0;
        CK(hipMemcpyAsync(&c, cnt, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (c) { ++bad; words += c; }
      }
      printf("%s: %d of %d launches differ from the first (%llu dwords in all)\n", names[mode], bad, iters, words);
      fflush(stdout);
    }
  return 0;
}
