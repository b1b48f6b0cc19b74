// norm_bwd_repro.hip -- dev experiment (not part of libumx.so): is k_norm_bwd bitwise reproducible while another process uses the GPU?
// usage: norm_bwd_repro [iterations] [rows]     (start two of them at the same time)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "umx_kernels_pl.h"
#include "umx_kernels.h"
using namespace umx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_diff(const unsigned int* a, const unsigned int* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const long nn = argc > 2 ? atol(argv[2]) : 780;
  const size_t n = (size_t)nn * ROW;
  std::vector<float> h(n);
  float *x, *gy, *gres, *gx, *ref, *aw; unsigned long long* cnt;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&gy, n * 4)); CK(hipMalloc(&gres, n * 4)); CK(hipMalloc(&gx, n * 4)); CK(hipMalloc(&ref, n * 4));
  CK(hipMalloc(&aw, 3 * C * 4)); CK(hipMalloc(&cnt, 8));
  srand(1);
  auto fill = [&](float* d, size_t m, float sc) { for (size_t i = 0; i < m; ++i) h[i] = sc * ((rand() % 20001) / 10000.0f - 1.0f); CK(hipMemcpy(d, h.data(), m * 4, hipMemcpyHostToDevice)); };
  fill(x, n, 1.0f); fill(gy, n, 1e-3f); fill(gres, n, 1e-3f); fill(aw, 3 * C, 1.0f);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_norm_bwd, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, s, gy, x, aw, gres, ref, nn);
  CK(hipStreamSynchronize(s));
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    CK(hipMemsetAsync(cnt, 0, 8, s));
    hipLaunchKernelGGL(k_norm_bwd, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, s, gy, x, aw, gres, gx, nn);
    hipLaunchKernelGGL(k_diff, dim3(64), dim3(256), 0, s, reinterpret_cast<const unsigned int*>(gx), reinterpret_cast<const unsigned int*>(ref), n, cnt);
    unsigned long long c = 0;
    CK(hipMemcpyAsync(&c, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (c) {
      if (++bad <= 6) {
        printf("iteration %d: %llu dwords differ\n", it, c);
        std::vector<float> a(n), b(n);
        CK(hipMemcpy(a.data(), gx, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, n * 4, hipMemcpyDeviceToHost));
        int shown = 0;
        for (long r = 0; r < nn && shown < 3; ++r) {
          long nd = 0; for (int j = 0; j < ROW; ++j) nd += a[r * ROW + j] != b[r * ROW + j];
          if (!nd) continue;
          ++shown;
          printf("  row %ld (workgroup %ld, wave %ld): %ld of %d values differ;", r, r / 4, r % 4, nd, ROW);
          for (int j : {0, 1, 128, 129, 640, 1151}) printf("  [%d] %.9g vs %.9g", j, a[r * ROW + j], b[r * ROW + j]);
          printf("\n");
        }
      }
    }
  }
  printf("k_norm_bwd: %d of %d launches differ from the first\n", bad, iters);
  fflush(stdout);
  return 0;
}
