// overlap_bench.hip -- dev experiment (not part of libumx.so): can an HBM-bound streaming kernel run BESIDE the LDS-filling
// split-bf16 GEMM on the same CUs?  Times the Q3 256x256 forward GEMM alone, a copy kernel alone (classic one-item-per-thread
// grid, and persistent low-footprint grids of G blocks), and both launched together on two streams.
// usage: overlap_bench [M] [N] [K] [copy_GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include "umx_gemm_pl.h"
#include "umx_gemm_q.h"
using namespace umx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy_full(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// persistent: G blocks, each thread keeps U 16-B loads in flight
template <int U>
__global__ __launch_bounds__(256) void k_copy_pers(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256 * U;
  for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = base + (long)u * 256; v[u] = i < n ? src[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = base + (long)u * 256; if (i < n) dst[i] = v[u]; }
  }
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// "pmc" mode: one launch of every large split-bf16 GEMM of a c3 layer at the real shapes (8-image chunk = 1.14 M edges), for
// rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, LDS and wait counters); operands are random finite planes (bit patterns 0x3c00-0x3fff with random sign: 1-2 as fp16, 0.0078-0.031 as bf16).
static int pmc_mode(int reps, bool x3 = false) {
  const long M = 1139068;
  const long M4 = (M + 3) / 4 * 4;
  unsigned char *y1, *w; float* C;
  CK(hipMalloc(&y1, (size_t)M4 * 2304 * 6)); CK(hipMalloc(&w, (size_t)1536 * 768 * 6)); CK(hipMalloc(&C, (size_t)M * 2304 * 4));
  std::vector<unsigned short> h(1 << 22);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (size_t o = 0; o < (size_t)M4 * 2304 * 6; o += h.size() * 2) CK(hipMemcpy(y1 + o, h.data(), std::min(h.size() * 2, (size_t)M4 * 2304 * 6 - o), hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data() + 17, (size_t)1536 * 768 * 6, hipMemcpyHostToDevice));
  auto mk = [&](int P, int a_cols, int offA0, int offA1, int bHalf, long ldc, int offC, int offCi, int N, int K) {
    GemmPL q; std::memset(&q, 0, sizeof(q));
    q.Apl = reinterpret_cast<const unsigned short*>(y1); q.lda = (long)a_cols * P; q.offA0 = offA0; q.offA1 = offA1;
    q.Bpl = reinterpret_cast<const unsigned short*>(w); q.ldb = (long)K * P; q.bHalf = bHalf; q.Cp = C; q.ldc = ldc; q.offC = offC; q.offCi = offCi;
    q.conj = 1.f; q.cscale = 1.f; q.M = (int)M; q.N = N; q.K = K;
    return q;
  };
  auto grid = [&](int cplx, int wide, int N) {
    const int bmr = cplx ? 128 : 256, bnc = wide ? (cplx ? 128 : 256) : (cplx ? 64 : 128);
    const long nM = (M + bmr - 1) / bmr, nN = (N + bnc - 1) / bnc;
    return dim3((unsigned)(((nM + 7) / 8) * 8 * nN));
  };
  if (x3) {
    // the DEFAULT mode since round 4 (bf16x3): 6 plane products in both passes; forward and conv^T A operands as float32 quad-row blocks split in
    // registers (AF = 1; the buffer's bit patterns read as finite floats ~0.01-0.03), weights as three bf16 planes; fc3^T on PL planes
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 2, 0, 6, 3, 1, 2, 1>), grid(0, 0, 640), dim3(512), 0, 0, mk(3, 2304, 0, 0, 0, 1408, 0, 0, 640, 768));      // forward plain products: the LS + aligned-planes forms the engine launches (umx_gemm_q.h)
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 256), dim3(512), 0, 0, mk(3, 2304, 768, 1280, 256, 1408, 640, 896, 256, 512));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 128), dim3(512), 0, 0, mk(3, 2304, 1792, 2048, 128, 1408, 1152, 1280, 128, 256));
      hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 2, 0, 6, 3, 1, 2, 1>), grid(0, 0, 384), dim3(512), 0, 0, mk(3, 1152, 0, 0, 0, 1152, 0, 0, 384, 384));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 256), dim3(512), 0, 0, mk(3, 1152, 384, 640, 256, 1152, 384, 640, 256, 256));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 128), dim3(512), 0, 0, mk(3, 1152, 896, 1024, 128, 1152, 896, 1024, 128, 128));
      hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 3, 2, 0, 6, 3, 1, 1, 1>), grid(0, 1, 1536), dim3(512), 0, 0, mk(3, 128, 0, 0, 0, 1536, 0, 0, 1536, 128));
      hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 2, 0, 6, 3, 1>), grid(0, 0, 384), dim3(512), 0, 0, mk(3, 1152, 0, 0, 0, 1152, 0, 0, 384, 384));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 256), dim3(512), 0, 0, mk(3, 1152, 384, 640, 256, 1152, 384, 640, 256, 256));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 128), dim3(512), 0, 0, mk(3, 1152, 896, 1024, 128, 1152, 896, 1024, 128, 128));
      hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 3, 2, 0, 6, 3, 1>), grid(0, 1, 768), dim3(512), 0, 0, mk(3, 1408, 0, 0, 0, 2304, 0, 0, 768, 640));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 512), dim3(512), 0, 0, mk(3, 1408, 640, 896, 512, 2304, 768, 1280, 512, 256));
      hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 3, 2, 0, 6, 3, 1>), grid(1, 1, 256), dim3(512), 0, 0, mk(3, 1408, 1152, 1280, 256, 2304, 1792, 2048, 256, 128));
      { GemmPL qg = mk(3, 1536, 0, 0, 0, 128, 0, 0, 128, 1536); qg.lda = 2L * 1536; hipLaunchKernelGGL((umx_gemm_pl16_kernel<0, 3, 2, 4, 2, 2, 2, 0, 1>), grid(0, 0, 128), dim3(512), 0, 0, qg); }   // g_rad as float32 rows
    }
    CK(hipDeviceSynchronize());
    printf("pmc3 mode: %d x 14 GEMM launches done\n", reps);
    return 0;
  }
  for (int r = 0; r < reps; ++r) {
    // forward (quad-row layout, fp16: 2 activation planes x 3 weight planes, 4 products): conv-1 m0 / m1 / m2, conv-2 m0 / m1 / m2, radial fc3
    hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 2, 2, 1, 4, 3>), grid(0, 0, 640), dim3(512), 0, 0, mk(2, 2304, 0, 0, 0, 1408, 0, 0, 640, 768));
    hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 2, 2, 1, 4, 3>), grid(1, 1, 256), dim3(512), 0, 0, mk(2, 2304, 768, 1280, 256, 1408, 640, 896, 256, 512));
    hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 2, 2, 1, 4, 3>), grid(1, 1, 128), dim3(512), 0, 0, mk(2, 2304, 1792, 2048, 128, 1408, 1152, 1280, 128, 256));
    hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 2, 2, 1, 4, 3>), grid(0, 0, 384), dim3(512), 0, 0, mk(2, 1152, 0, 0, 0, 1152, 0, 0, 384, 384));
    hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 2, 2, 1, 4, 3>), grid(1, 1, 256), dim3(512), 0, 0, mk(2, 1152, 384, 640, 256, 1152, 384, 640, 256, 256));
    hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 2, 2, 1, 4, 3>), grid(1, 1, 128), dim3(512), 0, 0, mk(2, 1152, 896, 1024, 128, 1152, 896, 1024, 128, 128));
    hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 3>), grid(0, 1, 1536), dim3(512), 0, 0, mk(2, 128, 0, 0, 0, 1536, 0, 0, 1536, 128));
    // reverse (PL, P=2): conv-2^T m0 / m1 / m2, conv-1^T m0 / m1 / m2, fc3^T
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<0, 2, 3, 4, 2, 2, 2>), grid(0, 0, 384), dim3(512), 0, 0, mk(2, 1152, 0, 0, 0, 1152, 0, 0, 384, 384));
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 256), dim3(512), 0, 0, mk(2, 1152, 384, 640, 256, 1152, 384, 640, 256, 256));
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 128), dim3(512), 0, 0, mk(2, 1152, 896, 1024, 128, 1152, 896, 1024, 128, 128));
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<0, 2, 2, 4, 2, 2, 4>), grid(0, 1, 768), dim3(512), 0, 0, mk(2, 1408, 0, 0, 0, 2304, 0, 0, 768, 640));
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 512), dim3(512), 0, 0, mk(2, 1408, 640, 896, 512, 2304, 768, 1280, 512, 256));
    hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 256), dim3(512), 0, 0, mk(2, 1408, 1152, 1280, 256, 2304, 1792, 2048, 256, 128));
    hipLaunchKernelGGL((umx_gemm_pl_kernel<0, 2, 3, 4, 2, 2, 2>), grid(0, 0, 128), dim3(512), 0, 0, mk(2, 1536, 0, 0, 0, 128, 0, 0, 128, 1536));
  }
  CK(hipDeviceSynchronize());
  printf("pmc mode: %d x 14 GEMM launches done\n", reps);
  return 0;
}

__global__ void k_count_diff(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b, size_t n, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

// verify mode: is every split GEMM bitwise reproducible when another kernel shares the chip?  Each of the 14 launches of an evaluation is
// run alone (reference C), then `iters` times beside a full-grid copy kernel on a second stream, and C is compared dword by dword.
static int verify_mode(long M, int iters) {
  const long M4 = (M + 3) / 4 * 4;
  unsigned char *y1, *w; float *C, *Cref;
  const size_t cbytes = (size_t)M * 2304 * 4;
  CK(hipMalloc(&y1, (size_t)M4 * 2304 * 6)); CK(hipMalloc(&w, (size_t)1536 * 768 * 6)); CK(hipMalloc(&C, cbytes)); CK(hipMalloc(&Cref, cbytes));
  std::vector<unsigned short> h(1 << 22);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (size_t o = 0; o < (size_t)M4 * 2304 * 6; o += h.size() * 2) CK(hipMemcpy(y1 + o, h.data(), std::min(h.size() * 2, (size_t)M4 * 2304 * 6 - o), hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data() + 17, (size_t)1536 * 768 * 6, hipMemcpyHostToDevice));
  const long n4 = (long)(2.0e9 / 16);
  float4 *src, *dst; unsigned long long* d_cnt;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 1, n4 * 16)); CK(hipMalloc(&d_cnt, 8));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  auto mk = [&](int P, int a_cols, int offA0, int offA1, int bHalf, long ldc, int offC, int offCi, int N, int K) {
    GemmPL q; std::memset(&q, 0, sizeof(q));
    q.Apl = reinterpret_cast<const unsigned short*>(y1); q.lda = (long)a_cols * P; q.offA0 = offA0; q.offA1 = offA1;
    q.Bpl = reinterpret_cast<const unsigned short*>(w); q.ldb = (long)K * P; q.bHalf = bHalf; q.Cp = C; q.ldc = ldc; q.offC = offC; q.offCi = offCi;
    q.conj = 1.f; q.cscale = 1.f; q.M = (int)M; q.N = N; q.K = K;
    return q;
  };
  auto grid = [&](int cplx, int wide, int N) {
    const int bmr = cplx ? 128 : 256, bnc = wide ? (cplx ? 128 : 256) : (cplx ? 64 : 128);
    const long nM = (M + bmr - 1) / bmr, nN = (N + bnc - 1) / bnc;
    return dim3((unsigned)(((nM + 7) / 8) * 8 * nN));
  };
  struct L { const char* name; std::function<void()> f; };
  std::vector<L> ls;
  ls.push_back({"fwd conv1 m0  q<0,0>", [&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 2, 2, 1, 4, 3>), grid(0, 0, 640), dim3(512), 0, sa, mk(2, 2304, 0, 0, 0, 1408, 0, 0, 640, 768)); }});
  ls.push_back({"fwd conv1 m1  q<1,1>", [&] { hipLaunchKernelGGL((umx_gemm_q_kernel<1, 1, 2, 2, 1, 4, 3>), grid(1, 1, 256), dim3(512), 0, sa, mk(2, 2304, 768, 1280, 256, 1408, 640, 896, 256, 512)); }});
  ls.push_back({"fwd fc3       q<0,1>", [&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 3>), grid(0, 1, 1536), dim3(512), 0, sa, mk(2, 128, 0, 0, 0, 1536, 0, 0, 1536, 128)); }});
  ls.push_back({"rev conv2T m0 pl16<0,2,3,4,2,2,2>", [&] { hipLaunchKernelGGL((umx_gemm_pl16_kernel<0, 2, 3, 4, 2, 2, 2>), grid(0, 0, 384), dim3(512), 0, sa, mk(2, 1152, 0, 0, 0, 1152, 0, 0, 384, 384)); }});
  ls.push_back({"rev conv2T m1 pl16<1,2,2,4,2,2,4>", [&] { hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 256), dim3(512), 0, sa, mk(2, 1152, 384, 640, 256, 1152, 384, 640, 256, 256)); }});
  ls.push_back({"rev conv1T m0 pl16<0,2,2,4,2,2,4>", [&] { hipLaunchKernelGGL((umx_gemm_pl16_kernel<0, 2, 2, 4, 2, 2, 4>), grid(0, 1, 768), dim3(512), 0, sa, mk(2, 1408, 0, 0, 0, 2304, 0, 0, 768, 640)); }});
  ls.push_back({"rev conv1T m1 pl16<1,2,2,4,2,2,4>", [&] { hipLaunchKernelGGL((umx_gemm_pl16_kernel<1, 2, 2, 4, 2, 2, 4>), grid(1, 1, 512), dim3(512), 0, sa, mk(2, 1408, 640, 896, 512, 2304, 768, 1280, 512, 256)); }});
  ls.push_back({"rev fc3T      pl<0,2,3,4,2,2,2>", [&] { hipLaunchKernelGGL((umx_gemm_pl_kernel<0, 2, 3, 4, 2, 2, 2>), grid(0, 0, 128), dim3(512), 0, sa, mk(2, 1536, 0, 0, 0, 128, 0, 0, 128, 1536)); }});
  for (auto& l : ls) {
    CK(hipMemset(C, 0, cbytes)); CK(hipDeviceSynchronize());
    l.f(); CK(hipDeviceSynchronize());
    CK(hipMemcpy(Cref, C, cbytes, hipMemcpyDeviceToDevice));
    int bad_runs = 0; unsigned long long worst = 0;
    for (int it = 0; it < iters; ++it) {
      CK(hipMemset(C, 0, cbytes)); CK(hipMemset(d_cnt, 0, 8)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_copy_full, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, sb, src, dst, n4);
      l.f();
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const unsigned int*>(C), reinterpret_cast<const unsigned int*>(Cref), cbytes / 4, d_cnt);
      unsigned long long c = 0; CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
      if (c) { ++bad_runs; worst = std::max(worst, c); }
    }
    printf("%-36s beside a copy kernel: %d of %d runs differ from the solo result (worst: %llu dwords)\n", l.name, bad_runs, iters, worst);
    fflush(stdout);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "pmc")) return pmc_mode(argc > 2 ? atoi(argv[2]) : 2);
  if (argc > 1 && !strcmp(argv[1], "pmc3")) return pmc_mode(argc > 2 ? atoi(argv[2]) : 2, true);
  if (argc > 1 && !strcmp(argv[1], "verify")) return verify_mode(argc > 2 ? atol(argv[2]) : 100000, argc > 3 ? atoi(argv[3]) : 40);
  const long M = argc > 1 ? atol(argv[1]) : 569632; const int N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const double copy_gb = argc > 4 ? atof(argv[4]) : 8.0;
  const int R = 8;
  unsigned char *Aq, *Bq; float* C;
  const long Mp = (M + 3) / 4 * 4;
  CK(hipMalloc(&Aq, (size_t)Mp * K * 6)); CK(hipMalloc(&Bq, (size_t)N * K * 6)); CK(hipMalloc(&C, (size_t)M * N * 4));
  {  // random bf16 planes (finite values): fill with a byte pattern of small-exponent bf16 numbers
    std::vector<unsigned short> h(1 << 22);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    for (size_t o = 0; o < (size_t)Mp * K * 6; o += h.size() * 2) CK(hipMemcpy(Aq + o, h.data(), std::min(h.size() * 2, (size_t)Mp * K * 6 - o), hipMemcpyHostToDevice));
    CK(hipMemcpy(Bq, h.data() + 17, (size_t)N * K * 6, hipMemcpyHostToDevice));
  }
  if (getenv("ZERO")) {   // DVFS check: the same kernel on all-zero operands (less switching power -> higher clock, MI355X_MICROARCH.md give-back item 1)
    CK(hipMemset(Aq, 0, (size_t)Mp * K * 6)); CK(hipMemset(Bq, 0, (size_t)N * K * 6));
    printf("operands: ALL ZERO\n");
  }
  const long n4 = (long)(copy_gb * 1e9 / 16);
  float4 *src, *dst;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 1, n4 * 16));
  GemmPL gq; std::memset(&gq, 0, sizeof(gq)); gq.conj = 1.f;
  gq.Apl = reinterpret_cast<const unsigned short*>(Aq); gq.lda = 3L * K; gq.Bpl = reinterpret_cast<const unsigned short*>(Bq); gq.ldb = 3L * K;
  gq.Cp = C; gq.ldc = N; gq.M = (int)M; gq.N = N; gq.K = K;
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  auto gridq = [&](int bn) { const long nm = (M + 255) / 256, nn = (N + bn - 1) / bn; return dim3((unsigned)(((nm + 7) / 8) * 8 * nn)); };
  auto gemm_wide = [&] { for (int r = 0; r < R; ++r) hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1>), gridq(256), dim3(512), 0, sa, gq); };
  auto gemm_narrow = [&] { for (int r = 0; r < R; ++r) hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0>), gridq(128), dim3(512), 0, sa, gq); };
  auto wall = [&](auto f) { CK(hipDeviceSynchronize()); const double t0 = now_ms(); f(); CK(hipDeviceSynchronize()); return now_ms() - t0; };
  auto best = [&](auto f) { double b = 1e30; for (int i = 0; i < 3; ++i) b = std::min(b, wall(f)); return b; };
  const double flops = 2.0 * M * N * K * 6 * R;
  printf("GEMM Q3 P=3 M=%ld N=%d K=%d x%d; copy %.1f GB read + %.1f GB write\n", M, N, K, R, copy_gb, copy_gb);
  auto gemm_wide3 = [&] { for (int r = 0; r < R; ++r) hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 3, 3>), gridq(256), dim3(512), 0, sa, gq); };
  auto gemm_narrow3 = [&] { for (int r = 0; r < R; ++r) hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 3>), gridq(128), dim3(512), 0, sa, gq); };
  auto gemm_narrow4 = [&] { for (int r = 0; r < R; ++r) hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 4>), gridq(128), dim3(512), 0, sa, gq); };
  const char* names[5] = {"wide S=2  ", "wide S=3  ", "narrow S=2", "narrow S=3", "narrow S=4"};
  for (int wide = 0; wide < 5; ++wide) {
    auto gemm = [&] { if (wide == 0) gemm_wide(); else if (wide == 1) gemm_wide3(); else if (wide == 2) gemm_narrow(); else if (wide == 3) gemm_narrow3(); else gemm_narrow4(); };
    gemm(); CK(hipDeviceSynchronize());
    const double tg = best(gemm);
    printf("%s GEMM alone: %.2f ms (%.0f TFLOP/s executed)\n", names[wide], tg, flops / tg / 1e9);
    struct V { const char* name; std::function<void()> f; };
    std::vector<V> vs;
    vs.push_back({"copy full grid            ", [&] { hipLaunchKernelGGL(k_copy_full, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, sb, src, dst, n4); }});
    for (int G : {256, 512})
      for (int U : {4}) {
        char* nm = new char[64]; snprintf(nm, 64, "copy persistent G=%4d U=%2d", G, U);
        if (U == 4) vs.push_back({nm, [&, G] { hipLaunchKernelGGL(k_copy_pers<4>, dim3(G), dim3(256), 0, sb, src, dst, n4); }});
        if (U == 8) vs.push_back({nm, [&, G] { hipLaunchKernelGGL(k_copy_pers<8>, dim3(G), dim3(256), 0, sb, src, dst, n4); }});
        if (U == 16) vs.push_back({nm, [&, G] { hipLaunchKernelGGL(k_copy_pers<16>, dim3(G), dim3(256), 0, sb, src, dst, n4); }});
      }
    for (auto& v : vs) {
      const double tc = best(v.f);
      const double tb = best([&] { v.f(); gemm(); });
      const double tb2 = best([&] { gemm(); v.f(); });
      printf("  %s alone %6.2f ms (%5.2f TB/s) | together %6.2f / %6.2f ms (copy first / gemm first) | sum %6.2f max %6.2f | overlap eff %.2f\n", v.name, tc,
             2 * copy_gb / tc, tb, tb2, tg + tc, std::max(tg, tc), (tg + tc - std::min(tb, tb2)) / std::min(tg, tc));
      fflush(stdout);
    }
  }
  return 0;
}
