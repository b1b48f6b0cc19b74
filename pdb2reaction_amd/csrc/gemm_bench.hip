// gemm_bench.hip -- dev micro-benchmark / correctness check for the GEMM kernels (not part of libumx.so).
// usage: gemm_bench M N K
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "umx_gemm_pl.h"
#include "umx_gemm_q.h"
using namespace umx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

// fp32 [rows][ld] columns [0,K) -> PL layout (P planes interleaved per 32-column block), RNE split with exact residuals
template <int P>
__global__ void k_split(const float* __restrict__ src, long rows, long ld, int K, unsigned short* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i % K);
  float x = src[r * ld + k];
  for (int q = 0; q < P; ++q) { const __bf16 h = (__bf16)x; dst[r * K * P + (k / 32) * 32 * P + q * 32 + (k % 32)] = __builtin_bit_cast(unsigned short, h); x -= (float)h; }
}

// fp32 [rows][ld] columns [0,K) -> Q3 layout (umx_gemm_q.h)
template <int P>
__global__ void k_split_q(const float* __restrict__ src, long rows, long ld, int K, unsigned char* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i % K);
  float x = src[r * ld + k];
  unsigned short* d = reinterpret_cast<unsigned short*>(dst + ((r / 4) * (K / 16) + k / 16) * (128 * P) + (r % 4) * (32 * P) + (k % 16) * 2);
  for (int q = 0; q < P; ++q) { const __bf16 hh = (__bf16)x; d[q * 16] = __builtin_bit_cast(unsigned short, hh); x -= (float)hh; }
}

// fp32 -> two IEEE-half planes in the Q2 layout: hi = RNE(x), lo = RNE(x - hi) (subnormal halves kept)
template <int PQ>
__global__ void k_split_q_f16(const float* __restrict__ src, long rows, long ld, int K, unsigned char* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i % K);
  float x = src[r * ld + k];
  unsigned short* d = reinterpret_cast<unsigned short*>(dst + ((r / 4) * (K / 16) + k / 16) * (128 * PQ) + (r % 4) * (32 * PQ) + (k % 16) * 2);
  for (int q = 0; q < PQ; ++q) { const _Float16 hq = (_Float16)x; d[16 * q] = __builtin_bit_cast(unsigned short, hq); x -= (float)hq; }
}

// fp32 -> the two 8-bit planes of the X8 form (umx_gemm_q.h, O8 layout).  WEIGHT = 0 (activations): x1' = bf8(2^10 lo), x2' = bf8(2^20 (x - hi - lo));
// WEIGHT = 1: w0' = bf8(w), w1' = bf8(2^10 lo)
template <int WEIGHT>
__global__ void k_split_o8(const float* __restrict__ src, long rows, long ld, int K, unsigned char* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i % K);
  const float x = src[r * ld + k];
  const _Float16 h0 = (_Float16)x; const _Float16 h1 = (_Float16)(x - (float)h0);
  const float r2 = (x - (float)h0) - (float)h1;
  const float xc = fminf(fmaxf(x, -57344.f), 57344.f), l1 = (float)h1 * (float)(1 << Q8_SHIFT1);
  const int pk = WEIGHT ? __builtin_amdgcn_cvt_pk_bf8_f32(xc, l1, 0, false) : __builtin_amdgcn_cvt_pk_bf8_f32(l1, r2 * (float)(1 << Q8_SHIFT), 0, false);
  unsigned char* d = dst + (r * (K / 64) + k / 64) * 128 + (k % 64);
  d[0] = (unsigned char)(pk & 0xff); d[64] = (unsigned char)((pk >> 8) & 0xff);
}

// accuracy + speed of the fp16 two-plane forms against the bf16 three-plane form, with a float64 host reference on the first rows.
// Row r of A is scaled by 2^-(r % 28) when `ragged` to put the low planes into the half subnormal range.
static int f16_mode(long M, int N, int K) {
  float *A, *B, *C; unsigned char *Aq3, *Bq3, *Aq2, *Bq2;
  const long Mp = (M + 3) / 4 * 4;
  CK(hipMalloc(&A, M * (long)K * 4)); CK(hipMalloc(&B, (long)N * K * 4)); CK(hipMalloc(&C, M * (long)N * 4));
  CK(hipMalloc(&Aq3, (size_t)Mp * K * 6)); CK(hipMalloc(&Bq3, (size_t)N * K * 6)); CK(hipMalloc(&Aq2, (size_t)Mp * K * 4)); CK(hipMalloc(&Bq2, (size_t)N * K * 4));
  unsigned char* Bq3h; CK(hipMalloc(&Bq3h, (size_t)N * K * 6));
  unsigned char *A8, *B8; CK(hipMalloc(&A8, (size_t)Mp * K * 2)); CK(hipMalloc(&B8, (size_t)N * K * 2));
  const int R = 256;
  for (int ragged = 0; ragged < 2; ++ragged) {
    std::vector<float> h((size_t)1 << 22); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
    std::vector<float> ha((size_t)R * K), hb((size_t)N * K);
    for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) ha[(size_t)r * K + k] = h[(size_t)r * K + k] * (ragged ? std::ldexp(1.f, -(r % 28)) : 1.f);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = 0.1f * h[(1 << 21) + i];
    for (long o = 0; o < M * (long)K; o += h.size()) CK(hipMemcpy(A + o, h.data(), std::min<long>(h.size(), M * (long)K - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Aq3, 0, (size_t)Mp * K * 6)); CK(hipMemset(Aq2, 0, (size_t)Mp * K * 4));
    hipLaunchKernelGGL(k_split_q<3>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, (long)K, K, Aq3);
    hipLaunchKernelGGL(k_split_q<3>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bq3);
    hipLaunchKernelGGL(k_split_q_f16<2>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, (long)K, K, Aq2);
    hipLaunchKernelGGL(k_split_q_f16<2>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bq2);
    hipLaunchKernelGGL(k_split_q_f16<3>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bq3h);
    CK(hipMemset(A8, 0, (size_t)Mp * K * 2));
    hipLaunchKernelGGL(k_split_o8<0>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, (long)K, K, A8);
    hipLaunchKernelGGL(k_split_o8<1>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, B8);
    CK(hipDeviceSynchronize());
    std::vector<double> ref((size_t)R * N), nrm(R, 0.0);
    for (int r = 0; r < R; ++r) for (int n = 0; n < N; ++n) {
      double sacc = 0; for (int k = 0; k < K; ++k) sacc += (double)ha[(size_t)r * K + k] * (double)hb[(size_t)n * K + k];
      ref[(size_t)r * N + n] = sacc; nrm[r] = std::max(nrm[r], std::fabs(sacc));
    }
    GemmPL g3; std::memset(&g3, 0, sizeof(g3)); g3.conj = 1.f;
    g3.Apl = reinterpret_cast<const unsigned short*>(Aq3); g3.lda = 3L * K; g3.Bpl = reinterpret_cast<const unsigned short*>(Bq3); g3.ldb = 3L * K;
    g3.Cp = C; g3.ldc = N; g3.M = (int)M; g3.N = N; g3.K = K;
    g3.cscale = 1.f;
    GemmPL g2 = g3; g2.Apl = reinterpret_cast<const unsigned short*>(Aq2); g2.lda = 2L * K; g2.Bpl = reinterpret_cast<const unsigned short*>(Bq2); g2.ldb = 2L * K;
    GemmP pf; std::memset(&pf, 0, sizeof(pf)); pf.conj = 1.f; pf.A = A; pf.lda = K; pf.B = B; pf.ldb = K; pf.Cp = C; pf.ldc = N; pf.M = (int)M; pf.N = N; pf.K = K;
    const long nm = (M + 255) / 256; const dim3 gw((unsigned)(((nm + 7) / 8) * 8 * ((N + 255) / 256)));
    const long nM = (M + 127) / 128, nN = (N + 127) / 128; const dim3 gf((unsigned)(((nM + 7) / 8) * 8 * nN));
    const double fl = 2.0 * M * N * K;
    auto report = [&](const char* name, float ms) {
      std::vector<float> c((size_t)R * N); CK(hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0, rms = 0, shrink = 0, wsum = 0;     // error relative to the largest entry of the row (rows differ in scale when ragged)
      for (int r = 0; r < R; ++r) for (int n = 0; n < N; ++n) {
        const double rf = ref[(size_t)r * N + n], d = c[(size_t)r * N + n] - rf, e = std::fabs(d) / nrm[r];
        worst = std::max(worst, e); rms += e * e;
        shrink += d * rf / (nrm[r] * nrm[r]); wsum += rf * rf / (nrm[r] * nrm[r]);     // least-squares fit c = (1 + s) ref
      }
      printf("%s %-30s %8.3f ms %7.1f alg-TF/s   max rel err %.2e  rms %.2e  fitted gain-1 %+.2e\n", ragged ? "[ragged]" : "[flat]  ", name, ms, fl / ms / 1e9, worst, std::sqrt(rms / ((double)R * N)), shrink / wsum);
      fflush(stdout);
    };
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("fp32 MFMA", timeit([&] { hipLaunchKernelGGL((umx_gemm_kernel<A_PLAIN, 0, E_BIAS>), gf, dim3(256), 0, 0, pf); }));
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("Q3 bf16 x3 planes, 6 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1>), gw, dim3(512), 0, 0, g3); }));
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("Q2 f16 x2 planes, 4 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 2>), gw, dim3(512), 0, 0, g2); }));
    if (K % 64 == 0) {
      GemmPL g8 = g2; g8.A8 = A8; g8.lda8 = 2L * K; g8.B8 = B8; g8.Bpl = reinterpret_cast<const unsigned short*>(Bq3h); g8.ldb = 3L * K;
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("f16 x2 + bf8 / exact weights (X8)", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 3, 1>), gw, dim3(512), 0, 0, g8); }));
      // what does each 8-bit product add?  (C with it) - (C without it) against the host sum of the decoded planes
      std::vector<unsigned char> ha8((size_t)R * K * 2), hb8((size_t)N * K * 2);
      CK(hipMemcpy(ha8.data(), A8, ha8.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(hb8.data(), B8, hb8.size(), hipMemcpyDeviceToHost));
      auto dec = [](unsigned char b) { const int e = (b >> 2) & 31, m = b & 3; const double v = e ? std::ldexp(1.0 + m / 4.0, e - 15) : std::ldexp(m / 4.0, -14); return (b & 0x80) ? -v : v; };
      auto at = [&](const std::vector<unsigned char>& v, int r, int k, int q) { return dec(v[((size_t)r * (K / 64) + k / 64) * 128 + q * 64 + k % 64]); };
      std::vector<float> cs[4];
      for (int skip = 0; skip < 4; ++skip) {
        g8.x8_skip = skip; CK(hipMemset(C, 0, M * (long)N * 4));
        hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 3, 1>), gw, dim3(512), 0, 0, g8); CK(hipDeviceSynchronize());
        cs[skip].resize((size_t)R * N); CK(hipMemcpy(cs[skip].data(), C, cs[skip].size() * 4, hipMemcpyDeviceToHost));
      }
      for (int pr = 0; pr < 2; ++pr) {
        double sxy = 0, sxx = 0, syy = 0;
        for (int r = 0; r < 64; ++r) for (int n = 0; n < N; ++n) {
          double want = 0; for (int k = 0; k < K; ++k) want += at(ha8, r, k, pr) * at(hb8, n, k, 1 - pr);
          want = std::ldexp(want, -Q8_SHIFT);       // (2^-10 * 2^-10 and 2^-20 * 1)
          const double got = (double)cs[pr ? 1 : 2][(size_t)r * N + n] - (double)cs[3][(size_t)r * N + n];     // skip = 2 keeps product 0, skip = 1 keeps product 1
          sxy += want * got; sxx += want * want; syy += got * got;
        }
        printf("         8-bit product %d: (C with) - (C without) = %.4f x host sum of the decoded planes  (correlation %.4f, rms host %.3e)\n", pr, sxy / sxx, sxy / std::sqrt(sxx * syy), std::sqrt(sxx / (64.0 * N)));
      }
    }
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("Q2 f16 x2 planes, 3 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 3, 2>), gw, dim3(512), 0, 0, g2); }));
    CK(hipMemset(C, 0, M * (long)N * 4));
    { GemmPL g23 = g2; g23.Bpl = reinterpret_cast<const unsigned short*>(Bq3h); g23.ldb = 3L * K; g23.cscale = 1.f;
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("f16 A x2 / B x3 exact, 4 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 3>), gw, dim3(512), 0, 0, g23); }));
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("f16 A x2 / B x3, 5 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 5, 3>), gw, dim3(512), 0, 0, g23); })); }
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("Q2 f16 x2, 4 products, ring 3", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 3, 1, 4, 2>), gw, dim3(512), 0, 0, g2); }));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "f16")) return f16_mode(argc > 2 ? atol(argv[2]) : 569632, argc > 3 ? atoi(argv[3]) : 512, argc > 4 ? atoi(argv[4]) : 512);
  const long M = argc > 1 ? atol(argv[1]) : 569632; const int N = argc > 2 ? atoi(argv[2]) : 640, K = argc > 3 ? atoi(argv[3]) : 768;
  const long lda = 2304;
  float *A, *C, *C2, *B; unsigned short *Bp, *Ap, *Bp2, *Ap2;
  CK(hipMalloc(&A, M * lda * 4)); CK(hipMalloc(&C, M * (long)N * 4)); CK(hipMalloc(&C2, M * (long)N * 4)); CK(hipMalloc(&B, (long)N * K * 4));
  CK(hipMalloc(&Bp, 3L * N * K * 2)); CK(hipMalloc(&Ap, 3L * M * K * 2)); CK(hipMalloc(&Bp2, 2L * N * K * 2)); CK(hipMalloc(&Ap2, 2L * M * K * 2));
  std::vector<float> h(1 << 22); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (long o = 0; o < M * lda; o += h.size()) CK(hipMemcpy(A + o, h.data() + (o % 977), std::min<long>(h.size() - 977, M * lda - o) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data() + 13, (long)N * K * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_split<3>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, lda, K, Ap);
  hipLaunchKernelGGL(k_split<3>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bp);
  hipLaunchKernelGGL(k_split<2>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, lda, K, Ap2);
  hipLaunchKernelGGL(k_split<2>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bp2);
  CK(hipDeviceSynchronize());
  // the register-staged kernels take separate planes for B: give them fp32-derived planes via a second split (not timed for accuracy)
  GemmP p; std::memset(&p, 0, sizeof(p)); p.conj = 1.f;
  p.A = A; p.lda = lda; p.B = B; p.ldb = K; p.Bpl = Bp; p.bplane = (long)N * K; p.Cp = C; p.ldc = N; p.M = (int)M; p.N = N; p.K = K;
  GemmPL q; std::memset(&q, 0, sizeof(q)); q.conj = 1.f;
  q.Apl = Ap; q.lda = 3L * K; q.Bpl = Bp; q.ldb = 3L * K; q.Cp = C2; q.ldc = N; q.M = (int)M; q.N = N; q.K = K;
  GemmPL q2 = q; q2.Apl = Ap2; q2.lda = 2L * K; q2.Bpl = Bp2; q2.ldb = 2L * K;
  const long nM = (M + 127) / 128, nN = (N + 127) / 128; dim3 grid((unsigned)(((nM + 7) / 8) * 8 * nN)), block(256);
  const double fl = 2.0 * M * N * K;
  auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f alg-TFLOP/s\n", name, ms, fl / ms / 1e9); fflush(stdout); };
  auto check = [&](const char* name) {
    const long rows = std::min<long>(M, 300);
    std::vector<float> c1(rows * N), c2(rows * N), c3(200L * N), c4(200L * N);
    CK(hipMemcpy(c1.data(), C, rows * N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, rows * N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c3.data(), C + (M - 200) * N, 200L * N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c4.data(), C2 + (M - 200) * N, 200L * N * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < c1.size(); ++i) { md = std::max(md, (double)std::fabs(c1[i] - c2[i])); mr = std::max(mr, (double)std::fabs(c1[i])); }
    for (size_t i = 0; i < c3.size(); ++i) md = std::max(md, (double)std::fabs(c3[i] - c4[i]));
    printf("   check %-30s max|diff| %.3e (max|ref| %.3e)\n", name, md, mr);
    if (getenv("DBG")) { for (int r : {0, 1, 33, 130}) { printf("     row %d ref:", r); for (int c : {0, 1, 31, 32, 64, 100}) printf(" %9.4f", c1[(size_t)r * N + c]); printf("\n     row %d got:", r); for (int c : {0, 1, 31, 32, 64, 100}) printf(" %9.4f", c2[(size_t)r * N + c]); printf("\n"); } }
  };
  rep("fp32 PLAIN", timeit([&] { hipLaunchKernelGGL((umx_gemm_kernel<A_PLAIN, 0, E_BIAS>), grid, block, 0, 0, p); }));
  auto launch_pl = [&](void (*k)(const GemmPL), dim3 g, const GemmPL& a, int nthr = 256) { hipLaunchKernelGGL(k, g, dim3(nthr), 0, 0, a); };
  auto gridpl = [&](int bn) { const long nm = (M + 255) / 256, nn = (N + bn - 1) / bn; return dim3((unsigned)(((nm + 7) / 8) * 8 * nn)); };
  CK(hipMemset(C2, 0, M * (long)N * 4));
  rep("PL 8w 256x128 P=3 S=2", timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 3, 2, 4, 2, 2, 2>, gridpl(128), q, 512); }));
  check("PL 8w 256x128 P=3 S=2 vs fp32");
  CK(hipMemset(C2, 0, M * (long)N * 4));
  rep("PL 8w 256x128 P=2 S=3", timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 2, 3, 4, 2, 2, 2>, gridpl(128), q2, 512); }));
  check("PL 8w 256x128 P=2 S=3 vs fp32");
  CK(hipMemset(C2, 0, M * (long)N * 4));
  rep("PL16 8w 256x128 P=3 S=2 (16x16x32)", timeit([&] { launch_pl(&umx_gemm_pl16_kernel<0, 3, 2, 4, 2, 2, 2>, gridpl(128), q, 512); }));
  check("PL16 P=3 vs fp32");
  CK(hipMemset(C2, 0, M * (long)N * 4));
  rep("PL16 8w 256x128 P=2 S=3 (16x16x32)", timeit([&] { launch_pl(&umx_gemm_pl16_kernel<0, 2, 3, 4, 2, 2, 2>, gridpl(128), q2, 512); }));
  check("PL16 P=2 vs fp32");
  {
    auto grid128 = [&](int bn) { const long nm = (M + 127) / 128, nn = (N + bn - 1) / bn; return dim3((unsigned)(((nm + 7) / 8) * 8 * nn)); };
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("PL16 4w 128x128 P=2 S=2 (2 blocks/CU)", timeit([&] { launch_pl(&umx_gemm_pl16_kernel<0, 2, 2, 2, 2, 2, 2>, grid128(128), q2, 256); }));
    check("PL16 4w 128x128 P=2 vs fp32");
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("PL32 4w 128x128 P=2 S=2 (2 blocks/CU)", timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 2, 2, 2, 2, 2, 2>, grid128(128), q2, 256); }));
    check("PL32 4w 128x128 P=2 vs fp32");
  }
  if (N % 256 == 0) {      // Q2: two-plane quad-row layout (experiment)
    unsigned char *Aq2, *Bq2;
    const long Mp2 = (M + 3) / 4 * 4;
    CK(hipMalloc(&Aq2, (size_t)Mp2 * K * 4)); CK(hipMalloc(&Bq2, (size_t)N * K * 4));
    CK(hipMemset(Aq2, 0, (size_t)Mp2 * K * 4));
    hipLaunchKernelGGL(k_split_q<2>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, lda, K, Aq2);
    hipLaunchKernelGGL(k_split_q<2>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bq2);
    CK(hipDeviceSynchronize());
    GemmPL g2; std::memset(&g2, 0, sizeof(g2)); g2.conj = 1.f;
    g2.Apl = reinterpret_cast<const unsigned short*>(Aq2); g2.lda = 2L * K; g2.Bpl = reinterpret_cast<const unsigned short*>(Bq2); g2.ldb = 2L * K;
    g2.Cp = C2; g2.ldc = N; g2.M = (int)M; g2.N = N; g2.K = K;
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("Q2 256x256 P=2 (quad-row, BK=16)", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2>), gridpl(256), dim3(512), 0, 0, g2); }));
    check("Q2 256x256 P=2 vs fp32");
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("Q2 256x128 P=2 (quad-row, BK=16)", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 2>), gridpl(128), dim3(512), 0, 0, g2); }));
    check("Q2 256x128 P=2 vs fp32");
    CK(hipFree(Aq2)); CK(hipFree(Bq2));
  }
  if (N % 256 == 0) {
    unsigned char *Aq, *Bq;
    const long Mp = (M + 3) / 4 * 4;
    CK(hipMalloc(&Aq, (size_t)Mp * K * 6)); CK(hipMalloc(&Bq, (size_t)N * K * 6));
    CK(hipMemset(Aq, 0, (size_t)Mp * K * 6));
    hipLaunchKernelGGL(k_split_q<3>, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, lda, K, Aq);
    hipLaunchKernelGGL(k_split_q<3>, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, B, (long)N, (long)K, K, Bq);
    CK(hipDeviceSynchronize());
    GemmPL gq; std::memset(&gq, 0, sizeof(gq)); gq.conj = 1.f;
    gq.Apl = reinterpret_cast<const unsigned short*>(Aq); gq.lda = 3L * K; gq.Bpl = reinterpret_cast<const unsigned short*>(Bq); gq.ldb = 3L * K;
    gq.Cp = C2; gq.ldc = N; gq.M = (int)M; gq.N = N; gq.K = K;
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("Q3 256x256 P=3 (quad-row layout, BK=16)", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1>), gridpl(256), dim3(512), 0, 0, gq); }));
    check("Q3 256x256 P=3 vs fp32");
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("Q3 256x128 P=3", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0>), gridpl(128), dim3(512), 0, 0, gq); }));
    check("Q3 256x128 P=3 vs fp32");
    CK(hipFree(Aq)); CK(hipFree(Bq));
  }
  if (N % 256 == 0) {
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("PL32 8w 256x256 P=2 S=2 (wave 128x64)", timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 2, 2, 2, 4, 4, 2>, gridpl(256), q2, 512); }));
    check("PL32 256x256 P=2 vs fp32");
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("PL16 8w 256x256 P=2 S=2 (wave 128x64)", timeit([&] { launch_pl(&umx_gemm_pl16_kernel<0, 2, 2, 2, 4, 4, 2>, gridpl(256), q2, 512); }));
    check("PL16 256x256 P=2 vs fp32");
    CK(hipMemset(C2, 0, M * (long)N * 4));
    rep("PL16 8w 256x256 P=2 S=2 (wave 64x128)", timeit([&] { launch_pl(&umx_gemm_pl16_kernel<0, 2, 2, 4, 2, 2, 4>, gridpl(256), q2, 512); }));
    check("PL16 256x256 b P=2 vs fp32");
  }
  if (getenv("ABLATE")) {
#define ABL3(flag) rep("  P=3 S=2 8w ABL=" #flag, timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 3, 2, 4, 2, 2, 2, flag>, gridpl(128), q, 512); }))
#define ABL2(flag) rep("  P=2 S=3 8w ABL=" #flag, timeit([&] { launch_pl(&umx_gemm_pl_kernel<0, 2, 3, 4, 2, 2, 2, flag>, gridpl(128), q2, 512); }))
    ABL3(1); ABL3(4); ABL3(8); ABL3(5); ABL3(9); ABL3(12); ABL3(13); ABL3(2); ABL3(6);
    ABL2(1); ABL2(4); ABL2(8); ABL2(5); ABL2(9); ABL2(12); ABL2(13); ABL2(2); ABL2(6);
  }
  return 0;
}
