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


// ---- experiment (round 4, NOTES.md section 10): the A operand as plain fp32 (4 B per element instead of the 6 B of three bf16 planes), split into
// its three bf16 planes IN REGISTERS by the GEMM (truncating split: x = p0 + p1 + p2 exactly, 8 + 8 + 8 bits) -- 11 VALU ops per two values.
// "QF" layout of A: blocks of 4 rows x 16 columns fp32 = 256 B; element (r, k) -> ((r/4) * (cols/16) + k/16) * 256 + (r%4) * 64 + (k%16) * 4.
// LDS image: the DMA's flat chunk order, the 16-B piece index of a row XOR-ed with (row group & 3) (conflict-free ds_read_b128).
__global__ void k_copy_qf(const float* __restrict__ src, long rows, long ld, int K, unsigned char* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i % K);
  *reinterpret_cast<float*>(dst + ((r / 4) * (K / 16) + k / 16) * 256 + (r % 4) * 64 + (k % 16) * 4) = src[r * ld + k];
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#ifndef QF_TRUNC
#define QF_TRUNC 0
#endif
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned int& p0, unsigned int& p1, unsigned int& p2) {
#if QF_TRUNC
  const unsigned int u0 = __builtin_bit_cast(unsigned int, x0), u1 = __builtin_bit_cast(unsigned int, x1);
  p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
  const unsigned int v0 = __builtin_bit_cast(unsigned int, r0), v1 = __builtin_bit_cast(unsigned int, r1);
  p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned int, s1), __builtin_bit_cast(unsigned int, s0), 0x07060302u);
#else
  // round-to-nearest planes: exactly what the producers' q_split2<0> writes today (bitwise the same GEMM inputs), with ONE v_cvt_pk_bf16_f32
  // per pair and plane: 11 VALU ops per two values
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  p0 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
  const float r0 = x0 - __builtin_bit_cast(float, p0 << 16), r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  const float s0 = r0 - __builtin_bit_cast(float, p1 << 16), s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
#endif
}
template <int WIDE, int S>
__global__ __launch_bounds__(512, 1) void gemm_qf_kernel(const GemmPL p) {
  constexpr int PB = 3, BM = 256, BN = WIDE ? 256 : 128;
  constexpr int BLKB = 128 * PB, CPBB = 8 * PB;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 32 * PB, STAGE = A_BYTES + B_BYTES;
  constexpr int TNW = WIDE ? 4 : 2;
  constexpr int JA = A_BYTES / 8192, JBF = B_BYTES / 8192, BHR = (B_BYTES % 8192) ? 1 : 0;
  static_assert(S * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int nN = (p.N + BN - 1) / BN, nM = (p.M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.Apl);
  const unsigned char* Bb = reinterpret_cast<const unsigned char*>(p.Bpl);
  const long a_blocks = p.lda / 16, b_blocks = p.K / 16;
  const long gA = ((long)p.M + 3) / 4;
  const int gN = p.N / 4;
  long a_off[JA], b_off[JBF + BHR];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int c = tid + 512 * j, g = c >> 4, s = c & 15;
    long grp = (long)mt * (BM / 4) + g;
    if (grp >= gA) grp = gA - 1;
    a_off[j] = grp * a_blocks * 256 + (s >> 2) * 64 + (((s & 3) ^ (g & 3)) * 16);
  }
#pragma unroll
  for (int j = 0; j < JBF + BHR; ++j) {
    const int c = tid + 512 * j, g = (c / CPBB) % (BN / 4), s = c % CPBB;
    int cg = nt * (BN / 4) + g; if (cg >= gN) cg = gN - 1;
    b_off[j] = (long)cg * b_blocks * BLKB + (s ^ q_swz<PB>(g)) * 16;
  }
  const int piece = __builtin_amdgcn_readfirstlane(wave * 1024);
  const bool b_tail = __builtin_amdgcn_readfirstlane(wave < 4 ? 1 : 0) != 0;
  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_ad[2][2], b_ad[TNW][PB];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = wm * 64 + t * 32 + l31, g = row >> 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) a_ad[t][i] = g * 256 + (row & 3) * 64 + (((h * 2 + i) ^ (g & 3)) * 16);
  }
#pragma unroll
  for (int t = 0; t < TNW; ++t) {
    const int row = wn * (32 * TNW) + t * 32 + l31;
#pragma unroll
    for (int q = 0; q < PB; ++q) b_ad[t][q] = A_BYTES + q_row_off<PB>(row) + ((q * 2 + h) ^ q_swz<PB>(row >> 2)) * 16;
  }
  const int nk = p.K / 16;
  constexpr int TAG = 77000 + S * 10 + WIDE;
  constexpr int GI = JA + JBF;
#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + t * STAGE, a_off, b_off, (long)t * 256, (long)t * BLKB, piece, b_tail);
  int st_cur = 0, st_nxt = S - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (S == 2 || kt + S - 2 >= nk) wait_vmcnt<0>();
    else if (BHR != 0 && b_tail) wait_vmcnt<(S - 2) * (GI + 1)>();
    else wait_vmcnt<(S - 2) * GI>();
    __builtin_amdgcn_s_barrier();
    if (kt + S - 1 < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + st_nxt * STAGE, a_off, b_off, (long)(kt + S - 1) * 256, (long)(kt + S - 1) * BLKB, piece, b_tail);
    const unsigned char* sb = ring + st_cur * STAGE;
    st_cur = st_cur + 1 == S ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == S ? 0 : st_nxt + 1;
    bf16x8_t a[2][3], b[TNW][PB];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(sb + a_ad[t][0]), hi = *reinterpret_cast<const f32x4_t*>(sb + a_ad[t][1]);
      unsigned int w[3][4];
      split3_pair(lo[0], lo[1], w[0][0], w[1][0], w[2][0]); split3_pair(lo[2], lo[3], w[0][1], w[1][1], w[2][1]);
      split3_pair(hi[0], hi[1], w[0][2], w[1][2], w[2][2]); split3_pair(hi[2], hi[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
      for (int q = 0; q < 3; ++q) { const u32x4_t v{w[q][0], w[q][1], w[q][2], w[q][3]}; a[t][q] = __builtin_bit_cast(bf16x8_t, v); }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q)
#pragma unroll
      for (int t = 0; t < TNW; ++t) b[t][q] = *reinterpret_cast<const bf16x8_t*>(sb + b_ad[t][q]);
#pragma unroll
    for (int ord = 4; ord >= 0; --ord)
#pragma unroll
      for (int qa = 0; qa < 3; ++qa) {
        const int qb = ord - qa;
        if (qb < 0 || qb >= PB || qa + qb >= 3) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int col = nt * BN + wn * (32 * TNW) + j * 32 + l31;
      const long row0 = (long)mt * BM + wm * 64 + i * 32 + 4 * h;
      float* c = p.Cp + row0 * p.ldc + col;
      if (col < p.N)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < p.M) c[(long)dr * p.ldc] = acc[i][j][r];
        }
    }
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
    {
      unsigned char* Aqf; CK(hipMalloc(&Aqf, (size_t)Mp * K * 4)); CK(hipMemset(Aqf, 0, (size_t)Mp * K * 4));
      hipLaunchKernelGGL(k_copy_qf, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M, (long)K, K, Aqf);
      CK(hipDeviceSynchronize());
      GemmPL gf = g3; gf.Apl = reinterpret_cast<const unsigned short*>(Aqf); gf.lda = K;
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("A fp32 split in registers, 6 products", timeit([&] { hipLaunchKernelGGL((gemm_qf_kernel<1, 2>), gw, dim3(512), 0, 0, gf); }));
      {   // bit for bit the plane form?  (library kernels: umx_gemm_q_kernel with AF = 1 against AF = 0, wide and narrow)
        std::vector<float> c0((size_t)4096 * N), c1((size_t)4096 * N);
        auto cmp = [&](const char* nm) { CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost)); size_t nd = 0; for (size_t i = 0; i < c0.size(); ++i) nd += std::memcmp(&c0[i], &c1[i], 4) != 0; printf("         %-44s differing from the plane form in %zu of %zu values\n", nm, nd, c0.size()); };
        CK(hipMemset(C, 0, M * (long)N * 4)); hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1>), gw, dim3(512), 0, 0, g3); CK(hipDeviceSynchronize());
        CK(hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(C, 0, M * (long)N * 4)); hipLaunchKernelGGL((gemm_qf_kernel<1, 2>), gw, dim3(512), 0, 0, gf); CK(hipDeviceSynchronize()); cmp("bench kernel gemm_qf_kernel<1, 2>");
        CK(hipMemset(C, 0, M * (long)N * 4)); GemmPL gl = gf; gl.lda = 3L * K;      // (the library kernels keep lda = 3 x columns for every six-product form)
        hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 3, 2, 0, 6, 3, 1>), gw, dim3(512), 0, 0, gl); CK(hipDeviceSynchronize()); cmp("library kernel AF = 1, 256x256");
        const dim3 gn2((unsigned)(((nm + 7) / 8) * 8 * ((N + 127) / 128)));
        CK(hipMemset(C, 0, M * (long)N * 4)); hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0, 3, 2, 0, 6, 3, 1>), gn2, dim3(512), 0, 0, gl); CK(hipDeviceSynchronize()); cmp("library kernel AF = 1, 256x128");
        CK(hipMemset(C, 0, M * (long)N * 4)); hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0>), gn2, dim3(512), 0, 0, g3); CK(hipDeviceSynchronize()); cmp("library kernel planes, 256x128");
      }
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("... ring 3", timeit([&] { hipLaunchKernelGGL((gemm_qf_kernel<1, 3>), gw, dim3(512), 0, 0, gf); }));
      const dim3 gn((unsigned)(((nm + 7) / 8) * 8 * ((N + 127) / 128)));
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("... 256x128 tiles", timeit([&] { hipLaunchKernelGGL((gemm_qf_kernel<0, 2>), gn, dim3(512), 0, 0, gf); }));
      CK(hipMemset(C, 0, M * (long)N * 4));
      report("Q3 bf16 x3, 256x128 tiles", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 0>), gn, dim3(512), 0, 0, g3); }));
      CK(hipFree(Aqf));
    }
    CK(hipMemset(C, 0, M * (long)N * 4));
    report("Q2 f16 x2 planes, 4 products", timeit([&] { hipLaunchKernelGGL((umx_gemm_q_kernel<0, 1, 2, 2, 1, 4, 2>), gw, dim3(512), 0, 0, g2); }));
    // (the "f16 x2 + bf8" X8 form measured here in round 4 was removed in round 5: profiles/r04_gemm_shape_times_f16x2b8.txt, NOTES.md section 10)
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
