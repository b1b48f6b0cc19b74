// umx_gemm_q.h -- forward (3-plane) split-bf16 GEMM on the "quad-row" operand layout Q3, which lets a 256 x 256 tile fit the LDS.
//
//   C[M x N] (fp32) = sum_{i+j<3} A_i[M x K] . B_j[N x K]^T ,   A_i, B_j bf16 planes
//
// LAYOUT Q3 of a matrix X[rows][cols]: blocks of 4 rows x 16 columns x 3 planes = 384 B = three whole 128-B lines,
//   element (r, k, plane q) -> byte ((r/4) * (cols/16) + k/16) * 384 + (r%4) * 96 + q * 32 + (k%16) * 2
// (rows padded to a multiple of 4).  A 16-column k-tile of a 256-row operand tile is 64 contiguous-by-block pieces = 24 KB, so
// BOTH operands of a 256 x 256 tile times two ring stages take 96 KB -- with the 32-column plane-interleaved rows of the PL layout
// (umx_gemm_pl.h) one stage of that tile is already 98 KB and the forward GEMMs were stuck at 256 x 128.  A wider tile needs a third
// less L2->LDS fill per FLOP, which co-limits these kernels: measured -8...10 % against the PL kernels.
// Every DMA instruction still fetches whole lines (24 consecutive lanes cover one 384-B block).  Fragment reads: lane = row, 16 B at
// (row/4)*384 + (row%4)*96 + q*32 + h*16.  A ds_read_b128 is served in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32),
// and the bank base of a row group is 32*(group & 1) dwords -- so in the plain image row groups 0/6 and 3/5 of a 32-row fragment
// collide (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE, profiles/r02_gemm_pmc_*).  Fix without touching the HBM layout: the two
// 16-B halves (k 0-7 / 8-15) of every plane piece are swapped in LDS for the row groups with bit 2 set, by swapping the SOURCE chunk
// of the DMA lane (q3_swz) and XOR-ing h in the fragment address -- conflict-free for every plane and fragment base.
//
// Structure as umx_gemm_pl.h: LDS-DMA ring (2 stages, BK = 16), one raw s_barrier per k-tile, 8 waves (4 x 2), each wave owning
// 64 rows x (BN/2) columns of v_mfma_f32_32x32x16_bf16 tiles; CPLX as there (rows = (re/im, edge), weight rows = (A/B half, channel)).
// WIDE = 1: 256 x 256 tile (N must fill whole tiles); WIDE = 0: 256 x 128.
#pragma once
#include "umx_gemm_pl.h"

namespace umx {

template <int P> __device__ __forceinline__ int q_row_off(int row) { return (row >> 2) * (128 * P) + (row & 3) * (32 * P); }
// LDS-image swizzle: rows whose row group has bit 2 set (tile rows 16-31, 48-63, ...) hold their two 16-B k-halves swapped
__device__ __forceinline__ int q3_swz_row(int row) { return (row >> 4) & 1; }
__device__ __forceinline__ int q3_swz_group(int g) { return (g >> 2) & 1; }

template <int JA, int JBF, int BHALF_ROUND, int A_BYTES, int TAG>
__device__ __forceinline__ void q3_issue(const unsigned char* A, const unsigned char* B, unsigned char* sbase, const long (&a_off)[JA],
                                         const long (&b_off)[JBF + BHALF_ROUND], long kofs, int piece, bool b_tail) {
#pragma unroll
  for (int j = 0; j < JA; ++j)
    __builtin_amdgcn_global_load_lds(A + a_off[j] + kofs, (__attribute__((address_space(3))) void*)(sbase + piece + j * 8192), 16, 0, 0);
#pragma unroll
  for (int j = 0; j < JBF; ++j)
    __builtin_amdgcn_global_load_lds(B + b_off[j] + kofs, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + piece + j * 8192), 16, 0, 0);
  if constexpr (BHALF_ROUND != 0)
    if (b_tail)                   // wave-uniform: the first four waves fetch the last half round
      __builtin_amdgcn_global_load_lds(B + b_off[JBF] + kofs, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + piece + JBF * 8192), 16, 0, 0);
}

// P = 3: the forward layout Q3 described above.  P = 2 ("Q2", 256-B blocks of 4 rows x 16 columns x 2 planes) exists for gemm_bench only.
// S = ring stages (2: request tile kt+1 while tile kt is consumed; 3: two tiles in flight -- more tolerant of HBM latency when
// other kernels load the memory system, at 144 KB of LDS for the wide tile).
template <int CPLX, int WIDE, int P = 3, int S = 2>
__global__ __launch_bounds__(512, 1) void umx_gemm_q_kernel(const GemmPL p) {
  constexpr int BM = 256, BN = WIDE ? 256 : 128;
  constexpr int BMR = CPLX ? BM / 2 : BM, BNC = CPLX ? BN / 2 : BN;
  constexpr int RB = 32 * P, BLK = 128 * P, CPB = 8 * P;      // bytes per row per block, bytes per block, 16-B chunks per block
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int TNW = WIDE ? 4 : 2;                       // 32-column MFMA tiles per wave
  constexpr int JA = A_BYTES / 8192;                      // DMA rounds of the whole block (512 lanes x 16 B)
  constexpr int JBF = B_BYTES / 8192, BHR = (B_BYTES % 8192) ? 1 : 0;
  static_assert(S * STAGE <= 160 * 1024 && A_BYTES % 8192 == 0 && (B_BYTES % 8192 == 0 || B_BYTES % 8192 == 4096), "tile geometry");
  static_assert(P == 2 || P == 3, "two or three planes");
  static_assert(S >= 2 && S <= 4, "ring depth");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int nN = (p.N + BNC - 1) / BNC, nM = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;

  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.Apl);
  const unsigned char* Bb = reinterpret_cast<const unsigned char*>(p.Bpl);
  const long a_blocks = p.lda / (16 * P);                  // 16-column blocks per row of A (lda = columns * P)
  const long b_blocks = p.K / 16;
  const long gA = ((long)p.M + 3) / 4;                     // row groups that exist (rows are padded to 4)
  const int gN = p.N / 4;
  long a_off[JA], b_off[JBF + BHR];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int c = tid + 512 * j, g = c / CPB, s = c % CPB;
    long grp; int offA;
    if (CPLX) { grp = (long)mt * (BMR / 4) + (g % (BMR / 4)); offA = (g / (BMR / 4)) ? p.offA1 : p.offA0; }
    else      { grp = (long)mt * (BM / 4) + g;                offA = p.offA0; }
    if (grp >= gA) grp = gA - 1;
    a_off[j] = (grp * a_blocks + offA / 16) * BLK + (s ^ q3_swz_group(g)) * 16;      // the 16-B half is the LSB of the chunk index
  }
#pragma unroll
  for (int j = 0; j < JBF + BHR; ++j) {
    const int c = tid + 512 * j, g = (c / CPB) % (BN / 4), s = c % CPB;   // (% keeps the unused lanes of a half round in range)
    long grp;
    if (CPLX) { int cg = nt * (BNC / 4) + (g % (BNC / 4)); if (cg >= gN) cg = gN - 1; grp = (long)(g / (BNC / 4)) * (p.bHalf / 4) + cg; }
    else      { int cg = nt * (BN / 4) + g; if (cg >= gN) cg = gN - 1; grp = cg; }
    b_off[j] = grp * b_blocks * BLK + (s ^ q3_swz_group(g)) * 16;
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

  int a_ad[2], b_ad[TNW];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = CPLX ? (t * BMR + wm * 32 + l31) : (wm * 64 + t * 32 + l31);
    a_ad[t] = q_row_off<P>(row) + (h ^ q3_swz_row(row)) * 16;
  }
#pragma unroll
  for (int t = 0; t < TNW; ++t) {
    const int row = CPLX ? ((t / (TNW / 2)) * BNC + wn * (16 * TNW) + (t % (TNW / 2)) * 32 + l31) : (wn * (32 * TNW) + t * 32 + l31);
    b_ad[t] = A_BYTES + q_row_off<P>(row) + (h ^ q3_swz_row(row)) * 16;
  }

  const int nk = p.K / 16;
  constexpr int TAG = 9000 + S * 100 + P * 10 + CPLX * 2 + WIDE;
  constexpr int GI = JA + JBF;                            // DMA instructions per tile per wave (+1 for the waves that fetch the half round)
#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + t * STAGE, a_off, b_off, (long)t * BLK, piece, b_tail);
  int st_cur = 0, st_nxt = S - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the requests of the S-2 younger tiles are outstanding (fewer near the tail: wait for all)
    if (S == 2 || kt + S - 2 >= nk) wait_vmcnt<0>();
    else if (BHR != 0 && b_tail) wait_vmcnt<(S - 2) * (GI + 1)>();
    else wait_vmcnt<(S - 2) * GI>();
    __builtin_amdgcn_s_barrier();   // tile kt landed everywhere; everyone finished reading tile kt-1
    if (kt + S - 1 < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + st_nxt * STAGE, a_off, b_off, (long)(kt + S - 1) * BLK, piece, b_tail);
    const unsigned char* sb = ring + st_cur * STAGE;
    st_cur = st_cur + 1 == S ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == S ? 0 : st_nxt + 1;
    bf16x8_t a[2][P], b[TNW][P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
#pragma unroll
      for (int t = 0; t < 2; ++t) a[t][q] = *reinterpret_cast<const bf16x8_t*>(sb + a_ad[t] + q * 32);
#pragma unroll
      for (int t = 0; t < TNW; ++t) b[t][q] = *reinterpret_cast<const bf16x8_t*>(sb + b_ad[t] + q * 32);
    }
#pragma unroll
    for (int ord = P - 1; ord >= 0; --ord)     // smallest terms first
#pragma unroll
      for (int qa = 0; qa <= ord; ++qa) {
        const int qb = ord - qa;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
      }
  }

  // ---- epilogue (C/D map of 32x32 tiles: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); block-uniform fast path
  const bool full = ((long)mt * BMR + BMR <= p.M) && (nt * BNC + BNC <= p.N);
  if (CPLX) {
#pragma unroll
    for (int cg = 0; cg < TNW / 2; ++cg) {
      const int chan = nt * BNC + wn * (16 * TNW) + cg * 32 + l31;
      const long e0 = (long)mt * BMR + wm * 32 + 4 * h;
      float* c = p.Cp + e0 * p.ldc + chan;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* cr = c + (long)((r & 3) + 8 * (r >> 2)) * p.ldc;
          cr[p.offC] = acc[0][cg][r] - p.conj * acc[1][TNW / 2 + cg][r];
          cr[p.offCi] = acc[1][cg][r] + p.conj * acc[0][TNW / 2 + cg][r];
        }
      } else if (chan < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (e0 + dr < p.M) {
            float* cr = c + (long)dr * p.ldc;
            cr[p.offC] = acc[0][cg][r] - p.conj * acc[1][TNW / 2 + cg][r];
            cr[p.offCi] = acc[1][cg][r] + p.conj * acc[0][TNW / 2 + cg][r];
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int col = nt * BN + wn * (32 * TNW) + j * 32 + l31;
        const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        const long row0 = (long)mt * BM + wm * 64 + i * 32 + 4 * h;
        float* c = p.Cp + row0 * p.ldc + p.offC + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r] + bv;
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M) c[(long)dr * p.ldc] = acc[i][j][r] + bv;
          }
        }
      }
  }
}

}  // namespace umx
