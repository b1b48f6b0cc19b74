// umx_gemm_q.h -- forward split-precision GEMM on the "quad-row" operand layouts, which let a 256 x 256 tile fit the LDS.
//
//   C[M x N] (fp32) = cscale * sum_{(i,j) in products} A_i[M x K] . B_j[N x K]^T ,   A_i, B_j 16-bit planes (q_use_product below)
//
// Two operand formats (producers: umx_kernels_pl.h QFmt; weights: umx_api.hip):
//   default : IEEE-half planes -- A = 2 planes of 16 x activation (256-B blocks), B = 3 planes of s x weight (exact; 384-B blocks),
//             4 products on v_mfma_f32_32x32x16_f16
//   "Q3"    : bf16 planes, 3 x 3 (384-B blocks both sides), 6 products on v_mfma_f32_32x32x16_bf16 (UMX_PRECISION=split-bf16)
//
// LAYOUT of a P-plane matrix X[rows][cols]: blocks of 4 rows x 16 columns x P planes = 128 P bytes = P whole 128-B lines,
//   element (r, k, plane q) -> byte ((r/4) * (cols/16) + k/16) * 128 P + (r%4) * 32 P + q * 32 + (k%16) * 2
// (rows padded to a multiple of 4).  A 16-column k-tile of a 256-row operand tile is 64 contiguous-by-block pieces = 8 P KB, so
// BOTH operands of a 256 x 256 tile times two ring stages take 80 KB (fp16 form) / 96 KB (Q3) -- with the 32-column plane-interleaved
// rows of the PL layout (umx_gemm_pl.h) one 3-plane stage of that tile is already 98 KB and the forward GEMMs were stuck at 256 x 128.
// A wider tile needs a third less L2->LDS fill per FLOP, which co-limits these kernels (NOTES.md section 5): measured -8...10 %
// against the PL kernels.  Every DMA instruction still fetches whole lines (8 P consecutive lanes cover one block).
// Fragment reads: lane = row, 16 B (8 k-values of one plane) per ds_read_b128, which is served in the lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32).  In the plain P = 3 image the bank base of a row group is 32*(group & 1) dwords,
// so row groups 0/6 and 3/5 of a 32-row fragment collide (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE before the fix); in the
// plain P = 2 image every row group starts on the same bank (4-way).  Fix without touching the HBM layout: the 16-B chunk index
// (plane * 2 + k-half) of a row piece is XOR-ed with q_swz<P>(row group) in LDS -- by swapping the SOURCE chunk of the DMA lane and
// XOR-ing the chunk in the fragment address: conflict-free for every plane and fragment base (profiles/r02_gemm_pmc_counters.json).
//
// Structure as umx_gemm_pl.h: LDS-DMA ring (2 stages, BK = 16), one raw s_barrier per k-tile, 8 waves (4 x 2), each wave owning
// 64 rows x (BN/2) columns of 32x32x16 MFMA tiles; CPLX as there (rows = (re/im, edge), weight rows = (A/B half, channel)).
// WIDE = 1: 256 x 256 tile (N must fill whole tiles); WIDE = 0: 256 x 128.
#pragma once
#include "umx_gemm_pl.h"

namespace umx {

template <int P> __device__ __forceinline__ int q_row_off(int row) { return (row >> 2) * (128 * P) + (row & 3) * (32 * P); }
// LDS-image swizzle: the 16-B chunk index inside a row piece (plane-major, k-half = LSB) is XOR-ed with q_swz<P>(row group).
// P = 3: rows whose row group has bit 2 set (tile rows 16-31, 48-63, ...) hold their two k-halves swapped.  P = 2: a row piece is
// 4 chunks = 16 banks and the four row groups one ds_read_b128 pass serves ({0,3,5,6} or {1,2,4,7} of a 32-row fragment) must land
// on four different chunks -> XOR with (g >> 1) & 3.
template <int P> __device__ __forceinline__ int q_swz(int g) { return P == 3 ? ((g >> 2) & 1) : ((g >> 1) & 3); }
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <int JA, int JBF, int BHALF_ROUND, int A_BYTES, int TAG>
__device__ __forceinline__ void q3_issue(const unsigned char* A, const unsigned char* B, unsigned char* sbase, const long (&a_off)[JA],
                                         const long (&b_off)[JBF + BHALF_ROUND], long kofs_a, long kofs_b, int piece, bool b_tail) {
#pragma unroll
  for (int j = 0; j < JA; ++j)
    __builtin_amdgcn_global_load_lds(A + a_off[j] + kofs_a, (__attribute__((address_space(3))) void*)(sbase + piece + j * 8192), 16, 0, 0);
#pragma unroll
  for (int j = 0; j < JBF; ++j)
    __builtin_amdgcn_global_load_lds(B + b_off[j] + kofs_b, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + piece + j * 8192), 16, 0, 0);
  if constexpr (BHALF_ROUND != 0)
    if (b_tail)                   // wave-uniform: the first four waves fetch the last half round
      __builtin_amdgcn_global_load_lds(B + b_off[JBF] + kofs_b, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + piece + JBF * 8192), 16, 0, 0);
}

// (Round 4 also had an "X8" form -- two fp16 planes + two bf8 planes of the activations, the two 2^-22-order products on
// v_mfma_scale_f32_32x32x64_f8f6f4 -- worth 0.7 % at c3 and removed in round 5; NOTES.md section 10 keeps the measurements and what was
// learnt about that instruction's inner sum.)
// ---- AF = 1: the A operand as plain float32, split into its three bf16 planes IN REGISTERS (round 4) ------------------------------------
// "QF" layout of A[rows][cols]: blocks of 4 rows x 16 columns fp32 = 256 B (the block geometry of the two-plane half format):
//   element (r, k) -> byte ((r/4) * (cols/16) + k/16) * 256 + (r%4) * 64 + (k%16) * 4
// LDS image: the DMA's flat chunk order with the 16-B piece index of a row XOR-ed with (row group & 3) -- conflict-free ds_read_b128.
// The lane's 8 k-values of a fragment are two 16-B pieces; qf_split2 turns a pair of floats into the three packed bf16 dwords with ONE
// v_cvt_pk_bf16_f32 per plane (11 VALU ops per pair: 88 per k-tile and wave beside 48 MFMAs) -- round to nearest, bit for bit the planes
// the producers' q_split2<0> wrote, so C is bitwise what the plane form gives.  Why: 4 B instead of 6 B per A element in HBM (producer
// write + GEMM read), 16 KB instead of 24 KB per k-tile through L2->LDS and the LDS read ports; csrc/gemm_bench.hip f16: 0.88-0.93 of the
// plane form's time on 256 x 256 tiles, the same on 256 x 128 tiles.  (A truncating split is no cheaper and gives every plane the sign
// of x -- the dropped terms would then be one-sided.)  The weights stay pre-split planes: their fragments are twice as many per wave.
// which plane products A_qa . B_qb a kernel accumulates:
//   bf16, 3 x 3 planes, 6 products: qa + qb < 3 (everything down to 2^-16 of the leading term; dropped terms are 2^-24)
//   fp16, 2 x 3 planes, 4 products: hh, hl, lh and A_hi . B_lo2 -- the weights (B, 33 bits in three half planes) are EXACT, so their
//         rounding cannot bias every atom the same way (two-plane weights shift the c3 energy by +2.5e-8 eV/atom, NOTES.md section 5);
//         the activations keep 22 bits with unbiased per-element rounding; dropped: A_lo . B_mid (2^-22, random sign) and below
//   fp16, 2 x 2 planes, 3 or 4 products: hh, hl, lh (+ ll)
__host__ __device__ constexpr bool q_use_product(int PA, int PB, int NPROD, int qa, int qb) {
  if (PA == 3 && PB == 3) return qa + qb < 3;
  if (PA == 2 && PB == 3) return NPROD == 5 ? (qa + qb < 3) : (qa + qb < 2 || (qa == 0 && qb == 2));
  return NPROD == 4 ? true : (qa + qb < 2);
}

// P / PB = planes of the A / B operand (PB defaults to P).  F16 = 1: the planes are IEEE half (11-bit significands) and the products
// run on v_mfma_f32_32x32x16_f16, else bf16.  NPROD = number of plane products (q_use_product).  The engine uses <.., 3, 2, 0, 6, 3, 1>
// (bf16x3 / split-bf16: A as float32, AF) and <.., 2, 2, 1, 4, 3> (split: fp16 planes, exact weights); the other forms are gemm_bench's.
// S = ring stages (2: request tile kt+1 while tile kt is consumed; 3: two tiles in flight -- more tolerant of HBM latency when
// other kernels load the memory system, at 144 KB of LDS for the wide tile).
// AL = 1 (AF kernels, forward products): "aligned planes" -- the leading plane of A is quantised to the lane's pass group (qf_align_magic,
// umx_gemm_pl.h); the weights' leading plane is quantised the same way when the planes are built (umx_api.hip).
template <int CPLX, int WIDE, int P = 3, int S = 2, int F16 = 0, int NPROD = (P == 3 ? 6 : 3), int PB = P, int AF = 0, int LS = 0, int AL = 0>
__global__ __launch_bounds__(512, 1) void umx_gemm_q_kernel(const GemmPL p) {
  static_assert(!AL || AF, "AL: the float32-A kernels");
  static_assert(!LS || (P == 3 && PB == 3 && !F16 && NPROD == 6), "LS: the six-product bf16 form");
  static_assert(LS != 2 || !WIDE, "LS = 2 (second accumulator set): 256 x 128 tiles only");
  static_assert(!AF || (P == 3 && PB == 3 && F16 == 0), "AF: the six-product bf16 form with A as float32");
  static_assert((P == 3 && PB == 3 && NPROD == 6) || (P == 2 && PB == 2 && (NPROD == 3 || NPROD == 4)) || (P == 2 && PB == 3 && (NPROD == 4 || NPROD == 5)), "plane products");
  constexpr int BM = 256, BN = WIDE ? 256 : 128;
  constexpr int BMR = CPLX ? BM / 2 : BM, BNC = CPLX ? BN / 2 : BN;
  constexpr int BLK = AF ? 256 : 128 * P, CPB = AF ? 16 : 8 * P;   // A: bytes per block, 16-B chunks per block (32 P bytes per row per block; AF: 64)
  constexpr int BLKB = 128 * PB, CPBB = 8 * PB;           // B likewise
  constexpr int A_BYTES = AF ? BM * 64 : BM * 32 * P, B_BYTES = BN * 32 * PB, STAGE = A_BYTES + B_BYTES;
  constexpr int TNW = WIDE ? 4 : 2;                       // 32-column MFMA tiles per wave
  constexpr int JA = A_BYTES / 8192;                      // DMA rounds of the whole block (512 lanes x 16 B)
  constexpr int JBF = B_BYTES / 8192, BHR = (B_BYTES % 8192) ? 1 : 0;
  static_assert(S * STAGE <= 160 * 1024 && A_BYTES % 8192 == 0 && (B_BYTES % 8192 == 0 || B_BYTES % 8192 == 4096), "tile geometry");
  static_assert((P == 2 || P == 3) && (PB == 2 || PB == 3), "two or three planes");
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
  const long b_blocks = p.K / 16;                          // (ldb is implied: K * PB)
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
    if (AF) a_off[j] = (grp * a_blocks + offA / 16) * BLK + (s >> 2) * 64 + (((s & 3) ^ (g & 3)) * 16);   // s = row in group * 4 + piece
    else    a_off[j] = (grp * a_blocks + offA / 16) * BLK + (s ^ q_swz<P>(g)) * 16;          // the 16-B half is the LSB of the chunk index
  }
#pragma unroll
  for (int j = 0; j < JBF + BHR; ++j) {
    const int c = tid + 512 * j, g = (c / CPBB) % (BN / 4), s = c % CPBB; // (% keeps the unused lanes of a half round in range)
    long grp;
    if (CPLX) { int cg = nt * (BNC / 4) + (g % (BNC / 4)); if (cg >= gN) cg = gN - 1; grp = (long)(g / (BNC / 4)) * (p.bHalf / 4) + cg; }
    else      { int cg = nt * (BN / 4) + g; if (cg >= gN) cg = gN - 1; grp = cg; }
    b_off[j] = grp * b_blocks * BLKB + (s ^ q_swz<PB>(g)) * 16;
  }
  const int piece = __builtin_amdgcn_readfirstlane(wave * 1024);
  const bool b_tail = __builtin_amdgcn_readfirstlane(wave < 4 ? 1 : 0) != 0;

  // The accumulators START from the bias (real GEMMs; scaled and signed the way the epilogue un-scales them, both exact): a bias added
  // to the finished float32 sum is "a value on the float32 grid plus a constant", whose rounding error is the SAME for every row whose
  // result shares a binade -- up to half an ulp per column, coherent over all edges, i.e. an energy error that grows with N (NOTES 11).
  // Started from the bias, every rounding of the chain sees edge-dependent low bits and the errors are zero-mean.
  const float cs = F16 ? p.cscale : 1.f;          // a power of two: exact (bf16 planes are unscaled, the multiply folds away)
  const float cso = p.odd_sign < 0.f ? -cs : cs;  // odd rows: the producer stored them negated (sign-alternating rows, umx_kernels_pl.h); row parity = r & 1
  f32x16 acc[2][TNW];
  f32x16 low[LS == 2 ? 2 : 1][LS == 2 ? TNW : 1];      // LS = 2 (below)
  if constexpr (LS == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) low[i][j][r] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    float b0 = 0.f;
    if (!CPLX) {
      const int col = nt * BN + wn * (32 * TNW) + j * 32 + l31;
      b0 = (p.bias && col < p.N) ? p.bias[col] / cs : 0.f;
    }
    const float b1 = p.odd_sign < 0.f ? -b0 : b0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (r & 1) ? b1 : b0;
  }

  int a_ad[2][P], b_ad[TNW][PB];       // byte address of this lane's 16-B fragment piece, per plane (chunk = plane * 2 + half, swizzled)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = CPLX ? (t * BMR + wm * 32 + l31) : (wm * 64 + t * 32 + l31);
#pragma unroll
    for (int q = 0; q < P; ++q)
      a_ad[t][q] = AF ? ((row >> 2) * 256 + (row & 3) * 64 + ((((h * 2 + (q & 1)) ^ ((row >> 2) & 3))) * 16))     // AF: entries 0 / 1 = the lane's two 16-B pieces
                      : (q_row_off<P>(row) + ((q * 2 + h) ^ q_swz<P>(row >> 2)) * 16);
  }
#pragma unroll
  for (int t = 0; t < TNW; ++t) {
    const int row = CPLX ? ((t / (TNW / 2)) * BNC + wn * (16 * TNW) + (t % (TNW / 2)) * 32 + l31) : (wn * (32 * TNW) + t * 32 + l31);
#pragma unroll
    for (int q = 0; q < PB; ++q) b_ad[t][q] = A_BYTES + q_row_off<PB>(row) + ((q * 2 + h) ^ q_swz<PB>(row >> 2)) * 16;
  }

  const int nk = p.K / 16;
  constexpr int TAG = 9000 + S * 100 + P * 10 + CPLX * 2 + WIDE + F16 * 1000 + NPROD * 10000 + PB * 100000 + AF * 10000000 + LS * 100000000 + AL * 500000000;   // one q3_issue instance per kernel
  constexpr int GI = JA + JBF;                            // DMA instructions per tile per wave (+1 for the waves that fetch the half round)
#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + t * STAGE, a_off, b_off, (long)t * BLK, (long)t * BLKB, piece, b_tail);
  int st_cur = 0, st_nxt = S - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the requests of the S-2 younger tiles are outstanding (fewer near the tail: wait for all)
    if (S == 2 || kt + S - 2 >= nk) wait_vmcnt<0>();
    else if (BHR != 0 && b_tail) wait_vmcnt<(S - 2) * (GI + 1)>();
    else wait_vmcnt<(S - 2) * GI>();
    __builtin_amdgcn_s_barrier();   // tile kt landed everywhere; everyone finished reading tile kt-1
    if (kt + S - 1 < nk) q3_issue<JA, JBF, BHR, A_BYTES, TAG>(Ab, Bb, ring + st_nxt * STAGE, a_off, b_off, (long)(kt + S - 1) * BLK, (long)(kt + S - 1) * BLKB, piece, b_tail);
    const unsigned char* sb = ring + st_cur * STAGE;
    st_cur = st_cur + 1 == S ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == S ? 0 : st_nxt + 1;
    bf16x8_t a[2][P], b[TNW][PB];
    if constexpr (AF) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4q_t lo = *reinterpret_cast<const f32x4q_t*>(sb + a_ad[t][0]), hi = *reinterpret_cast<const f32x4q_t*>(sb + a_ad[t][1]);
        unsigned int w[3][4];
        if constexpr (AL) {           // the lane's 8 k-values ARE one pass group of this row
          const float c = qf_align_magic(lo, hi);
          qf_split2q(lo[0], lo[1], qf_round_q(lo[0], c), qf_round_q(lo[1], c), w[0][0], w[1][0], w[2][0]);
          qf_split2q(lo[2], lo[3], qf_round_q(lo[2], c), qf_round_q(lo[3], c), w[0][1], w[1][1], w[2][1]);
          qf_split2q(hi[0], hi[1], qf_round_q(hi[0], c), qf_round_q(hi[1], c), w[0][2], w[1][2], w[2][2]);
          qf_split2q(hi[2], hi[3], qf_round_q(hi[2], c), qf_round_q(hi[3], c), w[0][3], w[1][3], w[2][3]);
        } else {
          qf_split2(lo[0], lo[1], w[0][0], w[1][0], w[2][0]); qf_split2(lo[2], lo[3], w[0][1], w[1][1], w[2][1]);
          qf_split2(hi[0], hi[1], w[0][2], w[1][2], w[2][2]); qf_split2(hi[2], hi[3], w[0][3], w[1][3], w[2][3]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { const u32x4q_t v{w[q][0], w[q][1], w[q][2], w[q][3]}; a[t][q] = __builtin_bit_cast(bf16x8_t, v); }
      }
    } else {
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t][q] = *reinterpret_cast<const bf16x8_t*>(sb + a_ad[t][q]);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q)
#pragma unroll
      for (int t = 0; t < TNW; ++t) b[t][q] = *reinterpret_cast<const bf16x8_t*>(sb + b_ad[t][q]);
    if constexpr (LS == 1) {
      // LS (round 5): the three plane products of order 2^-16 (a0.b2, a1.b1, a2.b0) accumulate APART from the large ones and join them
      // through one float32 add.  Why: the 16-bit matrix cores align the 16 products of an MFMA to its largest addend -- the accumulator
      // -- and drop what falls below ~2^-32 of it.  Products of order 2^-16 of the sum (16-bit significands) lose their last bits
      // there: a one-sided part (floor; the sign-alternating rows cancel it) AND a part that follows the sign of the product
      // (profiles/r04_mfma_adder_rounding.txt: all products positive -1.5e-10, all negative +2.2e-10 per MFMA).  Where the columns of A
      // are one-signed -- fc3's SiLU outputs, gated scalars, element embeddings -- sign(a0.b2) is the sign of the weight's third plane:
      // the same for every edge, a fixed offset per output column, an energy error that grows with N (tools/gpu_fc3_error_form.py: 835 of
      // 1536 fc3 columns off by > 4 standard errors, the float32 MFMA: 1; with LS: 67).  Accumulated among themselves these products keep
      // every bit; the three leading products end at 2^-24 of the sum and were never cut.
      //   LS = 1 (256 x 256 tiles, 239 VGPRs): one spare accumulator, folded in per tile and k-step (16 v_add_f32 behind an MFMA wait)
      //   LS = 2 (256 x 128 tiles, 190 VGPRs): a second accumulator set for the whole k loop, folded in once
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          f32x16 lo = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], lo, 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {      // scalar adds (an opaque asm per element): a vector add would be selected as v_pk_add_f32 (NOTES 5 item 14)
            float v = lo[r];
            asm("" : "+v"(v));
            acc[i][j][r] += v;
          }
        }
    } else
#pragma unroll
    for (int ord = P + PB - 2; ord >= 0; --ord)   // smallest terms first
#pragma unroll
      for (int qa = 0; qa < P; ++qa) {
        const int qb = ord - qa;
        if (qb < 0 || qb >= PB || !q_use_product(P, PB, NPROD, qa, qb)) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j) {
            if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[i][qa]), __builtin_bit_cast(f16x8_t, b[j][qb]), acc[i][j], 0, 0, 0);
            else if constexpr (LS == 2) {      // narrow tiles: the 2^-16-order products have accumulators of their own for the whole k loop (64 VGPRs)
              if (qa + qb == 2) low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], low[i][j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
            } else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
          }
      }
  }
  if constexpr (LS == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {        // scalar adds (an opaque asm per element: no v_pk_add_f32, NOTES 5 item 14)
          float v = low[i][j][r];
          asm("" : "+v"(v));
          acc[i][j][r] += v;
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
          cr[p.offC] = ((r & 1) ? cso : cs) * (acc[0][cg][r] - p.conj * acc[1][TNW / 2 + cg][r]);
          cr[p.offCi] = ((r & 1) ? cso : cs) * (acc[1][cg][r] + p.conj * acc[0][TNW / 2 + cg][r]);
        }
      } else if (chan < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (e0 + dr < p.M) {
            float* cr = c + (long)dr * p.ldc;
            cr[p.offC] = ((r & 1) ? cso : cs) * (acc[0][cg][r] - p.conj * acc[1][TNW / 2 + cg][r]);
            cr[p.offCi] = ((r & 1) ? cso : cs) * (acc[1][cg][r] + p.conj * acc[0][TNW / 2 + cg][r]);
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
        const long row0 = (long)mt * BM + wm * 64 + i * 32 + 4 * h;
        float* c = p.Cp + row0 * p.ldc + p.offC + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = ((r & 1) ? cso : cs) * acc[i][j][r];
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M) c[(long)dr * p.ldc] = ((r & 1) ? cso : cs) * acc[i][j][r];
          }
        }
      }
  }
}

}  // namespace umx
