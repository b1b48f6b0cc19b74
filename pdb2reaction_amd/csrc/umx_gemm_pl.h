// umx_gemm_pl.h -- multi-plane bf16 MFMA GEMM with an asynchronous LDS-DMA ring (gfx950).
//
//   C[M x N] (fp32) = sum_{i+j<P} A_i[M x K] . B_j[N x K]^T ,   A_i, B_j bf16 planes in HBM
//
// Both operands arrive already split into P bf16 planes (x = x0 + x1 (+ x2), exact): weights once at
// load time, activations by the HBM-bound producer kernel that writes them (one conversion per element
// instead of one per N tile).  P = 3 -> 6 MFMAs per product (fp32-equivalent, forward pass),
// P = 2 -> 3 MFMAs (reverse pass).
//
// PLANE-INTERLEAVED ROW LAYOUT ("PL" layout).  A matrix X[rows][K] is stored as rows of K*P bf16:
//   element (r, k, plane q)  ->  r * (K*P) + (k / 32) * (32*P) + q * 32 + (k % 32)
// so the P planes of one 32-column block of a row are contiguous (128 B for P=2, 192 B for P=3): one
// global_load_lds wave-instruction then fetches whole 128-B lines.  (With separate planes and a 16-wide
// k-step every line was requested four separate times and the fill, not the MFMA, set the pace.)
//
// Structure (cdna_hip_programming.md section 5; measured steps in NOTES.md section 5):
//  * 256 x (64*WNT) block tile, BK = 32, 4 waves (2x2), ONE block per CU, each wave owns a
//    128 x (32*WNT) C tile = 4 x WNT v_mfma_f32_32x32x16_bf16 tiles (up to 256 accumulator registers;
//    the wave has the whole 512-entry register file).
//  * Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no VALU): a ring of S
//    stages, tile kt+S-1 is requested while tile kt is consumed; the only waits are a counted
//    s_waitcnt vmcnt((S-2)*G) and ONE raw s_barrier per k-step (never __syncthreads(), which would drain
//    the DMA queue).  All LDS lives in one __shared__ array.
//  * LDS image = the DMA's flat chunk order (row-major, 4P 16-B slots per row, no padding possible);
//    bank conflicts are removed by permuting the SOURCE chunk with a per-row XOR and applying the same
//    involution to fragment reads (conflict-free ds_read_b128).
//  * Rows past M / N are clamped on the source side (their results are masked in the epilogue).
//  * CPLX: rows are (re/im, edge), weight rows (A/B half, channel); every wave holds both re/im row
//    groups and both A/B column groups of its edges x channels and combines them in the accumulators.
#pragma once
#include "umx_gemm.h"

namespace umx {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmPL {
  const unsigned short* Apl; long lda; int offA0, offA1;   // PL layout: lda = row pitch in bf16 (= K_total*P); offsets in COLUMNS (multiples of 32)
  const unsigned short* Bpl; long ldb; int bHalf;          // weights, PL layout (ldb = K*P)
  float* Cp; long ldc; int offC, offCi;
  const float* bias;
  float conj;
  int M, N, K;       // CPLX: M = edges, N = channels per half
  float cscale;      // fp16-plane kernels only (umx_gemm_q.h, F16 = 1): C = cscale * (A' . B'^T) undoes the power-of-two operand scales
  float odd_sign;    // +1, or -1 when the producer stored the A rows of ODD index negated (sign-alternating rows, umx_kernels_pl.h): the
                     // epilogue multiplies the accumulators of odd rows by it, so C is what it would be -- with the matrix core's one-sided
                     // rounding error reversed on every second row.  0 is read as +1 (dev programs that memset the struct).
};

// float32 A operands (AF = 1 kernels here and in umx_gemm_q.h): a pair of floats -> the three packed bf16 plane dwords, round to nearest,
// ONE v_cvt_pk_bf16_f32 per plane (11 VALU ops per pair) -- bit for bit the planes the producers' split writes
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4q_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4q_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void qf_split2(float x0, float x1, unsigned int& p0, unsigned int& p1, unsigned int& p2) {
  p0 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
  const float r0 = x0 - __builtin_bit_cast(float, p0 << 16), r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  const float s0 = r0 - __builtin_bit_cast(float, p1 << 16), s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}


// "Aligned planes" (round 6; the bit-exact model of the matrix core's adder, tools/mfma_emul.c / NOTES.md section 12).  One pass of a 16-bit
// MFMA takes the 8 products of one lane's 8 k-values, cuts each of them TOWARD ZERO at 2^-24 of the pass's largest product exponent and only
// then adds them: a product more than 2^-10 below the largest loses low bits with an error that follows ITS SIGN -- coherent over all edges
// where an activation column is one-signed and consistently small (a dead SiLU unit), i.e. an energy error that grows with N.  The cure
// loses no bit: the value that goes into the LEADING plane is first rounded to a multiple of Q = 2^(e_max - 12), e_max = exponent of the
// largest of the lane's 8 values (= the pass group of this row); the remainder x - plane0 goes down the planes as before.  Elements within
// 2^-5 of the group's largest keep their 8 leading bits, smaller ones get fewer (down to none).  The weights' leading plane is quantised the
// same way at load time (umx_api.hip), so the lowest bit of every leading product lies at or above 2^(e_a,max + e_b,max - 24) >= 2^(epmax - 24)
// and stage 1 has nothing to cut.  Magic-number rounding: (x + c) - c with c = 1.5 * 2^(e_max + 11) rounds x to a multiple of 2^(e_max - 12),
// to nearest even; 4 + 3 + 16 VALU operations per 8 values.  (Scalar code element by element: a vector form would be selected as v_pk_add_f32.)
__device__ __forceinline__ float qf_align_magic(const f32x4q_t& lo, const f32x4q_t& hi) {
  float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(lo[0]), __builtin_fabsf(lo[1])), __builtin_fabsf(lo[2]));
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(lo[3])), __builtin_fabsf(hi[0]));
  m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(hi[1])), __builtin_fabsf(hi[2]));
  m = __builtin_fmaxf(m, __builtin_fabsf(hi[3]));
  const unsigned int cb = ((__builtin_bit_cast(unsigned int, m) & 0x7f800000u) + 0x05800000u) | 0x00400000u;
  return __builtin_bit_cast(float, cb);
}
__device__ __forceinline__ float qf_round_q(float x, float c) {
  float t = x + c;
  asm("" : "+v"(t));          // keep the two roundings apart (and out of a packed-fp32 instruction)
  return t - c;
}
// qf_split2 with the leading plane taken from (q0, q1) = the aligned values of (x0, x1)
__device__ __forceinline__ void qf_split2q(float x0, float x1, float q0, float q1, unsigned int& p0, unsigned int& p1, unsigned int& p2) {
  p0 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{q0, q1}, bf16x2_t));
  const float r0 = x0 - __builtin_bit_cast(float, p0 << 16), r1 = x1 - __builtin_bit_cast(float, p0 & 0xffff0000u);
  p1 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  const float s0 = r0 - __builtin_bit_cast(float, p1 << 16), s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// per-row slot permutation (an involution): LDS slot s of row r holds source chunk pl_perm(s, r)
template <int P> __device__ __forceinline__ int pl_perm(int s, int r) {
  return P == 2 ? (s ^ ((r >> 1) & 7)) : ((s & ~3) | ((s & 3) ^ ((r >> 2) & 3)));
}

// request one k-tile (32 columns, all planes) of A and B into the ring stage at `sbase`.
// (A plain __device__ function: a lambda calling the LDS-DMA builtin silently drops the host-side kernel stub, and
//  hipcc 7.2 rejects a second kernel re-using one specialization of such a function -- hence the TAG parameter.)
template <int JA, int JB, int TA_B, int WSTR, int TAG, int AUXA = 0>
__device__ __forceinline__ void pl_issue(const GemmPL& p, unsigned char* sbase, const long (&a_off)[JA], const long (&b_off)[JB], long kofs, int piece, long kofs_a = -1) {
  if (kofs_a < 0) kofs_a = kofs;      // (float32 A rows advance by 64 shorts per k-tile, the planes of B by 32 P)
#pragma unroll
  for (int j = 0; j < JA; ++j)
    __builtin_amdgcn_global_load_lds(p.Apl + a_off[j] + kofs_a, (__attribute__((address_space(3))) void*)(sbase + piece + j * WSTR), 16, 0, AUXA);
#pragma unroll
  for (int j = 0; j < JB; ++j)
    __builtin_amdgcn_global_load_lds(p.Bpl + b_off[j] + kofs, (__attribute__((address_space(3))) void*)(sbase + TA_B + piece + j * WSTR), 16, 0, 0);
}

// Geometry: WVM x WVN waves, each owning WMT x WNT MFMA tiles (32x32): block tile = (32*WVM*WMT) x (32*WVN*WNT).
//   <.., WVM=2, WVN=2, WMT=4, WNT=2>  256x128, 4 waves (one per SIMD, 128 accumulator registers)
//   <.., WVM=4, WVN=2, WMT=2, WNT=2>  256x128, 8 waves (two per SIMD: one wave's LDS-read latency hides under the other's MFMAs)
//   <.., WVM=2, WVN=4, WMT=4, WNT=2>  256x256, 8 waves (half the fill bytes per FLOP; P=2 only, LDS)
// ABL (dev only): 1 = no DMA in the main loop, 2 = no MFMA, 4 = no C stores, 8 = no LDS fragment reads, 16 = nt A stream
template <int CPLX, int P, int S, int WVM, int WVN, int WMT, int WNT, int ABL = 0>
__global__ __launch_bounds__(64 * WVM * WVN, 1) void umx_gemm_pl_kernel(const GemmPL p) {
  constexpr int NT = 64 * WVM * WVN;           // threads
  constexpr int BM = 32 * WVM * WMT;           // block tile rows (A rows)
  constexpr int BN = 32 * WVN * WNT;           // block tile columns (B rows)
  static_assert(!CPLX || (WMT % 2 == 0 && WNT % 2 == 0), "complex tiles need re/im and A/B halves in every wave");
  constexpr int SEG = 4 * P;                   // 16-B chunks per row per k-tile
  constexpr int ROWB = SEG * 16;               // bytes per row per k-tile (128 or 192)
  constexpr int TA_B = BM * ROWB;
  constexpr int TB_B = BN * ROWB;
  constexpr int STAGE_B = TA_B + TB_B;
  static_assert(S * STAGE_B <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[S * STAGE_B];
  constexpr int BMR = CPLX ? BM / 2 : BM;      // logical rows (edges) per block
  constexpr int BNC = CPLX ? BN / 2 : BN;      // logical cols (channels) per block
  constexpr int JA = BM * SEG / NT;            // A chunks per lane per k-tile
  constexpr int JB = BN * SEG / NT;
  static_assert(BM * SEG % NT == 0 && BN * SEG % NT == 0, "tile does not divide over the threads");
  constexpr int G = JA + JB;                   // global_load_lds instructions per wave per k-tile
  constexpr int WSTR = NT * 16;                // LDS bytes covered by one DMA round of the whole block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WVN, wn = wave % WVN;
  const int l31 = lane & 31, h = lane >> 5;

  const int nN = (p.N + BNC - 1) / BNC;
  const int nM = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;

  // ---- per-lane source offsets (bf16 elements): flat chunk id c = tid + 256 j -> row = c / SEG, LDS slot = c % SEG
  long a_off[JA], b_off[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int c = tid + NT * j, trow = c / SEG, s = c % SEG;
    long grow; int offA;
    if (CPLX) { grow = (long)mt * BMR + (trow % BMR); offA = (trow / BMR) ? p.offA1 : p.offA0; }
    else      { grow = (long)mt * BM + trow;          offA = p.offA0; }
    if (grow >= p.M) grow = p.M - 1;
    a_off[j] = grow * p.lda + (long)offA * P + pl_perm<P>(s, trow) * 8;
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int c = tid + NT * j, trow = c / SEG, s = c % SEG;
    int brow;
    if (CPLX) { int cc = nt * BNC + (trow % BNC); if (cc >= p.N) cc = p.N - 1; brow = (trow / BNC) * p.bHalf + cc; }
    else      { brow = nt * BN + trow; if (brow >= p.N) brow = p.N - 1; }
    b_off[j] = (long)brow * p.ldb + pl_perm<P>(s, trow) * 8;
  }
  const int piece = __builtin_amdgcn_readfirstlane(wave * 1024);   // this wave's 1-KiB piece inside one DMA round

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows inside the tile
  // CPLX: tile t of a wave = (half = t / (W?T/2), group = t % (W?T/2)); halves are re/im rows and A/B weight rows
  int a_row[WMT], b_row[WNT];
#pragma unroll
  for (int t = 0; t < WMT; ++t)
    a_row[t] = CPLX ? ((t / (WMT / 2)) * BMR + wm * (16 * WMT) + (t % (WMT / 2)) * 32 + l31) : (wm * (32 * WMT) + t * 32 + l31);
#pragma unroll
  for (int t = 0; t < WNT; ++t)
    b_row[t] = CPLX ? ((t / (WNT / 2)) * BNC + wn * (16 * WNT) + (t % (WNT / 2)) * 32 + l31) : (wn * (32 * WNT) + t * 32 + l31);

  const int nk = p.K / 32;
  constexpr int TAG = ((((ABL * 2 + CPLX) * 4 + P) * 8 + S) * 8 + WVM) * 64 + WVN * 16 + WMT * 2 + WNT / 2;
  constexpr int AUXA = (ABL & 16) ? 2 : 0;   // dev: non-temporal A stream
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) pl_issue<JA, JB, TA_B, WSTR, TAG, AUXA>(p, ring + s * STAGE_B, a_off, b_off, (long)s * 32 * P, piece);

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + S - 2 < nk) wait_vmcnt<(S - 2) * G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's tile kt landed; everyone finished reading tile kt-1
    if (kt + S - 1 < nk && !(ABL & 1))
      pl_issue<JA, JB, TA_B, WSTR, TAG, AUXA>(p, ring + ((kt + S - 1) % S) * STAGE_B, a_off, b_off, (long)(kt + S - 1) * 32 * P, piece);
    const unsigned char* sbase = ring + (kt % S) * STAGE_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[WMT][P], b[WNT][P];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const int u = q * 4 + ks * 2 + h;      // source chunk wanted: plane q, k-chunk 2 ks + h
#pragma unroll
        for (int t = 0; t < WMT; ++t) {
          if (ABL & 8) { for (int z = 0; z < 8; ++z) a[t][q][z] = (__bf16)(float)(kt + t); }
          else a[t][q] = *reinterpret_cast<const bf16x8_t*>(sbase + a_row[t] * ROWB + pl_perm<P>(u, a_row[t]) * 16);
        }
#pragma unroll
        for (int t = 0; t < WNT; ++t) {
          if (ABL & 8) { for (int z = 0; z < 8; ++z) b[t][q][z] = (__bf16)(float)(kt - t); }
          else b[t][q] = *reinterpret_cast<const bf16x8_t*>(sbase + TA_B + b_row[t] * ROWB + pl_perm<P>(u, b_row[t]) * 16);
        }
      }
#pragma unroll
      for (int ord = P - 1; ord >= 0; --ord)     // smallest terms first
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          const int qb = ord - qa;
#pragma unroll
          for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
              if (ABL & 2) acc[i][j][0] += (float)a[i][qa][0] * (float)b[j][qb][1];
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
            }
        }
    }
  }

  // ---- epilogue (C/D map of 32x32 tiles: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)); straight-line
  //      stores on the block-uniform fast path, see gemm_epilogue in umx_gemm.h
  const bool full = ((long)mt * BMR + BMR <= p.M) && (nt * BNC + BNC <= p.N);
  const float osg = p.odd_sign < 0.f ? -1.0f : 1.0f;
  if (ABL & 4) {
    float sum = 0.f;
    for (int i = 0; i < WMT; ++i) for (int j = 0; j < WNT; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 1.2345f) p.Cp[0] = sum;
    return;
  }
  if (CPLX) {
#pragma unroll
    for (int eg = 0; eg < WMT / 2; ++eg)
#pragma unroll
      for (int cg = 0; cg < WNT / 2; ++cg) {
        const int chan = nt * BNC + wn * (16 * WNT) + cg * 32 + l31;
        const long e0 = (long)mt * BMR + wm * (16 * WMT) + eg * 32 + 4 * h;
        float* c = p.Cp + e0 * p.ldc + chan;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* cr = c + (long)((r & 3) + 8 * (r >> 2)) * p.ldc;
            const float sg = (r & 1) ? osg : 1.0f;                         // row parity = r & 1 (tile origins are even)
            cr[p.offC] = sg * (acc[eg][cg][r] - p.conj * acc[WMT / 2 + eg][WNT / 2 + cg][r]);
            cr[p.offCi] = sg * (acc[WMT / 2 + eg][cg][r] + p.conj * acc[eg][WNT / 2 + cg][r]);
          }
        } else if (chan < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (e0 + dr < p.M) {
              float* cr = c + (long)dr * p.ldc;
              const float sg = (r & 1) ? osg : 1.0f;
              cr[p.offC] = sg * (acc[eg][cg][r] - p.conj * acc[WMT / 2 + eg][WNT / 2 + cg][r]);
              cr[p.offCi] = sg * (acc[WMT / 2 + eg][cg][r] + p.conj * acc[eg][WNT / 2 + cg][r]);
            }
          }
        }
      }
  } else {
    float bv[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = nt * BN + wn * (32 * WNT) + j * 32 + l31;
      bv[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int col = nt * BN + wn * (32 * WNT) + j * 32 + l31;
        const long row0 = (long)mt * BM + wm * (32 * WMT) + i * 32 + 4 * h;
        float* c = p.Cp + row0 * p.ldc + p.offC + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = ((r & 1) ? osg : 1.0f) * acc[i][j][r] + bv[j];
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (row0 + dr < p.M) c[(long)dr * p.ldc] = ((r & 1) ? osg : 1.0f) * acc[i][j][r] + bv[j];
          }
        }
      }
  }
}

// Same kernel on v_mfma_f32_16x16x32_bf16 (one MFMA k-step per 32-wide k-tile; the chip may hold a different clock on this shape).
// AF = 1 (round 4): the A operand as plain float32 ROWS (row pitch lda shorts = 2 x columns; what the node-centric k_modrot_bwd_pl<0>
// writes edge by edge), split into the three bf16 planes in registers as in umx_gemm_q.h.  A row of a k-tile is 32 floats = 128 B = the
// geometry of the two-plane rows: the LDS slot of the lane's piece i (k = 8 h + 4 i ...) is the two-plane image's slot of (plane i,
// k-chunk h), so the proven conflict-free pl_perm<2> serves; source chunk = 2 h + i.
template <int CPLX, int P, int S, int WVM, int WVN, int WMT, int WNT, int ABL = 0, int AF = 0>
__global__ __launch_bounds__(64 * WVM * WVN, 1) void umx_gemm_pl16_kernel(const GemmPL p) {
  static_assert(!AF || P == 3, "AF: the six-product form");
  constexpr int NT = 64 * WVM * WVN;           // threads
  constexpr int BM = 32 * WVM * WMT;           // block tile rows (A rows)
  constexpr int BN = 32 * WVN * WNT;           // block tile columns (B rows)
  static_assert(!CPLX || (WMT % 2 == 0 && WNT % 2 == 0), "complex tiles need re/im and A/B halves in every wave");
  constexpr int SEG = 4 * P;                   // 16-B chunks per row per k-tile
  constexpr int ROWB = SEG * 16;               // bytes per row per k-tile (128 or 192)
  constexpr int SEGA = AF ? 8 : SEG, ROWA = SEGA * 16;
  constexpr int TA_B = BM * ROWA;
  constexpr int TB_B = BN * ROWB;
  constexpr int STAGE_B = TA_B + TB_B;
  static_assert(S * STAGE_B <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[S * STAGE_B];
  constexpr int BMR = CPLX ? BM / 2 : BM;      // logical rows (edges) per block
  constexpr int BNC = CPLX ? BN / 2 : BN;      // logical cols (channels) per block
  constexpr int JA = BM * SEGA / NT;           // A chunks per lane per k-tile
  constexpr int JB = BN * SEG / NT;
  static_assert(BM * SEGA % NT == 0 && BN * SEG % NT == 0, "tile does not divide over the threads");
  constexpr int G = JA + JB;                   // global_load_lds instructions per wave per k-tile
  constexpr int WSTR = NT * 16;                // LDS bytes covered by one DMA round of the whole block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WVN, wn = wave % WVN;
  const int l15 = lane & 15, h = lane >> 4;       // 16x16x32: lane supplies row l15, k-chunk h (8 bf16 = 16 B)
  constexpr int TM = 2 * WMT, TN = 2 * WNT;       // 16-row / 16-column MFMA tiles per wave

  const int nN = (p.N + BNC - 1) / BNC;
  const int nM = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;

  // ---- per-lane source offsets (bf16 elements): flat chunk id c = tid + 256 j -> row = c / SEG, LDS slot = c % SEG
  long a_off[JA], b_off[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int c = tid + NT * j, trow = c / SEGA, s = c % SEGA;
    long grow; int offA;
    if (CPLX) { grow = (long)mt * BMR + (trow % BMR); offA = (trow / BMR) ? p.offA1 : p.offA0; }
    else      { grow = (long)mt * BM + trow;          offA = p.offA0; }
    if (grow >= p.M) grow = p.M - 1;
    if (AF) { const int u = pl_perm<2>(s, trow); a_off[j] = grow * p.lda + (long)offA * 2 + (2 * (u & 3) + (u >> 2)) * 8; }
    else a_off[j] = grow * p.lda + (long)offA * P + pl_perm<P>(s, trow) * 8;
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int c = tid + NT * j, trow = c / SEG, s = c % SEG;
    int brow;
    if (CPLX) { int cc = nt * BNC + (trow % BNC); if (cc >= p.N) cc = p.N - 1; brow = (trow / BNC) * p.bHalf + cc; }
    else      { brow = nt * BN + trow; if (brow >= p.N) brow = p.N - 1; }
    b_off[j] = (long)brow * p.ldb + pl_perm<P>(s, trow) * 8;
  }
  const int piece = __builtin_amdgcn_readfirstlane(wave * 1024);   // this wave's 1-KiB piece inside one DMA round

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // fragment rows inside the tile
  // CPLX: tile t of a wave = (half = t / (W?T/2), group = t % (W?T/2)); halves are re/im rows and A/B weight rows
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
    a_row[t] = CPLX ? ((t / (TM / 2)) * BMR + wm * (16 * WMT) + (t % (TM / 2)) * 16 + l15) : (wm * (32 * WMT) + t * 16 + l15);
#pragma unroll
  for (int t = 0; t < TN; ++t)
    b_row[t] = CPLX ? ((t / (TN / 2)) * BNC + wn * (16 * WNT) + (t % (TN / 2)) * 16 + l15) : (wn * (32 * WNT) + t * 16 + l15);

  const int nk = p.K / 32;
  constexpr int TAG = 1000000 + AF * 4000000 + ((((ABL * 2 + CPLX) * 4 + P) * 8 + S) * 8 + WVM) * 64 + WVN * 16 + WMT * 2 + WNT / 2;
  constexpr int AUXA = (ABL & 16) ? 2 : 0;   // dev: non-temporal A stream
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) pl_issue<JA, JB, TA_B, WSTR, TAG, AUXA>(p, ring + s * STAGE_B, a_off, b_off, (long)s * 32 * P, piece, AF ? (long)s * 64 : -1L);

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + S - 2 < nk) wait_vmcnt<(S - 2) * G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's tile kt landed; everyone finished reading tile kt-1
    if (kt + S - 1 < nk && !(ABL & 1))
      pl_issue<JA, JB, TA_B, WSTR, TAG, AUXA>(p, ring + ((kt + S - 1) % S) * STAGE_B, a_off, b_off, (long)(kt + S - 1) * 32 * P, piece, AF ? (long)(kt + S - 1) * 64 : -1L);
    const unsigned char* sbase = ring + (kt % S) * STAGE_B;
    bf16x8_t a[TM][P], b[TN][P];         // one MFMA k-step covers the whole 32-wide k-tile
    if constexpr (AF) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const f32x4q_t lo = *reinterpret_cast<const f32x4q_t*>(sbase + a_row[t] * ROWA + pl_perm<2>(h, a_row[t]) * 16);
        const f32x4q_t hi = *reinterpret_cast<const f32x4q_t*>(sbase + a_row[t] * ROWA + pl_perm<2>(4 + h, a_row[t]) * 16);
        unsigned int w[3][4];
        qf_split2(lo[0], lo[1], w[0][0], w[1][0], w[2][0]); qf_split2(lo[2], lo[3], w[0][1], w[1][1], w[2][1]);
        qf_split2(hi[0], hi[1], w[0][2], w[1][2], w[2][2]); qf_split2(hi[2], hi[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
        for (int q = 0; q < 3; ++q) { const u32x4q_t v{w[q][0], w[q][1], w[q][2], w[q][3]}; a[t][q] = __builtin_bit_cast(bf16x8_t, v); }
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int u = q * 4 + h;             // source chunk wanted: plane q, k-chunk h
      if constexpr (!AF) {
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t][q] = *reinterpret_cast<const bf16x8_t*>(sbase + a_row[t] * ROWB + pl_perm<P>(u, a_row[t]) * 16);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t][q] = *reinterpret_cast<const bf16x8_t*>(sbase + TA_B + b_row[t] * ROWB + pl_perm<P>(u, b_row[t]) * 16);
    }
#pragma unroll
    for (int ord = P - 1; ord >= 0; --ord)     // smallest terms first
#pragma unroll
      for (int qa = 0; qa <= ord; ++qa) {
        const int qb = ord - qa;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
      }
  }

  // ---- epilogue (C/D map of 16x16 tiles: col = lane&15, row = reg + 4*(lane>>4): row parity = reg & 1)
  const bool full = ((long)mt * BMR + BMR <= p.M) && (nt * BNC + BNC <= p.N);
  const float osg = p.odd_sign < 0.f ? -1.0f : 1.0f;
  if (CPLX) {
#pragma unroll
    for (int eg = 0; eg < TM / 2; ++eg)
#pragma unroll
      for (int cg = 0; cg < TN / 2; ++cg) {
        const int chan = nt * BNC + wn * (16 * WNT) + cg * 16 + l15;
        const long e0 = (long)mt * BMR + wm * (16 * WMT) + eg * 16 + 4 * h;
        float* c = p.Cp + e0 * p.ldc + chan;
        if (full) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* cr = c + (long)r * p.ldc;
            const float sg = (r & 1) ? osg : 1.0f;
            cr[p.offC] = sg * (acc[eg][cg][r] - p.conj * acc[TM / 2 + eg][TN / 2 + cg][r]);
            cr[p.offCi] = sg * (acc[TM / 2 + eg][cg][r] + p.conj * acc[eg][TN / 2 + cg][r]);
          }
        } else if (chan < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (e0 + r < p.M) {
              float* cr = c + (long)r * p.ldc;
              const float sg = (r & 1) ? osg : 1.0f;
              cr[p.offC] = sg * (acc[eg][cg][r] - p.conj * acc[TM / 2 + eg][TN / 2 + cg][r]);
              cr[p.offCi] = sg * (acc[TM / 2 + eg][cg][r] + p.conj * acc[eg][TN / 2 + cg][r]);
            }
          }
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = nt * BN + wn * (32 * WNT) + j * 16 + l15;
        const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        const long row0 = (long)mt * BM + wm * (32 * WMT) + i * 16 + 4 * h;
        float* c = p.Cp + row0 * p.ldc + p.offC + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 4; ++r) c[(long)r * p.ldc] = ((r & 1) ? osg : 1.0f) * acc[i][j][r] + bv;
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) c[(long)r * p.ldc] = ((r & 1) ? osg : 1.0f) * acc[i][j][r] + bv;
        }
      }
  }
}

}  // namespace umx
