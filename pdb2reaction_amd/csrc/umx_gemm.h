// umx_gemm.h -- fp32-input MFMA GEMM family for the SO(2) / radial / atom-wise linears.
//
// C[M x N] = epilogue( prologue(A)[M x K] . B[N x K]^T ),  B = nn.Linear weight layout [out][in].
//
// gfx950 design notes (MI355X_MICROARCH.md / cdna_hip_programming.md section 3):
//  * v_mfma_f32_32x32x2_f32: exact f32 (k-ordered fmaf chain), 64 FLOP/clk/SIMD.  A/B operands
//    are one VGPR per lane: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31].  Because the sum
//    over k is order-free, an 8-wide K chunk is fed as 4 MFMAs where lane-half h supplies
//    k = 4h + r (r = 0..3): ONE ds_read_b128 per lane yields the operands of 4 MFMAs.
//  * 128x128x32 block tile, 4 waves as 2x2, each wave a 64x64 C tile = 2x2 MFMA tiles
//    (64 accumulator VGPRs).  LDS rows are padded to 36 floats so the 16-lane groups of
//    ds_read_b128 hit 16 distinct 4-bank slots (conflict free).  Double-buffered LDS
//    (73,728 B -> 2 blocks/CU), register-staged global loads because the prologue
//    (radial modulation / gaussian basis / SiLU) runs on the staged registers.
//  * XCD-aware block -> tile map: the 8 XCDs walk 8 different M tiles while consecutive blocks
//    of one XCD sweep the N tiles of the SAME M tile, so an A tile is fetched from HBM once and
//    re-read from that XCD's L2; weights (<= 2 MB per GEMM) stay L2 resident on every XCD.
//  * CPLX=1 implements the SO(2) m>0 "complex" linear without doubling K: rows are
//    (edge, re/im), weight rows are (A-half, B-half); each wave owns the 2x2 MFMA tiles
//    {re,im} x {A,B} of the same 32 edges x 32 channels, so y_re = P[re][A] - s P[im][B] and
//    y_im = P[im][A] + s P[re][B] combine element-wise in the accumulators (s=+1 forward,
//    s=-1 for the transposed/backward product).
#pragma once
#include "umx_common.h"

namespace umx {

enum AMode { A_PLAIN = 0, A_MODUL = 1, A_SILU = 3 };
enum EMode { E_BIAS = 0 };     // (round 5: the gaussian-basis prologue and the element-table epilogue of the unfused radial layers went with them, umx_radial.h)

struct GemmP {
  // A operand: row r at A + r*lda + offA{0,1} (+ blockIdx.z * zA); offA1 = imaginary rows (CPLX)
  const float* A; long lda; int offA0, offA1;
  const float* R; long ldr; int offR;            // A_MODUL: A .* R
  // B operand: weights [rows][ldb]; CPLX row of kind ab: ab*bHalf + n
  const float* B; long ldb; int bHalf;
  const unsigned short* Bpl; long bplane;        // split-bf16 kernels: plane q of the weights at Bpl + q*bplane (same [row][ldb] layout)
  // C: row r at C + r*ldc + offC (+ blockIdx.z * zC); offCi = imaginary output (CPLX)
  float* Cp; long ldc; int offC, offCi;
  const float* bias;                             // [N] or null
  const float* resid; long ldres; int offRes;    // optional residual added to the output (plain)
  float conj;                                    // CPLX combine sign
  long zA, zC, zRes;
  long zBl;                                      // != 0: SO(3)-linear mode -- blockIdx.z = l-primary coefficient (0..8), weights of degree l(z) at B + l*zBl,
                                                 //       bias only on z = 0 (one launch instead of one per degree)
  int M, N, K;                                   // CPLX: M = edges, N = channels per half
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BK = 32;
constexpr int G_LDK = 36;   // padded LDS row (floats)

// C/D map of 32x32 MFMA tiles (dtype independent on gfx950): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Stores are issued as straight-line code on a block-uniform fast path (tile fully inside M x N): a guarded store per
// element puts every store in its own basic block behind a compiler-inserted s_waitcnt vmcnt(0), and vmcnt counts
// stores too, so the 64+ stores of a thread would each wait for the previous one's round trip.
template <int CPLX, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[2][2], int mt, int nt, int wm, int wn, int l31, int h) {
  if (CPLX) {
    const int chan = nt * 64 + wn * 32 + l31;
    const long e0 = (long)mt * 64 + wm * 32 + 4 * h;
    const bool full = ((long)mt * 64 + 64 <= p.M) && (nt * 64 + 64 <= p.N);
    float* c = p.Cp + e0 * p.ldc + chan;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* cr = c + (long)((r & 3) + 8 * (r >> 2)) * p.ldc;
        cr[p.offC] = acc[0][0][r] - p.conj * acc[1][1][r];
        cr[p.offCi] = acc[1][0][r] + p.conj * acc[0][1][r];
      }
    } else if (chan < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (e0 + dr < p.M) {
          float* cr = c + (long)dr * p.ldc;
          cr[p.offC] = acc[0][0][r] - p.conj * acc[1][1][r];
          cr[p.offCi] = acc[1][0][r] + p.conj * acc[0][1][r];
        }
      }
    }
  } else {
    const long zoffC = (long)blockIdx.z * p.zC, zoffR = (long)blockIdx.z * p.zRes;
    const bool full = ((long)mt * 128 + 128 <= p.M);      // all rows of the block exist; columns are predicated once per lane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = nt * 128 + wn * 64 + j * 32 + l31;
        const long row0 = (long)mt * 128 + wm * 64 + i * 32 + 4 * h;
        float* c = p.Cp + row0 * p.ldc + p.offC + zoffC + col;
        if (full && col < p.N && EPI == E_BIAS && !p.resid) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r];
        } else if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long row = row0 + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
              float v = acc[i][j][r];                                  // (the bias is where the accumulator started, below)
              if (p.resid) v += p.resid[row * p.ldres + p.offRes + zoffR + col];
              p.Cp[row * p.ldc + p.offC + zoffC + col] = v;
            }
          }
        }
      }
  }
}

template <int AMODE, int CPLX, int EPI>
__global__ __launch_bounds__(256, 2) void umx_gemm_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][128][G_LDK];
  constexpr int BMR = CPLX ? 64 : 128;   // logical rows (edges) per block
  constexpr int BNC = CPLX ? 64 : 128;   // logical cols (channels) per block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int nN = (p.N + BNC - 1) / BNC;
  const int nM = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;
  const long zoffA = (long)blockIdx.z * p.zA;
  const float* Bz = p.B + (p.zBl ? (blockIdx.z == 0 ? 0 : blockIdx.z < 4 ? 1 : 2) * p.zBl : 0);

  // ---- staging assignment: thread -> 4 rows x one float4 of A and of B ------------------------
  const int k4 = (tid & 7) * 4;
  const int trow0 = tid >> 3;   // + 32*r
  float4 ra[4], rb[4];

  auto gload = [&](int kt) {
    const int k0 = kt * G_BK + k4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = trow0 + 32 * r;
      long grow; int offA;
      if (CPLX) { grow = (long)mt * 64 + (trow & 63); offA = (trow >> 6) ? p.offA1 : p.offA0; }
      else      { grow = (long)mt * 128 + trow;       offA = p.offA0; }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < p.M) {
        v = *reinterpret_cast<const float4*>(p.A + grow * p.lda + offA + zoffA + k0);
        if (AMODE == A_MODUL) {
          const float4 m = *reinterpret_cast<const float4*>(p.R + grow * p.ldr + p.offR + k0);
          v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        if (AMODE == A_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
      }
      ra[r] = v;
      int brow; bool ok;
      if (CPLX) { const int c = nt * 64 + (trow & 63); ok = c < p.N; brow = (trow >> 6) * p.bHalf + c; }
      else      { brow = nt * 128 + trow; ok = brow < p.N; }
      rb[r] = ok ? *reinterpret_cast<const float4*>(Bz + (long)brow * p.ldb + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = trow0 + 32 * r;
      *reinterpret_cast<float4*>(&lds[buf][0][trow][k4]) = ra[r];
      *reinterpret_cast<float4*>(&lds[buf][1][trow][k4]) = rb[r];
    }
  };

  // Real GEMMs start their accumulators FROM the bias: added to the finished sum it would be "float32-grid value + constant", a rounding
  // error shared by every row (coherent over edges -> an energy error that grows with N; umx_gemm_q.h, NOTES 11).
  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = nt * 128 + wn * 64 + j * 32 + l31;
    const float b0 = (!CPLX && p.bias && col < p.N && (p.zBl == 0 || blockIdx.z == 0)) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = b0;
  }

  int arow[2], brow_l[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    arow[t]   = CPLX ? (t * 64 + wm * 32 + l31) : (wm * 64 + t * 32 + l31);
    brow_l[t] = CPLX ? (t * 64 + wn * 32 + l31) : (wn * 64 + t * 32 + l31);
  }

  const int nk = p.K / G_BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      float4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const float4*>(&lds[buf][0][arow[t]][kc * 8 + 4 * h]);
        b[t] = *reinterpret_cast<const float4*>(&lds[buf][1][brow_l[t]][kc * 8 + 4 * h]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<CPLX, EPI>(p, acc, mt, nt, wm, wn, l31, h);
}


// ---- float64-accumulate variant for the NODE-level linears (atom-wise SO(3) linears, scalar MLP, readout and their transposes) ----
// Same operands and epilogue as umx_gemm_kernel<A_PLAIN|A_SILU, 0, E_BIAS>, but every product and the whole k-sum are carried in
// double and rounded to float32 ONCE (after bias and residual).  Node-level GEMMs are < 1 % of the work, so this costs nothing
// measurable, while it removes their share of the one-signed energy drift that grows with the number of atoms (a float32 dot
// product of 128 terms errs by ~3e-8 relative; rounded once the error is 3e-8 of the RESULT only and unbiased).  Used when the
// engine runs large systems (NOTES.md section 5, "energy error vs N").  64 x 64 tile, 16 x 16 threads, 4 x 4 outputs per thread.
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int AMODE>
__global__ __launch_bounds__(256) void k_gemm_f64acc(const GemmP p) {
  // v_mfma_f64_16x16x4_f64 (same peak as the f64 VALU on this part, but a quarter of the LDS operand traffic per FLOP: a VALU version
  // with 4 x 4 register tiles ran LDS-bound at 26 % of the f64 peak).  64 x 64 tile, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles;
  // operands converted to double once, while staging.  LDS tiles are ROW-major [64 rows][32 k] with rows of 34 doubles (68 dwords = 4 banks
  // past a multiple of 64): a staging store writes 32 consecutive doubles of one row per half-wave (all 64 banks once), a fragment read
  // takes 16 rows x 2 k per half-wave = bank pairs 4 i + 2 k, all distinct.  (The first version staged k-major with rows of 80 doubles:
  // conflict-free reads, but the 32 k of a staging store fell on 2 bank pairs -- 16-way conflicts that made the kernel LDS-write bound
  // at 28 % of the f64 matrix peak.)
  // Lane maps (cdna_hip_programming.md): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D: col = l & 15, row = (l >> 4) + 4 reg.
  constexpr int LD = 34;
  __shared__ double As[64][LD];
  __shared__ double Bs[64][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lk = lane >> 4;
  const int nN = (p.N + 63) / 64;
  const int mt = blockIdx.x / nN, nt = blockIdx.x % nN;
  const long zoffA = (long)blockIdx.z * p.zA;
  const float* Bz = p.B + (p.zBl ? (blockIdx.z == 0 ? 0 : blockIdx.z < 4 ? 1 : 2) * p.zBl : 0);
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  // k-steps of 32 with the NEXT step's operands fetched into registers before this step's MFMAs: small node counts (c1: 400 rows,
  // 7 row tiles) leave one workgroup per CU, so nothing else hides the global-load latency of the 4 ... 16 steps.
  float ra[8], rb[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, r = idx >> 5, kk = idx & 31;
      const long grow = (long)mt * 64 + r;
      ra[i] = grow < p.M ? p.A[grow * p.lda + p.offA0 + zoffA + k0 + kk] : 0.f;
      const int brow = nt * 64 + r;
      rb[i] = brow < p.N ? Bz[(long)brow * p.ldb + k0 + kk] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < p.K; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, r = idx >> 5, kk = idx & 31;
      As[r][kk] = (double)(AMODE == A_SILU ? silu_f(ra[i]) : ra[i]);
      Bs[r][kk] = (double)rb[i];
    }
    __syncthreads();
    if (k0 + 32 < p.K) fetch(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[wm * 32 + t * 16 + l15][ks * 4 + lk];
        b[t] = Bs[wn * 32 + t * 16 + l15][ks * 4 + lk];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const long zoffC = (long)blockIdx.z * p.zC, zoffR = (long)blockIdx.z * p.zRes;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = nt * 64 + wn * 32 + j * 16 + l15;
      if (col >= p.N) continue;
      const double bv = (p.bias && (p.zBl == 0 || blockIdx.z == 0)) ? (double)p.bias[col] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = (long)mt * 64 + wm * 32 + i * 16 + lk + 4 * r;
        if (row >= p.M) continue;
        double v = acc[i][j][r] + bv;
        if (p.resid) v += (double)p.resid[row * p.ldres + p.offRes + zoffR + col];
        p.Cp[row * p.ldc + p.offC + zoffC + col] = (float)v;
      }
    }
}

}  // namespace umx
