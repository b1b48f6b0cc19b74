// umx_gemm_bf16.h -- split-bf16 MFMA GEMM: fp32-equivalent products at the bf16 matrix rate.
//
// Every fp32 operand is split into P bf16 planes x = x0 + x1 (+ x2) (round-to-nearest, exact
// residuals), and C = sum_{i+j<P} A_i . B_j^T is accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//   P = 3 -> 6 MFMAs per product ("bf16x6", 24 significant bits: as accurate as fp32 MFMA -- the
//            energy tolerance of 1e-4 eV on 2000 atoms needs this, tools/precision_study.py);
//   P = 2 -> 3 MFMAs per product ("bf16x3", ~16 bits: reverse pass only; forces move by ~1e-5 eV/A).
// bf16 MFMA runs at 16x the fp32-MFMA rate, so the ceilings are 2.67x / 5.33x the fp32 roofline.
//
// Weights are split once at load time (planes in HBM as bf16); activations are split by the staging
// threads on their way into LDS (after the fp32 prologue: radial modulation), so each element is
// converted once per block.  Same 128x128x32 block tile / 2x2 waves / 64x64 wave tile / C layout /
// XCD-aware tile map / complex-in-accumulator trick as umx_gemm.h.  LDS rows are padded to 40 bf16
// (80 B) so the 16-lane groups of ds_read_b128 fall on 16 distinct 16-B slots.
#pragma once
#include "umx_gemm.h"

namespace umx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int GB_LDK = 40;   // padded LDS row (bf16 elements)

template <int P>
__device__ __forceinline__ void split_store(float4 v, __bf16* dst, int plane_stride) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < P; ++q) {
    bf16x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const __bf16 hq = (__bf16)x[c];
      o[c] = hq;
      x[c] -= (float)hq;
    }
    *reinterpret_cast<bf16x4*>(dst + q * plane_stride) = o;
  }
}

// ABL (dev only, tools/gemm_bench.hip): 1 = no fp32->bf16 split (raw truncation), 2 = no global A/R loads, 4 = no MFMA, 8 = no B loads
template <int AMODE, int CPLX, int P, int ABL = 0>
__global__ __launch_bounds__(256, 2) void umx_gemm_bf16_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2][P][128][GB_LDK];   // [A|B][plane][row][k]
  constexpr int BMR = CPLX ? 64 : 128;
  constexpr int BNC = CPLX ? 64 : 128;
  constexpr int PLANE = 128 * GB_LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int nN = (p.N + BNC - 1) / BNC;
  const int nM = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / nN) * 8 + xcd, nt = slot % nN;
  if (mt >= nM) return;

  const int k4 = (tid & 7) * 4, trow0 = tid >> 3;        // A staging: 4 rows x one float4
  const int k8 = (tid & 3) * 8, brow0 = tid >> 2;        // B staging: 2 rows x 8 bf16 per plane
  float4 ra[4];
  uint4 rb[P][2];

  auto gload = [&](int kt) {
    const int k0 = kt * G_BK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = trow0 + 32 * r;
      long grow; int offA;
      if (CPLX) { grow = (long)mt * 64 + (trow & 63); offA = (trow >> 6) ? p.offA1 : p.offA0; }
      else      { grow = (long)mt * 128 + trow;       offA = p.offA0; }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < p.M && !(ABL & 2)) {
        v = *reinterpret_cast<const float4*>(p.A + grow * p.lda + offA + k0 + k4);
        if (AMODE == A_MODUL) {
          const float4 m = *reinterpret_cast<const float4*>(p.R + grow * p.ldr + p.offR + k0 + k4);
          v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
      }
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int trow = brow0 + 64 * r;
      int brow; bool ok;
      if (CPLX) { const int c = nt * 64 + (trow & 63); ok = c < p.N; brow = (trow >> 6) * p.bHalf + c; }
      else      { brow = nt * 128 + trow; ok = brow < p.N; }
#pragma unroll
      for (int q = 0; q < P; ++q)
        rb[q][r] = (ok && !(ABL & 8)) ? *reinterpret_cast<const uint4*>(p.Bpl + q * p.bplane + (long)brow * p.ldb + k0 + k8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ABL & 1) {
#pragma unroll
        for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(&lds[0][q][trow0 + 32 * r][k4]) = make_uint2(__float_as_uint(ra[r].x) + q, __float_as_uint(ra[r].z));
      } else split_store<P>(ra[r], &lds[0][0][trow0 + 32 * r][k4], PLANE);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<uint4*>(&lds[1][q][brow0 + 64 * r][k8]) = rb[q][r];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int arow[2], brow_l[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    arow[t]   = CPLX ? (t * 64 + wm * 32 + l31) : (wm * 64 + t * 32 + l31);
    brow_l[t] = CPLX ? (t * 64 + wn * 32 + l31) : (wn * 64 + t * 32 + l31);
  }

  const int nk = p.K / G_BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][P], b[2][P];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < P; ++q) {
          a[t][q] = *reinterpret_cast<const bf16x8*>(&lds[0][q][arow[t]][ks * 16 + 8 * h]);
          b[t][q] = *reinterpret_cast<const bf16x8*>(&lds[1][q][brow_l[t]][ks * 16 + 8 * h]);
        }
      // smallest terms first: order i+j descending
#pragma unroll
      for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
        for (int qa = 0; qa <= ord; ++qa) {
          const int qb = ord - qa;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (ABL & 4) { acc[i][j][0] += (float)a[i][qa][0] * (float)b[j][qb][0]; }
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) lstore();
    __syncthreads();
  }
  gemm_epilogue<CPLX, E_BIAS>(p, acc, mt, nt, wm, wn, l31, h);
}

}  // namespace umx
