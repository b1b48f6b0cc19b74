// umx_radial.h -- the small layers of the radial MLP as ONE kernel per direction (gfx950).
//
// RadialMLP(x_edge) = fc3( SiLU(LN( fc2( SiLU(LN( fc1(x_edge) )) ) )) ),  x_edge = [64 gaussians of d | emb(Z_src) | emb(Z_dst)].
// fc3 (128 -> 1536, the large GEMM) stays on the split-bf16 path; everything in front of it -- and, in the reverse pass,
// everything behind fc3^T -- used to be 4 / 5 launches with an HBM round trip of a 128-float row between each of them:
//     forward  gaussians+fc1 (fp32 GEMM) -> LN+SiLU -> fc2 (fp32 GEMM) -> LN+SiLU+split          1.55 ms per 8-image c3 chunk
//     reverse  LN+SiLU bwd -> fc2^T (fp32 GEMM) -> LN+SiLU bwd -> fc1^T (fp32 GEMM) -> d/dd dot   1.8 ms
// Both chains are row-local (a 128-float row per edge), so here a workgroup keeps a 64-edge tile in LDS from the gaussians to the
// fc3 operand (forward) and from fc3^T's output to the scalar dE/dd (reverse).  The weights (128x64 and 128x128 fp32) live in
// REGISTERS: wave w owns output columns [32w, 32w+32) of both linears, and a lane holds, for its column, the k-values its
// v_mfma_f32_32x32x2_f32 operand slots need (lane-half h supplies k = 8c + 4h + r, as in umx_gemm.h) -- 96 / 128 VGPRs, loaded once
// per (persistent) workgroup.  LDS holds only activations: two 64 x 132-float buffers (row pad 4 floats: the 16-lane groups of
// ds_read_b128 hit 16 distinct 4-bank slots).  Arithmetic is the fp32 MFMA's (exact fp32 fma chains), i.e. what umx_gemm_kernel does.
// HBM traffic per edge: forward 20 B in, 1024 B (h1pre, h2pre for the reverse pass) + 768 B (Q3 planes) out; reverse 1.5 KB in, 4 B out.
#pragma once
#include "umx_common.h"
#include "umx_gemm.h"
#include "umx_gemm_q.h"
#include "umx_kernels_pl.h"

namespace umx {

constexpr int R_LD = 132;      // padded LDS row of a 128-wide activation tile (floats)
constexpr int R_LDG = 68;      // padded LDS row of the 64-wide gaussian tile
constexpr int R_TR = 2;        // 32-row MFMA tiles per workgroup tile (64 edges; two workgroups per CU)

// (dpp_add / row16_sum / wave_sum_dpp: umx_common.h)

// Transcendentals: the libm-accurate functions of umx_common.h.  These kernels are VALU-bound on exactly them (two SiLU per element pair, 64
// gaussians per edge); rounds 3-4 measured the hardware approximations (raw: -4 ms per c3 iteration but a one-signed energy shift; with
// argument reduction / a Newton step: -0.8 ms, as accurate per call, but a SiLU bias 2.3x libm's that shows at 20 000 atoms) and fp16 plane
// products for fc1 / fc2 (-2.5 ms, energy error at 20 000 atoms x3) -- both removed in round 5, NOTES.md sections 5 and 9 keep the numbers.
__device__ __forceinline__ float r_silu(float x) { return x * sigmoid_f(x); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it also waits for every
// GLOBAL store in flight (h1pre / h2pre / the fc3 operand rows), i.e. it exposes a full HBM write round trip at each of the seven
// barriers of a tile.  Every barrier of these kernels protects LDS buffers only (round 3; ablation: the stores cost 4.4 of 12.8 ms).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LayerNorm(128) + SiLU of one row held as 2 values per lane (the arithmetic of k_ln_silu_fwd)
__device__ __forceinline__ float2 ln_silu_row(float2 v, float2 ww, float2 bb) {
  const float mu = wave_sum_dpp(v.x + v.y) * (1.0f / RH);
  v.x -= mu; v.y -= mu;
  const float var = wave_sum_dpp(v.x * v.x + v.y * v.y) * (1.0f / RH);
  const Rstd rstd = rstd_eps(var, LN_EPS);            // compensated sum, in every mode (umx_common.h: `var + 1e-5f` rounds one way)
  return make_float2(r_silu(scale_rstd(v.x, rstd) * ww.x + bb.x), r_silu(scale_rstd(v.y, rstd) * ww.y + bb.y));
}
// The same on a row given as hi + lo (lo = what the float32 rounding of the row dropped, |lo| <= ulp(hi) / 2): the mean is taken of both, and the
// CENTRED value is formed in double before it is rounded -- fc1's row is "float32 MFMA sum + table constant of the element pair", and rounding
// THAT sum to float32 is one error for every edge of the pair (the "grid value + constant" trap of umx_gemm_q.h: tools/gpu_fc1_table_form.py,
// +1.0e-6 eV at 700 atoms on one weight set, 6.7 sigma; a CPU emulation reproduces it and shows the unrounded sum clean).  x - mean has the
// edge's own mean in it, so ITS rounding is not coherent.  The stored h1pre stays the float32 hi (the reverse pass differentiates through it).
__device__ __forceinline__ float2 ln_silu_row_hl(float2 hi, float2 lo, float2 ww, float2 bb) {
  const float mu = (wave_sum_dpp(hi.x + hi.y) + wave_sum_dpp(lo.x + lo.y)) * (1.0f / RH);
  float2 v;
  v.x = (float)(((double)hi.x - (double)mu) + (double)lo.x);
  v.y = (float)(((double)hi.y - (double)mu) + (double)lo.y);
  const float var = wave_sum_dpp(v.x * v.x + v.y * v.y) * (1.0f / RH);
  const Rstd rstd = rstd_eps(var, LN_EPS);
  return make_float2(r_silu(scale_rstd(v.x, rstd) * ww.x + bb.x), r_silu(scale_rstd(v.y, rstd) * ww.y + bb.y));
}
// backward of it (the arithmetic of k_ln_silu_bwd): go = dE/d(output), v = the pre-LayerNorm row
__device__ __forceinline__ float2 ln_silu_row_bwd(float2 go, float2 v, float2 ww, float2 bb) {
  const float mu = wave_sum_dpp(v.x + v.y) * (1.0f / RH);
  v.x -= mu; v.y -= mu;
  const float var = wave_sum_dpp(v.x * v.x + v.y * v.y) * (1.0f / RH);
  const float rstd = rstd_eps(var, LN_EPS).y;         // (reverse pass: feeds dE/dd only)
  const float xh0 = v.x * rstd, xh1 = v.y * rstd;
  const float gw0 = go.x * silu_grad_f(xh0 * ww.x + bb.x) * ww.x;
  const float gw1 = go.y * silu_grad_f(xh1 * ww.y + bb.y) * ww.y;
  const float m1 = wave_sum_dpp(gw0 + gw1) * (1.0f / RH);
  const float m2 = wave_sum_dpp(gw0 * xh0 + gw1 * xh1) * (1.0f / RH);
  return make_float2(rstd * (gw0 - m1 - xh0 * m2), rstd * (gw1 - m1 - xh1 * m2));
}

// acc[i] += A[rows i*32 + l31][k] . W[col][k] over NC chunks of 8 k-values; A in LDS (row pitch LD), W in registers
template <int NC, int LD, int TR>
__device__ __forceinline__ void rad_mma(const float* __restrict__ a_lds, const float4 (&wr)[NC], f32x16 (&acc)[TR], int l31, int h) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float4 a[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) a[i] = *reinterpret_cast<const float4*>(a_lds + (i * 32 + l31) * LD + c * 8 + 4 * h);
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, wr[c].x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, wr[c].y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, wr[c].z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, wr[c].w, acc[i], 0, 0, 0);
    }
  }
}

// ---- forward: d, Z_src, Z_dst  ->  h1pre, h2pre (kept for the reverse pass) and the fc3 operand --------------------------------
// OUTQ = 2: the fc3 operand as fp16 two-plane "Q2H" planes (QFmt<1>, UMX_PRECISION=split), 4: float32 quad-row blocks (QFmt<3>: bf16x3 / split-bf16),
// 0: fp32 rows (fp32 precision mode).   ts / tt: per-element tables of the embedding part of fc1 (tt includes the fc1 bias).
template <int OUTQ>
__global__ __launch_bounds__(256, 2) void k_radial_head(const float* __restrict__ evec, const int* __restrict__ ez, double gcoef,
                                                        const double* __restrict__ gmu, const float* __restrict__ w1g,
                                                        const double* __restrict__ ts, const double* __restrict__ tt,
                                                        const float* __restrict__ ln1w, const float* __restrict__ ln1b,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ ln2w, const float* __restrict__ ln2b,
                                                        float* __restrict__ h1pre, float* __restrict__ h2pre, void* __restrict__ out, long ne,
                                                        float odd_sign) {
  static_assert(OUTQ == 0 || OUTQ == 2 || OUTQ == 4, "fc3 operand format");
  constexpr int TR = R_TR, RT = 32 * TR;
  __shared__ __attribute__((aligned(16))) float bufA[RT * R_LD];
  __shared__ __attribute__((aligned(16))) float bufB[RT * R_LD];
  __shared__ float dbuf[RT];
  __shared__ int zbuf[RT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int col = wave * 32 + l31;                   // the output column this lane owns in both linears
  float4 W1[NG / 8], W2[RH / 8];
#pragma unroll
  for (int c = 0; c < NG / 8; ++c) W1[c] = *reinterpret_cast<const float4*>(w1g + col * NG + c * 8 + 4 * h);
#pragma unroll
  for (int c = 0; c < RH / 8; ++c) W2[c] = *reinterpret_cast<const float4*>(w2 + col * RH + c * 8 + 4 * h);
  const float bias2 = b2[col];
  const float2 l1w = *reinterpret_cast<const float2*>(ln1w + 2 * lane), l1b = *reinterpret_cast<const float2*>(ln1b + 2 * lane);
  const float2 l2w = *reinterpret_cast<const float2*>(ln2w + 2 * lane), l2b = *reinterpret_cast<const float2*>(ln2b + 2 * lane);
  const long ntiles = (ne + RT - 1) / RT;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long e0 = tile * RT;
    lds_barrier();                                 // the previous tile's buffers are free
    if (tid < RT) {
      const long e = e0 + tid < ne ? e0 + tid : ne - 1;
      dbuf[tid] = evec[e * 4 + 3];
      zbuf[tid] = ez[e];
    }
    lds_barrier();
    {   // gaussian basis of the tile -> bufA as [RT][R_LDG]: 256 / RT threads per row, RT / 4 columns each
      constexpr int TPR = 256 / RT;
      const int row = tid / TPR, c0 = (tid % TPR) * (RT / 4);
      const float d = dbuf[row];
#pragma unroll
      for (int q = 0; q < RT / 16; ++q) {
        // centres and coefficient in DOUBLE, the exponent rounded to float32 once (round 3): a float32 centre is off by up to 2.4e-7 A,
        // i.e. up to 4e-6 relative on a gaussian, the SAME for every edge -- the largest systematic term of the energy error ~ N
        const double dd = (double)d;
        float4 v; double t;
        t = dd - gmu[c0 + 4 * q + 0]; v.x = exp_f((float)(gcoef * t * t));
        t = dd - gmu[c0 + 4 * q + 1]; v.y = exp_f((float)(gcoef * t * t));
        t = dd - gmu[c0 + 4 * q + 2]; v.z = exp_f((float)(gcoef * t * t));
        t = dd - gmu[c0 + 4 * q + 3]; v.w = exp_f((float)(gcoef * t * t));
        *reinterpret_cast<float4*>(bufA + row * R_LDG + c0 + 4 * q) = v;
      }
    }
    lds_barrier();
    f32x16 acc[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    rad_mma<NG / 8, R_LDG, TR>(bufA, W1, acc, l31, h);
    // epilogue 1: the raw fc1 tile to bufB; the row pass below adds the element tables, writes h1pre (whole 512-B rows) and normalises
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * R_LD + col] = acc[i][r];
    lds_barrier();
    // Pass 1a: + element tables, back into LDS -- global LOADS only.  gfx9 counts loads and stores on one in-order counter (vmcnt), so a
    // table load issued behind an h1pre store cannot be waited for without waiting for that store as well: with the stores in this
    // loop every pair of rows exposed a full HBM write round trip.  Pass 1b below has the stores and no loads.
#pragma unroll 4
    for (int rr = 0; rr < RT / 4; ++rr) {            // wave w owns rows 16w .. 16w+15
      const int row = wave * (RT / 4) + rr;
      const int zz = zbuf[row];
      float* p = bufB + row * R_LD + 2 * lane;
      float2 v = *reinterpret_cast<const float2*>(p);
      // element tables in DOUBLE, the sum rounded once (round 3): a float32 table entry is off by a fixed 3e-8 relative, the same for every
      // edge of that element pair -- a pattern the LayerNorm does not remove, measured as a -2e-8 gain on the radial output
      const double2 a = *reinterpret_cast<const double2*>(ts + (zz & 0xffff) * RH + 2 * lane);
      const double2 b = *reinterpret_cast<const double2*>(tt + (zz >> 16) * RH + 2 * lane);
      const double sx = (double)v.x + (a.x + b.x), sy = (double)v.y + (a.y + b.y);
      v.x = (float)sx; v.y = (float)sy;
      *reinterpret_cast<float2*>(p) = v;
      // what the rounding dropped, for the LayerNorm below (bufA: the gaussian tile is dead, every wave is past the barrier behind fc1)
      *reinterpret_cast<float2*>(bufA + row * R_LD + 2 * lane) = make_float2((float)(sx - (double)v.x), (float)(sy - (double)v.y));
    }
#pragma unroll 2
    for (int rr = 0; rr < RT / 4; ++rr) {            // pass 1b (same rows, same wave: no barrier): h1pre out (whole 512-B rows), LN + SiLU in place
      const int row = wave * (RT / 4) + rr;
      float* p = bufB + row * R_LD + 2 * lane;
      const float2 v = *reinterpret_cast<const float2*>(p);
      const float2 vlo = *reinterpret_cast<const float2*>(bufA + row * R_LD + 2 * lane);
      if (e0 + row < ne) *reinterpret_cast<float2*>(h1pre + (e0 + row) * RH + 2 * lane) = v;
      *reinterpret_cast<float2*>(p) = ln_silu_row_hl(v, vlo, l1w, l1b);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bias2;        // the chain STARTS from the bias: added last it is "grid value + constant", one rounding error for every edge (umx_gemm_q.h)
    rad_mma<RH / 8, R_LD, TR>(bufB, W2, acc, l31, h);
    // epilogue 2: fc2 tile + bias to bufA (the gaussian tile is dead: every wave passed the barriers behind fc1)
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * R_LD + col] = acc[i][r];
    lds_barrier();
#pragma unroll 2
    for (int rr = 0; rr < RT / 4; ++rr) {            // h2pre out, LN + SiLU -> the fc3 operand
      const int row = wave * (RT / 4) + rr;
      const long e = e0 + row;
      const float2 v = *reinterpret_cast<const float2*>(bufA + row * R_LD + 2 * lane);
      const float2 o = ln_silu_row(v, l2w, l2b);
      if (e < ne) {
        *reinterpret_cast<float2*>(h2pre + e * RH + 2 * lane) = v;
        if (OUTQ) q_store2<(OUTQ == 4 ? 3 : 1)>(reinterpret_cast<unsigned short*>(out), e, RH, 2 * lane, row_sign(e, odd_sign) * o.x, row_sign(e, odd_sign) * o.y);
        else *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + e * RH + 2 * lane) = o;
      }
    }
  }
}

// ---- reverse: dE/d(fc3 operand) -> dE/dd (accumulated into dedd) ------------------------------------------------------------------
// ga2 = output of the fc3^T GEMM; w2T = W2^T ([k][j]), w1gT = W1g^T ([64 gaussians][128]).
__global__ __launch_bounds__(256, 2) void k_radial_tail(const float* __restrict__ ga2, const float* __restrict__ h2pre,
                                                        const float* __restrict__ h1pre, const float* __restrict__ evec, double gcoef,
                                                        const double* __restrict__ gmu, const float* __restrict__ ln2w,
                                                        const float* __restrict__ ln2b, const float* __restrict__ ln1w,
                                                        const float* __restrict__ ln1b, const float* __restrict__ w2T,
                                                        const float* __restrict__ w1gT, float* __restrict__ dedd, long ne) {
  constexpr int TR = R_TR, RT = 32 * TR;
  __shared__ __attribute__((aligned(16))) float bufA[RT * R_LD];
  __shared__ __attribute__((aligned(16))) float bufB[RT * R_LD];
  __shared__ float dbuf[RT];
  __shared__ float part[2][RT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int col = wave * 32 + l31;
  const int gi = wave >> 1, gj = wave & 1;           // last linear (128 -> 64): wave -> (row tile, column tile)
  float4 W2T[RH / 8], W1T[RH / 8];
#pragma unroll
  for (int c = 0; c < RH / 8; ++c) W2T[c] = *reinterpret_cast<const float4*>(w2T + col * RH + c * 8 + 4 * h);
#pragma unroll
  for (int c = 0; c < RH / 8; ++c) W1T[c] = *reinterpret_cast<const float4*>(w1gT + (gj * 32 + l31) * RH + c * 8 + 4 * h);
  const float2 l1w = *reinterpret_cast<const float2*>(ln1w + 2 * lane), l1b = *reinterpret_cast<const float2*>(ln1b + 2 * lane);
  const float2 l2w = *reinterpret_cast<const float2*>(ln2w + 2 * lane), l2b = *reinterpret_cast<const float2*>(ln2b + 2 * lane);
  const double mu_col = gmu[gj * 32 + l31];
  const long ntiles = (ne + RT - 1) / RT;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long e0 = tile * RT;
    lds_barrier();
    if (tid < RT) dbuf[tid] = evec[(e0 + tid < ne ? e0 + tid : ne - 1) * 4 + 3];
#pragma unroll 4
    for (int rr = 0; rr < RT / 4; ++rr) {            // LN2 + SiLU backward: g_h2 -> bufA
      const int row = wave * (RT / 4) + rr;
      const long e = e0 + row < ne ? e0 + row : ne - 1;
      const float2 go = *reinterpret_cast<const float2*>(ga2 + e * RH + 2 * lane);
      const float2 x = *reinterpret_cast<const float2*>(h2pre + e * RH + 2 * lane);
      *reinterpret_cast<float2*>(bufA + row * R_LD + 2 * lane) = ln_silu_row_bwd(go, x, l2w, l2b);
    }
    lds_barrier();
    f32x16 acc[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    rad_mma<RH / 8, R_LD, TR>(bufA, W2T, acc, l31, h);   // g_a1 = g_h2 . W2
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * R_LD + col] = acc[i][r];
    lds_barrier();
#pragma unroll 4
    for (int rr = 0; rr < RT / 4; ++rr) {            // LN1 + SiLU backward: g_h1 -> bufA (its fc2^T reads finished at the barrier above)
      const int row = wave * (RT / 4) + rr;
      const long e = e0 + row < ne ? e0 + row : ne - 1;
      const float2 go = *reinterpret_cast<const float2*>(bufB + row * R_LD + 2 * lane);
      const float2 x = *reinterpret_cast<const float2*>(h1pre + e * RH + 2 * lane);
      *reinterpret_cast<float2*>(bufA + row * R_LD + 2 * lane) = ln_silu_row_bwd(go, x, l1w, l1b);
    }
    lds_barrier();
    if (gi < TR) {   // g_gauss = g_h1 . W1g (128 -> 64): one 32 x 32 tile per wave (wave-uniform), then dE/dd = sum_k g_gauss[k] d/dd exp(gcoef (d - mu_k)^2)
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
      for (int c = 0; c < RH / 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(bufA + (gi * 32 + l31) * R_LD + c * 8 + 4 * h);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, W1T[c].x, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, W1T[c].y, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, W1T[c].z, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, W1T[c].w, a1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = gi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const double t = (double)dbuf[row] - mu_col;
        // sum over the 32 columns of this half-wave (two 16-lane rows), for both halves at once
        const int rs = __builtin_bit_cast(int, row16_sum(a1[r] * exp_f((float)(gcoef * t * t)) * (float)(2.0 * gcoef * t)));
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rs, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(rs, 16));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rs, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(rs, 48));
        if (lane == 0) { part[gj][row] = s0; part[gj][row + 4] = s1; }      // lane 0 sees h = 0: `row` is the first half's row, +4 the second's
      }
    }
    lds_barrier();
    if (tid < RT && e0 + tid < ne) dedd[e0 + tid] += part[0][tid] + part[1][tid];
  }
}

}  // namespace umx
