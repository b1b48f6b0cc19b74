// umx_common.h -- shared constants and device helpers of the UMA-S engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace umx {

// UMA-S shapes (pdb2reaction_amd/weights.py)
constexpr int C = 128;            // sphere channels
constexpr int H = 128;            // hidden channels
constexpr int S = 9;              // (lmax+1)^2
constexpr int NL = 4;             // layers
constexpr int NG = 64;            // gaussian basis
constexpr int NZ = 100;           // max elements
constexpr int RH = 128;           // radial hidden
constexpr int ROW = S * C;        // 1152 floats: one node / one local-frame message
constexpr int XROT = S * 2 * C;   // 2304 floats: rotated [src|dst] message
constexpr int RAD = 1536;         // radial weights of SO(2) conv 1
constexpr int HG = 2 * H + ROW;   // 1408 floats: [gate scalars (256) | conv-1 output (9x128)]
constexpr int FRAME = 36;         // per-edge frame record: R[9], D2[25], env, denv
constexpr float NORM_EPS = 1e-5f;
constexpr float LN_EPS = 1e-5f;
constexpr float DEG_RESCALE = 5.0f;
constexpr float SQRT3 = 1.7320508075688772f;
constexpr int Q8_SHIFT = 20;      // 8-bit operand planes (umx_gemm_q.h, X8): x2' = bf8(2^Q8_SHIFT x (residual of the two half planes)),
constexpr int Q8_SHIFT1 = 10;     //                                           x1' = bf8(2^Q8_SHIFT1 x (the low half plane))

// Accurate (<= 1 ulp, unbiased) transcendentals on purpose: every atom shares the same weights, so the
// deterministic error of the fast v_exp/v_rsq approximations does not average out over atoms -- it showed up
// as a same-sign per-atom energy bias of ~1e-7 eV that grows linearly with N.  These ops live in HBM-bound
// kernels (and in GEMM prologues with ample VALU slack), so the extra instructions are free.
#ifdef UMX_EXP_DOUBLE      // dev experiment: correctly rounded transcendentals, to measure what the float ones contribute to the energy error
__device__ __forceinline__ float exp_f(float x) { return (float)exp((double)x); }
#else
__device__ __forceinline__ float exp_f(float x) { return expf(x); }
#endif
__device__ __forceinline__ float rsqrt_f(float x) { return 1.0f / sqrtf(x); }
// 1 / sqrt(x + eps) for the normalisations (round 3).  In float32 `x + 1e-5f` is a grid value plus a constant: within a binade the
// sum always lands on the same fraction of an ulp, so its rounding error is the SAME for every row -- measured +2.1e-8 relative on
// var + eps at var ~ 0.65, i.e. a -1.1e-8 gain on every LayerNorm output of the radial MLP (a -2e-8 gain on its output for every
// edge) and likewise on every RMS norm: an energy error that grows with the number of atoms.  Cure: carry the rounding error of the
// sum along (Fast2Sum: dl = eps - ((x + eps) - x) is exact for x >= eps): 1/sqrt(s + dl) = y (1 + c) with c = -dl y^2 / 2 ~ 1e-8.
// The factor cannot be folded into y -- y sits on the float grid and y (1 + c) rounds straight back to it (the same trap) -- so it
// travels with y and is applied where the row is scaled, inside ONE fma: v y (1 + c) = fma(v, y, (v y) c), rounded once.  Two
// extra instructions per element.  (A double sqrt + divide per row did the same but cost the fused radial kernels 2.7 ms per c3
// iteration: every lane of the wave computes it.)  func_bias.hip: LayerNorm + SiLU gain -1.22e-8 -> see NOTES.md section 5.
struct Rstd { float y, c; };
__device__ __forceinline__ Rstd rstd_eps(float x, float eps) {
  const float s = __fadd_rn(x, eps);                       // (explicitly rounded ops: must not be re-associated into dl = 0)
  const float dl = __fsub_rn(eps, __fsub_rn(s, x));
  const float y = 1.0f / sqrtf(s);
  return Rstd{y, -0.5f * dl * y * y};
}
__device__ __forceinline__ float scale_rstd(float v, Rstd r) { return fmaf(v, r.y, (v * r.y) * r.c); }
#ifdef UMX_EXP_DOUBLE
__device__ __forceinline__ float silu_f(float x) { return (float)((double)x / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
#else
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif
__device__ __forceinline__ float silu_grad_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// Grid-stride forms ("virtual blocks"): the kernel body is the loop body, so ANY grid size does the same work in the same per-item
// arithmetic order.  Launched with the full grid it is one iteration per wave (as UMX_WAVE_ITEM); launched with a capped grid
// (engine "throttled" mode: a few resident workgroups per CU) the kernel leaves room on every CU for the other lane's GEMM
// workgroups instead of flooding the chip (NOTES.md section 5, two-lane execution).   Usage:  UMX_WAVE_LOOP(idx, count) { body }
#define UMX_WAVE_LOOP(idx, count)                                                     \
  const int lane = threadIdx.x & 63;                                                  \
  for (long _it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); _it < (count); _it += (long)gridDim.x * 4) \
    if (const long idx = __builtin_amdgcn_readfirstlane((int)_it); true)

// Wave-wide sum on the VALU (DPP) instead of six dependent ds_bpermute round trips: the wave-per-item kernels run at few waves per
// SIMD, so the ~600-cycle latency of a __shfl_xor butterfly is NOT hidden by other waves (measured in the fused radial kernels: the
// LayerNorm passes took as long as the MFMAs).  quad_perm / row_ror adds leave every lane with the sum of its 16-lane row; the four
// row sums are combined through v_readlane.  Round 3: every wave_sum of the engine is this form (the per-edge torque sums of the
// reverse edge kernels were 18-24 ds_bpermute per edge).
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);       // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);       // quad_perm:[2,3,0,1]
  v = dpp_add<0x124>(v);      // row_ror:4
  v = dpp_add<0x128>(v);      // row_ror:8
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  const int r = __builtin_bit_cast(int, row16_sum(v));
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48)));
}
__device__ __forceinline__ float wave_sum_shfl(float v) {       // the butterfly form (UMX_WAVE_SUM_SHFL builds: A/B only)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#ifdef UMX_WAVE_SUM_SHFL
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_shfl(v); }
#else
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
#endif
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- frame application: W (rows m-primary, cols l-primary) and its transpose -------------------
// f[0..8] = R row-major (R nhat = +y), f[9..33] = D2 row-major.  m-primary row r holds l-primary
// coefficient TO_M[r] = {0,2,6,3,7,1,5,8,4}.
__device__ __forceinline__ void rot_fwd(const float* __restrict__ f, const float x[9], float v[9]) {
  v[0] = x[0];
  v[5] = f[0] * x[1] + f[1] * x[2] + f[2] * x[3];   // lp1
  v[1] = f[3] * x[1] + f[4] * x[2] + f[5] * x[3];   // lp2
  v[3] = f[6] * x[1] + f[7] * x[2] + f[8] * x[3];   // lp3
  float t[5];
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float* d = f + 9 + a * 5;
    t[a] = d[0] * x[4] + d[1] * x[5] + d[2] * x[6] + d[3] * x[7] + d[4] * x[8];
  }
  v[8] = t[0]; v[6] = t[1]; v[2] = t[2]; v[4] = t[3]; v[7] = t[4];
}
// x += scale * W^T v
__device__ __forceinline__ void rot_bwd_acc(const float* __restrict__ f, const float v[9], float scale, float x[9]) {
  x[0] += scale * v[0];
  const float u0 = scale * v[5], u1 = scale * v[1], u2 = scale * v[3];
  x[1] += f[0] * u0 + f[3] * u1 + f[6] * u2;
  x[2] += f[1] * u0 + f[4] * u1 + f[7] * u2;
  x[3] += f[2] * u0 + f[5] * u1 + f[8] * u2;
  const float w0 = scale * v[8], w1 = scale * v[6], w2 = scale * v[2], w3 = scale * v[4], w4 = scale * v[7];
  const float* d = f + 9;
#pragma unroll
  for (int b = 0; b < 5; ++b)
    x[4 + b] += d[b] * w0 + d[5 + b] * w1 + d[10 + b] * w2 + d[15 + b] * w3 + d[20 + b] * w4;
}
// per-channel torque partials tau_k += <g, L_k a> (m-primary rows); generators from
// tools/gen_generators.py (L_x, L_z; L_y only for the gauge check)
__device__ __forceinline__ void torque_acc(const float g[9], const float a[9], float sgn, float& tx, float& ty, float& tz) {
  tx += sgn * (-g[1] * a[3] + g[3] * a[1] + SQRT3 * (g[4] * a[2] - g[2] * a[4]) - g[4] * a[7] + g[7] * a[4] - g[6] * a[8] + g[8] * a[6]);
  ty += sgn * (-g[3] * a[5] + g[5] * a[3] - g[4] * a[6] + g[6] * a[4] + 2.0f * (g[8] * a[7] - g[7] * a[8]));
  tz += sgn * (g[1] * a[5] - g[5] * a[1] + SQRT3 * (g[2] * a[6] - g[6] * a[2]) + g[4] * a[8] - g[8] * a[4] - g[6] * a[7] + g[7] * a[6]);
}

}  // namespace umx
