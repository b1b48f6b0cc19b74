// umx_kernels_pl.h -- producer kernels of the split-precision path: every A operand of the large SO(2)/radial GEMMs is
// written ONCE, already split into 16-bit planes.  Reverse-pass operands (and the dev layout UMX_Q3=0) use the plane-interleaved
// row layout of umx_gemm_pl.h:
//   element (row, k, plane q) -> row * (K*P) + (k / 32) * (32*P) + q * 32 + (k % 32)        P = 2 bf16 planes (P = 3 forward, dev)
// forward operands the quad-row layouts of umx_gemm_q.h (QFmt below: two fp16 planes by default, three bf16 planes in split-bf16 mode).
// Fusions relative to the fp32 path: the radial modulation is applied by the gather/rotate producer (the conv-1 GEMM
// no longer reads `rad`), and the reverse pass re-derives the rotated message inside the modulation-backward kernel
// instead of round-tripping a 9 KB/edge buffer through HBM.
#pragma once
#include "umx_common.h"

namespace umx {

// SIGN-ALTERNATING ROWS (round 4).  The adder of the 16-bit matrix cores aligns the products of one MFMA against the fp32 accumulator and
// drops the bits below ~2^-32 of the largest addend with a FLOOR (toward -infinity, whatever the signs: csrc/mfma_bias.hip) -- a one-sided
// error of a few 1e-9 of the accumulator per MFMA, i.e. -1e-8 ... -3e-8 relative on a GEMM output after 50-300 MFMAs, the SAME sign for
// every edge: a systematic energy error that grows like N.  floor(-S) = -ceil(S): with every operand row of ODD edge index stored negated
// (`odd_sign` = -1 in the producers below) and the sign restored in the GEMM epilogue (GemmPL::odd_sign), the error of odd rows has the
// opposite sign, and what was a bias is now noise that cancels between neighbouring edges.  Free: a sign flip before the split, a
// multiply by +-1 in the epilogue.
__device__ __forceinline__ float row_sign(long e, float odd_sign) { return (e & 1) ? odd_sign : 1.0f; }

template <int P> __device__ __forceinline__ long pl_index(int k) { return (long)(k >> 5) * (32 * P) + (k & 31); }

// split 2 adjacent values into P planes and store them (k even): 4-byte store per plane
// dev switch (-DUMX_NT=1): non-temporal stores for the write-once operand planes (A/B measurement in NOTES.md section 9)
#ifndef UMX_NT
#define UMX_NT 0
#endif
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_stream(unsigned int* p, unsigned int v) {
  if (UMX_NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
  if (UMX_NT) __builtin_nontemporal_store(u32x2_t{v.x, v.y}, reinterpret_cast<u32x2_t*>(p)); else *p = v;
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
  if (UMX_NT) __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p)); else *p = v;
}
template <int P> __device__ __forceinline__ void pl_store2(unsigned short* row, int k, float x0, float x1) {
  unsigned short* d = row + pl_index<P>(k);
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
    const unsigned int pk = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
    st_stream(reinterpret_cast<unsigned int*>(d + q * 32), pk);
    x0 -= (float)h0; x1 -= (float)h1;
  }
}
// split 4 adjacent values (k multiple of 4): 8-byte store per plane
template <int P> __device__ __forceinline__ void pl_store4(unsigned short* row, int k, float4 v) {
  unsigned short* d = row + pl_index<P>(k);
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < P; ++q) {
    unsigned short hb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const __bf16 h = (__bf16)x[c]; hb[c] = __builtin_bit_cast(unsigned short, h); x[c] -= (float)h; }
    st_stream(reinterpret_cast<uint2*>(d + q * 32), make_uint2((unsigned int)hb[0] | ((unsigned int)hb[1] << 16), (unsigned int)hb[2] | ((unsigned int)hb[3] << 16)));
  }
}

// Quad-row layouts of the FORWARD operands (umx_gemm_q.h), two formats:
//   FMT 0 ("Q3"):  three bf16 planes of x (exact 24-bit split), 384-B blocks of 4 rows x 16 columns x 3 planes
//   FMT 1 ("Q2H"): two IEEE-half planes of QF16_SCALE * x (hi = RNE, lo = RNE(residual): 22 significant bits, and below
//                  2^-14 / QF16_SCALE a fixed absolute resolution of 2^-25 / QF16_SCALE -- half subnormals are kept by the
//                  conversion and by the MFMA), 256-B blocks.  |x| >= 65520 / QF16_SCALE converts to inf, the low plane to -inf
//                  and the GEMM output to NaN: a range violation cannot pass silently (the energy comes out non-finite).
// element (row, k, plane q) of a matrix with `cols` columns -> byte ((row/4) * (cols/16) + k/16) * 128 P + (row%4) * 32 P + q * 32 + (k%16) * 2.
constexpr float QF16_SCALE = 16.f;
//   FMT 3 ("QF"):  plain float32, 256-B blocks of 4 rows x 16 columns (row r of a block = 64 B): the A operand of the bf16x3 GEMMs since
//                  round 4 -- umx_gemm_q.h (AF = 1) splits it into the three bf16 planes in registers, bit for bit the planes of FMT 0,
//                  at 4 B instead of 6 B per element through HBM, L2 and LDS.
template <int FMT> struct QFmt { static constexpr int P = FMT ? 2 : 3; static constexpr int BLK = 128 * P; static constexpr int ROWB = 32 * P; static constexpr bool F32 = (FMT == 3); };
// LDS staging of one row piece (16 columns of one row of a block = 8 P dwords at `rowbase`): two / four adjacent values at column k15 of the piece
template <int FMT> __device__ __forceinline__ void q_split2(float x0, float x1, unsigned int (&out)[QFmt<FMT>::P]);
template <int FMT> __device__ __forceinline__ void q_stage2(unsigned int* rowbase, int k15, float x0, float x1) {
  if constexpr (QFmt<FMT>::F32) *reinterpret_cast<uint2*>(rowbase + k15) = make_uint2(__builtin_bit_cast(unsigned int, x0), __builtin_bit_cast(unsigned int, x1));
  else {
    unsigned int w[QFmt<FMT>::P];
    q_split2<FMT>(x0, x1, w);
#pragma unroll
    for (int q = 0; q < QFmt<FMT>::P; ++q) rowbase[(k15 >> 1) + q * 8] = w[q];
  }
}
template <int FMT> __device__ __forceinline__ void q_stage4(unsigned int* rowbase, int k15, const float (&x)[4]) {
  if constexpr (QFmt<FMT>::F32)
    *reinterpret_cast<uint4*>(rowbase + k15) = make_uint4(__builtin_bit_cast(unsigned int, x[0]), __builtin_bit_cast(unsigned int, x[1]), __builtin_bit_cast(unsigned int, x[2]), __builtin_bit_cast(unsigned int, x[3]));
  else {
    unsigned int wa[QFmt<FMT>::P], wb[QFmt<FMT>::P];
    q_split2<FMT>(x[0], x[1], wa); q_split2<FMT>(x[2], x[3], wb);
#pragma unroll
    for (int q = 0; q < QFmt<FMT>::P; ++q) *reinterpret_cast<uint2*>(rowbase + (k15 >> 1) + q * 8) = make_uint2(wa[q], wb[q]);
  }
}
// two adjacent values -> one packed dword per plane
template <int FMT> __device__ __forceinline__ void q_split2(float x0, float x1, unsigned int (&out)[QFmt<FMT>::P]) {
  if constexpr (FMT == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
      out[q] = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
      x0 -= (float)h0; x1 -= (float)h1;
    }
  } else {
    x0 *= QF16_SCALE; x1 *= QF16_SCALE;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    out[0] = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
    out[1] = (unsigned int)__builtin_bit_cast(unsigned short, l0) | ((unsigned int)__builtin_bit_cast(unsigned short, l1) << 16);
  }
}
template <int FMT> __device__ __forceinline__ unsigned short* q_ptr(unsigned short* base, long row, int cols, int k) {
  return reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(base) + ((row >> 2) * (cols >> 4) + (k >> 4)) * QFmt<FMT>::BLK +
                                           (row & 3) * QFmt<FMT>::ROWB + (k & 15) * (QFmt<FMT>::F32 ? 4 : 2));
}
template <int FMT> __device__ __forceinline__ void q_store2(unsigned short* base, long row, int cols, int k, float x0, float x1) {
  unsigned short* d = q_ptr<FMT>(base, row, cols, k);
  if constexpr (QFmt<FMT>::F32) { st_stream(reinterpret_cast<uint2*>(d), make_uint2(__builtin_bit_cast(unsigned int, x0), __builtin_bit_cast(unsigned int, x1))); return; }
  unsigned int w[QFmt<FMT>::P];
  q_split2<FMT>(x0, x1, w);
#pragma unroll
  for (int q = 0; q < QFmt<FMT>::P; ++q) st_stream(reinterpret_cast<unsigned int*>(d + q * 16), w[q]);
}
template <int FMT> __device__ __forceinline__ void q_store4(unsigned short* base, long row, int cols, int k, float4 v) {
  unsigned short* d = q_ptr<FMT>(base, row, cols, k);
  if constexpr (QFmt<FMT>::F32) { st_stream(reinterpret_cast<uint4*>(d), make_uint4(__builtin_bit_cast(unsigned int, v.x), __builtin_bit_cast(unsigned int, v.y), __builtin_bit_cast(unsigned int, v.z), __builtin_bit_cast(unsigned int, v.w))); return; }
  unsigned int a[QFmt<FMT>::P], b[QFmt<FMT>::P];
  q_split2<FMT>(v.x, v.y, a); q_split2<FMT>(v.z, v.w, b);
#pragma unroll
  for (int q = 0; q < QFmt<FMT>::P; ++q) st_stream(reinterpret_cast<uint2*>(d + q * 16), make_uint2(a[q], b[q]));
}

#define UMX_WAVE_ITEM_PL(idx, count)                                                  \
  const int lane = threadIdx.x & 63;                                                  \
  const long idx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))); \
  if (idx >= (count)) return;

// same, but consecutive work items go to the SAME XCD (blocks are dealt round-robin over the 8 XCDs, each with its own L2):
// XCD x walks the contiguous range [x * per, (x+1) * per) -- rows gathered by neighbouring items are then shared in one L2.
// The grid must be a multiple of 8 blocks (nblk8()).
#define UMX_WAVE_ITEM_PL_XCD(idx, count)                                              \
  const int lane = threadIdx.x & 63;                                                  \
  const long _per = gridDim.x >> 3;                                                   \
  const long _blk = (long)(blockIdx.x & 7) * _per + (blockIdx.x >> 3);                \
  const long idx = __builtin_amdgcn_readfirstlane((int)(_blk * 4 + (threadIdx.x >> 6))); \
  if (idx >= (count)) return;

// grid-stride forms of the two macros above (see UMX_WAVE_LOOP in umx_kernels.h): body = loop body, any grid size.  The XCD form
// keeps the contiguous-range-per-XCD mapping in terms of VIRTUAL blocks: virtual block v = blockIdx.x + k * gridDim.x has
// v & 7 == blockIdx.x & 7 (gridDim.x is a multiple of 8), so a physical workgroup only ever works for its own XCD's range.
#define UMX_WAVE_LOOP_PL_XCD(idx, count)                                              \
  const int lane = threadIdx.x & 63;                                                  \
  const long _nvb = ((((count) + 3) / 4 + 7) / 8) * 8;                                \
  const long _per = _nvb >> 3;                                                        \
  for (long _vb = blockIdx.x; _vb < _nvb; _vb += gridDim.x)                           \
    if (const long idx = __builtin_amdgcn_readfirstlane((int)(((_vb & 7) * _per + (_vb >> 3)) * 4 + (threadIdx.x >> 6))); idx < (count))

// K7a for the quad-row operand layouts (umx_gemm_q.h; FMT as QFmt).  A workgroup = the four edges of one row group.  Per m-primary
// row r the four waves put their 256 modulated columns x P planes into LDS in exactly the byte order of the 16 consecutive blocks that
// hold (row group, columns r*256 ... r*256+255), and the whole workgroup then writes those 6 KB (4 KB) with coalesced 16-B stores
// (direct stores from the compute layout would be 32-B pieces at a 384-B stride: measured 60 ms instead of 42 ms per iteration).
template <int FMT>
__global__ __launch_bounds__(256) void k_gather_rotate_mod_q3(const float* __restrict__ xn, const int* __restrict__ esrc,
                                                              const int* __restrict__ edst, const float* __restrict__ frame,
                                                              const float* __restrict__ rad, unsigned short* __restrict__ y1, long ne, float odd_sign) {
  constexpr int P = QFmt<FMT>::P, BLK = QFmt<FMT>::BLK;
  __shared__ __attribute__((aligned(16))) unsigned int stage[2][16][4][8 * P];   // [buffer][16-column block][row in group][q*8 + pair]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nvb = (((ne + 3) / 4 + 7) / 8) * 8;                                 // virtual blocks = row groups, padded to the 8 XCDs
  const long per = nvb >> 3;                                                     // XCD-contiguous groups (see UMX_WAVE_LOOP_PL_XCD)
  for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {                        // grid-stride: any grid size (multiple of 8) works
  const long grp = (vb & 7) * per + (vb >> 3);
  const long e0 = grp * 4;
  if (e0 >= ne) continue;                                                        // block-uniform
  __syncthreads();                                                               // the previous virtual block's last stage buffer is free
  const long e = e0 + wave;
  const bool valid = e < ne;
  const int c0 = lane * 2;
  const float sg = row_sign(e, odd_sign);
  float ps[9], qs[9], pd[9], qd[9];
  const float* rd = rad + (valid ? e : e0) * RAD;
  {
    const long ee = valid ? e : e0;
    const float* f = frame + ee * FRAME;
    const long js = esrc[ee], jd = edst[ee];
    float sx[9], sy[9], dx[9], dy[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const float2 a = *reinterpret_cast<const float2*>(xn + js * ROW + r * C + c0);
      const float2 b = *reinterpret_cast<const float2*>(xn + jd * ROW + r * C + c0);
      sx[r] = a.x; sy[r] = a.y; dx[r] = b.x; dy[r] = b.y;
    }
    rot_fwd(f, sx, ps); rot_fwd(f, sy, qs); rot_fwd(f, dx, pd); rot_fwd(f, dy, qd);
  }
  const int ridx[9] = {0, 1, 2, 3, 4, 3, 4, 5, 5};   // radial row of each m-primary row
  unsigned char* gbase = reinterpret_cast<unsigned char*>(y1) + grp * (long)(XROT / 16) * BLK;
  auto put = [&](int buf, int col, float x0, float x1) {
    q_stage2<FMT>(&stage[buf][col >> 4][wave][0], col & 15, x0, x1);
  };
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int buf = r & 1;
    const float2 ms = *reinterpret_cast<const float2*>(rd + ridx[r] * 2 * C + c0);
    const float2 md = *reinterpret_cast<const float2*>(rd + ridx[r] * 2 * C + C + c0);
    put(buf, c0, sg * (ps[r] * ms.x), sg * (qs[r] * ms.y));          // (the product rounded as before, then the sign: bitwise the old planes, negated)
    put(buf, C + c0, sg * (pd[r] * md.x), sg * (qd[r] * md.y));
    __syncthreads();
    // 16 blocks x 128 P bytes = 128 P chunks of 16 B: thread t copies chunk t and (P = 3), for t < 128, chunk 256 + t
    const uint4* src = reinterpret_cast<const uint4*>(&stage[buf][0][0][0]);
    uint4* dst = reinterpret_cast<uint4*>(gbase + (long)r * 16 * BLK);
    dst[threadIdx.x] = src[threadIdx.x];
    if (P == 3 && threadIdx.x < 128) dst[256 + threadIdx.x] = src[256 + threadIdx.x];
  }
  }
}

// SO(2) gate for the quad-row operand layouts: a workgroup = 8 edges = two row groups; per m-primary row the gated 128 columns x P
// planes of the 8 edges are staged in LDS in the byte order of their 2 x 8 consecutive blocks and written with coalesced
// 16-B stores (same reason as k_gather_rotate_mod_q3).
template <int FMT>
__global__ __launch_bounds__(256) void k_gate_edge_fwd_q3(const float* __restrict__ hg, unsigned short* __restrict__ hid, long ne, float odd_sign) {
  constexpr int P = QFmt<FMT>::P, BLK = QFmt<FMT>::BLK;
  __shared__ __attribute__((aligned(16))) unsigned int stage[2][2][8][4][8 * P]; // [buffer][row group][16-column block][row][q*8 + pair]
  const long nvb = (ne + 7) / 8;
  for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {                        // grid-stride over groups of 8 edges: any grid size works
  const long e0 = vb * 8;
  __syncthreads();                                                               // the previous virtual block's last stage buffer is free
  const int le = threadIdx.x >> 5;                                               // edge within the workgroup
  const int c = (threadIdx.x & 31) * 4;
  const long e = (e0 + le < ne) ? e0 + le : e0;                                  // tail lanes recompute edge e0 (their rows are padding)
  const float* p = hg + e * HG;
  const float4 g1 = *reinterpret_cast<const float4*>(p + c), g2 = *reinterpret_cast<const float4*>(p + H + c);
  const float4 s1 = make_float4(sigmoid_f(g1.x), sigmoid_f(g1.y), sigmoid_f(g1.z), sigmoid_f(g1.w));
  const float4 s2 = make_float4(sigmoid_f(g2.x), sigmoid_f(g2.y), sigmoid_f(g2.z), sigmoid_f(g2.w));
  unsigned char* gbase = reinterpret_cast<unsigned char*>(hid) + (e0 >> 2) * (long)(ROW / 16) * BLK;
  float4 vr[9];                                                                  // all nine rows requested up front: the barriers in the loop
#pragma unroll                                                                   // below are memory fences the compiler will not move a load across
  for (int r = 0; r < 9; ++r) vr[r] = *reinterpret_cast<const float4*>(p + 2 * H + r * H + c);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int buf = r & 1;
    const float4 v = vr[r];
    float x[4];
    if (r == 0) { x[0] = silu_f(v.x); x[1] = silu_f(v.y); x[2] = silu_f(v.z); x[3] = silu_f(v.w); }
    else {
      const bool l1 = (r == 1 || r == 3 || r == 5);
      const float4 sg = l1 ? s1 : s2;
      x[0] = v.x * sg.x; x[1] = v.y * sg.y; x[2] = v.z * sg.z; x[3] = v.w * sg.w;
    }
    const float rs = row_sign(le, odd_sign);                                      // e0 is a multiple of 8: the edge's parity is le's
    x[0] *= rs; x[1] *= rs; x[2] *= rs; x[3] *= rs;
    q_stage4<FMT>(&stage[buf][le >> 2][c >> 4][le & 3][0], c & 15, x);
    __syncthreads();
    // per row group: 8 blocks x 128 P bytes = 64 P chunks of 16 B; 128 P chunks in all
    constexpr int CPG = 64 * P;
#pragma unroll
    for (int it = 0; it < (P == 3 ? 2 : 1); ++it) {
      const int ch = threadIdx.x + 256 * it;
      const int g = ch / CPG, o = ch % CPG;
      if (ch < 2 * CPG && e0 + 4 * g < ne) {      // the second row group may lie entirely beyond the (4-row padded) buffer
        const uint4 val = reinterpret_cast<const uint4*>(&stage[buf][g][0][0][0])[o];
        st_stream(reinterpret_cast<uint4*>(gbase + (long)g * (ROW / 16) * BLK + (long)r * 8 * BLK) + o, val);
      }
    }
  }
  }
}

// backward of K7b for the SO(2) messages: g_msg[e] = env_e (W_e g[dst e]) as PL planes; dedd/tau as the fp32 kernel
template <int P>
__global__ __launch_bounds__(256) void k_rotate_back_bwd_pl(const float* __restrict__ gnode, const float* __restrict__ msg,
                                                            const float* __restrict__ frame, const int* __restrict__ edst,
                                                            unsigned short* __restrict__ gmsg, float* __restrict__ dedd,
                                                            float* __restrict__ tau, long ne, float odd_sign) {
  UMX_WAVE_LOOP_PL_XCD(e, ne) {
  const int c0 = lane * 2;
  const float* f = frame + e * FRAME;
  const long jd = edst[e];
  float gx[9], gy[9], lx[9], ly[9], mx[9], my[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float2 t = *reinterpret_cast<const float2*>(gnode + jd * ROW + r * C + c0);
    gx[r] = t.x; gy[r] = t.y;
    const float2 m = *reinterpret_cast<const float2*>(msg + e * ROW + r * C + c0);
    mx[r] = m.x; my[r] = m.y;
  }
  rot_fwd(f, gx, lx); rot_fwd(f, gy, ly);
  const float sc = f[34];
  float s = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    s += lx[r] * mx[r] + ly[r] * my[r];
    lx[r] *= sc; ly[r] *= sc;
  }
  torque_acc(lx, mx, -1.0f, tx, ty, tz);
  torque_acc(ly, my, -1.0f, tx, ty, tz);
  unsigned short* o = gmsg + e * (long)(ROW * P);
  const float sg = row_sign(e, odd_sign);
#pragma unroll
  for (int r = 0; r < 9; ++r) pl_store2<P>(o, r * C + c0, sg * lx[r], sg * ly[r]);
  s = wave_sum(s); tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
  if (lane == 0) {
    dedd[e] += f[35] * s;
    tau[e * 4 + 0] += tx; tau[e * 4 + 1] += ty; tau[e * 4 + 2] += tz;
  }
  }
}

// The same for the quad-row (Q3, three bf16 planes) operand layout of umx_gemm_q.h -- the reverse pass of the bf16x3 mode (round 4): the
// conv-2^T GEMMs then run on the 256 x 256-tile kernel like the forward ones.  A workgroup = the four edges of one row group (one wave
// each, XCD-contiguous row groups); per m-primary row the four waves stage their 128 columns x 3 planes in LDS in the byte order of the
// 8 consecutive 384-B blocks and the workgroup writes them with coalesced 16-B stores (see k_gather_rotate_mod_q3).
template <int FMT>             // 0: three bf16 planes, 3: float32 rows (QFmt)
__global__ __launch_bounds__(256) void k_rotate_back_bwd_q3(const float* __restrict__ gnode, const float* __restrict__ msg,
                                                            const float* __restrict__ frame, const int* __restrict__ edst,
                                                            unsigned short* __restrict__ gmsg, float* __restrict__ dedd,
                                                            float* __restrict__ tau, long ne, float odd_sign) {
  constexpr int P = QFmt<FMT>::P, BLK = QFmt<FMT>::BLK;
  __shared__ __attribute__((aligned(16))) unsigned int stage[2][8][4][8 * P];    // [buffer][16-column block][row in group][q*8 + pair]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nvb = (((ne + 3) / 4 + 7) / 8) * 8;
  const long per = nvb >> 3;
  for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  const long grp = (vb & 7) * per + (vb >> 3);
  const long e0 = grp * 4;
  if (e0 >= ne) continue;                                                        // block-uniform
  __syncthreads();
  const long e = e0 + wave;
  const bool valid = e < ne;
  const long ee = valid ? e : e0;
  const int c0 = lane * 2;
  const float* f = frame + ee * FRAME;
  const long jd = edst[ee];
  float gx[9], gy[9], lx[9], ly[9], mx[9], my[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float2 t = *reinterpret_cast<const float2*>(gnode + jd * ROW + r * C + c0);
    gx[r] = t.x; gy[r] = t.y;
    const float2 m = *reinterpret_cast<const float2*>(msg + ee * ROW + r * C + c0);
    mx[r] = m.x; my[r] = m.y;
  }
  rot_fwd(f, gx, lx); rot_fwd(f, gy, ly);
  const float sc = f[34];
  float s = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    s += lx[r] * mx[r] + ly[r] * my[r];
    lx[r] *= sc; ly[r] *= sc;
  }
  torque_acc(lx, mx, -1.0f, tx, ty, tz);
  torque_acc(ly, my, -1.0f, tx, ty, tz);
  s = wave_sum(s); tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
  if (lane == 0 && valid) {
    dedd[e] += f[35] * s;
    tau[e * 4 + 0] += tx; tau[e * 4 + 1] += ty; tau[e * 4 + 2] += tz;
  }
  const float sg = row_sign(ee, odd_sign);
  unsigned char* gbase = reinterpret_cast<unsigned char*>(gmsg) + grp * (long)(ROW / 16) * BLK;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int buf = r & 1;
    q_stage2<FMT>(&stage[buf][c0 >> 4][wave][0], c0 & 15, sg * lx[r], sg * ly[r]);
    __syncthreads();
    // 8 blocks x 128 P bytes = 64 P chunks of 16 B
    if (threadIdx.x < 64 * P) {
      const uint4 val = reinterpret_cast<const uint4*>(&stage[buf][0][0][0])[threadIdx.x];
      st_stream(reinterpret_cast<uint4*>(gbase + (long)r * 8 * BLK) + threadIdx.x, val);
    }
  }
  }
}

// backward of the edge gate: ghid (9x128 fp32), hg (forward, fp32) -> g_hg = [ggate | ghpre] as PL planes (1408 columns)
template <int P>
__global__ void k_gate_edge_bwd_pl(const float* __restrict__ ghid, const float* __restrict__ hg, unsigned short* __restrict__ ghg, long ne, float odd_sign) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ne * (H / 4); i += (long)gridDim.x * blockDim.x) {   // grid-stride
  const long e = i / (H / 4);
  const float sg = row_sign(e, odd_sign);
  const int c = (int)(i % (H / 4)) * 4;
  const float* p = hg + e * HG;
  const float4 g1 = *reinterpret_cast<const float4*>(p + c), g2 = *reinterpret_cast<const float4*>(p + H + c);
  const float s1[4] = {sigmoid_f(g1.x), sigmoid_f(g1.y), sigmoid_f(g1.z), sigmoid_f(g1.w)};
  const float s2[4] = {sigmoid_f(g2.x), sigmoid_f(g2.y), sigmoid_f(g2.z), sigmoid_f(g2.w)};
  unsigned short* o = ghg + e * (long)(HG * P);
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float4 gv4 = *reinterpret_cast<const float4*>(ghid + e * ROW + r * H + c);
    const float4 hv4 = *reinterpret_cast<const float4*>(p + 2 * H + r * H + c);
    const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
    float w[4];
    const bool l1 = (r == 1 || r == 3 || r == 5);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r == 0) w[k] = gv[k] * silu_grad_f(hv[k]);
      else {
        w[k] = gv[k] * (l1 ? s1[k] : s2[k]);
        if (l1) a1[k] += gv[k] * hv[k]; else a2[k] += gv[k] * hv[k];
      }
    }
    pl_store4<P>(o, 2 * H + r * H + c, make_float4(sg * w[0], sg * w[1], sg * w[2], sg * w[3]));
  }
  pl_store4<P>(o, c, make_float4(sg * (a1[0] * s1[0] * (1.0f - s1[0])), sg * (a1[1] * s1[1] * (1.0f - s1[1])), sg * (a1[2] * s1[2] * (1.0f - s1[2])), sg * (a1[3] * s1[3] * (1.0f - s1[3]))));
  pl_store4<P>(o, H + c, make_float4(sg * (a2[0] * s2[0] * (1.0f - s2[0])), sg * (a2[1] * s2[1] * (1.0f - s2[1])), sg * (a2[2] * s2[2] * (1.0f - s2[2])), sg * (a2[3] * s2[3] * (1.0f - s2[3]))));
  }
}

// The same for the quad-row (Q3) layout (bf16x3 reverse pass): a workgroup = 8 edges = two row groups; per 128-column row (the nine
// m-primary rows of g_hpre, then the two gate rows) the values of the 8 edges are staged in LDS in block order and written with coalesced
// 16-B stores (see k_gate_edge_fwd_q3).  Column layout of g_hg as everywhere: [ggate l1 (128) | ggate l2 (128) | ghpre 9 x 128].
template <int FMT>             // 0: three bf16 planes, 3: float32 rows (QFmt)
__global__ __launch_bounds__(256) void k_gate_edge_bwd_q3(const float* __restrict__ ghid, const float* __restrict__ hg, unsigned short* __restrict__ ghg,
                                                          long ne, float odd_sign) {
  constexpr int P = QFmt<FMT>::P, BLK = QFmt<FMT>::BLK;
  __shared__ __attribute__((aligned(16))) unsigned int stage[2][2][8][4][8 * P]; // [buffer][row group][16-column block][row][q*8 + pair]
  const long nvb = (ne + 7) / 8;
  for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  const long e0 = vb * 8;
  __syncthreads();
  const int le = threadIdx.x >> 5;
  const int c = (threadIdx.x & 31) * 4;
  const long e = (e0 + le < ne) ? e0 + le : e0;
  const float* p = hg + e * HG;
  const float4 g1 = *reinterpret_cast<const float4*>(p + c), g2 = *reinterpret_cast<const float4*>(p + H + c);
  const float s1[4] = {sigmoid_f(g1.x), sigmoid_f(g1.y), sigmoid_f(g1.z), sigmoid_f(g1.w)};
  const float s2[4] = {sigmoid_f(g2.x), sigmoid_f(g2.y), sigmoid_f(g2.z), sigmoid_f(g2.w)};
  const float rs = row_sign(le, odd_sign);                                       // e0 is a multiple of 8
  unsigned char* gbase = reinterpret_cast<unsigned char*>(ghg) + (e0 >> 2) * (long)(HG / 16) * BLK;
  float4 gv4[9], hv4[9];                                                         // all loads up front (the barriers below fence loads)
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    gv4[r] = *reinterpret_cast<const float4*>(ghid + e * ROW + r * H + c);
    hv4[r] = *reinterpret_cast<const float4*>(p + 2 * H + r * H + c);
  }
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  auto emit = [&](int it, int colblk, const float (&x)[4]) {                      // colblk: first 16-column block of this 128-column row
    const int buf = it & 1;
    const float xs[4] = {rs * x[0], rs * x[1], rs * x[2], rs * x[3]};
    q_stage4<FMT>(&stage[buf][le >> 2][c >> 4][le & 3][0], c & 15, xs);
    __syncthreads();
    constexpr int CPG = 64 * P;                                                  // 16-B chunks per row group: 8 blocks x 8 P
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int ch = threadIdx.x + 256 * k2;
      const int g = ch / CPG, o = ch % CPG;
      if (ch < 2 * CPG && e0 + 4 * g < ne) {
        const uint4 val = reinterpret_cast<const uint4*>(&stage[buf][g][0][0][0])[o];
        st_stream(reinterpret_cast<uint4*>(gbase + (long)g * (HG / 16) * BLK + (long)colblk * BLK) + o, val);
      }
    }
  };
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float gv[4] = {gv4[r].x, gv4[r].y, gv4[r].z, gv4[r].w}, hv[4] = {hv4[r].x, hv4[r].y, hv4[r].z, hv4[r].w};
    float w[4];
    const bool l1 = (r == 1 || r == 3 || r == 5);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r == 0) w[k] = gv[k] * silu_grad_f(hv[k]);
      else {
        w[k] = gv[k] * (l1 ? s1[k] : s2[k]);
        if (l1) a1[k] += gv[k] * hv[k]; else a2[k] += gv[k] * hv[k];
      }
    }
    emit(r, (2 * H + r * H) / 16, w);
  }
  const float q1[4] = {a1[0] * s1[0] * (1.0f - s1[0]), a1[1] * s1[1] * (1.0f - s1[1]), a1[2] * s1[2] * (1.0f - s1[2]), a1[3] * s1[3] * (1.0f - s1[3])};
  const float q2[4] = {a2[0] * s2[0] * (1.0f - s2[0]), a2[1] * s2[1] * (1.0f - s2[1]), a2[2] * s2[2] * (1.0f - s2[2]), a2[3] * s2[3] * (1.0f - s2[3])};
  emit(9, 0, q1);
  emit(10, H / 16, q2);
  }
}

// backward of the radial modulation with the rotated message RE-DERIVED in place (gather + rotate, no HBM round trip):
// gy1 (9x256 fp32, in/out -> gxrot = gy1 .* rad), g_rad (1536) as PL planes, tau += <gxrot, L xrot>
template <int P>
__global__ __launch_bounds__(256) void k_modulate_bwd_pl(float* __restrict__ gy1, const float* __restrict__ xn,
                                                         const int* __restrict__ esrc, const int* __restrict__ edst,
                                                         const float* __restrict__ frame, const float* __restrict__ rad,
                                                         unsigned short* __restrict__ grad, float* __restrict__ tau, long ne, float odd_sign) {
  UMX_WAVE_ITEM_PL(e, ne)
  const int c0 = lane * 4;                     // column of the 256-wide [src | dst] row; lanes 0-31 own the source half
  const float* f = frame + e * FRAME;
  const long node = (lane < 32) ? esrc[e] : edst[e];
  const int cn = c0 & (C - 1);
  float4 xv[9];
  {
    float4 raw[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) raw[r] = *reinterpret_cast<const float4*>(xn + node * ROW + r * C + cn);
    float in[9], out[9];
#pragma unroll
    for (int comp = 0; comp < 4; ++comp) {
#pragma unroll
      for (int r = 0; r < 9; ++r) in[r] = comp == 0 ? raw[r].x : comp == 1 ? raw[r].y : comp == 2 ? raw[r].z : raw[r].w;
      rot_fwd(f, in, out);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (comp == 0) xv[r].x = out[r]; else if (comp == 1) xv[r].y = out[r]; else if (comp == 2) xv[r].z = out[r]; else xv[r].w = out[r];
      }
    }
  }
  float* g = gy1 + e * XROT + c0;
  const float* rd = rad + e * RAD + c0;
  float4 gv[9], rv[6], gacc[6];
#pragma unroll
  for (int r = 0; r < 9; ++r) gv[r] = *reinterpret_cast<const float4*>(g + r * 2 * C);
#pragma unroll
  for (int k = 0; k < 6; ++k) { rv[k] = *reinterpret_cast<const float4*>(rd + k * 2 * C); gacc[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
  const int ridx[9] = {0, 1, 2, 3, 4, 3, 4, 5, 5};
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = ridx[r];
    gacc[k].x += gv[r].x * xv[r].x; gacc[k].y += gv[r].y * xv[r].y; gacc[k].z += gv[r].z * xv[r].z; gacc[k].w += gv[r].w * xv[r].w;
    gv[r].x *= rv[k].x; gv[r].y *= rv[k].y; gv[r].z *= rv[k].z; gv[r].w *= rv[k].w;
    *reinterpret_cast<float4*>(g + r * 2 * C) = gv[r];
  }
  unsigned short* gr = grad + e * (long)(RAD * P);
  const float sg = row_sign(e, odd_sign);
#pragma unroll
  for (int k = 0; k < 6; ++k) pl_store4<P>(gr, k * 2 * C + c0, make_float4(sg * gacc[k].x, sg * gacc[k].y, sg * gacc[k].z, sg * gacc[k].w));
  float tx = 0.f, ty = 0.f, tz = 0.f;
  float ga[9], xa[9];
#pragma unroll
  for (int comp = 0; comp < 4; ++comp) {
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      ga[r] = comp == 0 ? gv[r].x : comp == 1 ? gv[r].y : comp == 2 ? gv[r].z : gv[r].w;
      xa[r] = comp == 0 ? xv[r].x : comp == 1 ? xv[r].y : comp == 2 ? xv[r].z : xv[r].w;
    }
    torque_acc(ga, xa, 1.0f, tx, ty, tz);
  }
  tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
  if (lane == 0) { tau[e * 4 + 0] += tx; tau[e * 4 + 1] += ty; tau[e * 4 + 2] += tz; }
}

// Reverse pass of K7a in ONE node-centric kernel (replaces k_modulate_bwd_pl + k_gather_rotate_bwd): for every node n the wave
// walks its incoming edges (n is the target: columns C..2C of the 256-wide [src | dst] rows) and its outgoing edges (n is the
// source: columns 0..C).  In both walks the un-rotated feature is the node's OWN xn[n], so the rotated message is re-derived
// in registers, and
//   g_rad[e][k]  = sum_{rows r of block k} gy1[e][r] * xrot[r]        -> PL planes (A operand of the w3^T GEMM; P = 0: float32 rows, fp32 mode)
//   g_xrot[e][r] = gy1[e][r] * rad[e][k(r)]                            (never leaves the registers: -18 KB/edge of HBM traffic)
//   tau[e]      += <g_xrot, L xrot>                                    (target half -> tau, source half -> tau2: two writers)
//   g_xn[n]      = sum_e W_e^T g_xrot[e][half]                         (deterministic segmented sum, fixed edge order)
template <int P>
__global__ __launch_bounds__(256) void k_modrot_bwd_pl(const float* __restrict__ gy1, const float* __restrict__ xn,
                                                       const float* __restrict__ frame, const float* __restrict__ rad,
                                                       const int* __restrict__ row_ptr, const int* __restrict__ out_ptr,
                                                       const int* __restrict__ out_edge, unsigned short* __restrict__ grad,
                                                       float* __restrict__ tau, float* __restrict__ tau2, float* __restrict__ gxn, long nt, float odd_sign) {
  // one BLOCK per node: its four waves take every fourth edge of the two lists (4x shorter dependent loops, 4x more loads in
  // flight) and their partial g_xn rows are added through LDS in wave order -- still a fixed summation order
  __shared__ float part[3][9][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long node = blockIdx.x;
  if (node >= nt) return;
  const int c0 = lane * 2;
  float xx[9], xy[9], ax[9], ay[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float2 t = *reinterpret_cast<const float2*>(xn + node * ROW + r * C + c0);
    xx[r] = t.x; xy[r] = t.y; ax[r] = 0.f; ay[r] = 0.f;
  }
  const int ridx[9] = {0, 1, 2, 3, 4, 3, 4, 5, 5};
  auto one_edge = [&](long e, int half, float* __restrict__ tdst) {
    const float* f = frame + e * FRAME;
    const float* g = gy1 + e * XROT + half + c0;
    const float* rd = rad + e * RAD + half + c0;
    float2 gv[9], rv[6];
#pragma unroll
    for (int r = 0; r < 9; ++r) gv[r] = *reinterpret_cast<const float2*>(g + r * 2 * C);
#pragma unroll
    for (int k = 0; k < 6; ++k) rv[k] = *reinterpret_cast<const float2*>(rd + k * 2 * C);
    float px[9], py[9];
    rot_fwd(f, xx, px); rot_fwd(f, xy, py);
    float gax[6], gay[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { gax[k] = 0.f; gay[k] = 0.f; }
    float hx[9], hy[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int k = ridx[r];
      gax[k] += gv[r].x * px[r]; gay[k] += gv[r].y * py[r];
      hx[r] = gv[r].x * rv[k].x; hy[r] = gv[r].y * rv[k].y;
    }
    if constexpr (P == 0) {      // g_rad as plain float32 rows: the A operand of the fp32-MFMA fc3^T GEMM (fp32 mode, odd_sign = +1) or of
      float* gr = reinterpret_cast<float*>(grad) + e * (long)RAD;      // umx_gemm_pl16_kernel<.., AF = 1> (bf16x3, sign-alternating rows)
      const float sg = row_sign(e, odd_sign);
#pragma unroll
      for (int k = 0; k < 6; ++k) *reinterpret_cast<float2*>(gr + k * 2 * C + half + c0) = make_float2(sg * gax[k], sg * gay[k]);
    } else {
      unsigned short* gr = grad + e * (long)(RAD * P);
      const float sg = row_sign(e, odd_sign);
#pragma unroll
      for (int k = 0; k < 6; ++k) pl_store2<(P ? P : 2)>(gr, k * 2 * C + half + c0, sg * gax[k], sg * gay[k]);
    }
    float tx = 0.f, ty = 0.f, tz = 0.f;
    torque_acc(hx, px, 1.0f, tx, ty, tz); torque_acc(hy, py, 1.0f, tx, ty, tz);
    tx = wave_sum(tx); ty = wave_sum(ty); tz = wave_sum(tz);
    if (lane == 0) { tdst[e * 4 + 0] += tx; tdst[e * 4 + 1] += ty; tdst[e * 4 + 2] += tz; }
    rot_bwd_acc(f, hx, 1.0f, ax); rot_bwd_acc(f, hy, 1.0f, ay);
  };
  for (int e = row_ptr[node] + wave; e < row_ptr[node + 1]; e += 4) one_edge((long)e, C, tau);
  for (int k = out_ptr[node] + wave; k < out_ptr[node + 1]; k += 4) one_edge((long)out_edge[k], 0, tau2);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 9; ++r) *reinterpret_cast<float2*>(&part[wave - 1][r][c0]) = make_float2(ax[r], ay[r]);
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      float sx = ax[r], sy = ay[r];
#pragma unroll
      for (int w = 0; w < 3; ++w) { const float2 t = *reinterpret_cast<const float2*>(&part[w][r][c0]); sx += t.x; sy += t.y; }
      *reinterpret_cast<float2*>(gxn + node * ROW + r * C + c0) = make_float2(sx, sy);
    }
  }
}

// tau += tau2 (the two halves of k_modrot_bwd_pl's torque are written by different waves)
__global__ void k_add4(float* __restrict__ a, const float* __restrict__ b, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 x = *reinterpret_cast<float4*>(a + i * 4);
  const float4 y = *reinterpret_cast<const float4*>(b + i * 4);
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
  *reinterpret_cast<float4*>(a + i * 4) = x;
}

}  // namespace umx
