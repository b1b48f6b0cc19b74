// Dev tool (not part of libumx.so): what would the two third-plane products cost on the block-scaled 8/6-bit matrix instruction?
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_rate pdb2reaction_amd/csrc/mfma_rate.hip && build/mfma_rate
// The 24-bit plane split needs six products per k-step (umx_gemm_q.h).  Two of them (a0.b2, a2.b0) only carry ~2 significant bits of the
// result when the planes are fp16 (11 + 11 + 2 bits), so they would fit v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 / bf6 operands, E8M0 scale 2^-22
// on the third plane, the same fp32 accumulator).  This program times register-resident loops that issue, per K = 64 chunk of a 64x64 wave
// tile, exactly the matrix instructions of each plan on random operands -- the rate and the power behaviour of the pipe, no memory system:
//   mode 0  six bf16 products                       (24 x v_mfma_f32_32x32x16_bf16 per accumulator)          = today's bf16x3
//   mode 1  four fp16 products + two bf8 products   (16 x v_mfma_f32_32x32x16_f16 + 2 x ..._f8f6f4 bf8)
//   mode 2  four fp16 products + two bf6 products   (16 x f16 + 2 x ..._f8f6f4 bf6)
//   mode 3  four fp16 products                      (16 x f16)                                                = today's fast forward pass
//   mode 4  mode 0 on zero operands                 (how much of the rate is the power limit)
//   mode 5 / 6  only the scaled instruction, 64 per chunk, bf8 / bf6  (its issue rate; with `blocks` small the clock is not power-limited)
//   usage: mfma_rate [chunks = 40000] [repeats = 3] [blocks = 512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ops16: [2 sides][2 tiles][3 planes][64 lanes] of 8 x 16 bit; ops8: [2 sides][2 tiles][2 kinds][64 lanes] of 8 x 32 bit
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(const bf16x8* ops16, const i32x8* ops8, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[2][3], b[2][3];
  i32x8 a8[2][2], b8[2][2];
  for (int i = 0; i < 2; ++i)
    for (int p = 0; p < 3; ++p) { a[i][p] = ops16[((0 * 2 + i) * 3 + p) * 64 + lane]; b[i][p] = ops16[((1 * 2 + i) * 3 + p) * 64 + lane]; }
  for (int i = 0; i < 2; ++i)
    for (int p = 0; p < 2; ++p) { a8[i][p] = ops8[((0 * 2 + i) * 2 + p) * 64 + lane]; b8[i][p] = ops8[((1 * 2 + i) * 2 + p) * 64 + lane]; }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (MODE >= 5) {
          } else if (MODE == 0 || MODE == 4) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][0]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][1]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][0]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][1]), acc[i][j], 0, 0, 0);
          }
        }
    if (MODE == 1 || MODE == 2 || MODE >= 5) {
      constexpr int FMT = (MODE == 1 || MODE == 5) ? 1 : 3;      // 1 = bf8 (e5m2), 3 = bf6 (e3m2)
#pragma unroll
      for (int rep = 0; rep < (MODE >= 5 ? 8 : 1); ++rep)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i][0], b8[j][1], acc[i][j], FMT, FMT, 0, 127, 0, 127 - 22);
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i][1], b8[j][0], acc[i][j], FMT, FMT, 0, 127 - 22, 0, 127);
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;       // keeps the loop alive, (almost) never stores
}

static unsigned short to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static unsigned short to_f16(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }

template <int MODE>
static double run(const bf16x8* d16, const i32x8* d8, float* dout, int iters, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<MODE><<<blocks, 256>>>(d16, d8, dout, iters / 8);              // warm
  CK(hipEventRecord(e0));
  k_rate<MODE><<<blocks, 256>>>(d16, d8, dout, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 40000, reps = argc > 2 ? atoi(argv[2]) : 3, blocks = argc > 3 ? atoi(argv[3]) : 512;
  std::mt19937_64 rng(11);
  std::normal_distribution<float> nd(0.f, 1.f);
  // three planes of random float32 values: bf16 split for mode 0, fp16 split for the others; random bytes for the 8/6-bit operands
  // (the exponent fields are kept mid-range so no product is inf / nan)
  std::vector<unsigned short> bf(2 * 2 * 3 * 64 * 8), fh(bf.size()), zero(bf.size(), 0);
  for (int side = 0; side < 2; ++side)
    for (int t = 0; t < 2; ++t)
      for (int l = 0; l < 64 * 8; ++l) {
        float v = nd(rng) * (side ? 0.05f : 1.f), r = v;
        for (int p = 0; p < 3; ++p) {
          unsigned short h = to_bf16(r); uint32_t u = (uint32_t)h << 16; float back; memcpy(&back, &u, 4);
          bf[((side * 2 + t) * 3 + p) * 512 + l] = h; r -= back;
        }
        r = v;
        for (int p = 0; p < 3; ++p) {
          _Float16 h = (_Float16)r; fh[((side * 2 + t) * 3 + p) * 512 + l] = to_f16(r); r -= (float)h;
        }
      }
  std::vector<uint32_t> o8(2 * 2 * 2 * 64 * 8);
  for (auto& w : o8) {
    uint32_t x = 0;
    for (int k = 0; k < 4; ++k) { uint32_t byte = ((rng() & 1) << 7) | ((12 + (rng() % 6)) << 2) | (rng() & 3); x |= byte << (8 * k); }   // e5m2: sign, exp 12..17, 2 mantissa bits
    w = x;
  }
  bf16x8 *dbf, *dfh, *dz; i32x8* d8; float* dout;
  CK(hipMalloc(&dbf, bf.size() * 2)); CK(hipMalloc(&dfh, bf.size() * 2)); CK(hipMalloc(&dz, bf.size() * 2)); CK(hipMalloc(&d8, o8.size() * 4));
  CK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
  CK(hipMemcpy(dbf, bf.data(), bf.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dfh, fh.data(), bf.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dz, zero.data(), bf.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d8, o8.data(), o8.size() * 4, hipMemcpyHostToDevice));
  const double waves = (double)blocks * 4, alg = waves * iters * 4.0 * 2.0 * 32 * 32 * 64;     // algorithmic flops: one product per accumulator
  printf("%d blocks x 4 waves, %d K=64 chunks of a 64x64 wave tile each; algorithmic = ONE product per element\n", blocks, iters);
  printf("%-52s %10s %14s %14s\n", "plan", "ms", "alg. TFLOP/s", "vs bf16x3");
  for (int rep = 0; rep < reps; ++rep) {
    const double t0 = run<0>(dbf, d8, dout, iters, blocks), t1 = run<1>(dfh, d8, dout, iters, blocks), t2 = run<2>(dfh, d8, dout, iters, blocks),
                 t3 = run<3>(dfh, d8, dout, iters, blocks), t4 = run<4>(dz, d8, dout, iters, blocks), t5 = run<5>(dfh, d8, dout, iters, blocks),
                 t6 = run<6>(dfh, d8, dout, iters, blocks);
    const char* names[7] = {"0: six bf16 products (today)", "1: four fp16 + two bf8 (scaled, K = 64)", "2: four fp16 + two bf6 (scaled, K = 64)", "3: four fp16 products only",
                            "4: six bf16 products on ZERO operands", "5: 64 scaled bf8 instructions (K = 64) only", "6: 64 scaled bf6 instructions (K = 64) only"};
    const double ts[7] = {t0, t1, t2, t3, t4, t5, t6};
    for (int m = 0; m < 7; ++m) printf("%-52s %10.2f %14.1f %14.3f\n", names[m], ts[m], alg / ts[m] * 1e-9, ts[m] / t0);
  }
  return 0;
}
