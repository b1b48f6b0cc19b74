// Dev tool (not part of libumx.so): how does the matrix core round?  Signed error of accumulation chains on the 16-bit MFMAs against
// the exact float64 sum of the same products.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_bias pdb2reaction_amd/csrc/mfma_bias.hip && build/mfma_bias
// One wave computes a 32x32 tile C = sum over `steps` k-steps of A_t . B_t^T with v_mfma_f32_32x32x16_{bf16,f16}; every C element is compared with
// the double sum.  Variants: operands as drawn / A negated (does the bias follow the sign of the data or is it one-sided?) / products that all
// have the same sign (a running sum that grows) / a small product added to a large accumulator.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A, B: [steps][32 rows][16 k] as 16-bit patterns; one block = one wave; tiles indexed by blockIdx.x
template <int F16>
__global__ void k_chain(const unsigned short* A, const unsigned short* B, float* Cout, int steps, float c0) {
  const int lane = threadIdx.x, row = lane & 31, h = lane >> 5;
  const size_t tile = blockIdx.x;
  const unsigned short* a = A + tile * (size_t)steps * 512;
  const unsigned short* b = B + tile * (size_t)steps * 512;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  for (int t = 0; t < steps; ++t) {
    bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (size_t)t * 512 + row * 16 + h * 8);
    bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (size_t)t * 512 + row * 16 + h * 8);
    if (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
  // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int r = 0; r < 16; ++r) Cout[tile * 1024 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = acc[r];
}

static unsigned short to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float from_bf16(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short to_f16(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }
static float from_f16(unsigned short s) { _Float16 h; memcpy(&h, &s, 2); return (float)h; }

int main() {
  const int tiles = 512;
  std::mt19937_64 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  printf("%-34s %6s %6s  %12s %12s %12s   (errors relative to the rms of the exact result)\n", "case", "type", "steps", "mean err", "rms err", "mean|err|");
  for (int f16 = 0; f16 < 2; ++f16)
    for (int variant = 0; variant < 6; ++variant)
      for (int steps : {1, 8, 24, 48}) {
        std::vector<unsigned short> A((size_t)tiles * steps * 512), B(A.size());
        std::vector<float> Af(A.size()), Bf(A.size());
        for (size_t i = 0; i < A.size(); ++i) {
          float a = nd(rng), b = nd(rng) * 0.1f;
          if (variant == 2) { a = fabsf(a); b = fabsf(b); }                 // all products positive: the sum grows
          if (variant == 3) { a = -fabsf(a); b = fabsf(b); }                // all products negative
          if (variant == 4) { a *= 1e-3f; }                                 // small products ...
          if (variant == 5) { a *= 1e-3f; }
          unsigned short ha = f16 ? to_f16(a) : to_bf16(a), hb = f16 ? to_f16(b) : to_bf16(b);
          if (variant == 1) ha ^= 0x8000;                                   // A negated
          A[i] = ha; B[i] = hb;
          Af[i] = f16 ? from_f16(ha) : from_bf16(ha); Bf[i] = f16 ? from_f16(hb) : from_bf16(hb);
        }
        const float c0 = variant == 4 ? 1.2345678f : variant == 5 ? -1.2345678f : 0.f;   // ... onto a large positive / negative accumulator
        unsigned short *dA, *dB; float* dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, A.size() * 2)); CK(hipMalloc(&dC, (size_t)tiles * 1024 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), A.size() * 2, hipMemcpyHostToDevice));
        if (f16) k_chain<1><<<tiles, 64>>>(dA, dB, dC, steps, c0); else k_chain<0><<<tiles, 64>>>(dA, dB, dC, steps, c0);
        CK(hipDeviceSynchronize());
        std::vector<float> C((size_t)tiles * 1024);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double se = 0, se2 = 0, sa = 0, sr2 = 0; size_t n = 0;
        std::vector<double> ref(C.size());
        for (int tl = 0; tl < tiles; ++tl)
          for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
              double s = c0;
              for (int t = 0; t < steps; ++t)
                for (int k = 0; k < 16; ++k) s += (double)Af[((size_t)tl * steps + t) * 512 + i * 16 + k] * (double)Bf[((size_t)tl * steps + t) * 512 + j * 16 + k];
              ref[(size_t)tl * 1024 + i * 32 + j] = s; sr2 += s * s;
            }
        const double rms = std::sqrt(sr2 / ref.size());
        for (size_t i = 0; i < C.size(); ++i) { const double e = ((double)C[i] - ref[i]) / rms; se += e; se2 += e * e; sa += std::fabs(e); ++n; }
        const char* names[6] = {"random signs", "random signs, A negated", "all products positive", "all products negative", "small products onto +1.23", "small products onto -1.23"};
        printf("%-34s %6s %6d  %+12.3e %12.3e %12.3e\n", names[variant], f16 ? "f16" : "bf16", steps, se / n, std::sqrt(se2 / n), sa / n);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
      }
  return 0;
}
