// Dev tool (not part of libumx.so): how precise is the dot product INSIDE v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 operands)?
//   hipcc --offload-arch=gfx950 -O2 -o build/f8_inner_sum pdb2reaction_amd/csrc/f8_inner_sum.hip && build/f8_inner_sum
// Element C[0][0] of ONE instruction: row 0 of A = (2^p, s, s, ..., s), column 0 of B = (1, 1, ..., 1): exact result c0 + 2^p + n s.  The n small
// products sit either in the same 32-k half as the big one (lane 0) or in the other half (lane 32).  What comes back shows how many bits
// below the largest product survive the instruction's internal alignment, and in which direction the rest is rounded.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_one(const unsigned char* A, const unsigned char* B, float* out, float c0) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  const int* pa = reinterpret_cast<const int*>(A + lane * 32);
  const int* pb = reinterpret_cast<const int*>(B + lane * 32);
  for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 1, 1, 0, 127, 0, 127);      // bf8 x bf8, both scales 2^0
  // EVERY lane stores (and the operands stay alive): with `if (lane == 0) out[0] = acc[0]` alone the compiler sinks the matrix instruction into
  // the branch, it then runs with one active lane and the operand bytes of the other lanes read as zero -- a probe artefact, not the hardware
  out[lane] = acc[0];
  int live = 0;
  for (int i = 0; i < 8; ++i) live ^= a[i] ^ b[i];
  if (live == 0x5a5a5a5a) out[64] = 1.f;
}

static unsigned char enc(double v) {    // e5m2 of a power of two times {1, 1.25, 1.5, 1.75}
  const unsigned char s = v < 0 ? 0x80 : 0;
  v = std::fabs(v);
  if (v == 0) return s;
  int e; const double f = std::frexp(v, &e);
  return s | (unsigned char)(((e - 1 + 15) << 2) | (int)((f * 2 - 1) * 4 + 0.5));
}

int main() {
  unsigned char *dA, *dB; float* dout;
  CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dout, 65 * 4));
  for (int other_half = 0; other_half < 2; ++other_half)
    for (double small : {1.0, -1.0})
      for (float c0 : {0.f, 1048576.f}) {
        printf("small products %+g x %d in the %s half, accumulator starts at %g\n", small, other_half ? 32 : 31, other_half ? "OTHER" : "same", c0);
        for (int p = 0; p <= 24; p += 2) {
          unsigned char hA[2048], hB[2048];
          memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB);
          hA[0] = enc(std::ldexp(1.0, p / 2)); hB[0] = enc(std::ldexp(1.0, p - p / 2));      // k = 0: lane 0, byte 0
          int n = 0;
          for (int kk = 1; kk < 64; ++kk) {
            const bool second = kk >= 32;
            if (second != (other_half != 0)) continue;
            const int off = (second ? 32 : 0) * 32 + kk % 32;                                 // lane 32 holds k = 32 ... 63 of row 0
            hA[off] = enc(small); hB[off] = enc(1.0); ++n;
          }
          CK(hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice));
          float r = -1.f;
          CK(hipMemset(dout, 0, 65 * 4));
          k_one<<<1, 64>>>(dA, dB, dout, c0);
          CK(hipDeviceSynchronize());
          CK(hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost));
          const double want = (double)c0 + std::ldexp(1.0, p) + n * small;
          printf("   big 2^%-2d + %d x %+g : got %-12.1f exact %-12.1f  small part that arrived: %+.1f of %+.1f\n", p, n, small, r, want, r - (double)c0 - std::ldexp(1.0, p), n * small);
        }
      }
  return 0;
}
