// Dev tool (not part of libumx.so): raw results of the 16-bit matrix-core instructions on operand tiles read from a file, so that an
// arithmetic model of the adder (alignment width, truncation rule, final rounding) can be fitted OFFLINE, bit for bit
// (tools/mfma_model.py).  Round 6: the "tie rule" hypothesis of NOTES.md section 11 is tested on this probe, not on the engine.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe pdb2reaction_amd/csrc/mfma_probe.hip
//   build/mfma_probe <bf16_32|f16_32|bf16_16|f32_32> in.bin out.bin
// in.bin : int32 T, int32 steps, then A[T][steps][R][K] u16 (f32_32: float), B[T][steps][R][K] likewise, C0[T][R][R] float32
//          (R, K) = (32, 16) for *_32, (16, 32) for bf16_16, (32, 2) for f32_32 (v_mfma_f32_32x32x2_f32)
// out.bin: C[T][R][R] float32 = C0 + sum over the steps of A_t . B_t^T, accumulated by chained MFMAs (one wave per tile)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// 32x32x16: lane -> row = lane & 31, k = 8 * (lane >> 5) ... + 7; C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <int F16>
__global__ void k32(const unsigned short* A, const unsigned short* B, const float* C0, float* C, int steps) {
  const int lane = threadIdx.x, row = lane & 31, h = lane >> 5;
  const size_t tile = blockIdx.x;
  const unsigned short* a = A + tile * (size_t)steps * 512;
  const unsigned short* b = B + tile * (size_t)steps * 512;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = C0[tile * 1024 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row];
  for (int t = 0; t < steps; ++t) {
    bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (size_t)t * 512 + row * 16 + h * 8);
    bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (size_t)t * 512 + row * 16 + h * 8);
    if (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[tile * 1024 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = acc[r];
}

// 16x16x32: lane -> row = lane & 15, k = 8 * (lane >> 4) ... + 7; C/D: col = lane & 15, row = 4 * (lane >> 4) + reg
__global__ void k16(const unsigned short* A, const unsigned short* B, const float* C0, float* C, int steps) {
  const int lane = threadIdx.x, row = lane & 15, g = lane >> 4;
  const size_t tile = blockIdx.x;
  const unsigned short* a = A + tile * (size_t)steps * 512;
  const unsigned short* b = B + tile * (size_t)steps * 512;
  f32x4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C0[tile * 256 + (size_t)(4 * g + r) * 16 + row];
  for (int t = 0; t < steps; ++t) {
    bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (size_t)t * 512 + row * 32 + g * 8);
    bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (size_t)t * 512 + row * 32 + g * 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[tile * 256 + (size_t)(4 * g + r) * 16 + row] = acc[r];
}

// v_mfma_f32_32x32x2_f32: lane -> row = lane & 31, k = lane >> 5
__global__ void kf32(const float* A, const float* B, const float* C0, float* C, int steps) {
  const int lane = threadIdx.x, row = lane & 31, h = lane >> 5;
  const size_t tile = blockIdx.x;
  const float* a = A + tile * (size_t)steps * 64;
  const float* b = B + tile * (size_t)steps * 64;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = C0[tile * 1024 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row];
  for (int t = 0; t < steps; ++t)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(size_t)t * 64 + row * 2 + h], b[(size_t)t * 64 + row * 2 + h], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[tile * 1024 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = acc[r];
}

int main(int argc, char** argv) {
  if (argc != 4) { printf("usage: mfma_probe <bf16_32|f16_32|bf16_16|f32_32> in.bin out.bin\n"); return 2; }
  const std::string kind = argv[1];
  FILE* f = fopen(argv[2], "rb");
  if (!f) { printf("cannot open %s\n", argv[2]); return 1; }
  int hdr[2];
  if (fread(hdr, 4, 2, f) != 2) return 1;
  const int T = hdr[0], steps = hdr[1];
  const bool f32 = kind == "f32_32";
  const int R = kind == "bf16_16" ? 16 : 32;
  const size_t opnd_elems = (size_t)T * steps * (f32 ? 64 : 512), opnd_bytes = opnd_elems * (f32 ? 4 : 2), cn = (size_t)T * R * R;
  std::vector<char> A(opnd_bytes), B(opnd_bytes);
  std::vector<float> C0(cn), C(cn);
  if (fread(A.data(), 1, opnd_bytes, f) != opnd_bytes || fread(B.data(), 1, opnd_bytes, f) != opnd_bytes || fread(C0.data(), 4, cn, f) != cn) {
    printf("short read\n"); return 1;
  }
  fclose(f);
  char *dA, *dB; float *dC0, *dC;
  CK(hipMalloc(&dA, opnd_bytes)); CK(hipMalloc(&dB, opnd_bytes)); CK(hipMalloc(&dC0, cn * 4)); CK(hipMalloc(&dC, cn * 4));
  CK(hipMemcpy(dA, A.data(), opnd_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), opnd_bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC0, C0.data(), cn * 4, hipMemcpyHostToDevice));
  if (kind == "bf16_32") k32<0><<<T, 64>>>((const unsigned short*)dA, (const unsigned short*)dB, dC0, dC, steps);
  else if (kind == "f16_32") k32<1><<<T, 64>>>((const unsigned short*)dA, (const unsigned short*)dB, dC0, dC, steps);
  else if (kind == "bf16_16") k16<<<T, 64>>>((const unsigned short*)dA, (const unsigned short*)dB, dC0, dC, steps);
  else if (f32) kf32<<<T, 64>>>((const float*)dA, (const float*)dB, dC0, dC, steps);
  else { printf("unknown kind %s\n", kind.c_str()); return 2; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, cn * 4, hipMemcpyDeviceToHost));
  f = fopen(argv[3], "wb");
  if (!f || fwrite(C.data(), 4, cn, f) != cn) { printf("cannot write %s\n", argv[3]); return 1; }
  fclose(f);
  printf("%s: %d tiles x %d steps -> %s\n", kind.c_str(), T, steps, argv[3]);
  return 0;
}
