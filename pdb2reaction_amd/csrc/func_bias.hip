// Dev tool: mean SIGNED relative error (a bias) of the float32 device functions the kernels use, against double.
//   hipcc --offload-arch=gfx950 -O3 -o build/func_bias pdb2reaction_amd/csrc/func_bias.hip && build/func_bias
#include <hip/hip_runtime.h>
#include "umx_radial.h"
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>

__global__ void k_eval(const float* x, double* out, int n, int which) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const double d = (double)v;
  float f = 0.f; double r = 1.0;
  switch (which) {
    case 0: f = expf(-v); r = exp(-d); break;
    case 1: f = 1.0f / (1.0f + expf(-v)); r = 1.0 / (1.0 + exp(-d)); break;
    case 2: f = v / (1.0f + expf(-v)); r = d / (1.0 + exp(-d)); break;
    case 3: f = v * (1.0f / (1.0f + expf(-v))); r = d / (1.0 + exp(-d)); break;
    case 4: { const float a = fabsf(v) + 1e-3f; f = 1.0f / sqrtf(a); r = 1.0 / sqrt((double)a); break; }
    case 5: { const float a = fabsf(v) + 1e-3f; f = 1.0f / a; r = 1.0 / (double)a; break; }
    case 6: { const float a = fabsf(v) + 1e-3f; f = sqrtf(a); r = sqrt((double)a); break; }
    case 7: f = __expf(-v); r = exp(-d); break;
    case 9: { const float a = 0.6f + 0.1f * (fabsf(v) - floorf(fabsf(v))); f = 1.0f / sqrtf(a); r = 1.0 / sqrt((double)a); break; }
    case 10: { const float a = 0.6f + 0.1f * (fabsf(v) - floorf(fabsf(v))); f = sqrtf(a); r = sqrt((double)a); break; }
    case 11: { const float a = 0.77f + 0.06f * (fabsf(v) - floorf(fabsf(v))); f = 1.0f / a; r = 1.0 / (double)a; break; }
    case 12: { const float a = 0.6f + 0.1f * (fabsf(v) - floorf(fabsf(v))); f = (float)(1.0 / sqrt((double)a)); r = 1.0 / sqrt((double)a); break; }
    case 13: { const float a = 0.6f + 0.1f * (fabsf(v) - floorf(fabsf(v))); f = umx::r_rsqrt<0>(a + umx::LN_EPS); r = 1.0 / sqrt((double)a + (double)umx::LN_EPS); break; }
    case 14: f = umx::r_exp<2>(-v); r = exp(-d); break;
    case 15: f = umx::r_sigmoid<2>(v); r = 1.0 / (1.0 + exp(-d)); break;
    case 16: f = umx::r_silu<2>(v); r = d / (1.0 + exp(-d)); break;
    case 17: f = umx::r_silu<0>(v); r = d / (1.0 + exp(-d)); break;
    case 18: { const float a = fabsf(v) + 1e-3f; f = umx::r_rcp<2>(a); r = 1.0 / (double)a; break; }
    case 19: f = umx::r_silu_grad<2>(v); r = 0; { const double sd = 1.0 / (1.0 + exp(-d)); r = sd * (1.0 + d * (1.0 - sd)); } break;
    case 20: f = umx::r_silu_grad<0>(v); r = 0; { const double sd = 1.0 / (1.0 + exp(-d)); r = sd * (1.0 + d * (1.0 - sd)); } break;
    case 21: { const float dd = 1.0f + umx::r_exp<2>(-v); float y = __frcp_rn(dd); y = fmaf(fmaf(-dd, y, 1.0f), y, y); y = fmaf(fmaf(-dd, y, 1.0f), y, y); f = v * y; r = d / (1.0 + exp(-d)); break; }
    case 22: { const float dd = 1.0f + umx::r_exp<2>(-v); f = v / dd; r = d / (1.0 + exp(-d)); break; }
    case 23: { const float dd = 1.0f + umx::r_exp<2>(-v); float y = __frcp_rn(dd); const float q = v * y; f = fmaf(fmaf(-dd, q, v), y, q); r = d / (1.0 + exp(-d)); break; }
    case 24: { const float a = 0.6f + 0.1f * (fabsf(v) - floorf(fabsf(v))); f = umx::scale_rstd(1.2345678f, umx::rstd_eps(a, umx::LN_EPS)); r = 1.2345678f / sqrt((double)a + (double)umx::LN_EPS); break; }
    case 8: { const float s = 1.0f / (1.0f + expf(-v)); f = s * (1.0f + v * (1.0f - s)); const double sd = 1.0 / (1.0 + exp(-d)); r = sd * (1.0 + d * (1.0 - sd)); break; }
  }
  out[i] = ((double)f - r) / fabs(r);
}

// LayerNorm + SiLU of 128-wide rows exactly as the fused radial kernels do it (one wave per row, two values per lane)
__global__ void k_lnsilu(const float* x, const float* w, const float* b, float* out, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float2 v = *reinterpret_cast<const float2*>(x + (long)row * 128 + 2 * lane);
  const float2 ww = *reinterpret_cast<const float2*>(w + 2 * lane), bb = *reinterpret_cast<const float2*>(b + 2 * lane);
  *reinterpret_cast<float2*>(out + (long)row * 128 + 2 * lane) = umx::ln_silu_row<0>(v, ww, bb);
}

__global__ void k_lnparts(const float* x, float* out, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float2 v = *reinterpret_cast<const float2*>(x + (long)row * 128 + 2 * lane);
  const float mu = umx::wave_sum_dpp(v.x + v.y) * (1.0f / 128);
  v.x -= mu; v.y -= mu;
  const float var = umx::wave_sum_dpp(v.x * v.x + v.y * v.y) * (1.0f / 128);
  const float rstd = umx::r_rsqrt<0>(var + umx::LN_EPS);
  const float var2 = umx::wave_sum_shfl(v.x * v.x + v.y * v.y) * (1.0f / 128);      // butterfly order instead of DPP order
  if (lane == 0) { out[row * 4 + 0] = mu; out[row * 4 + 1] = var; out[row * 4 + 2] = rstd; out[row * 4 + 3] = var2; }
}

static void lnparts_test(const std::vector<float>& x, int rows, float* dx) {
  float* dout; hipMalloc(&dout, rows * 16);
  k_lnparts<<<rows / 4, 256>>>(dx, dout, rows);
  std::vector<float> o(rows * 4);
  hipMemcpy(o.data(), dout, rows * 16, hipMemcpyDeviceToHost);
  double em = 0, ev = 0, er = 0, ev2 = 0;
  for (int r = 0; r < rows; ++r) {
    double mu = 0, var = 0;
    for (int c = 0; c < 128; ++c) mu += x[(size_t)r * 128 + c];
    mu /= 128;
    for (int c = 0; c < 128; ++c) { const double d = x[(size_t)r * 128 + c] - mu; var += d * d; }
    var /= 128;
    const double rstd = 1.0 / std::sqrt(var + 1e-5);
    em += (o[r * 4] - mu) / std::fabs(mu); ev += (o[r * 4 + 1] - var) / var; er += (o[r * 4 + 2] - rstd) / rstd; ev2 += (o[r * 4 + 3] - var) / var;
  }
  std::printf("LayerNorm parts: mean signed rel err  mu %+.3e   var(dpp) %+.3e   var(butterfly) %+.3e   rstd %+.3e\n", em / rows, ev / rows, ev2 / rows, er / rows);
}

static void lnsilu_test() {
  const int rows = 1 << 15;
  std::vector<float> x((size_t)rows * 128), w(128), b(128), o((size_t)rows * 128);
  std::mt19937 g(7);
  std::normal_distribution<float> nd(0.f, 1.0f);
  for (auto& v : x) v = 0.3f + 0.8f * nd(g);
  for (auto& v : w) v = 1.0f + 0.2f * nd(g);
  for (auto& v : b) v = 0.1f * nd(g);
  float *dx, *dw, *db, *dout;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, 512); hipMalloc(&db, 512); hipMalloc(&dout, x.size() * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  lnparts_test(x, rows, dx);
  k_lnsilu<<<rows / 4, 256>>>(dx, dw, db, dout, rows);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  double dot_ar = 0, dot_rr = 0, msum = 0;
  for (int r = 0; r < rows; ++r) {
    double mu = 0, var = 0;
    for (int c = 0; c < 128; ++c) mu += x[(size_t)r * 128 + c];
    mu /= 128;
    for (int c = 0; c < 128; ++c) { const double d = x[(size_t)r * 128 + c] - mu; var += d * d; }
    var /= 128;
    const double rstd = 1.0 / std::sqrt(var + 1e-5);
    for (int c = 0; c < 128; ++c) {
      const double y = (x[(size_t)r * 128 + c] - mu) * rstd * w[c] + b[c];
      const double ref = y / (1.0 + std::exp(-y));
      const double a = o[(size_t)r * 128 + c];
      dot_ar += a * ref; dot_rr += ref * ref; msum += a - ref;
    }
  }
  std::printf("ln_silu_row<0> on %d rows: gain %+.3e   mean diff %+.3e\n", rows, dot_ar / dot_rr - 1.0, msum / ((double)rows * 128));
}

int main() {
  lnsilu_test();
  const int n = 1 << 22;
  std::vector<float> h(n);
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.5f);
  for (auto& v : h) v = nd(g);
  float* dx; double* dout;
  hipMalloc(&dx, n * sizeof(float)); hipMalloc(&dout, n * sizeof(double));
  hipMemcpy(dx, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  std::vector<double> o(n);
  const char* names[] = {"expf(-x)", "sigmoid = 1/(1+expf(-x))", "silu = x/(1+expf(-x))", "silu = x*(1/(1+expf(-x)))", "1/sqrtf(a)", "1/a", "sqrtf(a)", "__expf(-x)", "silu_grad", "1/sqrtf(a), a in [0.6,0.7]", "sqrtf(a), a in [0.6,0.7]", "1/a, a in [0.77,0.83]", "(float)(1/sqrt((double)a))", "r_rsqrt<0>(a + LN_EPS)", "r_exp<2>(-x)", "r_sigmoid<2>", "r_silu<2>", "r_silu<0>", "r_rcp<2>", "r_silu_grad<2>", "r_silu_grad<0>", "silu: exp<2>, rcp 2 Newton", "silu: exp<2>, true division", "silu: exp<2>, q = x*rcp + 1 residual step", "scale_rstd(c, rstd_eps(a, LN_EPS)), a in [0.6,0.7]"};
  for (int w = 0; w < 25; ++w) {
    k_eval<<<(n + 255) / 256, 256>>>(dx, dout, n, w);
    hipMemcpy(o.data(), dout, n * sizeof(double), hipMemcpyDeviceToHost);
    double m = 0, a = 0, mx = 0;
    long cnt = 0;
    for (double v : o) { if (!std::isfinite(v)) continue; m += v; a += std::fabs(v); mx = std::fmax(mx, std::fabs(v)); ++cnt; }
    std::printf("%-28s mean signed rel err %+.3e   mean |rel err| %.3e   max %.3e  (float eps/2 = 2.98e-08, %ld samples)\n", names[w], m / cnt, a / cnt, mx, cnt);
  }
  return 0;
}
