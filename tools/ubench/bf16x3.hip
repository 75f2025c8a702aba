// Error-free fp32 = bf16 + bf16 + bf16 split on the bf16 MFMA of gfx950: layout and accuracy check.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x3 bf16x3.hip && ./bf16x3
// D[32 ch][32 pos] = W[32][K] (int8, exact in bf16) x X[K][32] (fp32), K = 384, three ways:
//   (a) v_mfma_f32_32x32x2_f32 (what the fused step kernel used up to round 2),
//   (b) v_mfma_f32_32x32x16_bf16 on the three bf16 pieces of X (hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = the rest:
//       x = hi + mid + lo exactly, every product w * piece is exact in fp32, the accumulation is fp32),
//   (c) v_mfma_f32_16x16x32_bf16 the same way (first 16 x 16 block),
// each compared with a double-precision dot product on the host.  Lane layout pinned here:
//   32x32x16: A lane l = W[i = l & 31][k = 8 (l >> 5) .. + 7], B lane l = X[k = 8 (l >> 5) .. + 7][j = l & 31]
//   16x16x32: A lane l = W[i = l & 15][k = 8 (l >> 4) .. + 7], B lane l = X[k = 8 (l >> 4) .. + 7][j = l & 15]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 384;

__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned xb = __builtin_bit_cast(unsigned, x);
  hi = xb & 0xffff0000u;
  const float r = x - __builtin_bit_cast(float, hi);
  const unsigned rb = __builtin_bit_cast(unsigned, r);
  mid = rb & 0xffff0000u;
  const float l = r - __builtin_bit_cast(float, mid);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ unsigned pk(unsigned a_hi16, unsigned b_hi16) { return (a_hi16 >> 16) | (b_hi16 & 0xffff0000u); }

__global__ void kern(const signed char* W, const float* X, float* Da, float* Db, float* Dc, unsigned* lo_low_bits) {
  const int l = threadIdx.x;
  // (a) fp32 MFMA
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int k = 0; k < K; k += 2) {
    const float a = static_cast<float>(W[(l & 31) * K + k + (l >> 5)]);
    const float b = X[(k + (l >> 5)) * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) Da[((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[e];
  // (b) 32x32x16 bf16 x 3
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  unsigned bad = 0;
  for (int k = 0; k < K; k += 16) {
    u32x4 a, bh, bm, bl;
    for (int q = 0; q < 4; ++q) {
      const int k0 = k + 8 * (l >> 5) + 2 * q;
      const float w0 = static_cast<float>(W[(l & 31) * K + k0]), w1 = static_cast<float>(W[(l & 31) * K + k0 + 1]);
      a[q] = pk(__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1));
      unsigned h0, m0, l0, h1, m1, l1;
      split3(X[k0 * 32 + (l & 31)], h0, m0, l0);
      split3(X[(k0 + 1) * 32 + (l & 31)], h1, m1, l1);
      bad |= (l0 | l1) & 0xffffu;
      bh[q] = pk(h0, h1); bm[q] = pk(m0, m1); bl[q] = pk(l0, l1);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) Db[((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[e];
  atomicOr(lo_low_bits, bad);
  // (c) 16x16x32 bf16 x 3, separate accumulators per piece
  f32x4 ch = {0, 0, 0, 0}, cm = ch, cl = ch;
  for (int k = 0; k < K; k += 32) {
    u32x4 a, bh, bm, bl;
    for (int q = 0; q < 4; ++q) {
      const int k0 = k + 8 * (l >> 4) + 2 * q;
      const float w0 = static_cast<float>(W[(l & 15) * K + k0]), w1 = static_cast<float>(W[(l & 15) * K + k0 + 1]);
      a[q] = pk(__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1));
      unsigned h0, m0, l0, h1, m1, l1;
      split3(X[k0 * 32 + (l & 15)], h0, m0, l0);
      split3(X[(k0 + 1) * 32 + (l & 15)], h1, m1, l1);
      bh[q] = pk(h0, h1); bm[q] = pk(m0, m1); bl[q] = pk(l0, l1);
    }
    ch = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bh), ch, 0, 0, 0);
    cm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bm), cm, 0, 0, 0);
    cl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bl), cl, 0, 0, 0);
  }
  for (int e = 0; e < 4; ++e) Dc[((l >> 4) * 4 + e) * 32 + (l & 15)] = ch[e] + (cm[e] + cl[e]);
}

int main() {
  std::vector<signed char> W(32 * K);
  std::vector<float> X(K * 32);
  srand(7);
  for (auto& w : W) w = static_cast<signed char>(rand() % 255 - 127);
  for (size_t i = 0; i < X.size(); ++i) {
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    X[i] = static_cast<float>(std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v) * ((i % 7 == 0) ? 1e-3 : 1.0));
  }
  signed char* dW; float *dX, *dA, *dB, *dC; unsigned* dbad;
  hipMalloc(&dW, W.size()); hipMalloc(&dX, X.size() * 4); hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dC, 4096); hipMalloc(&dbad, 4);
  hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4); hipMemset(dC, 0, 4096);
  kern<<<1, 64>>>(dW, dX, dA, dB, dC, dbad);
  std::vector<float> A(1024), B(1024), C(1024); unsigned bad;
  hipMemcpy(A.data(), dA, 4096, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, 4096, hipMemcpyDeviceToHost); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  double ea = 0, eb = 0, ec = 0, ef = 0, ref2 = 0, ma = 0, mb = 0, mc = 0; int nc = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double r = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) { r += static_cast<double>(W[i * K + k]) * X[k * 32 + j]; f = fmaf(static_cast<float>(W[i * K + k]), X[k * 32 + j], f); }
      const double da = A[i * 32 + j] - r, db = B[i * 32 + j] - r, df = f - r;
      ea += da * da; eb += db * db; ef += df * df; ref2 += r * r;
      ma = fmax(ma, fabs(da)); mb = fmax(mb, fabs(db));
      if (i < 16 && j < 16) { const double dc = C[i * 32 + j] - r; ec += dc * dc; mc = fmax(mc, fabs(dc)); ++nc; }
    }
  const double rr = std::sqrt(ref2 / 1024);
  printf("K = %d, int8 weights x fp32 activations, rms of the exact result %.3f\n", K, rr);
  printf("host fmaf chain (fp32)                 rms error %.3e  (relative %.3e)\n", std::sqrt(ef / 1024), std::sqrt(ef / 1024) / rr);
  printf("(a) v_mfma_f32_32x32x2_f32             rms error %.3e  (relative %.3e)  max %.3e\n", std::sqrt(ea / 1024), std::sqrt(ea / 1024) / rr, ma);
  printf("(b) 3 x v_mfma_f32_32x32x16_bf16       rms error %.3e  (relative %.3e)  max %.3e\n", std::sqrt(eb / 1024), std::sqrt(eb / 1024) / rr, mb);
  printf("(c) 3 x v_mfma_f32_16x16x32_bf16       rms error %.3e  (relative %.3e)  max %.3e   (16 x 16 block, one accumulator per piece)\n",
         std::sqrt(ec / nc), std::sqrt(ec / nc) / rr, mc);
  printf("low 16 bits of any `lo` piece set: %s (the split is %s)\n", bad ? "YES" : "no", bad ? "NOT exact" : "exact");
  return 0;
}
