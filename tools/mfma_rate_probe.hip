// Bare MFMA issue-rate probe: how fast does ONE wave per SIMD issue v_mfma_f32_16x16x32_bf16 over 32 independent accumulators with the
// operands in registers -- accumulators in arch VGPRs ("v") or in AGPRs ("a")?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe && tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, const short* in) {
  bf16x8 a[4], b[8];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) a[i][e] = in[(threadIdx.x * 8 + e + i * 64) & 4095];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) b[i][e] = in[(threadIdx.x * 8 + e + i * 192 + 77) & 4095];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 0) MFMA_V(acc[i][j], a[j], b[i]);
        else MFMA_A(acc[i][j], a[j], b[i]);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const short* in, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<MODE><<<256, 256>>>(out, 10, in);
  (void)hipEventRecord(e0);
  probe<MODE><<<256, 256>>>(out, iters, in);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 32;
  printf("%s accumulators: %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s\n", MODE ? "AGPR" : "VGPR", ms, ms * 1e6 / mfmas,
         256.0 * 4 * mfmas * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  short* in;
  float* out;
  (void)hipMalloc(&in, 4096 * 2);
  (void)hipMalloc(&out, 256 * 512 * 4);
  short h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (short)(0x3f00 + (i * 37 % 251));
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(in, out);
  run<1>(in, out);
  run<0>(in, out);
  run<1>(in, out);
  return 0;
}
