// Feasibility probe (not part of the library): 256x256 workgroup tile, 4 waves x (128x128) with the accumulators in AGPRs,
// NT bf16 GEMM, interior tiles only.  Build: hipcc --offload-arch=gfx950 -O3 tools/gemm256_probe.hip -o tools/gemm256_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int TM = 256, TN = 256, RS = 144, OPB = 256 * RS, STAGE = 2 * OPB;

__device__ __forceinline__ bf16_t f2b(float f) {
  uint32_t u = __float_as_uint(f);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ __launch_bounds__(256, 1) void gemm256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                  bf16_t* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  int pid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 4;
  const int group = pid / (GM * tiles_n), first_m = group * GM, gm = min(GM, tiles_m - first_m);
  const int tm = first_m + (pid - group * GM * tiles_n) % gm, tn = (pid - group * GM * tiles_n) / gm;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;

  unsigned goff[8], soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + i * 256, row = c >> 3, kc = c & 7;
    goff[i] = (unsigned)(row * K * 2 + kc * 16);
    soff[i] = row * RS + kc * 16;
  }
  const char* ab = reinterpret_cast<const char*>(A + m0 * K);
  const char* bb = reinterpret_cast<const char*>(B + n0 * K);
  u32x4 ra[8], rb[8];
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ab + goff[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const u32x4*>(bb + goff[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + soff[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + OPB + soff[i]) = rb[i];
  __syncthreads();
  const int fa = (wr * 128 + (l & 31)) * RS + (l >> 5) * 16;
  const int fb = OPB + (wc * 128 + (l & 31)) * RS + (l >> 5) * 16;
  bf16x8 a0[4], b0[4], a1[4], b1[4];
#define LOAD_FRAGS(AF, BF, ST, KS)                                                                         \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) AF[i] = *reinterpret_cast<const bf16x8*>((ST) + fa + i * 32 * RS + (KS) * 32); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) BF[j] = *reinterpret_cast<const bf16x8*>((ST) + fb + j * 32 * RS + (KS) * 32); \
  }
#define MMA(AF, BF)                                                                                        \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], acc[i][j], 0, 0, 0);             \
  }
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  LOAD_FRAGS(a0, b0, smem, 0)
  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    const char* st = smem + cur * STAGE;
    char* nx = smem + (cur ^ 1) * STAGE;
    ab += 128; bb += 128;
    // ks = 0: MFMAs on set 0, global loads of the next tile, fragments of ks = 1 into set 1
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ab + goff[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const u32x4*>(bb + goff[i]);
    LOAD_FRAGS(a1, b1, st, 1)
    MMA(a0, b0)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x020, 1); SGB(0x100, 1); SGB(0x008, 1); SGB(0x020, 1); }
    LOAD_FRAGS(a0, b0, st, 2)
    MMA(a1, b1)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
    LOAD_FRAGS(a1, b1, st, 3)
    MMA(a0, b0)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
    // ks = 3: MFMAs on set 1 with the LDS writes of the next tile in between
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(nx + soff[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(nx + OPB + soff[i]) = rb[i];
    MMA(a1, b1)
#pragma unroll
    for (int i = 0; i < 16; ++i) { SGB(0x008, 1); SGB(0x200, 1); }
    __syncthreads();
    LOAD_FRAGS(a0, b0, nx, 0)
    cur ^= 1;
  }
  {
    const char* st = smem + cur * STAGE;
    LOAD_FRAGS(a1, b1, st, 1)
    MMA(a0, b0)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
    LOAD_FRAGS(a0, b0, st, 2)
    MMA(a1, b1)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
    LOAD_FRAGS(a1, b1, st, 3)
    MMA(a0, b0)
#pragma unroll
    for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
    MMA(a1, b1)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        const int64_t col = n0 + wc * 128 + j * 32 + (l & 31);
        C[row * N + col] = f2b(acc[i][j][e]);
      }
}

static float b2f(bf16_t v) {
  uint32_t u = ((uint32_t)v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 122880, K = argc > 2 ? atoi(argv[2]) : 2560, N = argc > 3 ? atoi(argv[3]) : 7680;
  if (M % 256 || N % 256 || K % 64) { printf("bad shape\n"); return 1; }
  std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hA) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
  for (auto& v : hB) { float f = rnd() * 0.1f; uint32_t u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
  bf16_t *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const int lds = 2 * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tiles_m = M / TM, tiles_n = N / TN;
  auto go = [&]() { hipLaunchKernelGGL(gemm256, dim3(tiles_m * tiles_n), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_m, tiles_n); };
  go();
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int iters = 10;
  for (int i = 0; i < iters; ++i) go();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("M=%d K=%d N=%d  %.3f ms  %.1f TFLOP/s\n", M, K, N, ms, 2.0 * M * N * K / ms / 1e9);
  // spot check
  std::vector<bf16_t> hC(64);
  double worst = 0;
  for (int t = 0; t < 64; ++t) {
    const int64_t r = ((int64_t)t * 7919 + 13) % M, c = ((int64_t)t * 104729 + 7) % N;
    bf16_t got;
    hipMemcpy(&got, dC + r * N + c, 2, hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)b2f(hA[r * K + k]) * b2f(hB[c * K + k]);
    const double err = fabs(ref - b2f(got)) / (fabs(ref) + 1e-2);
    if (err > worst) worst = err;
  }
  printf("worst rel err over 64 samples: %.4f\n", worst);
  return worst < 0.02 ? 0 : 2;
}
