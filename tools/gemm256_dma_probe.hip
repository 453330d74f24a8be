// Feasibility probe (not part of the library): the 256x256 / 4 x (128x128) AGPR tiling fed by LDS-DMA.
//   K tile = 32 bf16 (64-byte rows), FOUR LDS stages of 2 x 16 KiB, unpadded XOR-swizzled images written by
//   `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write), tile kt+4 requested in the middle of step kt,
//   one barrier per K tile placed between its two k-steps so no fragment read is exposed behind it.
// NT bf16, interior tiles only.  Build: hipcc --offload-arch=gfx950 -O3 tools/gemm256_dma_probe.hip -o tools/gemm256_dma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int TM = 256, TN = 256, BK = 32, NST = 4, OPB = 256 * 64, STAGE = 2 * OPB;

#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory", "m0");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(((size_t)p) >> 32) & 0xffff);
  r[2] = -1;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ bf16_t f2b(float f) {
  uint32_t u = __float_as_uint(f);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__global__ __launch_bounds__(256, 1) void gemm256(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                  bf16_t* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wv = __builtin_amdgcn_readfirstlane(wave);
  int pid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 4;
  const int group = pid / (GM * tiles_n), first_m = group * GM, gm = min(GM, tiles_m - first_m);
  const int tm = first_m + (pid - group * GM * tiles_n) % gm, tn = (pid - group * GM * tiles_n) / gm;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;

  // DMA: instruction j (0..3) of wave w fills LDS bytes [(4w + j) KiB, +1 KiB) of an operand image = rows 16 (4w + j) .. +15;
  // lane l -> row + (l >> 2), 16-byte slot l & 3, which holds k-chunk (l & 3) ^ ((row >> 2) & 3)
  const unsigned voff = (unsigned)((wave * 64 + (l >> 2)) * K * 2 + (((l & 3) ^ ((l >> 4) & 3)) * 16));
  const unsigned jstride = (unsigned)(16 * K * 2);
  const i32x4 rsa = make_rsrc(A + m0 * K), rsb = make_rsrc(B + n0 * K);
  // fragments: row (w0 + 32 i + r), k-step ks: chunk (2 ks + h) ^ ((r >> 2) & 3)
  const int r_ = l & 31, h = l >> 5, fr = (r_ >> 2) & 3;
  const int fa = (wr * 128 + r_) * 64, fb = OPB + (wc * 128 + r_) * 64;
  const int o0 = ((0 + h) ^ fr) * 16, o1 = ((2 + h) ^ fr) * 16;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
#define ISSUE(KT)                                                                                      \
  {                                                                                                    \
    const unsigned sbase = lds0 + ((KT) & (NST - 1)) * STAGE + wv * 4096;                              \
    const unsigned koff = (unsigned)(KT) * 64;                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) dma16(rsa, voff, koff + j * jstride, sbase + j * 1024); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) dma16(rsb, voff, koff + j * jstride, sbase + OPB + j * 1024); \
  }
#define FRAGS(AF, BF, ST, OFF)                                                                         \
  {                                                                                                    \
    AF[0] = *reinterpret_cast<const bf16x8*>((ST) + fa + (OFF));                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) BF[j] = *reinterpret_cast<const bf16x8*>((ST) + fb + j * 2048 + (OFF)); \
    _Pragma("unroll") for (int i = 1; i < 4; ++i) AF[i] = *reinterpret_cast<const bf16x8*>((ST) + fa + i * 2048 + (OFF)); \
  }
#define MMA(AF, BF)                                                                                    \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], acc[i][j], 0, 0, 0);         \
  }
#define PACE() { _Pragma("unroll") for (int i = 0; i < 8; ++i) { SGB(0x008, 2); SGB(0x100, 1); } }
  // s_waitcnt immediates (gfx9): vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
#define WAIT_GROUPS_LEFT(G)                                                                            \
  {                                                                                                    \
    if ((G) >= 3) __builtin_amdgcn_s_waitcnt(0x4f78);      /* vmcnt(24) */                             \
    else if ((G) == 2) __builtin_amdgcn_s_waitcnt(0x4f70); /* vmcnt(16) */                             \
    else if ((G) == 1) __builtin_amdgcn_s_waitcnt(0x0f78); /* vmcnt(8)  */                             \
    else __builtin_amdgcn_s_waitcnt(0x0f70);               /* vmcnt(0)  */                             \
  }

  const int pre = nk < NST ? nk : NST;
  for (int t = 0; t < pre; ++t) ISSUE(t)
  WAIT_GROUPS_LEFT(pre - 1)
  __syncthreads();
  bf16x8 a0[4], b0[4], a1[4], b1[4];
  FRAGS(a0, b0, smem, o0)
  for (int kt = 0; kt < nk - 1; ++kt) {
    const char* st = smem + (kt & (NST - 1)) * STAGE;
    FRAGS(a1, b1, st, o1)
    MMA(a0, b0)
    PACE()
    // tile kt+1 must have landed: the groups requested after it are tiles kt+2 .. min(kt+3, nk-1)
    const int last = kt + 3 < nk - 1 ? kt + 3 : nk - 1;
    WAIT_GROUPS_LEFT(last - (kt + 1))
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's reads of tile kt are complete
    __syncthreads();
    if (kt + NST < nk) ISSUE(kt + NST)
    const char* nx = smem + ((kt + 1) & (NST - 1)) * STAGE;
    FRAGS(a0, b0, nx, o0)
    MMA(a1, b1)
    PACE()
  }
  {
    const char* st = smem + ((nk - 1) & (NST - 1)) * STAGE;
    FRAGS(a1, b1, st, o1)
    MMA(a0, b0)
    PACE()
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
  if (M % 256 || N % 256 || K % 32) { printf("bad shape\n"); return 1; }
  std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hA) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
  for (auto& v : hB) { float f = rnd() * 0.1f; uint32_t u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
  bf16_t *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const int lds = NST * STAGE;
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
