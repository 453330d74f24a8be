// Feasibility probe (not part of the library): 256x256 output tile, EIGHT waves (2 x 4, 128x64 of C each, two per SIMD),
// v_mfma_f32_16x16x32_bf16, K tile 64, operands brought in by LDS-DMA (buffer_load_dwordx4 ... lds) into two 64 KiB stages.
//   * the two wave rows run one barrier apart ("ping-pong"): while waves 0-3 multiply (16 MFMAs = one 64x32 quadrant x K 64),
//     waves 4-7 issue their fragment reads and their share of the next DMA half-tile, and vice versa, so each SIMD's matrix
//     pipe always has exactly one wave feeding it;
//   * LDS image [256 rows][128 B], 16-byte chunk c of row r stored at slot c ^ ((r >> 1) & 7): DMA pieces stay full 128-byte
//     global lines (8 rows x 128 B per wave-instruction) and every ds_read_b128 of a 16x32 fragment is bank-conflict free;
//   * prefetch distance 1.5 K tiles in two stages: during K tile kt the four phases issue A-lo(kt+1), A-hi(kt+1) -> other
//     stage, B-lo(kt+2), B-hi(kt+2) -> this stage (its B images are retired after phase 1, its A images after phase 2);
//     one counted wait (vmcnt 4) per K tile.
// NT bf16 (A [M][K], B [N][K], C [M][N]), interior tiles only.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gemm8w_probe.hip -o tools/gemm8w_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int TM = 256, TN = 256, BK = 64, OPB = 256 * 128, STAGE = 2 * OPB;

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
#define FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define BARRIER() { FENCE() __builtin_amdgcn_s_barrier(); FENCE() }

__global__ __launch_bounds__(512, 2) void gemm8w(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                 int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
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

  // DMA: a half-tile (128 rows x 128 B) is 16 pieces of 8 rows; wave w moves pieces 2w and 2w+1 (rows 16 w .. 16 w + 15).
  // lane l of a piece -> row l >> 3, LDS slot l & 7, global chunk (l & 7) ^ ((row >> 1) & 7)
  const int prow0 = wave * 16 + (l >> 3), prow1 = prow0 + 8;
  const unsigned voff0 = (unsigned)(prow0 * K * 2 + (((l & 7) ^ ((prow0 >> 1) & 7)) * 16));
  const unsigned voff1 = (unsigned)(prow1 * K * 2 + (((l & 7) ^ ((prow1 >> 1) & 7)) * 16));
  const unsigned half_g = (unsigned)(128 * K * 2);  // global bytes between the two halves of an operand tile
  const i32x4 rsa = make_rsrc(A + m0 * K), rsb = make_rsrc(B + n0 * K);
  // issue half-tile `h` (0 / 1) of operand image at LDS byte `img` for K tile kt
#define ISSUE_HALF(RS, IMG, H, KT)                                                              \
  {                                                                                             \
    const unsigned sb = lds0 + (IMG) + (H) * 16384 + wv * 2048;                                 \
    const unsigned so = (unsigned)(KT) * 128 + (H) * half_g;                                    \
    dma16(RS, voff0, so, sb);                                                                   \
    dma16(RS, voff1, so, sb + 1024);                                                            \
  }

  // fragments: row r = l & 15 of a 16-row block, chunk (l >> 4) + 4 kh at slot chunk ^ (r >> 1)
  const int fr = l & 15, swz = fr >> 1;
  const int fo0 = fr * 128 + ((((l >> 4) + 0) ^ swz) * 16), fo1 = fr * 128 + ((((l >> 4) + 4) ^ swz) * 16);
  const int fa = wr * 16384, fb = OPB + wc * 8192;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  // prologue: A(0), B(0), B(1)
  ISSUE_HALF(rsa, 0, 0, 0)
  ISSUE_HALF(rsa, 0, 1, 0)
  ISSUE_HALF(rsb, OPB, 0, 0)
  ISSUE_HALF(rsb, OPB, 1, 0)
  if (nk > 1) {
    ISSUE_HALF(rsb, STAGE + OPB, 0, 1)
    ISSUE_HALF(rsb, STAGE + OPB, 1, 1)
    __builtin_amdgcn_s_waitcnt(0x0f74);  // vmcnt(4)
  } else {
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  }
  BARRIER()
  if (wr == 1) BARRIER()  // the second wave row runs one barrier behind the first

  bf16x8 a[4][2], b[2][2][2];
#define READ_A(ST, MH)                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
      a[i][0] = *reinterpret_cast<const bf16x8*>((ST) + fa + ((MH) * 4 + i) * 2048 + fo0);      \
      a[i][1] = *reinterpret_cast<const bf16x8*>((ST) + fa + ((MH) * 4 + i) * 2048 + fo1);      \
    }                                                                                           \
  }
#define READ_B(ST, NH)                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
      b[NH][j][0] = *reinterpret_cast<const bf16x8*>((ST) + fb + ((NH) * 2 + j) * 2048 + fo0);  \
      b[NH][j][1] = *reinterpret_cast<const bf16x8*>((ST) + fb + ((NH) * 2 + j) * 2048 + fo1);  \
    }                                                                                           \
  }
#define MMA(MH, NH)                                                                             \
  {                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
          acc[(MH) * 4 + i][(NH) * 2 + j] =                                                     \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kh], b[NH][j][kh], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                              \
  }
#define LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)

  for (int kt = 0; kt < nk; ++kt) {
    const int s = kt & 1;
    const char* st = smem + s * STAGE;
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    // phase 0: A rows 0-63 + B cols 0-31 | DMA A-lo(kt+1) -> other stage
    READ_A(st, 0)
    READ_B(st, 0)
    if (n1) ISSUE_HALF(rsa, (s ^ 1) * STAGE, 0, kt + 1)
    BARRIER()
    MMA(0, 0)
    BARRIER()
    // phase 1: B cols 32-63 | DMA A-hi(kt+1); the B reads are retired BEFORE the barrier: phase 2 overwrites this stage's B
    READ_B(st, 1)
    if (n1) ISSUE_HALF(rsa, (s ^ 1) * STAGE, 1, kt + 1)
    LGKM0();
    BARRIER()
    MMA(0, 1)
    BARRIER()
    // phase 2: A rows 64-127 | DMA B-lo(kt+2) -> this stage; A reads retired before the barrier (next K tile's phase 0 DMA)
    READ_A(st, 1)
    if (n2) ISSUE_HALF(rsb, s * STAGE + OPB, 0, kt + 2)
    LGKM0();
    BARRIER()
    MMA(1, 1)
    BARRIER()
    // phase 3: no reads | DMA B-hi(kt+2); everything K tile kt+1 needs has landed once at most B(kt+2) is outstanding
    if (n2) {
      ISSUE_HALF(rsb, s * STAGE + OPB, 1, kt + 2)
      __builtin_amdgcn_s_waitcnt(0x0f74);  // vmcnt(4)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    }
    BARRIER()
    MMA(1, 0)
    BARRIER()
  }
  if (wr == 0) BARRIER()

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = m0 + wr * 128 + i * 16 + 4 * (l >> 4) + e;
        const int64_t col = n0 + wc * 64 + j * 16 + (l & 15);
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
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8w), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tiles_m = M / TM, tiles_n = N / TN;
  auto go = [&]() { hipLaunchKernelGGL(gemm8w, dim3(tiles_m * tiles_n), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_m, tiles_n); };
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
  for (int t = 0; t < 256; ++t) {
    const int64_t r = ((int64_t)t * 7919 + 13) % M, c = ((int64_t)t * 104729 + 7) % N;
    bf16_t got;
    hipMemcpy(&got, dC + r * N + c, 2, hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)b2f(hA[r * K + k]) * b2f(hB[c * K + k]);
    const double err = fabs(ref - b2f(got)) / (fabs(ref) + 1e-2);
    if (err > worst) worst = err;
  }
  printf("worst rel err over 256 samples: %.4f\n", worst);
  return worst < 0.02 ? 0 : 2;
}
