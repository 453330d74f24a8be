// K3: strided-batched MFMA GEMM for gfx950 (MI355X), C = epilogue(alpha * op(A) op(B)).
//
// One kernel family covers every contraction on the CaSE path: nn.Linear forward (NT), its input
// gradient (NN, weight is k-major), its weight gradient (TN, both operands k-major, split-K with f32
// atomics), attention QK^T / PV and their gradients (two-level batch strides address heads in place
// inside the packed QKV projection), the Interaction bmm chain and the dense pointer map.
//
// Tile: 128x128 per 256-thread workgroup (4 waves, 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles),
// 128 bytes of K per LDS row (64 bf16 / 32 f32), two LDS stages, register-staged global loads issued
// before the MFMA phase of the previous tile and written after it (one barrier per K tile).
//   bf16 : v_mfma_f32_32x32x16_bf16, f32 accumulate.
//   f32  : v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) -- the parity mode.
// LDS images
//   k-contiguous operand : [128 rows][128 B + 16 B pad]  (row stride 36 dwords -> conflict-free ds_read_b128)
//   k-major operand      : [BK k-rows][128 cols + pad]; bf16 fragments are gathered with the gfx950
//                          transposing read ds_read_b64_tr_b16 (row stride 320 B -> 4 k-rows land on disjoint
//                          bank windows); f32 fragments are plain ds_read_b32 (lane = column).
// Workgroup ids are remapped so each XCD (8, private L2) walks a contiguous range of tiles, N fastest:
// neighbouring workgroups on an XCD share the A row panel and the (small) weight panel in L2.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Args {
  const void* A; const void* B; void* C;
  const float* bias_col; const float* bias_row; const void* aux; void* aux_out;
  int64_t M, N, K, lda, ldb, ldc, ld_aux;
  int64_t batch2, sa1, sa2, sb1, sb2, sc1, sc2, saux1, saux2;
  int tiles_m, tiles_n, split_k, k_tiles_per_split;
  int nwg;
  float alpha, drop_p;
  uint64_t seed, offset;
  const CaseStepState* state;  // nullable: offset += state->rng_base (ABI 600)
  int vec_a, vec_b;  // 16-byte global loads are legal for the operand
  int vec_c;         // 16-byte accesses are legal for C / aux / aux_out / bias_col
  float* rowsum;     // case_gemm_dw_bias: pre-zeroed f32 [M], receives sum_k op(A)[m, k] (the bias gradient of a weight-gradient GEMM)
  float* slabs;      // case_gemm_dw_slabs: split s of a split-K call stores its partial tile into slabs + s * slab_stride (no atomics)
  int64_t slab_stride;
  // case_gemm_ln: LayerNorm of the A rows as a prologue of the small-problem kernel (K = 512); ln_out [M, K] receives LN(A) (nullable)
  const float* ln_gamma; const float* ln_beta; void* ln_out; float ln_eps;
};

// The kernel body is parameterised by the workgroup size (gemm_impl.inc): 256 threads = 4 waves x (64x64) per 128x128
// tile, two workgroups per CU.  (A 512-thread / 8-wave x (32x64) build at four waves per SIMD was measured in round 1:
// +5..10 % on the k-major weight-gradient shapes, -1..3 % on the others, register-bound at 128 VGPRs -- not shipped.)
#define GEMM_NS gemm_w4
#define GEMM_NTHREADS 256
#include "gemm_impl.inc"
#undef GEMM_NS
#undef GEMM_NTHREADS

#include "gemm8w.inc"
#include "gemm_small.inc"

namespace {
constexpr int BM = 128, BN = 128, ROWB = 128;

int device_cus() { return case_device_cus(); }
}  // namespace

namespace {
// Validates the call and fills the kernel arguments; *tile receives the tiling (128 or 256) the call runs on.  Pure: depends on
// its arguments (and the CU count of the current device) only.
int prepare(const CaseGemmDesc* d, const void* A, const void* B, void* C, const float* bias_col, const float* bias_row,
            const void* aux, void* aux_out, Args& a, int* tile) {
  CASE_REQUIRE(d && A && B && C, "case_gemm: null argument");
  CASE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "case_gemm: empty problem M=%lld N=%lld K=%lld", (long long)d->M,
               (long long)d->N, (long long)d->K);
  CASE_REQUIRE(d->batch1 > 0 && d->batch2 > 0, "case_gemm: batch must be positive");
  CASE_REQUIRE(d->tile == 0 || d->tile == 64 || d->tile == 128 || d->tile == 256, "case_gemm: tile must be 0, 64, 128 or 256");
  const int epi = d->epilogue;
  CASE_REQUIRE(!(epi & CASE_EPI_BIAS_COL) || bias_col, "case_gemm: BIAS_COL without bias_col");
  CASE_REQUIRE(!(epi & CASE_EPI_BIAS_ROW) || bias_row, "case_gemm: BIAS_ROW without bias_row");
  CASE_REQUIRE(!(epi & (CASE_EPI_RESIDUAL | CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU)) || aux,
               "case_gemm: epilogue needs aux");
  const int split = d->split_k < 1 ? 1 : d->split_k;
  CASE_REQUIRE(split == 1 || ((epi & CASE_EPI_ATOMIC) && d->out_dtype == CASE_F32),
               "case_gemm: split_k > 1 needs CASE_EPI_ATOMIC and f32 output");
  CASE_REQUIRE(!((epi & CASE_EPI_RESIDUAL) && (epi & (CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU))),
               "case_gemm: RESIDUAL and MUL_D* share the aux operand and cannot be combined");
  CASE_REQUIRE(!(epi & CASE_EPI_DROPOUT) || (d->drop_p > 0.f && d->drop_p < 1.f), "case_gemm: drop_p out of range");
  CASE_REQUIRE(!(epi & CASE_EPI_ATOMIC) ||
                   !(epi & (CASE_EPI_GELU | CASE_EPI_RELU | CASE_EPI_RESIDUAL | CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU |
                            CASE_EPI_DROPOUT)),
               "case_gemm: non-linear epilogue cannot be combined with split-K accumulation");
  CASE_REQUIRE((d->in_dtype == CASE_BF16 || d->in_dtype == CASE_F32) && (d->out_dtype == CASE_F32 || (d->out_dtype == CASE_BF16 && d->in_dtype == CASE_BF16)),
               "case_gemm: dtype combination in=%d out=%d", d->in_dtype, d->out_dtype);
  const int esz = d->in_dtype == CASE_BF16 ? 2 : 4;
  const int ept = 16 / esz;
  a.A = A; a.B = B; a.C = C; a.bias_col = bias_col; a.bias_row = bias_row; a.aux = aux; a.aux_out = aux_out;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ld_aux = d->ld_aux;
  a.batch2 = d->batch2; a.sa1 = d->sa1; a.sa2 = d->sa2; a.sb1 = d->sb1; a.sb2 = d->sb2; a.sc1 = d->sc1;
  a.sc2 = d->sc2; a.saux1 = d->saux1; a.saux2 = d->saux2;
  a.tiles_m = (int)((d->M + BM - 1) / BM);
  a.tiles_n = (int)((d->N + BN - 1) / BN);
  const int bke = ROWB / esz;
  const int kt = (int)((d->K + bke - 1) / bke);
  a.split_k = split > kt ? kt : split;
  a.k_tiles_per_split = (kt + a.split_k - 1) / a.split_k;
  a.split_k = (kt + a.k_tiles_per_split - 1) / a.k_tiles_per_split;  // no empty splits
  const int64_t nwg = (int64_t)a.tiles_m * a.tiles_n * a.split_k * d->batch1 * d->batch2;
  CASE_REQUIRE(nwg < (1ll << 31), "case_gemm: grid too large");
  a.nwg = (int)nwg;
  a.alpha = d->alpha;
  a.rowsum = nullptr;
  a.slabs = nullptr;
  a.slab_stride = 0;
  a.ln_gamma = a.ln_beta = nullptr;
  a.ln_out = nullptr;
  a.ln_eps = 0.f;
  a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset; a.state = d->state;
  // 16-byte loads need the contiguous extent, every leading stride and the base to be 16-byte multiples
  auto aligned = [&](const void* p, int64_t ld, int64_t s1, int64_t s2, int64_t extent) {
    return ((uintptr_t)p % 16 == 0) && (ld % ept == 0) && (s1 % ept == 0) && (s2 % ept == 0) && (extent % ept == 0);
  };
  a.vec_a = aligned(A, d->lda, d->sa1, d->sa2, d->a_kmajor ? d->M : d->K);
  a.vec_b = aligned(B, d->ldb, d->sb1, d->sb2, d->b_kmajor ? d->N : d->K);
  {
    const int eo = d->out_dtype == CASE_BF16 ? 8 : 4;  // elements per 16 bytes of C
    auto ok = [&](const void* p, int64_t ld, int64_t s1, int64_t s2, int e) {
      return p == nullptr || (((uintptr_t)p % 16 == 0) && ld % e == 0 && s1 % e == 0 && s2 % e == 0);
    };
    a.vec_c = ok(C, d->ldc, d->sc1, d->sc2, eo) && ok(aux, d->ld_aux, d->saux1, d->saux2, ept) &&
              ok(aux_out, d->ld_aux, d->saux1, d->saux2, ept) && ok(bias_col, 4, 0, 0, 4);
  }
  *tile = 128;
  // interior bf16 problems large enough to fill the chip with 256x256 tiles go to the large-tile kernel
  if (d->tile != 128 && d->in_dtype == CASE_BF16 && d->batch1 * d->batch2 == 1 && d->M % 256 == 0 && d->N % 256 == 0 && d->K % 64 == 0 &&
      a.vec_a && a.vec_b && a.vec_c && d->lda < (1 << 22) && d->ldb < (1 << 22) && d->ldc < (1 << 22) && d->ld_aux < (1 << 22) &&
      (!(epi & CASE_EPI_ATOMIC) || (epi == CASE_EPI_ATOMIC && d->out_dtype == CASE_F32))) {
    const int64_t t256 = (d->M / 256) * (d->N / 256) * a.split_k;
    // k-major operands advance by 64 leading-dimension rows per K tile through a 32-bit DMA offset
    const bool span_ok = (!d->a_kmajor || (d->K / a.split_k + 64) * d->lda < (1ll << 30)) &&
                         (!d->b_kmajor || (d->K / a.split_k + 64) * d->ldb < (1ll << 30));
    if (span_ok && d->tile != 64 && (d->tile == 256 || gemm_t8w::prefer(nwg, t256, device_cus()))) *tile = 256;
  }
  // small problems (the 128x128 tiling cannot fill the chip once): 64x64 tiles with the whole K panel resident in LDS
  if (*tile == 128 && d->tile != 128 && d->tile != 256 && d->in_dtype == CASE_BF16 && d->batch1 * d->batch2 == 1 && d->M % 64 == 0 &&
      d->N % 64 == 0 && d->K % 64 == 0 && a.k_tiles_per_split <= gemm_sm::MAXKT && a.vec_a && a.vec_b && d->lda < (1 << 22) &&
      d->ldb < (1 << 22) && (!(epi & CASE_EPI_ATOMIC) || d->out_dtype == CASE_F32)) {
    const int64_t t64 = (d->M / 64) * (d->N / 64) * a.split_k;
    const bool span_ok = (!d->a_kmajor || (int64_t)a.k_tiles_per_split * 64 * d->lda < (1ll << 30)) &&
                         (!d->b_kmajor || (int64_t)a.k_tiles_per_split * 64 * d->ldb < (1ll << 30));
    if (span_ok && (d->tile == 64 || (nwg < device_cus() && t64 <= 2 * device_cus()))) *tile = 64;
  }
  return 0;
}
}  // namespace

#ifdef G8_CLOCK_STAMPS  // measurement builds only: (shader cycles, 100 MHz ticks) workgroup 0 of the LAST 8-wave launch spent between its first and last instruction
extern "C" int case_debug_gemm_clock(unsigned long long* out2) {
  unsigned long long h[4];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gemm_t8w::g_clock_stamps), sizeof(h)) != hipSuccess) return 1;
  out2[0] = h[2] - h[0];
  out2[1] = h[3] - h[1];
  return 0;
}
#endif
extern "C" int case_gemm_tile_for(const CaseGemmDesc* d, const void* A, const void* B, const void* C, const float* bias_col,
                                  const void* aux, const void* aux_out) {
  Args a;
  int tile = 0;
  const float one = 0.f;  // stands in for bias_row: only its presence is validated
  const int rc = prepare(d, A, B, const_cast<void*>(C), bias_col, (d && (d->epilogue & CASE_EPI_BIAS_ROW)) ? &one : nullptr, aux,
                         const_cast<void*>(aux_out), a, &tile);
  return rc ? rc : tile;
}

extern "C" int case_gemm_dw_bias(const CaseGemmDesc* d, const void* A, const void* B, void* C, float* d_bias, case_stream_t stream) {
  Args a;
  int tile = 0;
  const int rc = prepare(d, A, B, C, nullptr, nullptr, nullptr, nullptr, a, &tile);
  if (rc) return rc;
  CASE_REQUIRE(d_bias, "case_gemm_dw_bias: null d_bias");
  if (!(tile == 256 && d->a_kmajor && d->epilogue == CASE_EPI_ATOMIC && d->out_dtype == CASE_F32))
    return case_set_error(CASE_E_UNSUPPORTED, "case_gemm_dw_bias: needs the 256x256 tiling (see case_gemm_tile_for), a k-major A, "
                                              "the bare ATOMIC epilogue and f32 output");
  a.rowsum = d_bias;
  return gemm_t8w::launch<float, 1>(a, d->epilogue, d->a_kmajor, d->b_kmajor, case_persistent_cus(), (hipStream_t)stream);
}

namespace {
// C[m, n] (+)= sum over the splits, in split order, of the f32 slabs a case_gemm_dw_slabs GEMM left behind: every output element is summed
// by one thread in one fixed order, so the weight gradient is bit-identical from run to run (the atomic form's is not).  One thread
// per 4 consecutive outputs; the slab reads of a workgroup are 4 KiB runs.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, const int64_t quads,
                                                            const int splits, const int64_t stride, const int accumulate) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + q;
  const int64_t sq = stride / 4;
  f32x4 acc = accumulate ? reinterpret_cast<const f32x4*>(C)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (int64_t)(s + u) * sq);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < splits; ++s) acc += __builtin_nontemporal_load(src + (int64_t)s * sq);
  reinterpret_cast<f32x4*>(C)[q] = acc;
}
}  // namespace

extern "C" int64_t case_gemm_dw_slab_bytes(const CaseGemmDesc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
  const int kt = (int)((d->K + 63) / 64);
  int split = d->split_k < 1 ? 1 : d->split_k;
  split = split > kt ? kt : split;
  const int per = (kt + split - 1) / split;
  split = (kt + per - 1) / per;
  return split < 2 ? 0 : (int64_t)split * d->M * d->N * 4;
}

extern "C" int case_gemm_dw_slabs(const CaseGemmDesc* d, const void* A, const void* B, void* C, float* d_bias, void* slabs,
                                  int64_t slab_bytes, case_stream_t stream) {
  Args a;
  int tile = 0;
  const int rc = prepare(d, A, B, C, nullptr, nullptr, nullptr, nullptr, a, &tile);
  if (rc) return rc;
  if (!(tile == 256 && d->epilogue == CASE_EPI_ATOMIC && d->out_dtype == CASE_F32 && d->ldc == d->N && a.split_k > 1 && (!d_bias || d->a_kmajor)))
    return case_set_error(CASE_E_UNSUPPORTED, "case_gemm_dw_slabs: needs the 256x256 tiling (see case_gemm_tile_for), split_k > 1, the bare ATOMIC "
                                              "epilogue, f32 output with ldc == N (and a k-major A for d_bias)");
  CASE_REQUIRE(slabs && (uintptr_t)slabs % 16 == 0 && slab_bytes >= (int64_t)a.split_k * d->M * d->N * 4,
               "case_gemm_dw_slabs: workspace of %lld bytes, %lld needed (case_gemm_dw_slab_bytes)", (long long)slab_bytes,
               (long long)a.split_k * d->M * d->N * 4);
  a.rowsum = d_bias;
  a.slabs = reinterpret_cast<float*>(slabs);
  a.slab_stride = d->M * d->N;
  hipStream_t s = (hipStream_t)stream;
  const int rg = gemm_t8w::launch<float, 2>(a, d->epilogue, d->a_kmajor, d->b_kmajor, case_persistent_cus(), s);
  if (rg) return rg;
  const int64_t quads = d->M * d->N / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a.slabs, reinterpret_cast<float*>(C), quads,
                     a.split_k, a.slab_stride, 1);
  return case_check_launch("case_gemm_dw_slabs");
}

// C = epilogue(LN(A) B^T ...): the LayerNorm that feeds a small projection (the greedy step's LN1 -> QKV, LN2 -> cross-attention query,
// LN3 -> feed-forward; common/TransformerDecoder.py:76-89 at one position per sequence) as a PROLOGUE of the 64 x 64 small-problem kernel,
// whose A panel (64 rows x the whole K = 512) sits in LDS anyway: every workgroup normalises its 64 rows in place (two-pass mean / variance
// in f32, as case_layernorm_fwd) before the MFMA loop; the workgroups of the first column tile also store LN(A) to ln_out (the residual of
// the layer's next GEMM reads it).  One launch of ~10 us per LayerNorm gone from the step (24 per greedy step at 4 + 4 layers).
extern "C" int case_gemm_ln(const CaseGemmDesc* d, const void* A, const float* gamma, const float* beta, float eps, void* ln_out, const void* B,
                            void* C, const float* bias_col, const void* aux, case_stream_t stream) {
  Args a;
  int tile = 0;
  CASE_REQUIRE(d && gamma && beta, "case_gemm_ln: null argument");
  CaseGemmDesc dd = *d;
  dd.tile = 64;
  const int rc = prepare(&dd, A, B, C, bias_col, nullptr, aux, nullptr, a, &tile);
  if (rc) return rc;
  if (!(tile == 64 && d->K == 512 && d->lda == 512 && !d->a_kmajor && a.split_k == 1 && d->in_dtype == CASE_BF16 && d->batch1 * d->batch2 == 1 &&
        !(d->epilogue & (CASE_EPI_ATOMIC | CASE_EPI_BIAS_ROW | CASE_EPI_DROPOUT))))
    return case_set_error(CASE_E_UNSUPPORTED, "case_gemm_ln: built for bf16 rows of K = lda = 512 (k-contiguous A), M and N multiples of 64, "
                                              "unsplit, unbatched, without dropout (run case_layernorm_fwd + case_gemm)");
  CASE_REQUIRE((uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0 && (ln_out == nullptr || (uintptr_t)ln_out % 16 == 0),
               "case_gemm_ln: gamma / beta / ln_out must be 16-byte aligned");
  a.ln_gamma = gamma;
  a.ln_beta = beta;
  a.ln_out = ln_out;
  a.ln_eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (d->out_dtype == CASE_BF16) return gemm_sm::launch<bf16_t, true>(a, d->epilogue, false, d->b_kmajor, s);
  return gemm_sm::launch<float, true>(a, d->epilogue, false, d->b_kmajor, s);
}

extern "C" int case_gemm(const CaseGemmDesc* d, const void* A, const void* B, void* C, const float* bias_col,
                         const float* bias_row, const void* aux, void* aux_out, case_stream_t stream) {
  Args a;
  int tile = 0;
  const int rc = prepare(d, A, B, C, bias_col, bias_row, aux, aux_out, a, &tile);
  if (rc) return rc;
  const int epi = d->epilogue;
  hipStream_t s = (hipStream_t)stream;
  if (tile == 256) {
    const int cus = case_persistent_cus();
    if (d->out_dtype == CASE_BF16) return gemm_t8w::launch<bf16_t, 0>(a, epi, d->a_kmajor, d->b_kmajor, cus, s);
    if (epi & CASE_EPI_ATOMIC) return gemm_t8w::launch<float, 1>(a, epi, d->a_kmajor, d->b_kmajor, cus, s);
    return gemm_t8w::launch<float, 0>(a, epi, d->a_kmajor, d->b_kmajor, cus, s);
  }
  if (tile == 64) {
    if (d->out_dtype == CASE_BF16) return gemm_sm::launch<bf16_t, false>(a, epi, d->a_kmajor, d->b_kmajor, s);
    return gemm_sm::launch<float, false>(a, epi, d->a_kmajor, d->b_kmajor, s);
  }
  if (d->in_dtype == CASE_BF16 && d->out_dtype == CASE_BF16) return gemm_w4::launch<bf16_t, bf16_t>(a, epi, d->a_kmajor, d->b_kmajor, s);
  if (d->in_dtype == CASE_BF16 && d->out_dtype == CASE_F32) return gemm_w4::launch<bf16_t, float>(a, epi, d->a_kmajor, d->b_kmajor, s);
  if (d->in_dtype == CASE_F32 && d->out_dtype == CASE_F32) return gemm_w4::launch<float, float>(a, epi, d->a_kmajor, d->b_kmajor, s);
  return case_set_error(CASE_E_UNSUPPORTED, "case_gemm: dtype combination in=%d out=%d", d->in_dtype, d->out_dtype);
}
