// K15  optimizer-side multi-tensor kernels: everything the reference's loop does to the parameters between two backward passes
// (common/CumulativeTrainer.py:70-76) -- clip_grad_norm_(params, 1), optim.Adam.step(), EMA.update() (common/EMA.py:13-18) --
// plus the refresh of the bf16 operand copies, as TWO launches over all 365 parameter tensors:
//   case_optim_sumsq     sum of squares of every gradient (the global L2 norm): one partial per chunk, then ONE workgroup adds
//                        the partials in a fixed order -- the scalar is bit-reproducible run to run and identical on every
//                        data-parallel rank (f32 atomics from 12k workgroups were neither, and the clip coefficient derived
//                        from it feeds Adam on every rank: replicas would drift apart in the last ulp)
//   case_optim_adam_ema  clip coefficient from that scalar (no host round trip), Adam moments and update, EMA lerp, bf16 copy;
//                        the step-dependent scalars ride in the table entry (torch.optim.Adam keeps a step per parameter), and
//                        an entry without a gradient only gets its EMA shadow moved (EMA.update touches every parameter)
// ~7 GB of traffic at H = 512 (p, g, m, v, shadow read; p, m, v, shadow, bf16 written) instead of 6 passes and ~200 launches.
// The tensors are addressed through a device table (one entry per tensor) and a chunk list (tensor index, chunk index), the
// usual multi-tensor-apply layout; both are built by the host binding (case_rg_amd/optim.py).
#include "common.h"

namespace {
constexpr int OPT_CHUNK = 16384;  // elements per workgroup
constexpr int OPT_THREADS = 256;

__global__ __launch_bounds__(OPT_THREADS) void optim_sumsq_kernel(const CaseOptTensor* __restrict__ table, const int32_t* __restrict__ chunks,
                                                                  float* __restrict__ partials) {
  __shared__ float red[32];
  const CaseOptTensor t = table[chunks[2 * blockIdx.x]];
  if (t.g == nullptr) {  // shadow-only entry (a parameter that received no gradient this step)
    if (threadIdx.x == 0) partials[blockIdx.x] = 0.f;
    return;
  }
  const int64_t begin = (int64_t)chunks[2 * blockIdx.x + 1] * OPT_CHUNK;
  const int64_t end = begin + OPT_CHUNK < t.numel ? begin + OPT_CHUNK : t.numel;
  const float* g = reinterpret_cast<const float*>(t.g) + begin;
  const int64_t n = end - begin;
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const int64_t nv = n / 4;
    for (int64_t i = threadIdx.x; i < nv; i += OPT_THREADS) {
      const float4 v = *reinterpret_cast<const float4*>(g + 4 * i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = 4 * nv + threadIdx.x; i < n; i += OPT_THREADS) s += g[i] * g[i];
  } else {
    for (int64_t i = threadIdx.x; i < n; i += OPT_THREADS) s += g[i] * g[i];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// fixed-order sum of the per-chunk partials: thread t adds partials[t], [t + 1024], ... in index order, then the block tree
constexpr int OPT_FINAL_THREADS = 1024;
__global__ __launch_bounds__(OPT_FINAL_THREADS) void optim_sumsq_final_kernel(const float* __restrict__ partials, int64_t n,
                                                                              float* __restrict__ out) {
  __shared__ float red[32];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += OPT_FINAL_THREADS) s += partials[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s;
}

struct AdamArgs {
  const float* sumsq;  // null: no clipping
  // derived on the host in double, as torch's _single_tensor_adam does: 1 - beta1, 1 - beta2; the step-dependent lr / (1 - beta1^t)
  // and sqrt(1 - beta2^t) are per tensor (CaseOptTensor.step_size / .bc2_sqrt)
  float max_norm, one_m_b1, beta2, one_m_b2, eps, ema_w;
  const CaseStepState* state;  // ABI 600, nullable: step_size / bc2_sqrt of the entries with a gradient come from the device struct
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float clip, float step_size,
                                         float bc2_sqrt) {
  g *= clip;
  m = m + a.one_m_b1 * (g - m);                    // exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + a.one_m_b2 * (g * g);          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / bc2_sqrt + a.eps;
  p -= step_size * (m / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(OPT_THREADS) void optim_adam_ema_kernel(const CaseOptTensor* __restrict__ table, const int32_t* __restrict__ chunks,
                                                                     const AdamArgs a) {
  const CaseOptTensor t = table[chunks[2 * blockIdx.x]];
  const int64_t begin = (int64_t)chunks[2 * blockIdx.x + 1] * OPT_CHUNK;
  const int64_t end = begin + OPT_CHUNK < t.numel ? begin + OPT_CHUNK : t.numel;
  float clip = 1.f;
  if (a.sumsq) clip = fminf(1.f, a.max_norm / (sqrtf(*a.sumsq) + 1e-6f));  // torch.nn.utils.clip_grad_norm_
  float* p = reinterpret_cast<float*>(t.p);
  const float* g = reinterpret_cast<const float*>(t.g);
  float* m = reinterpret_cast<float*>(t.m);
  float* v = reinterpret_cast<float*>(t.v);
  float* sh = reinterpret_cast<float*>(t.shadow);
  bf16_t* lp = reinterpret_cast<bf16_t*>(t.p_bf16);
  if (g == nullptr) {  // no gradient this step: the parameter and its moments stay, the EMA shadow still moves (common/EMA.py:13-18)
    if (sh && a.ema_w > 0.f)
      for (int64_t i = begin + threadIdx.x; i < end; i += OPT_THREADS) sh[i] = sh[i] + a.ema_w * (p[i] - sh[i]);
    return;
  }
  float step_size = t.step_size, bc2_sqrt = t.bc2_sqrt;
  if (a.state) {  // a captured step: the table is static, the step-dependent scalars are read when the kernel runs
    step_size = a.state->step_size;
    bc2_sqrt = a.state->bc2_sqrt;
  }
  // Round 6: 16-byte non-temporal accesses where every stream of the entry is 16-byte aligned (a chunk starts at a multiple of 16 384 elements):
  // parameter, gradient, both moments and the shadow are read and written ONCE per step and next touched a step later -- they need not displace
  // the bf16 operand copies (the one output the next forward pass reads) from the Infinity Cache.  The arithmetic per element is adam_one's, in the
  // same order: results are bit-identical to the scalar loop below, which stays for unaligned views (gradients inside a flat bucket).
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  const uintptr_t align = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                          reinterpret_cast<uintptr_t>(sh) | (reinterpret_cast<uintptr_t>(lp) << 1);
  const bool ema = sh && a.ema_w > 0.f;
  int64_t done = begin;
#ifndef CASE_STREAM_DEFAULT_POLICY
  if ((align & 15) == 0) {
    const int64_t nv = (end - begin) / 4;
    for (int64_t q = threadIdx.x; q < nv; q += OPT_THREADS) {
      const int64_t i = begin + 4 * q;
      f32x4_nt pv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p + i));
      const f32x4_nt gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(g + i));
      f32x4_nt mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(m + i));
      f32x4_nt vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(v + i));
      f32x4_nt sv = {0.f, 0.f, 0.f, 0.f};
      if (ema) sv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(sh + i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        adam_one(pe, gv[e], me, ve, a, clip, step_size, bc2_sqrt);
        pv[e] = pe; mv[e] = me; vv[e] = ve;
        if (ema) sv[e] = sv[e] + a.ema_w * (pe - sv[e]);
      }
      __builtin_nontemporal_store(pv, reinterpret_cast<f32x4_nt*>(p + i));
      __builtin_nontemporal_store(mv, reinterpret_cast<f32x4_nt*>(m + i));
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_nt*>(v + i));
      if (ema) __builtin_nontemporal_store(sv, reinterpret_cast<f32x4_nt*>(sh + i));
      if (lp) *reinterpret_cast<uint2*>(lp + i) = make_uint2(f32x2_to_bf16x2(pv[0], pv[1]), f32x2_to_bf16x2(pv[2], pv[3]));
    }
    done = begin + 4 * nv;
  }
#endif
  for (int64_t i = done + threadIdx.x; i < end; i += OPT_THREADS) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, a, clip, step_size, bc2_sqrt);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (sh && a.ema_w > 0.f) sh[i] = sh[i] + a.ema_w * (pi - sh[i]);  // torch lerp (weight < 0.5): shadow + w (p - shadow)
    if (lp) lp[i] = f32_to_bf16(pi);
  }
}
}  // namespace

extern "C" int case_optim_sumsq(const CaseOptTensor* table, const int32_t* chunks, int64_t nchunks, float* partials, float* sumsq,
                                case_stream_t stream) {
  CASE_REQUIRE(table && chunks && partials && sumsq && nchunks > 0 && nchunks < (1ll << 31), "case_optim_sumsq: bad argument");
  hipLaunchKernelGGL(optim_sumsq_kernel, dim3((unsigned)nchunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, table, chunks, partials);
  hipLaunchKernelGGL(optim_sumsq_final_kernel, dim3(1), dim3(OPT_FINAL_THREADS), 0, (hipStream_t)stream, partials, nchunks, sumsq);
  return case_check_launch("case_optim_sumsq");
}

extern "C" int case_optim_adam_ema(const CaseOptTensor* table, const int32_t* chunks, int64_t nchunks, const float* sumsq, float max_norm,
                                   double beta1, double beta2, double eps, double ema_w, const CaseStepState* state, case_stream_t stream) {
  CASE_REQUIRE(table && chunks && nchunks > 0 && nchunks < (1ll << 31), "case_optim_adam_ema: bad argument");
  CASE_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., "case_optim_adam_ema: bad hyper-parameters");
  const AdamArgs a = {sumsq, max_norm, (float)(1. - beta1), (float)beta2, (float)(1. - beta2), (float)eps, (float)ema_w, state};
  hipLaunchKernelGGL(optim_adam_ema_kernel, dim3((unsigned)nchunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, table, chunks, a);
  return case_check_launch("case_optim_adam_ema");
}

extern "C" int case_optim_chunk_elems(void) { return OPT_CHUNK; }

// ---- ABI 600: the device-resident step state ------------------------------------------------------------------------------------
namespace {
__global__ void step_advance_kernel(CaseStepState* st, uint64_t rng_stride, double beta1, double beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int32_t step = st->step + 1;
  st->step = step;
  st->rng_base += rng_stride;
  // lr / (1 - beta1^step) and sqrt(1 - beta2^step) in double, rounded once: the host forms the table entries the same way (optim.py)
  st->step_size = (float)((double)st->lr / (1.0 - pow(beta1, (double)step)));
  st->bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
}
}  // namespace

extern "C" int case_sizeof_step_state(void) { return (int)sizeof(CaseStepState); }

extern "C" int case_step_advance(CaseStepState* state, uint64_t rng_stride, double beta1, double beta2, case_stream_t stream) {
  CASE_REQUIRE(state != nullptr && (rng_stride & 1) == 0, "case_step_advance: null state or odd RNG stride (the kernels hash element pairs)");
  CASE_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., "case_step_advance: bad hyper-parameters");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, rng_stride, beta1, beta2);
  return case_check_launch("case_step_advance");
}
