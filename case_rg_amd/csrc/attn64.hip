// K18: head_dim-64 self-attention with the (sequence, head) RESIDENT in one workgroup (gfx950, bf16 in, f32 accumulate).
//
// nn.MultiheadAttention at head_dim 64 over the 384-token passages (common/TransformerEncoder.py:67, the H-wide blocks of
// common/TransformerBlock.py:26) is the attention the training step and the encoder-forward point spend their time in.  The
// flash-style kernels of attention.hip (four waves, one 64-key tile per barrier, register staging) ran it latency-bound at
// 0.15-0.22 of the MFMA peak with the vector pipe 67 % busy.  A (sequence, head) is small -- Q, K, V are 48 KiB each -- so
// here ONE workgroup of twelve waves owns a whole (sequence, head): every byte of Q, K and V enters the chip once, by LDS-DMA,
// on a stream that runs ahead of the arithmetic across items, and the twelve waves are a STATIC three-stage pipeline:
//
//   wave w owns queries 32 w .. 32 w + 31 (lane = query, as in attention.hip) and walks the keys in chunks of 96;
//   per chunk it runs   QK  : S^T = K Q^T (12 MFMAs, K rows from LDS) + key mask + chunk maximum
//                       SM  : online-softmax decision, 48 exponentials per lane, row sums, dropout
//                       PV  : pack P to bf16, O^T += V^T P^T (12 MFMAs, V^T by ds_read_b64_tr_b16)
//   one phase per INTERVAL (= one s_barrier); wave group r = w / 4 (one wave on each SIMD) runs r intervals behind group 0,
//   so in every interval each SIMD has one wave in each phase: the exponentials of one wave always sit beside the MFMAs of
//   the other two instead of all waves multiplying together and then all exponentiating together.
//
// Data movement is static too: 144 KiB (K, V, the next item's Q) per item = 144 pieces of 1 KiB, twelve per wave and item, issued
// by buffer_load_dwordx4 ... lds at fixed places of the wave's own phase sequence (K piece in QK, V piece in PV, Q pieces in the
// first two SM phases of an item).  K and V live in rings of four 12 KiB chunk slots (a K chunk is read in three consecutive
// intervals, once by each group; a V chunk likewise two intervals later), the next item's Q rows replace the current ones as
// soon as every group has taken its fragments.  A piece is issued >= 6 intervals before its first reader; every wave ends an
// interval with a COUNTED s_waitcnt -- an immediate, because the schedule repeats every item -- that leaves exactly the
// vector-memory operations of the last five intervals in flight, so the barrier that opens interval t publishes everything
// issued before t - 4.  (One wave issues about one instruction per 4.4 cycles whatever its kind: the first version of this
// kernel computed the schedule at run time and spent 40 % of every phase in scalar bookkeeping.)
//
// LDS images are plain 128-byte rows, swizzled on the SOURCE side of the DMA (the destination of an LDS-DMA is lane-linear):
//   K, Q (row reads, ds_read_b128):  16-byte chunk c of row r at slot c ^ ((r >> 1) & 7)  -- conflict-free fragments
//   V (transposed reads):            chunk c of row r at slot c ^ (4 ((r >> 1) & 1))       -- the four key rows of a
//                                    ds_read_b64_tr_b16 block land on four different 64-byte quarters of the 256-byte bank row
//
// Scope (fa64::fwd_ok): head_dim 64, not causal, 288 < Lk <= 384 (four chunks), Lq <= 384, 16-byte aligned operands whose
// sequences fit 32-bit buffer offsets.  Everything else stays with attention.hip.  Same arithmetic points as there: scores
// in f32, base-2 online softmax with lazy rescale (here at most three decisions per row), P rounded to bf16 once, the same
// counter RNG and element index for dropout (common.h), LSE in natural log.
#include "common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace fa64 {

constexpr int CK = 96, NCH = 4, LMAX = 384, SLOT = CK * 128, NKS = 4, NVS = 4, LAK = 3, LAV = 3, DEPTH = 5, NTHR = 768;
constexpr int K_OFF = 0, V_OFF = NKS * SLOT, Q_OFF = V_OFF + NVS * SLOT, M_OFF = Q_OFF + LMAX * 128, LDS_BYTES = M_OFF + 2 * 512;  // 148 480
// s_setprio per phase: measured flat (0 / 2 / 1, 1 / 3 / 2, 0 / 1 / 2, 2 / 1 / 0 against none: 0.157-0.162 ms all) -- not emitted
#ifndef FA64_PRIO_QK
#define FA64_PRIO_QK 0
#define FA64_PRIO_SM 0
#define FA64_PRIO_PV 0
#endif
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.69314718055994531f, RESCALE_THR = 8.f * LOG2E;

struct Args {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  int64_t ldq, ldk, ldv, sq, sk, sv;
  bf16_t* o; int64_t ldo, so;
  float* lse;
  const uint8_t* key_valid;
  int N, heads, Lq, Lk, nitems;
  float scale, drop_p;
  uint64_t seed, offset;
  const CaseStepState* state;  // nullable: offset += state->rng_base (ABI 600)
};

__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(((size_t)p) >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one LDS-DMA piece: 64 lanes x 16 bytes -> lds_addr + 16 lane.  Every byte of Q, K and V is read once chip-wide: streaming policy.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}
// 64 lanes x 4 bytes -> lds_addr + 4 lane (the key-validity bytes of an item: an ordinary load would make hipcc wait vmcnt(0) at
// its use and drain the DMA stream, and an inline-asm register load leaves its destination unprotected against compiler copies)
__device__ __forceinline__ void dma4(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
// all but the N youngest vector-memory operations of this wave are complete
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Vector-memory operations a wave issues at step s = 3 c + phase of an item (QK, SM, PV of chunks 0 .. 3) and the counted wait that
// closes the step: the operations of the last DEPTH steps stay in flight.  STEADY: an item with a successor (step 0: the previous
// item's five stores and K chunk 3; step 1: Q rows 0-95 and the two validity pieces; step 4: Q rows 96-383; QK: a K piece, PV: a
// V piece).  LAST: the final item of the workgroup (K and V chunk 3 only), behind a STEADY item.  Smaller actual counts (first
// item: everything older came from the drained prologue) only make the wait a no-op.
struct Sched {
  int steady[12], last[12];
};
constexpr Sched make_sched() {
  const int ps[12] = {6, 3, 1, 1, 3, 1, 1, 0, 1, 1, 0, 1}, pl[12] = {6, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  Sched r = {};
  for (int s = 0; s < 12; ++s) {
    int ns = 0, nl = 0;
    for (int k = 0; k < DEPTH; ++k) {
      const int t = s - k;
      ns += ps[(t + 12) % 12];
      nl += t >= 0 ? pl[t] : ps[t + 12];
    }
    r.steady[s] = ns;
    r.last[s] = nl;
  }
  return r;
}
constexpr Sched SCHED = make_sched();
static_assert(SCHED.steady[4] == 14 && SCHED.steady[11] == 3 && SCHED.last[0] == 9 && SCHED.last[7] == 0, "schedule table");
#pragma clang diagnostic pop

// accumulator registers 8 s2 .. 8 s2 + 7 -> B-operand fragment of k-step s2 (k order: row 16 s2 + 8 (j >> 2) + 4 half + (j & 3))
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
  u32x4 w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = f32x2_to_bf16x2(a[8 * s2 + 2 * j], a[8 * s2 + 2 * j + 1]);
  return *reinterpret_cast<bf16x8*>(&w);
}

// value of the OTHER 32-lane half's lane (l ^ 32) combined with this lane's: both halves hold partial statistics of one query
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifdef FA64_STAMPS
// diagnostic build only (tools/fa64_stamps.py): s_memtime of workgroup 0 at the end of each phase's work, after the counted wait and
// after the barrier, intervals FA64_T0 .. FA64_T0 + 35, every wave.  No stamp exists in the shipped library.
#ifndef FA64_T0
#define FA64_T0 26
#endif
__device__ uint64_t g_fa64_stamps[12 * 36 * 3];
#define FA64_STAMP(k) { if (blockIdx.x == 0 && tau >= FA64_T0 && tau < FA64_T0 + 36) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
    if (l == 0) g_fa64_stamps[(wave * 36 + (tau - FA64_T0)) * 3 + (k)] = t_; } }
#else
#define FA64_STAMP(k)
#endif

template <bool DROP>
__global__ __launch_bounds__(NTHR) void fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, r32 = l & 31, half = l >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // persistent: XCD x (blockIdx.x & 7) owns a contiguous range of (sequence, head) items and cuts it evenly over its workgroups; a
  // workgroup walks ITS range in order (head fastest), so the per-item bookkeeping is an increment
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nwx = ((int)gridDim.x - xcd + 7) >> 3;
  const int per = a.nitems >> 3, rem = a.nitems & 7;
  const int xfirst = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per, xcount = per + (xcd < rem ? 1 : 0);
  const int wper = xcount / nwx, wrem = xcount - wper * nwx;
  const int my_items = wper + (wslot < wrem ? 1 : 0);
  if (my_items == 0) return;
  const int pid0 = xfirst + (wslot < wrem ? wslot * (wper + 1) : wrem * (wper + 1) + (wslot - wrem) * wper);

  // descriptor of one (sequence, head) slice
  // (only the two address words of a descriptor are kept per slice: sixteen scalar registers less than whole descriptors)
  auto slice = [&](const void* base, int64_t seq_stride, int n, int head) {
    return ((uint64_t)base + (uint64_t)(((int64_t)n * seq_stride + head * 64) * 2)) & 0x0000ffffffffffffull;
  };
  auto desc = [&](uint64_t p, uint32_t bytes) {
    i32x4 r;
    r[0] = (int)(uint32_t)p;
    r[1] = (int)(uint32_t)(p >> 32);
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
  };
  const uint32_t qbytes = (uint32_t)(((int64_t)(a.Lq - 1) * a.ldq + 64) * 2), kbytes = (uint32_t)(((int64_t)(a.Lk - 1) * a.ldk + 64) * 2),
                 vbytes = (uint32_t)(((int64_t)(a.Lk - 1) * a.ldv + 64) * 2);
  const bool masked = a.key_valid != nullptr;
  const uint32_t mbytes = masked ? (uint32_t)a.Lk : 0u;  // bytes beyond Lk read as zeros; without a mask: an empty range, nobody reads the zeros
  auto mask_slice = [&](int n) {
    return (masked ? (uint64_t)a.key_valid + (uint64_t)((int64_t)n * a.Lk) : (uint64_t)a.q) & 0x0000ffffffffffffull;
  };

  // ---- DMA lane offsets: this wave's piece of a 96-row chunk = rows 8 wave .. 8 wave + 7, lane -> row (l >> 3), slot l & 7
  const int wr = 8 * wave + (l >> 3);
  const unsigned vK = (unsigned)(wr * a.ldk * 2 + (((l & 7) ^ ((wr >> 1) & 7)) << 4));
  const unsigned vQ = (unsigned)(wr * a.ldq * 2 + (((l & 7) ^ ((wr >> 1) & 7)) << 4));
  const unsigned vV = (unsigned)(wr * a.ldv * 2 + (((l & 7) ^ (4 * ((wr >> 1) & 1))) << 4));
  const unsigned vM = (unsigned)(4 * l);
  const unsigned kstep = (unsigned)(CK * a.ldk * 2), vstep = (unsigned)(CK * a.ldv * 2), qstep = (unsigned)(CK * a.ldq * 2);
  const unsigned piece = (unsigned)wave * 1024u;

  // ---- fragment lane offsets
  // row reads (K tiles, Q): lane (r32, half) takes chunk 2 s + half of row r32 at slot (2 s + half) ^ ((r32 >> 1) & 7)
  int roff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) roff[s] = r32 * 128 + (((2 * s + half) ^ ((r32 >> 1) & 7)) << 4);
  // transposed reads (V): 16-lane group (l >> 4) = (dhalf, half); lane 4 q + p of it addresses key row 4 half + q, columns 4 p .. 4 p + 3
  // of the 16-column block 32 dt + 16 dhalf: chunk 4 dt + 2 dhalf + (p >> 1), at slot chunk ^ 4 ((q >> 1) & 1)
  const int tq = (l & 15) >> 2, tp = l & 3, dhalf = (l >> 4) & 1;
  int toff[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
    toff[dt] = (4 * half + tq) * 128 + (((4 * dt + 2 * dhalf + (tp >> 1)) ^ (4 * ((tq >> 1) & 1))) << 4) + (tp & 1) * 8;

  // the item this wave computes (its group runs `grp` intervals behind group 0) and its successor in the workgroup's range
  int my_n = pid0 / a.heads, my_head = pid0 - my_n * a.heads;
  uint64_t ck, cv, nk, nv, nq, nm;  // K / V slices of this item; K / V / Q / validity slices of the next
  ck = slice(a.k, a.sk, my_n, my_head);
  cv = slice(a.v, a.sv, my_n, my_head);
  nk = ck; nv = cv; nq = ck; nm = ck;
  // byte step from the last head of a sequence to head 0 of the next (48-bit addresses: no carry into the descriptor's upper bits for
  // tensors below 2^47)
  const int64_t seqstep_k = (a.sk - (int64_t)(a.heads - 1) * 64) * 2, seqstep_v = (a.sv - (int64_t)(a.heads - 1) * 64) * 2,
                seqstep_q = (a.sq - (int64_t)(a.heads - 1) * 64) * 2;

  // ---- prologue: K chunks 0 .. 2, V chunks 0 .. 2, all of Q, the validity bytes of the first item
  {
    const i32x4 rq = desc(slice(a.q, a.sq, my_n, my_head), qbytes), rm = desc(mask_slice(my_n), mbytes);
#pragma unroll
    for (int c = 0; c < LAK; ++c) dma16(desc(ck, kbytes), vK, c * kstep, lds0 + K_OFF + c * SLOT + piece);
#pragma unroll
    for (int c = 0; c < LAV; ++c) dma16(desc(cv, vbytes), vV, c * vstep, lds0 + V_OFF + c * SLOT + piece);
#pragma unroll
    for (int c = 0; c < NCH; ++c) dma16(rq, vQ, c * qstep, lds0 + Q_OFF + c * SLOT + piece);
    dma4(rm, vM, lds0 + M_OFF);
    dma4(rm, vM + 256u, lds0 + M_OFF + 256);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float scale2 = a.scale * LOG2E;
  const float keep_scale = (DROP && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t thr = rng_threshold(a.drop_p);
#ifdef FA64_STAMPS
  int tau = grp;
#endif

  // group r idles r intervals (one barrier each) before its first phase and 2 - r after its last: every wave meets 12 items + 2 barriers
  for (int k = 0; k < grp; ++k) __builtin_amdgcn_s_barrier();

  const int q0 = 32 * wave, qi = q0 + r32;
  bf16x8 qf[4];
  f32x16 o[2], st[3];
  float m = -INFINITY, lsum = 0.f, mx = 0.f;
  uint32_t row_key = 0;
  bool has_next = false, all_valid = true;

  auto store_item = [&]() {
    // O = O^T keep_scale / l as 16-byte pieces: lane pairs (l, l + 32) hold columns 8 g + {0..3} / {4..7}; after the half swap the
    // lower lane owns all 8 columns of an even g and the upper lane those of g + 1.  Always 5 store instructions (the counted waits
    // rely on it): rows beyond Lq fall outside the descriptors.
    // (the lane's row / column offsets are recomputed from an opaque lane id: hoisted out of the item loop they cost six registers
    // that hipcc spilled to scratch -- and a scratch reload waits vmcnt(0), draining the DMA stream)
    int lane_ = l;
    asm volatile("" : "+v"(lane_));
    const int half = lane_ >> 5, qi = 32 * wave + (lane_ & 31);
    const float tot = half_sum(lsum);
    const float inv = tot > 0.f ? keep_scale / tot : 0.f;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.o + (int64_t)my_n * a.so + my_head * 64), 0, (int)(((int64_t)(a.Lq - 1) * a.ldo + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.lse + ((int64_t)my_n * a.heads + my_head) * a.Lq), 0, (int)(a.Lq * 4), 0x00020000);
    const int orow = qi * (int)a.ldo * 2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        uint32_t x[2], y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          x[i] = f32x2_to_bf16x2(o[dt][8 * gp + 2 * i] * inv, o[dt][8 * gp + 2 * i + 1] * inv);          // g = 2 gp
          y[i] = f32x2_to_bf16x2(o[dt][8 * gp + 4 + 2 * i] * inv, o[dt][8 * gp + 4 + 2 * i + 1] * inv);  // g = 2 gp + 1
          const auto sw = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false);
          x[i] = sw[0];
          y[i] = sw[1];
        }
        const int vo = qi < a.Lq ? orow + (32 * dt + 16 * gp + 8 * half) * 2 : 0x7fffffff;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], y[0], y[1]}, ro, vo, 0, 0);
      }
    const float lse_v = tot > 0.f ? m * LN2 + __logf(tot) : -INFINITY;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lse_v), rl, (half == 0 && qi < a.Lq) ? qi * 4 : 0x7fffffff, 0, 0);
  };

  // closing step S of an item: counted wait, then the barrier that publishes what has landed
  auto close = [&](auto step) {
    constexpr int S = decltype(step)::value;
    __builtin_amdgcn_sched_barrier(0);  // (with the empty asm at the end of each phase: the phase's arithmetic stays in its interval)
    FA64_STAMP(0)
    if (has_next) wait_vm<SCHED.steady[S]>();
    else wait_vm<SCHED.last[S]>();
    FA64_STAMP(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FA64_STAMP(2)
    __builtin_amdgcn_sched_barrier(0);
#ifdef FA64_STAMPS
    ++tau;
#endif
  };

  // one chunk of 96 keys: three phases, each one interval
  auto chunk = [&](auto cc, int item) {
    constexpr int c = decltype(cc)::value;
    // ================================================ QK ================================================
    if constexpr (FA64_PRIO_QK != 0 || FA64_PRIO_SM != 0 || FA64_PRIO_PV != 0) __builtin_amdgcn_s_setprio(FA64_PRIO_QK);
    if constexpr (c == 0) {
      if (item > 0) {
        store_item();
        ++my_head;
        if (my_head == a.heads) {
          my_head = 0;
          ++my_n;
        }
        ck = nk;
        cv = nv;
      }
      has_next = item + 1 < my_items;
      if (has_next) {  // the successor's slices: the next head of this sequence, or head 0 of the next sequence
        const bool wrap = my_head + 1 == a.heads;
        nk = ck + (uint64_t)(wrap ? seqstep_k : 128);
        nv = cv + (uint64_t)(wrap ? seqstep_v : 128);
        nq = (item == 0 ? slice(a.q, a.sq, my_n, my_head) : nq) + (uint64_t)(wrap ? seqstep_q : 128);
        nm = (item == 0 ? mask_slice(my_n) : nm) + (uint64_t)(wrap && masked ? a.Lk : 0);
      }
      dma16(desc(ck, kbytes), vK, 3u * kstep, lds0 + K_OFF + 3 * SLOT + piece);  // K chunk 3 of this item
    } else {
      if (has_next) dma16(desc(nk, kbytes), vK, (unsigned)(c - 1) * kstep, lds0 + K_OFF + (c - 1) * SLOT + piece);  // K chunk c - 1 of the next
    }
    if constexpr (c == 0) {
      const char* qb = smem + Q_OFF + q0 * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qb + roff[s]);
      m = -INFINITY;
      lsum = 0.f;
      if (DROP) row_key = rng_row_key(a.seed, a.offset + rng_base_of(a.state) + (uint64_t)(((int64_t)my_n * a.heads + my_head) * a.Lq + qi));
      // every key of the item valid?  (bytes beyond Lk were written as zeros)
      if (masked) {
        int lane_ = l;
        asm volatile("" : "+v"(lane_));  // (not hoisted out of the item loop: see store_item)
        const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + M_OFF + (item & 1) * 512);
        all_valid = __ballot(mw[lane_] == 0x01010101u && mw[64 + (lane_ & 31)] == 0x01010101u) == ~0ull;  // 96 dwords = 384 keys
      } else {
        all_valid = a.Lk == LMAX;
      }
    }
    {
      const char* kb = smem + K_OFF + c * SLOT;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kb + t * 4096 + roff[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s)
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kb + t * 4096 + roff[s]), qf[s], st[t], 0, 0, 0);
      }
    }
    if (!all_valid) {
      // validity of this chunk's 96 keys as wave-uniform words
      uint64_t km0;
      uint32_t km1;
      if (masked) {
        const uint8_t* mbuf = reinterpret_cast<const uint8_t*>(smem + M_OFF + (item & 1) * 512 + CK * c);
        km0 = __ballot(mbuf[l] != 0);
        km1 = (uint32_t)__ballot(l < 32 && mbuf[64 + (l & 31)] != 0);
      } else {
        km0 = __ballot(CK * c + l < a.Lk);
        km1 = (uint32_t)__ballot(l < 32 && CK * c + 64 + l < a.Lk);
      }
      const uint64_t m0 = km0 >> (4 * half);
      const uint32_t m1 = km1 >> (4 * half);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int bit = (e & 3) + 8 * (e >> 2);
          const bool ok = t < 2 ? ((m0 >> (bit + 32 * t)) & 1ull) != 0ull : ((m1 >> bit) & 1u) != 0u;
          st[t][e] = ok ? st[t][e] : -INFINITY;
        }
    }
    mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = (t == 0 ? 3 : 0); e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, st[t][e]), st[t][e + 1]);
    mx = fmaxf(fmaxf(mx, st[1][15]), st[2][15]);  // (tile 0 ends on an odd count: its register 15 was taken in the loop)
    // hipcc moves register-only arithmetic across s_barrier (IR-level sinking towards the use): values that must exist when the
    // interval closes pass through an empty asm
    asm volatile("" : "+v"(mx), "+v"(st[0]), "+v"(st[1]), "+v"(st[2]));
    close(std::integral_constant<int, 3 * c>{});

    // ================================================ SM ================================================
    if constexpr (FA64_PRIO_QK != 0 || FA64_PRIO_SM != 0 || FA64_PRIO_PV != 0) __builtin_amdgcn_s_setprio(FA64_PRIO_SM);
    if constexpr (c == 0) {
      if (has_next) {  // the next item's Q rows 0 .. 95 (group 0 took its fragments an interval ago) and its validity bytes
        dma16(desc(nq, qbytes), vQ, 0u, lds0 + Q_OFF + piece);
        const unsigned mb = lds0 + M_OFF + ((item + 1) & 1) * 512;
        const i32x4 rm = desc(nm, mbytes);
        dma4(rm, vM, mb);
        dma4(rm, vM + 256u, mb + 256);
      }
    } else if constexpr (c == 1) {
      if (has_next) {  // rows 96 .. 383 (every group has its fragments)
        const i32x4 rq = desc(nq, qbytes);
        dma16(rq, vQ, qstep, lds0 + Q_OFF + SLOT + piece);
        dma16(rq, vQ, 2u * qstep, lds0 + Q_OFF + 2 * SLOT + piece);
        dma16(rq, vQ, 3u * qstep, lds0 + Q_OFF + 3 * SLOT + piece);
      }
    }
    {
      mx = half_max(mx) * scale2;  // scale2 > 0: the maximum commutes with the scaling
      if constexpr (c == 0) {
        m = mx;
      } else {
        if (__any(mx > m + RESCALE_THR)) {
          // lazy rescale: the reference maximum moves only when some query's chunk maximum exceeds it by more than THR
          const float m_new = fmaxf(m, mx);
          const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
          lsum *= alpha;
          m = m_new;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
      }
      const float mref = (m == -INFINITY) ? 0.f : m;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[t][e] = __builtin_amdgcn_exp2f(fmaf(st[t][e], scale2, -mref));
      if constexpr (DROP) {
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) ps += st[t][e];
        lsum += ps;
      }
      if constexpr (DROP) {  // lane = one row of the probability matrix; registers = columns CK c + 32 t + 8 gg + 4 half + {0..3}
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const uint32_t base = (((uint32_t)(CK * c + 32 * t) >> 1) + 2u * (uint32_t)half) * RNG_C1;
#pragma unroll
          for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const uint32_t bits = rng_pair_bits_pre(row_key, base + (uint32_t)(4 * gg + i) * RNG_C1);
              st[t][4 * gg + 2 * i] = (bits & 0xffffu) >= thr ? st[t][4 * gg + 2 * i] : 0.f;
              st[t][4 * gg + 2 * i + 1] = (bits >> 16) >= thr ? st[t][4 * gg + 2 * i + 1] : 0.f;
            }
          __builtin_amdgcn_sched_barrier(0);  // one tile's hashes at a time (interleaving all 24 spills registers)
        }
      }
    }
    asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(lsum));
    close(std::integral_constant<int, 3 * c + 1>{});

    // ================================================ PV ================================================
    if constexpr (FA64_PRIO_QK != 0 || FA64_PRIO_SM != 0 || FA64_PRIO_PV != 0) __builtin_amdgcn_s_setprio(FA64_PRIO_PV);
    if constexpr (c == 0) {
      dma16(desc(cv, vbytes), vV, 3u * vstep, lds0 + V_OFF + 3 * SLOT + piece);  // V chunk 3 of this item
    } else {
      if (has_next) dma16(desc(nv, vbytes), vV, (unsigned)(c - 1) * vstep, lds0 + V_OFF + (c - 1) * SLOT + piece);  // V chunk c - 1 of the next
    }
    {
      // row sum of the probabilities
      if constexpr (!DROP) {  // (with dropout the sum runs in SM, in front of the mask)
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) ps += st[t][e];
        lsum += ps;
      }
      const char* vb = smem + V_OFF + c * SLOT;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_acc(st[t], s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const char* p = vb + t * 4096 + s2 * 2048 + toff[dt];
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 1024));
            bf16x8 vf;
            vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
            vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
            if (c == 0 && t == 0 && s2 == 0) {  // an item's first product starts the accumulators (no zero fill)
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, zero, 0, 0, 0);
            } else {
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
            }
          }
        }
    }
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(lsum));
    close(std::integral_constant<int, 3 * c + 2>{});
  };

  for (int item = 0; item < my_items; ++item) {
    chunk(std::integral_constant<int, 0>{}, item);
    chunk(std::integral_constant<int, 1>{}, item);
    chunk(std::integral_constant<int, 2>{}, item);
    chunk(std::integral_constant<int, 3>{}, item);
  }
  store_item();
  for (int k = 0; k < 2 - grp; ++k) __builtin_amdgcn_s_barrier();
}

bool fwd_ok(const CaseAttnDesc* d) {
  if (d->head_dim != 64 || d->causal || d->Lk <= 288 || d->Lk > LMAX || d->Lk % 4 || d->Lq > LMAX || d->Lq < 1) return false;
  const int64_t span_q = ((d->Lq - 1) * d->ldq + 64) * 2, span_k = ((d->Lk - 1) * d->ldk + 64) * 2, span_v = ((d->Lk - 1) * d->ldv + 64) * 2;
  const int64_t span_o = ((d->Lq - 1) * d->ldo + 64) * 2;
  return span_q < (1ll << 31) && span_k < (1ll << 31) && span_v < (1ll << 31) && span_o < (1ll << 31) && d->ldo % 8 == 0 && d->so % 8 == 0 &&
         d->N * d->heads < (1ll << 30);
}

int launch_fwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out, float* lse,
               hipStream_t s) {
  Args a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.sq = d->sq; a.sk = d->sk; a.sv = d->sv;
  a.o = (bf16_t*)out; a.ldo = d->ldo; a.so = d->so; a.lse = lse; a.key_valid = key_valid;
  a.N = (int)d->N; a.heads = (int)d->heads; a.Lq = (int)d->Lq; a.Lk = (int)d->Lk; a.nitems = (int)(d->N * d->heads);
  a.scale = d->scale; a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset; a.state = d->state;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_attention_fwd: cannot raise the dynamic LDS limit");
    attr = true;
  }
  const int cus = case_persistent_cus();
  const int grid = a.nitems < cus ? a.nitems : cus;
  if (a.drop_p > 0.f) hipLaunchKernelGGL(fwd_kernel<true>, dim3(grid), dim3(NTHR), LDS_BYTES, s, a);
  else hipLaunchKernelGGL(fwd_kernel<false>, dim3(grid), dim3(NTHR), LDS_BYTES, s, a);
  return case_check_launch("case_attention_fwd (resident)");
}


// =====================================================================================================================================
// K19: single-pass backward of the same attention.  One workgroup = twelve waves = one (sequence, head); wave w keeps keys 32 w ..
// 32 w + 31 STATIONARY (lane = key: K and V fragments and the dK^T / dV^T accumulators in registers, as attention.hip's dK / dV kernel)
// while the workgroup sweeps the queries in tiles of 32.  Per tile and wave:
//   step 1: S = Q K^T and dP = dO V^T (8 MFMAs; the accumulators START at -lse / scale and -delta, so p = 2^(scale2 S') and
//           dS = p dP' need no subtraction and no per-row constants in registers), the exponentials, dS; the bf16 dS^T tile goes to LDS
//   step 2: dV^T += dO^T P, dK^T += Q^T dS (8 MFMAs, the A operands by transposed reads of the SAME Q / dO tile images), and the
//           tile's dQ = dS K: eight pieces of 16 queries x 16 head-dim columns (waves 0 .. 7, one each), each a full contraction
//           over the 384 keys (12 x v_mfma_f32_16x16x32_bf16) from the dS^T tile and the K image -- no partial sums, no atomics,
//           ONE exponential per score and no second pass over Q / K / V / dO (the flash-style pair of kernels recomputes S, the
//           exponential and the dropout hash for dQ).
// Two barriers per tile (dS^T is the only thing that crosses waves).  Q / dO tiles, their -lse / scale and -delta rows and the NEXT
// item's K image arrive by LDS-DMA three tiles ahead (waves 0 .. 7: one Q or dO piece per tile; waves 8 .. 11: the K image and the
// statistics); the K image is double buffered, so the sweep runs on across items; the next item's V fragments are requested into
// dead registers during the last tile.  Images: 128-byte rows, 16-byte chunk c of row r at slot c ^ f((r >> 1) & 7),
// f(y) = (y0 << 2) | (y2 << 1) | y1: conflict-free for the ds_read_b128 row fragments, the 32 x 32 transposed fragments AND the
// 16 x 16 transposed fragments of the dQ product.  dS^T rows are 64 bytes (32 queries): 8-byte unit u of key row k at u ^ ((k >> 1) & 7)
// (conflict-free for the producers' ds_write_b64 -- sixteen consecutive keys per lane group -- and for the transposed reads).
// =====================================================================================================================================
namespace bwd {

constexpr int TQ = 32, NT = 768, KIMG = LMAX * 128, DS_OFF = 2 * KIMG, DS_BYTES = LMAX * 64, RING_OFF = DS_OFF + DS_BYTES;
constexpr int SL_Q = 0, SL_DO = 4096, SL_NL = 8192, SL_ND = 8448, SL_RK = 8704, SLOT_B = 8832, NSLOT = 4, LOOK = 3;
constexpr int LDS_B = RING_OFF + NSLOT * SLOT_B;  // 158 208

struct BArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;
  int64_t ldq, ldk, ldv, lddo, sq, sk, sv, sdo;
  const float* negl; const float* negd;  // [N, heads, Lq]: -lse / scale, -delta
  const uint8_t* key_valid;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int N, heads, Lq, Lk, nitems, ntiles;
  float scale, drop_p;
  uint64_t seed, offset;
  const CaseStepState* state;  // nullable: offset += state->rng_base (ABI 600)
};

__device__ __forceinline__ int swz(int row) {  // f((row >> 1) & 7)
  const int y = (row >> 1) & 7;
  return ((y & 1) << 2) | ((y >> 2) << 1) | ((y >> 1) & 1);
}

// -lse / scale and -delta = -rowsum(dO * O) per (sequence, head, query): the accumulator initial values of S and dP
__global__ __launch_bounds__(256) void stat_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, const float* __restrict__ lse,
                                                   float* __restrict__ negl, float* __restrict__ negd, int64_t rows, int heads, int Lq,
                                                   float inv_scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (n Lq + q) heads + head
  if (i >= rows * heads) return;
  const int64_t row = i / heads;
  const int head = (int)(i - row * heads);
  const bf16_t* a = dout + row * heads * 64 + head * 64;
  const bf16_t* b = out + row * heads * 64 + head * 64;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 64; c += 8) {
    const uint4 x = *reinterpret_cast<const uint4*>(a + c), y = *reinterpret_cast<const uint4*>(b + c);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      acc += __uint_as_float(xs[k] << 16) * __uint_as_float(ys[k] << 16) + __uint_as_float(xs[k] & 0xffff0000u) * __uint_as_float(ys[k] & 0xffff0000u);
  }
  const int64_t n = row / Lq, q = row - n * Lq;
  const int64_t o = (n * heads + head) * Lq + q;
  negd[o] = -acc;
  negl[o] = -lse[o] * inv_scale;  // lse = -inf (no valid key): +inf -> handled as p = 0 through the key mask of such a sequence
}

template <int N>
__device__ __forceinline__ void wait_vmk() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_small(int n) {  // all but the n youngest vector-memory operations are complete (n uniform, <= 15)
  if (n < 4) {
    if (n < 2) { if (n == 0) wait_vmk<0>(); else wait_vmk<1>(); } else { if (n == 2) wait_vmk<2>(); else wait_vmk<3>(); }
  } else if (n < 8) {
    if (n < 6) { if (n == 4) wait_vmk<4>(); else wait_vmk<5>(); } else { if (n == 6) wait_vmk<6>(); else wait_vmk<7>(); }
  } else if (n < 12) {
    if (n < 10) { if (n == 8) wait_vmk<8>(); else wait_vmk<9>(); } else { if (n == 10) wait_vmk<10>(); else wait_vmk<11>(); }
  } else {
    if (n < 14) { if (n == 12) wait_vmk<12>(); else wait_vmk<13>(); } else { if (n == 14) wait_vmk<14>(); else wait_vmk<15>(); }
  }
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 tr_pair(const char* p, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + hi_off));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

#ifdef FA64_STAMPS
// diagnostic build only (tools/fa64_stamps.py bwd): workgroup 0, tiles FA64B_T0 .. + 23, every wave: s_memtime at the tile's start, in front of
// and behind the first barrier, behind the dV / dK products, behind the dQ piece, in front of and behind the second barrier
#ifndef FA64B_T0
#define FA64B_T0 14
#endif
__device__ uint64_t g_fa64b_stamps[12 * 24 * 7];
#define FA64B_STAMP(k) { if (blockIdx.x == 0 && T >= FA64B_T0 && T < FA64B_T0 + 24) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
    if (l == 0) g_fa64b_stamps[(wave * 24 + (T - FA64B_T0)) * 7 + (k)] = t_; } }
#else
#define FA64B_STAMP(k)
#endif

template <bool DROP>
__global__ __launch_bounds__(NT) void bwd_kernel(const BArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = l & 31, half = l >> 5;  // (the tile loop recomputes them from an opaque lane id)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // persistent: the item ranges of K18's forward
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nwx = ((int)gridDim.x - xcd + 7) >> 3;
  const int per = a.nitems >> 3, rem = a.nitems & 7;
  const int xfirst = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per, xcount = per + (xcd < rem ? 1 : 0);
  const int wper = xcount / nwx, wrem = xcount - wper * nwx;
  const int my_items = wper + (wslot < wrem ? 1 : 0);
  if (my_items == 0) return;
  const int pid0 = xfirst + (wslot < wrem ? wslot * (wper + 1) : wrem * (wper + 1) + (wslot - wrem) * wper);
  const int ntiles = a.ntiles;

  auto slice = [&](const void* base, int64_t seq_stride, int n, int head) {
    return ((uint64_t)base + (uint64_t)(((int64_t)n * seq_stride + head * 64) * 2)) & 0x0000ffffffffffffull;
  };
  auto desc = [&](uint64_t p, uint32_t bytes) {
    i32x4 r;
    r[0] = (int)(uint32_t)p;
    r[1] = (int)(uint32_t)(p >> 32);
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
  };
  const uint32_t qbytes = (uint32_t)(((int64_t)(a.Lq - 1) * a.ldq + 64) * 2), kbytes = (uint32_t)(((int64_t)(a.Lk - 1) * a.ldk + 64) * 2),
                 obytes = (uint32_t)(((int64_t)(a.Lq - 1) * a.lddo + 64) * 2), sbytes = (uint32_t)(a.Lq * 4);

  // ---- DMA duty: waves 8 .. 11 (they have no dQ piece) issue everything, at the end of a tile's second step: wave 8 + j the Q piece and
  // the dO piece of tile rows 8 j .. 8 j + 7 and two pieces of the next item's K image per tile (first six tiles of an item); wave 8 also
  // the statistics rows, wave 9 the dropout row keys.  All per-item addresses advance by increments (head fastest inside the range).
  const int prow = 8 * (wave & 3) + (l >> 3);  // row of the 32-row tile
  const unsigned vTq = (unsigned)(prow * a.ldq * 2 + (((l & 7) ^ swz(prow)) << 4)), vTo = (unsigned)(prow * a.lddo * 2 + (((l & 7) ^ swz(prow)) << 4));
  const unsigned qtstep = (unsigned)(TQ * a.ldq * 2), otstep = (unsigned)(TQ * a.lddo * 2);
  // K image piece p = rows 8 p .. 8 p + 7; the pieces a wave issues all have the parity of the wave (wave + 12 i in the prologue, wave - 8 + 4 j
  // in the sweep), and the swizzle of row 8 p + r depends on p through its parity only
  const int krow = l >> 3;
  const unsigned vKi = (unsigned)(krow * a.ldk * 2 + (((l & 7) ^ swz(8 * (wave & 1) + krow)) << 4));
  const unsigned kpiece = (unsigned)(8 * a.ldk * 2);

  int my_n = pid0 / a.heads, my_head = pid0 - my_n * a.heads;  // the item being computed
  // the DMA stream runs LOOK tiles ahead: its item's head index, tile, and the base addresses of the operands in that item
  int dh = my_head, dtile = 0, ditem = 0;
  const int64_t seqstep_q = (a.sq - (int64_t)(a.heads - 1) * 64) * 2, seqstep_o = (a.sdo - (int64_t)(a.heads - 1) * 64) * 2;
  uint64_t qbase = slice(a.q, a.sq, my_n, my_head), obase = slice(a.dout, a.sdo, my_n, my_head);
  uint64_t soff = (uint64_t)(((int64_t)my_n * a.heads + my_head) * a.Lq) * 4ull;  // byte offset of the item's statistics rows; row index for the keys
  int n_cur = 0, n_prev = 0;
  // the pieces of the stream's tile into ring slot `sl` (waves 8 .. 11), then the stream moves on by one tile
  auto issue_tile = [&](int sl) {
    const unsigned sb = lds0 + RING_OFF + sl * SLOT_B;
    if (ditem < my_items) {
      dma16(desc(qbase, qbytes), vTq, (unsigned)dtile * qtstep, sb + SL_Q + (wave & 3) * 1024);
      dma16(desc(obase, obytes), vTo, (unsigned)dtile * otstep, sb + SL_DO + (wave & 3) * 1024);
      n_cur += 2;
      if (wave == 8) {
        dma4(desc(((uint64_t)a.negl + soff) & 0x0000ffffffffffffull, sbytes), (unsigned)(dtile * 128 + 4 * l), sb + SL_NL);
        dma4(desc(((uint64_t)a.negd + soff) & 0x0000ffffffffffffull, sbytes), (unsigned)(dtile * 128 + 4 * l), sb + SL_ND);
        n_cur += 2;
      } else if (DROP && wave == 9) {  // the dropout row keys of the tile's queries
        if (l < 32) {
          const uint32_t rk = rng_row_key(a.seed, a.offset + rng_base_of(a.state) + (soff >> 2) + (uint64_t)(dtile * TQ + l));
          *reinterpret_cast<uint32_t*>(smem + RING_OFF + sl * SLOT_B + SL_RK + 4 * l) = rk;
        }
      }
    }
    if (++dtile == ntiles) {
      dtile = 0;
      ++ditem;
      const bool w = dh + 1 == a.heads;
      qbase += (uint64_t)(w ? seqstep_q : 128);
      obase += (uint64_t)(w ? seqstep_o : 128);
      dh = w ? 0 : dh + 1;
      soff += (uint64_t)a.Lq * 4ull;
    }
  };

  // ---- prologue: the first item's K image (4 pieces per wave), tiles 0 .. 2
  uint64_t kcur = slice(a.k, a.sk, my_n, my_head);
  {
    const i32x4 rk = desc(kcur, kbytes);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave + 12 * i;
      dma16(rk, vKi, (unsigned)pc * kpiece, lds0 + pc * 1024);
    }
    if (wave >= 8)
      for (int i = 0; i < LOOK; ++i) issue_tile(i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  n_cur = 0;

  const float scale2 = a.scale * LOG2E;
  const float keep_scale = (DROP && a.drop_p > 0.f) ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t thr = rng_threshold(a.drop_p);
  const int k0 = 32 * wave, ki = k0 + r32;

  bf16x8 kf[4], vf[4];
  f32x16 dk[2], dv[2];
  bool key_ok = true, keys_all_ok = true;
  uint32_t kbyte = 1u;
  const uint32_t jc1 = ((uint32_t)ki >> 1) * RNG_C1, field_shift = 16u * ((uint32_t)ki & 1u);
  const int vrow = ki < a.Lk ? ki : a.Lk - 1;  // clamped: such a key is masked

  // V fragments and the validity byte of this lane's key in item (n, head): plain loads, issued a whole step before their first use
  auto request_stationary = [&](int n, int head) {
    const bf16_t* vp = a.v + (int64_t)n * a.sv + head * 64 + (int64_t)vrow * a.ldv + 8 * half;
#pragma unroll
    for (int s = 0; s < 4; ++s) vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s);
    kbyte = a.key_valid ? (uint32_t)a.key_valid[(int64_t)n * a.Lk + vrow] : 1u;
  };
  auto take_stationary = [&](int buf) {  // K fragments from the item's image; the validity flags
    if constexpr (!DROP) {  // (the dropout instantiation has no registers left: it reads the K fragments from the image every tile)
      const char* kb = smem + buf * KIMG + k0 * 128 + r32 * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(kb + (((2 * s + half) ^ swz(r32)) << 4));
    }
    key_ok = ki < a.Lk && kbyte != 0u;
    keys_all_ok = __all(key_ok);
  };
  auto store_item = [&]() {  // dK = scale dK^T^T, dV = dV^T^T as 16-byte pieces (K18's store_item); rows beyond Lk fall outside the descriptors
    int lane_ = l;
    asm volatile("" : "+v"(lane_));
    const int hf = lane_ >> 5, key = 32 * wave + (lane_ & 31);
    const __amdgpu_buffer_rsrc_t rdk = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dk + (int64_t)my_n * a.sk + my_head * 64), 0, (int)kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.dv + (int64_t)my_n * a.sv + my_head * 64), 0, (int)(((int64_t)(a.Lk - 1) * a.ldv + 64) * 2), 0x00020000);
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const f32x16& acc = which ? dv[dt] : dk[dt];
          const float mul = which ? 1.f : a.scale;
          uint32_t x[2], y[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            x[i] = f32x2_to_bf16x2(acc[8 * gp + 2 * i] * mul, acc[8 * gp + 2 * i + 1] * mul);
            y[i] = f32x2_to_bf16x2(acc[8 * gp + 4 + 2 * i] * mul, acc[8 * gp + 4 + 2 * i + 1] * mul);
            const auto sw = __builtin_amdgcn_permlane32_swap(x[i], y[i], false, false);
            x[i] = sw[0];
            y[i] = sw[1];
          }
          const int ld = which ? (int)a.ldv : (int)a.ldk;
          const int vo = key < a.Lk ? key * ld * 2 + (32 * dt + 16 * gp + 8 * hf) * 2 : 0x7fffffff;
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], y[0], y[1]}, which ? rdv : rdk, vo, 0, 0);
        }
    n_cur += 8;
  };

  request_stationary(my_n, my_head);
  take_stationary(0);
  const int64_t seqstep_k = (a.sk - (int64_t)(a.heads - 1) * 64) * 2;
  int T = 0;
  for (int item = 0; item < my_items; ++item) {
    const int kbuf = item & 1;
    const bool has_next = item + 1 < my_items;
    const bool wrap = my_head + 1 == a.heads;
    const int nn = wrap ? my_n + 1 : my_n, nh = wrap ? 0 : my_head + 1;
    const uint64_t knext = kcur + (uint64_t)(wrap ? seqstep_k : 128);
    const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dq + (int64_t)my_n * a.sq + my_head * 64), 0, (int)qbytes, 0x00020000);
    for (int t = 0; t < ntiles; ++t, ++T) {
      const char* slot = smem + RING_OFF + (T & 3) * SLOT_B;
      // fragment lane offsets, recomputed per tile from an opaque lane id: hoisted out of the loops they (and the 48 addresses of the dQ
      // product) cost more registers than the kernel has -- hipcc spilled 100 of them
      int ln = l;
      asm volatile("" : "+v"(ln));
      const int r32 = ln & 31, half = ln >> 5;
      // row reads (Q / dO tiles): chunk 2 s + half of row r32 at slot (2 s + half) ^ f(r32): s = 0 .. 3 differ by XOR 2 s
      const int rbase = r32 * 128, rsw = swz(r32) ^ half;
      // 32 x 32 transposed reads (Q^T, dO^T): 16-lane group (dhalf, half); lane 4 q + p: tile row 4 half + q (+ 8), chunk 4 dt + 2 dhalf + (p >> 1)
      const int tq = (ln & 15) >> 2, tp = ln & 3, dhalf = (ln >> 4) & 1, trow = 4 * half + tq;
      const int tlo = trow * 128 + (tp & 1) * 8, tcl = (2 * dhalf + (tp >> 1)) ^ swz(trow), tch = (2 * dhalf + (tp >> 1)) ^ swz(trow + 8);
      // ======================================== step 1: S, dP, the exponentials, dS ========================================
      FA64B_STAMP(0)
      f32x16 st, dp;
      {
        const float* nl = reinterpret_cast<const float*>(slot + SL_NL) + 4 * half;
        const float* nd = reinterpret_cast<const float*>(slot + SL_ND) + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(nl + 8 * g);
          st[4 * g] = x[0]; st[4 * g + 1] = x[1]; st[4 * g + 2] = x[2]; st[4 * g + 3] = x[3];
          if (!DROP) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(nd + 8 * g);
            dp[4 * g] = y[0]; dp[4 * g + 1] = y[1]; dp[4 * g + 2] = y[2]; dp[4 * g + 3] = y[3];
          }
        }
        if (TQ * t + TQ > a.Lq) {  // rows beyond Lq (their statistics read as zeros): p = 0
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (TQ * t + (e & 3) + 8 * (e >> 2) + 4 * half >= a.Lq) st[e] = -INFINITY;
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int ro = rbase + (((2 * s) ^ rsw) << 4);
          const bf16x8 kfs = DROP ? *reinterpret_cast<const bf16x8*>(smem + kbuf * KIMG + k0 * 128 + ro) : kf[s];
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(slot + SL_Q + ro), kfs, st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(slot + SL_DO + ro), vf[s], (DROP && s == 0) ? zero : dp, 0, 0, 0);
        }
      }
      // p = 2^(scale2 S'); the dropped-out probabilities (for dV) and dS / scale (the factor goes on the finished dK and dQ)
      if constexpr (!DROP) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          st[e] = __builtin_amdgcn_exp2f(st[e] * scale2);
          if (!keys_all_ok) st[e] = key_ok ? st[e] : 0.f;
          dp[e] *= st[e];
        }
      } else {
        const float* nd = reinterpret_cast<const float*>(slot + SL_ND) + 4 * half;
        const uint32_t* rkp = reinterpret_cast<const uint32_t*>(slot + SL_RK) + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(nd + 8 * g);
          const u32x4 rk4 = *reinterpret_cast<const u32x4*>(rkp + 8 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            float pr = __builtin_amdgcn_exp2f(st[e] * scale2);
            if (!keys_all_ok) pr = key_ok ? pr : 0.f;
            const float kp = ((rng_pair_bits_pre(rk4[i], jc1) >> field_shift) & 0xffffu) >= thr ? keep_scale : 0.f;
            dp[e] = pr * fmaf(dp[e], kp, d4[i]);  // P (keep / (1 - p) dP - delta)
            st[e] = pr * kp;
          }
          __builtin_amdgcn_sched_barrier(0);  // four rows at a time (interleaving all sixteen hashes spills)
        }
      }
      const bf16x8 pf0 = pack_acc(st, 0), pf1 = pack_acc(st, 1), df0 = pack_acc(dp, 0), df1 = pack_acc(dp, 1);
      {
        // dS^T[key][query]: this lane's key row, four queries (8 bytes) per unit 2 g + half
        char* dsr = smem + DS_OFF + (k0 + r32) * 64;
        const int sx = (r32 >> 1) & 7;  // (k0 is a multiple of 32)
        const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&df0);
        const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&df1);
        *reinterpret_cast<uint2*>(dsr + (((0 + half) ^ sx) << 3)) = make_uint2(w0[0], w0[1]);
        *reinterpret_cast<uint2*>(dsr + (((2 + half) ^ sx) << 3)) = make_uint2(w0[2], w0[3]);
        *reinterpret_cast<uint2*>(dsr + (((4 + half) ^ sx) << 3)) = make_uint2(w1[0], w1[1]);
        *reinterpret_cast<uint2*>(dsr + (((6 + half) ^ sx) << 3)) = make_uint2(w1[2], w1[3]);
      }
      FA64B_STAMP(1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      FA64B_STAMP(2)
      // the next item's V fragments and validity byte (K and V fragments are dead from here to the end of the item: the loads fly
      // under this tile's second step and the gradient stores)
#ifndef FA64B_LATE_STATIONARY
      if (t == ntiles - 1 && has_next) {
        request_stationary(nn, nh);
        n_cur += a.key_valid ? 5 : 4;
      }
#endif
      // ======================================== step 2: dV, dK, this tile's dQ ========================================
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = s2 ? pf1 : pf0, df = s2 ? df1 : df0;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int lo = tlo + (((4 * dt) ^ tcl) << 4), hi = tlo + 1024 + (((4 * dt) ^ tch) << 4) - lo;
          const bf16x8 ao = tr_pair(slot + SL_DO + s2 * 2048 + lo, hi);
          const bf16x8 aq = tr_pair(slot + SL_Q + s2 * 2048 + lo, hi);
          if (t == 0 && s2 == 0) {  // an item's first products start the accumulators (no zero fill)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, pf, zero, 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, df, zero, 0, 0, 0);
          } else {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, df, dk[dt], 0, 0, 0);
          }
        }
      }
#ifdef FA64_STAMPS
      asm volatile("" : "+v"(dk[0]), "+v"(dk[1]), "+v"(dv[0]), "+v"(dv[1]));
#endif
      FA64B_STAMP(3)
      if (wave < 8) {
        // dQ^T piece: head-dim columns 16 db .. (MFMA rows), queries 16 qb .. (MFMA columns), all keys.
        // lane addresses of k-step 0; a k-step is 32 keys further: + 4096 in the K image, + 2048 in dS^T (the swizzles repeat every 16 keys)
        const int qb = wave >> 2, db = wave & 3, lr = ln & 15, lg = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
        const int key_lo = 8 * lg + q4, ca = 2 * db + (p4 >> 1), ub = 4 * qb + p4;
        const char* ka = smem + kbuf * KIMG + key_lo * 128 + (p4 & 1) * 8;
        const int ka_lo = (ca ^ swz(key_lo)) << 4, ka_hi = 512 + ((ca ^ swz(key_lo + 4)) << 4);
        const char* da = smem + DS_OFF + key_lo * 64;
        const int da_lo = (ub ^ ((key_lo >> 1) & 7)) << 3, da_hi = 256 + ((ub ^ (((key_lo + 4) >> 1) & 7)) << 3);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
          const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ka + ks * 4096 + ka_lo));
          const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ka + ks * 4096 + ka_hi));
          const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(da + ks * 2048 + da_lo));
          const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(da + ks * 2048 + da_hi));
          bf16x8 af, bfr;
          af[0] = alo[0]; af[1] = alo[1]; af[2] = alo[2]; af[3] = alo[3]; af[4] = ahi[0]; af[5] = ahi[1]; af[6] = ahi[2]; af[7] = ahi[3];
          bfr[0] = blo[0]; bfr[1] = blo[1]; bfr[2] = blo[2]; bfr[3] = blo[3]; bfr[4] = bhi[0]; bfr[5] = bhi[1]; bfr[6] = bhi[2]; bfr[7] = bhi[3];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc, 0, 0, 0);
        }
        const int q = TQ * t + 16 * qb + lr;
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        const u32x2 w = {f32x2_to_bf16x2(acc[0] * a.scale, acc[1] * a.scale), f32x2_to_bf16x2(acc[2] * a.scale, acc[3] * a.scale)};
        __builtin_amdgcn_raw_buffer_store_b64(w, rdq, q < a.Lq ? q * (int)a.ldq * 2 + (16 * db + 4 * lg) * 2 : 0x7fffffff, 0, 0);
      } else {
        // the DMA pieces of tile T + 3 (its slot held tile T - 1: free since that tile's second barrier) and two pieces of the next
        // item's K image: (wave - 8) + 4 (2 t + i)
        issue_tile((T + LOOK) & 3);
        if (has_next && t < 6) {
          const i32x4 rk = desc(knext, kbytes);
          const int pc = (wave - 8) + 8 * t;
          dma16(rk, vKi, (unsigned)pc * kpiece, lds0 + (kbuf ^ 1) * KIMG + pc * 1024);
          dma16(rk, vKi, (unsigned)(pc + 4) * kpiece, lds0 + (kbuf ^ 1) * KIMG + (pc + 4) * 1024);
          n_cur += 2;
        }
      }
      FA64B_STAMP(4)
      // everything a DMA wave issued before the previous tile has landed (counted wait); the barrier publishes it
      if (wave >= 8) wait_vm_small(n_cur + n_prev);
      FA64B_STAMP(5)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      FA64B_STAMP(6)
      n_prev = n_cur;
      n_cur = 0;
    }
#ifdef FA64B_LATE_STATIONARY
    if (has_next) request_stationary(nn, nh);
#endif
    store_item();
    if (has_next) {
      my_n = nn;
      my_head = nh;
      kcur = knext;
      take_stationary(kbuf ^ 1);
    }
  }
}

bool ok(const CaseAttnDesc* d) {
  if (!fwd_ok(d) || d->Lq <= 256) return false;
  return d->ldo == d->heads * 64 && d->so == d->Lq * d->ldo;
}

int launch(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid, const void* out, const float* lse,
           const void* dout, float* scratch, void* dq, void* dk, void* dv, hipStream_t s) {
  const int64_t rows = d->N * d->Lq, nstat = d->N * d->heads * d->Lq;
  float* negl = scratch;
  float* negd = scratch + nstat;
  hipLaunchKernelGGL(stat_kernel, dim3((unsigned)((rows * d->heads + 255) / 256)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)out, lse, negl,
                     negd, rows, (int)d->heads, (int)d->Lq, 1.f / d->scale);
  BArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.lddo = d->ldo; a.sq = d->sq; a.sk = d->sk; a.sv = d->sv; a.sdo = d->so;
  a.negl = negl; a.negd = negd; a.key_valid = key_valid;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.N = (int)d->N; a.heads = (int)d->heads; a.Lq = (int)d->Lq; a.Lk = (int)d->Lk; a.nitems = (int)(d->N * d->heads);
  a.ntiles = (int)((d->Lq + TQ - 1) / TQ);
  a.scale = d->scale; a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset; a.state = d->state;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B) != hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_attention_bwd: cannot raise the dynamic LDS limit");
    attr = true;
  }
  const int cus = case_persistent_cus();
  const int grid = a.nitems < cus ? a.nitems : cus;
  if (a.drop_p > 0.f) hipLaunchKernelGGL(bwd_kernel<true>, dim3(grid), dim3(NT), LDS_B, s, a);
  else hipLaunchKernelGGL(bwd_kernel<false>, dim3(grid), dim3(NT), LDS_B, s, a);
  return case_check_launch("case_attention_bwd (resident)");
}

}  // namespace bwd
}  // namespace fa64


#ifdef FA64_STAMPS
extern "C" int case_attention_resident_stamps(uint64_t* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fa64::g_fa64_stamps), sizeof(fa64::g_fa64_stamps)) == hipSuccess ? 0 : -1;
}
extern "C" int case_attention_resident_bwd_stamps(uint64_t* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fa64::bwd::g_fa64b_stamps), sizeof(fa64::bwd::g_fa64b_stamps)) == hipSuccess ? 0 : -1;
}
#endif

// used by attention.hip's dispatch
int case_attention_resident_ok(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const void* out) {
  return fa64::fwd_ok(d) && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)out % 16 == 0;
}
int case_attention_resident_fwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out,
                                float* lse, hipStream_t s) {
  return fa64::launch_fwd(d, q, k, v, key_valid, out, lse, s);
}
int case_attention_resident_bwd_ok(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const void* out, const void* dout,
                                   const void* dq, const void* dk, const void* dv) {
  return fa64::bwd::ok(d) && ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0;
}
int case_attention_resident_bwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid, const void* out,
                                const float* lse, const void* dout, float* scratch, void* dq, void* dk, void* dv, hipStream_t s) {
  return fa64::bwd::launch(d, q, k, v, key_valid, out, lse, dout, scratch, dq, dk, dv, s);
}
