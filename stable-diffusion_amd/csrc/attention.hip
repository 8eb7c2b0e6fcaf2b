// Flash-style attention on the gfx950 matrix cores (fp16 / bf16 operands, fp32 accumulation and
// softmax statistics); the score matrix never leaves registers.
//
// Work split: one workgroup = 4 (or 8) waves = 128 (256) queries of one (batch, head); each wave owns 32 queries.
// K/V are walked in tiles of 64 keys, staged global -> registers -> LDS (two LDS buffers, the loads
// of tile t+1 are issued before the MFMAs of tile t and written after them; one barrier per tile).
//
// Per tile and wave, with v_mfma_f32_32x32x16:
//   S^T = K Q^T   (A = K rows from LDS by ds_read_b128, B = the wave's Q fragments, kept in registers
//                  for the whole kernel) -> a lane holds, for ITS query (lane & 31), 16 of the 32 keys
//                  of a 32-key block, the other 16 sit in lane ^ 32: the softmax row statistics need
//                  one cross-half exchange per tile and no LDS;
//   P^T          = exp2(S^T * c - m) converted to 16 bit in place: the accumulator registers 8s..8s+7
//                  are exactly the B operand of k-step s of the next product (k order permuted, so
//                  the other operand is fetched in the same order);
//   O^T += V^T P^T (A = V^T read from the row-major V tile with ds_read_b64_tr_b16, the hardware
//                  transpose read) -> the output accumulator has the query on the lane as well, so the
//                  running-max rescale is a per-lane scalar multiply.
// Head sizes that are not multiples of 16 / 32 (40, 80) are zero-padded in the Q fragments only; K rows
// stay unpadded in LDS (row stride = odd multiple of 16 bytes => conflict-free ds_read_b128) and the V
// row stride is 64 or 192 (mod 256) bytes so the four key rows of a transpose read hit disjoint banks.
#include "ops.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

int attention_scalar(const AttentionArgs& a, hipStream_t stream);
int attention_f32(const AttentionArgs& a, hipStream_t stream);     // attention_f32.hip
int attention_f32_init();
int attention_init();

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma32;
template <> struct Mfma32<half_t> {
  typedef half_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32<bf16_t> {
  typedef bf16_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// max(a, b, c) without the NaN-quieting pre-pass fmaxf gets on matrix-core outputs (scores are finite or -inf)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// A vector instruction may read MFMA accumulators only once the matrix pipe has written them (8 passes plus 4 issue
// slots after a 32x32x16 product on gfx950; reading earlier returns the OLD register contents, not a stall).  The
// compiler keeps that distance for the instructions it emits itself but does not look inside inline assembly, so
// the first max3f that reads fresh accumulators is preceded by this.
__device__ __forceinline__ void mfma_settle(f32x16& a, f32x16& b) {
  asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void mfma_settle(f32x16& a) {
  asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a));
}
// max over the two 32-lane halves of the wave, left in every lane (one lane-swap instead of an LDS round trip)
__device__ __forceinline__ float max_halves(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max3f(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
}

template <typename T> struct Mfma16;
template <> struct Mfma16<half_t> {
  typedef half_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma16<bf16_t> {
  typedef bf16_t v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

struct FlashParams {
  const char* q; const char* k; const char* v; char* out;
  int ldq, ldk, ldv, ldo;      // in elements
  int B, heads, Nq, Nk;
  int qblocks;                 // 128-query blocks per (batch, head)
  float c;                     // scale * log2(e)
  int causal;                  // keys beyond the query's own index are masked (selects the kernel variant)
};

constexpr int KV_TILE = 64;

template <int D> struct Geo {
  static constexpr int KSTEPS = (D + 15) / 16;            // 16-wide contraction steps of K Q^T
  static constexpr int DT = (D + 31) / 32;                // 32-row output tiles of V^T P^T
  static constexpr int CPR = D / 8;                       // 16-byte chunks per K/V row
  static constexpr int NCH = KV_TILE * CPR;               // chunks per tile
  // Head sizes that leave spare contraction slots in the last 16-wide step (D % 16 == 8: 40) carry the
  // running maximum through the matrix cores: K rows get two extra columns holding 1, the query fragment
  // the two halves (hi, lo) of -m, and Q is scaled by scale * log2(e) when it is loaded, so that the
  // accumulators come out as the exponent itself (no multiply-subtract per score).
  static constexpr bool FOLD = (D % 16) == 8;
  static constexpr int SK = FOLD ? ((KSTEPS * 32) | 16)                 // room for the extra columns
                                 : (((D * 2 + 15) / 16 * 16) | 16);     // K row stride: odd multiple of 16 B
  static constexpr int sv_min = DT * 64;                  // bytes the transpose reads touch per row
  static constexpr int SV = (sv_min <= 64) ? 64 : (sv_min <= 192 ? 192 : (sv_min <= 320 ? 320 : (sv_min <= 448 ? 448 : 576)));
  static constexpr int KBUF = KV_TILE * SK + 32;          // + zeroed tail for the last row's over-read
  static constexpr int VBUF = KV_TILE * SV;
  static constexpr int STAGE = KBUF + VBUF;
  static constexpr bool ONES = (D % 32) != 0;             // a spare V column exists for the row sums
  // The last 32-row output tile holds at most 16 live rows (head dims and the ones column): it is computed as a
  // 16-row tile with v_mfma_f32_16x16x32 (two query halves x two 32-key steps per tile, 64 matrix-pipe cycles in
  // place of 128).  The P^T fragments come out of the accumulators laid out for 32x32x16; one
  // v_permlane16_swap per register pair turns the two k-steps of a 32-key block into the 16x16x32 B operands of the
  // two query halves (lane rows {q 0-15 | q 16-31} x {keys of half 0 | half 1} -> {q 0-15: four key octets}).
  static constexpr bool NARROW = ONES && (D % 32) + 1 <= 16;
  static constexpr int DT32 = NARROW ? DT - 1 : DT;       // full 32-row tiles
};

// d = 40 (the 64x64 level of the UNet, where most of the attention time goes): the K/V tile goes global -> LDS by
// LDS-DMA (`buffer_load_dwordx4 ... lds`, ten 1-KiB pieces per 64-key tile shared out over the waves) instead of
// through registers: no staging registers, no ds_write, one instruction per piece.  A piece lands lane-linearly,
// so the images are dense and the constant columns the kernel relies on live beside them, written once:
//   K   [64][80 B]   rows of 5 chunks (odd: conflict-free ds_read_b128); the chunk that carries the (1, 1) columns of
//                    the folded maximum is ONE constant chunk (KC, and a copy 32 rows further for the second key
//                    block), which the lanes of the upper half read in the last k-step instead of a row chunk
//   V03 [64][64 B]   head dims 0..31 (the 32-row output tile)
//   V4  [64][16 B]   head dims 32..39, and VC [64][16 B] = (1, 0, ..., 0) per key: the ones column and the zero
//                    padding of the 16-row tile; a lane of the transpose read takes its 8 bytes from V4 or from VC
// The descriptors end with the last key, so the rows a ragged last tile reads beyond it land as zeros.
struct Dma40 {
  static constexpr int V03 = 5120, V4 = 9216, VC = 10304, KC = 11328, KC2 = KC + 32 * 80, STAGE = KC2 + 16;
  static constexpr int PIECES = 10;       // K: 0..4, V03: 5..8, V4: 9 -- piece p lands at byte 1024 p of the stage
  static_assert(STAGE % 16 == 0 && VC % 256 == 64 && V4 % 256 == 0, "bank placement of the constant rows");
};

// NW waves per workgroup (4 or 8): 32 * NW queries share every staged K/V tile
template <typename T, int D, int NW, bool CAUSAL>
__global__ __launch_bounds__(64 * NW, (D <= 40 ? 16 / NW : (D <= 80 ? 8 / NW : 1))) void attn_flash_kernel(FlashParams p) {
  typedef Geo<D> G;
  constexpr int NT = 64 * NW, KC = (G::NCH + NT - 1) / NT;
  typedef typename Mfma32<T>::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware order: consecutive workgroup ids go round the 8 XCDs, so the ids dealt to one XCD (id % 8)
  // are given a contiguous range of work items, query block fastest: the query blocks of a (batch, head)
  // then run on the same XCD and its K/V stays in that XCD's L2.
  int qblk, bh;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, x = bid & 7;
    const int id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (bid >> 3);
    // (the division runs on the vector ALU: tell the compiler that its result is uniform)
    bh = __builtin_amdgcn_readfirstlane(id / p.qblocks);
    qblk = id - bh * p.qblocks;
  }
  const int b = bh / p.heads, hd = bh - b * p.heads;
  const int qi = qblk * (32 * NW) + wid * 32 + r;
  const int qrow = min(qi, p.Nq - 1);

  // ---- Q fragments (B operand of K Q^T): dims 16s + 8h .. +7 of this lane's query ---------------
  v8 qf[G::KSTEPS];
  {
    const char* qp = p.q + ((size_t)((size_t)b * p.Nq + qrow) * p.ldq + (size_t)hd * D) * sizeof(T);
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
      const int d0 = 16 * s + 8 * h;
      if (d0 < D) {
        qf[s] = *reinterpret_cast<const v8*>(qp + d0 * sizeof(T));
        if constexpr (G::FOLD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) qf[s][e] = (T)((float)qf[s][e] * p.c);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (T)0.f;
      }
    }
  }

  // ---- K/V staging -----------------------------------------------------------------------------
  constexpr bool DMA = D == 40 && !CAUSAL;
  constexpr int STAGE_B = DMA ? Dma40::STAGE : G::STAGE;
  constexpr int KCR = DMA ? 1 : KC;
  u32x4 kreg[KCR], vreg[KCR];
  const char* kbase = p.k + ((size_t)b * p.Nk * p.ldk + (size_t)hd * D) * sizeof(T);
  const char* vbase = p.v + ((size_t)b * p.Nk * p.ldv + (size_t)hd * D) * sizeof(T);
  // chunk geometry of this thread (clamped so every staging register is always written): byte offsets of its
  // chunks inside a 64-key tile in global memory and in the two LDS images.  The tile's base address is
  // uniform (scalar registers), so a load costs no vector address arithmetic.
  unsigned kgo[KCR], vgo[KCR], kso[KCR], vso[KCR];
  bool ch_ok[KCR];
  if constexpr (!DMA) {
#pragma unroll
    for (int i = 0; i < KCR; ++i) {
      const int c = tid + i * NT;
      ch_ok[i] = c < G::NCH;
      const int cc = ch_ok[i] ? c : G::NCH - 1;
      const int key = cc / G::CPR, piece = cc - key * G::CPR;
      kgo[i] = (unsigned)((key * p.ldk + piece * 8) * sizeof(T));
      vgo[i] = (unsigned)((key * p.ldv + piece * 8) * sizeof(T));
      kso[i] = key * G::SK + piece * 16;
      vso[i] = G::KBUF + key * G::SV + piece * 16;
    }
  }
  const int ntiles = (p.Nk + KV_TILE - 1) / KV_TILE;
  // K / V of this (batch, head) through buffer descriptors that end with the last key's row: the tile offset
  // rides in a scalar register, the load needs no vector address arithmetic, and the rows a ragged last tile
  // reads beyond the sequence come back as zeros (their scores are masked below)
  const unsigned ktile = (unsigned)(KV_TILE * p.ldk * sizeof(T)), vtile = (unsigned)(KV_TILE * p.ldv * sizeof(T));
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)kbase, 0, (int)(((size_t)(p.Nk - 1) * p.ldk + D) * sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)vbase, 0, (int)(((size_t)(p.Nk - 1) * p.ldv + D) * sizeof(T)), 0x00020000);
#define LOAD_TILE(t_)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < KCR; ++i) {                                                    \
    kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(krs, kgo[i], (t_) * ktile, 0);                        \
    vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vgo[i], (t_) * vtile, 0);                        \
  }
#define STORE_TILE(buf_)                                                                                  \
  {                                                                                                       \
    char* sb_ = smem + (buf_) * G::STAGE;                                                                 \
    _Pragma("unroll") for (int i = 0; i < KCR; ++i) {                                                  \
      if (ch_ok[i]) {                                                                                     \
        *reinterpret_cast<u32x4*>(sb_ + kso[i]) = kreg[i];                                                \
        *reinterpret_cast<u32x4*>(sb_ + vso[i]) = vreg[i];                                                \
      }                                                                                                   \
    }                                                                                                     \
  }
  // LDS-DMA path: piece q = wave + i NW of the tile belongs to this wave; a lane's source offset inside the tile
  constexpr int PW = (Dma40::PIECES + NW - 1) / NW;
  unsigned pofs[PW];
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  if constexpr (DMA) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wu + i * NW;
      if (q < 5) {                                       // K: chunk (64 q + lane) of the dense [64][5] image
        const int slot = 64 * q + lane, row = slot / 5;
        pofs[i] = (unsigned)((row * p.ldk) * sizeof(T) + (slot - 5 * row) * 16);
      } else if (q < 9) {                                // V03: four chunks per key
        pofs[i] = (unsigned)(((16 * (q - 5) + (lane >> 2)) * p.ldv) * sizeof(T) + (lane & 3) * 16);
      } else {                                           // V4: the fifth chunk of every key
        pofs[i] = (unsigned)((lane * p.ldv) * sizeof(T) + 64);
      }
    }
  }
  auto dma_tile = [&](const int buf, const int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wu + i * NW;
      if (q < Dma40::PIECES) {
        auto dst = (__attribute__((address_space(3))) void*)(smem + buf * Dma40::STAGE + q * 1024);
#if defined(__HIP_DEVICE_COMPILE__)        // (the host pass of the compiler does not know this builtin inside a template)
        if (q < 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, dst, 16, pofs[i], t * ktile, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, dst, 16, pofs[i], t * vtile, 0, 0);
#else
        (void)dst;
#endif
      }
    }
  };
  if constexpr (DMA) {
    // the constant rows of both stages: VC = (1, 0, ..., 0) per key, KC / KC2 = (1, 1, 0, ..., 0)
    if (tid < 2 * KV_TILE) {
      T* vc = reinterpret_cast<T*>(smem + (tid >> 6) * Dma40::STAGE + Dma40::VC + (tid & 63) * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) vc[e] = (T)(e == 0 ? 1.f : 0.f);
    }
    if (tid < 4) {
      T* kc = reinterpret_cast<T*>(smem + (tid >> 1) * Dma40::STAGE + ((tid & 1) ? Dma40::KC2 : Dma40::KC));
#pragma unroll
      for (int e = 0; e < 8; ++e) kc[e] = (T)(e < 2 ? 1.f : 0.f);
    }
  } else {
  // zero the over-read tails of both K buffers once (finite operands for the zero-padded Q dims)
  if (tid < 4) {
    const int buf = tid >> 1, part = tid & 1;
    *reinterpret_cast<u32x4*>(smem + buf * G::STAGE + KV_TILE * G::SK + part * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
  if constexpr (G::FOLD) {
    // columns D, D+1 of every K row = 1 (they meet the (hi, lo) halves of -m in the query fragment), the
    // other spare columns 0; the staging never writes there
    if (tid < 2 * KV_TILE) {
      T* krow = reinterpret_cast<T*>(smem + (tid >> 6) * G::STAGE + (tid & 63) * G::SK + D * 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) krow[e] = (T)(e < 2 ? 1.f : 0.f);
    }
  }
  // Head sizes with spare columns in the last 32-wide output tile: V column D is set to 1 in both
  // buffers (the staging never writes there), so output row D accumulates the softmax denominator on
  // the matrix cores instead of 32 adds per tile and lane; the other spare columns are zeroed.
  if constexpr (G::ONES) {
    if (tid < 2 * KV_TILE) {
      char* vrow = smem + (tid >> 6) * G::STAGE + G::KBUF + (tid & 63) * G::SV + D * 2;
#pragma unroll
      for (int e = 0; e < G::DT * 32 - D; ++e) reinterpret_cast<T*>(vrow)[e] = (T)(e == 0 ? 1.f : 0.f);
    }
  }
  }

  f32x16 o[G::DT32 > 0 ? G::DT32 : 1];
#pragma unroll
  for (int dt = 0; dt < G::DT32; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  // narrow last tile: rows 32 DT32 + 4 (lane / 16) + i of query (lane & 15) [0] and 16 + (lane & 15) [1]
  f32x4 on[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float m_run = -INFINITY, l_run = 0.f;

  if constexpr (DMA) {
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    LOAD_TILE(0)
    STORE_TILE(0)
  }
  __syncthreads();

  // transpose-read lane geometry: 16-lane group g covers d columns (g&1)*16.., key rows 4*(g>>1)..;
  // lane i of the group supplies the address of key row i>>2, columns 4*(i&3)..
  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = ((lane >> 4) & 1) * 16 + gp * 4;
  // LDS-DMA images: lane parts of the fragment addresses (everything else is an immediate)
  const int kl = r * 80 + h * 16;                                   // K row chunk 2 s + h, k-steps 0 and 1
  const int kl2 = h ? Dma40::KC : r * 80 + 64;                      // last k-step: dims 32..39 | the (1, 1) chunk
  const int vl = Dma40::V03 + (4 * h + gq) * 64 + gcol * 2;         // 32-row tile: key 4 h + gq (+ immediates)
  const int vnl = (gp < 2 ? Dma40::V4 + gp * 8 : Dma40::VC + (gp - 2) * 8) +
                  (((lane >> 4) & 1) * 16 + 4 * (lane >> 5) + gq) * 16;   // 16-row tile

  // One key tile; BUF (the LDS buffer it reads, compile time) makes every LDS address an immediate offset.
  auto tile = [&](auto bufc, const int t) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    if (t + 1 < ntiles) {
      if constexpr (DMA) dma_tile(BUF ^ 1, t + 1);
      else { LOAD_TILE(t + 1) }
    }
    const char* kb = smem + BUF * STAGE_B;
    const char* vb = kb + G::KBUF;

    // ---- S^T = K Q^T for the two 32-key blocks ---------------------------------------------------
    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
      v8 a0, a1;
      if constexpr (DMA) {
        const int ko = s == G::KSTEPS - 1 ? kl2 : kl + s * 32;
        a0 = *reinterpret_cast<const v8*>(kb + ko);
        a1 = *reinterpret_cast<const v8*>(kb + ko + 32 * 80);
      } else {
        a0 = *reinterpret_cast<const v8*>(kb + r * G::SK + s * 32 + h * 16);
        a1 = *reinterpret_cast<const v8*>(kb + (32 + r) * G::SK + s * 32 + h * 16);
      }
      s0 = Mfma32<T>::mma(a0, qf[s], s0);
      s1 = Mfma32<T>::mma(a1, qf[s], s1);
    }
    if (t == ntiles - 1 && (p.Nk & (KV_TILE - 1)) != 0) {
      const int kbase_i = t * KV_TILE + 4 * h;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_i + (i & 3) + 8 * (i >> 2);
        if (key >= p.Nk) s0[i] = -INFINITY;
        if (key + 32 >= p.Nk) s1[i] = -INFINITY;
      }
    }

    // the causal mask is a compile-time variant: as a run-time flag the compiler folds it into 32 selects that
    // every tile of every launch pays
    if (CAUSAL && t * KV_TILE + KV_TILE - 1 > qblk * (32 * NW) + wid * 32) {
      const int kbase_i = t * KV_TILE + 4 * h;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_i + (i & 3) + 8 * (i >> 2);
        if (key > qrow) s0[i] = -INFINITY;
        if (key + 32 > qrow) s1[i] = -INFINITY;
      }
    }

    // ---- online softmax (per lane = per query; partner lane ^ 32 holds the other keys) ----------
    // The running maximum only moves when some query of the wave outgrows it by more than 2^8: P stays
    // below 256 (exact range of both 16-bit formats, fp32 sums), and almost every tile skips the
    // rescale of the output accumulators.
    mfma_settle(s0, s1);
    float mx = max3f(s0[0], s0[1], s0[2]), mx1 = max3f(s1[0], s1[1], s1[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) { mx = max3f(mx, s0[i], s0[i + 1]); mx1 = max3f(mx1, s1[i], s1[i + 1]); }
    mx = max3f(mx, s0[15], s1[15]);
    mx = max3f(mx, mx1, mx1);
    v8 pf[4];        // P^T fragments: block 0 k-steps 0,1 then block 1 k-steps 0,1
    if constexpr (G::FOLD) {
      // the accumulators already hold score * c - m_run
      mx = max_halves(mx);
      if (t == 0 || __any(mx > 8.0f)) {
        // first tile: adopt the row maximum; later: move up by the overshoot
        const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < G::DT32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
          if constexpr (G::NARROW) {
            on[0] *= __shfl(alpha, lane & 15, 64);
            on[1] *= __shfl(alpha, 16 + (lane & 15), 64);
          }
        }
        m_run = (t == 0 ? 0.f : m_run) + delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }
        if (h == 1) {
          const T hi = (T)(-m_run);
          const T lo = (T)(-m_run - (float)hi);
          qf[G::KSTEPS - 1][0] = hi;
          qf[G::KSTEPS - 1][1] = lo;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(s0[i]);
        const float e1 = __builtin_amdgcn_exp2f(s1[i]);
        pf[i >> 3][i & 7] = (T)e0;
        pf[2 + (i >> 3)][i & 7] = (T)e1;
      }
    } else {
      mx = max_halves(mx * p.c);
      if (__any(mx > m_run + 8.0f)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < G::DT32; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        if constexpr (G::NARROW) {
          on[0] *= __shfl(alpha, lane & 15, 64);
          on[1] *= __shfl(alpha, 16 + (lane & 15), 64);
        }
      }
      const float m_new = m_run;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(s0[i], p.c, -m_new));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(s1[i], p.c, -m_new));
        if constexpr (!G::ONES) l_run += e0 + e1;
        pf[i >> 3][i & 7] = (T)e0;
        pf[2 + (i >> 3)][i & 7] = (T)e1;
      }
    }

    // ---- O^T += V^T P^T ---------------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int key0 = (f >> 1) * 32 + (f & 1) * 16 + 4 * h + gq;      // first 4-key block; second is +8
#pragma unroll
      for (int dt = 0; dt < G::DT32; ++dt) {
        // (DMA images: dt is 0, key0's lane part sits in vl and rows are 64 bytes apart)
        const char* a0p = DMA ? kb + vl + ((f >> 1) * 32 + (f & 1) * 16) * 64 : vb + key0 * G::SV + (dt * 32 + gcol) * 2;
        constexpr int HI = DMA ? 8 * 64 : 8 * G::SV;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0p + HI));
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = Mfma32<T>::mma(__builtin_bit_cast(v8, both), pf[f], o[dt]);
      }
    }

    if constexpr (G::NARROW) {
      const int g = lane >> 4;
#pragma unroll
      for (int bk = 0; bk < 2; ++bk) {
        u32x4 xu = __builtin_bit_cast(u32x4, pf[2 * bk]), yu = __builtin_bit_cast(u32x4, pf[2 * bk + 1]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const auto sw = __builtin_amdgcn_permlane16_swap(xu[w], yu[w], false, false);
          xu[w] = sw[0]; yu[w] = sw[1];
        }
        // lane group g now holds, for query (lane & 15) of either half, the keys 32 bk + 16 (g & 1) + 4 (g >> 1)
        // + {0..3, 8..11}; the V^T operand is fetched in the same order, rows 32 DT32 + (lane & 15)
        const int key0 = bk * 32 + (g & 1) * 16 + 4 * (g >> 1) + gq;
        const char* a0p = DMA ? kb + vnl + bk * 32 * 16 : vb + key0 * G::SV + (G::DT32 * 32 + gp * 4) * 2;
        constexpr int HI = DMA ? 8 * 16 : 8 * G::SV;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0p + HI));
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const v8 av = __builtin_bit_cast(v8, both);
        on[0] = Mfma16<T>::mma(av, __builtin_bit_cast(v8, xu), on[0]);
        on[1] = Mfma16<T>::mma(av, __builtin_bit_cast(v8, yu), on[1]);
      }
    }

    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the next tile have landed
    } else {
      if (t + 1 < ntiles) STORE_TILE(BUF ^ 1)
    }
    __syncthreads();
  };
  {
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
      tile(std::integral_constant<int, 0>{}, t);
      tile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < ntiles) tile(std::integral_constant<int, 0>{}, t);
  }

  // ---- normalise and store: lane holds d = 32 dt + 8 (i>>2) + 4 h + (i&3) of its query --------------
  if constexpr (G::NARROW) {
    // the ones column is row D - 32 DT32 of the narrow tile: lanes 16 ((D % 32) / 4) .. of register (D % 32) % 4,
    // query (lane & 15) of either half
    constexpr int ROW = D % 32;
    const float la = __shfl(on[0][ROW % 4], 16 * (ROW / 4) + (lane & 15), 64);
    const float lb = __shfl(on[1][ROW % 4], 16 * (ROW / 4) + (lane & 15), 64);
    l_run = (r < 16) ? la : lb;
  } else if constexpr (G::ONES) {
    // output row D (the ones column) sits in the h = 0 half of the wave, register 4 * ((D % 32) / 8)
    static_assert(D % 8 == 0 && (D % 32) % 8 == 0, "ones column must start an 8-row group");
    l_run = __shfl(o[G::DT - 1][4 * ((D % 32) / 8)], r, 64);
  } else {
    l_run += __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_run;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  if (qi < p.Nq) {
    char* op = p.out + ((size_t)((size_t)b * p.Nq + qi) * p.ldo + (size_t)hd * D) * sizeof(T);
#pragma unroll
    for (int dt = 0; dt < G::DT32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = dt * 32 + 8 * g4 + 4 * h;
        if (d0 < D) {
          Tx4 v = {(T)(o[dt][g4 * 4 + 0] * inv), (T)(o[dt][g4 * 4 + 1] * inv), (T)(o[dt][g4 * 4 + 2] * inv),
                   (T)(o[dt][g4 * 4 + 3] * inv)};
          *reinterpret_cast<Tx4*>(op + d0 * sizeof(T)) = v;
        }
      }
  }
  if constexpr (G::NARROW) {
    // lane group g holds dims 32 DT32 + 4 g .. +3 of query (lane & 15) [on[0]] and 16 + (lane & 15) [on[1]]
    const int g = lane >> 4, n = lane & 15;
    const float inv_a = __shfl(inv, n, 64), inv_b = __shfl(inv, 16 + n, 64);
    const int d0 = G::DT32 * 32 + 4 * g;
    if (d0 < D) {
      const int qbase = qi - r;                       // first query of the wave
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int q = qbase + 16 * half + n;
        if (q < p.Nq) {
          const float iv = half ? inv_b : inv_a;
          char* op = p.out + ((size_t)((size_t)b * p.Nq + q) * p.ldo + (size_t)hd * D) * sizeof(T);
          Tx4 v = {(T)(on[half][0] * iv), (T)(on[half][1] * iv), (T)(on[half][2] * iv), (T)(on[half][3] * iv)};
          *reinterpret_cast<Tx4*>(op + d0 * sizeof(T)) = v;
        }
      }
    }
  }
}

#undef LOAD_TILE
#undef STORE_TILE

// Cross-attention over a short key sequence (d = 40, at most 128 keys: the 77 prompt tokens at the 64x64 level).
// With one or two key tiles per query block the kernel above is a chain of latencies per workgroup (query rows, K/V
// staging, barrier, stores) around a microsecond of arithmetic.  Here a workgroup stages BOTH key tiles of its
// (batch, head) once, keeps them in LDS and walks `qpw` consecutive query blocks with them: no staging wait or
// barrier inside the walk, and the query rows of block i + 1 are in flight while block i is multiplied.  Images,
// fragments and arithmetic are those of attn_flash_kernel's d = 40 path (tile t lives in stage t).
// T1: what the second key tile holds -- 0 nothing (at most 64 keys), 1 its first 32-key block only (at most 96
// keys: the 77 prompt tokens; the other block's products are skipped), 2 both blocks.
template <typename T, int NW, int T1>
__global__ __launch_bounds__(64 * NW, 16 / NW) void xattn_resident_kernel(FlashParams p, int qpw, int chunks) {
  constexpr int D = 40;
  typedef Geo<D> G;
  typedef typename Mfma32<T>::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // same XCD-aware order as above, over (batch, head, chunk of query blocks)
  int chunk, bh;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, x = bid & 7;
    const int id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (bid >> 3);
    bh = __builtin_amdgcn_readfirstlane(id / chunks);
    chunk = id - bh * chunks;
  }
  const int b = bh / p.heads, hd = bh - b * p.heads;
  constexpr int ntiles = T1 == 0 ? 1 : 2;

  const char* kbase = p.k + ((size_t)b * p.Nk * p.ldk + (size_t)hd * D) * sizeof(T);
  const char* vbase = p.v + ((size_t)b * p.Nk * p.ldv + (size_t)hd * D) * sizeof(T);
  const unsigned ktile = (unsigned)(KV_TILE * p.ldk * sizeof(T)), vtile = (unsigned)(KV_TILE * p.ldv * sizeof(T));
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)kbase, 0, (int)(((size_t)(p.Nk - 1) * p.ldk + D) * sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)vbase, 0, (int)(((size_t)(p.Nk - 1) * p.ldv + D) * sizeof(T)), 0x00020000);

  // the constant rows of both stages (see Dma40), then both tiles by LDS-DMA, piece q = wave + i NW
  if (tid < 2 * KV_TILE) {
    T* vc = reinterpret_cast<T*>(smem + (tid >> 6) * Dma40::STAGE + Dma40::VC + (tid & 63) * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) vc[e] = (T)(e == 0 ? 1.f : 0.f);
  }
  if (tid < 4) {
    T* kc = reinterpret_cast<T*>(smem + (tid >> 1) * Dma40::STAGE + ((tid & 1) ? Dma40::KC2 : Dma40::KC));
#pragma unroll
    for (int e = 0; e < 8; ++e) kc[e] = (T)(e < 2 ? 1.f : 0.f);
  }
  {
    constexpr int PW = (Dma40::PIECES + NW - 1) / NW;
    const int wu = __builtin_amdgcn_readfirstlane(wid);
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wu + i * NW;
      if (q < Dma40::PIECES) {
        unsigned ofs;
        if (q < 5) {
          const int slot = 64 * q + lane, row = slot / 5;
          ofs = (unsigned)((row * p.ldk) * sizeof(T) + (slot - 5 * row) * 16);
        } else if (q < 9) {
          ofs = (unsigned)(((16 * (q - 5) + (lane >> 2)) * p.ldv) * sizeof(T) + (lane & 3) * 16);
        } else {
          ofs = (unsigned)((lane * p.ldv) * sizeof(T) + 64);
        }
        for (int t = 0; t < ntiles; ++t) {
          auto dst = (__attribute__((address_space(3))) void*)(smem + t * Dma40::STAGE + q * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
          if (q < 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, dst, 16, ofs, t * ktile, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, dst, 16, ofs, t * vtile, 0, 0);
#else
          (void)dst;
#endif
        }
      }
    }
  }

  // Query rows and output rows travel as 16-byte chunks in memory order (chunk c = lane + 64 i of the wave's
  // 32 rows x 80 bytes: a load or store instruction covers 13 rows' contiguous 80 bytes instead of 64 separate
  // lines) and change to and from the fragment layout through a 2560-byte LDS image owned by the wave.
  const int qb_begin = chunk * qpw, qb_end = min(p.qblocks, qb_begin + qpw);
  const char* qhead = p.q + ((size_t)b * p.Nq * p.ldq + (size_t)hd * D) * sizeof(T);
  char* const ohead = p.out + ((size_t)b * p.Nq * p.ldo + (size_t)hd * D) * sizeof(T);
  char* const wimg = smem + 2 * Dma40::STAGE + wid * (32 * 80);
  constexpr int QCH = 3;                                     // 160 chunks = 2.5 x 64 lanes
  int crow[QCH], cofs[QCH];
#pragma unroll
  for (int i = 0; i < QCH; ++i) {
    const int c = lane + 64 * i;
    crow[i] = c / 5;
    cofs[i] = (c - 5 * crow[i]) * 16;
  }
  u32x4 qraw[QCH];
#pragma unroll
  for (int i = 0; i < QCH; ++i) qraw[i] = (u32x4){0u, 0u, 0u, 0u};
  auto load_q = [&](const int qb) __attribute__((always_inline)) {
    const int row0 = qb * (32 * NW) + wid * 32;
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
      if (lane + 64 * i < 160) {
        const int qrow = min(row0 + crow[i], p.Nq - 1);
        qraw[i] = *reinterpret_cast<const u32x4*>(qhead + (size_t)qrow * p.ldq * sizeof(T) + cofs[i]);
      }
    }
  };
  load_q(qb_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = ((lane >> 4) & 1) * 16 + gp * 4;
  const int kl = r * 80 + h * 16;
  const int kl2 = h ? Dma40::KC : r * 80 + 64;
  const int vl = Dma40::V03 + (4 * h + gq) * 64 + gcol * 2;
  const int vnl = (gp < 2 ? Dma40::V4 + gp * 8 : Dma40::VC + (gp - 2) * 8) +
                  (((lane >> 4) & 1) * 16 + 4 * (lane >> 5) + gq) * 16;
  typedef T Tx4 __attribute__((ext_vector_type(4)));

  for (int qb = qb_begin; qb < qb_end; ++qb) {
    const int row0 = qb * (32 * NW) + wid * 32;
#pragma unroll
    for (int i = 0; i < QCH; ++i)
      if (lane + 64 * i < 160) *reinterpret_cast<u32x4*>(wimg + 16 * (lane + 64 * i)) = qraw[i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    v8 qf[G::KSTEPS];
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
      const int d0 = 16 * s + 8 * h;
      if (d0 < D) {
        const v8 raw = *reinterpret_cast<const v8*>(wimg + r * 80 + d0 * sizeof(T));
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (T)((float)raw[e] * p.c);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (T)0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (qb + 1 < qb_end) load_q(qb + 1);

    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    f32x4 on[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float m_run = 0.f;

    // HALF: only the first 32 keys of the tile exist (77 keys: 13 of them in tile 1), the second key block is skipped
    auto tile = [&](auto tc, auto hc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr bool HALF = decltype(hc)::value;
      const char* kb = smem + t * Dma40::STAGE;
      f32x16 s0, s1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < G::KSTEPS; ++s) {
        const int ko = s == G::KSTEPS - 1 ? kl2 : kl + s * 32;
        const v8 a0 = *reinterpret_cast<const v8*>(kb + ko);
        const v8 a1 = *reinterpret_cast<const v8*>(kb + ko + 32 * 80);
        s0 = Mfma32<T>::mma(a0, qf[s], s0);
        if constexpr (!HALF) s1 = Mfma32<T>::mma(a1, qf[s], s1);
      }
      if (t == ntiles - 1 && (p.Nk & (KV_TILE - 1)) != 0) {
        const int kbase_i = t * KV_TILE + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kbase_i + (i & 3) + 8 * (i >> 2);
          if (key >= p.Nk) s0[i] = -INFINITY;
          if (key + 32 >= p.Nk) s1[i] = -INFINITY;
        }
      }
      // the accumulators hold score * c - m_run (the maximum rides through the last k-step, see Geo::FOLD)
      if constexpr (HALF) mfma_settle(s0); else mfma_settle(s0, s1);
      float mx = max3f(s0[0], s0[1], s0[2]);
      if constexpr (HALF) {
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = max3f(mx, s0[i], s0[i + 1]);
        mx = max3f(mx, s0[15], s0[15]);
      } else {
        float mx1 = max3f(s1[0], s1[1], s1[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) { mx = max3f(mx, s0[i], s0[i + 1]); mx1 = max3f(mx1, s1[i], s1[i + 1]); }
        mx = max3f(mx, s0[15], s1[15]);
        mx = max3f(mx, mx1, mx1);
      }
      mx = max_halves(mx);
      if (t == 0 || __any(mx > 8.0f)) {
        const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int i = 0; i < 16; ++i) o[i] *= alpha;
          on[0] *= __shfl(alpha, lane & 15, 64);
          on[1] *= __shfl(alpha, 16 + (lane & 15), 64);
        }
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] -= delta; if constexpr (!HALF) s1[i] -= delta; }
        if (h == 1) {
          const T hi = (T)(-m_run);
          const T lo = (T)(-m_run - (float)hi);
          qf[G::KSTEPS - 1][0] = hi;
          qf[G::KSTEPS - 1][1] = lo;
        }
      }
      v8 pf[4];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        pf[i >> 3][i & 7] = (T)__builtin_amdgcn_exp2f(s0[i]);
        if constexpr (!HALF) pf[2 + (i >> 3)][i & 7] = (T)__builtin_amdgcn_exp2f(s1[i]);
      }
#pragma unroll
      for (int f = 0; f < (HALF ? 2 : 4); ++f) {
        const char* a0p = kb + vl + ((f >> 1) * 32 + (f & 1) * 16) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0p + 8 * 64));
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o = Mfma32<T>::mma(__builtin_bit_cast(v8, both), pf[f], o);
      }
#pragma unroll
      for (int bk = 0; bk < (HALF ? 1 : 2); ++bk) {
        u32x4 xu = __builtin_bit_cast(u32x4, pf[2 * bk]), yu = __builtin_bit_cast(u32x4, pf[2 * bk + 1]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const auto sw = __builtin_amdgcn_permlane16_swap(xu[w], yu[w], false, false);
          xu[w] = sw[0]; yu[w] = sw[1];
        }
        const char* a0p = kb + vnl + bk * 32 * 16;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0p + 8 * 16));
        s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const v8 av = __builtin_bit_cast(v8, both);
        on[0] = Mfma16<T>::mma(av, __builtin_bit_cast(v8, xu), on[0]);
        on[1] = Mfma16<T>::mma(av, __builtin_bit_cast(v8, yu), on[1]);
      }
    };
    tile(std::integral_constant<int, 0>{}, std::false_type{});
    if constexpr (T1 == 1) tile(std::integral_constant<int, 1>{}, std::true_type{});
    // (two full tiles in one basic block make the register allocator spill: the branch, always taken, keeps them apart)
    if constexpr (T1 == 2) { if (p.Nk > KV_TILE) tile(std::integral_constant<int, 1>{}, std::false_type{}); }

    // normalise and store (the ones column is row 8 of the narrow tile: lanes 32.., register 0)
    const float la = __shfl(on[0][0], 32 + (lane & 15), 64);
    const float lb = __shfl(on[1][0], 32 + (lane & 15), 64);
    const float inv = 1.0f / ((r < 16) ? la : lb);
    // fragments -> the wave's image (rows of queries that do not exist are written too and never stored)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      Tx4 v = {(T)(o[g4 * 4 + 0] * inv), (T)(o[g4 * 4 + 1] * inv), (T)(o[g4 * 4 + 2] * inv), (T)(o[g4 * 4 + 3] * inv)};
      *reinterpret_cast<Tx4*>(wimg + r * 80 + (8 * g4 + 4 * h) * sizeof(T)) = v;
    }
    {
      const int g = lane >> 4, n = lane & 15;
      const float inv_a = __shfl(inv, n, 64), inv_b = __shfl(inv, 16 + n, 64);
      if (g < 2) {                                   // dims 32 + 4 g .. + 3 of queries n and 16 + n
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float iv = half ? inv_b : inv_a;
          Tx4 v = {(T)(on[half][0] * iv), (T)(on[half][1] * iv), (T)(on[half][2] * iv), (T)(on[half][3] * iv)};
          *reinterpret_cast<Tx4*>(wimg + (16 * half + n) * 80 + (32 + 4 * g) * sizeof(T)) = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
      if (lane + 64 * i < 160 && row0 + crow[i] < p.Nq) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(wimg + 16 * (lane + 64 * i));
        *reinterpret_cast<u32x4*>(ohead + (size_t)(row0 + crow[i]) * p.ldo * sizeof(T) + cofs[i]) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

template <typename T, int D>
int launch_flash(const AttentionArgs& a, hipStream_t stream) {
  typedef Geo<D> G;
  static_assert(D % 8 == 0 && G::DT * 64 <= G::SV && D * 2 <= G::SV, "bad head size geometry");
  static_assert(!G::FOLD || (G::ONES && G::KSTEPS * 32 <= G::SK), "folded maximum needs spare K and V columns");
  FlashParams p;
  p.q = (const char*)a.q; p.k = (const char*)a.k; p.v = (const char*)a.v; p.out = (char*)a.out;
  p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
  p.B = a.B; p.heads = a.heads; p.Nq = a.Nq; p.Nk = a.Nk;
  p.c = a.scale * 1.4426950408889634f;
  p.causal = a.causal;
  // (d = 40 non-causal: the LDS-DMA images; the causal variant keeps the register-staged layout)
  const int lds = 2 * ((D == 40 && !a.causal) ? Dma40::STAGE : G::STAGE);
  // d = 40 at 1024..4096 tokens: 8 waves (256 queries) share every staged K/V tile -- half the staging traffic and
  // barriers per query, measured -5 % at 4096 tokens (16 waves: no better; other head sizes and 9216 tokens: equal or
  // worse).  SDMI_ATTN_WAVES=4/8 overrides for head sizes up to 80 (experiments; read on every call)
  if constexpr (D == 40) {
    if (!a.causal && a.Nk <= 2 * KV_TILE && a.ldo % 8 == 0) {
      // short key sequences with enough query blocks to give every workgroup several: K/V stay in LDS for a walk of
      // qpw blocks (about 1024 workgroups in all, four per CU).  SDMI_XATTN_QPW overrides (experiments; 1 = off)
      p.qblocks = (a.Nq + 127) / 128;
      const long blocks = (long)p.qblocks * a.B * a.heads;
      int qpw = (int)(blocks / 1024);
      if (qpw > 8) qpw = 8;
      if (const char* e = getenv("SDMI_XATTN_QPW")) { const int v = atoi(e); if (v >= 1 && v <= 64) qpw = v; }
      if (qpw >= 2) {
        const int chunks = (p.qblocks + qpw - 1) / qpw;
        const dim3 grid(chunks * a.B * a.heads), block(256);
        const int lds = 2 * Dma40::STAGE + 4 * 32 * 80;
        if (a.Nk <= KV_TILE) hipLaunchKernelGGL((xattn_resident_kernel<T, 4, 0>), grid, block, lds, stream, p, qpw, chunks);
        else if (a.Nk <= KV_TILE + 32) hipLaunchKernelGGL((xattn_resident_kernel<T, 4, 1>), grid, block, lds, stream, p, qpw, chunks);
        else hipLaunchKernelGGL((xattn_resident_kernel<T, 4, 2>), grid, block, lds, stream, p, qpw, chunks);
        SDMI_CHECK_HIP(hipGetLastError());
        return SDMI_OK;
      }
    }
  }
  int nw = (D == 40 && a.Nq >= 1024 && a.Nq <= 4096 && a.Nk >= 1024) ? 8 : 4;
  if (const char* e = getenv("SDMI_ATTN_WAVES")) { const int v = atoi(e); if ((v == 4 || v == 8) && D <= 80) nw = v; }
  if (a.causal) nw = 4;                    // the causal variant exists for 4-wave workgroups only
  p.qblocks = (a.Nq + 32 * nw - 1) / (32 * nw);
  dim3 grid(p.qblocks * a.B * a.heads);
  if (a.causal) {
    hipLaunchKernelGGL((attn_flash_kernel<T, D, 4, true>), grid, dim3(256), lds, stream, p);
  } else if (nw == 8) {
    if constexpr (D <= 80) hipLaunchKernelGGL((attn_flash_kernel<T, D, 8, false>), grid, dim3(512), lds, stream, p);
  } else {
    hipLaunchKernelGGL((attn_flash_kernel<T, D, 4, false>), grid, dim3(256), lds, stream, p);
  }
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

template <typename T>
int dispatch_flash(const AttentionArgs& a, hipStream_t stream, bool* handled) {
  *handled = true;
  switch (a.d) {
    case 40: return launch_flash<T, 40>(a, stream);
    case 64: return launch_flash<T, 64>(a, stream);
    case 80: return launch_flash<T, 80>(a, stream);
    case 128: return launch_flash<T, 128>(a, stream);
    case 160: return launch_flash<T, 160>(a, stream);
    default: *handled = false; return SDMI_OK;
  }
}

}  // namespace

int attention(const AttentionArgs& a, hipStream_t stream) {
  SDMI_REQUIRE(a.q && a.k && a.v && a.out, "attention: null pointer");
  SDMI_REQUIRE(a.B > 0 && a.heads > 0 && a.Nq > 0 && a.Nk > 0 && a.d > 0, "attention: empty problem");
  char label[64];
  snprintf(label, sizeof(label), "B%d h%d Nq%d Nk%d d%d", a.B, a.heads, a.Nq, a.Nk, a.d);
  ProfScope prof(PROF_ATTENTION, stream, 4.0 * a.B * a.heads * (double)a.Nq * a.Nk * a.d, 0.0, label);
  if (!(a.impl & 1) && a.dtype != SDMI_F32) {
    SDMI_TRY(attention_init());
    SDMI_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0,
                 "attention: row strides must be multiples of 8 elements");
    SDMI_REQUIRE((size_t)a.Nk * (size_t)(a.ldk > a.ldv ? a.ldk : a.ldv) * 2 < ((size_t)1 << 31),
                 "attention: one batch element's K or V spans 2 GiB or more");
    bool handled = false;
    int rc;
    if (a.dtype == SDMI_F16) rc = dispatch_flash<half_t>(a, stream, &handled);
    else rc = dispatch_flash<bf16_t>(a, stream, &handled);
    if (handled) return rc;
  }
  if (!(a.impl & 1) && a.dtype == SDMI_F32) return attention_f32(a, stream);
  return attention_scalar(a, stream);
}

namespace {
template <typename T, int D>
int init_one() {
  const int lds = 2 * Geo<D>::STAGE;
  if (lds > 48 * 1024)
  {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_flash_kernel<T, D, 4, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_flash_kernel<T, D, 4, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (D <= 80)
      SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_flash_kernel<T, D, 8, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  return SDMI_OK;
}
template <typename T>
int init_all() {
  SDMI_TRY((init_one<T, 40>())); SDMI_TRY((init_one<T, 64>())); SDMI_TRY((init_one<T, 80>()));
  SDMI_TRY((init_one<T, 128>())); SDMI_TRY((init_one<T, 160>()));
  return SDMI_OK;
}
}  // namespace

// Raise dynamic-LDS limits once, outside of stream capture (called from finalize and lazily).
int attention_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  SDMI_TRY(init_all<half_t>());
  SDMI_TRY(init_all<bf16_t>());
  SDMI_TRY(attention_f32_init());
  done = true;
  return SDMI_OK;
}
