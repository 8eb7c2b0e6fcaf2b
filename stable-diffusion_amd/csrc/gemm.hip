// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4).
//
//   out[m][n] = epilogue( sum_k A(m,k) * Wp[n][k] )
//
// m walks output pixels (b, oy, ox) of an NHWC activation, n output channels, k = (tap, input
// channel); the input may be the channel-concatenation of two tensors (UNet skip connections), may
// be read with stride 2 (Downsample) or through an on-the-fly nearest x2 upsample.  taps = 1 makes
// it a plain token GEMM (Linear, 1x1 conv).
//
// Tile: 128(m) x 128(n) per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA
// 16x16 tiles with fp32 accumulators.  K-tile = 8 chunks of 16 bytes per row (64 halves / 32
// floats).  Both operand tiles are staged global -> LDS by `global_load_lds_dwordx4` (no VGPR round
// trip); out-of-image taps and out-of-range rows read a zero page instead, so the halo costs no
// branch in the MFMA loop.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with
// (row & 7) -- applied on the per-lane *source* address, because LDS-DMA writes lane-linearly -- which
// makes the ds_read_b128 fragment reads bank-conflict free.  Two LDS buffers: the DMA of tile k+1
// is in flight while tile k feeds the MFMAs; one barrier per K-tile.
//
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so a lane ends
// up holding 4 consecutive output channels of one pixel: the epilogue (bias, per-sample time-
// embedding add, residual add, GEGLU gate) works on 8/16-byte vectors.
//
// Small-M layers (8x8 / 16x16 feature maps) cannot fill 256 CUs with 128x128 tiles, so the K range
// can be split over blockIdx.y; partial sums go to an fp32 slab and a second kernel reduces them
// and applies the epilogue.
#include "ops.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

// Block shapes: WM x WN waves, each wave owns a 64x64 output tile; NSTAGE LDS stages of
// (BM + BN) rows x 128 bytes.  The DMA of tile k+NSTAGE-1 is issued while tile k is multiplied.
template <int WM_, int WN_, int NSTAGE_> struct Shape {
  static constexpr int WM = WM_, WN = WN_, NSTAGE = NSTAGE_;
  static constexpr int NW = WM * WN, NTHREADS = 64 * NW;
  static constexpr int BM = 64 * WM, BN = 64 * WN;
  static constexpr int AP = (BM / 8) / NW, WP = (BN / 8) / NW;     // 1-KiB DMA pieces per wave and tile
  static constexpr int GL = AP + WP;                                // global_load_lds per wave and tile
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide evenly over the waves");
};
typedef Shape<2, 2, 2> ShapeSmall;      // 128x128, 4 waves, 64 KiB: two blocks per CU (small-M layers)


typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmParams {
  const char* a0; const char* a1; int c0, c1;
  const char* e0; const char* e1; int ec0, ec1;   // extra pointwise sources behind the taps: K = taps (c0 + c1) + ec0 + ec1,
                                                  // their rows are the OUTPUT pixels (a 1x1 convolution summed into this one)
  const char* w; char* out; int ldo;
  const float* bias; const float* rowadd; const char* resid; int ldr; int ldra; int out_f32;
  int B, Hin, Win, Ho, Wo, taps, stride, up;
  int pad;            // zero rows/columns ahead of the image (3x3 taps): 1, or 0 for the VAE encoder's (0,1,0,1) padding
  int M, N, K, geglu;
  int nkt, kt_per_split;
  float* partial;
  int tiles_m, tiles_n;
  const char* zero;   // 256 zero bytes (halo / out-of-range rows read these)
  int dup_rows;       // > 0: rows are stored a second time dup_rows further down
  int wide_store;     // large-tile kernel: 16-bit output written through the LDS strips
  const float* ln_stats;   // LayerNorm folded in: per row (rstd, rstd * mean); null = plain GEMM
  const float* ln_colsum;  // ... and per column sum_k w[n][k]; `bias` is then b' and is applied after the scaling
  int ln_nparts;           // 0: ln_stats holds (rstd, rstd * mean) per row; P > 0: P pairs (sum, sum of squares) per
  float ln_inv_d, ln_eps;  //    row, as a producing GEMM left them (row length 1 / ln_inv_d)
  float2* rs_out;          // this launch leaves row sums of its OUTPUT for a LayerNorm behind it: rs_parts pairs
  int rs_parts;            //    (sum, sum of squares of the stored values) per row; null = none
  int w_rows;              // > 0: one weight matrix per sample, rows m belong to sample m / w_rows (a tile never
  size_t w_stride;         //    straddles two samples: the launcher picks BM | w_rows); bytes between the matrices
  int vec_stride;          //    floats between the samples' bias / ln_colsum vectors (0: one vector for all)
  int softmax_seg;         // 1: every 80-column segment of an output row is replaced by its softmax (base 2)
};

// (rstd, rstd * mean) of row m for a GEMM with a folded LayerNorm
__device__ __forceinline__ float2 ln_row(const GemmParams& p, int m) {
  const float2* st = reinterpret_cast<const float2*>(p.ln_stats);
  if (p.ln_nparts == 0) return st[m];
  st += (size_t)m * p.ln_nparts;
  float s, q;
  if (p.ln_nparts == 1) {
    s = st[0].x; q = st[0].y;
  } else {
    // an even number of pairs, 16 at most: eight 16-byte loads in flight (the ones beyond the row's pairs re-read
    // its last and count for nothing), summed in slot order
    const float4* st4 = reinterpret_cast<const float4*>(st);
    const int last = p.ln_nparts / 2 - 1;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = st4[min(k, last)];
    s = 0.f; q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k <= last) { s += v[k].x; q += v[k].y; s += v[k].z; q += v[k].w; }
    }
  }
  const float mean = s * p.ln_inv_d;
  const float rstd = rsqrtf(fmaxf(q * p.ln_inv_d - mean * mean, 0.f) + p.ln_eps);
  return make_float2(rstd, rstd * mean);
}

template <typename T> struct Frag;      // 16 bytes of T as an MFMA operand fragment
template <> struct Frag<half_t> { typedef half_t type __attribute__((ext_vector_type(8))); };
template <> struct Frag<bf16_t> { typedef bf16_t type __attribute__((ext_vector_type(8))); };
template <> struct Frag<float>  { typedef float  type __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ f32x4 mma(Frag<half_t>::type a, Frag<half_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(Frag<bf16_t>::type a, Frag<bf16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// fp32: one 16-byte fragment = 4 k-values per lane = 4 MFMA 16x16x4 steps.  Lane group q (=lane>>4)
// of step s covers k = 4q+s on both operands, so the permuted k order cancels out.
__device__ __forceinline__ f32x4 mma(Frag<float>::type a, Frag<float>::type b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// Shared epilogue: v[4] are fp32 sums for columns n..n+3 of row m.
template <typename T>
__device__ __forceinline__ f32x4 epilogue_value(const GemmParams& p, int m, int n, f32x4 v) {
  if (p.bias) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
    v += b;
  }
  if (p.rowadd) {
    const int bidx = m / (p.Ho * p.Wo);
    v += *reinterpret_cast<const f32x4*>(p.rowadd + (size_t)bidx * p.ldra + n);
  }
  if (p.resid) {
    const T* r = reinterpret_cast<const T*>(p.resid) + (size_t)m * p.ldr + n;
    v.x += to_f32(r[0]); v.y += to_f32(r[1]); v.z += to_f32(r[2]); v.w += to_f32(r[3]);
  }
  return v;
}
template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, f32x4 v) {
  if (p.bias) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
    v += b;
  }
  if (p.rowadd) {
    const int bidx = m / (p.Ho * p.Wo);
    v += *reinterpret_cast<const f32x4*>(p.rowadd + (size_t)bidx * p.ldra + n);
  }
  if (p.resid) {
    const T* r = reinterpret_cast<const T*>(p.resid) + (size_t)m * p.ldr + n;
    v.x += to_f32(r[0]); v.y += to_f32(r[1]); v.z += to_f32(r[2]); v.w += to_f32(r[3]);
  }
  if (p.out_f32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n) = v;
    return;
  }
  T* o = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  Tx4 ov = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
  *reinterpret_cast<Tx4*>(o) = ov;
  if (p.dup_rows > 0) *reinterpret_cast<Tx4*>(o + (size_t)p.dup_rows * p.ldo) = ov;
}

// GEGLU: value and gate sums for the same 4 output columns.
template <typename T>
__device__ __forceinline__ void epilogue_store_geglu(const GemmParams& p, int m, int n_val, int n_gate,
                                                     int n_out, f32x4 val, f32x4 gate) {
  if (p.bias) {
    val += *reinterpret_cast<const f32x4*>(p.bias + n_val);
    gate += *reinterpret_cast<const f32x4*>(p.bias + n_gate);
  }
  const f32x4 v = geglu4_for<T>(val, gate);
  T* o = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n_out;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  Tx4 ov = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
  *reinterpret_cast<Tx4*>(o) = ov;
}

// LayerNorm folded into the GEMM: the operand rows were the raw x and the weights carry gamma, so the sums
// S_mn become rstd_m * (S_mn - mean_m * colsum_n) + b'_n.  Applied to the accumulators in registers ahead of
// the epilogue proper; acc[i][j] holds columns n_lane + 16 i .. +3 of row m_lane + 16 j.
template <int NI, int MJ>
__device__ __forceinline__ void ln_fold_acc(const GemmParams& p, f32x4 (&acc)[NI][MJ], int m_lane, int n_lane) {
  float r[MJ], rm[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const float2 st = ln_row(p, min(m_lane + 16 * j, p.M - 1));
    r[j] = st.x; rm[j] = st.y;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = min(n_lane + 16 * i, p.N - 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = r[j] * acc[i][j] + (b - rm[j] * c);
  }
}

// Epilogue of one wave: acc[i][j] holds columns n_wave + 16 i + 4 fq .. +3 of row m_lane + 16 j.
template <typename T, int NI, int MJ>
__device__ __forceinline__ void store_wave_tile(const GemmParams& p, f32x4 (&acc)[NI][MJ], int m_lane, int n_wave,
                                                int fq) {
  const int nbase = n_wave + fq * 4;
  if (p.partial) {
    float* slab = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int m = m_lane + j * 16;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = nbase + i * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  if constexpr (NI % 2 == 0) {
    if (p.geglu) {
      const int obase = n_wave / 2 + fq * 4;
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int m = m_lane + j * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
          const int n = nbase + i * 16;
          if (n < p.N) epilogue_store_geglu<T>(p, m, n, n + 16, obase + (i / 2) * 16, acc[i][j], acc[i + 1][j]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int m = m_lane + j * 16;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nbase + i * 16;
      if (n < p.N) epilogue_store<T>(p, m, n, acc[i][j]);
    }
  }
}

// Epilogue through LDS: the MFMA layout gives a lane 4 channels of one pixel, so direct stores reach
// memory as 8- or 16-byte pieces scattered over 16 rows.  Here the wave parks 16 rows at a time in a
// private fp32 LDS strip and writes them back out as 16-byte pieces that run along the channel rows
// (GEGLU gate applied on the way in, the residual on the way out, in fp32, one rounding).  OutT is
// the stored type: T, or float for split-K slabs and fp32 outputs.  `strip` must hold
// 16 * (16 * NO + 4) floats, NO = output fragments per row.
// With `psum` (16 * CH float2 of LDS private to the wave) the wave also leaves the row sums of what it stores,
// slot `rs_slot` of the rs_parts pairs of every row: a lane sums its pieces, parks the pair in LDS and lanes 0..15
// fold one row each in piece order.
// The same for 16-bit outputs WITHOUT a residual: nothing is added between the accumulators and the store, so the
// values are rounded on the way INTO the strip (same results, bit for bit) and the strip holds 16-bit rows: half the
// LDS bytes, 8-byte ds_write in place of the 16-byte one (6-8 cycles against 13 per four values), and the pieces
// come back out as stored vectors with nothing left to convert.
template <typename T, int NI, bool G, int MJ>
__device__ __forceinline__ void store_wave_tile_lds16(const GemmParams& p, f32x4 (&acc)[NI][MJ], T* strip, T* out,
                                                      int ldo, int m_wave, int n_wave, int lane, float2* psum,
                                                      int rs_slot) {
  constexpr int NO = G ? NI / 2 : NI;
  constexpr int ROWH = NO * 16 + 8;             // row stride in elements (16-byte aligned rows)
  constexpr int CH = NO * 2;                    // 16-byte pieces per row
  constexpr int TOTAL = 16 * CH;
  constexpr int KI = (TOTAL + 63) / 64;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  typedef T Tx8 __attribute__((ext_vector_type(8)));
  const int frow = lane & 15, fq = lane >> 4;
  const int n_out = G ? n_wave / 2 : n_wave;
  int c_row[KI], c_col[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int c = min(lane + 64 * k, TOTAL - 1);
    c_row[k] = c / CH;
    c_col[k] = (c - c_row[k] * CH) * 8;
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
#pragma unroll
    for (int io = 0; io < NO; ++io) {
      f32x4 v;
      if constexpr (G) v = geglu4_for<T>(acc[2 * io][j], acc[2 * io + 1][j]);
      else v = acc[io][j];
      const Tx4 h = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
      *reinterpret_cast<Tx4*>(strip + frow * ROWH + io * 16 + fq * 4) = h;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int m = m_wave + j * 16 + c_row[k];
      const Tx8 o = *reinterpret_cast<const Tx8*>(strip + c_row[k] * ROWH + c_col[k]);
      if (psum) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = to_f32(o[e]); ps += f; pq = fmaf(f, f, pq); }
        if (lane + 64 * k < TOTAL) psum[lane + 64 * k] = make_float2(ps, pq);
      }
      if (lane + 64 * k < TOTAL && m < p.M) {
        *reinterpret_cast<Tx8*>(out + (size_t)m * ldo + n_out + c_col[k]) = o;
        if (p.dup_rows > 0) *reinterpret_cast<Tx8*>(out + (size_t)(m + p.dup_rows) * ldo + n_out + c_col[k]) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (psum) {
      if (lane < 16) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) { const float2 v = psum[lane * CH + c]; ps += v.x; pq += v.y; }
        const int m = m_wave + j * 16 + lane;
        if (m < p.M) {
          p.rs_out[(size_t)m * p.rs_parts + rs_slot] = make_float2(ps, pq);
          if (p.dup_rows > 0) p.rs_out[(size_t)(m + p.dup_rows) * p.rs_parts + rs_slot] = make_float2(ps, pq);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename T, typename OutT, int NI, bool G, int MJ = 4>
__device__ __forceinline__ void store_wave_tile_lds(const GemmParams& p, f32x4 (&acc)[NI][MJ], float* strip,
                                                    OutT* out, int ldo, int m_wave, int n_wave, int lane,
                                                    float2* psum = nullptr, int rs_slot = 0) {
  if constexpr (sizeof(OutT) == 2 && sizeof(T) == 2) {
    if (p.resid == nullptr) {
      store_wave_tile_lds16<T, NI, G, MJ>(p, acc, reinterpret_cast<T*>(strip), reinterpret_cast<T*>(out), ldo, m_wave,
                                          n_wave, lane, psum, rs_slot);
      return;
    }
  }
  constexpr int NO = G ? NI / 2 : NI;
  constexpr int ROWF = NO * 16 + 4;
  constexpr int EPC = 16 / (int)sizeof(OutT);   // elements per 16-byte piece
  constexpr int CH = NO * 16 / EPC;             // pieces per row
  constexpr int TOTAL = 16 * CH;
  constexpr int KI = (TOTAL + 63) / 64;         // pieces per lane and 16-row strip
  typedef T Tres __attribute__((ext_vector_type(EPC)));
  typedef OutT Tout __attribute__((ext_vector_type(EPC)));
  const int frow = lane & 15, fq = lane >> 4;
  const int n_out = G ? n_wave / 2 : n_wave;
  // Bias and time-embedding terms are already in the accumulators (the kernel starts from them).
  // The residual loads are issued unconditionally on clamped addresses, a strip's worth at a time,
  // ahead of the LDS round trip that hides their latency.
  int c_row[KI], c_col[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int c = min(lane + 64 * k, TOTAL - 1);
    c_row[k] = c / CH;
    c_col[k] = (c - c_row[k] * CH) * EPC;
  }
  const bool has_resid = p.resid != nullptr;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    Tres res[KI];
    if (has_resid) {
#pragma unroll
      for (int k = 0; k < KI; ++k) {
        const int m = min(m_wave + j * 16 + c_row[k], p.M - 1);
        res[k] = *reinterpret_cast<const Tres*>(reinterpret_cast<const T*>(p.resid) + (size_t)m * p.ldr + n_out +
                                                c_col[k]);
      }
    }
#pragma unroll
    for (int io = 0; io < NO; ++io) {
      f32x4 v;
      if constexpr (G) {
        const f32x4 val = acc[2 * io][j], gate = acc[2 * io + 1][j];
        v = geglu4_for<T>(val, gate);
      } else {
        v = acc[io][j];
      }
      *reinterpret_cast<f32x4*>(strip + frow * ROWF + io * 16 + fq * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int m = m_wave + j * 16 + c_row[k];
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(strip + c_row[k] * ROWF + c_col[k] + e);
        v[e] = q.x; v[e + 1] = q.y; v[e + 2] = q.z; v[e + 3] = q.w;
      }
      if (has_resid) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] += to_f32(res[k][e]);
      }
      Tout o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) o[e] = from_f32<OutT>(v[e]);
      if (psum) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float f = to_f32(o[e]); ps += f; pq = fmaf(f, f, pq); }
        if (lane + 64 * k < TOTAL) psum[lane + 64 * k] = make_float2(ps, pq);
      }
      if (lane + 64 * k < TOTAL && m < p.M) {
        *reinterpret_cast<Tout*>(out + (size_t)m * ldo + n_out + c_col[k]) = o;
        if (p.dup_rows > 0) *reinterpret_cast<Tout*>(out + (size_t)(m + p.dup_rows) * ldo + n_out + c_col[k]) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (psum) {
      if (lane < 16) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) { const float2 v = psum[lane * CH + c]; ps += v.x; pq += v.y; }
        const int m = m_wave + j * 16 + lane;
        if (m < p.M) {
          p.rs_out[(size_t)m * p.rs_parts + rs_slot] = make_float2(ps, pq);
          if (p.dup_rows > 0) p.rs_out[(size_t)(m + p.dup_rows) * p.rs_parts + rs_slot] = make_float2(ps, pq);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(S::NTHREADS) void conv_gemm_kernel(GemmParams p) {
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                // k elements per tile row (128 bytes)
  constexpr int BM = S::BM, BN = S::BN, AP = S::AP, WP = S::WP, NSTAGE = S::NSTAGE;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / S::WN, wn = wid - wm * S::WN;

  // XCD-aware tile order: workgroups dealt to the same XCD (bid % 8) take a contiguous range of
  // tile ids, n fastest, so that they share A panels and keep the weight panel in that XCD's L2.
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_n = id % p.tiles_n, tile_m = id / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int kt_end = min(p.nkt, kt_begin + p.kt_per_split);

  // ---- loader state: AP activation rows + WP weight rows per thread ---------------------------
  const int lrow = lane >> 3;                        // row within an 8-row DMA piece
  const int sc = (lane & 7) ^ lrow;                  // swizzled source chunk for LDS chunk lane&7
  const int hw = p.Ho * p.Wo;
  const bool pointwise = (p.taps == 1 && p.stride == 1 && !p.up);
  int a_b[AP], a_oy[AP], a_ox[AP];
  bool a_ok[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + (wid * AP + i) * 8 + lrow;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (pointwise) {
      a_b[i] = 0; a_oy[i] = 0; a_ox[i] = mm;          // pointwise: pixel index = row index
    } else {
      const int b = mm / hw, rem = mm - b * hw;
      a_b[i] = b; a_oy[i] = rem / p.Wo; a_ox[i] = rem - (rem / p.Wo) * p.Wo;
    }
  }
  const char* wptr[WP];
  int wstep[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int n = n0 + (wid * WP + i) * 8 + lrow;
    const bool wok = n < p.N;
    wptr[i] = wok ? p.w + ((size_t)n * p.K) * sizeof(T) + sc * 16 : p.zero + sc * 16;
    wstep[i] = wok ? 128 : 0;
  }
  const int tiles_c0 = p.c0 / BK, tiles_c1 = p.c1 / BK, tiles_tap = tiles_c0 + tiles_c1;
  const int Hup = p.up ? 2 * p.Hin : p.Hin, Wup = p.up ? 2 * p.Win : p.Win;
  const char* const zero = p.zero + sc * 16;

  // Prefetch cursor: K-tiles are visited segment by segment (tap-major, then source tensor); inside
  // a segment every row pointer just advances by one 128-byte tile row, so the steady-state cost of
  // issuing a tile is one 64-bit add per DMA.  Row pointers are rebuilt only at segment changes.
  const char* acur[AP];
  int astep[AP];
  int pf_tap, pf_src, pf_left;      // current segment of the prefetcher and tiles left in it
  auto set_segment = [&](int tap, int src, int cseg) {
    pf_tap = tap; pf_src = src;
    pf_left = (src ? tiles_c1 : tiles_c0) - cseg;
    const char* base = src ? p.a1 : p.a0;
    const int cs = src ? p.c1 : p.c0;
    const int dy = p.taps == 9 ? tap / 3 - p.pad : 0, dx = p.taps == 9 ? tap - (tap / 3) * 3 - p.pad : 0;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      bool ok = a_ok[i];
      size_t pix;
      if (pointwise) {
        pix = (size_t)a_ox[i];
      } else {
        int iy = a_oy[i] * p.stride + dy, ix = a_ox[i] * p.stride + dx;
        ok = ok && iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
        if (p.up) { iy >>= 1; ix >>= 1; }
        pix = ((size_t)a_b[i] * p.Hin + iy) * p.Win + ix;
      }
      acur[i] = ok ? base + (pix * cs + (size_t)cseg * BK) * sizeof(T) + sc * 16 : zero;
      astep[i] = ok ? 128 : 0;
    }
  };
  {
    const int tap = kt_begin / tiles_tap, rem = kt_begin - tap * tiles_tap;
    const int src = rem >= tiles_c0 ? 1 : 0;
    set_segment(tap, src, src ? rem - tiles_c0 : rem);
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) wptr[i] += (size_t)kt_begin * wstep[i];

  // issue the DMA of the cursor's tile into stage `buf`, then advance the cursor
  auto prefetch = [&](int buf) {
    char* abuf = smem + buf * S::STAGE_BYTES;
    char* wbuf = abuf + S::A_BYTES;
#pragma unroll
    for (int i = 0; i < AP; ++i) glds16(acur[i], abuf + (wid * AP + i) * 1024);
#pragma unroll
    for (int i = 0; i < WP; ++i) glds16(wptr[i], wbuf + (wid * WP + i) * 1024);
#pragma unroll
    for (int i = 0; i < AP; ++i) acur[i] += astep[i];
#pragma unroll
    for (int i = 0; i < WP; ++i) wptr[i] += wstep[i];
    if (--pf_left == 0) {
      int tap = pf_tap, src = pf_src;
      if (src == 0 && tiles_c1 > 0) src = 1;
      else { src = 0; ++tap; }
      if (tap < p.taps) set_segment(tap, src, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a tile: row*128 + ((chunk ^ (row&7)) * 16)
  const int frow = lane & 15, fq = lane >> 4;
  const int sw = frow & 7;      // (row & 7) for every fragment row of this lane (16 | row offsets)
  const int coff0 = (fq ^ sw) * 16, coff1 = ((4 + fq) ^ sw) * 16;
  const int w_base = (wn * 64 + frow) * 128 + S::A_BYTES, a_base = (wm * 64 + frow) * 128;

  // ---- main loop: NSTAGE-deep ring.  At the top of iteration `it` the wave waits until its own DMA
  // of that tile has landed (leaving the younger tiles' DMAs in flight: counted vmcnt), the barrier
  // makes every wave's part visible and proves that everyone has finished reading the stage of the
  // previous tile, which is then refilled with tile it+NSTAGE-1.  One barrier per K-tile.  Inside an
  // iteration the first half's fragments are read before the prefetch is issued and the second
  // half's before the first half's MFMAs, so LDS latency hides behind address work and MFMAs.
  const int nk = kt_end - kt_begin;
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i)
    if (i < nk) prefetch(i);
  for (int it = 0; it < nk; ++it) {
    // younger tiles already issued and allowed to stay in flight: min(nk-1-it, NSTAGE-2)
    const int younger = min(nk - 1 - it, NSTAGE - 2);
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::GL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S::GL) : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sbuf = smem + (it % NSTAGE) * S::STAGE_BYTES;
    frag_t wf0[4], af0[4], wf1[4], af1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf0[i] = *reinterpret_cast<const frag_t*>(sbuf + w_base + i * 2048 + coff0);
#pragma unroll
    for (int j = 0; j < 4; ++j) af0[j] = *reinterpret_cast<const frag_t*>(sbuf + a_base + j * 2048 + coff0);
    if (it + NSTAGE - 1 < nk) prefetch((it + NSTAGE - 1) % NSTAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) wf1[i] = *reinterpret_cast<const frag_t*>(sbuf + w_base + i * 2048 + coff1);
#pragma unroll
    for (int j = 0; j < 4; ++j) af1[j] = *reinterpret_cast<const frag_t*>(sbuf + a_base + j * 2048 + coff1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mma(wf0[i], af0[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mma(wf1[i], af1[j], acc[i][j]);
  }

  // ---- epilogue -----------------------------------------------------------------------------
  if (p.ln_stats && !p.partial) {
    ln_fold_acc<4, 4>(p, acc, m0 + wm * 64 + frow, n0 + wn * 64 + fq * 4);
    GemmParams pe = p;
    pe.bias = nullptr;                       // b' went in with the fold
    store_wave_tile<T, 4, 4>(pe, acc, m0 + wm * 64 + frow, n0 + wn * 64, fq);
    return;
  }
  store_wave_tile<T, 4, 4>(p, acc, m0 + wm * 64 + frow, n0 + wn * 64, fq);
}

// ---------------------------------------------------------------------------------------------
// Large-tile kernel: 256(m) x 32*NFW(n) per 512-thread workgroup, one workgroup per CU.
//
// A 128x128 tile moves 32 KiB of operands L2 -> LDS per 2.1 MFLOP, which is what bounds the kernel
// above; a 256x320 tile moves 72 KiB per 10.5 MFLOP, less than half of that per flop.  With a single
// workgroup on the CU the overlap of LDS traffic and MFMAs has to be built by hand: the 8 waves form
// two groups of four (one wave of each group per SIMD) that run the same program one segment
// apart.  A K-tile (64 deep) is 2 NPH phases per wave; a phase is a read segment (ds_read of the
// phase's fragments + a share of a later tile's LDS-DMA, retired with lgkmcnt(0)), a barrier, an
// MFMA segment (4 x NFW/NPH MFMAs), a barrier.  Group 1 enters the loop one barrier late, so in
// every segment one group feeds the matrix cores while the other one reads.
//
// Wave (wr = wid & 3, grp = wid >> 2) owns rows wr*64..+63 and columns grp*16*NFW.. (NFW fragments);
// phase (ks, h) multiplies k-step ks of the tile into column part h (NFW/NPH fragments).
//
// Shapes: NFW 10 / NPH 2 (256x320) and NFW 8 / NPH 1 (256x256: its fragments fit one phase per k-step, 32 MFMAs
// per segment, +1..3 %) with two LDS stages; NFW 5 / NPH 1
// (256x160, for layers with too few 256x320 tiles to fill the chip) with three.
//
// Hazards.  The stage of K-tile t is read in t's segments; the last reads (group 1, last phase)
// retire before the barrier that opens group 0's first segment of tile t+1, and the DMA of tile
// t+NSTAGE into that stage is issued no earlier than that segment.  Every wave waits for its own
// DMA of tile t+1 (counted vmcnt: with three stages tile t+2 stays in flight) before the last
// barrier that precedes group 0's first read of it.
template <int NFW_, int NPH_, int NSTAGE_, int MJ_ = 4> struct PPShape {
  static constexpr int NFW = NFW_, NPH = NPH_, NSTAGE = NSTAGE_, MJ = MJ_;   // MJ: 16-row fragments per wave
  static constexpr int NFP = NFW / NPH;                             // fragments per phase
  // one or two row fragments per wave: a K-tile is so little matrix work (2 NFW MJ MFMAs per wave) that the two-group
  // schedule's eight synchronisations cost more than they hide; every wave runs read -> multiply -> one barrier
  static constexpr bool LOCKSTEP = MJ_ <= 2;
  static_assert(!LOCKSTEP || NSTAGE_ <= 5, "the lock-step loop keeps at most three tiles in flight behind the next one");
  static constexpr int BM = 64 * MJ, BN = 32 * NFW, NTHREADS = 512;
  static constexpr int AP = MJ;                                     // 1-KiB activation pieces per wave and tile
  static constexpr int WPT = BN / 8, WP_LO = WPT / 8, WP_REM = WPT % 8;   // weight pieces: waves < WP_REM take one more
  static constexpr int WP_MAX = WP_LO + (WP_REM ? 1 : 0);
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
  static_assert(NFW % NPH == 0 && (NSTAGE == 2 || NSTAGE == 3 || LOCKSTEP), "unsupported shape");
  static_assert(LDS_BYTES <= 160 * 1024, "stages do not fit in LDS");
};
typedef PPShape<10, 2, 2> PP320;
typedef PPShape<8, 1, 2> PP256;
typedef PPShape<5, 1, 3> PP160;
typedef PPShape<5, 1, 3, 2> PP160H;     // 128x160: layers whose 256-row tiles would leave CUs idle
typedef PPShape<4, 1, 3> PP128;         // 256x128: 128-channel layers (the VAE decoder's 512x512 stage)
typedef PPShape<5, 1, 3, 1> PP160Q;     // 64x160: small problems (UNet batch 2) whose 128-row tiles leave most CUs idle

// Softmax (base 2: the scores arrive scaled by log2 e) over each 80-column segment of the wave's rows, in the
// accumulators: acc[i][j] holds columns 16 i + 4 fq .. +3 of row (lane & 15) + 16 j, so a row's segment is five
// fragments in the four lanes that differ in fq.  Cross-attention scores against a text context of up to 80
// tokens, one segment per head; columns beyond the context carry a large negative bias and come out as zeros.
template <int NFW, int MJ>
__device__ __forceinline__ void softmax_seg80(f32x4 (&acc)[NFW][MJ]) {
  static_assert(NFW % 5 == 0, "a wave must own whole 80-column segments");
#pragma unroll
  for (int sg = 0; sg < NFW / 5; ++sg)
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const f32x4 v = acc[5 * sg + i][j];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x4 v = acc[5 * sg + i][j];
        v.x = __builtin_amdgcn_exp2f(v.x - mx); v.y = __builtin_amdgcn_exp2f(v.y - mx);
        v.z = __builtin_amdgcn_exp2f(v.z - mx); v.w = __builtin_amdgcn_exp2f(v.w - mx);
        sum += (v.x + v.y) + (v.z + v.w);
        acc[5 * sg + i][j] = v;
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[5 * sg + i][j] *= inv;
    }
}

// MODE selects what a launch needs beyond the plain GEMM, as instantiations of their own because a few extra
// branches or address terms in the loader cost every other launch 0.6-4 % (measured): 1 (EX) = the K loop continues
// behind the taps into extra pointwise sources (p.e0 / p.e1); 2 (PS) = one weight matrix (and bias / column-sum
// vector) per sample, optional softmax over 80-column segments in the epilogue.
enum { GEMM_PLAIN = 0, GEMM_EX = 1, GEMM_PS = 2 };
template <typename T, typename S, bool TI, int MODE = GEMM_PLAIN>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(GemmParams p) {
  constexpr bool EX = MODE == GEMM_EX, PS = MODE == GEMM_PS;
  static_assert(sizeof(T) == 2, "16-bit operands only");
  constexpr int BK = 64, AP = S::AP, NFW = S::NFW, NFP = S::NFP, NPH = S::NPH, NSTAGE = S::NSTAGE, MJ = S::MJ;
  constexpr int WROWS = 16 * MJ;                                     // rows of a wave's tile
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = wid & 3;

  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_n = id % p.tiles_n, tile_m = id / p.tiles_n;
  const int m0 = tile_m * S::BM, n0 = tile_n * S::BN;
  const int kt_begin = blockIdx.y * p.kt_per_split;
  const int nk = min(p.nkt, kt_begin + p.kt_per_split) - kt_begin;
  // per-sample weights (and bias / column sums): this tile's sample
  const int smp = PS && p.w_rows > 0 ? __builtin_amdgcn_readfirstlane(m0 / p.w_rows) : 0;
  const char* const wmat = PS ? p.w + (size_t)smp * p.w_stride : p.w;
  const float* const biasv = PS && p.bias ? p.bias + (size_t)smp * p.vec_stride : p.bias;
  const float* const colsumv = PS && p.ln_colsum ? p.ln_colsum + (size_t)smp * p.vec_stride : p.ln_colsum;

  // ---- loader state ----------------------------------------------------------------------------
  const int lrow = lane >> 3;
  const int sc = (lane & 7) ^ lrow;
  const int hw = p.Ho * p.Wo;
  const bool pointwise = (p.taps == 1 && p.stride == 1 && !p.up);
  // Two K orders.  3x3 convolutions without upsampling walk the nine taps of a 64-channel slice in
  // consecutive K-tiles (kt = slice * 9 + tap): the shifted views of the same activation rows are then
  // fetched back to back and meet in L2 (the 32 workgroups of an XCD touch about 1 MB per K-tile; a
  // tap-major walk puts 5 MB between two views and every tap goes out to the fabric again).  Pointwise
  // problems and upsampling convolutions keep the tap-major walk (kt = tap * slices + slice).
  constexpr bool tapinner = TI;       // the launcher picks TI only for taps == 9 without upsampling
  // per row, tap-major walk: sample index (-1 = row beyond M) and the output pixel packed as
  // (oy << 16 | ox), for pointwise problems the row index itself; tap-inner walk: bit mask of the taps
  // that fall inside the image (0 = row beyond M) and the linear index of the centre input pixel, the
  // one tap (pad, pad) reads
  int a_b[AP], a_px[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + (wid * AP + i) * 8 + lrow;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    if (pointwise) {
      a_b[i] = ok ? 0 : -1; a_px[i] = mm;
    } else {
      const int b = mm / hw, rem = mm - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      if constexpr (tapinner) {
        const int cy = oy * p.stride, cx = ox * p.stride;
        unsigned my = 0, mx = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (cy + d - p.pad >= 0 && cy + d - p.pad < p.Hin) my |= 1u << d;
          if (cx + d - p.pad >= 0 && cx + d - p.pad < p.Win) mx |= 1u << d;
        }
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          if (((my >> (t / 3)) & 1u) && ((mx >> (t % 3)) & 1u)) mask |= 1u << t;
        a_b[i] = ok ? (int)mask : 0;
        a_px[i] = (b * p.Hin + cy) * p.Win + cx;
      } else {
        a_b[i] = ok ? b : -1; a_px[i] = (oy << 16) | ox;
      }
    }
  }
  // weight rows: the launcher guarantees N % BN == 0, so every row exists.  One 32-bit byte offset per
  // lane; the pieces of a tile differ by a wave-uniform stride (scalar base + vector offset loads).
  const int wp_first = wid * S::WP_LO + min(wid, S::WP_REM);           // first weight piece of this wave
  const bool wp_extra = S::WP_REM != 0 && wid < S::WP_REM;             // ... and whether it has WP_LO + 1
  unsigned wofs = (unsigned)(((size_t)(n0 + wp_first * 8 + lrow) * p.K + (size_t)kt_begin * BK) * sizeof(T)) + sc * 16;
  const size_t wpitch8 = (size_t)8 * p.K * sizeof(T);
  const int tiles_c0 = p.c0 / BK, tiles_c1 = p.c1 / BK, tiles_tap = tiles_c0 + tiles_c1;
  const int Hup = p.up ? 2 * p.Hin : p.Hin, Wup = p.up ? 2 * p.Win : p.Win;
  const char* const zero = p.zero + sc * 16;

  const char* acur[AP];         // rows outside the image / beyond M sit on the zero page and do not advance
  int pf_tap, pf_src, pf_left;
  const int tiles_e0 = EX ? p.ec0 / BK : 0, tiles_e1 = EX ? p.ec1 / BK : 0;      // K-tiles of the extra pointwise sources
  auto set_segment = [&](int tap, int src, int cseg) {
    pf_tap = tap; pf_src = src;
    if (EX && tap >= p.taps) {
      // behind the taps: extra source `src`, read at the output pixel itself (no halo, no stride)
      pf_left = (src ? tiles_e1 : tiles_e0) - cseg;
      const char* base = src ? p.e1 : p.e0;
      const int cs = src ? p.ec1 : p.ec0;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const size_t pix = pointwise ? (size_t)a_px[i]
                                     : ((size_t)a_b[i] * p.Ho + (a_px[i] >> 16)) * p.Wo + (a_px[i] & 0xffff);
        acur[i] = a_b[i] >= 0 ? base + (pix * cs + (size_t)cseg * BK) * sizeof(T) + sc * 16 : zero;
      }
      return;
    }
    pf_left = (src ? tiles_c1 : tiles_c0) - cseg;
    const char* base = src ? p.a1 : p.a0;
    const int cs = src ? p.c1 : p.c0;
    const int dy = p.taps == 9 ? tap / 3 - p.pad : 0, dx = p.taps == 9 ? tap - (tap / 3) * 3 - p.pad : 0;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      bool ok = a_b[i] >= 0;
      size_t pix;
      if (pointwise) {
        pix = (size_t)a_px[i];
      } else {
        int iy = (a_px[i] >> 16) * p.stride + dy, ix = (a_px[i] & 0xffff) * p.stride + dx;
        ok = ok && iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
        if (p.up) { iy >>= 1; ix >>= 1; }
        pix = ((size_t)a_b[i] * p.Hin + iy) * p.Win + ix;
      }
      acur[i] = ok ? base + (pix * cs + (size_t)cseg * BK) * sizeof(T) + sc * 16 : zero;
    }
  };
  // tap-inner walk: wave-uniform position (slice, tap = dy * 3 + dx), byte offset of the tap relative
  // to the centre pixel, and the step of the weight offset to the next K-tile
  int ti_chunk = 0, ti_tap = 0, ti_dy = 0, ti_dx = 0, ti_cs2 = 0, ti_rowpitch = 0;
  long ti_soff = 0;
  unsigned w_step = 128;
  const int ctot2 = (p.c0 + p.c1) * (int)sizeof(T);
  auto set_source = [&](int chunk) {      // acur = centre pixel of every row in slice `chunk`
    const int src = chunk >= tiles_c0 ? 1 : 0;
    const char* base = src ? p.a1 : p.a0;
    const int cs = src ? p.c1 : p.c0;
    const int cseg = src ? chunk - tiles_c0 : chunk;
    ti_cs2 = cs * (int)sizeof(T);
    ti_rowpitch = p.Win * ti_cs2;
    ti_soff = (long)(ti_dy - p.pad) * ti_rowpitch + (long)(ti_dx - p.pad) * ti_cs2;
#pragma unroll
    for (int i = 0; i < AP; ++i)
      acur[i] = base + ((size_t)a_px[i] * cs + (size_t)cseg * BK) * sizeof(T) + sc * 16;
  };
  // tap-inner walk behind its last slice: the extra sources, one K-tile after the other (ti_ex = tiles left in the
  // current one, 0 while the taps are walked); the rows sit on their own pixel (a_px, stride 1) and the weight
  // columns follow those of the taps, so the weight offset simply advances by one tile
  int ti_ex = 0, ti_ex_src = 0;
  auto set_extra = [&](int src) {
    ti_ex_src = src;
    ti_ex = src ? tiles_e1 : tiles_e0;
    const char* base = src ? p.e1 : p.e0;
    const int cs = src ? p.ec1 : p.ec0;
    ti_soff = 0;
#pragma unroll
    for (int i = 0; i < AP; ++i) acur[i] = base + ((size_t)a_px[i] * cs) * sizeof(T) + sc * 16;
  };
  if constexpr (tapinner) {
    // (the launcher sends launches with extra sources here only unsplit: kt_begin = 0)
    ti_chunk = kt_begin / 9;
    ti_tap = kt_begin - ti_chunk * 9;
    ti_dy = ti_tap / 3;
    ti_dx = ti_tap - ti_dy * 3;
    set_source(ti_chunk);
    wofs = (unsigned)(((size_t)(n0 + wp_first * 8 + lrow) * p.K) * sizeof(T)) + sc * 16 +
           (unsigned)(ti_tap * ctot2 + ti_chunk * (BK * (int)sizeof(T)));
  } else {
    int tap = kt_begin / tiles_tap, rem = kt_begin - tap * tiles_tap;
    if (EX && tap >= p.taps) {             // a split that starts among the extra sources
      rem = kt_begin - p.taps * tiles_tap;
      const int src = rem >= tiles_e0 ? 1 : 0;
      set_segment(p.taps, src, src ? rem - tiles_e0 : rem);
    } else {
      const int src = rem >= tiles_c0 ? 1 : 0;
      set_segment(tap, src, src ? rem - tiles_c0 : rem);
    }
  }
  auto prefetch_a = [&](int stage) {
    char* abuf = smem + stage * S::STAGE_BYTES;
    if constexpr (tapinner) {
      if (EX && ti_ex > 0) {
        // an extra source: every row inside M reads its own pixel (the centre tap's validity bit)
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          glds16(((unsigned)a_b[i] & 16u) ? acur[i] : zero, abuf + (wid * AP + i) * 1024);
          acur[i] += BK * sizeof(T);
        }
        w_step = (unsigned)(BK * (int)sizeof(T));
        if (--ti_ex == 0 && ti_ex_src == 0 && tiles_e1 > 0) set_extra(1);
        return;
      }
      const unsigned bit = 1u << ti_tap;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const char* src = acur[i] + ti_soff;
        glds16(((unsigned)a_b[i] & bit) ? src : zero, abuf + (wid * AP + i) * 1024);
      }
      if (ti_tap == 8) {
        w_step = (unsigned)(BK * (int)sizeof(T) - 8 * ctot2);
        ti_tap = 0; ti_dy = 0; ti_dx = 0;
        ++ti_chunk;
        if (EX && ti_chunk == tiles_tap) {
          // the taps are done: on to the extra sources, whose weight columns start one tile further
          w_step = (unsigned)(BK * (int)sizeof(T));
          if (tiles_e0 > 0) set_extra(0);
        } else if (ti_chunk == tiles_c0 && tiles_c1 > 0) {
          set_source(ti_chunk);
        } else {
          ti_soff = -(long)p.pad * (ti_rowpitch + ti_cs2);
#pragma unroll
          for (int i = 0; i < AP; ++i) acur[i] += BK * sizeof(T);
        }
      } else {
        w_step = (unsigned)ctot2;
        ++ti_tap;
        if (ti_dx == 2) { ti_dx = 0; ++ti_dy; ti_soff += ti_rowpitch - 2 * ti_cs2; }
        else { ++ti_dx; ti_soff += ti_cs2; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        glds16(acur[i], abuf + (wid * AP + i) * 1024);
        acur[i] += acur[i] != zero ? 128 : 0;
      }
      if (--pf_left == 0) {
        int tap = pf_tap, src = pf_src;
        if (src == 0 && ((EX && tap >= p.taps) ? tiles_e1 : tiles_c1) > 0) src = 1;
        else { src = 0; ++tap; }
        if (tap < p.taps || (EX && tap == p.taps && tiles_e0 > 0)) set_segment(tap, src, 0);
      }
    }
  };
  auto prefetch_w = [&](int stage) {      // follows prefetch_a of the same tile (w_step)
    char* wbuf = smem + stage * S::STAGE_BYTES + S::A_BYTES + wp_first * 1024;
#pragma unroll
    for (int i = 0; i < S::WP_LO; ++i) glds16(wmat + i * wpitch8 + wofs, wbuf + i * 1024);
    if (wp_extra) glds16(wmat + S::WP_LO * wpitch8 + wofs, wbuf + S::WP_LO * 1024);
    if constexpr (tapinner) wofs += w_step;
    else wofs += 128;
  };
  // wait until at most one tile's worth of this wave's DMA is still in flight (three-stage ring)
  auto wait_keep_one_tile = [&]() {
    if (wp_extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + S::WP_LO + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + S::WP_LO) : "memory");
  };

  // ... at most n tiles' worth (n <= 3, deeper rings of the lock-step loop)
  auto wait_keep_tiles = [&](int n) {
    const int e = wp_extra ? 1 : 0;
    constexpr int PER = AP + S::WP_LO;
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) { if (e) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); }
    else if (n == 2) { if (e) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER + 2) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); }
    else { if (e) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER + 3) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); }
  };

  const int frow = lane & 15, fq = lane >> 4;
  // The accumulators start from the per-column bias and the per-(sample, column) time-embedding term
  // (when this launch owns the epilogue): their loads overlap the first tile's DMA and the epilogue
  // is left with the residual only.
  f32x4 acc[NFW][MJ];
  const bool own_epilogue = p.partial == nullptr;
  // LayerNorm folded in (and this launch owns the epilogue): y = rstd (S - mean c) + b'.  The accumulators start
  // from b'_n / rstd_m - mean_m c_n, so that all that is left for the epilogue is the row scaling by rstd_m.
  const bool ln_mode = own_epilogue && p.ln_stats != nullptr;
  float ln_r[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) ln_r[j] = 1.f;
  {
    const int ncol = n0 + grp * (S::BN / 2) + fq * 4;
    if (ln_mode) {
      // every lane fetches the statistics of one of the wave's rows; the rows a lane multiplies come by shuffle
      const float2 mine = ln_row(p, min(m0 + wr * WROWS + (lane % WROWS), p.M - 1));
      float inv[MJ], mean[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const float rs = __shfl(mine.x, j * 16 + frow, 64), rm = __shfl(mine.y, j * 16 + frow, 64);
        ln_r[j] = rs;
        inv[j] = 1.0f / rs;
        mean[j] = rm * inv[j];
      }
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(colsumv + ncol + i * 16);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (biasv) b = *reinterpret_cast<const f32x4*>(biasv + ncol + i * 16);
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = b * inv[j] - mean[j] * c;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NFW; ++i) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (own_epilogue && biasv) bv = *reinterpret_cast<const f32x4*>(biasv + ncol + i * 16);
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = bv;
      }
      if (own_epilogue && p.rowadd) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = min(m0 + wr * WROWS + j * 16 + frow, p.M - 1);
          const float* ra = p.rowadd + (size_t)(m / hw) * p.ldra + ncol;
#pragma unroll
          for (int i = 0; i < NFW; ++i) acc[i][j] += *reinterpret_cast<const f32x4*>(ra + i * 16);
          __builtin_amdgcn_sched_barrier(0);        // one row group's loads in flight at a time (registers)
        }
      }
    }
  }
  const int sw = frow & 7;
  const int a_base = (wr * WROWS + frow) * 128;
  const int w_base = S::A_BYTES + (grp * (S::BN / 2) + frow) * 128;

  frag_t af[MJ], wf[NFP];

  // ---- prologue: the first NSTAGE-1 tiles are put in flight, tile 0 is made visible to everyone,
  // then the groups are staggered ----------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nk) { prefetch_a(t); prefetch_w(t); }
  if constexpr (S::LOCKSTEP) wait_keep_tiles(min(NSTAGE - 2, nk - 1));
  else if (NSTAGE == 3 && nk > 1) wait_keep_one_tile();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!S::LOCKSTEP && grp == 1) __builtin_amdgcn_s_barrier();

  int st_rd = 0, st_wr = NSTAGE - 1;          // stage of the tile being multiplied / of the tile being fetched
  if constexpr (S::LOCKSTEP) {
    // All waves in step, one barrier per K-tile: read the tile's fragments, put tile kt + NSTAGE - 1 in flight (its
    // stage held tile kt - 1, whose reads every wave retired before the barrier that ended that iteration),
    // multiply, wait for this wave's share of tile kt + 1, barrier.
    static_assert(NPH == 1, "lock-step loop reads whole k-steps");
    for (int kt = 0; kt < nk; ++kt) {
      const char* sb = smem + st_rd * S::STAGE_BYTES;
      const bool more = kt + NSTAGE - 1 < nk;
      frag_t af2[2][MJ], wf2[2][NFW];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((4 * ks + fq) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < MJ; ++j) af2[ks][j] = *reinterpret_cast<const frag_t*>(sb + a_base + j * 2048 + coff);
#pragma unroll
        for (int i = 0; i < NFW; ++i) wf2[ks][i] = *reinterpret_cast<const frag_t*>(sb + w_base + i * 2048 + coff);
      }
      if (more) { prefetch_a(st_wr); prefetch_w(st_wr); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = mma(wf2[ks][i], af2[ks][j], acc[i][j]);
      // tile kt + 1 must have landed; the tiles issued after it (NSTAGE - 2 of them in the steady state) stay in flight
      wait_keep_tiles(min(nk - 1, kt + NSTAGE - 1) - (kt + 1));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      st_rd = st_rd + 1 == NSTAGE ? 0 : st_rd + 1;
      st_wr = st_wr + 1 == NSTAGE ? 0 : st_wr + 1;
    }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + st_rd * S::STAGE_BYTES;
    const bool more = kt + NSTAGE - 1 < nk;           // a tile to put in flight during this one
    const bool keep = NSTAGE == 3 && kt + 2 < nk;     // ... that may stay in flight past the end of this one
#pragma unroll
    for (int ph = 0; ph < 2 * NPH; ++ph) {
      const int ks = ph / NPH, h = ph % NPH;
      const int coff = ((4 * ks + fq) ^ sw) * 16;
      // read segment
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) af[j] = *reinterpret_cast<const frag_t*>(sb + a_base + j * 2048 + coff);
      }
#pragma unroll
      for (int i = 0; i < NFP; ++i)
        wf[i] = *reinterpret_cast<const frag_t*>(sb + w_base + (h * NFP + i) * 2048 + coff);
      if (ph == 0 && more) prefetch_a(st_wr);
      // (two stages and two phases: both operands go out in the first phase, a full phase ahead of the wait)
      if (ph == ((NSTAGE == 2 && 2 * NPH == 2) ? 0 : 1) && more) prefetch_w(st_wr);
      if (ph == 2 * NPH - 1 && grp == 1) {      // the next tile's DMA is retired before the barrier that
        if (keep) wait_keep_one_tile();         // precedes group 0's first read of it
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // MFMA segment
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NFP; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[h * NFP + i][j] = mma(wf[i], af[j], acc[h * NFP + i][j]);
      __builtin_amdgcn_s_setprio(0);
      if (ph == 2 * NPH - 1 && grp == 0) {
        if (keep) wait_keep_one_tile();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    st_rd = st_rd + 1 == NSTAGE ? 0 : st_rd + 1;
    st_wr = st_wr + 1 == NSTAGE ? 0 : st_wr + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  }

  GemmParams pe = p;
  if (own_epilogue) { pe.bias = nullptr; pe.rowadd = nullptr; }      // already in the accumulators
  const int m_wave = m0 + wr * WROWS, n_wave = n0 + grp * (S::BN / 2);
  if (ln_mode) {
    // rstd is fetched again rather than kept in registers across the K loop
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] *= ln_r[j];
  }
  if constexpr (PS && NFW == 5) {     // (the 160-column shapes only: the launcher never sends score GEMMs elsewhere)
    if (p.softmax_seg) softmax_seg80<NFW, MJ>(acc);
  }
  if (p.wide_store) {
    // all stage reads and DMAs are retired behind the last barrier: the stages are free
    float* strip = reinterpret_cast<float*>(smem + wid * (16 * (16 * NFW + 4) * 4));
    static_assert(8 * 16 * (16 * NFW + 4) * 4 <= S::LDS_BYTES, "strips do not fit");
    if (p.partial) {
      pe.resid = nullptr; pe.dup_rows = 0;                           // applied by the reduction pass
      store_wave_tile_lds<T, float, NFW, false, MJ>(pe, acc, strip, p.partial + (size_t)blockIdx.y * p.M * p.N, p.N,
                                                    m_wave, n_wave, lane);
    } else if (p.out_f32) {
      store_wave_tile_lds<T, float, NFW, false, MJ>(pe, acc, strip, reinterpret_cast<float*>(p.out), p.ldo, m_wave,
                                                    n_wave, lane);
    } else if (NFW % 2 == 0 && p.geglu) {
      if constexpr (NFW % 2 == 0)
        store_wave_tile_lds<T, T, NFW, true, MJ>(pe, acc, strip, reinterpret_cast<T*>(p.out), p.ldo, m_wave, n_wave,
                                                 lane);
    } else {
      // row sums for a LayerNorm behind this GEMM: one slot per (tile along N, wave group)
      constexpr int STRIPS = 8 * 16 * (16 * NFW + 4) * 4, PSUM = 16 * 2 * NFW * 8;
      static_assert(STRIPS + 8 * PSUM <= S::LDS_BYTES, "row-sum scratch does not fit");
      float2* psum = p.rs_out ? reinterpret_cast<float2*>(smem + STRIPS + wid * PSUM) : nullptr;
      store_wave_tile_lds<T, T, NFW, false, MJ>(pe, acc, strip, reinterpret_cast<T*>(p.out), p.ldo, m_wave, n_wave,
                                                lane, psum, 2 * tile_n + grp);
    }
    return;
  }
  store_wave_tile<T, NFW, MJ>(pe, acc, m_wave + frow, n_wave, fq);
}

// split-K second pass: sum the fp32 slabs and run the same epilogue.  One thread per 4 columns.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int splits) {
  const int n4 = p.N / 4;
  const size_t total = (size_t)p.M * n4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / n4), n = (int)(idx - (size_t)m * n4) * 4;
    if (p.geglu) {
      // columns are laid out [16 value | 16 gate] per 32; handle the value quad and fetch its gate
      if ((n & 16) != 0) continue;
      f32x4 val = {0, 0, 0, 0}, gate = {0, 0, 0, 0};
      {
        const float* s0 = p.partial + (size_t)m * p.N + n;
        const size_t zs = (size_t)p.M * p.N;
        int z = 0;
        for (; z + 1 < splits; z += 2) {        // two slabs (four loads) in flight; the sums keep slab order
          const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + z * zs), ag = *reinterpret_cast<const f32x4*>(s0 + z * zs + 16);
          const f32x4 b = *reinterpret_cast<const f32x4*>(s0 + (z + 1) * zs), bg = *reinterpret_cast<const f32x4*>(s0 + (z + 1) * zs + 16);
          val += a; gate += ag; val += b; gate += bg;
        }
        for (; z < splits; ++z) {
          val += *reinterpret_cast<const f32x4*>(s0 + z * zs);
          gate += *reinterpret_cast<const f32x4*>(s0 + z * zs + 16);
        }
      }
      if (p.ln_stats) {
        const float2 st = ln_row(p, m);
        val = st.x * val + (*reinterpret_cast<const f32x4*>(p.bias + n) - st.y * *reinterpret_cast<const f32x4*>(p.ln_colsum + n));
        gate = st.x * gate + (*reinterpret_cast<const f32x4*>(p.bias + n + 16) -
                              st.y * *reinterpret_cast<const f32x4*>(p.ln_colsum + n + 16));
        GemmParams pe = p;
        pe.bias = nullptr;
        epilogue_store_geglu<T>(pe, m, n, n + 16, (n >> 5) * 16 + (n & 15), val, gate);
        continue;
      }
      epilogue_store_geglu<T>(p, m, n, n + 16, (n >> 5) * 16 + (n & 15), val, gate);
    } else {
      f32x4 v = {0, 0, 0, 0};
      {
        const float* s0 = p.partial + (size_t)m * p.N + n;
        const size_t zs = (size_t)p.M * p.N;
        int z = 0;
        for (; z + 3 < splits; z += 4) {        // four slabs in flight; the sum keeps slab order
          const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + z * zs), b = *reinterpret_cast<const f32x4*>(s0 + (z + 1) * zs);
          const f32x4 d = *reinterpret_cast<const f32x4*>(s0 + (z + 2) * zs), e = *reinterpret_cast<const f32x4*>(s0 + (z + 3) * zs);
          v += a; v += b; v += d; v += e;
        }
        for (; z < splits; ++z) v += *reinterpret_cast<const f32x4*>(s0 + z * zs);
      }
      if (p.ln_stats) {
        const float2 st = ln_row(p, m);
        f32x4 b = {0, 0, 0, 0};
        if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v = st.x * v + (b - st.y * *reinterpret_cast<const f32x4*>(p.ln_colsum + n));
        GemmParams pe = p;
        pe.bias = nullptr;
        epilogue_store<T>(pe, m, n, v);
        continue;
      }
      epilogue_store<T>(p, m, n, v);
    }
  }
}

// split-K second pass that also leaves the row sums of its output (one pair per row).  A thread owns ONE quad of
// columns of one row (all its slab loads are independent and in flight together); a row is TPR threads, a whole
// number of waves, and its sums are folded lanes first, then the row's waves through LDS in wave order.
template <typename T>
__global__ __launch_bounds__(512) void splitk_reduce_rows_kernel(GemmParams p, int splits, int tpr) {
  __shared__ float2 wsum[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpb = (int)blockDim.x / tpr, wpr = tpr >> 6;         // rows per block, waves per row
  const int n4 = p.N / 4;
  const int lr = (int)threadIdx.x / tpr, c = (int)threadIdx.x - lr * tpr;
  for (int m0 = blockIdx.x * rpb; m0 < p.M; m0 += gridDim.x * rpb) {
    const int m = m0 + lr;
    const bool live = m < p.M && c < n4;
    float ps = 0.f, pq = 0.f;
    if (live) {
      const int n = c * 4;
      const float* src = p.partial + (size_t)m * p.N + n;
      const size_t zs = (size_t)p.M * p.N;
      f32x4 v = {0, 0, 0, 0};
      int z = 0;
      for (; z + 3 < splits; z += 4) {           // four slabs in flight; the sum keeps slab order
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + z * zs), b = *reinterpret_cast<const f32x4*>(src + (z + 1) * zs);
        const f32x4 d = *reinterpret_cast<const f32x4*>(src + (z + 2) * zs), e = *reinterpret_cast<const f32x4*>(src + (z + 3) * zs);
        v += a; v += b; v += d; v += e;
      }
      for (; z < splits; ++z) v += *reinterpret_cast<const f32x4*>(src + z * zs);
      GemmParams pe = p;
      if (p.ln_stats) {
        const float2 st = ln_row(p, m);
        f32x4 b = {0, 0, 0, 0};
        if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v = st.x * v + (b - st.y * *reinterpret_cast<const f32x4*>(p.ln_colsum + n));
        pe.bias = nullptr;
      }
      v = epilogue_value<T>(pe, m, n, v);
      pe.bias = nullptr; pe.rowadd = nullptr; pe.resid = nullptr;
      epilogue_store<T>(pe, m, n, v);
      if (!p.out_f32) { v.x = to_f32(from_f32<T>(v.x)); v.y = to_f32(from_f32<T>(v.y)); v.z = to_f32(from_f32<T>(v.z)); v.w = to_f32(from_f32<T>(v.w)); }
      ps = (v.x + v.y) + (v.z + v.w);
      pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    ps = wave_sum(ps); pq = wave_sum(pq);
    if (wpr > 1) {
      if (lane == 0) wsum[wave] = make_float2(ps, pq);
      __syncthreads();
      ps = 0.f; pq = 0.f;
      for (int w = 0; w < wpr; ++w) { const float2 t = wsum[lr * wpr + w]; ps += t.x; pq += t.y; }
      __syncthreads();
    }
    if (m < p.M && c < p.rs_parts) {
      const float2 o = c == 0 ? make_float2(ps, pq) : make_float2(0.f, 0.f);
      p.rs_out[(size_t)m * p.rs_parts + c] = o;
      if (p.dup_rows > 0) p.rs_out[(size_t)(m + p.dup_rows) * p.rs_parts + c] = o;
    }
  }
}

// row sums of a stored 16-bit / fp32 output, for the launches whose kernel does not leave them itself
template <typename T>
__global__ __launch_bounds__(256) void row_sums_kernel(GemmParams p) {
  const int lane = threadIdx.x & 63;
  const int n4 = p.N / 4;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < p.M; m += gridDim.x * 4) {
    float ps = 0.f, pq = 0.f;
    for (int c = lane; c < n4; c += 64) {
      const Tx4 r = *reinterpret_cast<const Tx4*>(reinterpret_cast<const T*>(p.out) + (size_t)m * p.ldo + c * 4);
      const float x = to_f32(r[0]), y = to_f32(r[1]), z = to_f32(r[2]), w = to_f32(r[3]);
      ps += (x + y) + (z + w);
      pq += (x * x + y * y) + (z * z + w * w);
    }
    ps = wave_sum(ps); pq = wave_sum(pq);
    if (lane < p.rs_parts) {
      const float2 o = lane == 0 ? make_float2(ps, pq) : make_float2(0.f, 0.f);
      p.rs_out[(size_t)m * p.rs_parts + lane] = o;
      if (p.dup_rows > 0) p.rs_out[(size_t)(m + p.dup_rows) * p.rs_parts + lane] = o;
    }
  }
}

}  // namespace

// Raise the dynamic-LDS limit of every instantiation once (must not first happen under stream capture).
namespace {
template <typename T>
int set_lds_limits() {
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_kernel<T, ShapeSmall>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, ShapeSmall::LDS_BYTES));
  if constexpr (sizeof(T) == 2) {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP320, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP320::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP320, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP320::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP320, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP320::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP320, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP320::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP320, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP320::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP256, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP256::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP256, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP256::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP256, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP256::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP256, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP256::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP256, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP256::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160H, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160H::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160H, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160H::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160H, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160H::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160H, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160H::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160H, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160H::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160Q, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160Q::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160Q, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160Q::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160Q, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160Q::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160Q, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160Q::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP160Q, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP160Q::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP128, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP128::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP128, false, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP128::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP128, false, GEMM_PS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP128::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP128, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP128::LDS_BYTES));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_pp_kernel<T, PP128, true, GEMM_EX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PP128::LDS_BYTES));
  }
  return SDMI_OK;
}
}  // namespace

static char* g_zero_page = nullptr;     // device buffer of zeros, allocated once per process

int conv_gemm_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  if (!g_zero_page) {
    SDMI_CHECK_HIP(hipMalloc((void**)&g_zero_page, 4096));
    SDMI_CHECK_HIP(hipMemset(g_zero_page, 0, 4096));
  }
  SDMI_TRY(set_lds_limits<float>());
  SDMI_TRY(set_lds_limits<half_t>());
  SDMI_TRY(set_lds_limits<bf16_t>());
  done = true;
  return SDMI_OK;
}

namespace {

// ---- kernel / split selection --------------------------------------------------------------------
// A plan is (block shape, number of K splits).  Each candidate is priced with a two-constant model:
// the workgroups are dealt to the CUs in rounds (512 small blocks or 256 large ones per round), a
// round takes (K-tiles per split) x (time of one K-tile of that shape), and split-K adds the slab
// round trip through HBM plus a launch.  The K-tile times are measured on MI355X with full grids.
enum { KIND_SMALL = 0, KIND_PP320 = 1, KIND_PP256 = 2, KIND_PP160 = 3, KIND_PP160H = 4, KIND_PP128 = 5, KIND_PP160Q = 6 };
struct Plan { int kind, splits; };

// SDMI_GEMM_SHAPE (tests, experiments; read on every call): 1 = 128x128 tiles only, 2 = 256x320 /
// 256x256 tiles wherever they are legal, 3 = 256x160 tiles wherever they are legal, 4 = 128x160 likewise,
// 5 = never 128x160 / 64x160, 6 = 256x128 wherever it is legal, 7 = 64x160 wherever it is legal.
int forced_kind() {
  const char* e = getenv("SDMI_GEMM_SHAPE");
  return e ? atoi(e) : 0;
}
// SDMI_GEMM_DIRECT_STORE=1 (tests; read on every call): large-tile kernels store straight from the
// accumulator layout, the path otherwise taken only for outputs that are not 16-byte addressable
bool direct_store_forced() {
  const char* e = getenv("SDMI_GEMM_DIRECT_STORE");
  return e && atoi(e) != 0;
}

constexpr double T_KT_SMALL_US = 1.40;    // 128x128x64 K-tile, two blocks per CU
constexpr double T_KT_PP320_US = 2.10;    // 256x320x64 K-tile, one block per CU
constexpr double T_KT_PP256_US = 1.75;    // 256x256x64
constexpr double T_KT_PP160_US = 1.20;    // 256x160x64
constexpr double T_KT_PP160H_US = 0.80;   // 128x160x64 (only offered for short K loops, see make_plan)
constexpr double T_KT_PP128_US = 1.05;    // 256x128x64
constexpr double T_KT_PP160Q_US = 0.55;   // 64x160x64 (short K loops only, like 128x160)

double split_cost_us(int M, int N, int splits) {
  if (splits <= 1) return 0.0;
  return 3.0 + (double)(splits + 1) * M * N * 4.0 / 4.0e6;      // slab write + read at ~4 TB/s, + launch
}

int small_splits(int M, int N, int nkt) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (tiles >= 192) return 1;
  int s = (384 + tiles - 1) / tiles;
  int smax = nkt / 4;
  if (smax < 1) smax = 1;
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

constexpr double kind_tk(int kind) {
  return kind == KIND_PP320 ? T_KT_PP320_US : (kind == KIND_PP256 ? T_KT_PP256_US :
         (kind == KIND_PP160 ? T_KT_PP160_US : (kind == KIND_PP128 ? T_KT_PP128_US :
         (kind == KIND_PP160Q ? T_KT_PP160Q_US : T_KT_PP160H_US))));
}

// Launches with one weight matrix per sample (row_unit = rows per sample > 0): large-tile kernels whose row count
// divides row_unit, no split-K; with `seg` (softmax over 80-column segments in the epilogue) only the 160-column
// shapes, whose waves own exactly one segment.  kind = -1: no shape fits.
Plan make_plan_per_sample(int M, int N, int K, int row_unit, bool seg) {
  const int nkt = K / 64;
  Plan best;
  best.kind = -1; best.splits = 1;
  double best_t = 0.0;
  for (int kind = KIND_PP320; kind <= KIND_PP160Q; ++kind) {
    const int bn = kind == KIND_PP320 ? 320 : (kind == KIND_PP256 ? 256 : (kind == KIND_PP128 ? 128 : 160));
    const int bm = kind == KIND_PP160H ? 128 : (kind == KIND_PP160Q ? 64 : 256);
    if (N % bn != 0 || row_unit % bm != 0 || M % bm != 0) continue;
    if (seg && bn != 160) continue;
    const long tiles = (long)(M / bm) * (N / bn);
    const double t = (double)((tiles + 255) / 256) * (nkt + 3.0) * kind_tk(kind);
    if (best.kind < 0 || t < best_t) { best.kind = kind; best_t = t; }
  }
  return best;
}

Plan make_plan(int M, int N, int K, int dtype, bool geglu, int row_unit = 0, bool seg = false) {
  if (row_unit > 0) {
    if (dtype == SDMI_F32 || geglu) { Plan none; none.kind = -1; none.splits = 1; return none; }
    return make_plan_per_sample(M, N, K, row_unit, seg);
  }
  const int BK = dtype == SDMI_F32 ? 32 : 64;
  const int nkt = K / BK;
  Plan best;
  best.kind = KIND_SMALL;
  best.splits = small_splits(M, N, nkt);
  if (dtype == SDMI_F32 || forced_kind() == 1 || nkt < 2) return best;
  double best_t;
  {
    const long blocks = (long)((M + 127) / 128) * ((N + 127) / 128) * best.splits;
    const int per = (nkt + best.splits - 1) / best.splits;
    best_t = (double)((blocks + 511) / 512) * (per + 1.5) * T_KT_SMALL_US + split_cost_us(M, N, best.splits);
  }
  bool found = false;
  for (int kind = KIND_PP320; kind <= KIND_PP160Q; ++kind) {
    const int bn = kind == KIND_PP320 ? 320 : (kind == KIND_PP256 ? 256 : (kind == KIND_PP128 ? 128 : 160));
    const int bm = kind == KIND_PP160H ? 128 : (kind == KIND_PP160Q ? 64 : 256);
    if (N % bn != 0) continue;
    const bool odd = kind == KIND_PP160 || kind == KIND_PP160H || kind == KIND_PP160Q;   // odd fragment count: no (value, gate) pairs
    if (odd && (geglu || forced_kind() == 2)) continue;
    if ((kind == KIND_PP160H || kind == KIND_PP160Q) && forced_kind() == 5) continue;
    // the 128-column shape moves more operand bytes per flop than the wider ones: only where none of them fits
    if (kind == KIND_PP128 && (N % 160 == 0 || N % 256 == 0) && forced_kind() != 6) continue;
    const double tk = kind == KIND_PP320 ? T_KT_PP320_US : (kind == KIND_PP256 ? T_KT_PP256_US :
                      (kind == KIND_PP160 ? T_KT_PP160_US : (kind == KIND_PP128 ? T_KT_PP128_US :
                      (kind == KIND_PP160Q ? T_KT_PP160Q_US : T_KT_PP160H_US))));
    const long tiles = (long)((M + bm - 1) / bm) * (N / bn);
    for (int s = 1; s <= 16; ++s) {
      if (s > 1 && nkt / s < 8) break;
      const int per = (nkt + s - 1) / s;
      // the 128-row shape pays its extra weight traffic back only while the per-tile fixed cost dominates:
      // measured against 256x160 with split-K it wins up to about 32 K-tiles per workgroup
      if (kind == KIND_PP160H && per > 32 && forced_kind() != 4) continue;
      // the 64-row shape likewise, and sooner: measured ahead of 128x160 up to about 16 K-tiles per workgroup
      if (kind == KIND_PP160Q && per > 16 && forced_kind() != 7) continue;
      const double t = (double)((tiles * s + 255) / 256) * (per + 3.0) * tk + split_cost_us(M, N, s);
      // forced shapes 3 / 4 / 6 / 7: the first legal plan of that kind is taken and only plans of that kind replace it
      const int want = forced_kind() == 3 ? KIND_PP160 : (forced_kind() == 4 ? KIND_PP160H :
                       (forced_kind() == 6 ? KIND_PP128 : (forced_kind() == 7 ? KIND_PP160Q : -1)));
      const bool force = (forced_kind() == 2 && !found) || (kind == want && best.kind != kind);
      const bool locked = want >= 0 && best.kind == want && kind != want;
      if ((t < best_t && !locked) || force) {
        best_t = t; best.kind = kind; best.splits = s; found = true;
      }
    }
  }
  return best;
}

// K order of the large-tile kernels.  Taps innermost pays when a workgroup is the only reader of its
// activation rows (one tile along N): measured -10 % on the 320-channel 3x3 convolutions at 64x64.
// With several tiles along N the neighbouring workgroups of the XCD already share those rows in L2 and
// the tap-major walk, whose loader is a little cheaper, stays ahead (+4..11 % the other way).
// SDMI_GEMM_TAP_INNER=0/1 overrides where legal (tests and experiments; read on every call).
bool use_tap_inner(const GemmParams& p) {
  if (p.taps != 9 || p.up) return false;
  if (p.e0 && p.kt_per_split < p.nkt) return false;      // the tap-inner walk reaches the extra sources only from tile 0
  const char* e = getenv("SDMI_GEMM_TAP_INNER");
  if (e) return atoi(e) != 0;
  return p.tiles_n == 1;
}

// kernel for an explicit split count (callers size the slab with conv_gemm_pick_splits)
int pick_kind(int M, int N, int K, int dtype, bool geglu, int row_unit = 0, bool seg = false) {
  return make_plan(M, N, K, dtype, geglu, row_unit, seg).kind;
}

int kind_bn(int kind) {
  return kind == KIND_PP320 ? 320 : (kind == KIND_PP256 ? 256 : (kind == KIND_SMALL || kind == KIND_PP128 ? 128 : 160));
}
// pairs per row a launch with row_stats_out writes: the large-tile kernels leave one per (tile along N, wave
// group) from their epilogue, everything else a single one
int row_stat_parts(int M, int N, int K, int dtype, bool geglu, int splits, int row_unit = 0) {
  const int kind = pick_kind(M, N, K, dtype, geglu, row_unit);
  if (kind == KIND_SMALL || splits > 1 || geglu || dtype == SDMI_F32) return 1;
  return 2 * (N / kind_bn(kind));
}

template <typename T, typename S>
void launch_pp(GemmParams& p, int splits, hipStream_t stream) {
  p.tiles_m = (p.M + S::BM - 1) / S::BM; p.tiles_n = p.N / S::BN;
  const dim3 grid(p.tiles_m * p.tiles_n, splits), block(S::NTHREADS);
  const bool ti = use_tap_inner(p);
  if (p.w_rows > 0) {
    hipLaunchKernelGGL((conv_gemm_pp_kernel<T, S, false, GEMM_PS>), grid, block, S::LDS_BYTES, stream, p);
  } else if (p.e0) {
    if (ti) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, S, true, GEMM_EX>), grid, block, S::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_gemm_pp_kernel<T, S, false, GEMM_EX>), grid, block, S::LDS_BYTES, stream, p);
  } else if (ti) {
    hipLaunchKernelGGL((conv_gemm_pp_kernel<T, S, true>), grid, block, S::LDS_BYTES, stream, p);
  } else {
    hipLaunchKernelGGL((conv_gemm_pp_kernel<T, S, false>), grid, block, S::LDS_BYTES, stream, p);
  }
}

template <typename T>
int launch(const ConvGemmArgs& a, hipStream_t stream) {
  constexpr int BK = 8 * (16 / (int)sizeof(T));
  GemmParams p;
  p.a0 = (const char*)a.a0; p.a1 = (const char*)a.a1; p.c0 = a.c0; p.c1 = a.a1 ? a.c1 : 0;
  p.w = (const char*)a.w; p.out = (char*)a.out; p.ldo = a.ldo;
  p.bias = a.bias; p.rowadd = a.rowadd; p.resid = (const char*)a.resid; p.ldr = a.ldr;
  p.ldra = a.ldra > 0 ? a.ldra : (a.ldra < 0 ? 0 : a.N); p.out_f32 = a.out_f32;     // < 0: one row for every sample
  p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Ho = a.Ho; p.Wo = a.Wo;
  p.taps = a.taps; p.stride = a.stride; p.up = a.up; p.pad = a.taps == 9 ? a.pad_lo : 0;
  p.e0 = (const char*)a.e0; p.e1 = (const char*)a.e1; p.ec0 = a.e0 ? a.ec0 : 0; p.ec1 = a.e1 ? a.ec1 : 0;
  p.M = a.B * a.Ho * a.Wo; p.N = a.N; p.K = a.taps * (p.c0 + p.c1) + p.ec0 + p.ec1; p.geglu = a.geglu;
  p.dup_rows = a.dup_rows;
  p.ln_stats = a.ln_stats; p.ln_colsum = a.ln_stats ? a.ln_colsum : nullptr;
  p.ln_nparts = a.ln_nparts; p.ln_eps = a.ln_eps;
  p.ln_inv_d = 1.0f / (float)(a.taps * (p.c0 + p.c1));
  p.rs_out = nullptr; p.rs_parts = 0;
  p.w_rows = a.w_rows; p.w_stride = a.w_rows > 0 ? (size_t)a.N * a.taps * (a.c0 + (a.a1 ? a.c1 : 0)) * sizeof(T) : 0;
  p.vec_stride = a.w_rows > 0 && a.vec_per_sample ? a.N : 0; p.softmax_seg = a.softmax_seg;
  SDMI_REQUIRE(a.w_rows >= 0 && (a.w_rows == 0 || (a.taps == 1 && !a.a1 && !a.rowadd && !a.geglu && !a.out_f32 &&
                                                   sizeof(T) == 2 && a.splits <= 1)),
               "conv_gemm: per-sample weights need a single-source 16-bit token GEMM without rowadd / GEGLU / split-K");
  SDMI_REQUIRE(!a.softmax_seg || (a.w_rows > 0 && a.N % 80 == 0 && !a.resid && !a.row_stats_out && !a.dup_rows),
               "conv_gemm: the segment softmax belongs to per-sample score GEMMs (N a multiple of 80, plain store)");
  SDMI_REQUIRE(a.ln_nparts >= 0 && a.ln_nparts <= 16 && (a.ln_nparts == 1 || a.ln_nparts % 2 == 0),
               "conv_gemm: ln_nparts must be 0, 1 or an even number up to 16");
  SDMI_REQUIRE(!a.row_stats_out || (!a.geglu && !a.out_f32), "conv_gemm: row sums are not available with GEGLU / fp32 outputs");
  SDMI_REQUIRE(!a.ln_stats || (a.ln_colsum && a.taps == 1 && !a.a1 && !a.rowadd && (!a.geglu || a.bias)),
               "conv_gemm: a folded LayerNorm needs ln_colsum, a single-source pointwise problem without rowadd "
               "(and a bias with GEGLU)");
  SDMI_REQUIRE(a.dup_rows == 0 || (a.dup_rows >= p.M && !a.geglu && !a.out_f32),
               "conv_gemm: dup_rows must skip past the computed rows and excludes GEGLU / fp32 outputs");
  SDMI_REQUIRE(a.a0 && a.w && a.out, "conv_gemm: null pointer");
  SDMI_REQUIRE((!a.e1 || a.e0) && p.ec0 % BK == 0 && p.ec1 % BK == 0 && (!a.e0 || (p.ec0 > 0 && !a.ln_stats && !a.w_rows)),
               "conv_gemm: extra sources (%d,%d) must be multiples of %d channels, the first one present", p.ec0,
               p.ec1, BK);
  SDMI_REQUIRE(a.taps == 1 || a.taps == 9, "conv_gemm: taps must be 1 or 9 (got %d)", a.taps);
  SDMI_REQUIRE(a.stride == 1 || a.stride == 2, "conv_gemm: stride must be 1 or 2");
  SDMI_REQUIRE(p.c0 % BK == 0 && p.c1 % BK == 0 && p.c0 > 0,
               "conv_gemm: input channels (%d,%d) must be multiples of %d", p.c0, p.c1, BK);
  SDMI_REQUIRE(p.N % 4 == 0 && (!a.geglu || p.N % 32 == 0), "conv_gemm: N=%d not vectorisable", p.N);
  SDMI_REQUIRE(p.M > 0 && p.N > 0, "conv_gemm: empty problem");
  SDMI_REQUIRE(a.ldo % 4 == 0 && (!a.resid || a.ldr % 4 == 0), "conv_gemm: ldo/ldr must be multiples of 4");
  if (a.taps == 1 && a.stride == 1 && !a.up)
    SDMI_REQUIRE(a.Hin * a.Win == a.Ho * a.Wo, "conv_gemm: pointwise needs equal in/out sizes");
  else {
    const int Hup = a.up ? 2 * a.Hin : a.Hin, Wup = a.up ? 2 * a.Win : a.Win;
    const int pad2 = a.taps == 9 ? a.pad_lo + 1 : 0, ks = a.taps == 9 ? 3 : 1;      // trailing pad is always 1
    SDMI_REQUIRE(a.pad_lo == 0 || a.pad_lo == 1, "conv_gemm: pad_lo must be 0 or 1");
    SDMI_REQUIRE((Hup + pad2 - ks) / a.stride + 1 == a.Ho && (Wup + pad2 - ks) / a.stride + 1 == a.Wo,
                 "conv_gemm: output size %dx%d inconsistent with input %dx%d", a.Ho, a.Wo, a.Hin, a.Win);
  }
  p.nkt = p.K / BK;
  int splits = a.splits < 1 ? 1 : a.splits;
  if (splits > p.nkt) splits = p.nkt;
  p.kt_per_split = (p.nkt + splits - 1) / splits;
  splits = (p.nkt + p.kt_per_split - 1) / p.kt_per_split;
  p.partial = splits > 1 ? a.partial : nullptr;
  SDMI_REQUIRE(splits == 1 || a.partial, "conv_gemm: split-K needs a partial buffer");
  SDMI_TRY(conv_gemm_init());
  p.zero = g_zero_page;
  const int kind = pick_kind(p.M, p.N, p.K, a.dtype, a.geglu != 0, a.w_rows, a.softmax_seg != 0);
  SDMI_REQUIRE(kind >= 0, "conv_gemm: no tile shape fits %d rows per sample with N = %d", a.w_rows, p.N);
  SDMI_REQUIRE(!a.e0 || (kind != KIND_SMALL && sizeof(T) == 2),
               "conv_gemm: extra sources need a large-tile kernel (ask conv_gemm_extra_ok first)");
  const int rs_parts = a.row_stats_out ? row_stat_parts(p.M, p.N, p.K, a.dtype, a.geglu != 0, a.splits, a.w_rows) : 0;
  // the epilogue leaves the row sums itself when it is the large-tile kernel's wide store (decided below)
  bool rs_in_kernel = false;
  p.wide_store = a.ldo % 8 == 0 && (!a.resid || a.ldr % 8 == 0) && ((uintptr_t)a.out % 16 == 0) &&
                 (!a.resid || (uintptr_t)a.resid % 16 == 0) && (!p.partial || (uintptr_t)p.partial % 16 == 0) &&
                 !(p.out_f32 && p.geglu) && !direct_store_forced();
  SDMI_REQUIRE(a.w_rows == 0 || p.wide_store, "conv_gemm: per-sample launches need 16-byte addressable operands");
  {
    char label[64];
    snprintf(label, sizeof(label), "M%d N%d K%d t%d s%d u%d g%d x%d%s k%d", p.M, p.N, p.K, p.taps, p.stride, p.up,
             p.geglu, splits, p.softmax_seg ? " sm" : (p.w_rows ? " ps" : (p.e0 ? " +e" : "")), kind);
    ProfScope prof(PROF_CONV_GEMM, stream, 2.0 * p.M * (double)p.N * p.K,
                   ((double)p.M * p.K / (p.taps == 9 ? 9 : 1) + (double)p.N * p.K + (double)p.M * p.N) * sizeof(T),
                   label);
    if (rs_parts > 0) {
      p.rs_parts = rs_parts;
      rs_in_kernel = kind != KIND_SMALL && sizeof(T) == 2 && splits == 1 && p.wide_store &&
                     rs_parts == 2 * (p.N / kind_bn(kind));
      if (rs_in_kernel || splits > 1) p.rs_out = reinterpret_cast<float2*>(a.row_stats_out);
    }
    if constexpr (sizeof(T) == 2) {
      if (kind == KIND_PP320) launch_pp<T, PP320>(p, splits, stream);
      else if (kind == KIND_PP256) launch_pp<T, PP256>(p, splits, stream);
      else if (kind == KIND_PP160) launch_pp<T, PP160>(p, splits, stream);
      else if (kind == KIND_PP160H) launch_pp<T, PP160H>(p, splits, stream);
      else if (kind == KIND_PP128) launch_pp<T, PP128>(p, splits, stream);
      else if (kind == KIND_PP160Q) launch_pp<T, PP160Q>(p, splits, stream);
    }
    if (kind == KIND_SMALL || sizeof(T) != 2) {
      typedef ShapeSmall S;
      p.tiles_m = (p.M + S::BM - 1) / S::BM; p.tiles_n = (p.N + S::BN - 1) / S::BN;
      hipLaunchKernelGGL((conv_gemm_kernel<T, S>), dim3(p.tiles_m * p.tiles_n, splits), dim3(S::NTHREADS),
                         S::LDS_BYTES, stream, p);
    }
  }
  SDMI_CHECK_HIP(hipGetLastError());
  if (splits > 1) {
    const size_t total = (size_t)p.M * (p.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (p.rs_out && !p.geglu && p.N / 4 <= 512) {
      // threads per row: the row's column quads rounded up to whole waves; rows per block: what fits 256 threads
      const int tpr = ((p.N / 4 + 63) / 64) * 64;
      const int rpb = tpr >= 256 ? 1 : 256 / tpr;
      int rb = (p.M + rpb - 1) / rpb;
      if (rb > 16384) rb = 16384;
      hipLaunchKernelGGL(splitk_reduce_rows_kernel<T>, dim3(rb), dim3(rpb * tpr), 0, stream, p, splits, tpr);
    } else {
      hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, stream, p, splits);
      if (p.rs_out && !p.geglu) {               // rows wider than one block: sums from the stored output
        int rb = (p.M + 3) / 4;
        if (rb > 8192) rb = 8192;
        hipLaunchKernelGGL(row_sums_kernel<T>, dim3(rb), dim3(256), 0, stream, p);
      }
    }
    SDMI_CHECK_HIP(hipGetLastError());
  } else if (rs_parts > 0 && !rs_in_kernel) {
    p.rs_out = reinterpret_cast<float2*>(a.row_stats_out);
    int rb = (p.M + 3) / 4;
    if (rb > 8192) rb = 8192;
    hipLaunchKernelGGL(row_sums_kernel<T>, dim3(rb), dim3(256), 0, stream, p);
    SDMI_CHECK_HIP(hipGetLastError());
  }
  return SDMI_OK;
}

}  // namespace

int conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(a.dtype, T, return launch<T>(a, stream));
  return SDMI_OK;
}

int conv_gemm_pick_splits(int M, int N, int K, int dtype, bool geglu) { return make_plan(M, N, K, dtype, geglu).splits; }

int conv_gemm_row_stat_parts(int M, int N, int K, int dtype, bool geglu, int splits, int rows_per_sample) {
  return row_stat_parts(M, N, K, dtype, geglu, splits, rows_per_sample);
}

bool conv_gemm_extra_ok(int M, int N, int K, int dtype) {
  return dtype != SDMI_F32 && make_plan(M, N, K, dtype, false).kind != KIND_SMALL;
}

bool conv_gemm_per_sample_ok(int M, int N, int K, int dtype, int rows_per_sample, bool seg) {
  return rows_per_sample > 0 && make_plan(M, N, K, dtype, false, rows_per_sample, seg).kind >= 0;
}
