// Flash-style attention in full fp32 on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32): the attention of the
// fp32 parity mode, where the 16-bit kernel of attention.hip cannot be used and the vector-ALU kernel of
// attention_scalar.hip ran at a quarter of this one's rate.  fp32 operands, fp32 products, fp32 accumulation and
// softmax statistics; the score matrix never leaves registers.
//
// Work split: one workgroup = 4 waves = 128 queries of one (batch, head); a wave owns two 16-query tiles (one, 64
// queries per workgroup, for small problems).  K and V
// are walked in tiles of 64 (wide heads: 32) keys staged through LDS (rows of D + 4 floats), 16-key blocks inside.
//
// Per 16-key block and query tile:
//   S^T = K Q^T    A = K rows (lane: key lane % 16, four consecutive head dims 16 t + 4 (lane / 16) .. + 3 by one
//                  ds_read_b128, used as the k-slices of four MFMAs), B = the wave's Q fragments, loaded the same way
//                  once and kept in registers (pre-multiplied by scale * log2 e).  The contraction order inside a
//                  group of 16 head dims is permuted alike on both operands, which a sum does not see.
//                  -> lane (g = lane / 16, n = lane % 16) holds the scores of query n against keys 4 g .. 4 g + 3.
//   P^T = exp2(S^T - m): the running maximum of a query needs the lane's four values and two lane exchanges (xor 16,
//                  xor 32); the row sum is kept per lane and folded once at the end.
//   O^T += V^T P^T A = V^T (lane: head dim 16 dt + lane % 16 of key 4 g + r, a ds_read_b32 shared by the wave's two
//                  query tiles), B = P^T: k-step r takes key 4 g + r from lane group g, which is the lane's own
//                  accumulator r -- no data movement between the two products.
//                  -> lane (g, n) holds head dims 16 dt + 4 g .. + 3 of query n: 16-byte stores.
#include "ops.h"

int attention_scalar(const AttentionArgs& a, hipStream_t stream);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct F32AttnParams {
  const float* q; const float* k; const float* v; float* out;
  int ldq, ldk, ldv, ldo;
  int B, heads, Nq, Nk;
  float c;                     // scale * log2(e)
  int causal;
};

// keys per LDS tile: 64, or 32 for the wide heads (three workgroups per CU instead of one)
template <int D> struct F32Tile { static constexpr int KT = D > 80 ? 32 : 64; };

// QT: 16-query tiles per wave (2: 128 queries per workgroup; 1: 64, for problems that would leave CUs idle)
template <int D, int QT>
__global__ __launch_bounds__(256) void attn_mfma_f32_kernel(F32AttnParams p) {
  static_assert(D % 8 == 0, "head size must be a multiple of 8");
  constexpr int G16 = D / 16;                 // full groups of 16 head dims (one ds_read_b128 per lane each)
  constexpr bool TAIL8 = (D % 16) == 8;       // ... and a last group of 8 (ds_read_b64: two MFMAs)
  constexpr int DT = (D + 15) / 16;           // 16-row tiles of the output's head dims
  constexpr int LD = D + 4;                   // LDS row stride in floats (16-byte aligned; = 4 mod 8)
  constexpr int F32_KT = F32Tile<D>::KT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ks = smem;
  float* vs = smem + F32_KT * LD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);    // first query of this wave

  // Q fragments of the two query tiles, scaled
  f32x4 qf[QT][G16 > 0 ? G16 : 1];
  f32x2 qt[QT];
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) {
    const int qi = min(q0 + 16 * t2 + n, p.Nq - 1);
    const float* qp = p.q + ((size_t)b * p.Nq + qi) * p.ldq + h * D;
#pragma unroll
    for (int t = 0; t < G16; ++t) qf[t2][t] = *reinterpret_cast<const f32x4*>(qp + 16 * t + 4 * g) * p.c;
    if constexpr (TAIL8) qt[t2] = *reinterpret_cast<const f32x2*>(qp + 16 * G16 + 2 * g) * p.c;
  }
  f32x4 o[QT][DT];
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[t2][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) { m_run[t2] = -INFINITY; l_run[t2] = 0.f; }

  // the pad columns of the V rows are read by the last output tile of head sizes that are not multiples of 16
  for (int i = tid; i < F32_KT * 4; i += 256) {
    ks[(i >> 2) * LD + D + (i & 3)] = 0.f;
    vs[(i >> 2) * LD + D + (i & 3)] = 0.f;
  }
  if (tid < 16) vs[F32_KT * LD + tid] = 0.f;                    // ... and the last row's reads run past it
  // causal: keys beyond the workgroup's last query contribute nothing
  const int nk_end = p.causal ? min(p.Nk, (int)(blockIdx.x + 1) * (64 * QT)) : p.Nk;
  // K/V tiles go global -> registers -> LDS; the loads of tile t + 1 are in flight while tile t is multiplied
  constexpr int VPR = D / 4;                                    // 16-byte pieces per row
  constexpr int NP = (F32_KT * VPR + 255) / 256;                // pieces per thread and operand
  f32x4 kreg[NP], vreg[NP];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / VPR, cidx = (idx - r * VPR) * 4;
      kreg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      vreg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (idx < F32_KT * VPR && k0 + r < p.Nk) {
        const size_t row = (size_t)b * p.Nk + k0 + r;
        kreg[i] = *reinterpret_cast<const f32x4*>(p.k + row * p.ldk + h * D + cidx);
        vreg[i] = *reinterpret_cast<const f32x4*>(p.v + row * p.ldv + h * D + cidx);
      }
    }
  };
  if (nk_end > 0) load_tile(0);
  for (int k0 = 0; k0 < nk_end; k0 += F32_KT) {
    __syncthreads();                                            // everyone is done with the previous tile
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / VPR, cidx = (idx - r * VPR) * 4;
      if (idx < F32_KT * VPR) {
        *reinterpret_cast<f32x4*>(ks + r * LD + cidx) = kreg[i];
        *reinterpret_cast<f32x4*>(vs + r * LD + cidx) = vreg[i];
      }
    }
    __syncthreads();
    if (k0 + F32_KT < nk_end) load_tile(k0 + F32_KT);
#pragma unroll 1
    for (int kb = 0; kb < F32_KT / 16; ++kb) {
      const int key0 = k0 + kb * 16;
      if (key0 >= nk_end) break;
      // ---- S^T = K Q^T for both query tiles ---------------------------------------------------------
      f32x4 s[QT];
#pragma unroll
      for (int t2 = 0; t2 < QT; ++t2) s[t2] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* kr = ks + (kb * 16 + n) * LD;
#pragma unroll
      for (int t = 0; t < G16; ++t) {
        const f32x4 ka = *reinterpret_cast<const f32x4*>(kr + 16 * t + 4 * g);
#pragma unroll
        for (int t2 = 0; t2 < QT; ++t2) {
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qf[t2][t].x, s[t2], 0, 0, 0);
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qf[t2][t].y, s[t2], 0, 0, 0);
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qf[t2][t].z, s[t2], 0, 0, 0);
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qf[t2][t].w, s[t2], 0, 0, 0);
        }
      }
      if constexpr (TAIL8) {
        const f32x2 ka = *reinterpret_cast<const f32x2*>(kr + 16 * G16 + 2 * g);
#pragma unroll
        for (int t2 = 0; t2 < QT; ++t2) {
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qt[t2].x, s[t2], 0, 0, 0);
          s[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qt[t2].y, s[t2], 0, 0, 0);
        }
      }
      // ---- online softmax ------------------------------------------------------------------------------
      f32x4 pr[QT];
#pragma unroll
      for (int t2 = 0; t2 < QT; ++t2) {
        const int qi = q0 + 16 * t2 + n;
        float sv[4] = {s[t2].x, s[t2].y, s[t2].z, s[t2].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + 4 * g + r;
          if (key >= p.Nk || (p.causal && key > qi)) sv[r] = -INFINITY;
        }
        float mb = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_run[t2], mb);
        // a query that has seen no live key yet keeps everything at zero (exp2(-inf - -inf) would be NaN)
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float corr = __builtin_amdgcn_exp2f(m_run[t2] - m_use);
        m_run[t2] = m_new;
        pr[t2] = (f32x4){__builtin_amdgcn_exp2f(sv[0] - m_use), __builtin_amdgcn_exp2f(sv[1] - m_use),
                         __builtin_amdgcn_exp2f(sv[2] - m_use), __builtin_amdgcn_exp2f(sv[3] - m_use)};
        l_run[t2] = l_run[t2] * corr + ((pr[t2].x + pr[t2].y) + (pr[t2].z + pr[t2].w));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t2][dt] *= corr;
      }
      // ---- O^T += V^T P^T ------------------------------------------------------------------------------
      const float* vr = vs + (kb * 16 + 4 * g) * LD + n;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const float v0 = vr[16 * dt], v1 = vr[LD + 16 * dt], v2 = vr[2 * LD + 16 * dt], v3 = vr[3 * LD + 16 * dt];
#pragma unroll
        for (int t2 = 0; t2 < QT; ++t2) {
          o[t2][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, pr[t2].x, o[t2][dt], 0, 0, 0);
          o[t2][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, pr[t2].y, o[t2][dt], 0, 0, 0);
          o[t2][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, pr[t2].z, o[t2][dt], 0, 0, 0);
          o[t2][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, pr[t2].w, o[t2][dt], 0, 0, 0);
        }
      }
    }
  }
  // ---- normalise and store: lane (g, n) holds head dims 16 dt + 4 g .. + 3 of query n ---------------------
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) {
    float l = l_run[t2];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qi = q0 + 16 * t2 + n;
    if (qi < p.Nq) {
      float* op = p.out + ((size_t)b * p.Nq + qi) * p.ldo + h * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d0 = 16 * dt + 4 * g;
        if (d0 < D) *reinterpret_cast<f32x4*>(op + d0) = o[t2][dt] * inv;
      }
    }
  }
}

template <int D>
int launch_f32(const AttentionArgs& a, hipStream_t stream) {
  F32AttnParams p;
  p.q = (const float*)a.q; p.k = (const float*)a.k; p.v = (const float*)a.v; p.out = (float*)a.out;
  p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
  p.B = a.B; p.heads = a.heads; p.Nq = a.Nq; p.Nk = a.Nk;
  p.c = a.scale * 1.4426950408889634f;
  p.causal = a.causal;
  const size_t lds = (size_t)(2 * F32Tile<D>::KT * (D + 4) + 16) * sizeof(float);
  // two query tiles per wave share every K / V fragment read; with fewer than two workgroups per CU that way, one
  // tile per wave doubles the waves instead
  if ((long)((a.Nq + 127) / 128) * a.B * a.heads >= 512) {
    hipLaunchKernelGGL((attn_mfma_f32_kernel<D, 2>), dim3((a.Nq + 127) / 128, a.B * a.heads), dim3(256), lds, stream, p);
  } else {
    hipLaunchKernelGGL((attn_mfma_f32_kernel<D, 1>), dim3((a.Nq + 63) / 64, a.B * a.heads), dim3(256), lds, stream, p);
  }
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

template <int D>
int init_f32() {
  const int lds = (2 * F32Tile<D>::KT * (D + 4) + 16) * (int)sizeof(float);
  if (lds > 48 * 1024) {
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_mfma_f32_kernel<D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_mfma_f32_kernel<D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  return SDMI_OK;
}

}  // namespace

// Raise dynamic-LDS limits once, outside of stream capture (called from attention_init()).
int attention_f32_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  SDMI_TRY(init_f32<40>()); SDMI_TRY(init_f32<64>()); SDMI_TRY(init_f32<80>());
  SDMI_TRY(init_f32<128>()); SDMI_TRY(init_f32<160>());
  done = true;
  return SDMI_OK;
}

// fp32 attention: the matrix-core kernel for the head sizes it is built for, else the vector-ALU kernel
int attention_f32(const AttentionArgs& a, hipStream_t stream) {
  if (a.ldq % 4 != 0 || a.ldk % 4 != 0 || a.ldv % 4 != 0 || a.ldo % 4 != 0 || ((uintptr_t)a.q | (uintptr_t)a.k |
      (uintptr_t)a.v | (uintptr_t)a.out) % 16 != 0)
    return attention_scalar(a, stream);
  SDMI_TRY(attention_f32_init());
  switch (a.d) {
    case 40: return launch_f32<40>(a, stream);
    case 64: return launch_f32<64>(a, stream);
    case 80: return launch_f32<80>(a, stream);
    case 128: return launch_f32<128>(a, stream);
    case 160: return launch_f32<160>(a, stream);
    default: return attention_scalar(a, stream);
  }
}
