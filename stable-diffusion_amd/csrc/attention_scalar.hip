// Exact-arithmetic attention kernel (fp32 math on the vector ALU, online softmax, no score matrix).
//
// Used (a) as the attention of the fp32 parity mode and (b) for head sizes the MFMA kernel does not
// cover.  A query row is owned by LPQ = d / DPL consecutive lanes, each holding DPL dims of q and of
// the output accumulator in registers; K/V tiles are staged in LDS and read as broadcasts.  Scores
// are processed in chunks of 8 keys so the running-max rescale costs one exp per chunk.
#include "ops.h"

namespace {

struct AttnParams {
  const void* q; const void* k; const void* v; void* out;
  int ldq, ldk, ldv, ldo;
  int B, heads, Nq, Nk, d;
  float scale;
  int causal;
  int tk;           // keys per LDS tile
};

template <typename T, int DPL>
__global__ __launch_bounds__(256) void attn_scalar_kernel(AttnParams p) {
  constexpr int EPV = VecOf<T>::N;
  constexpr int NV = DPL / EPV;
  constexpr int KC = 8;
  typedef typename VecOf<T>::type V;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lpq = p.d / DPL;                       // lanes per query (power of two)
  const int qpb = 256 / lpq;                       // queries per block
  const int sub = threadIdx.x % lpq;
  const int qi = blockIdx.x * qpb + threadIdx.x / lpq;
  const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
  const bool qok = qi < p.Nq;
  const int qrow = qok ? qi : p.Nq - 1;

  float q[DPL], acc[DPL];
  {
    const T* qp = (const T*)p.q + ((size_t)b * p.Nq + qrow) * p.ldq + h * p.d + sub * DPL;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const V v = *reinterpret_cast<const V*>(qp + i * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) q[i * EPV + e] = to_f32<T>(v[e]) * p.scale;
    }
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
  }
  float mrun = -INFINITY, lrun = 0.f;

  T* ks = reinterpret_cast<T*>(smem);
  T* vs = ks + (size_t)p.tk * p.d;
  const int vec_per_row = p.d / EPV;
  for (int k0 = 0; k0 < p.Nk; k0 += p.tk) {
    const int nk = min(p.tk, p.Nk - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nk * vec_per_row; idx += 256) {
      const int r = idx / vec_per_row, c = (idx - r * vec_per_row) * EPV;
      const size_t row = (size_t)b * p.Nk + k0 + r;
      *reinterpret_cast<V*>(ks + r * p.d + c) = *reinterpret_cast<const V*>((const T*)p.k + row * p.ldk + h * p.d + c);
      *reinterpret_cast<V*>(vs + r * p.d + c) = *reinterpret_cast<const V*>((const T*)p.v + row * p.ldv + h * p.d + c);
    }
    __syncthreads();
    for (int j0 = 0; j0 < nk; j0 += KC) {
      float s[KC];
      float cmax = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < KC; ++jj) {
        const int j = j0 + jj;
        float part = 0.f;
        if (j < nk) {
          const T* kr = ks + j * p.d + sub * DPL;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const V v = *reinterpret_cast<const V*>(kr + i * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) part += q[i * EPV + e] * to_f32<T>(v[e]);
          }
        }
        for (int o = 1; o < lpq; o <<= 1) part += __shfl_xor(part, o, 64);
        s[jj] = (j < nk && !(p.causal && k0 + j > qrow)) ? part : -INFINITY;
        cmax = fmaxf(cmax, s[jj]);
      }
      const float mnew = fmaxf(mrun, cmax);
      const float corr = __expf(mrun - mnew);        // exp(-inf) = 0 on the first chunk
      lrun *= corr;
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[e] *= corr;
#pragma unroll
      for (int jj = 0; jj < KC; ++jj) {
        const int j = j0 + jj;
        if (j < nk) {
          const float pj = __expf(s[jj] - mnew);
          lrun += pj;
          const T* vr = vs + j * p.d + sub * DPL;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const V v = *reinterpret_cast<const V*>(vr + i * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[i * EPV + e] += pj * to_f32<T>(v[e]);
          }
        }
      }
      mrun = mnew;
    }
  }
  if (qok) {
    const float inv = 1.0f / lrun;
    T* op = (T*)p.out + ((size_t)b * p.Nq + qi) * p.ldo + h * p.d + sub * DPL;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      V v;
#pragma unroll
      for (int e = 0; e < EPV; ++e) v[e] = from_f32<T>(acc[i * EPV + e] * inv);
      *reinterpret_cast<V*>(op + i * EPV) = v;
    }
  }
}

template <typename T, int DPL>
int launch_dpl(const AttentionArgs& a, hipStream_t stream) {
  AttnParams p;
  p.q = a.q; p.k = a.k; p.v = a.v; p.out = a.out;
  p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo;
  p.B = a.B; p.heads = a.heads; p.Nq = a.Nq; p.Nk = a.Nk; p.d = a.d; p.scale = a.scale; p.causal = a.causal;
  const int lpq = a.d / DPL;
  int tk = 64;
  while (tk > 8 && (size_t)2 * tk * a.d * sizeof(T) > 48 * 1024) tk >>= 1;
  p.tk = tk;
  const size_t lds = (size_t)2 * tk * a.d * sizeof(T);
  const int qpb = 256 / lpq;
  dim3 grid((a.Nq + qpb - 1) / qpb, a.B * a.heads);
  hipLaunchKernelGGL((attn_scalar_kernel<T, DPL>), grid, dim3(256), lds, stream, p);
  SDMI_CHECK_HIP(hipGetLastError());
  return SDMI_OK;
}

template <typename T>
int launch(const AttentionArgs& a, hipStream_t stream) {
  constexpr int EPV = VecOf<T>::N;
  SDMI_REQUIRE(a.q && a.k && a.v && a.out, "attention: null pointer");
  SDMI_REQUIRE(a.B > 0 && a.heads > 0 && a.Nq > 0 && a.Nk > 0 && a.d > 0, "attention: empty problem");
  SDMI_REQUIRE(a.ldq % EPV == 0 && a.ldk % EPV == 0 && a.ldv % EPV == 0 && a.ldo % EPV == 0 && a.d % EPV == 0,
               "attention: strides and head size must be multiples of %d elements", EPV);
  auto pow2_le64 = [](int x) { return x >= 1 && x <= 64 && (x & (x - 1)) == 0; };
  if (a.d % 40 == 0 && pow2_le64(a.d / 40)) return launch_dpl<T, 40>(a, stream);
  if (a.d % 32 == 0 && pow2_le64(a.d / 32)) return launch_dpl<T, 32>(a, stream);
  sdmi_set_error("attention: head size %d not supported (need 40*2^k or 32*2^k)", a.d);
  return SDMI_EINVAL;
}

}  // namespace

int attention_scalar(const AttentionArgs& a, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(a.dtype, T, return launch<T>(a, stream));
  return SDMI_OK;
}
