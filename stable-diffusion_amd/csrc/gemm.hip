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

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 128;           // one operand tile in LDS
constexpr int LDS_BYTES = 4 * TILE_BYTES;       // 2 buffers x (A + W)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];   // never written

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmParams {
  const char* a0; const char* a1; int c0, c1;
  const char* w; char* out; int ldo;
  const float* bias; const float* rowadd; const char* resid; int ldr; int ldra; int out_f32;
  int B, Hin, Win, Ho, Wo, taps, stride, up;
  int M, N, K, geglu;
  int nkt, kt_per_split;
  float* partial;
  int tiles_m, tiles_n;
};

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
}

// GEGLU: value and gate sums for the same 4 output columns.
template <typename T>
__device__ __forceinline__ void epilogue_store_geglu(const GemmParams& p, int m, int n_val, int n_gate,
                                                     int n_out, f32x4 val, f32x4 gate) {
  if (p.bias) {
    val += *reinterpret_cast<const f32x4*>(p.bias + n_val);
    gate += *reinterpret_cast<const f32x4*>(p.bias + n_gate);
  }
  f32x4 v = {val.x * gelu_erf_f(gate.x), val.y * gelu_erf_f(gate.y), val.z * gelu_erf_f(gate.z),
             val.w * gelu_erf_f(gate.w)};
  T* o = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n_out;
  typedef T Tx4 __attribute__((ext_vector_type(4)));
  Tx4 ov = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
  *reinterpret_cast<Tx4*>(o) = ov;
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void conv_gemm_kernel(GemmParams p) {
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                // k elements per tile row (128 bytes)
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

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

  // ---- loader state: 4 activation rows + 4 weight rows per thread ---------------------------
  const int lrow = lane >> 3;                        // row within an 8-row DMA piece
  const int sc = (lane & 7) ^ lrow;                  // swizzled source chunk for LDS chunk lane&7
  const int hw = p.Ho * p.Wo;
  int a_b[4], a_oy[4], a_ox[4];
  bool a_ok[4];
  const char* wptr[4];
  int wstep[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + lrow;
    const int m = m0 + r;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (p.taps == 1 && p.stride == 1 && !p.up) {
      a_b[i] = 0; a_oy[i] = 0; a_ox[i] = mm;          // pointwise: pixel index = row index
    } else {
      const int b = mm / hw, rem = mm - b * hw;
      a_b[i] = b; a_oy[i] = rem / p.Wo; a_ox[i] = rem - (rem / p.Wo) * p.Wo;
    }
    const int n = n0 + r;
    const bool wok = n < p.N;
    wptr[i] = wok ? p.w + ((size_t)n * p.K) * sizeof(T) + sc * 16 : (const char*)g_zero_page + sc * 16;
    wstep[i] = wok ? 128 : 0;
  }
  const bool pointwise = (p.taps == 1 && p.stride == 1 && !p.up);
  const int tiles_c0 = p.c0 / BK, tiles_c1 = p.c1 / BK, tiles_tap = tiles_c0 + tiles_c1;
  const int Hup = p.up ? 2 * p.Hin : p.Hin, Wup = p.up ? 2 * p.Win : p.Win;

  const char* aptr[4];
  int astep[4];
  int seg_tap = -1, seg_src = -1;

  auto set_segment = [&](int tap, int src) {
    seg_tap = tap; seg_src = src;
    const char* base = src ? p.a1 : p.a0;
    const int cs = src ? p.c1 : p.c0;
    const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
      aptr[i] = ok ? base + pix * cs * sizeof(T) + sc * 16 : (const char*)g_zero_page + sc * 16;
      astep[i] = ok ? 128 : 0;
    }
  };

  auto stage = [&](int kt, int buf) {
    const int tap = kt / tiles_tap, rem = kt - tap * tiles_tap;
    const int src = rem >= tiles_c0 ? 1 : 0;
    const int cseg = src ? rem - tiles_c0 : rem;
    if (tap != seg_tap || src != seg_src) set_segment(tap, src);
    char* abuf = smem + buf * 2 * TILE_BYTES;
    char* wbuf = abuf + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(aptr[i] + (size_t)cseg * astep[i], abuf + (wid * 4 + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(wptr[i] + (size_t)kt * wstep[i], wbuf + (wid * 4 + i) * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a tile: row*128 + ((chunk ^ (row&7)) * 16)
  const int frow = lane & 15, fq = lane >> 4;
  int w_off[4], a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w_off[i] = (wn * 64 + i * 16 + frow) * 128;
    a_off[i] = (wm * 64 + i * 16 + frow) * 128;
  }
  const int sw = frow & 7;      // (row & 7) for every fragment row of this lane (16 | row offsets)

  if (kt_begin < kt_end) stage(kt_begin, 0);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt_end) stage(kt + 1, buf ^ 1);
    const char* abuf = smem + buf * 2 * TILE_BYTES;
    const char* wbuf = abuf + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ sw) * 16;
      frag_t wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const frag_t*>(wbuf + w_off[i] + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const frag_t*>(abuf + a_off[j] + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma(wf[i], af[j], acc[i][j]);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------
  const int nbase = n0 + wn * 64 + fq * 4;
  const int mbase = m0 + wm * 64 + frow;
  if (p.partial) {
    float* slab = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mbase + j * 16;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nbase + i * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (p.geglu) {
    const int obase = (n0 + wn * 64) / 2 + fq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mbase + j * 16;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const int n = nbase + i * 16;
        if (n < p.N) epilogue_store_geglu<T>(p, m, n, n + 16, obase + (i / 2) * 16, acc[i][j], acc[i + 1][j]);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = mbase + j * 16;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nbase + i * 16;
      if (n < p.N) epilogue_store<T>(p, m, n, acc[i][j]);
    }
  }
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
      for (int z = 0; z < splits; ++z) {
        const float* s = p.partial + ((size_t)z * p.M + m) * p.N + n;
        val += *reinterpret_cast<const f32x4*>(s);
        gate += *reinterpret_cast<const f32x4*>(s + 16);
      }
      epilogue_store_geglu<T>(p, m, n, n + 16, (n >> 5) * 16 + (n & 15), val, gate);
    } else {
      f32x4 v = {0, 0, 0, 0};
      for (int z = 0; z < splits; ++z)
        v += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)z * p.M + m) * p.N + n);
      epilogue_store<T>(p, m, n, v);
    }
  }
}

}  // namespace

// Raise the dynamic-LDS limit of every instantiation once (must not first happen under stream capture).
int conv_gemm_init() {
  static bool done = false;
  if (done) return SDMI_OK;
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_kernel<float>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_kernel<half_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  SDMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_gemm_kernel<bf16_t>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  done = true;
  return SDMI_OK;
}

namespace {

template <typename T>
int launch(const ConvGemmArgs& a, hipStream_t stream) {
  constexpr int BK = 8 * (16 / (int)sizeof(T));
  GemmParams p;
  p.a0 = (const char*)a.a0; p.a1 = (const char*)a.a1; p.c0 = a.c0; p.c1 = a.a1 ? a.c1 : 0;
  p.w = (const char*)a.w; p.out = (char*)a.out; p.ldo = a.ldo;
  p.bias = a.bias; p.rowadd = a.rowadd; p.resid = (const char*)a.resid; p.ldr = a.ldr;
  p.ldra = a.ldra > 0 ? a.ldra : a.N; p.out_f32 = a.out_f32;
  p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Ho = a.Ho; p.Wo = a.Wo;
  p.taps = a.taps; p.stride = a.stride; p.up = a.up;
  p.M = a.B * a.Ho * a.Wo; p.N = a.N; p.K = a.taps * (p.c0 + p.c1); p.geglu = a.geglu;
  SDMI_REQUIRE(a.a0 && a.w && a.out, "conv_gemm: null pointer");
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
    const int pad = a.taps == 9 ? 1 : 0, ks = a.taps == 9 ? 3 : 1;
    SDMI_REQUIRE((Hup + 2 * pad - ks) / a.stride + 1 == a.Ho && (Wup + 2 * pad - ks) / a.stride + 1 == a.Wo,
                 "conv_gemm: output size %dx%d inconsistent with input %dx%d", a.Ho, a.Wo, a.Hin, a.Win);
  }
  p.nkt = p.K / BK;
  int splits = a.splits < 1 ? 1 : a.splits;
  if (splits > p.nkt) splits = p.nkt;
  p.kt_per_split = (p.nkt + splits - 1) / splits;
  splits = (p.nkt + p.kt_per_split - 1) / p.kt_per_split;
  p.partial = splits > 1 ? a.partial : nullptr;
  SDMI_REQUIRE(splits == 1 || a.partial, "conv_gemm: split-K needs a partial buffer");
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;

  SDMI_TRY(conv_gemm_init());
  ProfScope prof(PROF_CONV_GEMM, stream, 2.0 * p.M * (double)p.N * p.K,
                 ((double)p.M * p.K / (p.taps == 9 ? 9 : 1) + (double)p.N * p.K + (double)p.M * p.N) * sizeof(T));
  dim3 grid(p.tiles_m * p.tiles_n, splits);
  hipLaunchKernelGGL(conv_gemm_kernel<T>, grid, dim3(NTHREADS), LDS_BYTES, stream, p);
  SDMI_CHECK_HIP(hipGetLastError());
  if (splits > 1) {
    const size_t total = (size_t)p.M * (p.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, stream, p, splits);
    SDMI_CHECK_HIP(hipGetLastError());
  }
  return SDMI_OK;
}

}  // namespace

int conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
  SDMI_DISPATCH_DTYPE(a.dtype, T, return launch<T>(a, stream));
  return SDMI_OK;
}

int conv_gemm_pick_splits(int M, int N, int K, int dtype) {
  const int BK = dtype == SDMI_F32 ? 32 : 64;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int nkt = K / BK;
  if (tiles >= 192) return 1;
  int s = (384 + tiles - 1) / tiles;
  int smax = nkt / 4;
  if (smax < 1) smax = 1;
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
