// MFMA implicit-GEMM kernels for gfx950 (see gemm.h).  256 threads = 4 waves, 16x16x32 bf16 MFMA,
// BK = 64, register-staged double-buffered LDS tiles with XOR-swizzled 16-byte chunks
// (conflict-free ds_read_b128 / ds_read_b64_tr_b16), one barrier per K-step, XCD-aware 1-D grid.
#include "gemm.h"

#include <algorithm>
#include <vector>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----------
// slots: 0..3 = gemm_nt <128,128> <128,64> <64,128> <64,64>; 4..7 = gemm_tn <128,128> <128,64> <64,128> <64,64>;
// 8 = conv3x3_halo2<128,14>, 9 = conv3x3_halo2<128,28>, 10 = conv3x3_halo2<64,*>, 11 = conv3x3_halo (v1, all)
namespace {
struct ProfSlot {
  std::vector<hipEvent_t> ev;   // start/stop pairs
  double flops = 0.0;
  long long launches = 0;
};
bool g_prof_on = false;
ProfSlot g_prof[12];
inline int prof_slot(bool tn, int a, int b) { return (tn ? 4 : 0) + (a == 128 ? 0 : 2) + (b == 128 ? 0 : 1); }
struct ProfScope {
  ProfSlot* s = nullptr;
  hipStream_t st;
  ProfScope(int slot, double flops, hipStream_t st_) : st(st_) {
    if (!g_prof_on) return;
    s = &g_prof[slot];
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { s = nullptr; return; }
    s->ev.push_back(a);
    s->ev.push_back(b);
    s->flops += flops;
    s->launches += 1;
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (s) (void)hipEventRecord(s->ev.back(), st);
  }
};
}  // namespace

void gemm_profile_enable(int on) {
  for (auto& s : g_prof) {
    for (auto e : s.ev) (void)hipEventDestroy(e);
    s.ev.clear();
    s.flops = 0.0;
    s.launches = 0;
  }
  g_prof_on = on != 0;
}
// caller must have synchronised the stream(s).  Returns 0 and fills totals for `slot`.
int gemm_profile_read(int slot, double* total_ms, long long* launches, double* flops) {
  if (slot < 0 || slot >= 12) return -1;
  ProfSlot& s = g_prof[slot];
  double ms = 0.0;
  for (size_t i = 0; i + 1 < s.ev.size(); i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.ev[i], s.ev[i + 1]) != hipSuccess) return -2;
    ms += t;
  }
  *total_ms = ms;
  *launches = s.launches;
  *flops = s.flops;
  return 0;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// Tile loads are BRANCH-FREE raw buffer loads: an out-of-range lane gets voffset = num_records and the hardware
// bounds check returns zeros.  (Predicated `if (ok) v = *p` loads made hipcc emit s_waitcnt vmcnt(0) after every
// load, serialising the 8 loads of a K-step: 2x slower.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// bijective XCD remap (blocks b and b+8 share an XCD): gives every XCD a contiguous range of logical ids
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = id & 7, loc = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
}

// =====================================================================================================
// NT kernel
// =====================================================================================================
template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p) {
  constexpr int AI = BM / 32, BI = BN / 32;             // 16-B chunks per thread per tile
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;   // 16x16 fragments per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + NBUF * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int split = blockIdx.y;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kt0 = split * p.ksteps_per_split;
  const int kt1 = min(kt0 + p.ksteps_per_split, p.ksteps_total);

  const int ch = tid & 7, rbase = tid >> 3;
  int a_hb[AI], a_wb[AI], a_pix[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + rbase + 32 * i;
    a_ok[i] = m < p.M;
    if (p.mode == 1) {
      const int mm = a_ok[i] ? m : 0;
      const int hw = p.Ho * p.Wo;
      const int img = mm / hw, rem = mm - img * hw;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hb[i] = ho * p.stride - p.pad;
      a_wb[i] = wo * p.stride - p.pad;
      a_pix[i] = img * p.H * p.W;
    } else {
      a_hb[i] = a_wb[i] = 0;
      a_pix[i] = m;
    }
  }
  // tap state for k-step kt: (r, s, cc)
  int tap = kt0 / p.cpt, cc = kt0 - tap * p.cpt;
  int r = tap / p.S, s = tap - r * p.S;

  uint4 ra[AI], rb[BI];

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const int upm = p.up - 1, ups = p.up >> 1;     // up in {1,2}: parity mask / shift
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      unsigned off;
      bool ok = a_ok[i];
      if (p.mode == 1) {
        int hp = a_hb[i] + r, wp = a_wb[i] + s;
        ok = ok && (((hp | wp) & upm) == 0);
        hp >>= ups;
        wp >>= ups;
        ok = ok && (unsigned)hp < (unsigned)p.H && (unsigned)wp < (unsigned)p.W;
        off = ((unsigned)(a_pix[i] + hp * p.W + wp) * (unsigned)p.C + (unsigned)(cc * 64 + ch * 8)) * 2u;
      } else {
        const int k = kt * 64 + ch * 8;
        ok = ok && k < p.K;
        off = ((unsigned)a_pix[i] * (unsigned)p.lda + (unsigned)k) * 2u;
      }
      ra[i] = buf_load16(rsA, ok ? off : p.a_bytes);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rbase + 32 * i;
      const int k = kt * 64 + ch * 8;
      const unsigned off = ((unsigned)n * (unsigned)p.K + (unsigned)k) * 2u;
      rb[i] = buf_load16(rsB, (n < p.N && k < p.K) ? off : p.b_bytes);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sA + buf * A_BYTES + row * 128 + ((ch ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sB + buf * B_BYTES + row * 128 + ((ch ^ (row & 7)) << 4)) = rb[i];
    }
  };
  auto advance = [&]() {
    if (++cc == p.cpt) {
      cc = 0;
      if (++s == p.S) {
        s = 0;
        ++r;
      }
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();

  const int l15 = lane & 15, lg = lane >> 4;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
    const bool more = kt + 1 < kt1;
    if (more) {
      advance();
      load_tiles(kt + 1);
    }
        const unsigned char* cA = sA + buf * A_BYTES;
    const unsigned char* cB = sB + buf * B_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + lg;
      bf16x8_t fb[TN], fa[TM];
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int row = wn * (BN / WN) + ni * 16 + l15;
        fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = wm * (BM / WM) + mi * 16 + l15;
        fa[mi] = *reinterpret_cast<const bf16x8_t*>(cA + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
    }
    if (NBUF == 1) __syncthreads();          // every wave is done reading the single buffer
    if (more) store_tiles(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue.  acc[ni][mi][reg]: n = wn*(BN/WN)+ni*16+lg*4+reg ; m = wm*(BM/WM)+mi*16+l15 ----
  if (p.Cf) {
    float* slab = p.Cf + (size_t)split * p.M * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4;
        if (m < p.M && n < p.N) *reinterpret_cast<float4*>(slab + (size_t)m * p.N + n) =
            make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
      }
    return;
  }
  constexpr int CST = BN * 2 + 16;   // staged C row stride in bytes
  unsigned char* sC = smem;          // all waves are past the last barrier of the K loop
  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int ml = wm * (BM / WM) + mi * 16 + l15;
      const int nl = wn * (BN / WN) + ni * 16 + lg * 4;
      bf16_t h[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = f2bf(acc[ni][mi][q]);
        const float v = bf2f(h[q]);
        ssum[ni][q] += v;
        ssq[ni][q] += v * v;
      }
      uint2 pk;
      pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
      pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
      *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
    }
  if (p.stats) {
    float* prow = p.stats + (size_t)(bm * WM + wm) * 2 * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = ssum[ni][q], b = ssq[ni][q];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0 && n < p.N) {
          prow[n] = a;
          prow[p.N + n] = b;
        }
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;   // 16-B chunks per staged row
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int m = m0 + row, n = n0 + c * 8;
    if (m < p.M && n < p.N)
      *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
  }
}

int g_nt_nbuf = 2;   // option "nt_nbuf": LDS stages of the NT kernel (1 -> 4 blocks/CU, 2 -> one barrier per K-step)

template <int BM, int BN, int WM, int WN, int NBUF>
static int launch_nt_impl(const GemmNT& p0, int splits, hipStream_t st);

template <int BM, int BN, int WM, int WN>
static int launch_nt(const GemmNT& p0, int splits, hipStream_t st) {
  if (g_nt_nbuf == 1) return launch_nt_impl<BM, BN, WM, WN, 1>(p0, splits, st);
  return launch_nt_impl<BM, BN, WM, WN, 2>(p0, splits, st);
}

template <int BM, int BN, int WM, int WN, int NBUF>
static int launch_nt_impl(const GemmNT& p0, int splits, hipStream_t st) {
  GemmNT p = p0;
  const int nbm = ceil_div(p.M, BM);
  p.nbn = ceil_div(p.N, BN);
  p.ksteps_per_split = ceil_div(p.ksteps_total, splits);
  const int real_splits = ceil_div(p.ksteps_total, p.ksteps_per_split);
  FEDFR_REQUIRE(real_splits == splits, "gemm_nt: splits=%d leaves an empty split (ksteps=%d)", splits, p.ksteps_total);
  constexpr size_t kStage = (size_t)(BM + BN) * 128, kEpi = (size_t)BM * (BN * 2 + 16);
  const size_t lds = (NBUF * kStage > kEpi) ? NBUF * kStage : kEpi;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, WM, WN, NBUF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(nbm * p.nbn, splits, 1);
  ProfScope prof(prof_slot(false, BM, BN), 2.0 * p.M * p.N * (double)p.K, st);
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, NBUF>), grid, dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_nt");
  return FEDFR_OK;
}

// =====================================================================================================
// 3x3 / stride-1 / pad-1 convolution (fwd, and dgrad with the flipped shadow) with an LDS-resident HALO tile.
// The generic kernel above re-fetches the 128-pixel activation tile for each of the 9 taps; here the
// 128 + 2(W+1) consecutive NHWC pixels a tile can touch are staged ONCE per 64-channel chunk and every tap reads
// its shifted window from LDS (row = pixel + r*W + s), masked per lane for image borders.  Activation traffic
// through L2->CU and VGPR->LDS drops ~7x (14x14) .. 3x (112x112); the weight tile [BN][64] per (tap, chunk) stays
// register-staged and double buffered.  K order: chunk outer, tap inner (only the fp32 summation order changes).
// =====================================================================================================
template <int BN, int AH>   // AH = halo 16-B chunks per thread = ceil((128 + 2W + 2) * 8 / 256)
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(GemmNT p, int a_bytes_lds) {
  constexpr int BM = 128, WM = 2, WN = 2;
  constexpr int BI = BN / 32;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + a_bytes_lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int W = p.W, NR = BM + 2 * W + 2;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ch = tid & 7, rbase = tid >> 3;
  const int npix = p.M;                       // stride 1: input pixels == output pixels
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

  // per-lane 9-bit tap validity for each of the TM fragment rows this lane feeds
  unsigned vmask[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
    unsigned msk = 0;
    if (m < p.M) {
      const int hw = p.H * W;
      const int rem = m % hw;
      const int h = rem / W, w = rem - h * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if ((unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)W) msk |= 1u << t;
      }
    }
    vmask[mi] = msk;
  }

  uint4 rh[AH], rb[BI];
  auto load_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 3, c = e & 7;
      const int pix = m0 - (W + 1) + rl;
      const bool ok = rl < NR && (unsigned)pix < (unsigned)npix;
      const unsigned off = ((unsigned)pix * (unsigned)p.C + (unsigned)(cc * 64 + c * 8)) * 2u;
      rh[i] = buf_load16(rsA, ok ? off : p.a_bytes);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 3, c = e & 7;
      if (rl < NR) *reinterpret_cast<uint4*>(sA + rl * 128 + ((c ^ (rl & 7)) << 4)) = rh[i];
    }
  };
  auto load_b = [&](int tap, int cc) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rbase + 32 * i;
      const unsigned off = ((unsigned)n * (unsigned)p.K + (unsigned)(tap * p.C + cc * 64 + ch * 8)) * 2u;
      rb[i] = buf_load16(rsB, n < p.N ? off : p.b_bytes);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sB + buf * B_BYTES + row * 128 + ((ch ^ (row & 7)) << 4)) = rb[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int cpt = p.C >> 6;
  load_halo(0);
  load_b(0, 0);
  store_halo();
  store_b(0);
  __syncthreads();
  int buf = 0;
  const bf16x8_t zfrag = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
  for (int cc = 0; cc < cpt; ++cc) {
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_tap = tap == 8;
      const bool more = !(last_tap && cc + 1 == cpt);
      if (more) load_b(last_tap ? 0 : tap + 1, last_tap ? cc + 1 : cc);
      if (last_tap && cc + 1 < cpt) load_halo(cc + 1);
      const int r = tap / 3, sft = r * W + (tap - 3 * r);
      const unsigned char* cB = sB + buf * B_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + lg;
        bf16x8_t fb[TN], fa[TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          const int row = wn * (BN / WN) + ni * 16 + l15;
          fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + row * 128 + ((c ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          const int row = wm * (BM / WM) + mi * 16 + l15 + sft;
          const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(sA + row * 128 + ((c ^ (row & 7)) << 4));
          fa[mi] = ((vmask[mi] >> tap) & 1u) ? v : zfrag;
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
      }
      if (more) store_b(buf ^ 1);
      __syncthreads();                       // next weight tile visible; everyone is done with this tap's reads
      if (last_tap && cc + 1 < cpt) {
        store_halo();                        // safe: all waves passed the barrier above => no reader of the old halo
        __syncthreads();
      }
      buf ^= 1;
    }
  }

  // ---- epilogue (identical to gemm_nt_kernel's bf16 path) ----
  constexpr int CST = BN * 2 + 16;
  unsigned char* sC = smem;
  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int ml = wm * (BM / WM) + mi * 16 + l15;
      const int nl = wn * (BN / WN) + ni * 16 + lg * 4;
      bf16_t h[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = f2bf(acc[ni][mi][q]);
        const float v = bf2f(h[q]);
        ssum[ni][q] += v;
        ssq[ni][q] += v * v;
      }
      uint2 pk;
      pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
      pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
      *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
    }
  if (p.stats) {
    float* prow = p.stats + (size_t)(bm * WM + wm) * 2 * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = ssum[ni][q], b = ssq[ni][q];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0 && n < p.N) {
          prow[n] = a;
          prow[p.N + n] = b;
        }
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int m = m0 + row, n = n0 + c * 8;
    if (m < p.M && n < p.N)
      *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
  }
}

int g_halo_waves = 4;  // option "halo_waves": 4 or 8 waves per 128x128 tile in the halo2 kernel
int g_halo_bn64 = 0;   // option "halo_bn64": 64-wide N tiles in the halo2 kernel (more, smaller blocks)
int g_conv_halo = 2;   // option "conv_halo": 0 generic gather kernel, 1 halo v1 (masked, swizzled), 2 zero-padded image (W=14/28) else v1

template <int BN, int AH>
static int launch_halo(GemmNT p, hipStream_t st) {
  const int nbm = ceil_div(p.M, 128);
  p.nbn = ceil_div(p.N, BN);
  const int NR = 128 + 2 * p.W + 2;
  const int a_lds = (int)align_up((size_t)NR * 128, 256);
  constexpr size_t kEpi = (size_t)128 * (BN * 2 + 16);
  size_t lds = (size_t)a_lds + 2 * (size_t)BN * 128;
  if (lds < kEpi) lds = kEpi;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN, AH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  ProfScope prof(11, 2.0 * p.M * p.N * (double)p.K, st);
  hipLaunchKernelGGL((conv3x3_halo_kernel<BN, AH>), dim3(nbm * p.nbn), dim3(256), lds, st, p, a_lds);
  FEDFR_LAUNCH_CHECK("conv3x3_halo");
  return FEDFR_OK;
}

// =====================================================================================================
// halo kernel v2 (W = 14 / 28, i.e. 84 of iresnet100's 103 convs): the LDS image is laid out in ZERO-PADDED image
// coordinates — every image row gets a zero pixel left and right, every image a zero row above and below — so a
// filter tap is a pure constant shift ((r*(W+2) + s) rows) with NO per-lane border masks, and with a linear
// 160-byte row stride (conflict-free for ds_read_b128 without XOR) + W as a template constant the 9 tap offsets
// are instruction immediates.  v1 spent 136 VALU instructions per 32 MFMAs on masks and swizzled addresses.
// =====================================================================================================
template <int BN, int W_, int WN>   // WN = 2: 4 waves (64x64 wave tiles at BN=128); WN = 4: 8 waves (64x32)
__global__ __launch_bounds__(128 * WN) void conv3x3_halo2_kernel(GemmNT p, int nr_rows) {
  constexpr int BM = 128, WM = 2, PW = W_ + 2, RS = 160, NT = 64 * WM * WN;
  constexpr int BI = BN * 8 / NT, BROWS = NT / 8;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int B_BYTES = BN * 128;
  constexpr int NSRC = BM + 2 * W_ + 2;                 // source pixels a tile can touch
  constexpr int AH = (NSRC * 8 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  const int a_bytes_lds = (nr_rows * RS + 255) & ~255;
  unsigned char* sB = smem + a_bytes_lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = lid % p.nbn, bm = lid / p.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ch = tid & 7, rbase = tid >> 3;
  const int npix = p.M, HW = p.H * W_, PIMG = (p.H + 2) * PW;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  auto qof = [&](int pix) {                             // padded coordinate of a flattened (img, h, w) pixel
    const int img = pix / HW, rem = pix - img * HW;
    const int h = rem / W_, w = rem - h * W_;
    return img * PIMG + (h + 1) * PW + w + 1;
  };
  const int qb = qof(m0) - (PW + 1);

  // zero the whole image once: padding positions are never written afterwards
  for (int i = tid * 16; i < a_bytes_lds; i += NT * 16) *reinterpret_cast<uint4*>(sA + i) = make_uint4(0, 0, 0, 0);

  // staging plan (same rows every chunk): source byte offset (without the channel-chunk term) and LDS destination
  unsigned src_off[AH];
  int dst_off[AH];
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    const int e = tid + NT * i;
    const int rl = e >> 3, c = e & 7;
    const int pix = m0 - (W_ + 1) + rl;
    bool ok = rl < NSRC && (unsigned)pix < (unsigned)npix;
    int row = 0;
    if (ok) {
      row = qof(pix) - qb;
      ok = (unsigned)row < (unsigned)nr_rows;
    }
    src_off[i] = ok ? ((unsigned)pix * (unsigned)p.C + (unsigned)(c * 8)) * 2u : 0xffffffffu;
    dst_off[i] = ok ? row * RS + c * 16 : -1;
  }
  // A-fragment base addresses (tap (0,0)); rows >= M are clamped (their results are masked in the epilogue)
  int a_addr[TM];
  bool m_ok[TM];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int m = m0 + wm * (BM / WM) + mi * 16 + l15;
    m_ok[mi] = m < p.M;
    a_addr[mi] = (qof(m_ok[mi] ? m : p.M - 1) - (PW + 1) - qb) * RS + lg * 16;
  }

  uint4 rh[AH], rb[BI];
  auto load_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < AH; ++i)
      rh[i] = buf_load16(rsA, src_off[i] == 0xffffffffu ? p.a_bytes : src_off[i] + (unsigned)(cc * 128));
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < AH; ++i)
      if (dst_off[i] >= 0) *reinterpret_cast<uint4*>(sA + dst_off[i]) = rh[i];
  };
  const unsigned b_row0 = ((unsigned)(n0 + rbase) * (unsigned)p.K + (unsigned)(ch * 8)) * 2u;
  auto load_b = [&](int tap, int cc) {
    const unsigned koff = (unsigned)(tap * p.C + cc * 64) * 2u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rbase + BROWS * i;
      rb[i] = buf_load16(rsB, n < p.N ? b_row0 + (unsigned)(BROWS * i) * (unsigned)p.K * 2u + koff : p.b_bytes);
    }
  };
  const int b_st = rbase * 128 + ((ch ^ (rbase & 7)) << 4);
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<uint4*>(sB + buf * B_BYTES + b_st + i * BROWS * 128) = rb[i];
  };
  int b_addr[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int row = wn * (BN / WN) + ni * 16 + l15;
    b_addr[ni] = row * 128 + ((lg ^ (row & 7)) << 4);      // ks = 1 flips chunk bit 2: XOR 64
  }

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int cpt = p.C >> 6;
  load_halo(0);
  load_b(0, 0);
  __syncthreads();                   // zero fill complete before real pixels land
  store_halo();
  store_b(0);
  __syncthreads();
  int buf = 0;
  for (int cc = 0; cc < cpt; ++cc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      constexpr int kDummy = 0;
      (void)kDummy;
      const bool last_tap = tap == 8;
      const bool more = !(last_tap && cc + 1 == cpt);
      if (more) load_b(last_tap ? 0 : tap + 1, last_tap ? cc + 1 : cc);
      if (last_tap && cc + 1 < cpt) load_halo(cc + 1);
      const int toff = ((tap / 3) * PW + (tap % 3)) * RS;          // compile-time per unrolled tap
      const unsigned char* cB = sB + buf * B_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fb[TN], fa[TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(cB + (b_addr[ni] ^ (ks * 64)));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(sA + a_addr[mi] + toff + ks * 64);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = MFMA16(fb[ni], fa[mi], acc[ni][mi]);
      }
      if (more) store_b(buf ^ 1);
      __syncthreads();
      if (last_tap && cc + 1 < cpt) {
        store_halo();
        __syncthreads();
      }
      buf ^= 1;
    }
  }

  // ---- epilogue: as gemm_nt_kernel's bf16 path; rows >= M contribute nothing to the statistics ----
  constexpr int CST = BN * 2 + 16;
  unsigned char* sC = smem;
  float ssum[TN][4], ssq[TN][4];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[ni][q] = ssq[ni][q] = 0.f;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int ml = wm * (BM / WM) + mi * 16 + l15;
      const int nl = wn * (BN / WN) + ni * 16 + lg * 4;
      bf16_t h[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = f2bf(acc[ni][mi][q]);
        const float v = m_ok[mi] ? bf2f(h[q]) : 0.f;
        ssum[ni][q] += v;
        ssq[ni][q] += v * v;
      }
      uint2 pk;
      pk.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
      pk.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
      *reinterpret_cast<uint2*>(sC + ml * CST + nl * 2) = pk;
    }
  if (p.stats) {
    float* prow = p.stats + (size_t)(bm * WM + wm) * 2 * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = ssum[ni][q], b = ssq[ni][q];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = n0 + wn * (BN / WN) + ni * 16 + lg * 4 + q;
        if (l15 == 0 && n < p.N) {
          prow[n] = a;
          prow[p.N + n] = b;
        }
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;
  if (p.bpart == nullptr) {
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      const int row = idx / CPR, c = idx - row * CPR;
      const int m = m0 + row, n = n0 + c * 8;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
    }
    return;
  }
  // ---- fused BN-backward reduction: this thread owns chunk column c (8 channels) of rows rg, rg + NT/CPR, ... ----
  constexpr int RG = NT / CPR;
  const int c = tid % CPR, rg = tid / CPR;
  const int n = n0 + c * 8;
  const bool n_ok = n < p.N;
  float mean[8], rstd[8], ga[8], be[8], al[8], s1[8], s2[8], s3[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int nn = n_ok ? n + q : 0;
    mean[q] = p.bmean[nn]; rstd[q] = p.brstd[nn];
    ga[q] = p.bgamma ? p.bgamma[nn] : 1.f; be[q] = p.bbeta ? p.bbeta[nn] : 0.f; al[q] = p.balpha ? p.balpha[nn] : 1.f;
    s1[q] = s2[q] = s3[q] = 0.f;
  }
  const bool has_alpha = p.balpha != nullptr;
  for (int row = rg; row < BM; row += RG) {
    const int m = m0 + row;
    if (m < p.M && n_ok) {
      const uint4 dv = *reinterpret_cast<const uint4*>(sC + row * CST + c * 16);
      *reinterpret_cast<uint4*>(p.Cb + (size_t)m * p.ldc + n) = dv;
      float dy[8], xv[8];
      unpack8(dv, dy);
      unpack8(*reinterpret_cast<const uint4*>(p.bx + (size_t)m * p.N + n), xv);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xh = (xv[q] - mean[q]) * rstd[q];
        float dz = dy[q];
        if (has_alpha) {
          const float z = ga[q] * xh + be[q];
          if (z <= 0.f) {
            s3[q] += dy[q] * z;
            dz = dy[q] * al[q];
          }
        }
        s1[q] += dz;
        s2[q] += dz * xh;
      }
    }
  }
  __syncthreads();                                   // everyone is done reading the staged C tile
  float* red = reinterpret_cast<float*>(smem);       // [RG][3][BN]
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    red[(rg * 3 + 0) * BN + c * 8 + q] = s1[q];
    red[(rg * 3 + 1) * BN + c * 8 + q] = s2[q];
    red[(rg * 3 + 2) * BN + c * 8 + q] = s3[q];
  }
  __syncthreads();
  for (int i = tid; i < 3 * BN; i += NT) {
    const int which = i / BN, col = i - which * BN;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) t += red[(r * 3 + which) * BN + col];
    if (n0 + col < p.N) p.bpart[((size_t)bm * 3 + which) * p.N + n0 + col] = t;
  }
}

template <int BN, int W_, int WN>
static int launch_halo2(GemmNT p, hipStream_t st) {
  const int nbm = ceil_div(p.M, 128);
  p.nbn = ceil_div(p.N, BN);
  // exact LDS image height: max over tiles of the padded-coordinate span, plus the halo on both sides
  const int PW = W_ + 2, HW = p.H * W_, PIMG = (p.H + 2) * PW;
  auto qof = [&](int pix) {
    const int img = pix / HW, rem = pix - img * HW;
    return img * PIMG + (rem / W_ + 1) * PW + rem % W_ + 1;
  };
  int span = 0;
  for (int t = 0; t < nbm; ++t) {
    const int a = t * 128, b = (a + 127 < p.M - 1) ? a + 127 : p.M - 1;
    span = std::max(span, qof(b) - qof(a));
  }
  const int nr = span + 2 * (PW + 1) + 1;
  const size_t a_lds = ((size_t)nr * 160 + 255) & ~(size_t)255;
  constexpr size_t kEpi = (size_t)128 * (BN * 2 + 16), kRed = (size_t)(128 * WN / (BN / 8)) * 3 * BN * sizeof(float);
  size_t lds = a_lds + 2 * (size_t)BN * 128;
  if (lds < kEpi) lds = kEpi;
  if (lds < kRed) lds = kRed;
  FEDFR_REQUIRE(lds <= 160 * 1024, "conv3x3_halo2: LDS image too large (%zu bytes)", lds);
  if (p.bpart) {
    FEDFR_REQUIRE(p.bx && p.bmean && p.brstd && p.ldc == p.N, "conv3x3_halo2: fused BN-bwd reduction needs bx/mean/rstd and ldc == N");
    if (p.bwd_fused) *p.bwd_fused = nbm;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo2_kernel<BN, W_, WN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  ProfScope prof(BN == 64 ? 10 : (W_ == 14 ? 8 : 9), 2.0 * p.M * p.N * (double)p.K, st);
  hipLaunchKernelGGL((conv3x3_halo2_kernel<BN, W_, WN>), dim3(nbm * p.nbn), dim3(128 * WN), lds, st, p, nr);
  FEDFR_LAUNCH_CHECK("conv3x3_halo2");
  return FEDFR_OK;
}

static inline int nt_bm(int M, int N) {
  // 128-row tiles unless that leaves fewer than ~1.5 tiles per CU (256 CUs)
  const int bn = (N <= 64) ? 64 : 128;
  const long long tiles128 = (long long)ceil_div(M, 128) * ceil_div(N, bn);
  return (tiles128 >= 384) ? 128 : 64;
}

int gemm_nt_pick_splits(int M, int N, int K) {
  const int BM = nt_bm(M, N);
  const int bn = (N <= 64) ? 64 : 128;
  const int tiles = ceil_div(M, BM) * ceil_div(N, bn);
  const int ksteps = ceil_div(K, 64);
  int splits = ceil_div(512, tiles);
  if (splits > ksteps / 4) splits = ksteps / 4;
  if (splits < 1) splits = 1;
  const int per = ceil_div(ksteps, splits);
  return ceil_div(ksteps, per);
}

int gemm_nt_stat_rows(int M, int N) {
  // must mirror the tile choice in gemm_nt_launch: rows = ceil(M/BM) * WM
  const int BM = nt_bm(M, N);
  const int WM = (BM == 128) ? 2 : ((N <= 64) ? 2 : 1);
  return ceil_div(M, BM) * WM;
}

int gemm_nt_launch(GemmNT p, int splits, hipStream_t st) {
  FEDFR_REQUIRE(p.A && p.B && p.M > 0 && p.N > 0 && p.K > 0, "gemm_nt: null/empty operand");
  FEDFR_REQUIRE((p.K & 7) == 0, "gemm_nt: K=%d must be a multiple of 8", p.K);
  FEDFR_REQUIRE((p.Cb != nullptr) != (p.Cf != nullptr), "gemm_nt: exactly one of bf16 / fp32-slab outputs");
  if (p.Cb) FEDFR_REQUIRE((p.N & 7) == 0 && (p.ldc & 7) == 0 && splits == 1, "gemm_nt: bf16 output needs N%%8==0, ldc%%8==0, splits==1");
  if (p.Cf) FEDFR_REQUIRE((p.N & 3) == 0, "gemm_nt: fp32 output needs N%%4==0");
  if (p.mode == 1) {
    FEDFR_REQUIRE((p.C & 63) == 0, "gemm_nt: gather needs C%%64==0 (C=%d)", p.C);
    p.cpt = p.C / 64;
    FEDFR_REQUIRE(p.K % p.C == 0 && p.S > 0 && (p.K / p.C) % p.S == 0, "gemm_nt: K must be taps*C");
    FEDFR_REQUIRE(p.up == 1 || p.up == 2, "gemm_nt: up must be 1 or 2");
  } else {
    FEDFR_REQUIRE((p.lda & 7) == 0, "gemm_nt: lda%%8");
    p.cpt = 1 << 30;
    p.S = 1;
  }
  p.ksteps_total = ceil_div(p.K, 64);
  {
    const unsigned long long ab = p.mode == 1 ? 2ull * ((unsigned long long)ceil_div(p.M, p.Ho * p.Wo)) * p.H * p.W * p.C
                                              : 2ull * (unsigned long long)p.M * p.lda;
    const unsigned long long bb = 2ull * (unsigned long long)p.N * p.K;
    FEDFR_REQUIRE(ab < (1ull << 32) - 64 && bb < (1ull << 32) - 64, "gemm_nt: operand larger than 4 GiB (32-bit buffer offsets)");
    p.a_bytes = (unsigned)ab;
    p.b_bytes = (unsigned)bb;
  }
  const int BM = nt_bm(p.M, p.N);
  if (g_conv_halo && BM == 128 && p.mode == 1 && p.S == 3 && p.K == 9 * p.C && p.stride == 1 && p.pad == 1 && p.up == 1 &&
      p.H == p.Ho && p.W == p.Wo && p.Cb && splits == 1 && p.W <= 126) {
    if (g_conv_halo >= 2 && p.H == p.W && (p.W == 14 || p.W == 28)) {
      const bool bn64 = p.N <= 64 || g_halo_bn64;
      if (p.W == 14) return bn64 ? launch_halo2<64, 14, 2>(p, st) : (g_halo_waves == 8 ? launch_halo2<128, 14, 4>(p, st) : launch_halo2<128, 14, 2>(p, st));
      return bn64 ? launch_halo2<64, 28, 2>(p, st) : (g_halo_waves == 8 ? launch_halo2<128, 28, 4>(p, st) : launch_halo2<128, 28, 2>(p, st));
    }
    const int ah = ceil_div((128 + 2 * p.W + 2) * 8, 256);
    if (p.N <= 64) {
      if (ah <= 6) return launch_halo<64, 6>(p, st);
      if (ah <= 8) return launch_halo<64, 8>(p, st);
      return launch_halo<64, 12>(p, st);
    }
    if (ah <= 6) return launch_halo<128, 6>(p, st);
    if (ah <= 8) return launch_halo<128, 8>(p, st);
    return launch_halo<128, 12>(p, st);
  }
  if (BM == 128) {
    if (p.N <= 64) return launch_nt<128, 64, 2, 2>(p, splits, st);
    return launch_nt<128, 128, 2, 2>(p, splits, st);
  }
  if (p.N <= 64) return launch_nt<64, 64, 2, 2>(p, splits, st);
  return launch_nt<64, 128, 1, 4>(p, splits, st);
}

// =====================================================================================================
// TN kernel
// =====================================================================================================
template <int RB>   // tile row bytes (128 or 256): chunk swizzle that makes tr-reads and b128 writes conflict-free
__device__ __forceinline__ int tn_swz(int row) {
  if (RB == 256) return ((row & 3) << 1) | (((row >> 3) & 1) << 3);
  return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
}

// per-lane byte offset (inside one LDS stage) of the first tr-read of the fragment for column block `colblk`;
// the second read is +4*RB, the ks=1 half +32*RB (the swizzle only depends on row bits 0-1 and 3, which those
// offsets do not touch), so every in-loop address is base + register + immediate.
template <int RB>
__device__ __forceinline__ int tn_frag_off(int colblk, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int col = colblk + 4 * pp;
  const int r0 = 8 * g + q;
  return r0 * RB + (((col >> 3) ^ tn_swz<RB>(r0)) << 4) + ((col >> 2) & 1) * 8;
}
template <int RB>
__device__ __forceinline__ bf16x8_t tn_frag_tr(const unsigned char* stage, int off, int ks) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const unsigned char* a0 = stage + off + ks * 32 * RB;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * RB));
  s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
template <int RB>   // validation fallback: scalar LDS gathers (no transpose-read instruction)
__device__ __forceinline__ bf16x8_t tn_frag_scalar(const unsigned char* tile, int ks, int colblk, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int col = colblk + li;
  const int c = col >> 3, e = col & 7;
  s16x8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int rr = ks * 32 + 8 * g + j;
    v[j] = *reinterpret_cast<const short*>(tile + rr * RB + ((c ^ tn_swz<RB>(rr)) << 4) + e * 2);
  }
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int TI, int TJ, bool USE_TR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
  constexpr int WI = 2, WJ = 2;
  constexpr int RBP = TI * 2, RBQ = TJ * 2;             // tile row bytes
  constexpr int CPRP = TI / 8, CPRQ = TJ / 8;           // chunks per row
  constexpr int PI = 64 * CPRP / 256, QI = 64 * CPRQ / 256;   // chunks per thread = CONSECUTIVE rows per thread
  constexpr int FI = TI / WI / 16, FJ = TJ / WJ / 16;
  constexpr int P_BYTES = 64 * RBP, Q_BYTES = 64 * RBQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sP = smem;
  unsigned char* sQ = smem + 2 * P_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;
  // 1-D grid over (split, tile), split-major, XCD-remapped: each XCD owns whole K-splits, i.e. 1/8 of the pixel rows for
  // ALL output tiles -> the dy / x rows it streams (a few MB) stay in its private L2 while ~36 tiles re-read them
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / p.ntiles, tile = lid - split * p.ntiles;
  const int bj = tile % p.nbj, bi = tile / p.nbj;
  const int i0 = bi * TI, j0 = bj * TJ;
  const int kt0 = split * p.ksteps_per_split;
  const int kt1 = min(kt0 + p.ksteps_per_split, p.ksteps_total);

  // thread -> (chunk column, PI/QI consecutive rows)
  const int pc = tid % CPRP, prow0 = (tid / CPRP) * PI;
  const int qc = tid % CPRQ, qrow0 = (tid / CPRQ) * QI;
  // filter tap of THIS THREAD's column chunk (a 128-wide j tile spans two taps when Cin == 64)
  int r = 0, s = 0, cj0 = j0 + qc * 8;
  if (p.mode == 1) {
    const int col = min(j0 + qc * 8, p.NJ - 8);
    const int tap = col / p.C;
    cj0 = col - tap * p.C;
    r = tap / p.S;
    s = tap - r * p.S;
  }
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.P, p.p_bytes), rsQ = make_rsrc(p.Q, p.q_bytes);
  uint4 rp[PI], rq[QI];
  int pst[PI], qst[QI];                                  // LDS store offsets (loop invariant)
#pragma unroll
  for (int i = 0; i < PI; ++i) pst[i] = (prow0 + i) * RBP + ((pc ^ tn_swz<RBP>(prow0 + i)) << 4);
#pragma unroll
  for (int i = 0; i < QI; ++i) qst[i] = (qrow0 + i) * RBQ + ((qc ^ tn_swz<RBQ>(qrow0 + i)) << 4);
  const bool pcol_ok = i0 + pc * 8 < p.NI;
  unsigned poff = ((unsigned)(kt0 * 64 + prow0) * (unsigned)p.ldp + (unsigned)(i0 + pc * 8)) * 2u;   // row prow0 of step kt0
  const unsigned pstep = 64u * (unsigned)p.ldp * 2u, prow_b = (unsigned)p.ldp * 2u;
  // gather state of this thread's first Q row: pixel (img, ho, wo) of m = kt*64 + qrow0, advanced by 64 pixels per step
  int q_img = 0, q_ho = 0, q_wo = 0;
  const int dW = 64 % max(p.Wo, 1), dH = 64 / max(p.Wo, 1);
  unsigned qoff_plain = 0;
  const bool qcol_ok = j0 + qc * 8 < p.NJ;
  if (p.mode == 1) {
    const unsigned m = (unsigned)(kt0 * 64 + qrow0);
    const unsigned img = fdiv(m, p.dHoWo), rem = m - img * p.dHoWo.d;
    const unsigned ho = fdiv(rem, p.dWo);
    q_img = (int)img; q_ho = (int)ho; q_wo = (int)(rem - ho * p.dWo.d);
  } else {
    qoff_plain = ((unsigned)(kt0 * 64 + qrow0) * (unsigned)p.ldq + (unsigned)(j0 + qc * 8)) * 2u;
  }
  const unsigned qstep = 64u * (unsigned)p.ldq * 2u, qrow_b = (unsigned)p.ldq * 2u;
  const unsigned qchan = (unsigned)cj0;

  auto load_tiles = [&](int kt) {
    const int mrow = kt * 64;
#pragma unroll
    for (int i = 0; i < PI; ++i)
      rp[i] = buf_load16(rsP, (pcol_ok && mrow + prow0 + i < p.Kp) ? poff + (unsigned)i * prow_b : p.p_bytes);
    poff += pstep;
    if (p.mode == 1) {
      int img = q_img, ho = q_ho, wo = q_wo;
#pragma unroll
      for (int i = 0; i < QI; ++i) {
        const int hp = ho * p.stride + r - p.pad, wp = wo * p.stride + s - p.pad;
        const bool ok = qcol_ok && mrow + qrow0 + i < p.Kp && (unsigned)hp < (unsigned)p.H && (unsigned)wp < (unsigned)p.W;
        const unsigned off = (((unsigned)(img * p.H + hp) * (unsigned)p.W + (unsigned)wp) * (unsigned)p.C + qchan) * 2u;
        rq[i] = buf_load16(rsQ, ok ? off : p.q_bytes);
        if (++wo == p.Wo) {                      // next consecutive output pixel
          wo = 0;
          if (++ho == p.Ho) { ho = 0; ++img; }
        }
      }
      // advance the first row by 64 pixels
      q_wo += dW;
      int carry = q_wo >= p.Wo ? 1 : 0;
      q_wo -= carry * p.Wo;
      q_ho += dH + carry;
      const unsigned t = fdiv((unsigned)q_ho, p.dHo);
      q_ho -= (int)t * p.Ho;
      q_img += (int)t;
    } else {
#pragma unroll
      for (int i = 0; i < QI; ++i)
        rq[i] = buf_load16(rsQ, (qcol_ok && mrow + qrow0 + i < p.Kp) ? qoff_plain + (unsigned)i * qrow_b : p.q_bytes);
      qoff_plain += qstep;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PI; ++i) *reinterpret_cast<uint4*>(sP + buf * P_BYTES + pst[i]) = rp[i];
#pragma unroll
    for (int i = 0; i < QI; ++i) *reinterpret_cast<uint4*>(sQ + buf * Q_BYTES + qst[i]) = rq[i];
  };

  int foq[FJ], fop[FI];
#pragma unroll
  for (int tj = 0; tj < FJ; ++tj) foq[tj] = tn_frag_off<RBQ>(wj * (TJ / WJ) + tj * 16, lane);
#pragma unroll
  for (int ti = 0; ti < FI; ++ti) fop[ti] = tn_frag_off<RBP>(wi * (TI / WI) + ti * 16, lane);

  f32x4_t acc[FJ][FI];
#pragma unroll
  for (int a = 0; a < FJ; ++a)
#pragma unroll
    for (int b = 0; b < FI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) load_tiles(kt + 1);
    const unsigned char* cP = sP + buf * P_BYTES;
    const unsigned char* cQ = sQ + buf * Q_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fq[FJ], fp[FI];
#pragma unroll
      for (int tj = 0; tj < FJ; ++tj)
        fq[tj] = USE_TR ? tn_frag_tr<RBQ>(cQ, foq[tj], ks) : tn_frag_scalar<RBQ>(cQ, ks, wj * (TJ / WJ) + tj * 16, lane);
#pragma unroll
      for (int ti = 0; ti < FI; ++ti)
        fp[ti] = USE_TR ? tn_frag_tr<RBP>(cP, fop[ti], ks) : tn_frag_scalar<RBP>(cP, ks, wi * (TI / WI) + ti * 16, lane);
#pragma unroll
      for (int tj = 0; tj < FJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < FI; ++ti) acc[tj][ti] = MFMA16(fq[tj], fp[ti], acc[tj][ti]);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  }
  // D[row = j][col = i]: j = j0 + wj*(TJ/WJ) + tj*16 + (lane>>4)*4 + reg ; i = i0 + wi*(TI/WI) + ti*16 + (lane&15)
  float* slab = p.out + (size_t)split * p.NI * p.NJ;
#pragma unroll
  for (int tj = 0; tj < FJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < FI; ++ti) {
      const int i = i0 + wi * (TI / WI) + ti * 16 + (lane & 15);
      const int j = j0 + wj * (TJ / WJ) + tj * 16 + (lane >> 4) * 4;
      if (i < p.NI && j < p.NJ) *reinterpret_cast<float4*>(slab + (size_t)i * p.NJ + j) =
          make_float4(acc[tj][ti][0], acc[tj][ti][1], acc[tj][ti][2], acc[tj][ti][3]);
    }
}

void gemm_tn_tiles(int NI, int NJ, int C, int* TI, int* TJ) {
  *TI = (NI <= 64) ? 64 : 128;
  (void)C;                                   // tiles may straddle taps: the tap is decoded per thread column chunk
  *TJ = (NJ <= 64) ? 64 : 128;
}

int g_tn_target_blocks = 416;   // option "tn_target_blocks"

int gemm_tn_pick_splits(int Kp, int NI, int NJ, int C) {
  int TI, TJ;
  gemm_tn_tiles(NI, NJ, C, &TI, &TJ);
  const int tiles = ceil_div(NI, TI) * ceil_div(NJ, TJ);
  const int ksteps = ceil_div(Kp, 64);
  int splits = ceil_div(g_tn_target_blocks, tiles);
  if (splits > ksteps) splits = ksteps;
  // keep >= 4 k-steps per split so the slab write does not dominate
  while (splits > 1 && ksteps / splits < 4) --splits;
  if (splits < 1) splits = 1;
  // no empty trailing split
  const int per = ceil_div(ksteps, splits);
  return ceil_div(ksteps, per);
}

template <int TI, int TJ, bool USE_TR>
static int launch_tn(GemmTN p, int splits, hipStream_t st) {
  const int nbi = ceil_div(p.NI, TI);
  p.nbj = ceil_div(p.NJ, TJ);
  p.ksteps_total = ceil_div(p.Kp, 64);
  p.ksteps_per_split = ceil_div(p.ksteps_total, splits);
  FEDFR_REQUIRE(ceil_div(p.ksteps_total, p.ksteps_per_split) == splits, "gemm_tn: splits=%d leaves an empty split", splits);
  const size_t lds = 2 * (size_t)64 * (TI + TJ) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<TI, TJ, USE_TR>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  p.ntiles = nbi * p.nbj;
  dim3 grid(p.ntiles * splits, 1, 1);
  ProfScope prof(prof_slot(true, TI, TJ), 2.0 * p.NI * p.NJ * (double)p.Kp, st);
  hipLaunchKernelGGL((gemm_tn_kernel<TI, TJ, USE_TR>), grid, dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_tn");
  return FEDFR_OK;
}

int gemm_tn_launch(GemmTN p, int splits, hipStream_t st) {
  FEDFR_REQUIRE(p.P && p.Q && p.out && p.Kp > 0 && p.NI > 0 && p.NJ > 0, "gemm_tn: null/empty operand");
  FEDFR_REQUIRE((p.NI & 7) == 0 && (p.NJ & 7) == 0 && (p.ldp & 7) == 0, "gemm_tn: NI, NJ, ldp must be multiples of 8");
  int TI, TJ;
  {
    const unsigned long long pb = 2ull * (unsigned long long)p.Kp * p.ldp;
    const unsigned long long qb = p.mode == 1 ? 2ull * ((unsigned long long)ceil_div(p.Kp, p.Ho * p.Wo)) * p.H * p.W * p.C
                                              : 2ull * (unsigned long long)p.Kp * p.ldq;
    FEDFR_REQUIRE(pb < (1ull << 32) - 64 && qb < (1ull << 32) - 64, "gemm_tn: operand larger than 4 GiB (32-bit buffer offsets)");
    p.p_bytes = (unsigned)pb;
    p.q_bytes = (unsigned)qb;
  }
  if (p.mode == 1) {
    FEDFR_REQUIRE((p.C & 63) == 0 && p.NJ % p.C == 0, "gemm_tn: gather needs C%%64==0 and NJ=taps*C");
    p.dHoWo = make_fastdiv((unsigned)(p.Ho * p.Wo));
    p.dWo = make_fastdiv((unsigned)p.Wo);
    p.dHo = make_fastdiv((unsigned)p.Ho);
    FEDFR_REQUIRE((long long)p.Kp * (long long)(p.Ho * p.Wo) < (1ll << 40), "gemm_tn: fastdiv range");
    gemm_tn_tiles(p.NI, p.NJ, p.C, &TI, &TJ);
  } else {
    FEDFR_REQUIRE((p.ldq & 7) == 0, "gemm_tn: ldq%%8");
    gemm_tn_tiles(p.NI, p.NJ, 0, &TI, &TJ);
  }
#define TN_CASE(a, b)                                                \
  if (TI == a && TJ == b) {                                          \
    if (p.use_tr) return launch_tn<a, b, true>(p, splits, st);       \
    return launch_tn<a, b, false>(p, splits, st);                    \
  }
  TN_CASE(128, 128)
  TN_CASE(128, 64)
  TN_CASE(64, 128)
  TN_CASE(64, 64)
#undef TN_CASE
  fedfr_set_error("gemm_tn: no tile for TI=%d TJ=%d", TI, TJ);
  return FEDFR_ERR_UNSUPPORTED;
}
