// MFMA implicit-GEMM kernels for gfx950 (see gemm.h).  256 threads = 4 waves, 16x16x32 bf16 MFMA,
// BK = 64, register-staged double-buffered LDS tiles with XOR-swizzled 16-byte chunks
// (conflict-free ds_read_b128 / ds_read_b64_tr_b16), one barrier per K-step, XCD-aware 1-D grid.
#include "gemm_dev.h"
#include "gemm_tn_dev.h"
#include "nt_epilogue.h"

#include <algorithm>
#include <vector>


// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----------
// slots: 0..3 = gemm_nt <128,128> <128,64> <64,128> <64,64>; 4..7 = gemm_tn <128,128> <128,64> <64,128> <64,64>;
// 8..11 = (retired: the register-staged LDS-halo 3x3 kernels of rounds 1-3),
// 12 = conv3x3_glds<14,14>, 13 = conv3x3_glds<28,7>, 14 = gemm_tn_glds<128,128>, 15 = conv3x3_glds<56,4> (64 channels), 16 = wgrad9,
// 17 = gemm_nt_glds (all tiles)
namespace {
struct ProfSlot {
  std::vector<hipEvent_t> ev;   // start/stop pairs
  double flops = 0.0, bytes = 0.0;
  long long launches = 0;
};
bool g_prof_on = false;
ProfSlot g_prof[32];   // 0..17: MFMA GEMM kernels (slot = flops); 20..27: HBM-bound kernels (slot 'flops' = algorithmic BYTES)
inline int prof_slot(bool tn, int a, int b) { return (tn ? 4 : 0) + (a == 128 ? 0 : 2) + (b == 128 ? 0 : 1); }
}  // namespace

ProfScope::ProfScope(int slot, double flops, hipStream_t st, double bytes) : st_(st) {
  if (!g_prof_on) return;
  ProfSlot* s = &g_prof[slot];
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  s->ev.push_back(a);
  s->ev.push_back(b);
  s->flops += flops;
  s->bytes += bytes;
  s->launches += 1;
  (void)hipEventRecord(a, st);
  slot_ = s;
}
ProfScope::~ProfScope() {
  if (slot_) (void)hipEventRecord(static_cast<ProfSlot*>(slot_)->ev.back(), st_);
}

void gemm_profile_enable(int on) {
  for (auto& s : g_prof) {
    for (auto e : s.ev) (void)hipEventDestroy(e);
    s.ev.clear();
    s.flops = s.bytes = 0.0;
    s.launches = 0;
  }
  g_prof_on = on != 0;
}
// caller must have synchronised the stream(s).  Returns 0 and fills totals for `slot`.
int gemm_profile_read(int slot, double* total_ms, long long* launches, double* flops) {
  if (slot < 0 || slot >= 32) return -1;
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



int gemm_profile_read_bytes(int slot, double* bytes) {
  if (slot < 0 || slot >= 32 || !bytes) return -1;
  *bytes = g_prof[slot].bytes;
  return 0;
}
// algorithmic HBM bytes of an NT / TN launch: every operand read once, the output written once (16-bit activations, fp32 slabs)
static double nt_alg_bytes(const GemmNT& p, int splits) {
  const double images = p.mode == 1 ? (double)ceil_div(p.M, p.Ho * p.Wo) : 0.0;
  // (a 1x1 / stride-2 conv touches only the pixels it keeps)
  const double a = p.mode == 1 ? 2.0 * images * (p.S == 1 && p.up == 1 ? (double)p.Ho * p.Wo : (double)p.H * p.W) * p.C : 2.0 * (double)p.M * p.K;
  (void)splits;                                          // (split-K slabs are overhead, not algorithmic: the output counts once)
  const double out = p.Cb ? 2.0 * (double)p.M * p.N * (p.par_on == 2 ? 4.0 : 1.0) : 4.0 * (double)p.M * p.N;
  return a + 2.0 * (double)p.N * p.K + out;
}
double gemm_nt_alg_bytes(const GemmNT& p, int splits) { return nt_alg_bytes(p, splits); }
double gemm_tn_alg_bytes(const GemmTN& p, int splits) {
  const double q = p.mode == 1 ? 2.0 * (double)ceil_div(p.Kp, p.Ho * p.Wo) * (p.S == 1 ? (double)p.Ho * p.Wo : (double)p.H * p.W) * p.C : 2.0 * (double)p.Kp * p.NJ;
  (void)splits;
  return 2.0 * (double)p.Kp * p.NI + q + 4.0 * (double)p.NI * p.NJ;
}

// =====================================================================================================
// NT kernel
// =====================================================================================================
template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p_) {
  GemmNT p = p_;
  if (p.par_on == 2) {                                  // all four output-parity classes of a stride-2 dgrad in one launch: class = blockIdx.z,
    const int cls = 3 - (int)blockIdx.z;                // the four-tap class first (its workgroups run 4x as long as the one-tap class's)
    p.par_h = cls >> 1; p.par_w = cls & 1;
    p.ksteps_total = (1 + p.par_h) * (1 + p.par_w) * p.cpt;
    p.ksteps_per_split = p.ksteps_total;
  }
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
      int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      if (p.par_on) {                                   // class row (img, h2, w2) -> output pixel (2 h2 + par_h, 2 w2 + par_w)
        ho = 2 * ho + p.par_h;
        wo = 2 * wo + p.par_w;
      }
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
  const int s0 = 1 - p.par_w, tstep = p.par_on ? 2 : 1;   // parity class: taps r = 1 - par_h (+2), s = 1 - par_w (+2) only
  if (p.par_on) {
    const int nS = 1 + p.par_w;
    r = (1 - p.par_h) + 2 * (tap / nS);
    s = s0 + 2 * (tap % nS);
  }

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
      const int k = (p.mode == 1 ? ((r * p.S + s) * p.cpt + cc) * 64 : kt * 64) + ch * 8;   // == kt * 64 unless taps are skipped
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
      s += tstep;
      if (s >= p.S) {
        s = p.par_on ? s0 : 0;
        r += tstep;
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

  nt_epilogue<BM, BN, WM, WN, 256>(p, acc, smem, bm, m0, n0, split, wm, wn, tid, lane);   // all waves are past the last barrier of the K loop
}

// (the single-stage variant of the NT kernel, option "nt_nbuf" = 1, lost every sweep of rounds 2-5 and was removed in round 6: two LDS stages, one
// barrier per K-step)

template <int BM, int BN, int WM, int WN, int NBUF>
static int launch_nt_impl(const GemmNT& p0, int splits, hipStream_t st);

template <int BM, int BN, int WM, int WN>
static int launch_nt(const GemmNT& p0, int splits, hipStream_t st) {
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
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, WM, WN, NBUF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  dim3 grid(nbm * p.nbn, splits, p.par_on == 2 ? 4 : 1);
  ProfScope prof(prof_slot(false, BM, BN), 2.0 * p.M * p.N * (p.par_on == 2 ? 64.0 * 9 * p.cpt : p.par_on ? 64.0 * p.ksteps_total : (double)p.K), st,
                 nt_alg_bytes(p, splits));
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, NBUF>), grid, dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_nt");
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

// rows of gemm_nt_stat_rows that carry data for a conv with BatchNorm statistics in its epilogue (the rest are zero rows the kernel
// writes so that a finalize over the 128-pixel-tile row count stays right): the 196-pixel-tile LDS-DMA kernels leave 2 per tile
bool gemm_nt_conv_epilogue_ok(int W, int C, int N, int M, int ksize, int stride);
int g_conv28_tpw2 = 2;   // option "conv28_tpw2": 28x28 convs run two image tiles per workgroup -- 1: the forward launches (one BatchNorm partial row per workgroup: 256 instead of 1024), 2: the dgrad launches too (256 workgroups that stay instead of 512 that are dispatched in two rounds between the weight-gradient workgroups: 17.72 -> 17.54 ms/step same-box)
static bool glds28_two_tiles_shape(int M) { return g_conv28_tpw2 && (M / 196) % 2 == 0; }
// the 28x28 dgrad with the BatchNorm-backward reduction in its epilogue runs two tiles per workgroup as well (one partial row each)
bool gemm_nt_fused28_two_tiles(int M) { return g_conv28_tpw2 >= 2 && glds28_two_tiles_shape(M); }
static bool glds28_two_tiles(const GemmNT& p) {          // option value 2: the dgrad launches (no statistics) too
  return (p.stats || g_conv28_tpw2 >= 2) && p.W == 28 && !p.esc && !p.eadd && !p.Cb2 && glds28_two_tiles_shape(p.M);
}
int gemm_nt_stat_rows_live(int M, int N, int C, int W, int ksize, int stride) {
  if ((W == 14 || W == 28) && gemm_nt_conv_epilogue_ok(W, C, N, M, ksize, stride))
    return (W == 28 && glds28_two_tiles_shape(M)) ? M / 196 / 2 : M / 196 * 2;
  // the persistent 64-channel kernel: two rows per workgroup, at most one workgroup per CU (conv_c64p.hip: launch_c64p) — 512 rows instead
  // of 6 272 / 25 088, few enough for the finalize kernel to take without the staging launch in front of it
  if (g_conv_c64p && ksize == 3 && stride == 1 && C == 64 && N == 64 && (W == 56 || W == 112) && M % (W * W) == 0) return 2 * conv_c64p_grid(M);
  return gemm_nt_stat_rows(M, N);
}

int g_dgrad_parity = 2;   // option "dgrad_parity": stride-2 3x3 dgrad as 4 output-parity classes (9/4 instead of 9 taps per output pixel); 2: the four classes in one launch

// shapes whose conv runs on a plain LDS-DMA kernel instantiation (mirrors the dispatch in gemm_nt_launch_one): those implement the
// eval-mode output epilogue (GemmNT::esc / eadd / Cb2)
bool gemm_nt_conv_epilogue_ok(int W, int C, int N, int M, int ksize, int stride) {
  if (ksize != 3 || stride != 1 || W <= 0 || M % (W * W) != 0 || nt_bm(M, N) != 128) return false;
  if (W == 112) return C == 64 && N == 64;
  if (W == 56) return (C == 64 && (N == 64 || N == 128)) || (C == 128 && N == 64);
  if (W == 14 || W == 28) return N % 128 == 0 && C % 128 == 0;
  return false;
}

static int gemm_nt_launch_one(GemmNT p, int splits, hipStream_t st);
int gemm_nt_launch(GemmNT p, int splits, hipStream_t st) {
  if (g_dgrad_parity && p.mode == 1 && p.up == 2 && p.S == 3 && p.pad == 1 && p.stride == 1 && p.Cb && !p.stats && splits == 1 &&
      !(p.Ho & 1) && !(p.Wo & 1) && p.Ho == 2 * p.H && p.Wo == 2 * p.W && p.M % (p.Ho * p.Wo) == 0 && !p.par_on) {
    if (g_dgrad_parity >= 2) {                          // one launch, class = blockIdx.z
      GemmNT q = p;
      q.par_on = 2; q.par_h = 1; q.par_w = 1;           // (the largest class sizes the checks; the kernel sets its own)
      q.outH = p.Ho; q.outW = p.Wo;
      q.Ho = p.Ho / 2; q.Wo = p.Wo / 2; q.M = p.M / 4;
      return gemm_nt_launch_one(q, 1, st);
    }
    for (int cls = 0; cls < 4; ++cls) {
      GemmNT q = p;
      q.par_on = 1; q.par_h = cls >> 1; q.par_w = cls & 1;
      q.outH = p.Ho; q.outW = p.Wo;
      q.Ho = p.Ho / 2; q.Wo = p.Wo / 2; q.M = p.M / 4;
      FEDFR_TRY(gemm_nt_launch_one(q, 1, st));
    }
    return FEDFR_OK;
  }
  return gemm_nt_launch_one(p, splits, st);
}
static int gemm_nt_launch_one(GemmNT p, int splits, hipStream_t st) {
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
  p.ksteps_total = p.par_on ? (1 + p.par_h) * (1 + p.par_w) * p.cpt : ceil_div(p.K, 64);
  {
    const unsigned long long ab = p.mode == 1 ? 2ull * ((unsigned long long)ceil_div(p.M, p.Ho * p.Wo)) * p.H * p.W * p.C
                                              : 2ull * (unsigned long long)p.M * p.lda;
    const unsigned long long bb = 2ull * (unsigned long long)p.N * p.K;
    FEDFR_REQUIRE(ab < (1ull << 32) - 64 && bb < (1ull << 32) - 64, "gemm_nt: operand larger than 4 GiB (32-bit buffer offsets)");
    p.a_bytes = (unsigned)ab;
    p.b_bytes = (unsigned)bb;
  }
  if (p.esc || p.eadd || p.Cb2)
    FEDFR_REQUIRE(p.mode == 1 && p.up == 1 && p.pad == 1 && p.H == p.W && p.H == p.Ho && p.W == p.Wo && p.Cb && splits == 1 && !p.bpart &&
                  !p.stats && !p.par_on && gemm_nt_conv_epilogue_ok(p.W, p.C, p.N, p.M, p.S, p.stride), "gemm_nt: output epilogue is not available for this convolution");
  if (splits == 1 && conv_c64p_applies(p)) return launch_conv_c64p(p, st);
  const int BM = nt_bm(p.M, p.N);
  // 3x3 / stride-1 / pad-1 layers on the LDS-DMA kernels (conv_glds_impl.h: one translation unit per instantiation); every other shape — odd map
  // sizes, channel counts that are not multiples of 128 on the 14x14 / 28x28 maps — takes the generic gather GEMM below
  if (BM == 128 && p.mode == 1 && p.S == 3 && p.K == 9 * p.C && p.stride == 1 && p.pad == 1 && p.up == 1 &&
      p.H == p.Ho && p.W == p.Wo && p.Cb && splits == 1 && p.W <= 126) {
    if (p.H == 112 && p.W == 112 && !p.bpart && p.M % (112 * 112) == 0 && p.C == 64 && p.N == 64) return launch_conv_glds_w112(p, st);
    if (p.H == 56 && p.W == 56 && !p.bpart && p.M % (56 * 56) == 0) {
      if (p.C == 64 && p.N == 64) return launch_conv_glds_w56(p, st);
      if (p.C == 64 && p.N == 128) return launch_conv_glds_w56_c64_n128(p, st);
      if (p.C == 128 && p.N == 64) return launch_conv_glds_w56_c128_n64(p, st);
    }
    if (p.H == p.W && (p.W == 14 || p.W == 28) && p.N % 128 == 0 && p.C % 128 == 0 && p.M % (p.H * p.W) == 0) {
      if (p.bpart && p.ldc == p.N && p.bmom == 2)         // sphnet: PReLU-apply epilogue
        return p.W == 14 ? launch_conv_glds8_fused_w14_papply(p, st)
                         : (gemm_nt_fused28_two_tiles(p.M) && !p.stats ? launch_conv_glds8_fused_w28s_papply(p, st) : launch_conv_glds8_fused_w28_papply(p, st));
      if (p.bpart && p.ldc == p.N)
        return p.W == 14 ? launch_conv_glds8_fused_w14(p, st)
                         : (gemm_nt_fused28_two_tiles(p.M) && !p.stats ? launch_conv_glds8_fused_w28s(p, st) : launch_conv_glds8_fused_w28(p, st));
      if (!p.bpart)
        return p.W == 14 ? launch_conv_glds8_w14(p, st) : (glds28_two_tiles(p) ? launch_conv_glds8_w28_stats(p, st) : launch_conv_glds8_w28(p, st));
    }
  }
  if (gemm_nt_glds_applies(p, BM, splits)) return launch_nt_glds(p, BM, splits, 17, st);
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

static const int g_tn_target_blocks = 416;   // workgroups a register-staged weight-gradient GEMM aims for (256 / 832 measured neutral in rounds 4-5: no switch)
int g_tn_glds = 2;              // option "tn_glds": LDS-DMA wgrad kernel for 128-multiple conv shapes: 2 = 8 waves (two per SIMD: one wave's VALU / DMA issue hides
                                // behind the other's MFMAs, +20 % over 1 = 4 waves), 0 = register-staged kernel

// K-split count of a weight-gradient problem given WHICH kernel family serves it (use_w9: the nine-tap kernel, use_glds: the LDS-DMA ring GEMM)
static int tn_pick_splits_for(int Kp, int NI, int NJ, int C, int Wo, bool use_w9, bool use_glds) {
  if (use_w9) return wgrad9_pick_splits(Kp, NI, NJ, Wo);
  if (use_glds) return gemm_tn_glds_pick_splits(Kp, NI, NJ);
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
int gemm_tn_pick_splits(int Kp, int NI, int NJ, int C, int Wo, int stride) {
  return tn_pick_splits_for(Kp, NI, NJ, C, Wo, Wo > 0 && wgrad9_applies_shape(Kp, NI, NJ, C, Wo, stride), C > 0 && gemm_tn_glds_applies(NI, NJ, C, 1));
}

// slab count to size a workspace for: the largest any kernel choice (options can be toggled after a plan was created) would use.  Round 6: computed
// from the shape predicates alone — the earlier version toggled the process-global switches in a loop and restored them, which a second host thread
// (Server.train with parallel_clients, thread-ranks) could observe mid-flight: "wgrad9_pair: unsupported problem pair" out of a concurrent backward pass
int gemm_tn_max_splits(int Kp, int NI, int NJ, int C, int Wo, int stride) {
  const bool w9 = Wo > 0 && wgrad9_shape_ok(Kp, NI, NJ, C, Wo, stride), gl = C > 0 && gemm_tn_glds_shape_ok(NI, NJ, C, 1);
  int m = tn_pick_splits_for(Kp, NI, NJ, C, Wo, false, false);
  if (gl) m = std::max(m, tn_pick_splits_for(Kp, NI, NJ, C, Wo, false, true));
  if (w9) m = std::max(m, tn_pick_splits_for(Kp, NI, NJ, C, Wo, true, false));
  return m;
}

template <int TI, int TJ, bool USE_TR>
static int launch_tn(GemmTN p, int splits, hipStream_t st) {
  const int nbi = ceil_div(p.NI, TI);
  p.nbj = ceil_div(p.NJ, TJ);
  p.ksteps_total = ceil_div(p.Kp, 64);
  p.ksteps_per_split = ceil_div(p.ksteps_total, splits);
  FEDFR_REQUIRE(ceil_div(p.ksteps_total, p.ksteps_per_split) == splits, "gemm_tn: splits=%d leaves an empty split", splits);
  const size_t lds = 2 * (size_t)64 * (TI + TJ) * 2;
  static PerDeviceOnce attr_once;     // hipFuncSetAttribute is per device (a Server process may drive several)
  attr_once.run([&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<TI, TJ, USE_TR>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  p.ntiles = nbi * p.nbj;
  dim3 grid(p.ntiles * splits, 1, 1);
  ProfScope prof(prof_slot(true, TI, TJ), 2.0 * p.NI * p.NJ * (double)p.Kp, st, gemm_tn_alg_bytes(p, splits));
  hipLaunchKernelGGL((gemm_tn_kernel<TI, TJ, USE_TR>), grid, dim3(256), lds, st, p);
  FEDFR_LAUNCH_CHECK("gemm_tn");
  return FEDFR_OK;
}

// operand ranges and conv-gather dividers shared by the single and the paired launch
static int tn_prepare(GemmTN& p) {
  FEDFR_REQUIRE(p.P && p.Q && p.out && p.Kp > 0 && p.NI > 0 && p.NJ > 0, "gemm_tn: null/empty operand");
  FEDFR_REQUIRE((p.NI & 7) == 0 && (p.NJ & 7) == 0 && (p.ldp & 7) == 0, "gemm_tn: NI, NJ, ldp must be multiples of 8");
  const unsigned long long pb = 2ull * (unsigned long long)p.Kp * p.ldp;
  const unsigned long long qb = p.mode == 1 ? 2ull * ((unsigned long long)ceil_div(p.Kp, p.Ho * p.Wo)) * p.H * p.W * p.C
                                            : 2ull * (unsigned long long)p.Kp * p.ldq;
  FEDFR_REQUIRE(pb < (1ull << 32) - 64 && qb < (1ull << 32) - 64, "gemm_tn: operand larger than 4 GiB (32-bit buffer offsets)");
  p.p_bytes = (unsigned)pb;
  p.q_bytes = (unsigned)qb;
  if (p.mode == 1) {
    FEDFR_REQUIRE((p.C & 63) == 0 && p.NJ % p.C == 0, "gemm_tn: gather needs C%%64==0 and NJ=taps*C");
    p.dHoWo = make_fastdiv((unsigned)(p.Ho * p.Wo));
    p.dWo = make_fastdiv((unsigned)p.Wo);
    p.dHo = make_fastdiv((unsigned)p.Ho);
    FEDFR_REQUIRE((long long)p.Kp * (long long)(p.Ho * p.Wo) < (1ll << 40), "gemm_tn: fastdiv range");
  } else {
    FEDFR_REQUIRE((p.ldq & 7) == 0, "gemm_tn: ldq%%8");
  }
  return FEDFR_OK;
}

// the two 3x3 / stride-1 weight gradients of a residual block on the paired nine-tap kernel (wgrad9p.hip); *splits = slabs per layer
bool gemm_tn_w9pair_ok(const GemmTN& a, const GemmTN& b) { return wgrad9p_applies(a, b); }
int gemm_tn_w9pair_splits(const GemmTN& a) { return wgrad9p_pick_splits(a.Kp, a.NI, a.NJ, a.W); }
bool gemm_tn_w9pair_job_ok(const GemmTN& a, int splits, const W9PJob& job) { return wgrad9p_job_ok(a, splits, job); }
int gemm_tn_launch_w9pair(GemmTN a, GemmTN b, int splits, hipStream_t st, const W9PJob* job) {
  FEDFR_TRY(tn_prepare(a));
  FEDFR_TRY(tn_prepare(b));
  return launch_wgrad9_pair(a, b, splits, st, job);
}

#ifndef TN_ABLATE
#define TN_ABLATE 0     // timing experiments only (WRONG results): 1 = no weight gradient of the 64 -> 64 stride-2 3x3 layer (112 -> 56)
#endif
int gemm_tn_launch(GemmTN p, int splits, hipStream_t st) {
  FEDFR_TRY(tn_prepare(p));
  if ((TN_ABLATE & 1) && p.mode == 1 && p.stride == 2 && p.C == 64 && p.S == 3) return FEDFR_OK;
  int TI, TJ;
  if (p.mode == 1) {
    if (wgrad9_applies(p)) return launch_wgrad9(p, splits, st);
    if (p.use_tr && gemm_tn_glds_applies(p.NI, p.NJ, p.C, 1)) return launch_tn_glds(p, splits, st);
    gemm_tn_tiles(p.NI, p.NJ, p.C, &TI, &TJ);
  } else {
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
