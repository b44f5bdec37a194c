// gemm_f32.hip -- fp32 GEMM family on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chain,
// 157 TFLOP/s peak = 1/16 of bf16 MFMA; there is no TF32/xf32 on CDNA4).
//
// One kernel template covers every dense contraction of the SM3Det backbone hot path
// (reference: mmrotate/models/backbones/convnext_moe.py FFN.forward :397-405, the expert loop :244,
// CosineTopKGate projection :101, stem / downsample convs as patch GEMMs :533-558,783-791) and their backward:
//
//   MODE_NT : C[M,N] = A[M,K] . B[N,K]^T        (nn.Linear forward: x @ W^T)
//   MODE_NN : C[M,N] = A[M,K] . B[K,N]          (dgrad: dY @ W)
//   MODE_TN : C[M,N] = A[Kt,M]^T . B[Kt,N]      (wgrad: dY^T @ X, split-K over token rows, partials to workspace)
//
// Grouped (MoE experts): rows of A/C (NT, NN) or the reduction rows (TN) are partitioned into `num_groups`
// contiguous segments by a DEVICE prefix array `offsets[G+1]` (expert-major slot order); group g uses weight block
// g.  The tile->group map is computed in the kernel from `offsets`, so ragged expert loads never sync the host
// (the reference does `.cpu()` per MoE block: convnext_moe.py:259).
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128x128x32, wave tile 64x64 = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs), operands staged k-major in LDS (conflict-free ds_read_b32: lanes 0-31 read 32 consecutive
// floats of row k, lanes 32-63 of row k+1 -- exactly the A[i][k]/B[k][j] fragment of the 32x32x2 instruction),
// register-prefetched double-buffered LDS (one barrier per k-step), XCD-aware block remap so the N-tiles that
// share an A row-panel land on one XCD's L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MODE_NT = 0, MODE_NN = 1, MODE_TN = 2;
constexpr int BM = 128, BN = 128;
constexpr int BK_MAX = 32;
// leading dim of a tile written transposed (k-contiguous source), chosen so the 4-byte scatter writes of one
// half-wave hit 32 distinct banks: BK=32 -> 8 k-quads x 4 rows need LD = 1 (mod 8); BK=16 -> 4 k-quads x 8 rows need 2.
template <int BK> struct LdT { static constexpr int v = (BK == 32) ? BM + 1 : BM + 2; };
constexpr int LD_D = BM + 4;  // leading dim of a tile written directly (k-major source): 16-B aligned rows
constexpr int NTHREADS = 256;

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;  // TN: M,N = output dims, K = total reduction rows (ignored when grouped: offsets decide)
  int lda, ldb, ldc;
  // grouping
  const int32_t* offsets;  // device, num_groups+1 (NULL => one group covering all rows)
  int num_groups;
  long strideB;     // elements between consecutive groups' B blocks (NT/NN)
  long strideBias;  // elements between groups' bias vectors
  long strideC;     // TN: elements between groups' outputs in the partial workspace (= M*N)
  int splits;       // TN: split-K factor per group
  // epilogue operands
  const float* bias;      // [N] (per group)
  const float* aux_in;    // EPI_GELU_BWD: gelu'(h)[M,N];  EPI_BIAS_SCALE_RES: residual[M,N]
  float* aux_out;         // EPI_BIAS_GELU: gelu'(h)[M,N]; EPI_BIAS_SCALE_RES: y[M,N]
  float* colpart;         // EPI_GELU_BWD: per-row-tile column sums [row tiles][N] (bias gradient partials) or NULL
  const float* gamma;     // [N] layer scale
  const float* rowscale;  // [M / rows_per_scale] (stochastic-depth keep/keep_prob per image) or NULL
  int rows_per_scale;
  int ld_aux;
  // GATHER (implicit-GEMM 3x3 convolution over NHWC tokens, padding 1): the gathered operand has `cC` channels per tap,
  // its rows live on an (sH, sW) grid per image; the GEMM's own rows (NT/NN: A/C rows, TN: reduction rows) live on an
  // (rH, rW) grid.  cT = 0: source = row * cS + d - 1 (forward / weight gradient);  cT = 1: source = (row - d + 1) / cS
  // where divisible (input gradient = transposed convolution).  cInv: (kt * cInv) >> 16 == kt / (cC / BK).
  int cC, sH, sW, rH, rW, cS, cT;
  int kTilesPerSplit;  // GATHER NT/NN: > 0 -> blockIdx.z owns k-tiles [z*kTilesPerSplit, ...) and writes slab z of C
  unsigned cInv, mRW, mRH;  // mRW/mRH: ceil(2^32 / rW), ceil(2^32 / rH) for exact umulhi division of row indices
};

// exact n / d for n * d < 2^32 with m = floor(2^32 / d) + 1 (d >= 2); d == 1 passes through
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned m) { return d == 1 ? n : __umulhi(n, m); }

// source coordinate of one axis for tap offset dd in {0,1,2}; returns -1 when the tap falls outside / between pixels
__device__ __forceinline__ int gather_coord(int r, int dd, int stride, int transposed, int lim) {
  if (!transposed) {
    const int v = r * stride + dd - 1;
    return (v >= 0 && v < lim) ? v : -1;
  }
  const int t = r - dd + 1;  // stride is 1 or 2 (checked on the host)
  const int q = stride == 2 ? (t >> 1) : t;
  const bool ok = t >= 0 && (stride == 1 || (t & 1) == 0) && q < lim;
  return ok ? q : -1;
}

// GELU(erf) and its derivative share one exponential: y = h*Phi(h), y' = Phi(h) + h*phi(h).  The forward epilogue stores
// y' so the backward epilogue is a single multiply (no transcendental on the dgrad critical path).
// Phi through erfc(u) = t*(a1 + t*(a2 + ... a5 t))*exp(-u^2), t = 1/(1 + p u), u = |h|/sqrt(2) (Abramowitz & Stegun
// 7.1.26, |error| <= 1.5e-7 on erf, i.e. <= 7.5e-8 on Phi -- fp32 rounding level): ~16 VALU ops per element instead of
// ~55 for ocml's erff + expf; the epilogue was costing 35-120 us per launch on the 25-50 M-element expert/FFN outputs.
__device__ __forceinline__ void gelu_erf_both(float h, float& y, float& dy) {
  const float e = __expf(-0.5f * h * h);  // exp(-u^2)
  const float u = fabsf(h) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float half_erfc = 0.5f * q * t * e;  // 0.5 * erfc(u) = Phi(-|h|)
  const float cdf = h >= 0.f ? 1.0f - half_erfc : half_erfc;
  const float pdf = 0.39894228040143267794f * e;
  y = h * cdf;
  dy = fmaf(h, pdf, cdf);
}

// EPI codes (must match include/sm3det_hip.h)
constexpr int EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_SCALE_RES = 3, EPI_GELU_BWD = 4;
constexpr int EPI_BIAS_RELU = 5;  // conv3x3 + bias + F.relu (RPN tower); only instantiated for the gather variant

template <int MODE, int EPI, int BK, int GATHER = 0>
__global__ __launch_bounds__(NTHREADS, (BK == 32 ? 2 : 4)) void gemm_f32_kernel(GemmParams p) {
  constexpr int LD_T = LdT<BK>::v;
  constexpr int NLD = BK / 8;  // global->LDS passes per operand tile
  // both operand regions are sized for the wider leading dimension (LD_D); 2 stages each:
  // BK=32: 67.6 KB -> 2 blocks/CU;  BK=16: 33.8 KB -> 4 blocks/CU (more prologue/epilogue overlap for short-K GEMMs)
  __shared__ __attribute__((aligned(16))) float smem[4 * BK * LD_D];
  // A tile uses LD_T when its source is k-contiguous (NT, NN), LD_D when k-major (TN).
  // B tile uses LD_T when its source is k-contiguous (NT),     LD_D when k-major (NN, TN).
  constexpr bool A_TRANS = (MODE != MODE_TN);
  constexpr bool B_TRANS = (MODE == MODE_NT);
  constexpr int LDA_S = A_TRANS ? LD_T : LD_D;
  constexpr int LDB_S = B_TRANS ? LD_T : LD_D;
  float* As = smem;                  // [2][BK][LDA_S]
  float* Bs = smem + 2 * BK * LD_D;  // [2][BK][LDB_S]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64;
  const int wn0 = (wave & 1) * 64;
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // ---- XCD-aware remap of the linear block id (speed only): consecutive logical tiles share an XCD ----------
  const int ntn = (p.N + BN - 1) / BN;
  int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / kNumXCD, r = nblk % kNumXCD;
    const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % ntn;
  int tile_m = bid / ntn;

  // ---- group / row-range resolution -------------------------------------------------------------------------
  int g = 0;
  int row0, row_end;  // NT/NN: rows of A and C handled by this block; TN: reduction rows [row0,row_end)
  int m0;             // TN: first output row of this tile
  if (MODE == MODE_TN) {
    g = blockIdx.z / p.splits;
    const int split = blockIdx.z - g * p.splits;
    int seg0 = 0, seg1 = p.K;
    if (p.offsets) {
      seg0 = p.offsets[g];
      seg1 = p.offsets[g + 1];
    }
    const int cnt = seg1 - seg0;
    int chunk = (cnt + p.splits - 1) / p.splits;
    chunk = (chunk + BK - 1) / BK * BK;
    row0 = seg0 + split * chunk;
    row_end = min(seg1, row0 + chunk);
    m0 = tile_m * BM;
  } else {
    if (p.offsets) {
      int base = 0;
      bool found = false;
      for (int gg = 0; gg < p.num_groups; gg++) {
        const int o0 = p.offsets[gg], o1 = p.offsets[gg + 1];
        const int nt = (o1 - o0 + BM - 1) / BM;
        if (tile_m < base + nt) {
          g = gg;
          row0 = o0 + (tile_m - base) * BM;
          row_end = o1;
          found = true;
          break;
        }
        base += nt;
      }
      if (!found) return;
    } else {
      row0 = tile_m * BM;
      row_end = p.M;
      if (row0 >= row_end) return;
    }
    m0 = row0;
  }
  const int n0 = tile_n * BN;
  const float* __restrict__ Ag = p.A;
  const float* __restrict__ Bg = p.B + (MODE == MODE_TN ? 0 : (long)g * p.strideB);

  // ---- loaders ----------------------------------------------------------------------------------------------
  // transposed loader: source rows are k-contiguous: thread -> (row t_r + T_ROWS i, k quad t_kq)
  const int t_kq = tid % (BK / 4), t_r = tid / (BK / 4);
  constexpr int T_ROWS = NTHREADS / (BK / 4);  // rows covered per pass of the transposed loader
  // direct loader: source is k-major: thread -> (k row d_kk + 8 i, column quad d_nq)
  const int d_nq = tid & 31, d_kk = tid >> 5;
  int nk = (MODE == MODE_TN) ? (max(row_end - row0, 0) + BK - 1) / BK : p.K / BK;
  int kbase = 0;  // first k-tile of this block (split-K over the taps of small implicit-GEMM convolutions)
  if (GATHER && MODE != MODE_TN && p.kTilesPerSplit > 0) {
    kbase = blockIdx.z * p.kTilesPerSplit;
    nk = min(p.kTilesPerSplit, nk - kbase);
  }

  // Per-thread source pointers, computed once.  Out-of-range rows / columns are CLAMPED to a valid address instead
  // of branched around (the garbage they bring only reaches output rows/columns the epilogue masks); the one place
  // where zeros are required -- reduction rows past the segment end in TN -- uses a select after the load.
  const float* pa[NLD];
  const float* pb[NLD];
  int gy[NLD], gx[NLD];  // GATHER (NT/NN): grid coordinates of this thread's A rows
  int tn_tap = 0;        // GATHER (TN): the tap this N-tile belongs to (cC % BN == 0)
#pragma unroll
  for (int i = 0; i < NLD; i++) {
    gy[i] = gx[i] = 0;
    if (MODE == MODE_TN) {
      const int mc = min(m0 + 4 * d_nq, p.M - 4), nc = min(n0 + 4 * d_nq, p.N - 4);
      pa[i] = Ag + mc;
      if (GATHER) {
        tn_tap = n0 / p.cC;
        pb[i] = Bg + (nc - tn_tap * p.cC);  // channel offset inside the tap; the row part is added per load
      } else {
        pb[i] = Bg + nc;
      }
    } else {
      const int r = min(row0 + t_r + T_ROWS * i, row_end - 1);
      if (GATHER) {
        const unsigned t = fast_div((unsigned)r, (unsigned)p.rW, p.mRW);
        gx[i] = r - (int)t * p.rW;
        const unsigned b = fast_div(t, (unsigned)p.rH, p.mRH);
        gy[i] = (int)t - (int)b * p.rH;
        pa[i] = Ag + (long)b * p.sH * p.sW * p.cC + 4 * t_kq;
      } else {
        pa[i] = Ag + (long)r * p.lda + 4 * t_kq;
      }
      if (MODE == MODE_NT) {
        const int n = min(n0 + t_r + T_ROWS * i, p.N - 1);
        pb[i] = Bg + (long)n * p.ldb + 4 * t_kq;
      } else {
        const int nc = min(n0 + 4 * d_nq, p.N - 4);
        pb[i] = Bg + (long)(d_kk + 8 * i) * p.ldb + nc;
      }
    }
  }
  // piece q in [0, 2*NLD): q < NLD -> A piece q, else B piece q - NLD
  auto load_piece = [&](f32x4 (&ra)[NLD], f32x4 (&rb)[NLD], int q, int kt) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_TN) {
      const int i = q < NLD ? q : q - NLD;
      const int kr = row0 + kt * BK + d_kk + 8 * i;
      const int krc = min(kr, row_end - 1);
      if (q < NLD) {
        f32x4 v = *reinterpret_cast<const f32x4*>(pa[i] + (long)krc * p.lda);
        ra[i] = kr < row_end ? v : zero4;
      } else if (GATHER) {
        // reduction row = output position (b, oy, ox); B row = the input pixel this N-tile's tap reads for it
        const unsigned t = fast_div((unsigned)krc, (unsigned)p.rW, p.mRW);
        const int ox = krc - (int)t * p.rW;
        const unsigned b = fast_div(t, (unsigned)p.rH, p.mRH);
        const int oy = (int)t - (int)b * p.rH;
        const int sy = gather_coord(oy, tn_tap / 3, p.cS, p.cT, p.sH);
        const int sx = gather_coord(ox, tn_tap % 3, p.cS, p.cT, p.sW);
        const bool ok = kr < row_end && sy >= 0 && sx >= 0;
        const long off = (((long)b * p.sH + max(sy, 0)) * p.sW + max(sx, 0)) * p.cC;
        f32x4 v = *reinterpret_cast<const f32x4*>(pb[i] + off);
        rb[i] = ok ? v : zero4;
      } else {
        f32x4 v = *reinterpret_cast<const f32x4*>(pb[i] + (long)krc * p.ldb);
        rb[i] = kr < row_end ? v : zero4;
      }
    } else if (q < NLD) {
      if (GATHER) {
        const int kg = kt + kbase;
        const int tap = (int)(((unsigned)kg * p.cInv) >> 16);  // k-tile -> tap (uniform), channel offset inside it
        const int c0 = kg * BK - tap * p.cC;
        const int sy = gather_coord(gy[q], tap / 3, p.cS, p.cT, p.sH);
        const int sx = gather_coord(gx[q], tap % 3, p.cS, p.cT, p.sW);
        const long off = ((long)max(sy, 0) * p.sW + max(sx, 0)) * p.cC + c0;
        f32x4 v = *reinterpret_cast<const f32x4*>(pa[q] + off);
        ra[q] = (sy >= 0 && sx >= 0) ? v : zero4;
      } else {
        ra[q] = *reinterpret_cast<const f32x4*>(pa[q] + kt * BK);
      }
    } else if (MODE == MODE_NT) {
      rb[q - NLD] = *reinterpret_cast<const f32x4*>(pb[q - NLD] + (kt + kbase) * BK);
    } else if (GATHER) {
      // NN gather (input gradient): B row k = (tap, co) lives at W[co][tap][:]  (ldb = 9 * Cin, + tap * N columns)
      const int kg = kt + kbase;
      const int tap = (int)(((unsigned)kg * p.cInv) >> 16);
      const int c0 = kg * BK - tap * p.cC;
      rb[q - NLD] = *reinterpret_cast<const f32x4*>(pb[q - NLD] + (long)c0 * p.ldb + tap * p.N);
    } else {
      rb[q - NLD] = *reinterpret_cast<const f32x4*>(pb[q - NLD] + (long)kt * BK * p.ldb);
    }
  };
  auto store_piece = [&](const f32x4 (&ra)[NLD], const f32x4 (&rb)[NLD], int q, int buf) {
    if (q < NLD) {
      float* a_s = As + buf * BK * LDA_S;
      if (A_TRANS) {
#pragma unroll
        for (int j = 0; j < 4; j++) a_s[(4 * t_kq + j) * LDA_S + t_r + T_ROWS * q] = ra[q][j];
      } else {
        *reinterpret_cast<f32x4*>(a_s + (d_kk + 8 * q) * LDA_S + 4 * d_nq) = ra[q];
      }
    } else {
      const int i = q - NLD;
      float* b_s = Bs + buf * BK * LDB_S;
      if (B_TRANS) {
#pragma unroll
        for (int j = 0; j < 4; j++) b_s[(4 * t_kq + j) * LDB_S + t_r + T_ROWS * i] = rb[i][j];
      } else {
        *reinterpret_cast<f32x4*>(b_s + (d_kk + 8 * i) * LDB_S + 4 * d_nq) = rb[i];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // Two register sets: while tile kt is multiplied out of LDS, the global loads of tile kt+2 are ISSUED into one set
  // (first half of the k-pairs) and tile kt+1 -- loaded one iteration earlier, long landed -- is WRITTEN to the other
  // LDS buffer from the other set (second half), one piece between each group of four MFMAs.  The matrix pipe never
  // waits for address arithmetic, a vmcnt drain or the LDS write pass; one barrier per k-step remains.
  f32x4 sa0[NLD], sb0[NLD], sa1[NLD], sb1[NLD];
  constexpr int NP = 2 * NLD;  // pieces per tile (== BK/4 == half of the BK/2 k-pairs)
  static_assert(NP * 2 == BK / 2, "piece schedule assumes 2*NP k-pairs per tile");
  if (nk > 0) {
#pragma unroll
    for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, 0);
#pragma unroll
    for (int q = 0; q < NP; q++) store_piece(sa0, sb0, q, 0);
#pragma unroll
    for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, min(1, nk - 1));
  }
  __syncthreads();

  auto k_step = [&](f32x4 (&ca)[NLD], f32x4 (&cb)[NLD], f32x4 (&na)[NLD], f32x4 (&nb)[NLD], int kt) {
    // ca/cb hold tile kt+1 (to be stored), na/nb receive tile kt+2
    const int buf = kt & 1;
    // branch-free on purpose: a conditional around a load makes hipcc drain vmcnt(0) at the join, serialising the
    // pipeline.  Past the end the last tile is simply re-loaded / re-stored into the idle buffer (never read).
    const int kt_load = min(kt + 2, nk - 1);
    const float* a_s = As + buf * BK * LDA_S + wm0 + l31;
    const float* b_s = Bs + buf * BK * LDB_S + wn0 + l31;
    float a0 = a_s[lh * LDA_S], a1 = a_s[lh * LDA_S + 32];
    float b0 = b_s[lh * LDB_S], b1 = b_s[lh * LDB_S + 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      float xa0 = 0.f, xa1 = 0.f, xb0 = 0.f, xb1 = 0.f;
      if (kk + 1 < BK / 2) {  // fragment reads of the NEXT k-pair go out before this k-pair's MFMAs
        const int krow = 2 * (kk + 1) + lh;
        xa0 = a_s[krow * LDA_S];
        xa1 = a_s[krow * LDA_S + 32];
        xb0 = b_s[krow * LDB_S];
        xb1 = b_s[krow * LDB_S + 32];
      }
      if (kk < NP) load_piece(na, nb, kk, kt_load);
      else store_piece(ca, cb, kk - NP, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);  // everything above is issued before this k-pair's MFMAs
      // operands swapped on purpose: D = (B fragment) x (A fragment) = the TRANSPOSED 32x32 tile, so that each lane
      // ends up with 4 consecutive output COLUMNS of one row -> 16-byte epilogue loads/stores
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
      a0 = xa0; a1 = xa1; b0 = xb0; b1 = xb1;
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    k_step(sa0, sb0, sa1, sb1, kt);
    if (kt + 1 < nk) k_step(sa1, sb1, sa0, sb0, kt + 1);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // acc[i][j][4q + e]: row = wm0 + 32 i + l31 ; col = wn0 + 32 j + 8 q + 4 lh + e   (e = 0..3 contiguous)
  float* __restrict__ Cg = p.C;
  long c_base = 0;
  int m_lim;
  if (MODE == MODE_TN) {
    c_base = (long)blockIdx.z * p.strideC;
    m_lim = p.M;
  } else {
    if (GATHER && p.kTilesPerSplit > 0) c_base = (long)blockIdx.z * p.strideC;
    m_lim = row_end;
  }
  const float* bias = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_SCALE_RES || EPI == EPI_BIAS_RELU)
                          ? p.bias + (long)g * p.strideBias
                          : nullptr;
  f32x4 csum[2][4];
  if (EPI == EPI_GELU_BWD) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) csum[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = m0 + wm0 + 32 * i + l31;
    if (row >= m_lim) continue;
    float rsc = 1.f;
    if (EPI == EPI_BIAS_SCALE_RES && p.rowscale) rsc = p.rowscale[row / p.rows_per_scale];
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int col = n0 + wn0 + 32 * j + 8 * q + 4 * lh;
        if (col >= p.N) continue;
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        float* cp = Cg + c_base + (long)row * p.ldc + col;
        const long ai = (long)row * p.ld_aux + col;
        if (EPI == EPI_NONE) {
          *reinterpret_cast<f32x4*>(cp) = v;
        } else if (EPI == EPI_BIAS) {
          *reinterpret_cast<f32x4*>(cp) = v + *reinterpret_cast<const f32x4*>(bias + col);
        } else if (EPI == EPI_BIAS_RELU) {
          f32x4 o = v + *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
          for (int e = 0; e < 4; e++) o[e] = fmaxf(o[e], 0.f);
          *reinterpret_cast<f32x4*>(cp) = o;
        } else if (EPI == EPI_BIAS_GELU) {
          const f32x4 h = v + *reinterpret_cast<const f32x4*>(bias + col);
          f32x4 y, dy;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float ye, de;
            gelu_erf_both(h[e], ye, de);
            y[e] = ye;
            dy[e] = de;
          }
          *reinterpret_cast<f32x4*>(p.aux_out + ai) = dy;
          *reinterpret_cast<f32x4*>(cp) = y;
        } else if (EPI == EPI_BIAS_SCALE_RES) {
          const f32x4 y = v + *reinterpret_cast<const f32x4*>(bias + col);
          *reinterpret_cast<f32x4*>(p.aux_out + ai) = y;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.gamma + col) * rsc;
          *reinterpret_cast<f32x4*>(cp) = *reinterpret_cast<const f32x4*>(p.aux_in + ai) + sc * y;
        } else if (EPI == EPI_GELU_BWD) {
          const f32x4 o = v * *reinterpret_cast<const f32x4*>(p.aux_in + ai);
          *reinterpret_cast<f32x4*>(cp) = o;
          csum[j][q] += o;
        }
      }
    }
  }
  if (EPI == EPI_GELU_BWD) {
    if (p.colpart) {  // uniform branch: column sums of this 128-row tile -> colpart[tile_m][n]
      constexpr int LDR = BN + 4;
      float* red = smem;  // the k-loop ended with a barrier: LDS is free.  [64 = 2 wave rows x 32 lanes][LDR]
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<f32x4*>(red + ((wave >> 1) * 32 + l31) * LDR + wn0 + 32 * j + 8 * q + 4 * lh) = csum[j][q];
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; r++) t += red[r * LDR + tid];
        p.colpart[(long)tile_m * p.N + n0 + tid] = t;
      }
    }
  }
}

// colpart [row tiles][N] -> out[g][n]: sum the tiles that belong to group g (same tile->group map as the GEMM).
// block = 64 columns x 4 tile lanes
__global__ __launch_bounds__(256) void tile_colsum_reduce_kernel(const float* __restrict__ colpart, int N, int M,
                                                                const int32_t* __restrict__ offsets,
                                                                int num_groups, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int g = blockIdx.y;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  int t0 = 0, t1 = (M + BM - 1) / BM;
  if (offsets) {
    int base = 0;
    for (int gg = 0; gg < g; gg++) base += (offsets[gg + 1] - offsets[gg] + BM - 1) / BM;
    t0 = base;
    t1 = base + (offsets[g + 1] - offsets[g] + BM - 1) / BM;
  }
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int t = t0 + rl;
    for (; t + 4 < t1; t += 8) {
      s0 += colpart[(long)t * N + n];
      s1 += colpart[(long)(t + 4) * N + n];
    }
    for (; t < t1; t += 4) s0 += colpart[(long)t * N + n];
  }
  red[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && n < N) out[(long)g * N + n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Sum the split-K partials of a TN GEMM: out[g][i] = sum_s ws[(g*splits+s)][i]
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long mn, int splits,
                                     int groups, const float* __restrict__ bias = nullptr, int ncols = 0,
                                     int relu = 0) {
  const long total = mn * groups;
  for (long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; idx < total;
       idx += (long)gridDim.x * blockDim.x * 4) {
    const long gidx = idx / mn;
    const long e = idx - gidx * mn;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (bias) s = *reinterpret_cast<const f32x4*>(bias + (e % ncols));  // rows of ncols (multiple of 4) columns
    for (int k = 0; k < splits; k++) s += *reinterpret_cast<const f32x4*>(ws + (gidx * splits + k) * mn + e);
    if (relu) {
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] = fmaxf(s[q], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + idx) = s;
  }
}

// Column sums over row segments (bias gradients): out[g][n] += sum_{r in seg g} X[r][n]; `out` is zeroed by the
// caller (memset on the same stream).  Block = 16 column quads (64 columns, 16 B per lane) x 16 row lanes; row
// chunks are combined with fp32 L2 atomics (<= COLSUM_SPLITS adds per address).
constexpr int COLSUM_SPLITS = 128;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, int N,
                                                    const int32_t* __restrict__ offsets, int M, int splits,
                                                    float* __restrict__ out) {
  const int g = blockIdx.z;
  int r0 = 0, r1 = M;
  if (offsets) {
    r0 = offsets[g];
    r1 = offsets[g + 1];
  }
  const int chunk = (r1 - r0 + splits - 1) / splits;
  r0 += blockIdx.y * chunk;
  r1 = min(r1, r0 + chunk);
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 16) s += *reinterpret_cast<const f32x4*>(X + (long)r * ld + c);
  __shared__ f32x4 red[16][16];
  red[rl][cq] = s;
  __syncthreads();
  if (threadIdx.x < 64 && r0 < r1) {
    const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) t += red[i][q][e];
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col < N) atomicAdd(out + (long)g * N + col, t);
  }
}

template <int MODE, int BK>
int launch_mode(const GemmParams& p, int epi, dim3 grid, hipStream_t st) {
#define SM3_LAUNCH(E)                                                  \
  case E:                                                              \
    gemm_f32_kernel<MODE, E, BK><<<grid, NTHREADS, 0, st>>>(p);        \
    return SM3_OK;
  switch (epi) {
    SM3_LAUNCH(EPI_NONE)
    SM3_LAUNCH(EPI_BIAS)
    SM3_LAUNCH(EPI_BIAS_GELU)
    SM3_LAUNCH(EPI_BIAS_SCALE_RES)
    SM3_LAUNCH(EPI_GELU_BWD)
    case EPI_BIAS_RELU:  // only the x.W^T form needs it (fully-connected + ReLU of the RoI head)
      if (MODE != MODE_NT) return SM3_ERR_INVALID_ARG;
      gemm_f32_kernel<MODE_NT, EPI_BIAS_RELU, BK><<<grid, NTHREADS, 0, st>>>(p);
      return SM3_OK;
  }
#undef SM3_LAUNCH
  return SM3_ERR_INVALID_ARG;
}

}  // namespace

extern "C" {

size_t sm3_gemm_f32_workspace_bytes(const sm3_gemm_desc* d) {
  if (!d) return 0;
  if (d->mode != MODE_TN) {
    if (d->epilogue == EPI_GELU_BWD && d->colsum_out) {
      const int ntm = (d->M + BM - 1) / BM + (d->group_offsets ? (d->num_groups > 0 ? d->num_groups : 1) : 0);
      return (size_t)ntm * d->N * sizeof(float);
    }
    return 0;
  }
  const int groups = d->num_groups > 0 ? d->num_groups : 1;
  const int splits = d->splits > 0 ? d->splits : 1;
  return (size_t)groups * splits * d->M * d->N * sizeof(float);
}

// k-step depth: 16 gives 4 resident blocks per CU (33.8 KB LDS, <=128 VGPRs), 32 gives 2.
static int choose_bk(const sm3_gemm_desc* d) {
  if (d->mode == MODE_TN) return 16;   // measured on MI355X (scripts/gemm_shapes.py): split-K wgrad always prefers 16
  return (d->K >= 192 && d->K <= 768) ? 16 : 32;  // mid-K GEMMs are prologue/epilogue bound: more resident blocks win
}

int sm3_gemm_f32(const sm3_gemm_desc* d, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (!d) return SM3_ERR_INVALID_ARG;
  if (d->M < 0 || d->N <= 0 || d->K < 0) return SM3_ERR_INVALID_ARG;
  if ((d->lda & 3) || (d->ldb & 3) || (d->N & 3)) return SM3_ERR_UNSUPPORTED;  // float4 loads
  if (d->mode != MODE_TN && (d->K % BK_MAX) != 0) return SM3_ERR_UNSUPPORTED;
  static const int env_bk = [] { const char* e = getenv("SM3_GEMM_BK"); return e ? atoi(e) : 0; }();
  int bk = env_bk == 16 || env_bk == 32 ? env_bk : choose_bk(d);
  if (d->mode == MODE_TN && (d->M & 3)) return SM3_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  GemmParams p;
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.offsets = d->group_offsets;
  p.num_groups = d->num_groups > 0 ? d->num_groups : 1;
  p.strideB = d->stride_b; p.strideBias = d->stride_bias;
  p.strideC = (long)d->M * d->N;
  p.splits = d->splits > 0 ? d->splits : 1;
  p.bias = d->bias; p.aux_in = d->aux_in; p.aux_out = d->aux_out;
  p.gamma = d->gamma; p.rowscale = d->rowscale;
  p.rows_per_scale = d->rows_per_scale > 0 ? d->rows_per_scale : 1;
  p.ld_aux = d->ld_aux;
  p.colpart = nullptr;
  const int ntn = (d->N + BN - 1) / BN;
  if (d->mode == MODE_TN) {
    if (d->epilogue != EPI_NONE) return SM3_ERR_INVALID_ARG;
    const size_t need = sm3_gemm_f32_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) return SM3_ERR_WORKSPACE;
    const int ntm = (d->M + BM - 1) / BM;
    float* out = d->C;
    p.C = (float*)workspace;
    p.ldc = d->N;
    dim3 grid(ntn * ntm, 1, p.num_groups * p.splits);
    if (bk == 16) gemm_f32_kernel<MODE_TN, EPI_NONE, 16><<<grid, NTHREADS, 0, st>>>(p);
    else gemm_f32_kernel<MODE_TN, EPI_NONE, 32><<<grid, NTHREADS, 0, st>>>(p);
    const long mn = (long)d->M * d->N;
    long nb = (mn * p.num_groups / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    splitk_reduce_kernel<<<(int)nb, 256, 0, st>>>((const float*)workspace, out, mn, p.splits, p.num_groups);
    return launch_status();
  }
  if (d->M == 0) return SM3_OK;
  // ragged groups: at most ceil(M/BM) + G row tiles exist; surplus blocks exit
  const int ntm = (d->M + BM - 1) / BM + (d->group_offsets ? p.num_groups : 0);
  dim3 grid(ntn * ntm, 1, 1);
  const bool want_colsum = (d->epilogue == EPI_GELU_BWD && d->colsum_out);
  if (want_colsum) {
    if (!workspace || workspace_bytes < (size_t)ntm * d->N * sizeof(float)) return SM3_ERR_WORKSPACE;
    p.colpart = (float*)workspace;
  }
  int rc;
  if (d->mode == MODE_NT)
    rc = bk == 16 ? launch_mode<MODE_NT, 16>(p, d->epilogue, grid, st) : launch_mode<MODE_NT, 32>(p, d->epilogue, grid, st);
  else if (d->mode == MODE_NN)
    rc = bk == 16 ? launch_mode<MODE_NN, 16>(p, d->epilogue, grid, st) : launch_mode<MODE_NN, 32>(p, d->epilogue, grid, st);
  else return SM3_ERR_INVALID_ARG;
  if (rc) return rc;
  if (want_colsum) {
    dim3 rg((d->N + 63) / 64, p.num_groups);
    tile_colsum_reduce_kernel<<<rg, 256, 0, st>>>(p.colpart, d->N, d->M, d->group_offsets, p.num_groups,
                                                 d->colsum_out);
  }
  return launch_status();
}

int sm3_colsum_f32(const float* x, int ld, int m, int n, const int32_t* group_offsets, int num_groups, float* out,
                   sm3_stream_t stream) {
  if (n <= 0 || m < 0 || !x || !out) return SM3_ERR_INVALID_ARG;
  const int groups = group_offsets ? num_groups : 1;
  const int rows_per_group = m / groups > 0 ? m / groups : 1;
  int splits = rows_per_group / 256;
  if (splits < 1) splits = 1;
  if (splits > COLSUM_SPLITS) splits = COLSUM_SPLITS;
  dim3 grid((n + 63) / 64, splits, groups);
  (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)groups * n, (hipStream_t)stream);
  colsum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ld, n, group_offsets, m, splits, out);
  return launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution (padding 1, stride 1 or 2) on NHWC tokens as implicit GEMMs on the same kernel (GATHER = 1): the
// operand that a materialised im2col would hold is gathered tap by tap while the tile is loaded, so no column buffer
// exists (the FPN 3x3 at 256^2 x 256 ch would need a 1.2 GB one).  Weight layout (Cout, 3, 3, Cin) = [Cout][9*Cin].
static void conv_geometry(GemmParams& p, int cC, int sH, int sW, int rH, int rW, int stride, int transposed, int bk) {
  p.cC = cC; p.sH = sH; p.sW = sW; p.rH = rH; p.rW = rW; p.cS = stride; p.cT = transposed;
  const unsigned tpt = (unsigned)(cC / bk);
  p.cInv = (65536u + tpt - 1) / tpt;
  p.mRW = (unsigned)((1ull << 32) / (unsigned)rW + 1);
  p.mRH = (unsigned)((1ull << 32) / (unsigned)rH + 1);
}

static GemmParams conv_params_zero() {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.num_groups = 1;
  p.splits = 1;
  p.rows_per_scale = 1;
  return p;
}

static bool conv_dims_ok(int B, int H, int W, int Cin, int Cout, int stride) {
  return B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) &&
         (long)B * H * W < (1l << 24);
}

// Small pyramid levels have fewer 128x128 output tiles than the chip has CUs while K = 9*C is long: split K across
// blockIdx.z (whole k-tiles, tap-aligned or not), partial slabs to the workspace, one reduce (+ bias).
static int conv_ksplits(long m_rows, int n_cols, int k_tiles) {
  const long tiles = ((m_rows + BM - 1) / BM) * ((n_cols + BN - 1) / BN);
  if (tiles >= 512) return 1;
  long s = (1024 + tiles - 1) / tiles;
  if (s > k_tiles / 4) s = k_tiles / 4;  // at least 4 k-tiles per split
  if (s > 16) s = 16;
  return s < 1 ? 1 : (int)s;
}

size_t sm3_conv3x3_nhwc_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride, int backward_input) {
  if (!conv_dims_ok(B, H, W, Cin, Cout, stride)) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long m = backward_input ? (long)B * H * W : (long)B * Ho * Wo;
  const int n = backward_input ? Cin : Cout, kc = backward_input ? Cout : Cin;
  const int ks = conv_ksplits(m, n, 9 * kc / 32);
  return ks > 1 ? (size_t)ks * m * n * sizeof(float) : 0;
}

static int conv_launch_nt_nn(GemmParams& p, int mode, const float* bias, int relu, float* out, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  const int k_tiles = p.K / 32;
  const int ks = conv_ksplits(p.M, p.N, k_tiles);
  const long mn = (long)p.M * p.N;
  if (ks > 1) {
    if (!workspace || workspace_bytes < (size_t)ks * mn * sizeof(float)) return SM3_ERR_WORKSPACE;
    p.kTilesPerSplit = (k_tiles + ks - 1) / ks;
    p.strideC = mn;
    p.C = (float*)workspace;
    p.bias = nullptr;
  } else {
    p.C = out;
  }
  const int zs = ks > 1 ? (k_tiles + p.kTilesPerSplit - 1) / p.kTilesPerSplit : 1;
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, zs);
  if (mode == MODE_NN) gemm_f32_kernel<MODE_NN, EPI_NONE, 32, 1><<<grid, NTHREADS, 0, st>>>(p);
  else if (p.bias && relu) gemm_f32_kernel<MODE_NT, EPI_BIAS_RELU, 32, 1><<<grid, NTHREADS, 0, st>>>(p);
  else if (p.bias) gemm_f32_kernel<MODE_NT, EPI_BIAS, 32, 1><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE_NT, EPI_NONE, 32, 1><<<grid, NTHREADS, 0, st>>>(p);
  if (ks > 1) {
    long nb = (mn / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    splitk_reduce_kernel<<<(int)nb, 256, 0, st>>>((const float*)workspace, out, mn, zs, 1, bias, p.N, relu);
  }
  return launch_status();
}

int sm3_conv3x3_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                         int Cout, int stride, int relu, void* workspace, size_t workspace_bytes,
                         sm3_stream_t stream) {
  if (!x || !w || !y || !conv_dims_ok(B, H, W, Cin, Cout, stride)) return SM3_ERR_INVALID_ARG;
  if (relu && !bias) return SM3_ERR_INVALID_ARG;
  if ((Cin % 32) || (Cout & 3)) return SM3_ERR_UNSUPPORTED;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmParams p = conv_params_zero();
  p.A = x; p.B = w;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.lda = Cin; p.ldb = 9 * Cin; p.ldc = Cout; p.ld_aux = Cout;
  p.bias = bias;
  conv_geometry(p, Cin, H, W, Ho, Wo, stride, 0, 32);
  return conv_launch_nt_nn(p, MODE_NT, bias, relu, y, workspace, workspace_bytes, (hipStream_t)stream);
}

int sm3_conv3x3_nhwc_bwd_input(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout,
                               int stride, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (!dy || !w || !dx || !conv_dims_ok(B, H, W, Cin, Cout, stride)) return SM3_ERR_INVALID_ARG;
  if ((Cout % 32) || (Cin & 3)) return SM3_ERR_UNSUPPORTED;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmParams p = conv_params_zero();
  p.A = dy; p.B = w;
  p.M = B * H * W; p.N = Cin; p.K = 9 * Cout;
  p.lda = Cout; p.ldb = 9 * Cin; p.ldc = Cin; p.ld_aux = Cin;
  conv_geometry(p, Cout, Ho, Wo, H, W, stride, 1, 32);
  return conv_launch_nt_nn(p, MODE_NN, nullptr, 0, dx, workspace, workspace_bytes, (hipStream_t)stream);
}

static int conv_wgrad_splits(int B, int H, int W, int Cin, int Cout, int stride) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long rows = (long)B * Ho * Wo;
  const long tiles = (long)((Cout + BM - 1) / BM) * ((9 * Cin + BN - 1) / BN);
  long s = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU in total
  const long smax = (rows + 255) / 256;         // at least 256 reduction rows per split
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  return (int)s;
}

size_t sm3_conv3x3_nhwc_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride) {
  if (!conv_dims_ok(B, H, W, Cin, Cout, stride)) return 0;
  return (size_t)conv_wgrad_splits(B, H, W, Cin, Cout, stride) * Cout * 9 * Cin * sizeof(float);
}

int sm3_conv3x3_nhwc_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                int stride, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (!x || !dy || !dw || !conv_dims_ok(B, H, W, Cin, Cout, stride)) return SM3_ERR_INVALID_ARG;
  if ((Cin % BN) || (Cout & 3)) return SM3_ERR_UNSUPPORTED;  // an N-tile (128 columns of [9*Cin]) must sit in one tap
  const size_t need = sm3_conv3x3_nhwc_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, stride);
  if (!workspace || workspace_bytes < need) return SM3_ERR_WORKSPACE;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmParams p = conv_params_zero();
  p.A = dy; p.B = x; p.C = (float*)workspace;
  p.M = Cout; p.N = 9 * Cin; p.K = B * Ho * Wo;
  p.lda = Cout; p.ldb = Cin; p.ldc = 9 * Cin; p.ld_aux = 9 * Cin;
  p.splits = conv_wgrad_splits(B, H, W, Cin, Cout, stride);
  p.strideC = (long)p.M * p.N;
  conv_geometry(p, Cin, H, W, Ho, Wo, stride, 0, 16);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, p.splits);
  gemm_f32_kernel<MODE_TN, EPI_NONE, 16, 1><<<grid, NTHREADS, 0, st>>>(p);
  const long mn = (long)p.M * p.N;
  long nb = (mn / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  splitk_reduce_kernel<<<(int)nb, 256, 0, st>>>((const float*)workspace, dw, mn, p.splits, 1);
  return launch_status();
}

}  // extern "C"
