// gemm_f32_kernel.h -- the fp32 GEMM kernel template of libsm3det_hip.so (gfx950 matrix cores,
// v_mfma_f32_32x32x2_f32: exact f32 FMA chain, 157 TFLOP/s peak; there is no TF32/xf32 on CDNA4).
//
// One template covers every dense contraction of the SM3Det backbone hot path (reference:
// mmrotate/models/backbones/convnext_moe.py FFN.forward :397-405, the expert loop :244, CosineTopKGate projection :101,
// stem / downsample convs as patch GEMMs :533-558,783-791), their backward, and the 3x3 NHWC convolutions of the neck
// and heads as implicit GEMMs (GATHER):
//
//   MODE_NT : C[M,N] = A[M,K] . B[N,K]^T        (nn.Linear forward: x @ W^T)
//   MODE_NN : C[M,N] = A[M,K] . B[K,N]          (dgrad: dY @ W)
//   MODE_TN : C[M,N] = A[Kt,M]^T . B[Kt,N]      (wgrad: dY^T @ X, reduction over token rows)
//
// Grouped (MoE experts): rows of A/C (NT, NN) or the reduction rows (TN) are partitioned into `num_groups` contiguous
// segments by a DEVICE prefix array `offsets[G+1]` (expert-major slot order); group g uses weight block g.  The
// tile->group map is computed in the kernel, so ragged expert loads never sync the host (the reference does `.cpu()`
// per MoE block: convnext_moe.py:259).
//
// Tiling: 256 threads = 4 waves arranged WM x WN, each wave owns TI x TJ MFMA tiles of 32x32 (block tile
// BM = 32*WM*TI, BN = 32*WN*TJ).  Shapes instantiated: 128x128 (2,2,2,2), 128x96 (4,1,1,3), 96x128 (1,4,3,1),
// 128x192 (2,2,2,3), 192x128 (2,2,3,2), 64x128 (2,2,1,2): the ConvNeXt channel widths 96 / 192 are not multiples of
// 128 and a 128-wide tile wastes a quarter of its MFMA work on them.  Operands are staged k-major in LDS so that one
// conflict-free ds_read_b32 per lane fetches exactly the A[i][k] / B[k][j] fragment of the 32x32x2 instruction
// (lanes 0-31: row k, lanes 32-63: row k+1); two register sets software-pipeline the loop (loads of tile kt+2 and LDS
// writes of tile kt+1 interleaved between the MFMAs of tile kt, one barrier per k-step).
//
// Split-K with an in-kernel fix-up (all modes): blockIdx.z slices K; every slice writes its accumulators as a
// fragment-ordered slab (thread-major 16-byte stores), publishes it with ONE agent-scope release + ticket, and the
// last-arriving block of a tile sums the slabs in slice order (deterministic) and runs the epilogue -- no second
// kernel, no second pass over the output (cdna_hip_programming.md, in-launch split-K recipe).  The ticket counters
// are a caller-provided zeroed array that the kernel leaves zeroed.
//
// The k-loop carries no vector-ALU work (every VALU instruction issued between the MFMAs of a SIMD takes issue time from
// them): operands are addressed as buffer descriptor (SGPRs) + scalar k offset + a per-lane 32-bit offset fixed for the
// whole loop; clamps, the TN row mask and the zero stage of an odd trip count live in a `tail` variant of the step that
// runs the last 3-4 k-tiles only; two steps per iteration with no branch between them; lanes past a tile edge repeat the
// tile's last row into the same LDS slot.  GATHER (implicit-GEMM 3x3 convolution) splits the source pixel into a per-row
// part and a per-tap part that is equal for all lanes; masked taps read past the descriptor's extent (hardware zero).
// F16 = 1: fp16 image in LDS (v_mfma_f32_32x32x16_f16, fp32 accumulation); IO flags say which tensors are fp16 in HBM.
// CSUM = 1 (TN): the workgroups of the first column tile also sum the rows of A they load (bias gradient).
#pragma once
#include "common.h"

namespace sm3gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MODE_NT = 0, MODE_NN = 1, MODE_TN = 2;
constexpr int NTHREADS = 256;
// EPI codes (must match include/sm3det_hip.h)
constexpr int EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_SCALE_RES = 3, EPI_GELU_BWD = 4;
constexpr int EPI_BIAS_RELU = 5;
constexpr int kCounterSlots = 1 << 16;

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
  long strideC;     // TN: elements between (group, split) outputs (= M*N)
  // split-K
  int splits;          // slices of K per tile (TN: per group)
  int fixup;           // 1: fragment-ordered slabs + in-kernel last-arriver reduction; 0: slice z -> C + z*strideC, reduced later
  int kTilesPerSplit;  // NT/NN: k-tiles per slice
  float* slabs;        // fixup: [tile][split][BM*BN]
  int* counters;       // fixup: one ticket counter per tile, zero on entry, zero on exit
  // epilogue operands
  const float* bias;      // [N] (per group)
  const float* aux_in;    // EPI_GELU_BWD: gelu'(h)[M,N];  EPI_BIAS_SCALE_RES: residual[M,N]
  float* aux_out;         // EPI_BIAS_GELU: gelu'(h)[M,N]; EPI_BIAS_SCALE_RES: y[M,N]
  float* colpart;         // EPI_GELU_BWD: per-row-tile column sums [row tiles][N] (bias gradient partials) or NULL
  float* csum;            // TN (fp32): column sums of A over this launch's reduction rows -> csum[z * csum_stride + m],
  long csum_stride;       //   z = group * splits + slice (the bias gradient as a by-product of the weight gradient) or NULL
  const float* gamma;     // [N] layer scale
  const float* rowscale;  // [M / rows_per_scale] (stochastic-depth keep/keep_prob per image) or NULL
  int rows_per_scale;
  int ld_aux;
  // GATHER (implicit-GEMM 3x3 convolution over NHWC tokens, padding 1): the gathered operand has `cC` channels per tap,
  // its rows live on an (sH, sW) grid per image; the GEMM's own rows (NT/NN: A/C rows, TN: reduction rows) live on an
  // (rH, rW) grid.  cT = 0: source = row * cS + d - 1 (forward / weight gradient);  cT = 1: source = (row - d + 1) / cS
  // where divisible (input gradient = transposed convolution).  cInv: (kt * cInv) >> 16 == kt / (cC / BK).
  int cC, sH, sW, rH, rW, cS, cT;
  int cU;  // per-tap source offset u(d), d = 0..2, 4 bits each: d (forward), 2 - d (input gradient), 1 - d/2 (same, stride 2)
  unsigned cInv, mRW, mRH;  // mRW/mRH: ceil(2^32 / rW), ceil(2^32 / rH) for exact umulhi division of row indices
  int eq_prio;  // lower the wave priority with the workgroup's progress through K (see the k-loop)
#ifdef SM3_TRACE  // measurement build only (python -m sm3det_amd.build --variant trace): per-workgroup phase timestamps
  unsigned long long* trace;  // [blocks][8]: s_memtime at entry / loop start / loop end / after fix-up / exit, HW_ID|XCC_ID<<32, nk, s_memrealtime
#endif
};

#ifdef SM3_TRACE
#define SM3_TR(i)                                                                                    \
  do {                                                                                               \
    if (p.trace && threadIdx.x == 0 && sm3_first_pass)                                               \
      p.trace[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SM3_TR(i) do {} while (0)
#endif

template <int WM_, int WN_, int TI_, int TJ_>
struct Tile {
  static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_;
  static constexpr int BM = 32 * WM_ * TI_, BN = 32 * WN_ * TJ_;
  static_assert(WM_ * WN_ == 4, "4 waves per workgroup");
};
using T128x128 = Tile<2, 2, 2, 2>;
using T128x96 = Tile<4, 1, 1, 3>;
using T96x128 = Tile<1, 4, 3, 1>;
using T128x192 = Tile<2, 2, 2, 3>;
using T192x128 = Tile<2, 2, 3, 2>;
using T64x128 = Tile<2, 2, 1, 2>;

// resident workgroups per CU the launch bounds ask for (VGPR budget 512 / waves-per-SIMD)
#ifndef SM3_F16_OCC
#define SM3_F16_OCC 3  // A/B builds: python -m sm3det_amd.build --variant f16_occ2
#endif
#ifndef SM3_B3_SIGNED
#define SM3_B3_SIGNED 2  // bf16x3 form, the bf16 MFMA's downward accumulation bias (see `SIGNED` / `CHECKER` in the kernel):
#endif                   // 2 = one accumulator set, the SIGN of the accumulated value alternates over the 32-row blocks of the
                         // token rows of the output (round 6, default); 1 = odd k-tiles accumulate negated products in a second
                         // set (round 5; --variant b3_two_sets); 0 = one set, no cancellation (--variant b3_unsigned)
#ifndef SM3_B3_OCC
#define SM3_B3_OCC (SM3_B3_SIGNED == 1 ? 2 : 3)  // bf16x3 form: workgroups per CU the launch bounds ask for
#endif
#ifndef SM3_PL_AB_OCC
#define SM3_PL_AB_OCC 2  // both operands as planes at 128x128: six 16-byte pieces per register set spill at 168 VGPRs
#endif
template <class TL, int BK, int F16 = 0, int CSUM = 0, int IO = 0, int MODE = -1>
constexpr int occupancy() {
  if (F16 == 2 && IO == 48 && TL::TI * TL::TJ >= 4) return SM3_PL_AB_OCC;
#ifndef SM3_F16_KEEP_SPILLS  // (A/B: --variant f16_spills restores round 5's bounds)
  // fp16 operands: the instantiations that do not fit 168 registers (round 5: 4-40 spilled VGPRs, reloaded inside the tail
  // loop = the WHOLE loop of the K = 96 / 192 launches, and inside the bulk loop of the column-sum weight gradients) run at
  // two workgroups per CU instead -- the loop is bound by operand delivery, which did not care (7.07 vs 7.01 ms, round 3)
  if (F16 == 1 && ((MODE == MODE_NT && TL::TI * TL::TJ >= 4 && BK == 32) || CSUM)) return 2;
#endif
  // F16: the fp16 LDS image of a k-step-32 tile is 34 KB (the fp32 image 66 KB), so three workgroups fit a CU and the
  // loop -- bound by load latency, not by the matrix pipe -- gets a third wave per SIMD to hide it; k-step 64 (68 KB,
  // twice the MFMAs per barrier) runs two
  // (128x128 with the column-sum by-product: 8 more live registers; at 168 it spills INTO the k-loop)
  if (F16 == 2) return (TL::TI * TL::TJ >= 6 || (CSUM && TL::TI * TL::TJ >= 4)) ? 2 : SM3_B3_OCC;
  if (F16) return BK == 64 ? 2 : SM3_F16_OCC;
  // 128x128 at k-step 16 needs ~150 VGPRs to keep its LDS read bases out of the loop: 3 waves per SIMD without spills
  // instead of 4 with scratch reloads and address arithmetic between the MFMAs
  return (TL::TI * TL::TJ >= 6 || BK == 32) ? 2 : (TL::TI * TL::TJ >= 3 ? 3 : 4);
}

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
// ~55 for ocml's erff + expf.
__device__ __forceinline__ void gelu_erf_both(float h, float& y, float& dy) {
#ifdef SM3_GELU_EXACT  // A/B build only (python -m sm3det_amd.build --variant gelu_exact): ocml erff / expf, ~55 VALU ops
  const float cdf_ = 0.5f * (1.0f + erff(h * 0.70710678118654752440f));
  y = h * cdf_;
  dy = fmaf(h, 0.39894228040143267794f * expf(-0.5f * h * h), cdf_);
  return;
#endif
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

// IO (F16 = 1 only): which tensors live in HBM as fp16 -- the AMP data path (mmcv wrap_fp16_model / autocast: the outputs of
// nn.Linear are half tensors, mmcv/mmcv/runner/fp16_utils.py:71-149).  Weights, biases, the residual stream and every
// C-wide gradient stay fp32; the LayerNorm output that feeds the GEMMs and the three 4C-wide tensors of a block (GELU
// output, GELU', their gradient) are fp16: half the bytes of the launches that move them.
constexpr int IO_A16 = 1, IO_B16 = 2, IO_C16 = 4, IO_X16 = 8;  // A operand, B operand, C output, aux_in / aux_out
// IO (F16 = 2, NT / NN): which operands arrive as bf16x3 PLANES (planes.hip: planes[p][k / 8][row][8], one 16-byte granule per
// plane, k-octet and row) instead of fp32 -- split once by their producer, moved by the loader as they are (one 16-byte load
// and one 16-byte LDS store per granule, no arithmetic).  A: `A` addresses the planes of A[M][K], lda = rows per octet block;
// B: the planes of the K-CONTIGUOUS form of B -- B[N][K] in NT, B^T in NN -- ldb = rows per octet block, strideB = rows
// between groups.  Bit-identical results to the fp32-operand launch (the split is the same arithmetic).
constexpr int IO_APL = 16, IO_BPL = 32;

template <int MODE, int EPI, int BK, class TL, int GATHER, int F16 = 0, int CSUM = 0, int IO = 0>
__global__ __launch_bounds__(NTHREADS, (occupancy<TL, BK, F16, CSUM, IO, MODE>())) void gemm_f32_kernel(GemmParams p) {
  constexpr int BM = TL::BM, BN = TL::BN, TI = TL::TI, TJ = TL::TJ, WN = TL::WN, WM = TL::WM;
  constexpr bool A16 = (IO & IO_A16) != 0, B16 = (IO & IO_B16) != 0, C16 = (IO & IO_C16) != 0, X16 = (IO & IO_X16) != 0;
  constexpr bool APL = (IO & IO_APL) != 0, BPL = (IO & IO_BPL) != 0;
  static_assert(IO == 0 || (F16 == 1 && !GATHER && (IO & ~15) == 0) ||
                    (F16 == 2 && !GATHER && MODE != MODE_TN && (IO & ~(IO_APL | IO_BPL)) == 0),
                "fp16 storage only with fp16 operands; bf16x3 planes only in the plain NT / NN launches");
  // B16 in NT / NN: the B operand is an fp16 SHADOW of the weights (what wrap_fp16_model's half model holds), see gemm_h16.hip
  static_assert(!X16 || EPI == EPI_BIAS_GELU || EPI == EPI_GELU_BWD, "fp16 auxiliary tensor = GELU' only");
  constexpr int EA = A16 ? 2 : 4, EB = B16 ? 2 : 4;  // bytes per element in HBM
  // A tile is written transposed (k-contiguous source) in NT/NN, directly (k-major source) in TN; B transposed in NT.
  // Leading dim of a transposed tile: the 4-byte scatter writes of one half-wave must hit 32 distinct banks:
  // BK=32 -> 8 k-quads x 4 rows need LD = 1 (mod 8); BK=16 -> 4 k-quads x 8 rows need LD = 2 (mod 8).
  constexpr bool A_TRANS = (MODE != MODE_TN);
  constexpr bool B_TRANS = (MODE == MODE_NT);
  constexpr int PADT = (BK == 32) ? 1 : 2;
  constexpr int LDA_S = A_TRANS ? BM + PADT : BM + 4;
  constexpr int LDB_S = B_TRANS ? BN + PADT : BN + 4;
  constexpr int A_STAGE = BK * LDA_S, B_STAGE = BK * LDB_S;
  // F16: the LDS image holds fp16, converted once when a piece is stored.  The 8 consecutive k of one row (one MFMA
  // fragment of a lane) are 16 contiguous bytes: element (k, r) lives in half ((k / 8) * LD16 + r) * 8 + (k & 7), so a
  // fragment is ONE conflict-free ds_read_b128 with no conversion or permute between it and the MFMA (the fp32 image
  // needed 8 ds_read_b32 + 4 conversions per fragment).  LD16 = rows + 4 = 4 (mod 16) rows keeps the 8-byte transposed
  // stores of a k-step-32 tile on distinct banks.
  // B3 (F16 == 2): fp32-equivalent arithmetic on the bf16 matrix pipe.  Every fp32 operand element is split EXACTLY into
  // three bf16 pieces x = x0 + x1 + x2 (round-to-nearest each: 3 x 8 significant bits + signs >= the 24 of fp32) when its
  // piece is stored to LDS; the image holds the three pieces as three fp16-style planes, and a k-step issues the six
  // products a_i . b_j with i + j <= 2 on v_mfma_f32_32x32x16_bf16 (fp32 accumulation; each bf16 x bf16 product is exact in
  // fp32, the dropped terms are <= 2^-25 |a||b|).  6 instructions of 32 cycles replace 8 of 64 per 16 k: the fp32 matrix
  // instruction runs at the VECTOR rate (1/16 of the bf16 rate), this form at 6/16 of it.  k-step 16 only (the three
  // planes of a k-step-32 tile would not leave room for two workgroups per CU).
  constexpr bool B3 = (F16 == 2);
  static_assert(!B3 || (BK == 16 && (IO & 15) == 0 && !(MODE == MODE_TN && GATHER == 1)), "bf16x3 form: k-step 16, fp32 tensors or planes; TN gather only in the row-aligned form (GATHER == 2)");
  constexpr int NIMG = B3 ? 3 : 1;
  constexpr int LDA16 = BM + 4, LDB16 = BN + 4;                          // rows per k-octet
  constexpr int A_ST16 = (BK / 8) * LDA16 * 4, B_ST16 = (BK / 8) * LDB16 * 4;  // dwords per plane and stage
  constexpr int A_STG = NIMG * A_ST16, B_STG = NIMG * B_ST16;                // dwords per stage
  // the allocation is the image of the instantiation's own precision (round 2 gave F16 the fp32 size, which capped it at
  // two workgroups per CU); never below the epilogue's per-wave staging patches (4 x 32 x 36 floats)
  static_assert(!(CSUM && A16), "the column-sum by-product adds the fp32 values of the A pieces");
  constexpr int SMEM_IMAGE = F16 ? 2 * (A_STG + B_STG) : 2 * (A_STAGE + B_STAGE);
  constexpr int SMEM_WORDS = SMEM_IMAGE > 4 * 32 * 36 ? SMEM_IMAGE : 4 * 32 * 36;
  static_assert(!(F16 && CSUM) || (BK / 4) * BM <= SMEM_WORDS, "column-sum scratch of the fp16-operand TN form");
  __shared__ __attribute__((aligned(16))) float smem[SMEM_WORDS];
  float* As = smem;                // [2][BK][LDA_S]
  uint32_t* Aw = reinterpret_cast<uint32_t*>(smem);  // [2][BK/8][LDA16][4 dwords], then B
  uint32_t* Bw = Aw + 2 * A_STG;
  float* Bs = smem + 2 * A_STAGE;  // [2][BK][LDB_S]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
#ifdef SM3_TRACE
  bool sm3_first_pass = true;  // the timestamps describe the workgroup's first item
  unsigned sm3_items = 0;
  SM3_TR(0);
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) * 8;
    tr[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |      // HW_REG_HW_ID
            ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
    tr[7] = __builtin_amdgcn_s_memrealtime();
    tr[1] = tr[2] = tr[3] = tr[4] = 0;
  }
#endif
#ifdef SM3_PRIO  // measurement build only: static wave priority by residency slot (co-resident workgroups get different ones)
  {
    const unsigned lb = blockIdx.x + gridDim.x * blockIdx.z;
    const unsigned cls = SM3_PRIO == 1 ? (lb >> 8) % 3u : 2u - (lb >> 8) % 3u;
    if (cls == 1) __builtin_amdgcn_s_setprio(1);
    else if (cls == 2) __builtin_amdgcn_s_setprio(2);
  }
#endif
  const int wm0 = (wave / WN) * (TI * 32);
  const int wn0 = (wave % WN) * (TJ * 32);
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // ---- work items ------------------------------------------------------------------------------------------------
  // One item = one output tile (x one k-slice: blockIdx.z), one item per workgroup (grid.x = items).
  const int nitems = (int)gridDim.x;
  const int ntn = (p.N + BN - 1) / BN;
  // group offsets of a grouped NT / NN launch in ONE vector load: lane l holds offsets[l] and the tile -> group look-up is
  // scalar ALU over v_readlane (the look-up used to issue up to 2 * G dependent scalar loads per workgroup)
  const bool off_in_reg = MODE != MODE_TN && p.offsets != nullptr && p.num_groups < 64;
  int off_v = 0;
  if (off_in_reg) off_v = p.offsets[min(lane, p.num_groups)];

  // state of the current item (block-uniform)
  int bid = 0, tile_n = 0, tile_m = 0;
  int g = 0, split = 0;
  int row0 = 0, row_end = 0;  // NT/NN: rows of A and C handled by this item; TN: reduction rows [row0,row_end)
  int m0 = 0, n0 = 0;         // first output row / column of this tile
  int nk = 0, kbase = 0;      // k-tiles of this item, first k-tile
  bool valid = true;          // false: surplus item of a ragged grouped launch (no rows)
  const float* __restrict__ Ag = p.A;
  const float* __restrict__ Bg = p.B;

  // ---- loaders ----------------------------------------------------------------------------------------------
  // transposed loader (source rows k-contiguous): thread -> (row t_r + T_ROWS i, k quad t_kq)
  constexpr int KQ = BK / 4;
  constexpr int T_ROWS = NTHREADS / KQ;
  const int t_kq = tid % KQ, t_r = tid / KQ;
  // direct loader (source k-major) of an operand with R columns: quad index tid + 256 i -> (k row, column quad)
  // (direct pieces of an fp16-stored operand: one piece = 4 consecutive k of TWO adjacent columns, four 4-byte loads)
  // planes: piece q = plane q of the tile: (BK / 8) k-octets x rows granules of 16 bytes, one per thread
  static_assert(!(APL || BPL) || (BK == 16 && 2 * BM <= NTHREADS && 2 * BN <= NTHREADS), "plane pieces: one granule per thread");
  constexpr int PA = APL ? 3 : A_TRANS ? (BM + T_ROWS - 1) / T_ROWS
                             : ((A16 ? (BK / 4) * (BM / 2) : BK * (BM / 4)) + NTHREADS - 1) / NTHREADS;
  constexpr int PB = BPL ? 3 : B_TRANS ? (BN + T_ROWS - 1) / T_ROWS
                             : ((B16 ? (BK / 4) * (BN / 2) : BK * (BN / 4)) + NTHREADS - 1) / NTHREADS;
  constexpr int NP = PA + PB;  // pieces (one 16-byte load per thread each) per k-tile
  constexpr int KP = BK / 2;   // k-pairs = MFMA groups per k-tile
  static_assert(NP <= KP, "piece schedule: one load and one store slot per k-pair");

  // Per-thread source offsets (per item) and LDS offsets (once).  Out-of-range rows / columns are CLAMPED to a valid
  // address instead of branched around (the garbage they bring only reaches output rows/columns the epilogue masks);
  // reduction rows past the segment end in TN are zeroed by a select after the load.
  // Addresses: GATHER keeps one 64-bit pointer per piece; the plain GEMMs split every address into a block-uniform base
  // (a buffer descriptor in SGPRs; the k-tile advance is the instruction's scalar offset) plus a per-thread 32-bit byte
  // offset fixed for the whole k-loop, so a load is `buffer_load_dwordx4 v, v_off, s[rsrc], s_koff offen` with no vector
  // arithmetic: VALU instructions issued from the k-loop take issue slots from the MFMAs of the same SIMD (measured:
  // 46 extra v_cndmask per two k-steps cost 5-7 % on the long-K shapes; hipcc turns 64-bit per-thread pointers into
  // one v_lshl_add_u64 per load).  All offsets from the block's base stay below 2^31 (checked on the host).
  const float* pa[PA];
  const float* pb[PB];
  unsigned oa[PA], ob[PB];  // byte offsets from a_base / b_base
  const char* a_base = nullptr;
  const char* b_base = nullptr;
  int sa[PA], sb[PB];    // LDS offset of the piece (lanes beyond the tile repeat its last row: no exec mask on the stores)
  int ha[PA], hb[PB];    // F16 image: dword offset (transposed pieces, 8-byte stores) / half offset (direct pieces)
  int ka[PA], kb[PB];    // direct loader: k row of the piece
  int gy[PA], gx[PA];    // GATHER (NT/NN): grid coordinates of this thread's A rows
  unsigned m9[PA];       // GATHER (NT/NN): bit t set <=> tap t of this row reads a pixel inside the image
  int tn_tap = 0;        // GATHER (TN): the tap this N-tile belongs to (cC % BN == 0)
  // bf16x3 CHECKER: sign word of the A pieces of this thread in NT / NN launches -- A rows (= token rows of the output) that
  // lie in an odd 32-row block of the tile are stored NEGATED (see `CHECKER` below).  The transposed loader gives a wave 16
  // consecutive rows per piece, so the sign is wave-uniform: it lives in an SGPR and costs no vector register.
  uint32_t ga[PA];
  const uint32_t gapl = APL ? (((uint32_t)((tid % BM) & 32) << 10) | ((uint32_t)((tid % BM) & 32) << 26)) : 0u;  // planes: per lane
  auto blk_sign = [](int r) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(r & 32) << 26)); };  // 0x80000000 on odd blocks
  // direct pieces of the fp16 / bf16x3 images: unit idx -> (k-quad g4, column c) of an operand with R columns, column
  // fastest: a wave's 4-byte loads then cover 256 contiguous bytes of a k-row.  (Measured for the bf16x3 form: a k-quad-
  // parity-fastest order, whose 8-byte LDS stores are bank-conflict free, is 0.45 ms per step SLOWER -- two 128-byte row
  // segments per load instruction cost more than the 2-way store conflict; profiles/r05/gemm_b3_variants.txt)
  auto unit_gc = [](int idx, int R, int& g4, int& c) {
    g4 = min(idx / R, BK / 4 - 1);  // surplus units repeat the last k-quad
    c = idx % R;
  };
  // item-independent part: where a piece lands in the LDS image
#pragma unroll
  for (int i = 0; i < PA; i++) {
    gy[i] = gx[i] = 0;
    m9[i] = 0;
    ka[i] = 0;
    ha[i] = 0;
    pa[i] = nullptr;
    ga[i] = 0;
    if (APL) {  // plane piece i: granule (k-octet tid / BM, tile row tid % BM) of plane i (surplus threads repeat a granule)
      sa[i] = 0;
      ha[i] = (min(tid / BM, BK / 8 - 1) * LDA16 + tid % BM) * 4;
      continue;
    }
    if (A_TRANS) {
      const int rl = t_r + T_ROWS * i;
      sa[i] = (4 * t_kq) * LDA_S + min(rl, BM - 1);  // surplus lanes repeat row BM-1 (same data, same slot)
      ha[i] = ((t_kq >> 1) * LDA16 + min(rl, BM - 1)) * 4 + 2 * (t_kq & 1);
      ga[i] = blk_sign(min(rl, BM - 1));
    } else {
      constexpr int QR = BM / 4;
      const int idx = tid + NTHREADS * i;
      const int kk = idx / QR, cq = idx - kk * QR;
      ka[i] = min(kk, BK - 1);  // surplus lanes (kk >= BK) repeat k-row BK-1: same data into the same slot
      sa[i] = min(kk, BK - 1) * LDA_S + 4 * cq;
      if (F16) {
        // F16: a piece is FOUR CONSECUTIVE k OF ONE COLUMN (four coalesced 4-byte loads, lanes along the columns), so
        // that it lands in the fp16 image as 8 contiguous bytes like a transposed piece; a piece of four columns at
        // one k would scatter 2-byte stores 16 B apart (8-way bank conflicts: measured 13.5 vs 9.4 ms per step)
        int g4, c;
        unit_gc(idx, BM, g4, c);
        ha[i] = ((g4 >> 1) * LDA16 + c) * 4 + 2 * (g4 & 1);
        if (A16) {
          // unit = (k-octet, column pair, k-quad parity), parity fastest: the two 8-byte LDS stores of 16 consecutive
          // lanes then spread over 16 banks (2-way) instead of 8 (column pairs are 32 B apart in the image)
          const int par = idx & 1, u = idx >> 1;
          const int g8 = min(u / (BM / 2), BK / 8 - 1), cc = 2 * (u % (BM / 2));
          ha[i] = (g8 * LDA16 + cc) * 4 + 2 * par;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PB; i++) {
    kb[i] = 0;
    hb[i] = 0;
    pb[i] = nullptr;
    if (BPL) {
      sb[i] = 0;
      hb[i] = (min(tid / BN, BK / 8 - 1) * LDB16 + tid % BN) * 4;
      continue;
    }
    if (B_TRANS) {
      const int rl = t_r + T_ROWS * i;
      sb[i] = (4 * t_kq) * LDB_S + min(rl, BN - 1);
      hb[i] = ((t_kq >> 1) * LDB16 + min(rl, BN - 1)) * 4 + 2 * (t_kq & 1);
    } else {
      constexpr int QR = BN / 4;
      const int idx = tid + NTHREADS * i;
      const int kk = idx / QR, cq = idx - kk * QR;
      kb[i] = min(kk, BK - 1);
      sb[i] = min(kk, BK - 1) * LDB_S + 4 * cq;
      if (F16) {
        if (B16) {
          const int par = idx & 1, u = idx >> 1;
          const int g8 = min(u / (BN / 2), BK / 8 - 1), cc = 2 * (u % (BN / 2));
          hb[i] = (g8 * LDB16 + cc) * 4 + 2 * par;
        } else {
          int g4, c;
          unit_gc(idx, BN, g4, c);
          hb[i] = ((g4 >> 1) * LDB16 + c) * 4 + 2 * (g4 & 1);
        }
      }
    }
  }
  auto make_rsrc = [](const char* base, long valid_floats, int esize) {
    // readfirstlane: base and extent are block-uniform by construction, this makes them provably so (no waterfall
    // loop).  The extent is the operand's valid span seen from the base: a stray offset reads 0 instead of faulting.
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const long vb = valid_floats > 0 ? valid_floats * esize : 0;
    const int nrec = __builtin_amdgcn_readfirstlane((int)(vb < 0x7fffffffl ? vb : 0x7fffffffl));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nrec,
                                             0x00020000);
  };
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;

  // ---- everything that depends on the item: tile coordinates, group, row range, k-slice, source offsets, descriptors ----
  auto setup_item = [&](int item) {
    // XCD-aware remap of the linear item id (speed only): consecutive logical tiles share an XCD (hardware block b runs
    // on XCD b % 8)
    bid = item;
    {
      const int nblk = nitems;
      const int q = nblk / kNumXCD, r = nblk % kNumXCD;
      const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    tile_n = bid % ntn;
    tile_m = bid / ntn;
    valid = true;
    g = 0;
    // group / row-range / k-slice resolution
    if (MODE == MODE_TN) {
      g = blockIdx.z / p.splits;
      split = blockIdx.z - g * p.splits;
      int seg0 = 0, seg1 = p.K;
      if (p.offsets) {
        seg0 = p.offsets[g];
        seg1 = p.offsets[g + 1];
      }
      const int cnt = seg1 - seg0;
      int chunk = (cnt + p.splits - 1) / p.splits;
      chunk = (chunk + BK - 1) / BK * BK;
      // slices whose first rows sit a large power of two apart start on the same HBM channels and crawl (measured: 256
      // slices of 512 rows: 287 us, 192 or 341 slices: 121 / 162 us): keep the slice length off multiples of 128 rows
      if (p.splits > 1 && (chunk & 127) == 0) chunk += BK;
      row0 = min(seg1, seg0 + split * chunk);
      row_end = min(seg1, row0 + chunk);
      m0 = tile_m * BM;
    } else {
      split = blockIdx.z;
      if (p.offsets) {
        int base = 0;
        bool found = false;
        int o0 = off_in_reg ? __builtin_amdgcn_readlane(off_v, 0) : p.offsets[0];
        for (int gg = 0; gg < p.num_groups; gg++) {
          const int o1 = off_in_reg ? __builtin_amdgcn_readlane(off_v, gg + 1) : p.offsets[gg + 1];
          const int nt = (o1 - o0 + BM - 1) / BM;
          if (tile_m < base + nt) {
            g = gg;
            row0 = o0 + (tile_m - base) * BM;
            row_end = o1;
            found = true;
            break;
          }
          base += nt;
          o0 = o1;
        }
        if (!found) {  // surplus item of a ragged launch: no rows (a plain launch leaves, see below)
          valid = false;
          row0 = row_end = 0;
        }
      } else {
        row0 = tile_m * BM;
        row_end = p.M;
        if (row0 >= row_end) {
          valid = false;
          row0 = row_end = 0;
        }
      }
      m0 = row0;
    }
    n0 = tile_n * BN;
    Bg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.B) +
                                        (MODE == MODE_TN ? 0 : (long)g * p.strideB * (B16 ? 2 : 4)));
    kbase = 0;
    if (MODE == MODE_TN) {
      nk = (max(row_end - row0, 0) + BK - 1) / BK;
    } else {
      nk = p.K / BK;
      if (p.splits > 1) {
        kbase = split * p.kTilesPerSplit;
        nk = max(0, min(p.kTilesPerSplit, nk - kbase));
      }
      if (!valid) nk = 0;
    }
    // block-uniform by construction; readfirstlane makes it provable, so loop counters and k-offsets live in SGPRs (the
    // TN kernel kept its trip count in a spilled VGPR and drained vmcnt(0) every iteration to reload it)
    nk = __builtin_amdgcn_readfirstlane(nk);
    kbase = __builtin_amdgcn_readfirstlane(kbase);
    row0 = __builtin_amdgcn_readfirstlane(row0);
    row_end = __builtin_amdgcn_readfirstlane(row_end);
    m0 = __builtin_amdgcn_readfirstlane(m0);
    n0 = __builtin_amdgcn_readfirstlane(n0);
    g = __builtin_amdgcn_readfirstlane(g);
#pragma unroll
    for (int i = 0; i < PA; i++) {
      if (APL) {  // byte offset of this thread's granule inside an octet block pair, rows past the tile's end repeat the last
        const int rr = max(min(tid % BM, row_end - 1 - row0), 0);
        oa[i] = (unsigned)(((long)min(tid / BM, BK / 8 - 1) * p.lda + rr) * 16);
        continue;
      }
      if (A_TRANS) {
        const int rl = t_r + T_ROWS * i;
        const int r = min(row0 + min(rl, BM - 1), row_end - 1);
        if (GATHER) {
          const unsigned t = fast_div((unsigned)r, (unsigned)p.rW, p.mRW);
          gx[i] = r - (int)t * p.rW;
          const unsigned b = fast_div(t, (unsigned)p.rH, p.mRH);
          gy[i] = (int)t - (int)b * p.rH;
          pa[i] = Ag + (long)b * p.sH * p.sW * p.cC + 4 * t_kq;
          // Implicit im2col without per-load coordinate arithmetic: source pixel = (ay + uy(dy) - 1, ax + ux(dx) - 1) with a
          // per-row part (ay, ax) and a per-tap part that is the same for every lane:
          //   forward / weight gradient (cT = 0): ay = gy * stride,          uy(d) = d
          //   input gradient, stride 1:           ay = gy,                   uy(d) = 2 - d
          //   input gradient, stride 2:           ay = (gy + 1) >> 1,        uy(d) = 1 - d / 2   (taps of the wrong parity
          //                                                                                      are masked in m9)
          // so the address is  [base - (sW + 1) * cC]  +  lane offset (fixed)  +  scalar tap offset, and a lane whose tap
          // falls outside the image swaps its offset for one beyond the descriptor's extent: the load returns 0.
          const int ay = p.cT ? (p.cS == 2 ? (gy[i] + 1) >> 1 : gy[i]) : gy[i] * p.cS;
          const int ax = p.cT ? (p.cS == 2 ? (gx[i] + 1) >> 1 : gx[i]) : gx[i] * p.cS;
          oa[i] = (unsigned)(((((long)b * p.sH + ay) * p.sW + ax) * p.cC + 4 * t_kq) * 4);
          m9[i] = 0;
#pragma unroll
          for (int tp = 0; tp < 9; tp++) {
            const int sy = gather_coord(gy[i], tp / 3, p.cS, p.cT, p.sH);
            const int sx = gather_coord(gx[i], tp % 3, p.cS, p.cT, p.sW);
            if (sy >= 0 && sx >= 0) m9[i] |= 1u << tp;
          }
        } else {
          pa[i] = Ag + (long)r * p.lda + 4 * t_kq + (long)kbase * BK;
          oa[i] = (unsigned)(((long)(r - row0) * p.lda + 4 * t_kq) * EA);
        }
      } else {
        constexpr int QR = BM / 4;
        const int idx = tid + NTHREADS * i;
        const int kk = idx / QR, cq = idx - kk * QR;
        pa[i] = Ag + min(m0 + 4 * cq, p.M - 4);
        oa[i] = (unsigned)(((long)min(kk, BK - 1) * p.lda + min(m0 + 4 * cq, p.M - 4)) * 4);  // kk >= BK: lane without an element
        if (F16) {
          int g4, c;
          unit_gc(idx, BM, g4, c);
          oa[i] = (unsigned)(((long)(4 * g4) * p.lda + min(m0 + c, p.M - 1)) * 4);
          if (A16) {
            const int par = idx & 1, u = idx >> 1;
            const int g8 = min(u / (BM / 2), BK / 8 - 1), cc = 2 * (u % (BM / 2));
            oa[i] = (unsigned)(((long)(8 * g8 + 4 * par) * p.lda + min(m0 + cc, p.M - 2)) * 2);
          }
        }
      }
    }
    if (APL) a_base = reinterpret_cast<const char*>(Ag) + (long)row0 * 16;
    else if (MODE == MODE_TN) a_base = reinterpret_cast<const char*>(Ag) + (long)row0 * p.lda * EA;
    else if (GATHER) a_base = reinterpret_cast<const char*>(Ag) - (long)(p.sW + 1) * p.cC * 4;
    else a_base = reinterpret_cast<const char*>(Ag) + ((long)row0 * p.lda + (long)kbase * BK) * EA;
#pragma unroll
    for (int i = 0; i < PB; i++) {
      if (BPL) {
        const int nn = max(min(tid % BN, p.N - 1 - n0), 0);
        ob[i] = (unsigned)(((long)min(tid / BN, BK / 8 - 1) * p.ldb + nn) * 16);
        continue;
      }
      if (B_TRANS) {
        const int rl = t_r + T_ROWS * i;
        const int n = min(n0 + min(rl, BN - 1), p.N - 1);
        pb[i] = Bg + (long)n * p.ldb + 4 * t_kq + (long)kbase * BK;
        ob[i] = (unsigned)(((long)(n - n0) * p.ldb + 4 * t_kq) * EB);
      } else {
        constexpr int QR = BN / 4;
        const int idx = tid + NTHREADS * i;
        const int kk = idx / QR, cq = idx - kk * QR;
        const int nc = min(n0 + 4 * cq, p.N - 4);
        if (MODE == MODE_TN) {
          if (GATHER) {
            tn_tap = n0 / p.cC;
            pb[i] = Bg + (nc - tn_tap * p.cC);  // channel offset inside the tap; the row part is added per load
            // GATHER == 2 (rW % BK == 0: a k-tile of BK output positions lies inside one image row, so its first source
            // pixel is the same for every lane): lane part = this lane's position inside the k-tile + channel offset
            ob[i] = (unsigned)(((long)min(kk, BK - 1) * p.cS * p.cC + (nc - tn_tap * p.cC)) * 4);
          } else {
            pb[i] = Bg + nc;
            ob[i] = (unsigned)(((long)min(kk, BK - 1) * p.ldb + nc) * 4);
          }
        } else {  // NN: B[K,N] rows are k
          pb[i] = Bg + (long)min(kk, BK - 1) * p.ldb + nc + (GATHER ? 0 : (long)kbase * BK * p.ldb);
          ob[i] = (unsigned)(((long)min(kk, BK - 1) * p.ldb + nc) * 4);
        }
        if (F16) {
          if (B16) {
            const int par = idx & 1, u = idx >> 1;
            const int g8 = min(u / (BN / 2), BK / 8 - 1), cc = 2 * (u % (BN / 2));
            ob[i] = (unsigned)(((long)(8 * g8 + 4 * par) * p.ldb + min(n0 + cc, p.N - 2)) * 2);
          } else {
            int g4, c;
            unit_gc(idx, BN, g4, c);
            ob[i] = (unsigned)(((long)(4 * g4) * p.ldb + min(n0 + c, p.N - 1)) * 4);
            if (MODE == MODE_TN && GATHER == 2) {
              // unit = four consecutive output positions (k) of one channel of this N-tile's tap: positions are cS * cC
              // floats apart in the gathered tensor; kb = the unit's first position inside the k-tile (x-range test)
              kb[i] = 4 * g4;
              ob[i] = (unsigned)(((long)(4 * g4) * p.cS * p.cC + (min(n0 + c, p.N - 1) - tn_tap * p.cC)) * 4);
            }
          }
        }
      }
    }
    if (BPL) b_base = reinterpret_cast<const char*>(p.B) + ((long)g * p.strideB + n0) * 16;
    else if (MODE == MODE_NT) b_base = reinterpret_cast<const char*>(Bg) + ((long)n0 * p.ldb + (long)kbase * BK) * EB;
    else if (MODE == MODE_NN) b_base = reinterpret_cast<const char*>(Bg) + (GATHER ? 0 : (long)kbase * BK * p.ldb) * EB;
    else if (GATHER == 2) b_base = reinterpret_cast<const char*>(Bg) - (long)(p.sW + 1) * p.cC * 4;
    else b_base = reinterpret_cast<const char*>(Bg) + (long)row0 * p.ldb * EB;
    long a_valid, b_valid;
    if (MODE == MODE_TN) {
      a_valid = (long)(row_end - 1 - row0) * p.lda + p.M;
      b_valid = (long)(row_end - 1 - row0) * p.ldb + p.N;
      if (row_end <= row0) a_valid = b_valid = 0;
      if (GATHER == 2) b_valid = (long)(p.K / (p.rH * p.rW)) * p.sH * p.sW * p.cC + (long)(p.sW + 1) * p.cC;
    } else if (GATHER) {
      // gathered tensor: (M / (rH * rW)) images of sH x sW x cC, seen from the shifted base; NN weights: cC rows of ldb
      a_valid = (long)(p.M / (p.rH * p.rW)) * p.sH * p.sW * p.cC + (long)(p.sW + 1) * p.cC;
      b_valid = MODE == MODE_NT ? (long)(p.N - 1 - n0) * p.ldb + (p.K - (long)kbase * BK) : (long)p.cC * p.ldb;
    } else {
      a_valid = (long)(row_end - 1 - row0) * p.lda + (p.K - (long)kbase * BK);
      b_valid = MODE == MODE_NT ? (long)(p.N - 1 - n0) * p.ldb + (p.K - (long)kbase * BK)
                                : (long)(p.K - (long)kbase * BK - 1) * p.ldb + p.N;
      if (!valid) a_valid = 0;
    }
    // planes: the extent is what is left of the three planes behind the base (in 4-byte units)
    if (APL) a_valid = valid ? ((long)3 * (p.K >> 3) * p.lda - row0) * 4 : 0;
    if (BPL) b_valid = ((long)3 * (p.K >> 3) * p.ldb - ((long)g * p.strideB + n0)) * 4;
    a_rsrc = make_rsrc(a_base, a_valid, APL ? 4 : EA);
    b_rsrc = make_rsrc(b_base, b_valid, BPL ? 4 : EB);
  };
  int item = blockIdx.x;
  setup_item(item);
  if (!valid) return;  // surplus block of a ragged plain launch (all slices of it leave: no ticket is ever drawn)
  // piece q in [0, NP): q < PA -> A piece q, else B piece q - PA.  tail == false (the bulk of the k-loop): tile kt is a
  // complete tile strictly before the last one, so no clamp and no select is issued; tail == true: the last steps, where
  // kt may be clamped and TN reduction rows past the segment end are zeroed.
  auto ldg = [&](const __amdgpu_buffer_rsrc_t& rs, long soff, unsigned voff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
  };
  auto ldg1 = [&](const __amdgpu_buffer_rsrc_t& rs, long soff, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
  };
  auto ldg2 = [&](const __amdgpu_buffer_rsrc_t& rs, long soff, unsigned voff) {  // 8 bytes = 4 stored halves
    typedef unsigned u32x2l __attribute__((ext_vector_type(2)));
    // bit_cast, not an implicit conversion: whatever 64-bit type the builtin returns (a scalar would be SPLAT into a
    // 2-vector by a plain assignment, which silently narrows the load to one dword)
    const u32x2l v = __builtin_bit_cast(u32x2l, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
    // (through scalar temporaries: hipcc 7.2 evaluates __builtin_bit_cast applied DIRECTLY to a vector-element
    // expression `v[1]` as element 0 -- the load was narrowed to one dword and both words of the piece were the same)
    const unsigned w0 = v[0], w1 = v[1];
    return f32x4{__builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1), 0.f, 0.f};
  };
  auto load_piece = [&](f32x4 (&ra)[PA], f32x4 (&rb)[PB], int q, int kt, bool tail) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (APL && q < PA) {  // plane q, k-octets 2 (kbase + kt) and + 1: scalar offset, one 16-byte granule per thread
      ra[q] = ldg(a_rsrc, ((long)q * (p.K >> 3) + 2 * (kt + kbase)) * p.lda * 16, oa[q]);
      return;
    }
    if (BPL && q >= PA) {
      rb[q - PA] = ldg(b_rsrc, ((long)(q - PA) * (p.K >> 3) + 2 * (kt + kbase)) * p.ldb * 16, ob[q - PA]);
      return;
    }
    if (q < PA) {
      if (MODE == MODE_TN && F16) {
        // rows past the segment end lie beyond the descriptor's extent and read 0: no clamp, no select, bulk or tail
#pragma unroll
        for (int kk = 0; kk < 4; kk++) ra[q][kk] = ldg1(a_rsrc, ((long)kt * BK + kk) * p.lda * EA, oa[q]);
      } else if (MODE == MODE_TN) {
        if (!tail) {
          ra[q] = ldg(a_rsrc, (long)kt * BK * p.lda * 4, oa[q]);
        } else {
          const int kr = row0 + kt * BK + ka[q];
          const int krc = min(kr, row_end - 1);
          f32x4 v = *reinterpret_cast<const f32x4*>(pa[q] + (long)krc * p.lda);
          ra[q] = kr < row_end ? v : zero4;
        }
      } else if (GATHER) {
        const int kg = kt + kbase;
        const int tap = (int)(((unsigned)kg * p.cInv) >> 16);  // k-tile -> tap (uniform), channel offset inside it
        const int c0 = kg * BK - tap * p.cC;
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int uy = (p.cU >> (4 * dy)) & 15, ux = (p.cU >> (4 * dx)) & 15;  // branch-free (see GemmParams::cU)
        const long soff = ((long)(uy * p.sW + ux) * p.cC + c0) * 4;           // scalar
        const unsigned voff = (m9[q] & (1u << tap)) ? oa[q] : 0x7fff0000u;   // 3 VALU per load
        ra[q] = ldg(a_rsrc, soff, voff);
      } else if (A16) {
        ra[q] = ldg2(a_rsrc, (long)kt * BK * 2, oa[q]);
      } else {
        ra[q] = ldg(a_rsrc, (long)kt * BK * 4, oa[q]);
      }
    } else {
      const int i = q - PA;
      if (MODE == MODE_TN) {
        if (GATHER == 2) {
          // the k-tile's first output position and its source pixel on the scalar unit; per lane only the x-range test
          const int r0 = row0 + kt * BK;
          const unsigned t = fast_div((unsigned)r0, (unsigned)p.rW, p.mRW);
          const int oxb = r0 - (int)t * p.rW;
          const unsigned b = fast_div(t, (unsigned)p.rH, p.mRH);
          const int oy = (int)t - (int)b * p.rH;
          const int dy = tn_tap / 3, dx = tn_tap - 3 * dy;
          const int syp = oy * p.cS + dy;  // source row + 1 (the base is shifted by one row and one pixel)
          const bool yok = syp >= 1 && syp <= p.sH;
          const long soff = ((((long)b * p.sH + syp) * p.sW + oxb * p.cS + dx) * p.cC) * 4;
          if (F16) {  // four positions of one channel: one 4-byte load each, masked on its own x
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
              const int sxk = (oxb + kb[i] + kk) * p.cS + dx - 1;
              rb[i][kk] = ldg1(b_rsrc, soff + (long)kk * p.cS * p.cC * 4, (yok && sxk >= 0 && sxk < p.sW) ? ob[i] : 0x7fff0000u);
            }
          } else {
          const int sx = (oxb + kb[i]) * p.cS + dx - 1;
          rb[i] = ldg(b_rsrc, soff, (yok && sx >= 0 && sx < p.sW) ? ob[i] : 0x7fff0000u);
          }
        } else if (GATHER) {
          const int kr = row0 + kt * BK + kb[i];
          const int krc = min(kr, row_end - 1);
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
        } else if (F16) {
#pragma unroll
          for (int kk = 0; kk < 4; kk++) rb[i][kk] = ldg1(b_rsrc, ((long)kt * BK + kk) * p.ldb * EB, ob[i]);
        } else if (!tail) {
          rb[i] = ldg(b_rsrc, (long)kt * BK * p.ldb * 4, ob[i]);
        } else {
          const int kr = row0 + kt * BK + kb[i];
          const int krc = min(kr, row_end - 1);
          f32x4 v = *reinterpret_cast<const f32x4*>(pb[i] + (long)krc * p.ldb);
          rb[i] = kr < row_end ? v : zero4;
        }
      } else if (MODE == MODE_NT) {
        if (B16) rb[i] = ldg2(b_rsrc, (long)kt * BK * 2, ob[i]);  // four stored halves of one weight row
        else rb[i] = ldg(b_rsrc, (long)kt * BK * 4, ob[i]);
      } else if (GATHER) {
        // NN gather (input gradient): B row k = (tap, co) lives at W[co][tap][:]  (ldb = 9 * Cin, + tap * N columns)
        const int kg = kt + kbase;
        const int tap = (int)(((unsigned)kg * p.cInv) >> 16);
        const int c0 = kg * BK - tap * p.cC;
        if (F16) {  // four consecutive k (output channels of the tap) of one column
#pragma unroll
          for (int kk = 0; kk < 4; kk++)
            rb[i][kk] = ldg1(b_rsrc, ((long)(c0 + kk) * p.ldb + (long)tap * p.N) * 4, ob[i]);
        } else {
        rb[i] = ldg(b_rsrc, ((long)c0 * p.ldb + (long)tap * p.N) * 4, ob[i]);
        }
      } else if (F16) {  // NN: B rows are k
#pragma unroll
        for (int kk = 0; kk < 4; kk++) rb[i][kk] = ldg1(b_rsrc, ((long)kt * BK + kk) * p.ldb * EB, ob[i]);
      } else {
        rb[i] = ldg(b_rsrc, (long)kt * BK * p.ldb * 4, ob[i]);
      }
    }
  };
  // TN by-product: column sums of A (= the bias gradient next to the weight gradient dY^T X).  Only the workgroups of
  // the first column tile accumulate (uniform branch around four vector adds per piece: no loads inside it); every
  // piece passes through store_piece exactly once, already masked / zeroed where the tile does not exist.
  // (CSUM is a template flag: hipcc turns the branch into predicated adds, which the plain TN launches must not pay)
  const bool do_cs = CSUM && MODE == MODE_TN && p.csum != nullptr && tile_n == 0;
  f32x4 csa[(CSUM && !F16) ? PA : 1];  // fp32 image: a piece = four COLUMNS at one k -> four running sums per piece
#pragma unroll
  for (int i = 0; i < ((CSUM && !F16) ? PA : 1); i++) csa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fp16 / bf16x3 images: a piece = four consecutive k of ONE column, so one running sum per piece is enough (the four
  // lanes-of-a-vector form held PA x 4 registers: up to 40 spilled VGPRs in the k-step-64 weight-gradient kernels)
  float cs1[(CSUM && F16) ? PA : 1];
#pragma unroll
  for (int i = 0; i < ((CSUM && F16) ? PA : 1); i++) cs1[i] = 0.f;
  // live == false (uniform, tail steps only): the tile does not exist (index >= nk) and zeros are stored instead -- the
  // k-loop runs an even number of steps without a branch between them, so the step after the last tile of an odd nk
  // multiplies this all-zero stage
  auto store_piece = [&](const f32x4 (&ra_)[PA], const f32x4 (&rb_)[PB], int q, int buf, bool live, bool negate = false) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[PA], rb[PB];
    if (q < PA) ra[q] = live ? ra_[q] : z4;
    else rb[q - PA] = live ? rb_[q - PA] : z4;
    if (CSUM && !F16 && q < PA && do_cs) csa[(CSUM && !F16) ? q : 0] += ra[q];
    if (CSUM && F16 && q < PA && do_cs) cs1[(CSUM && F16) ? q : 0] += (ra[q][0] + ra[q][1]) + (ra[q][2] + ra[q][3]);
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    auto pack2 = [](float x, float y) {  // round-to-nearest-even, like a torch .half() cast
      return __builtin_bit_cast(uint32_t, f16x2{(_Float16)x, (_Float16)y});
    };
    auto bits = [](float x) { return __builtin_bit_cast(uint32_t, x); };
    // a direct piece of an fp16-stored operand holds w[k] = (column c, column c + 1) at four consecutive k: regroup into
    // the two columns' k-quads (v_perm_b32: low halves / high halves of two words)
    auto lo2 = [&](float w0, float w1) { return __builtin_amdgcn_perm(bits(w1), bits(w0), 0x05040100u); };
    auto hi2 = [&](float w0, float w1) { return __builtin_amdgcn_perm(bits(w1), bits(w0), 0x07060302u); };
    // B3: exact three-way bf16 split of the four k-values of a piece -> 8 bytes into each of the three planes
    auto cvt2 = [](float lo, float hi) {  // v_cvt_pk_bf16_f32: two round-to-nearest-even bf16 in one dword
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
    };
    auto split3 = [&](const f32x4& v, uint32_t* dst, int plane, uint32_t sgn) {  // sgn: 0 or 0x80008000 (both halves of a pair)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      auto lo_f = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
      auto hi_f = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); };
#ifdef SM3_ABL_NOCVT  // ablation: the three 8-byte stores without the conversion arithmetic (wrong results, timing only)
      *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1])};
      *reinterpret_cast<u32x2*>(dst + plane) = u32x2{__builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])};
      *reinterpret_cast<u32x2*>(dst + 2 * plane) = u32x2{__builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2])};
      return;
#endif
      const uint32_t h0 = cvt2(v[0], v[1]), h1 = cvt2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0 ^ sgn, h1 ^ sgn};
      const float r0 = v[0] - lo_f(h0), r1 = v[1] - hi_f(h0), r2 = v[2] - lo_f(h1), r3 = v[3] - hi_f(h1);  // exact
      const uint32_t m0 = cvt2(r0, r1), m1 = cvt2(r2, r3);
      *reinterpret_cast<u32x2*>(dst + plane) = u32x2{m0 ^ sgn, m1 ^ sgn};
      const float s0 = r0 - lo_f(m0), s1 = r1 - hi_f(m0), s2 = r2 - lo_f(m1), s3 = r3 - hi_f(m1);          // exact
      *reinterpret_cast<u32x2*>(dst + 2 * plane) = u32x2{cvt2(s0, s1) ^ sgn, cvt2(s2, s3) ^ sgn};
    };
    if (B3) {
      constexpr bool CHK = SM3_B3_SIGNED == 2 && MODE != MODE_TN;
      typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
      if (APL && q < PA) {  // a granule of plane q as it is (sign of the row block applied to all eight pieces)
        const float t0 = ra[q][0], t1 = ra[q][1], t2 = ra[q][2], t3 = ra[q][3];
        const uint32_t sg = CHK ? gapl : 0u;
        *reinterpret_cast<u32x4s*>(Aw + buf * A_STG + q * A_ST16 + ha[q]) =
            u32x4s{bits(t0) ^ sg, bits(t1) ^ sg, bits(t2) ^ sg, bits(t3) ^ sg};
      } else if (BPL && q >= PA) {
        const int i = q - PA;
        const float t0 = rb[i][0], t1 = rb[i][1], t2 = rb[i][2], t3 = rb[i][3];
        *reinterpret_cast<u32x4s*>(Bw + buf * B_STG + i * B_ST16 + hb[i]) = u32x4s{bits(t0), bits(t1), bits(t2), bits(t3)};
      } else if (q < PA) {
        f32x4 va = ra[q];
        if (CHK) {  // -x splits into exactly the negated pieces of x (round-to-nearest-even is symmetric): four v_xor with an
                    // SGPR operand before the split instead of six on the packed planes after it
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float t = va[e];  // (a scalar temporary: see ldg2 on __builtin_bit_cast of a vector element)
            va[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, t) ^ ga[q]);
          }
        }
        split3(va, Aw + buf * A_STG + ha[q], A_ST16, (!CHK && negate) ? 0x80008000u : 0u);
      } else {
        split3(rb[q - PA], Bw + buf * B_STG + hb[q - PA], B_ST16, 0u);
      }
    } else if (q < PA) {
      if (F16 && A16 && A_TRANS) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(Aw + buf * A_STG + ha[q]) = u32x2{bits(ra[q][0]), bits(ra[q][1])};  // already halves
      } else if (F16 && A16) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        uint32_t* d = Aw + buf * A_STG + ha[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{lo2(ra[q][0], ra[q][1]), lo2(ra[q][2], ra[q][3])};
        *reinterpret_cast<u32x2*>(d + 4) = u32x2{hi2(ra[q][0], ra[q][1]), hi2(ra[q][2], ra[q][3])};
      } else if (F16) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        // every F16 piece is four consecutive k of one row / column -> 8 contiguous bytes
        *reinterpret_cast<u32x2*>(Aw + buf * A_STG + ha[q]) = u32x2{pack2(ra[q][0], ra[q][1]), pack2(ra[q][2], ra[q][3])};
      } else {
        float* a_s = As + buf * A_STAGE;
        if (A_TRANS) {
#pragma unroll
          for (int j = 0; j < 4; j++) a_s[sa[q] + j * LDA_S] = ra[q][j];
        } else {
          *reinterpret_cast<f32x4*>(a_s + sa[q]) = ra[q];
        }
      }
    } else {
      const int i = q - PA;
      if (F16 && B16 && B_TRANS) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(Bw + buf * B_STG + hb[i]) = u32x2{bits(rb[i][0]), bits(rb[i][1])};  // already halves
      } else if (F16 && B16) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        uint32_t* d = Bw + buf * B_STG + hb[i];
        *reinterpret_cast<u32x2*>(d) = u32x2{lo2(rb[i][0], rb[i][1]), lo2(rb[i][2], rb[i][3])};
        *reinterpret_cast<u32x2*>(d + 4) = u32x2{hi2(rb[i][0], rb[i][1]), hi2(rb[i][2], rb[i][3])};
      } else if (F16) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(Bw + buf * B_STG + hb[i]) = u32x2{pack2(rb[i][0], rb[i][1]), pack2(rb[i][2], rb[i][3])};
      } else {
        float* b_s = Bs + buf * B_STAGE;
        if (B_TRANS) {
#pragma unroll
          for (int j = 0; j < 4; j++) b_s[sb[i] + j * LDB_S] = rb[i][j];
        } else {
          *reinterpret_cast<f32x4*>(b_s + sb[i]) = rb[i];
        }
      }
    }
  };

  f32x16 acc[TI][TJ];
  // bf16x3, SM3_B3_SIGNED: v_mfma_f32_32x32x16_bf16 adds its 16 products to the accumulator with a truncation TOWARDS MINUS
  // INFINITY (measured, scripts/probes/b3_bias.py: mean signed error -0.1 ulp at K = 384, -0.4 ulp at K = 3072, the same
  // sign for positive and negative results; the native fp32 instruction is unbiased) -- harmless per element, but coherent
  // over the ~1e5 tokens a bias / LayerNorm gradient sums.  Odd k-tiles therefore store the NEGATED A pieces and accumulate
  // into a second set: it holds -S_odd with the same downward bias, and S = S_even - (-S_odd) cancels the two biases.
  // Cost (same box, training step): 16.83 ms against 16.36 ms with one set -- 230 VGPRs instead of 166 at 128x128, two
  // workgroups per CU instead of three.  Benefit: mean signed error +0.002 / +0.03 ulp (K = 384 / 3072; one set: -0.12 / -0.38),
  // mean |error| 0.63 / 1.5 ulp (one set 1.2 / 2.8, the native fp32 instruction 1.5 / 4.2); worst full-size gradient
  // 2.9e-4 instead of 6-8e-4 of the 1e-3 bar (profiles/r05/gemm_b3_bias.txt).  The alternative that keeps one set --
  // flipping the sign of the whole accumulation (64 v_xor in the shadow of a body's fragment reads, 161 VGPRs, three
  // workgroups per CU) -- measured slower than both, every four k-tiles and every k-tile alike: same box 17.61 ms (flip per
  // tile) / 17.19 (two sets) / 16.77 (one set), and its |error| stays the one-set form's (1.17 ulp at K = 384): the xors
  // take the issue slots of the wave's own MFMAs.
  constexpr bool SIGNED = B3 && SM3_B3_SIGNED == 1;
  // CHECKER (round 6, the default): ONE accumulator set; the bias is cancelled ACROSS the token rows of the output instead of
  // inside every element.  What made the bias visible was its coherence over the ~1e5 token rows that a bias / LayerNorm
  // gradient sums (a -0.1 ulp mean over 131 072 rows is 36 sigma of the rows' rounding noise); everything else sums a few
  // hundred or thousand terms, where it is 1e-7 relative.  So in the NT / NN launches (output rows = tokens) the A rows of the
  // odd 32-row blocks of a tile are stored NEGATED: those blocks accumulate -S with the same downward truncation, and after
  // the sign is restored (once, after the k-loop) their error has mean +b where the even blocks have -b -- a sum over token
  // rows adds equal numbers of both.  TN launches (weight gradients: the token sum is the k-loop itself, split over up to 256
  // slices of a few hundred rows each) need nothing.  Cost: 8 v_xor per body in NT / NN (SGPR sign word), none in TN.
  // 168 VGPRs at 128x128: three workgroups per CU (the two-set form: 230, two).  Per-element mean |error| is the one-set
  // form's (1.2 / 2.8 ulp at K = 384 / 3072; two sets 0.63 / 1.5; the native fp32 instruction 1.5 / 4.2):
  // tests/test_gemm_gpu.py guardrail, scripts/probes/b3_bias.py for the column sums.
  constexpr bool CHECKER = B3 && SM3_B3_SIGNED == 2 && MODE != MODE_TN;
  f32x16 accn[SIGNED ? TI : 1][SIGNED ? TJ : 1];

  // Two register sets: while tile kt is multiplied out of LDS, the global loads of tile kt+2 are ISSUED into one set
  // (first k-pairs) and tile kt+1 -- loaded one iteration earlier, long landed -- is WRITTEN to the other LDS buffer
  // from the other set (last k-pairs), one piece between each group of MFMAs.  The matrix pipe never waits for address
  // arithmetic, a vmcnt drain or the LDS write pass; one barrier per k-step remains.
  f32x4 sa0[PA], sb0[PB], sa1[PA], sb1[PB];

  auto k_step = [&](f32x4 (&ca)[PA], f32x4 (&cb)[PB], f32x4 (&na)[PA], f32x4 (&nb)[PB], int kt, bool tail) {
    // ca/cb hold tile kt+1 (to be stored), na/nb receive tile kt+2
    const int buf = kt & 1;
    // branch-free on purpose: a conditional around a load makes hipcc drain vmcnt(0) at the join, serialising the
    // pipeline.  Past the end (tail steps) the last tile is simply re-loaded; a tile index >= nk is stored as zeros.
    const int kt_load = tail ? min(kt + 2, nk - 1) : kt + 2;
    const bool live = tail ? kt + 1 < nk : true;
    const float* a_s = As + buf * A_STAGE + wm0 + l31;
    const float* b_s = Bs + buf * B_STAGE + wn0 + l31;
    float a[TI], b[TJ];
#pragma unroll
    for (int i = 0; i < TI; i++) a[i] = a_s[lh * LDA_S + 32 * i];
#pragma unroll
    for (int j = 0; j < TJ; j++) b[j] = b_s[lh * LDB_S + 32 * j];
#pragma unroll
    for (int kk = 0; kk < KP; kk++) {
      float xa[TI], xb[TJ];
#pragma unroll
      for (int i = 0; i < TI; i++) xa[i] = 0.f;
#pragma unroll
      for (int j = 0; j < TJ; j++) xb[j] = 0.f;
      if (kk + 1 < KP) {  // fragment reads of the NEXT k-pair go out before this k-pair's MFMAs
        const int krow = 2 * (kk + 1) + lh;
#pragma unroll
        for (int i = 0; i < TI; i++) xa[i] = a_s[krow * LDA_S + 32 * i];
#pragma unroll
        for (int j = 0; j < TJ; j++) xb[j] = b_s[krow * LDB_S + 32 * j];
      }
      if (kk < NP) load_piece(na, nb, kk, kt_load, tail);
      if (kk >= KP - NP) store_piece(ca, cb, kk - (KP - NP), buf ^ 1, live);
      __builtin_amdgcn_sched_barrier(0);  // everything above is issued before this k-pair's MFMAs
      // operands swapped on purpose: D = (B fragment) x (A fragment) = the TRANSPOSED 32x32 tile, so that each lane
      // ends up with 4 consecutive output COLUMNS of one row -> 16-byte epilogue loads/stores
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TI; i++) a[i] = xa[i];
#pragma unroll
      for (int j = 0; j < TJ; j++) b[j] = xb[j];
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  // F16 (mixed precision, the reference's `fp16 = dict(loss_scale='dynamic')` configs): same tiles and loaders; operands
  // stored as fp32 in HBM (weights, the residual stream, C-wide gradients) are rounded to fp16 when a piece is stored to
  // LDS (see Aw above), operands the AMP data path already stores as fp16 (IO_A16 / IO_B16: LayerNorm output, dispatched
  // expert inputs, GELU output, the 4C-wide gradient) arrive as halves and are stored as they are; the fragments (8
  // k-values per lane and operand = 4 dwords) go from LDS straight into v_mfma_f32_32x32x16_f16 with fp32 accumulation --
  // 16x the matrix rate of the fp32 instruction, so the loop is bound by the operand stream, not by the matrix pipe.  No
  // cast kernels exist.  A and B use the same (lane-half, element) -> k assignment, so the sum over k is complete
  // whatever order the hardware walks it in.
  auto k_step16 = [&](f32x4 (&ca)[PA], f32x4 (&cb)[PB], f32x4 (&na)[PA], f32x4 (&nb)[PB], int kt, bool tail) {
    const int buf = kt & 1;
    const int kt_load = tail ? min(kt + 2, nk - 1) : kt + 2;
    const bool live = tail ? kt + 1 < nk : true;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t* a_w = Aw + buf * A_STG + (wm0 + l31) * 4;
    const uint32_t* b_w = Bw + buf * B_STG + (wn0 + l31) * 4;
#ifndef SM3_ABL_NOLOAD  // ablation builds (sm3det_amd/build.py VARIANTS): measurement aids, never the default library
#pragma unroll
    for (int q = 0; q < NP; q++) load_piece(na, nb, q, kt_load, tail);
#endif
    // fragments of sub-step ks + 1 are requested before the MFMAs of sub-step ks (one LDS round trip per step instead of
    // one per sub-step); the LDS writes of tile kt + 1 go out after the first sub-step's reads
    constexpr int KS = BK / 16;
    f16x8 a[2][TI], b[2][TJ];
    auto frag = [&](int ks, f16x8 (&fa)[TI], f16x8 (&fb)[TJ]) {
#pragma unroll
      for (int i = 0; i < TI; i++)
        fa[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(a_w + ((2 * ks + lh) * LDA16 + 32 * i) * 4));
#pragma unroll
      for (int j = 0; j < TJ; j++)
        fb[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(b_w + ((2 * ks + lh) * LDB16 + 32 * j) * 4));
    };
    frag(0, a[0], b[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      if (ks + 1 < KS) frag(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
#ifndef SM3_ABL_NOSTORE
      if (ks == 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) store_piece(ca, cb, q, buf ^ 1, live);
      }
#endif
#ifdef SM3_ABL_NOMFMA
#pragma unroll
      for (int i = 0; i < TI; i++) asm volatile("" ::"v"(a[ks & 1][i]));
#pragma unroll
      for (int j = 0; j < TJ; j++) asm volatile("" ::"v"(b[ks & 1][j]));
#else
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0);
#endif
    }
    __syncthreads();
  };
  // B3 (bf16x3, see the LDS image above).  One body = the 16 k of one tile: the fragments of its three planes are read at
  // the top (12 ds_read_b128 at 128x128), then six products a_s . b_t (s + t <= 2) of TI x TJ MFMAs each, a_1 . b_1 first
  // (the smallest terms lead).  Tile t + 1 is written to the other stage -- split into its three bf16 planes: 22 VALU
  // instructions and three 8-byte stores per piece, one piece per product slot, interleaved with that slot's MFMAs -- and the
  // registers a piece came from are refilled at once with tile t + 3 (two bodies of prefetch on two register sets: a body
  // is 768 matrix-pipe cycles, a third of the fp32 form's).  230 VGPRs at 128x128 with the second accumulator set (below): two
  // workgroups per CU (166 VGPRs and three with one set).
  // Measured alternatives, all within +-3 % of this form or worse (profiles/r05/gemm_b3_variants.txt): fragment reads one
  // tile ahead with plane 0 double-buffered (207 VGPRs, two workgroups per CU), compiler-ordered body, no interleave hint.
  // Phase ablations (profiles/r05/gemm_b3_ablations.txt): the conversion arithmetic is free beside the bf16 MFMAs; the
  // family is latency-bound -- global loads 2.0 ms, LDS stores 1.8 ms, epilogues 1.2 ms, MFMAs 2.5 ms of 12.3 ms add up.
  bf16x8 fa[3][TI], fb[3][TJ];
  typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
  auto rd_a = [&](int buf, int s, int dst) {
    const uint32_t* a_w = Aw + buf * A_STG + s * A_ST16 + (lh * LDA16 + wm0 + l31) * 4;
#pragma unroll
    for (int i = 0; i < TI; i++) fa[dst][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(a_w + 32 * i * 4));
  };
  auto rd_b = [&](int buf, int s, int dst) {
    const uint32_t* b_w = Bw + buf * B_STG + s * B_ST16 + (lh * LDB16 + wn0 + l31) * 4;
#pragma unroll
    for (int j = 0; j < TJ; j++) fb[dst][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(b_w + 32 * j * 4));
  };
  // operands swapped as in the fp32 form: D = (B fragment) x (A fragment) = the transposed 32x32 tile
  auto prod = [&](int sa_, int sb_, int par) {
#ifdef SM3_ABL_NOMFMA
#pragma unroll
    for (int i = 0; i < TI; i++) asm volatile("" ::"v"(fa[sa_][i]));
#pragma unroll
    for (int j = 0; j < TJ; j++) asm volatile("" ::"v"(fb[sb_][j]));
#else
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        if (SIGNED && par)
          accn[SIGNED ? i : 0][SIGNED ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              fb[sb_][j], fa[sa_][i], accn[SIGNED ? i : 0][SIGNED ? j : 0], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[sb_][j], fa[sa_][i], acc[i][j], 0, 0, 0);
      }
#endif
  };
  static_assert(!B3 || NP <= 6, "bf16x3 body: one piece per product slot, six slots (fp32 operands use four)");
  auto b3_piece = [&](f32x4 (&ra)[PA], f32x4 (&rb)[PB], int q, int stage, bool live, int kt_load, bool tail, bool neg) {
    if (q >= NP) return;
#ifndef SM3_ABL_NOSTORE
    store_piece(ra, rb, q, stage, live, neg);
#endif
#ifndef SM3_ABL_NOLOAD
    load_piece(ra, rb, q, kt_load, tail);
#endif
  };
  // interleave hint for one product slot: after each MFMA a share of the slot's vector-ALU work and one LDS operation
  auto slot_mix = [&]() {
#ifndef SM3_B3_NOMIX
#pragma unroll
    for (int m = 0; m < TI * TJ; m++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, (22 + TI * TJ - 1) / (TI * TJ), 0);  // VALU
      __builtin_amdgcn_sched_group_barrier(0x200, (3 + TI * TJ - 1) / (TI * TJ), 0);   // DS write
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);  // the refill loads last
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  auto body_b3 = [&](f32x4 (&ra)[PA], f32x4 (&rb)[PB], int t, int par, bool tail) {
    const int st = par, nx = par ^ 1;
    const bool live = tail ? t + 1 < nk : true;
    const int kt_load = tail ? min(t + 3, nk - 1) : t + 3;
    const bool neg_store = SIGNED && par == 0;  // tile t + 1 (stored by this body) is odd <=> t is even
#pragma unroll
    for (int s_ = 0; s_ < 3; s_++) {
      rd_a(st, s_, s_);
      rd_b(st, s_, s_);
    }
    __builtin_amdgcn_sched_barrier(0);
    prod(1, 1, par);
    b3_piece(ra, rb, 0, nx, live, kt_load, tail, neg_store);
    slot_mix();
    prod(2, 0, par);
    b3_piece(ra, rb, 1, nx, live, kt_load, tail, neg_store);
    slot_mix();
    prod(0, 2, par);
    b3_piece(ra, rb, 2, nx, live, kt_load, tail, neg_store);
    slot_mix();
    prod(1, 0, par);
    b3_piece(ra, rb, 3, nx, live, kt_load, tail, neg_store);
    slot_mix();
    prod(0, 1, par);
    if (NP > 4) {
      b3_piece(ra, rb, 4, nx, live, kt_load, tail, neg_store);
      slot_mix();
    }
    prod(0, 0, par);
    if (NP > 5) b3_piece(ra, rb, 5, nx, live, kt_load, tail, neg_store);
    __syncthreads();
  };
  if (B3) {
    // bf16x3 prologue: tiles 0 and 1 into the two register sets (tile 0 goes to stage 0 below, its set then takes tile 2)
    if (nk > 0) {
#pragma unroll
      for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, 0, true);
#pragma unroll
      for (int q = 0; q < NP; q++) load_piece(sa1, sb1, q, min(1, nk - 1), true);
    }
  } else if (nk > 0) {  // first k-tile
#pragma unroll
    for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, 0, true);
  }
  {
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    if (SIGNED) {
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) accn[SIGNED ? i : 0][SIGNED ? j : 0][r] = 0.f;
    }
    if (B3) {
      // tile 0 -> stage 0; the register sets then hold tiles 1 (sa1: stored by body 0) and 2 (sa0: stored by body 1)
      if (nk > 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) store_piece(sa0, sb0, q, 0, true);
#pragma unroll
        for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, min(2, nk - 1), true);
      }
      __syncthreads();
    } else {
    if (nk > 0) {
#pragma unroll
      for (int q = 0; q < NP; q++) store_piece(sa0, sb0, q, 0, true);
#pragma unroll
      for (int q = 0; q < NP; q++) load_piece(sa0, sb0, q, min(1, nk - 1), true);
    }
    __syncthreads();
    }
#ifdef SM3_STAGGER  // measurement build only: the workgroups of the first dispatch round start their k-loops a third of a
    {                // tile-time apart per residency slot (tests whether co-resident workgroups run in lockstep)
      const unsigned lb = blockIdx.x + gridDim.x * blockIdx.z;
      if (lb < 768u) {
        const long long wait = (long long)((lb >> 8) % 3u) * nk * (BK * TI * TJ * 32);  // cls x MFMA cycles of the tile alone
        const long long t_in = __builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t_in < wait) __builtin_amdgcn_s_sleep(32);
      }
    }
#endif
    SM3_TR(1);
#ifdef SM3_TRACE
    if (p.trace && tid == 0 && sm3_first_pass)
      p.trace[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) * 8 + 6] = (unsigned long long)nk;
#endif

  // Two steps per iteration WITHOUT a branch between them (a conditional second step made hipcc drain vmcnt(0) at the
  // loop header).  The bulk loop carries no clamp / select / zeroing at all; the last three or four steps run the
  // `tail` variant, which also absorbs an odd nk by one extra step on an all-zero stage (see store_piece).
  // (bf16x3: a bulk body loads tile t + 3 unclamped and stores tile t + 1 as live)
  const int nk_bulk = max(0, nk - 3) & ~1;
  int kt = 0;
  // eq_prio (single-round launches, host-decided): the issue arbiter prefers the OLDEST wave, so of the three workgroups
  // that enter a CU together the first runs its loop in ~0.65 of the time of the last (17 / 27 / 37 us on 8192x1536x384,
  // profiles/r06/gemm_trace.txt) and the CU spends the last third of the launch with one k-loop alive.  With no further
  // workgroup to take a freed slot the shortest launch is the one where all three finish TOGETHER: every workgroup lowers
  // its waves' priority as it advances (3 -> 0), so whoever is behind issues first.
  // (quarters of the K range; geometric thresholds on the remaining k-tiles -- meeting again at 75 / 87 / 94 % -- measured
  // the same: the loops of a CU end 9-11 us apart instead of 21, profiles/r06/gemm_trace_eq_prio.txt)
  const int q1 = p.eq_prio ? (nk >> 2) & ~1 : -1, q2 = p.eq_prio ? (nk >> 1) & ~1 : -1, q3 = p.eq_prio ? (3 * nk >> 2) & ~1 : -1;
  if (p.eq_prio) __builtin_amdgcn_s_setprio(3);
  for (; kt < nk_bulk; kt += 2) {
    if (kt == q1) __builtin_amdgcn_s_setprio(2);
    else if (kt == q2) __builtin_amdgcn_s_setprio(1);
    else if (kt == q3) __builtin_amdgcn_s_setprio(0);
    if (B3) {
      body_b3(sa1, sb1, kt, 0, false);
      body_b3(sa0, sb0, kt + 1, 1, false);
    } else if (F16) {
      k_step16(sa0, sb0, sa1, sb1, kt, false);
      k_step16(sa1, sb1, sa0, sb0, kt + 1, false);
    } else {
      k_step(sa0, sb0, sa1, sb1, kt, false);
      k_step(sa1, sb1, sa0, sb0, kt + 1, false);
    }
  }
  for (; kt < nk; kt += 2) {
    if (B3) {
      body_b3(sa1, sb1, kt, 0, true);
      body_b3(sa0, sb0, kt + 1, 1, true);
    } else if (F16) {
      k_step16(sa0, sb0, sa1, sb1, kt, true);
      k_step16(sa1, sb1, sa0, sb0, kt + 1, true);
    } else {
      k_step(sa0, sb0, sa1, sb1, kt, true);
      k_step(sa1, sb1, sa0, sb0, kt + 1, true);
    }
  }

  if (SIGNED) {  // S = S_even - (-S_odd)
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) acc[i][j] -= accn[SIGNED ? i : 0][SIGNED ? j : 0];
  }
  if (CHECKER) {  // the row blocks of this wave that lie in odd 32-row blocks of the tile hold -S
    const int pw = (wm0 >> 5) & 1;  // wave-uniform
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        // (a multiplication by +-1, exact; NOT a bit_cast of `acc[i][j][r]` xor a sign word: hipcc 7.2 evaluates
        // __builtin_bit_cast applied directly to a vector-element expression as element 0 -- see ldg2 above)
        acc[i][j] *= ((pw ^ i) & 1) ? -1.0f : 1.0f;
      }
  }
  SM3_TR(2);
  if (CSUM && do_cs && F16) {  // an F16 piece = four k of ONE column (unit idx = k-quad * BM + column): [BK / 4][BM]
    float* red = smem;
#pragma unroll
    for (int i = 0; i < PA; i++) {
      const int idx = tid + NTHREADS * i;
      int g4, c;
      unit_gc(idx, BM, g4, c);
      if (idx < (BK / 4) * BM) red[g4 * BM + c] = cs1[(CSUM && F16) ? i : 0];
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float t = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < BK / 4; g4++) t += red[g4 * BM + tid];
      p.csum[(long)blockIdx.z * p.csum_stride + m0 + tid] = t;
    }
    __syncthreads();  // the epilogue stages through the same memory
  } else if (CSUM && do_cs) {  // fold the k rows of the tile through LDS (free after the loop)
    constexpr int QRc = BM / 4;
    float* red = smem;  // [BK][BM + 4]
#pragma unroll
    for (int i = 0; i < PA; i++) {
      const int idx = tid + NTHREADS * i;
      if (idx < BK * QRc) *reinterpret_cast<f32x4*>(red + (idx / QRc) * (BM + 4) + 4 * (idx % QRc)) = csa[(CSUM && !F16) ? i : 0];
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float t = 0.f;
#pragma unroll
      for (int kk = 0; kk < BK; kk++) t += red[kk * (BM + 4) + tid];
      p.csum[(long)blockIdx.z * p.csum_stride + m0 + tid] = t;
    }
    __syncthreads();  // the epilogue stages through the same memory
  }

#ifdef SM3_ABL_NOEPI
  if (F16) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) t += acc[i][j][r];
    if (t == 12345.678f) p.C[tid] = t;
    return;
  }
#endif
  const int e_m0 = m0, e_n0 = n0, e_row_end = row_end, e_g = g, e_tile_m = tile_m, e_bid = bid, e_split = split;
  // ---- split-K fix-up: publish this slice, the last arriver of the tile sums all slices in order ------------
  if (p.splits > 1 && p.fixup) {
    constexpr int NF = TI * TJ * 4;  // float4 fragments per thread
    const long tile_id = (MODE == MODE_TN ? (long)e_g * gridDim.x : 0) + e_bid;
    float* tile_slabs = p.slabs + tile_id * p.splits * (long)(BM * BN);
    f32x4* mine = reinterpret_cast<f32x4*>(tile_slabs + (long)e_split * (BM * BN));
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          mine[((i * TJ + j) * 4 + q) * NTHREADS + tid] =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its slab stores have left the CU
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);  // the k-loop ended with a barrier: LDS is free
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (ticket == p.splits - 1);
    }
    __syncthreads();
    if (!*flag) {
      SM3_TR(3);
      return;
    }
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.splits; s++) {  // slice order, whoever arrived last: deterministic sum
      const f32x4* sl = reinterpret_cast<const f32x4*>(tile_slabs + (long)s * (BM * BN));
      f32x4 v[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) v[f] = sl[f * NTHREADS + tid];
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
          for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][4 * q + e] += v[(i * TJ + j) * 4 + q][e];
    }
    __syncthreads();  // `flag` word is reused by the column-sum scratch below
  }

  SM3_TR(3);
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // acc[i][j][4q + e]: row = wm0 + 32 i + l31 ; col = wn0 + 32 j + 8 q + 4 lh + e   (e = 0..3 contiguous).
  // Stored straight from the fragments, one wave instruction would touch 32 rows x 32 bytes: a quarter of each cache
  // line.  Measured on the epilogue-bound shapes (K = 96 / 192 with GELU and a second output): 201 -> 145 us.
  float* __restrict__ Cg = p.C;
  long c_base = 0;
  int m_lim;
  if (p.splits > 1 && !p.fixup) c_base = (long)blockIdx.z * p.strideC;  // raw slice z of a later reduce pass
  else if (MODE == MODE_TN) c_base = (long)e_g * p.strideC;
  m_lim = (MODE == MODE_TN) ? p.M : e_row_end;
  const float* bias = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_SCALE_RES || EPI == EPI_BIAS_RELU)
                          ? p.bias + (long)e_g * p.strideBias
                          : nullptr;
  // Staged epilogue: every 32x32 accumulator tile goes through a per-wave LDS patch (32 rows x 36 floats) and comes
  // back row-contiguous: one wave instruction then touches 8 rows x 128 contiguous bytes (full cache lines) of C and of
  // the auxiliary tensors.  LDS operations of one wave execute in order, so the write -> read hand-over needs no
  // barrier, only a compiler fence.
  const int lane_e = tid & 63, wave_e = tid >> 6;
  float* stg = smem + wave_e * (32 * 36);
  const int sr = lane_e >> 3, sc = (lane_e & 7) * 4;
  f32x4 cs[TJ];
  if (EPI == EPI_GELU_BWD) {
#pragma unroll
    for (int j = 0; j < TJ; j++) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // The auxiliary input (gelu' / residual) runs two 32x32 tiles ahead of the tile being staged and stored (three rotating
  // register buffers; the request for tile t + 2 goes out once tile t's accumulators have been written to LDS and their
  // registers are free): otherwise every tile exposes one full HBM round trip -- the compiler cannot hoist the loads
  // over the LDS hand-over.
  constexpr bool AUX_IN = (EPI == EPI_GELU_BWD || EPI == EPI_BIAS_SCALE_RES);
  constexpr int NT_ = TI * TJ;
  constexpr int AUX_DEPTH = 1;  // tiles of look-ahead (2 spills at 128 VGPRs: measured slower)
  f32x4 pre[AUX_DEPTH + 1][4];
  auto aux_fetch = [&](int t, f32x4 (&dst)[4]) {
    const int j = t / TI, i = t - j * TI;
    const int col = e_n0 + wn0 + 32 * j + sc;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = e_m0 + wm0 + 32 * i + sr + 8 * it;
      const bool ok = row < m_lim && col < p.N;
      const long ai = ok ? (long)row * p.ld_aux + col : 0;
      if (X16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.aux_in) + ai);
        dst[it] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        dst[it] = *reinterpret_cast<const f32x4*>(p.aux_in + ai);
      }
    }
  };
  // output stores: fp32, or fp16 (round-to-nearest-even, like a torch .half() cast) where the tensor is stored as half
  auto st_c = [&](long ci, const f32x4& v) {
    if (C16) {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(Cg) + ci) =
          f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else {
      *reinterpret_cast<f32x4*>(Cg + ci) = v;
    }
  };
  auto st_x = [&](long ai, const f32x4& v) {
    if (X16) {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(p.aux_out) + ai) =
          f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else {
#ifndef SM3_AUX_TEMPORAL  // (A/B: --variant aux_temporal restores the plain store)
      // GELU' is not read again before the backward pass: a non-temporal store keeps it from evicting the activations the
      // next GEMM reads (stage 0: act and GELU' are 201 MB each, the MALL holds 256 MB) -- 16.25 -> 16.16 ms per step, same box
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.aux_out + ai));
#else
      *reinterpret_cast<f32x4*>(p.aux_out + ai) = v;
#endif
    }
  };
  if (AUX_IN) {
    aux_fetch(0, pre[0]);
    if (AUX_DEPTH > 1 && NT_ > 1) aux_fetch(1, pre[1]);
  }
#pragma unroll
  for (int j = 0; j < TJ; j++) {
    const int col = e_n0 + wn0 + 32 * j + sc;
    const bool col_ok = col < p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = bv;
    if (bias && col_ok) bv = *reinterpret_cast<const f32x4*>(bias + col);
    if (EPI == EPI_BIAS_SCALE_RES && col_ok) gv = *reinterpret_cast<const f32x4*>(p.gamma + col);
#pragma unroll
    for (int i = 0; i < TI; i++) {
      const int t = j * TI + i;
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<f32x4*>(stg + l31 * 36 + 8 * q + 4 * lh) =
            f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      asm volatile("" ::: "memory");
      if (AUX_IN && t + AUX_DEPTH < NT_) aux_fetch(t + AUX_DEPTH, pre[(t + AUX_DEPTH) % (AUX_DEPTH + 1)]);
      f32x4 v[4];
#pragma unroll
      for (int it = 0; it < 4; it++) v[it] = *reinterpret_cast<const f32x4*>(stg + (sr + 8 * it) * 36 + sc);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int row = e_m0 + wm0 + 32 * i + sr + 8 * it;
        if (row >= m_lim || !col_ok) continue;
        const long ci = c_base + (long)row * p.ldc + col;
        const long ai = (long)row * p.ld_aux + col;
        if (EPI == EPI_NONE) {
          st_c(ci, v[it]);
        } else if (EPI == EPI_BIAS) {
          st_c(ci, v[it] + bv);
        } else if (EPI == EPI_BIAS_RELU) {
          f32x4 o = v[it] + bv;
#pragma unroll
          for (int e = 0; e < 4; e++) o[e] = fmaxf(o[e], 0.f);
          st_c(ci, o);
        } else if (EPI == EPI_BIAS_GELU) {
          const f32x4 h = v[it] + bv;
          f32x4 y, dy;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float ye, de;
            gelu_erf_both(h[e], ye, de);
            y[e] = ye;
            dy[e] = de;
          }
          st_x(ai, dy);
          st_c(ci, y);
        } else if (EPI == EPI_BIAS_SCALE_RES) {
          const f32x4 y = v[it] + bv;
#ifndef SM3_AUX_TEMPORAL
          __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(p.aux_out + ai));  // y: saved for the backward pass only
#else
          *reinterpret_cast<f32x4*>(p.aux_out + ai) = y;  // y and the residual stream are fp32 in every data path
#endif
          const float rsc = p.rowscale ? p.rowscale[row / p.rows_per_scale] : 1.f;
          st_c(ci, pre[t % (AUX_DEPTH + 1)][it] + (gv * rsc) * y);
        } else if (EPI == EPI_GELU_BWD) {
          const f32x4 o = v[it] * pre[t % (AUX_DEPTH + 1)][it];
          st_c(ci, o);
          cs[j] += o;
        }
      }
    }
  }
  if (EPI == EPI_GELU_BWD && p.colpart) {
    __syncthreads();  // every wave is done with its staging patch: the column-sum scratch overlays it
    float* red = smem;  // [4 waves][TJ*32]
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      f32x4 t = cs[j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; e++) t[e] += __shfl_xor(t[e], o, 64);
      if (sr == 0) *reinterpret_cast<f32x4*>(red + wave * (TJ * 32) + 32 * j + sc) = t;
    }
    __syncthreads();
    if (tid < BN && e_n0 + tid < p.N) {
      const int wn = tid / (TJ * 32), c = tid - wn * (TJ * 32);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; w++) t += red[(w * WN + wn) * (TJ * 32) + c];
      p.colpart[(long)e_tile_m * p.N + e_n0 + tid] = t;
    }
  }
#ifdef SM3_TRACE
  if (sm3_first_pass) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores of this wave have left the CU
  SM3_TR(4);
  sm3_first_pass = false;
  sm3_items++;
#endif
  }
}

// launchers (one translation unit per mode: gemm_f32.hip = NT, gemm_f32_nn.hip, gemm_f32_tn.hip)
// tile: 0 128x128, 1 128x96, 2 96x128, 3 128x192, 4 192x128, 5 64x128 ; bk: 16 | 32 ; returns SM3_* code
int launch_nt(const GemmParams& p, int epi, int tile, int bk, int gather, dim3 grid, hipStream_t st);
int launch_nn(const GemmParams& p, int epi, int tile, int bk, int gather, dim3 grid, hipStream_t st);
int launch_tn(const GemmParams& p, int tile, int bk, int gather, dim3 grid, hipStream_t st);
// fp16-operand variants (gemm_f16.hip): k-step 16 | 32 | 64, tiles 0 / 1 / 5 (NT, NN), 0 / 1 / 2 (TN)
// io: IO_* bits = which tensors are stored as fp16 (0: all fp32 in HBM, rounded in the loader).  gemm_h16.hip holds the
// io != 0 instantiations: NT {A16: bias / bias+scale+residual / none; A16|C16|X16: bias+GELU}, NN {A16: none;
// C16|X16: GELU'}, TN {B16; A16|B16}
int launch_nt16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st);
int launch_nn16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st);
int launch_tn16(const GemmParams& p, int tile, int bk, int io, dim3 grid, hipStream_t st);
int launch_nt_h16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st);
int launch_nn_h16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st);
int launch_tn_h16(const GemmParams& p, int tile, int bk, int io, dim3 grid, hipStream_t st);

// bf16x3 form (gemm_b3_{nt,nn,tn}.hip): fp32 tensors, k-step 16, tiles 0 / 1 / 5 (NT, NN), 0 / 1 / 2 (TN)
int launch_nt_b3(const GemmParams& p, int epi, int tile, dim3 grid, hipStream_t st);
int launch_nn_b3(const GemmParams& p, int epi, int tile, dim3 grid, hipStream_t st);
int launch_tn_b3(const GemmParams& p, int tile, dim3 grid, hipStream_t st);
// ... with operands as bf16x3 planes (io = IO_APL | IO_BPL bits; gemm_b3_pl_{nt,nn}.hip)
int launch_nt_b3_pl(const GemmParams& p, int epi, int tile, int io, dim3 grid, hipStream_t st);
int launch_nn_b3_pl(const GemmParams& p, int epi, int tile, int io, dim3 grid, hipStream_t st);
// implicit-GEMM 3x3 convolutions in the bf16x3 form (128x128 tile, k-step 16; weight gradient: the row-aligned gather only)
int launch_nt_b3_conv(const GemmParams& p, int epi, dim3 grid, hipStream_t st);
int launch_nn_b3_conv(const GemmParams& p, dim3 grid, hipStream_t st);
int launch_tn_b3_conv(const GemmParams& p, dim3 grid, hipStream_t st);


inline void tile_dims(int tile, int& bm, int& bn) {
  static const int d[6][2] = {{128, 128}, {128, 96}, {96, 128}, {128, 192}, {192, 128}, {64, 128}};
  bm = d[tile][0];
  bn = d[tile][1];
}

}  // namespace sm3gemm
