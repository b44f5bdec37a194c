// gemm_f32.hip -- host side of the fp32 GEMM family (tile / k-step / split-K selection, workspaces, the reductions that
// follow a GEMM) + the MODE_NT instantiations of the kernel template in gemm_f32_kernel.h (MODE_NN / MODE_TN live in
// gemm_f32_nn.hip / gemm_f32_tn.hip so the three compile in parallel).
#include <stdlib.h>
#include <string.h>

#include "gemm_f32_kernel.h"

using namespace sm3gemm;

namespace sm3gemm {

template <int EPI, int BK, class TL>
static void go_nt(const GemmParams& p, dim3 grid, hipStream_t st) {
  gemm_f32_kernel<MODE_NT, EPI, BK, TL, 0><<<grid, NTHREADS, 0, st>>>(p);
}

template <int EPI>
static int nt_by_tile(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile * 100 + bk) {
    case 16: go_nt<EPI, 16, T128x128>(p, grid, st); return SM3_OK;
    case 32: go_nt<EPI, 32, T128x128>(p, grid, st); return SM3_OK;
    case 116: go_nt<EPI, 16, T128x96>(p, grid, st); return SM3_OK;
    case 132: go_nt<EPI, 32, T128x96>(p, grid, st); return SM3_OK;
    case 316: go_nt<EPI, 16, T128x192>(p, grid, st); return SM3_OK;
    case 516: go_nt<EPI, 16, T64x128>(p, grid, st); return SM3_OK;
    case 532: go_nt<EPI, 32, T64x128>(p, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nt(const GemmParams& p, int epi, int tile, int bk, int gather, dim3 grid, hipStream_t st) {
  if (gather) {
    if (tile != 0 || bk != 32) return SM3_ERR_INVALID_ARG;
    if (epi == EPI_NONE) gemm_f32_kernel<MODE_NT, EPI_NONE, 32, T128x128, 1><<<grid, NTHREADS, 0, st>>>(p);
    else if (epi == EPI_BIAS) gemm_f32_kernel<MODE_NT, EPI_BIAS, 32, T128x128, 1><<<grid, NTHREADS, 0, st>>>(p);
    else if (epi == EPI_BIAS_RELU) gemm_f32_kernel<MODE_NT, EPI_BIAS_RELU, 32, T128x128, 1><<<grid, NTHREADS, 0, st>>>(p);
    else return SM3_ERR_INVALID_ARG;
    return SM3_OK;
  }
  switch (epi) {
    case EPI_NONE: return nt_by_tile<EPI_NONE>(p, tile, bk, grid, st);
    case EPI_BIAS: return nt_by_tile<EPI_BIAS>(p, tile, bk, grid, st);
    case EPI_BIAS_GELU: return nt_by_tile<EPI_BIAS_GELU>(p, tile, bk, grid, st);
    case EPI_BIAS_SCALE_RES: return nt_by_tile<EPI_BIAS_SCALE_RES>(p, tile, bk, grid, st);
    case EPI_BIAS_RELU: return nt_by_tile<EPI_BIAS_RELU>(p, tile, bk, grid, st);
  }
  return SM3_ERR_INVALID_ARG;
}

}  // namespace sm3gemm

namespace {

// colpart [row tiles][N] -> out[g][n]: sum the row tiles that belong to group g (same tile->group map as the GEMM, whose
// row tile height is `bm`).  block = 16 column quads (64 columns) x 64 tile lanes, 8 independent 16-byte loads in flight
// per thread: 512 row tiles per round trip (1024 exist at stage 0; four lanes walking them two at a time took 128 trips)
constexpr int TCR_THREADS = 1024;
__global__ __launch_bounds__(TCR_THREADS) void tile_colsum_reduce_kernel(const float* __restrict__ colpart, int N, int M,
                                                                        int bm, const int32_t* __restrict__ offsets,
                                                                        int num_groups, float* __restrict__ out) {
  __shared__ f32x4 red[64][17];
  const int g = blockIdx.y;
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + 4 * cq;  // N is a multiple of 4
  int t0 = 0, t1 = (M + bm - 1) / bm;
  if (offsets) {
    int base = 0;
    for (int gg = 0; gg < g; gg++) base += (offsets[gg + 1] - offsets[gg] + bm - 1) / bm;
    t0 = base;
    t1 = base + (offsets[g + 1] - offsets[g] + bm - 1) / bm;
  }
  f32x4 s[8];
#pragma unroll
  for (int u = 0; u < 8; u++) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n < N && t1 > t0) {
    for (int t = t0 + rl; t < t1; t += 512) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = t + 64 * u;
        const f32x4 v = *reinterpret_cast<const f32x4*>(colpart + (long)min(r, t1 - 1) * N + n);
        if (r < t1) s[u] += v;
      }
    }
  }
  red[rl][cq] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (threadIdx.x < 256) {
    const int col = threadIdx.x >> 2, part = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) t += red[16 * part + i][col >> 2][col & 3];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    const int oc = blockIdx.x * 64 + col;
    if (part == 0 && oc < N) out[(long)g * N + oc] = t;
  }
}

// Sum the raw split-K slices of a GEMM: out[g][i] = sum_s ws[(g*splits+s)][i] (+ bias, relu).  A workgroup is
// 256 / SL output quads x SL slice lanes: slice lane l adds slices l, l+SL, ... (4 independent loads in flight), the
// lanes meet in LDS in a fixed order.  Round 1 ran one thread per quad over all slices: 341 dependent loads per thread
// and 36 workgroups on the stage-0 weight gradients (124 us per call).
template <int SL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           long mn, int splits, int groups,
                                                           const float* __restrict__ bias, int ncols, int relu,
                                                           float* __restrict__ out2, long mn1) {
  constexpr int QB = 256 / SL;
  __shared__ f32x4 red[SL][QB];
  const int ql = threadIdx.x % QB, sl = threadIdx.x / QB;
  const long total = mn * groups;
  const long idx = ((long)blockIdx.x * QB + ql) * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  long e = 0;
  if (idx < total) {
    const long gidx = idx / mn;
    e = idx - gidx * mn;
    const float* base = ws + gidx * splits * mn + e;
    int k = sl;
    for (; k + 3 * SL < splits; k += 4 * SL) {
      s0 += *reinterpret_cast<const f32x4*>(base + (long)k * mn);
      s1 += *reinterpret_cast<const f32x4*>(base + (long)(k + SL) * mn);
      s2 += *reinterpret_cast<const f32x4*>(base + (long)(k + 2 * SL) * mn);
      s3 += *reinterpret_cast<const f32x4*>(base + (long)(k + 3 * SL) * mn);
    }
    for (; k < splits; k += SL) s0 += *reinterpret_cast<const f32x4*>(base + (long)k * mn);
  }
  f32x4 s = (s0 + s1) + (s2 + s3);
  if (SL > 1) {
    red[sl][ql] = s;
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int l = 1; l < SL; l++) s += red[l][ql];
  }
  if (idx < total) {
    if (bias) s += *reinterpret_cast<const f32x4*>(bias + (e % ncols));  // rows of ncols (multiple of 4) columns
    if (relu) {
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] = fmaxf(s[q], 0.f);
    }
    // out2 != NULL: every slice carries a tail of mn - mn1 floats (TN column sums of A) that goes to its own tensor
    if (out2 == nullptr) *reinterpret_cast<f32x4*>(out + idx) = s;
    else if (e < mn1) *reinterpret_cast<f32x4*>(out + (idx / mn) * mn1 + e) = s;
    else *reinterpret_cast<f32x4*>(out2 + (idx / mn) * (mn - mn1) + (e - mn1)) = s;
  }
}

void launch_splitk_reduce(const float* ws, float* out, long mn, int splits, int groups, const float* bias, int ncols,
                          int relu, hipStream_t st, float* out2 = nullptr, long mn1 = 0) {
  const long quads = mn * groups / 4;
  if (splits >= 64) {
    splitk_reduce_kernel<16><<<(int)((quads + 15) / 16), 256, 0, st>>>(ws, out, mn, splits, groups, bias, ncols, relu,
                                                                      out2, mn1);
  } else if (splits >= 8) {
    splitk_reduce_kernel<4><<<(int)((quads + 63) / 64), 256, 0, st>>>(ws, out, mn, splits, groups, bias, ncols, relu,
                                                                     out2, mn1);
  } else {
    splitk_reduce_kernel<1><<<(int)((quads + 255) / 256), 256, 0, st>>>(ws, out, mn, splits, groups, bias, ncols, relu,
                                                                       out2, mn1);
  }
}

// Column sums over row segments (bias gradients): out[g][n] += sum_{r in seg g} X[r][n]; `out` is zeroed by the
// wrapper below (a zero-fill kernel on the same stream).  Block = 16 column quads (64 columns, 16 B per lane) x 16 row lanes; row
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
  if (c < N) {
    f32x4 s1 = s, s2 = s, s3 = s;  // four independent loads in flight per thread
    int r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
      s += *reinterpret_cast<const f32x4*>(X + (long)r * ld + c);
      s1 += *reinterpret_cast<const f32x4*>(X + (long)(r + 16) * ld + c);
      s2 += *reinterpret_cast<const f32x4*>(X + (long)(r + 32) * ld + c);
      s3 += *reinterpret_cast<const f32x4*>(X + (long)(r + 48) * ld + c);
    }
    for (; r < r1; r += 16) s += *reinterpret_cast<const f32x4*>(X + (long)r * ld + c);
    s = (s + s1) + (s2 + s3);
  }
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

// ---- configuration: tile shape, k-step, split-K ---------------------------------------------------------------
constexpr int COUNTER_SLOTS = kCounterSlots;

struct Cfg {
  int tile, bk, splits, fixup;
  int bm, bn, ntm, ntn, groups;
  int ktps;  // NT/NN: k-tiles per slice
};

inline int pad_to(int n, int b) { return (n + b - 1) / b * b; }

// blocks -> rounds of 256 CUs actually paid for / rounds of work (>= 1): the matrix pipes of a CU are shared by its
// resident workgroups, so a launch costs as many tile-times as the fullest CU holds
inline double quant_cost(long blocks) {
  const double pc = (double)blocks / kNumCU;
  const long rounds = (blocks + kNumCU - 1) / kNumCU;
  return pc > 0 ? (double)rounds / pc : 1.0;
}

// Tile / k-step / split-K selection: a small cost model fitted to scripts/gemm_sweep2.py on MI355X (tile x k-step x slices
// over every GEMM shape of the ConvNeXt-T e8t2 training step; profiles/r02/gemm_sweep_r02b.txt is the sweep of the
// current kernel, refit: scripts/fit_gemm_nt.py, scripts/fit_gemm_tn.py):
//  * NT/NN: (tile in 128x128, 128x96, 64x128) x (1..8 k-slices, only with < 256 tiles and a long K, in-kernel fix-up);
//    cost = CU-round quantisation x padded-width waste x tile handicap x penalty for CUs with < 2 workgroups x 10 % per
//    extra slice.  k-step 32 from K = 1024.
//  * TN: (128x128, 128x96, 96x128) x slices 1..192 reduced by the second pass (never the in-kernel fix-up: on 64 KB slabs
//    it measured slower); cost = quantisation x waste x handicap + the slab round trip, 60*G/K of the GEMM's own time per
//    slice.  Since the k-loop lost its vector-ALU work the 96-wide tiles are as fast per FLOP as 128x128.
// d->tuning (benchmarking aid, 0 in production): bits 0-3 tile+1, 4-7 k-step (1 = 16, 2 = 32, 3 = 64: fp16 only), 8-15 slices,
// bit 16: TN slices summed by the in-kernel fix-up instead of the second pass (so a literal 256 in the slices field
// reads as `automatic + fix-up`: scripts/gemm_sweep2.py)
Cfg choose_cfg(const sm3_gemm_desc* d) {
  Cfg c;
  memset(&c, 0, sizeof(c));
  c.groups = d->num_groups > 0 ? d->num_groups : 1;
  const unsigned tune = (unsigned)d->tuning;
  const int t_tile = (int)(tune & 15) - 1, t_bk = (int)((tune >> 4) & 15),
            t_splits = (int)((tune >> 8) & 255) | (int)(((tune >> 20) & 15) << 8);  // bits 20-23: slices / 256
  const bool have_counters = d->counters != nullptr;
  const int G = c.groups;
  if (d->mode != MODE_TN) {
    // joint choice of (tile, k-slices): cost = CU-round quantisation x padded-width waste x tile handicap x a penalty for
    // CUs left with fewer than two workgroups x 10 % per extra slice (fix-up traffic); refitted on
    // profiles/r02/gemm_sweep_r02b.txt (regret 0.3 % of the summed NT/NN time of the training step)
    static const int cand[3] = {0, 1, 5};
    static const double handicap32[3] = {1.0, 1.04, 1.02};
    // fp16 operands: the loop is bound by operand delivery, not by the matrix pipe, so the tile with the most reuse wins
    // unless CU-round quantisation costs more than 15 % (cold-operand sweep, profiles/r03/amp_gemm/gemm_sweep_amp_cold.txt)
    static const double handicap16[3] = {1.0, 1.04, 1.15};
    const double* handicap = d->compute == 1 ? handicap16 : handicap32;
    int bk = d->K >= 1024 ? 32 : 16;
    if (t_bk) bk = t_bk == 1 ? 16 : 32;
    if (d->compute == 1) {
      // k-step 64 (two workgroups per CU, half the barriers, 128 contiguous bytes per fp16 row and step) pays on the long
      // reductions and on the stage-0 FC2 shape (N <= 128 columns, an M-long stream of fp16 rows); same sweep
      bk = (d->K >= 3072 || (d->mode == MODE_NT && d->N <= 128 && d->K >= 384 && d->M >= 65536)) ? 64 : 32;
      if (t_bk) bk = t_bk == 1 ? 16 : (t_bk == 3 ? 64 : 32);
    }
    if (d->compute == 2) bk = 16;  // bf16x3 form: k-step 16 only (three LDS planes per operand)
    if (d->K % bk) bk = (bk == 64 && d->K % 32 == 0) ? 32 : 16;
    const int kt = d->K / bk;
    double best = 0;
    int best_s = 1;
    c.tile = -1;
    for (int i = 0; i < 3; i++) {
      if (t_tile >= 0 && t_tile != cand[i]) continue;
      int bm, bn;
      tile_dims(cand[i], bm, bn);
      const long tiles = (long)((d->M + bm - 1) / bm + (d->group_offsets ? G / 2 : 0)) * ((d->N + bn - 1) / bn);
      const long slots = (long)((d->M + bm - 1) / bm + (d->group_offsets ? G : 0)) * ((d->N + bn - 1) / bn);
      const double waste = (double)pad_to(d->N, bn) / d->N;
      for (int s = 1; s <= 8; s++) {
        if (s > 1 && (!have_counters || slots > COUNTER_SLOTS - 16 || d->splits == 1 || tiles >= kNumCU || kt < 24 ||
                      kt / s < 6 || t_tile >= 0))
          break;
        const double pc = (double)tiles * s / kNumCU;
        const double occ = pc < 1.0 ? 1.2 : (pc < 2.0 ? 1.0 + 0.05 * (2.0 - pc) : 1.0);
        const double cost = quant_cost(tiles * s) * handicap[i] * waste * occ * (1.0 + 0.1 * (s - 1));
        if (c.tile < 0 || cost < best - 1e-9) { best = cost; c.tile = cand[i]; best_s = s; }
      }
    }
    if (c.tile < 0) { c.tile = t_tile >= 0 ? t_tile : 0; best_s = 1; }
    c.bk = bk;
    if (c.tile == 3) c.bk = 16;
    if (d->compute == 1 && c.tile != 0 && c.tile != 1 && c.tile != 5) c.tile = 0;  // fp16: 128x128 / 128x96 / 64x128
    if (d->compute == 2 && c.tile != 0 && c.tile != 1 && c.tile != 5 && c.tile != 3) c.tile = 0;  // bf16x3: + 128x192
    if (d->compute == 2) c.bk = 16;
    tile_dims(c.tile, c.bm, c.bn);
    c.ntn = (d->N + c.bn - 1) / c.bn;
    // ragged groups: at most ceil(M/BM) + G row tiles exist; surplus blocks exit
    c.ntm = (d->M + c.bm - 1) / c.bm + (d->group_offsets ? G : 0);
    const int ktc = d->K / c.bk;
    const long tiles = (long)c.ntm * c.ntn;
    int s = 1;
    if (have_counters && tiles <= COUNTER_SLOTS - 16) {
      s = best_s;
      if (d->splits > 1) s = d->splits;
      if (t_splits) s = t_splits;
    }
    if (s > ktc) s = ktc > 0 ? ktc : 1;
    c.ktps = (ktc + s - 1) / s;
    if (s > 1 && (c.ktps & 1)) c.ktps++;  // even number of k-tiles per slice: no phantom k-step (gemm_f32_kernel.h)
    s = c.ktps > 0 ? (ktc + c.ktps - 1) / c.ktps : 1;  // every slice owns >= 1 k-tile
    c.splits = s;
    c.fixup = s > 1 ? 1 : 0;
    return c;
  }
  // ---- TN
  static const int cand[3] = {0, 1, 2};
  static const double handicap32[3] = {1.0, 0.97, 0.97}, handicap16[3] = {1.0, 1.1, 1.1};  // fp16: see above
  const double* handicap = d->compute == 1 ? handicap16 : handicap32;
  const int rows = d->K / G > 0 ? d->K / G : 1;
  // bf16x3: a k-loop is a third as long, the launch is latency-bound and more, shorter slices win on the long reductions
  // (stage-0 / stage-1 weight gradients: 256 slices 78 us vs 89 us at the fp32 form's 85; profiles/r05/gemm_b3_sweep.txt)
  const double pen = d->K > 0 ? (d->compute == 2 ? 40.0 : 60.0) * G / d->K : 0.0;
  double best = 0;
  int best_s = 1;
  c.tile = -1;
  for (int i = 0; i < 3; i++) {
    int bm, bn;
    tile_dims(cand[i], bm, bn);
    if (t_tile >= 0 && cand[i] != t_tile) continue;
    const long tiles = (long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn) * G;
    const double waste = (double)pad_to(d->M, bm) * pad_to(d->N, bn) / ((double)d->M * d->N);
    for (int s = 1; s <= 512; s++) {
      if (s > 1 && rows / s < 64) break;
      if (d->splits > 0 && s != d->splits) continue;
      if (d->splits <= 0 && s > (d->compute == 2 ? 256 : 192)) break;  // beyond ~200 slices the second pass and the short k-loops cost more than
                                             // the extra workgroups bring (stage-0 weight gradients: 128-192 best)
      const long blocks = tiles * s;
      const double cost = quant_cost(blocks) * handicap[i] * waste + pen * (s > 1 ? s : 0);
      if (c.tile < 0 || cost < best - 1e-9) { best = cost; c.tile = cand[i]; best_s = s; }
    }
  }
  if (c.tile < 0) { c.tile = t_tile >= 0 ? t_tile : 0; best_s = d->splits > 0 ? d->splits : 1; }
  c.bk = 16;
  if (t_bk) c.bk = t_bk == 1 ? 16 : 32;
  if (c.tile >= 3) c.bk = 16;
  if (d->compute == 1) {
    c.bk = t_bk == 1 ? 16 : (t_bk == 3 ? 64 : 32);
    if (c.tile > 2) c.tile = 0;
  }
  if (d->compute == 2) {
    c.bk = 16;
    if (c.tile > 4) c.tile = 0;  // (128x192 / 192x128: through the tuning override)
  }
  tile_dims(c.tile, c.bm, c.bn);
  c.ntn = (d->N + c.bn - 1) / c.bn;
  c.ntm = (d->M + c.bm - 1) / c.bm;
  c.splits = t_splits ? t_splits : best_s;
  const long tiles = (long)c.ntm * c.ntn * G;
  c.fixup = (c.splits > 1 && ((tune >> 16) & 1) && have_counters && tiles <= COUNTER_SLOTS - 16) ? 1 : 0;
  return c;
}

size_t slab_bytes(const sm3_gemm_desc* d, const Cfg& c) {
  if (c.splits <= 1) return 0;
  if (c.fixup) {
    const long tiles = (long)c.ntm * c.ntn * (d->mode == MODE_TN ? c.groups : 1);
    return (size_t)tiles * c.splits * c.bm * c.bn * sizeof(float);
  }
  // TN raw slices in output layout (+ a tail of M column sums of A per slice when colsum_out is given)
  return (size_t)c.groups * c.splits * ((size_t)d->M * d->N + (d->colsum_out ? d->M : 0)) * sizeof(float);
}

size_t colpart_bytes(const sm3_gemm_desc* d, const Cfg& c) {
  if (d->mode == MODE_NN && d->epilogue == EPI_GELU_BWD && d->colsum_out) return (size_t)c.ntm * d->N * sizeof(float);
  return 0;
}

}  // namespace

// arithmetic of the convolutions (sm3_conv3x3_set_arith): 0 = v_mfma_f32_32x32x2_f32, 2 = the bf16x3 form (three exact bf16
// pieces per fp32 operand element, six bf16 MFMA products, fp32 accumulation: sm3_gemm_desc.compute == 2), k-step 16
static int g_conv_arith = 0;
static inline int conv_bk() { return g_conv_arith == 2 ? 16 : 32; }

#ifdef SM3_TRACE
static unsigned long long* g_trace = nullptr;
extern "C" void sm3_gemm_set_trace(void* buf) { g_trace = (unsigned long long*)buf; }  // measurement build only
#endif

extern "C" {

int sm3_conv3x3_set_arith(int compute) {
  if (compute != 0 && compute != 2) return SM3_ERR_INVALID_ARG;
  g_conv_arith = compute;
  return SM3_OK;
}

int sm3_gemm_f32_counter_slots(void) { return COUNTER_SLOTS; }

size_t sm3_gemm_f32_workspace_bytes(const sm3_gemm_desc* d) {
  if (!d || d->M < 0 || d->N <= 0 || d->K < 0) return 0;
  const Cfg c = choose_cfg(d);
  return align_up(slab_bytes(d, c), 256) + colpart_bytes(d, c);
}

int sm3_gemm_f32(const sm3_gemm_desc* d, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (!d) return SM3_ERR_INVALID_ARG;
  if (d->M < 0 || d->N <= 0 || d->K < 0) return SM3_ERR_INVALID_ARG;
  if ((d->N & 3) || (!(d->compute == 2 && (d->io & IO_APL)) && (d->lda & 3)) || (!(d->compute == 2 && (d->io & IO_BPL)) && (d->ldb & 3)))
    return SM3_ERR_UNSUPPORTED;  // float4 loads (plane operands: lda / ldb count rows, any value)
  if (d->mode != MODE_TN && (d->K % 32) != 0) return SM3_ERR_UNSUPPORTED;
  if (d->mode == MODE_TN && (d->M & 3)) return SM3_ERR_UNSUPPORTED;
  if (d->mode == MODE_TN && d->epilogue != EPI_NONE) return SM3_ERR_INVALID_ARG;
  if (d->mode != MODE_NT && d->mode != MODE_NN && d->mode != MODE_TN) return SM3_ERR_INVALID_ARG;
  if (d->compute != 0 && d->compute != 1 && d->compute != 2) return SM3_ERR_INVALID_ARG;
  const int pl = d->compute == 2 ? (d->io & (IO_APL | IO_BPL)) : 0;  // operands as bf16x3 planes (planes.hip)
  if (d->io != 0 && !pl && (d->compute != 1 || d->io < 0 || d->io > 15)) return SM3_ERR_INVALID_ARG;
  if (pl) {
    if (d->io != pl || d->mode == MODE_TN) return SM3_ERR_UNSUPPORTED;
    if (((pl & IO_APL) && (reinterpret_cast<uintptr_t>(d->A) & 15)) || ((pl & IO_BPL) && (reinterpret_cast<uintptr_t>(d->B) & 15)))
      return SM3_ERR_INVALID_ARG;  // 16-byte granules
    // lda / ldb = rows per k-octet block of the planes; the three planes must stay inside a 32-bit byte offset
    if (((pl & IO_APL) && (d->lda < d->M || 3l * (d->K / 8) * d->lda * 16 >= (1l << 31) - 65536)) ||
        ((pl & IO_BPL) && (d->ldb < d->N || 3l * (d->K / 8) * d->ldb * 16 >= (1l << 31) - 65536)))
      return SM3_ERR_UNSUPPORTED;
  }
  // fp16-stored operands: 8-byte loads of k-quads (lda % 4, checked above) / 4-byte loads of column pairs (even dims)
  if (d->mode == MODE_TN && (((d->io & 1) && (d->M & 1)) || ((d->io & 2) && (d->N & 1)))) return SM3_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const Cfg c = choose_cfg(d);
  {  // the kernel addresses each operand as block base + 32-bit byte offset (buffer loads): keep the spans below 2^31.
     // TN rebases per k-slice (a_base = A + row0 * lda), so what must fit is ONE slice's rows (a ragged group may hold all
     // of K), not the whole reduction: stage-0 weight gradients of any batch size pass as long as the slices are short.
    const long lim = (1l << 31) - 65536;
    const long ld = d->lda > d->ldb ? d->lda : d->ldb;
    const long slice_rows = (d->K + c.splits - 1) / (c.splits > 0 ? c.splits : 1) + 4 * c.bk;
    const long span = d->mode == MODE_TN ? slice_rows * ld * 4
                                         : ((long)256 * ld + d->K) * 4 + (d->mode == MODE_NN ? (long)d->K * d->ldb * 4 : 0);
    if (span >= lim) return SM3_ERR_UNSUPPORTED;
  }
  // every argument check happens before the first launch
  if (d->mode == MODE_TN && c.splits > 1 && !c.fixup && d->ldc != d->N) return SM3_ERR_UNSUPPORTED;
  const size_t sb = align_up(slab_bytes(d, c), 256), cb = colpart_bytes(d, c);
  if ((sb + cb) && (!workspace || workspace_bytes < sb + cb)) return SM3_ERR_WORKSPACE;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.offsets = d->group_offsets;
  p.num_groups = c.groups;
  p.strideB = d->stride_b; p.strideBias = d->stride_bias;
  p.strideC = (long)d->M * d->N;
  p.splits = c.splits; p.fixup = c.fixup; p.kTilesPerSplit = c.ktps;
  p.slabs = (float*)workspace;
  p.counters = (int*)d->counters;
  p.bias = d->bias; p.aux_in = d->aux_in; p.aux_out = d->aux_out;
  p.gamma = d->gamma; p.rowscale = d->rowscale;
  p.rows_per_scale = d->rows_per_scale > 0 ? d->rows_per_scale : 1;
  p.ld_aux = d->ld_aux;
  p.colpart = cb ? (float*)((char*)workspace + sb) : nullptr;
  {
    // progress-ordered wave priority (gemm_f32_kernel.h, k-loop): only when every workgroup of the launch is resident at
    // once on more than one workgroup per CU -- with further rounds to come the oldest-first order is the better pipeline
    static const int mode = [] { const char* e = getenv("SM3_EQ_PRIO"); return e ? atoi(e) : 2; }();  // 0 off, 1 always, 2 auto
    const long blocks = (long)c.ntn * c.ntm * (d->mode == MODE_TN ? (long)c.groups * c.splits : c.splits);
    static const int maxr = [] { const char* e = getenv("SM3_EQ_PRIO_MAXR"); return e ? atoi(e) : 3; }();  // workgroups per CU (A/B aid)
    static const int forms = [] { const char* e = getenv("SM3_EQ_PRIO_FORMS"); return e ? atoi(e) : 6; }();  // bit per d->compute
    p.eq_prio = ((forms >> d->compute) & 1) && (mode == 1 || (mode == 2 && blocks > kNumCU && blocks <= (long)maxr * kNumCU));
  }
#ifdef SM3_TRACE
  p.trace = g_trace;
#endif
  if (d->mode == MODE_TN) {
    float* out = d->C;
    const long mn1 = (long)d->M * d->N;
    const bool cs = d->colsum_out != nullptr;  // bias gradient = column sums of A, a by-product of the A loader
    // (an fp16-STORED A operand has no fp32 values to add: the caller keeps the separate column-sum kernel there)
    if (cs && (c.fixup || (d->io & 1))) return SM3_ERR_UNSUPPORTED;
    const long mn = mn1 + (cs ? d->M : 0);
    if (c.splits > 1 && !c.fixup) {  // raw slices to the workspace, separate reduce pass
      p.C = (float*)workspace;
      p.ldc = d->N;
      p.strideC = mn;
      p.csum = cs ? (float*)workspace + mn1 : nullptr;
      p.csum_stride = mn;
    } else {
      p.csum = d->colsum_out;  // one slice: the sums are final
      p.csum_stride = d->M;
    }
    dim3 grid(c.ntn * c.ntm, 1, c.groups * c.splits);
    const int rc = d->compute == 1 ? launch_tn16(p, c.tile, c.bk, d->io, grid, st)
                   : d->compute == 2 ? launch_tn_b3(p, c.tile, grid, st)
                                     : launch_tn(p, c.tile, c.bk, 0, grid, st);
    if (rc) return rc;
    if (c.splits > 1 && !c.fixup) {
      launch_splitk_reduce((const float*)workspace, out, mn, c.splits, c.groups, nullptr, 0, 0, st,
                           cs ? d->colsum_out : nullptr, mn1);
    }
    return launch_status();
  }
  if (d->M == 0) return SM3_OK;
  dim3 grid(c.ntn * c.ntm, 1, c.splits);
  int rc;
  if (d->compute == 1) {
    rc = d->mode == MODE_NT ? launch_nt16(p, d->epilogue, c.tile, c.bk, d->io, grid, st)
                            : launch_nn16(p, d->epilogue, c.tile, c.bk, d->io, grid, st);
  } else if (d->compute == 2 && pl) {
    rc = d->mode == MODE_NT ? launch_nt_b3_pl(p, d->epilogue, c.tile, pl, grid, st)
                            : launch_nn_b3_pl(p, d->epilogue, c.tile, pl, grid, st);
  } else if (d->compute == 2) {
    rc = d->mode == MODE_NT ? launch_nt_b3(p, d->epilogue, c.tile, grid, st)
                            : launch_nn_b3(p, d->epilogue, c.tile, grid, st);
  } else {
    rc = d->mode == MODE_NT ? launch_nt(p, d->epilogue, c.tile, c.bk, 0, grid, st)
                            : launch_nn(p, d->epilogue, c.tile, c.bk, 0, grid, st);
  }
  if (rc) return rc;
  if (cb) {
    dim3 rg((d->N + 63) / 64, c.groups);
    tile_colsum_reduce_kernel<<<rg, TCR_THREADS, 0, st>>>(p.colpart, d->N, d->M, c.bm, d->group_offsets, c.groups,
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
  sm3_zero_async(out, sizeof(float) * (size_t)groups * n, (hipStream_t)stream);
  colsum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ld, n, group_offsets, m, splits, out);
  return launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution (padding 1, stride 1 or 2) on NHWC tokens as implicit GEMMs on the same kernel (GATHER = 1): the
// operand that a materialised im2col would hold is gathered tap by tap while the tile is loaded, so no column buffer
// exists (the FPN 3x3 at 256^2 x 256 ch would need a 1.2 GB one).  Weight layout (Cout, 3, 3, Cin) = [Cout][9*Cin].
static void conv_geometry(GemmParams& p, int cC, int sH, int sW, int rH, int rW, int stride, int transposed, int bk) {
  p.cC = cC; p.sH = sH; p.sW = sW; p.rH = rH; p.rW = rW; p.cS = stride; p.cT = transposed;
  p.cU = !transposed ? (0 | 1 << 4 | 2 << 8) : (stride == 2 ? (1 | 1 << 4 | 0 << 8) : (2 | 1 << 4 | 0 << 8));
  const unsigned tpt = (unsigned)(cC / bk);
  p.cInv = (65536u + tpt - 1) / tpt;
  p.mRW = (unsigned)((1ull << 32) / (unsigned)rW + 1);
  p.mRH = (unsigned)((1ull << 32) / (unsigned)rH + 1);
}

constexpr int BM = 128, BN = 128;  // the implicit-GEMM convolutions run on the 128x128 tile


static GemmParams conv_params_zero() {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.num_groups = 1;
  p.splits = 1;
  p.rows_per_scale = 1;
  return p;
}

static bool conv_dims_ok(int B, int H, int W, int Cin, int Cout, int stride) {
  // the gathered operand is addressed as descriptor base + 32-bit offsets, masked taps use an offset of 0x7fff0000
  return B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) &&
         (long)B * H * W < (1l << 24) && (long)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < (1l << 30);
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
  const int ks = conv_ksplits(m, n, 9 * kc / conv_bk());
  return ks > 1 ? (size_t)ks * m * n * sizeof(float) : 0;
}

static int conv_launch_nt_nn(GemmParams& p, int mode, const float* bias, int relu, float* out, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  const int k_tiles = p.K / conv_bk();
  const int ks = conv_ksplits(p.M, p.N, k_tiles);
  const long mn = (long)p.M * p.N;
  if (ks > 1) {  // raw k-slices to the workspace, one reduce pass adds them (+ bias, ReLU)
    if (!workspace || workspace_bytes < (size_t)ks * mn * sizeof(float)) return SM3_ERR_WORKSPACE;
    p.kTilesPerSplit = (k_tiles + ks - 1) / ks;
    p.strideC = mn;
    p.C = (float*)workspace;
    p.bias = nullptr;
  } else {
    p.C = out;
  }
  const int zs = ks > 1 ? (k_tiles + p.kTilesPerSplit - 1) / p.kTilesPerSplit : 1;
  p.splits = zs;
  p.fixup = 0;
  dim3 grid(((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM), 1, zs);
  const int epi = mode == MODE_NN ? EPI_NONE : (p.bias && relu ? EPI_BIAS_RELU : (p.bias ? EPI_BIAS : EPI_NONE));
  int rc;
  if (g_conv_arith == 2) rc = mode == MODE_NN ? launch_nn_b3_conv(p, grid, st) : launch_nt_b3_conv(p, epi, grid, st);
  else rc = mode == MODE_NN ? launch_nn(p, epi, 0, 32, 1, grid, st) : launch_nt(p, epi, 0, 32, 1, grid, st);
  if (rc) return rc;
  if (ks > 1) launch_splitk_reduce((const float*)workspace, out, mn, zs, 1, bias, p.N, relu, st);
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
  conv_geometry(p, Cin, H, W, Ho, Wo, stride, 0, conv_bk());
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
  conv_geometry(p, Cout, Ho, Wo, H, W, stride, 1, conv_bk());
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
  p.fixup = 0;
  // bf16x3: the row-aligned gather only (a k-tile inside one image row); other widths keep the native fp32 form
  const int rc = (g_conv_arith == 2 && (Wo % 16) == 0) ? launch_tn_b3_conv(p, grid, st)
                                                      : launch_tn(p, 0, 16, (Wo % 16) == 0 ? 2 : 1, grid, st);
  if (rc) return rc;
  launch_splitk_reduce((const float*)workspace, dw, (long)p.M * p.N, p.splits, 1, nullptr, 0, 0, st);
  return launch_status();
}

}  // extern "C"
