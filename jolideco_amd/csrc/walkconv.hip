// Strip-walk form of the separable 'same' convolution (rank-1 PSFs -- every sampled Gaussian -- whose non-zero taps fit a
// frame of 17 or of 33 taps per direction; the text below describes the 17-tap frame, the 33-tap frame is the same walk on
// 36 accumulator rows and two columns per lane).
//
// The tile kernel of sepconv.hip stages a (32 + halo) x (64 + halo) window in LDS, runs the row pass over ALL window
// rows (1.63 x the output rows) into a second LDS image, then the column pass: ~1000 vector instructions and ~200 KB of
// LDS traffic per 2048-pixel tile make it issue / LDS bound at 55-60 % of the HBM rate.  Here ONE WAVE owns a strip of
// 64 C columns (C = 2 or 4 per lane) and walks down R rows of it:
//   * per image row: the lanes multiply flux x exposure for their C columns, exchange the products through a
//     (64 C + 16)-float LDS row (one store, (16 + C) / 4 reads per lane), and run the 17-tap row pass on registers;
//   * the column pass is in SCATTER form: the finished row-pass value h[r] is added into the 17 outputs r - 8 .. r + 8
//     it contributes to, which live in 18 rotating accumulator slots of C registers per lane (the loop is unrolled over
//     the 18 rotation states so that every register index is static and the prefetch registers, period 2 or 3, rotate
//     with it); output row r - 8 is complete after row r and leaves the wave through the epilogue (Poisson pass, or
//     scale + accumulate for the adjoint).  At C = 4 the column pass runs on packed FMAs (two columns per instruction).
// No second LDS image, no block barriers, no halo rows recomputed inside a tile: 34 FMAs, 1 + (16 + C) / C LDS floats
// and one set of streaming loads per pixel; the only overhead is the 16 warm-up rows of a tile (R = 74: 22 % more row
// passes).  Loads of row r + P are issued P steps ahead (registers), row taps live in SGPRs.  The launches follow their
// own instruction stream, not the memory system.
//
// Batched adjoint (the gradient of a joint step, sum over the datasets of E_d x corr(g_d, psf_d)): a block is one wave
// PER DATASET walking the same strip; finished rows go to an LDS exchange buffer in groups of 6, 3 or 2, and the waves
// add them in dataset order -- the additions of the per-dataset launches, bit for bit -- into the gradient image, which
// is read and written once.  Several flux components: walk_multi_kernel (forward, one wave per component) and the same
// adjoint over a grid of components x tiles with up to 16 waves per block.
#include <cmath>
#include <utility>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {

// The walk's FRAME: WK taps per direction, out[y] = sum_t tu[t] h[y - WH + t], and WS accumulator slots (the rotation
// period of the unrolled walk: > WK, and a multiple of every prefetch depth and exchange group size).  Two frames are
// compiled: 17 taps (PSFs up to 17 x 17, the common case) and 33 taps (up to 33 x 33; twice the arithmetic per pixel, two
// columns per lane so that its 36 accumulator rows fit the register file at two waves per SIMD).  Which frame an
// operator takes follows from the support of ITS taps (SepOpInfo), not from the plan's (kh, kw): datasets whose PSFs
// were embedded in a common larger array walk in the frame their own PSF needs.
template <int WKT> struct Frame;
template <> struct Frame<17> { static constexpr int WK = 17, WH = 8, WS = 18; };
template <> struct Frame<33> { static constexpr int WK = 33, WH = 16, WS = 36; };
constexpr int WK = Frame<17>::WK, WH = Frame<17>::WH, WS = Frame<17>::WS;  // (the 17-tap frame: multi-component and joint kernels)
constexpr int XG_MAX = 6;  // rows per exchange group of the batched adjoint: 2, 3 or 6 (a divisor of WS, <= waves)
constexpr int XW = 8;   // waves (= datasets) per block of the batched adjoint
#ifndef JD_WALK_PREFETCH
#define JD_WALK_PREFETCH 2
#endif
constexpr int WALK_PREFETCH = JD_WALK_PREFETCH;  // rows the streaming loads run ahead of the arithmetic
// the batched adjoint on 4 columns per lane has one block of 8 waves per CU: its 32 KB in flight per CU at two rows
// ahead are the latency-bandwidth product of its 3.9 TB/s; three rows ahead: 80 -> 75 us in the fit (the forward launch
// loses 3 % with the extra registers: it stays at two)
constexpr int WALK_PREFETCH_ADJ = 3;
// Launches below this many (pixel, dataset) pairs stay with the tile kernel, which has four times the waves for the
// same image: measured on MI355X (tools/walk_check.py) forward + adjoint of one 2048^2 dataset 38.9 us (tile) against
// 51.8 (walk), of two 65 / 77, of four 129 / 112, of eight 282 / 199, of one 4096^2 dataset 138 / 118
constexpr size_t WALK_MIN_PIXELS = (size_t)1 << 24;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int C> struct Vec;
template <> struct Vec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float T __attribute__((ext_vector_type(4))); };

struct WalkArgs {
  const float* in;         // forward: flux; plain / adjoint: the image to convolve
  const float* in_scale;   // forward: exposure
  const float* op;         // operator buffer of sepconv.hip
  float* out;              // POISSON: g = d loss / d conv; plain: the result (+= when accumulate)
  const float* out_scale;  // plain: nullable
  const float* background;
  const float* counts;
  float* npred_out;  // nullable
  double* partials;  // POISSON: one per (dataset, tile)
  int H, W, strips, tiles_y, rows;
  int taps_u, taps_v;          // op offsets of the first row tap / first column tap of this direction
  int kh, kw, oy0, ox0;        // the plan's PSF size; image offset of its first stored tap (frame position WH + oy0)
  // batched forward launch over operators of BOTH frames: blocks < blocks17 walk the 17-tap datasets
  // table->order[0 .. n17) with the tiling above, the others the 33-tap datasets table->order[n17 ..) with this one
  int blocks17, n17;
  int ilv17, ilv33;            // > 0 (tuning, JD_SEP_INTERLEAVE): the ilv datasets of a frame are neighbours in the launch order
  int strips33, tiles_y33, rows33;
  int part_stride;             // POISSON: partial sums per dataset in `partials` (>= tiles of either tiling; the rest zeroed)
  float coef;
  int accumulate;
  float eps, inv_n;
  int write_grad;
  int n_batch;                 // > 0: per-dataset pointers come from `table`
  const SepBatchTable* table;
  int d_base;                  // batched adjoint: first dataset of this launch (wave w = dataset d_base + w)
  int n_comp, comp;            // batched adjoint: components per dataset and the one of this launch (table entry d * n_comp + comp)
  const double* fin_partials;  // batched adjoint: blocks < fin_n finalise the losses of the forward launch
  double fin_scale;
  int fin_count, fin_n;
  int* guard;                  // host-mapped: set when an operator is not rank 1
  int comp_blocks;             // batched adjoint over ALL components: blocks per component (0: component `comp` only)
  float* out_comp[4];          //   and their gradient images
};

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only keeps the COMPILER from moving them across
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// acc (+)= tap * h on packed fp32 (two pixels per instruction), the tap being the low (SEL = 0) or high (SEL = 1) half
// of a register pair broadcast to both elements through the operand-select bits -- hipcc has no pattern for that and
// otherwise spends a register pair per tap.  The same fused multiply-add (or plain product) per element as fmaf / *.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int SEL>
__device__ __forceinline__ v2f pk_fma_tap(v2f taps, v2f h, v2f acc) {
  if constexpr (SEL == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(taps), "v"(h));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(taps), "v"(h));
  return acc;
}
template <int SEL>
__device__ __forceinline__ v2f pk_mul_tap(v2f taps, v2f h) {
  v2f r;
  if constexpr (SEL == 0)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(taps), "v"(h));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(taps), "v"(h));
  return r;
}

// The same with the tap pair in SGPRs (uniform values: the row taps): no vector registers for the taps at all.
template <int SEL>
__device__ __forceinline__ v2f pk_fma_stap(v2f taps, v2f h, v2f acc) {
  if constexpr (SEL == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(taps), "v"(h));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(taps), "v"(h));
  return acc;
}
template <int SEL>
__device__ __forceinline__ v2f pk_mul_stap(v2f taps, v2f h) {
  v2f r;
  if constexpr (SEL == 0)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "s"(taps), "v"(h));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "s"(taps), "v"(h));
  return r;
}

#ifndef JD_WALK_PKROW
#define JD_WALK_PKROW 0  // 1: the row pass on packed FMAs (measured: no gain, see DESIGN_LOG.md); 0: the scalar row pass
#endif

// XG > 0: the batched adjoint (one wave per dataset -- at most XWT of them --, rows exchanged in groups of XG, wave
// w < XG adds up row w of a group); with a.comp_blocks > 0 the grid covers all flux components, comp_blocks blocks each.
// `bid`: the block's index within the blocks of its frame; (strips, tiles_y, rows): their tiling; `slot0`: position of
// the frame's first dataset in table->order (batched forward launches).
template <int WKT, int C, int P, bool POISSON, bool IN_SCALE, int XG, int XWT>
__device__ __forceinline__ void walk_body(const WalkArgs& a, int bid, const int strips, const int tiles_y, const int rows,
                                          const int slot0) {
#pragma clang fp contract(off)  // every fused multiply-add below is an explicit fmaf: batched and per-dataset paths round alike
  constexpr int WK = Frame<WKT>::WK, WH = Frame<WKT>::WH, WS = Frame<WKT>::WS;
  constexpr bool XCHG = XG > 0;
  static_assert(!XCHG || (WS % XG == 0 && XG <= XG_MAX), "the exchange group must divide the rotation period");
  typedef typename Vec<C>::T vC;
  constexpr int NX = 2 * WH / C;       // lanes that also load the right-hand halo piece
  constexpr int NWIN = 2 * WH + C;     // window floats per lane
  static_assert(64 * C + C * 64 <= 2 * 64 * C && NX <= 64, "row buffer");
  __shared__ __attribute__((aligned(16))) float rowbuf[XCHG ? XWT : 1][2 * 64 * C];  // 64 C + 2 WH floats used, the rest is a dump
  __shared__ __attribute__((aligned(16))) float xbuf[XCHG ? 2 * XG * XWT * 64 * C : 4];
  vC oprev;  // XCHG: the row of the gradient image this wave adds a group's row to, requested at the group's start
  __shared__ double fin_red[4];

  const int lane = threadIdx.x & 63;
  const int wv = XCHG ? (int)(threadIdx.x >> 6) : 0;
  const int nb = XCHG ? (int)(blockDim.x >> 6) : 1;

  if (XCHG && a.fin_partials && (int)blockIdx.x < a.fin_n) {  // (block-uniform; the host asks only with >= 256 threads)
    // finalize_rows_kernel's summation order: 256 strided threads, wave butterflies, the four wave sums in order
    if (threadIdx.x < 256) {
      const double* row = a.fin_partials + (size_t)blockIdx.x * a.fin_count;
      double s = 0.0;
      for (int i = threadIdx.x; i < a.fin_count; i += 256) s += row[i];
      s = wave_sum(s);
      if (lane == 0) fin_red[wv] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) total += fin_red[i];
      a.table->loss_out[blockIdx.x][0] = (float)(a.fin_scale * total + (double)a.table->loss_offset[blockIdx.x]);
    }
  }

  // consecutive tiles of one XCD (blockIdx % 8) are vertical neighbours in a strip: their halo rows hit in that L2
  const int n_tiles = strips * tiles_y;
  const int per_xcd = (n_tiles + 7) / 8;
  int comp = a.comp;
  if (XCHG && a.comp_blocks) comp = bid / a.comp_blocks, bid -= comp * a.comp_blocks;  // (comp_blocks: a multiple of 8)
  const int q = bid / 8;
  // batched forward launch: dataset-major, the datasets of this frame in the table's order
  // (JD_SEP_INTERLEAVE: tile-major instead -- the waves of one tile's datasets start together and read the flux rows they
  // share within a row or two of each other)
  const int ilv = !XCHG && a.n_batch > 0 ? (slot0 ? a.ilv33 : a.ilv17) : 0;
  const int dsel = (!XCHG && a.n_batch > 0) ? a.table->order[slot0 + (ilv ? q % ilv : q / per_xcd)] : 0;
  const int tile = (bid % 8) * per_xcd + (ilv ? q / ilv : q % per_xcd);
  if (tile >= n_tiles) return;  // (block-uniform)
  const int sx = tile / tiles_y, ty = tile - sx * tiles_y;

  // (global address space stated: pointers that come out of the table are generic to the compiler, and generic loads are
  // `flat_` instructions, which complete out of order and force a full s_waitcnt at every step)
  typedef const float __attribute__((address_space(1)))* gcp;
  typedef float __attribute__((address_space(1)))* gp;
  typedef const vC __attribute__((address_space(1)))* gcv;
  typedef vC __attribute__((address_space(1)))* gv;
  const bool batch = a.n_batch > 0;
  const int d = XCHG ? a.d_base + wv : dsel;
  const int slot = XCHG ? d * a.n_comp + comp : d;  // (a batched forward launch of this kernel has one component)
  const gcp in = (gcp)(batch && !POISSON ? a.table->g[slot] : a.in);
  const gcp in_scale = (gcp)(batch ? a.table->scale[slot] : a.in_scale);
  const gcp op = (gcp)(batch ? a.table->op[slot] : a.op);
  const gcp out_scale = (gcp)(batch ? a.table->scale[slot] : a.out_scale);
  const gcp background = (gcp)(batch ? a.table->bkg[d] : a.background);
  const gcp counts = (gcp)(batch ? a.table->cnt[d] : a.counts);
  const gp out = (gp)(POISSON && batch ? a.table->g[d] : XCHG && a.comp_blocks ? a.out_comp[comp] : a.out);
  const gp npred_out = (gp)a.npred_out;

  // taps -> SGPRs
  float tu[WK], tv[WK];
  {
    if (((int)op[0] != 1 || (int)op[1] > WK) && lane == 0) *a.guard = 1;  // not a rank-1 operator, or one whose own header asks for a wider frame (a registered buffer overwritten in place): the host reports it at its next call
    // (frame position t holds the stored tap t - WH - oy0: an operator whose PSF was embedded in a larger array of
    // zeros has its non-zero taps inside the frame the host chose for it, the taps outside are the zeros)
    float mu = 0.f, mv = 0.f;
    const int iu = lane - (WH + a.oy0), iv = lane - (WH + a.ox0);
    if (iu >= 0 && iu < a.kh) mu = op[a.taps_u + iu];
    if (iv >= 0 && iv < a.kw) mv = op[a.taps_v + iv];
#pragma unroll
    for (int t = 0; t < WK; ++t) {
      tu[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), t));
      tv[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mv), t));
    }
  }

  // the column taps once more, two per VGPR pair, for the packed FMAs of the column pass at 4 columns per lane (two
  // columns per lane in the 17-tap frame: scalar FMAs on the SGPR taps -- those kernels live on 128 registers; the
  // 33-tap frame has the registers of two waves per SIMD anyway); the empty asm keeps the compiler from moving the
  // (uniform) values back to SGPRs
  constexpr bool PK = C == 4 || WKT == 33;
  v2f tup[PK ? (WK + 1) / 2 : 1];
  if constexpr (PK) {
#pragma unroll
    for (int j = 0; j < (WK + 1) / 2; ++j) {
      float lo = tu[2 * j], hi = 2 * j + 1 < WK ? tu[2 * j + 1] : 0.f;
      asm volatile("" : "+v"(lo), "+v"(hi));
      tup[j] = v2f{lo, hi};
    }
  }

  const int X0 = sx * 64 * C;
  const int xm = X0 - WH + C * lane;           // image column of the lane's piece of the window
  const int xe = X0 - WH + 64 * C + C * lane;  // right-hand halo piece (lanes < NX)
  const int xo = X0 + C * lane;                // the lane's output columns
  const bool vm = xm >= 0 && xm < a.W, ve = lane < NX && xe < a.W, vo = xo < a.W;
  // lanes without a halo piece re-load the piece of lane (lane % NX) (same cache lines) and store it behind the window:
  // every lane issues the same loads and stores in every step, no branch -- the compiler can then count the loads in
  // flight exactly (s_waitcnt vmcnt(n)) instead of draining them at every merge point
  const int xd = X0 - WH + 64 * C + C * (lane % NX);
  const unsigned om = vm ? xm : 0, oe = ve ? xe : (xd < a.W ? xd : 0), oo = vo ? xo : 0;
  const int Y0 = ty * rows, y_end = min(Y0 + rows, a.H);
  const int r_begin = Y0 - WH, r_end = min(y_end + WH, a.H);  // image rows >= H contribute nothing
  float* rb = rowbuf[wv];

  struct Row { vC a, s, xa, xs; };
  struct Epi { vC p, q; };  // POISSON: background, counts; plain: out_scale, previous out
  auto row_ok = [&](int rr) { return rr >= 0 && rr < r_end; };
  // (rows outside [0, r_end) are loaded from the nearest row inside -- cache hits -- and never used)
  auto load_row = [&](int rr, Row& w) {
    const size_t base = (size_t)min(max(rr, 0), r_end - 1) * a.W;
    const gcp pin = in + base;
    w.a = *(gcv)(pin + om);
    if (IN_SCALE) w.s = *(gcv)(in_scale + base + om);
#ifdef JD_WALK_MASKED_HALO
    if (lane < NX)  // only the NX lanes with a halo piece load (and the compiler's load count becomes a bound)
#endif
    {
      w.xa = *(gcv)(pin + oe);
      if (IN_SCALE) w.xs = *(gcv)(in_scale + base + oe);
    }
  };
  auto epi_ok = [&](int y) { return y >= Y0 && y < y_end; };
  auto load_epi = [&](int y, Epi& e) {
    const size_t base = (size_t)min(max(y, Y0), y_end - 1) * a.W;
    if (POISSON) {
      e.p = *(gcv)(background + base + oo);
      e.q = *(gcv)(counts + base + oo);
    } else {
      if (out_scale) e.p = *(gcv)(out_scale + base + oo);
      if (!XCHG && a.accumulate) e.q = *(gcv)(out + base + oo);
    }
  };

  // WS = 18 accumulator slots for the 17 live outputs: with a rotation period of 18 the prefetch registers (period P,
  // P | 18) rotate statically too -- a register that is the target of a load in flight is never moved
  v2f acc[WS][C / 2];  // (pairs of columns: the operands of the packed FMAs)
#pragma unroll
  for (int s = 0; s < WS; ++s)
#pragma unroll
    for (int c = 0; c < C; ++c) acc[s][c / 2][c % 2] = 0.f;
  static_assert(WS % P == 0, "prefetch depth must divide the rotation period");
  Row pf[P];
  Epi ep[P];
  double loss = 0.0;

#pragma unroll
  for (int p = 0; p < P; ++p) {
    load_row(r_begin + p, pf[p]);
    load_epi(r_begin + p - WH, ep[p]);
  }

#if JD_WALK_PKROW
  // The ROW PASS on packed FMAs (round 4): two neighbouring outputs per instruction.  The operand pair of outputs
  // (c, c + 1) and tap t is (w[c + t], w[c + t + 1]): an ALIGNED register pair of the window for even c + t, a pair that
  // straddles two of them for odd.  So the window is read from LDS twice, at both alignments -- Wp[k] = (w[2k], w[2k+1])
  // and Wq[k] = (w[2k+1], w[2k+2]) -- which costs LDS reads (cheap: the launches follow their vector instructions, DESIGN
  // section 5) and no registers: the two alignments take the place of the two window sets of the round-3 software pipeline.
  // The taps are SGPR pairs, broadcast through the operand-select bits (pk_fma_stap).  Per element the same chain
  // tv[0] w[c], fma(tv[1], w[c + 1], .), ... as before: same bits.  17 taps, 4 columns: 34 instead of 68 instructions.
  // The pipeline: the window of row rr + 1 is requested right after the row pass of row rr has consumed the registers, and
  // arrives behind the column pass of row rr.
  constexpr int NP = NWIN / 2, NQ = NWIN / 2 - 1;
  typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
  v2f Wp[NP], Wq[NQ];
  // (the address of the shifted pairs goes through an empty asm: otherwise the compiler, which sees that they overlap the
  // aligned reads, builds most of them from those registers with two v_mov each -- ten vector instructions per row)
  int qoff = C * lane + 1;
  asm volatile("" : "+v"(qoff));
  const float* rbq = rb + qoff;
  v2f tvp[(WK + 1) / 2];
#pragma unroll
  for (int j = 0; j < (WK + 1) / 2; ++j) tvp[j] = v2f{tv[2 * j], 2 * j + 1 < WK ? tv[2 * j + 1] : 0.f};
  // (rows outside the image go through the exchange too -- clamped loads, never used: every step issues the same LDS
  // operations, so the compiler counts the ones in flight exactly instead of waiting for all of them at a join)
  auto produce = [&](int rr1, Row& nx) {
    // ---- products of the row -> LDS ----------------------------------------------------------------------------
    vC prod = nx.a;
    if (IN_SCALE) prod = prod * nx.s;
#pragma unroll
    for (int c = 0; c < C; ++c) prod[c] = vm ? prod[c] : 0.f;
    *reinterpret_cast<vC*>(rb + C * lane) = prod;
    vC px = nx.xa;
    if (IN_SCALE) px = px * nx.xs;
#pragma unroll
    for (int c = 0; c < C; ++c) px[c] = ve ? px[c] : 0.f;
    *reinterpret_cast<vC*>(rb + 64 * C + C * lane) = px;  // (lanes >= NX: behind the window, never read)
    load_row(rr1 + P, nx);  // the row P steps ahead takes this row's registers
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NWIN / C; ++k) {
      const vC t = *reinterpret_cast<const vC*>(rb + C * lane + C * k);
#pragma unroll
      for (int c = 0; c < C; ++c) Wp[(C * k + c) / 2][c % 2] = t[c];
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) Wq[k] = *reinterpret_cast<const v2u*>(rbq + 2 * k);
    wave_lds_fence();
  };
  produce(r_begin, pf[0]);
#else
  // The row exchange is software-pipelined: in step i the products of row rr + 1 go to LDS and its window is READ BACK
  // at once into the other of two window register sets, and only then the FMAs of row rr run on the window that was
  // requested one step earlier -- the LDS round trip (write, read, ~200 cycles that two waves per SIMD do not hide)
  // overlaps with 100-200 FMAs.  Measured with the read-back removed (wrong results, timing only): forward launch of
  // 2048^2 x 8 125 -> 85 us, adjoint 74 -> 58 us -- the round trip was a third of the launch.  LDS operations of one wave
  // execute in order, so one row buffer per wave still does (the next row's write cannot pass this row's read).
  float W[2][NWIN];
  // (rows outside the image go through the exchange too -- clamped loads, never used: every step issues the same LDS
  // operations, so the compiler counts the ones in flight exactly instead of waiting for all of them at a join)
  auto produce = [&](int rr1, Row& nx, float (&w)[NWIN]) {
    // ---- products of the row -> LDS ----------------------------------------------------------------------------
    vC prod = nx.a;
    if (IN_SCALE) prod = prod * nx.s;
#pragma unroll
    for (int c = 0; c < C; ++c) prod[c] = vm ? prod[c] : 0.f;
    *reinterpret_cast<vC*>(rb + C * lane) = prod;
    vC px = nx.xa;
    if (IN_SCALE) px = px * nx.xs;
#pragma unroll
    for (int c = 0; c < C; ++c) px[c] = ve ? px[c] : 0.f;
    *reinterpret_cast<vC*>(rb + 64 * C + C * lane) = px;  // (lanes >= NX: behind the window, never read)
    load_row(rr1 + P, nx);  // the row P steps ahead takes this row's registers
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NWIN / C; ++k) {
      const vC t = *reinterpret_cast<const vC*>(rb + C * lane + C * k);
#pragma unroll
      for (int c = 0; c < C; ++c) w[C * k + c] = t[c];
    }
    wave_lds_fence();
  };
  produce(r_begin, pf[0], W[0]);

#endif

  for (int r0 = r_begin; r0 < y_end + WH + (XCHG ? XG - 1 : 0); r0 += WS) {  // (+: the last group's flush step)
    static_for<WS>([&](auto ic) {  // (the 18 rotation states as compile-time constants: see walk_multi_kernel)
      constexpr int i = decltype(ic)::value;
      const int rr = r0 + i;
      Epi& ce = ep[i % P];
      const bool live = row_ok(rr);
#if JD_WALK_PKROW
      // ---- row pass: h[c] = sum_t tv[t] w[c + t], outputs in pairs ---------------------------------------------------
      v2f hp[C / 2];
      if (live) {
        // (two chains per output pair -- the even taps on the aligned window, the odd taps on the shifted one -- added
        // at the end: dependent packed FMAs issue a wait state apart, and one chain of 33 of them is latency, not work)
        v2f ho[C / 2];
        static_for<WK>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          static_for<C / 2>([&](auto jc) {
            constexpr int j = decltype(jc)::value, e = 2 * j + t;  // c + t of the pair's first output
            if constexpr (t == 0) hp[j] = pk_mul_stap<0>(tvp[0], Wp[e / 2]);
            else if constexpr (t == 1) ho[j] = pk_mul_stap<1>(tvp[0], Wq[e / 2]);
            else if constexpr (t % 2 == 0) hp[j] = pk_fma_stap<0>(tvp[t / 2], Wp[e / 2], hp[j]);
            else ho[j] = pk_fma_stap<1>(tvp[t / 2], Wq[e / 2], ho[j]);
          });
        });
#pragma unroll
        for (int j = 0; j < C / 2; ++j) hp[j] = hp[j] + ho[j];
      }
      produce(rr + 1, pf[(i + 1) % P]);
      if (live) {
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = hp[c / 2][c % 2];
#else
      produce(rr + 1, pf[(i + 1) % P], W[(i + 1) & 1]);
      if (live) {
        const float (&w)[NWIN] = W[i & 1];
        // ---- row pass: h[c] = sum_t tv[t] w[c + t] -----------------------------------------------------------
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = tv[0] * w[c];
#pragma unroll
        for (int t = 1; t < WK; ++t)
#pragma unroll
          for (int c = 0; c < C; ++c) h[c] = fmaf(tv[t], w[c + t], h[c]);
        v2f hp[C / 2];
#pragma unroll
        for (int c = 0; c < C; ++c) hp[c / 2][c % 2] = h[c];
#endif
        // ---- column pass, scatter form: out[rr + 8 - t] += tu[t] h; 4 columns per lane: on packed FMAs (the same fused
        // operation per element; with one or two waves per SIMD a packed FMA costs well under two scalar ones)
        static_for<WK>([&](auto tc) {
          constexpr int t = decltype(tc)::value, s = (i + WH - t + WS) % WS;
          if constexpr (PK) {
#pragma unroll
            for (int j = 0; j < C / 2; ++j)
              acc[s][j] = t == 0 ? pk_mul_tap<0>(tup[0], hp[j]) : pk_fma_tap<t % 2>(tup[t / 2], hp[j], acc[s][j]);
          } else {
#pragma unroll
            for (int c = 0; c < C; ++c)
              acc[s][c / 2][c % 2] = t == 0 ? tu[0] * h[c] : fmaf(tu[t], h[c], acc[s][c / 2][c % 2]);
          }
        });
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[(i + WH) % WS][c / 2][c % 2] = 0.f;
      }

      // ---- output row y = rr - 8 is complete ------------------------------------------------------------------
      const int y = rr - WH;
      if (epi_ok(y)) {
        vC conv;
#pragma unroll
        for (int c = 0; c < C; ++c) conv[c] = acc[(i + WS - WH) % WS][c / 2][c % 2];
        const size_t off = (size_t)y * a.W;
        if (POISSON) {
          vC gvec, nvec;
          float rowsum = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float n = fmaxf(conv[c], 0.f) + ce.p[c];  // clip, then the un-convolved background (npred.py:191)
            float term, g;
            poisson_point(n, ce.q[c], a.eps, a.inv_n, term, g);
            rowsum += term;
            nvec[c] = n;
            gvec[c] = conv[c] >= 0.f ? g : 0.f;  // clamp backward
          }
          if (vo) {
            loss += (double)rowsum;
            if (a.write_grad) *(gv)(out + off + oo) = gvec;
            if (npred_out) *(gv)(npred_out + off + oo) = nvec;
          }
        } else {
          vC pv;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            pv[c] = a.coef * conv[c];
            if (out_scale) pv[c] = pv[c] * ce.p[c];
          }
          if (!XCHG) {
            if (a.accumulate) pv = ce.q + pv;
            if (vo) *(gv)(out + off + oo) = pv;
          } else if constexpr (XCHG) {
            const int k = (y - Y0) / XG, gy = (i + WS - 2 * WH) % XG;  // (tiles start on a group boundary: gy is static)
            *reinterpret_cast<vC*>(xbuf + (size_t)((((k & 1) * XG + gy) * XWT + wv) * 64 + lane) * C) = pv;
          }
        }
      }
      if constexpr (XCHG) {
        // Every vector-memory operation of the exchange sits at a STATIC place of the unrolled loop and is issued by every
        // wave, needed or not (rows clamped into the tile): the compiler counts the loads in flight exactly and never has
        // to drain the prefetched rows to get at the gradient row.
        const int gy = (i + WS - 2 * WH) % XG;
        const int mine = wv < XG ? wv : XG - 1;  // the row of a group this wave adds up (waves >= XG: unused duplicates)
        if (gy == 0) {
          const int yy = min(max(y + mine, Y0), y_end - 1);
          oprev = *(gcv)(out + (size_t)yy * a.W + oo);
        }
        if (gy == XG - 1 && y - gy >= Y0 && y - gy < y_end) {  // (block-uniform) the group is complete
          __syncthreads();
          const int k = (y - Y0) / XG, yy = y - gy + mine;
          vC res = oprev;
          for (int dd = 0; dd < nb; ++dd) {
            const vC pd = *reinterpret_cast<const vC*>(xbuf + (size_t)((((k & 1) * XG + mine) * XWT + dd) * 64 + lane) * C);
            res = dd == 0 && !a.accumulate ? pd : res + pd;
          }
          if (vo && wv < XG && yy < y_end) *(gv)(out + (size_t)yy * a.W + oo) = res;
        }
      }
      load_epi(y + P, ce);
    });
  }
  if (POISSON) {
    loss = wave_sum(loss);
    // (a launch over both frames has two tilings: every dataset's row of partial sums is `part_stride` long, and the
    // entries beyond its own tile count are zeroed by its own waves -- exact in any summation order)
    const int stride = a.part_stride ? a.part_stride : n_tiles;
    if (lane == 0) {
      double* row = a.partials + (size_t)d * stride;
      row[tile] = loss;
      for (int t = tile + n_tiles; t < stride; t += n_tiles) row[t] = 0.0;
    }
  }
}

template <int WKT, int C, int P, bool POISSON, bool IN_SCALE, int XG, int XWT = XW>
__global__ __launch_bounds__(XG ? 64 * XWT : 64) void walk_kernel(WalkArgs a) {
  walk_body<WKT, C, P, POISSON, IN_SCALE, XG, XWT>(a, (int)blockIdx.x, a.strips, a.tiles_y, a.rows, 0);
}

// Batched forward launch over operators of both frames: the 17-tap datasets at C17 columns per lane, the 33-tap
// datasets behind them at two (every wave of the launch resident at once; the host sizes the two tilings so that the
// waves of both kinds take about as long).
template <int C17, int P>
__global__ __launch_bounds__(64) void walk_mixed_kernel(WalkArgs a) {
  if ((int)blockIdx.x < a.blocks17)  // (block-uniform)
    walk_body<17, C17, P, true, true, 0, XW>(a, (int)blockIdx.x, a.strips, a.tiles_y, a.rows, 0);
  else
    walk_body<33, 2, P, true, true, 0, XW>(a, (int)blockIdx.x - a.blocks17, a.strips33, a.tiles_y33, a.rows33, a.n17);
}

// ------------------------------------------------------------------------------------------------------------------
// Forward models + Poisson pass of datasets with SEVERAL flux components (models/npred.py:241-261: the components are
// clipped one by one, summed, then the background is added).  A block is one wave PER COMPONENT walking the same strip of
// the same dataset: every wave convolves its component exactly as walk_kernel does (same arithmetic, same bits), parks
// the finished rows in LDS in groups of MG, and after a barrier every wave runs the Poisson pass of the group's rows on
// the sum of all components' clipped rows and writes the gradient image of ITS component (masked where its own
// convolution is negative).  Background and counts are fetched by every wave (the second fetch is an L2 hit) so that
// each wave issues the same loads in every step (see walk_kernel).
constexpr int MG = 6;         // rows per group (a divisor of WS)
constexpr int MULTI_MAX = 4;  // components (= waves per block)

struct MultiArgs {
  const float* flux[MULTI_MAX];
  double* partials;
  int H, W, strips, tiles_y, rows;
  int taps_u, taps_v, kh, kw, oy0, ox0;  // (image offset of the first stored tap: frame position WH + oy0, see walk_body)
  float eps, inv_n;
  int write_grad, n_comp;
  const SepBatchTable* table;
  int* guard;
};

// (round 4: the frame is a template parameter of the WAVE's body -- a dataset whose components need different frames has
// waves of both kinds in its block; they meet at the same barriers, one pair per group of MG output rows)
template <int WKT, int C, int P>
__device__ __forceinline__ void walk_multi_body(const MultiArgs& a, float* multi_lds) {
#pragma clang fp contract(off)
  constexpr int WK = Frame<WKT>::WK, WH = Frame<WKT>::WH, WS = Frame<WKT>::WS;
  typedef typename Vec<C>::T vC;
  static_assert(WS % MG == 0 && WS % P == 0, "group size and prefetch depth must divide the rotation period");
  constexpr int NX = 2 * WH / C;
  constexpr int NWIN = 2 * WH + C;
  // LDS (dynamic, n_comp x 64 C x (2 + 3 MG) floats): per wave a row buffer; [row of the group][component][lane * C]
  // finished convolution rows; per wave [row][background | counts][lane * C]
  const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6), nc = a.n_comp;
  float* const cbuf = multi_lds + nc * 2 * 64 * C;
  float* const bcbase = cbuf + MG * nc * 64 * C;
  const int n_tiles = a.strips * a.tiles_y;
  const int per_xcd = (n_tiles + 7) / 8;
  const int q = blockIdx.x / 8;
  const int d = q / per_xcd;  // dataset-major grid
  const int tile = (blockIdx.x % 8) * per_xcd + q % per_xcd;
  if (tile >= n_tiles) return;  // (block-uniform)
  const int sx = tile / a.tiles_y, ty = tile - sx * a.tiles_y;

  typedef const float __attribute__((address_space(1)))* gcp;
  typedef float __attribute__((address_space(1)))* gp;
  typedef const vC __attribute__((address_space(1)))* gcv;
  typedef vC __attribute__((address_space(1)))* gv;
  const int slot = d * nc + wv;
  const gcp in = (gcp)(wv == 0 ? a.flux[0] : wv == 1 ? a.flux[1] : wv == 2 ? a.flux[2] : a.flux[3]);
  const gcp in_scale = (gcp)a.table->scale[slot];
  const gcp op = (gcp)a.table->op[slot];
  const gcp background = (gcp)a.table->bkg[d];
  const gcp counts = (gcp)a.table->cnt[d];
  const gp out = (gp)a.table->g[slot];

  float tu[WK], tv[WK];
  {
    if (((int)op[0] != 1 || (int)op[1] > WK) && lane == 0) *a.guard = 1;
    float mu = 0.f, mv = 0.f;
    const int iu = lane - (WH + a.oy0), iv = lane - (WH + a.ox0);
    if (iu >= 0 && iu < a.kh) mu = op[a.taps_u + iu];
    if (iv >= 0 && iv < a.kw) mv = op[a.taps_v + iv];
#pragma unroll
    for (int t = 0; t < WK; ++t) {
      tu[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), t));
      tv[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mv), t));
    }
  }

#ifndef JD_WALK_MULTI_PK
#define JD_WALK_MULTI_PK 1
#endif
  constexpr bool PK = (C == 4 && JD_WALK_MULTI_PK) || WKT == 33;  // packed FMAs in the column pass: see walk_kernel
  v2f tup[PK ? (WK + 1) / 2 : 1];
  if constexpr (PK) {
#pragma unroll
    for (int j = 0; j < (WK + 1) / 2; ++j) {
      float lo = tu[2 * j], hi = 2 * j + 1 < WK ? tu[2 * j + 1] : 0.f;
      asm volatile("" : "+v"(lo), "+v"(hi));
      tup[j] = v2f{lo, hi};
    }
  }

  const int X0 = sx * 64 * C;
  const int xm = X0 - WH + C * lane, xe = X0 - WH + 64 * C + C * lane, xo = X0 + C * lane;
  const bool vm = xm >= 0 && xm < a.W, ve = lane < NX && xe < a.W, vo = xo < a.W;
  const int xd = X0 - WH + 64 * C + C * (lane % NX);
  const unsigned om = vm ? xm : 0, oe = ve ? xe : (xd < a.W ? xd : 0), oo = vo ? xo : 0;
  const int Y0 = ty * a.rows, y_end = min(Y0 + a.rows, a.H);
  const int r_begin = Y0 - WH, r_end = min(y_end + WH, a.H);
  float* rb = multi_lds + wv * 2 * 64 * C;
  float* bc = bcbase + wv * MG * 2 * 64 * C;

  struct Row { vC a, s, xa, xs; };
  struct Epi { vC p, q; };
  auto row_ok = [&](int rr) { return rr >= 0 && rr < r_end; };
  auto load_row = [&](int rr, Row& w) {
    const size_t base = (size_t)min(max(rr, 0), r_end - 1) * a.W;
    w.a = *(gcv)(in + base + om);
    w.s = *(gcv)(in_scale + base + om);
    w.xa = *(gcv)(in + base + oe);
    w.xs = *(gcv)(in_scale + base + oe);
  };
  auto epi_ok = [&](int y) { return y >= Y0 && y < y_end; };
  auto load_epi = [&](int y, Epi& e) {
    const size_t base = (size_t)min(max(y, Y0), y_end - 1) * a.W;
    e.p = *(gcv)(background + base + oo);
    e.q = *(gcv)(counts + base + oo);
  };

  v2f acc[WS][C / 2];
#pragma unroll
  for (int s = 0; s < WS; ++s)
#pragma unroll
    for (int c = 0; c < C; ++c) acc[s][c / 2][c % 2] = 0.f;
  Row pf[P];
  Epi ep[P];
  double loss = 0.0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    load_row(r_begin + p, pf[p]);
    load_epi(r_begin + p - WH, ep[p]);
  }

  for (int r0 = r_begin; r0 < y_end + WH + MG - 1; r0 += WS) {
    // (the 18 rotation states as a compile-time sequence: with `#pragma unroll` this kernel's accumulators ended up in
    // scratch at C = 4 -- the unrolling came after the last scalar-replacement pass)
    static_for<WS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int rr = r0 + i;
      Row& cur = pf[i % P];
      Epi& ce = ep[i % P];
      const bool live = row_ok(rr);
      if (live) {
        vC prod = cur.a * cur.s;
#pragma unroll
        for (int c = 0; c < C; ++c) prod[c] = vm ? prod[c] : 0.f;
        *reinterpret_cast<vC*>(rb + C * lane) = prod;
        vC px = cur.xa * cur.xs;
#pragma unroll
        for (int c = 0; c < C; ++c) px[c] = ve ? px[c] : 0.f;
        *reinterpret_cast<vC*>(rb + 64 * C + C * lane) = px;
      }
      load_row(rr + P, cur);
      if (live) {
        wave_lds_fence();
        float w[NWIN];
#pragma unroll
        for (int k = 0; k < NWIN / C; ++k) {
          const vC t = *reinterpret_cast<const vC*>(rb + C * lane + C * k);
#pragma unroll
          for (int c = 0; c < C; ++c) w[C * k + c] = t[c];
        }
        wave_lds_fence();
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = tv[0] * w[c];
#pragma unroll
        for (int t = 1; t < WK; ++t)
#pragma unroll
          for (int c = 0; c < C; ++c) h[c] = fmaf(tv[t], w[c + t], h[c]);
        v2f hp[C / 2];
#pragma unroll
        for (int c = 0; c < C; ++c) hp[c / 2][c % 2] = h[c];
        static_for<WK>([&](auto tc) {
          constexpr int t = decltype(tc)::value, s = (i + WH - t + WS) % WS;
          if constexpr (PK) {
#pragma unroll
            for (int j = 0; j < C / 2; ++j)
              acc[s][j] = t == 0 ? pk_mul_tap<0>(tup[0], hp[j]) : pk_fma_tap<t % 2>(tup[t / 2], hp[j], acc[s][j]);
          } else {
#pragma unroll
            for (int c = 0; c < C; ++c)
              acc[s][c / 2][c % 2] = t == 0 ? tu[0] * h[c] : fmaf(tu[t], h[c], acc[s][c / 2][c % 2]);
          }
        });
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[(i + WH) % WS][c / 2][c % 2] = 0.f;
      }

      // ---- output row y = rr - 8 of this component is complete: park it (and this wave's copy of background, counts)
      const int y = rr - WH;
      constexpr int gy = (i + WS - 2 * WH) % MG;  // (y - Y0 = i - 16 mod 18: tiles start on a group boundary, static)
      if (epi_ok(y)) {
        vC conv;
#pragma unroll
        for (int c = 0; c < C; ++c) conv[c] = acc[(i + WS - WH) % WS][c / 2][c % 2];
        *reinterpret_cast<vC*>(cbuf + (size_t)((gy * nc + wv) * 64 + lane) * C) = conv;
        *reinterpret_cast<vC*>(bc + (size_t)((gy * 2 + 0) * 64 + lane) * C) = ce.p;
        *reinterpret_cast<vC*>(bc + (size_t)((gy * 2 + 1) * 64 + lane) * C) = ce.q;
      }
      load_epi(y + P, ce);
      if (gy == MG - 1 && y - gy >= Y0 && y - gy < y_end) {  // (block-uniform) the group is complete in every wave
        __syncthreads();
#pragma unroll
        for (int g2 = 0; g2 < MG; ++g2) {
          const int yy = y - gy + g2;
          if (yy >= y_end) continue;
          vC nsum;
#pragma unroll
          for (int c = 0; c < C; ++c) nsum[c] = 0.f;
          vC own = nsum;
          for (int k = 0; k < nc; ++k) {  // 0 + clip(conv_0) + clip(conv_1) + ...: the order of poisson_fused_kernel
            const vC cv = *reinterpret_cast<const vC*>(cbuf + (size_t)((g2 * nc + k) * 64 + lane) * C);
#pragma unroll
            for (int c = 0; c < C; ++c) nsum[c] += fmaxf(cv[c], 0.f);
            if (k == wv) own = cv;
          }
          const vC b = *reinterpret_cast<const vC*>(bc + (size_t)((g2 * 2 + 0) * 64 + lane) * C);
          const vC cn = *reinterpret_cast<const vC*>(bc + (size_t)((g2 * 2 + 1) * 64 + lane) * C);
          vC gvec;
          float rowsum = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float n = nsum[c] + b[c];
            float term, g;
            poisson_point(n, cn[c], a.eps, a.inv_n, term, g);
            rowsum += term;
            gvec[c] = own[c] >= 0.f ? g : 0.f;
          }
          if (vo) {
            if (wv == 0) loss += (double)rowsum;
            if (a.write_grad) *(gv)(out + (size_t)yy * a.W + oo) = gvec;
          }
        }
        __syncthreads();  // the group's LDS rows may be overwritten
      }
    });
  }
  if (wv == 0) {
    loss = wave_sum(loss);
    if (lane == 0) a.partials[(size_t)d * n_tiles + tile] = loss;
  }
}

// FRAMES = 17: every operator of the launch walks in the 17-tap frame; 33: some need the 33-tap frame (two columns per
// lane only) -- table->frame33 says which (bit d * n_comp + c)
template <int C, int P, int FRAMES>
__global__ __launch_bounds__(64 * MULTI_MAX) void walk_multi_kernel(MultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float multi_lds[];
  if constexpr (FRAMES == 33) {
    static_assert(C == 2, "the 33-tap frame runs two columns per lane");
    const int per_xcd = (a.strips * a.tiles_y + 7) / 8;
    const int d = (int)(blockIdx.x / 8) / per_xcd, wv = (int)(threadIdx.x >> 6);
    if ((a.table->frame33 >> (d * a.n_comp + wv)) & 1ull)  // (wave-uniform)
      walk_multi_body<33, C, P>(a, multi_lds);
    else
      walk_multi_body<17, C, P>(a, multi_lds);
  } else {
    walk_multi_body<17, C, P>(a, multi_lds);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The whole likelihood step of a joint fit in ONE launch: forward model, Poisson pass, adjoint convolution and the sum
// over the datasets.  A wave walks its strip once and carries TWO convolutions: the forward one (input row r -> output
// row r - 8) feeds the Poisson pass, whose g row goes -- through a second LDS row -- straight into the adjoint one
// (g row r - 8 -> gradient row r - 16).  The g images (4 B written + 4 B read per pixel and dataset) never exist and the
// exposure is fetched from HBM once (its second use, 16 rows later, is an L2 hit): 13 B per (pixel, dataset) from HBM
// instead of 29.  The adjoint needs g in an 8-pixel halo around its outputs, so a strip of 64 C forward columns yields
// 64 C - 16 gradient columns and a tile of R gradient rows walks R + 32 input rows; the redundant forward work in the
// halo is the price.  Same arithmetic in the same order as the two-launch path (walk_kernel forward + adjoint), so the
// gradient is the same bit for bit; the loss is summed over other tiles (equal to rounding).
// XG = 0: one dataset per launch (single wave per block, no exchange).
struct JointArgs {
  const float* flux;
  // one dataset (n_batch == 0)
  const float* exposure;
  const float* op;
  const float* background;
  const float* counts;
  float* grad;          // (+)= coef * sum_d E_d corr(g_d)
  double* partials;     // [dataset][tile] sums of n - c log(n + eps) over the tile's own pixels
  int H, W, strips, tiles_y, rows;
  int taps_u, taps_v, kh, kw, offy, offx;  // FORWARD taps in the 17-tap frame (the adjoint uses them reversed)
  float coef;
  int accumulate;
  float eps, inv_n;
  int n_batch, d_base;
  const SepBatchTable* table;
  int* guard;
};

#ifndef JD_JOINT_PREFETCH
#define JD_JOINT_PREFETCH 2
#endif
constexpr int JOINT_PREFETCH = JD_JOINT_PREFETCH;  // rows in flight per stream (a divisor of WS)

template <int C, int P, int XG>
__global__ __launch_bounds__(XG ? 64 * XW : 64) void walk_joint_kernel(JointArgs a) {
#pragma clang fp contract(off)
  typedef typename Vec<C>::T vC;
  constexpr bool XCHG = XG > 0;
  static_assert(!XCHG || (WS % XG == 0 && XG <= XG_MAX), "the exchange group must divide the rotation period");
  static_assert(WS % P == 0, "prefetch depth must divide the rotation period");
  constexpr int NX = 2 * WH / C;
  constexpr int NWIN = 2 * WH + C;
  constexpr int SW = 64 * C - 2 * WH;  // gradient columns per strip
  __shared__ __attribute__((aligned(16))) float rowbuf[XCHG ? XW : 1][2 * 64 * C];
  __shared__ __attribute__((aligned(16))) float gbuf[XCHG ? XW : 1][64 * C + 2 * WH];
  __shared__ __attribute__((aligned(16))) float xbuf[XCHG ? 2 * XG * XW * 64 * C : 4];

  const int lane = threadIdx.x & 63;
  const int wv = XCHG ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // (scalar: table pointers stay in SGPRs)
  const int nb = XCHG ? (int)(blockDim.x >> 6) : 1;
  const int n_tiles = a.strips * a.tiles_y;
  const int per_xcd = (n_tiles + 7) / 8;
  const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || tile >= n_tiles) return;  // (block-uniform)
  const int sx = tile / a.tiles_y, ty = tile - sx * a.tiles_y;

  typedef const float __attribute__((address_space(1)))* gcp;
  typedef float __attribute__((address_space(1)))* gp;
  typedef const vC __attribute__((address_space(1)))* gcv;
  typedef vC __attribute__((address_space(1)))* gv;
  typedef const char __attribute__((address_space(1)))* gcb;
  typedef char __attribute__((address_space(1)))* gb8;
  const bool batch = a.n_batch > 0;
  const int d = a.d_base + wv;
  const gcp flux = (gcp)a.flux;
  const gcp expo = (gcp)(batch ? a.table->scale[d] : a.exposure);
  const gcp op = (gcp)(batch ? a.table->op[d] : a.op);
  const gcp background = (gcp)(batch ? a.table->bkg[d] : a.background);
  const gcp counts = (gcp)(batch ? a.table->cnt[d] : a.counts);
  const gp out = (gp)a.grad;

  float tu[WK], tv[WK];
  {
    if (((int)op[0] != 1 || (int)op[1] > WK) && lane == 0) *a.guard = 1;
    float mu = 0.f, mv = 0.f;
    const int iu = lane - a.offy, iv = lane - a.offx;
    if (iu >= 0 && iu < a.kh) mu = op[a.taps_u + iu];
    if (iv >= 0 && iv < a.kw) mv = op[a.taps_v + iv];
#pragma unroll
    for (int t = 0; t < WK; ++t) {
      tu[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), t));
      tv[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mv), t));
    }
  }

  static_assert(C == 2, "the packed column passes below take one register pair per accumulator row");
  v2f tup[(WK + 1) / 2];  // the column taps in VGPR pairs for the packed FMAs (see walk_kernel)
#pragma unroll
  for (int j = 0; j < (WK + 1) / 2; ++j) {
    float lo = tu[2 * j], hi = 2 * j + 1 < WK ? tu[2 * j + 1] : 0.f;
    asm volatile("" : "+v"(lo), "+v"(hi));
    tup[j] = v2f{lo, hi};
  }

  const int X0 = sx * SW - WH;                 // first FORWARD-output column of the strip
  const int xm = X0 - WH + C * lane;           // the lane's piece of the product window
  const int xe = X0 - WH + 64 * C + C * lane;  // right-hand halo piece (lanes < NX)
  const int xo = X0 + C * lane;                // the lane's forward-output (g) columns = its gradient columns if inner
  const bool vm = xm >= 0 && xm < a.W, ve = lane < NX && xe >= 0 && xe < a.W, vg = xo >= 0 && xo < a.W;
  const bool inner = C * lane >= WH && C * lane < 64 * C - WH;
  const bool vo = inner && vg;
  const int xd = X0 - WH + 64 * C + C * (lane % NX);
  const unsigned om = vm ? xm : 0, oe = ve ? xe : (xd >= 0 && xd < a.W ? xd : 0), oo = vg ? xo : 0;
  // (uniform row pointer + 32-bit lane byte offset: the loads take the scalar-base addressing form, no 64-bit VALU adds)
  const unsigned omb = om * 4u, oeb = oe * 4u, oob = oo * 4u;
  auto ld = [](gcp p, size_t row, unsigned off) -> vC { return *(gcv)((gcb)(p + row) + off); };
  auto st = [](gp p, size_t row, unsigned off, vC v) { *(gv)((gb8)(p + row) + off) = v; };
  const int Y0 = ty * a.rows, y_end = min(Y0 + a.rows, a.H);
  const int r_begin = Y0 - 2 * WH, r_end = min(y_end + 2 * WH, a.H);
  const int g_lo = max(Y0 - WH, 0), g_hi = min(y_end + WH, a.H);  // g rows this tile needs that exist
  float* rb = rowbuf[wv];
  float* gb = gbuf[wv];

  struct Row { vC a, s, xa, xs; };
  struct EpiF { vC b, c; };
  auto row_ok = [&](int rr) { return rr >= 0 && rr < r_end; };
  auto load_row = [&](int rr, Row& w) {
    const size_t base = (size_t)min(max(rr, 0), r_end - 1) * a.W;
    w.a = ld(flux, base, omb);
    w.s = ld(expo, base, omb);
    w.xa = ld(flux, base, oeb);
    w.xs = ld(expo, base, oeb);
  };
  auto load_epf = [&](int y, EpiF& e) {
    const size_t base = (size_t)min(max(y, g_lo), g_hi - 1) * a.W;
    e.b = ld(background, base, oob);
    e.c = ld(counts, base, oob);
  };
  struct EpiA { vC e, o; };  // exposure and (one dataset per launch, accumulate) the previous gradient of a gradient row
  auto load_epa = [&](int y, EpiA& e) {
    const size_t base = (size_t)min(max(y, Y0), y_end - 1) * a.W;
    e.e = ld(expo, base, oob);
    if (!XCHG && a.accumulate) e.o = ld((gcp)out, base, oob);
  };

  vC accF[WS], accA[WS];
#pragma unroll
  for (int s = 0; s < WS; ++s)
#pragma unroll
    for (int c = 0; c < C; ++c) accF[s][c] = 0.f, accA[s][c] = 0.f;
  Row pf[P];
  EpiF epf[P];
  EpiA epa[P];
  vC oprev;
  double loss = 0.0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    load_row(r_begin + p, pf[p]);
    load_epf(r_begin + p - WH, epf[p]);
    load_epa(r_begin + p - 2 * WH, epa[p]);
  }
#pragma unroll
  for (int c = 0; c < C; ++c) oprev[c] = 0.f;

  for (int r0 = r_begin; r0 < y_end + 2 * WH + (XCHG ? XG - 1 : 0); r0 += WS) {
    static_for<WS>([&](auto ic) {  // (compile-time rotation state: the accumulators stay in registers)
      constexpr int i = decltype(ic)::value;
      const int rr = r0 + i;
      Row& cur = pf[i % P];
      EpiF& cf = epf[i % P];
      EpiA& ca = epa[i % P];
      // ---- forward convolution: input row rr ------------------------------------------------------------------
      const bool live = row_ok(rr);
      if (live) {
        vC prod = cur.a * cur.s;
#pragma unroll
        for (int c = 0; c < C; ++c) prod[c] = vm ? prod[c] : 0.f;
        *reinterpret_cast<vC*>(rb + C * lane) = prod;
        vC px = cur.xa * cur.xs;
#pragma unroll
        for (int c = 0; c < C; ++c) px[c] = ve ? px[c] : 0.f;
        *reinterpret_cast<vC*>(rb + 64 * C + C * lane) = px;
      }
      load_row(rr + P, cur);
      if (live) {
        wave_lds_fence();
        float w[NWIN];
#pragma unroll
        for (int k = 0; k < NWIN / C; ++k) {
          const vC t = *reinterpret_cast<const vC*>(rb + C * lane + C * k);
#pragma unroll
          for (int c = 0; c < C; ++c) w[C * k + c] = t[c];
        }
        wave_lds_fence();
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = tv[0] * w[c];
#pragma unroll
        for (int t = 1; t < WK; ++t)
#pragma unroll
          for (int c = 0; c < C; ++c) h[c] = fmaf(tv[t], w[c + t], h[c]);
        vC hv;
#pragma unroll
        for (int c = 0; c < C; ++c) hv[c] = h[c];
        static_for<WK>([&](auto tc) {  // (packed FMAs: the same fused operation per element)
          constexpr int t = decltype(tc)::value, s = (i + WH - t + WS) % WS;
          accF[s] = t == 0 ? pk_mul_tap<0>(tup[0], hv) : pk_fma_tap<t % 2>(tup[t / 2], hv, accF[s]);
        });
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) accF[(i + WH) % WS][c] = 0.f;
      }

      // ---- Poisson pass on forward-output row y1 = rr - 8; g is zero outside the image --------------------------
      const int y1 = rr - WH;
      const bool grow = y1 >= g_lo && y1 < g_hi;  // (rows above g_lo only feed gradient rows this tile does not own)
      if (grow) {
        const vC conv = accF[(i + WS - WH) % WS];
        vC gvec;
        float rowsum = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float n = fmaxf(conv[c], 0.f) + cf.b[c];
          float term, g;
          poisson_point(n, cf.c[c], a.eps, a.inv_n, term, g);
          rowsum += term;
          gvec[c] = vg && conv[c] >= 0.f ? g : 0.f;
        }
        if (vo && y1 >= Y0 && y1 < y_end) loss += (double)rowsum;
        *reinterpret_cast<vC*>(gb + WH + C * lane) = gvec;
      }
      load_epf(y1 + P, cf);
      // ---- adjoint convolution: g row y1 -> gradient rows y1 - 8 .. y1 + 8 (taps reversed) -----------------------
      if (grow) {
        wave_lds_fence();
        float w[NWIN];
#pragma unroll
        for (int k = 0; k < NWIN / C; ++k) {
          const vC t = *reinterpret_cast<const vC*>(gb + C * lane + C * k);
#pragma unroll
          for (int c = 0; c < C; ++c) w[C * k + c] = t[c];
        }
        wave_lds_fence();
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = tv[WK - 1] * w[c];
#pragma unroll
        for (int t = 1; t < WK; ++t)
#pragma unroll
          for (int c = 0; c < C; ++c) h[c] = fmaf(tv[WK - 1 - t], w[c + t], h[c]);
        vC hv;
#pragma unroll
        for (int c = 0; c < C; ++c) hv[c] = h[c];
        static_for<WK>([&](auto tc) {
          constexpr int t = decltype(tc)::value, s = (i - t + WS) % WS;
          constexpr int r = WK - 1 - t;  // (taps reversed)
          accA[s] = t == 0 ? pk_mul_tap<r % 2>(tup[r / 2], hv) : pk_fma_tap<r % 2>(tup[r / 2], hv, accA[s]);
        });
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) accA[i % WS][c] = 0.f;
      }

      // ---- gradient row y2 = rr - 16 is complete --------------------------------------------------------------
      const int y2 = rr - 2 * WH;
      if (y2 >= Y0 && y2 < y_end) {
        const vC corr = accA[(i + WS - 2 * WH) % WS];
        vC pv;
#pragma unroll
        for (int c = 0; c < C; ++c) pv[c] = (a.coef * corr[c]) * ca.e[c];
        if constexpr (!XCHG) {
          if (a.accumulate) pv = ca.o + pv;
          if (vo) st(out, (size_t)y2 * a.W, oob, pv);
        } else {
          const int k = (y2 - Y0) / XG, gy = (i + 2 * WS - 4 * WH) % XG;
          *reinterpret_cast<vC*>(xbuf + (size_t)((((k & 1) * XG + gy) * XW + wv) * 64 + lane) * C) = pv;
        }
      }
      if constexpr (XCHG) {
        const int gy = (i + 2 * WS - 4 * WH) % XG;  // (y2 - Y0 = i - 32 mod 18: tiles start on a group boundary, static)
        const int mine = wv < XG ? wv : XG - 1;
        if (gy == 0) {
          const int yy = min(max(y2 + mine, Y0), y_end - 1);
          oprev = ld((gcp)out, (size_t)yy * a.W, oob);
        }
        if (gy == XG - 1 && y2 - gy >= Y0 && y2 - gy < y_end) {  // (block-uniform) the group is complete
          __syncthreads();
          const int k = (y2 - Y0) / XG, yy = y2 - gy + mine;
          vC res = oprev;
          for (int dd = 0; dd < nb; ++dd) {
            const vC pd = *reinterpret_cast<const vC*>(xbuf + (size_t)((((k & 1) * XG + mine) * XW + dd) * 64 + lane) * C);
            res = dd == 0 && !(a.accumulate || a.d_base > 0) ? pd : res + pd;
          }
          if (vo && wv < XG && yy < y_end) st(out, (size_t)yy * a.W, oob, res);
        }
      }
      load_epa(y2 + P, ca);
    });
  }
  loss = wave_sum(loss);
  if (lane == 0) a.partials[(size_t)d * n_tiles + tile] = loss;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Do the walk kernels take a launch over n datasets of (H, W) pixels at all?  Option JD_SEP_WALK: 0 never, 1 whenever
// the operators allow (then independent of n: a batched step and the per-dataset calls it stands for choose alike and
// agree bit for bit), unset: where they are the faster kernels -- which depends on n, so that by default a batched step
// at a large size and its per-dataset form agree to fp32 rounding only.
bool walk_enabled(int H, int W, int n) {
  const int mode = opt_value(OPT_SEP_WALK, -1);
  if (mode == 0 || W % 4 != 0) return false;
  if (mode < 0 && (size_t)H * W * n < WALK_MIN_PIXELS) return false;
  return true;
}

// Do taps with the support `info` of a plan (kh, kw, oy, ox) fit a frame of WKf taps around position WHf, in both
// directions?  (Frame position of the stored row tap i: WHf + oy0 + i, of the stored column tap j: WHf + ox0 + j.)
bool frame_fits(const SepOpInfo& info, int kh, int kw, int oy, int ox, int WHf, int WKf) {
  for (int adj = 0; adj < 2; ++adj) {
    const SepGeom g = sep_geom(kh, kw, oy, ox, adj != 0);
    if (WHf + g.oy0 + info.ulo[adj] < 0 || WHf + g.oy0 + info.uhi[adj] > WKf) return false;
    if (WHf + g.ox0 + info.vlo[adj] < 0 || WHf + g.ox0 + info.vhi[adj] > WKf) return false;
  }
  return true;
}

int frame_of_info(const SepOpInfo& info, int kh, int kw, int oy, int ox) {
  if (info.rank != 1) return 0;
  if (frame_fits(info, kh, kw, oy, ox, Frame<17>::WH, Frame<17>::WK)) return 17;
  if (frame_fits(info, kh, kw, oy, ox, Frame<33>::WH, Frame<33>::WK)) return 33;
  return 0;
}

// The frame (17 or 33 taps) the walk kernels run a REGISTERED rank-1 operator of the plan geometry in, 0: none
int walk_frame(const void* op, int kh, int kw, int oy, int ox) {
  SepOpInfo info;
  if (!sep_operator_info(op, &info)) return 0;
  return frame_of_info(info, kh, kw, oy, ox);
}

// the widest support a plan of this geometry can hold (a full kh x kw PSF)
int plan_frame(int kh, int kw, int oy, int ox) {
  SepOpInfo full;
  full.rank = 1;
  for (int adj = 0; adj < 2; ++adj) {
    const SepGeom g = sep_geom(kh, kw, oy, ox, adj != 0);
    full.ulo[adj] = 0, full.uhi[adj] = kh, full.vlo[adj] = g.shiftx, full.vhi[adj] = g.shiftx + kw;
  }
  if (frame_fits(full, kh, kw, oy, ox, Frame<17>::WH, Frame<17>::WK)) return 17;
  if (frame_fits(full, kh, kw, oy, ox, Frame<33>::WH, Frame<33>::WK)) return 33;
  return 0;
}

int device_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}

// tap addressing of one direction (frame independent: the kernels place the taps in their frame themselves)
void walk_setup(WalkArgs& a, int kh, int kw, int oy, int ox, int adjoint) {
  const SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
  const int taps_off = 4 + (adjoint ? (int)((sep_conv_operator_floats() - 4) / 2) : 0);
  a.kh = kh, a.kw = kw;
  a.oy0 = g.oy0, a.ox0 = g.ox0 + g.shiftx;
  a.taps_u = taps_off, a.taps_v = taps_off + g.khp + g.shiftx;
}

void walk_tiles(WalkArgs& a, int C, int rows) {
  a.rows = rows;
  a.strips = (a.W + 64 * C - 1) / (64 * C);
  a.tiles_y = (a.H + rows - 1) / rows;
}

// (C, rows) of a launch over `n` datasets.  Measured inside the fit (tools/ab.py, MI355X): the launch is fastest with as
// many waves as are resident at once -- 8 per CU at C = 4 (218 registers: two per SIMD) -- and no second round: forward
// + Poisson launch of 2048^2 x 8 at C = 4 (after the packed column pass), rows 56 / 65 / 74 / 83 / 92 / 110 / 128 =
// 137 / 118 / 119 / 128 / 122 / 123 / 125 us (9.25 / 8 / 7 / 6.25 / 5.75 / 4.75 / 4 waves per CU); 4096^2 x 1 forward,
// rows 38 / 56 / 74 = 77 / 81 / 86 us (6.75 waves per CU at 38).  Rows = k WS - 2 WH: the walk advances in rounds of WS
// rows (18 k - 16 in the 17-tap frame, 36 k - 32 in the 33-tap frame, which always runs two columns per lane).
void walk_shape(const WalkArgs& a, int n, bool adjoint, int frame, int* C, int* rows) {
  const long want = (long)(7.25 * device_cus());
  auto waves = [&](int c, int r) { return (long)((a.W + 64 * c - 1) / (64 * c)) * ((a.H + r - 1) / r) * n; };
  const int ws = frame == 33 ? Frame<33>::WS : Frame<17>::WS;
  if (frame == 33) {
    *C = 2, *rows = 2 * ws - 2 * Frame<33>::WH;
  } else {
    *C = waves(4, 38) >= want * 5 / 8 ? 4 : 2;
    *rows = 38;
  }
  while (*rows < 4096 && waves(*C, *rows) > want) *rows += ws;
  const int oc = opt_value(adjoint ? OPT_SEP_WALK_ADJ_COLS : OPT_SEP_WALK_COLS, 0);
  const int orows = opt_value(adjoint ? OPT_SEP_WALK_ADJ_ROWS : OPT_SEP_WALK_ROWS, 0);
  if ((oc == 2 || oc == 4) && frame != 33) *C = oc;
  if (orows >= 20 && orows <= 4096) *rows = orows;
}

template <bool POISSON, bool IN_SCALE>
int launch_walk(WalkArgs a, int frame, int C, int n_grid, hipStream_t stream) {
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  const int n_tiles = a.strips * a.tiles_y;
  const unsigned blocks = (unsigned)(((n_tiles + 7) / 8) * 8 * n_grid);
  ProfScope prof(POISSON ? JD_KERNEL_POISSON_FUSED : JD_KERNEL_SEP_CONV, stream);
  if (frame == 33)
    hipLaunchKernelGGL((walk_kernel<33, 2, WALK_PREFETCH, POISSON, IN_SCALE, 0>), dim3(blocks), dim3(64), 0, stream, a);
  else if (C == 4)
    hipLaunchKernelGGL((walk_kernel<17, 4, WALK_PREFETCH, POISSON, IN_SCALE, 0>), dim3(blocks), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, POISSON, IN_SCALE, 0>), dim3(blocks), dim3(64), 0, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// frame of one (dataset, component) entry of a batch: operator registered with rank 1 and a support that fits a frame,
// every image 16-byte aligned; 0: not a walk case
int dataset_frame(const SepBatchTable& t, int i, int d, int kh, int kw, int oy, int ox) {
  if (!(aligned16(t.scale[i]) && aligned16(t.g[i]) && aligned16(t.bkg[d]) && aligned16(t.cnt[d]))) return 0;
  return walk_frame(t.op[i], kh, kw, oy, ox);
}

}  // namespace

bool walk_takes_launch(int H, int W, int n_datasets, int kh, int kw, int oy, int ox) {
  return walk_enabled(H, W, n_datasets) && plan_frame(kh, kw, oy, ox) != 0;
}

int walk_operator_frame(const float* op, int kh, int kw, int oy, int ox) { return walk_frame(op, kh, kw, oy, ox); }

int walk_info_frame(const SepOpInfo& info, int kh, int kw, int oy, int ox) { return frame_of_info(info, kh, kw, oy, ox); }

// order <- the datasets with the 17-tap operators first (in dataset order), then the others; *n17 <- how many
void walk_batch_order(SepBatchTable& table, int n, int n_comp, int kh, int kw, int oy, int ox) {
  int k = 0;
  table.n17 = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int d = 0; d < n; ++d) {
      const bool is17 = n_comp != 1 || walk_frame(table.op[d], kh, kw, oy, ox) != 33;
      if (is17 == (pass == 0)) table.order[k++] = d;
      if (is17 && pass == 0) ++table.n17;
    }
  for (; k < SEP_MAX_BATCH; ++k) table.order[k] = 0;
  table.frame33 = 0;
  for (int i = 0; i < n * n_comp && i < 64; ++i)
    if (walk_frame(table.op[i], kh, kw, oy, ox) == 33) table.frame33 |= 1ull << i;
}

bool sep_batch_is_mixed(int n, int n_comp, const SepBatchTable& table, int H, int W, int kh, int kw, int oy, int ox) {
  // (the per-dataset calls a mixed batch falls back to decide with n = 1)
  if (n_comp > MULTI_MAX || !walk_enabled(H, W, n * n_comp)) return false;  // no dataset takes the walk kernels
  int yes = 0;
  for (int d = 0; d < n; ++d)
    for (int c = 0; c < n_comp; ++c) {
      const int f = dataset_frame(table, d * n_comp + c, d, kh, kw, oy, ox);
      yes += f != 0 ? 1 : 0;
    }
  return yes != 0 && yes != n * n_comp;
}

// out (+)= coef * out_scale * conv/corr_same(in * in_scale, psf)      [launch_sep_conv's contract]
int walk_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H, int W,
              int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, hipStream_t stream) {
  if (!walk_enabled(H, W, 1)) return JD_WALK_NOT_TAKEN;
  const int frame = walk_frame(op, kh, kw, oy, ox);
  if (!frame) return JD_WALK_NOT_TAKEN;
  if (!aligned16(in) || !aligned16(in_scale) || !aligned16(out) || !aligned16(out_scale)) return JD_WALK_NOT_TAKEN;
  WalkArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = out, a.out_scale = out_scale;
  a.H = H, a.W = W, a.coef = coef, a.accumulate = accumulate;
  walk_setup(a, kh, kw, oy, ox, adjoint);
  int C, rows;
  walk_shape(a, 1, adjoint != 0, frame, &C, &rows);
  walk_tiles(a, C, rows);
  return in_scale ? launch_walk<false, true>(a, frame, C, 1, stream) : launch_walk<false, false>(a, frame, C, 1, stream);
}

// launch_sep_conv_poisson's contract; *n_partials = partial sums written
int walk_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh, int kw,
                      int oy, int ox, const float* background, const float* counts, float* npred_out, double* partials,
                      float eps, float inv_n, int write_grad, int* n_partials, hipStream_t stream) {
  if (!walk_enabled(H, W, 1) || !in_scale) return JD_WALK_NOT_TAKEN;
  const int frame = walk_frame(op, kh, kw, oy, ox);
  if (!frame) return JD_WALK_NOT_TAKEN;
  if (!aligned16(in) || !aligned16(in_scale) || !aligned16(g_out) || !aligned16(background) || !aligned16(counts) ||
      !aligned16(npred_out))
    return JD_WALK_NOT_TAKEN;
  WalkArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = g_out;
  a.H = H, a.W = W, a.coef = 1.f;
  a.background = background, a.counts = counts, a.npred_out = npred_out, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad;
  walk_setup(a, kh, kw, oy, ox, 0);
  int C, rows;
  walk_shape(a, 1, false, frame, &C, &rows);
  walk_tiles(a, C, rows);
  *n_partials = a.strips * a.tiles_y;
  return launch_walk<true, true>(a, frame, C, 1, stream);
}

// One flux component: all forward models + Poisson passes of a joint step in one launch (dataset-major grid, the
// datasets in table.order: 17-tap operators first); *n_partials = partial sums per dataset,
// partials[d * *n_partials + tile]
int walk_conv_poisson_batch(int n, const float* flux, const SepBatchTable& table, const SepBatchTable* table_dev, int H,
                            int W, int kh, int kw, int oy, int ox, double* partials, float eps, float inv_n,
                            int write_grad, int* n_partials, hipStream_t stream) {
  if (!walk_enabled(H, W, n) || !aligned16(flux)) return JD_WALK_NOT_TAKEN;
  int n17 = 0, n33 = 0;
  for (int d = 0; d < n; ++d) {
    const int f = dataset_frame(table, d, d, kh, kw, oy, ox);
    if (!f) return JD_WALK_NOT_TAKEN;
    (f == 17 ? n17 : n33)++;
  }
  if (n17 != table.n17) return fail(JD_ERR_INVALID, "walk_conv_poisson_batch: the table's dataset order is stale");
  WalkArgs a{};
  a.in = flux, a.H = H, a.W = W, a.coef = 1.f, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad, a.n_batch = n, a.table = table_dev;
  walk_setup(a, kh, kw, oy, ox, 0);
  int C, rows;
  if (n17 == 0 || n33 == 0) {
    const int frame = n33 ? 33 : 17;
    walk_shape(a, n, false, frame, &C, &rows);
    walk_tiles(a, C, rows);
    *n_partials = a.strips * a.tiles_y;
    if (opt_is_set(OPT_SEP_INTERLEAVE)) a.ilv17 = n;
    return launch_walk<true, true>(a, frame, C, n, stream);
  }
  // Both frames in one launch, all waves resident at once.  A 17-tap wave at 4 columns per lane issues ~226 vector
  // instructions per row of 256 pixels, a 33-tap wave at 2 columns ~COST33 per row of 128; the two tilings are chosen so
  // that tile time (cost per row x (rows + warm-up rows)) is about equal and the waves fill the chip once.  Measured
  // inside the fit (tools/ab.py, 2048^2, 6 + 2 datasets): COST33 = 150 / 190 / 230 / 260 / 280 -> 139.5 / 124.4-126.9 /
  // 123.0-124.9 / 122.8-124.4 / 125.0 us.
  const double cost17_4 = 226.0, cost17_2 = 150.0;
  const double cost33 = (double)opt_value(OPT_SEP_WALK_COST33, 230);
  const long want = (long)(7.25 * device_cus());
  auto tiles = [&](int c, int r) { return (long)((W + 64 * c - 1) / (64 * c)) * ((H + r - 1) / r); };
  C = tiles(4, 38) * n17 + tiles(2, 40) * n33 >= want * 5 / 8 ? 4 : 2;
  const int oc = opt_value(OPT_SEP_WALK_COLS, 0);
  if (oc == 2 || oc == 4) C = oc;
  const double cost17 = C == 4 ? cost17_4 : cost17_2;
  int rows33 = 40;
  rows = 38;
  for (int r17 = 38; r17 < 4096; r17 += Frame<17>::WS) {
    // the 33-tap tile height (36 k - 32) whose tile time is nearest that of the 17-tap tile
    const double t17 = cost17 * (r17 + 2 * Frame<17>::WH);
    int k = (int)std::lround(t17 / cost33 / Frame<33>::WS);  // (rows + 2 WH = k WS)
    if (k < 2) k = 2;
    const int r33 = k * Frame<33>::WS - 2 * Frame<33>::WH;
    rows = r17, rows33 = r33;
    if (tiles(C, r17) * n17 + tiles(2, r33) * n33 <= want) break;
  }
  const int orows = opt_value(OPT_SEP_WALK_ROWS, 0), orows33 = opt_value(OPT_SEP_WALK_ROWS33, 0);
  if (orows >= 20 && orows <= 4096) rows = orows;
  if (orows33 >= 20 && orows33 <= 4096) rows33 = orows33;
  walk_tiles(a, C, rows);
  a.rows33 = rows33, a.strips33 = (W + 127) / 128, a.tiles_y33 = (H + rows33 - 1) / rows33;
  const int t17 = a.strips * a.tiles_y, t33 = a.strips33 * a.tiles_y33;
  a.n17 = n17;
  if (opt_is_set(OPT_SEP_INTERLEAVE)) a.ilv17 = n17, a.ilv33 = n33;
  a.blocks17 = ((t17 + 7) / 8) * 8 * n17;
  a.part_stride = t17 > t33 ? t17 : t33;
  *n_partials = a.part_stride;
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  const unsigned blocks = (unsigned)(a.blocks17 + ((t33 + 7) / 8) * 8 * n33);
  ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);
  if (C == 4)
    hipLaunchKernelGGL((walk_mixed_kernel<4, WALK_PREFETCH>), dim3(blocks), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((walk_mixed_kernel<2, WALK_PREFETCH>), dim3(blocks), dim3(64), 0, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// Several flux components: forward models + Poisson passes of all datasets in one launch, one wave per component
// (walk_multi_kernel); *n_partials = partial sums per dataset
int walk_conv_poisson_batch_multi(int n, int n_comp, const float* const* flux, const SepBatchTable& table,
                                  const SepBatchTable* table_dev, int H, int W, int kh, int kw, int oy, int ox,
                                  double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                                  hipStream_t stream) {
  if (n_comp < 2 || n_comp > MULTI_MAX) return JD_WALK_NOT_TAKEN;
  for (int c = 0; c < n_comp; ++c)
    if (!aligned16(flux[c])) return JD_WALK_NOT_TAKEN;
  if (!walk_enabled(H, W, n * n_comp) || n * n_comp > 64) return JD_WALK_NOT_TAKEN;
  bool any33 = false;
  for (int d = 0; d < n; ++d)
    for (int c = 0; c < n_comp; ++c) {
      const int f = dataset_frame(table, d * n_comp + c, d, kh, kw, oy, ox);
      if (!f) return JD_WALK_NOT_TAKEN;
      any33 = any33 || f == 33;
    }
  MultiArgs a{};
  for (int c = 0; c < n_comp; ++c) a.flux[c] = flux[c];
  a.partials = partials, a.H = H, a.W = W, a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad, a.n_comp = n_comp;
  a.table = table_dev;
  const SepGeom g = sep_geom(kh, kw, oy, ox, false);
  a.oy0 = g.oy0, a.ox0 = g.ox0 + g.shiftx;
  a.kh = kh, a.kw = kw, a.taps_u = 4, a.taps_v = 4 + g.khp + g.shiftx;
  // 4 columns per lane (184 registers: 2 waves per SIMD) unless that leaves CUs without a block; a launch with operators
  // of the 33-tap frame: 2 columns per lane for every block (its 36 accumulator rows: 2 waves per SIMD at that width)
  int C = opt_value(OPT_SEP_WALK_COLS, 0);
  const int per_cu4 = 8 / n_comp;
  auto blocks_of = [&](int c, int r) { return (long)((W + 64 * c - 1) / (64 * c)) * ((H + r - 1) / r) * n; };
  if (C != 2 && C != 4) C = blocks_of(4, 36) >= (long)device_cus() * per_cu4 / 2 ? 4 : 2;
  if (any33) C = 2;
  const int per_cu = (C == 4 || any33 ? 8 : 16) / n_comp;  // (126 registers at C = 2 in the 17-tap frame: 4 waves per SIMD)
  // Rows per tile (a multiple of MG).  Measured inside the fit, 2048^2 x 16 x 2 components, C = 4: 72 / 96 / 120 / 144 /
  // 168 / 192 / 240 / 294 rows = 432 / 434 / 456 / 416-436 / 465 / 505 / 550 / 510 us: unlike the one-component launch this
  // one prefers SEVERAL rounds of short blocks to one round of long ones, and loses what its last round leaves empty
  // (120 rows: 2.25 rounds).  The tallest tile of 72-168 rows whose last round is at least 90 % full.
  const long slots = (long)device_cus() * (per_cu < 1 ? 1 : per_cu);
  int rows = 72;
  double best = 0.0;
  for (int r = 168; r >= 72; r -= MG) {
    const double rounds = (double)blocks_of(C, r) / (double)slots;
    const double eff = rounds / std::ceil(rounds);
    if (eff >= 0.9) {
      rows = r;
      break;
    }
    if (eff > best) best = eff, rows = r;
  }
  if (rows > (H + MG - 1) / MG * MG) rows = (H + MG - 1) / MG * MG;
  const int orows = opt_value(OPT_SEP_WALK_ROWS, 0);
  if (orows >= 18) rows = orows;
  rows = (rows + MG - 1) / MG * MG;
  a.rows = rows, a.strips = (W + 64 * C - 1) / (64 * C), a.tiles_y = (H + rows - 1) / rows;
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  const int n_tiles = a.strips * a.tiles_y;
  *n_partials = n_tiles;
  const unsigned blocks = (unsigned)(((n_tiles + 7) / 8) * 8 * n);
  const size_t lds = (size_t)n_comp * 64 * C * (2 + 3 * MG) * sizeof(float);
  ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);
  if (C == 4) {
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
      JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(walk_multi_kernel<4, WALK_PREFETCH, 17>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_set = lds;
    }
    hipLaunchKernelGGL((walk_multi_kernel<4, WALK_PREFETCH, 17>), dim3(blocks), dim3(64 * n_comp), lds, stream, a);
  } else if (any33) {
    hipLaunchKernelGGL((walk_multi_kernel<2, WALK_PREFETCH, 33>), dim3(blocks), dim3(64 * n_comp), lds, stream, a);
  } else {
    hipLaunchKernelGGL((walk_multi_kernel<2, WALK_PREFETCH, 17>), dim3(blocks), dim3(64 * n_comp), lds, stream, a);
  }
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// grad (+)= coef * sum_d scale[d] * corr_same(g[d], psf_d), the datasets added in order: one wave per dataset, up to 8
// datasets per launch, later chunks accumulate (the same additions in the same order).  Datasets whose operators walk
// in different frames go to different launches: the batch is cut into runs of consecutive datasets of one frame (the
// benchmark's 6 + 2), so the order of the additions stays the dataset order.  With fin_partials, blocks d < n of the
// first launch also turn the fin_count partial sums of dataset d into its loss (*fin_done <- 1)
int walk_conv_adjoint_batch(int n, int n_comp, int comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                            float* grad, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                            hipStream_t stream, const double* fin_partials, double fin_scale, int fin_count, int* fin_done) {
  *fin_done = 0;
  if (!walk_enabled(H, W, n * n_comp) || !aligned16(grad)) return JD_WALK_NOT_TAKEN;
  int frames[SEP_MAX_BATCH];
  for (int d = 0; d < n; ++d)
    for (int c = 0; c < n_comp; ++c) {  // (all components: the forward launch of the step must have been a walk launch too)
      const int f = dataset_frame(table, d * n_comp + c, d, kh, kw, oy, ox);
      if (!f) return JD_WALK_NOT_TAKEN;
      if (c == comp) frames[d] = f;
    }
  WalkArgs a{};
  a.out = grad, a.H = H, a.W = W, a.coef = coef, a.table = table_dev, a.n_comp = n_comp, a.comp = comp;
  walk_setup(a, kh, kw, oy, ox, 1);
  // Block shape.  6-8 datasets: 4 columns per lane (1 KB per row and stream), exchange groups of 6 rows: 112 KB of LDS, ONE
  // block of 6-8 waves per CU; fewer datasets: 2 columns per lane, as many blocks per CU as LDS (exchange buffer) and
  // registers (112-127: 4 waves per SIMD) allow.  Rows per tile: a multiple of every exchange group size, the smallest that
  // leaves all blocks resident at once.  Measured inside the fit at 2048^2 x 8 (tools/ab.py): C = 4, rows 36 / 66 / 72 /
  // 84 = 98 / 77.5 / 85 / 89 us (66 rows: 8 strips x 32 tiles = one block for every CU); C = 2 (two blocks per CU), rows
  // 54 / 72 / 90 = 111 / 85 / 97 us.  The 33-tap frame: two columns per lane, two waves per SIMD.
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  for (int d0 = 0; d0 < n;) {
    const int frame = frames[d0];
    int m = 1;
    while (d0 + m < n && m < XW && frames[d0 + m] == frame) ++m;  // datasets (= waves) of this launch: its block shape follows
    if (frame == 33 && opt_value(OPT_SEP_WALK_ADJ33, 0) == 1) m = 1;  // (tuning: one plain accumulating launch per dataset)
    const int xg = m >= 6 ? 6 : m >= 3 ? 3 : 2;
    const bool wide = frame == 17 && m >= 6 && opt_value(OPT_SEP_WALK_ADJ_COLS, 4) != 2;
    const int C = wide ? 4 : 2;
    const int lds = (2 * xg * XW * 64 * C + XW * 128 * C) * 4;  // exchange buffer + row buffers
    int per_cu = 160 * 1024 / lds;
    // (204 registers at C = 4 and in the 33-tap frame: 2 waves per SIMD; 112-127 at C = 2 in the 17-tap frame: 4)
    const int by_regs = (wide || frame == 33 ? 8 : 16) / m;
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu < 1) per_cu = 1;
    const long slots = (long)device_cus() * per_cu * (per_cu > 1 ? 15 : 16) / 16;
    const int strips = (W + 64 * C - 1) / (64 * C);
    int rows = 36;
    while (rows < 4096 && (long)strips * ((H + rows - 1) / rows) > slots) rows += 6;
    const int orows = opt_value(frame == 33 ? OPT_SEP_WALK_ADJ_ROWS33 : OPT_SEP_WALK_ADJ_ROWS, 0);
    if (orows >= 18) rows = orows;
    rows = (rows + 5) / 6 * 6;
    walk_tiles(a, C, rows);
    const int n_tiles = a.strips * a.tiles_y;
    const unsigned blocks = (unsigned)(((n_tiles + 7) / 8) * 8);
    a.d_base = d0, a.n_batch = m, a.accumulate = d0 == 0 ? accumulate : 1;
    a.fin_partials = nullptr;
    if (d0 == 0 && fin_partials && m >= 4 && (int)blocks >= n) {  // (the fold needs 256 threads and a block per dataset)
      a.fin_partials = fin_partials, a.fin_scale = fin_scale, a.fin_count = fin_count, a.fin_n = n;
      *fin_done = 1;
    }
    ProfScope prof(JD_KERNEL_SEP_CONV, stream);
    if (frame == 33 && m >= 6)
      hipLaunchKernelGGL((walk_kernel<33, 2, WALK_PREFETCH, false, false, 6>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (frame == 33 && m >= 3)
      hipLaunchKernelGGL((walk_kernel<33, 2, WALK_PREFETCH, false, false, 3>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (frame == 33 && m == 2)
      hipLaunchKernelGGL((walk_kernel<33, 2, WALK_PREFETCH, false, false, 2>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m >= 6 && wide)
      hipLaunchKernelGGL((walk_kernel<17, 4, WALK_PREFETCH_ADJ, false, false, 6>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m >= 6)
      hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 6>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m >= 3)
      hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 3>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m == 2)
      hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 2>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else {  // a single dataset: the plain walk (same arithmetic: out (+)= (coef * corr) * scale)
      WalkArgs b = a;
      const int slot = d0 * n_comp + comp;
      b.n_batch = 0, b.table = nullptr, b.in = table.g[slot], b.op = table.op[slot], b.out_scale = table.scale[slot];
      int C1, r1;
      walk_shape(b, 1, true, frame, &C1, &r1);
      if (frame == 33 && orows >= 18) r1 = orows;
      walk_tiles(b, C1, r1);
      const unsigned blocks1 = (unsigned)(((b.strips * b.tiles_y + 7) / 8) * 8);
      if (frame == 33)
        hipLaunchKernelGGL((walk_kernel<33, 2, WALK_PREFETCH, false, false, 0>), dim3(blocks1), dim3(64), 0, stream, b);
      else if (C1 == 4)
        hipLaunchKernelGGL((walk_kernel<17, 4, WALK_PREFETCH, false, false, 0>), dim3(blocks1), dim3(64), 0, stream, b);
      else
        hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 0>), dim3(blocks1), dim3(64), 0, stream, b);
    }
    JD_LAUNCH_CHECK();
    d0 += m;
  }
  return JD_OK;
}

// The same sum for ALL flux components of up to 16 datasets in ONE launch: grads[c] (+)= coef * sum_d scale[d, c] *
// corr_same(g[d, c], psf_(d, c)).  The grid is n_comp x tiles, a block is one wave per dataset (up to 16: a full 1024
// threads, two columns per lane so that 128 registers do), the datasets are added in order as above -- the same bits as
// the per-component launches in chunks of 8, which at the benchmark's two-component case (1024^2 x 16 x 2) meant four
// launches that each filled under half of the chip.  Option JD_SEP_WALK_ADJ_ALL = 0: never.
constexpr int XW2 = 16;
int walk_conv_adjoint_batch_all(int n, int n_comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                                float* const* grads, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                                hipStream_t stream, const double* fin_partials, double fin_scale, int fin_count,
                                int* fin_done) {
  static_assert(MULTI_MAX <= 4, "WalkArgs::out_comp");
  *fin_done = 0;
  if (opt_value(OPT_SEP_WALK_ADJ_ALL, 1) == 0) return JD_WALK_NOT_TAKEN;
  if (n < 2 || n > XW2 || n_comp < 1 || n_comp > MULTI_MAX || (n <= XW && n_comp < 2) || !table_dev) return JD_WALK_NOT_TAKEN;
  // One component, 9-16 datasets: two launches of the 4-column kernel are the faster form where they fill the chip
  // (2048^2 x 16: 2 x 75.7 us against 163 us), this one where even 36-row tiles leave CUs without a block (1024^2 x
  // 16: 60 us against 2 x 49 us).  Two components at 2048^2 x 16: 296 us against 4 x 78 us.
  if (n_comp == 1 && (long)((W + 255) / 256) * ((H + 35) / 36) >= device_cus()) return JD_WALK_NOT_TAKEN;
  WalkArgs a{};
  for (int c = 0; c < n_comp; ++c) {
    if (!aligned16(grads[c])) return JD_WALK_NOT_TAKEN;
    a.out_comp[c] = grads[c];
  }
  if (!walk_enabled(H, W, n * n_comp)) return JD_WALK_NOT_TAKEN;
  for (int d = 0; d < n; ++d)
    for (int c = 0; c < n_comp; ++c)  // (one launch = one frame: the 17-tap one)
      if (dataset_frame(table, d * n_comp + c, d, kh, kw, oy, ox) != 17) return JD_WALK_NOT_TAKEN;
  a.out = grads[0], a.H = H, a.W = W, a.coef = coef, a.table = table_dev, a.n_comp = n_comp, a.comp = 0;
  walk_setup(a, kh, kw, oy, ox, 1);
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  const bool big = n > XW;
  const int xg = n >= 6 ? 6 : n >= 3 ? 3 : 2;
  const bool wide = !big && n >= 6 && opt_value(OPT_SEP_WALK_ADJ_COLS, 4) != 2;
  const int C = wide ? 4 : 2;
  const int xw = big ? XW2 : XW;
  const int lds = (2 * xg * xw * 64 * C + xw * 128 * C) * 4;
  int per_cu = 160 * 1024 / lds;
  const int by_regs = (big ? 16 : wide ? 8 : 16) / n;
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  const long slots = (long)device_cus() * per_cu * (per_cu > 1 ? 15 : 16) / 16;
  const int strips = (W + 64 * C - 1) / (64 * C);
  int rows = 36;
  while (rows < 4096 && (long)strips * ((H + rows - 1) / rows) * n_comp > slots) rows += 6;
  const int orows = opt_value(OPT_SEP_WALK_ADJ_ROWS, 0);
  if (orows >= 18) rows = orows;
  rows = (rows + 5) / 6 * 6;
  walk_tiles(a, C, rows);
  const int n_tiles = a.strips * a.tiles_y;
  a.comp_blocks = ((n_tiles + 7) / 8) * 8;
  const unsigned blocks = (unsigned)a.comp_blocks * n_comp;
  a.d_base = 0, a.n_batch = n, a.accumulate = accumulate;
  if (fin_partials && n >= 4 && (int)blocks >= n) {
    a.fin_partials = fin_partials, a.fin_scale = fin_scale, a.fin_count = fin_count, a.fin_n = n;
    *fin_done = 1;
  }
  ProfScope prof(JD_KERNEL_SEP_CONV, stream);
  if (big)
    hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 6, XW2>), dim3(blocks), dim3(64 * n), 0, stream, a);
  else if (wide)
    hipLaunchKernelGGL((walk_kernel<17, 4, WALK_PREFETCH_ADJ, false, false, 6>), dim3(blocks), dim3(64 * n), 0, stream, a);
  else if (n >= 6)
    hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 6>), dim3(blocks), dim3(64 * n), 0, stream, a);
  else if (n >= 3)
    hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 3>), dim3(blocks), dim3(64 * n), 0, stream, a);
  else
    hipLaunchKernelGGL((walk_kernel<17, 2, WALK_PREFETCH, false, false, 2>), dim3(blocks), dim3(64 * n), 0, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// The likelihood step of n datasets (one flux component) in one launch per 8 datasets: loss partial sums (*n_partials per
// dataset) and grad (+)= coef * sum_d E_d corr(g_d).  JD_WALK_NOT_TAKEN when the fused kernel is not eligible or not the
// faster choice (option JD_SEP_JOINT: 0 never, 1 whenever eligible).
int walk_joint_step(int n, const float* flux, const SepBatchTable& table, const SepBatchTable* table_dev, float* grad, int H,
                    int W, int kh, int kw, int oy, int ox, double* partials, float eps, float inv_n, float coef,
                    int accumulate, int* n_partials, hipStream_t stream) {
  // Measured inside the fit (tools/ab.py, 2048^2 x 8): 201-205 us against 125 + 77.5 us for the forward and the adjoint
  // launch -- the kernel saves 45 % of the HBM traffic but issues 40 % more instructions (halo, two columns per lane), and
  // with its 150 registers only two waves per SIMD hide each other's latencies; 4 datasets: 134 against 63 + 49 us.
  // Packed FMAs in the two column passes (13 % fewer vector instructions) took the kernel alone from 201 to 197 us:
  // it is not the instruction count that bounds it.  So it runs only on request (option JD_SEP_JOINT = 1); its results
  // are those of the two-launch path bit for bit.
  const int mode = opt_value(OPT_SEP_JOINT, 0);
  if (mode != 1 || !aligned16(flux) || !aligned16(grad)) return JD_WALK_NOT_TAKEN;
  if (!walk_enabled(H, W, n)) return JD_WALK_NOT_TAKEN;
  for (int d = 0; d < n; ++d)
    if (dataset_frame(table, d, d, kh, kw, oy, ox) != 17) return JD_WALK_NOT_TAKEN;
  if (!table_dev && n != 1) return JD_WALK_NOT_TAKEN;  // (without a device table: one dataset, pointers by value)
  JointArgs a{};
  a.flux = flux, a.grad = grad, a.partials = partials, a.H = H, a.W = W, a.coef = coef, a.eps = eps, a.inv_n = inv_n;
  const SepGeom g = sep_geom(kh, kw, oy, ox, false);
  a.offy = WH + g.oy0, a.offx = WH + g.ox0 + g.shiftx;
  a.kh = kh, a.kw = kw, a.taps_u = 4, a.taps_v = 4 + g.khp + g.shiftx;
  constexpr int C = 2;
  a.strips = (W + (64 * C - 2 * WH) - 1) / (64 * C - 2 * WH);
  // datasets per block (= waves; option JD_SEP_JOINT_CHUNK): two launches of 4 were measured slower than one of 8
  // (2 x 134 against 201 us)
  int chunk = opt_value(OPT_SEP_JOINT_CHUNK, 0);
  if (chunk < 1 || chunk > XW) chunk = n < XW ? n : XW;
  const int m0 = n < chunk ? n : chunk;
  // rows per tile (a multiple of 6): all blocks resident at once
  int per_cu = 12 / m0;
  const int xg = m0 >= 6 ? 6 : m0 >= 3 ? 3 : m0 >= 2 ? 2 : 0;
  if (xg) {
    const int lds = (2 * xg * XW * 64 * C + XW * (2 * 64 * C + 64 * C + 2 * WH)) * 4;
    if (per_cu > 160 * 1024 / lds) per_cu = 160 * 1024 / lds;
  }
  long slots = (long)device_cus() * per_cu;
  if (m0 == 1) slots = slots * 7 / 12;  // single waves: ~7 per CU (see walk_shape)
  else slots = slots * 31 / 32;
  int rows = 36;
  while (rows < 8192 && (long)a.strips * ((H + rows - 1) / rows) > slots) rows += 6;
  const int orows = opt_value(OPT_SEP_JOINT_ROWS, 0);
  if (orows >= 18) rows = (orows + 5) / 6 * 6;
  a.rows = rows, a.tiles_y = (H + rows - 1) / rows;
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  const int n_tiles = a.strips * a.tiles_y;
  *n_partials = n_tiles;
  const unsigned blocks = (unsigned)(((n_tiles + 7) / 8) * 8);
  for (int d0 = 0; d0 < n; d0 += chunk) {
    const int m = n - d0 < chunk ? n - d0 : chunk;
    a.d_base = d0, a.n_batch = m, a.table = table_dev, a.accumulate = d0 == 0 ? accumulate : 1;
    if (!table_dev) {
      a.n_batch = 0, a.exposure = table.scale[0], a.op = table.op[0], a.background = table.bkg[0], a.counts = table.cnt[0];
    }
    ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);
    if (m >= 6)
      hipLaunchKernelGGL((walk_joint_kernel<C, JOINT_PREFETCH, 6>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m >= 3)
      hipLaunchKernelGGL((walk_joint_kernel<C, JOINT_PREFETCH, 3>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else if (m == 2)
      hipLaunchKernelGGL((walk_joint_kernel<C, JOINT_PREFETCH, 2>), dim3(blocks), dim3(64 * m), 0, stream, a);
    else
      hipLaunchKernelGGL((walk_joint_kernel<C, JOINT_PREFETCH, 0>), dim3(blocks), dim3(64), 0, stream, a);
    JD_LAUNCH_CHECK();
  }
  return JD_OK;
}

}  // namespace jd
