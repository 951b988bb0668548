// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   forward / data-gradient : conv_mfma_kernel   GEMM M = output pixels, N = out channels, K = taps x in channels
//                             (tiles 128x128 [16 waves], 128x64 [8 waves], 128x32 and 64x64 [4 waves])
//   weight gradient         : wgrad_mfma_kernel  GEMM M = anchor channels, N = gathered channels, K = pixels
//
// Activations are NHWC so a 16-byte global load fetches 4 consecutive input channels of one
// pixel; tiles are staged through LDS ([row][k] with k fastest, row stride BK+4 floats which makes
// ds_read_b128 fragment reads conflict free) and double buffered with register prefetch so that one
// barrier per K step suffices. The K order inside a step is permuted identically for A and B so one
// ds_read_b128 feeds four MFMAs (see the comment at the fragment reads).
#include "hwg_common.h"
#include "wino_pack.h"
#include <stdlib.h>

namespace {

struct ConvK {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q;
  int mode;        // 0: convolution gather, 1: fractionally strided (transposed, stride>1) by parity class, 2: fractionally strided with the parity
                   // classes MERGED into the GEMM's N dimension (R % sh == 0, S % sw == 0: every class has R/sh x S/sw taps, and - with each class's
                   // block grid shifted by floor((class + pad) / stride) - they all read the SAME input pixels: one stride-1 (R/sh x S/sw)-tap
                   // convolution to classes * K channels, written depth-to-space. The operand is read once instead of once per class and a
                   // 64-channel layer fills 128 x 128 tiles)
  int KN;          // extent of the GEMM's N dimension: K, in mode 2 classes * K
  int Pm, Qm;      // mode 2: rows / columns of the shared block grid
  int accumulate;
  int ntm;         // M tiles of the largest parity class
  int ntm_pad;     // ntm rounded up to a multiple of 8 (XCD remap)
  int nsplit;      // split-K: the taps x channel-steps loop is cut into nsplit ranges, one workgroup each (blockIdx.y % nsplit)
  float* part;     // nsplit > 1: partial outputs [nsplit][N*P*Q*K] in y's layout, summed (+bias) by the last wavefront to arrive (cnt) or by conv_split_reduce_kernel
  int* cnt;        // nsplit > 1: arrival counters, one per (tile, wavefront sub-tile), zero between launches; nullptr = separate reduce launch
  int dbg;         // ABL kernel only (HWG_CONV_DBG, timing ablations of the 128 x 128 tile - results are garbage): 1 no global loads in the loop, 2 no LDS
                   // stores, 4 no barrier, 8 no MFMAs / fragment reads, 16 no fragment reads (MFMAs on stale registers), 32 nothing (the ABL build itself).
                   // Round 5 (tools/probes/probe_r5_abl7.txt, warm clocks, 4x66x1026x64->128 4x4 stride 2): 147.6 us as shipped; MFMAs alone 130 us
                   // (132 TFLOP/s - the ceiling of this tiling at the clock the chip sustains, 0.84 of the nominal peak); the shipped kernel runs at
                   // 0.88 of that ceiling. A single-basic-block rolled pipeline (burst interleaved with the MFMAs, barrier behind the last fragment
                   // read) and a start offset between co-resident workgroups both measured +-1 % and were removed again.
};

// PF = register prefetch depth in K steps. 1: the loads of step t+1 are issued before the MFMAs of step t and consumed right after them
// (their latency has ONE step of matrix-core work to hide behind: enough for the 16-wave 128x128 tile, whose co-resident waves cover
// for each other). 2: the loads of step t+2 are issued before the MFMAs of step t (two register sets): small layers run one 4-wave
// workgroup per CU, a step is ~1000 matrix-core cycles and an L2 / HBM round trip 2000-4000, so with depth 1 such a workgroup sits in
// s_waitcnt for most of every step.
// WK > 1: WK wavefronts share every (M, N) sub-tile and split each K step between them (their partial sums meet in LDS once, at the end):
// a second wavefront per SIMD for layers whose grid is one small workgroup per CU, without a second launch to add partial images.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int PF = 1, int WK = 1, bool ABL = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WK) void conv_mfma_kernel(ConvK a) {
  constexpr int NT = 64 * WAVES_M * WAVES_N * WK;  // 4, 8 or 16 wavefronts per workgroup
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  static_assert(MI >= 1 && NI >= 1, "wave tile must be at least 32x32");
  constexpr int LD = BK + 4;
  constexpr int KC = BK / 4;
  constexpr int A_F4 = BM * KC, B_F4 = BN * KC;
  constexpr int A_IT = (A_F4 + NT - 1) / NT, B_IT = (B_F4 + NT - 1) / NT;

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
  // fractionally strided modes: (sample, block row, block column) of the tile's BM rows, decoded once per workgroup for the epilogue (every lane
  // used to decode each of its 16 x MI rows itself: three divisions by run-time values per row)
  __shared__ int row_n[BM], row_p[BM], row_q[BM];
  float* As = smem;
  float* Bs = smem + 2 * BM * LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;

  // XCD-aware tile order: the dispatcher places block b on XCD b%8, give each XCD a contiguous run
  // of M tiles so neighbouring tiles (shared halo rows, same weights) hit the same L2.
  // (ranges differ by at most one tile: XCD i owns tiles [i*ntm/8, (i+1)*ntm/8))
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int tile_m = (xcd * a.ntm >> 3) + (bid >> 3);
  if (tile_m >= ((xcd + 1) * a.ntm >> 3)) return;
  const int split = blockIdx.y % a.nsplit;
  const int n0 = (blockIdx.y / a.nsplit) * BN;

  int cp = 0, cq = 0, Pc = a.P, Qc = a.Q;
  int r0 = 0, s0 = 0, tr = 1, ts = 1, nr = a.R, ns = a.S;
  int ah, bh, ch, aw, bw, cw;
  if (a.mode == 0) {
    ah = a.sh; bh = -a.ph; ch = a.dh;
    aw = a.sw; bw = -a.pw; cw = a.dw;
  } else if (a.mode == 2) {
    Pc = a.Pm; Qc = a.Qm;
    tr = a.sh; ts = a.sw;
    nr = a.R / a.sh; ns = a.S / a.sw;
    ah = 1; bh = 0; ch = -1;
    aw = 1; bw = 0; cw = -1;
  } else {
    const int cls = blockIdx.z;
    cp = cls / a.sw; cq = cls % a.sw;
    Pc = (a.P - cp + a.sh - 1) / a.sh;
    Qc = (a.Q - cq + a.sw - 1) / a.sw;
    r0 = (cp + a.ph) % a.sh; s0 = (cq + a.pw) % a.sw;
    tr = a.sh; ts = a.sw;
    nr = (a.R - r0 + a.sh - 1) / a.sh; if (nr < 0) nr = 0;
    ns = (a.S - s0 + a.sw - 1) / a.sw; if (ns < 0) ns = 0;
    ah = 1; bh = (cp + a.ph - r0) / a.sh; ch = -1;
    aw = 1; bw = (cq + a.pw - s0) / a.sw; cw = -1;
  }
  if (Pc <= 0 || Qc <= 0) return;
  const int Mc = a.N * Pc * Qc;
  const int m0 = tile_m * BM;
  if (m0 >= Mc) return;

  // per-thread gather rows
  int a_hb[A_IT], a_wb[A_IT], a_nb[A_IT], a_row[A_IT], a_chk[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int f = tid + it * NT;
    const int row = f / KC;
    a_row[it] = row;
    a_chk[it] = f % KC;
    const int m = m0 + row;
    const bool ok = (f < A_F4) && (m < Mc);
    a_ok[it] = ok;
    const int mm = ok ? m : 0;
    const int qi = mm % Qc;
    const int t = mm / Qc;
    const int pi = t % Pc;
    const int n = t / Pc;
    a_hb[it] = ah * pi + bh;
    a_wb[it] = aw * qi + bw;
    a_nb[it] = n * a.H * a.W;
  }
  int b_row[B_IT], b_chk[B_IT], b_base[B_IT];     // b_base: row of the [tap][K][C] filter image this B row reads at the class's first tap
  bool b_ok[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int f = tid + it * NT;
    b_row[it] = f / KC;
    b_chk[it] = f % KC;
    const int col = n0 + b_row[it];
    b_ok[it] = (f < B_F4) && (col < a.KN);
    if (a.mode == 2) {
      const int cls = col / a.K, kk = col - cls * a.K;
      const int rc = (cls / a.sw + a.ph) % a.sh, sc = (cls % a.sw + a.pw) % a.sw;
      b_base[it] = (rc * a.S + sc) * a.K + kk;
    } else {
      b_base[it] = (r0 * a.S + s0) * a.K + col;
    }
  }

  const int csteps = a.C / BK;
  const int Tall = nr * ns * csteps;
  const int t_begin = (int)((long long)Tall * split / a.nsplit);
  const int T = (int)((long long)Tall * (split + 1) / a.nsplit) - t_begin;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  constexpr int NSET = PF == 2 ? 2 : 1;
  float4 ra[NSET][A_IT], rb[NSET][B_IT];
  bool ma[NSET][A_IT];   // validity of the A loads in flight (applied when the tile is stored, so the wait for the loads sits after the MFMAs)
  // coordinates of the tile being LOADED (this workgroup's K range starts at step t_begin)
  int c0 = (t_begin % csteps) * BK;
  int js = (t_begin / csteps) % (ns > 0 ? ns : 1);
  int jr = (t_begin / csteps) / (ns > 0 ? ns : 1);
  const float* __restrict__ xg = a.x;
  const float* __restrict__ wg = a.w;

  auto load_tile = [&](float4* ra, float4* rb, bool* ma) {
    const int tap = tr * jr * a.S + ts * js;      // tap offset from the class's first tap (which b_base holds)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int ih = a_hb[it] + ch * jr;
      const int iw = a_wb[it] + cw * js;
      const bool v = a_ok[it] && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      // always issue the load (from a valid dummy address when masked): no exec-mask branches between the loads of one tile, so they
      // all go out back to back; the zero fill happens in store_tile
      const long long off = v ? ((long long)(a_nb[it] + ih * a.W + iw)) * a.C + c0 + a_chk[it] * 4 : 0;
      ra[it] = *reinterpret_cast<const float4*>(xg + off);
      ma[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const long long off = b_ok[it] ? ((long long)tap * a.K + b_base[it]) * a.C + c0 + b_chk[it] * 4 : 0;
      rb[it] = *reinterpret_cast<const float4*>(wg + off);
    }
    // advance to the next tile
    c0 += BK;
    if (c0 >= a.C) {
      c0 = 0;
      if (++js >= ns) { js = 0; ++jr; }
    }
  };
  auto store_tile = [&](int buf, const float4* ra, const float4* rb, const bool* ma) {
    float* Ab = As + buf * BM * LD;
    float* Bb = Bs + buf * BN * LD;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      float4 v = ra[it];
      if (!ma[it]) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
      if (tid + it * NT < A_F4) *reinterpret_cast<float4*>(Ab + a_row[it] * LD + a_chk[it] * 4) = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      float4 v = rb[it];
      if (!b_ok[it]) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
      if (tid + it * NT < B_F4) *reinterpret_cast<float4*>(Bb + b_row[it] * LD + b_chk[it] * 4) = v;
    }
  };

  constexpr int WMN = WAVES_M * WAVES_N;
  const int wk = wid / WMN, wmn = wid % WMN;
  const int wm0 = (wmn / WAVES_N) * WM;
  const int wn0 = (wmn % WAVES_N) * WN;
  const int l31 = lane & 31;
  const int lhi = lane >> 5;

  // one K step: fragment reads + MFMAs on LDS buffer `buf`
  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * LD;
    const float* Bb = Bs + buf * BN * LD;
    // lane (i = lane&31, half = lane>>5) reads 4 consecutive k values starting at 4*(2*sg+half); the j-th of them
    // is the operand of the j-th MFMA of this group. A and B use the same mapping, so MFMA j contracts
    // k = {8sg+j, 8sg+4+j}: a permutation of the K order, which the sum does not care about.
    // Software pipeline: the fragment reads of group sg+1 are issued before the MFMAs of group sg.
    constexpr int NSG = BK / 8 / WK;        // groups of 8 k values of this wavefront's share of the step
    static_assert(NSG >= 1, "K step too short for the K-split wavefronts");
    float4 af[2][MI], bf[2][NI];
    if constexpr (ABL) {
      for (int s_ = 0; s_ < 2; ++s_) {
        for (int mi = 0; mi < MI; ++mi) af[s_][mi] = make_float4(lane * 1e-3f, 1.f, 2.f, 3.f);
        for (int ni = 0; ni < NI; ++ni) bf[s_][ni] = make_float4(1.f, lane * 1e-3f, 2.f, 3.f);
      }
    }
    auto frag = [&](int sg, int slot) {
      const int koff = 4 * (2 * (wk * NSG + sg) + lhi);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[slot][mi] = *reinterpret_cast<const float4*>(Ab + (wm0 + mi * 32 + l31) * LD + koff);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[slot][ni] = *reinterpret_cast<const float4*>(Bb + (wn0 + ni * 32 + l31) * LD + koff);
    };
    if (!ABL || !(a.dbg & 16)) frag(0, 0);
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) {
      if (sg + 1 < NSG && (!ABL || !(a.dbg & 16))) frag(sg + 1, (sg + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float4 a4 = af[sg & 1][mi];
          const float av = j == 0 ? a4.x : j == 1 ? a4.y : j == 2 ? a4.z : a4.w;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const float4 b4 = bf[sg & 1][ni];
            const float bv = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (PF == 1) {
    if (T > 0) {
      load_tile(ra[0], rb[0], ma[0]);
      store_tile(0, ra[0], rb[0], ma[0]);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const int buf = t & 1;
      if (t + 1 < T) load_tile(ra[0], rb[0], ma[0]);  // global loads in flight during the MFMAs below
      compute(buf);
      if (t + 1 < T) store_tile(buf ^ 1, ra[0], rb[0], ma[0]);
      __syncthreads();
    }
  } else if constexpr (PF == 3) {
    // "store first": tile t+1 goes to LDS at the START of step t (its loads were issued at the start of step t-1), then the loads of tile
    // t+2 are issued, then the MFMAs of tile t run. The LDS-store latency sits under the MFMAs instead of between them and the barrier,
    // and right after the barrier the next step's fragments are already in LDS (one register set, two LDS buffers as before).
    if (T > 0) {
      load_tile(ra[0], rb[0], ma[0]);
      store_tile(0, ra[0], rb[0], ma[0]);
    }
    if (T > 1) load_tile(ra[0], rb[0], ma[0]);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T && (!ABL || !(a.dbg & 2))) store_tile((t + 1) & 1, ra[0], rb[0], ma[0]);
      if (t + 2 < T && (!ABL || !(a.dbg & 1))) load_tile(ra[0], rb[0], ma[0]);
      if (!ABL || !(a.dbg & 8)) compute(t & 1);
      if (!ABL || !(a.dbg & 4)) __syncthreads();
    }
  } else {
    // register set (t & 1) carries tile t between its loads (issued during step t-2) and its LDS store (after the MFMAs of step t-1);
    // the loop is unrolled by two so that the sets are addressed with constants (they must stay in registers)
    if (T > 0) load_tile(ra[0], rb[0], ma[0]);
    if (T > 1) load_tile(ra[1], rb[1], ma[1]);
    if (T > 0) store_tile(0, ra[0], rb[0], ma[0]);
    __syncthreads();
    for (int t = 0; t < T; t += 2) {
      if (t + 2 < T) load_tile(ra[0], rb[0], ma[0]);
      compute(0);
      if (t + 1 < T) store_tile(1, ra[1], rb[1], ma[1]);
      __syncthreads();
      if (t + 1 >= T) break;
      if (t + 3 < T) load_tile(ra[1], rb[1], ma[1]);
      compute(1);
      if (t + 2 < T) store_tile(0, ra[0], rb[0], ma[0]);
      __syncthreads();
    }
  }

  if (a.mode != 0) {
    for (int r = tid; r < BM; r += NT) {
      const int m = m0 + r;
      int n = -1, pi = 0, qi = 0;
      if (m < Mc) {
        qi = m % Qc;
        const int t2 = m / Qc;
        pi = t2 % Pc;
        n = t2 / Pc;
      }
      row_n[r] = n; row_p[r] = pi; row_q[r] = qi;
    }
    if constexpr (WK == 1) __syncthreads();      // (WK > 1: the barrier of the partial-sum exchange below orders it)
  }
  if constexpr (WK > 1) {
    // the K-split wavefronts' partial sums meet in LDS (the tile buffers are free after the loop's last barrier); wavefront group 0 adds
    // them in group order and writes the output
    float* red = smem;
    if (wk > 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((((wk - 1) * WMN + wmn) * MI * NI + mi * NI + ni) * 16 + e) * 64 + lane] = acc[mi][ni][e];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int w = 1; w < WK; ++w)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] += red[((((w - 1) * WMN + wmn) * MI * NI + mi * NI + ni) * 16 + e) * 64 + lane];
  }
  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  float bv[NI];
  int ck[NI], c_p[NI], c_q[NI], c_bh[NI], c_bw[NI];     // mode 2: channel, parity class and block shift of this lane's columns
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l31;
    ck[ni] = col; c_p[ni] = c_q[ni] = c_bh[ni] = c_bw[ni] = 0;
    if (a.mode == 2) {
      const int cls = col / a.K;
      ck[ni] = col - cls * a.K;
      c_p[ni] = cls / a.sw; c_q[ni] = cls % a.sw;
      c_bh[ni] = (c_p[ni] + a.ph) / a.sh; c_bw[ni] = (c_q[ni] + a.pw) / a.sw;
    }
    bv[ni] = (a.bias && a.nsplit == 1 && col < a.KN) ? a.bias[ck[ni]] : 0.f;
  }
  // every (accumulator element, output offset) pair this lane owns - the same set for every split of a tile
  auto for_each_out = [&](auto&& f) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const int m = m0 + row;
        if (m >= Mc) continue;
        if (a.mode == 2) {
          const int qi = row_q[row], pi = row_p[row], n = row_n[row];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn0 + ni * 32 + l31;
            const int p = a.sh * (pi - c_bh[ni]) + c_p[ni], q = a.sw * (qi - c_bw[ni]) + c_q[ni];
            if (col < a.KN && pi >= c_bh[ni] && p < a.P && qi >= c_bw[ni] && q < a.Q) f(mi, ni, e, (((long long)n * a.P + p) * a.Q + q) * a.K + ck[ni]);
          }
          continue;
        }
        long long obase;
        if (a.mode == 0) {
          obase = (long long)m * a.K;
        } else {
          const int qi = row_q[row], pi = row_p[row], n = row_n[row];
          obase = (((long long)n * a.P + (cp + a.sh * pi)) * a.Q + (cq + a.sw * qi)) * a.K;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int col = n0 + wn0 + ni * 32 + l31;
          if (col < a.K) f(mi, ni, e, obase + col);
        }
      }
    }
  };
  if (a.nsplit == 1) {
    float* __restrict__ yg = a.y;
    const bool accum = a.accumulate;
    for_each_out([&](int mi, int ni, int e, long long o) {
      float v = acc[mi][ni][e] + bv[ni];
      if (accum) v += yg[o];
      yg[o] = v;
    });
    return;
  }
  const long long total = (long long)a.N * a.P * a.Q * a.K;
  {
    float* __restrict__ pg = a.part + (long long)split * total;
    if (!a.cnt) {                                      // partial images summed by conv_split_reduce_kernel
      for_each_out([&](int mi, int ni, int e, long long o) { pg[o] = acc[mi][ni][e]; });
      return;
    }
    for_each_out([&](int mi, int ni, int e, long long o) { hwg_store_agent(pg + o, acc[mi][ni][e]); });
  }
  // The wavefront that delivers the LAST of a sub-tile's nsplit partial images sums them itself - in split order, then bias, then the old
  // output: the arithmetic of conv_split_reduce_kernel, bit for bit - and writes the output: no second launch, the partials are read where
  // they were written. Every split's wavefront of a sub-tile owns the same output offsets, so nothing but the arrival counter crosses
  // wavefronts (hwg_split_arrive_wave, hwg_common.h: the partials travel as agent-scope stores / loads).
  if (!hwg_split_arrive_wave(a.cnt + (((size_t)blockIdx.z * (gridDim.y / a.nsplit) + blockIdx.y / a.nsplit) * gridDim.x + blockIdx.x) * WMN + wmn, a.nsplit)) return;
  float bl[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bl[ni] = (a.bias && n0 + wn0 + ni * 32 + l31 < a.KN) ? a.bias[ck[ni]] : 0.f;
  const float* pr = a.part;
  float* yg = a.y;
  const int nsplit = a.nsplit;
  const bool accum = a.accumulate, has_bias = a.bias != nullptr;
  for_each_out([&](int, int ni, int, long long o) {
    float v = hwg_load_agent(pr + o);
    for (int sp = 1; sp < nsplit; ++sp) v += hwg_load_agent(pr + sp * total + o);
    if (has_bias) v += bl[ni];
    if (accum) v += yg[o];
    yg[o] = v;
  });
}

// y[i] = (accumulate ? y[i] : 0) + bias[i % K] + sum_s part[s][i]   (split-K epilogue; fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float* part, const float* bias, float* y, long long total, int K, int nsplit,
                                                                int accumulate) {
  if ((K & 3) == 0) {
    const long long n4 = total >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
      int sp = 1;
      for (; sp + 2 < nsplit; sp += 3) {               // splits are 2, 4 or 8: the first image plus groups of three, all loads of a group in flight
        float4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) v[u] = *reinterpret_cast<const float4*>(part + (sp + u) * total + 4 * i);
#pragma unroll
        for (int u = 0; u < 3; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; sp < nsplit; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(part + sp * total + 4 * i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (bias) {
        const int kb = total < (1ll << 32) ? (int)((unsigned)(4 * i) % (unsigned)K) : (int)((4 * i) % K);   // 64-bit division: ~40 VALU instructions
        const float4 b = *reinterpret_cast<const float4*>(bias + kb);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      if (accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(y + 4 * i);
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      *reinterpret_cast<float4*>(y + 4 * i) = s;
    }
  } else {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      float s = part[i];
      for (int sp = 1; sp < nsplit; ++sp) s += part[sp * total + i];
      if (bias) s += bias[(int)(i % K)];
      y[i] = accumulate ? y[i] + s : s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
struct WgK {
  const float* u;  // anchor  [Mtot][K]
  const float* v;  // gathered [N,H,W,C]
  float* part;     // [nsplit][R*S][K][C]
  int N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q;
  int Mtot, chunk, tiles_v;
  int bias_on;          // also produce the column sums of u (the layer's bias gradient) from the tiles that pass through LDS anyway
  long long pstride;    // floats per partial image: R*S*K*C (+K bias partials)
  // gradient sets (hwg_conv_wgrad_sets): u holds `sets` anchors of Mtot rows each, back to back, v is shared. Workgroup blockIdx.x works on set
  // blockIdx.x / gx as workgroup blockIdx.x % gx of a gx-wide launch and writes partial image blockIdx.x. sets == 1: gx == gridDim.x.
  int sets, gx;
  int set_on_v;         // 0: the sets differ in u (v shared); 1: the sets differ in v ([N,H,W,C] each, back to back) and share u (transposed layers)
};

// TAPN: single-channel gathered tensor (C == 1, first layers): the GEMM's N dimension is the taps (<= BNV) instead of C, so one
// workgroup produces dw[k][all taps] for its pixel chunk from an im2col tile built on the fly (v loads are scalar, L1/L2 resident).
template <int BMU, int BNV, int BKP, int WAVES_M, int WAVES_N, int WAVES_K, bool TAPN = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K) void wgrad_mfma_kernel(WgK a) {
  constexpr int NT = 64 * WAVES_M * WAVES_N * WAVES_K;
  constexpr int WM = BMU / WAVES_M, WN = BNV / WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int UC = BMU / 4, VC = BNV / 4;
  constexpr int U_F4 = BKP * UC, V_F4 = BKP * VC;
  constexpr int U_IT = (U_F4 + NT - 1) / NT, V_IT = (V_F4 + NT - 1) / NT;
  constexpr int TILE = BKP * (BMU + BNV);
  constexpr int RED = (WAVES_K > 1) ? WAVES_K * WAVES_M * WAVES_N * MI * NI * 16 * 64 : 0;
  constexpr int SM = (2 * TILE > RED) ? 2 * TILE : RED;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  float* Us = smem;              // [2][BKP][BMU]
  float* Vs = smem + 2 * BKP * BMU;  // [2][BKP][BNV]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tap = TAPN ? 0 : blockIdx.z;
  const int r = tap / a.S, s = tap % a.S;
  const int RS = a.R * a.S;
  const int tu = blockIdx.y / a.tiles_v, tv = blockIdx.y % a.tiles_v;
  const int k0 = tu * BMU, c0 = tv * BNV;
  const int set = a.sets > 1 ? (int)blockIdx.x / a.gx : 0;
  const int bx = (int)blockIdx.x - set * a.gx;
  const float* __restrict__ au = a.u + (a.set_on_v ? 0ll : (long long)set * a.Mtot * a.K);
  const float* __restrict__ av = a.v + (a.set_on_v ? (long long)set * a.N * a.H * a.W * a.C : 0ll);
  const int pb = bx * a.chunk;
  const int pe = min(pb + a.chunk, a.Mtot);

  int u_kp[U_IT], u_c4[U_IT];
  bool u_ok[U_IT];
#pragma unroll
  for (int it = 0; it < U_IT; ++it) {
    const int f = tid + it * NT;
    u_kp[it] = f / UC;
    u_c4[it] = f % UC;
    u_ok[it] = (f < U_F4) && (k0 + u_c4[it] * 4 < a.K);
  }
  int v_kp[V_IT], v_c4[V_IT];
  bool v_ok[V_IT];
#pragma unroll
  for (int it = 0; it < V_IT; ++it) {
    const int f = tid + it * NT;
    v_kp[it] = f / VC;
    v_c4[it] = f % VC;
    v_ok[it] = (f < V_F4) && (c0 + v_c4[it] * 4 < (TAPN ? RS : a.C));
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ru[U_IT], rv[V_IT];
  int pbase = pb;  // first pixel of the tile being loaded
  // (n, p, q) of the pixel each V load slot reads, advanced incrementally (BKP pixels per step) instead of two integer divisions per load
  int v_q[V_IT], v_p[V_IT], v_n[V_IT];
#pragma unroll
  for (int it = 0; it < V_IT; ++it) {
    const int m = pb + v_kp[it];
    v_q[it] = m % a.Q;
    const int t = m / a.Q;
    v_p[it] = t % a.P;
    v_n[it] = t / a.P;
  }
  const bool one_wrap = a.Q >= BKP;
  auto load_tile = [&]() {
#pragma unroll
    for (int it = 0; it < U_IT; ++it) {
      const int m = pbase + u_kp[it];
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u_ok[it] && m < pe) val = *reinterpret_cast<const float4*>(au + (long long)m * a.K + k0 + u_c4[it] * 4);
      ru[it] = val;
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
      const int m = pbase + v_kp[it];
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v_ok[it] && m < pe) {
        if (TAPN) {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int tp = c0 + v_c4[it] * 4 + j;
            const int rr = tp / a.S, ss = tp - rr * a.S;
            const int ih = v_p[it] * a.sh - a.ph + rr * a.dh;
            const int iw = v_q[it] * a.sw - a.pw + ss * a.dw;
            const bool ok = tp < RS && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            e[j] = ok ? av[((long long)v_n[it] * a.H + ih) * a.W + iw] : 0.f;
          }
          val = make_float4(e[0], e[1], e[2], e[3]);
        } else {
          const int ih = v_p[it] * a.sh - a.ph + r * a.dh;
          const int iw = v_q[it] * a.sw - a.pw + s * a.dw;
          if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
            val = *reinterpret_cast<const float4*>(av + (((long long)v_n[it] * a.H + ih) * a.W + iw) * a.C + c0 + v_c4[it] * 4);
        }
      }
      rv[it] = val;
      if (one_wrap) {
        v_q[it] += BKP;
        if (v_q[it] >= a.Q) {
          v_q[it] -= a.Q;
          if (++v_p[it] >= a.P) { v_p[it] = 0; ++v_n[it]; }
        }
      } else {
        const int m2 = m + BKP;
        v_q[it] = m2 % a.Q;
        const int t2 = m2 / a.Q;
        v_p[it] = t2 % a.P;
        v_n[it] = t2 / a.P;
      }
    }
    pbase += BKP;
  };
  auto store_tile = [&](int buf) {
    float* Ub = Us + buf * BKP * BMU;
    float* Vb = Vs + buf * BKP * BNV;
#pragma unroll
    for (int it = 0; it < U_IT; ++it)
      if (tid + it * NT < U_F4) *reinterpret_cast<float4*>(Ub + u_kp[it] * BMU + u_c4[it] * 4) = ru[it];
#pragma unroll
    for (int it = 0; it < V_IT; ++it)
      if (tid + it * NT < V_F4) *reinterpret_cast<float4*>(Vb + v_kp[it] * BNV + v_c4[it] * 4) = rv[it];
  };

  const int T = (pe > pb) ? (pe - pb + BKP - 1) / BKP : 0;
  // wave -> (K slice, M sub-tile, N sub-tile)
  constexpr int WMN = WAVES_M * WAVES_N;
  const int wk = wid / WMN, wmn = wid % WMN;
  const int wm0 = (wmn / WAVES_N) * WM;
  const int wn0 = (wmn % WAVES_N) * WN;
  const int kbeg = wk * (BKP / WAVES_K);
  const int l31 = lane & 31, lhi = lane >> 5;

  // bias gradient = column sums of u: taken by the workgroups of the first tap / first v tile from their LDS copy of the u tile
  const bool do_bias = a.bias_on && tap == 0 && tv == 0 && tid < BMU;
  double bsum = 0.0;   // a bias gradient is the difference of large, nearly cancelling partial sums (real vs fake lines in a `disc` lesson): fp64

  if (T > 0) { load_tile(); store_tile(0); }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    if (t + 1 < T) load_tile();
    const float* Ub = Us + buf * BKP * BMU;
    const float* Vb = Vs + buf * BKP * BNV;
    if (do_bias) {
#pragma unroll
      for (int kp = 0; kp < BKP; ++kp) bsum += (double)Ub[kp * BMU + tid];
    }
    // software pipeline over groups of PF k pairs: the LDS reads of group g+1 are issued before the MFMAs of group g
    // (sched_barrier keeps the compiler from re-serialising them into read-wait-MFMA triples)
    constexpr int KPAIRS = BKP / WAVES_K / 2;
    constexpr int PF = KPAIRS >= 4 ? 4 : KPAIRS;
    constexpr int NG = KPAIRS / PF;
    float af[2][PF][MI], bf[2][PF][NI];
    auto lds_group = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kk = kbeg + 2 * (g * PF + u) + lhi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[slot][u][mi] = Ub[kk * BMU + wm0 + mi * 32 + l31];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[slot][u][ni] = Vb[kk * BNV + wn0 + ni * 32 + l31];
      }
    };
    lds_group(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) lds_group(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][u][mi], bf[g & 1][u][ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + 1 < T) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (do_bias && k0 + tid < a.K) a.part[(long long)blockIdx.x * a.pstride + (long long)RS * a.K * a.C + k0 + tid] = (float)bsum;

  if (WAVES_K > 1) {
    // cross-wave reduction of the per-wave K slices (tile buffers are free now)
    float* red = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((wid * MI * NI + mi * NI + ni) * 16 + e) * 64 + lane] = acc[mi][ni][e];
    __syncthreads();
    if (wk != 0) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float sacc = 0.f;
          for (int w = 0; w < WAVES_K; ++w) sacc += red[(((w * WMN + wmn) * MI * NI + mi * NI + ni) * 16 + e) * 64 + lane];
          acc[mi][ni][e] = sacc;
        }
  }

  float* pout = a.part + (long long)blockIdx.x * a.pstride + (long long)tap * a.K * a.C;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + wm0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      if (k >= a.K) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int c = c0 + wn0 + ni * 32 + l31;
        if (TAPN) {
          if (c < RS) pout[(long long)c * a.K + k] = acc[mi][ni][e];   // part layout [tap][K][C=1]
        } else {
          if (c < a.C) pout[(long long)k * a.C + c] = acc[mi][ni][e];
        }
      }
    }
}

// dw = (accumulate ? dw : 0) + sum over the partial images; entries past R*S*K*C of a partial image are the K bias partials.
// The four reduce schedules below are __device__ bodies parameterised by (block id, block count) so that the same code runs as a launch of
// its own (one weight gradient) and as one entry of the table-driven launch that sums the partial images of a whole backward pass
// (wgrad_reduce_multi_kernel). Every output element is summed over the partial images in a fixed order that does not depend on the grid.
// Four consecutive elements per thread (16-byte loads from every partial image); requires C % 4 == 0 (the kernel's own requirement).
__device__ __forceinline__ void wgrad_reduce_vec4_body(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS, int S, int K, int C,
                                                       long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                       long long pstride, float* dbias, int bias_accumulate, int bid, int nb) {
  const long long total = (long long)RS * K * C;
  const long long all = total + (dbias ? K : 0);
  const long long n4 = all >> 2;     // total, K and pstride are multiples of 4
  for (long long j = bid * (long long)blockDim.x + threadIdx.x; j < n4; j += (long long)nb * blockDim.x) {
    const long long i = j << 2;
    float4 sum = *reinterpret_cast<const float4*>(part + i);
    int sp = 1;
    for (; sp + 3 < nsplit; sp += 4) {                 // four partial images in flight, added in image order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(part + (long long)(sp + u) * pstride + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
    for (; sp < nsplit; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long long)sp * pstride + i);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float e[4] = {sum.x, sum.y, sum.z, sum.w};
    if (i >= total) {
      const int k = (int)(i - total);
#pragma unroll
      for (int u = 0; u < 4; ++u) dbias[k + u] = bias_accumulate ? dbias[k + u] + e[u] : e[u];
      continue;
    }
    const unsigned iu = (unsigned)i, t = iu / (unsigned)C;       // 32-bit: a weight image has far fewer than 2^31 elements
    const int c = (int)(iu - t * C);
    const int tap = (int)(t / (unsigned)K);
    const int k = (int)(t - (unsigned)tap * K);
    const long long o = k * sa + c * sb + (tap / S) * sr + (tap % S) * ss;
#pragma unroll
    for (int u = 0; u < 4; ++u) dw[o + u * sb] = accumulate ? dw[o + u * sb] + e[u] : e[u];
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS, int S, int K, int C,
                                                           long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                           long long pstride, float* dbias, int bias_accumulate) {
  wgrad_reduce_vec4_body(part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss, accumulate, pstride, dbias, bias_accumulate, blockIdx.x, gridDim.x);
}
// scalar variant for the taps-as-N kernel (C == 1)
__device__ __forceinline__ void wgrad_reduce_scalar_body(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS, int S, int K,
                                                         int C, long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                         long long pstride, float* dbias, int bias_accumulate, int bid, int nb) {
  const long long total = (long long)RS * K * C;
  const long long all = total + (dbias ? K : 0);
  for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < all; i += (long long)nb * blockDim.x) {
    float sum = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) sum += part[(long long)sp * pstride + i];
    if (i >= total) {
      const int k = (int)(i - total);
      dbias[k] = bias_accumulate ? dbias[k] + sum : sum;
      continue;
    }
    const unsigned iu = (unsigned)i, t = iu / (unsigned)C;
    const int c = (int)(iu - t * C);
    const int tap = (int)(t / (unsigned)K);
    const int k = (int)(t - (unsigned)tap * K);
    const long long o = k * sa + c * sb + (tap / S) * sr + (tap % S) * ss;
    dw[o] = accumulate ? dw[o] + sum : sum;
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS, int S, int K,
                                                                  int C, long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                                  long long pstride, float* dbias, int bias_accumulate) {
  wgrad_reduce_scalar_body(part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss, accumulate, pstride, dbias, bias_accumulate, blockIdx.x, gridDim.x);
}

// same sum with SL "split lanes" per output: used when there are few outputs and many partial images (thread t of a workgroup owns
// output t % OUT and partials t / OUT, t / OUT + SL, ...; the lanes are combined through LDS in lane order, so still deterministic)
template <int SL>
__device__ __forceinline__ void wgrad_reduce_lanes_body(const float* part, float* dw, int nsplit, int RS, int S, int K, int C,
                                                        long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                        long long pstride, float* dbias, int bias_accumulate, int bid, float* red) {
  constexpr int OUT = 256 / SL;           // red: [SL][OUT] floats of LDS
  const long long total = (long long)RS * K * C;
  const long long all = total + (dbias ? K : 0);
  const int ol = threadIdx.x % OUT, lane = threadIdx.x / OUT;
  const long long i = (long long)bid * OUT + ol;
  float sum = 0.f;
  if (i < all) {
    int sp = lane;
    for (; sp + 3 * SL < nsplit; sp += 4 * SL) {      // four partial images in flight, added in image order
      const float v0 = part[(long long)sp * pstride + i], v1 = part[(long long)(sp + SL) * pstride + i];
      const float v2 = part[(long long)(sp + 2 * SL) * pstride + i], v3 = part[(long long)(sp + 3 * SL) * pstride + i];
      sum += v0; sum += v1; sum += v2; sum += v3;
    }
    for (; sp < nsplit; sp += SL) sum += part[(long long)sp * pstride + i];
  }
  red[lane * OUT + ol] = sum;
  __syncthreads();
  if (lane == 0 && i < all) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < SL; ++l) t += red[l * OUT + ol];
    if (i >= total) {
      const int kb = (int)(i - total);
      dbias[kb] = bias_accumulate ? dbias[kb] + t : t;
      return;
    }
    const unsigned iu = (unsigned)i, q = iu / (unsigned)C;
    const int c = (int)(iu - q * C);
    const int tap = (int)(q / (unsigned)K);
    const int k = (int)(q - (unsigned)tap * K);
    const long long o = k * sa + c * sb + (tap / S) * sr + (tap % S) * ss;
    dw[o] = accumulate ? dw[o] + t : t;
  }
}
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_lanes_kernel(const float* part, float* dw, int nsplit, int RS, int S, int K, int C,
                                                                 long long sa, long long sb, long long sr, long long ss, int accumulate,
                                                                 long long pstride, float* dbias, int bias_accumulate) {
  __shared__ float red[256];
  wgrad_reduce_lanes_body<SL>(part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss, accumulate, pstride, dbias, bias_accumulate, blockIdx.x, red);
}

// Tap-contiguous weights (ss == 1, sr == S, sb == R*S - every weight this package owns): the sums of 256 / SL neighbouring (k, c) pairs go
// through LDS and leave as whole runs of R*S*(256/SL) consecutive floats. The plain kernel above writes a tap at a time, 4 bytes every
// R*S*4, and was bound by those scattered stores (27 us for 512x512x3x3 with 4 partial images, 38 MB read + 9 MB written).
// Thread t: pair t % KC, partial images t / KC, t / KC + SL, ... (combined through LDS in lane order: a fixed order, deterministic).
template <int SL, int RSC>
__device__ __forceinline__ void wgrad_reduce_rows_body(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS_rt, int K, int C,
                                                       long long sa, int accumulate, long long pstride, float* dbias, int bias_accumulate,
                                                       int main_blocks, int bid, float* red) {
  constexpr int KC = 256 / SL;
  const int RS = RSC ? RSC : RS_rt;        // red: [SL][KC][RS] (+1 float of padding per pair) of LDS
  const int ldp = RS | 1;
  const long long KCtot = (long long)K * C;
  if (bid >= main_blocks) {    // the K bias sums behind every partial image
    const int k = (bid - main_blocks) * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(long long)sp * pstride + RS * KCtot + k];
    dbias[k] = bias_accumulate ? dbias[k] + s : s;
    return;
  }
  const int pl = threadIdx.x % KC, sl = threadIdx.x / KC;
  const long long base = (long long)bid * KC;
  const long long kc = base + pl;
  if (kc < KCtot) {
    if (RSC) {
      float acc[RSC ? RSC : 1];
#pragma unroll
      for (int t = 0; t < RSC; ++t) acc[t] = 0.f;
      for (int sp = sl; sp < nsplit; sp += SL) {
        const float* src = part + (long long)sp * pstride + kc;
#pragma unroll
        for (int t = 0; t < RSC; ++t) acc[t] += src[t * KCtot];
      }
#pragma unroll
      for (int t = 0; t < RSC; ++t) red[(sl * KC + pl) * ldp + t] = acc[t];
    } else {
      for (int t = 0; t < RS; ++t) {
        float a = 0.f;
        for (int sp = sl; sp < nsplit; sp += SL) a += part[(long long)sp * pstride + t * KCtot + kc];
        red[(sl * KC + pl) * ldp + t] = a;
      }
    }
  }
  __syncthreads();
  const int n_out = KC * RS;
  for (int j = threadIdx.x; j < n_out; j += 256) {
    const int p = j / RS, t = j - p * RS;
    const long long q = base + p;
    if (q >= KCtot) break;
    float s = red[p * ldp + t];
#pragma unroll
    for (int l = 1; l < SL; ++l) s += red[(l * KC + p) * ldp + t];
    const long long k = q / C;
    const long long o = k * sa + (q - k * C) * RS + t;
    dw[o] = accumulate ? dw[o] + s : s;
  }
}
template <int SL, int RSC>
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit, int RS_rt, int K, int C,
                                                                long long sa, int accumulate, long long pstride, float* dbias, int bias_accumulate,
                                                                int main_blocks) {
  extern __shared__ float red_rows[];
  wgrad_reduce_rows_body<SL, RSC>(part, dw, nsplit, RS_rt, K, C, sa, accumulate, pstride, dbias, bias_accumulate, main_blocks, blockIdx.x, red_rows);
}

// ---- one launch for the partial images of a whole backward pass ----------------------------------------------------------------
// A backward pass leaves one set of partial images per weight gradient (65 per training step); summed layer by layer that was 65 launches
// of 5..25 us, most of them too small to stream at memory rate. With deferral on (hwg_wgrad_defer_next before the weight-gradient call, the
// caller keeps the partial images alive in its own arena) the reduce of a weight gradient is only queued; hwg_wgrad_defer_flush sums
// everything queued so far with one launch per 32 entries: block -> (entry, block of that entry's own schedule), entries by value in the
// kernel arguments (no table upload). Each entry runs exactly the schedule it would have run alone, so results are bit-identical.
// Two queued gradients of the SAME tensor (a network applied twice in one pass) go to consecutive launches, in queue order.
enum { RED_VEC4 = 0, RED_SCALAR = 1, RED_LANES32 = 2, RED_LANES4 = 3, RED_ROWS = 4 /* + 2 * rsc_index + (SL == 4) */ };
struct RedEntry {
  const float* part; float* dw; float* dbias;
  long long pstride, sa, sb, sr, ss;
  int nsplit, RS, S, K, C, accumulate, bias_accumulate, variant, first_block, nblocks, main_blocks, tag;
};
constexpr int RED_MAX = 32;
struct RedTable { int n, pad; RedEntry e[RED_MAX]; };

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RedTable t) {
  extern __shared__ float red_multi[];
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {   // last entry whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (t.e[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RedEntry& e = t.e[lo];
  const int bid = (int)blockIdx.x - e.first_block;
  switch (e.variant) {
    case RED_VEC4: wgrad_reduce_vec4_body(e.part, e.dw, e.nsplit, e.RS, e.S, e.K, e.C, e.sa, e.sb, e.sr, e.ss, e.accumulate, e.pstride, e.dbias, e.bias_accumulate, bid, e.nblocks); break;
    case RED_SCALAR: wgrad_reduce_scalar_body(e.part, e.dw, e.nsplit, e.RS, e.S, e.K, e.C, e.sa, e.sb, e.sr, e.ss, e.accumulate, e.pstride, e.dbias, e.bias_accumulate, bid, e.nblocks); break;
    case RED_LANES32: wgrad_reduce_lanes_body<32>(e.part, e.dw, e.nsplit, e.RS, e.S, e.K, e.C, e.sa, e.sb, e.sr, e.ss, e.accumulate, e.pstride, e.dbias, e.bias_accumulate, bid, red_multi); break;
    case RED_LANES4: wgrad_reduce_lanes_body<4>(e.part, e.dw, e.nsplit, e.RS, e.S, e.K, e.C, e.sa, e.sb, e.sr, e.ss, e.accumulate, e.pstride, e.dbias, e.bias_accumulate, bid, red_multi); break;
#define HWG_ROWS_CASE(ID, SL, RSC) \
    case RED_ROWS + ID: wgrad_reduce_rows_body<SL, RSC>(e.part, e.dw, e.nsplit, e.RS, e.K, e.C, e.sa, e.accumulate, e.pstride, e.dbias, e.bias_accumulate, e.main_blocks, bid, red_multi); break;
    HWG_ROWS_CASE(0, 1, 9) HWG_ROWS_CASE(1, 4, 9) HWG_ROWS_CASE(2, 1, 16) HWG_ROWS_CASE(3, 4, 16) HWG_ROWS_CASE(4, 1, 3) HWG_ROWS_CASE(5, 4, 3)
    HWG_ROWS_CASE(6, 1, 0) HWG_ROWS_CASE(7, 4, 0)
#undef HWG_ROWS_CASE
    default: break;
  }
}

__global__ void pack_weight_kernel(const float* src, float* dst, int A, int B, int Bpad, int R, int S,
                                   long long sa, long long sb, long long sr, long long ss, int flip) {
  const long long total = (long long)R * S * A * Bpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i % Bpad);
    const long long t = i / Bpad;
    const int aa = (int)(t % A);
    const int tap = (int)(t / A);
    int r = tap / S, s = tap % S;
    if (flip) { r = R - 1 - r; s = S - 1 - s; }
    dst[i] = (b < B) ? src[aa * sa + b * sb + r * sr + s * ss] : 0.f;
  }
}

// all cached weight images of one optimizer group re-packed in ONE launch after the Adam step (table built by the host once)
struct PackEntry {
  const float* src;
  float* dst;
  int A, B, Bpad, R, S, flip;
  long long sa, sb, sr, ss, total, first_block;
  int mode, Apad;   // mode 1: Winograd filter transform (conv_wino.hip), total = Apad*Bpad threads; mode 2: the F(3x3,2x2) image of a 4x4 stride-2 layer
};

constexpr int PACK_PER_BLOCK = 1024;
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackEntry* table, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last entry whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = table[lo];
  const long long base = ((long long)blockIdx.x - e.first_block) * PACK_PER_BLOCK;
#pragma unroll
  for (int j = 0; j < PACK_PER_BLOCK / 256; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i >= e.total) break;
    if (e.mode == 1) {
      wino_pack_one(e.src, e.dst, i, e.A, e.Apad, e.B, e.Bpad, e.sa, e.sb, e.sr, e.ss, e.flip);
      continue;
    }
    if (e.mode == 2) {      // F(3x3,2x2) image of a 4x4 stride-2 layer: A / B = the convolution's output / input channels, flip = data-gradient image
      wino_s2_pack_one(e.src, e.dst, i, e.A, e.B, e.sa, e.sb, e.flip);
      continue;
    }
    const int b = (int)(i % e.Bpad);
    const long long t = i / e.Bpad;
    const int aa = (int)(t % e.A);
    const int tap = (int)(t / e.A);
    int r = tap / e.S, s2 = tap % e.S;
    if (e.flip) { r = e.R - 1 - r; s2 = e.S - 1 - s2; }
    e.dst[i] = (b < e.B) ? e.src[aa * e.sa + b * e.sb + r * e.sr + s2 * e.ss] : 0.f;
  }
}

// The same for weights that are a device-side scalar multiple of a stored tensor - the spectral-norm layers' W_bar / sigma (reference:
// model/discriminator_ap.py:31-32, recomputed on every forward): every image a forward / backward pass of the network will ask for (direct and
// mirrored tap order, Winograd domain) is written straight from W_bar with the factor applied on the way, in ONE launch per forward pass,
// instead of one scale launch plus two or three pack launches per layer. dst = dst_base + dst_off floats, factor = scale_base[scale_idx]: the
// table is static, the image buffer and the sigma vector are fresh per forward pass.
struct PackEntryS {
  PackEntry e;
  long long dst_off;
  int scale_idx, pad;
};
__global__ __launch_bounds__(256) void pack_weight_multi_scaled_kernel(const PackEntryS* table, int n, float* dst_base, const float* scale_base) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].e.first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = table[lo].e;
  float* dst = dst_base + table[lo].dst_off;
  const float sc = scale_base[table[lo].scale_idx];
  const long long base = ((long long)blockIdx.x - e.first_block) * PACK_PER_BLOCK;
#pragma unroll
  for (int j = 0; j < PACK_PER_BLOCK / 256; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i >= e.total) break;
    if (e.mode == 1) {
      wino_pack_one(e.src, dst, i, e.A, e.Apad, e.B, e.Bpad, e.sa, e.sb, e.sr, e.ss, e.flip, sc);
      continue;
    }
    if (e.mode == 2) {
      wino_s2_pack_one(e.src, dst, i, e.A, e.B, e.sa, e.sb, e.flip, sc);
      continue;
    }
    const int b = (int)(i % e.Bpad);
    const long long t = i / e.Bpad;
    const int aa = (int)(t % e.A);
    const int tap = (int)(t / e.A);
    int r = tap / e.S, s2 = tap % e.S;
    if (e.flip) { r = e.R - 1 - r; s2 = e.S - 1 - s2; }
    dst[i] = (b < e.B) ? __fmul_rn(e.src[aa * e.sa + b * e.sb + r * e.sr + s2 * e.ss], sc) : 0.f;
  }
}

// column sums: x[rows][C] -> part[chunks][C]  (single chunk: straight into out)
// V = 4: block = cgn float4 column groups x (256 / cgn) row lanes, cgn = min(16, C / 4): with fewer than 64 columns (the generator's 16 / 32
// channel bias gradients on 250 k rows: 19 us at 0.8 TB/s when 3/4 of the block idled on 128 blocks) the spare threads become row lanes;
// four rows per lane in flight. V = 1: 64 columns x 4 row lanes
template <int V>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* x, long long rows, int C, float* part, long long rows_per_chunk,
                                                             float* direct_out, int accumulate) {
  __shared__ double red[256][V];
  const int cleft = C - (int)blockIdx.y * 64;                          // columns this block column covers (<= 64 of them)
  const int cgn = V == 4 ? min(16, (cleft + 3) / 4) : min(64, cleft);  // column groups per block
  const int RL = 256 / cgn;                                            // row lanes
  const int cg = threadIdx.x % cgn, rl = threadIdx.x / cgn;
  const int c = blockIdx.y * 64 + cg * V;
  const long long rb = blockIdx.x * rows_per_chunk;
  const long long re = min(rb + rows_per_chunk, rows);
  double s[V];        // fp64: column sums feed bias gradients, which are small differences of large sums
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.0;
  if (rl < RL && c < C) {
    if (V == 4) {
      for (long long rr = rb + rl; rr < re; rr += 4ll * RL) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long r2 = rr + (long long)u * RL;
          v[u] = *reinterpret_cast<const float4*>(x + (r2 < re ? r2 : rr) * C + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (rr + (long long)u * RL >= re) break;
          s[0] += (double)v[u].x; s[1 % V] += (double)v[u].y; s[2 % V] += (double)v[u].z; s[3 % V] += (double)v[u].w;
        }
      }
    } else {
      for (long long rr = rb + rl; rr < re; rr += RL) s[0] += (double)x[rr * C + c];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if ((int)threadIdx.x < min(64, cleft)) {
    const int cc = blockIdx.y * 64 + threadIdx.x;
    double v = 0.0;
    for (int r = 0; r < RL; ++r) v += red[r * cgn + (int)threadIdx.x / V][threadIdx.x % V];
    if (direct_out) direct_out[cc] = accumulate ? direct_out[cc] + (float)v : (float)v;
    else part[(long long)blockIdx.x * C + cc] = (float)v;
  }
}
// C <= 8 (bias gradients of the 1- and 2-channel heads): the kernel above would keep 4 of its 256 threads busy. Here every thread owns whole
// rows (C consecutive floats, neighbouring lanes neighbouring rows), the block combines its 256 row lanes in a fixed tree.
template <int CC>
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ x, long long rows, float* part, long long rows_per_chunk,
                                                            float* direct_out, int accumulate) {
  __shared__ double red[256][CC];
  const long long rb = blockIdx.x * rows_per_chunk;
  const long long re = min(rb + rows_per_chunk, rows);
  double s[CC];
#pragma unroll
  for (int c = 0; c < CC; ++c) s[c] = 0.0;
  for (long long r = rb + threadIdx.x; r < re; r += 256) {
#pragma unroll
    for (int c = 0; c < CC; ++c) s[c] += (double)x[r * CC + c];
  }
#pragma unroll
  for (int c = 0; c < CC; ++c) red[threadIdx.x][c] = s[c];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int c = 0; c < CC; ++c) red[threadIdx.x][c] += red[threadIdx.x + w][c];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < CC) {
    const float v = (float)red[0][threadIdx.x];
    if (direct_out) direct_out[threadIdx.x] = accumulate ? direct_out[threadIdx.x] + v : v;
    else part[(long long)blockIdx.x * CC + threadIdx.x] = v;
  }
}
// second stage: 16 columns x 16 chunk lanes per workgroup (fixed combination order -> deterministic)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, int chunks, int C, float* out, int accumulate) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C)
    for (int i = lane; i < chunks; i += 16) s += part[(long long)i * C + c];
  red[lane][cl] = s;
  __syncthreads();
  if (lane == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

template <int BM, int BN, int BK, int WM_, int WN_>
void launch_conv(const ConvK& k, dim3 grid, hipStream_t st, int pf) {
  if constexpr (BM == 64 && BN == 64) {
    // K-split wavefront pairs on the 4-wavefront 64 x 64 tile (8 wavefronts): 42.2 -> 37.6 us on 8x8x122x128->128, 32.8 -> 29.4 us on
    // the recogniser's 1x3 layers, 20.9 -> 19.2 us on the generator's 4x4 stride-2 layer (tools/probes/probe_r3_wk.txt); four-way splits and
    // the 128 x 32 tile measured no better (HWG_CONV_WK=1 restores the 4-wavefront kernel for A/B runs)
    if (hwg_tune().conv_wk >= 2) {
      hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM_, WN_, 3, 2>), grid, dim3(128 * WM_ * WN_), 0, st, k);
      return;
    }
  }
  if constexpr (BM == 128 && BN == 128 && BK == 32) {
    if (k.dbg) { hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM_, WN_, 3, 1, true>), grid, dim3(64 * WM_ * WN_), 0, st, k); return; }
  }
  if (pf >= 3) hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM_, WN_, 3>), grid, dim3(64 * WM_ * WN_), 0, st, k);
  else if (pf == 2) hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM_, WN_, 2>), grid, dim3(64 * WM_ * WN_), 0, st, k);
  else hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM_, WN_, 1>), grid, dim3(64 * WM_ * WN_), 0, st, k);
}

struct WgPlan {
  int cfg;      // 0: 128x128, 1: 64x64, 2: 32x32 (k-split waves)
  int bmu, bnv, bkp;
  int tiles_u, tiles_v, nsplit, chunk;
};
static void wg_split(WgPlan& p, long long Mtot, long long ns) {
  const long long max_ns = (Mtot + 4LL * p.bkp - 1) / (4LL * p.bkp);  // at least 4 k-steps per split
  if (ns > max_ns) ns = max_ns;
  if (ns < 1) ns = 1;
  long long chunk = (Mtot + ns - 1) / ns;
  chunk = (chunk + p.bkp - 1) / p.bkp * p.bkp;
  ns = (Mtot + chunk - 1) / chunk;
  if (ns < 1) ns = 1;
  p.nsplit = (int)ns;
  p.chunk = (int)chunk;
}
static void wg_tile(WgPlan& p, int cfg, int K, int C) {
  p.cfg = cfg;
  if (cfg == 0) { p.bmu = 128; p.bnv = 128; p.bkp = 32; }
  else if (cfg == 1) { p.bmu = 64; p.bnv = 64; p.bkp = 32; }
  else { p.bmu = 32; p.bnv = 32; p.bkp = 32; }
  p.tiles_u = hwg_cdiv(K, p.bmu);
  p.tiles_v = hwg_cdiv(C, p.bnv);
}
// Schedule = (tile, number of pixel chunks) with the smallest modelled time; same quantum model as plan_conv (parameters fitted with
// tools/wgrad_model_fit.py): ceil(blocks/256) workgroup quanta of (K steps + overhead) steps each, plus the pass that sums the per-chunk
// partial weight images.
static WgPlan plan_wgrad_model(const hwg_conv_desc* d);
WgPlan plan_wgrad(const hwg_conv_desc* d) {
  static thread_local HwgPlanCache<WgPlan> cache;
  return cache.get(d, plan_wgrad_model);
}
static WgPlan plan_wgrad_model(const hwg_conv_desc* d) {
  WgPlan p;
  const int K = d->K, C = d->C;
  const long long Mtot = (long long)d->N * d->P * d->Q;
  if (K <= 32 && C <= 32) {
    wg_tile(p, 2, K, C);
    const long long base = (long long)d->R * d->S * p.tiles_u * p.tiles_v;
    wg_split(p, Mtot, (1024 + base - 1) / base);
    return p;
  }
  if (const char* f = hwg_tune().wgrad_force; *f) {  // tuning aid: "cfg,target_blocks"
    int fc = -1, ft = 0;
    if (sscanf(f, "%d,%d", &fc, &ft) == 2 && ft > 0 && (fc == 1 || (fc == 0 && K >= 128 && C >= 128))) {
      wg_tile(p, fc, K, C);
      const long long base = (long long)d->R * d->S * p.tiles_u * p.tiles_v;
      wg_split(p, Mtot, (ft + base - 1) / base);
      return p;
    }
  }
  static const int splits[14] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128};
  static const double tflops[2] = {106.0, 90.0}, overhead[2] = {1.0, 1.0};
  static const int waves_per_block[2] = {16, 8};
  const double wbytes = 4.0 * d->R * d->S * K * C;
  double best = 1e30;
  WgPlan bp = p;
  for (int cfg = (K >= 128 && C >= 128) ? 0 : 1; cfg <= 1; ++cfg) {
    WgPlan c;
    wg_tile(c, cfg, K, C);
    const double base = (double)d->R * d->S * c.tiles_u * c.tiles_v;
    int last_ns = 0;
    for (int si = 0; si < 14; ++si) {
      wg_split(c, Mtot, splits[si]);
      if (c.nsplit == last_ns) continue;
      last_ns = c.nsplit;
      if (wbytes * c.nsplit > 2.0e9) break;
      const double q = base * c.nsplit / 256.0;
      const double quanta = q <= 2.0 ? ceil(q) : q + 0.5;
      // sustained rate grows with the wavefronts resident per CU (latency hiding): 65 % with one 8-wave workgroup, full at >= 24 waves
      double resident = (q > 1.0 ? q : 1.0);
      const double max_blocks = 32.0 / waves_per_block[cfg];
      if (resident > max_blocks) resident = max_blocks;
      resident *= waves_per_block[cfg];
      const double rate = tflops[cfg] * (0.65 + 0.35 * (resident >= 24.0 ? 1.0 : resident / 24.0));
      const double step_s = 2.0 * c.bmu * c.bnv * c.bkp / (rate * 1e12 / 256.0);
      const double tm = quanta * ((double)c.chunk / c.bkp + overhead[cfg]) * step_s + (c.nsplit + 1) * wbytes / 8.0e12 + 3e-6;
      if (tm < best) { best = tm; bp = c; }
    }
  }
  return bp;
}

}  // namespace

extern "C" int hwg_conv_pack_weight(const float* src, float* dst, int A, int B, int Bpad, int R, int S,
                                    long long sa, long long sb, long long sr, long long ss, int flip, void* stream) {
  HWG_REQUIRE(src && dst && A > 0 && B > 0 && Bpad >= B && R > 0 && S > 0, "conv_pack_weight: bad arguments");
  const long long total = (long long)R * S * A * Bpad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, dst, A, B, Bpad, R, S, sa, sb, sr, ss, flip);
  HWG_LAUNCH_CHECK("conv_pack_weight");
  return HWG_OK;
}

extern "C" int hwg_conv_pack_weight_multi(const void* table, int n_entries, long long total_blocks, void* stream) {
  HWG_REQUIRE(table && n_entries > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "conv_pack_weight_multi: bad arguments");
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)table, n_entries);
  HWG_LAUNCH_CHECK("conv_pack_weight_multi");
  return HWG_OK;
}

extern "C" int hwg_conv_pack_weight_multi_scaled(const void* table, int n_entries, long long total_blocks, float* dst_base, const float* scale_base,
                                                 void* stream) {
  HWG_REQUIRE(table && dst_base && scale_base && n_entries > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "conv_pack_weight_multi_scaled: bad arguments");
  hipLaunchKernelGGL(pack_weight_multi_scaled_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackEntryS*)table, n_entries,
                     dst_base, scale_base);
  HWG_LAUNCH_CHECK("conv_pack_weight_multi_scaled");
  return HWG_OK;
}

// direct kernels for single-channel ends live in conv_direct.hip
int hwg_conv_c1_fwd_impl(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int accumulate, hipStream_t st);
int hwg_conv_to1_fwd_impl(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int accumulate, hipStream_t st);
int hwg_conv_wgrad_direct_impl(const hwg_conv_desc* d, const float* u, const float* v, float* dw, long long sa, long long sb,
                               long long sr, long long ss, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
size_t hwg_conv_wgrad_direct_workspace(const hwg_conv_desc* d);

static int check_desc(const hwg_conv_desc* d, const char* who) {
  HWG_REQUIRE(d, "%s: null descriptor", who);
  HWG_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->P > 0 && d->Q > 0,
              "%s: non-positive dimension", who);
  HWG_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0, "%s: bad stride/dilation", who);
  return HWG_OK;
}

// shared with conv_wino.hip: the fixed-order sum of split partial outputs
int hwg_conv_split_reduce_launch(const float* part, const float* bias, float* y, long long total, int K, int nsplit, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(conv_split_reduce_kernel, dim3(hwg_stream_grid(total / 4 + 1, 256)), dim3(256), 0, st, part, bias, y, total, K, nsplit, accumulate);
  HWG_LAUNCH_CHECK("conv_split_reduce");
  return HWG_OK;
}

struct ConvPlan {
  bool mfma;
  int bm, bn, bk, nsplit, classes;
  long long Mc;
  int merged, KN, Pm, Qm;   // transposed with R % sh == 0 and S % sw == 0: parity classes merged into the GEMM's N dimension (ConvK.mode 2)
  double model_s;   // modelled duration of the chosen schedule (seconds)
};
static ConvPlan plan_conv_model(const hwg_conv_desc* d);
static ConvPlan plan_conv(const hwg_conv_desc* d) {
  static thread_local HwgPlanCache<ConvPlan> cache;
  return cache.get(d, plan_conv_model);
}
static ConvPlan plan_conv_model(const hwg_conv_desc* d) {
  ConvPlan p;
  p.mfma = !((!d->transposed) && (d->K <= 2 || d->C == 1));
  p.classes = d->transposed ? d->stride_h * d->stride_w : 1;
  p.Mc = (long long)d->N * d->P * d->Q;
  if (d->transposed) p.Mc = (long long)d->N * hwg_cdiv(d->P, d->stride_h) * hwg_cdiv(d->Q, d->stride_w);
  p.merged = 0; p.KN = d->K; p.Pm = p.Qm = 0;
  // Measured (tools/probes/probe_r5_merge.txt, profiles/r05_probe_merge.txt): the merged form only pays where a class alone cannot fill the narrowest
  // tile - 8x32x244x32->16 (the generator's last up-convolution) 48.8 -> 25.4 us; at K >= 32 it is equal (style extractor's 4x4 stride-2 data
  // gradients 214 vs 216 us: those launches are bound by their 2.07 rounds of workgroups, not by the operand re-reads) or slower where the
  // class grid had the better tile count (8x8x61x128->64 20.5 -> 35 us). HWG_CONV_MERGE: 0 never, 1 (default) K < 32, 2 always (tests).
  if (d->transposed && p.classes > 1 && d->R % d->stride_h == 0 && d->S % d->stride_w == 0 &&
      (hwg_tune().conv_merge == 2 || (hwg_tune().conv_merge == 1 && d->K < 32))) {
    // block grid shared by all classes: class c's blocks are shifted by floor((c + pad) / stride)
    for (int c = 0; c < d->stride_h; ++c) p.Pm = std::max(p.Pm, hwg_cdiv(d->P - c, d->stride_h) + (c + d->pad_h) / d->stride_h);
    for (int c = 0; c < d->stride_w; ++c) p.Qm = std::max(p.Qm, hwg_cdiv(d->Q - c, d->stride_w) + (c + d->pad_w) / d->stride_w);
    p.merged = 1; p.KN = p.classes * d->K; p.Mc = (long long)d->N * p.Pm * p.Qm; p.classes = 1;
  }
  // Schedule = (tile, split-K factor) with the smallest modelled time. The model (fitted to tools/conv_sweep.py measurements):
  //  * a CU retires one workgroup "quantum" at a time (co-resident workgroups share its matrix cores) and every XCD (32 CUs) owns
  //    a fixed 1/8 of the M tiles, so the makespan is ceil(blocks_per_XCD / 32) quanta (beyond 8 quanta the tail evens out);
  //  * a workgroup costs (K steps + fill/epilogue overhead) x step time at the tile's sustained rate;
  //  * split-K adds one pass that reads the nsplit partial images and writes the output.
  const int bk = (d->C % 32 == 0) ? 32 : 16;
  const int taps_total = d->R * d->S;
  const double taps_per_class = (double)taps_total / (d->transposed ? d->stride_h * d->stride_w : 1);
  const double T_total = taps_per_class * (d->C / bk);
  const int min_taps = d->transposed ? (d->R / d->stride_h) * (d->S / d->stride_w) : taps_total;
  const int min_steps = (min_taps > 0 ? min_taps : 1) * (d->C / bk);
  const double out_bytes = 4.0 * d->N * d->P * d->Q * d->K;
  struct Tile { int bm, bn; double tflops, overhead; };
  static const Tile tiles[4] = {{128, 128, 108.0, 1.0}, {128, 64, 94.0, 1.25}, {64, 64, 88.0, 1.5}, {128, 32, 62.0, 1.5}};
  static const int splits[8] = {1, 2, 3, 4, 6, 8, 12, 16};
  int bm = 128, bn = 32, ns = 1;
  double best = 1e30;
  for (int ti = 0; ti < 4; ++ti) {
    const Tile& t = tiles[ti];
    if (p.KN <= 32 ? t.bn != 32 : (t.bn == 32 || (t.bn == 128 && p.KN < 96))) continue;
    const double step_s = 2.0 * t.bm * t.bn * bk / (t.tflops * 1e12 / 256.0);
    const double per_xcd = (double)hwg_cdiv(hwg_cdiv(p.Mc, t.bm), 8) * hwg_cdiv(p.KN, t.bn) * p.classes;
    for (int si = 0; si < 8; ++si) {
      const int n = splits[si];
      if (n > 1 && (min_steps / n < 3 || out_bytes * n > 1.5e9)) break;
      const double q = per_xcd * n / 32.0;
      const double quanta = q <= 8.0 ? ceil(q) : q + 0.5;
      double tm = quanta * (T_total / n + t.overhead) * step_s;
      // the 4-wave 64x64 workgroup is rated for a CU that holds several of them (they cover each other's barrier / LDS phases);
      // alone on its CU it measures 1.0 us per 32-channel step instead of 0.65 (tools/conv_probe.py: 8x8x122x128->128 3x3 44 us at
      // 244 workgroups, 27 us per 244 at 976)
      if (t.bm == 64 && q < 2.0) tm /= (q <= 1.0 ? 0.65 : 0.65 + 0.35 * (q - 1.0));
      if (n > 1) tm += (n + 1) * out_bytes / 8.0e12 + 6e-6;
      if (tm < best) { best = tm; bm = t.bm; bn = t.bn; ns = n; }
    }
  }
  if (const char* f = hwg_tune().conv_force; *f) {  // tuning aid: "bm,bn,bk[,nsplit]" (ignored when the shape cannot use it)
    int fm = 0, fn = 0, fk = 0, fs = 1;
    if (sscanf(f, "%d,%d,%d,%d", &fm, &fn, &fk, &fs) >= 3) {
      if ((fm == 128 && (fn == 128 || fn == 64 || fn == 32)) || (fm == 64 && fn == 64)) { bm = fm; bn = fn; }
      if (fs >= 1 && fs <= 32) ns = fs;
      while (ns > 1 && min_steps / ns < 3) --ns;
    }
  }
  p.bm = bm; p.bn = bn; p.bk = bk; p.nsplit = ns;
  p.model_s = best;
  return p;
}

// modelled duration of the direct implicit-GEMM schedule (conv_wino.hip weighs its own model against it)
double hwg_conv_direct_model_seconds(const hwg_conv_desc* d) {
  if (!d || d->C % 16 != 0) return 1e30;
  const ConvPlan p = plan_conv(d);
  return p.mfma ? p.model_s : 1e30;
}

extern "C" size_t hwg_conv_fwd_workspace(const hwg_conv_desc* d) {
  if (!d || d->C % 16 != 0) return 0;
  const ConvPlan p = plan_conv(d);
  if (!p.mfma || p.nsplit <= 1) return 0;
  return (size_t)p.nsplit * d->N * d->P * d->Q * d->K * sizeof(float);
}

extern "C" int hwg_conv_fwd(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                            int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_desc(d, "conv_fwd");
  if (rc) return rc;
  HWG_REQUIRE(x && w && y, "conv_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (!d->transposed) {
    // geometry must be consistent
    const int Pexp = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
    const int Qexp = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
    HWG_REQUIRE(Pexp == d->P && Qexp == d->Q, "conv_fwd: output size %dx%d does not match geometry (%dx%d)", d->P, d->Q, Pexp, Qexp);
    if (d->K <= 2) return hwg_conv_to1_fwd_impl(d, x, w, bias, y, accumulate, st);
    if (d->C == 1) return hwg_conv_c1_fwd_impl(d, x, w, bias, y, accumulate, st);
  } else {
    HWG_REQUIRE(d->dil_h == 1 && d->dil_w == 1, "conv_fwd: transposed mode needs dilation 1");
    const int Pexp = (d->H - 1) * d->stride_h - 2 * d->pad_h + d->R;
    const int Qexp = (d->W - 1) * d->stride_w - 2 * d->pad_w + d->S;
    HWG_REQUIRE(Pexp <= d->P && d->P < Pexp + d->stride_h && Qexp <= d->Q && d->Q < Qexp + d->stride_w,
                "conv_fwd: transposed output size %dx%d inconsistent with geometry (%dx%d)", d->P, d->Q, Pexp, Qexp);
  }
  HWG_REQUIRE(d->C % 16 == 0, "conv_fwd: MFMA path needs C %% 16 == 0 (got C=%d); pad the channels", d->C);

  const ConvPlan p = plan_conv(d);
  const size_t need = hwg_conv_fwd_workspace(d);
  if (need && (!workspace || workspace_bytes < need)) {
    hwg_set_error("conv_fwd: workspace too small (%zu < %zu), size it with hwg_conv_fwd_workspace()", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  ConvK k;
  k.x = x; k.w = w; k.bias = bias; k.y = y;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
  k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
  k.P = d->P; k.Q = d->Q;
  k.mode = d->transposed ? (p.merged ? 2 : 1) : 0;
  k.KN = p.KN; k.Pm = p.Pm; k.Qm = p.Qm;
  k.accumulate = accumulate;
  k.nsplit = p.nsplit;
  k.part = (float*)workspace;
  k.cnt = nullptr;
  k.dbg = hwg_tune().conv_dbg;
  const int bm = p.bm, bn = p.bn, bk = p.bk;
  k.ntm = hwg_cdiv(p.Mc, bm);
  k.ntm_pad = (k.ntm + 7) / 8 * 8;
  dim3 grid(k.ntm_pad, hwg_cdiv(p.KN, bn) * p.nsplit, p.classes);
  if (p.nsplit > 1 && hwg_tune().split_inkernel) {
    const int wmn = bm == 64 ? 4 : 4 * (bn / 32);                      // wavefront sub-tiles per tile (WAVES_M x WAVES_N of the cases below)
    if ((long long)k.ntm_pad * hwg_cdiv(p.KN, bn) * p.classes * wmn <= HWG_SPLIT_COUNTERS) {
      k.cnt = hwg_split_counters(st);
      if (!k.cnt) return HWG_ERR_LAUNCH;
    }
  }
  // algorithmic work: every output pixel x K x C x taps (transposed: every input pixel feeds RxS outputs)
  const double pix = d->transposed ? (double)d->N * d->H * d->W : (double)d->N * d->P * d->Q;
  const int prof = hwg_prof_open(HWG_PROF_CONV, 2.0 * pix * d->K * d->C * d->R * d->S, st);
  const int pf = hwg_tune().conv_pf;
#define HWG_CONV_CASE(BM_, BN_, WMW, WNW)                                   \
  if (bm == BM_ && bn == BN_) {                                             \
    if (bk == 32) launch_conv<BM_, BN_, 32, WMW, WNW>(k, grid, st, pf);     \
    else launch_conv<BM_, BN_, 16, WMW, WNW>(k, grid, st, pf);              \
  } else
  // 16 / 8 wavefronts per workgroup on the big tiles: same LDS footprint, twice / four times the resident waves per SIMD to overlap
  // the gather phase of one wave with the MFMA phase of another (measured +8..15 % over 4-wave workgroups)
  HWG_CONV_CASE(128, 128, 4, 4)
  HWG_CONV_CASE(128, 64, 4, 2)
  HWG_CONV_CASE(128, 32, 4, 1)
  HWG_CONV_CASE(64, 64, 2, 2)
  { hwg_set_error("conv_fwd: no tile config for bm=%d bn=%d", bm, bn); return HWG_ERR_ARG; }
#undef HWG_CONV_CASE
  hwg_prof_close(prof, st);
  hwg_note_plan(HWG_PROF_CONV, bm * 1000 + bn + (p.merged ? 1000000 : 0), p.nsplit);      // (+ 1000000: merged parity classes)
  HWG_LAUNCH_CHECK("conv_fwd");
  if (p.nsplit > 1 && !k.cnt) {
    const long long total = (long long)d->N * d->P * d->Q * d->K;
    const int prof2 = hwg_prof_open(HWG_PROF_CONV_REDUCE, 4.0 * total * (p.nsplit + 1), st);
    hipLaunchKernelGGL(conv_split_reduce_kernel, dim3(hwg_stream_grid(total / 4 + 1, 256)), dim3(256), 0, st, (const float*)workspace, bias, y, total,
                       d->K, p.nsplit, accumulate);
    hwg_prof_close(prof2, st);
    HWG_LAUNCH_CHECK("conv_split_reduce");
  }
  return HWG_OK;
}

// C == 1 with at most 64 taps: taps-as-N MFMA kernel; other single/double channel ends: direct kernels (conv_direct.hip)
// ---- narrow layers (K, C <= 32: the generator's 16 / 32-channel tail at 64x488 and 32x244) -----------------------------------------
// These products are memory bound (36 FLOP/B at 16x16 channels) and fill a quarter of the 32-wide tiles of the kernel above, which also
// re-reads u and v once per tap. Here one wavefront owns a 16 (k) x 16 (c) block for ALL taps (R*S accumulator blocks of
// v_mfma_f32_16x16x4_f32, contraction = 4 output pixels per instruction) and walks a range of output pixels: dy is read once, every tap's
// operand is the same input neighbourhood (L1 / L2 hits after the first tap). The wavefronts of a workgroup that share a block are summed
// through LDS in a fixed tree, one partial image per workgroup goes to the usual [split][tap][K][C] (+ K bias sums) layout.
// Measured (8 x 64 x 488, 16 -> 16, 3x3): 34 + 8 us against 87 + 8 us; with 2 or 4 channel blocks the block-sharing wavefronts re-read each
// other's operands and the 32 x 32 tiles of the kernel above are no longer half empty - those layers stay there (HWG_WGRAD_NARROW=2 forces
// this kernel for them, used by the tests).
template <int R, int S>
__global__ __launch_bounds__(512) void wgrad_narrow_kernel(WgK a, int kbn, int cbn) {
  constexpr int NT = R * S;
  __shared__ __attribute__((aligned(16))) float red[4 * (NT * 256 + 16)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = kbn * cbn, role = wid % nblk, stream = wid / nblk, streams = 8 / nblk;
  const int kb = role / cbn, cb = role % cbn;
  const int ch = lane & 15, j = lane >> 4;
  const int kch = min(kb * 16 + ch, a.K - 1), cch = min(cb * 16 + ch, a.C - 1);      // clamped: ragged blocks are masked at the store
  const long long G4 = ((long long)a.Mtot + 3) >> 2;
  const int set = a.sets > 1 ? (int)blockIdx.x / a.gx : 0;
  const float* __restrict__ au = a.u + (a.set_on_v ? 0ll : (long long)set * a.Mtot * a.K);
  const float* __restrict__ avv = a.v + (a.set_on_v ? (long long)set * a.N * a.H * a.W * a.C : 0ll);
  const long long gs = (long long)((int)blockIdx.x - set * a.gx) * streams + stream, GS = (long long)a.gx * streams;
  const long long g0 = G4 * gs / GS, g1 = G4 * (gs + 1) / GS;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int PQ = a.P * a.Q;
  // operands of one group of 4 pixels: loads from clamped coordinates (no control flow), validity as a bit mask applied at use;
  // the next group's loads are issued before the current group's MFMAs (a wavefront only walks ~15-60 groups: without the prefetch
  // every one of them would expose a full memory latency)
  constexpr int GB = 1;                    // groups fetched together (measured: 1 -> 34 us, 4 -> 40 us on 8 x 64 x 488 x 16 x 16: occupancy wins)
  float av_c[GB], av_n[GB], bv_c[GB][NT], bv_n[GB][NT];
  int ok_c[GB], ok_n[GB];
  // pixel coordinates of this lane's pixel in the group to fetch next: decoded once, then advanced by 4 pixels per group (the integer
  // divisions of a per-group decode cost more VALU time than the nine MFMAs of the group)
  int cn, cp, cq;
  {
    const long long m0 = g0 * 4 + j;
    const int mm0 = m0 < a.Mtot ? (int)m0 : 0;
    cn = mm0 / PQ;
    const int pq0 = mm0 - cn * PQ;
    cp = pq0 / a.Q;
    cq = pq0 - cp * a.Q;
  }
  auto fetch = [&](long long g, float& av, float (&bv)[NT], int& okm) {
    const long long m = g * 4 + j;
    const bool mv = m < a.Mtot;
    const int mm = mv ? (int)m : 0;
    const int n = mv ? cn : 0, p = mv ? cp : 0, q = mv ? cq : 0;
    // advance to the pixel of the next group (rows shorter than 4 pixels may wrap more than once)
    cq += 4;
    while (cq >= a.Q) {
      cq -= a.Q;
      if (++cp == a.P) { cp = 0; ++cn; }
    }
    av = au[(long long)mm * a.K + kch];
    const int ih0 = p * a.sh - a.ph, iw0 = q * a.sw - a.pw;
    int roff[R], coff[S], rokm = 0, cokm = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ih = ih0 + r;
      rokm |= (ih >= 0 && ih < a.H ? 1 : 0) << r;
      roff[r] = ((n * a.H + min(max(ih, 0), a.H - 1)) * a.W) * a.C + cch;
    }
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) {
      const int iw = iw0 + s2;
      cokm |= (iw >= 0 && iw < a.W ? 1 : 0) << s2;
      coff[s2] = min(max(iw, 0), a.W - 1) * a.C;
    }
    int m2 = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) bv[r * S + s2] = avv[roff[r] + coff[s2]];
      m2 |= ((rokm >> r) & 1 ? cokm : 0) << (r * S);
    }
    okm = mv ? (m2 | (1 << NT)) : 0;
  };
#pragma unroll
  for (int u = 0; u < GB; ++u) fetch(g0 + u, av_c[u], bv_c[u], ok_c[u]);       // groups past g1 are masked by their pixel index only when
  // they also pass Mtot; the range check below drops the others
  for (long long g = g0; g < g1; g += GB) {
#pragma unroll
    for (int u = 0; u < GB; ++u) fetch(g + GB + u, av_n[u], bv_n[u], ok_n[u]);
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int okm = g + u < g1 ? ok_c[u] : 0;
      const float av = (okm >> NT) & 1 ? av_c[u] : 0.f;
      bsum += av;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = (okm >> t) & 1 ? bv_c[u][t] : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      av_c[u] = av_n[u]; ok_c[u] = ok_n[u];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv_c[u][t] = bv_n[u][t];
    }
  }
  // bias: sum over the four pixel groups of the wavefront -> lanes 0..15 hold the column sums of their channel
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  // fixed-order tree over the streams that share a block: the upper half writes, the lower half adds
  for (int half = streams >> 1; half >= 1; half >>= 1) {
    float* slot = red + ((stream - half) * nblk + role) * (NT * 256 + 16);
    if (stream >= half && stream < 2 * half) {
#pragma unroll
      for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(slot + t * 256 + lane * 4) = acc[t];
      if (lane < 16) slot[NT * 256 + lane] = bsum;
    }
    __syncthreads();
    if (stream < half) {
      const float* src = red + (stream * nblk + role) * (NT * 256 + 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(src + t * 256 + lane * 4);
        acc[t] += o;
      }
      if (lane < 16) bsum += src[NT * 256 + lane];
    }
    __syncthreads();
  }
  if (stream == 0) {
    float* pout = a.part + (long long)blockIdx.x * a.pstride;
    const int c = cb * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kb * 16 + (lane >> 4) * 4 + e;
        if (k < a.K && c < a.C) pout[((long long)t * a.K + k) * a.C + c] = acc[t][e];
      }
    if (a.bias_on && cb == 0 && lane < 16 && kb * 16 + lane < a.K) pout[(long long)NT * a.K * a.C + kb * 16 + lane] = bsum;
  }
}
// ---- single gathered channel, 64 anchor channels (the first layers: recogniser 3x3, style extractor 5x5, discriminator 7x7) --------------
// dw[k][tap] = sum_pix dy[pix][k] * x[pix + tap] is memory bound on paper (2 * taps FLOP per 4 bytes of dy), so a VALU kernel that reads dy once
// looked attractive next to the taps-as-N MFMA kernel (7.7-40 TFLOP/s). Built and MEASURED SLOWER in the training step (profiles/r06_census_c1.txt):
// lane = channel k for the dy loads, lane = pixel for the gather of the group's R x S tap values, the walk over the group's 64 pixels broadcasts
// each tap value with v_readlane into an FMA - 2 x taps VALU instructions of 4 cycles each per pixel and wavefront: 119 us for the
// discriminator's 7x7 layer (MFMA kernel: 74), 65 us for the 5x5 (42-71), 34 us for the 3x3 (39). OFF by default (HWG_WGRAD_C1=1 runs it;
// the parity cases c1valu_* keep it correct). Partial images [split][tap][K] (+K bias sums) as every other weight-gradient kernel writes them.
template <int R, int S>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(WgK a) {
  constexpr int T = R * S, NW = 4, DEPTH = 8;
  __shared__ float red[NW][(T + 1) * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = a.sets > 1 ? (int)blockIdx.x / a.gx : 0;
  const float* __restrict__ au = a.u + (a.set_on_v ? 0ll : (long long)set * a.Mtot * 64);
  const float* __restrict__ av = a.v + (a.set_on_v ? (long long)set * a.N * a.H * a.W : 0ll);
  const long long G = ((long long)a.Mtot + 63) >> 6;
  const long long w = (long long)((int)blockIdx.x - set * a.gx) * NW + wid, Wt = (long long)a.gx * NW;
  const long long g0 = G * w / Wt, g1 = G * (w + 1) / Wt;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  float bsum = 0.f;
  const int PQ = a.P * a.Q;
  for (long long g = g0; g < g1; ++g) {
    const int pix0 = (int)(g << 6);
    // this lane's pixel of the group: its R x S tap values (zero outside the image / past the last pixel)
    float xv[T];
    {
      const int mp = pix0 + lane;
      const bool mv = mp < a.Mtot;
      const int mc = mv ? mp : a.Mtot - 1;
      const int n = mc / PQ, rem = mc - n * PQ;
      const int pp = rem / a.Q, qq = rem - pp * a.Q;
      const int ih0 = pp * a.sh - a.ph, iw0 = qq * a.sw - a.pw;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ih = ih0 + r * a.dh;
        const bool rok = mv && ih >= 0 && ih < a.H;
        const int rbase = (n * a.H + min(max(ih, 0), a.H - 1)) * a.W;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
          const int iw = iw0 + s2 * a.dw;
          const float x = av[rbase + min(max(iw, 0), a.W - 1)];
          xv[r * S + s2] = (rok && iw >= 0 && iw < a.W) ? x : 0.f;
        }
      }
    }
    const int npix = min(64, a.Mtot - pix0);
    for (int j0 = 0; j0 < npix; j0 += DEPTH) {
      float d[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int j = j0 + u < npix ? j0 + u : j0;
        d[u] = au[(long long)(pix0 + j) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int j = j0 + u;                     // wave-uniform
        if (j >= npix) break;
        const float dv = d[u];
        bsum += dv;
#pragma unroll
        for (int t = 0; t < T; ++t)
          acc[t] = fmaf(dv, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[t]), j)), acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) red[wid][t * 64 + lane] = acc[t];
  red[wid][T * 64 + lane] = bsum;
  __syncthreads();
  float* pout = a.part + (long long)blockIdx.x * a.pstride;
  for (int i = tid; i < (T + 1) * 64; i += 256) {
    const float sres = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    if (i < T * 64) pout[i] = sres;                        // [tap][K = 64][C = 1]
    else if (a.bias_on) pout[i] = sres;                    // K bias sums behind the image
  }
}
// ---- the same layers on the matrix cores with the input ROWS staged in LDS (round 6, the default) ------------------------------------------------
// dw[k][tap] = sum_pix dy[pix][k] * x[pix + tap] as a GEMM with M = 64 filters, N = taps, contraction = pixels. The taps-as-N kernel above
// builds its im2col tile tap by tap (a clamped 4-byte gather with its own coordinate arithmetic per element) and is bound by that gather. Here a
// workgroup walks 128-pixel segments of ONE gradient row: the R input rows under it are staged in LDS once (coalesced, zero-filled outside the
// image), lane n owns tap n for the whole launch, so its B operand of pixel pair j is rows[off(n) + j] - one ds_read_b32 at an address that
// only advances - and the A operand (dy[pixel][k]) comes straight from global memory, 128 contiguous bytes per half-wavefront, all loads of
// a segment issued before the first MFMA. v_mfma_f32_32x32x2_f32, 2 x ceil(taps / 32) per pixel pair; the bias gradient is the column sum of
// the A operands. One partial image [tap][64] (+64 bias sums) per workgroup, waves summed in a fixed order.
template <int R, int S>
__global__ __launch_bounds__(256) void wgrad_c1_rows_kernel(WgK a, int qtiles, int nseg) {
  constexpr int T = R * S, NB = (T + 31) / 32, QT = 128, RW = QT + S - 1, STEPS = 16;
  __shared__ float rows[R * RW];
  __shared__ float red[3][2 * NB * 16 * 64 + 128];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = a.sets > 1 ? (int)blockIdx.x / a.gx : 0;
  const float* __restrict__ au = a.u + (a.set_on_v ? 0ll : (long long)set * a.Mtot * 64);
  const float* __restrict__ av = a.v + (a.set_on_v ? (long long)set * a.N * a.H * a.W : 0ll);
  const int b0 = (int)blockIdx.x - set * a.gx;
  int toff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int tap = min(nb * 32 + l31, T - 1);                 // (lanes past the last tap repeat it: their columns are never stored)
    toff[nb] = (tap / S) * RW + tap % S + wid * 32 + lhi;
  }
  f32x16 acc[2][NB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
  float bsum[2] = {0.f, 0.f};
  for (int seg = b0; seg < nseg; seg += a.gx) {
    const int qt = seg % qtiles, t2 = seg / qtiles;
    const int p = t2 % a.P, n = t2 / a.P;
    const int q0 = qt * QT;
    // A operands of this wavefront's 32 pixels: dy[(n, p, q0 + 32 wid + 2 st + lhi)][l31 (+32)], zero past the row's end
    float af[STEPS][2];
    const float* dyrow = au + ((long long)(n * a.P + p) * a.Q) * 64;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int q = q0 + wid * 32 + 2 * st + lhi;
      const bool ok = q < a.Q;
      const float* src = dyrow + (long long)(ok ? q : 0) * 64 + l31;
      const float v0 = src[0], v1 = src[32];
      af[st][0] = ok ? v0 : 0.f; af[st][1] = ok ? v1 : 0.f;
    }
    __syncthreads();                                    // the previous segment's operand reads are done
    for (int i = tid; i < R * RW; i += 256) {
      const int r = i / RW, c = i - r * RW;
      const int ih = p - a.ph + r, iw = q0 - a.pw + c;
      const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      rows[i] = ok ? av[((long long)n * a.H + ih) * a.W + iw] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      bsum[0] += af[st][0]; bsum[1] += af[st][1];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float bfv = rows[toff[nb] + 2 * st];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[st][mb], bfv, acc[mb][nb], 0, 0, 0);
      }
    }
  }
  // fixed-order sum over the four wavefronts: 3, 2, 1 park their accumulators, wavefront 0 adds them in that order and writes the image
  bsum[0] += __shfl_xor(bsum[0], 32, 64); bsum[1] += __shfl_xor(bsum[1], 32, 64);
  __syncthreads();
  if (wid > 0) {
    float* slot = red[wid - 1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) slot[((mb * NB + nb) * 16 + e) * 64 + lane] = acc[mb][nb][e];
    if (lane < 32) { slot[2 * NB * 16 * 64 + lane] = bsum[0]; slot[2 * NB * 16 * 64 + 32 + lane] = bsum[1]; }
  }
  __syncthreads();
  if (wid == 0) {
    float* pout = a.part + (long long)blockIdx.x * a.pstride;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = ((mb * NB + nb) * 16 + e) * 64 + lane;
          const float v = ((acc[mb][nb][e] + red[0][i]) + red[1][i]) + red[2][i];
          // C/D layout: col (tap) = lane & 31, row (k) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
          const int tap = nb * 32 + l31, k = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (tap < T) pout[tap * 64 + k] = v;
        }
    if (a.bias_on && lane < 32) {
      const int o = 2 * NB * 16 * 64;
      pout[T * 64 + lane] = ((bsum[0] + red[0][o + lane]) + red[1][o + lane]) + red[2][o + lane];
      pout[T * 64 + 32 + lane] = ((bsum[1] + red[0][o + 32 + lane]) + red[1][o + 32 + lane]) + red[2][o + 32 + lane];
    }
  }
}
static bool wgrad_is_c1_rows(const hwg_conv_desc* d) {
  return hwg_tune().wgrad_c1 != 1 && hwg_tune().wgrad_c1_rows && d->C == 1 && d->K == 64 && d->R == d->S && (d->R == 3 || d->R == 5 || d->R == 7) &&
         d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && (long long)d->N * d->P * d->Q >= 4096 &&
         (long long)d->N * d->P * hwg_cdiv(d->Q, 128) < (1ll << 31) && (long long)d->N * d->H * d->W < (1ll << 31);
}
static bool wgrad_is_c1_valu(const hwg_conv_desc* d) {
  return hwg_tune().wgrad_c1 && d->C == 1 && d->K == 64 && d->R == d->S && (d->R == 3 || d->R == 5 || d->R == 7) &&
         (long long)d->N * d->P * d->Q >= 4096 && (long long)d->N * d->P * d->Q * 64 < (1ll << 40) && (long long)d->N * d->H * d->W < (1ll << 31);
}
static bool wgrad_is_narrow(const hwg_conv_desc* d) {
  const int mode = hwg_tune().wgrad_narrow;               // 0: never, 2: also layers with 2 / 4 channel blocks (tests)
  const bool off = mode == 0, all = mode == 2;
  const bool taps = (d->R == 3 && d->S == 3) || (d->R == 4 && d->S == 4);
  // LDS tree needs streams = 8 / blocks >= 1 and a power of two: 1, 2 or 4 blocks of 16 x 16
  const int kbn = hwg_cdiv(d->K, 16), cbn = hwg_cdiv(d->C, 16), nb = kbn * cbn;
  return !off && taps && d->dil_h == 1 && d->dil_w == 1 && d->C >= 4 && d->K > 2 && (nb == 1 || (all && (nb == 2 || nb == 4))) &&
         (long long)d->N * d->P * d->Q >= 16384 && (long long)d->N * d->H * d->W * d->C < (1ll << 31) && !d->transposed;
}
static int narrow_blocks(const hwg_conv_desc* d) {
  const int nb = hwg_cdiv(d->K, 16) * hwg_cdiv(d->C, 16), streams = 8 / nb;
  const long long G4 = ((long long)d->N * d->P * d->Q + 3) / 4;
  long long blocks = G4 / ((long long)streams * 8);          // at least 8 MFMA groups per wavefront
  if (blocks > 512) blocks = 512;                           // two workgroups per CU
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static bool wgrad_is_tapn(const hwg_conv_desc* d) { return d->C == 1 && d->K > 2 && d->K % 4 == 0 && d->R * d->S <= 64; }
static bool wgrad_is_direct(const hwg_conv_desc* d) { return (d->K <= 2 || d->C <= 2) && !wgrad_is_tapn(d); }
static WgPlan plan_wgrad_tapn(const hwg_conv_desc* d) {
  WgPlan p;
  p.cfg = 3; p.bmu = 64; p.bnv = 64; p.bkp = 32;
  p.tiles_u = hwg_cdiv(d->K, 64); p.tiles_v = 1;
  if (wgrad_is_c1_rows(d)) {      // wgrad_c1_rows_kernel: one partial image per workgroup, the workgroups stride over the 128-pixel segments
    p.cfg = 5;
    const long long nseg = (long long)d->N * d->P * hwg_cdiv(d->Q, 128);
    p.nsplit = (int)(nseg < 512 ? nseg : 512);
    p.chunk = 0;
    return p;
  }
  if (wgrad_is_c1_valu(d)) {      // wgrad_c1_kernel: one partial image per workgroup of four wavefronts, three workgroups per CU
    p.cfg = 4;
    wg_split(p, (long long)d->N * d->P * d->Q, 768);
    return p;
  }
  wg_split(p, (long long)d->N * d->P * d->Q, hwg_cdiv(1024, p.tiles_u));
  return p;
}

// sums the partial images part[split][tap][K][C] (+ K bias sums at the end of every image, pstride floats apart) in a fixed order and writes
// dw through the weight's strides (shared by the direct and the Winograd weight-gradient kernels)
struct RedSchedule { int variant, nblocks, main_blocks; size_t lds; };
static RedSchedule reduce_schedule(int nsplit, int RS, int S, int K, int C, long long sb, long long sr, long long ss, bool has_bias) {
  RedSchedule r; r.main_blocks = 0; r.lds = 0;
  const long long total = (long long)RS * K * C + (has_bias ? K : 0);
  // large filters cut into few ranges (the 256..512-channel layers): row-contiguous stores. Measured (tools/probes/probe_r3_reduce.txt, whole weight
  // gradient): 512x512x3x3 / 4 images 144 -> 138 us, 256x256x3x3 / 16 images 48.7 -> 45.0 us, 512x512x1x3 / 4 images 32.6 -> 31.0 us; with many
  // images of a small filter the lane-split kernels below stay ahead (128x128x3x3 / 62 images 38 vs 44 us).
  const long long pairs = (long long)K * C;
  if (ss == 1 && sr == S && sb == RS && hwg_tune().wgrad_reduce_rows && RS <= 49 && pairs >= 65536 && nsplit <= 16) {
    const int sl = nsplit >= 8 && pairs <= 65536 ? 4 : 1;
    const int kc = 256 / sl;
    r.main_blocks = (int)hwg_cdiv(pairs, kc);
    r.nblocks = r.main_blocks + (has_bias ? hwg_cdiv(K, 256) : 0);
    r.lds = (size_t)256 * (RS | 1) * sizeof(float);
    const int rsc = RS == 9 ? 0 : RS == 16 ? 1 : RS == 3 ? 2 : 3;
    r.variant = RED_ROWS + 2 * rsc + (sl == 4 ? 1 : 0);
    return r;
  }
  if (nsplit >= 64 && total <= 65536) { r.variant = RED_LANES32; r.nblocks = hwg_cdiv(total, 8); r.lds = 1024; }
  else if (nsplit >= 8 && total <= 262144) { r.variant = RED_LANES4; r.nblocks = hwg_cdiv(total, 64); r.lds = 1024; }
  else if (C % 4 == 0) { r.variant = RED_VEC4; r.nblocks = hwg_stream_grid(total / 4, 256); }
  else { r.variant = RED_SCALAR; r.nblocks = hwg_stream_grid(total, 256); }
  return r;
}

#include <mutex>
#include <vector>
namespace {
thread_local int g_defer_next = 0;          // set by hwg_wgrad_defer_next, consumed (and cleared) by the next weight-gradient entry point
std::mutex g_defer_mu;                       // queued from the autograd engine's thread, flushed from the caller's
std::vector<RedEntry> g_defer_queue;
}  // namespace
int hwg_prof_current_tag();
void hwg_prof_add_child(int parent, int kind, int tag, double work);
int hwg_pg_defer_flush(hipStream_t st, int* launches);
bool hwg_wgrad_defer_take() { const bool d = g_defer_next != 0; g_defer_next = 0; return d; }
extern "C" int hwg_wgrad_defer_next(void) { g_defer_next = 1; return HWG_OK; }
extern "C" long long hwg_wgrad_defer_pending(void) { std::lock_guard<std::mutex> lock(g_defer_mu); return (long long)g_defer_queue.size(); }

int hwg_wgrad_reduce_launch(const float* part, float* dw, int nsplit, int RS, int S, int K, int C, long long sa, long long sb, long long sr,
                            long long ss, int accumulate, long long pstride, float* dbias, int bias_accumulate, hipStream_t st, bool defer) {
  const long long total = (long long)RS * K * C + (dbias ? K : 0);
  if (total >= (1ll << 31)) { hwg_set_error("wgrad_reduce: weight image too large for 32-bit element indices"); return HWG_ERR_ARG; }
  const RedSchedule r = reduce_schedule(nsplit, RS, S, K, C, sb, sr, ss, dbias != nullptr);
  if (defer) {
    RedEntry e;
    e.part = part; e.dw = dw; e.dbias = dbias; e.pstride = pstride; e.sa = sa; e.sb = sb; e.sr = sr; e.ss = ss;
    e.nsplit = nsplit; e.RS = RS; e.S = S; e.K = K; e.C = C; e.accumulate = accumulate; e.bias_accumulate = bias_accumulate;
    e.variant = r.variant; e.first_block = 0; e.nblocks = r.nblocks; e.main_blocks = r.main_blocks; e.tag = hwg_prof_current_tag();
    std::lock_guard<std::mutex> lock(g_defer_mu);
    g_defer_queue.push_back(e);
    return HWG_OK;
  }
  const int v = r.variant;
  if (v >= RED_ROWS) {
#define HWG_ROWS(SL, RSC)                                                                                                                          \
  hipLaunchKernelGGL((wgrad_reduce_rows_kernel<SL, RSC>), dim3(r.nblocks), dim3(256), r.lds, st, part, dw, nsplit, RS, K, C, sa, accumulate, pstride, \
                     dbias, bias_accumulate, r.main_blocks)
    switch (v - RED_ROWS) {
      case 0: HWG_ROWS(1, 9); break;  case 1: HWG_ROWS(4, 9); break;  case 2: HWG_ROWS(1, 16); break;  case 3: HWG_ROWS(4, 16); break;
      case 4: HWG_ROWS(1, 3); break;  case 5: HWG_ROWS(4, 3); break;  case 6: HWG_ROWS(1, 0); break;   default: HWG_ROWS(4, 0); break;
    }
#undef HWG_ROWS
  } else if (v == RED_LANES32)
    hipLaunchKernelGGL(wgrad_reduce_lanes_kernel<32>, dim3(r.nblocks), dim3(256), 0, st, part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss,
                       accumulate, pstride, dbias, bias_accumulate);
  else if (v == RED_LANES4)
    hipLaunchKernelGGL(wgrad_reduce_lanes_kernel<4>, dim3(r.nblocks), dim3(256), 0, st, part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss,
                       accumulate, pstride, dbias, bias_accumulate);
  else if (v == RED_VEC4)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(r.nblocks), dim3(256), 0, st, part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss,
                       accumulate, pstride, dbias, bias_accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3(r.nblocks), dim3(256), 0, st, part, dw, nsplit, RS, S, K, C, sa, sb, sr, ss,
                       accumulate, pstride, dbias, bias_accumulate);
  HWG_LAUNCH_CHECK("conv_wgrad_reduce");
  return HWG_OK;
}

// sums every queued set of partial images (see wgrad_reduce_multi_kernel); returns the number of launches made through *launches
extern "C" int hwg_wgrad_defer_flush(void* stream, int* launches) {
  hipStream_t st = (hipStream_t)stream;
  std::vector<RedEntry> q;
  {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    q.swap(g_defer_queue);
  }
  int made = 0;
  std::vector<int> pass(q.size(), 0);
  int npass = 0;
  for (size_t i = 0; i < q.size(); ++i) {          // the k-th queued gradient of a tensor goes to pass k (same accumulation order as undeferred)
    for (size_t j = 0; j < i; ++j)
      if (q[j].dw == q[i].dw || (q[i].dbias && q[j].dbias == q[i].dbias)) pass[i] = pass[j] + 1 > pass[i] ? pass[j] + 1 : pass[i];
    if (pass[i] + 1 > npass) npass = pass[i] + 1;
  }
  for (int ps = 0; ps < npass; ++ps) {
    size_t i = 0;
    while (i < q.size()) {
      RedTable t; t.n = 0; t.pad = 0;
      int blocks = 0; size_t lds = 1024; double bytes = 0.0;
      int member[RED_MAX]; double mbytes[RED_MAX];
      for (; i < q.size() && t.n < RED_MAX; ++i) {
        if (pass[i] != ps) continue;
        RedEntry e = q[i];
        e.first_block = blocks;
        blocks += e.nblocks;
        const RedSchedule r = reduce_schedule(e.nsplit, e.RS, e.S, e.K, e.C, e.sb, e.sr, e.ss, e.dbias != nullptr);
        if (r.lds > lds) lds = r.lds;
        const double b = 4.0 * ((double)e.RS * e.K * e.C + (e.dbias ? e.K : 0)) * (e.nsplit + 1 + (e.accumulate ? 1 : 0));
        member[t.n] = e.tag; mbytes[t.n] = b; bytes += b;
        t.e[t.n++] = e;
      }
      if (t.n == 0) break;
      const int prof = hwg_prof_open(HWG_PROF_WGRAD_REDUCE, bytes, st);
      hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), lds, st, t);
      hwg_prof_close(prof, st);
      if (prof >= 0)                               // the launch's time is shared out over its entries' layers in proportion to their bytes
        for (int m = 0; m < t.n; ++m) hwg_prof_add_child(prof, HWG_PROF_WGRAD_REDUCE, member[m], mbytes[m]);
      HWG_LAUNCH_CHECK("wgrad_reduce_multi");
      ++made;
    }
  }
  if (launches) *launches = made;
  return hwg_pg_defer_flush(st, launches);      // the generator epilogues' queued parameter-gradient sums (norm_act.hip)
}

extern "C" size_t hwg_conv_wgrad_workspace(const hwg_conv_desc* d) {
  if (!d) return 0;
  if (wgrad_is_direct(d)) return hwg_conv_wgrad_direct_workspace(d);
  if (wgrad_is_narrow(d)) return (size_t)narrow_blocks(d) * ((size_t)d->R * d->S * d->K * d->C + d->K) * sizeof(float);
  WgPlan p = wgrad_is_tapn(d) ? plan_wgrad_tapn(d) : plan_wgrad(d);
  return (size_t)p.nsplit * ((size_t)d->R * d->S * d->K * d->C + d->K) * sizeof(float);
}

// Pixel ranges per set when `sets` gradients run as one launch: the schedule is planned for the whole launch (sets x the pixels) and its ranges
// are shared out over the sets - the per-set plan would fill the chip once per set, i.e. `sets` times as many partial images to write and sum.
static WgPlan plan_wgrad_sets(const hwg_conv_desc* d, int sets, bool tapn, bool narrow) {
  if (sets <= 1) {
    WgPlan p = tapn ? plan_wgrad_tapn(d) : plan_wgrad(d);
    if (narrow) p.nsplit = narrow_blocks(d);
    return p;
  }
  hwg_conv_desc all = *d;
  all.N *= sets;
  WgPlan p = tapn ? plan_wgrad_tapn(&all) : plan_wgrad(&all);
  const long long Mset = (long long)d->N * d->P * d->Q;
  if (narrow) {
    int nb = narrow_blocks(&all) / sets;
    p.nsplit = nb < 1 ? 1 : nb;
    return p;
  }
  wg_split(p, Mset, hwg_cdiv(p.nsplit, sets));
  return p;
}
extern "C" size_t hwg_conv_wgrad_sets_workspace(const hwg_conv_desc* d, int sets) {
  if (!d || sets < 1 || wgrad_is_direct(d)) return 0;
  const WgPlan p = plan_wgrad_sets(d, sets, wgrad_is_tapn(d), wgrad_is_narrow(d));
  return (size_t)sets * p.nsplit * ((size_t)d->R * d->S * d->K * d->C + d->K) * sizeof(float);
}

// one weight gradient, or `sets` of them that share the gathered tensor v (u = the sets' anchors back to back): one launch, one partial-image
// range and one (possibly deferred) sum per set
static int conv_wgrad_run(const hwg_conv_desc* d, const float* u, const float* v, int sets, int set_on_v, float* const* dws,
                          long long sa, long long sb, long long sr, long long ss, int accumulate,
                          float* const* dbiases, int bias_accumulate, void* workspace, size_t workspace_bytes, hipStream_t st, bool defer) {
  int rc = check_desc(d, "conv_wgrad");
  if (rc) return rc;
  HWG_REQUIRE(u && v && dws && dws[0] && sets >= 1, "conv_wgrad: null pointer");
  float* const dbias = dbiases ? dbiases[0] : nullptr;
  if (wgrad_is_direct(d)) {
    HWG_REQUIRE(sets == 1, "conv_wgrad_sets: not available on the direct (K<=2 / C<=2) path (hwg_conv_wgrad_sets_supported)");
    HWG_REQUIRE(!dbias, "conv_wgrad: the fused bias gradient is not available on the direct (K<=2 / C<=2) path, use hwg_colsum");
    return hwg_conv_wgrad_direct_impl(d, u, v, dws[0], sa, sb, sr, ss, accumulate, workspace, workspace_bytes, st);
  }
  const bool tapn = wgrad_is_tapn(d);
  HWG_REQUIRE(d->K % 4 == 0 && (tapn || d->C % 4 == 0), "conv_wgrad: channels must be multiples of 4 (K=%d C=%d)", d->K, d->C);
  const size_t need = sets == 1 ? hwg_conv_wgrad_workspace(d) : hwg_conv_wgrad_sets_workspace(d, sets);
  if (!workspace || workspace_bytes < need) {
    hwg_set_error("conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  const bool narrow = wgrad_is_narrow(d);
  const WgPlan p = plan_wgrad_sets(d, sets, tapn, narrow);
  WgK k;
  k.u = u; k.v = v; k.part = (float*)workspace;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
  k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
  k.P = d->P; k.Q = d->Q;
  k.Mtot = d->N * d->P * d->Q;
  k.chunk = p.chunk;
  k.tiles_v = p.tiles_v;
  k.bias_on = dbias ? 1 : 0;
  k.pstride = (long long)d->R * d->S * d->K * d->C + d->K;
  k.sets = sets; k.gx = p.nsplit; k.set_on_v = set_on_v;
  const int gx = p.nsplit * sets;
  dim3 grid(gx, p.tiles_u * p.tiles_v, tapn ? 1 : d->R * d->S);
  const int prof = hwg_prof_open(HWG_PROF_WGRAD, 2.0 * sets * k.Mtot * d->K * d->C * d->R * d->S, st);
  if (narrow && d->R == 3) hipLaunchKernelGGL((wgrad_narrow_kernel<3, 3>), dim3(gx), dim3(512), 0, st, k, hwg_cdiv(d->K, 16), hwg_cdiv(d->C, 16));
  else if (narrow) hipLaunchKernelGGL((wgrad_narrow_kernel<4, 4>), dim3(gx), dim3(512), 0, st, k, hwg_cdiv(d->K, 16), hwg_cdiv(d->C, 16));
  // 16 waves and 32-pixel K steps on the big tile: +10 % over 8 waves x 16 pixels (331 -> 299 us on 512x512x3x3 at 6096 pixels)
  else if (p.cfg == 0) hipLaunchKernelGGL((wgrad_mfma_kernel<128, 128, 32, 4, 4, 1>), grid, dim3(1024), 0, st, k);
  // 64x64: two wave groups split every 32-pixel K step between them (8 waves; 5..10 % over 4 waves on every measured shape)
  else if (p.cfg == 1) hipLaunchKernelGGL((wgrad_mfma_kernel<64, 64, 32, 2, 2, 2>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 5) {
    const int qtiles = hwg_cdiv(d->Q, 128), nseg = d->N * d->P * qtiles;
    if (d->R == 3) hipLaunchKernelGGL((wgrad_c1_rows_kernel<3, 3>), dim3(gx), dim3(256), 0, st, k, qtiles, nseg);
    else if (d->R == 5) hipLaunchKernelGGL((wgrad_c1_rows_kernel<5, 5>), dim3(gx), dim3(256), 0, st, k, qtiles, nseg);
    else hipLaunchKernelGGL((wgrad_c1_rows_kernel<7, 7>), dim3(gx), dim3(256), 0, st, k, qtiles, nseg);
  }
  else if (p.cfg == 4 && d->R == 3) hipLaunchKernelGGL((wgrad_c1_kernel<3, 3>), dim3(gx), dim3(256), 0, st, k);
  else if (p.cfg == 4 && d->R == 5) hipLaunchKernelGGL((wgrad_c1_kernel<5, 5>), dim3(gx), dim3(256), 0, st, k);
  else if (p.cfg == 4) hipLaunchKernelGGL((wgrad_c1_kernel<7, 7>), dim3(gx), dim3(256), 0, st, k);
  else if (p.cfg == 3) hipLaunchKernelGGL((wgrad_mfma_kernel<64, 64, 32, 2, 2, 2, true>), grid, dim3(512), 0, st, k);
  else hipLaunchKernelGGL((wgrad_mfma_kernel<32, 32, 32, 1, 1, 4>), grid, dim3(256), 0, st, k);
  hwg_prof_close(prof, st);
  hwg_note_plan(HWG_PROF_WGRAD, narrow ? 100 + d->R : (tapn ? 10 + p.cfg : p.cfg), p.nsplit);   // 103 / 104: all-taps narrow kernel, 1x: taps-as-N
  HWG_LAUNCH_CHECK("conv_wgrad");
  for (int s_ = 0; s_ < sets; ++s_) {
    const int prof2 = hwg_prof_open(HWG_PROF_WGRAD_REDUCE, 4.0 * ((double)d->R * d->S * d->K * d->C + (dbias ? d->K : 0)) * (p.nsplit + 1), st);
    rc = hwg_wgrad_reduce_launch((const float*)workspace + (long long)s_ * p.nsplit * k.pstride, dws[s_], p.nsplit, d->R * d->S, d->S, d->K, d->C, sa, sb,
                                 sr, ss, accumulate, k.pstride, dbiases ? dbiases[s_] : nullptr, bias_accumulate, st, defer);
    hwg_prof_close(prof2, st);
    if (rc) return rc;
  }
  return HWG_OK;
}

extern "C" int hwg_conv_wgrad(const hwg_conv_desc* d, const float* u, const float* v, float* dw,
                              long long sa, long long sb, long long sr, long long ss, int accumulate,
                              float* dbias, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  const bool defer = hwg_wgrad_defer_take();      // consumed here whatever path runs: a stale flag must never reach a later call
  float* dws[1] = {dw};
  float* dbs[1] = {dbias};
  return conv_wgrad_run(d, u, v, 1, 0, dws, sa, sb, sr, ss, accumulate, dbias ? dbs : nullptr, bias_accumulate, workspace, workspace_bytes,
                        (hipStream_t)stream, defer);
}

extern "C" int hwg_conv_wgrad_sets_supported(const hwg_conv_desc* d) {
  if (!d || wgrad_is_direct(d)) return 0;
  return d->K % 4 == 0 && (wgrad_is_tapn(d) || d->C % 4 == 0);
}

extern "C" int hwg_conv_wgrad_sets(const hwg_conv_desc* d, const float* u, const float* v, int sets, int set_on_v, const long long* dw_ptrs,
                                   long long sa, long long sb, long long sr, long long ss, int accumulate,
                                   const long long* dbias_ptrs, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  const bool defer = hwg_wgrad_defer_take();
  HWG_REQUIRE(sets >= 1 && sets <= 8 && dw_ptrs, "conv_wgrad_sets: 1..8 sets");
  float* dws[8]; float* dbs[8];
  for (int i = 0; i < sets; ++i) { dws[i] = (float*)(uintptr_t)dw_ptrs[i]; dbs[i] = dbias_ptrs ? (float*)(uintptr_t)dbias_ptrs[i] : nullptr; }
  HWG_REQUIRE(!(set_on_v && dbias_ptrs), "conv_wgrad_sets: the fused bias gradient sums the anchor u, which the sets share here");
  return conv_wgrad_run(d, u, v, sets, set_on_v, dws, sa, sb, sr, ss, accumulate, dbias_ptrs ? dbs : nullptr, bias_accumulate, workspace, workspace_bytes,
                        (hipStream_t)stream, defer);
}

static long long colsum_chunks(long long rows) {
  // ~16 rows per thread (16 row lanes per block); keep the second stage short (<= 256 partials per column; 128 until round 6: a quarter of
  // the chip on the generator's 250 k-row tensors; 512 made the second stage slower than the first got faster)
  long long chunks = (rows + 255) / 256;
  if (chunks > 256) chunks = 256;
  if (chunks < 1) chunks = 1;
  return chunks;
}
extern "C" size_t hwg_colsum_workspace(long long rows, int C) { return (size_t)colsum_chunks(rows) * C * sizeof(float); }

extern "C" int hwg_colsum(const float* x, long long rows, int C, float* out, int accumulate,
                          void* workspace, size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(x && out && rows > 0 && C > 0, "colsum: bad arguments");
  const size_t need = hwg_colsum_workspace(rows, C);
  if (!workspace || workspace_bytes < need) {
    hwg_set_error("colsum: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long chunks = colsum_chunks(rows);
  const long long rpc = (rows + chunks - 1) / chunks;
  float* direct = chunks == 1 ? out : (float*)nullptr;
  if (C == 1)
    hipLaunchKernelGGL(colsum_narrow_kernel<1>, dim3((unsigned)chunks), dim3(256), 0, st, x, rows, (float*)workspace, rpc, direct, accumulate);
  else if (C == 2)
    hipLaunchKernelGGL(colsum_narrow_kernel<2>, dim3((unsigned)chunks), dim3(256), 0, st, x, rows, (float*)workspace, rpc, direct, accumulate);
  else if (C == 3)
    hipLaunchKernelGGL(colsum_narrow_kernel<3>, dim3((unsigned)chunks), dim3(256), 0, st, x, rows, (float*)workspace, rpc, direct, accumulate);
  else if (C % 4 == 0)
    hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3((unsigned)chunks, hwg_cdiv(C, 64)), dim3(256), 0, st, x, rows, C, (float*)workspace, rpc,
                       chunks == 1 ? out : (float*)nullptr, accumulate);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3((unsigned)chunks, hwg_cdiv(C, 64)), dim3(256), 0, st, x, rows, C, (float*)workspace, rpc,
                       chunks == 1 ? out : (float*)nullptr, accumulate);
  HWG_LAUNCH_CHECK("colsum_partial");
  if (chunks == 1) return HWG_OK;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(hwg_cdiv(C, 16)), dim3(256), 0, st, (const float*)workspace, (int)chunks, C, out, accumulate);
  HWG_LAUNCH_CHECK("colsum_final");
  return HWG_OK;
}
