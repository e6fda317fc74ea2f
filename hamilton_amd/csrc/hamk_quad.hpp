// hamk_quad.hpp -- FOUR LANES PER TRAJECTORY: the kernels for 17 <= n <= 32 generalized coordinates whose coordinate
// map has a sparse Jacobian (BASELINE.json config 5, the N-link chain at N = 32), and for smaller n when the ensemble
// is too small to fill the chip with one trajectory per lane.
//
// Why this mapping.  What one right-hand side (Hamilton.hs:370-387) costs at n = 32 is the factorisation of
// K = J^T M J (n^3/3 flops) -- IF K itself is cheap.  With one trajectory per lane it is: the AD seeds are compile-time
// constants, the compiler deletes every structural zero and shares every repeated entry of J (a chain's dx_k/dq_j is
// one value for all k >= j), and with re-association allowed the sum over the 2n cartesian outputs collapses to
// "count x product" (hamk_device.hpp mass_matrix): ~3 flops per entry of K instead of 2 x 2n.  But one lane cannot
// hold K (528 doubles at n = 32), which is why round 1-2 mapped a trajectory to 32 lanes with one AD DIRECTION per
// lane (hamk_wave.hpp) -- and paid for it: lane-dependent seeds make J dense (K = 64 x 32 x 32 on the matrix cores),
// and the factorisation lives in LDS (one round trip per pivot pair; the LDS unit was the busiest resource).
// Here both are kept: EVERY lane of a quad runs the per-trajectory sweeps with compile-time seeds (4x redundant, but
// sparse: a few hundred instructions), and only the storage and factorisation of K are shared -- row a of K lives in
// lane a % 4 (register slot a / 4), 144 doubles per lane at n = 32 -- with the pivot column exchanged by DPP
// quad_perm broadcasts: no LDS in the factorisation at all, no dependent memory round trips, pure VALU.
//   per lane and right-hand side at n = 32 (one stage of the RK4 loop, counted from the code object: 6.5 k instructions):
//     sincos of the lane's 8 angles + sweep 1 + dU/dq      ~0.6 k
//     the lane's rows of K (re-associated: count x product)  ~0.9 k
//     Cholesky (2.1 k fp64 + 1.1 k DPP moves + 32 rsqrt)   ~3.5 k
//     back substitution                                    ~0.7 k
//     reverse sweep for dT/dq (redundant x 4) + the stage   ~0.9 k
//   for 16 trajectories per wavefront (~400 per trajectory) against ~3.4 k per wavefront for 2 trajectories (~1700 per
//   trajectory) of the wave-cooperative kernels.  Half of it is fp64 arithmetic; the rest is what the mapping costs: 1.2 k DPP
//   moves, 0.7 k v_accvgpr moves (K alone is 288 of the 256 architectural VGPRs), 0.3 k selects.
// LDS holds only what the four lanes share of the state: q, qd and the sincos pairs of the trajectory, [row][64
// trajectories of the block] -- 64 KiB per 256-thread block at n = 32 -- read with immediate offsets from one base.
//
// All eight kernels of the path are provided (round 3, second half: the first version had the four of the hot path and left
// the rest to the system's wave-cooperative module).
#pragma once
#include "hamk_device.hpp"

namespace hamk {
// a constant the compiler cannot recognise twice (the generated code of a dense map writes shared `constant x value` products out
// at every use: hamk_codegen.cpp g_emit_unshare_scales)
HAMK_DEV double opaque_const(double c) {
#ifndef HAMK_HOST_EMULATION
  asm volatile("" : "+s"(c));
#endif
  return c;
}
namespace quad {

template <int N> struct Geo {
  static constexpr int NR = (N + 3) / 4;      // rows of K (= coordinates of the state) per lane
  static constexpr int NP4 = 4 * NR;          // N rounded up to a multiple of four; rows >= N are identity padding
  static constexpr int TPB = 64;              // trajectories per 256-thread block
};

// [row][64]: component `a` of the block's trajectory `tl` at a * 64 + tl (a wavefront reads 16 consecutive doubles, each
// by the four lanes of a quad: conflict-free broadcasts)
template <class S> struct Lds {
  static constexpr int NP4 = Geo<S::N>::NP4;
  static constexpr int Q = 0, V = NP4 * 64, SQ = 2 * NP4 * 64, CQ = 3 * NP4 * 64, GU = 4 * NP4 * 64, TOTAL = 5 * NP4 * 64;
};
#define HAMK_QUAD_SMEM(S) __shared__ double smem[hamk::quad::Lds<S>::TOTAL]

#ifdef HAMK_HOST_EMULATION
// tests/host_emulation/wave_shim.hpp: one OS thread per lane; quad primitives through a per-quad barrier
#define HAMK_QUAD_SYNC() emu_quad_barrier()
#else
HAMK_DEV void quad_sync_dev() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_sched_barrier(0);
}
#define HAMK_QUAD_SYNC() hamk::quad::quad_sync_dev()
#endif
// The phases of an evaluation (sweep, panels of the factorisation, back substitution, reverse sweep) must not be
// interleaved by the machine scheduler: each needs most of the register file for itself, and overlapped they spill.
#ifdef HAMK_HOST_EMULATION
#define HAMK_PHASE() ((void)0)
#elif defined(HAMK_PROBE_PHASE)
// probe builds (scripts/quad_phases.py): a numbered s_setprio at every phase boundary, so the instructions of an evaluation
// can be attributed to its phases from the disassembly
#define HAMK_PHASE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(__COUNTER__ & 3); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HAMK_PHASE() __builtin_amdgcn_sched_barrier(0)
#endif

// value of lane SRC of the caller's quad, in all four lanes: two v_mov_b32 with DPP quad_perm [SRC, SRC, SRC, SRC]
template <int SRC> HAMK_DEV double qbcast(double x) {
#ifdef HAMK_HOST_EMULATION
  return emu_quad_read(x, SRC, 0);
#else
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), SRC * 0x55, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), SRC * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#endif
}
// value of lane (own ^ MASK), MASK = 1: quad_perm [1, 0, 3, 2] = 0xB1; MASK = 2: [2, 3, 0, 1] = 0x4E
template <int MASK> HAMK_DEV double qxor(double x) {
#ifdef HAMK_HOST_EMULATION
  return emu_quad_read(x, MASK, 1);
#else
  constexpr int ctrl = (MASK == 1) ? 0xB1 : 0x4E;
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#endif
}
template <int MASK> HAMK_DEV int qxor_i(int x) {
#ifdef HAMK_HOST_EMULATION
  return emu_quad_read_i(x, MASK);
#else
  return __builtin_amdgcn_mov_dpp(x, (MASK == 1) ? 0xB1 : 0x4E, 0xf, 0xf, true);
#endif
}
// the same bits in all four lanes (both levels add the same two numbers in both orders: + is commutative)
HAMK_DEV double qsum(double x) { x += qxor<1>(x); x += qxor<2>(x); return x; }
HAMK_DEV double qmax(double x) {
  double y = qxor<1>(x); x = (y > x) ? y : x;
  y = qxor<2>(x); return (y > x) ? y : x;
}
HAMK_DEV int qor(int x) { x |= qxor_i<1>(x); x |= qxor_i<2>(x); return x; }

// candidate `r` of four (r = lane & 3): what "row 4 i + r" means in code that all four lanes execute
HAMK_DEV double sel4(int r, double a, double b, double c, double d) {
  const double ab = (r & 1) ? b : a, cd = (r & 1) ? d : c;
  return (r & 2) ? cd : ab;
}

// The same choice by ARITHMETIC: m[rr] = (lane == rr) ? 1 : 0, candidate a m[0] + b m[1] + c m[2] + d m[3] -- four fp64 instructions
// instead of six v_cndmask, and one per candidate that is not a compile-time zero (the rows of K inside the diagonal block pick
// from (a, 0, 0, 0), (a, b, 0, 0), (a, b, c, 0): 1 / 2 / 3 instructions instead of 4 / 4 / 6).  Exact for finite candidates; a
// non-finite one reaches all four rows instead of one -- of a trajectory that is lost either way (ST_NONFINITE).  Written
// OUTSIDE the re-association regions: the sum is a leaf there.
struct LaneMask {
  double m[4];
  HAMK_DEV explicit LaneMask(int r) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) m[rr] = (r == rr) ? 1.0 : 0.0;
  }
};
HAMK_DEV double msel4(const LaneMask& k, double a, double b, double c, double d) { return a * k.m[0] + b * k.m[1] + c * k.m[2] + d * k.m[3]; }

// one component of the trajectory's shared state in LDS
struct LdsVec {
  const double* p;
  HAMK_DEV double operator[](int j) const { return p[j * 64]; }
  HAMK_DEV double at(int j) const { return p[j * 64]; }
};
// The generated reverse sweep (S::dT_reverse) reads q, v and the sincos pairs through reverse_vec / reverse_trig in its second
// half.  Where they are LDS rows read in place -- q always, v and the pairs in the kernels whose sites are not inputs -- that is
// a second look through a pointer the compiler cannot connect with the first (the reverse half LOADS them again instead of
// keeping them alive in accumulation registers from the forward half); where they were loaded in one burst (TrigRegsQ,
// VecRegsQ below) the copies are kept: re-loading those too is fewer instructions and was measured slower (chain32 2.82e8 vs
// 2.89e8, chain24 4.97e8 vs 5.48e8 steps/s, profiles/r04h_ab.jsonl) -- a wavefront alone on its SIMD pays the extra LDS round
// trip in full and the moves at one issue slot each.
HAMK_DEV const double* relaunder(const double* p) {
#ifndef HAMK_HOST_EMULATION
  // (the 32-bit LDS offset is what passes through the opaque statement: a laundered generic pointer would be read with flat loads)
  typedef const __attribute__((address_space(3))) double* lds_ptr;
  unsigned off = (unsigned)(unsigned long long)(lds_ptr)p;
  asm volatile("" : "+v"(off));
  return (const double*)(lds_ptr)(unsigned long long)off;
#else
  return p;
#endif
}
HAMK_DEV LdsVec reverse_vec(const LdsVec& x) { return LdsVec{relaunder(x.p)}; }
// sweep inputs with COMPILE-TIME seeds: q_j with d/dq_i = delta_ij (j is a literal after inlining)
template <int N> struct InJet1 {
  const double* q;
  HAMK_DEV Jet1<N> operator[](int j) const {
    Jet1<N> r; r.v = q[j * 64];
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (i == j) ? 1.0 : 0.0;
    return r;
  }
};
template <int N> struct InDouble {
  const double* q;
  HAMK_DEV double operator[](int j) const { return q[j * 64]; }
};
// sincos pairs of the trajectory's inputs in LDS, addressed by SITE (same member syntax as TrigCache: tc.s[k], tc.c[k])
template <class S> struct TrigSite {
  const double* base;
  HAMK_DEV double operator[](int k) const { return base[S::trig_input(k) * 64]; }
};
template <class S> struct TrigLdsQ {
  TrigSite<S> s, c;
  double* ax; double* as; double* ac;       // unused (TRIG_REUSE never touches them)
};
// ... every read an opaque copy (dense maps, the sweep that evaluates all m outputs' VALUES in one piece: with shared values the
// compiler hoists every product a term shares with another output's -- 11 x 2 n of them for the benchmark maps -- to its first use
// and parks them in scratch until the last; distinct copies leave nothing to share)
template <class S> struct TrigSiteOpaque {
  const double* base;
  HAMK_DEV double operator[](int k) const {
    double x = base[S::trig_input(k) * 64];
#ifndef HAMK_HOST_EMULATION
    asm volatile("" : "+v"(x));
#endif
    return x;
  }
};
template <class S> struct TrigLdsOpaqueQ {
  TrigSiteOpaque<S> s, c;
  double* ax; double* as; double* ac;
};
// The same rows read in ONE BURST into registers.  A wavefront alone on its SIMD pays every LDS round trip it waits for, and
// left to itself the compiler issues each ds_read a few instructions before its first use (it schedules for register
// pressure): the sweeps then stop ~50 (first sweep) + ~100 (reverse sweep) times per right-hand side for ~100 cycles -- a
// quarter of the kernel's time by PMC (SQ_WAIT_ANY).  Loading the n sincos pairs and the n velocities together, behind a
// scheduling fence, pays ONE round trip per sweep; the registers are there (K is not yet assembled / already dead).  The
// reverse sweep keeps what it needs of them for its second half (re-loading them there instead is fewer instructions and was
// measured slower: profiles/r04h_ab.jsonl).  chain32 2.66e8 -> 2.89e8, chain24 4.93e8 -> 5.48e8 steps/s.
HAMK_DEV void burst_fence() {
#ifndef HAMK_HOST_EMULATION
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <class S> struct TrigRegsQ {
  static constexpr int NT = (S::NTRIG_F > 0) ? S::NTRIG_F : 1;
  double s[NT], c[NT];
  double* ax; double* as; double* ac;       // unused (TRIG_REUSE never touches them)
  HAMK_DEV void load(const double* sbase, const double* cbase) {
    ax = as = ac = nullptr;
#pragma unroll
    for (int k = 0; k < NT; ++k) { s[k] = sbase[S::trig_input(k) * 64]; c[k] = cbase[S::trig_input(k) * 64]; }
    burst_fence();
  }
};
template <class S> HAMK_DEV const TrigRegsQ<S>& reverse_trig(const TrigRegsQ<S>& t) { return t; }      // (kept alive, not re-loaded)
template <int N> struct VecRegsQ {
  double x[N];
  HAMK_DEV void load(const double* base) {
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = base[j * 64];
    burst_fence();
  }
  HAMK_DEV double operator[](int j) const { return x[j]; }
  HAMK_DEV double at(int j) const { return x[j]; }
};
template <int N> HAMK_DEV const VecRegsQ<N>& reverse_vec(const VecRegsQ<N>& v) { return v; }
// (what velocity() does between staging the inputs and the first sweep)
template <class S, class TC> HAMK_DEV void trig_fill(TC&, const double*, const double*) {}
template <class S> HAMK_DEV void trig_fill(TrigRegsQ<S>& t, const double* sb, const double* cb) { t.load(sb, cb); }

template <class S> struct Ctx {
  static constexpr int N = S::N, NR = Geo<N>::NR, NP4 = Geo<N>::NP4;
  double* smem;      // the block's __shared__ array
  int tl;            // the quad's trajectory within the block: row a of array X at smem[X + a * 64 + tl]
  int r;             // lane within the quad
  HAMK_DEV double* q() const { return smem + Lds<S>::Q + tl; }
  HAMK_DEV double* v() const { return smem + Lds<S>::V + tl; }
  HAMK_DEV double* sq() const { return smem + Lds<S>::SQ + tl; }
  HAMK_DEV double* cq() const { return smem + Lds<S>::CQ + tl; }
  HAMK_DEV double* gu() const { return smem + Lds<S>::GU + tl; }        // dU/dq of the lane's rows waits here for the end of the evaluation
  HAMK_DEV TrigLdsQ<S> trig() const { TrigLdsQ<S> t; t.s.base = sq(); t.c.base = cq(); t.ax = t.as = t.ac = nullptr; return t; }
#ifdef HAMK_HOST_EMULATION
  HAMK_DEV Ctx launder() const { return *this; }
#else
  // the trajectory's offset re-defined opaquely per evaluation: every LDS address is smem + tl + literal, the literal
  // folds into the DS offset field -- and loop-invariant code motion cannot hoist hundreds of addresses out of the
  // stepping loop into registers (the lesson of hamk_wave.hpp's Ctx::launder)
  // (r & 3 again after the barrier: the predicates "row 4 i + r is below pivot j" fold to always / never for all but three
  // values of j - 4 i only if the compiler still knows that r < 4 -- without it every (i, j) pair keeps its own lane mask
  // in an SGPR pair, ~1000 spilled SGPRs at n = 32)
  HAMK_DEV Ctx launder() const { Ctx c = *this; asm volatile("" : "+v"(c.tl), "+v"(c.r)); c.r &= 3; c.tl &= 63; return c; }
#endif
};

// entry a of a gradient, 0 beyond N (a is a literal after unrolling; the clamp keeps the abstract machine in bounds)
template <int N> HAMK_DEV double dget(const double (&d)[N], int a) { return (a < N) ? d[(a < N) ? a : 0] : 0.0; }

// ---- the lane's rows of K = J^T M J, accumulated while the sweep runs ----------------------------------------------
// acc[i][b] = K[4 i + r][b] for b <= 4 i + 3 (the last four columns include the entries above the diagonal: K is
// symmetric, the factorisation updates them consistently and never reads them).  The sum over the outputs is free to be
// re-associated: for a chain the compiler turns it into count x (S_i A_b + S'_i B_b), S_i = the lane's pick of four
// columns of J.
template <class S> struct SinkK {
  static constexpr int N = S::N, NR = Geo<N>::NR, NP4 = Geo<N>::NP4;
  double acc[NR][NP4];
  int r;
  LaneMask lm{0};
  HAMK_DEV void init(int r_) {
    r = r_; lm = LaneMask(r_);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int b = 0; b < NP4; ++b) acc[i][b] = 0.0;
  }
  template <int K, int SEQ> HAMK_DEV void put(const Jet1<N>& x) {
#pragma clang fp reassociate(on)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const double xs = S::inertia(K) * msel4(lm, dget<N>(x.d, 4 * i), dget<N>(x.d, 4 * i + 1), dget<N>(x.d, 4 * i + 2), dget<N>(x.d, 4 * i + 3));
#pragma unroll
      for (int b = 0; b < 4 * i + 4; ++b)
        if (b < N) acc[i][b] += xs * x.d[(b < N) ? b : 0];
    }
  }
  // rows >= N (n not a multiple of four): identity, so that the factorisation of the padded matrix is that of K
  HAMK_DEV void pad() {
    if constexpr (NP4 != N) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (4 * (NR - 1) + rr >= N && r == rr) acc[NR - 1][4 * (NR - 1) + rr] = 1.0;
    }
  }
};

// (frsqrt: hamk_device.hpp)

// CHOLESKY K = G G^T of the quad's K in registers, IN PLACE, the forward substitution of one right-hand side riding along.
// On return: Kp[i][j], j < 4 i + r: G; Kp[i][4 i + r] = 1 / G_aa (a = 4 i + r); z[i] = w_a = p_a - sum_(k < a) G[a][k] y_k with
// y_k = w_k / G_kk (the owner keeps the UNSCALED w: solve_back divides twice).
// Pivot j lives in lane j % 4, slot j / 4.  LEFT-LOOKING IN PANELS OF FOUR PIVOTS -- one slot of rows, the quad's own granularity:
// when a panel's turn comes its four columns receive the updates of ALL finished columns, K[a][k] -= sum_(j < J0) G[a][j] G[k][j]
// (row k of the panel broadcast from its owner, slot jb of lane k % 4: 8 DPP moves per finished column, then four FMAs per row
// slot), and its four pivots are then eliminated one after the other inside the panel's columns (d_j and the three or fewer
// column entries below it broadcast by DPP, every lane scaling its own rows).  Every entry of K is read and written ONCE and
// the finished columns are only read: at n = 32 the lane's 144 doubles of K exceed the 256 architectural VGPRs, the rest lives
// in AGPRs, and a touch of such an entry costs two v_accvgpr moves each way (right-looking in rank-4 panels, round 3, read and
// wrote every trailing entry once per panel).
// Cholesky rather than LDL^T (round 4): with G = L sqrt(D) the update K[a][k] -= G[a][j] G[k][j] multiplies the lane's own
// entry by the broadcast one -- both the SAME scaled column, stored where it will stay -- whereas LDL^T needs the column
// twice while a panel is open (L[a][j] and d_j L[k][j]: 32 more doubles per lane at n = 32, all of them AGPR traffic),
// and a pass of selects per panel to put L in place afterwards.  1 / sqrt costs what 1 / d did.
// The code is the same for the four lanes: slot i is updated over columns up to 4 i + 3 whichever row of the slot the lane
// owns; the entries beyond the lane's diagonal are the symmetric ones and never read.
// (Measured against this order on one box and not kept, profiles/r04g_ab.jsonl, r04h_ab.jsonl, r04i_ab.jsonl: the right-looking
// order of the same in-place Cholesky, -1.7 ... -5 %; a look-ahead -- the next panel collecting the finished columns while
// this panel's pivots are eliminated -- +3 % at n = 32, -1 % at n = 24 and 17.  git history: 4c91209.
// Round 6, profiles/r06g_chol_lds_ab.jsonl: the update's broadcasts through LDS -- every lane leaves its row of the finished columns, all
// four read the panel's four rows, 2.5 LDS instructions per column instead of 8 moves, written a panel ahead and loaded two column pairs
// ahead: 9 % fewer VALU instructions and -7.5 ... +0.6 % steps/s at n = 32 (a wavefront alone on its SIMD waits for what it does not
// issue).  git history: bdf3ae2.)
template <class S>
HAMK_DEV void chol(int r, double (&Kp)[Geo<S::N>::NR][Geo<S::N>::NP4], double (&z)[Geo<S::N>::NR], int& st) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  bool ok = true;
#pragma unroll
  for (int jb = 0; jb < NR; ++jb) {
    HAMK_PHASE();
    const int J0 = 4 * jb, J1 = (4 * jb + 4 < N) ? 4 * jb + 4 : N;        // this panel's pivots [J0, J1)
#pragma unroll
    for (int j = 0; j < J0; ++j) {
      const double c0 = qbcast<0>(Kp[jb][j]), c1 = qbcast<1>(Kp[jb][j]), c2 = qbcast<2>(Kp[jb][j]), c3 = qbcast<3>(Kp[jb][j]);
#pragma unroll
      for (int i = jb; i < NR; ++i) {
        const double g = Kp[i][j];
        Kp[i][J0] = fma(-g, c0, Kp[i][J0]);
        if (J0 + 1 < J1) Kp[i][J0 + 1] = fma(-g, c1, Kp[i][J0 + 1]);
        if (J0 + 2 < J1) Kp[i][J0 + 2] = fma(-g, c2, Kp[i][J0 + 2]);
        if (J0 + 3 < J1) Kp[i][J0 + 3] = fma(-g, c3, Kp[i][J0 + 3]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = J0 + jj;
      if (j >= J1) continue;
      double d, wj;
      switch (jj) {                                      // (a literal after unrolling: one case survives)
        case 0: d = qbcast<0>(Kp[jb][j]); wj = qbcast<0>(z[jb]); break;
        case 1: d = qbcast<1>(Kp[jb][j]); wj = qbcast<1>(z[jb]); break;
        case 2: d = qbcast<2>(Kp[jb][j]); wj = qbcast<2>(z[jb]); break;
        default: d = qbcast<3>(Kp[jb][j]); wj = qbcast<3>(z[jb]); break;
      }
      ok = ok && (d > 0.0);
      const double rs = frsqrt(d);
      const double yj = wj * rs;
      // column j, scaled where it stays: the rows below the pivot; the pivot's own lane keeps 1 / G_jj; rows above keep what they hold
#pragma unroll
      for (int i = jb + 1; i < NR; ++i) Kp[i][j] *= rs;
      const double below = Kp[jb][j] * rs;
      Kp[jb][j] = (r > jj) ? below : ((r == jj) ? rs : Kp[jb][j]);
      const double lm = (r > jj) ? below : 0.0;         // the slot's multiplier: rows at or above the pivot take no update
      // inside the panel: the columns (j, J1) of every row below the pivot
#pragma unroll
      for (int k = j + 1; k < J1; ++k) {
        double c;
        switch (k & 3) {
          case 1: c = qbcast<1>(Kp[jb][j]); break;
          case 2: c = qbcast<2>(Kp[jb][j]); break;
          default: c = qbcast<3>(Kp[jb][j]); break;
        }
        Kp[jb][k] = fma(-lm, c, Kp[jb][k]);
#pragma unroll
        for (int i = jb + 1; i < NR; ++i) Kp[i][k] = fma(-Kp[i][j], c, Kp[i][k]);
      }
      z[jb] = fma(-lm, yj, z[jb]);
#pragma unroll
      for (int i = jb + 1; i < NR; ++i) z[i] = fma(-Kp[i][j], yj, z[i]);
    }
  }
  if (!ok) st |= ST_SINGULAR;                            // every inertia positive (HAMK_INSTANTIATE_QUAD asserts it): a non-positive pivot IS singular
}

// G y = w is done (z holds w, y_a = w_a / G_aa); G^T v = y here; returns the lane's v_(4 i + r).  Row-oriented: G[k][a] is in
// the lane that owns row k, so the lanes accumulate partial sums s[a] = sum over their own solved rows k > a of G[k][a] v_k and
// the four partial sums meet in a quad reduction when v_a is due.
template <class S>
HAMK_DEV void solve_back(int r, const double (&Kp)[Geo<S::N>::NR][Geo<S::N>::NP4], const double (&z)[Geo<S::N>::NR], double (&v)[Geo<S::N>::NR]) {
  constexpr int N = S::N, NR = Geo<N>::NR, NP4 = Geo<N>::NP4;
  double s[NP4];
#pragma unroll
  for (int a = 0; a < NP4; ++a) s[a] = 0.0;
#pragma unroll
  for (int i = NR - 1; i >= 0; --i) {
    HAMK_PHASE();
    double vi = 0.0;
#pragma unroll
    for (int rr = 3; rr >= 0; --rr) {
      const int a = 4 * i + rr;
      if (a >= N) continue;
      // (meaningful in lane rr, whose Kp[i][a] is 1 / G_aa: zero elsewhere -- ONE select per row; the other lanes' factors below
      // are finite entries of the same block, and the lane's own v is the sum of its one non-zero)
      const double full = fma(z[i], Kp[i][a], -qsum(s[a])) * Kp[i][a];     // (the quad reduction is executed by all four lanes)
      const double va = (r == rr) ? full : 0.0;
      vi += va;
      // row a's entries inside the diagonal block feed the rows of the same slot still to come
#pragma unroll
      for (int a2 = 4 * i; a2 < a; ++a2) s[a2] = fma(Kp[i][a2], va, s[a2]);
    }
    v[i] = vi;
#pragma unroll
    for (int a2 = 0; a2 < 4 * i; ++a2) s[a2] = fma(Kp[i][a2], vi, s[a2]);
  }
}

// How an evaluation gets its sincos pairs.  When every site of f takes an input as operand (angles) each lane evaluates
// the pairs of its own coordinates once and the quad shares them through LDS (TRIG_REUSE in every sweep); otherwise every
// lane evaluates the sites itself in the first sweep and keeps them in registers for the later ones.
template <class S, bool LUT> struct Trig {
  static constexpr bool shared = S::TRIG_ALL_INPUTS && S::NTRIG_F > 0;
  static constexpr int mode1 = shared ? TRIG_REUSE : (LUT ? TRIG_LUT : TRIG_FULL);
};

// One factorisation ships: K assembled whole by ONE sweep, Cholesky in place, left-looking in panels of four pivots (chol above).
// Its history, each step measured or counted (docs/NOTEBOOK.md): round 3 shipped LDL^T right-looking in rank-4 panels (8.4 k
// instructions per lane and right-hand side at n = 32, 160 spilled registers) after a left-looking Cholesky with K assembled
// panel by panel by eight sweeps had lost to it (no scratch, but 9.5 k instructions: profiles/r03_quad_ab.jsonl, the instruction
// count decides); round 4 kept the one sweep and took the rest of that variant -- Cholesky, so that an open panel exists once
// (not as L and as d L), and the left-looking order, so that finished columns are only read: 6.5 k instructions, no scratch in
// the stepping loop.

// q of the quad's trajectory to LDS and, when every sincos site of f takes an input as operand, the pairs of the lane's
// own coordinates with it (each lane evaluates its n/4 angles once; all sweeps of the evaluation read them).
template <class S, bool LUT>
HAMK_DEV void stage_inputs(const Ctx<S>& c, const double (&qi)[Geo<S::N>::NR]) {
  constexpr int NR = Geo<S::N>::NR;
  const int r = c.r;
  HAMK_QUAD_SYNC();                                       // readers of the previous evaluation are done
#pragma unroll
  for (int i = 0; i < NR; ++i) c.q()[(4 * i + r) * 64] = qi[i];
  if constexpr (Trig<S, LUT>::shared) {
    bool far = false;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      double sv, cv;
      if constexpr (LUT) sincos_lut_fast(qi[i], sv, cv, LutLiterals()); else sincos_f64_fast(qi[i], sv, cv);
      c.sq()[(4 * i + r) * 64] = sv; c.cq()[(4 * i + r) * 64] = cv;
      far = far || !(fabs(qi[i]) < 1.6e6);
    }
#ifndef HAMK_PROBE_NO_SLOWPATH
    // huge, NaN, Inf: the library path -- ONE copy for the lane's angles (a rolled loop over what is already in LDS;
    // one inlined copy per angle is ~2000 instructions and a dozen SGPR pairs of exec masks each)
    if (far) {
#pragma unroll 1
      for (int i = 0; i < NR; ++i) {
        const double x = c.q()[(4 * i + r) * 64];
        if (!(fabs(x) < 1.6e6)) { c.sq()[(4 * i + r) * 64] = ::sin(x); c.cq()[(4 * i + r) * 64] = ::cos(x); }
      }
    }
#endif
  }
  HAMK_QUAD_SYNC();
}

// ---- DENSE coordinate maps on this mapping (S::QUAD_DENSE, round 6) ---------------------------------------------------------
// A map whose Jacobian has ~m n distinct entries (x = 2 q + A sin q + B cos q) used to leave this mapping for the wave-cooperative
// kernels, where every lane evaluates the WHOLE tape at one-direction jets with a run-time seed: dense32 63.9 k VALU instructions
// per wavefront-step for TWO trajectories.  Compile-time seeds delete every structural zero of such a map too (a term that depends
// on one input has one derivative), so a lane here evaluates J at ~2 FMAs per entry -- what stops the one-sweep SinkK above is
// REGISTERS: its 144 accumulators (n = 32) are updated by every output, and beyond 256 VGPRs each update is a round trip through
// AGPRs (first build, forced: 8 759 v_accvgpr moves around 5 640 FMAs, and the whole Jacobian parked in scratch ahead of them).
// Here K is accumulated in TILES -- a group of row slots x a block of 16 columns -- whose accumulators fit the VGPRs NEXT TO the
// entries of J the tile needs and their sincos pairs (a lane has 128 doubles of VGPRs; n = 32: rows 0..15 x columns 0..15, rows 16..23
// and 24..31 x columns 0..15, rows 16..31 x columns 16..31 -- 40 / 32 / 32 / 40 accumulators with 16 / 24 / 24 / 16 entries of J per
// output).  Every tile is a sweep of its own over the outputs: the generated code computes a whole row of J per output, the tile's
// sink reads only its columns and its rows' entries, and everything else of the row is dead code -- 80 entries of J per output over
// the four tiles instead of 4 x 32.  Each sweep runs behind a freshly laundered context (so that the compiler cannot keep one tile's
// entries for the next), and every output ends with its accumulators passing through an opaque statement and a scheduling fence
// (a fence alone orders only what has side effects: instruction selection emitted all m fences first and the arithmetic after them
// in one block).  No re-association licence: the sum over the outputs must stay the chain of FMAs it is written as.
// dU/dq of a potential over the cartesian coordinates is J^T (dU/dx), tiled the same way (SinkG: the chain rule by hand -- handing
// the jets of all m outputs to the potential keeps m x n derivatives alive across the whole sweep).
template <class T> HAMK_DEV void opaque_copy(T& x) {
#ifndef HAMK_HOST_EMULATION
  asm volatile("" : "+v"(x));
#endif
}
// accumulators acc[i - I0][b - B0] = K[4 i + r][b] for slots [I0, I1) and columns [B0, min(B1, 4 i + 4))
template <class S, int I0, int I1, int B0, int B1> struct SinkKTile {
  static constexpr int N = S::N;
  double acc[I1 - I0][B1 - B0];
  LaneMask lm{0};
  HAMK_DEV void init(int r_) {
    lm = LaneMask(r_);
#pragma unroll
    for (int i = 0; i < I1 - I0; ++i)
#pragma unroll
      for (int b = 0; b < B1 - B0; ++b) acc[i][b] = 0.0;
  }
  static constexpr bool has(int i, int b) { return b >= B0 && b < B1 && b < 4 * i + 4 && b < N; }
  template <int K, int SEQ> HAMK_DEV void put(const Jet1<N>& x) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const double xs = S::inertia(K) * msel4(lm, dget<N>(x.d, 4 * i), dget<N>(x.d, 4 * i + 1), dget<N>(x.d, 4 * i + 2), dget<N>(x.d, 4 * i + 3));
#pragma unroll
      for (int b = B0; b < B1; ++b)
        if (has(i, b)) acc[i - I0][b - B0] = fma(xs, x.d[(b < N) ? b : 0], acc[i - I0][b - B0]);
    }
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
      for (int b = B0; b < B1; ++b)
        if (has(i, b)) opaque_copy(acc[i - I0][b - B0]);
    HAMK_PHASE();
  }
  template <int NR, int NP4> HAMK_DEV void store(double (&Kp)[NR][NP4]) const {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
      for (int b = B0; b < B1; ++b)
        if (b < NP4 && b < 4 * i + 4) Kp[i][b] = (b < N) ? acc[i - I0][b - B0] : 0.0;
  }
};
// the lane's rows [I0, I1) of J^T w, w = dU/dx (a potential over the cartesian coordinates)
template <class S, int I0, int I1> struct SinkG {
  static constexpr int N = S::N;
  double g[I1 - I0];
  const double* w;                                          // dU/dx_k of the trajectory in LDS (stride 64): k < NP4 in the rows of the velocities,
  const double* w2;                                         // the others in the rows of dU/dq (both free while K is assembled)
  LaneMask lm{0};
  template <int K, int SEQ> HAMK_DEV void put(const Jet1<N>& x) {
    const double wk = (K < Geo<N>::NP4) ? w[(K < Geo<N>::NP4 ? K : 0) * 64] : w2[(K >= Geo<N>::NP4 ? K - Geo<N>::NP4 : 0) * 64];
#pragma unroll
    for (int i = I0; i < I1; ++i)
      g[i - I0] = fma(wk, msel4(lm, dget<N>(x.d, 4 * i), dget<N>(x.d, 4 * i + 1), dget<N>(x.d, 4 * i + 2), dget<N>(x.d, 4 * i + 3)), g[i - I0]);
#pragma unroll
    for (int i = I0; i < I1; ++i) opaque_copy(g[i - I0]);
    HAMK_PHASE();
  }
};
template <int NS> HAMK_DEV void launder_trig(TrigCache<NS>& t) {
  if constexpr (NS > 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) { opaque_copy(t.s[k]); opaque_copy(t.c[k]); }
  }
}
// one pass of the first-order sweep with `sink`, behind a freshly laundered context
template <class S, class TC, class Sink>
HAMK_DEV void dense_sweep(const Ctx<S>& c0, const TC& tc, Sink& sink) {
  constexpr int N = S::N;
  const Ctx<S> c = c0.launder();
  InJet1<N> in{c.q()};
  if constexpr (S::TRIG_ALL_INPUTS && S::NTRIG_F > 0) {
    TrigLdsQ<S> tl = c.trig();
    S::template coords_sink<Jet1<N>, TRIG_REUSE>(in, tl, sink);
  } else {
    TC t2 = tc;
    launder_trig(t2);
    S::template coords_sink<Jet1<N>, TRIG_REUSE>(in, t2, sink);
  }
  HAMK_PHASE();
}
// one tile of K: a sweep with SinkKTile, stored into Kp
template <class S, int I0, int I1, int B0, int B1, class TC>
HAMK_DEV void dense_tile(const Ctx<S>& c0, const TC& tc, double (&Kp)[Geo<S::N>::NR][Geo<S::N>::NP4]) {
  if constexpr (I0 < I1 && B0 < B1) {
    SinkKTile<S, I0, I1, B0, B1> sink;
    sink.init(c0.r);
    dense_sweep<S>(c0, tc, sink);
    sink.store(Kp);
  }
}
// the slots from H0 on, two at a time, against the first sixteen columns
template <class S, int H0, class TC>
HAMK_DEV void dense_tiles_low(const Ctx<S>& c0, const TC& tc, double (&Kp)[Geo<S::N>::NR][Geo<S::N>::NP4]) {
  constexpr int NR = Geo<S::N>::NR;
  if constexpr (H0 < NR) {
    constexpr int H1 = (H0 + 2 < NR) ? H0 + 2 : NR;
    dense_tile<S, H0, H1, 0, 16>(c0, tc, Kp);
    dense_tiles_low<S, H1>(c0, tc, Kp);
  }
}
template <class S, int I0, int I1, class TC>
HAMK_DEV void dense_grad_rows(const Ctx<S>& c0, const TC& tc, double (&gUi)[Geo<S::N>::NR]) {
  if constexpr (I0 < I1) {
    SinkG<S, I0, I1> sg;
    sg.w = c0.v(); sg.w2 = c0.gu();
    sg.lm = LaneMask(c0.r);
#pragma unroll
    for (int i = 0; i < I1 - I0; ++i) sg.g[i] = 0.0;
    dense_sweep<S>(c0, tc, sg);
#pragma unroll
    for (int i = I0; i < I1; ++i) gUi[i] = sg.g[i - I0];
  }
}
// K, U and the lane's rows of dU/dq for a dense map; tc: the sincos pairs of f's sites when they are not inputs (filled here)
template <class S, bool LUT, class TC>
HAMK_DEV void assemble_dense(const Ctx<S>& c, double (&Kp)[Geo<S::N>::NR][Geo<S::N>::NP4], double (&gUi)[Geo<S::N>::NR], double& U, TC& tc) {
  constexpr int N = S::N, M = S::M, NR = Geo<N>::NR, NP4 = Geo<N>::NP4;
  constexpr bool shared = S::TRIG_ALL_INPUTS && S::NTRIG_F > 0;
  constexpr int LO = NR < 4 ? NR : 4;                      // slots whose rows lie in the first sixteen
  static_assert(!S::U_CART || M <= 2 * NP4, "dU/dx waits in the LDS rows of the velocities and of dU/dq");
  const int r = c.r;
  TrigCache<S::NTRIG_U> tu;
  if constexpr (S::U_CART) {
    // values of the outputs (this sweep also fills tc where the sites are not inputs), dU/dx by first-order jets over x; every lane
    // computes all of it, lane 0 leaves dU/dx in LDS for the quad's sweeps below
    {
      double qv[N], xv[M];
#pragma unroll
      for (int j = 0; j < N; ++j) qv[j] = c.q()[j * 64];
      if constexpr (shared) {
        TrigLdsOpaqueQ<S> tl; tl.s.base = c.sq(); tl.c.base = c.cq(); tl.ax = tl.as = tl.ac = nullptr;
        S::template coords<double, TRIG_REUSE>(qv, xv, tl);
      } else S::template coords<double, LUT ? TRIG_LUT : TRIG_FULL>(qv, xv, tc);
      Jet1<M> xj[M];
#pragma unroll
      for (int k = 0; k < M; ++k) {
        xj[k].v = xv[k];
#pragma unroll
        for (int i = 0; i < M; ++i) xj[k].d[i] = (i == k) ? 1.0 : 0.0;
      }
      const Jet1<M> ux = S::template potential<Jet1<M>, TRIG_FULL>(xj, tu);
      U = ux.v;
      // (M <= NP4 rows in V; the rest, if any, in GU -- which receives dU/dq only after the sweeps that read dU/dx)
#pragma unroll
      for (int k = 0; k < M; ++k)
        if (r == (k & 3)) { if (k < NP4) c.v()[(k < NP4 ? k : 0) * 64] = ux.d[k]; else c.gu()[(k >= NP4 ? k - NP4 : 0) * 64] = ux.d[k]; }
    }
    HAMK_QUAD_SYNC();
    dense_grad_rows<S, 0, LO>(c, tc, gUi);
    dense_grad_rows<S, LO, NR>(c, tc, gUi);
  } else {
    double qv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) qv[j] = c.q()[j * 64];
    if constexpr (!shared && S::NTRIG_F > 0) { double xv[M]; S::template coords<double, LUT ? TRIG_LUT : TRIG_FULL>(qv, xv, tc); }      // (fills tc; the values are dead code)
    Jet1<N> qj[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      qj[j].v = qv[j];
#pragma unroll
      for (int i = 0; i < N; ++i) qj[j].d[i] = (i == j) ? 1.0 : 0.0;
    }
    const Jet1<N> u = S::template potential<Jet1<N>, TRIG_FULL>(qj, tu);
    U = u.v;
#pragma unroll
    for (int i = 0; i < NR; ++i) gUi[i] = sel4(r, dget<N>(u.d, 4 * i), dget<N>(u.d, 4 * i + 1), dget<N>(u.d, 4 * i + 2), dget<N>(u.d, 4 * i + 3));
  }
  HAMK_PHASE();
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int b = 0; b < NP4; ++b) Kp[i][b] = 0.0;
  dense_tile<S, 0, LO, 0, 16>(c, tc, Kp);
  dense_tiles_low<S, LO>(c, tc, Kp);
  dense_tile<S, LO, NR, 16, NP4>(c, tc, Kp);
  if constexpr (NP4 != N) {                                // rows >= N: identity (SinkK::pad)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (4 * (NR - 1) + rr >= N && r == rr) Kp[NR - 1][4 * (NR - 1) + rr] = 1.0;
  }
}

// Shared first half of every evaluation: q to LDS, sincos pairs, K and its factorisation with the forward substitution
// of p, back substitution.  Returns the lane's velocities; gU (own rows), U.
template <class S, bool LUT, class TC>
HAMK_DEV void velocity(const Ctx<S>& c, const double (&qi)[Geo<S::N>::NR], const double (&pi)[Geo<S::N>::NR],
                       double (&vi)[Geo<S::N>::NR], double (&gUi)[Geo<S::N>::NR], double& U, int& st, TC& tc) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  const int r = c.r;
  stage_inputs<S, LUT>(c, qi);
  if constexpr (S::QUAD_DENSE) {
    double Kp[NR][Geo<N>::NP4];
    assemble_dense<S, LUT>(c, Kp, gUi, U, tc);
    double z[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { z[i] = pi[i]; c.gu()[(4 * i + r) * 64] = gUi[i]; gUi[i] = 0.0; }
    HAMK_PHASE();
    chol<S>(r, Kp, z, st);
    HAMK_PHASE();
    solve_back<S>(r, Kp, z, vi);
    HAMK_PHASE();
  } else {
  trig_fill<S>(tc, c.sq(), c.cq());
  SinkK<S> sink;
  sink.init(r);
  InJet1<N> in{c.q()};
  TrigCache<S::NTRIG_U> tu;
  const Jet1<N> u = S::template coords_sink_u<Jet1<N>, Trig<S, LUT>::mode1>(in, tc, tu, sink);
  sink.pad();
  HAMK_PHASE();
  U = u.v;
#pragma unroll
  for (int i = 0; i < NR; ++i) gUi[i] = sel4(r, dget<N>(u.d, 4 * i), dget<N>(u.d, 4 * i + 1), dget<N>(u.d, 4 * i + 2), dget<N>(u.d, 4 * i + 3));
  double z[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { z[i] = pi[i]; c.gu()[(4 * i + r) * 64] = gUi[i]; gUi[i] = 0.0; }      // (dU/dq waits in LDS: Ctx::gu)
  HAMK_PHASE();
  chol<S>(r, sink.acc, z, st);
  HAMK_PHASE();
  solve_back<S>(r, sink.acc, z, vi);
  HAMK_PHASE();
  }
}

// hamEqs for the quad's trajectory: the lane returns (dq, dp) of its coordinates.         Hamilton.hs:370-387
template <class S, bool LUT>
HAMK_DEV void ham_eqs(const Ctx<S>& c0, const double (&qi)[Geo<S::N>::NR], const double (&pi)[Geo<S::N>::NR],
                      double (&dqi)[Geo<S::N>::NR], double (&dpi)[Geo<S::N>::NR], int& st) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  const Ctx<S> c = c0.launder();
  const int r = c.r;
  double vi[NR], gUi[NR], U;
  double dT[N];
  // dT/dq = -d/dq [sum_k m_k (J qd)_k (D_qd x_k)] with (J qd)_k held fixed: the generated reverse sweep, per trajectory,
  // every lane of the quad (compile-time sparsity; the cooperative alternative is dense in every direction)
  if constexpr (Trig<S, LUT>::shared) {
    {
      TrigRegsQ<S> tr;                                    // (filled by velocity() once the inputs are staged)
      velocity<S, LUT>(c, qi, pi, vi, gUi, U, st, tr);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) c.v()[(4 * i + r) * 64] = vi[i];
    HAMK_QUAD_SYNC();
    const Ctx<S> c2 = c.launder();
    LdsVec q{c2.q()};
    VecRegsQ<N> v; v.load(c2.v());
    TrigRegsQ<S> t2; t2.load(c2.sq(), c2.cq());
    S::dT_reverse(q, v, t2, dT);
  } else {
    TrigCache<S::NTRIG_F> tc;
    velocity<S, LUT>(c, qi, pi, vi, gUi, U, st, tc);
#pragma unroll
    for (int i = 0; i < NR; ++i) c.v()[(4 * i + r) * 64] = vi[i];
    HAMK_QUAD_SYNC();
    const Ctx<S> c2 = c.launder();
    LdsVec q{c2.q()}, v{c2.v()};
    S::dT_reverse(q, v, tc, dT);
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    dqi[i] = vi[i];
    const double gu = c.gu()[(4 * i + r) * 64];           // written by this lane during the sweep
    dpi[i] = -(sel4(r, dget<N>(dT, 4 * i), dget<N>(dT, 4 * i + 1), dget<N>(dT, 4 * i + 2), dget<N>(dT, 4 * i + 3)) + gu);
  }
}

// velocities alone (fromPhase, keP, hamiltonian)
template <class S, bool LUT>
HAMK_DEV void velocity_only(const Ctx<S>& c0, const double (&qi)[Geo<S::N>::NR], const double (&pi)[Geo<S::N>::NR],
                            double (&vi)[Geo<S::N>::NR], double& U, int& st) {
  constexpr int NR = Geo<S::N>::NR;
  const Ctx<S> c = c0.launder();
  double gUi[NR];
  if constexpr (Trig<S, LUT>::shared) { TrigRegsQ<S> tr; velocity<S, LUT>(c, qi, pi, vi, gUi, U, st, tr); }
  else { TrigCache<S::NTRIG_F> tc; velocity<S, LUT>(c, qi, pi, vi, gUi, U, st, tc); }
}

// ---- kernels --------------------------------------------------------------------------------------------------------
template <class S> struct Where {
  static constexpr int N = S::N, NR = Geo<N>::NR;
  i64 t;            // trajectory index (clamped to B-1 for the tail)
  bool real;        // the quad's trajectory exists
  Ctx<S> c;
  HAMK_DEV Where(double* smem, i64 B) {
    c.tl = threadIdx.x >> 2;
    c.r = threadIdx.x & 3;
    const i64 tt = (i64)blockIdx.x * Geo<N>::TPB + c.tl;
    real = tt < B;
    t = real ? tt : B - 1;
    c.smem = smem;
  }
  HAMK_DEV bool owns(int i) const { return 4 * i + c.r < N; }
  HAMK_DEV void load(const double* a, i64 B, double (&x)[NR]) const {
#pragma unroll
    for (int i = 0; i < NR; ++i) { const int j = owns(i) ? 4 * i + c.r : 0; const double y = a[(i64)j * B + t]; x[i] = owns(i) ? y : 0.0; }
  }
  HAMK_DEV void store(double* a, i64 B, const double (&x)[NR]) const {
#pragma unroll
    for (int i = 0; i < NR; ++i) if (real && owns(i)) a[(i64)(4 * i + c.r) * B + t] = x[i];
  }
};

template <class S, bool LUT> HAMK_DEV double energy(const Ctx<S>& c0, const double (&qi)[Geo<S::N>::NR], const double (&pi)[Geo<S::N>::NR], int& st) {
  constexpr int NR = Geo<S::N>::NR;
  double vi[NR], U;
  velocity_only<S, LUT>(c0, qi, pi, vi, U, st);
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NR; ++i) t = fma(vi[i], pi[i], t);
  return fma(0.5, qsum(t), U);
}

template <class S>
HAMK_DEV void rk4_body(double* smem, double* park, double* q, double* p, i64 B, double dt, int nsteps, double drift_tol, int* status) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  constexpr bool LUT = StageTrig<S>::lut;                 // the stepping kernel loads the sincos table (hamk_device.hpp)
  if constexpr (LUT) lut_load();
  Where<S> w(smem, B);
  double yq[NR], yp[NR];
  w.load(q, B, yq); w.load(p, B, yp);
  int st = 0;
  double H0 = 0.0;
  if (drift_tol > 0.0) H0 = energy<S, LUT>(w.c, yq, yp, st);
  const double h2 = 0.5 * dt, h6 = dt * (1.0 / 6.0), h3 = dt * (1.0 / 3.0);
  // The step's base point y and the running combination wait in LDS while a right-hand side runs ([component][lane],
  // conflict-free; 4 x NR x 2 KiB per block): the lane's quarter of K alone is more than the 256 architectural VGPRs at
  // n = 32, and what the compiler cannot keep it sends to scratch -- through the vector memory pipe, with one wavefront
  // per SIMD to hide it.
  double* py = park + threadIdx.x;                          // yq[i] at py[i * 256], yp[i] at py[(NR + i) * 256]
  double* pa = park + 2 * NR * 256 + threadIdx.x;           // the same for the combination
#pragma unroll
  for (int i = 0; i < NR; ++i) { py[i * 256] = yq[i]; py[(NR + i) * 256] = yp[i]; pa[i * 256] = yq[i]; pa[(NR + i) * 256] = yp[i]; }
  double kq[NR], kp[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { kq[i] = 0.0; kp[i] = 0.0; }
#pragma unroll 1
  for (int it = 0; it < 4 * nsteps; ++it) {
    const int sg = it & 3;
    const double a = (sg == 0) ? 0.0 : ((sg == 3) ? dt : h2);
    const double b = (sg == 0 || sg == 3) ? h6 : h3;
    double tq[NR], tp[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { tq[i] = fma(a, kq[i], py[i * 256]); tp[i] = fma(a, kp[i], py[(NR + i) * 256]); }
#ifndef HAMK_HOST_EMULATION
    __builtin_amdgcn_sched_barrier(0);
#endif
    ham_eqs<S, LUT>(w.c, tq, tp, kq, kp, st);
#ifndef HAMK_HOST_EMULATION
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (sg == 3) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const double vq = fma(b, kq[i], pa[i * 256]), vp = fma(b, kp[i], pa[(NR + i) * 256]);
        pa[i * 256] = vq; py[i * 256] = vq; pa[(NR + i) * 256] = vp; py[(NR + i) * 256] = vp;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) { pa[i * 256] = fma(b, kq[i], pa[i * 256]); pa[(NR + i) * 256] = fma(b, kp[i], pa[(NR + i) * 256]); }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) { yq[i] = py[i * 256]; yp[i] = py[(NR + i) * 256]; }
  if (drift_tol > 0.0) {
    int st1 = 0;
    const double H1 = energy<S, LUT>(w.c, yq, yp, st1);
    const double lim = drift_tol * fmax(1.0, fabs(H0));
    if (!(fabs(H1 - H0) <= lim) || is_nonfinite_bits(H1)) st |= ST_DRIFT;
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NR; ++i) bad = bad || (w.owns(i) && (is_nonfinite_bits(yq[i]) || is_nonfinite_bits(yp[i])));
  if (bad) st |= ST_NONFINITE;
  const int stq = qor(st);
  w.store(q, B, yq); w.store(p, B, yp);
  if (status && w.real && w.c.r == 0) status[w.t] = stq;
}

template <class S>
HAMK_DEV void hameqs_body(double* smem, const double* q, const double* p, double* dq, double* dp, i64 B, int* status) {
  constexpr int NR = Geo<S::N>::NR;
  Where<S> w(smem, B);
  double qi[NR], pi[NR], a[NR], b[NR];
  w.load(q, B, qi); w.load(p, B, pi);
  int st = 0;
  ham_eqs<S, false>(w.c, qi, pi, a, b, st);
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NR; ++i) bad = bad || (w.owns(i) && (is_nonfinite_bits(a[i]) || is_nonfinite_bits(b[i])));
  if (bad) st |= ST_NONFINITE;
  const int stq = qor(st);
  w.store(dq, B, a); w.store(dp, B, b);
  if (status && w.real && w.c.r == 0) status[w.t] = stq;
}

template <class S>
HAMK_DEV void from_phase_body(double* smem, const double* q, const double* p, double* qd, i64 B, int* status) {
  constexpr int NR = Geo<S::N>::NR;
  Where<S> w(smem, B);
  double qi[NR], pi[NR], vi[NR], U;
  w.load(q, B, qi); w.load(p, B, pi);
  int st = 0;
  velocity_only<S, false>(w.c, qi, pi, vi, U, st);
  const int stq = qor(st);
  w.store(qd, B, vi);
  if (status && w.real && w.c.r == 0) status[w.t] = stq;
}

template <class S>
HAMK_DEV void observe_body(double* smem, const double* q, const double* p, double* ke, double* pe, double* h, i64 B, int* status) {
  constexpr int NR = Geo<S::N>::NR;
  Where<S> w(smem, B);
  double qi[NR], pi[NR], vi[NR], U;
  w.load(q, B, qi);
  if (p) w.load(p, B, pi);
  else {
#pragma unroll
    for (int i = 0; i < NR; ++i) pi[i] = 0.0;
  }
  int st = 0;
  velocity_only<S, false>(w.c, qi, pi, vi, U, st);
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NR; ++i) t = fma(vi[i], pi[i], t);
  t = 0.5 * qsum(t);
  const int stq = qor(st);
  if (w.real && w.c.r == 0) {
    if (ke) ke[w.t] = t;
    if (pe) pe[w.t] = U;
    if (h) h[w.t] = t + U;
    if (status) status[w.t] = (ke || h) ? stq : 0;
  }
}


// ---- the entry points around the hot path -----------------------------------------------------------------------------
// momenta (Hamilton.hs:262-269): p_a = sum_k J[k][a] m_k (J qd)_k for the lane's rows, one sparse sweep; U rides along.
template <class S> struct SinkP {
  static constexpr int N = S::N, NR = Geo<N>::NR;
  double p[NR];
  const double* v;      // qd of the trajectory in LDS
  int r;
  template <int K, int SEQ> HAMK_DEV void put(const Jet1<N>& x) {
    double w = 0.0;
#pragma unroll
    for (int b = 0; b < N; ++b) w = fma(x.d[b], v[b * 64], w);
    w *= S::inertia(K);
#pragma unroll
    for (int i = 0; i < NR; ++i)
      p[i] = fma(sel4(r, dget<N>(x.d, 4 * i), dget<N>(x.d, 4 * i + 1), dget<N>(x.d, 4 * i + 2), dget<N>(x.d, 4 * i + 3)), w, p[i]);
  }
};
template <class S>
HAMK_DEV void momentum(const Ctx<S>& c0, const double (&qi)[Geo<S::N>::NR], const double (&vi)[Geo<S::N>::NR],
                       double (&pi)[Geo<S::N>::NR], double& U) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  const Ctx<S> c = c0.launder();
  const int r = c.r;
  stage_inputs<S, false>(c, qi);
#pragma unroll
  for (int i = 0; i < NR; ++i) c.v()[(4 * i + r) * 64] = vi[i];
  HAMK_QUAD_SYNC();
  SinkP<S> sink;
  sink.v = c.v(); sink.r = r;
#pragma unroll
  for (int i = 0; i < NR; ++i) sink.p[i] = 0.0;
  InJet1<N> in{c.q()};
  TrigCache<S::NTRIG_U> tu;
  Jet1<N> u;
  if constexpr (Trig<S, false>::shared) { TrigLdsQ<S> tl = c.trig(); u = S::template coords_sink_u<Jet1<N>, TRIG_REUSE>(in, tl, tu, sink); }
  else { TrigCache<S::NTRIG_F> tc; u = S::template coords_sink_u<Jet1<N>, TRIG_FULL>(in, tc, tu, sink); }
  U = u.v;
#pragma unroll
  for (int i = 0; i < NR; ++i) pi[i] = sink.p[i];
}

template <class S>
HAMK_DEV void to_phase_body(double* smem, const double* q, const double* qd, double* p, i64 B) {
  constexpr int NR = Geo<S::N>::NR;
  Where<S> w(smem, B);
  double qi[NR], vi[NR], pi[NR], U;
  w.load(q, B, qi); w.load(qd, B, vi);
  momentum<S>(w.c, qi, vi, pi, U);
  w.store(p, B, pi);
}

// keC / lagrangian (Hamilton.hs:288-309)
template <class S>
HAMK_DEV void observe_config_body(double* smem, const double* q, const double* qd, double* ke, double* lag, i64 B) {
  constexpr int NR = Geo<S::N>::NR;
  Where<S> w(smem, B);
  double qi[NR], vi[NR], pi[NR], U;
  w.load(q, B, qi); w.load(qd, B, vi);
  momentum<S>(w.c, qi, vi, pi, U);
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NR; ++i) t = fma(vi[i], pi[i], t);
  t = 0.5 * qsum(t);
  if (w.real && w.c.r == 0) {
    if (ke) ke[w.t] = t;
    if (lag) lag[w.t] = t - U;
  }
}

// underlyingPos (Hamilton.hs:174-178): every lane evaluates f, lane r writes the outputs k = r mod 4
template <class S> struct SinkX {
  double* x; i64 B, t; int r; bool real;
  template <int K, int SEQ> HAMK_DEV void put(double v) const { if (real && (K & 3) == r) x[(i64)K * B + t] = v; }
};
template <class S> HAMK_DEV void coords_body(double* smem, const double* q, double* x, i64 B) {
  constexpr int N = S::N, NR = Geo<N>::NR;
  Where<S> w(smem, B);
  double qi[NR];
  w.load(q, B, qi);
  const Ctx<S> c = w.c.launder();
  stage_inputs<S, false>(c, qi);
  SinkX<S> sink{x, B, w.t, c.r, w.real};
  InDouble<N> in{c.q()};
  if constexpr (Trig<S, false>::shared) { TrigLdsQ<S> tl = c.trig(); S::template coords_sink<double, TRIG_REUSE>(in, tl, sink); }
  else { TrigCache<S::NTRIG_F> tc; S::template coords_sink<double, TRIG_FULL>(in, tc, sink); }
}

// evolveHam / stepHam: the GSL semantics of hamk::rkf45_body (rkf45.c, cstd.c with a_y = a_dydt = 1, evolve.c, gsl-ode.c
// under either binding), one trajectory per quad.  t, h and the accept / reject decision are uniform within a quad (the
// error norm is a quad-wide max by DPP) but differ between the quads of a wavefront: every lane always executes the
// attempt and a quad that has reached its output time does not commit (the structure of hamk::wave::rkf45_body).
#ifndef HAMK_QUAD_RKF_PARK
#define HAMK_QUAD_RKF_PARK 0       /* generated: hamk_options::rkf_park */
#endif

// The body for the larger systems (HAMK_QUAD_RKF_PARK; the library's default from n = 17).  The stepper's nine vectors (y,
// dydt, k2..k6, the trial state and its derivative: 18 NR doubles per lane, 288 registers at n = 32) only WAIT while a
// right-hand side runs, and one right-hand side alone wants the whole register file; left to the register allocator they
// compete with K and the spill code lands inside the factorisation (chain32: 797 spilled registers).  As in
// hamk::rkf45_body_parked: y and dydt (and k2.. where the CU's LDS allows) wait in LDS next to the quad's exchange
// arrays, the other k's, the trial state and the error combination in one private array that is written at a run-time
// row (the stage counter: it stays in scratch memory), and every right-hand side's result is used from the registers
// by the stage -- or the error norm and the commit -- that follows it.  Same expressions (the compiler contracts them
// into FMAs on its own terms: the two bodies agree to roundoff, with identical sub-step counts).  Measured on MI355X,
// stepHam calls/s at B = 16 384, registers / parked (profiles/r03_quad_rkf_park.jsonl): chain32 4.14e6 / 6.63e6 (797 ->
// 74 spilled registers), chain24 1.40e7 / 1.74e7, chain20 2.70e7 / 2.94e7, chain17 3.34e7 / 3.48e7.
template <class S> struct RkfPark {
  static constexpr int D = 2 * Geo<S::N>::NR;
  static constexpr int BUDGET = (160 * 1024 - 8 * 1024 - 2 * 1024 - Lds<S>::TOTAL * 8) / 2048;       // doubles per lane left in the CU's LDS
  static constexpr int NL = (BUDGET / D) > 5 ? 5 : (BUDGET / D);                                        // y, dydt, then k2, k3, k4
  static_assert(NL >= 2, "the quad exchange arrays leave room for y and dydt up to n = 32");
};
template <class S>
HAMK_DEV void rkf45_body_parked(double* smem, const double* q0, const double* p0, double* qout, double* pout, i64 B, int nt,
                                const double* ts, double ts0, double ts1, double h0, double eps_abs, double eps_rel,
                                int flags, int max_sub, int* status, int* nsub, int ncalls, int it_every) {
  constexpr int N = S::N, NR = Geo<N>::NR, D = 2 * NR, NL = RkfPark<S>::NL;
  constexpr bool LUT = StageTrig<S>::lut;
  if constexpr (LUT) lut_load();
  const int row0 = flags & 1, inplace = (flags >> 8) & 3, gsl_api = (flags >> 16) & 3;
  const bool api2 = gsl_api != 1;
  const double sgn = (!api2 || h0 > 0.0) ? 1.0 : -1.0;
  bool failed = false;
  Where<S> w(smem, B);
  __shared__ double rows[NL * D * 256];
  // (one base pointer per LDS row, each "array + constant + lane": see hamk::rkf45_body_parked)
#define HAMK_RKF_LROW(r) (rows + (r) * D * 256 + threadIdx.x)
#define HAMK_RKF_K(KR, j) ((2 + (KR) < NL) ? HAMK_RKF_LROW(2 + (KR))[(j) * 256] : v[KR][j])      /* k_{2 + KR}[j] */
  double* py = rows + threadIdx.x;                         // y[j]    at py[j * 256]: [q of the lane's rows; p of the lane's rows]
  double* pf = rows + D * 256 + threadIdx.x;               // dydt[j] at pf[j * 256]
  // rows 0..3: k2..k5 (those that are not in LDS); the trial state and the error combination take over the rows of k2 and
  // k3 (see hamk::rkf45_body_parked)
  double v[4][D];
#define HAMK_RKF_YN(j) HAMK_RKF_K(0, j)
#define HAMK_RKF_E(j) HAMK_RKF_K(1, j)
  auto put_k = [&](int kr, const double (&x)[D]) {         // k_{2 + kr}; kr: a run-time value (the stage counter)
    if (NL > 2 && 2 + kr < NL) {
#pragma unroll
      for (int j = 0; j < D; ++j) HAMK_RKF_LROW(2 + kr)[j * 256] = x[j];
    } else {
#pragma unroll
      for (int j = 0; j < D; ++j) v[kr][j] = x[j];
    }
  };
  auto rhs = [&](const double (&yy)[D], double (&dy)[D], int& st_) {
    double a[NR], b[NR], da[NR], db[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { a[i] = yy[i]; b[i] = yy[NR + i]; }
    __builtin_amdgcn_sched_barrier(0);                     // no row is fetched early into the right-hand side
    ham_eqs<S, LUT>(w.c, a, b, da, db, st_);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NR; ++i) { dy[i] = da[i]; dy[NR + i] = db[i]; }
  };
  int st = 0, attempts = 0;
  double t = ts ? ts[0] : ts0, h = h0;
  {
    double a[NR], b[NR];
    w.load(q0, B, a); w.load(p0, B, b);
    if (row0 == 0) { w.store(qout, B, a); w.store(pout, B, b); }
#pragma unroll
    for (int i = 0; i < NR; ++i) { py[i * 256] = a[i]; py[(NR + i) * 256] = b[i]; }
  }
  auto store_state = [&](double* qo, double* po) {
    double a[NR], b[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { a[i] = py[i * 256]; b[i] = py[(NR + i) * 256]; }
    w.store(qo, B, a); w.store(po, B, b);
  };
  double* fq = qout; double* fp = pout;                    // iterate: where the next frame goes
  int until_frame = it_every;
#pragma unroll 1
  for (int call = 0; call < ncalls; ++call) {
    int budget = max_sub;
    if (call > 0) { t = ts ? ts[0] : ts0; h = h0; failed = false; }
    {
      // dydt_in of EVERY call by the instructions a separate launch starts with (see hamk::rkf45_body_parked): `iterate`
      // is bit-identical to the calls one by one by construction, not by two inlined copies compiling alike
      double y0[D], f0[D];
#pragma unroll
      for (int j = 0; j < D; ++j) y0[j] = py[j * 256];
      rhs(y0, f0, st);
#pragma unroll
      for (int j = 0; j < D; ++j) pf[j * 256] = f0[j];
    }
    for (int rr = 1; rr < nt; ++rr) {
      const double ti = ts ? ts[rr] : ts1;
      for (;;) {
        const bool active = (sgn * (ti - t) > 0.0) && (budget > 0) && !failed;
        if (!__any(active)) break;                         // wave-uniform exit
        HAMK_MARK(1);                                      // (probe builds: scripts/isa_stats.py rkf45_attempt_stats)
        const double dt = ti - t;
        double hh = h;
        bool final_step = false;
        if ((dt >= 0.0 && hh > dt) || (dt < 0.0 && hh < dt)) { hh = dt; final_step = true; }
        int st_try = 0;
        double out[D];                                      // the last right-hand side's result: k_{sg + 1} at the top of stage sg
#pragma unroll
        for (int j = 0; j < D; ++j) out[j] = 0.0;
#pragma unroll 1
        for (int sg = 0; sg < 6; ++sg) {
          double yt[D];
          switch (sg) {
            case 0:
#pragma unroll
              for (int j = 0; j < D; ++j) yt[j] = py[j * 256] + (1.0 / 4.0) * hh * pf[j * 256];
              break;
            case 1:
#pragma unroll
              for (int j = 0; j < D; ++j) yt[j] = py[j * 256] + hh * ((3.0 / 32.0) * pf[j * 256] + (9.0 / 32.0) * out[j]);
              break;
            case 2:
#pragma unroll
              for (int j = 0; j < D; ++j)
                yt[j] = py[j * 256] + hh * ((1932.0 / 2197.0) * pf[j * 256] + (-7200.0 / 2197.0) * HAMK_RKF_K(0, j) + (7296.0 / 2197.0) * out[j]);
              break;
            case 3:
#pragma unroll
              for (int j = 0; j < D; ++j)
                yt[j] = py[j * 256] + hh * ((8341.0 / 4104.0) * pf[j * 256] + (-32832.0 / 4104.0) * HAMK_RKF_K(0, j) + (29440.0 / 4104.0) * HAMK_RKF_K(1, j) +
                                            (-845.0 / 4104.0) * out[j]);
              break;
            case 4:
#pragma unroll
              for (int j = 0; j < D; ++j)
                yt[j] = py[j * 256] + hh * ((-6080.0 / 20520.0) * pf[j * 256] + (41040.0 / 20520.0) * HAMK_RKF_K(0, j) + (-28352.0 / 20520.0) * HAMK_RKF_K(1, j) +
                                            (9295.0 / 20520.0) * HAMK_RKF_K(2, j) + (-5643.0 / 20520.0) * out[j]);
              break;
            default: {
              double ye[D];
#pragma unroll
              for (int j = 0; j < D; ++j) {
                const double f0 = pf[j * 256], k3 = HAMK_RKF_K(1, j), k4 = HAMK_RKF_K(2, j), k5 = HAMK_RKF_K(3, j), k6 = out[j];
                yt[j] = py[j * 256] + hh * ((902880.0 / 7618050.0) * f0 + (3953664.0 / 7618050.0) * k3 + (3855735.0 / 7618050.0) * k4 +
                                            (-1371249.0 / 7618050.0) * k5 + (277020.0 / 7618050.0) * k6);
                ye[j] = hh * ((1.0 / 360.0) * f0 + (-128.0 / 4275.0) * k3 + (-2197.0 / 75240.0) * k4 + (1.0 / 50.0) * k5 + (2.0 / 55.0) * k6);
              }
#pragma unroll
              for (int j = 0; j < D; ++j) HAMK_RKF_YN(j) = yt[j];
#pragma unroll
              for (int j = 0; j < D; ++j) HAMK_RKF_E(j) = ye[j];
              break;
            }
          }
          HAMK_MARK(3);
          HAMK_PIN(yt);
          rhs(yt, out, st_try);
          HAMK_PIN(out);
          HAMK_MARK(0);
          if (sg < 4) put_k(sg, out);                      // k2..k5 (k6 and dydt at the trial state are used from the registers)
        }
        if (active) st |= st_try;
        // cstd.c: std_control_hadjust, ord = 5; the norm runs over the trajectory's 2n components = the quad's lanes
        double yn[D];
        double rl = 2.2250738585072014e-308;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          yn[j] = HAMK_RKF_YN(j);
          if (!w.owns(j < NR ? j : j - NR)) continue;
          const double rj = fabs(HAMK_RKF_E(j)) / fabs(eps_rel * (fabs(yn[j]) + fabs(hh * out[j])) + eps_abs);
          rl = (rj > rl) ? rj : rl;
        }
        const double rmax = qmax(rl);
        const double tnew = final_step ? ti : t + hh;
        const double h_old = hh;
        bool reject = false, fail_now = false;
        if (rmax > 1.1) {
          double rr5 = 0.9 * rpow_inv<5>(rmax);
          if (rr5 < 0.2) rr5 = 0.2;
          const double hdec = rr5 * h_old;
          if (fabs(hdec) < fabs(h_old) && (tnew + hdec) != tnew) { reject = true; hh = hdec; }
          else if (api2) { fail_now = true; hh = hdec; }     // GSL_FAILURE; y and t stay advanced
        } else if (rmax < 0.5) {
          double rr6 = 0.9 * rpow_inv<6>(rmax);
          if (rr6 > 5.0) rr6 = 5.0;
          if (rr6 < 1.0) rr6 = 1.0;
          hh = rr6 * h_old;
        }
        if (active) {                                      // evolve.c: accept or undo
          ++attempts; --budget;
          if (fail_now) { failed = true; st |= ST_UNDERFLOW; }
          if (reject || fail_now || !api2 || !final_step) h = hh;
          if (!reject) {
            if (!(sgn * (tnew - t) > 0.0)) st |= ST_UNDERFLOW;
            t = tnew;
#pragma unroll
            for (int j = 0; j < D; ++j) { py[j * 256] = yn[j]; pf[j * 256] = out[j]; }
          }
        }
        HAMK_MARK(2);
      }
      if (sgn * (ti - t) > 0.0 && !failed) st |= ST_MAXSTEPS;
      if (rr >= row0 && call == ncalls - 1) {
        double* qo = (inplace == 2) ? const_cast<double*>(q0) : (inplace ? qout : qout + (i64)rr * N * B);
        double* po = (inplace == 2) ? const_cast<double*>(p0) : (inplace ? pout : pout + (i64)rr * N * B);
        store_state(qo, po);
      }
    }
    if (it_every > 0 && --until_frame == 0) {
      until_frame = it_every;
      store_state(fq, fp);
      fq += (i64)N * B; fp += (i64)N * B;
    }
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NR; ++i) bad = bad || (w.owns(i) && (is_nonfinite_bits(py[i * 256]) || is_nonfinite_bits(py[(NR + i) * 256])));
  if (bad) st |= ST_NONFINITE;
  const int stq = qor(st);
  if (w.real && w.c.r == 0) {
    if (status) status[w.t] = stq;
    if (nsub) nsub[w.t] = attempts;
  }
#undef HAMK_RKF_YN
#undef HAMK_RKF_E
#undef HAMK_RKF_K
#undef HAMK_RKF_LROW
}

template <class S>
HAMK_DEV void rkf45_body(double* smem, const double* q0, const double* p0, double* qout, double* pout, i64 B, int nt,
                         const double* ts, double ts0, double ts1, double h0, double eps_abs, double eps_rel,
                         int flags, int max_sub, int* status, int* nsub, int ncalls, int it_every) {
  if constexpr (HAMK_QUAD_RKF_PARK != 0) {
    rkf45_body_parked<S>(smem, q0, p0, qout, pout, B, nt, ts, ts0, ts1, h0, eps_abs, eps_rel, flags, max_sub, status, nsub, ncalls, it_every);
    return;
  }
  constexpr int N = S::N, NR = Geo<N>::NR, D = 2 * NR;
  constexpr bool LUT = StageTrig<S>::lut;
  if constexpr (LUT) lut_load();
  const int row0 = flags & 1, inplace = (flags >> 8) & 3, gsl_api = (flags >> 16) & 3;      // see hamk::rkf45_body
  const bool api2 = gsl_api != 1;
  const double sgn = (!api2 || h0 > 0.0) ? 1.0 : -1.0;
  bool failed = false;
  Where<S> w(smem, B);
  double y[D], f[D];                                       // [q of the lane's rows; p of the lane's rows]
  {
    double a[NR], b[NR];
    w.load(q0, B, a); w.load(p0, B, b);
#pragma unroll
    for (int i = 0; i < NR; ++i) { y[i] = a[i]; y[NR + i] = b[i]; }
    if (row0 == 0) { w.store(qout, B, a); w.store(pout, B, b); }
  }
  auto rhs = [&](const double (&yy)[D], double (&dy)[D], int& st_) {
    double a[NR], b[NR], da[NR], db[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { a[i] = yy[i]; b[i] = yy[NR + i]; }
    ham_eqs<S, LUT>(w.c, a, b, da, db, st_);
#pragma unroll
    for (int i = 0; i < NR; ++i) { dy[i] = da[i]; dy[NR + i] = db[i]; }
  };
  int st = 0, attempts = 0;
  double t = ts ? ts[0] : ts0, h = h0;
  double* fq = qout; double* fp = pout;                    // iterate: where the next frame goes
  int until_frame = it_every;
#pragma unroll 1
  for (int call = 0; call < ncalls; ++call) {
    int budget = max_sub;
    if (call > 0) { t = ts ? ts[0] : ts0; h = h0; failed = false; }
    rhs(y, f, st);                                         // dydt_in of EVERY call by the instructions a separate launch starts with (see hamk::rkf45_body)
    for (int rr = 1; rr < nt; ++rr) {
      const double ti = ts ? ts[rr] : ts1;
      for (;;) {
        const bool active = (sgn * (ti - t) > 0.0) && (budget > 0) && !failed;
        if (!__any(active)) break;                         // wave-uniform exit
        const double dt = ti - t;
        double hh = h;
        bool final_step = false;
        if ((dt >= 0.0 && hh > dt) || (dt < 0.0 && hh < dt)) { hh = dt; final_step = true; }
        double k2[D], k3[D], k4[D], k5[D], k6[D], yn[D], fn[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { k2[j] = k3[j] = k4[j] = k5[j] = k6[j] = 0.0; yn[j] = y[j]; fn[j] = 0.0; }
        int st_try = 0;
#pragma unroll 1
        for (int sg = 0; sg < 6; ++sg) {
          double yt[D], out[D];
          switch (sg) {
            case 0:
#pragma unroll
              for (int j = 0; j < D; ++j) yt[j] = y[j] + (1.0 / 4.0) * hh * f[j];
              break;
            case 1:
#pragma unroll
              for (int j = 0; j < D; ++j) yt[j] = y[j] + hh * ((3.0 / 32.0) * f[j] + (9.0 / 32.0) * k2[j]);
              break;
            case 2:
#pragma unroll
              for (int j = 0; j < D; ++j) yt[j] = y[j] + hh * ((1932.0 / 2197.0) * f[j] + (-7200.0 / 2197.0) * k2[j] + (7296.0 / 2197.0) * k3[j]);
              break;
            case 3:
#pragma unroll
              for (int j = 0; j < D; ++j)
                yt[j] = y[j] + hh * ((8341.0 / 4104.0) * f[j] + (-32832.0 / 4104.0) * k2[j] + (29440.0 / 4104.0) * k3[j] + (-845.0 / 4104.0) * k4[j]);
              break;
            case 4:
#pragma unroll
              for (int j = 0; j < D; ++j)
                yt[j] = y[j] + hh * ((-6080.0 / 20520.0) * f[j] + (41040.0 / 20520.0) * k2[j] + (-28352.0 / 20520.0) * k3[j] +
                                     (9295.0 / 20520.0) * k4[j] + (-5643.0 / 20520.0) * k5[j]);
              break;
            default:
#pragma unroll
              for (int j = 0; j < D; ++j) {
                yn[j] = y[j] + hh * ((902880.0 / 7618050.0) * f[j] + (3953664.0 / 7618050.0) * k3[j] + (3855735.0 / 7618050.0) * k4[j] +
                                     (-1371249.0 / 7618050.0) * k5[j] + (277020.0 / 7618050.0) * k6[j]);
                yt[j] = yn[j];
              }
              break;
          }
          rhs(yt, out, st_try);
          switch (sg) {
            case 0:
#pragma unroll
              for (int j = 0; j < D; ++j) k2[j] = out[j];
              break;
            case 1:
#pragma unroll
              for (int j = 0; j < D; ++j) k3[j] = out[j];
              break;
            case 2:
#pragma unroll
              for (int j = 0; j < D; ++j) k4[j] = out[j];
              break;
            case 3:
#pragma unroll
              for (int j = 0; j < D; ++j) k5[j] = out[j];
              break;
            case 4:
#pragma unroll
              for (int j = 0; j < D; ++j) k6[j] = out[j];
              break;
            default:
#pragma unroll
              for (int j = 0; j < D; ++j) fn[j] = out[j];
              break;
          }
        }
        if (active) st |= st_try;
        // cstd.c: std_control_hadjust, ord = 5; the norm runs over the trajectory's 2n components = the quad's lanes
        double rl = 2.2250738585072014e-308;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          if (!w.owns(j < NR ? j : j - NR)) continue;
          const double e = hh * ((1.0 / 360.0) * f[j] + (-128.0 / 4275.0) * k3[j] + (-2197.0 / 75240.0) * k4[j] + (1.0 / 50.0) * k5[j] + (2.0 / 55.0) * k6[j]);
          const double rj = fabs(e) / fabs(eps_rel * (fabs(yn[j]) + fabs(hh * fn[j])) + eps_abs);
          rl = (rj > rl) ? rj : rl;
        }
        const double rmax = qmax(rl);
        const double tnew = final_step ? ti : t + hh;
        const double h_old = hh;
        bool reject = false, fail_now = false;
        if (rmax > 1.1) {
          double rr5 = 0.9 * rpow_inv<5>(rmax);
          if (rr5 < 0.2) rr5 = 0.2;
          const double hdec = rr5 * h_old;
          if (fabs(hdec) < fabs(h_old) && (tnew + hdec) != tnew) { reject = true; hh = hdec; }
          else if (api2) { fail_now = true; hh = hdec; }     // GSL_FAILURE; y and t stay advanced
        } else if (rmax < 0.5) {
          double rr6 = 0.9 * rpow_inv<6>(rmax);
          if (rr6 > 5.0) rr6 = 5.0;
          if (rr6 < 1.0) rr6 = 1.0;
          hh = rr6 * h_old;
        }
        if (active) {                                      // evolve.c: accept or undo
          ++attempts; --budget;
          if (fail_now) { failed = true; st |= ST_UNDERFLOW; }
          if (reject || fail_now || !api2 || !final_step) h = hh;
          if (!reject) {
            if (!(sgn * (tnew - t) > 0.0)) st |= ST_UNDERFLOW;
            t = tnew;
#pragma unroll
            for (int j = 0; j < D; ++j) { y[j] = yn[j]; f[j] = fn[j]; }
          }
        }
      }
      if (sgn * (ti - t) > 0.0 && !failed) st |= ST_MAXSTEPS;
      if (rr >= row0 && call == ncalls - 1) {
        double a[NR], b[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { a[i] = y[i]; b[i] = y[NR + i]; }
        double* qo = (inplace == 2) ? const_cast<double*>(q0) : (inplace ? qout : qout + (i64)rr * N * B);
        double* po = (inplace == 2) ? const_cast<double*>(p0) : (inplace ? pout : pout + (i64)rr * N * B);
        w.store(qo, B, a); w.store(po, B, b);
      }
    }
    if (it_every > 0 && --until_frame == 0) {
      until_frame = it_every;
      double a[NR], b[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) { a[i] = y[i]; b[i] = y[NR + i]; }
      w.store(fq, B, a); w.store(fp, B, b);
      fq += (i64)N * B; fp += (i64)N * B;
    }
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NR; ++i) bad = bad || (w.owns(i) && (is_nonfinite_bits(y[i]) || is_nonfinite_bits(y[NR + i])));
  if (bad) st |= ST_NONFINITE;
  const int stq = qor(st);
  if (w.real && w.c.r == 0) {
    if (status) status[w.t] = stq;
    if (nsub) nsub[w.t] = attempts;
  }
}

}  // namespace quad
}  // namespace hamk

// The eight kernels of the path on the quad mapping (the names of HAMK_INSTANTIATE).
#define HAMK_INSTANTIATE_QUAD(S)                                                                                 \
  static_assert(S::INERTIA_POS, "the four-lane mapping factorises K without pivoting: every inertia must be positive (libhamk routes other systems to the lane / wave-cooperative kernels, which pivot)"); \
  HAMK_SCRIBBLE_KERNEL                                                                                           \
  extern "C" __global__ void __launch_bounds__(256) hamk_rk4_steps_k(double* q, double* p, long long B,          \
                                                          double dt, int nsteps, double drift_tol, int* status) { \
    HAMK_QUAD_SMEM(S);                                                                                           \
    __shared__ double park[4 * hamk::quad::Geo<S::N>::NR * 256];                                                 \
    hamk::quad::rk4_body<S>(smem, park, q, p, B, dt, nsteps, drift_tol, status);                                 \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_hameqs_k(const double* q, const double* p, double* dq,  \
                                                                   double* dp, long long B, int* status) {       \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::hameqs_body<S>(smem, q, p, dq, dp, B, status);                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_from_phase_k(const double* q, const double* p,          \
                                                                       double* qd, long long B, int* status) {   \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::from_phase_body<S>(smem, q, p, qd, B, status);                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_observe_k(const double* q, const double* p, double* ke, \
                                                                    double* pe, double* h, long long B,          \
                                                                    int* status) {                               \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::observe_body<S>(smem, q, p, ke, pe, h, B, status);                                               \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_coords_k(const double* q, double* x, long long B) {     \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::coords_body<S>(smem, q, x, B);                                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_to_phase_k(const double* q, const double* qd,           \
                                                                     double* p, long long B) {                   \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::to_phase_body<S>(smem, q, qd, p, B);                                                             \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_observe_config_k(const double* q, const double* qd,     \
                                                                           double* ke, double* lag,              \
                                                                           long long B) {                        \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::observe_config_body<S>(smem, q, qd, ke, lag, B);                                                 \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_rkf45_k(                                                \
      const double* q0, const double* p0, double* qout, double* pout, long long B, int nt, const double* ts,     \
      double ts0, double ts1, double h0, double eps_abs, double eps_rel, int flags, int max_sub,                 \
      int* status, int* nsub, int ncalls, int it_every) {                                                        \
    HAMK_QUAD_SMEM(S);                                                                                           \
    hamk::quad::rkf45_body<S>(smem, q0, p0, qout, pout, B, nt, ts, ts0, ts1, h0, eps_abs, eps_rel, flags,        \
                              max_sub, status, nsub, ncalls, it_every);                                          \
  }
