// hamk_wave.hpp -- wave-cooperative kernels for systems with many generalized coordinates
// (17 <= n <= 64 by default; BASELINE.json config 5, the N-link chain), where one trajectory no longer fits
// the registers of one lane (second-order jets of 2N outputs in N directions).
//
// Mapping: one trajectory per group of NP = 16, 32 or 64 lanes (G = 64/NP trajectories per
// wavefront; n > 32: the whole wavefront, sixteen 16x16 MFMA blocks of which the ten on or above the diagonal are kept); LANE i OF THE GROUP CARRIES AD DIRECTION e_i.  Every lane runs the same
// generated f/U code -- uniform control flow, the SIMT-friendly way to run forward mode --
// at the one-direction jets Jet1<1> / Jet2<1>, with its own seed:
//
//   sweep 1  Jet1<1>, seed d = delta(i, lane): lane i gets column i of J (and dU/dq_i), one row
//            J[k][.] at a time: coords_sink_u hands each output to a sink the moment it is defined;
//   K        K = J^T M J on the f64 matrix cores: four rows are staged in LDS and consumed by
//            v_mfma_f64_16x16x4_f64 (SinkK) -- one 8-byte LDS read per lane per 16-column block per
//            four rows.  No J tile, no second pass over it.
//   solve    LDL^T with rows distributed over lanes, in panels of 16 pivots (factor_blocked): inside a panel every lane
//            drops its entry of the pivot column into an LDS buffer and reads what it needs back as broadcast loads,
//            the forward substitution riding along; the trailing blocks are updated on the matrix cores; back
//            substitution reads L^T from the packed triangle.  Systems with a non-positive inertia: LU with partial
//            pivoting instead (solve_pivoted), as the reference's `inv`.
//   sweep 2  Jet2<1> along the common runtime direction qd with own e_i: lane i accumulates
//            dT/dq_i = -sum_k m_k (J qd)_k ((dJ/dq_i) qd)_k directly in the sink -- the m x n x n
//            Hessian tensor of the reference (Hamilton.hs:222, 512 KiB per point at N = 32)
//            never exists, not even one slice of it.
//
//   sincos   when every sincos site of f takes an input as operand (angles), lane j evaluates
//            sincos(q_j) once and the pairs live in LDS: the sweeps read them (TRIG_REUSE)
//            instead of every lane recomputing all n of them.
//
// State stays SoA in HBM (q[j*B + t]); a group loads/stores one value per lane.  LDS per
// trajectory: NP*(NP+1)/2 (row staging, then K, then L, as a packed lower triangle) + 4*NP + 2*NTRIG
// doubles (5.6 KiB at N = 32, 45 KiB per 256-thread block).  Two wavefronts per SIMD by registers
// (228 VGPRs); a third was measured and does not pay -- the LDS unit is the busy resource.
#pragma once
#include "hamk_device.hpp"

namespace hamk {
namespace wave {

template <int N> struct Geo {
  static constexpr int NP = (N <= 16) ? 16 : (N <= 32) ? 32 : 64;   // lanes per trajectory
  static constexpr int G = 64 / NP;                    // trajectories per wavefront
  static constexpr int WAVES = 4;                      // wavefronts per 256-thread block
};

template <class S> struct Lds {
  static constexpr int NP = Geo<S::N>::NP;
  static constexpr int NT = (S::NTRIG_F > 0) ? S::NTRIG_F : 1;
  // K, then L, as a packed LOWER TRIANGLE: entry (i, j), j <= i, at i (i+1)/2 + j.  Half the footprint
  // of a padded square tile (5.6 instead of 9.75 KiB per trajectory at NP = 32); row k, read across the lanes in the back substitution, is
  // contiguous.  The first 4*NP doubles double as the staging rows of sweep 1.
  static constexpr int TILE = NP * (NP + 1) / 2;
  HAMK_DEV static constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
  static constexpr int NPIV = NP;                           // the pivots d_j (factor_blocked)
  static constexpr int PER_TRAJ = TILE + 4 * NP + 2 * NT + NPIV;   // + 2*NP scratch row (sincos exchange, pivot column + z) + two all-gather buffers + sincos pairs
};

// sincos pairs of the trajectory's current point, resident in LDS (same member syntax as
// hamk::TrigCache, so the generated code and trig_pair<> work on it unchanged)
struct TrigLds { double* s; double* c; double* ax; double* as; double* ac; };

// Where the code relies on "DS operations of one wavefront execute in order" to reuse a buffer
// without a second lds_sync (all lanes have issued their reads before any lane issues the next
// write), the host emulation of tests/host_emulation -- one OS thread per lane -- needs a real
// barrier; on the device this is nothing.
#ifdef HAMK_HOST_EMULATION
#define HAMK_LOCKSTEP() emu_wave_barrier()
#else
#define HAMK_LOCKSTEP() ((void)0)
#endif

// LDS written by some lanes of a wavefront and read by others of the same wavefront: DS
// operations of one wave complete in order, so only the compiler has to be kept from moving them.
HAMK_DEV void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // also a scheduling barrier: the phases of an evaluation must not be interleaved by the
  // machine scheduler (it otherwise overlaps them freely and needs ~400 VGPRs for a stream
  // whose phases need at most ~200 each)
  __builtin_amdgcn_sched_barrier(0);
}

// Value of lane `src` of the caller's group, for all lanes of the group.  grp4 = 4 * (first lane
// of the group): the byte index ds_bpermute wants is grp4 + 4*src, and with src a literal the
// 4*src goes into the instruction's offset field -- no per-source index registers.
HAMK_DEV double bcast4(double x, int grp4, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(grp4 + 4 * src, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(grp4 + 4 * src, __double2hiint(x));
  return __hiloint2double(hi, lo);
}

// Value of lane `lane` (wave-uniform, here always a literal) for every lane of the wavefront: two v_readlane_b32 into a scalar
// register pair -- no LDS, no counter to wait for.  One trajectory per wavefront (n > 32) exchanges its pivot rows this way.
HAMK_DEV double read_lane(double x, int lane) {
#ifdef HAMK_HOST_EMULATION
  return emu_read_lane(x, lane);
#else
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
#endif
}

template <int NP> HAMK_DEV double group_max(double x) {
#pragma unroll
  for (int off = NP / 2; off > 0; off >>= 1) { const double y = __shfl_xor(x, off, NP); x = (y > x) ? y : x; }
  return x;
}

template <int NP> HAMK_DEV double group_sum(double x) {
#pragma unroll
  for (int off = NP / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, NP);
  return x;
}

template <int N, int NP> HAMK_DEV void allgather(double* buf, int li, double mine, double (&all)[N]) {
  lds_sync();                       // earlier readers of this buffer are done
  buf[li] = mine;
  lds_sync();
#pragma unroll
  for (int j = 0; j < N; ++j) all[j] = buf[j];
}

// ---- inputs of a sweep, built on the fly from the LDS-resident q (and qd): no register arrays ----
struct InJet1 {                                    // q_j with seed d = delta(j, lane)
  const double* q; int li;
  HAMK_DEV Jet1<1> operator[](int j) const { Jet1<1> r; r.v = q[j]; r.d[0] = (j == li) ? 1.0 : 0.0; return r; }
};
struct InJet2 {                                    // q_j along the common direction v with own e_lane
  const double* q; const double* v; int li;
  HAMK_DEV Jet2<1> operator[](int j) const {
    Jet2<1> r; r.v = q[j]; r.dv = v[j]; r.d[0] = (j == li) ? 1.0 : 0.0; r.dd[0] = 0.0; return r;
  }
};

// ---- sinks for coords_sink -------------------------------------------------------------------
// Sweep 1: K = sum_k m_k J[k][.]^T J[k][.] on the matrix cores.  Lane i of a group learns J[k][i]
// the moment x_k is defined; four consecutive rows are staged in LDS (in the tile, which is free
// during the sweep) and consumed by v_mfma_f64_16x16x4_f64, whose operand layout is exactly
// "four rows, sixteen columns": A[i][kk] and B[kk][j] both sit in lane 16*kk + i -- ONE 8-byte LDS
// read per lane per 16-column block per four rows, where the FMA formulation needs every lane to
// read every column of every row (17 reads per row in circulant form: LDS bandwidth, not FP64 rate,
// bounded it at 58 % of the evaluation).  The f64 MFMA rate itself is lower than the VALU's on this
// chip (profiles/r01_mfma_f64_probe.txt); what the matrix core buys here is the operand broadcast.
// A wavefront holds G = 64/NP trajectories; each MFMA serves one of them with all 64 lanes, the
// accumulators of all G stay in registers: D[i][j] of block (ib, jb) is K[16 ib + i][16 jb + j],
// lane l / register r holding i = 4 r + l/16, j = l%16 (scripts/probes/mfma_f64_layout.hip).
#ifdef HAMK_HOST_EMULATION                                 // tests/host_emulation: g++ spelling of the same vector type
typedef double mfma_d4 __attribute__((vector_size(32)));
#else
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
#endif

template <class S, int NP> struct SinkK {
  static constexpr int G = 64 / NP, NB = NP / 16, NBLK = NB * (NB + 1) / 2;
  static constexpr int PER = Lds<S>::PER_TRAJ;
  HAMK_DEV static constexpr int blk_of(int ib, int jb) { return ib * NB - ib * (ib - 1) / 2 + (jb - ib); }   // ib <= jb
  double* mine;            // staging rows of this lane's trajectory: [4][NP], + li
  const double* rd;        // staging of the wave's first trajectory, + (lane/16)*NP + lane%16
  int kq;                  // lane / 16: which of the four staged rows this lane feeds to the MFMA
  double m0, m1, m2, m3;   // inertias of the staged rows (scalars: an array invites a dynamically indexed load)
  mfma_d4 acc[G][NBLK];
  HAMK_DEV void init(double* smem, int off, int offw, int li, int lw) {
    mine = smem + off + li;
    kq = lw >> 4;
    rd = smem + offw + kq * NP + (lw & 15);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[g][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  }
  template <int R> HAMK_DEV void set_m(double m) {
    if constexpr (R == 0) m0 = m; else if constexpr (R == 1) m1 = m; else if constexpr (R == 2) m2 = m; else m3 = m;
  }
  // SEQ: position of this output in the order of emission (the generator numbers its puts)
  template <int K, int SEQ> HAMK_DEV void put(const Jet1<1>& v) {
#ifdef HAMK_PROBE_SKIP_KACC                                // timing probes (scripts/wave_attrib.py): wrong results
    acc[0][0][K & 3] += v.d[0]; return;
#endif
    mine[(SEQ & 3) * NP] = v.d[0];
    set_m<SEQ & 3>(S::inertia(K));
    if constexpr ((SEQ & 3) == 3) flush<cols_lo(SEQ - 3, SEQ), cols_hi(SEQ - 3, SEQ)>();
  }
  HAMK_DEV void finish() {                                 // M not a multiple of four: zero rows
    if constexpr ((S::M & 3) != 0) {
#pragma unroll
      for (int r = (S::M & 3); r < 4; ++r) mine[r * NP] = 0.0;
      if constexpr ((S::M & 3) <= 1) m1 = 0.0;
      if constexpr ((S::M & 3) <= 2) m2 = 0.0;
      m3 = 0.0;
      flush<cols_lo(S::M - (S::M & 3), S::M - 1), cols_hi(S::M - (S::M & 3), S::M - 1)>();
    }
  }
  // Columns in which the rows delivered by puts s0..s1 can be non-zero: output k = S::seq_out(s) depends on the inputs
  // S::dep_lo(k)..S::dep_hi(k) (generated tables: the structure of the coordinate map, known when the kernels are specialised).
  HAMK_DEV static constexpr int cols_lo(int s0, int s1) {
    int lo = NP;
    for (int s = s0; s <= s1; ++s) { const int k = S::seq_out(s); if (S::dep_hi(k) >= 0 && S::dep_lo(k) < lo) lo = S::dep_lo(k); }
    return lo;
  }
  HAMK_DEV static constexpr int cols_hi(int s0, int s1) {
    int hi = -1;
    for (int s = s0; s <= s1; ++s) { const int k = S::seq_out(s); if (S::dep_hi(k) > hi) hi = S::dep_hi(k); }
    return hi;
  }
  // LO..HI: the columns in which the four staged rows can be non-zero.  A 16-column block outside it multiplies zeros: its
  // operand is not read and its matrix instructions are not issued -- K += J[rows]^T M J[rows] touches only the blocks
  // (ib, jb) inside the range.  An N-link chain (x_k, y_k depend on q_0..q_k) issues half the MFMAs of a dense map; the
  // accumulation is bound by the f64 matrix rate (64 cycles per 16x16x4 block product), so that is half its time.
  template <int LO, int HI> HAMK_DEV void flush() {
    constexpr int B0 = (HI < 0) ? 0 : LO / 16, B1 = (HI < 0) ? -1 : HI / 16;      // blocks B0..B1 (none: the rows are constants)
    lds_sync();
    // inertia of the row this lane feeds (folds when all are equal).  All four are read
    // unconditionally and blended as VALUES: a conditional read is turned into one read through a
    // selected address before inlining, and that dynamic index then pins the whole sink in scratch.
    const double a0 = m0, a1 = m1, a2 = m2, a3 = m3;
    const double m01 = (kq & 1) ? a1 : a0, m23 = (kq & 1) ? a3 : a2;
    const double m = (kq & 2) ? m23 : m01;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      double a[NB], am[NB];
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) if (cb >= B0 && cb <= B1) { a[cb] = rd[g * PER + 16 * cb]; am[cb] = m * a[cb]; }
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int jb = ib; jb < NB; ++jb)
          if (ib >= B0 && jb <= B1)
            acc[g][blk_of(ib, jb)] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[ib], a[jb], acc[g][blk_of(ib, jb)], 0, 0, 0);
    }
    lds_sync();                                            // the next rows overwrite the staging area
  }
  // K of every trajectory of the wave into its packed lower triangle.  Block (ib, jb), ib <= jb, holds
  // K[16 ib + 4 r + kq][16 jb + l16]; an entry above the diagonal goes to its mirror position (K is
  // symmetric, so in a diagonal block two lanes write the same number to the same place) -- no
  // predicated stores, which would cost an exec-mask round trip each.
  HAMK_DEV void store(double* smem, int offw, int lw) const {
    const int kq_ = lw >> 4, l16 = lw & 15;
    double* base = smem + offw;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int jb = ib; jb < NB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ib + 4 * r + kq_, col = 16 * jb + l16;
            const int hi = (ib != jb || col > row) ? col : row, lo = (ib != jb || col > row) ? row : col;
            base[g * PER + hi * (hi + 1) / 2 + lo] = acc[g][blk_of(ib, jb)][r];
          }
    }
  }
};
template <class S> struct SinkT {                  // sweep 2: dT/dq_i = -sum_k m_k x_k.dv x_k.dd
  double dT = 0.0;
  template <int K, int SEQ> HAMK_DEV void put(const Jet2<1>& v) { dT = fma(-(S::inertia(K) * v.dv), v.dd[0], dT); }
};
template <class S> struct SinkP {                  // momenta: p_i = sum_k J[k][i] m_k (J qd)_k
  double p = 0.0;
  template <int K, int SEQ> HAMK_DEV void put(const Jet2<1>& v) { p = fma(S::inertia(K) * v.dv, v.d[0], p); }
};

// ---- per-group context -------------------------------------------------------------------------
// LDS addresses are formed as smem + off + constant with `off` (and the lane id) re-defined
// opaquely at the top of every evaluation (`launder`): the constant then folds into the DS
// instruction's offset field, and loop-invariant code motion cannot hoist hundreds of distinct
// LDS addresses out of the stepping loop into VGPRs (measured before: 365 spilled registers and
// 1.4 KiB of scratch per lane, 78 % of the wave cycles waiting).
template <class S> struct Ctx {
  static constexpr int N = S::N, M = S::M, NP = Geo<N>::NP;
  double* smem;      // the block's __shared__ array
  int off;           // this trajectory's offset into it (doubles)
  int offw;          // offset of the wavefront's first trajectory
  int lw;            // lane within the wavefront
  int li;            // lane within the group = AD direction
  HAMK_DEV double* tile() const { return smem + off; }                                   // [TILE] K, then L
  HAMK_DEV double* rowbuf() const { return smem + off + Lds<S>::TILE; }                  // [2*NP] exchange buffer
  HAMK_DEV double* ga() const { return smem + off + Lds<S>::TILE + 2 * NP; }             // [NP] q
  HAMK_DEV double* gb() const { return smem + off + Lds<S>::TILE + 3 * NP; }             // [NP] qd
  HAMK_DEV TrigLds trig() const {
    TrigLds t; t.s = smem + off + Lds<S>::TILE + 4 * NP; t.c = t.s + Lds<S>::NT; t.ax = t.as = t.ac = nullptr; return t;
  }
  HAMK_DEV double* piv() const { return smem + off + Lds<S>::TILE + 4 * NP + 2 * Lds<S>::NT; }   // [NP] d_j (blocked factorisation)
  int g4;            // 4 * (first lane of the group within the wavefront)
  HAMK_DEV int grp4() const { return g4; }
#ifdef HAMK_HOST_EMULATION
  HAMK_DEV Ctx launder() const { return *this; }
#else
  HAMK_DEV Ctx launder() const { Ctx c = *this; asm volatile("" : "+v"(c.off), "+v"(c.li), "+v"(c.g4), "+v"(c.offw), "+v"(c.lw)); return c; }
#endif
};

// Fill the LDS-resident sincos pairs cooperatively when every site's operand is an input.
// Returns the TRIG mode the first sweep must use.
template <class S> HAMK_DEV void cooperative_trig(const Ctx<S>& c, double qi) {
  if constexpr (S::TRIG_ALL_INPUTS && S::NTRIG_F > 0) {
    constexpr int NP = Ctx<S>::NP;
    double sv, cv;
    sincos_f64(qi, sv, cv);                               // lane j: sincos(q_j), once per trajectory
    lds_sync();
    c.rowbuf()[c.li] = sv; c.rowbuf()[NP + c.li] = cv;        // (the row buffer is free here)
    lds_sync();
    if (c.li < S::NTRIG_F) {                              // lane k fills site k
      const int src = S::trig_input(c.li);
      c.trig().s[c.li] = c.rowbuf()[src];
      c.trig().c[c.li] = c.rowbuf()[NP + src];
    }
    static_assert(S::NTRIG_F <= NP, "more sincos sites than lanes in a group");
    lds_sync();
  }
}

// LDL^T in panels of 16 pivots, left-looking: before panel b is factorised, block column b of what is
// left of K (rows >= 16 b) receives the contribution of all earlier panels at once,
//   K[16 ib + i][16 b + j] -= sum_c L[16 ib + i][c] d_c L[16 b + j][c],   c < 16 b,
// on the matrix cores, operands read from the packed triangle where the panels left L (one 8-byte read
// per lane per four columns -- the same operand layout as the accumulation of K in SinkK), the block
// read from and written back to its place in the triangle.  Inside a panel the column broadcast of
// `factor` below runs unchanged, but its trailing update stops at the panel's edge: a lane reads
// (16 - 1 - j mod 16) values per pivot instead of (N - 1 - j), and carries 16 entries of its row in
// registers instead of N.  Broadcast reads per factorisation at N = 32: 240 wide reads -> 112, plus
// 32 narrow ones for the one off-diagonal block.
template <class S>
HAMK_DEV void factor_blocked(const Ctx<S>& c, double& dinv, int& st, double& z) {
  constexpr int N = S::N, NP = Ctx<S>::NP, G = 64 / NP, PER = Lds<S>::PER_TRAJ, NBR = (N + 15) / 16;
  auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
  const int li = c.li;
  const int kq = c.lw >> 4, l16 = c.lw & 15;
  bool ok = true;
  dinv = 0.0;
  double* Lrow = c.tile() + li * (li + 1) / 2;
  double* cA = c.rowbuf();
  double* cB = c.rowbuf() + NP;
  double* cZ = c.gb();
#pragma unroll
  for (int pb = 0; pb < NBR; ++pb) {
    const int J0 = 16 * pb, J1 = (16 * pb + 16 < N) ? 16 * pb + 16 : N;     // this panel's pivots [J0, J1)
    if (pb > 0) {
      HAMK_LOCKSTEP();
      lds_sync();                                            // (the pivots of the earlier panels went to c.piv() as they were found)                                            // L of the earlier panels and their pivots are in LDS
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double* T = c.smem + c.offw + g * PER;         // trajectory g of the wavefront: all 64 lanes serve it
        const double* D = T + Lds<S>::TILE + 4 * NP + 2 * Lds<S>::NT;
        double xb[4 * (NBR > 1 ? NBR - 1 : 1)], xd[4 * (NBR > 1 ? NBR - 1 : 1)];
#pragma unroll
        for (int cm = 0; cm < 4 * pb; ++cm) {                // columns 4 cm + kq of the earlier panels
          xb[cm] = T[tri(J0 + l16, 4 * cm + kq)];
          xd[cm] = D[4 * cm + kq];
        }
#pragma unroll
        for (int ib = pb; ib < NBR; ++ib) {
          mfma_d4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ib + 4 * r + kq, col = J0 + l16;
            const int hi = (ib != pb || row > col) ? row : col, lo = (ib != pb || row > col) ? col : row;
            acc[r] = T[hi * (hi + 1) / 2 + lo];
          }
#pragma unroll
          for (int cm = 0; cm < 4 * pb; ++cm) {
            const double xa = (ib == pb) ? xb[cm] : T[tri(16 * ib + l16, 4 * cm + kq)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-(xa * xd[cm]), xb[cm], acc, 0, 0, 0);
          }
          double* Tw = c.smem + c.offw + g * PER;
          // In the diagonal block lanes (row, col) and (col, row) hold the same entry of the symmetric update, each from
          // its own MFMA -- fl(fl(L_i d) L_j) in one, fl(fl(L_j d) L_i) in the other, equal up to the last bit -- and both
          // would address the same packed slot: only the lane on or below the diagonal stores, so what the triangle holds
          // never depends on the order in which a wavefront's stores land (four predicated stores per panel).
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ib + 4 * r + kq, col = J0 + l16;
            if (ib != pb || row >= col) Tw[row * (row + 1) / 2 + col] = acc[r];
          }
        }
      }
      lds_sync();
    }
    if constexpr (G == 1) {
      // ONE TRAJECTORY PER WAVEFRONT (n > 32): lane i = row i, and the pivot rows travel through SCALAR registers.  Every lane
      // keeps its WHOLE row segment of the panel -- S[i][c] for the panel's 16 columns c, on both sides of the diagonal (the
      // mirrored entry of the packed triangle where c > i) -- and eliminates it in full: after the steps before j, lane j's
      // segment is row j of the Schur complement, and "column j as seen by every row" is that row read lane by lane
      // (v_readlane_b32, the lane a literal): S[k][j] = S[j][k].  No LDS exchange, no lds_sync per pivot pair -- the panel
      // version below pays a write -> fence -> broadcast-read round trip per pair of pivots, 32 of them in a row at n = 64, each
      // of which also drains the scratch traffic of the kernel's spilled registers (the fence waits for every counter):
      // profiles/r05d_wave_probe.jsonl, the factorisation was 63 % of chain64's launch.  (Rows eliminate with their OWN
      // multipliers against the pivot ROW: plain Gaussian elimination, L D L^T up to the roundoff by which the two copies of a
      // symmetric entry differ; the results agree with the exchange version to roundoff, not bitwise.)
      double row[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int cidx = J0 + b;
        const int hi_ = (li > cidx) ? li : cidx, lo_ = (li > cidx) ? cidx : li;      // (v_max / v_min: no lane mask per column)
        row[b] = (cidx < N) ? c.tile()[hi_ * (hi_ + 1) / 2 + lo_] : 0.0;
      }
#pragma unroll
      for (int j = J0; j + 1 < J1; j += 2) {
        const double a = read_lane(row[j - J0], j), b = read_lane(row[j + 1 - J0], j), cc = read_lane(row[j + 1 - J0], j + 1);
        const double zj = read_lane(z, j), zj1 = read_lane(z, j + 1);
        const double det = fma(a, cc, -(b * b));
        ok = ok && (a > 0.0) && (det > 0.0);
        const double inv_a = frcp(a), inv_det = frcp(det);
        const double inv_c = a * inv_det;
        const double l = b * inv_a;
        const double zj1p = fma(-l, zj, zj1);
        const double l0 = (li > j) ? row[j - J0] * inv_a : 0.0;
        const double l1 = (li > j + 1) ? fma(-l0, b, row[j + 1 - J0]) * inv_c : 0.0;
        // the two pivots are the same number in every lane (they came through scalar registers): every lane stores them into the
        // trajectory's pivot buffer, and lane i picks d_i up ONCE after the last panel -- instead of a chain of 2 x n lane-masked
        // selects that carries (dinv, dmine) through the whole factorisation and keeps a scalar pair per pivot alive for it
        c.piv()[j] = a;
        c.piv()[j + 1] = det * inv_a;
        z = fma(-l1, zj1p, fma(-l0, zj, z));
        const double al = fma(-l1, l, l0);
#pragma unroll
        for (int k = j + 2; k < J1; ++k) {
          const double ak = read_lane(row[k - J0], j), bk = read_lane(row[k - J0], j + 1);
          row[k - J0] = fma(-al, ak, fma(-l1, bk, row[k - J0]));
#ifndef HAMK_HOST_EMULATION
          // four columns' worth of scalar pairs at a time: left alone the scheduler reads the whole pivot rows first (56 scalar
          // registers on top of the kernel's own) and the kernel spills SGPRs by the hundred
          if (((k - j) & 3) == 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        Lrow[(li > j) ? j : li] = l0;
        Lrow[(li > j + 1) ? j + 1 : li] = l1;
      }
      if (((J1 - J0) & 1) != 0) {                            // last pivot of an odd N: nothing below it
        const double dj = read_lane(row[J1 - 1 - J0], J1 - 1);
        ok = ok && (dj > 0.0);
        c.piv()[J1 - 1] = dj;
      }
      continue;
    }
    double row[16];                                          // this lane's entries in the panel's columns (entries beyond li: never used)
#pragma unroll
    for (int b = 0; b < 16; ++b) row[b] = (J0 + b < N) ? Lrow[J0 + b] : 0.0;
#pragma unroll
    for (int j = J0; j + 1 < J1; j += 2) {
      HAMK_LOCKSTEP();
      cA[li] = row[j - J0];
      cB[li] = row[j + 1 - J0];
      cZ[li] = z;
      lds_sync();
      const double a = cA[j], b = cA[j + 1], cc = cB[j + 1], zj = cZ[j], zj1 = cZ[j + 1];
      const double det = fma(a, cc, -(b * b));
      ok = ok && (a > 0.0) && (det > 0.0);
      const double inv_a = frcp(a), inv_det = frcp(det);
      const double inv_c = a * inv_det;
      const double l = b * inv_a;
      const double zj1p = fma(-l, zj, zj1);
      const double l0 = (li > j) ? row[j - J0] * inv_a : 0.0;
      const double l1 = (li > j + 1) ? fma(-l0, b, row[j + 1 - J0]) * inv_c : 0.0;
      c.piv()[j] = a;                                         // (group-uniform: every lane of the trajectory stores the same two numbers;
      c.piv()[j + 1] = det * inv_a;                           //  lane i reads d_i after the last panel -- see the one-trajectory branch above)
      z = fma(-l1, zj1p, fma(-l0, zj, z));
      const double al = fma(-l1, l, l0);
#pragma unroll
      for (int k = j + 2; k < J1; ++k) row[k - J0] = fma(-al, cA[k], fma(-l1, cB[k], row[k - J0]));
      Lrow[(li > j) ? j : li] = l0;
      Lrow[(li > j + 1) ? j + 1 : li] = l1;
    }
    if (((J1 - J0) & 1) != 0) {                              // last pivot of an odd N: nothing below it
      HAMK_LOCKSTEP();
      cA[li] = row[J1 - 1 - J0];
      lds_sync();
      const double dj = cA[J1 - 1];
      ok = ok && (dj > 0.0);
      c.piv()[J1 - 1] = dj;
    }
  }
  if (!ok && li < N) st |= ST_SINGULAR;
  lds_sync();
  dinv = (li < N) ? frcp(c.piv()[li]) : 0.0;               // lane i's 1 / d_i (see the pivot stores above)
}

template <int NP> HAMK_DEV int group_min(int x) {
#pragma unroll
  for (int off = NP / 2; off > 0; off >>= 1) { const int y = __shfl_xor(x, off, NP); x = (y < x) ? y : x; }
  return x;
}

// K v = rhs by LU WITH PARTIAL PIVOTING -- what the reference does for every K: hmatrix `inv` is LAPACK's
// dgesv against the identity (Hamilton.hs:321, :381).  Used where an inertia is not positive (S::INERTIA_POS
// false): K = J^T M J is then symmetric but need not be definite, LDL^T without pivoting can meet a zero or
// tiny pivot on an invertible matrix, and "flag every lane singular" would be a different answer from the
// reference's on the same input.  Rows stay distributed over the lanes (lane i: row i of K, all N entries,
// read from the packed triangle through the mirrored index); per column the group finds the largest
// remaining |a[r][c]| (shuffles), the lane that holds it publishes its row and right-hand side through LDS
// and retires, everybody else eliminates.  No row ever moves: `mycol` remembers which column a lane's row
// pivoted.  Back substitution column by column through LDS.  N LDS round trips and N(N+1)/2 FMAs per lane
// each way: a slow path by design -- systems with a non-positive inertia are rare, and correct beats fast.
// Returns v_li; an exactly zero (or non-finite) pivot column sets ST_SINGULAR and yields NaN, where the
// reference raises.
template <class S>
HAMK_DEV double solve_pivoted(const Ctx<S>& c, int& st, double rhs) {
  constexpr int N = S::N, NP = Ctx<S>::NP;
  const int li = c.li;
  const double* T = c.tile();
  double* cR = c.rowbuf();                                 // [NP] the pivot row; [NP]: its right-hand side
  double* cX = c.gb();                                     // [NP] the solution (the qd buffer is free here)
  double row[N];
#pragma unroll
  for (int b = 0; b < N; ++b) {
    const int hi = (li > b) ? li : b, lo = (li > b) ? b : li;
    row[b] = T[hi * (hi + 1) / 2 + lo];                    // lanes li >= N read padding: they never pivot, never update
  }
  bool used = li >= N, singular = false;
  int mycol = -1;
  double z = rhs;
#pragma unroll
  for (int c0 = 0; c0 < N; ++c0) {
    double cand = fabs(row[c0]);
    if (used) cand = -1.0;
    else if (is_nonfinite_bits(cand)) cand = 0.0;          // every lane must see the same maximum: no NaN in the reduction
    const double m = group_max<NP>(cand);
    const int who = group_min<NP>((!used && cand == m) ? li : NP);     // the first row that attains it
    singular = singular || !(m > 0.0);
    HAMK_LOCKSTEP();
    if (li == who) {
#pragma unroll
      for (int k = c0; k < N; ++k) cR[k] = row[k];
      cR[NP] = z;
      used = true;
      mycol = c0;
    }
    lds_sync();
    if (!used) {
      const double l = row[c0] / cR[c0];
#pragma unroll
      for (int k = c0 + 1; k < N; ++k) row[k] = fma(-l, cR[k], row[k]);
      z = fma(-l, cR[NP], z);
    }
  }
  // U x = z: the lane with mycol == c0 holds row c0 of U (entries c0..N-1) and z_c0
#pragma unroll
  for (int c0 = N - 1; c0 >= 0; --c0) {
    HAMK_LOCKSTEP();
    if (mycol == c0) cX[c0] = z / row[c0];
    lds_sync();
    const double xc = cX[c0];
    if (mycol >= 0 && mycol < c0) z = fma(-row[c0], xc, z);
  }
  lds_sync();
  double v = (li < N) ? cX[li] : 0.0;
  if (singular) {
    if (li < N) st |= ST_SINGULAR;
    v = quiet_nan();
  }
  lds_sync();                                              // the buffers are free again
  return v;
}

// Sweep 1 (with K accumulated in the sink) + the solve's first half: LDL^T in panels of 16 pivots with the forward
// substitution of one right-hand side riding along (factor_blocked; a system of n <= 16 is one panel) -- or, where an
// inertia is not positive, the whole pivoted solve (solve_pivoted).  On return: dinv = 1/d_li, z = (L^-1 rhs)_li,
// gU = dU/dq_li; the packed triangle holds L.
// (Until round 4 a flat, unblocked column-broadcast LDL^T lived here as the alternative for n > 16 -- measured behind the
// panel version, chain32 5.06e7 vs 5.66e7 and chain64 4.6e6 vs 7.6e6 RK4 steps/s, profiles/r02_wave_blocked.jsonl -- removed.)
template <class S>
HAMK_DEV void factor(const Ctx<S>& c, double qi, double& dinv, double& gU, double& U, int& st,
                     double& z) {
  constexpr int N = S::N, NP = Ctx<S>::NP;
  constexpr int TRIG1 = (S::TRIG_ALL_INPUTS && S::NTRIG_F > 0) ? TRIG_REUSE : TRIG_FULL;
  const int li = c.li;
  cooperative_trig<S>(c, qi);
  {
    InJet1 qj{c.ga(), li};                                  // q lives in the gather buffer a
    SinkK<S, NP> sink;
    lds_sync();                                             // readers of the tile (L of the last solve) are done:
    sink.init(c.smem, c.off, c.offw, li, c.lw);             //   its first 4*NP doubles stage the rows of J
    TrigCache<S::NTRIG_U> tu;
    TrigLds tl = c.trig();
    const Jet1<1> u = S::template coords_sink_u<Jet1<1>, TRIG1>(qj, tl, tu, sink);
    gU = u.d[0]; U = u.v;
    sink.finish();
    sink.store(c.smem, c.offw, c.lw);                       // accumulators -> K in the packed triangles
  }
  lds_sync();
  if constexpr (!S::INERTIA_POS) {                          // K need not be definite: the reference's pivoted LU (solve_pivoted)
    z = solve_pivoted<S>(c, st, z);                         // z = v_li already; solve_back passes it through
    dinv = 1.0;
    return;
  }
#ifdef HAMK_PROBE_SKIP_FACTOR                              // timing probe (scripts/wave_probe.py): wrong results
  dinv = 1.0;
  return;
#endif
  factor_blocked<S>(c, dinv, st, z);
}

// Finish K v = rhs after `factor` (which left z = L^-1 rhs): D y = z, L^T v = y; returns v_li.
// Four unknowns per LDS round trip: the lanes exchange their partial sums through a buffer and each
// resolves the 4 x 4 triangle at the head of the block itself (six FMAs on broadcast reads of L).
template <class S>
HAMK_DEV double solve_back(const Ctx<S>& c, double dinv, double z) {
  constexpr int N = S::N, R = N & 3;
  auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
  const int li = c.li;
  if constexpr (!S::INERTIA_POS) return z;                  // solve_pivoted has done the whole solve
#ifdef HAMK_PROBE_SKIP_SOLVE
  return z * dinv;
#endif
  double v = z * dinv;                                     // D y = z
  double* cV = c.gb();
  const double* T = c.tile();
  if constexpr (Geo<N>::G == 1) {
    // one trajectory per wavefront: the unknowns already resolved travel through scalar registers (read_lane), L^T is read
    // from the packed triangle, which is final -- no exchange buffer, no fence between the blocks of four
#pragma unroll
    for (int k = N - 1; k >= N - R; --k) {
      const double vk = read_lane(v, k);
      if (li < k) v = fma(-T[tri(k, li)], vk, v);
    }
#pragma unroll
    for (int kb = N - R - 4; kb >= 0; kb -= 4) {
#ifndef HAMK_HOST_EMULATION
      // one block of four at a time: L^T is final, so nothing orders its ten reads per block against the chain of v -- left alone
      // the scheduler issues the reads of MANY blocks ahead (they hide latency) and pays for it in registers the kernel does not have
      __builtin_amdgcn_sched_barrier(0);
#endif
      const double p0 = read_lane(v, kb), p1 = read_lane(v, kb + 1), p2 = read_lane(v, kb + 2), v3 = read_lane(v, kb + 3);
      const double v2 = fma(-T[tri(kb + 3, kb + 2)], v3, p2);
      const double v1 = fma(-T[tri(kb + 2, kb + 1)], v2, fma(-T[tri(kb + 3, kb + 1)], v3, p1));
      const double v0 = fma(-T[tri(kb + 1, kb)], v1, fma(-T[tri(kb + 2, kb)], v2, fma(-T[tri(kb + 3, kb)], v3, p0)));
      if (li < kb) {
        v = fma(-T[tri(kb + 3, li)], v3, v);
        v = fma(-T[tri(kb + 2, li)], v2, v);
        v = fma(-T[tri(kb + 1, li)], v1, v);
        v = fma(-T[tri(kb, li)], v0, v);
      } else {
        if (li == kb) v = v0;
        if (li == kb + 1) v = v1;
        if (li == kb + 2) v = v2;
      }
    }
    return v;
  }
#pragma unroll
  for (int k = N - 1; k >= N - R; --k) {                   // the top N mod 4 unknowns one at a time
    HAMK_LOCKSTEP();
    cV[li] = v;
    lds_sync();
    const double vk = cV[k];
    if (li < k) v = fma(-T[tri(k, li)], vk, v);
  }
#pragma unroll
  for (int kb = N - R - 4; kb >= 0; kb -= 4) {
    HAMK_LOCKSTEP();
    cV[li] = v;
    lds_sync();
    const double p0 = cV[kb], p1 = cV[kb + 1], p2 = cV[kb + 2], v3 = cV[kb + 3];
    const double v2 = fma(-T[tri(kb + 3, kb + 2)], v3, p2);
    const double v1 = fma(-T[tri(kb + 2, kb + 1)], v2, fma(-T[tri(kb + 3, kb + 1)], v3, p1));
    const double v0 = fma(-T[tri(kb + 1, kb)], v1, fma(-T[tri(kb + 2, kb)], v2, fma(-T[tri(kb + 3, kb)], v3, p0)));
    if (li < kb) {
      v = fma(-T[tri(kb + 3, li)], v3, v);
      v = fma(-T[tri(kb + 2, li)], v2, v);
      v = fma(-T[tri(kb + 1, li)], v1, v);
      v = fma(-T[tri(kb, li)], v0, v);
    } else {
      if (li == kb) v = v0;
      if (li == kb + 1) v = v1;
      if (li == kb + 2) v = v2;
    }
  }
  return v;
}

// hamEqs for the group's trajectory: lane i returns (dq_i, dp_i).           Hamilton.hs:370-387
template <class S>
HAMK_DEV void ham_eqs(const Ctx<S>& c0, double qi, double pi, double& dqi, double& dpi, int& st) {
  constexpr int N = S::N;
  const Ctx<S> c = c0.launder();
  double dinv, gU, U;
  lds_sync();
  c.ga()[c.li] = qi;                                        // all-gather q through LDS; it stays there
  lds_sync();
  double z = pi;
  factor<S>(c, qi, dinv, gU, U, st, z);
  const double vi = solve_back<S>(c, dinv, z);
  lds_sync();
  c.gb()[c.li] = vi;                                        // ... and qd
  lds_sync();
  // Fresh opaque addresses for the second sweep: with provably identical LDS loads the compiler
  // merges the value/gradient parts of sweep 2 into sweep 1 and keeps ~150 registers alive across
  // the whole factorisation to save a few hundred cheap instructions.
  const Ctx<S> c2 = c.launder();
  InJet2 q2{c2.ga(), c2.gb(), c2.li};
  SinkT<S> sink;
  TrigLds tl = c2.trig();
#ifndef HAMK_PROBE_SKIP_SWEEP2
  S::template coords_sink<Jet2<1>, TRIG_REUSE>(q2, tl, sink);   // sincos pairs of sweep 1, from LDS
#endif
  dqi = vi;
  dpi = -(sink.dT + gU);
}

// ---- kernels ---------------------------------------------------------------------------------------
template <class S> struct Where {
  static constexpr int N = S::N, NP = Geo<N>::NP, G = Geo<N>::G, WAVES = Geo<N>::WAVES;
  i64 t;            // trajectory index (clamped to B-1 for the tail)
  bool real;        // the group's trajectory exists (not the padded tail of the last block)
  bool live;        // this lane owns a real (trajectory, coordinate)
  Ctx<S> c;
  HAMK_DEV Where(double* smem, i64 B) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = lane / NP;
    c.li = lane % NP;
    const i64 tt = ((i64)blockIdx.x * WAVES + wv) * G + grp;
    real = tt < B;
    live = real && (c.li < N);
    t = (tt < B) ? tt : B - 1;
    c.smem = smem;
    c.off = (wv * G + grp) * Lds<S>::PER_TRAJ;
    c.offw = wv * G * Lds<S>::PER_TRAJ;
    c.lw = lane;
    c.g4 = 4 * grp * NP;
  }
};

#define HAMK_WAVE_SMEM(S) __shared__ double smem[hamk::wave::Geo<S::N>::WAVES * hamk::wave::Geo<S::N>::G * hamk::wave::Lds<S>::PER_TRAJ]

template <class S> HAMK_DEV double velocity(const Ctx<S>& c0, double qi, double pi, double& U, int& st);

// hamiltonian of the group's trajectory (every lane returns it)
template <class S> HAMK_DEV double energy(const Ctx<S>& c, double qi, double pi, int& st) {
  constexpr int N = S::N, NP = Geo<N>::NP;
  double U;
  const double vi = velocity<S>(c, qi, pi, U, st);
  return fma(0.5, group_sum<NP>((c.li < N) ? vi * pi : 0.0), U);
}

// drift_tol: see hamk::rk4_body (ST_DRIFT when the launch loses its invariant)
template <class S>
HAMK_DEV void rk4_body(double* smem, double* q, double* p, i64 B, double dt, int nsteps, double drift_tol, int* status) {
  constexpr int N = S::N;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  double yq = q[(i64)j * B + w.t], yp = p[(i64)j * B + w.t];
  int st = 0;
  double H0 = 0.0;
  if (drift_tol > 0.0) H0 = energy<S>(w.c, yq, yp, st);
  const double h2 = 0.5 * dt, h6 = dt * (1.0 / 6.0), h3 = dt * (1.0 / 3.0);
  double kq = 0.0, kp = 0.0, aq = yq, ap = yp;
#pragma unroll 1
  for (int it = 0; it < 4 * nsteps; ++it) {
    const int sg = it & 3;
    const double a = (sg == 0) ? 0.0 : ((sg == 3) ? dt : h2);
    const double b = (sg == 0 || sg == 3) ? h6 : h3;
    const double tq = fma(a, kq, yq), tp = fma(a, kp, yp);
    ham_eqs<S>(w.c, tq, tp, kq, kp, st);
    aq = fma(b, kq, aq); ap = fma(b, kp, ap);
    if (sg == 3) { yq = aq; yp = ap; }
  }
  if (drift_tol > 0.0) {
    int st1 = 0;
    const double H1 = energy<S>(w.c, yq, yp, st1);
    const double lim = drift_tol * fmax(1.0, fabs(H0));
    if (!(fabs(H1 - H0) <= lim) || is_nonfinite_bits(H1)) st |= ST_DRIFT;
  }
  const bool bad = is_nonfinite_bits(yq) || is_nonfinite_bits(yp);
  if (bad && w.c.li < N) st |= ST_NONFINITE;
  // status of the trajectory = OR over its lanes
  int stg = st;
#pragma unroll
  for (int off = Where<S>::NP / 2; off > 0; off >>= 1) stg |= __shfl_xor(stg, off, Where<S>::NP);
  if (w.live) { q[(i64)j * B + w.t] = yq; p[(i64)j * B + w.t] = yp; }
  if (status && w.live && w.c.li == 0) status[w.t] = stg;
}

template <class S>
HAMK_DEV void hameqs_body(double* smem, const double* q, const double* p, double* dq, double* dp, i64 B, int* status) {
  constexpr int N = S::N;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  const double qi = q[(i64)j * B + w.t], pi = p[(i64)j * B + w.t];
  double a, b;
  int st = 0;
  ham_eqs<S>(w.c, qi, pi, a, b, st);
  if ((is_nonfinite_bits(a) || is_nonfinite_bits(b)) && w.c.li < N) st |= ST_NONFINITE;
  int stg = st;
#pragma unroll
  for (int off = Where<S>::NP / 2; off > 0; off >>= 1) stg |= __shfl_xor(stg, off, Where<S>::NP);
  if (w.live) { dq[(i64)j * B + w.t] = a; dp[(i64)j * B + w.t] = b; }
  if (status && w.live && w.c.li == 0) status[w.t] = stg;
}

// momenta / toPhase: one Jet2<1> sweep along qd                                  Hamilton.hs:262-284
template <class S>
HAMK_DEV double momentum(const Ctx<S>& c0, double qi, double vi) {
  const Ctx<S> c = c0.launder();
  constexpr int TRIG1 = (S::TRIG_ALL_INPUTS && S::NTRIG_F > 0) ? TRIG_REUSE : TRIG_FULL;
  lds_sync();
  c.ga()[c.li] = qi; c.gb()[c.li] = vi;
  lds_sync();
  cooperative_trig<S>(c, qi);
  InJet2 q2{c.ga(), c.gb(), c.li};
  SinkP<S> sink;
  TrigLds tl = c.trig();
  S::template coords_sink<Jet2<1>, TRIG1>(q2, tl, sink);
  return sink.p;
}

template <class S>
HAMK_DEV void to_phase_body(double* smem, const double* q, const double* qd, double* p, i64 B) {
  constexpr int N = S::N;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  const double pi = momentum<S>(w.c, q[(i64)j * B + w.t], qd[(i64)j * B + w.t]);
  if (w.live) p[(i64)j * B + w.t] = pi;
}

// velocities / fromPhase                                                          Hamilton.hs:316-337
template <class S>
HAMK_DEV double velocity(const Ctx<S>& c0, double qi, double pi, double& U, int& st) {
  const Ctx<S> c = c0.launder();
  constexpr int N = S::N, NP = Ctx<S>::NP;
  double dinv, gU;
  lds_sync();
  c.ga()[c.li] = qi;
  lds_sync();
  double z = pi;
  factor<S>(c, qi, dinv, gU, U, st, z);
  return solve_back<S>(c, dinv, z);
}

template <class S>
HAMK_DEV void from_phase_body(double* smem, const double* q, const double* p, double* qd, i64 B, int* status) {
  constexpr int N = S::N;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  int st = 0;
  double U;
  const double vi = velocity<S>(w.c, q[(i64)j * B + w.t], p[(i64)j * B + w.t], U, st);
  int stg = st;
#pragma unroll
  for (int off = Where<S>::NP / 2; off > 0; off >>= 1) stg |= __shfl_xor(stg, off, Where<S>::NP);
  if (w.live) qd[(i64)j * B + w.t] = vi;
  if (status && w.live && w.c.li == 0) status[w.t] = stg;
}

template <class S> HAMK_DEV double potential_only(const Ctx<S>& c, double qi) {
  constexpr int N = S::N, NP = Ctx<S>::NP;
  double q[N];
  allgather<N, NP>(c.ga(), c.li, qi, q);
  return potential_value<S>(q);
}

template <class S>
HAMK_DEV void observe_body(double* smem, const double* q, const double* p, double* ke, double* pe, double* h, i64 B,
                           int* status) {
  constexpr int N = S::N, NP = Geo<N>::NP;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  int st = 0;
  double U, t = 0.0;
  const double qi = q[(i64)j * B + w.t];
  if (ke || h) {
    const double pi = p[(i64)j * B + w.t];
    const double vi = velocity<S>(w.c, qi, pi, U, st);
    t = 0.5 * group_sum<NP>((w.c.li < N) ? vi * pi : 0.0);
  } else {
    U = potential_only<S>(w.c, qi);
  }
  int stg = st;
#pragma unroll
  for (int off = NP / 2; off > 0; off >>= 1) stg |= __shfl_xor(stg, off, NP);
  if (w.live && w.c.li == 0) {
    if (ke) ke[w.t] = t;
    if (pe) pe[w.t] = U;
    if (h) h[w.t] = t + U;
    if (status) status[w.t] = stg;
  }
}

template <class S>
HAMK_DEV void observe_config_body(double* smem, const double* q, const double* qd, double* ke, double* lag, i64 B) {
  constexpr int N = S::N, NP = Geo<N>::NP;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  const double qi = q[(i64)j * B + w.t], vi = qd[(i64)j * B + w.t];
  const double pi = momentum<S>(w.c, qi, vi);
  const double t = 0.5 * group_sum<NP>((w.c.li < N) ? vi * pi : 0.0);
  double U = 0.0;
  if (lag) U = potential_only<S>(w.c, qi);
  if (w.live && w.c.li == 0) {
    if (ke) ke[w.t] = t;
    if (lag) lag[w.t] = t - U;
  }
}

template <class S> HAMK_DEV void coords_body(double* smem, const double* q, double* x, i64 B) {
  constexpr int N = S::N, M = S::M, NP = Geo<N>::NP;
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  double qq[N], xx[M];
  allgather<N, NP>(w.c.ga(), w.c.li, q[(i64)j * B + w.t], qq);
  TrigCache<S::NTRIG_F> tc;
  S::template coords<double, TRIG_FULL>(qq, xx, tc);
  // every lane of the group holds all M outputs; lane li writes outputs li, li + NP, ...
#pragma unroll
  for (int k = 0; k < M; ++k)
    if (w.real && (k % NP) == w.c.li) x[(i64)k * B + w.t] = xx[k];
}


// evolveHam / stepHam on the wave path: the same GSL semantics as hamk::rkf45_body (rkf45.c,
// cstd.c with a_y = a_dydt = 1, evolve.c, gsl-ode.c; see hamk_device.hpp), one trajectory per
// lane group.  t, h and the accept/reject decision are uniform within a group (the error norm
// is a group-wide max by shuffles) but differ between the groups of a wavefront: every lane
// always executes the attempt -- the cooperative shuffles need all lanes -- and a group that has
// already reached the output time simply does not commit.  One inlined right-hand side serves
// the six evaluations of an attempt through a stage switch.
template <class S>
HAMK_DEV void rkf45_body(double* smem, const double* q0, const double* p0, double* qout, double* pout, i64 B, int nt,
                         const double* ts, double ts0, double ts1, double h0, double eps_abs, double eps_rel,
                         int flags, int max_sub, int* status, int* nsub, int ncalls, int it_every) {
  constexpr int N = S::N, NP = Geo<N>::NP;
  const int row0 = flags & 1, inplace = (flags >> 8) & 3, gsl_api = (flags >> 16) & 3;      // see hamk::rkf45_body
  const bool api2 = gsl_api != 1;
  const double sgn = (!api2 || h0 > 0.0) ? 1.0 : -1.0;    // odeiv2 driver.c: direction = sign of the initial step
  bool failed = false;                                    // odeiv2: GSL_FAILURE (uniform within a group)
  Where<S> w(smem, B);
  const int j = (w.c.li < N) ? w.c.li : 0;
  const bool mine = w.c.li < N;
  double yq = q0[(i64)j * B + w.t], yp = p0[(i64)j * B + w.t];
  if (row0 == 0 && w.live) { qout[(i64)j * B + w.t] = yq; pout[(i64)j * B + w.t] = yp; }
  int st = 0, attempts = 0;
  double t = ts ? ts[0] : ts0, h = h0;
  double fq, fp;
  // ncalls > 1: `iterate (stepHam dt)` in one launch -- see hamk::rkf45_body
#pragma unroll 1
  for (int call = 0; call < ncalls; ++call) {
  int budget = max_sub;
  if (call > 0) { t = ts ? ts[0] : ts0; h = h0; failed = false; }
  ham_eqs<S>(w.c, yq, yp, fq, fp, st);                    // dydt_in of EVERY call by the instructions a separate launch starts with
  for (int r = 1; r < nt; ++r) {
    const double ti = ts ? ts[r] : ts1;
    for (;;) {
      const bool active = (sgn * (ti - t) > 0.0) && (budget > 0) && !failed;
      if (!__any(active)) break;                           // wave-uniform exit
      const double dt = ti - t;
      double hh = h;
      bool final_step = false;
      if ((dt >= 0.0 && hh > dt) || (dt < 0.0 && hh < dt)) { hh = dt; final_step = true; }
      double k2q = 0, k2p = 0, k3q = 0, k3p = 0, k4q = 0, k4p = 0, k5q = 0, k5p = 0, k6q = 0, k6p = 0;
      double ynq = yq, ynp = yp, fnq = 0, fnp = 0;
#pragma unroll 1
      for (int sg = 0; sg < 6; ++sg) {
        double tq, tp;
        switch (sg) {
          case 0:
            tq = yq + (1.0 / 4.0) * hh * fq; tp = yp + (1.0 / 4.0) * hh * fp; break;
          case 1:
            tq = yq + hh * ((3.0 / 32.0) * fq + (9.0 / 32.0) * k2q);
            tp = yp + hh * ((3.0 / 32.0) * fp + (9.0 / 32.0) * k2p); break;
          case 2:
            tq = yq + hh * ((1932.0 / 2197.0) * fq + (-7200.0 / 2197.0) * k2q + (7296.0 / 2197.0) * k3q);
            tp = yp + hh * ((1932.0 / 2197.0) * fp + (-7200.0 / 2197.0) * k2p + (7296.0 / 2197.0) * k3p); break;
          case 3:
            tq = yq + hh * ((8341.0 / 4104.0) * fq + (-32832.0 / 4104.0) * k2q + (29440.0 / 4104.0) * k3q + (-845.0 / 4104.0) * k4q);
            tp = yp + hh * ((8341.0 / 4104.0) * fp + (-32832.0 / 4104.0) * k2p + (29440.0 / 4104.0) * k3p + (-845.0 / 4104.0) * k4p); break;
          case 4:
            tq = yq + hh * ((-6080.0 / 20520.0) * fq + (41040.0 / 20520.0) * k2q + (-28352.0 / 20520.0) * k3q +
                            (9295.0 / 20520.0) * k4q + (-5643.0 / 20520.0) * k5q);
            tp = yp + hh * ((-6080.0 / 20520.0) * fp + (41040.0 / 20520.0) * k2p + (-28352.0 / 20520.0) * k3p +
                            (9295.0 / 20520.0) * k4p + (-5643.0 / 20520.0) * k5p); break;
          default:
            ynq = yq + hh * ((902880.0 / 7618050.0) * fq + (3953664.0 / 7618050.0) * k3q + (3855735.0 / 7618050.0) * k4q +
                             (-1371249.0 / 7618050.0) * k5q + (277020.0 / 7618050.0) * k6q);
            ynp = yp + hh * ((902880.0 / 7618050.0) * fp + (3953664.0 / 7618050.0) * k3p + (3855735.0 / 7618050.0) * k4p +
                             (-1371249.0 / 7618050.0) * k5p + (277020.0 / 7618050.0) * k6p);
            tq = ynq; tp = ynp; break;
        }
        double oq, op;
        int st_try = 0;
        ham_eqs<S>(w.c, tq, tp, oq, op, st_try);
        if (active) st |= st_try;
        switch (sg) {
          case 0: k2q = oq; k2p = op; break;
          case 1: k3q = oq; k3p = op; break;
          case 2: k4q = oq; k4p = op; break;
          case 3: k5q = oq; k5p = op; break;
          case 4: k6q = oq; k6p = op; break;
          default: fnq = oq; fnp = op; break;
        }
      }
      // cstd.c: std_control_hadjust, ord = 5; the norm runs over the group's 2n components
      const double eq = hh * ((1.0 / 360.0) * fq + (-128.0 / 4275.0) * k3q + (-2197.0 / 75240.0) * k4q + (1.0 / 50.0) * k5q + (2.0 / 55.0) * k6q);
      const double ep = hh * ((1.0 / 360.0) * fp + (-128.0 / 4275.0) * k3p + (-2197.0 / 75240.0) * k4p + (1.0 / 50.0) * k5p + (2.0 / 55.0) * k6p);
      const double rq = fabs(eq) / fabs(eps_rel * (fabs(ynq) + fabs(hh * fnq)) + eps_abs);
      const double rp = fabs(ep) / fabs(eps_rel * (fabs(ynp) + fabs(hh * fnp)) + eps_abs);
      double rl = (rq > rp) ? rq : rp;
      if (!mine || !(rl > 2.2250738585072014e-308)) rl = 2.2250738585072014e-308;
      const double rmax = group_max<NP>(rl);
      const double tnew = final_step ? ti : t + hh;
      const double h_old = hh;
      bool reject = false, fail_now = false;
      if (rmax > 1.1) {
        double rr = 0.9 * rpow_inv<5>(rmax);
        if (rr < 0.2) rr = 0.2;
        const double hdec = rr * h_old;
        if (fabs(hdec) < fabs(h_old) && (tnew + hdec) != tnew) { reject = true; hh = hdec; }
        else if (api2) { fail_now = true; hh = hdec; }       // GSL_FAILURE; y and t stay advanced
      } else if (rmax < 0.5) {
        double rr = 0.9 * rpow_inv<6>(rmax);
        if (rr > 5.0) rr = 5.0;
        if (rr < 1.0) rr = 1.0;
        hh = rr * h_old;
      }
      if (active) {                                        // evolve.c: accept or undo
        ++attempts; --budget;
        if (fail_now) { failed = true; st |= ST_UNDERFLOW; }
        // the suggested step: always written back by gsl_odeiv; by gsl_odeiv2 not on a final step
        if (reject || fail_now || !api2 || !final_step) h = hh;
        if (!reject) {
          if (!(sgn * (tnew - t) > 0.0)) st |= ST_UNDERFLOW;
          t = tnew; yq = ynq; yp = ynp; fq = fnq; fp = fnp;
        }
      }
    }
    if (sgn * (ti - t) > 0.0 && !failed) st |= ST_MAXSTEPS;
    if (r >= row0 && w.live && call == ncalls - 1) {
      double* qo = (inplace == 2) ? const_cast<double*>(q0) : (inplace ? qout : qout + (i64)r * N * B);
      double* po = (inplace == 2) ? const_cast<double*>(p0) : (inplace ? pout : pout + (i64)r * N * B);
      qo[(i64)j * B + w.t] = yq; po[(i64)j * B + w.t] = yp;
    }
  }
  if (it_every > 0 && (call + 1) % it_every == 0 && w.live) {
    const i64 row = (i64)((call + 1) / it_every - 1);
    qout[(row * N + j) * B + w.t] = yq; pout[(row * N + j) * B + w.t] = yp;
  }
  }
  if ((is_nonfinite_bits(yq) || is_nonfinite_bits(yp)) && mine) st |= ST_NONFINITE;
  if (!mine) st = 0;
  int stg = st;
#pragma unroll
  for (int off = NP / 2; off > 0; off >>= 1) stg |= __shfl_xor(stg, off, NP);
  if (w.live && w.c.li == 0) {
    if (status) status[w.t] = stg;
    if (nsub) nsub[w.t] = attempts;
  }
}

}  // namespace wave
}  // namespace hamk

// Same eight kernel names as HAMK_INSTANTIATE, wave-cooperative bodies.
#ifndef HAMK_RK4_MIN_WAVES
#define HAMK_RK4_MIN_WAVES 2
#endif
#ifndef HAMK_RK4_MIN_WAVES_BIG                             // n > 32 (one trajectory per wavefront)
#define HAMK_RK4_MIN_WAVES_BIG 1
#endif
// n > 32 (one trajectory per wavefront): the adaptive stepper capped like the RK4 kernel -- two wavefronts per SIMD with
// spills instead of one with everything in registers (codegen defines HAMK_RKF_MIN_WAVES for those systems; round 3)
#ifdef HAMK_RKF_MIN_WAVES
#define HAMK_RKF_BOUNDS __launch_bounds__(256, HAMK_RKF_MIN_WAVES)
#else
#define HAMK_RKF_BOUNDS __launch_bounds__(256)
#endif
#define HAMK_INSTANTIATE_WAVE(S)                                                                                 \
  HAMK_SCRIBBLE_KERNEL                                                                                           \
  extern "C" __global__ void __launch_bounds__(256, (S::N > 32) ? HAMK_RK4_MIN_WAVES_BIG : HAMK_RK4_MIN_WAVES) hamk_rk4_steps_k(double* q, double* p, long long B, \
                                                          double dt, int nsteps, double drift_tol, int* status) { \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::rk4_body<S>(smem, q, p, B, dt, nsteps, drift_tol, status);                                       \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_hameqs_k(const double* q, const double* p, double* dq,  \
                                                                   double* dp, long long B, int* status) {       \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::hameqs_body<S>(smem, q, p, dq, dp, B, status);                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_coords_k(const double* q, double* x, long long B) {     \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::coords_body<S>(smem, q, x, B);                                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_to_phase_k(const double* q, const double* qd,           \
                                                                     double* p, long long B) {                   \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::to_phase_body<S>(smem, q, qd, p, B);                                                             \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_from_phase_k(const double* q, const double* p,          \
                                                                       double* qd, long long B, int* status) {   \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::from_phase_body<S>(smem, q, p, qd, B, status);                                                   \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_observe_k(const double* q, const double* p, double* ke, \
                                                                    double* pe, double* h, long long B,          \
                                                                    int* status) {                               \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::observe_body<S>(smem, q, p, ke, pe, h, B, status);                                               \
  }                                                                                                              \
  extern "C" __global__ void __launch_bounds__(256) hamk_observe_config_k(const double* q, const double* qd,     \
                                                                           double* ke, double* lag,              \
                                                                           long long B) {                        \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::observe_config_body<S>(smem, q, qd, ke, lag, B);                                                 \
  }                                                                                                              \
  extern "C" __global__ void HAMK_RKF_BOUNDS hamk_rkf45_k(                                                       \
      const double* q0, const double* p0, double* qout, double* pout, long long B, int nt, const double* ts,     \
      double ts0, double ts1, double h0, double eps_abs, double eps_rel, int flags, int max_sub,                 \
      int* status, int* nsub, int ncalls, int it_every) {                                                        \
    HAMK_WAVE_SMEM(S);                                                                                           \
    hamk::wave::rkf45_body<S>(smem, q0, p0, qout, pout, B, nt, ts, ts0, ts1, h0, eps_abs, eps_rel, flags,        \
                              max_sub, status, nsub, ncalls, it_every);                                          \
  }
