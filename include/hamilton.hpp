// hamilton.hpp -- header-only C++17 host mirror of `Numeric.Hamilton` over the C ABI of
// libhamk.so (include/hamk.h).
//
// The reference is compiled Haskell; its toolchain is absent from this image, so the host
// side above the C ABI is written in C++ with the reference's own names and argument
// meaning (src/Numeric/Hamilton.hs:28-70): mkSystem, mkSystem' (mkSystemP here), Config,
// Phase, toPhase, fromPhase, momenta, velocities, keC, keP, pe, lagrangian, hamiltonian,
// hamEqs, stepHam, evolveHam, evolveHam' (evolveHamL), stepHamC, evolveHamC, underlyingPos.
// The extension over the reference: a Config/Phase holds an ENSEMBLE (B trajectories,
// structure of arrays); B = 1 is the reference's case.
//
// User functions are written once against a generic number type `A` -- the C++ spelling of
// `forall a. RealFloat a => V.Vector n a -> V.Vector m a` (Hamilton.hs:212):
//
//     auto f = [](const std::vector<hamilton::Var>& q) {            // generic lambda works too
//       using hamilton::sin; using hamilton::cos;
//       return std::vector<hamilton::Var>{sin(q[0]), 0.5 - cos(q[0])};
//     };
//
// and are instantiated here at hamilton::Var, which records the expression tape that
// crosses the ABI (the reference instantiates at ad's Forward/Sparse/Reverse types,
// Hamilton.hs:220-224).
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <array>
#include <vector>

#include "hamk.h"

namespace hamilton {

// ---------------------------------------------------------------------------------------
// recording number type
// ---------------------------------------------------------------------------------------
class Tape {
 public:
  explicit Tape(int n_in) : n_in_(n_in) {}
  int32_t emit(int32_t op, int32_t a = 0, int32_t b = 0, double c = 0.0) {
    auto key = std::make_tuple(op, a, b, bits(c));
    auto it = memo_.find(key);
    if (it != memo_.end()) return it->second;           // hash-consing of identical subexpressions
    hamk_op o{op, a, b, 0, c};
    ops_.push_back(o);
    int32_t id = (int32_t)ops_.size() - 1;
    memo_[key] = id;
    return id;
  }
  bool is_const(int32_t id, double* c = nullptr) const {
    if (ops_[id].op != HAMK_OP_CONST) return false;
    if (c) *c = ops_[id].c;
    return true;
  }
  const std::vector<hamk_op>& ops() const { return ops_; }
  int n_in() const { return n_in_; }
  // The same recording in CANONICAL form: only the values the outputs depend on, numbered in
  // depth-first post-order from the outputs (first operand before second, outputs in order) -- a
  // function of the expression alone, whatever order a recorder emitted constants in.  Every host
  // shim ships this form: byte-identical tapes for the same function (tests/test_recorders.py).
  // `outs` is rewritten to the new ids.
  std::vector<hamk_op> canonical(std::vector<int32_t>& outs) const {
    auto kids = [](const hamk_op& o) {
      if (o.op == HAMK_OP_CONST || o.op == HAMK_OP_INPUT) return 0;
      return (o.op == HAMK_OP_ADD || o.op == HAMK_OP_SUB || o.op == HAMK_OP_MUL || o.op == HAMK_OP_DIV ||
              o.op == HAMK_OP_POW || o.op == HAMK_OP_ATAN2) ? 2 : 1;
    };
    std::vector<int32_t> new_id(ops_.size(), -1);
    std::vector<hamk_op> out;
    std::vector<std::pair<int32_t, int>> stack;
    for (int32_t root : outs) {
      stack.push_back({root, 0});
      while (!stack.empty()) {
        auto [node, phase] = stack.back();
        stack.pop_back();
        if (new_id[node] >= 0) continue;
        const hamk_op& o = ops_[node];
        const int nk = kids(o);
        if (phase == 0) {
          stack.push_back({node, 1});
          if (nk == 2 && new_id[o.b] < 0) stack.push_back({o.b, 0});
          if (nk >= 1 && new_id[o.a] < 0) stack.push_back({o.a, 0});
        } else {
          hamk_op c = o;
          if (nk >= 1) c.a = new_id[o.a];
          if (nk == 2) c.b = new_id[o.b];
          new_id[node] = (int32_t)out.size();
          out.push_back(c);
        }
      }
    }
    for (int32_t& r : outs) r = new_id[r];
    return out;
  }

 private:
  static uint64_t bits(double c) { uint64_t u; std::memcpy(&u, &c, 8); return u; }
  int n_in_;
  std::vector<hamk_op> ops_;
  std::map<std::tuple<int32_t, int32_t, int32_t, uint64_t>, int32_t> memo_;
};

class Var {
 public:
  Var() : tape_(nullptr), id_(-1), c_(0.0) {}
  Var(double c) : tape_(nullptr), id_(-1), c_(c) {}          // realToFrac / fromInteger (late-bound constant)
  Var(Tape* t, int32_t id) : tape_(t), id_(id), c_(0.0) {}
  Tape* tape() const { return tape_; }
  bool is_const(double* c = nullptr) const {
    if (!tape_) { if (c) *c = c_; return true; }
    return tape_->is_const(id_, c);
  }
  int32_t id_on(Tape* t) const {
    if (!tape_) return t->emit(HAMK_OP_CONST, 0, 0, c_);
    if (tape_ != t) throw std::logic_error("hamilton::Var: mixing values of two recordings");
    return id_;
  }

 private:
  Tape* tape_;
  int32_t id_;
  double c_;
};

namespace detail {
inline Tape* tape_of(const Var& a, const Var& b) {
  if (a.tape() && b.tape() && a.tape() != b.tape()) throw std::logic_error("hamilton::Var: mixing values of two recordings");
  return a.tape() ? a.tape() : b.tape();
}
inline double powi(double x, int k) {
  if (k < 0) return 1.0 / powi(x, -k);
  double r = 1.0, b = x;
  while (k) { if (k & 1) r *= b; b *= b; k >>= 1; }
  return r;
}
inline Var neg(const Var& a) {                      // -c folds, -(-x) = x
  double c;
  if (a.is_const(&c)) return Var(-c);
  Tape* t = a.tape();
  const hamk_op& o = t->ops()[a.id_on(t)];
  if (o.op == HAMK_OP_NEG) return Var(t, o.a);
  return Var(t, t->emit(HAMK_OP_NEG, a.id_on(t)));
}
inline Var binary(int32_t op, const Var& a, const Var& b) {
  double ca = 0.0, cb = 0.0;
  const bool ka = a.is_const(&ca), kb = b.is_const(&cb);
  if (ka && kb) {
    switch (op) {
      case HAMK_OP_ADD: return Var(ca + cb);
      case HAMK_OP_SUB: return Var(ca - cb);
      case HAMK_OP_MUL: return Var(ca * cb);
      default: return Var(ca / cb);
    }
  }
  Tape* t = tape_of(a, b);
  // exact identities only: never change a result bit
  if (op == HAMK_OP_ADD) { if (ka && ca == 0.0) return b; if (kb && cb == 0.0) return a; }
  if (op == HAMK_OP_SUB) { if (kb && cb == 0.0) return a; if (ka && ca == 0.0) return neg(b); }
  if (op == HAMK_OP_MUL) {
    if (ka && ca == 1.0) return b;
    if (kb && cb == 1.0) return a;
    if (ka && ca == -1.0) return neg(b);
    if (kb && cb == -1.0) return neg(a);
  }
  if (op == HAMK_OP_DIV) {
    if (kb && cb == 1.0) return a;
    if (ka && ca == 1.0) return Var(t, t->emit(HAMK_OP_RECIP, b.id_on(t)));
  }
  const int32_t ia = a.id_on(t), ib = b.id_on(t);   // operands in the order written: the tape is a function of
  return Var(t, t->emit(op, ia, ib));               // the expression alone (see canonical())
}
inline Var unary(int32_t op, const Var& x, double (*f)(double)) {
  double c;
  if (x.is_const(&c)) return Var(f(c));
  return Var(x.tape(), x.tape()->emit(op, x.id_on(x.tape())));
}
}  // namespace detail

inline Var operator+(const Var& a, const Var& b) { return detail::binary(HAMK_OP_ADD, a, b); }
inline Var operator-(const Var& a, const Var& b) { return detail::binary(HAMK_OP_SUB, a, b); }
inline Var operator*(const Var& a, const Var& b) { return detail::binary(HAMK_OP_MUL, a, b); }
inline Var operator/(const Var& a, const Var& b) { return detail::binary(HAMK_OP_DIV, a, b); }
inline Var operator-(const Var& a) { return detail::neg(a); }
inline Var operator+(const Var& a) { return a; }

#define HAMILTON_UNARY(name, OP) \
  inline Var name(const Var& x) { return detail::unary(OP, x, [](double v) { return std::name(v); }); }
HAMILTON_UNARY(sin, HAMK_OP_SIN)
HAMILTON_UNARY(cos, HAMK_OP_COS)
HAMILTON_UNARY(tan, HAMK_OP_TAN)
HAMILTON_UNARY(asin, HAMK_OP_ASIN)
HAMILTON_UNARY(acos, HAMK_OP_ACOS)
HAMILTON_UNARY(atan, HAMK_OP_ATAN)
HAMILTON_UNARY(sinh, HAMK_OP_SINH)
HAMILTON_UNARY(cosh, HAMK_OP_COSH)
HAMILTON_UNARY(tanh, HAMK_OP_TANH)
HAMILTON_UNARY(asinh, HAMK_OP_ASINH)
HAMILTON_UNARY(acosh, HAMK_OP_ACOSH)
HAMILTON_UNARY(atanh, HAMK_OP_ATANH)
HAMILTON_UNARY(exp, HAMK_OP_EXP)
HAMILTON_UNARY(log, HAMK_OP_LOG)
HAMILTON_UNARY(sqrt, HAMK_OP_SQRT)
#undef HAMILTON_UNARY
// Num's abs and signum (derivative of |x|: signum x, as `ad` has it)
inline Var abs(const Var& x) { return detail::unary(HAMK_OP_ABS, x, [](double v) { return std::fabs(v); }); }
inline Var signum(const Var& x) { return detail::unary(HAMK_OP_SIGNUM, x, [](double v) { return (double)((v > 0) - (v < 0)); }); }

// x ^ k (Haskell `^` / `^^`)
inline Var powi(const Var& x, int k) {
  double c;
  if (x.is_const(&c)) return Var(detail::powi(c, k));
  if (k == 0) return Var(1.0);
  if (k == 1) return x;
  return Var(x.tape(), x.tape()->emit(HAMK_OP_POWI, x.id_on(x.tape()), k));
}
// x ** y (Haskell `**`): constant integral exponents stay valid for negative bases (Examples.hs:154)
inline Var pow(const Var& x, const Var& y) {
  double cx, cy;
  const bool kx = x.is_const(&cx), ky = y.is_const(&cy);
  if (kx && ky) return Var(std::pow(cx, cy));
  if (ky) {
    if (cy == std::floor(cy) && std::fabs(cy) <= 64) return powi(x, (int)cy);
    return Var(x.tape(), x.tape()->emit(HAMK_OP_POWC, x.id_on(x.tape()), 0, cy));
  }
  Tape* t = detail::tape_of(x, y);
  return Var(t, t->emit(HAMK_OP_POW, x.id_on(t), y.id_on(t)));
}
inline Var atan2(const Var& y, const Var& x) {
  double cy, cx;
  if (y.is_const(&cy) && x.is_const(&cx)) return Var(std::atan2(cy, cx));
  Tape* t = detail::tape_of(y, x);
  return Var(t, t->emit(HAMK_OP_ATAN2, y.id_on(t), x.id_on(t)));
}

using VecFn = std::function<std::vector<Var>(const std::vector<Var>&)>;   // V.Vector n a -> V.Vector m a
using ScalarFn = std::function<Var(const std::vector<Var>&)>;              // V.Vector k a -> a

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
struct HamkError : std::runtime_error {
  int code;
  HamkError(int c, const std::string& m) : std::runtime_error("libhamk error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) { if (rc != HAMK_OK) throw HamkError(rc, hamk_last_error()); }

// ---------------------------------------------------------------------------------------
// states: SoA ensembles, host memory ([n][B])                      Hamilton.hs:103-145
// ---------------------------------------------------------------------------------------
struct Config { int n = 0; int64_t B = 0; std::vector<double> positions, velocities; };
struct Phase  { int n = 0; int64_t B = 0; std::vector<double> positions, momenta; };

inline Config Cfg(std::vector<double> q, std::vector<double> qd) {         // one trajectory
  Config c; c.n = (int)q.size(); c.B = 1; c.positions = std::move(q); c.velocities = std::move(qd); return c;
}
inline Phase Phs(std::vector<double> q, std::vector<double> p) {
  Phase s; s.n = (int)q.size(); s.B = 1; s.positions = std::move(q); s.momenta = std::move(p); return s;
}

// ---------------------------------------------------------------------------------------
// System                                                            Hamilton.hs:160-254
// ---------------------------------------------------------------------------------------
class System {
 public:
  // opt: the library's choices fixed by the caller (hamk.h hamk_options; nullptr = all defaults)
  System(int m, int n, const std::vector<double>& inertia, const VecFn& f, const ScalarFn& u, int u_space,
         const hamk_options* opt = nullptr)
      : m_(m), n_(n) {
    if ((int)inertia.size() != m) throw std::invalid_argument("inertia must have m entries");
    Tape tf(n), tu(u_space == HAMK_U_CARTESIAN ? m : n);
    f_outs_ = record(tf, n, [&](const std::vector<Var>& q) { return f(q); }, m);
    u_outs_ = record(tu, tu.n_in(), [&](const std::vector<Var>& z) { return std::vector<Var>{u(z)}; }, 1);
    f_ops_ = tf.canonical(f_outs_);
    u_ops_ = tu.canonical(u_outs_);
    hamk_system* h = nullptr;
    check(hamk_system_create_ex(m, n, inertia.data(), f_ops_.data(), (int32_t)f_ops_.size(), f_outs_.data(),
                                u_ops_.data(), (int32_t)u_ops_.size(), u_outs_[0], u_space, opt, &h));
    h_.reset(h, hamk_system_destroy);
  }
  // the tapes that crossed the ABI (canonical form)
  const std::vector<hamk_op>& f_tape() const { return f_ops_; }
  const std::vector<int32_t>& f_outs() const { return f_outs_; }
  const std::vector<hamk_op>& u_tape() const { return u_ops_; }
  const std::vector<int32_t>& u_outs() const { return u_outs_; }
  int m() const { return m_; }
  int n() const { return n_; }
  hamk_system* handle() const { return h_.get(); }
  // every choice resolved, for a launch over B trajectories (B < 0: a large ensemble)
  hamk_options options(int64_t B = -1) const { hamk_options o; check(hamk_system_get_options(h_.get(), B, &o)); return o; }
  std::string source() const { return hamk_system_source(h_.get()); }
  std::vector<int32_t> last_status;

 private:
  template <class F> static std::vector<int32_t> record(Tape& t, int n_in, F fn, int n_out) {
    std::vector<Var> in;
    for (int j = 0; j < n_in; ++j) in.emplace_back(&t, t.emit(HAMK_OP_INPUT, j));
    std::vector<Var> out = fn(in);
    if ((int)out.size() != n_out) throw std::invalid_argument("function returned the wrong number of values");
    std::vector<int32_t> ids;
    for (auto& v : out) ids.push_back(v.id_on(&t));
    return ids;
  }
  int m_, n_;
  std::vector<hamk_op> f_ops_, u_ops_;
  std::vector<int32_t> f_outs_, u_outs_;
  std::shared_ptr<hamk_system> h_;
};

// mkSystem: potential over generalized coordinates                   Hamilton.hs:201-225
inline System mkSystem(const std::vector<double>& inertia, int n, const VecFn& f, const ScalarFn& u, const hamk_options* opt = nullptr) {
  return System((int)inertia.size(), n, inertia, f, u, HAMK_U_GENERALIZED, opt);
}
// mkSystem': potential over the underlying cartesian coordinates      Hamilton.hs:238-254
inline System mkSystemP(const std::vector<double>& inertia, int n, const VecFn& f, const ScalarFn& u, const hamk_options* opt = nullptr) {
  return System((int)inertia.size(), n, inertia, f, u, HAMK_U_CARTESIAN, opt);
}

// ---------------------------------------------------------------------------------------
// state functions
// ---------------------------------------------------------------------------------------
inline std::vector<double> underlyingPos(const System& s, const std::vector<double>& q, int64_t B = 1) {   // :174-178
  std::vector<double> x((size_t)s.m() * B);
  check(hamk_coords_batch(s.handle(), B, q.data(), x.data(), HAMK_MEM_HOST));
  return x;
}
inline std::vector<double> momenta(const System& s, const Config& c) {                                     // :262-269
  std::vector<double> p((size_t)s.n() * c.B);
  check(hamk_to_phase_batch(s.handle(), c.B, c.positions.data(), c.velocities.data(), p.data(), HAMK_MEM_HOST));
  return p;
}
inline Phase toPhase(const System& s, const Config& c) {                                                   // :279-284
  Phase ph; ph.n = c.n; ph.B = c.B; ph.positions = c.positions; ph.momenta = momenta(s, c); return ph;
}
inline std::vector<double> velocities(System& s, const Phase& ph) {                                        // :316-324
  std::vector<double> v((size_t)s.n() * ph.B);
  s.last_status.assign((size_t)ph.B, 0);
  check(hamk_from_phase_batch(s.handle(), ph.B, ph.positions.data(), ph.momenta.data(), v.data(), s.last_status.data(), HAMK_MEM_HOST));
  return v;
}
inline Config fromPhase(System& s, const Phase& ph) {                                                      // :332-337
  Config c; c.n = ph.n; c.B = ph.B; c.positions = ph.positions; c.velocities = velocities(s, ph); return c;
}
namespace detail {
inline std::vector<double> observe(System& s, const Phase& ph, int which) {
  std::vector<double> out((size_t)ph.B);
  s.last_status.assign((size_t)ph.B, 0);
  check(hamk_observe_batch(s.handle(), ph.B, ph.positions.data(), which == 1 ? nullptr : ph.momenta.data(),
                           which == 0 ? out.data() : nullptr, which == 1 ? out.data() : nullptr,
                           which == 2 ? out.data() : nullptr, s.last_status.data(), HAMK_MEM_HOST));
  return out;
}
inline std::vector<double> observe_config(const System& s, const Config& c, int which) {
  std::vector<double> out((size_t)c.B);
  check(hamk_observe_config_batch(s.handle(), c.B, c.positions.data(), c.velocities.data(),
                                  which == 0 ? out.data() : nullptr, which == 1 ? out.data() : nullptr, HAMK_MEM_HOST));
  return out;
}
}  // namespace detail
inline std::vector<double> keP(System& s, const Phase& ph) { return detail::observe(s, ph, 0); }            // :341-349
inline std::vector<double> pe(System& s, const std::vector<double>& q, int64_t B = 1) {                    // :182-186
  Phase ph; ph.n = s.n(); ph.B = B; ph.positions = q; return detail::observe(s, ph, 1);
}
inline std::vector<double> hamiltonian(System& s, const Phase& ph) { return detail::observe(s, ph, 2); }    // :353-361
inline std::vector<double> keC(const System& s, const Config& c) { return detail::observe_config(s, c, 0); }        // :288-296
inline std::vector<double> lagrangian(const System& s, const Config& c) { return detail::observe_config(s, c, 1); } // :301-309

// hamEqs: (dH/dp, -dH/dq)                                                                                  :370-387
inline std::pair<std::vector<double>, std::vector<double>> hamEqs(System& s, const Phase& ph) {
  std::vector<double> dq((size_t)s.n() * ph.B), dp((size_t)s.n() * ph.B);
  s.last_status.assign((size_t)ph.B, 0);
  check(hamk_hameqs_batch(s.handle(), ph.B, ph.positions.data(), ph.momenta.data(), dq.data(), dp.data(),
                          s.last_status.data(), HAMK_MEM_HOST));
  return {dq, dp};
}

// ---------------------------------------------------------------------------------------
// time stepping
// ---------------------------------------------------------------------------------------
inline Phase stepHam(double r, System& s, const Phase& ph) {                                               // :390-402
  Phase out = ph;
  s.last_status.assign((size_t)ph.B, 0);
  check(hamk_step_ham_batch(s.handle(), ph.B, out.positions.data(), out.momenta.data(), r, s.last_status.data(), nullptr, HAMK_MEM_HOST));
  return out;
}
// `iterate (stepHam r s)` (README.md:150, app/Examples.hs:429): ncalls consecutive stepHam r in one launch, bit-identical to
// ncalls calls of stepHam; frames (optional) receives the Phase after every `every`-th call.
inline Phase iterateStepHam(double r, int ncalls, System& s, const Phase& ph, int every = 0, std::vector<Phase>* frames = nullptr) {
  Phase out = ph;
  s.last_status.assign((size_t)ph.B, 0);
  const size_t cnt = (size_t)s.n() * ph.B;
  const size_t rows = (every > 0 && frames) ? (size_t)(ncalls / every) : 0;
  std::vector<double> fq(cnt * rows), fp(cnt * rows);
  check(hamk_step_ham_iterate(s.handle(), ph.B, out.positions.data(), out.momenta.data(), r, ncalls, rows ? every : 0,
                              rows ? fq.data() : nullptr, rows ? fp.data() : nullptr, s.last_status.data(), nullptr, HAMK_MEM_HOST));
  if (frames) {
    frames->assign(rows, Phase{});
    for (size_t k = 0; k < rows; ++k) {
      (*frames)[k].n = ph.n; (*frames)[k].B = ph.B;
      (*frames)[k].positions.assign(fq.begin() + k * cnt, fq.begin() + (k + 1) * cnt);
      (*frames)[k].momenta.assign(fp.begin() + k * cnt, fp.begin() + (k + 1) * cnt);
    }
  }
  return out;
}
inline std::vector<Phase> evolveHam(System& s, const Phase& p0, const std::vector<double>& ts) {           // :433-462
  if (ts.size() < 2) throw std::invalid_argument("evolveHam needs at least two solution times (2 <= s)");
  const size_t cnt = (size_t)s.n() * p0.B;
  std::vector<double> qo(cnt * ts.size()), po(cnt * ts.size());
  s.last_status.assign((size_t)p0.B, 0);
  check(hamk_evolve_ham_batch(s.handle(), p0.B, p0.positions.data(), p0.momenta.data(), (int32_t)ts.size(), ts.data(),
                              qo.data(), po.data(), 0.0, 0.0, 0.0, s.last_status.data(), nullptr, HAMK_MEM_HOST));
  std::vector<Phase> rows(ts.size());
  for (size_t r = 0; r < ts.size(); ++r) {
    rows[r].n = p0.n; rows[r].B = p0.B;
    rows[r].positions.assign(qo.begin() + r * cnt, qo.begin() + (r + 1) * cnt);
    rows[r].momenta.assign(po.begin() + r * cnt, po.begin() + (r + 1) * cnt);
  }
  return rows;
}
// evolveHam' (list front-end): [] -> []; [x] -> evolve over [0, x], drop the first        :409-429
inline std::vector<Phase> evolveHamL(System& s, const Phase& p0, const std::vector<double>& ts) {
  if (ts.empty()) return {};
  if (ts.size() == 1) { auto rows = evolveHam(s, p0, {0.0, ts[0]}); rows.erase(rows.begin()); return rows; }
  return evolveHam(s, p0, ts);
}
inline Config stepHamC(double r, System& s, const Config& c) { return fromPhase(s, stepHam(r, s, toPhase(s, c))); }   // :502-515
inline std::vector<Config> evolveHamC(System& s, const Config& c0, const std::vector<double>& ts) {                    // :486-500
  std::vector<Config> out;
  for (auto& ph : evolveHam(s, toPhase(s, c0), ts)) out.push_back(fromPhase(s, ph));
  return out;
}
// classic fixed-step RK4 (BASELINE.json north_star; no reference counterpart)
inline Phase rk4Steps(double dt, int nsteps, System& s, const Phase& ph) {
  Phase out = ph;
  s.last_status.assign((size_t)ph.B, 0);
  check(hamk_rk4_steps(s.handle(), ph.B, out.positions.data(), out.momenta.data(), dt, nsteps, s.last_status.data(), HAMK_MEM_HOST));
  return out;
}

// ---------------------------------------------------------------------------------------
// Ensembles resident in HBM (no reference counterpart: the reference holds one trajectory on the
// Haskell heap).  DevicePhase owns SoA device arrays on the device that was current when it was
// made; the steppers advance it in place, asynchronously on the System's stream.  One System per
// device + hamk_set_device + gather() is the single-process form of the node-level sharding.
// ---------------------------------------------------------------------------------------
class DeviceArray {
 public:
  DeviceArray() = default;
  explicit DeviceArray(int64_t bytes) : bytes_(bytes) { void* p = nullptr; check(hamk_device_malloc(&p, bytes)); p_ = p; }
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  DeviceArray(DeviceArray&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  DeviceArray& operator=(DeviceArray&& o) noexcept { if (this != &o) { release(); p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.bytes_ = 0; } return *this; }
  ~DeviceArray() { release(); }
  template <class T> T* as() const { return static_cast<T*>(p_); }
  int64_t bytes() const { return bytes_; }
  void upload(const void* host) { check(hamk_memcpy(p_, host, bytes_, HAMK_COPY_H2D)); }
  void download(void* host) const { check(hamk_memcpy(host, p_, bytes_, HAMK_COPY_D2H)); }
 private:
  void release() { if (p_) hamk_device_free(p_); p_ = nullptr; }
  void* p_ = nullptr;
  int64_t bytes_ = 0;
};

struct DevicePhase {
  int n = 0; int64_t B = 0;
  DeviceArray positions, momenta, status;
  DevicePhase() = default;
  DevicePhase(int n_, int64_t B_) : n(n_), B(B_), positions(8 * (int64_t)n_ * B_), momenta(8 * (int64_t)n_ * B_), status(4 * B_) {}
  explicit DevicePhase(const Phase& h) : DevicePhase(h.n, h.B) { positions.upload(h.positions.data()); momenta.upload(h.momenta.data()); }
  Phase download() const {
    Phase h; h.n = n; h.B = B; h.positions.resize((size_t)n * B); h.momenta.resize((size_t)n * B);
    positions.download(h.positions.data()); momenta.download(h.momenta.data());
    return h;
  }
  std::vector<int32_t> download_status() const { std::vector<int32_t> st((size_t)B); status.download(st.data()); return st; }
};

// toPhase on the device: q, qd uploaded, p computed there                              :279-284
inline DevicePhase toPhaseDevice(const System& s, const Config& c) {
  DevicePhase d(c.n, c.B);
  DeviceArray qd(8 * (int64_t)c.n * c.B);
  d.positions.upload(c.positions.data()); qd.upload(c.velocities.data());
  check(hamk_to_phase_batch(s.handle(), c.B, d.positions.as<double>(), qd.as<double>(), d.momenta.as<double>(), HAMK_MEM_DEVICE));
  check(hamk_synchronize(s.handle()));                      // qd is freed on return
  return d;
}
// Initial Configs of trajectories first_index .. first_index + B - 1 drawn ON THE DEVICE from the global trajectory index
// (hamk_sample_batch), then toPhase there: a shard of an ensemble without a host array or a scatter.
// box: n (lo, hi) pairs each for positions and velocities.
struct Box { std::vector<double> q_lo, q_hi, qd_lo, qd_hi; };
inline DevicePhase samplePhaseDevice(const System& s, const Box& box, int64_t first_index, int64_t B, uint64_t seed) {
  int32_t m = 0, n = 0;
  check(hamk_system_dims(s.handle(), &m, &n));
  if ((int)box.q_lo.size() != n || (int)box.q_hi.size() != n || (int)box.qd_lo.size() != n || (int)box.qd_hi.size() != n)
    throw std::invalid_argument("samplePhaseDevice: the box needs n (lo, hi) pairs");
  DevicePhase d(n, B);
  DeviceArray qd(8 * (int64_t)n * B);
  check(hamk_sample_batch(s.handle(), B, first_index, seed, box.q_lo.data(), box.q_hi.data(), box.qd_lo.data(), box.qd_hi.data(),
                          d.positions.as<double>(), qd.as<double>(), HAMK_MEM_DEVICE));
  check(hamk_to_phase_batch(s.handle(), B, d.positions.as<double>(), qd.as<double>(), d.momenta.as<double>(), HAMK_MEM_DEVICE));
  check(hamk_synchronize(s.handle()));                      // qd is freed on return
  return d;
}
// The size of the WHOLE ensemble this System's launches are pieces of (hamk_system_set_ensemble_size): every shard on the
// mapping chosen for the whole, so any shard layout reproduces the one-launch bits.
inline void setEnsembleSize(System& s, int64_t B_total) { check(hamk_system_set_ensemble_size(s.handle(), B_total)); }
inline void rk4Steps(double dt, int nsteps, System& s, DevicePhase& d) {      // in place, asynchronous
  check(hamk_rk4_steps(s.handle(), d.B, d.positions.as<double>(), d.momenta.as<double>(), dt, nsteps, d.status.as<int32_t>(), HAMK_MEM_DEVICE));
}
inline void stepHam(double r, System& s, DevicePhase& d) {                     // :390-402, in place, asynchronous
  check(hamk_step_ham_batch(s.handle(), d.B, d.positions.as<double>(), d.momenta.as<double>(), r, d.status.as<int32_t>(), nullptr, HAMK_MEM_DEVICE));
}
inline void iterateStepHam(double r, int ncalls, System& s, DevicePhase& d) {  // README.md:150; in place, asynchronous, one launch
  check(hamk_step_ham_iterate(s.handle(), d.B, d.positions.as<double>(), d.momenta.as<double>(), r, ncalls, 0, nullptr, nullptr,
                              d.status.as<int32_t>(), nullptr, HAMK_MEM_DEVICE));
}
// ... with the launch checking its own energy invariant: HAMK_ST_DRIFT in d.status where
// |H_exit - H_entry| > drift_tol * max(1, |H_entry|)   (hamk_rk4_steps_checked)
inline void rk4StepsChecked(double dt, int nsteps, double drift_tol, System& s, DevicePhase& d) {
  check(hamk_rk4_steps_checked(s.handle(), d.B, d.positions.as<double>(), d.momenta.as<double>(), dt, nsteps, drift_tol, d.status.as<int32_t>(), HAMK_MEM_DEVICE));
}
inline void synchronize(const System& s) { check(hamk_synchronize(s.handle())); }

// Which binding of hmatrix-gsl's gsl-ode.c stepHam / evolveHam follow (hamk.h): 2 = gsl_odeiv2
// driver (default), 1 = old gsl_odeiv (-DGSLODE1)
inline void setGslApi(System& s, int api) { check(hamk_system_set_gsl_api(s.handle(), api)); }
inline int gslApi(const System& s) { return hamk_system_get_gsl_api(s.handle()); }

// Ensemble checkpoint / resume: the device-resident state to one flat file
// and back; a resumed run continues bit-identically (hamk.h)
struct CheckpointInfo { int32_t n = 0; int64_t B = 0, steps_done = 0; uint64_t seed = 0; double t = 0.0; };
inline void saveCheckpoint(const std::string& path, const System& s, const DevicePhase& d, int64_t steps_done, uint64_t seed, double t) {
  check(hamk_synchronize(s.handle()));
  check(hamk_checkpoint_write(path.c_str(), d.n, d.B, d.positions.as<double>(), d.momenta.as<double>(), HAMK_MEM_DEVICE, steps_done, seed, t));
}
inline void saveCheckpoint(const std::string& path, const Phase& h, int64_t steps_done, uint64_t seed, double t) {
  check(hamk_checkpoint_write(path.c_str(), h.n, h.B, h.positions.data(), h.momenta.data(), HAMK_MEM_HOST, steps_done, seed, t));
}
inline CheckpointInfo checkpointInfo(const std::string& path) {
  CheckpointInfo i;
  check(hamk_checkpoint_info(path.c_str(), &i.n, &i.B, &i.steps_done, &i.seed, &i.t));
  return i;
}
inline DevicePhase loadCheckpointDevice(const std::string& path, CheckpointInfo* info = nullptr) {
  const CheckpointInfo i = checkpointInfo(path);
  DevicePhase d(i.n, i.B);
  check(hamk_checkpoint_read(path.c_str(), i.n, i.B, d.positions.as<double>(), d.momenta.as<double>(), HAMK_MEM_DEVICE));
  if (info) *info = i;
  return d;
}
inline Phase loadCheckpoint(const std::string& path, CheckpointInfo* info = nullptr) {
  const CheckpointInfo i = checkpointInfo(path);
  Phase h; h.n = i.n; h.B = i.B; h.positions.resize((size_t)i.n * i.B); h.momenta.resize((size_t)i.n * i.B);
  check(hamk_checkpoint_read(path.c_str(), i.n, i.B, h.positions.data(), h.momenta.data(), HAMK_MEM_HOST));
  if (info) *info = i;
  return h;
}
inline std::vector<double> hamiltonian(System& s, const DevicePhase& d) {      // :353-361
  DeviceArray h(8 * d.B), st(4 * d.B);
  check(hamk_observe_batch(s.handle(), d.B, d.positions.as<double>(), d.momenta.as<double>(), nullptr, nullptr, h.as<double>(), st.as<int32_t>(), HAMK_MEM_DEVICE));
  check(hamk_synchronize(s.handle()));
  std::vector<double> out((size_t)d.B); h.download(out.data());
  return out;
}
// final gather of shards (possibly on several devices) into one host ensemble, part order;
// synchronize() the Systems that advance the shards first
inline Phase gather(const std::vector<const DevicePhase*>& parts) {
  Phase out;
  if (parts.empty()) return out;
  out.n = parts[0]->n;
  std::vector<int64_t> Bs; std::vector<const double*> qs, ps;
  for (auto* d : parts) { Bs.push_back(d->B); qs.push_back(d->positions.as<double>()); ps.push_back(d->momenta.as<double>()); out.B += d->B; }
  out.positions.resize((size_t)out.n * out.B); out.momenta.resize((size_t)out.n * out.B);
  check(hamk_gather_batch((int32_t)parts.size(), out.n, Bs.data(), qs.data(), out.positions.data(), HAMK_MEM_HOST));
  check(hamk_gather_batch((int32_t)parts.size(), out.n, Bs.data(), ps.data(), out.momenta.data(), HAMK_MEM_HOST));
  return out;
}

// The same final gather for a host that runs ONE PROCESS PER GPU: an RCCL communicator through the C ABI (hamk_comm_*).  Rank 0 draws the
// id and ships its 128 bytes to the other processes by whatever channel the launcher offers; every process selects its device first.
class Comm {
 public:
  using Id = std::array<char, HAMK_COMM_ID_BYTES>;
  static Id uniqueId() { Id id; check(hamk_comm_unique_id(id.data())); return id; }
  Comm(const Id& id, int world, int rank) : world_(world), rank_(rank) { check(hamk_comm_create(id.data(), world, rank, &c_)); }
  Comm(const Comm&) = delete;
  Comm& operator=(const Comm&) = delete;
  ~Comm() { hamk_comm_destroy(c_); }
  int world() const { return world_; }
  int rank() const { return rank_; }
  // every rank's shard, rank order, on this rank's device; Bs[g] = trajectories of rank g (the same vector on all ranks);
  // synchronize() the System that advanced `mine` first
  DevicePhase allGather(const DevicePhase& mine, const std::vector<int64_t>& Bs) const {
    if ((int)Bs.size() != world_ || Bs[(size_t)rank_] != mine.B) throw HamkError(HAMK_ERR_INVALID, "Comm::allGather: Bs must hold one size per rank, Bs[rank] = mine.B");
    int64_t total = 0;
    for (int64_t b : Bs) total += b;
    DevicePhase all(mine.n, total);
    check(hamk_comm_allgather_batch(c_, mine.n, Bs.data(), mine.positions.as<double>(), all.positions.as<double>()));
    check(hamk_comm_allgather_batch(c_, mine.n, Bs.data(), mine.momenta.as<double>(), all.momenta.as<double>()));
    return all;
  }
  // one double per rank (timings, counts), rank order, on the host
  std::vector<double> allGatherScalar(double x) const {
    DeviceArray mine(8), all(8 * (int64_t)world_);
    mine.upload(&x);
    const std::vector<int64_t> ones((size_t)world_, 1);
    check(hamk_comm_allgather_batch(c_, 1, ones.data(), mine.as<double>(), all.as<double>()));
    std::vector<double> out((size_t)world_);
    all.download(out.data());
    return out;
  }
 private:
  hamk_comm* c_ = nullptr;
  int world_ = 0, rank_ = 0;
};

}  // namespace hamilton
