// hamk_codegen.cpp -- expression tape -> system struct for the device library.
//
// The reference builds a `System` by instantiating the user's polymorphic
// functions at ad's number types (mkSystem, Hamilton.hs:217-225).  Here the
// functions arrive already recorded (hamk_op[], include/hamk.h) and are
// re-emitted as C++ member templates generic over the number type, so that
// hamk_device.hpp can instantiate them at double / Jet1 / JetH / Jet2.  Only
// the user's f and U are generated; every kernel around them is hand-written.
#include "hamk_internal.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <sstream>

namespace hamk_host {

static std::string lit(double c) {
  if (std::isnan(c)) return "hamk::quiet_nan()";
  if (std::isinf(c)) return c > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%a", c);       // hex float: exact round trip
  std::string s(buf);
  if (c < 0 || std::signbit(c)) s = "(" + s + ")";
  return s;
}

static const char* unary_name(int op) {
  switch (op) {
    case HAMK_OP_RECIP: return "recip";
    case HAMK_OP_SIN: return "sin";
    case HAMK_OP_COS: return "cos";
    case HAMK_OP_TAN: return "tan";
    case HAMK_OP_ASIN: return "asin";
    case HAMK_OP_ACOS: return "acos";
    case HAMK_OP_ATAN: return "atan";
    case HAMK_OP_SINH: return "sinh";
    case HAMK_OP_COSH: return "cosh";
    case HAMK_OP_TANH: return "tanh";
    case HAMK_OP_ASINH: return "asinh";
    case HAMK_OP_ACOSH: return "acosh";
    case HAMK_OP_ATANH: return "atanh";
    case HAMK_OP_EXP: return "exp";
    case HAMK_OP_LOG: return "log";
    case HAMK_OP_SQRT: return "sqrt";
    case HAMK_OP_ABS: return "abs";
    case HAMK_OP_SIGNUM: return "signum";
    default: return nullptr;
  }
}

static bool is_binary(int op) {
  return op == HAMK_OP_ADD || op == HAMK_OP_SUB || op == HAMK_OP_MUL || op == HAMK_OP_DIV || op == HAMK_OP_POW ||
         op == HAMK_OP_ATAN2;
}

std::string validate_tape(const hamk_op* ops, int nops, int n_in, const int32_t* outs, int n_out, const char* what) {
  std::ostringstream e;
  if (nops < 0 || (nops > 0 && !ops)) { e << what << ": null tape"; return e.str(); }
  for (int i = 0; i < nops; ++i) {
    const hamk_op& o = ops[i];
    if (o.op < 0 || o.op >= HAMK_OP__COUNT) { e << what << ": op " << i << " has unknown opcode " << o.op; return e.str(); }
    if (o.op == HAMK_OP_CONST) continue;
    if (o.op == HAMK_OP_INPUT) {
      if (o.a < 0 || o.a >= n_in) { e << what << ": op " << i << " reads input " << o.a << " of " << n_in; return e.str(); }
      continue;
    }
    if (o.a < 0 || o.a >= i) { e << what << ": op " << i << " operand a=" << o.a << " is not an earlier value"; return e.str(); }
    if (is_binary(o.op) && (o.b < 0 || o.b >= i)) {
      e << what << ": op " << i << " operand b=" << o.b << " is not an earlier value"; return e.str();
    }
    if (o.op == HAMK_OP_POWI && (o.b > 4096 || o.b < -4096)) { e << what << ": op " << i << " exponent out of range"; return e.str(); }
  }
  for (int k = 0; k < n_out; ++k)
    if (outs[k] < 0 || outs[k] >= nops) { e << what << ": output " << k << " refers to value " << outs[k]; return e.str(); }
  return std::string();
}

// Emit the body of one generic function.  Values are `const auto vI`; constants
// stay plain doubles so the jet overloads never multiply by a lifted zero jet.
// Returns the number of trig-cache slots (one per distinct sincos operand) the body uses.
// sink_outs (optional): value id -> list of output slots; each output is handed to
// `sink.template put<K, SEQ>(value)` right after the op that defines it (SEQ = 0, 1, ... in the
// order of emission), so a consumer that only accumulates never keeps all M outputs live (wave
// kernels, M up to 64).
// input_exprs (optional): expression to use for INPUT j instead of in[j] (composition u . f);
// tc_name: the trig cache variable the body's sincos sites use.
// Values of the tape being emitted that somebody outside emit_body reads by name (the tape's outputs): set by generate_source
// before every call.  A value in it is always materialised.
static std::vector<char> g_emit_is_output;
// Dense maps on the four-lane kernels (SystemDesc::quad_dense): products `constant x value` that SEVERAL outputs read are not
// materialised once but written out at every use with the constant behind an opaque copy -- a tape whose n^2 coefficients come from
// a small table (the benchmark maps dense24 / dense32: 11 x 7 values) shares each product a sin q_j between ~n / 11 outputs, and a
// shared value lives from its first reader to its last: hundreds of them at once, i.e. scratch (dense32: 3 666 spilled registers, the
// library went back to the wave kernels).  Unshared, such a tape compiles like one with distinct coefficients.
static bool g_emit_unshare_scales = false;
static void mark_outputs(size_t nops, const int32_t* outs, int n_out) {
  g_emit_is_output.assign(nops, 0);
  for (int k = 0; k < n_out; ++k) g_emit_is_output[(size_t)outs[k]] = 1;
}
static int emit_body(std::ostringstream& o, const hamk_op* ops, int nops, const char* pfx,
                     const std::vector<std::vector<int>>* sink_outs = nullptr,
                     const std::vector<std::string>* input_exprs = nullptr, const char* tc_name = "tc",
                     std::vector<int>* slot_operand = nullptr, const char* trig_mode = "TRIG",
                     const std::vector<int>* shared_f_slot = nullptr, std::vector<int>* put_order = nullptr) {
  // pair SIN/COS of a shared operand: one sincos
  std::vector<int> sin_of(nops, -1), cos_of(nops, -1);
  for (int i = 0; i < nops; ++i) {
    if (ops[i].op == HAMK_OP_SIN && sin_of[ops[i].a] < 0) sin_of[ops[i].a] = i;
    if (ops[i].op == HAMK_OP_COS && cos_of[ops[i].a] < 0) cos_of[ops[i].a] = i;
  }
  std::vector<char> done(nops, 0);
  // 1 / sqrt(x) -- every inverse distance of a gravitational or electrostatic potential -- as ONE chain through hamk::rsqrt_of
  // (v_rsq_f64 + one third-order step, derivatives -r^3 / 2 and 3 r^5 / 4 from the same r) instead of a correctly rounded sqrt,
  // its derivative rule, a reciprocal and ITS rule: where the square root has no other reader and is not an output.  Round 6:
  // threeBodyPolar 1 398 -> see DESIGN section 4 instructions per RK4 step.
  std::vector<int> uses(nops, 0), fused_rsqrt(nops, 0);
  for (int i = 0; i < nops; ++i) {
    const hamk_op& p = ops[i];
    if (p.op == HAMK_OP_CONST || p.op == HAMK_OP_INPUT) continue;
    ++uses[p.a];
    if (is_binary(p.op)) ++uses[p.b];
  }
  for (int i = 0; i < nops; ++i)
    if (ops[i].op == HAMK_OP_RECIP && ops[ops[i].a].op == HAMK_OP_SQRT && uses[ops[i].a] == 1 &&
        !((size_t)ops[i].a < g_emit_is_output.size() && g_emit_is_output[(size_t)ops[i].a]) && !(sink_outs && !(*sink_outs)[ops[i].a].empty())) {
      fused_rsqrt[i] = 1;
      done[ops[i].a] = 2;                                   // the square root itself is never emitted
    }
  // exp(a x + b2) = exp(a x + b1) * exp(b2 - b1): two exponentials whose arguments are the SAME affine function of the same tape value
  // up to a constant offset -- the two walls of a rail (Examples.hs:155-156: logistic(-1.5, ...) r and logistic(1.5, ...) r), the four walls
  // of a room -- share one evaluation; the second is the first times a constant (its jet follows by scaling: the chain rule is exact).
  // ~35 instructions per evaluation saved; the product carries the first exponential's rounding plus the constant's (2 ulp), and
  // overflows / underflows where the direct evaluation would be within e^+-700 of doing so (offsets beyond 600 are left alone).
  struct Affine { int base; double slope, off; };
  std::vector<Affine> aff((size_t)nops);
  std::vector<int> exp_of(nops, -1);                       // exp node -> the earlier exp node it is a constant multiple of
  std::vector<double> exp_factor(nops, 1.0);
  {
    auto isc = [&](int i) { return ops[i].op == HAMK_OP_CONST; };
    for (int i = 0; i < nops; ++i) {
      const hamk_op& p = ops[i];
      Affine a{i, 1.0, 0.0};
      switch (p.op) {
        case HAMK_OP_ADD: if (isc(p.b)) a = {aff[p.a].base, aff[p.a].slope, aff[p.a].off + ops[p.b].c}; else if (isc(p.a)) a = {aff[p.b].base, aff[p.b].slope, aff[p.b].off + ops[p.a].c}; break;
        case HAMK_OP_SUB: if (isc(p.b)) a = {aff[p.a].base, aff[p.a].slope, aff[p.a].off - ops[p.b].c}; else if (isc(p.a)) a = {aff[p.b].base, -aff[p.b].slope, ops[p.a].c - aff[p.b].off}; break;
        case HAMK_OP_MUL: if (isc(p.b)) a = {aff[p.a].base, aff[p.a].slope * ops[p.b].c, aff[p.a].off * ops[p.b].c}; else if (isc(p.a)) a = {aff[p.b].base, aff[p.b].slope * ops[p.a].c, aff[p.b].off * ops[p.a].c}; break;
        case HAMK_OP_DIV: if (isc(p.b) && ops[p.b].c != 0.0) a = {aff[p.a].base, aff[p.a].slope / ops[p.b].c, aff[p.a].off / ops[p.b].c}; break;
        case HAMK_OP_NEG: a = {aff[p.a].base, -aff[p.a].slope, -aff[p.a].off}; break;
        default: break;
      }
      if (p.op == HAMK_OP_CONST) a = {-1, 0.0, p.c};
      aff[(size_t)i] = a;
    }
    for (int i = 0; i < nops; ++i) {
      if (ops[i].op != HAMK_OP_EXP) continue;
      const Affine& ai = aff[(size_t)ops[i].a];
      if (ai.base < 0) continue;
      for (int j = 0; j < i; ++j) {
        if (ops[j].op != HAMK_OP_EXP || exp_of[j] >= 0) continue;
        const Affine& aj = aff[(size_t)ops[j].a];
        const double delta = ai.off - aj.off;
        if (aj.base == ai.base && aj.slope == ai.slope && std::fabs(delta) <= 600.0) { exp_of[i] = j; exp_factor[i] = std::exp(delta); break; }
      }
    }
  }
  int put_seq = 0;
  std::vector<int> slot_of(nops, -1);          // operand value id -> trig cache slot
  int nslots = 0;
  auto slot = [&](int operand) {
    if (slot_of[operand] < 0) slot_of[operand] = nslots++;
    return std::string(tc_name) + ", " + std::to_string(slot_of[operand]);
  };
  // shared_f_slot (potential over generalized coordinates, evaluated right after f at the same q):
  // a sincos of INPUT j for which f has a site with the same operand reads f's pair instead of
  // evaluating it again (spring, Examples.hs:144-162: cos theta in f and in U)
  auto shared = [&](int operand) {
    if (!shared_f_slot || ops[operand].op != HAMK_OP_INPUT) return -1;
    return (*shared_f_slot)[ops[operand].a];
  };
  // INPUT values are not materialised: every use reads `in[j]` in place.  With array inputs
  // that is a reference; with the wave kernels' LDS-backed proxies it keeps 32 input jets from
  // all being live from the top of the function (the tape lists its inputs first).
  std::vector<char> inline_scale(nops, 0);
  if (g_emit_unshare_scales) {
    std::vector<int> nuse(nops, 0);
    for (int i = 0; i < nops; ++i) {
      const hamk_op& p = ops[i];
      if (p.op == HAMK_OP_CONST || p.op == HAMK_OP_INPUT) continue;
      ++nuse[p.a];
      if (is_binary(p.op)) ++nuse[p.b];
    }
    for (int i = 0; i < nops; ++i)
      if (ops[i].op == HAMK_OP_MUL && nuse[i] > 1 && (ops[ops[i].a].op == HAMK_OP_CONST) != (ops[ops[i].b].op == HAMK_OP_CONST) &&
          !((size_t)i < g_emit_is_output.size() && g_emit_is_output[(size_t)i]) && !(sink_outs && !(*sink_outs)[i].empty()))
        inline_scale[i] = 1;
  }
  std::function<std::string(int)> v = [&](int i) -> std::string {
    if (ops[i].op == HAMK_OP_INPUT)
      return input_exprs ? (*input_exprs)[ops[i].a] : std::string("in[") + std::to_string(ops[i].a) + "]";
    if (inline_scale[i]) {
      const bool ca = ops[ops[i].a].op == HAMK_OP_CONST;
      const int c = ca ? ops[i].a : ops[i].b, x = ca ? ops[i].b : ops[i].a;
      return "(hamk::opaque_const(" + lit(ops[c].c) + ") * " + v(x) + ")";
    }
    return std::string(pfx) + std::to_string(i);
  };
  for (int i = 0; i < nops; ++i) {
    if (done[i]) continue;
    const hamk_op& p = ops[i];
    if (inline_scale[i]) { done[i] = 2; continue; }
    o << "    ";
    switch (p.op) {
      case HAMK_OP_CONST: o << "const double " << v(i) << " = " << lit(p.c) << ";\n"; break;
      case HAMK_OP_INPUT: o << "// input " << p.a << "\n"; break;
      case HAMK_OP_ADD: o << "const auto " << v(i) << " = " << v(p.a) << " + " << v(p.b) << ";\n"; break;
      case HAMK_OP_SUB: o << "const auto " << v(i) << " = " << v(p.a) << " - " << v(p.b) << ";\n"; break;
      case HAMK_OP_MUL: o << "const auto " << v(i) << " = " << v(p.a) << " * " << v(p.b) << ";\n"; break;
      case HAMK_OP_DIV: o << "const auto " << v(i) << " = " << v(p.a) << " / " << v(p.b) << ";\n"; break;
      case HAMK_OP_NEG: o << "const auto " << v(i) << " = -" << v(p.a) << ";\n"; break;
      case HAMK_OP_POWC: o << "const auto " << v(i) << " = hamk::powc(" << v(p.a) << ", " << lit(p.c) << ");\n"; break;
      case HAMK_OP_POWI: o << "const auto " << v(i) << " = hamk::powi<" << p.b << ">(" << v(p.a) << ");\n"; break;
      case HAMK_OP_POW: o << "const auto " << v(i) << " = hamk::pow(" << v(p.a) << ", " << v(p.b) << ");\n"; break;
      case HAMK_OP_ATAN2: o << "const auto " << v(i) << " = hamk::atan2(" << v(p.a) << ", " << v(p.b) << ");\n"; break;
      case HAMK_OP_SIN:
      case HAMK_OP_COS: {
        const int si = sin_of[p.a], ci = cos_of[p.a];
        const int fs = shared(p.a);
        const std::string where = fs >= 0 ? "tcf, " + std::to_string(fs) : slot(p.a);
        const std::string mode = fs >= 0 ? "hamk::TRIG_REUSE" : trig_mode;
        if (si >= 0 && ci >= 0 && (si == i || ci == i) && !done[si] && !done[ci]) {
          o << "hamk::bare_t<decltype(" << v(p.a) << ")> " << v(si) << ", " << v(ci)
            << "; hamk::sincos<" << mode << ">(" << v(p.a) << ", " << v(si) << ", " << v(ci) << ", " << where << ");\n";
          done[si] = done[ci] = 1;
        } else {
          o << "const auto " << v(i) << " = hamk::" << (p.op == HAMK_OP_SIN ? "sin" : "cos") << "<" << mode << ">(" << v(p.a)
            << ", " << where << ");\n";
        }
      } break;
      default:
        if (p.op == HAMK_OP_EXP && exp_of[i] >= 0) o << "const auto " << v(i) << " = " << v(exp_of[i]) << " * " << lit(exp_factor[i]) << ";\n";
        else if (fused_rsqrt[i]) o << "const auto " << v(i) << " = hamk::rsqrt_of(" << v(ops[p.a].a) << ");\n";
        else o << "const auto " << v(i) << " = hamk::" << unary_name(p.op) << "(" << v(p.a) << ");\n";
        break;
    }
    done[i] = 1;
    if (sink_outs) {
      // a fused sincos defines two values at once: flush every defined value's outputs
      for (int j = 0; j < nops; ++j)
        if (done[j] == 1) {
          for (int k : (*sink_outs)[j]) {
            o << "    sink.template put<" << k << ", " << put_seq++ << ">(hamk::lift<A>(" << v(j) << "));\n";
            if (put_order) put_order->push_back(k);
          }
          done[j] = 2;
        }
    }
  }
  if (slot_operand) {
    slot_operand->assign(nslots, -1);
    for (int i = 0; i < nops; ++i)
      if (slot_of[i] >= 0) (*slot_operand)[slot_of[i]] = i;
  }
  return nslots;
}

// Name of value `id` in an emitted body: inputs are never materialised (see emit_body).
static std::string value_name(const std::vector<hamk_op>& ops, const char* pfx, int id,
                              const std::vector<std::string>* input_exprs = nullptr) {
  if (ops[id].op == HAMK_OP_INPUT)
    return input_exprs ? (*input_exprs)[ops[id].a] : std::string("in[") + std::to_string(ops[id].a) + "]";
  return std::string(pfx) + std::to_string(id);
}


// ---------------------------------------------------------------------------------------------
// Reverse sweep (MODE_R).  The quantity hamEqs needs from second derivatives is a GRADIENT:
//   dT/dq_i = -d/dq_i [ sum_k u_k (D_v x_k)(q) ],   u_k = m_k (J qd)_k held fixed, v = qd,
// so one forward pass of (value, tangent along v) followed by one reverse pass over the tape
// yields all n components at O(tape) cost, against the O(n * tape) of carrying n directions in
// forward mode (Jet2<N>).  Emitted as explicit scalar code; the elementary derivative rules are
// the d2_* helpers of hamk_device.hpp, sincos pairs come from the TrigCache of the first sweep.
// ---------------------------------------------------------------------------------------------
namespace {
struct RevTape {
  std::vector<hamk_op> ops;          // DIV expanded into RECIP + MUL
  std::vector<int> slot;             // per op: trig cache slot (SIN/COS), else -1
  std::vector<int> outs;             // output k -> value id
};

double host_ipow(double x, int k) {
  unsigned int e = (k < 0) ? (unsigned int)(-(long long)k) : (unsigned int)k;
  double r = 1.0, b = x;
  while (e) { if (e & 1u) r *= b; b *= b; e >>= 1; }
  return (k < 0) ? 1.0 / r : r;
}

// value of an op whose operands are all constants (folded on the host, fp64 as written)
double fold_const(const hamk_op& p, double a, double b) {
  switch (p.op) {
    case HAMK_OP_ADD: return a + b;   case HAMK_OP_SUB: return a - b;   case HAMK_OP_MUL: return a * b;
    case HAMK_OP_DIV: return a / b;   case HAMK_OP_NEG: return -a;      case HAMK_OP_RECIP: return 1.0 / a;
    case HAMK_OP_SIN: return std::sin(a);   case HAMK_OP_COS: return std::cos(a);   case HAMK_OP_TAN: return std::tan(a);
    case HAMK_OP_ASIN: return std::asin(a); case HAMK_OP_ACOS: return std::acos(a); case HAMK_OP_ATAN: return std::atan(a);
    case HAMK_OP_SINH: return std::sinh(a); case HAMK_OP_COSH: return std::cosh(a); case HAMK_OP_TANH: return std::tanh(a);
    case HAMK_OP_ASINH: return std::asinh(a); case HAMK_OP_ACOSH: return std::acosh(a); case HAMK_OP_ATANH: return std::atanh(a);
    case HAMK_OP_EXP: return std::exp(a);   case HAMK_OP_LOG: return std::log(a);   case HAMK_OP_SQRT: return std::sqrt(a);
    case HAMK_OP_POWC: return std::pow(a, p.c); case HAMK_OP_POWI: return host_ipow(a, p.b);
    case HAMK_OP_POW: return std::pow(a, b);    case HAMK_OP_ATAN2: return std::atan2(a, b);
    case HAMK_OP_ABS: return std::fabs(a);      case HAMK_OP_SIGNUM: return (double)((a > 0) - (a < 0));
    default: return std::nan("");
  }
}

RevTape expand_for_reverse(const SystemDesc& d) {
  RevTape r;
  const int n0 = (int)d.f_ops.size();
  std::vector<int> remap(n0, -1), slot_of_operand(n0, -1);
  int nslots = 0;
  for (int i = 0; i < n0; ++i) {
    hamk_op p = d.f_ops[i];
    int slot = -1;
    if (p.op == HAMK_OP_SIN || p.op == HAMK_OP_COS) {          // same numbering as emit_body
      if (slot_of_operand[p.a] < 0) slot_of_operand[p.a] = nslots++;
      slot = slot_of_operand[p.a];
    }
    const bool unary_or_binary = p.op != HAMK_OP_CONST && p.op != HAMK_OP_INPUT;
    const bool two = p.op == HAMK_OP_ADD || p.op == HAMK_OP_SUB || p.op == HAMK_OP_MUL || p.op == HAMK_OP_DIV ||
                     p.op == HAMK_OP_POW || p.op == HAMK_OP_ATAN2;
    if (unary_or_binary) p.a = remap[p.a];
    if (two) p.b = remap[p.b];
    if (unary_or_binary && r.ops[p.a].op == HAMK_OP_CONST && (!two || r.ops[p.b].op == HAMK_OP_CONST)) {
      hamk_op c{HAMK_OP_CONST, 0, 0, 0, fold_const(p, r.ops[p.a].c, two ? r.ops[p.b].c : 0.0)};
      r.ops.push_back(c); r.slot.push_back(-1);
    } else if (p.op == HAMK_OP_DIV && r.ops[p.b].op == HAMK_OP_CONST) {      // x / c = x * (1/c)
      hamk_op rc{HAMK_OP_CONST, 0, 0, 0, 1.0 / r.ops[p.b].c};
      r.ops.push_back(rc); r.slot.push_back(-1);
      hamk_op mu{HAMK_OP_MUL, p.a, (int32_t)r.ops.size() - 1, 0, 0.0};
      r.ops.push_back(mu); r.slot.push_back(-1);
    } else if (p.op == HAMK_OP_DIV) {
      hamk_op rc{HAMK_OP_RECIP, p.b, 0, 0, 0.0};
      r.ops.push_back(rc); r.slot.push_back(-1);
      hamk_op mu{HAMK_OP_MUL, p.a, (int32_t)r.ops.size() - 1, 0, 0.0};
      r.ops.push_back(mu); r.slot.push_back(-1);
    } else {
      r.ops.push_back(p); r.slot.push_back(slot);
    }
    remap[i] = (int)r.ops.size() - 1;
  }
  for (int k = 0; k < d.m; ++k) r.outs.push_back(remap[d.f_outs[k]]);
  return r;
}

const char* d2_name(int op) {
  switch (op) {
    case HAMK_OP_RECIP: return "recip"; case HAMK_OP_TAN: return "tan"; case HAMK_OP_ASIN: return "asin";
    case HAMK_OP_ACOS: return "acos"; case HAMK_OP_ATAN: return "atan"; case HAMK_OP_SINH: return "sinh";
    case HAMK_OP_COSH: return "cosh"; case HAMK_OP_TANH: return "tanh"; case HAMK_OP_ASINH: return "asinh";
    case HAMK_OP_ACOSH: return "acosh"; case HAMK_OP_ATANH: return "atanh"; case HAMK_OP_EXP: return "exp";
    case HAMK_OP_LOG: return "log"; case HAMK_OP_SQRT: return "sqrt"; case HAMK_OP_ABS: return "abs";
    case HAMK_OP_SIGNUM: return "signum"; default: return nullptr;
  }
}
}  // namespace

static void emit_reverse(std::ostringstream& o, const SystemDesc& d) {
  const RevTape r = expand_for_reverse(d);
  const int n = (int)r.ops.size();
  auto is_const = [&](int i) { return r.ops[i].op == HAMK_OP_CONST; };
  auto is_input = [&](int i) { return r.ops[i].op == HAMK_OP_INPUT; };
  std::vector<char> active(n, 0);                    // depends on an input
  for (int i = 0; i < n; ++i) {
    const hamk_op& p = r.ops[i];
    if (p.op == HAMK_OP_INPUT) active[i] = 1;
    else if (p.op != HAMK_OP_CONST) {
      active[i] = active[p.a];
      if (p.op == HAMK_OP_ADD || p.op == HAMK_OP_SUB || p.op == HAMK_OP_MUL || p.op == HAMK_OP_POW || p.op == HAMK_OP_ATAN2)
        active[i] = active[p.a] || active[p.b];
    }
  }
  auto val = [&](int i) -> std::string {             // primal value
    if (is_const(i)) return lit(r.ops[i].c);
    if (is_input(i)) return "q[" + std::to_string(r.ops[i].a) + "]";
    return "r" + std::to_string(i);
  };
  auto tan_ = [&](int i) -> std::string {            // tangent along v
    if (!active[i]) return "0.0";
    if (is_input(i)) return "v[" + std::to_string(r.ops[i].a) + "]";
    return "t" + std::to_string(i);
  };
  auto valr = [&](int i) -> std::string {            // the same, as the reverse pass reads them
    if (is_input(i)) return "qr.at(" + std::to_string(r.ops[i].a) + ")";
    return val(i);
  };
  auto tanr = [&](int i) -> std::string {
    if (active[i] && is_input(i)) return "vr.at(" + std::to_string(r.ops[i].a) + ")";
    return tan_(i);
  };
  auto adj = [&](int i, char w) -> std::string {     // adjoint accumulators: a = of value, b = of tangent
    if (is_input(i)) return std::string(1, w) + "q" + std::to_string(r.ops[i].a);
    return std::string(1, w) + std::to_string(i);
  };
  o << "  // dT[i] = -d/dq_i sum_k inertia(k) * (J v)_k * (D_v x_k)(q), (J v)_k held fixed: forward (value, tangent),\n";
  o << "  // then adjoints in reverse tape order\n";
  o << "  template <class QA, class VA, class TC> __device__ __forceinline__ static void dT_reverse(const QA& q, const VA& v, TC& tc, double (&dT)[N]) {\n";
  // ---- forward -----------------------------------------------------------------------------
  for (int i = 0; i < n; ++i) {
    const hamk_op& p = r.ops[i];
    if (p.op == HAMK_OP_CONST || p.op == HAMK_OP_INPUT) continue;
    const std::string I = std::to_string(i);
    o << "    ";
    if (!active[i]) {                                // constant subexpression (kept exact, no tangent)
      switch (p.op) {
        case HAMK_OP_ADD: o << "const double r" << I << " = " << val(p.a) << " + " << val(p.b) << ";\n"; break;
        case HAMK_OP_SUB: o << "const double r" << I << " = " << val(p.a) << " - " << val(p.b) << ";\n"; break;
        case HAMK_OP_MUL: o << "const double r" << I << " = " << val(p.a) << " * " << val(p.b) << ";\n"; break;
        case HAMK_OP_NEG: o << "const double r" << I << " = -" << val(p.a) << ";\n"; break;
        default: o << "const double r" << I << " = hamk::quiet_nan(); // constant subexpression of a function: fold on the host\n"; break;
      }
      continue;
    }
    switch (p.op) {
      case HAMK_OP_ADD:
        o << "const double r" << I << " = " << val(p.a) << " + " << val(p.b) << ", t" << I << " = " << tan_(p.a) << " + " << tan_(p.b) << ";\n"; break;
      case HAMK_OP_SUB:
        o << "const double r" << I << " = " << val(p.a) << " - " << val(p.b) << ", t" << I << " = " << tan_(p.a) << " - " << tan_(p.b) << ";\n"; break;
      case HAMK_OP_MUL:
        o << "const double r" << I << " = " << val(p.a) << " * " << val(p.b) << ", t" << I << " = " << val(p.a) << " * " << tan_(p.b)
          << " + " << tan_(p.a) << " * " << val(p.b) << ";\n"; break;
      case HAMK_OP_NEG:
        o << "const double r" << I << " = -" << val(p.a) << ", t" << I << " = -" << tan_(p.a) << ";\n"; break;
      case HAMK_OP_SIN:
        o << "const double r" << I << " = tc.s[" << r.slot[i] << "], g" << I << " = tc.c[" << r.slot[i] << "], h" << I << " = -tc.s[" << r.slot[i]
          << "], t" << I << " = g" << I << " * " << tan_(p.a) << ";\n"; break;
      case HAMK_OP_COS:
        o << "const double r" << I << " = tc.c[" << r.slot[i] << "], g" << I << " = -tc.s[" << r.slot[i] << "], h" << I << " = -tc.c[" << r.slot[i]
          << "], t" << I << " = g" << I << " * " << tan_(p.a) << ";\n"; break;
      case HAMK_OP_POWI:
        o << "double r" << I << ", g" << I << ", h" << I << "; hamk::d2_powi<" << p.b << ">(" << val(p.a) << ", r" << I << ", g" << I << ", h" << I
          << "); const double t" << I << " = g" << I << " * " << tan_(p.a) << ";\n"; break;
      case HAMK_OP_POWC:
        o << "double r" << I << ", g" << I << ", h" << I << "; hamk::d2_powc(" << val(p.a) << ", " << lit(p.c) << ", r" << I << ", g" << I << ", h" << I
          << "); const double t" << I << " = g" << I << " * " << tan_(p.a) << ";\n"; break;
      case HAMK_OP_POW:
      case HAMK_OP_ATAN2:
        o << "double r" << I << ", fa" << I << ", fb" << I << ", faa" << I << ", fab" << I << ", fbb" << I << "; hamk::d2_"
          << (p.op == HAMK_OP_POW ? "pow" : "atan2") << "(" << val(p.a) << ", " << val(p.b) << ", r" << I << ", fa" << I << ", fb" << I << ", faa" << I
          << ", fab" << I << ", fbb" << I << "); const double t" << I << " = fa" << I << " * " << tan_(p.a) << " + fb" << I << " * " << tan_(p.b) << ";\n";
        break;
      default:
        o << "double r" << I << ", g" << I << ", h" << I << "; hamk::d2_" << d2_name(p.op) << "(" << val(p.a) << ", r" << I << ", g" << I << ", h" << I
          << "); const double t" << I << " = g" << I << " * " << tan_(p.a) << ";\n";
        break;
    }
  }
  // ---- adjoints -----------------------------------------------------------------------------
  // (the reverse pass reads q, v and the sincos pairs through reverse_vec / reverse_trig: the identity where they are registers; where they
  // are LDS rows -- the four-lane kernels -- a re-laundered pointer, so that the pass loads them again instead of keeping 3 n
  // doubles alive from the forward pass in registers the forward pass's own intermediates need)
  o << "    using hamk::reverse_vec; using hamk::reverse_trig;\n";
  o << "    const auto qr = reverse_vec(q); const auto vr = reverse_vec(v); const auto& tcr = reverse_trig(tc);\n";
  o << "    double";
  for (int j = 0; j < d.n; ++j) o << (j ? "," : "") << " aq" << j << " = 0.0, bq" << j << " = 0.0";
  o << ";\n";
  for (int i = 0; i < n; ++i)
    if (active[i] && !is_input(i)) o << "    double a" << i << " = 0.0, b" << i << " = 0.0;\n";
  for (int k = 0; k < d.m; ++k) {                    // seeds: g = sum_k u_k t_k, u_k = m_k t_k (fixed)
    const int id = r.outs[k];
    if (!active[id]) continue;
    o << "    " << adj(id, 'b') << " += " << lit(d.inertia[k]) << " * " << tan_(id) << ";\n";
  }
  for (int i = n - 1; i >= 0; --i) {
    const hamk_op& p = r.ops[i];
    if (!active[i] || is_input(i)) continue;
    const std::string I = std::to_string(i), aI = "a" + I, bI = "b" + I;
    auto acc = [&](int x, char w, const std::string& expr) {
      if (active[x]) o << "    " << adj(x, w) << " += " << expr << ";\n";
    };
    switch (p.op) {
      case HAMK_OP_ADD: acc(p.a, 'a', aI); acc(p.a, 'b', bI); acc(p.b, 'a', aI); acc(p.b, 'b', bI); break;
      case HAMK_OP_SUB: acc(p.a, 'a', aI); acc(p.a, 'b', bI); acc(p.b, 'a', "-" + aI); acc(p.b, 'b', "-" + bI); break;
      case HAMK_OP_NEG: acc(p.a, 'a', "-" + aI); acc(p.a, 'b', "-" + bI); break;
      case HAMK_OP_MUL:
        acc(p.a, 'a', aI + " * " + valr(p.b) + (active[p.b] ? " + " + bI + " * " + tanr(p.b) : ""));
        acc(p.a, 'b', bI + " * " + valr(p.b));
        acc(p.b, 'a', aI + " * " + valr(p.a) + (active[p.a] ? " + " + bI + " * " + tanr(p.a) : ""));
        acc(p.b, 'b', bI + " * " + valr(p.a));
        break;
      case HAMK_OP_POW:
      case HAMK_OP_ATAN2:
        acc(p.a, 'a', aI + " * fa" + I + " + " + bI + " * (faa" + I + " * " + tanr(p.a) + " + fab" + I + " * " + tanr(p.b) + ")");
        acc(p.a, 'b', bI + " * fa" + I);
        acc(p.b, 'a', aI + " * fb" + I + " + " + bI + " * (fab" + I + " * " + tanr(p.a) + " + fbb" + I + " * " + tanr(p.b) + ")");
        acc(p.b, 'b', bI + " * fb" + I);
        break;
      case HAMK_OP_SIN: {                             // g = cos, h = -sin, read again from the cache's view
        const std::string sl = std::to_string(r.slot[i]);
        acc(p.a, 'a', aI + " * tcr.c[" + sl + "] - " + bI + " * tcr.s[" + sl + "] * " + tanr(p.a));
        acc(p.a, 'b', bI + " * tcr.c[" + sl + "]");
      } break;
      case HAMK_OP_COS: {                             // g = -sin, h = -cos
        const std::string sl = std::to_string(r.slot[i]);
        acc(p.a, 'a', "-(" + aI + " * tcr.s[" + sl + "] + " + bI + " * tcr.c[" + sl + "] * " + tanr(p.a) + ")");
        acc(p.a, 'b', "-(" + bI + " * tcr.s[" + sl + "])");
      } break;
      default:                                        // every other unary function: y = g(x), ty = g'(x) tx
        acc(p.a, 'a', aI + " * g" + I + " + " + bI + " * h" + I + " * " + tanr(p.a));
        acc(p.a, 'b', bI + " * g" + I);
        break;
    }
  }
  for (int j = 0; j < d.n; ++j) o << "    dT[" << j << "] = -aq" << j << "; (void)bq" << j << ";\n";
  o << "  }\n";
}

// How many DISTINCT non-zero entries the Jacobian of the coordinate map has, as expressions: first-order forward mode run
// symbolically over the tape with hash-consing (0 and 1 folded, + commutative).  A chain's dx_k/dq_j is ONE expression for
// every k >= j (2n distinct entries in an m x n = 2n x n matrix); a generic dense map has m n.  This is what decides
// whether a lane can run the per-trajectory sweep of a large system with compile-time seeds (hamk_quad.hpp): the compiler
// keeps one register pair per distinct entry, not per entry.
int distinct_jacobian_entries(const SystemDesc& d) {
  const int n = d.n, nops = (int)d.f_ops.size();
  enum : long long { ZERO = 0, ONE = 1 };
  std::map<std::array<long long, 4>, long long> table;
  long long next = 2;
  auto node = [&](long long kind, long long a, long long b, long long c) -> long long {
    const std::array<long long, 4> key = {kind, a, b, c};
    auto it = table.find(key);
    if (it != table.end()) return it->second;
    table[key] = next;
    return next++;
  };
  auto add = [&](long long a, long long b) { if (a == ZERO) return b; if (b == ZERO) return a; return node(1, std::min(a, b), std::max(a, b), 0); };
  auto neg = [&](long long a) { return a == ZERO ? (long long)ZERO : node(2, a, 0, 0); };
  auto scale = [&](long long factor_kind, long long factor_id, long long a) {       // (run-time factor) * a
    if (a == ZERO) return (long long)ZERO;
    return node(3, factor_kind * (1LL << 40) + factor_id, a, 0);                    // a == ONE: the factor itself
  };
  std::vector<std::vector<long long>> g(nops, std::vector<long long>(n, ZERO));
  for (int i = 0; i < nops; ++i) {
    const hamk_op& p = d.f_ops[i];
    switch (p.op) {
      case HAMK_OP_CONST: break;
      case HAMK_OP_INPUT: g[i][p.a] = ONE; break;
      case HAMK_OP_ADD: for (int j = 0; j < n; ++j) g[i][j] = add(g[p.a][j], g[p.b][j]); break;
      case HAMK_OP_SUB: for (int j = 0; j < n; ++j) g[i][j] = add(g[p.a][j], neg(g[p.b][j])); break;
      case HAMK_OP_NEG: for (int j = 0; j < n; ++j) g[i][j] = neg(g[p.a][j]); break;
      case HAMK_OP_MUL: {
        const bool ca = d.f_ops[p.a].op == HAMK_OP_CONST, cb = d.f_ops[p.b].op == HAMK_OP_CONST;
        for (int j = 0; j < n; ++j) {
          // value(a) * db + da * value(b); a constant factor is identified by its value (l * cos q_j is one node for all k)
          const long long t1 = cb ? (long long)ZERO : scale(ca ? 5 : 4, ca ? (long long)std::hash<double>{}(d.f_ops[p.a].c) & ((1LL << 40) - 1) : p.a, g[p.b][j]);
          const long long t2 = ca ? (long long)ZERO : scale(cb ? 5 : 4, cb ? (long long)std::hash<double>{}(d.f_ops[p.b].c) & ((1LL << 40) - 1) : p.b, g[p.a][j]);
          g[i][j] = add(t1, t2);
        }
      } break;
      case HAMK_OP_DIV: case HAMK_OP_POW: case HAMK_OP_ATAN2:
        for (int j = 0; j < n; ++j) g[i][j] = (g[p.a][j] == ZERO && g[p.b][j] == ZERO) ? (long long)ZERO : node(6, i, g[p.a][j], g[p.b][j]);
        break;
      default:                                             // every unary function: g'(x) * dx, g' identified by (opcode, operand)
        for (int j = 0; j < n; ++j) g[i][j] = scale(7 + p.op, p.a, g[p.a][j]);
        break;
    }
  }
  std::set<long long> distinct;
  for (int k = 0; k < d.m; ++k)
    for (int j = 0; j < n; ++j)
      if (g[d.f_outs[k]][j] != ZERO) distinct.insert(g[d.f_outs[k]][j]);
  return (int)distinct.size();
}

// What one first-order forward sweep of f costs a lane that runs it with COMPILE-TIME seeds (hamk_quad.hpp): the dependency set of
// every tape value is known here, and an operation computes only the gradient entries its operands make non-zero -- a product
// with a constant or a function of one value costs |support| operations, a sum costs the OVERLAP of its operands' supports (an
// entry only one operand has is passed through), a product of two values their union.  x = 2 q + A sin q + B cos q: 2 n per
// output although the Jacobian is dense; a map whose every operation depends on every input: (tape length) x n.
long long forward_gradient_work(const SystemDesc& d) {
  const int n = d.n, nops = (int)d.f_ops.size();
  std::vector<std::vector<char>> sup(nops, std::vector<char>(n, 0));
  auto count = [&](const std::vector<char>& v) { long long c = 0; for (char x : v) c += x; return c; };
  long long work = 0;
  for (int i = 0; i < nops; ++i) {
    const hamk_op& p = d.f_ops[i];
    if (p.op == HAMK_OP_CONST) continue;
    if (p.op == HAMK_OP_INPUT) { sup[i][p.a] = 1; continue; }
    if (is_binary(p.op)) {
      long long both = 0, any = 0;
      for (int j = 0; j < n; ++j) { const char a = sup[p.a][j], b = sup[p.b][j]; sup[i][j] = a | b; both += a & b; any += a | b; }
      const bool ca = d.f_ops[p.a].op == HAMK_OP_CONST, cb = d.f_ops[p.b].op == HAMK_OP_CONST;
      if (p.op == HAMK_OP_ADD || p.op == HAMK_OP_SUB) work += both;
      else if (ca || cb) work += any;                       // scaling
      else work += count(sup[p.a]) + count(sup[p.b]);       // value(a) db + da value(b)
    } else {
      sup[i] = sup[p.a];
      work += (p.op == HAMK_OP_NEG) ? 0 : count(sup[i]) + 2;
    }
  }
  return work;
}

// ---------------------------------------------------------------------------------------------
// Symbolic mass matrix (round 6).  K = J^T M J of the reference (Hamilton.hs:380) is a sum of products of Jacobian entries, and for
// the systems people write it simplifies: a point on a circle contributes m (cos^2 + sin^2) = m, polar coordinates make the
// r-phi entries cancel, a double pendulum's K is [[m1 + m2, m2/2 cos(t1 - t2)], [., m2/4]].  The compiler sees only fp64
// arithmetic and may not use sin^2 + cos^2 = 1; the generator can.  Where f is a polynomial in its inputs and in sincos of
// polynomial arguments (ADD, SUB, MUL, NEG, POWI, division by constants, SIN, COS), the Jacobian is derived here as polynomials
// over {q_j, s_k, c_k} (s_k, c_k: the sincos pair of trig-cache slot k), K[a][b] = sum_k m_k J[k][a] J[k][b] is expanded, even
// powers of s_k are rewritten through 1 - c_k^2, like terms are collected (terms whose coefficient cancelled to rounding are
// dropped) -- and if the result is cheaper than the numerical sum, the module gets `mass_matrix_sym` and the lane kernels use
// it.  doublePendulum: K00 = 2, K11 = 1/4 (constants: the 2 x 2 solve folds around them); twoBody: diag(mu, mu r^2);
// threeBodyPolar: diag(1, r1^2, 1, r2^2, 1, r3^2) -- the 6 x 6 LDL^T becomes six divisions.  Results differ from the numerical
// K by the rounding of the terms that cancel analytically (~1e-16 relative).
// ---------------------------------------------------------------------------------------------
namespace {
typedef std::vector<std::pair<int, int>> Mono;            // sorted (variable, power)
typedef std::map<Mono, double> Poly;
struct SymFail {};

Mono mono_mul(const Mono& a, const Mono& b) {
  Mono r; size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    if (j >= b.size() || (i < a.size() && a[i].first < b[j].first)) r.push_back(a[i++]);
    else if (i >= a.size() || b[j].first < a[i].first) r.push_back(b[j++]);
    else { r.push_back({a[i].first, a[i].second + b[j].second}); ++i; ++j; }
  }
  return r;
}
void poly_add(Poly& a, const Poly& b, double f = 1.0) {
  for (const auto& t : b) { a[t.first] += f * t.second; }
  if (a.size() > 20000) throw SymFail{};
}
Poly poly_mul(const Poly& a, const Poly& b) {
  Poly r;
  if (a.size() * b.size() > 200000) throw SymFail{};
  for (const auto& x : a) for (const auto& y : b) r[mono_mul(x.first, y.first)] += x.second * y.second;
  return r;
}
Poly poly_const(double c) { Poly p; if (c != 0.0) p[Mono{}] = c; return p; }
Poly poly_var(int v) { Poly p; p[Mono{{v, 1}}] = 1.0; return p; }
// s^2 -> 1 - c^2 for every sincos pair (sin variable ids are n + 2k, cos ids n + 2k + 1), then drop what cancelled
Poly poly_reduce(const Poly& a, int n, int hi = 1 << 30) {   // sincos variables: ids in [n, hi)
  Poly cur = a;
  for (bool again = true; again;) {
    again = false;
    Poly next;
    for (const auto& t : cur) {
      int at = -1;
      for (size_t i = 0; i < t.first.size(); ++i)
        if (t.first[i].first >= n && t.first[i].first < hi && ((t.first[i].first - n) & 1) == 0 && t.first[i].second >= 2) { at = (int)i; break; }
      if (at < 0) { next[t.first] += t.second; continue; }
      again = true;
      Mono rest = t.first;
      const int sv = rest[at].first;
      rest[at].second -= 2;
      if (rest[at].second == 0) rest.erase(rest.begin() + at);
      next[rest] += t.second;                               // ... * 1
      next[mono_mul(rest, Mono{{sv + 1, 2}})] -= t.second;  // ... * (-c^2)
    }
    cur.swap(next);
    if (cur.size() > 20000) throw SymFail{};
  }
  return cur;
}
void poly_prune(Poly& p, double scale) {
  for (auto it = p.begin(); it != p.end();) it = (std::fabs(it->second) <= 8.0 * 2.220446049250313e-16 * scale) ? p.erase(it) : std::next(it);
}
double poly_abs_sum(const Poly& p) { double s = 0.0; for (const auto& t : p) s += std::fabs(t.second); return s; }

struct SymK {
  bool ok = false;
  int n = 0;
  std::vector<Poly> k;          // upper triangle, row-major: (a, b), a <= b
  std::vector<Poly> dT;         // dT/dq_i = -1/2 v^T (dK/dq_i) v as a polynomial over {q, s, c, v}; variable ids of v_a: vbase + a
  int vbase = 0;
  bool dt_ok = false;
  long long sym_ops = 0, num_ops = 0, dt_ops = 0;
};

// slot_operand: trig-cache slot -> tape value that is the sincos operand (emit_body's numbering)
SymK symbolic_mass_matrix(const SystemDesc& d, const std::vector<int>& slot_operand) {
  SymK out;
  const int n = d.n, m = d.m, nops = (int)d.f_ops.size();
  out.n = n;
  // the small systems (configs 2-4, the reference's examples): K is a visible share of their right-hand side.  Counted for the chains beyond
  // (instructions per RK4 step with / without, round 6): chain8 2 720 / 2 708, chain12 6 044 / 6 312, chain16 12 308 / 12 232 -- the compiler's
  // re-associated numerical sum is already the closed form there
  if (n > 7 || d.mapping != HAMK_MAP_LANE) return out;
  std::vector<int> slot_of(nops, -1);
  for (size_t k = 0; k < slot_operand.size(); ++k) if (slot_operand[k] >= 0) slot_of[(size_t)slot_operand[k]] = (int)k;
  try {
    std::vector<Poly> V(nops);
    std::vector<std::vector<Poly>> G(nops, std::vector<Poly>(n));
    for (int i = 0; i < nops; ++i) {
      const hamk_op& p = d.f_ops[i];
      switch (p.op) {
        case HAMK_OP_CONST: V[i] = poly_const(p.c); break;
        case HAMK_OP_INPUT: V[i] = poly_var(p.a); G[i][p.a] = poly_const(1.0); break;
        case HAMK_OP_ADD: V[i] = V[p.a]; poly_add(V[i], V[p.b]); for (int j = 0; j < n; ++j) { G[i][j] = G[p.a][j]; poly_add(G[i][j], G[p.b][j]); } break;
        case HAMK_OP_SUB: V[i] = V[p.a]; poly_add(V[i], V[p.b], -1.0); for (int j = 0; j < n; ++j) { G[i][j] = G[p.a][j]; poly_add(G[i][j], G[p.b][j], -1.0); } break;
        case HAMK_OP_NEG: poly_add(V[i], V[p.a], -1.0); for (int j = 0; j < n; ++j) poly_add(G[i][j], G[p.a][j], -1.0); break;
        case HAMK_OP_MUL:
          V[i] = poly_reduce(poly_mul(V[p.a], V[p.b]), n);
          for (int j = 0; j < n; ++j) {
            if (!G[p.b][j].empty()) poly_add(G[i][j], poly_mul(V[p.a], G[p.b][j]));
            if (!G[p.a][j].empty()) poly_add(G[i][j], poly_mul(G[p.a][j], V[p.b]));
            G[i][j] = poly_reduce(G[i][j], n);
          }
          break;
        case HAMK_OP_DIV: {
          if (d.f_ops[p.b].op != HAMK_OP_CONST || d.f_ops[p.b].c == 0.0) throw SymFail{};
          const double r = 1.0 / d.f_ops[p.b].c;            // (x / c: the numerical path divides; the difference is one rounding of a constant)
          poly_add(V[i], V[p.a], r); for (int j = 0; j < n; ++j) poly_add(G[i][j], G[p.a][j], r);
        } break;
        case HAMK_OP_POWI: {
          if (p.b < 0 || p.b > 6) throw SymFail{};
          Poly acc = poly_const(1.0), dacc;                  // x^k and k x^(k-1)
          for (int e = 0; e < p.b; ++e) { dacc = acc; acc = poly_reduce(poly_mul(acc, V[p.a]), n); }
          V[i] = acc;
          if (p.b >= 1) for (int j = 0; j < n; ++j) if (!G[p.a][j].empty()) { Poly t = poly_mul(dacc, G[p.a][j]); poly_add(G[i][j], t, (double)p.b); G[i][j] = poly_reduce(G[i][j], n); }
        } break;
        case HAMK_OP_SIN: case HAMK_OP_COS: {
          const int k = slot_of[p.a];
          if (k < 0) throw SymFail{};
          const int sv = n + 2 * k, cv = sv + 1;
          const bool is_sin = p.op == HAMK_OP_SIN;
          V[i] = poly_var(is_sin ? sv : cv);
          for (int j = 0; j < n; ++j)
            if (!G[p.a][j].empty()) { Poly t = poly_mul(poly_var(is_sin ? cv : sv), G[p.a][j]); poly_add(G[i][j], t, is_sin ? 1.0 : -1.0); G[i][j] = poly_reduce(G[i][j], n); }
        } break;
        default: throw SymFail{};                            // sqrt, exp, recip ...: not a polynomial
      }
    }
    out.k.resize((size_t)n * (n + 1) / 2);
    size_t e = 0;
    for (int a = 0; a < n; ++a)
      for (int b = a; b < n; ++b, ++e) {
        Poly acc;
        double scale = 0.0;
        int both = 0;
        for (int k = 0; k < m; ++k) {
          const Poly& ga = G[d.f_outs[k]][a];
          const Poly& gb = G[d.f_outs[k]][b];
          if (ga.empty() || gb.empty()) continue;
          ++both;
          Poly t = poly_mul(ga, gb);
          scale += std::fabs(d.inertia[k]) * poly_abs_sum(t);
          poly_add(acc, t, d.inertia[k]);
        }
        acc = poly_reduce(acc, n);
        poly_prune(acc, scale);
        out.num_ops += 2LL * both;
        for (const auto& t : acc) {
          int deg = 0; for (const auto& vp : t.first) deg += vp.second;
          out.sym_ops += deg + ((std::fabs(t.second) != 1.0 && deg > 0) ? 1 : 0);
        }
        if (acc.size() > 1) out.sym_ops += (long long)acc.size() - 1;
        if (acc.size() > 24) throw SymFail{};
        out.k[e] = acc;
      }
    out.ok = out.sym_ops < out.num_ops;
    // dT/dq_i = -(M J v) . ((dJ/dq_i) v) = -1/2 v^T (dK/dq_i) v at fixed v (Hamilton.hs:382-385 in the form DESIGN.md section 3 derives):
    // with K a polynomial its derivative is one too -- d s_k = c_k d(arg_k), d c_k = -s_k d(arg_k) -- and the second-order sweep of f
    // (full second-order jets for n <= 3, the directional sweep above) is not needed at all
    if (out.ok) {
      const int nslots = (int)slot_operand.size();
      out.vbase = n + 2 * nslots;
      auto dvar = [&](int var, int i) -> Poly {               // d(var)/dq_i
        if (var < n) return (var == i) ? poly_const(1.0) : Poly();
        const int k = (var - n) / 2;
        const Poly& arg = G[slot_operand[(size_t)k]][i];
        if (arg.empty()) return Poly();
        Poly r = poly_mul(poly_var(((var - n) & 1) ? var - 1 : var + 1), arg);      // cos: -sin * d arg; sin: cos * d arg
        if ((var - n) & 1) { Poly neg; poly_add(neg, r, -1.0); return neg; }
        return r;
      };
      out.dT.assign((size_t)n, Poly());
      bool fits = true;
      for (int i = 0; i < n && fits; ++i) {
        Poly acc;
        double scale = 0.0;
        size_t e2 = 0;
        for (int a = 0; a < n; ++a)
          for (int b = a; b < n; ++b, ++e2) {
            const double w = (a == b) ? -0.5 : -1.0;
            for (const auto& t : out.k[e2]) {
              for (size_t f = 0; f < t.first.size(); ++f) {
                const Poly dv = dvar(t.first[f].first, i);
                if (dv.empty()) continue;
                Mono rest = t.first;
                const double pw = (double)rest[f].second;
                rest[f].second -= 1;
                if (rest[f].second == 0) rest.erase(rest.begin() + (long)f);
                Mono vv = (a == b) ? Mono{{out.vbase + a, 2}} : Mono{{out.vbase + a, 1}, {out.vbase + b, 1}};
                Poly term; term[mono_mul(rest, vv)] = w * pw * t.second;
                Poly prod = poly_mul(term, dv);
                scale += poly_abs_sum(prod);
                poly_add(acc, prod);
              }
            }
          }
        acc = poly_reduce(acc, n, out.vbase);
        poly_prune(acc, scale);
        for (const auto& t : acc) { int deg = 0; for (const auto& vp : t.first) deg += vp.second; out.dt_ops += deg + 1; }
        if (acc.size() > 64) fits = false;
        out.dT[(size_t)i] = acc;
      }
      // taken where it is SHORT (<= 12 n operations: doublePendulum 20, twoBody 4, spring 31, threeBodyPolar 12).  A chain's dT/dq has
      // n (n - 1) terms of degree four -- counted from the code objects, chain4 -3 % and chain6 +7 % instructions per step against the
      // directional second sweep, whose structural zeros the compiler already strips: the chains keep that sweep (and the symbolic K)
      out.dt_ok = fits && out.dt_ops <= 12LL * n;
    }
  } catch (const SymFail&) { out.ok = false; out.dt_ok = false; }
  return out;
}

static std::string sym_poly_expr(const Poly& p, int n, int vbase) {
  std::ostringstream x;
  bool first = true;
  for (const auto& t : p) {
    std::ostringstream f;
    bool unit = true;
    for (const auto& vp : t.first)
      for (int r = 0; r < vp.second; ++r) {
        f << (unit ? "" : " * ");
        unit = false;
        if (vp.first < n) f << "q[" << vp.first << "]";
        else if (vp.first >= vbase) f << "v[" << (vp.first - vbase) << "]";
        else f << "tc." << (((vp.first - n) & 1) ? "c" : "s") << "[" << (vp.first - n) / 2 << "]";
      }
    if (!first) x << " + ";
    first = false;
    if (unit) x << lit(t.second);
    else if (t.second == 1.0) x << f.str();
    else x << lit(t.second) << " * " << f.str();
  }
  if (first) x << "0.0";
  return x.str();
}

void emit_symbolic_dt(std::ostringstream& o, const SymK& sk) {
  o << "  // dT/dq = -1/2 v^T (dK/dq) v from the symbolic K (" << sk.dt_ops << " operations): no second-order sweep of f\n";
  o << "  template <class TC> __device__ __forceinline__ static void dT_sym(const double (&q)[N], const double (&v)[N], const TC& tc, double (&dT)[N]) {\n";
  for (int i = 0; i < sk.n; ++i) o << "    dT[" << i << "] = " << sym_poly_expr(sk.dT[(size_t)i], sk.n, sk.vbase) << ";\n";
  o << "  }\n";
}

void emit_symbolic_k(std::ostringstream& o, const SymK& sk) {
  const int n = sk.n;
  o << "  // K = J^T M J derived symbolically (sin^2 + cos^2 = 1 applied; " << sk.sym_ops << " operations against " << sk.num_ops << " for the numerical sum)\n";
  o << "  template <class TC> __device__ __forceinline__ static void mass_matrix_sym(const double (&q)[N], const TC& tc, double (&K)[N][N]) {\n";
  size_t e = 0;
  for (int a = 0; a < n; ++a)
    for (int b = a; b < n; ++b, ++e) {
      std::ostringstream x;
      bool first = true;
      for (const auto& t : sk.k[e]) {
        std::ostringstream f;
        bool unit = true;
        for (const auto& vp : t.first)
          for (int r = 0; r < vp.second; ++r) {
            f << (unit ? "" : " * ");
            unit = false;
            if (vp.first < n) f << "q[" << vp.first << "]";
            else f << "tc." << (((vp.first - n) & 1) ? "c" : "s") << "[" << (vp.first - n) / 2 << "]";
          }
        if (!first) x << " + ";
        first = false;
        if (unit) x << lit(t.second);
        else if (t.second == 1.0) x << f.str();
        else x << lit(t.second) << " * " << f.str();
      }
      if (first) x << "0.0";
      o << "    K[" << a << "][" << b << "] = " << x.str() << ";\n";
      if (a != b) o << "    K[" << b << "][" << a << "] = K[" << a << "][" << b << "];\n";
    }
  o << "  }\n";
}
}  // namespace

// Will a lane module of this system take K and dT/dq from the symbolic mass matrix?  (hamk_dispatch.cpp make_desc: such a right-hand
// side is short, which moves the sincos rule.)  Trig-cache slots numbered as emit_body numbers them: by first SIN / COS of an operand.
bool symbolic_rhs_applies(const SystemDesc& d0) {
  SystemDesc d = d0;
  d.mapping = HAMK_MAP_LANE;
  if (!d.k_symbolic) return false;
  std::vector<int> slot_of(d.f_ops.size(), -1), slot_operand;
  for (const hamk_op& p : d.f_ops)
    if ((p.op == HAMK_OP_SIN || p.op == HAMK_OP_COS) && slot_of[(size_t)p.a] < 0) { slot_of[(size_t)p.a] = (int)slot_operand.size(); slot_operand.push_back(p.a); }
  const SymK sk = symbolic_mass_matrix(d, slot_operand);
  return sk.ok && sk.dt_ok;
}

std::string generate_source(const SystemDesc& d) {
  std::ostringstream o;
  o << "// generated by libhamk (hamk_codegen.cpp) from the expression tape of one System " << d.m << " " << d.n << "\n";
  if (d.rk4_min_waves > 1 || d.wave) o << "#define HAMK_RK4_MIN_WAVES " << d.rk4_min_waves << "\n#define HAMK_RK4_MIN_WAVES_BIG " << d.rk4_min_waves << "\n";
  if (d.wave && d.n > 32 && d.rk4_min_waves > 1) o << "#define HAMK_RKF_MIN_WAVES " << d.rk4_min_waves << "\n";
  o << "#define HAMK_USE_LUT " << d.use_lut << "\n";
  o << "#define HAMK_K_REASSOC " << (d.k_reassoc ? 1 : 0) << "\n";
  if (d.rk4_park && !d.wave) o << "#define HAMK_RK4_PARK 1\n";
  if (d.mapping == HAMK_MAP_QUAD) o << "#define HAMK_QUAD_RKF_PARK " << (d.rkf_park ? 1 : 0) << "\n";
  if (d.mapping == HAMK_MAP_LANE && d.trig_const_vgpr && d.use_lut != 0) o << "#define HAMK_TRIG_CONST_VGPR 1\n";
  if (d.mapping == HAMK_MAP_LANE && d.pair_rows) o << "#define HAMK_PAIR_ROWS 1\n";
  if (d.mapping == HAMK_MAP_LANE && d.rkf_two_waves)
    o << "#define HAMK_RKF_MIN_WAVES_LANE 2\n#define HAMK_RKF_LDS_BUDGET 36\n#define HAMK_RKF_ROWS_IN_REGS 1\n";
  // (sin, cos)(i 2pi/512), correctly rounded from 80-bit: the constant data behind sincos_lut's LDS table
  o << "#ifdef HAMK_HOST_EMULATION\nstatic const double hamk_trig_lut_init[1024] = {\n#else\n__device__ const double hamk_trig_lut_init[1024] = {\n#endif\n";
  for (int i = 0; i < 512; ++i) {
    const long double a = (long double)i * (2.0L * 3.14159265358979323846264338327950288L / 512.0L);
    o << "  " << lit((double)sinl(a)) << ", " << lit((double)cosl(a)) << (i == 511 ? "\n" : ",\n");
  }
  o << "};\n";
  const bool quad = d.mapping == HAMK_MAP_QUAD;
  o << (d.wave ? "#include \"hamk_wave.hpp\"\n\n" : (quad ? "#include \"hamk_quad.hpp\"\n\n" : "#include \"hamk_device.hpp\"\n\n"));
  o << "struct HamkSys {\n";
  o << "  static constexpr int N = " << d.n << ";\n";
  o << "  static constexpr int M = " << d.m << ";\n";
  o << "  static constexpr bool U_CART = " << (d.u_space == HAMK_U_CARTESIAN ? "true" : "false") << ";\n";
  o << "  static constexpr bool MODE_H = " << (d.mode_h ? "true" : "false") << ";\n";
  o << "  static constexpr bool RK4_STAGE_LOOP = " << (d.rk4_stage_loop ? "true" : "false") << ";\n";
  o << "  static constexpr bool RKF_STAGE_LOOP = " << (d.rkf_stage_loop ? "true" : "false") << ";\n";
  o << "  static constexpr bool MODE_R = " << (d.mode_r ? "true" : "false") << ";\n";
  bool inertia_pos = true;
  for (double w : d.inertia) inertia_pos = inertia_pos && (w > 0.0);
  o << "  static constexpr bool INERTIA_POS = " << (inertia_pos ? "true" : "false") << ";\n";
  o << "  static constexpr bool QUAD_DENSE = " << ((quad && d.quad_dense) ? "true" : "false") << ";\n";
  o << "  __device__ __forceinline__ static constexpr double inertia(int k) {\n";
  o << "    constexpr double w[M] = {";
  for (int k = 0; k < d.m; ++k) o << (k ? ", " : "") << lit(d.inertia[k]);
  o << "};\n    return w[k];\n  }\n";
  // coordinate map f: generalized -> cartesian                       (_sysCoords, Hamilton.hs:220)
  o << "  template <class A, int TRIG, class TC> __device__ __forceinline__ static void coords(const A (&in)[N], A (&x)[M], TC& tc) {\n";
  std::vector<int> slot_operand;
  auto mark_f = [&] { mark_outputs(d.f_ops.size(), d.f_outs.data(), d.m); g_emit_unshare_scales = d.mapping == HAMK_MAP_QUAD && d.quad_dense; };
  auto mark_u = [&] { mark_outputs(d.u_ops.size(), &d.u_out, 1); g_emit_unshare_scales = false; };
  mark_f();
  const int ntrig_f = emit_body(o, d.f_ops.data(), (int)d.f_ops.size(), "f", nullptr, nullptr, "tc", &slot_operand);
  for (int k = 0; k < d.m; ++k) o << "    x[" << k << "] = hamk::lift<A>(" << value_name(d.f_ops, "f", d.f_outs[k]) << ");\n";
  o << "  }\n";
  // potential                                                         (_sysPotential, Hamilton.hs:223 / :254)
  const int nu = d.u_space == HAMK_U_CARTESIAN ? d.m : d.n;
  o << "  template <class A, int TRIG, class TC> __device__ __forceinline__ static A potential(const A (&in)[" << nu << "], TC& tc) {\n";
  mark_u();
  const int ntrig_u = emit_body(o, d.u_ops.data(), (int)d.u_ops.size(), "u");
  o << "    return hamk::lift<A>(" << value_name(d.u_ops, "u", d.u_out) << ");\n";
  o << "  }\n";
  // the same potential evaluated right after f at the same generalized coordinates: sincos sites of
  // inputs that f evaluates too read f's pair (tcf) instead of evaluating it again
  {
    std::vector<int> f_slot_of_input(d.n, -1);
    bool any = false;
    if (d.u_space == HAMK_U_GENERALIZED) {
      for (int k = 0; k < ntrig_f; ++k) {
        const int opi = slot_operand[k];
        if (opi >= 0 && d.f_ops[opi].op == HAMK_OP_INPUT) f_slot_of_input[d.f_ops[opi].a] = k;
      }
      for (const hamk_op& p : d.u_ops)
        if ((p.op == HAMK_OP_SIN || p.op == HAMK_OP_COS) && d.u_ops[p.a].op == HAMK_OP_INPUT && f_slot_of_input[d.u_ops[p.a].a] >= 0) any = true;
    }
    o << "  static constexpr bool U_SHARES_F_TRIG = " << (any ? "true" : "false") << ";\n";
    o << "  template <class A, int TRIG, class TC, class TCF> __device__ __forceinline__ static A potential_after_f(const A (&in)[" << nu
      << "], TC& tc, TCF& tcf) {\n";
    if (any) {
      mark_u();
      emit_body(o, d.u_ops.data(), (int)d.u_ops.size(), "u", nullptr, nullptr, "tc", nullptr, "TRIG", &f_slot_of_input);
      o << "    return hamk::lift<A>(" << value_name(d.u_ops, "u", d.u_out) << ");\n";
    } else {
      o << "    return potential<A, TRIG>(in, tc);\n";
    }
    o << "  }\n";
  }
  // the same map, delivering each output to a sink as soon as it is defined
  o << "  template <class A, int TRIG, class IN, class TC, class Sink> __device__ __forceinline__ static void coords_sink(const IN& in, TC& tc, Sink& sink) {\n";
  std::vector<std::vector<int>> sink_outs(d.f_ops.size());
  for (int k = 0; k < d.m; ++k) sink_outs[d.f_outs[k]].push_back(k);
  std::vector<int> put_order;                               // output index of the SEQ-th put
  mark_f();
  emit_body(o, d.f_ops.data(), (int)d.f_ops.size(), "f", &sink_outs, nullptr, "tc", nullptr, "TRIG", nullptr, &put_order);
  o << "  }\n";
  // f with a sink, followed by the potential on the same SSA values (u . f when U is cartesian):
  // no array of M outputs is ever materialised
  o << "  template <class A, int TRIG, class IN, class TC, class TCU, class Sink> __device__ __forceinline__ static A coords_sink_u(const IN& in, TC& tc, TCU& tcu, Sink& sink) {\n";
  mark_f();
  emit_body(o, d.f_ops.data(), (int)d.f_ops.size(), "f", &sink_outs);
  {
    std::vector<std::string> u_in;
    if (d.u_space == HAMK_U_CARTESIAN) for (int k = 0; k < d.m; ++k) u_in.push_back(value_name(d.f_ops, "f", d.f_outs[k]));
    else for (int j = 0; j < d.n; ++j) u_in.push_back("in[" + std::to_string(j) + "]");
    mark_u();
    emit_body(o, d.u_ops.data(), (int)d.u_ops.size(), "u", nullptr, &u_in, "tcu", nullptr, "hamk::TRIG_FULL");
    o << "    return hamk::lift<A>(" << value_name(d.u_ops, "u", d.u_out, &u_in) << ");\n";
  }
  o << "  }\n";
  // which input (or -1) each sincos site of f takes as its operand: sites fed by inputs can be
  // evaluated once per trajectory and shared (wave kernels)
  o << "  __device__ __forceinline__ static constexpr int trig_input(int slot) {\n    constexpr int w[" << (ntrig_f > 0 ? ntrig_f : 1) << "] = {";
  bool all_inputs = ntrig_f > 0;
  for (int k = 0; k < ntrig_f; ++k) {
    const int opi = slot_operand[k];
    const int inp = (opi >= 0 && d.f_ops[opi].op == HAMK_OP_INPUT) ? d.f_ops[opi].a : -1;
    if (inp < 0) all_inputs = false;
    o << (k ? ", " : "") << inp;
  }
  if (ntrig_f == 0) o << "-1";
  o << "};\n    return w[slot];\n  }\n";
  o << "  static constexpr bool TRIG_ALL_INPUTS = " << (all_inputs ? "true" : "false") << ";\n";
  if (d.wave) {
    // Structure of the Jacobian, for the wave kernels' accumulation of K = J^T M J on the matrix cores (hamk_wave.hpp SinkK):
    // which inputs output k depends on at all, as a range [dep_lo, dep_hi] of input indices (-1 / -1: none), and which output
    // the SEQ-th put of coords_sink delivers.  A 16-column block in which all four staged rows are structurally zero
    // contributes nothing and is skipped (an N-link chain: x_k, y_k depend on q_0..q_k -- half the blocks).
    const int nf = (int)d.f_ops.size();
    std::vector<int> lo(nf, 1 << 30), hi(nf, -1);
    for (int i = 0; i < nf; ++i) {
      const hamk_op& p = d.f_ops[i];
      if (p.op == HAMK_OP_CONST) continue;
      if (p.op == HAMK_OP_INPUT) { lo[i] = hi[i] = p.a; continue; }
      lo[i] = lo[p.a]; hi[i] = hi[p.a];
      if (is_binary(p.op)) { lo[i] = std::min(lo[i], lo[p.b]); hi[i] = std::max(hi[i], hi[p.b]); }
    }
    auto table = [&](const char* name, const std::vector<int>& w) {
      o << "  __device__ __forceinline__ static constexpr int " << name << "(int k) {\n    constexpr int w[" << w.size() << "] = {";
      for (size_t k = 0; k < w.size(); ++k) o << (k ? ", " : "") << w[k];
      o << "};\n    return w[k];\n  }\n";
    };
    std::vector<int> dlo(d.m), dhi(d.m);
    for (int k = 0; k < d.m; ++k) { const int v = d.f_outs[k]; dlo[k] = hi[v] < 0 ? -1 : lo[v]; dhi[k] = hi[v]; }
    table("dep_lo", dlo);
    table("dep_hi", dhi);
    table("seq_out", put_order);                            // (recorded by emit_body as it numbered the puts)
  }
  emit_reverse(o, d);
  {
    const SymK sk = d.k_symbolic ? symbolic_mass_matrix(d, slot_operand) : SymK();
    o << "  static constexpr bool HAS_SYM_K = " << (sk.ok ? "true" : "false") << ";\n";
    if (sk.ok) emit_symbolic_k(o, sk);
    else o << "  template <class TC> __device__ __forceinline__ static void mass_matrix_sym(const double (&)[N], const TC&, double (&)[N][N]) {}\n";
    o << "  static constexpr bool HAS_SYM_DT = " << ((sk.ok && sk.dt_ok) ? "true" : "false") << ";\n";
    if (sk.ok && sk.dt_ok) emit_symbolic_dt(o, sk);
    else o << "  template <class TC> __device__ __forceinline__ static void dT_sym(const double (&)[N], const double (&)[N], const TC&, double (&)[N]) {}\n";
  }
  o << "  static constexpr int NTRIG_F = " << ntrig_f << ";\n";
  o << "  static constexpr int NTRIG_U = " << ntrig_u << ";\n";
  o << "};\n\n";
  o << (d.wave ? "HAMK_INSTANTIATE_WAVE(HamkSys)\n" : (quad ? "HAMK_INSTANTIATE_QUAD(HamkSys)\n" : "HAMK_INSTANTIATE(HamkSys)\n"));
  return o.str();
}

}  // namespace hamk_host
