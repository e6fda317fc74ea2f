// The six example systems of the reference (app/Examples.hs:61-183, CLI defaults :230-359) written
// against include/hamilton.hpp -- the C++ spelling of `forall a. RealFloat a => ...` -- and their
// recorded tapes dumped as hex, one line per tape: "<name> f|u <outs...> : <24 bytes per op>".
// tests/test_recorders.py compares them byte for byte with the Python recorder's.  TEST INFRASTRUCTURE.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "hamilton.hpp"

using hamilton::Var;
using Vec = std::vector<Var>;

// logistic pos ht width x (Examples.hs:601-605); beta in fp64 as written
static Var logistic(double pos, double ht, double width, const Var& x) {
  const double beta = std::log(0.9 / (1 - 0.9)) / width;
  return ht / (1 + hamilton::exp(-(beta * (x - pos))));
}
static long choose(int n, int k) {
  auto fact = [](int v) { long r = 1; for (int i = 2; i <= v; ++i) r *= i; return r; };
  return fact(n) / (fact(n - k) * fact(k));
}
// bezierCurve (Examples.hs:607-627)
static Vec bezier_curve(const std::vector<std::pair<double, double>>& ps, const Var& t) {
  const int npts = (int)ps.size() - 1;
  Vec acc{Var(0.0), Var(0.0)};
  for (int i = 0; i <= npts; ++i) {
    const Var w = (double)choose(npts, i) * hamilton::powi(1 - t, npts - i) * hamilton::powi(t, i);
    acc = Vec{acc[0] + ps[i].first * w, acc[1] + ps[i].second * w};
  }
  return acc;
}

static void dump(const char* name, const char* which, const std::vector<hamk_op>& ops, const std::vector<int32_t>& outs) {
  std::printf("%s %s", name, which);
  for (int32_t o : outs) std::printf(" %d", o);
  std::printf(" :");
  for (const hamk_op& o : ops) {
    unsigned char b[sizeof(hamk_op)];
    hamk_op z; std::memset(&z, 0, sizeof z); z.op = o.op; z.a = o.a; z.b = o.b; z.c = o.c;
    std::memcpy(b, &z, sizeof z);
    std::printf(" ");
    for (unsigned char x : b) std::printf("%02x", x);
  }
  std::printf("\n");
}
static void dump(const char* name, const hamilton::System& s) {
  dump(name, "f", s.f_tape(), s.f_outs());
  dump(name, "u", s.u_tape(), s.u_outs());
}

int main() {
  using hamilton::sin; using hamilton::cos;
  {  // pendulum (Examples.hs:61-73)
    auto s = hamilton::mkSystemP({1.0, 1.0}, 1, [](const Vec& q) { return Vec{sin(q[0]), 0.5 - cos(q[0])}; },
                                 [](const Vec& x) { return x[1]; });
    dump("pendulum", s);
  }
  {  // doublePendulum 1 1 (:75-94)
    const double m1 = 1.0, m2 = 1.0;
    auto s = hamilton::mkSystemP({m1, m1, m2, m2}, 2,
                                 [](const Vec& q) {
                                   const Var &t1 = q[0], &t2 = q[1];
                                   return Vec{sin(t1), 1 - cos(t1), sin(t1) + sin(t2) / 2, 1 - cos(t1) - cos(t2) / 2};
                                 },
                                 [=](const Vec& x) { return 5 * (m1 * x[1] + m2 * x[3]); });
    dump("doublePendulum", s);
  }
  {  // room (:96-116)
    auto s = hamilton::mkSystem({1.0, 1.0}, 2, [](const Vec& q) { return Vec{q[0], q[1]}; },
                                [](const Vec& q) {
                                  const Var &x = q[0], &y = q[1];
                                  return 2 * y + (1 - logistic(-1, 10, 0.1, y)) + logistic(1, 10, 0.1, y) +
                                         (1 - logistic(-2, 10, 0.1, x)) + logistic(2, 10, 0.1, x);
                                });
    dump("room", s);
  }
  {  // twoBody 5 0.5 (:118-142)
    const double m1 = 5.0, m2 = 0.5, mT = m1 + m2;
    auto s = hamilton::mkSystem({m1, m1, m2, m2}, 2,
                                [=](const Vec& q) {
                                  const Var &r = q[0], &th = q[1];
                                  const Var r1 = r * (-(m2 / mT));
                                  const Var r2 = r * (m1 / mT);
                                  return Vec{r1 * cos(th), r1 * sin(th), r2 * cos(th), r2 * sin(th)};
                                },
                                [=](const Vec& q) { return -((m1 * m2) / q[0]); });
    dump("twoBody", s);
  }
  {  // spring 2 1 10 (:144-162)
    const double mB = 2.0, mW = 1.0, k = 10.0;
    auto s = hamilton::mkSystem({mB, mW, mW}, 3,
                                [](const Vec& q) {
                                  const Var &r = q[0], &x = q[1], &th = q[2];
                                  return Vec{r, r + (1 + x) * sin(th), (1 + x) * (-cos(th))};
                                },
                                [=](const Vec& q) {
                                  const Var &r = q[0], &x = q[1], &th = q[2];
                                  return k * hamilton::pow(x, 2) / 2 + (1 - logistic(-1.5, 25, 0.1, r)) + logistic(1.5, 25, 0.1, r) +
                                         mB * ((1 + x) * (-cos(th)));
                                });
    dump("spring", s);
  }
  {  // bezier (:164-183), default control points :350
    const std::vector<std::pair<double, double>> ps = {{-1, -1}, {-2, 1}, {0, 1}, {1, -1}, {2, 1}};
    auto s = hamilton::mkSystem({1.0, 1.0}, 1, [=](const Vec& q) { return bezier_curve(ps, q[0]); },
                                [](const Vec& q) { return (1 - logistic(0, 5, 0.05, q[0])) + logistic(1, 5, 0.05, q[0]); });
    dump("bezier", s);
  }
  return 0;
}
