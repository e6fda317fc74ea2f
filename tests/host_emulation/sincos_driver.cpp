// TEST INFRASTRUCTURE: exposes the library's own fp64 sincos routines (hamk_device.hpp) to the CPU
// suite through the host shim.
#include <cmath>
static double hamk_trig_lut_init[1024];          // what hamk_codegen.cpp emits into every generated system
namespace { struct LutInit { LutInit() {
  for (int i = 0; i < 512; ++i) {
    const long double a = (long double)i * (2.0L * 3.14159265358979323846264338327950288L / 512.0L);
    hamk_trig_lut_init[2 * i] = (double)sinl(a); hamk_trig_lut_init[2 * i + 1] = (double)cosl(a);
  } } } lut_init_; }
#include "hamk_device.hpp"
extern "C" {
void emu_sincos_lut(const double* x, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) hamk::sincos_lut(x[i], s[i], c[i]);
}
void emu_sincos(const double* x, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) hamk::sincos_f64(x[i], s[i], c[i]);
}
// anchor at xa (full evaluation), incremental evaluation at xa + d
void emu_sincos_incr(const double* xa, const double* d, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) {
    double sa, ca;
    hamk::sincos_f64(xa[i], sa, ca);
    hamk::sincos_incr(xa[i] + d[i], xa[i], sa, ca, s[i], c[i]);
  }
}
// anchor at xa (full evaluation), rotation to xa + d with the given range (0 wide 1/4, 1 narrow 1/8, 2 short 1/32)
void emu_sincos_incr_range(const double* xa, const double* d, int range, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) {
    double sa, ca;
    hamk::sincos_f64(xa[i], sa, ca);
    if (range == 0) hamk::sincos_incr<hamk::INCR_WIDE>(xa[i] + d[i], xa[i], sa, ca, s[i], c[i]);
    else if (range == 1) hamk::sincos_incr<hamk::INCR_NARROW>(xa[i] + d[i], xa[i], sa, ca, s[i], c[i]);
    else hamk::sincos_incr<hamk::INCR_SHORT>(xa[i] + d[i], xa[i], sa, ca, s[i], c[i]);
  }
}
// the four evaluations of one RK4 step through the library's own TRIG_DYN logic: full anchor at x,
// narrow rotation to x + d1 (becomes the anchor), short rotation to x + d1 + d2, narrow rotation to
// x + d1 + d3; s[4], c[4] per point
void emu_sincos_step(const double* x, const double* d1, const double* d2, const double* d3, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) {
    hamk::TrigCache<1> tc;
    const double pts[4] = {x[i], x[i] + d1[i], x[i] + d1[i] + d2[i], x[i] + d1[i] + d3[i]};
    const int modes[4] = {hamk::DYN_FULL_ANCHOR, hamk::DYN_NARROW_ANCHOR, hamk::DYN_SHORT, hamk::DYN_NARROW};
    for (int k = 0; k < 4; ++k) {
      tc.mode = modes[k];
      hamk::trig_pair<hamk::TRIG_DYN>(pts[k], tc, 0);
      s[4 * i + k] = tc.s[0]; c[4 * i + k] = tc.c[0];
    }
  }
}
void emu_frcp(const double* x, double* r, long long n) { for (long long i = 0; i < n; ++i) r[i] = hamk::frcp(x[i]); }
void emu_rpow(const double* x, double* r5, double* r6, long long n) {
  for (long long i = 0; i < n; ++i) { r5[i] = hamk::rpow_inv<5>(x[i]); r6[i] = hamk::rpow_inv<6>(x[i]); }
}
}
