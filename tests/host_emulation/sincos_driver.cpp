// TEST INFRASTRUCTURE: exposes the library's own fp64 sincos routines (hamk_device.hpp) to the CPU
// suite through the host shim.
#include "hamk_device.hpp"
extern "C" {
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
// a chain of `len` anchor rotations through the library's own TRIG_DYN logic: full anchor at x0, then
// x0 + d, x0 + 2 d, ... each obtained by rotating the previous anchor (which it then replaces);
// s, c: the pair after the last rotation
void emu_sincos_chain(const double* x0, const double* d, int len, double* s, double* c, long long n) {
  for (long long i = 0; i < n; ++i) {
    hamk::TrigCache<1> tc;
    tc.mode = hamk::DYN_FULL_ANCHOR;
    double x = x0[i];
    hamk::trig_pair<hamk::TRIG_DYN>(x, tc, 0);
    tc.mode = hamk::DYN_CHAIN;
    for (int k = 0; k < len; ++k) { x += d[i]; hamk::trig_pair<hamk::TRIG_DYN>(x, tc, 0); }
    s[i] = tc.s[0]; c[i] = tc.c[0];
  }
}
void emu_frcp(const double* x, double* r, long long n) { for (long long i = 0; i < n; ++i) r[i] = hamk::frcp(x[i]); }
void emu_rpow(const double* x, double* r5, double* r6, long long n) {
  for (long long i = 0; i < n; ++i) { r5[i] = hamk::rpow_inv<5>(x[i]); r6[i] = hamk::rpow_inv<6>(x[i]); }
}
}
