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
void emu_frcp(const double* x, double* r, long long n) { for (long long i = 0; i < n; ++i) r[i] = hamk::frcp(x[i]); }
void emu_rpow(const double* x, double* r5, double* r6, long long n) {
  for (long long i = 0; i < n; ++i) { r5[i] = hamk::rpow_inv<5>(x[i]); r6[i] = hamk::rpow_inv<6>(x[i]); }
}
}
