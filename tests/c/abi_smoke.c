/* Plain C99 client of include/hamk.h: the header must be usable without C++ (the Haskell FFI and a
 * cgo/ctypes binding see exactly this).  Builds the reference's pendulum (app/Examples.hs:61-73:
 * one coordinate theta, x = (sin theta, 0.5 - cos theta), U = y on the cartesian side) as a tape BY
 * HAND -- in the canonical form every recorder ships (depth-first post-order from the outputs;
 * tests/test_recorders.py checks it is byte for byte what the Python and C++ recorders emit) --
 * specialises it (hiprtc cross-compiles without a GPU), and -- with "run" and a GPU -- evaluates
 * hamEqs for one phase. */
#include <stdio.h>
#include <string.h>
#include "hamk.h"

static hamk_op op(int32_t code, int32_t a, int32_t b, double c) {
  hamk_op o;
  memset(&o, 0, sizeof o);
  o.op = code; o.a = a; o.b = b; o.c = c;
  return o;
}

int main(int argc, char** argv) {
  hamk_op f[5], u[1];
  int32_t f_outs[2] = {1, 4};
  double inertia[2] = {1.0, 1.0};
  hamk_system* s = NULL;
  int32_t m = 0, n = 0;
  int rc;
  f[0] = op(HAMK_OP_INPUT, 0, 0, 0.0);         /* theta                  */
  f[1] = op(HAMK_OP_SIN, 0, 0, 0.0);           /* x = sin theta          */
  f[2] = op(HAMK_OP_CONST, 0, 0, 0.5);
  f[3] = op(HAMK_OP_COS, 0, 0, 0.0);
  f[4] = op(HAMK_OP_SUB, 2, 3, 0.0);           /* y = 0.5 - cos theta    */
  u[0] = op(HAMK_OP_INPUT, 1, 0, 0.0);         /* U = y (cartesian)      */
  printf("%s, %d device(s)\n", hamk_version(), hamk_device_count());
  rc = hamk_system_create(2, 1, inertia, f, 5, f_outs, u, 1, 0, HAMK_U_CARTESIAN, &s);
  if (rc != HAMK_OK) { fprintf(stderr, "create: %d %s\n", rc, hamk_last_error()); return 1; }
  if (hamk_system_dims(s, &m, &n) != HAMK_OK || m != 2 || n != 1) return 2;
  printf("System %d %d, code %lld bytes\n", (int)m, (int)n, (long long)hamk_system_code_size(s));
  if (argc > 1 && strcmp(argv[1], "run") == 0) {
    double q = 0.3, p = 0.0, dq = 9.0, dp = 9.0;
    int32_t st = -1;
    rc = hamk_hameqs_batch(s, 1, &q, &p, &dq, &dp, &st, HAMK_MEM_HOST);
    if (rc != HAMK_OK) { fprintf(stderr, "hamEqs: %d %s\n", rc, hamk_last_error()); return 3; }
    printf("hamEqs dq = %.17g dp = %.17g status = %d\n", dq, dp, (int)st);   /* dq = 0, dp = -sin 0.3 */
  }
  hamk_system_destroy(s);
  return 0;
}
