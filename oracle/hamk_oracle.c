/*
 * hamk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C, CPU, one-trajectory-at-a-time restatement of the algorithm of
 * mstksg/hamilton's equations-of-motion path, following the reference's
 * evaluation literally (explicit inverse, full Hessian tensor, right-associated
 * mat-vec chain), so that the HIP path -- which deliberately evaluates an
 * algebraically equivalent but differently ordered form -- has something
 * independent to be compared against.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED.  The reference itself (Haskell + ad + hmatrix + hmatrix-gsl)
 * cannot be built or run in this image (no ghc/cabal/GSL) and its own test
 * suite is a stub (/root/reference/test/Spec.hs:1-2) -- there are no golden
 * vectors to pin against.  What pins this file instead: tests/golden/ (*.json),
 * produced by oracle/gen_golden.py from an INDEPENDENT symbolic derivation
 * (sympy Hamiltonian -> dH/dp, -dH/dq, evaluated with 50-digit mpmath) and
 * high-order Taylor integration of the same ODE.
 *
 * Third-party arithmetic restated here (not vendored in /root/reference;
 * hamilton.cabal:38-46 gives lower bounds only, no pinned versions):
 *   ad            (jacobianT / hessianF / grad, Hamilton.hs:221-224)
 *                   -> exact first/second derivatives; here a dense
 *                      second-order forward-mode tape interpreter.
 *   hmatrix       (inv, <>, #>, <.>, tr, diag; Hamilton.hs:267,321-324,377-387)
 *                   -> `inv` = LAPACK dgesv against the identity: LU with
 *                      partial (row) pivoting; restated below.
 *   hmatrix-gsl   (odeSolveV RKf45, Hamilton.hs:445) -> GSL: rkf45.c stepper,
 *                   cstd.c standard controller with a_y = a_dydt = 1, evolve.c
 *                   evolve_apply, and hmatrix-gsl's gsl-ode.c output loop.
 *                   gsl-ode.c carries TWO bindings, chosen at its build time:
 *                   `#ifdef GSLODE1` the old gsl_odeiv API (a `while (t < ti)
 *                   gsl_odeiv_evolve_apply` loop), otherwise -- the default --
 *                   gsl_odeiv2 through gsl_odeiv2_driver_apply.  They differ in
 *                   the evolve rule (see evolve_apply below); both are restated,
 *                   selected per system with orc_system_set_gsl_api (default 2).
 *                   Restated from the published algorithm (SURVEY.md section 8c box
 *                   for v1; DESIGN.md section 2.4 for what odeiv2 changes).
 *
 * Index conventions (Hamilton.hs:188-192, :221-222, :227-233):
 *   J[k][i]    = d f_k / d q_i                 (m rows, n columns)
 *   Hs[i][k][j] = d^2 f_k / d q_i d q_j        (`_sysHessian q !! i` = dJ/dq_i)
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* same layout and numbering as struct hamk_op / enum hamk_opcode (include/hamk.h) */
typedef struct { int32_t op, a, b, _pad; double c; } orc_op;
enum { O_CONST = 0, O_INPUT, O_ADD, O_SUB, O_MUL, O_DIV, O_NEG, O_RECIP, O_SIN, O_COS, O_TAN,
       O_ASIN, O_ACOS, O_ATAN, O_SINH, O_COSH, O_TANH, O_EXP, O_LOG, O_SQRT, O_POWC, O_POWI,
       O_POW, O_ATAN2, O_ASINH, O_ACOSH, O_ATANH, O_ABS, O_SIGNUM, O__COUNT };

typedef struct orc_system {
  int m, n, u_space;
  double* inertia;
  orc_op* f_ops; int f_nops; int32_t* f_outs;
  orc_op* u_ops; int u_nops; int32_t u_out;
  int gsl_api;   /* 1: gsl_odeiv (hmatrix-gsl built with -DGSLODE1), 2: gsl_odeiv2 driver (its default) */
} orc_system;


/* ------------------------------------------------------------------------ */
/* per-thread stack arena: the restatement allocates many small temporaries    */
/* per call; malloc contention would otherwise serialise the OpenMP ensemble    */
/* wrappers used for the bench's cpu_baseline leg.                              */
/* ------------------------------------------------------------------------ */
#define WS_MAXCHUNK 64
static __thread char* ws_chunk[WS_MAXCHUNK];
static __thread size_t ws_cap[WS_MAXCHUNK];
static __thread int ws_cur = 0;
static __thread size_t ws_top = 0;
typedef struct { int cur; size_t top; } ws_mark_t;

static ws_mark_t ws_mark(void) { ws_mark_t m = {ws_cur, ws_top}; return m; }
static void ws_release(ws_mark_t m) { ws_cur = m.cur; ws_top = m.top; }
static void* ws_alloc(size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  for (;;) {
    if (ws_chunk[ws_cur] && ws_top + bytes <= ws_cap[ws_cur]) {
      void* p = ws_chunk[ws_cur] + ws_top; ws_top += bytes; return p;
    }
    if (ws_chunk[ws_cur] && ws_cur + 1 < WS_MAXCHUNK) { ws_cur++; ws_top = 0; }
    if (!ws_chunk[ws_cur] || ws_cap[ws_cur] < bytes) {
      size_t cap = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20);
      free(ws_chunk[ws_cur]);
      ws_chunk[ws_cur] = (char*)malloc(cap); ws_cap[ws_cur] = cap; ws_top = 0;
    }
  }
}
#define WS_ENTER ws_mark_t mk_ = ws_mark()
#define WS_LEAVE ws_release(mk_)
static void* ws_calloc(size_t bytes) { void* p = ws_alloc(bytes); memset(p, 0, bytes); return p; }

/* ------------------------------------------------------------------------ */
/* second-order forward-mode numbers: value, gradient g[n], Hessian h[n*n]    */
/* ------------------------------------------------------------------------ */
typedef struct { double v; double* g; double* h; } hd;

static void hd_unary(int n, hd* y, const hd* x, double g0, double g1, double g2) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      y->h[i * n + j] = g1 * x->h[i * n + j] + g2 * x->g[i] * x->g[j];
  for (int i = 0; i < n; i++) y->g[i] = g1 * x->g[i];
  y->v = g0;
}

static void hd_binary(int n, hd* z, const hd* a, const hd* b, double f0, double fa, double fb,
                      double faa, double fab, double fbb) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      z->h[i * n + j] = fa * a->h[i * n + j] + fb * b->h[i * n + j] + faa * a->g[i] * a->g[j] +
                        fab * (a->g[i] * b->g[j] + a->g[j] * b->g[i]) + fbb * b->g[i] * b->g[j];
  for (int i = 0; i < n; i++) z->g[i] = fa * a->g[i] + fb * b->g[i];
  z->v = f0;
}

static double ipow(double x, int k) {
  if (k < 0) return 1.0 / ipow(x, -k);
  double r = 1.0, b = x;
  while (k) { if (k & 1) r *= b; b *= b; k >>= 1; }
  return r;
}

/* Interpret a tape on second-order numbers.  in[] are n_in numbers over n directions. */
static void tape_eval(int n, const orc_op* ops, int nops, const hd* in, hd* val) {
  for (int t = 0; t < nops; t++) {
    const orc_op* o = &ops[t];
    hd* y = &val[t];
    const hd* a = (o->op != O_CONST && o->op != O_INPUT) ? &val[o->a] : NULL;
    const hd* b = NULL;
    switch (o->op) {
      case O_ADD: case O_SUB: case O_MUL: case O_DIV: case O_POW: case O_ATAN2: b = &val[o->b]; break;
      default: break;
    }
    switch (o->op) {
      case O_CONST:
        y->v = o->c; memset(y->g, 0, sizeof(double) * n); memset(y->h, 0, sizeof(double) * n * n); break;
      case O_INPUT:
        y->v = in[o->a].v; memcpy(y->g, in[o->a].g, sizeof(double) * n);
        memcpy(y->h, in[o->a].h, sizeof(double) * n * n); break;
      case O_ADD: hd_binary(n, y, a, b, a->v + b->v, 1, 1, 0, 0, 0); break;
      case O_SUB: hd_binary(n, y, a, b, a->v - b->v, 1, -1, 0, 0, 0); break;
      case O_MUL: hd_binary(n, y, a, b, a->v * b->v, b->v, a->v, 0, 1, 0); break;
      case O_DIV: {
        double ib = 1.0 / b->v, q = a->v * ib;
        hd_binary(n, y, a, b, a->v / b->v, ib, -q * ib, 0, -ib * ib, 2 * q * ib * ib);
      } break;
      case O_NEG: hd_unary(n, y, a, -a->v, -1, 0); break;
      case O_RECIP: { double r = 1.0 / a->v; hd_unary(n, y, a, r, -r * r, 2 * r * r * r); } break;
      case O_SIN: { double s = sin(a->v), c = cos(a->v); hd_unary(n, y, a, s, c, -s); } break;
      case O_COS: { double s = sin(a->v), c = cos(a->v); hd_unary(n, y, a, c, -s, -c); } break;
      case O_TAN: { double t2 = tan(a->v), d = 1 + t2 * t2; hd_unary(n, y, a, t2, d, 2 * t2 * d); } break;
      case O_ASIN: { double w = 1 - a->v * a->v, r = 1 / sqrt(w); hd_unary(n, y, a, asin(a->v), r, a->v * r / w); } break;
      case O_ACOS: { double w = 1 - a->v * a->v, r = 1 / sqrt(w); hd_unary(n, y, a, acos(a->v), -r, -a->v * r / w); } break;
      case O_ATAN: { double w = 1 + a->v * a->v; hd_unary(n, y, a, atan(a->v), 1 / w, -2 * a->v / (w * w)); } break;
      case O_SINH: { double s = sinh(a->v), c = cosh(a->v); hd_unary(n, y, a, s, c, s); } break;
      case O_COSH: { double s = sinh(a->v), c = cosh(a->v); hd_unary(n, y, a, c, s, c); } break;
      case O_TANH: { double t2 = tanh(a->v), d = 1 - t2 * t2; hd_unary(n, y, a, t2, d, -2 * t2 * d); } break;
      case O_ASINH: { double w = a->v * a->v + 1, r = 1 / sqrt(w); hd_unary(n, y, a, asinh(a->v), r, -a->v * r / w); } break;
      case O_ACOSH: { double w = a->v * a->v - 1, r = 1 / sqrt(w); hd_unary(n, y, a, acosh(a->v), r, -a->v * r / w); } break;
      case O_ATANH: { double w = 1 - a->v * a->v; hd_unary(n, y, a, atanh(a->v), 1 / w, 2 * a->v / (w * w)); } break;
      case O_ABS: { double sg = (a->v > 0) - (a->v < 0); hd_unary(n, y, a, fabs(a->v), sg, 0.0); } break;        /* ad: abs' = signum */
      case O_SIGNUM: { double sg = (a->v > 0) - (a->v < 0); hd_unary(n, y, a, sg, 0.0, 0.0); } break;
      case O_EXP: { double e = exp(a->v); hd_unary(n, y, a, e, e, e); } break;
      case O_LOG: { double r = 1.0 / a->v; hd_unary(n, y, a, log(a->v), r, -r * r); } break;
      case O_SQRT: { double r = sqrt(a->v); hd_unary(n, y, a, r, 0.5 / r, -0.25 / (r * a->v)); } break;
      case O_POWC: {
        double c = o->c;
        hd_unary(n, y, a, pow(a->v, c), c * pow(a->v, c - 1), c * (c - 1) * pow(a->v, c - 2));
      } break;
      case O_POWI: {
        int k = o->b;
        hd_unary(n, y, a, ipow(a->v, k), k * ipow(a->v, k - 1), (double)k * (k - 1) * ipow(a->v, k - 2));
      } break;
      case O_POW: { /* z = a^b = exp(b log a) */
        double z = pow(a->v, b->v), la = log(a->v), ia = 1.0 / a->v;
        double fa = b->v * z * ia, fb = z * la;
        double faa = b->v * (b->v - 1) * z * ia * ia, fbb = z * la * la, fab = z * ia * (1 + b->v * la);
        hd_binary(n, y, a, b, z, fa, fb, faa, fab, fbb);
      } break;
      case O_ATAN2: { /* z = atan2(a, b): a is "y", b is "x" */
        double r2 = a->v * a->v + b->v * b->v, i2 = 1.0 / r2;
        double fa = b->v * i2, fb = -a->v * i2;
        double faa = -2 * a->v * b->v * i2 * i2, fbb = -faa, fab = (a->v * a->v - b->v * b->v) * i2 * i2;
        hd_binary(n, y, a, b, atan2(a->v, b->v), fa, fb, faa, fab, fbb);
      } break;
      default: y->v = NAN; break;
    }
  }
}

static hd* hd_alloc(int count, int n) {
  hd* a = (hd*)ws_alloc(sizeof(hd) * (count > 0 ? count : 1));
  double* buf = (double*)ws_calloc((size_t)(count > 0 ? count : 1) * (n + n * n) * sizeof(double));
  for (int i = 0; i < count; i++) { a[i].v = 0; a[i].g = buf + (size_t)i * (n + n * n); a[i].h = a[i].g + n; }
  return a;
}

/* All derivative objects of the System record at q (Hamilton.hs:160-169, :217-225). */
typedef struct {
  double *x, *J, *Hs, *gU; double U;
} derivs;

static void sys_derivs(const orc_system* s, const double* q, derivs* d) {
  WS_ENTER;
  int n = s->n, m = s->m;
  hd* in = hd_alloc(n, n);
  for (int j = 0; j < n; j++) { in[j].v = q[j]; in[j].g[j] = 1.0; }
  hd* fv = hd_alloc(s->f_nops, n);
  tape_eval(n, s->f_ops, s->f_nops, in, fv);
  for (int k = 0; k < m; k++) {
    const hd* xk = &fv[s->f_outs[k]];
    if (d->x) d->x[k] = xk->v;
    if (d->J) for (int i = 0; i < n; i++) d->J[k * n + i] = xk->g[i];
    if (d->Hs)
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) d->Hs[((size_t)i * m + k) * n + j] = xk->h[i * n + j];
  }
  /* potential: u over q (mkSystem) or u . f (mkSystem', Hamilton.hs:254) */
  hd* uv = hd_alloc(s->u_nops, n);
  if (s->u_space == 1) {
    hd* xin = (hd*)ws_alloc(sizeof(hd) * m);
    for (int k = 0; k < m; k++) xin[k] = fv[s->f_outs[k]];
    tape_eval(n, s->u_ops, s->u_nops, xin, uv);
  } else {
    tape_eval(n, s->u_ops, s->u_nops, in, uv);
  }
  d->U = uv[s->u_out].v;
  if (d->gU) for (int i = 0; i < n; i++) d->gU[i] = uv[s->u_out].g[i];
  WS_LEAVE;
}

/* ------------------------------------------------------------------------ */
/* hmatrix `inv`: LU with partial pivoting, solved against the identity.       */
/* returns 0 ok, 1 singular (exact zero pivot, LAPACK info > 0)                */
/* ------------------------------------------------------------------------ */
static int lu_inverse(int n, const double* A, double* Ainv) {
  WS_ENTER;
  double* a = (double*)ws_alloc(sizeof(double) * n * n);
  int* piv = (int*)ws_alloc(sizeof(int) * n);
  memcpy(a, A, sizeof(double) * n * n);
  int info = 0;
  for (int c = 0; c < n; c++) {
    int p = c; double best = fabs(a[c * n + c]);
    for (int r = c + 1; r < n; r++) if (fabs(a[r * n + c]) > best) { best = fabs(a[r * n + c]); p = r; }
    piv[c] = p;
    if (!(best > 0.0)) { info = 1; break; }
    if (p != c) for (int j = 0; j < n; j++) { double t = a[c * n + j]; a[c * n + j] = a[p * n + j]; a[p * n + j] = t; }
    double ip = 1.0 / a[c * n + c];
    for (int r = c + 1; r < n; r++) {
      double l = a[r * n + c] * ip; a[r * n + c] = l;
      for (int j = c + 1; j < n; j++) a[r * n + j] -= l * a[c * n + j];
    }
  }
  if (!info) {
    for (int col = 0; col < n; col++) {
      double* b = (double*)ws_alloc(sizeof(double) * n);
      for (int i = 0; i < n; i++) b[i] = (i == col) ? 1.0 : 0.0;
      for (int c = 0; c < n; c++) if (piv[c] != c) { double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
      for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) b[i] -= a[i * n + j] * b[j];
      for (int i = n - 1; i >= 0; i--) {
        for (int j = i + 1; j < n; j++) b[i] -= a[i * n + j] * b[j];
        b[i] /= a[i * n + i];
      }
      for (int i = 0; i < n; i++) Ainv[i * n + col] = b[i];
    }
  } else {
    for (int i = 0; i < n * n; i++) Ainv[i] = NAN;
  }
  WS_LEAVE; 
  return info;
}

/* hmatrix `inv` on its own (Hamilton.hs:321, :381), exported so the CPU suite can hold it against the
 * LAPACK dgesv that hmatrix really binds (numpy.linalg.inv calls the same routine):
 * tests/test_oracle_thirdparty.py.  A, Ainv row-major n x n; returns LAPACK's info > 0 as 1. */
int orc_inverse(int n, const double* A, double* Ainv) { return lu_inverse(n, A, Ainv); }

/* jmj = trj <> mm <> j   (Hamilton.hs:380, :324) */
static void mass_matrix(const orc_system* s, const double* J, double* K) {
  int n = s->n, m = s->m;
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++) {
      double acc = 0;
      for (int k = 0; k < m; k++) acc += J[k * n + a] * s->inertia[k] * J[k * n + b];
      K[a * n + b] = acc;
    }
}

/* ------------------------------------------------------------------------ */
/* public: construction                                                         */
/* ------------------------------------------------------------------------ */
orc_system* orc_system_create(int m, int n, const double* inertia, const orc_op* f_ops, int f_nops,
                              const int32_t* f_outs, const orc_op* u_ops, int u_nops, int32_t u_out,
                              int u_space) {
  orc_system* s = (orc_system*)calloc(1, sizeof(orc_system));
  s->m = m; s->n = n; s->u_space = u_space;
  s->inertia = (double*)malloc(sizeof(double) * m); memcpy(s->inertia, inertia, sizeof(double) * m);
  s->f_ops = (orc_op*)malloc(sizeof(orc_op) * (f_nops + 1)); memcpy(s->f_ops, f_ops, sizeof(orc_op) * f_nops);
  s->f_nops = f_nops;
  s->f_outs = (int32_t*)malloc(sizeof(int32_t) * m); memcpy(s->f_outs, f_outs, sizeof(int32_t) * m);
  s->u_ops = (orc_op*)malloc(sizeof(orc_op) * (u_nops + 1)); memcpy(s->u_ops, u_ops, sizeof(orc_op) * u_nops);
  s->u_nops = u_nops; s->u_out = u_out;
  s->gsl_api = 2;
  return s;
}
void orc_system_destroy(orc_system* s) {
  if (!s) return;
  free(s->inertia); free(s->f_ops); free(s->f_outs); free(s->u_ops); free(s);
}

/* ------------------------------------------------------------------------ */
/* public: System closures (Hamilton.hs:160-186, :217-225)                      */
/* ------------------------------------------------------------------------ */
void orc_coords(const orc_system* s, const double* q, double* x) {          /* _sysCoords   :220 */
  derivs d = {x, NULL, NULL, NULL, 0}; sys_derivs(s, q, &d);
}
void orc_jacobian(const orc_system* s, const double* q, double* J) {        /* _sysJacobian :221 */
  derivs d = {NULL, J, NULL, NULL, 0}; sys_derivs(s, q, &d);
}
void orc_hessian(const orc_system* s, const double* q, double* Hs) {        /* _sysHessian  :222 */
  derivs d = {NULL, NULL, Hs, NULL, 0}; sys_derivs(s, q, &d);
}
double orc_pe(const orc_system* s, const double* q) {                       /* pe :182-186, :223 */
  derivs d = {NULL, NULL, NULL, NULL, 0}; sys_derivs(s, q, &d); return d.U;
}
void orc_grad_pe(const orc_system* s, const double* q, double* g) {         /* _sysPotentialGrad :224 */
  derivs d = {NULL, NULL, NULL, g, 0}; sys_derivs(s, q, &d);
}

/* momenta: tr j #> diag m #> j #> qd, right-associated (Hamilton.hs:262-269) */
void orc_momenta(const orc_system* s, const double* q, const double* qd, double* p) {
  WS_ENTER;
  int n = s->n, m = s->m;
  double* J = (double*)ws_alloc(sizeof(double) * m * n);
  double* w = (double*)ws_alloc(sizeof(double) * m);
  orc_jacobian(s, q, J);
  for (int k = 0; k < m; k++) { double a = 0; for (int i = 0; i < n; i++) a += J[k * n + i] * qd[i]; w[k] = a; }
  for (int k = 0; k < m; k++) w[k] = s->inertia[k] * w[k];
  for (int i = 0; i < n; i++) { double a = 0; for (int k = 0; k < m; k++) a += J[k * n + i] * w[k]; p[i] = a; }
  WS_LEAVE; 
}

/* velocities: inv jmj #> p (Hamilton.hs:316-324); returns 1 if jmj singular */
int orc_velocities(const orc_system* s, const double* q, const double* p, double* qd) {
  WS_ENTER;
  int n = s->n, m = s->m;
  double* J = (double*)ws_alloc(sizeof(double) * m * n);
  double* K = (double*)ws_alloc(sizeof(double) * n * n);
  double* Ki = (double*)ws_alloc(sizeof(double) * n * n);
  orc_jacobian(s, q, J); mass_matrix(s, J, K);
  int info = lu_inverse(n, K, Ki);
  for (int i = 0; i < n; i++) { double a = 0; for (int j = 0; j < n; j++) a += Ki[i * n + j] * p[j]; qd[i] = a; }
  WS_LEAVE; 
  return info;
}

double orc_keC(const orc_system* s, const double* q, const double* qd) {     /* :288-296 */
  WS_ENTER;
  int n = s->n; double* p = (double*)ws_alloc(sizeof(double) * n);
  orc_momenta(s, q, qd, p);
  double a = 0; for (int i = 0; i < n; i++) a += qd[i] * p[i];
  WS_LEAVE; return a / 2;
}
double orc_lagrangian(const orc_system* s, const double* q, const double* qd) { /* :301-309 */
  return orc_keC(s, q, qd) - orc_pe(s, q);
}
double orc_keP(const orc_system* s, const double* q, const double* p) {      /* :341-349 */
  WS_ENTER;
  int n = s->n; double* v = (double*)ws_alloc(sizeof(double) * n);
  orc_velocities(s, q, p, v);
  double a = 0; for (int i = 0; i < n; i++) a += v[i] * p[i];
  WS_LEAVE; return a / 2;
}
double orc_hamiltonian(const orc_system* s, const double* q, const double* p) { /* :353-361 */
  return orc_keP(s, q, p) + orc_pe(s, q);
}

/* hamEqs (Hamilton.hs:370-387), evaluated as written:
 *   dTdq_i = -(p <.> ijmj #> trj #> mm #> djdq_i #> ijmj #> p)   all infixr 8
 *   dHdp = ijmj #> p ; dHdq = dTdq + gradU ; result (dHdp, -dHdq)
 * returns 1 if jmj is singular. */
int orc_hameqs(const orc_system* s, const double* q, const double* p, double* dq, double* dp) {
  WS_ENTER;
  int n = s->n, m = s->m;
  double* J = (double*)ws_alloc(sizeof(double) * m * n);
  double* Hs = (double*)ws_alloc(sizeof(double) * (size_t)n * m * n);
  double* gU = (double*)ws_alloc(sizeof(double) * n);
  double* K = (double*)ws_alloc(sizeof(double) * n * n);
  double* Ki = (double*)ws_alloc(sizeof(double) * n * n);
  double* w1 = (double*)ws_alloc(sizeof(double) * n);
  double* w2 = (double*)ws_alloc(sizeof(double) * m);
  double* w4 = (double*)ws_alloc(sizeof(double) * n);
  derivs d = {NULL, J, Hs, gU, 0};
  sys_derivs(s, q, &d);
  mass_matrix(s, J, K);
  int info = lu_inverse(n, K, Ki);
  for (int i = 0; i < n; i++) {
    const double* dj = Hs + (size_t)i * m * n;          /* djdq = _sysHessian q !! i */
    for (int a = 0; a < n; a++) { double t = 0; for (int b = 0; b < n; b++) t += Ki[a * n + b] * p[b]; w1[a] = t; }
    for (int k = 0; k < m; k++) { double t = 0; for (int j = 0; j < n; j++) t += dj[k * n + j] * w1[j]; w2[k] = t; }
    for (int k = 0; k < m; k++) w2[k] = s->inertia[k] * w2[k];
    for (int a = 0; a < n; a++) { double t = 0; for (int k = 0; k < m; k++) t += J[k * n + a] * w2[k]; w4[a] = t; }
    double dot = 0;
    for (int a = 0; a < n; a++) { double t = 0; for (int b = 0; b < n; b++) t += Ki[a * n + b] * w4[b]; dot += p[a] * t; }
    double dTdq = -dot;
    dp[i] = -(dTdq + gU[i]);
  }
  for (int a = 0; a < n; a++) { double t = 0; for (int b = 0; b < n; b++) t += Ki[a * n + b] * p[b]; dq[a] = t; }
  WS_LEAVE; 
  return info;
}

/* ------------------------------------------------------------------------ */
/* time stepping                                                                */
/* ------------------------------------------------------------------------ */
/* f = vjoin . hamEqs s . toPs on y = [q; p] (Hamilton.hs:449-458) */
static int rhs(const orc_system* s, const double* y, double* dy) {
  return orc_hameqs(s, y, y + s->n, dy, dy + s->n);
}

/* classic RK4 (build-only; no reference counterpart -- SURVEY.md a11) */
void orc_rk4_steps(const orc_system* s, double* q, double* p, double dt, int nsteps) {
  WS_ENTER;
  int n = s->n, d = 2 * n;
  double* y = (double*)ws_alloc(sizeof(double) * d * 6);
  double *k1 = y + d, *k2 = y + 2 * d, *k3 = y + 3 * d, *k4 = y + 4 * d, *t = y + 5 * d;
  memcpy(y, q, sizeof(double) * n); memcpy(y + n, p, sizeof(double) * n);
  for (int st = 0; st < nsteps; st++) {
    rhs(s, y, k1);
    for (int i = 0; i < d; i++) t[i] = y[i] + 0.5 * dt * k1[i];
    rhs(s, t, k2);
    for (int i = 0; i < d; i++) t[i] = y[i] + 0.5 * dt * k2[i];
    rhs(s, t, k3);
    for (int i = 0; i < d; i++) t[i] = y[i] + dt * k3[i];
    rhs(s, t, k4);
    for (int i = 0; i < d; i++) y[i] += dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
  }
  memcpy(q, y, sizeof(double) * n); memcpy(p, y + n, sizeof(double) * n);
  WS_LEAVE; 
}

/* GSL rkf45.c: embedded Runge-Kutta-Fehlberg 4(5); advances with the 5th-order
 * solution; yerr = h * sum ec_i k_i; dydt_out = f(t+h, y_new).                 */
static const double ah[] = {1.0 / 4.0, 3.0 / 8.0, 12.0 / 13.0, 1.0, 1.0 / 2.0};
static const double b3[] = {3.0 / 32.0, 9.0 / 32.0};
static const double b4[] = {1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0};
static const double b5[] = {8341.0 / 4104.0, -32832.0 / 4104.0, 29440.0 / 4104.0, -845.0 / 4104.0};
static const double b6[] = {-6080.0 / 20520.0, 41040.0 / 20520.0, -28352.0 / 20520.0, 9295.0 / 20520.0,
                            -5643.0 / 20520.0};
static const double c1 = 902880.0 / 7618050.0, c3 = 3953664.0 / 7618050.0, c4 = 3855735.0 / 7618050.0,
                    c5 = -1371249.0 / 7618050.0, c6 = 277020.0 / 7618050.0;
static const double ec[] = {0.0, 1.0 / 360.0, 0.0, -128.0 / 4275.0, -2197.0 / 75240.0, 1.0 / 50.0, 2.0 / 55.0};

typedef struct { double *k1, *k2, *k3, *k4, *k5, *k6, *y0, *ytmp, *yerr, *dydt_in, *dydt_out; } rkf_ws;

static void rkf45_apply(const orc_system* s, int dim, double h, double* y, double* yerr, const double* dydt_in,
                        double* dydt_out, rkf_ws* w, long* nrhs) {
  double *k1 = w->k1, *k2 = w->k2, *k3 = w->k3, *k4 = w->k4, *k5 = w->k5, *k6 = w->k6, *yt = w->ytmp;
  memcpy(k1, dydt_in, sizeof(double) * dim);
  for (int i = 0; i < dim; i++) yt[i] = y[i] + ah[0] * h * k1[i];
  rhs(s, yt, k2);
  for (int i = 0; i < dim; i++) yt[i] = y[i] + h * (b3[0] * k1[i] + b3[1] * k2[i]);
  rhs(s, yt, k3);
  for (int i = 0; i < dim; i++) yt[i] = y[i] + h * (b4[0] * k1[i] + b4[1] * k2[i] + b4[2] * k3[i]);
  rhs(s, yt, k4);
  for (int i = 0; i < dim; i++) yt[i] = y[i] + h * (b5[0] * k1[i] + b5[1] * k2[i] + b5[2] * k3[i] + b5[3] * k4[i]);
  rhs(s, yt, k5);
  for (int i = 0; i < dim; i++)
    yt[i] = y[i] + h * (b6[0] * k1[i] + b6[1] * k2[i] + b6[2] * k3[i] + b6[3] * k4[i] + b6[4] * k5[i]);
  rhs(s, yt, k6);
  for (int i = 0; i < dim; i++) {
    const double di = c1 * k1[i] + c3 * k3[i] + c4 * k4[i] + c5 * k5[i] + c6 * k6[i];
    y[i] += h * di;
  }
  rhs(s, y, dydt_out);
  for (int i = 0; i < dim; i++)
    yerr[i] = h * (ec[1] * k1[i] + ec[3] * k3[i] + ec[4] * k4[i] + ec[5] * k5[i] + ec[6] * k6[i]);
  *nrhs += 6;
}

/* GSL cstd.c std_control_hadjust with a_y = a_dydt = 1, ord = 5 (rkf45).
 * returns -1 decrease, +1 increase, 0 unchanged. */
static int std_hadjust(int dim, double eps_abs, double eps_rel, const double* y, const double* yerr,
                       const double* yp, double* h) {
  const double S = 0.9, ord = 5.0, h_old = *h;
  double rmax = DBL_MIN;
  for (int i = 0; i < dim; i++) {
    const double D0 = eps_rel * (1.0 * fabs(y[i]) + 1.0 * fabs(h_old * yp[i])) + eps_abs;
    const double r = fabs(yerr[i]) / fabs(D0);
    if (r > rmax) rmax = r;
  }
  if (rmax > 1.1) {
    double r = S / pow(rmax, 1.0 / ord);
    if (r < 0.2) r = 0.2;
    *h = r * h_old;
    return -1;
  } else if (rmax < 0.5) {
    double r = S / pow(rmax, 1.0 / (ord + 1.0));
    if (r > 5.0) r = 5.0;
    if (r < 1.0) r = 1.0;
    *h = r * h_old;
    return 1;
  }
  return 0;
}

/* GSL evolve.c: gsl_odeiv_evolve_apply (api 1) / gsl_odeiv2_evolve_apply (api 2), one call =
 * one ACCEPTED step (or, api 2 only, a failure).  The two differ in exactly two places:
 *   (a) the step size suggested for the next call: api 1 always writes the controller's h back;
 *       api 2 does not on a final (clipped-to-t1) step -- "that step can be very small compared to
 *       the previous step" -- so the h carried to the next output time is the last unclipped one;
 *   (b) when the controller asks for a smaller step but h cannot shrink any more (the decreased h
 *       no longer changes t): api 1 keeps the step size and accepts the step; api 2 leaves y
 *       advanced, reports h and returns GSL_FAILURE (-> returns 1 here).
 * api 2 also re-uses dydt_out of the previous step as dydt_in instead of evaluating f(t0, y) again
 * (e->count > 0): the same numbers, one evaluation fewer; nrhs counts what each API really does.
 * trace (optional): per attempt (t reached, h tried, accepted ? 1 : 0), at most trace_cap triples. */
typedef struct { double* buf; long cap, n; } orc_trace;
static void trace_put(orc_trace* tr, double t, double h, double acc) {
  if (tr && tr->buf && tr->n < tr->cap) { tr->buf[3 * tr->n] = t; tr->buf[3 * tr->n + 1] = h; tr->buf[3 * tr->n + 2] = acc; }
  if (tr) tr->n++;
}
static int evolve_apply(const orc_system* s, int dim, double eps_abs, double eps_rel, double* t, double t1,
                        double* h, double* y, rkf_ws* w, long* nrhs, long* nacc, long* nrej, int have_dydt,
                        orc_trace* tr) {
  const int api2 = s->gsl_api != 1;
  const double t0 = *t;
  double h0 = *h;
  int final_step = 0;
  const double dt = t1 - t0;
  memcpy(w->y0, y, sizeof(double) * dim);
  if (api2 && have_dydt) memcpy(w->dydt_in, w->dydt_out, sizeof(double) * dim);
  else { rhs(s, y, w->dydt_in); *nrhs += 1; }
  for (;;) {
    if ((dt >= 0.0 && h0 > dt) || (dt < 0.0 && h0 < dt)) { h0 = dt; final_step = 1; } else final_step = 0;
    rkf45_apply(s, dim, h0, y, w->yerr, w->dydt_in, w->dydt_out, w, nrhs);
    if (final_step) *t = t1; else *t = t0 + h0;
    const double h_old = h0;
    const int adj = std_hadjust(dim, eps_abs, eps_rel, y, w->yerr, w->dydt_out, &h0);
    if (adj < 0) {
      const volatile double t_curr = *t;
      const volatile double t_next = (*t) + h0;
      if (fabs(h0) < fabs(h_old) && t_next != t_curr) {
        trace_put(tr, *t, h_old, 0.0);
        memcpy(y, w->y0, sizeof(double) * dim);      /* undo step, retry with smaller h0 */
        *nrej += 1;
        continue;
      } else if (api2) {
        trace_put(tr, *t, h_old, -1.0);
        *h = h0;                                     /* "notify user of step-size which caused the failure" */
        return 1;                                    /* GSL_FAILURE; y and t stay advanced */
      } else {
        h0 = h_old;                                  /* keep current step size */
      }
    }
    trace_put(tr, *t, h_old, 1.0);
    break;
  }
  *nacc += 1;
  if (!api2 || !final_step) *h = h0;
  return 0;
}

/* evolveHam (Hamilton.hs:433-462) through hmatrix-gsl's gsl-ode.c loop:
 * out is [nt][2n] = rows [q; p]; row 0 = initial state; h carries across ts.
 * h0 > 0 is taken as given, otherwise the reference's (ts[1]-ts[0])/100 (Hamilton.hs:447);
 * eps <= 0 selects 1.49012e-08 (:448).
 * api 1: `for each ti: while (t < ti) gsl_odeiv_evolve_apply` -- a repeated or decreasing time does
 *        no stepping.
 * api 2: `for each ti: gsl_odeiv2_driver_apply(d, &t, ti, y)`; driver.c: the direction is the sign of
 *        the initial step (h > 0 ? +1 : -1), the loop is `while (sign (t1 - t) > 0)`, a ti on the
 *        wrong side is GSL_EINVAL, a failing evolve_apply ends the call -- gsl-ode.c then prints
 *        "error in ode" and leaves the remaining rows as they are (uninitialised memory in the
 *        reference; here: the last state reached).  hmin = 0, hmax = DBL_MAX, nmax = 0 (defaults).
 * counts (optional, 4 longs): rhs evaluations, accepted steps, rejected steps,
 *        failure (0 none, 1 GSL_FAILURE from evolve_apply, 2 GSL_EINVAL direction). */
void orc_evolve_ham_trace(const orc_system* s, const double* q0, const double* p0, int nt, const double* ts,
                          double* out, double h0, double eps_abs, double eps_rel, long* counts, double* trace,
                          long trace_cap) {
  WS_ENTER;
  int n = s->n, dim = 2 * n;
  if (!(h0 > 0)) h0 = (ts[1] - ts[0]) / 100.0;
  if (!(eps_abs > 0)) eps_abs = 1.49012e-08;
  if (!(eps_rel > 0)) eps_rel = 1.49012e-08;
  double* buf = (double*)ws_alloc(sizeof(double) * dim * 12);
  rkf_ws w = {buf, buf + dim, buf + 2 * dim, buf + 3 * dim, buf + 4 * dim, buf + 5 * dim,
              buf + 6 * dim, buf + 7 * dim, buf + 8 * dim, buf + 9 * dim, buf + 10 * dim};
  double* y = buf + 11 * dim;
  memcpy(y, q0, sizeof(double) * n); memcpy(y + n, p0, sizeof(double) * n);
  memcpy(out, y, sizeof(double) * dim);
  double t = ts[0], h = h0;
  long nrhs = 0, nacc = 0, nrej = 0, fail = 0;
  orc_trace tr = {trace, trace_cap, 0};
  const int api2 = s->gsl_api != 1;
  const double sign = (!api2 || h0 > 0.0) ? 1.0 : -1.0;
  int have_dydt = 0;
  for (int i = 1; i < nt; i++) {
    const double ti = ts[i];
    if (api2 && !fail && sign * (ti - t) < 0.0) fail = 2;
    while (!fail && sign * (ti - t) > 0.0) {
      if (evolve_apply(s, dim, eps_abs, eps_rel, &t, ti, &h, y, &w, &nrhs, &nacc, &nrej, have_dydt, &tr)) fail = 1;
      have_dydt = 1;
    }
    memcpy(out + (size_t)i * dim, y, sizeof(double) * dim);
  }
  if (counts) { counts[0] = nrhs; counts[1] = nacc; counts[2] = nrej; counts[3] = fail; }
  WS_LEAVE;
}
void orc_evolve_ham(const orc_system* s, const double* q0, const double* p0, int nt, const double* ts,
                    double* out, double h0, double eps_abs, double eps_rel, long* counts) {
  orc_evolve_ham_trace(s, q0, p0, nt, ts, out, h0, eps_abs, eps_rel, counts, NULL, 0);
}
void orc_system_set_gsl_api(orc_system* s, int api) { s->gsl_api = (api == 1) ? 1 : 2; }
int orc_system_get_gsl_api(const orc_system* s) { return s->gsl_api; }

/* stepHam r = evolveHam on (0, r), element 1 (Hamilton.hs:400-402) */
void orc_step_ham(const orc_system* s, double r, double* q, double* p, long* counts) {
  WS_ENTER;
  int n = s->n;
  double ts[2] = {0.0, r};
  double* out = (double*)ws_alloc(sizeof(double) * 4 * n);
  orc_evolve_ham(s, q, p, 2, ts, out, 0, 0, 0, counts);
  memcpy(q, out + 2 * n, sizeof(double) * n); memcpy(p, out + 3 * n, sizeof(double) * n);
  WS_LEAVE; 
}

/* ------------------------------------------------------------------------ */
/* SoA ensemble wrappers ([n][B] like the product ABI): parity tests and the    */
/* bench's cpu_baseline leg.  threads <= 0: all cores.                          */
/* ------------------------------------------------------------------------ */
static void set_threads(int threads) {
#ifdef _OPENMP
  omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());   /* <= 0: all cores, whatever an earlier call set */
#else
  (void)threads;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

#define GATHER(dst, src, n, B, i) for (int j_ = 0; j_ < (n); j_++) (dst)[j_] = (src)[(size_t)j_ * (B) + (i)]
#define SCATTER(dst, src, n, B, i) for (int j_ = 0; j_ < (n); j_++) (dst)[(size_t)j_ * (B) + (i)] = (src)[j_]

void orc_hameqs_batch(const orc_system* s, long B, const double* q, const double* p, double* dq, double* dp,
                      int32_t* status, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64], c[64], d[64];
    GATHER(a, q, n, B, i); GATHER(b, p, n, B, i);
    int st = orc_hameqs(s, a, b, c, d);
    SCATTER(dq, c, n, B, i); SCATTER(dp, d, n, B, i);
    if (status) status[i] = st;
  }
}

void orc_to_phase_batch(const orc_system* s, long B, const double* q, const double* qd, double* p, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64], c[64];
    GATHER(a, q, n, B, i); GATHER(b, qd, n, B, i);
    orc_momenta(s, a, b, c);
    SCATTER(p, c, n, B, i);
  }
}

void orc_from_phase_batch(const orc_system* s, long B, const double* q, const double* p, double* qd,
                          int32_t* status, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64], c[64];
    GATHER(a, q, n, B, i); GATHER(b, p, n, B, i);
    int st = orc_velocities(s, a, b, c);
    SCATTER(qd, c, n, B, i);
    if (status) status[i] = st;
  }
}

void orc_observe_batch(const orc_system* s, long B, const double* q, const double* p, double* ke, double* pe,
                       double* h, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64];
    GATHER(a, q, n, B, i); GATHER(b, p, n, B, i);
    double t = orc_keP(s, a, b), u = orc_pe(s, a);
    if (ke) ke[i] = t;
    if (pe) pe[i] = u;
    if (h) h[i] = t + u;
  }
}

void orc_observe_config_batch(const orc_system* s, long B, const double* q, const double* qd, double* ke,
                              double* lag, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64];
    GATHER(a, q, n, B, i); GATHER(b, qd, n, B, i);
    double t = orc_keC(s, a, b);
    if (ke) ke[i] = t;
    if (lag) lag[i] = t - orc_pe(s, a);
  }
}

void orc_coords_batch(const orc_system* s, long B, const double* q, double* x, int threads) {
  int n = s->n, m = s->m; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], c[128];
    GATHER(a, q, n, B, i);
    orc_coords(s, a, c);
    SCATTER(x, c, m, B, i);
  }
}

void orc_rk4_steps_batch(const orc_system* s, long B, double* q, double* p, double dt, int nsteps, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < B; i++) {
    double a[64], b[64];
    GATHER(a, q, n, B, i); GATHER(b, p, n, B, i);
    orc_rk4_steps(s, a, b, dt, nsteps);
    SCATTER(q, a, n, B, i); SCATTER(p, b, n, B, i);
  }
}

/* fail (optional, [B]): 0, or the api-2 failure code of orc_evolve_ham's counts[3] */
void orc_step_ham_batch(const orc_system* s, long B, double* q, double* p, double dt, int32_t* nsub, int32_t* fail, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < B; i++) {
    double a[64], b[64]; long counts[4];
    GATHER(a, q, n, B, i); GATHER(b, p, n, B, i);
    orc_step_ham(s, dt, a, b, counts);
    SCATTER(q, a, n, B, i); SCATTER(p, b, n, B, i);
    if (nsub) nsub[i] = (int32_t)(counts[1] + counts[2] + (counts[3] == 1));   /* attempts, the failing one included */
    if (fail) fail[i] = (int32_t)counts[3];
  }
}

/* qout/pout [nt][n][B] like hamk_evolve_ham_batch */
void orc_evolve_ham_batch(const orc_system* s, long B, const double* q0, const double* p0, int nt,
                          const double* ts, double* qout, double* pout, double h0, double eps_abs,
                          double eps_rel, int32_t* nsub, int32_t* fail, int threads) {
  int n = s->n; set_threads(threads);
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < B; i++) {
    double a[64], b[64]; long counts[4];
    double* out = (double*)malloc(sizeof(double) * 2 * n * nt);
    GATHER(a, q0, n, B, i); GATHER(b, p0, n, B, i);
    orc_evolve_ham(s, a, b, nt, ts, out, h0, eps_abs, eps_rel, counts);
    for (int r = 0; r < nt; r++) {
      SCATTER(qout + (size_t)r * n * B, out + (size_t)r * 2 * n, n, B, i);
      SCATTER(pout + (size_t)r * n * B, out + (size_t)r * 2 * n + n, n, B, i);
    }
    if (nsub) nsub[i] = (int32_t)(counts[1] + counts[2] + (counts[3] == 1));   /* attempts, the failing one included */
    if (fail) fail[i] = (int32_t)counts[3];
    free(out);
  }
}
