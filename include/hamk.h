/*
 * hamk.h -- C ABI of libhamk.so: the MI355X (gfx950) replacement for the
 * equations-of-motion hot path of mstksg/hamilton (Numeric.Hamilton).
 *
 * Everything the reference does between "user hands over coordinate map f and
 * potential U" and "here is the next Phase" is behind these entry points:
 *
 *   reference (src/Numeric/Hamilton.hs)                 this ABI
 *   -------------------------------------------------   -------------------------
 *   mkSystem  :201-225 / mkSystem' :238-254             hamk_system_create
 *   System record :160-169 (opaque, GC-owned)           hamk_system (opaque handle) / hamk_system_destroy
 *   underlyingPos :174-178                              hamk_coords_batch
 *   pe            :182-186                              hamk_observe_batch (pe output)
 *   momenta :262-269, toPhase :279-284                  hamk_to_phase_batch
 *   velocities :316-324, fromPhase :332-337             hamk_from_phase_batch
 *   keC :288-296, lagrangian :301-309                   hamk_observe_config_batch
 *   keP :341-349, hamiltonian :353-361                  hamk_observe_batch
 *   hamEqs :370-387                                     hamk_hameqs_batch
 *   stepHam :390-402                                    hamk_step_ham_batch   (adaptive RKF45, GSL semantics)
 *   iterate (stepHam dt)  README.md:150, Examples.hs:429 hamk_step_ham_iterate (ncalls calls, one launch)
 *   evolveHam :433-462, evolveHam' :409-429             hamk_evolve_ham_batch (adaptive RKF45, GSL semantics)
 *   odeSolveV's GSL binding (:445, hmatrix-gsl)         hamk_system_set_gsl_api (gsl_odeiv2 driver | old gsl_odeiv)
 *   (no counterpart; named by BASELINE.json north_star) hamk_rk4_steps        (classic fixed-step RK4)
 *   (no counterpart)            hamk_rk4_steps_checked, hamk_checkpoint_*
 *
 * The reference evaluates ONE trajectory per call on the CPU through `ad`
 * (AD), hmatrix (LAPACK/BLAS) and hmatrix-gsl (GSL odeiv).  This library
 * evaluates an ENSEMBLE of B independent trajectories per call on the GPU:
 * one trajectory per wavefront lane up to n = 16, four lanes per trajectory
 * (sparse coordinate maps up to n = 32, and small ensembles of mid-size
 * systems) or 16 / 32 / 64 lanes per trajectory beyond -- chosen per launch
 * from (n, B), see hamk_options::mapping.  B = 1 reproduces the reference API.
 *
 * Data layout: every state array is structure-of-arrays, fp64, component
 * major: q[j*B + i] is generalized coordinate j of trajectory i (j < n,
 * i < B).  Arrays are caller-owned; nothing is retained after return.
 * `mem` says where the caller's pointers live: HAMK_MEM_HOST (PCIe-inclusive:
 * small arrays go through a pinned arena the kernel reads and writes directly,
 * large ones are staged through device memory; the call returns when the
 * results are in the caller's arrays) or HAMK_MEM_DEVICE (pointers are HIP
 * device pointers on the current device -- from the host's own runtime or from
 * hamk_device_malloc; the launch is asynchronous on the handle's stream, see
 * hamk_set_stream).
 *
 * User functions cross the ABI as expression tapes (hamk_op[]): the host
 * shim instantiates the reference's rank-2 polymorphic functions
 * (`forall a. RealFloat a => Vector n a -> Vector m a`, Hamilton.hs:212,215)
 * at a recording number type and ships the recording.  The library
 * specialises its hand-written device kernels on the tape at
 * hamk_system_create time (hiprtc, gfx950) -- the device-side AD that
 * replaces `jacobianT`/`hessianF`/`grad` (Hamilton.hs:221-224).
 *
 * Error convention: every entry point returns 0 on success, <0 on API /
 * toolchain / HIP failure (text via hamk_last_error(), thread-local).  This
 * replaces the reference's `error`/`fromJust`/hmatrix exceptions
 * (Hamilton.hs:425,444,462,321,381).  Numerical trouble is per trajectory in
 * status[B] (bit mask, HAMK_ST_*), never an exception.
 *
 * Threading: one handle is used by one host thread at a time; distinct
 * handles may be used concurrently.  A handle runs on the device that is
 * current (hamk_set_device) when it is first used; one handle per device is the
 * way to drive several GPUs from one process.
 *
 * First use of a handle on a device runs a short self-check of its stepping
 * kernels against its hamEqs kernel (a JIT product does not take the code
 * generator's word for it); HAMK_SELFCHECK=0 skips it.
 */
#ifndef HAMK_H
#define HAMK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes ------------------------------------------------------ */
#define HAMK_OK               0
#define HAMK_ERR_INVALID     -1   /* bad argument (null pointer, n<=0, ...)  */
#define HAMK_ERR_TAPE        -2   /* malformed tape (forward ref, bad op)    */
#define HAMK_ERR_COMPILE     -3   /* hiprtc specialisation failed            */
#define HAMK_ERR_HIP         -4   /* HIP runtime error                       */
#define HAMK_ERR_NODEVICE    -5   /* no gfx950 device visible                */
#define HAMK_ERR_UNSUPPORTED -6   /* (m,n) outside what the kernels support  */

/* ---- per-trajectory status bits --------------------------------------- */
#define HAMK_ST_SINGULAR   1  /* mass matrix K = J^T M J not invertible (reference: hmatrix `inv` throws) */
#define HAMK_ST_NONFINITE  2  /* state became NaN/Inf                          */
#define HAMK_ST_UNDERFLOW  4  /* adaptive step could not advance time (h -> 0); with gsl_odeiv2 semantics:
                                 GSL_FAILURE out of evolve_apply -- the trajectory stops where it is         */
#define HAMK_ST_MAXSTEPS   8  /* adaptive stepper hit its sub-step budget      */
#define HAMK_ST_DRIFT     16  /* hamk_rk4_steps_checked: the launch lost its energy invariant */

/* ---- where caller pointers live ---------------------------------------- */
#define HAMK_MEM_HOST    0
#define HAMK_MEM_DEVICE  1

/* ---- which space the potential's tape is written in --------------------- */
#define HAMK_U_GENERALIZED 0   /* mkSystem : U(q), n inputs  (Hamilton.hs:201-225) */
#define HAMK_U_CARTESIAN   1   /* mkSystem': U(x), m inputs  (Hamilton.hs:238-254) */

/* ---- expression tape ----------------------------------------------------
 * Op i defines value i; operands a,b are indices of EARLIER values (SSA).
 * The set covers the Num/Fractional/Floating/RealFloat methods a traced
 * Haskell function can emit without comparisons (abs and signum included: they
 * are Num methods; what cannot be recorded is a BRANCH on a value).                            */
enum hamk_opcode {
  HAMK_OP_CONST = 0,  /* c                                  */
  HAMK_OP_INPUT = 1,  /* input[a]                           */
  HAMK_OP_ADD   = 2,  /* v[a] + v[b]                        */
  HAMK_OP_SUB   = 3,  /* v[a] - v[b]                        */
  HAMK_OP_MUL   = 4,  /* v[a] * v[b]                        */
  HAMK_OP_DIV   = 5,  /* v[a] / v[b]                        */
  HAMK_OP_NEG   = 6,  /* -v[a]                              */
  HAMK_OP_RECIP = 7,  /* 1 / v[a]                           */
  HAMK_OP_SIN   = 8,
  HAMK_OP_COS   = 9,
  HAMK_OP_TAN   = 10,
  HAMK_OP_ASIN  = 11,
  HAMK_OP_ACOS  = 12,
  HAMK_OP_ATAN  = 13,
  HAMK_OP_SINH  = 14,
  HAMK_OP_COSH  = 15,
  HAMK_OP_TANH  = 16,
  HAMK_OP_EXP   = 17,
  HAMK_OP_LOG   = 18,
  HAMK_OP_SQRT  = 19,
  HAMK_OP_POWC  = 20, /* v[a] ** c   (constant real exponent; valid for v[a]<0 when c is integral) */
  HAMK_OP_POWI  = 21, /* v[a] ^  b   (b = integer exponent, may be negative)                       */
  HAMK_OP_POW   = 22, /* v[a] ** v[b] (both variable; requires v[a] > 0)                           */
  HAMK_OP_ATAN2 = 23, /* atan2(v[a], v[b])                                                         */
  HAMK_OP_ASINH = 24,
  HAMK_OP_ACOSH = 25,
  HAMK_OP_ATANH = 26,
  HAMK_OP_ABS    = 27, /* |v[a]|     (Num.abs; derivative signum, second derivative 0: not differentiable at 0) */
  HAMK_OP_SIGNUM = 28, /* signum v[a] (Num.signum: -1, 0, +1; derivatives 0)                                   */
  HAMK_OP__COUNT
};

typedef struct hamk_op {
  int32_t op;   /* enum hamk_opcode */
  int32_t a;    /* first operand / input index */
  int32_t b;    /* second operand / integer exponent */
  int32_t _pad;
  double  c;    /* constant (CONST, POWC) */
} hamk_op;

typedef struct hamk_system hamk_system;   /* opaque */

/* ---- options of a system (hamk_system_create_ex) -----------------------------------------------------
 * Everything the library decides for itself when it specialises its kernels for a system can be fixed by the host
 * instead.  HAMK_AUTO (0) in a field leaves that decision to the library.  `size` AND `version` are mandatory: start from
 * hamk_options_init, which sets both and leaves every choice AUTO -- a zero-filled struct with only `size` set (or one built
 * against a header of another layout revision) is refused with HAMK_ERR_INVALID.  The environment variables of DESIGN.md section 7 are TEST
 * overrides: they are read only in a process that sets HAMK_TEST_OVERRIDES=1, and apply only where the field is
 * HAMK_AUTO.  hamk_system_get_options reports what was actually chosen.                                           */
#define HAMK_AUTO 0
#define HAMK_ON   1
#define HAMK_OFF  2
/* which lanes serve one trajectory */
#define HAMK_MAP_LANE 1   /* one trajectory per wavefront lane, everything in registers (n <= 16)                      */
#define HAMK_MAP_WAVE 2   /* wave-cooperative: 16 / 32 / 64 lanes per trajectory, one AD direction per lane, K and its
                             factorisation on the matrix cores and in LDS (any n <= 64; the only mapping for n > 32)   */
#define HAMK_MAP_QUAD 3   /* four lanes per trajectory: every lane runs the sparse per-trajectory AD sweeps, the rows of
                             K are dealt out over the four lanes and factorised in registers with DPP exchanges
                             (17 <= n <= 32 when the coordinate map's Jacobian is sparse, or dense but cheap to evaluate with
                             compile-time seeds -- K is then accumulated in tiles; also usable for n <= 16)              */
/* second-order AD strategy */
#define HAMK_AD_H 1       /* one sweep of full second-order jets                        */
#define HAMK_AD_D 2       /* first-order sweep, then a directional second-order sweep   */
#define HAMK_AD_R 3       /* first-order sweep, then a generated reverse sweep          */
/* stepping-loop bodies */
#define HAMK_BODY_UNROLLED   1
#define HAMK_BODY_STAGE_LOOP 2
/* sincos of the stepping kernels */
#define HAMK_TRIG_DIRECT       1   /* sincos_f64, no table                              */
#define HAMK_TRIG_TABLE        2   /* every evaluation through the 512-pair table in LDS */
#define HAMK_TRIG_TABLE_ROTATE 3   /* one table evaluation per step, stages 2-4 by rotation */
/* which of the two builds of a module its kernels are taken from */
#define HAMK_BUILD_DEFAULT 1
#define HAMK_BUILD_NOLICM  2       /* -mllvm -disable-machine-licm                      */

#define HAMK_OPTIONS_VERSION 0x484b0005u   /* layout revision of hamk_options ('H' 'K' 0x0005: round 5 dropped the dead field
                                              wave_blocked); a struct of another revision is refused with HAMK_ERR_INVALID   */
typedef struct hamk_options {
  uint32_t size;           /* sizeof(hamk_options) as the caller's header has it (hamk_options_init sets it)           */
  uint32_t version;        /* HAMK_OPTIONS_VERSION (hamk_options_init sets it): fields were removed since the first layout,
                              so `size` alone no longer identifies what the caller's header meant                       */
  int32_t mapping;         /* HAMK_MAP_*; AUTO: chosen PER LAUNCH from (n, ensemble size B): a small ensemble of a
                              mid-size system cannot fill the chip with one trajectory per lane.  Two mappings agree to
                              roundoff, not bitwise: pin the mapping where results must not depend on how an ensemble
                              is split into launches (hamk_system_get_options tells what AUTO picks for a size)         */
  int32_t ad_mode;         /* HAMK_AD_*                                                                                 */
  int32_t rk4_body;        /* HAMK_BODY_*                                                                               */
  int32_t rkf_body;        /* HAMK_BODY_*                                                                               */
  int32_t trig;            /* HAMK_TRIG_*                                                                               */
  int32_t gsl_api;         /* 1 | 2 (hamk_system_set_gsl_api); AUTO: 2                                                  */
  int32_t self_check;      /* ON | OFF: first-use self-check of the stepping kernels; AUTO: ON                          */
  int32_t build;           /* HAMK_BUILD_*; AUTO: per kernel, the build that spills fewer SGPRs                         */
  int32_t rk4_min_waves;   /* __launch_bounds__ waves per SIMD of the RK4 kernel; AUTO: measured default                */
  int32_t k_reassoc;       /* ON | OFF: K = J^T M J summed with re-association allowed (repeated Jacobian entries are
                              multiplied by their count instead of added up); AUTO: ON                                  */
  int32_t rk4_park;        /* ON | OFF: lane mapping, RK4 stage loop keeps y and the running combination in LDS across
                              the right-hand side; AUTO: n >= 14                                                        */
  int32_t max_substeps;    /* sub-step budget per stepHam / evolveHam CALL and trajectory (an evolveHam over nt times
                              shares one budget across its intervals; every call of `iterate` has its own); AUTO: 2^24  */
  int32_t cache;           /* ON | OFF: on-disk cache of compiled code objects; AUTO: ON                                */
  int32_t lanes_per_trajectory;  /* OUTPUT of hamk_system_get_options: 1, 4, 16, 32 or 64                               */
  int32_t rkf_park;        /* ON | OFF: QUAD mapping, the adaptive stepper's vectors (y, dydt, k2..k6, trial state) wait in
                              LDS and in a run-time-indexed private array instead of competing with the right-hand side
                              for registers; AUTO: n >= 17.  Lane mapping: follows rkf_body (the stage-loop body IS the
                              parked one since round 4): reported; a value that contradicts rkf_body is refused with
                              HAMK_ERR_UNSUPPORTED when the lane specialisation is built                                 */
  int32_t _align;          /* keeps ensemble_size 8-byte aligned; 0                                                     */
  int64_t ensemble_size;   /* mapping = AUTO only: the size of the WHOLE ensemble this handle's launches are pieces of (a shard
                              of a multi-GPU run, a chunk of a host loop).  AUTO picks the mapping from the ensemble size, and
                              two mappings agree to roundoff, not bitwise -- so a host that states the whole ensemble's size
                              ONCE (here, or hamk_system_set_ensemble_size) gets every piece computed by the mapping chosen
                              for the whole: any split of the ensemble, over any number of GPUs, reproduces the one-launch
                              bits.  0 (AUTO): every launch stands for itself (mapping from its own B)                     */
  int32_t reserved[12];           /* sizeof(hamk_options) = 128 */
} hamk_options;

/* Zero-fills *opt and sets opt->size.                                                                                 */
void hamk_options_init(hamk_options* opt);

/* ---- system construction (mkSystem / mkSystem') --------------------------
 * inertia[m]; coordinate map f: n inputs -> m outputs f_outs[m] (value ids in
 * f_ops); potential u: scalar output u_out (value id in u_ops) over n
 * (HAMK_U_GENERALIZED) or m (HAMK_U_CARTESIAN) inputs.  Compiles the device
 * module for gfx950; does not need a GPU until the first *_batch call.      */
int hamk_system_create(int32_t m, int32_t n, const double* inertia,
                       const hamk_op* f_ops, int32_t f_nops, const int32_t* f_outs,
                       const hamk_op* u_ops, int32_t u_nops, int32_t u_out,
                       int32_t u_space, hamk_system** out);
/* The same with options (NULL = all defaults = hamk_system_create).  Builds the specialisation the options name; with
 * mapping = HAMK_AUTO the one a large ensemble uses (others are built the first time a launch needs them).            */
int hamk_system_create_ex(int32_t m, int32_t n, const double* inertia,
                          const hamk_op* f_ops, int32_t f_nops, const int32_t* f_outs,
                          const hamk_op* u_ops, int32_t u_nops, int32_t u_out,
                          int32_t u_space, const hamk_options* opt, hamk_system** out);
/* What a launch over B trajectories uses (every field resolved, no HAMK_AUTO left; lanes_per_trajectory filled in).
 * Builds that specialisation if it does not exist yet.                                                                */
int hamk_system_get_options(hamk_system* s, int64_t B, hamk_options* resolved);
/* Makes the introspection entry points below (source, code_size, code_object, build_info, kernel_bytes) describe the
 * specialisation a launch over B trajectories uses (default: the one built at creation).                              */
int hamk_system_describe_batch(hamk_system* s, int64_t B);
/* hamk_options::ensemble_size after creation (0: back to per-launch choice).  A run sharded over G GPUs, or resumed from
 * a checkpoint with another G, calls this with the WHOLE ensemble's size on every handle: results are then independent of
 * the shard layout bit for bit.  No effect where the mapping is fixed by hamk_options::mapping. */
int hamk_system_set_ensemble_size(hamk_system* s, int64_t B_total);
void hamk_system_destroy(hamk_system* s);
int  hamk_system_dims(const hamk_system* s, int32_t* m, int32_t* n);

/* Launch on this HIP stream (hipStream_t as void*; NULL = default stream).  A stream belongs to one device: the
 * binding is made for the device that is CURRENT when this is called and used whenever the handle runs on that
 * device; other devices keep their own (default: the null stream).                                                 */
int hamk_set_stream(hamk_system* s, void* hip_stream);
/* Block until everything queued on the handle's stream of the current device has finished.                         */
int hamk_synchronize(hamk_system* s);

/* Which of the two GSL bindings in hmatrix-gsl's gsl-ode.c stepHam / evolveHam reproduce
 * (`odeSolveV`, Hamilton.hs:445):
 *   2 (default; gsl-ode.c's default build): gsl_odeiv2 -- gsl_odeiv2_driver_apply per output time.
 *     evolve_apply does not write the controller's step size back on a final (clipped) step; the
 *     direction of integration is the sign of the initial step, so a monotone DEcreasing time grid
 *     integrates backwards and a grid that changes direction is HAMK_ERR_INVALID (GSL_EINVAL); a
 *     step that must shrink but cannot is GSL_FAILURE: the trajectory stops, HAMK_ST_UNDERFLOW.
 *   1 (gsl-ode.c built with -DGSLODE1): old gsl_odeiv -- `while (t < ti) gsl_odeiv_evolve_apply`.
 *     h is written back after every accepted step; repeated / decreasing times do no stepping.
 * One trajectory over a single interval (stepHam from h0 = dt/100) takes the same steps under
 * both; they differ from the second output time of evolveHam on, at truncation level (~eps).
 * The environment variable HAMK_GSL_API=1|2 sets the default of new handles.                     */
int     hamk_system_set_gsl_api(hamk_system* s, int32_t api);
int32_t hamk_system_get_gsl_api(const hamk_system* s);

/* Generated HIP source of the specialised module (for inspection/tests).    */
const char* hamk_system_source(const hamk_system* s);
/* Number of bytes of gfx950 code object produced by the specialisation.     */
int64_t hamk_system_code_size(const hamk_system* s);
/* The gfx950 code object(s) the kernels are loaded from (which = 0: default build, 1: the build
 * without MachineLICM, empty unless a kernel is taken from it).  Returns the size in bytes and, if
 * buf != NULL and cap is large enough, copies the ELF into buf -- so a host can disassemble what
 * actually runs (bench.py counts the fp64 instructions of the stepping loop from it).            */
int64_t hamk_system_code_object(const hamk_system* s, int32_t which, void* buf, int64_t cap);
/* One line per kernel: which of the two builds it is taken from (default options, or without
 * MachineLICM when that spills fewer SGPRs), its code bytes and its spilled SGPRs.             */
const char* hamk_system_build_info(const hamk_system* s);
/* Machine-code bytes of one kernel of the module ("hamk_rk4_steps_k", ...); 0 if unknown.
 * kernel_name == NULL: number of function symbols in the module (9 = every device function
 * was inlined into the 8 kernels of the path; the ninth is the self-check's scribble kernel).                                                          */
int64_t hamk_system_kernel_bytes(const hamk_system* s, const char* kernel_name);

/* ---- state functions ------------------------------------------------------ */

/* underlyingPos: x[m][B] = f(q).                          Hamilton.hs:174-178 */
int hamk_coords_batch(hamk_system* s, int64_t B, const double* q, double* x, int32_t mem);

/* toPhase / momenta: p = J^T (M (J qd)).                  Hamilton.hs:262-284 */
int hamk_to_phase_batch(hamk_system* s, int64_t B, const double* q, const double* qd,
                        double* p, int32_t mem);

/* fromPhase / velocities: qd = (J^T M J)^-1 p.            Hamilton.hs:316-337 */
int hamk_from_phase_batch(hamk_system* s, int64_t B, const double* q, const double* p,
                          double* qd, int32_t* status, int32_t mem);

/* Phase-space observables; any output pointer may be NULL.
 * ke = keP (:341-349), pe = pe (:182-186), h = hamiltonian (:353-361).      */
int hamk_observe_batch(hamk_system* s, int64_t B, const double* q, const double* p,
                       double* ke, double* pe, double* h, int32_t* status, int32_t mem);

/* Config-space observables; any output pointer may be NULL.
 * ke = keC (:288-296), lag = lagrangian (:301-309).                          */
int hamk_observe_config_batch(hamk_system* s, int64_t B, const double* q, const double* qd,
                              double* ke, double* lag, int32_t mem);

/* hamEqs: (dq, dp) = (dH/dp, -dH/dq).                     Hamilton.hs:370-387 */
int hamk_hameqs_batch(hamk_system* s, int64_t B, const double* q, const double* p,
                      double* dq, double* dp, int32_t* status, int32_t mem);

/* ---- time stepping ---------------------------------------------------------- */

/* Classic fixed-step RK4 over hamEqs, nsteps steps of dt, IN PLACE on q,p.
 * (BASELINE.json metric: "RK4 phase-space steps/sec".)  status may be NULL.  */
int hamk_rk4_steps(hamk_system* s, int64_t B, double* q, double* p,
                   double dt, int32_t nsteps, int32_t* status, int32_t mem);

/* The same, with the launch checking its own invariant: H = hamiltonian (Hamilton.hs:353-361) is
 * evaluated at entry and exit (two extra evaluations per launch, nothing per step) and
 * HAMK_ST_DRIFT is set where |H_exit - H_entry| > drift_tol * max(1, |H_entry|).  A fixed step
 * through a near-singularity (close encounter of the gravitational systems) is
 * otherwise silently wrong -- the reference's analogous failure raises out of `inv`
 * (Hamilton.hs:321,381).  drift_tol <= 0 disables the check (= hamk_rk4_steps).               */
int hamk_rk4_steps_checked(hamk_system* s, int64_t B, double* q, double* p,
                           double dt, int32_t nsteps, double drift_tol, int32_t* status, int32_t mem);

/* stepHam dt: adaptive RKF45 with GSL's standard controller from 0 to dt,
 * h0 = dt/100, eps_abs = eps_rel = 1.49012e-08, IN PLACE.  Hamilton.hs:390-402,
 * :445-448.  nsub (optional, [B]) receives accepted+rejected sub-step counts. */
int hamk_step_ham_batch(hamk_system* s, int64_t B, double* q, double* p, double dt,
                        int32_t* status, int32_t* nsub, int32_t mem);

/* `iterate (stepHam dt)` (README.md:150; the demo's frame loop, app/Examples.hs:429): ncalls consecutive
 * stepHam dt in ONE launch, IN PLACE.  Every call is what a separate hamk_step_ham_batch would do -- a fresh
 * evolveHam over (0, dt): t = 0, h0 = dt/100 (Hamilton.hs:400-402, :447), its own sub-step budget -- and the
 * result is bit-identical to ncalls separate calls BY CONSTRUCTION on every mapping (every call starts with the
 * instructions a separate launch starts with, its own evaluation of dydt_in included); what is saved is
 * ncalls - 1 launches and stream synchronisations (one trajectory is launch-latency bound: BASELINE config 1).
 * out_every > 0: the state after every out_every-th call goes to qout/pout, [ncalls / out_every][n][B]
 * (the frames an animation shows); out_every == 0: qout/pout may be NULL.  status: OR over the calls;
 * nsub: sub-steps summed over the calls.                                                                 */
int hamk_step_ham_iterate(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t ncalls,
                          int32_t out_every, double* qout, double* pout,
                          int32_t* status, int32_t* nsub, int32_t mem);

/* evolveHam: states at each of the nt >= 2 requested times ts[] (host array);
 * qout/pout are [nt][n][B]; row 0 is the initial state (Hamilton.hs:443-462).
 * h0 = (ts[1]-ts[0])/100 and the step size carries across output times, as in
 * hmatrix-gsl's `odeSolveV` (how exactly: hamk_system_set_gsl_api).  Pass h0 <= 0 /
 * eps <= 0 for the reference defaults.                                         */
int hamk_evolve_ham_batch(hamk_system* s, int64_t B, const double* q0, const double* p0,
                          int32_t nt, const double* ts, double* qout, double* pout,
                          double h0, double eps_abs, double eps_rel,
                          int32_t* status, int32_t* nsub, int32_t mem);

/* ---- initial conditions of an ensemble, generated on the device ------------------------------------
 * q[j][i], qd[j][i] of trajectories first_index .. first_index + B - 1: uniform in [q_lo[j], q_hi[j]] and
 * [qd_lo[j], qd_hi[j]] (host arrays of n entries), from a counter-based generator keyed by (seed, GLOBAL
 * trajectory index, field) -- splitmix64, hamk_sample.hpp -- so a rank of a sharded run fills its own shard in
 * HBM without a host array or a scatter, and every shard layout draws bit-identical inputs.
 * Same bits as the Python sampler the CPU tests use (hamilton_amd/examples.py sample_config).  The reference has no
 * counterpart: its demo starts one trajectory from a command-line Config (app/Examples.hs:230-359).                */
int hamk_sample_batch(hamk_system* s, int64_t B, int64_t first_index, uint64_t seed,
                      const double* q_lo, const double* q_hi, const double* qd_lo, const double* qd_hi,
                      double* q, double* qd, int32_t mem);

/* ---- devices and device memory ---------------------------------------------
 * For hosts that do not link HIP themselves (the Haskell shim, plain C): with these an ensemble
 * can live in HBM across calls (HAMK_MEM_DEVICE) and one process can drive every GPU of a node --
 * one handle per device, the calling thread's current device selects where a handle runs.  The
 * reference has no counterpart (it is a single-trajectory CPU library); they exist because the
 * north_star's "state resident in HBM" and "shard over 8 GPUs, gather at the end" must be reachable
 * through the C ABI alone.                                                                      */
int hamk_set_device(int32_t device);                 /* current device of the calling thread      */
int hamk_get_device(int32_t* device);
int hamk_device_malloc(void** ptr, int64_t bytes);   /* on the current device                     */
int hamk_device_free(void* ptr);
#define HAMK_COPY_H2D 0
#define HAMK_COPY_D2H 1
#define HAMK_COPY_D2D 2   /* same or another device (peer copy over xGMI)                        */
int hamk_memcpy(void* dst, const void* src, int64_t bytes, int32_t kind);   /* synchronous        */
/* Final gather of a sharded ensemble: part g is a structure-of-arrays block [n][B_parts[g]] in
 * device memory (any device); out is [n][sum_g B_parts[g]], trajectories in part order -- in host
 * memory (out_mem = HAMK_MEM_HOST) or on the current device (HAMK_MEM_DEVICE; peer copies over
 * xGMI).  One array per call (q, then p).  Synchronous; it does not wait for launches still
 * running on other streams -- hamk_synchronize the handles that produce the parts first.        */
int hamk_gather_batch(int32_t nparts, int32_t n, const int64_t* B_parts, const double* const* parts,
                      double* out, int32_t out_mem);

/* The same final gather for hosts that run ONE PROCESS PER GPU ("RCCL over xGMI only for
 * the final gather"; the reference has no counterpart): a process has no peer pointers to hand to
 * hamk_gather_batch, it needs a communicator.  Rank 0 draws an id (HAMK_COMM_ID_BYTES opaque bytes) and ships
 * it to the other processes by whatever host channel the launcher offers (a file, a socket, MPI_Bcast); every
 * process selects its device (hamk_set_device) and calls hamk_comm_create with the same id and world and its
 * own rank -- collectively: the call returns when all ranks have arrived.  hamk_comm_allgather_batch is
 * collective too: part is this rank's structure-of-arrays block [n][B_parts[rank]] on the communicator's
 * device, out [n][sum_g B_parts[g]] on the same device, trajectories in rank order on EVERY rank; B_parts
 * (host array, world entries) must be the same on all ranks.  Equal shards take one ncclAllGather per row,
 * ragged ones one ncclBroadcast per (rank, row), fused in one RCCL group either way.  One array per call (q,
 * then p).  Runs on the NULL stream and returns when the data is there; hamk_synchronize the handle that
 * produced part first.  RCCL is loaded on first use: a box without librccl gets HAMK_ERR_UNSUPPORTED from
 * these four calls and loses nothing else.                                                                     */
#define HAMK_COMM_ID_BYTES 128
typedef struct hamk_comm hamk_comm;
int hamk_comm_unique_id(void* id);                                        /* HAMK_COMM_ID_BYTES bytes out   */
int hamk_comm_create(const void* id, int32_t world, int32_t rank, hamk_comm** out);
int hamk_comm_allgather_batch(hamk_comm* comm, int32_t n, const int64_t* B_parts, const double* part,
                              double* out);
int hamk_comm_destroy(hamk_comm* comm);                                   /* NULL is fine                   */

/* ---- ensemble checkpoint (no reference counterpart) -----------------------
 * One file = 64-byte header (magic "HAMKCKP1", n, B, steps_done, seed, t), q[n][B], p[n][B] as raw
 * little-endian fp64, SHA-256 of all of it.  q, p may be host or device arrays (mem); device state
 * is staged in 8 MiB pieces.  Written aside and renamed: a crash leaves the previous file.
 * steps_done / seed / t are the caller's bookkeeping (per-index splitmix64 seed of the initial
 * conditions, steps taken, model time) and come back from hamk_checkpoint_info.  Every kernel is a
 * pure function of the state, so a resumed run continues bit-identically -- on the same mapping: a run
 * resumed with a different shard layout states the whole ensemble's size (hamk_system_set_ensemble_size)
 * or pins hamk_options::mapping, as the original run must have.
 * Device state is copied on the NULL stream: hamk_synchronize the handle (or synchronise the stream that
 * produced / will consume q, p) before a write and before using the arrays after a read.  A header that does
 * not match the file's size is rejected before anything is allocated from it.                      */
int hamk_checkpoint_write(const char* path, int32_t n, int64_t B, const double* q, const double* p,
                          int32_t mem, int64_t steps_done, uint64_t seed, double t);
int hamk_checkpoint_info(const char* path, int32_t* n, int64_t* B, int64_t* steps_done,
                         uint64_t* seed, double* t);                    /* outputs may be NULL */
int hamk_checkpoint_read(const char* path, int32_t n, int64_t B, double* q, double* p, int32_t mem);

/* ---- diagnostics ---------------------------------------------------------- */
const char* hamk_last_error(void);
const char* hamk_version(void);
/* Number of visible HIP devices (0 if none / runtime missing).               */
int hamk_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* HAMK_H */
