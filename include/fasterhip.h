/* fasterhip.h — C ABI of the MI355X-native batched trajectory-optimisation core.
 *
 * This is the drop-in boundary for ONE hot path of mit-acl/faster: the Gurobi-backed
 * SolverGurobi::genNewTraj() / callOptimizer() / fillX()
 * (reference: faster/src/solverGurobi.cpp:426-477, :549-657, :122-168).
 *
 * The reference has no FFI for this path: the boundary there is the C++ class SolverGurobi
 * (faster/include/solverGurobi.hpp:61-186) held by value by Faster (faster/include/faster.hpp:74-75).
 * The replacement class `SolverHip` (faster_amd/host/solver_hip.hpp) keeps that class surface
 * and forwards to the entry points declared here; INTEGRATION.md shows the binding.
 *
 * Everything crossing this boundary is plain C: pointers, sizes, PODs. No torch, no HIP types
 * (a stream is passed as void*).  All arithmetic of the path is FP64.
 *
 * One `fh_problem` == one genNewTraj() call (the complete factor line search, every trial a
 * mixed-integer QP over the segment->polytope assignment).  Problems are independent.
 */
#ifndef FASTERHIP_H
#define FASTERHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_MAX_SEG 16         /* max N (segments); reference default N_=10 (solverGurobi.hpp:136), yaml 6 */
#define FH_MAX_POLY 8         /* max polytopes per problem (BASELINE config 5: <=8) */
#define FH_MAX_FACES 256      /* max faces per problem, summed over its polytopes */
#define FH_MAX_FACES_POLY 64  /* max faces of one polytope */

/* status codes in fh_result.status */
enum {
  FH_ST_OPTIMAL = 0,     /* some factor gave an optimal MIQP solution  (reference: GRB_OPTIMAL)      */
  FH_ST_INFEASIBLE = 1,  /* every factor in the window was infeasible   (reference: genNewTraj()==false) */
  FH_ST_NODE_LIMIT = 2,  /* branch-and-bound node cap hit (treated as not solved)                    */
  FH_ST_ITER_LIMIT = 3,  /* active-set iteration cap hit  (treated as not solved, cf. GRB_NUMERIC)   */
  FH_ST_BAD_INPUT = 4,   /* sizes out of range / non-finite input                                    */
  FH_ST_INTERRUPTED = 5  /* fh_request_stop() or the deadline ended the search (reference: GRB_INTERRUPTED after
                            mycallback::abort(), solverGurobi.cpp:15-28, :643-646); treated as not solved */
};

/* return codes of the entry points */
enum {
  FH_OK = 0,
  FH_ERR_ARG = -1,     /* null pointer / bad size */
  FH_ERR_DEVICE = -2,  /* HIP runtime error (message via fh_last_error) */
  FH_ERR_NOMEM = -3
};

/* One genNewTraj() call.  Mirrors the inputs stored by the SolverGurobi setters:
 *   setN (solverGurobi.cpp:60-63), setDC (:293-296), setBounds (:409-416),
 *   setFactorInitialAndFinalAndIncrement (:418-424), setForceFinalConstraint (:170-173),
 *   setX0/setXf (:298-330; order pos xyz, vel xyz, accel xyz), setPolytopes (:175-178). */
typedef struct fh_problem {
  int32_t n_seg;           /* N_  (1..FH_MAX_SEG)                                              */
  int32_t n_poly;          /* polytopes_.size() (0..FH_MAX_POLY); 0 => no corridor constraints */
  int32_t force_final_pos; /* forceFinalConstraint_: 1 whole trajectory, 0 safe trajectory      */
  int32_t face_begin;      /* index of this problem's first face in the batch face array        */
  int32_t face_off[FH_MAX_POLY + 1]; /* polytope p owns faces [face_begin+face_off[p], face_begin+face_off[p+1]) */
  uint32_t pin[2];         /* optional fixed binaries (BASELINE config 1 "fixed binaries (pure QP)"; the reference has no such
                              API — its b[t][p] are private): nibble t of the 64-bit word pin[0] | pin[1]<<32 is 0 to leave
                              segment t free, or p+1 to force b[t][p] = 1.  All zero = the reference's MIQP.            */
  int32_t reserved;
  double dc;               /* DC                                                               */
  double v_max, a_max, j_max;
  double f_init, f_final, f_inc; /* factor window; the loop accumulates `f += f_inc` in double   */
  double x0[9];            /* x0_[9]                                                           */
  double xf[9];            /* xf_[9] (pos used only for dt when force_final_pos==0)            */
} fh_problem;

/* A face is the row (a_x, a_y, a_z, b) of A x <= b  (LinearConstraint3D,
 * thirdparty/DecompROS/DecompUtil/include/decomp_geometry/polyhedron.h:115-185). 32 B, AoS. */
typedef struct fh_face {
  double a[3];
  double b;
} fh_face;

/* What genNewTraj() leaves behind: return value, trials_, factor_that_worked_, dt_, ObjVal,
 * the 12*N polynomial coefficients in the reference variable order
 * [ax ay az bx by bz cx cy cz dx dy dz] per segment (solverGurobi.cpp:70-84), and the binary matrix
 * collapsed to one polytope index per segment. */
typedef struct fh_result {
  int32_t solved;    /* 1 <=> genNewTraj() returned true                               */
  int32_t trials;    /* trials_                                                         */
  int32_t status;    /* FH_ST_*                                                         */
  int32_t nodes;     /* branch-and-bound nodes evaluated over all trials (diagnostic)   */
  int32_t qp_iters;  /* active-set iterations over all trials (diagnostic)              */
  int32_t kflops;    /* FP64 flop estimate of the solve in thousands, saturating (diagnostic; DESIGN.md roofline) */
  double factor;     /* factor_that_worked_ (valid iff solved)                          */
  double dt;         /* dt_ of the last trial                                           */
  double cost;       /* objective: sum_t sum_axis (6 a)^2  (solverGurobi.cpp:113-119)   */
  double coeff[FH_MAX_SEG][12];
  int8_t assign[FH_MAX_SEG]; /* polytope index per segment, -1 if n_poly==0          */
} fh_result;

/* Solver tolerances / limits (the counterpart of Gurobi parameters left at default by the
 * reference).  Obtain defaults with fh_default_params(). */
typedef struct fh_params {
  double feas_tol;    /* a row is violated iff  a.x - b > feas_tol   (default 1e-9)          */
  double dep_tol;     /* |z|/|g| below this => candidate row linearly dependent (1e-10)      */
  int32_t max_nodes;  /* per trial branch-and-bound node cap          (default 100000)       */
  int32_t max_iters;  /* per QP active-set iteration cap              (default 2000)         */
  int32_t max_work;   /* per PROBLEM cap on active-set iterations over all trials and nodes; 0 = unlimited (default).
                         The reference sets no Gurobi TimeLimit; a real-time caller can bound the worst case here:
                         a problem that exceeds it ends with FH_ST_ITER_LIMIT, solved = 0.  A work cap couples the
                         nodes of a problem, so problems are then solved by one wavefront each (no work sharing).   */
  int32_t share;      /* 1 (default): wavefronts that run out of problems take over untried subtrees of the branch-and-bound
                         trees still being explored (Gurobi explores one tree with all its threads, Threads = 0,
                         faster/param/faster.yaml:41); results are identical to share = 0 (one wavefront per problem);
                         nodes / qp_iters / kflops then count the work actually done by all wavefronts.  max_nodes and
                         max_iters apply per wavefront in a shared tree.  ONE documented difference: when a wavefront runs
                         into max_nodes / max_iters in the trial that holds the best leaf of a SHARED problem, the problem
                         is reported unsolved with that limit's status, where share = 0 would go on to the next factor
                         (the trials after it may already have been given away; the sequential continuation cannot be
                         reconstructed).  With the default limits (1e5 nodes, 2000 iterations per QP) no problem of the
                         BASELINE configurations comes near them.                                                     */
  double mip_gap;     /* 0 (default): the exact optimum over all assignments.  > 0: a node is pruned when its lower bound is
                         within this relative gap of the incumbent (Gurobi's MIPGap, default 1e-4, which the reference
                         leaves untouched); the result may then depend on exploration order, so work sharing is off.   */
  double deadline_ms; /* 0 (default): none.  > 0: wall-clock budget of a launch, measured on the device from the start of
                         each workgroup; problems not finished by then end with FH_ST_INTERRUPTED (the replan period of
                         the reference is 10 ms, faster/param/faster.yaml:5; Gurobi TimeLimit is not set there).         */
} fh_params;

/* One sample of fillX(): pos, vel, accel, jerk (faster_types.hpp:79-165 `state`, yaw/dyaw unused
 * by the solver).  12 doubles = 96 B. */
typedef struct fh_state {
  double pos[3], vel[3], accel[3], jerk[3];
} fh_state;

typedef struct fh_ctx fh_ctx;

/* ---- lifetime ------------------------------------------------------------------------- */
/* device < 0: use the current HIP device. Replaces the GRBEnv/GRBModel member initialisers
 * (solverGurobi.hpp:154-155). */
int fh_create(fh_ctx** out, int device);
void fh_destroy(fh_ctx* ctx);
const char* fh_last_error(const fh_ctx* ctx);
void fh_default_params(fh_params* p);
int fh_set_params(fh_ctx* ctx, const fh_params* p);
/* Kernels run on this stream (hipStream_t as void*); NULL => the context's own stream.
 * A context owns ONE work queue, ticket counter and branch-and-bound workspace: at most one of its launches may be in
 * flight at a time.  Launches issued through one context are ordered on its stream; switching streams synchronises with
 * the previous one.  Use one context per concurrent pipeline (as bench.py does). */
int fh_set_stream(fh_ctx* ctx, void* hip_stream);

/* Packed results: what crosses PCIe / xGMI.  An fh_result holds FH_MAX_SEG = 16 coefficient rows whatever n_seg is (1600 B; at
 * N = 10, 576 B of it are dead rows).  A packed record is the same words without the rows beyond n_seg:
 *   [ the 48-byte head: solved .. cost | coeff[0 .. n_seg-1][12] | assign[16] ]  =  64 + 96 n_seg bytes  (N = 10: 1024 B).
 * Lossless: fh_unpack_results() restores the fh_result records exactly (dead rows zero, as the kernels write them). */
size_t fh_packed_result_size(int n_seg);
/* d_results [n] fh_result -> d_packed [n] packed records; device pointers, asynchronous on the context stream */
int fh_pack_results_device(fh_ctx* ctx, const fh_result* d_results, int n, int n_seg, void* d_packed);
/* host side (no device needed): full records -> packed records, and back */
int fh_pack_results(const fh_result* results, int n, int n_seg, void* packed);
int fh_unpack_results(const void* packed, int n, int n_seg, fh_result* results);

/* The Bezier control points of every segment of every result: SolverGurobi::getCP0..getCP3 (solverGurobi.cpp:833-862) evaluated on
 * the result's coefficients with the reference's own expressions and operation order — cp0 = getPos(t, 0), cp1 = (Cn + 3 Dn) / 3,
 * cp2 = (Bn + 2 Cn + 3 Dn) / 3, cp3 = getPos(t, dt), with the normalised coefficients An = a dt^3, Bn = b dt^2, Cn = c dt, Dn = d
 * (:810-830) and getPos = a tau tau tau + b tau tau + c tau + d (:761-767).  These are the points the indicator rows of
 * setPolytopesConstraints constrain (:254-288) — with cost and the feasibility flag, the outputs BASELINE names.
 * cp: [n][n_seg][4][3] doubles (control point, axis); rows of unsolved results are zero.  Host side, no device needed (a format
 * conversion of fh_result like fh_unpack_results, not a solver). */
int fh_control_points(const fh_result* results, int n, int n_seg, double* cp);


/* Scheduling of a solve launch: how the persistent workgroups order and share the work of a batch.  NO RESULT FIELD DEPENDS ON
 * ANY OF THESE (tests/test_gpu_round2.py solves 8192 pairs with each of them switched off and compares bit for bit); only
 * nodes / qp_iters / kflops, which count the work actually done, and the time a launch takes.  Defaults: fh_default_sched(). */
typedef struct fh_sched {
  int32_t launch_order;       /* 1 (default): batches of >= 2048 units start with the corridors that have most polytopes — unless another context
                                 of this process has a solve launch in flight on the device (what the order is for, the END of a launch, is
                                 hidden then, and the two small sorting launches queue behind the resident grids); 2: always; 0: never   */
  int32_t publish_factor;     /* a problem that has used this many times the running mean of active-set iterations may publish
                                 frames ahead of the idle workgroups (default 4; 0: never)                                      */
  int32_t backlog;            /* frames that may be published ahead of the takers (default 32; 0: none)                        */
  int32_t waiting_workgroups; /* workgroups that keep waiting for frames when the fresh problems run out (0 = default: CUs / 64; CUs / 16 for batches of up to 8 problems per CU) */
  int32_t min_nodes;          /* a problem gives work to an IDLE workgroup only after this many nodes of its trees (default 2)  */
  int32_t cloud_blocks;       /* 1 (default): the decomposition skips blocks of 64 cloud points whose bounding box misses the
                                 local box of a segment                                                                         */
  int32_t workgroups_per_cu;  /* resident solves per CU (0 = default: as many as LDS and registers admit, 11 for the C4 kernel).
                               * 1..8 also selects the kernel build for two wavefronts per SIMD (all registers, no scratch): a
                               * batch that is alone on the device is done 13 % sooner (C4: 3.0 instead of 3.5 ms), batches
                               * streamed back to back 14 % later.  Same results bit for bit.  With 0, launches that cannot have
                               * more than 8 solves per CU anyway (N >= 15: LDS; batches of up to 8 problems per CU) run that build
                               * too; a value above 8 asks for the three-wavefront build whatever the batch.                        */
  int32_t no_child_bound;     /* 0 (default — also what a zero-initialised fh_sched asks for): a child of a branch-and-bound node is not
                                 visited when a lower bound of its QP that is known at the parent — the parent's multipliers plus one
                                 multiplier on the child's most violated row: cost* + v^2 / |n|^2 — already loses against the incumbent
                                 (it holds no better leaf: the result is unchanged up to ties of two leaves' costs below 1e-12 relative,
                                 the trees are half as large).  1: every child is visited, the tree of the CPU oracle.
                                 (Round 4 called this field child_bound with 1 = default; inverted so that a caller who fills a
                                 zero-initialised struct by hand keeps the default behaviour.)                                        */
  int32_t compact_results;    /* 1: a solve writes only the coefficient rows its kernel is built for (the smallest of 6, 10, 15, 16 segments that
                                 holds max_seg): rows fh_result.coeff[t] with t >= that count — they carry no information, a problem has
                                 n_seg <= max_seg segments — are left as the caller's buffer had them.  0 (default): every word of every
                                 fh_result is written (no memset needed).  At N = 10 a record is 1600 bytes of which 1024 are written
                                 with 1: what a caller that streams batches wants (bench.py sets it).                                     */
  int32_t pair_outputs;       /* fh_solve_pairs_device / fh_pool_solve_pairs: 1: d_safe and d_safe_faces are OUTPUTS, complete for every
                                 pair, exactly as fh_pair_glue_device leaves them (x0 = R, n_poly, face_off, face_begin, the rows; n_seg = 0
                                 marks a pair without a safe problem).  0 (default): they are scratch — the safe problem of a pair is built
                                 in the LDS of the wavefront that solves it and is written to these buffers only when it is handed to
                                 another workgroup (a tree that several wavefronts explore; a few dozen of 32768 pairs): what is in them
                                 after the launch is unspecified, the TEMPLATE fields of d_safe (n_seg, bounds, dc, factor window,
                                 force_final_pos, xf) are never written, and the results are the same bit for bit.  The safe problem's x0
                                 is the whole trajectory at sample k_safe (fh_sample_batch / fh_append_plans_device give it).              */
  int32_t look_every;         /* a branch-and-bound tree reads the launch's control words — stop request, deadline, is somebody out of work, do I hold
                                 tickets somebody else could start — every look_every-th node (a power of two, 2..1024; a shared problem twice as
                                 often).  0 (default): the library decides per launch — 8 for a launch that has the device to itself (what ends
                                 such a launch is its tail: its long trees must find help early), 16 when another context of this process has a
                                 solve launch in flight on the same device when this one is issued (the tail of a launch is then hidden behind
                                 the others, and every frame that changes hands is overhead: C4, twelve launches in flight, +2.5 %; one launch
                                 alone with 16: 2-5 % later).  fh_last_launch reports the value a launch ran with.                          */
  int32_t struct_size;        /* sizeof(fh_sched) as the CALLER was compiled (fh_default_sched sets it), or 0 = not stated.  fh_set_sched
                                 refuses any other value: the struct has changed between rounds (round 4's child_bound became
                                 no_child_bound with the opposite meaning at the same offset), and a binary built against an older
                                 header must fail loudly instead of silently switching a bound off.  See also fh_abi_version().         */
} fh_sched;
/* The layout generation of the structs in this header: bumped whenever a field changes its meaning, offset or size.  A caller compares
 * FH_ABI_VERSION (its compile time) with fh_abi_version() (the loaded library) once; SolverHip does. */
#define FH_ABI_VERSION 8
int fh_abi_version(void);
void fh_default_sched(fh_sched* s);
int fh_set_sched(fh_ctx* ctx, const fh_sched* s);

/* Cooperative cancellation (SolverGurobi::StopExecution / ResetToNormalState, solverGurobi.cpp:30-39; the reference polls
 * its flag inside Gurobi callbacks, :15-28).  fh_request_stop() may be called from ANY thread while a launch of the context
 * is running: it raises a word in mapped host memory that the workgroups poll between branch-and-bound nodes and when they
 * draw a problem; problems not finished by then report FH_ST_INTERRUPTED, solved = 0.  The request stays raised (later
 * launches return immediately with FH_ST_INTERRUPTED results) until fh_clear_stop(). */
int fh_request_stop(fh_ctx* ctx);
int fh_clear_stop(fh_ctx* ctx);

/* Work-sharing statistics of the most recent solve launch of the context (synchronises with it): frames given to the queue,
 * frames taken from it, donations refused because the queue / the record pool was full, workgroups launched.  Also the place
 * where a failed launch (watchdog, protocol error) is reported: returns FH_ERR_DEVICE and poisons nothing — the next launch
 * re-initialises the queue. */
typedef struct fh_share_stats {
  uint32_t donated, stolen, queue_full, records_full, records_used, error, interrupted, workgroups;
} fh_share_stats;
int fh_share_stats_read(fh_ctx* ctx, fh_share_stats* out);
/* Diagnostic builds only (-DFH_SHARE_PROFILE, scripts/share_diag.py): 16 words of in-kernel timers of the last launch (100 MHz
 * ticks / counts: look-around, donate, wait, frame copy, frame set-up, frame search, finish_part, idle tail).  All zero otherwise. */
int fh_share_profile_read(fh_ctx* ctx, unsigned long long* out16);

/* Measured FP64 vector-FMA peak of the device (independent v_fma_f64 chains, no memory traffic), in TFLOP/s — the
 * denominator of the compute roofline.  Runs a ~10 ms kernel on the context stream and synchronises. */
int fh_fp64_peak(fh_ctx* ctx, double* tflops);

/* ---- the hot path ---------------------------------------------------------------------- */
/* SolverGurobi::getDTInitial (solverGurobi.cpp:659-759) for a batch: the lower bound of the segment time from which genNewTraj's
 * factor loop starts (dt = factor * max(dt_initial, 2 DC), findDT :494-497) — per axis the time to cover |xf - x0| at v_max and the
 * smallest positive real roots of the constant-jerk cubic and the constant-acceleration quadratic, with the reference's float casts
 * and float / int division; > 10000 -> 0.  Reads x0, xf (positions), v_max, a_max, j_max and n_seg of each problem record; the same
 * device function the solve kernels call (their fh_result.dt = factor * max(this, 2 dc)).  dt [n] doubles.
 * Two things a caller should know (Eigen's PolynomialSolver, which the reference calls at :700-746, is not part of this library):
 *  - the roots come from single-precision starting points polished in double precision (closed forms in double where two roots are
 *    close); the value is the FLOAT the reference stores, and against the closed forms it can differ by one float ulp about once in
 *    1e8 roots (none in the 2.6 M problems of the test suite);
 *  - CONVENTION for xf == x0 exactly on an axis: the constant term of that axis' cubic is zero, so one root is exactly 0 and is
 *    NOT "the smallest positive root" (MinPositiveElement, solverGurobi_utils.hpp:19-32, tests v > 0); the candidates are the roots of
 *    (j/6) t^2 + (a0/2) t + v0.  A floating-point root finder returns that zero as +-1e-17 depending on its last bit — the
 *    reference's answer there is a property of the Eigen build it is linked against. */
int fh_dt_initial_batch(fh_ctx* ctx, const fh_problem* problems, int n, double* dt);
int fh_dt_initial_batch_device(fh_ctx* ctx, const fh_problem* d_problems, int n, double* d_dt);

/* Batch of genNewTraj() calls, inputs/outputs in HOST memory (copies in/out, synchronous).
 * Replaces SolverGurobi::genNewTraj() (solverGurobi.cpp:426-477) incl. every callOptimizer()
 * (:549-657) it issues.  `faces` holds n_faces rows addressed through fh_problem.face_begin. */
int fh_solve_batch(fh_ctx* ctx, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n,
                   fh_result* results);

/* Concurrent time-allocation search: same inputs, outputs and RESULTS as fh_solve_batch, but the factor
 * line search of genNewTraj (solverGurobi.cpp:445-446, `for (i = f_init; i <= f_final && !solved; i += f_inc)`) is run
 * `width` factors at a time, every factor as its own single-trial problem on its own wavefront, and the first (smallest)
 * feasible factor wins.  Trials of the reference are independent of one another (the model is rebuilt from scratch per
 * trial, :447-466), so this is the sequential first-feasible rule evaluated concurrently: factor, dt, coefficients, cost,
 * assignment and `trials` (= index of the winning factor + 1) are bit-identical to fh_solve_batch; nodes / qp_iters are
 * summed over the trials the sequential search would have run.  Meant for single problems and small batches (a replan is
 * ONE whole + ONE safe solve): it trades up to `width` times the work for the latency of one trial.  width <= 1, or a
 * work cap (fh_params.max_work > 0, which couples the trials), falls back to fh_solve_batch. */
int fh_solve_batch_speculative(fh_ctx* ctx, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n,
                               int width, fh_result* results);

/* Same, with every pointer already resident in device memory (HBM). Asynchronous on the
 * context's stream; call fh_sync() before reading results.  max_seg / max_faces are upper bounds on
 * n_seg and on the per-problem face count of the batch (they select the kernel instantiation and its
 * LDS carve; pass 0 for the library maxima FH_MAX_SEG / FH_MAX_FACES).  A problem exceeding them is
 * reported with FH_ST_BAD_INPUT. */
int fh_solve_batch_device(fh_ctx* ctx, const fh_problem* d_problems, const fh_face* d_faces, int n, int max_seg,
                          int max_faces, fh_result* d_results);

/* resetX()+fillX() (solverGurobi.cpp:382-388, :122-168) for a batch: problem i writes
 * counts[i] = max(2,(int)(N*dt/DC)) states to states[i*max_samples ...]; if counts[i] > max_samples
 * only max_samples are written (counts still reports the full size). Unsolved problems give 0. */
int fh_sample_batch(fh_ctx* ctx, const fh_problem* problems, const fh_result* results, int n, int max_samples,
                    fh_state* states, int32_t* counts);
int fh_sample_batch_device(fh_ctx* ctx, const fh_problem* d_problems, const fh_result* d_results, int n,
                           int max_samples, fh_state* d_states, int32_t* d_counts);

/* Whole -> safe hand-off of Faster::replan (faster/src/faster.cpp:456-475, :506-524) for synthetic
 * pairs, entirely on the device: for pair i take R = sample number
 * (int)(r_frac*count_i) of the whole trajectory (fillX semantics) as x0 of safe problem i (pos, vel,
 * accel).  The safe corridor is the run of up to max_safe_poly consecutive polytopes of the whole
 * corridor starting at the first one that (shrunk by `shrink` metres) contains R; its faces are written
 * to d_safe_faces at the SAME face_begin as the whole problem (d_safe_faces must be as large as
 * d_faces).  Everything else in d_safe[i] (n_seg, bounds, dc, factor window, force_final_pos, xf) is left
 * as the caller prepared it.  Unsolved whole problems mark the safe problem with n_seg = 0 (skipped:
 * its result reports FH_ST_BAD_INPUT).  Device pointers, asynchronous on the context stream. */
/* How the synthetic hand-off treats R (applies to fh_pair_glue_device and fh_solve_pairs_device of this context).
 * r_margin < 0 (default): SURVEY.md 8(d) to the letter, as described above — R may end up outside its shrunk corridor (22 % of the
 * C4 pairs: those safe problems are infeasible for every factor, which FASTER never poses).  r_margin >= 0: as in FASTER, where the
 * safe corridor is decomposed around R (faster.cpp:475-499), the corridor starts at the first polytope of the whole corridor that
 * contains R, and no face of that polytope is pulled closer to R than r_margin metres; the following polytopes are shrunk as before. */
int fh_set_pair_margin(fh_ctx* ctx, double r_margin);
/* WHICH sample of the whole trajectory becomes R (applies to fh_pair_glue_device, fh_solve_pairs_device and fh_pool_solve_pairs).
 * mode 0 (default): sample (int)(r_frac * count) — the synthetic pairing of SURVEY.md 8(d).
 * mode 1: FASTER's own rule, on the device, per pair:
 *   findIndexH (faster/src/faster.cpp:218-251): every 10th sample of the whole trajectory is tested against unknown space; the first
 *     one closer than drone_radius to it gives indexH = (int)(delta_h * i).  The reference asks a kd-tree of the mapper's unknown
 *     voxels; a batch of independent problems has no mapper, so unknown space is MODELLED here: everything farther than r_known from
 *     the start x0 of the whole problem is unknown (a vehicle that has seen what its sensor reaches and nothing else), i.e. the
 *     distance to unknown space is r_known - |pos - x0|.  No sample near unknown space: no safe trajectory is needed
 *     (needToComputeSafePath == false, faster.cpp:462-466) and the pair ends with its whole trajectory (safe n_seg = 0).
 *   findIndexR (faster.cpp:173-216): the first sample i <= indexH from which the vehicle can no longer brake before H — per axis
 *     x, y: sign(v (pH - p)) v^2 / (2 delta_a a_max) > |pH - p| — is R; if there is none, R = H.
 * mode 2: the same two functions against unknown space AS AN INPUT — the mapper's unknown voxels, given once per context with
 *   fh_set_unknown_grid_device: a sample is near unknown space iff an unknown voxel centre is closer than drone_radius to it, which is
 *   what the reference's `kdtree_unk_.nearestKSearch(p, 1, ...)` + `sqrt(d2) < drone_radius` decides (faster.cpp:236-240).  r_known
 *   is not used.  Every entry point that chooses R takes it: the staged ones (fh_pair_glue_device, fh_safe_corridor_batch_device,
 *   fh_append_plans_device) and, since round 5, the fused pair kernel (fh_solve_pairs_device, fh_pool_solve_pairs with
 *   fh_pool_set_unknown_grid): H, R and "is a safe trajectory needed" are then the reference's decisions inside the one launch; the
 *   safe corridor of the fused kernel stays the run of polytopes of the WHOLE corridor from the one that holds R — the corridor
 *   decomposed around R against unknown + occupied space, as FASTER builds it, is fh_safe_corridor_batch_device (the staged path).
 *   NOTE: the fused kernel's safe corridor is therefore an OCCUPIED-SPACE corridor, NOT FASTER's safe corridor (faster.cpp:475-499:
 *   cvxEllipsoidDecomp(JPS_safe, UNKOWN_AND_OCCUPIED_SPACE, ...)): its polytopes were decomposed against occupied points only, so its safe
 *   trajectories are NOT confined to known space — the property the safe trajectory exists for.  It is a different, easier problem
 *   (bench.py: 0.995 of its safe problems are solved, 0.387 of the faithful chain's); a caller that needs FASTER's guarantee runs the
 *   staged chain: fh_solve_batch_device -> fh_safe_corridor_batch_device -> fh_solve_batch_device.
 *   The fused mode-2 launch runs kernel instantiations of its own, built for two wavefronts per SIMD (<= 8 solves per CU).
 * r_frac of the calls is ignored in modes 1 and 2.  The safe corridor is built from R as described above in every mode. */
typedef struct fh_pair_rule {
  int32_t mode, reserved;
  double r_known;       /* [m] radius of known space around x0 (mode 1; unused in mode 2)                  */
  double drone_radius;  /* [m] par_.drone_radius (faster.yaml: 0.42)                                       */
  double delta_h;       /* par_.delta_H (faster.yaml: 1.0)                                                 */
  double delta_a;       /* par_.delta_a (faster.yaml: 0.5)                                                 */
} fh_pair_rule;
int fh_set_pair_rule(fh_ctx* ctx, const fh_pair_rule* rule);
/* Unknown space as an input (rule mode 2): d_flags[(iz ny + iy) nx + ix] != 0 marks cell (ix, iy, iz) of `grid` as unknown; the voxel
 * stands for its centre ((i + 0.5) res + origin) — the point the mapper's unknown cloud holds for it (FASTER: pclptr_unk_ feeds
 * kdtree_unk_, faster/src/faster.cpp:99-137, and, together with the occupied points, the decomposition: vec_uo_, jps_manager.cpp:91-98).
 * Device pointer, owned by the caller, read by every later launch of the context that uses rule mode 2 (it may be updated between
 * launches like any other input; it need not be the lattice of an fh_map).  d_flags = NULL: no unknown grid (mode 2 is refused). */
struct fh_voxel_grid;
int fh_set_unknown_grid_device(fh_ctx* ctx, const struct fh_voxel_grid* grid, const unsigned char* d_flags);
int fh_pair_glue_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results,
                        const fh_face* d_faces, int n, double r_frac, double shrink, int max_safe_poly,
                        fh_problem* d_safe, fh_face* d_safe_faces);
/* Faster::appendToPlan (faster/src/faster.cpp:606-648) for a batch of independent pairs whose plan holds only the start A
 * (k_end_whole = 0): plan i = the samples 0 .. k_safe of the whole trajectory (fillX semantics), then every sample of the safe
 * trajectory (:627-640).  k_safe is recomputed by the rule of the hand-off (r_frac, fh_set_pair_rule): the same sample that became x0
 * of the safe problem.  counts[i] = k_safe + 1 + samples of the safe trajectory; 0 when the pair commits nothing — the whole solve
 * failed (:427-431), or a safe trajectory was needed and not found (:529-533).  A pair that needs no safe trajectory (rule mode 1,
 * :462-466) commits its whole trajectory: k_safe = its last sample.  At most max_states states are written per pair (counts still
 * reports the full length); d_k_safe may be NULL.  Device pointers, asynchronous on the context stream. */
int fh_append_plans_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const fh_problem* d_safe,
                           const fh_result* d_safe_results, int n, double r_frac, int max_states, fh_state* d_plans, int32_t* d_counts,
                           int32_t* d_k_safe);

/* Faster::getNextGoal (faster/src/faster.cpp:699-723) for a batch of committed plans — the consumer of what fh_append_plans_device
 * wrote: plan i is d_plans[i][d_cursor[i] .. d_counts[i]); a call returns its front state and pops it unless it is the last one
 * (`next_goal = plan_.front(); if (plan_.size() > 1) plan_.pop_front();`), `ticks` calls in a row return the state of the last of them
 * and advance the cursor by up to `ticks`.  Yaw (getDesiredYaw) is not computed: yaw planning is out of scope.  A pair that committed
 * nothing (d_counts[i] == 0) gets a zero state and d_ok[i] = 0 (may be NULL).  d_cursor [n] starts at zero.  Asynchronous on the
 * context stream. */
int fh_next_goals_device(fh_ctx* ctx, const fh_state* d_plans, const int32_t* d_counts, int32_t* d_cursor, int n, int max_states, int ticks,
                         fh_state* d_goals, int32_t* d_ok);

/* ---- next row N1 on the device: convex decomposition around path segments ------------------------------ */
/* JPS_Manager::cvxEllipsoidDecomp (faster/src/jps_manager.cpp:80-127) for a batch of path segments that share one obstacle
 * cloud: per segment the rows [a | b] of its polytope (separating planes of the inflated obstacle points found by DecompUtil's
 * LineSegment3D::dilate, the 6 faces of the local bounding box, the ground plane -z <= -z_ground), oriented around the segment
 * midpoint.  cloud_xyz: [n_cloud][3]; segments: [n_segments][6] = p1, p2; faces: [n_segments][max_faces]; counts[i] = rows of
 * segment i, or -1 if it needs more than max_faces rows or has more than 16384 obstacle points inside its local box.
 * local_bbox must be positive (FASTER uses (2, 2, 1)).  The host version copies in and out and synchronises. */
int fh_decompose_batch_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_segments, int n_segments,
                              const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* d_faces,
                              int32_t* d_counts);
int fh_decompose_batch(fh_ctx* ctx, const double* cloud_xyz, int n_cloud, const double* segments, int n_segments,
                       const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* faces, int32_t* counts);

/* Corridors of a batch of paths, ready for fh_problem (faster.cpp:398-399: cvxEllipsoidDecomp of the path kept by createMoreVertexes /
 * deleteVertexes, then setPolytopes).  d_paths / d_n_points are what fh_map_plan_batch_device wrote with max_vertex_dist set; a path of
 * more than max_poly legs is cut to its first max_poly legs here (deleteVertexes, utils.cpp:1117-1124), so the caller can keep the whole
 * of JPS_in — fh_safe_corridor_batch_device marches along ALL of it, as the reference does (faster.cpp:446-452) — and still get the
 * corridor of JPS_whole.  Pair i gets one polytope per leg kept; its rows are stored back to back at
 * d_faces[i * faces_per_problem ...] with d_face_off[i][0..8] exactly as fh_problem.face_off wants them (face_begin =
 * i * faces_per_problem) and d_n_poly[i] = number of legs — 0 when the pair has no path, or a polytope exceeds FH_MAX_FACES_POLY rows,
 * or the rows do not fit faces_per_problem.  d_goal (may be NULL): the last vertex kept, the solver's E (faster.cpp:393-394).
 * Asynchronous on the context stream. */
int fh_corridor_batch_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_paths, const int32_t* d_n_points, int n,
                             int max_points, int max_poly, const double local_bbox[3], double drone_radius, double z_ground,
                             int faces_per_problem, fh_face* d_faces, int32_t* d_face_off, int32_t* d_n_poly, double* d_goal);

/* Problem records from corridors, on the device: the step between fh_corridor_batch_device and a solve (faster.cpp:393-404 for the whole
 * trajectory).  For every pair with a corridor (d_n_points[i] >= 2, d_n_poly[i] >= 1) record i gets its polytope table (n_poly, face_off,
 * face_begin = i * faces_per_problem), n_seg, and xf = the goal d_goals[i] when it lies in the LAST polytope, else the last vertex of the
 * path d_last_vertex[i] (E.pos, :399-400); everything else (x0, bounds, dc, factor window, force_final_pos) is left as the caller
 * prepared it.  Pairs without a corridor get n_seg = 0 (their results report FH_ST_BAD_INPUT).  Device pointers, asynchronous. */
int fh_corridor_problems_device(fh_ctx* ctx, const int32_t* d_n_points, const double* d_last_vertex, const double* d_goals, const fh_face* d_faces,
                                const int32_t* d_face_off, const int32_t* d_n_poly, int n, int faces_per_problem, int n_seg,
                                fh_problem* d_problems);

/* The SAFE corridor of Faster::replan, decomposed around R (faster/src/faster.cpp:446-524), for a batch of pairs whose whole
 * trajectories are solved — the faithful alternative to the hand-off of fh_pair_glue_device / fh_solve_pairs_device, which reuses
 * polytopes of the whole corridor.  Per pair:
 *   1. the path inside the sphere, JPS_in (d_paths [n][max_points][3], d_n_points [n]: what fh_map_plan_batch_device returns, first
 *      vertex = the start A; the WHOLE of JPS_in: plan with max_poly = 0, i.e. without deleteVertexes — the reference cuts its copy for
 *      the whole corridor only, faster.cpp:390-392), is cut where it first comes within drone_radius of unknown space and backed off by drone_radius
 *      (getFirstCollisionJPS(..., UNKNOWN_MAP, RETURN_INTERSECTION), :451-452);
 *   2. R = sample k_safe of the whole trajectory by the rule of the context (fh_set_pair_rule; r_frac in mode 0) becomes the first
 *      vertex and x0 of the safe problem; at most max_poly_safe legs are kept (:478-490); M = the last vertex;
 *   3. the path is decomposed against unknown + occupied points (:494): d_cloud_xyz are the occupied ones, the unknown ones are the
 *      voxels of `grid` (cell centres (i + 0.5) res + origin) farther than r_known from A, enumerated z-major and listed FIRST;
 *   4. d_safe[i]: x0 = R, xf = G (d_goals[i]) when G lies in the last polytope, else M (:498-499), n_poly / face_off / face_begin
 *      (= i * faces_per_problem, rows in d_safe_faces), n_seg = n_seg_safe; every other field is left as the caller prepared it.
 * Unknown space is MODELLED in rule modes 0 and 1 (a batch has no mapper): everything farther than fh_pair_rule.r_known from A.  In
 * rule mode 2 it is the caller's unknown voxel grid (fh_set_unknown_grid_device): step 1 marches along the path with the exact
 * distance to the nearest unknown voxel centre (the reference's kd-tree query), step 3 lists the unknown voxels of THAT grid first,
 * z-major, and `grid` of this call is not used.  A pair without a whole
 * trajectory, without a need for a safe one (rule mode 1) or without a corridor gets n_seg = 0.  Then solve d_safe with
 * fh_solve_batch_device and splice with fh_append_plans_device.  d_safe_paths [n][max_poly_safe + 1][3] / d_safe_n_points [n]: the
 * safe paths (may be NULL).  CPU restatement: oracle/pair_glue.py (safe_path, unknown_voxels) + the host decomposition.
 * Device pointers, asynchronous on the context stream. */
typedef struct fh_voxel_grid {
  double origin[3];
  double res;
  int32_t dims[3];
  int32_t reserved;
} fh_voxel_grid;
int fh_safe_corridor_batch_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const double* d_paths,
                                  const int32_t* d_n_points, int max_points, const double* d_goals, const double* d_cloud_xyz, int n_cloud,
                                  const fh_voxel_grid* grid, int n, double r_frac, int max_poly_safe, const double local_bbox[3],
                                  double drone_radius, double z_ground, int faces_per_problem, int n_seg_safe, fh_problem* d_safe,
                                  fh_face* d_safe_faces, double* d_safe_paths, int32_t* d_safe_n_points);

/* Whole solve -> hand-off -> safe solve of every pair in ONE launch: a wavefront takes a pair through fh_solve_batch_device,
 * fh_pair_glue_device and fh_solve_batch_device back to back (the per-pair dependency of Faster::replan, faster.cpp:427 -> :475 ->
 * :521-536), so no safe solve waits for the slowest whole solve of the batch.  Arguments and results are those of the three calls
 * (d_safe holds the safe problem templates on entry, as for fh_pair_glue_device); results are bit-identical to them.  The safe
 * problems must fit the same max_seg / max_faces bounds as the whole ones. */
int fh_solve_pairs_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_face* d_faces, int n, int max_seg, int max_faces,
                          double r_frac, double shrink, int max_safe_poly, fh_result* d_whole_results, fh_problem* d_safe,
                          fh_face* d_safe_faces, fh_result* d_safe_results);

int fh_sync(fh_ctx* ctx);

/* ---- one batch over several GPUs of a node ---------------------------------------------------------------
 * The reference is a single process on one CPU; its replacement for many independent start/goal/corridor problems shards ONE
 * batch over the devices of a node: device g of G solves the contiguous block [g*ceil(n/G), min(n, (g+1)*ceil(n/G))) (a
 * whole+safe pair never leaves its device), reading its block straight from the caller's host arrays (asynchronous H2D on its
 * own stream — the scatter needs no collective) and returning complete fh_result blocks.  Results are identical to the
 * single-device entry points (the problems are independent).  `devices`: n_devices HIP device indices (NULL: 0..n_devices-1;
 * n_devices <= 0: all visible devices; the same index may appear more than once — each entry gets its own context and stream).
 * No CPU fallback: FH_ERR_DEVICE without a device. */
typedef struct fh_pool fh_pool;
int fh_pool_create(fh_pool** out, const int* devices, int n_devices);
void fh_pool_destroy(fh_pool* pool);
int fh_pool_size(const fh_pool* pool);
const char* fh_pool_last_error(const fh_pool* pool);
int fh_pool_set_params(fh_pool* pool, const fh_params* p);
int fh_pool_set_pair_margin(fh_pool* pool, double r_margin);
int fh_pool_set_pair_rule(fh_pool* pool, const fh_pair_rule* rule);
/* fh_set_unknown_grid_device for every device of the pool: `flags` is HOST memory here (dims[0] * dims[1] * dims[2] bytes, x fastest);
 * each device gets its own copy (synchronous).  flags = NULL: no unknown grid. */
struct fh_voxel_grid;
int fh_pool_set_unknown_grid(fh_pool* pool, const struct fh_voxel_grid* grid, const unsigned char* flags);
/* fh_solve_batch over the pool (host pointers, synchronous).  The result blocks are gathered into `results` (host, may be NULL)
 * and/or into `d_results_root`, n records in the memory of pool device number `root`, with peer copies over xGMI
 * (hipMemcpyPeerAsync) — for a consumer that lives on that GPU.  At least one destination must be given. */
int fh_pool_solve_batch(fh_pool* pool, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n, fh_result* results,
                        int root, fh_result* d_results_root);
/* whole solve -> hand-off -> safe solve of n pairs over the pool (fh_solve_pairs_device per block; safe_templates as d_safe of
 * fh_pair_glue_device).  Both result arrays are gathered like fh_pool_solve_batch's. */
int fh_pool_solve_pairs(fh_pool* pool, const fh_problem* whole, const fh_face* faces, int64_t n_faces, int n,
                        const fh_problem* safe_templates, double r_frac, double shrink, int max_safe_poly, fh_result* whole_results,
                        fh_result* safe_results, int root, fh_result* d_whole_results_root, fh_result* d_safe_results_root);

/* ---- voxel map + batched path search on the device (first half) ------------------------------------------
 * fh_map replaces the pair JPS_Manager::updateJPSMap / JPS_Manager::solveJPS3D (faster/src/jps_manager.cpp:129-139, :141-200):
 *   fh_map_read[_device]   = MapUtil::readMap (faster/include/read_map.hpp:30-185): occupancy grid of cells[] cells of `res` metres
 *                            centred on `center` (x, y widened by 5*inflation/res cells, z clipped to [z_ground, z_max]); every cloud
 *                            point marks its cell and the cube of +-floor(inflation/res) cells around it;
 *   fh_map_plan_batch[_device] = solveJPS3D for n independent start/goal queries over that map: z clamped to >= 0, the cells around
 *                            start and goal freed (setFreeVoxelAndSurroundings; per query, the map itself is not modified), optimal
 *                            26-connected grid path with Euclidean costs, jps3d's clean-up (removeLinePts, removeCornerPts forwards and
 *                            backwards), ends forced onto the requested points.  One query per wavefront.
 *   With max_vertex_dist > 0 the vertices additionally go through Faster::createMoreVertexes (faster/src/faster.cpp:80-97) and, with
 *   max_poly > 0, deleteVertexes (faster/src/utils.cpp:1117-1124): what Faster::replan hands to the convex decomposition.
 * paths: [n][max_points][3]; n_points[i]: number of vertices, 0 = no path (start/goal outside the map or not connected),
 * -1 = more than max_points vertices, -2 = a search limit was hit (f >= 2040 cells, 131072 open entries, 4096 raw path cells).
 * expansions (may be NULL): cells expanded per query.  The CPU restatement the kernels are checked against vertex for vertex is
 * faster_amd/host/corridor_frontend.cpp (plan_path).  One stream per map; entry points of one map are not re-entrant.
 * No CPU fallback: FH_ERR_DEVICE without a device. */
typedef struct fh_map fh_map;
int fh_map_create(fh_map** out, int device);
void fh_map_destroy(fh_map* map);
const char* fh_map_last_error(const fh_map* map);
int fh_map_set_stream(fh_map* map, void* hip_stream);
/* Scheduling of the path search (results do not depend on it): wavefronts per CU that hold a search workspace (1..20; 0 = default:
 * 12 for the A* search, 20 for the jump point search),
 * and whether batches larger than the number of wavefronts start with the far-apart start/goal pairs (default 1). */
int fh_map_set_sched(fh_map* map, int waves_per_cu, int launch_order);
/* Which search fh_map_plan_batch* runs.  0 (default): A* with a total order of its own — an optimal path; equals the host restatement
 * plan_path bit for bit.  1: jump point search with jps3d's pruning rules, successor order, tolerance comparator and binary heap
 * (thirdparty/jps3d/src/jps_planner/graph_search.cpp:123-470) — of the optimal paths, the one FASTER itself gets from
 * planner_ptr_->plan(start, goal, 1, true) (faster/src/jps_manager.cpp:166); equals plan_path_jps bit for bit, which is pinned vertex
 * for vertex to the reference's compiled sources behind a test-only Boost.Heap shim (oracle/ref_frontend/shim: the sift discipline that
 * decides between equal-cost paths is the shim author's reading of Boost's, not Boost itself).  Limits in mode 1: 78311 open entries, 4096 jump points on the path.
 * Switching re-initialises the search workspace. */
int fh_map_set_search(fh_map* map, int mode);
/* How the jump point search (mode 1) keeps its per-cell records (g, parent, direction, closed) — jps3d's hm_ / seen_
 * (graph_search.h:258-259, sized by the map in the constructor, graph_search.cpp:25-26: one entry per cell of the map, allocated per query).  slots = 0: one 16-byte record per cell of
 * the map and wavefront, stamped with the query's serial number instead of being cleared.  slots = a power of two in [1024, 2^22]:
 * a hashed table of that many records per wavefront (39 bytes per slot with the heap levels that go with it, whatever the size of
 * the map), holding the cells the running query has reached; a query that reaches more than 3/4 of `slots` cells returns
 * n_points = -2 (raise slots, or use 0).  The same reads and writes in the same order either way: identical paths.
 * slots = -1 (default): per-cell records while the 48 GB workspace budget holds them for at least half of the wavefronts (faster on
 * small maps: 54 vs 66 ms for 65536 queries in 181 500 cells — and without a limit on the cells a query may reach), else 131072
 * hashed slots (1 452 000 cells: 298 ms and 6.5 GB with 32768 slots against 476 ms and 51.5 GB).  With the default, a query that ends
 * at the table's limit (an unreachable goal on a big map reaches more cells than 3/4 of 131072) is run AGAIN with per-cell records by
 * the host-pointer entry point fh_map_plan_batch (synchronous anyway; the workspace is rebuilt twice: the rare case), so that "no path"
 * (0) and a path come out as with slots = 0; fh_map_plan_batch_device is asynchronous and reports -2 for such a query — the caller
 * decides (fh_map_set_records(map, 0) and plan those queries again).
 * The A* search (mode 0) always uses per-cell records.
 * fh_map_workspace_bytes: size of the search workspace as allocated by the last search (0 before the first). */
int fh_map_set_records(fh_map* map, int slots);
long long fh_map_workspace_bytes(const fh_map* map);
/* JPS_in of Faster::replan (faster/src/faster.cpp:370-382): with ra > 0 every path of fh_map_plan_batch* is cut at its first crossing of
 * the sphere of radius min(|goal - start| - 0.001, ra) around its start (getFirstIntersectionWithSphere, utils.cpp:782-870, with the
 * reference's single-precision crossing), the crossing point E appended, BEFORE createMoreVertexes / deleteVertexes.  0 (default): off. */
int fh_map_set_sphere(fh_map* map, double ra);
int fh_map_sync(fh_map* map);
int fh_map_read(fh_map* map, const double* cloud_xyz, int n_cloud, const int32_t cells[3], double res, const double center[3],
                double z_ground, double z_max, double inflation);
int fh_map_read_device(fh_map* map, const double* d_cloud_xyz, int n_cloud, const int32_t cells[3], double res, const double center[3],
                       double z_ground, double z_max, double inflation);
int fh_map_dims(const fh_map* map, int32_t dims[3], double origin[3]);
int fh_map_occupancy(fh_map* map, int8_t* occ); /* [nz][ny][nx], 0 free / 100 occupied, as MapUtil stores it */
int fh_map_plan_batch(fh_map* map, const double* starts, const double* goals, int n, int max_points, double max_vertex_dist,
                      int max_poly, double* paths, int32_t* n_points, int64_t* expansions);
int fh_map_plan_batch_device(fh_map* map, const double* d_starts, const double* d_goals, int n, int max_points, double max_vertex_dist,
                             int max_poly, double* d_paths, int32_t* d_n_points, int64_t* d_expansions);

/* Timing of the solve kernel, measured with HIP events recorded around every solve-kernel launch on
 * the context stream (the same stream the kernel runs on).  fh_timing_reset() forgets recorded launches;
 * fh_timing_read() synchronises with the last recorded launch and writes the duration (ms) of up to
 * `cap` launches recorded since the reset, oldest first; returns the number recorded (may exceed cap)
 * or <0 on error.  fh_last_kernel_ms() is the duration of the most recent launch (<0 if none). */
int fh_timing_reset(fh_ctx* ctx);
int fh_timing_read(fh_ctx* ctx, double* ms, int cap);
double fh_last_kernel_ms(fh_ctx* ctx);

/* Which solve kernel the most recent solve launch of the context ran, as the profiler names it:
 * fh::solve_kernel<n_seg, pairs, waves_per_simd, unknown_space> — the instantiation (6 / 10 / 15 / 16 segments), the fused pair form, the
 * build (3 = three wavefronts per SIMD, 168 registers; 2 = two, all registers: fh_sched.workgroups_per_cu) and whether the hand-off asks
 * the caller's unknown voxels (fh_pair_rule mode 2) — with its grid, the resident solves per CU and the LDS bytes per workgroup.  Measurement only (bench.py matches its rocprofv3 summaries by this name);
 * returns FH_ERR_ARG before the first launch. */
typedef struct fh_launch_info {
  int32_t n_seg, pairs, waves_per_simd, grid, workgroups_per_cu, lds_bytes, unknown_space, look_every;
} fh_launch_info;
int fh_last_launch(const fh_ctx* ctx, fh_launch_info* out);

/* library / build identification, e.g. "fasterhip 0.1 gfx950" */
const char* fh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTERHIP_H */
