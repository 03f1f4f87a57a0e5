/*
 * treeqp_amd.h -- C-ABI of the MI355X (gfx950) device path of the tdunes hot path.
 *
 * This is the drop-in boundary: plain `extern "C"`, pointers and sizes only, no C++/torch types.
 * The reference has no FFI for this path (it is a static C library, Makefile:55-63); the entry
 * points below are what its own solver front end binds, one per reference interface:
 *
 *   tqgpu_create / tqgpu_destroy    <- treeqp_tdunes_calculate_size + treeqp_tdunes_create
 *                                      (dual_Newton_tree.c:1291-1407, 1411-1648): workspace sizing,
 *                                      setup_npar/setup_idxpos (:166-194) become device tables
 *   tqgpu_set_dynamics / _objective_diag / _bounds
 *                                   <- the QP values the solver reads from tree_qp_in every solve
 *                                      (tree_qp_common.h:85-115) + stage_qp_clipping_init
 *                                      (dual_Newton_tree_clipping.c:149-184: Qinv = 1/diag(Q))
 *   tqgpu_set_lambda                <- treeqp_tdunes_set_dual_initialization (dual_Newton_tree.c:1654-1663)
 *   tqgpu_solve                     <- the Newton loop of treeqp_tdunes_solve (dual_Newton_tree.c:1166-1228):
 *                                      solve_stage_problems, build_dual_problem, calculate_delta_lambda,
 *                                      line_search, all as HIP kernels
 *   tqgpu_get_solution              <- the export block of treeqp_tdunes_solve (:1235-1247) + export_mu
 *                                      (dual_Newton_tree_clipping.c:386-399)
 *
 * Flat data layout ("ltv" order of tree_qp_common.c:1952-2090):
 *   per node k: Qd,q,xmin,xmax (nx[k] doubles each, nodes concatenated); Rd,r,umin,umax (nu[k])
 *   per edge e=k-1: A (nx[k] x nx[dad(k)], column major), B (nx[k] x nu[dad(k)]), b (nx[k])
 *   lambda / lam / dlam: concatenation over k=1..Nn-1 of nx[k] doubles
 *
 * All functions return 0 on success, a negative TQGPU_E* code otherwise; tqgpu_last_error()
 * returns a human-readable message for the calling thread.  Nothing here falls back to the CPU.
 */
#ifndef TREEQP_AMD_H_
#define TREEQP_AMD_H_
#ifdef __cplusplus
extern "C" {
#endif

#define TQGPU_OK 0
#define TQGPU_ENODEVICE (-1)     /* no usable HIP device / HIP runtime error */
#define TQGPU_EINVAL (-2)        /* inconsistent tree / dimensions / options */
#define TQGPU_ENOMEM (-3)
#define TQGPU_EUNSUPPORTED (-4)  /* e.g. dual block too large for the LDS-resident kernels */
#define TQGPU_ECOMM (-5)         /* RCCL failure */
#define TQGPU_ETIMEOUT (-6)      /* a bounded wait inside the persistent launch gave up (its workgroups were not all resident: the device is
                                    shared); tqgpu_solve does not return this -- it redoes the solve on the launch-per-tier / per-level path */

typedef struct tqgpu_solver tqgpu_solver;

typedef struct tqgpu_opts {
    int maxIter;
    int termCondition;           /* termination_t value */
    double stationarityTolerance;
    int regType;                 /* regType_t value */
    double regTol, regValue;
    int lineSearchMaxIter;
    double lineSearchGamma, lineSearchBeta;
    int lineSearchRestartTrigger;
    int profile;                 /* 0: total device time only; 1: per-iteration event timing; 3: + per-phase event timing (launch-per-level path) */
    int checkLastActiveSet;      /* treeqp_tdunes_opts_t.checkLastActiveSet (dual_Newton_tree.c:98, default 1 there).  In the reference the option
                                    changes the work, never the result (a kept Cholesky factor is bit-identical to a rebuilt one), so 0 and 1 both
                                    run the default kernel, which rebuilds every pass.  2 selects the kernel variant that keeps the factor data of
                                    a workgroup (tier subtree) whose active set did not change and only substitutes (persistent path only); its
                                    steps agree with rebuilt ones to rounding, not bit for bit -- see DESIGN.md "checkLastActiveSet" */
} tqgpu_opts;

typedef struct tqgpu_result {
    int status;                  /* return_t value: 0 optimal, 1 max iterations, 2 not a descent direction */
    int iter;                    /* Newton iterations */
    int ls_total;                /* total line-search trials */
    int ls_last;                 /* trials of the last iteration */
    int n_launches;              /* kernels launched during the solve */
    double device_time;          /* seconds between the first and last kernel (HIP events) */
    double last_error_norm;      /* termination norm at exit */
    double last_fval;            /* dual function value at the last accepted point */
} tqgpu_result;

int tqgpu_device_count(void);
const char *tqgpu_last_error(void);
const char *tqgpu_version(void);

int tqgpu_create(tqgpu_solver **out, int device, int Nn, const int *nk, const int *nx, const int *nu);
void tqgpu_destroy(tqgpu_solver *s);

int tqgpu_set_dynamics(tqgpu_solver *s, const double *A, const double *B, const double *b);
int tqgpu_set_objective_diag(tqgpu_solver *s, const double *Qd, const double *Rd, const double *q, const double *r);
/* Dense objective + dense UNCONSTRAINED stage solver: replaces the reference's qpOASES stage backend
 * (dual_Newton_tree_qpoases.c:153-217 init/solve, :401-476 elimination matrices) for nodes without
 * bounds: z_k = H_k^-1 h_k, P_k = H_k^-1 with H_k = [Q S'; S R].  Flat layout as
 * tree_qp_in_set_ltv_objective_colmajor (tree_qp_common.c:2010-2050): per node Q (nx x nx), R (nu x nu),
 * S (nu x nx) column major, then q, r.  Bounds are ignored while it is selected. */
int tqgpu_set_objective_dense(tqgpu_solver *s, const double *Q, const double *R, const double *S, const double *q, const double *r);
/* the same with a per-node choice (opts->qp_solver[] of the reference, dual_Newton_tree.c:124-162): kind[k] = 0 clipping (the
 * diagonals of Q_k and R_k are the weights; off-diagonals and S_k must be zero), 1 dense unconstrained; NULL = all dense */
int tqgpu_set_objective_mixed(tqgpu_solver *s, const int *kind, const double *Q, const double *R, const double *S, const double *q, const double *r);
int tqgpu_set_bounds(tqgpu_solver *s, const double *xmin, const double *xmax, const double *umin, const double *umax);
/* Everything above in one call (NULL = leave alone) plus the starting duals: compared with a pinned host mirror of
 * what the device holds, only what changed is uploaded, without synchronisation.  This is what the drop-in
 * treeqp_tdunes_solve uses, which -- like the reference, dual_Newton_tree.c:1142-1160 -- re-reads qp_in at every solve. */
int tqgpu_set_problem(tqgpu_solver *s, const double *A, const double *B, const double *b,
                      const double *Qd, const double *Rd, const double *q, const double *r,
                      const double *xmin, const double *xmax, const double *umin, const double *umax, const double *lambda);
int tqgpu_set_lambda(tqgpu_solver *s, const double *lambda);   /* NULL = zeros */

int tqgpu_solve(tqgpu_solver *s, const tqgpu_opts *opts, tqgpu_result *res);

/* on != 0: tqgpu_solve enqueues the packing kernel and the download of the solution right behind a single persistent launch, while it
 * runs; the tqgpu_get_solution that follows only waits for the copy.  For callers that fetch the solution after every solve (the
 * drop-in front end, treeqp_tdunes_solve -> dual_Newton_tree.c:1235-1247); off by default (a caller that only wants the verdict pays
 * for nothing).  Results are the same either way. */
int tqgpu_set_export_ahead(tqgpu_solver *s, int on);

/* any output pointer may be NULL */
int tqgpu_get_solution(tqgpu_solver *s, double *x, double *u, double *lam, double *mu_x, double *mu_u, double *dlam);

/* 2: persistent single-launch solve (tdunes_persist.hpp); 1: tiered fused kernels (tdunes_fast.hpp);
 * 0: generic per-level kernels.  TREEQP_AMD_PATH=generic|tiered in the environment at create time
 * forces the lower paths. */
int tqgpu_uses_fused_path(const tqgpu_solver *s);
/* geometry of the persistent launch: block levels, tiers, workgroups per launch, co-resident workgroup capacity of the device, CUs */
int tqgpu_geometry(const tqgpu_solver *s, int *levels, int *tiers, int *workgroups, int *capacity, int *compute_units);
/* diagnostic: persistent launches of this mirror that timed out (device shared with other work) and were redone on another path */
int tqgpu_timeouts(const tqgpu_solver *s);

/* sizes of the flat arrays, for callers that did not keep them */
int tqgpu_dims(const tqgpu_solver *s, int *sum_nx, int *sum_nu, int *sum_lam, int *sum_A, int *sum_B);

/* per-iteration line-search counts and event times (seconds) of the last solve; arrays of
 * length >= iter; times are NaN unless opts.profile != 0 */
int tqgpu_get_iteration_log(tqgpu_solver *s, int *ls_iters, double *iter_times, int cap);

/* opts.profile >= 3: device time per Newton iteration of the reference's phases (treeqp/utils/profiling.h:58-67): build_dual,
 * newton_direction, line_search; stage_qps[0] = first sweep of the solve, 0 afterwards (phase S of a later iteration IS the accepted
 * trial sweep of the line search before it).  Arrays of length >= cap (any may be NULL); returns the number of iterations written. */
int tqgpu_get_phase_log(tqgpu_solver *s, double *stage_qps, double *build_dual, double *newton_direction, double *line_search, int cap);
/* roofline support: algorithmic bytes and flops of ONE Newton iteration with n_ls line-search
 * trials (closed form of SURVEY.md §8(d) generalised to per-node dimensions) */
int tqgpu_iteration_cost(const tqgpu_solver *s, int n_ls, double *bytes, double *flops);

/* The partition plan of the sharded mode without a device: which tier is the highest partitioned one, the boundary level, the
 * blocks above the bottom tier whose gradient / Hessian this rank computes (the first gh_counted of them enter its termination
 * partial) and the nodes it owns.  Host arithmetic only (the CPU tests compare it with treeqp_amd/sharding.py). */
int tqgpu_shard_plan(int md, int nx, int Nh, int nranks, int rank, int *part_top, int *boundary_level, int *gh_counted,
                     int *gh_list, int gh_cap, int *gh_n, int *owned_nodes, int owned_cap, int *owned_n);
/* ---- one tree sharded over several devices --------------------------------------
 * Every rank creates a mirror of the WHOLE tree and uploads the whole problem; tqgpu_shard_init then
 * restricts the rank's work to a contiguous range of subtrees (tiers above the partition boundary
 * are replicated).  Per Newton iteration two small in-place all-gathers over RCCL exchange the
 * boundary Schur records + termination partials and the {fval, dot} partials + boundary x/QinvCal.
 * id128: 128-byte RCCL unique id created by rank 0 (tqgpu_shard_unique_id) and broadcast by the
 * caller (e.g. through torch.distributed); NULL creates a "virtual rank" without a communicator. */
int tqgpu_shard_unique_id(void *id128);
int tqgpu_shard_init(tqgpu_solver *s, int rank, int nranks, const void *id128);
int tqgpu_shard_gather_solution(tqgpu_solver *s);
/* test / diagnostic: n virtual ranks of one tree in one process on one device, lock-step */
int tqgpu_solve_virtual_ranks(tqgpu_solver **ranks, int n, const tqgpu_opts *opts, tqgpu_result *res);

/* ---- one tree sharded over several devices INSIDE the persistent launch ----------
 * The workgroups of the single persistent launch (one per tier subtree, tdunes_persist.hpp) are dealt over one launch per
 * rank: tiers whose subtree count is a multiple of the number of ranks by contiguous subtree ranges (the independent subtrees
 * of dual_Newton_tree.c:668-775), the tiers above them to rank 0.  What crosses ranks -- the Schur push of
 * dual_Newton_tree.c:726-732, the forward pull of :768-769, the global scalars of :542, :944-945, :970 -- travels as the
 * same tagged words as on one device: every rank has a hand-over slab, a producer writes each word into every rank's slab
 * (system-scope stores into peer-mapped memory), a consumer polls its own.  No collective and no host in the loop; the
 * solution is collected afterwards by any transport (tqgpu_pshard_pack / _unpack: equal-sized buffers for an all-gather).
 *   tqgpu_pshard_init          this mirror (whole tree created, whole problem uploaded) becomes rank `rank` of `nranks` (<= 8)
 *   tqgpu_pshard_connect_local peer r is another mirror of this process (same device: rehearsal; other device: peer access)
 *   tqgpu_pshard_ipc_export / _ipc_connect   peers in other processes: 64-byte IPC handle of the slab out / of peer r in
 *   tqgpu_pshard_begin / _end  one solve: enqueue this rank's launch / wait for the verdict.  All ranks' launches must be in
 *                              flight together (they wait for each other: bounded, TQGPU_ETIMEOUT after 0.5 s) and every rank
 *                              must solve the same number of times (the launch number tags the words)
 *   tqgpu_pshard_solve_local   n mirrors of one process: connect, solve, collect the solution into every mirror */
/* the partition without a device (host arithmetic; tqgpu_pshard_init uses it): this rank's workgroups of the persistent launch */
int tqgpu_pshard_plan(int md, int Nh, int nranks, int rank, int *wgs, int cap, int *n, int *part_top, int *boundary_level);
int tqgpu_pshard_init(tqgpu_solver *s, int rank, int nranks);
int tqgpu_pshard_connect_local(tqgpu_solver *s, int r, tqgpu_solver *peer);
int tqgpu_pshard_ipc_export(tqgpu_solver *s, void *handle64);
int tqgpu_pshard_ipc_connect(tqgpu_solver *s, int r, const void *handle64);
int tqgpu_pshard_begin(tqgpu_solver *s, const tqgpu_opts *opts);
/* after 65535 sharded solves tqgpu_pshard_begin refuses (the 16-bit launch number tags the hand-over words): every rank then calls
 * tqgpu_pshard_rewind -- launch numbers back to zero, slab wiped, peers stay connected -- with no sharded solve in flight anywhere, i.e.
 * between two barriers of the caller's.  tqgpu_pshard_solve_local does it by itself. */
int tqgpu_pshard_rewind(tqgpu_solver *s);
int tqgpu_pshard_end(tqgpu_solver *s, tqgpu_result *res);
long tqgpu_pshard_pack_size(tqgpu_solver *s);
int tqgpu_pshard_pack(tqgpu_solver *s, double *out, long cap);
int tqgpu_pshard_unpack(tqgpu_solver *s, int src_rank, const double *in, long n_in);
int tqgpu_pshard_solve_local(tqgpu_solver **ranks, int n, const tqgpu_opts *opts, tqgpu_result *res);

/* n solves of the same problem from the same starting duals, one after the other, each waiting for its verdict: the timing loop
 * of the reference's drivers (examples/spring_mass_dual_newton_tree.c:135-140, `for (jj = 0; jj < NREP; jj++)
 * treeqp_tdunes_solve(...)`) in C.  res: the last solve; sums over the solves in *iter_sum, *ls_sum, *launch_sum (each may be NULL). */
int tqgpu_solve_n(tqgpu_solver *s, const tqgpu_opts *opts, int n, tqgpu_result *res, long *iter_sum, long *ls_sum, long *launch_sum);
/* the same loop for a batch: `steps` calls of tqgpu_solve_batch(solvers, n, ...) one after the other (the scenario sweep of
 * examples/fault_tolerance.c:486-530 repeated, e.g. once per MPC step); results: the n results of the last call; sums over all
 * calls and members in *iter_sum, *ls_sum, *launch_sum (each may be NULL) */
int tqgpu_solve_batch_n(tqgpu_solver **solvers, int n, const tqgpu_opts *opts, int steps, tqgpu_result *results, long *iter_sum, long *ls_sum, long *launch_sum);

/* Batched multi-tree solve: n independent mirrors with the same options (examples/fault_tolerance.c:486-530
 * holds one tree_qp_in / workspace pair per configuration and solves them one after the other).  Mirrors whose
 * solve is a single persistent launch run concurrently, as many as fit on the device at once. */
int tqgpu_solve_batch(tqgpu_solver **solvers, int n, const tqgpu_opts *opts, tqgpu_result *results);

/* Device times [s] of the last n solves (oldest first), measured with HIP events on the solver's
 * stream around everything a solve enqueues; synchronises the stream; returns the number written
 * (at most 512 solves back) or -1.  tqgpu_result.device_time of a single-launch (persistent) solve is
 * the kernel's own clock from launch start to verdict instead, so that tqgpu_solve can return as soon
 * as the verdict is in pinned host memory. */
int tqgpu_get_device_times(tqgpu_solver *s, double *out, int n);
/* Per-solve event pairs on (default) or off.  Off: a solve that is ONE persistent launch is enqueued with nothing around it
 * (two queue packets fewer per solve); tqgpu_get_device_times then reports NaN for such solves.  No counterpart in the
 * reference (its timers are host timers, treeqp/utils/timing.c:39-61). */
int tqgpu_set_event_timing(tqgpu_solver *s, int on);

/* diagnostic in-kernel time stamps of the last fused iteration (TREEQP_AMD_STAMPS=1) */
int tqgpu_get_stamps(tqgpu_solver *s, unsigned long long *out, int cap);
/* test support: a foreign kernel holding compute units (blocks x 256 threads, lds_kb KiB of LDS each, spinning for ms milliseconds
 * on its own stream; returns at once), and the wait for it */
int tqgpu_debug_occupy(int device, int blocks, int lds_kb, int ms);
int tqgpu_debug_occupy_wait(void);

/* sizeof() of the public structs (0 dmat, 1 dvec, 2 node, 3 tree_qp_in, 4 tree_qp_out,
 * 5 tdunes opts, 6 tdunes workspace, 7 profiling record, 8 qp_internal_t) for FFI self-checks */
int treeqp_amd_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_AMD_H_ */
