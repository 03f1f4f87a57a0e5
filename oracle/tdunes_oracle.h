/*
 * tdunes_oracle.h -- CPU restatement of treeQP's dual-Newton-on-tree ("tdunes") hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / CPU baseline.  The product path (treeqp_amd/csrc) never links or calls it.
 *
 * The oracle follows the reference algorithm phase by phase on flat (structure-of-arrays)
 * buffers; every function cites the reference file:line it restates (paths relative to the
 * reference repository root).  All floating-point work of the reference happens inside the
 * third-party BLASFEO library (giaf/blasfeo, un-vendored submodule external/blasfeo, pinned
 * commit unknown, API vintage 0.1.x); the BLAS-level operations are restated here from
 * BLASFEO's published semantics with a fixed, documented summation order (k ascending).
 *
 * Pinning (see DESIGN.md "Oracle"): the reference treeqp layer cannot be built in this image
 * without writing a stand-in for BLASFEO, so there is no oracle/_ref build of the solver.  The
 * oracle is pinned by the reference's own fixtures for this path:
 *   - examples/spring_mass_utils/{data.c,x0.txt,lambda0_tree.txt} with the assert
 *     KKT < 1e-8 of examples/spring_mass_dual_newton_tree.c:154-157,
 *   - the x0-eliminated case of examples/spring_mass.c:304-331 (KKT < 1e-10),
 *   - examples/thesis_example.c,
 *   - the six YALMIP/quadprog golden solutions examples/random_qp_utils/data0[0-5].json
 *     (max |x-xopt|,|u-uopt| < 1e-12, examples/random_qp.c:249-254) through the dense
 *     unconstrained stage solver oracle_tdunes_solve_dense().
 *
 * Flat data layout (identical to the reference's "ltv" setters, tree_qp_common.c:1952-2090):
 *   per node  k = 0..Nn-1 : Qd,q,xmin,xmax (nx[k] each, concatenated), Rd,r,umin,umax (nu[k])
 *   per edge  e = k-1     : A (nx[k] x nx[dad], column major), B (nx[k] x nu[dad]), b (nx[k])
 *   lambda / lam          : concatenation over k = 1..Nn-1 of nx[k] (block order == edge order)
 */
#ifndef TDUNES_ORACLE_H_
#define TDUNES_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* enums mirror treeqp/utils/types.h:46-85 and treeqp/src/dual_Newton_common.h:41-46 */
enum { ORC_SUMSQUAREDERRORS = 0, ORC_TWONORM = 1, ORC_INFNORM = 2 };
enum { ORC_NO_REG = 0, ORC_ALWAYS_LM = 1, ORC_ON_THE_FLY_LM = 2 };
enum { ORC_OPTIMAL = 0, ORC_MAXITER = 1, ORC_NOT_DESCENT = 2, ORC_INVALID_OPTION = 9 };

typedef struct {
    int maxIter;               /* dual_Newton_tree.c:94  (100)   */
    int termCondition;         /* :95   (INFNORM)                */
    double stationarityTolerance; /* :96 (1e-8)                  */
    int checkLastActiveSet;    /* :98   (1)                      */
    int lineSearchMaxIter;     /* :112  (50)                     */
    double lineSearchGamma;    /* :113  (0.1)                    */
    double lineSearchBeta;     /* :114  (0.6)                    */
    int lineSearchRestartTrigger; /* :115 (-1)                   */
    int regType;               /* :117  (ON_THE_FLY)             */
    double regTol;             /* :118  (1e-6)                   */
    double regValue;           /* :119  (1e-6)                   */
    int num_threads;           /* OpenMP threads for the cpu_baseline leg (<=1: serial) */
} oracle_opts_t;

typedef struct {
    int status;                /* return_t value                               */
    int iter;                  /* info.iter  (dual_Newton_tree.c:1248)         */
    int ls_total;              /* sum of lsIter over iterations                */
    int n_active;              /* number of active bounds at exit (diagnostic) */
    int n_regularized;         /* number of blocks that got on-the-fly / always LM shifts (diagnostic) */
    double solver_time;        /* seconds, Newton loop only                    */
    /* per-iteration traces, length maxIter+1 (caller may pass NULL) */
    double *trace_err;         /* termination norm checked at iteration i      */
    double *trace_fval;        /* accepted dual value after line search        */
    int *trace_ls;             /* line-search trials of iteration i            */
} oracle_info_t;

void oracle_opts_set_default(oracle_opts_t *opts);

/* ---- integer tree logic (must be bit-exact) ------------------------------------------- */
int oracle_ipow(int base, int exp);                                   /* utils.c:34-47   */
int oracle_calculate_number_of_nodes(int md, int Nr, int Nh);         /* tree.c:36-48    */
int oracle_number_of_nodes_from_nkids(const int *nk);                 /* tree.c:105-126  */
void oracle_setup_multistage_tree(int md, int Nr, int Nh, int *nk);   /* tree.c:247-280  */
/* tree.c:171-243: fills dad, nkids(=nk), stage, real, idxkid and kid0 (index of first child,
 * children are contiguous); returns number of parents Np (tree.c:52-61) */
int oracle_tree_create(int Nn, const int *nk, int *dad, int *stage, int *real, int *idxkid,
                       int *kid0);
/* dual_Newton_tree.c:177-194 and :166-173 */
void oracle_setup_idxpos(int Nn, const int *dad, const int *idxkid, const int *kid0,
                         const int *nx, int *idxpos);
void oracle_setup_npar(int Nn, const int *stage, int Nh, int *npar);

/* ---- LTI filler (tree_qp_common.c:1837-1949) on flat arrays ----------------------------
 * A,B,b hold `n_real` realizations (nx*nx, nx*nu, nx each); Q,q,P,p (nx), R,r (nu) diagonal
 * weights; bounds; x0.  Uniform nx (all nodes) and nu (parents; leaves get nu=0).  Outputs are
 * the flat per-edge / per-node arrays described at the top (caller allocates). */
void oracle_fill_lti_diag(int Nn, const int *nk, int nx, int nu,
                          const double *A, const double *B, const double *b,
                          const double *Qd, const double *q, const double *Pd, const double *p,
                          const double *Rd, const double *r,
                          const double *xmin, const double *xmax,
                          const double *umin, const double *umax, const double *x0,
                          double *oA, double *oB, double *ob,
                          double *oQd, double *oRd, double *oq, double *orr,
                          double *oxmin, double *oxmax, double *oumin, double *oumax);

/* ---- the solver (dual_Newton_tree.c:1104-1263 + clipping.c + dual_Newton_common.c) ------ */
int oracle_tdunes_solve(int Nn, const int *nk, const int *nx, const int *nu,
                        const double *A, const double *B, const double *b,
                        const double *Qd, const double *Rd, const double *q, const double *r,
                        const double *xmin, const double *xmax,
                        const double *umin, const double *umax,
                        const oracle_opts_t *opts, const double *lambda0,
                        double *x, double *u, double *lam, double *mu_x, double *mu_u,
                        oracle_info_t *info);

/* Same outer algorithm with a dense, unconstrained stage solver (H_k = [Q S';S R] Cholesky,
 * elimination matrix P = H^-1) standing in for the absent qpOASES backend on the
 * UNCONSTRAINED golden fixtures (dual_Newton_tree_qpoases.c:401-476 semantics when no bound
 * is active).  Q (nx x nx), R (nu x nu), S (nu x nx) column major, concatenated per node. */
int oracle_tdunes_solve_dense(int Nn, const int *nk, const int *nx, const int *nu,
                              const double *A, const double *B, const double *b,
                              const double *Q, const double *R, const double *S,
                              const double *q, const double *r,
                              const oracle_opts_t *opts, const double *lambda0,
                              double *x, double *u, double *lam, oracle_info_t *info);

/* ---- KKT residual (tree_qp_common.c:540-788), diagonal or dense weights ------------------
 * Q/R/S dense may be NULL (then Qd/Rd are used); returns max |.| over 3*nz+ne entries. */
double oracle_max_kkt(int Nn, const int *nk, const int *nx, const int *nu,
                      const double *A, const double *B, const double *b,
                      const double *Qd, const double *Rd,
                      const double *Q, const double *R, const double *S,
                      const double *q, const double *r,
                      const double *xmin, const double *xmax,
                      const double *umin, const double *umax,
                      const double *x, const double *u, const double *lam,
                      const double *mu_x, const double *mu_u);

#ifdef __cplusplus
}
#endif
#endif  /* TDUNES_ORACLE_H_ */
