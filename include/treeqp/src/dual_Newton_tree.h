/*
 * treeqp_amd: dual Newton strategy on the tree formulation ("tdunes"), MI355X build.
 *
 * Drop-in for the reference's treeqp/src/dual_Newton_tree.h:67-172: same option struct, same
 * entry points and signatures.  What is behind them differs by design:
 *   - treeqp_tdunes_create builds a device mirror (index tables + slabs in HBM) through the
 *     C-ABI in treeqp_amd.h; the caller-owned buffer only carries host-visible mirrors;
 *   - treeqp_tdunes_solve stages the QP values, runs the whole Newton loop as HIP kernels
 *     (phases S/G/H/F/L of SURVEY.md §3.2) and copies x,u,lambda,mu back;
 *   - there is NO host implementation of the solve: without a usable HIP device
 *     treeqp_tdunes_create prints the HIP error and exits(1) (the reference's convention for
 *     fatal configuration errors, dual_Newton_tree_clipping.c:70-74).
 * The reference's 13-entry per-node stage-QP vtable (dual_Newton_tree.h:48-63) is replaced by
 * one batched kernel family; opts->qp_solver[] is honoured as a per-node selector:
 * TREEQP_CLIPPING_SOLVER (diagonal weights, box bounds) and TREEQP_QPOASES_SOLVER restricted to
 * nodes WITHOUT bounds (dense unconstrained stage QP, solved on the device) mix freely in one
 * tree; a dense node with finite bounds would need an active-set stage solver and is rejected
 * at create time.
 */
#ifndef TREEQP_SRC_DUAL_NEWTON_TREE_H_
#define TREEQP_SRC_DUAL_NEWTON_TREE_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/src/dual_Newton_common.h"
#include "treeqp/src/tree_qp_common.h"
#include "treeqp/utils/types.h"
#include "treeqp/utils/profiling.h"
#include <blasfeo_target.h>
#include <blasfeo_common.h>

/* The reference's per-node plug-in interface for stage solvers (dual_Newton_tree.h:48-63), kept as a TYPE for source
 * compatibility.  This build has no per-node indirect calls: the entries' work is done for all nodes of a level by batched
 * kernels (solve_extended / solve / eval_dual_term -> the stage sweep; set_CmPnCmT / add_EPmE / add_CmPnCkT -> the dual-Hessian
 * kernels; init -> k_init / k_dense_init; export_mu -> the export kernel), selected per node by opts->qp_solver[]. */
typedef struct stage_qp_fcn_ptrs_ {
    answer_t (*is_applicable)(const tree_qp_in *qp_in, int idx);
    int (*calculate_size)(int nx, int nu, int nc);
    void (*assign_structs)(void **data, char **c_double_ptr);
    void (*assign_blasfeo_data)(int nx, int nu, void *data, char **c_double_ptr);
    void (*assign_data)(int nx, int nu, int nc, void *data, char **c_double_ptr);
    return_t (*init)(const tree_qp_in *qp_in, int idx, stage_qp_t solver_dad, void *work);
    return_t (*solve_extended)(const tree_qp_in *qp_in, int idx, void *work);
    return_t (*solve)(const tree_qp_in *qp_in, int idx, void *work);
    void (*set_CmPnCmT)(const tree_qp_in *qp_in, int idx, int idxdad, int offset, void *work_);
    void (*add_EPmE)(const tree_qp_in *qp_in, int idx, int idxdad, int offset, void *work_);
    void (*add_CmPnCkT)(const tree_qp_in *qp_in, int idx, int idxsib, int idxdad, int row_offset, int col_offset, void *work_);
    void (*eval_dual_term)(const tree_qp_in *qp_in, int idx, void *work_);
    void (*export_mu)(tree_qp_out *qp_out, int idx, void *work_);
} stage_qp_fcn_ptrs;

typedef struct treeqp_tdunes_opts_t_ {
    int maxIter;
    stage_qp_t *qp_solver;          /* per node */
    int checkLastActiveSet;         /* 0 / 1 (the reference's default): every block is rebuilt every iteration, which is bit-identical to the reference's skip logic;
                                     * 2: the persistent kernels keep the factors of workgroups whose active set did not change ("Active-set reuse") */
    double stationarityTolerance;
    termination_t termCondition;
    regType_t regType;
    double regTol;
    double regValue;
    int lineSearchMaxIter;
    double lineSearchGamma;
    double lineSearchBeta;
    int lineSearchRestartTrigger;
} treeqp_tdunes_opts_t;

struct tqgpu_solver;                /* opaque device mirror, see treeqp_amd.h */

typedef struct treeqp_tdunes_workspace_ {
    int Nn;
    int Np;
    int lsIter;                     /* line-search trials of the last Newton iteration */
    int lineSearchRestartCounter;
    int *npar;                      /* nodes per stage */
    int *idxpos;                    /* offset of node k inside its parent's dual block */

    /* host mirrors refreshed at the end of every solve (read by write_solution_to_txt and
     * by callers such as the reference's examples) */
    struct blasfeo_dvec *sx;            /* Nn */
    struct blasfeo_dvec *su;            /* Nn */
    struct blasfeo_dvec *slambda;       /* Np : block p = duals of p's children, concatenated */
    struct blasfeo_dvec *sDeltalambda;  /* Np */

    treeqp_profiling_t timings;

    /* --- MI355X extension fields --- */
    struct tqgpu_solver *device;    /* device mirror (released by treeqp_tdunes_destroy / atexit) */
    double *stage;                  /* pinned-size host staging slab for upload/download */
    int stage_doubles;
    int lsTotal;                    /* total line-search trials of the last solve */
    int maxIterAtCreate;
    int denseStageSolver;           /* 1: every node uses the dense unconstrained stage solver (TREEQP_QPOASES_SOLVER selector) */
} treeqp_tdunes_workspace;

int treeqp_tdunes_opts_calculate_size(int Nn);
void treeqp_tdunes_opts_create(int Nn, treeqp_tdunes_opts_t *opts, void *ptr);
void treeqp_tdunes_opts_set_default(int Nn, treeqp_tdunes_opts_t *opts);

int treeqp_tdunes_calculate_size(const tree_qp_in *qp_in, const treeqp_tdunes_opts_t *opts);
void treeqp_tdunes_create(const tree_qp_in *qp_in, const treeqp_tdunes_opts_t *opts, treeqp_tdunes_workspace *work, void *ptr);
void treeqp_tdunes_set_dual_initialization(const double *lambda, treeqp_tdunes_workspace *work);
return_t treeqp_tdunes_solve(const tree_qp_in *qp_in, tree_qp_out *qp_out, const treeqp_tdunes_opts_t *opts, treeqp_tdunes_workspace *work);

/* extension: release the device mirror explicitly (the reference API has no destroy call;
 * mirrors still alive at process exit are released by an atexit handler) */
void treeqp_tdunes_destroy(treeqp_tdunes_workspace *work);

void write_solution_to_txt(const tree_qp_in *qp_in, int Np, int iter, struct node *tree, treeqp_tdunes_workspace *work);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_SRC_DUAL_NEWTON_TREE_H_ */
