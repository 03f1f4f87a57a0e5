/*
 * treeqp_amd: tree-structured QP container (input, output, KKT check).
 *
 * Drop-in for the reference's treeqp/src/tree_qp_common.h:43-323: same struct/field names,
 * same function names, argument order and meaning, so that unmodified callers
 * (examples/spring_mass_dual_newton_tree.c, examples/thesis_example.c, ...) compile against
 * it.  The implementation (treeqp_amd/csrc/host/qp_container.c) is independent: all values
 * live in one flat column-major slab so that a solve can stage them to the MI355X with a
 * handful of contiguous copies.
 *
 * QP:   min  sum_k 1/2 [x;u]'[Q S';S R][x;u] + [q;r]'[x;u]
 *       s.t. x_k = A_{k-1} x_dad + B_{k-1} u_dad + b_{k-1},  bounds on x,u,  dmin <= Cx+Du <= dmax
 */
#ifndef TREEQP_SRC_TREE_OCP_QP_COMMON_H_
#define TREEQP_SRC_TREE_OCP_QP_COMMON_H_
#ifdef __cplusplus
extern "C" {
#endif
#include <blasfeo_target.h>
#include <blasfeo_common.h>
#include "treeqp/utils/types.h"

typedef struct treeqp_info_t_ {
    int Nn;
    int iter;                 /* Newton iterations (the converging check is not counted) */
    double total_time;        /* seconds */
    double solver_time;       /* Newton loop (device time on the MI355X path) */
    double interface_time;    /* staging in/out of the device + export */
} treeqp_info_t;

/* copies of the root-coupled data so that x0 can be changed after it was eliminated */
typedef struct qp_internal_t_ {
    int *is_A_initialized;
    int *is_b_initialized;
    int is_C_initialized;
    int is_dmin_initialized;
    int is_dmax_initialized;
    int is_S_initialized;
    int is_r_initialized;
    struct blasfeo_dvec x0;
    struct blasfeo_dmat *A0;
    struct blasfeo_dvec *b0;
    struct blasfeo_dmat C0;
    struct blasfeo_dvec dmax0;
    struct blasfeo_dvec dmin0;
    struct blasfeo_dmat S0;
    struct blasfeo_dvec r0;
} qp_internal_t;

typedef struct tree_qp_in_ {
    int N;                       /* number of nodes */
    int *nx, *nu, *nc;           /* per node */
    struct blasfeo_dmat *A, *B;  /* per edge (index = child node - 1) */
    struct blasfeo_dvec *b;
    struct blasfeo_dmat *Q, *R, *S;
    struct blasfeo_dvec *q, *r;
    struct blasfeo_dvec *xmin, *xmax, *umin, *umax;
    struct blasfeo_dmat *C, *D;
    struct blasfeo_dvec *dmin, *dmax;
    struct node *tree;
    qp_internal_t internal_memory;
} tree_qp_in;

typedef struct tree_qp_out_ {
    treeqp_info_t info;
    struct blasfeo_dvec *x, *u;  /* per node */
    struct blasfeo_dvec *lam;    /* per edge */
    struct blasfeo_dvec *mu_x, *mu_u, *mu_d;   /* + upper bound active, - lower */
} tree_qp_out;

/* ---- sizes ---- */
int total_number_of_states(const tree_qp_in *const qp_in);
int max_number_of_states(const tree_qp_in *const qp_in);
int total_number_of_controls(const tree_qp_in *const qp_in);
int max_number_of_controls(const tree_qp_in *const qp_in);
int total_number_of_general_constraints(const tree_qp_in *const qp_in);
int max_number_of_general_constraints(const tree_qp_in *const qp_in);
int total_number_of_primal_variables(const tree_qp_in *const qp_in);
int total_number_of_dynamic_constraints(const tree_qp_in *const qp_in);

/* ---- construction: calculate_size -> caller malloc -> create ---- */
int tree_qp_in_calculate_size(int Nn, const int *nx, const int *nu, const int *nc, const int *nk);
void tree_qp_in_create(int Nn, const int *nx, const int *nu, const int *nc, const int *nk, tree_qp_in *qp_in, void *ptr);
int tree_qp_out_calculate_size(const int Nn, const int *const nx, const int *const nu, const int *const nc);
void tree_qp_out_create(const int Nn, const int *const nx, const int *const nu, const int *const nc, tree_qp_out *const qp_out, void *ptr);

void tree_qp_in_eliminate_x0(tree_qp_in *const qp_in);
void tree_qp_out_eliminate_x0(tree_qp_out *const qp_out);

void tree_qp_out_calculate_KKT_res(const tree_qp_in *const qp_in, const tree_qp_out *const qp_out, double *res);
double tree_qp_out_max_KKT_res(const tree_qp_in *const qp_in, const tree_qp_out *const qp_out);

/* ---- accessors (generated) ----------------------------------------------------------------
 * edge index = child node - 1; matrices are column major; lda <= 0 means "tight". */
#define TQ_MAT_ACCESSORS(OBJ, KIND, NAME)                                                              \
    void tree_qp_in_set_##KIND##_##NAME##_colmajor(const double *const NAME, const int lda, OBJ *const qp_in, const int indx); \
    void tree_qp_in_get_##KIND##_##NAME##_colmajor(double *const NAME, const int lda, const OBJ *const qp_in, const int indx);
#define TQ_VEC_ACCESSORS(PFX, OBJ, KIND, NAME)                                                         \
    void PFX##_set_##KIND##_##NAME(const double *const NAME, OBJ *const qp, const int indx);           \
    void PFX##_get_##KIND##_##NAME(double *const NAME, const OBJ *const qp, const int indx);

TQ_MAT_ACCESSORS(tree_qp_in, edge, A)
TQ_MAT_ACCESSORS(tree_qp_in, edge, B)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, edge, b)
TQ_MAT_ACCESSORS(tree_qp_in, node, Q)
TQ_MAT_ACCESSORS(tree_qp_in, node, R)
TQ_MAT_ACCESSORS(tree_qp_in, node, S)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, q)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, r)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, xmin)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, xmax)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, umin)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, umax)
TQ_MAT_ACCESSORS(tree_qp_in, node, C)
TQ_MAT_ACCESSORS(tree_qp_in, node, D)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, dmin)
TQ_VEC_ACCESSORS(tree_qp_in, tree_qp_in, node, dmax)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, node, x)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, node, u)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, edge, lam)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, node, mu_x)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, node, mu_u)
TQ_VEC_ACCESSORS(tree_qp_out, tree_qp_out, node, mu_d)
#undef TQ_MAT_ACCESSORS
#undef TQ_VEC_ACCESSORS

/* grouped accessors */
void tree_qp_in_set_edge_dynamics_colmajor(const double *const A, const double *const B, const double *const b, tree_qp_in *const qp_in, const int indx);
void tree_qp_in_get_edge_dynamics_colmajor(double *const A, double *const B, double *const b, const tree_qp_in *const qp_in, const int indx);
void tree_qp_in_set_node_objective_colmajor(const double *const Q, const double *const R, const double *const S, const double *const q, const double *const r, tree_qp_in *const qp_in, const int indx);
void tree_qp_in_get_node_objective_colmajor(double *const Q, double *const R, double *const S, double *const q, double *const r, const tree_qp_in *const qp_in, const int indx);
void tree_qp_in_set_node_objective_diag(const double *const Qd, const double *const Rd, const double *const q, const double *const r, tree_qp_in *const qp_in, const int indx);
void tree_qp_in_set_node_bounds(const double *const xmin, const double *const xmax, const double *const umin, const double *const umax, tree_qp_in *const qp_in, const int indx);
void tree_qp_in_get_node_bounds(double *const xmin, double *const xmax, double *const umin, double *const umax, const tree_qp_in *const qp_in, const int indx);
void tree_qp_in_set_node_general_constraints(const double *const C, const double *const D, const double *const dmin, const double *const dmax, tree_qp_in *const qp_in, const int indx);
void tree_qp_in_get_node_general_constraints(double *const C, double *const D, double *const dmin, double *const dmax, const tree_qp_in *const qp_in, const int indx);

/* whole-tree setters: arguments are the per-edge / per-node pieces concatenated in index order */
void tree_qp_in_set_ltv_dynamics_colmajor(double *A, double *B, double *b, tree_qp_in *qp_in);
void tree_qp_in_set_ltv_objective_colmajor(double *Q, double *R, double *S, double *q, double *r, tree_qp_in *qp_in);
void tree_qp_in_set_ltv_objective_diag(double *Qd, double *Rd, double *q, double *r, tree_qp_in *qp_in);
void tree_qp_in_set_ltv_bounds(double *xmin, double *xmax, double *umin, double *umax, tree_qp_in *qp_in);
void tree_qp_in_set_const_bounds(double *xmin, double *xmax, double *umin, double *umax, tree_qp_in *qp_in);
void tree_qp_in_set_inf_bounds(tree_qp_in *qp_in);
void tree_qp_in_set_x0_strvec(tree_qp_in *qp_in, struct blasfeo_dvec *sx0);
void tree_qp_in_set_x0_colmaj(tree_qp_in *qp_in, double *x0);

/* LTI data replicated over the tree by realization id, diagonal weights scaled per stage */
void tree_qp_in_fill_lti_data_diag_weights(double *A, double *B, double *b,
    double *Q, double *q, double *P, double *p, double *R, double *r,
    double *xmin, double *xmax, double *umin, double *umax, double *x0,
    double *C, double *CN, double *D, double *dmin, double *dmax, tree_qp_in *qp_in);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_SRC_TREE_OCP_QP_COMMON_H_ */
