/*
 * treeqp_amd: regularisation options shared by the dual Newton solvers
 * (reference: treeqp/src/dual_Newton_common.h:41-60).  The solver's regularised Cholesky runs inside the
 * HIP factorisation kernels (p_factor_rows_first / p_refactor_rows, factor_body); the two host routines
 * below are the reference's utility entry points with the same semantics, for callers that link them.
 */
#ifndef DUAL_NEWTON_COMMON_H_
#define DUAL_NEWTON_COMMON_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"
#include <blasfeo_target.h>
#include <blasfeo_common.h>

/* NOTE: UNKNOWN and NO_REGULARIZATION share the value 0 in the reference as well. */
typedef enum {
    TREEQP_UNKNOWN_REGULARIZATION,
    TREEQP_NO_REGULARIZATION = 0,
    TREEQP_ALWAYS_LEVENBERG_MARQUARDT,
    TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT,
} regType_t;

typedef enum {
    TREEQP_NO_REGULARIZATION_ADDED = 0,
    TREEQP_REGULARIZATION_ADDED,
} reg_result_t;

/* Cholesky with the regularisation options of the dual Newton solvers (dual_Newton_common.c:36-123): NO: factorise; ALWAYS:
 * M += reg_val I, factorise; ON_THE_FLY: factorise, and if a diagonal entry of the factor is <= reg_tol: M += reg_val I,
 * factorise again.  M is modified when regularised (as in the reference).  _mn: M is m x n, m >= n. */
reg_result_t treeqp_dpotrf_l_with_reg_opts(struct blasfeo_dmat *M, struct blasfeo_dmat *CholM, regType_t reg_type, double reg_tol, double reg_val);
reg_result_t treeqp_dpotrf_l_mn_with_reg_opts(struct blasfeo_dmat *M, struct blasfeo_dmat *CholM, regType_t reg_type, double reg_tol, double reg_val);

#ifdef __cplusplus
}
#endif
#endif  /* DUAL_NEWTON_COMMON_H_ */
