/*
 * treeqp_amd: regularisation options shared by the dual Newton solvers
 * (reference: treeqp/src/dual_Newton_common.h:41-52).  The reference's
 * treeqp_dpotrf_l_with_reg_opts (dual_Newton_common.c:36-78) has no host implementation here:
 * the regularised Cholesky runs inside the HIP factorisation kernels
 * (treeqp_amd/csrc/device/tdunes_kernels.hip, potrf_reg()).
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

#ifdef __cplusplus
}
#endif
#endif  /* DUAL_NEWTON_COMMON_H_ */
