/*
 * treeqp_amd: scalar types and enums of the treeQP C API.
 * Values are part of the drop-in contract (callers compare status == 0 etc.); they restate
 * the reference's treeqp/utils/types.h:37-85.
 */
#ifndef TREEQP_UTILS_TYPES_H_
#define TREEQP_UTILS_TYPES_H_
#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned int uint;

#define TREEQP_INF 1e12   /* bound value meaning "unbounded" */

typedef enum { YES, NO } answer_t;

typedef enum {
    TREEQP_SUMSQUAREDERRORS = 0,
    TREEQP_TWONORM,
    TREEQP_INFNORM,
} termination_t;

typedef enum {
    /* common solver exits */
    TREEQP_OPTIMAL_SOLUTION_FOUND,          /* 0 */
    TREEQP_MAXIMUM_ITERATIONS_REACHED,      /* 1 */
    /* solver specific */
    TREEQP_DN_NOT_DESCENT_DIRECTION,        /* 2 */
    TREEQP_DN_STAGE_QP_INIT_FAILED,
    TREEQP_DN_STAGE_QP_SOLVE_FAILED,
    TREEQP_IP_MIN_STEP,
    TREEQP_IP_UNKNOWN_FLAG,
    /* misc */
    TREEQP_OK,                              /* 7 */
    TREEQP_FAILURE,
    TREEQP_INVALID_OPTION,                  /* 9 */
    TREEQP_ERROR_OPENING_FILE,
    TREEQP_UNKNOWN_ERROR,
} return_t;

typedef enum {
    TREEQP_CLIPPING_SOLVER = 0,
    TREEQP_QPOASES_SOLVER,
} stage_qp_t;

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_TYPES_H_ */
