/*
 * treeqp_amd: min-over-runs profiling record (reference: treeqp/utils/profiling.h:38-78).
 *
 * Difference by design: in the reference the *layout* of this struct (and therefore of
 * treeqp_tdunes_workspace) depends on the compile-time flag -DPROFILE=n.  Here the layout is
 * fixed (all fields always present) so that one shared library serves callers compiled with
 * any PROFILE value; the level only selects what timers_print shows and which device events
 * are recorded (see treeqp_amd_set_profiling_level in treeqp_amd.h).
 */
#ifndef TREEQP_UTILS_PROFILING_H_
#define TREEQP_UTILS_PROFILING_H_
#ifdef __cplusplus
extern "C" {
#endif
#include "treeqp/utils/types.h"
#include "treeqp/utils/timing.h"

typedef struct treeqp_profiling_t_ {
    int num_iter;
    int run_indx;
    /* level >= 1 */
    double total_time;
    double min_total_time;
    int total_ls_iter;
    /* level >= 2 : per Newton iteration */
    double *iter_times;
    double *min_iter_times;
    int *ls_iters;
    /* level >= 3 : per phase per iteration */
    double *stage_qps_times;
    double *min_stage_qps_times;
    double *build_dual_times;
    double *min_build_dual_times;
    double *newton_direction_times;
    double *min_newton_direction_times;
    double *line_search_times;
    double *min_line_search_times;
} treeqp_profiling_t;

int timers_calculate_size(int num_iter);
void timers_create(int num_iter, treeqp_profiling_t *timings, void *ptr);
void timers_initialize(treeqp_profiling_t *timings);
void timers_update(treeqp_profiling_t *timings);
void timers_print(treeqp_profiling_t *timings);

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_PROFILING_H_ */
