/*
 * treeqp_amd: wall-clock timer of the treeQP C API (reference: treeqp/utils/timing.h:59-61).
 * The reference uses gettimeofday (1 us resolution) which is too coarse for GPU iterations of
 * a few tens of us; this build keeps the struct name and tic/toc API over CLOCK_MONOTONIC.
 */
#ifndef TREEQP_UTILS_TIMING_H_
#define TREEQP_UTILS_TIMING_H_
#ifdef __cplusplus
extern "C" {
#endif
#include <time.h>
#include "treeqp/utils/types.h"

typedef struct treeqp_timer_ {
    struct timespec tic;
    struct timespec toc;
} treeqp_timer;

void treeqp_tic(treeqp_timer *t);
double treeqp_toc(treeqp_timer *t);   /* seconds since the matching tic */

#ifdef __cplusplus
}
#endif
#endif  /* TREEQP_UTILS_TIMING_H_ */
