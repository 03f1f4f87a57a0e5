/*
 * Minimal BLASFEO-compatible containers for the treeqp_amd host layer (MI355X build).
 *
 * The reference (dkouzoup/treeQP) stores every matrix/vector in giaf/blasfeo's
 * `struct blasfeo_dmat` / `struct blasfeo_dvec` and callers touch `pA/pa`, `m`, `n`, `memsize`
 * directly (e.g. tree_qp_common.c:457-524, memory.c:108,127) or through the element macros
 * BLASFEO_DMATEL / BLASFEO_DVECEL.  BLASFEO itself is an un-vendored submodule of the reference;
 * this header is an independent, from-scratch definition of the same *names* over a plain
 * column-major, unpadded layout (leading dimension == m).  That layout is what the device
 * upload path wants: a dmat/dvec here is just a typed view into one flat slab.
 */
#ifndef TREEQP_AMD_BLASFEO_COMMON_H_
#define TREEQP_AMD_BLASFEO_COMMON_H_

#ifdef __cplusplus
extern "C" {
#endif

struct blasfeo_dmat {
    double *pA;     /* column-major data, leading dimension m */
    int m;          /* rows */
    int n;          /* columns */
    int memsize;    /* bytes reserved by blasfeo_create_dmat */
};

struct blasfeo_dvec {
    double *pa;
    int m;
    int memsize;
};

#define BLASFEO_DMATEL(sA, ai, aj) ((sA)->pA[(ai) + (size_t)(aj) * (sA)->m])
#define BLASFEO_DVECEL(sa, ai) ((sa)->pa[(ai)])

#ifdef __cplusplus
}
#endif
#endif
