/* treeqp_amd BLASFEO-compat: untyped allocation helpers. */
#ifndef TREEQP_AMD_BLASFEO_V_AUX_EXT_DEP_H_
#define TREEQP_AMD_BLASFEO_V_AUX_EXT_DEP_H_
#ifdef __cplusplus
extern "C" {
#endif
void v_zeros(void **ptrA, int size);
void v_zeros_align(void **ptrA, int size);
void v_free(void *pA);
void v_free_align(void *pA);
#ifdef __cplusplus
}
#endif
#endif
