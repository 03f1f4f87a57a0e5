/* treeqp_amd BLASFEO-compat: target description (column-major reference layout, no panels). */
#ifndef TREEQP_AMD_BLASFEO_TARGET_H_
#define TREEQP_AMD_BLASFEO_TARGET_H_
#ifndef TARGET_GENERIC
#define TARGET_GENERIC
#endif
#ifndef LA_REFERENCE
#define LA_REFERENCE
#endif
#define TREEQP_AMD_BLASFEO_COMPAT 1
#endif
