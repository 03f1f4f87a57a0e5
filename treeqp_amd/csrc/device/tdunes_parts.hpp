/*
 * tdunes_parts.hpp -- how the device path is split over translation units, and the tables of instantiated shapes.
 *
 * tdunes_device.hip is compiled once per PART (treeqp_amd/build.py, all parts in parallel) with -DTQ_PARTS=<mask>: every part sees
 * every type and every kernel DECLARATION, but only the part that owns a kernel family compiles its bodies (and instantiates its
 * templates, explicitly, at the end of the family's header).  The host side of the C-ABI and the launch-per-phase kernels are
 * part TQP_HOST; it launches the other parts' kernels through their declarations (the host stub of a kernel is an ordinary
 * external symbol, registered by the part that defines it).  Without -DTQ_PARTS the file is one translation unit as before
 * (tools/vbuild.sh, single-file experiment builds).
 */
#pragma once

#define TQP_HOST    0x001   /* C-ABI host side + launch-per-phase kernels (k_*), k_pack_persist */
#define TQP_GP      0x002   /* g_persist, g_persist_batch (tdunes_gpersist.hpp) */
#define TQP_WIDE    0x004   /* k_hess_w, k_factor(_all)_w, k_forward(_all)_w (tdunes_wide.hpp) */
#define TQP_W3      0x008   /* k_sg, k_sgp_t, k_hf_w, k_fwd3(c) (tdunes_wide3.hpp) */
#define TQP_TIER    0x010   /* f_back, f_top, f_fwd, f_stage (tdunes_fast.hpp) */
#define TQP_PERSIST 0x020   /* f_persist, f_mpersist; sliced by shape: TQ_PERSIST_SLICE of TQ_PERSIST_NSLICES */
#define TQP_SHARD   0x040   /* f_persist_sh<.., 1>: compiled with -DTQ_LD_SCOPE=__HIP_MEMORY_SCOPE_SYSTEM (polls of a slab that peers write over xGMI) */
#define TQP_BATCH   0x080   /* f_persist_batch */
#define TQP_SHARD_AG 0x100  /* f_persist_sh<.., 0>: the same kernel with agent-scope polls (A/B on a node) */
#define TQP_ALL     0x1FF

#ifndef TQ_PARTS
#define TQ_PARTS TQP_ALL
#endif
#define TQ_HAS(p) (((TQ_PARTS) & (p)) != 0)

/* (NX, NU, MD) instantiations of the fused / persistent path: nx * md <= 16 and a multiple of 4, nx + nu <= 16 */
#ifdef TQ_SMALL_TABLE   /* experiment builds: only the BASELINE shapes, a fifth of the compile time */
#define FAST_TABLE(X) X(0, 8, 3, 2) X(2, 4, 1, 3)
#else
#define FAST_TABLE(X) X(0, 8, 3, 2) X(1, 4, 1, 2) X(2, 4, 1, 3) X(3, 2, 1, 2) X(4, 8, 2, 2) X(5, 6, 2, 2) X(6, 4, 1, 4) \
    X(7, 8, 1, 2) X(8, 8, 4, 2) X(9, 4, 2, 2) X(10, 4, 3, 2) X(11, 4, 2, 3) X(12, 4, 2, 4) X(13, 6, 1, 2) X(14, 6, 3, 2) X(15, 2, 1, 4) X(16, 2, 2, 2)
#endif

/* shapes with a sharded instantiation of the persistent kernel (f_persist_sh; one table line per shape) */
#ifdef TQ_SMALL_TABLE
#define SHARD_TABLE(X) X(0, 8, 3, 2)
#else
#define SHARD_TABLE(X) X(0, 8, 3, 2) X(1, 4, 1, 2) X(4, 8, 2, 2) X(5, 6, 2, 2)
#endif

/* (NX, NU, MD) instantiations of the multistage persistent kernel: the chain part works on blocks of NX rows,
 * which the MFMA Schur tile wants to be a multiple of 4 */
#ifdef TQ_SMALL_TABLE
#define MSTAGE_TABLE(X) X(0, 8, 3, 2) X(2, 4, 1, 3)
#else
#define MSTAGE_TABLE(X) X(0, 8, 3, 2) X(1, 4, 1, 2) X(2, 4, 1, 3) X(4, 8, 2, 2) X(6, 4, 1, 4) \
    X(7, 8, 1, 2) X(8, 8, 4, 2) X(9, 4, 2, 2) X(10, 4, 3, 2) X(11, 4, 2, 3) X(12, 4, 2, 4)
#endif

/* shapes with a batch kernel (f_persist_batch: one launch for a batch of trees of one shape); the BASELINE shapes and a few
 * neighbours -- any other (nx, nu, md) of FAST_TABLE / MSTAGE_TABLE is one line away and costs its compile time; without a line
 * the members of a batch are launched one by one */
#define BATCH_TABLE(X) X(0, 8, 3, 2, false) X(1, 4, 1, 3, true) X(2, 8, 3, 2, true) X(3, 4, 1, 2, false) X(4, 4, 1, 2, true) X(5, 8, 2, 2, false)

/* Slices of the persistent family: table index idx belongs to slice idx % TQ_PERSIST_NSLICES.  TQ_SL(idx, text) expands to `text`
 * in the part that owns the slice and to nothing elsewhere (explicit instantiations cannot be made conditional any other way). */
#ifndef TQ_PERSIST_NSLICES
#define TQ_PERSIST_NSLICES 1
#endif
#ifndef TQ_PERSIST_SLICE
#define TQ_PERSIST_SLICE 0
#endif
#define TQ_SL_YES(...) __VA_ARGS__
#define TQ_SL_NO(...)
#define TQ_SL_PICK(idx) ((idx) % TQ_PERSIST_NSLICES == TQ_PERSIST_SLICE)
#if TQ_SL_PICK(0)
#define TQ_SL_0 TQ_SL_YES
#else
#define TQ_SL_0 TQ_SL_NO
#endif
#if TQ_SL_PICK(1)
#define TQ_SL_1 TQ_SL_YES
#else
#define TQ_SL_1 TQ_SL_NO
#endif
#if TQ_SL_PICK(2)
#define TQ_SL_2 TQ_SL_YES
#else
#define TQ_SL_2 TQ_SL_NO
#endif
#if TQ_SL_PICK(3)
#define TQ_SL_3 TQ_SL_YES
#else
#define TQ_SL_3 TQ_SL_NO
#endif
#if TQ_SL_PICK(4)
#define TQ_SL_4 TQ_SL_YES
#else
#define TQ_SL_4 TQ_SL_NO
#endif
#if TQ_SL_PICK(5)
#define TQ_SL_5 TQ_SL_YES
#else
#define TQ_SL_5 TQ_SL_NO
#endif
#if TQ_SL_PICK(6)
#define TQ_SL_6 TQ_SL_YES
#else
#define TQ_SL_6 TQ_SL_NO
#endif
#if TQ_SL_PICK(7)
#define TQ_SL_7 TQ_SL_YES
#else
#define TQ_SL_7 TQ_SL_NO
#endif
#if TQ_SL_PICK(8)
#define TQ_SL_8 TQ_SL_YES
#else
#define TQ_SL_8 TQ_SL_NO
#endif
#if TQ_SL_PICK(9)
#define TQ_SL_9 TQ_SL_YES
#else
#define TQ_SL_9 TQ_SL_NO
#endif
#if TQ_SL_PICK(10)
#define TQ_SL_10 TQ_SL_YES
#else
#define TQ_SL_10 TQ_SL_NO
#endif
#if TQ_SL_PICK(11)
#define TQ_SL_11 TQ_SL_YES
#else
#define TQ_SL_11 TQ_SL_NO
#endif
#if TQ_SL_PICK(12)
#define TQ_SL_12 TQ_SL_YES
#else
#define TQ_SL_12 TQ_SL_NO
#endif
#if TQ_SL_PICK(13)
#define TQ_SL_13 TQ_SL_YES
#else
#define TQ_SL_13 TQ_SL_NO
#endif
#if TQ_SL_PICK(14)
#define TQ_SL_14 TQ_SL_YES
#else
#define TQ_SL_14 TQ_SL_NO
#endif
#if TQ_SL_PICK(15)
#define TQ_SL_15 TQ_SL_YES
#else
#define TQ_SL_15 TQ_SL_NO
#endif
#if TQ_SL_PICK(16)
#define TQ_SL_16 TQ_SL_YES
#else
#define TQ_SL_16 TQ_SL_NO
#endif
#define TQ_SL(idx, ...) TQ_SL_##idx(__VA_ARGS__)
