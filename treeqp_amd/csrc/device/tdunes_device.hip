/*
 * tdunes_device.hip -- MI355X (gfx950) device path of the tdunes hot path + its C-ABI.
 *
 * Replaces, as hand-written HIP kernels, the reference call sequences:
 *   k_stage   <- solve_stage_problems (dual_Newton_tree.c:218-330) + clipping solve_extended
 *                (dual_Newton_tree_clipping.c:188-227) fused with evaluate_dual_function
 *                (:823-918) + eval_dual_term (clipping.c:359-382) and the multiplier update of
 *                line_search (:964-968)
 *   k_grad    <- dual gradient loop of build_dual_problem (:519-539) + per-node norm partials
 *   k_check   <- calculate_error_in_residuals (:412-442) + termination test (:542-546)
 *   k_hess    <- dual Hessian loop (:551-615) with set_CmPnCmT / add_EPmE / add_CmPnCkT
 *                (clipping.c:264-355) as ONE C*P*C' product per parent block
 *   k_factor  <- backward sweep of calculate_delta_lambda (:668-752) with
 *                treeqp_dpotrf_l_with_reg_opts (dual_Newton_common.c:36-78): potrf, trsv_lnn and
 *                trsm_rltn done as one "tall" Cholesky of [W; resMod'; Ut], then dsyrk/dgemv Schur
 *                updates into the parent block
 *   k_forward <- forward sweep (:756-775) + partial of gradient_trans_times_direction (:808-820)
 *   k_ls_*    <- line_search control flow (:922-1019), decided on the device
 *
 * Design: one 64-lane wavefront owns one tree node (k_stage, k_grad) or one dual-Hessian block
 * (k_hess, k_factor, k_forward); the block lives in LDS while it is worked on; nodes/blocks of
 * one tree level form one grid.  All indices come from flat prefix-sum tables, so per-node
 * dimensions nx[k], nu[k], nk[k] are free (nx[0] = 0 after x0 elimination included).  The active
 * set bookkeeping of the reference (checkLastActiveSet) is not needed: every iteration rebuilds
 * every block, which yields bit-identical factors (DESIGN.md "checkLastActiveSet").
 *
 * Control flow lives in a device control block (struct Ctrl): every kernel first looks at it and
 * returns if its phase is not due, so the host may enqueue ahead without reading back.
 *
 * The kernels above are the launch-per-phase protocol (any tree).  Faster protocols for the trees that admit them are in the
 * headers included below: tdunes_fast.hpp / tdunes_persist.hpp (uniform and multistage trees: the whole solve as ONE launch, also
 * dealt over several devices: tqgpu_pshard_*), tdunes_wide.hpp / tdunes_wide3.hpp (dual blocks of 16 < d <= 64 rows: MFMA kernels,
 * three launches per Newton iteration), tdunes_gpersist.hpp (small trees of any shape: one workgroup per tree).
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "treeqp_amd.h"

#define WAVE 64
#include "tdunes_parts.hpp"

namespace tqd {

int fail(int code, const std::string &msg);      /* defined in the host part */



#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(TQGPU_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

/* ------------------------------------------------------------------------------------------ */
/* device-side tables                                                                         */
/* ------------------------------------------------------------------------------------------ */

struct Tree {            /* device pointers, passed by value */
    int Nn, Np, Nh, nx0;
    const int *dad, *nk, *kid0, *nx, *nu, *xoff, *uoff, *aoff, *boff, *pos, *bdim, *woff, *utoff;
    /* everything the workgroup-per-block kernels need to know about node k in ONE 128-byte record (a dependent
     * chain of table look-ups costs a global round trip, ~2.4 us, per hop):
     * [0] bdim [1] nx [2] nu [3] nk [4] kid0 [5] xoff [6] uoff [7] xoff[kid0] [8] woff [9] utoff [10] dad [11] pos
     * [12] bdim[dad] [13] woff[dad] [16 + 3u ..] nx, aoff, boff of child u < 4 */
    const int *desc;
};
#define DESC_INTS 32

struct Ctrl {
    int done;            /* all work finished (converged / max iterations / failure)        */
    int status;          /* return_t value                                                   */
    int iter;            /* completed Newton iterations                                      */
    int cur;             /* which lambda buffer holds the current point                      */
    int ls_pending;      /* line search wants another trial                                  */
    int ls_iter;         /* trial counter of the running line search                         */
    int ls_total;
    int ls_last;
    int restart_counter; /* work->lineSearchRestartCounter                                   */
    int n_reg;           /* blocks regularised (diagnostic)                                  */
    int pad0, pad1;
    double tau, tauPrev, fval0, fval, dot, err;
};

struct Data {            /* device pointers, passed by value */
    const double *A, *B, *b, *Qd, *Rd, *q, *r, *xmin, *xmax, *umin, *umax;
    double *Qinv, *Rinv;
    double *qmod, *rmod, *x, *u, *xUnc, *uUnc, *QinvCal, *RinvCal;
    /* unclipped solution of phase S of the iteration in progress: the first trial sweep of a line search saves it before it
     * overwrites xUnc / uUnc.  The reference's trial sweeps (stage_qp_clipping_solve, clipping.c:231-260) do not write xUnc,
     * so on a MAXIMUM_ITERATIONS exit export_mu (:386-399) pairs the x of the last trial with the xUnc of the last phase S. */
    double *xUncS, *uUncS;
    double *lam0, *lam1, *dlam, *res, *resMod;
    double *W, *CholW, *invd, *Ut, *CholUt;
    double *fval, *part_err, *part_dot;
    double *Sbuf, *ybuf;     /* fused path: Schur hand-off records across tiers; backward solution L^-1 resMod */
    unsigned long long *stamps;   /* diagnostic time stamps (written only when Opts.stamps != 0; never read by kernels) */
    Ctrl *ctrl;
    int *ls_log;
    int ls_log_cap;
    /* dense unconstrained stage solver (generic path only; dual_Newton_tree_qpoases.c restricted to no bounds):
     * per node the stage Hessian H = [Q S'; S R] ((nx+nu)^2, column major) and its inverse P = H^-1 */
    int strict;               /* TREEQP_AMD_STRICT_SUM=1: sums that feed decisions are taken in the reference's order (node by node, block by block), see strict_* below */
    int dense;                /* some nodes use the dense unconstrained stage solver */
    const int *kind;          /* [Nn] per node: 0 clipping, 1 dense unconstrained (read only when dense != 0) */
    const double *Hd;
    double *Pd;
    const int *poff;          /* [Nn+1] offsets of the (nx+nu)^2 blocks */
};

struct Opts {
    int maxIter, termCondition, regType, lsMaxIter, lsRestartTrigger, stamps;
    int reuse;               /* checkLastActiveSet: keep the factors of workgroups whose active set did not change (persistent path) */
    double tol, regTol, regValue, gamma, beta;
};

/* Tagged words: what crosses workgroups INSIDE a launch.  A double travels as two 64-bit relaxed agent-scope atomic stores,
 * each (tag << 32) | 32-bit half; the consumer polls the payload itself until every word carries the tag it expects (see
 * tdunes_persist.hpp).  Relaxed agent-scope accesses go to the memory side, so nothing depends on one XCD's L2 seeing another's. */
#ifndef TQ_WIDE_NAP
#define TQ_WIDE_NAP 1        /* s_sleep between two looks of a wait inside a fused sweep (x 64 cycles): 1 / 2 / 4 give 0.94 / 0.97 / 0.99 ms for one C5-class tree, no difference on C4; 16 and 32 are slower on both */
#endif
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
typedef unsigned long long u64;
__device__ __forceinline__ void st_tag(u64 *p, double v, unsigned tag) {
    const u64 t = (u64)tag << 32;
    __hip_atomic_store(p, t | (unsigned)__double2loint(v), RLX, AGENT);
    __hip_atomic_store(p + 1, t | (unsigned)__double2hiint(v), RLX, AGENT);
}
/* TQ_LD_SCOPE: the scope of the POLLS of hand-over words.  Agent scope everywhere but in the part that holds the sharded persistent kernel
 * (tdunes_parts.hpp, treeqp_amd/build.py): there the slab is written by peer devices over xGMI and the polls are system-scope loads. */
#ifndef TQ_LD_SCOPE
#define TQ_LD_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ double ld_tag(const u64 *p, unsigned tag, bool &ok) {
    const u64 a = __hip_atomic_load(p, RLX, TQ_LD_SCOPE), b = __hip_atomic_load(p + 1, RLX, TQ_LD_SCOPE);
    ok = ok && (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
    return __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
}
/* one tagged double, polled until it is there; gives up after 0.5 s (dead = true, value 0) */
__device__ __forceinline__ double wait_tag(const u64 *p, unsigned tag, bool &dead) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
        const double v = ld_tag(p, tag, ok);
        if (ok) return v;
        if (dead || wall_clock64() - t0 > 50000000ull) { dead = true; return 0.0; }      /* 100 MHz clock */
        __builtin_amdgcn_s_sleep(TQ_WIDE_NAP);
    }
}

/* Phase guards.  The host enqueues kernels ahead of the device's decisions and tags every launch
 * with the Newton iteration `h` (and line-search trial `t`) it belongs to; a kernel whose tag does
 * not match the device state is a no-op.  This keeps the host out of the loop: it only reads the
 * control block once per enqueued chunk. */
__device__ __forceinline__ bool phase_main(const Ctrl *c, int h) { return !c->done && c->iter == h && !c->ls_pending; }
__device__ __forceinline__ bool phase_trial(const Ctrl *c, int h, int t) { return !c->done && c->iter == h && c->ls_pending && c->ls_iter == t; }

/* Cross-lane sums without LDS round trips (all 64 lanes must be active):
 *   dpp_mov<row_ror:n>  rotate inside each row of 16 lanes (one VALU op per 32-bit half),
 *   v_permlane16/32_swap (gfx950) fold the rows.  Every lane ends with the full sum. */
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);
    return v;
}
__device__ __forceinline__ double row16_max(double v) {
    v = fmax(v, dpp_mov<0x128>(v)); v = fmax(v, dpp_mov<0x124>(v)); v = fmax(v, dpp_mov<0x122>(v)); v = fmax(v, dpp_mov<0x121>(v));
    return v;
}
/* value of lane l combined with lanes l ^ 16, l ^ 32, l ^ 48 */
template <bool IS_MAX = false>
__device__ __forceinline__ double rows_fold(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x0 = __hiloint2double((int)b[0], (int)a[0]), x1 = __hiloint2double((int)b[1], (int)a[1]);
    const double x = IS_MAX ? fmax(x0, x1) : x0 + x1;
    const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
    auto c = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
    const double y0 = __hiloint2double((int)d[0], (int)c[0]), y1 = __hiloint2double((int)d[1], (int)c[1]);
    return IS_MAX ? fmax(y0, y1) : y0 + y1;
}
__device__ __forceinline__ double wsum(double v) { return rows_fold<false>(row16_sum(v)); }
/* Maxima that feed the termination test must PROPAGATE a NaN (the reference's MAX(error, NaN) is NaN, `error < tol` is then
 * false and the line search ends the solve with NOT_DESCENT_DIRECTION, dual_Newton_tree.c:412-442, :949); fmax / v_max_f64
 * return the other operand.  nanmax: NaN if either operand is.  wmax: the fast v_max tree, then NaN if any lane's input was. */
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }
__device__ __forceinline__ double wmax(double v) {
    const double r = rows_fold<true>(row16_max(v));
    return __builtin_amdgcn_ballot_w64(v != v) ? __builtin_nan("") : r;
}

/* historical names, used by the generic kernels: same fixed-order reductions (a ds_bpermute butterfly cost
 * ~1.4 k cycles per call, five of them per node in the stage sweep) */
__device__ __forceinline__ double wave_sum(double v) { return wsum(v); }
__device__ __forceinline__ double wave_max(double v) { return wmax(v); }

/* The generic kernels work one wavefront per node / block on a private LDS window.  Their bodies are
 * device functions so that the same code runs (a) as one launch per phase and tree level, one wave per
 * workgroup, and (b) inside the single-workgroup persistent kernel g_persist, many waves side by side.
 * Inside a body the only synchronisation needed is among the lanes of ONE wave (they run in lockstep;
 * LDS operations of a wave complete in order), i.e. a compiler-level fence. */
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

/* ------------------------------------------------------------------------------------------ */
/* k_init: Qinv = 1/Qd, Rinv = 1/Rd  (stage_qp_clipping_init, clipping.c:163-170)             */
/* ------------------------------------------------------------------------------------------ */
#if TQ_HAS(TQP_HOST)
__global__ void k_init(int n_x, int n_u, Data D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_x) D.Qinv[i] = 1.0 / D.Qd[i];
    if (i < n_u) D.Rinv[i] = 1.0 / D.Rd[i];
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_dense_init: one wave per node: P = H^-1 through the Cholesky factor of H (stage_qp_qpoases */
/* restricted to unconstrained nodes, dual_Newton_tree_qpoases.c:153-217: the stage solution is */
/* z = H^-1 h and the elimination matrix is P = H^-1).  H in LDS, column by column; then lane j  */
/* solves L L' p_j = e_j.                                                                       */
/* ------------------------------------------------------------------------------------------ */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_dense_init(Tree T, Data D) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int k = blockIdx.x, lane = threadIdx.x;
    const int nz = T.nx[k] + T.nu[k];
    if (nz == 0 || !D.kind[k]) return;
    const double *H = D.Hd + D.poff[k];
    double *P = D.Pd + D.poff[k];
    double *Lm = lds;                      /* nz x nz, ld = nz */
    double *dinv = lds + (size_t)nz * nz;  /* nz */
    for (int e = lane; e < nz * nz; e += WAVE) Lm[e] = H[e];
    __syncthreads();
    for (int j = 0; j < nz; j++) {
        for (int i = j + lane; i < nz; i += WAVE) {
            double sacc = Lm[i + (size_t)j * nz];
            for (int c = 0; c < j; c++) sacc = fma(-Lm[i + (size_t)c * nz], Lm[j + (size_t)c * nz], sacc);
            Lm[i + (size_t)j * nz] = sacc;
        }
        __syncthreads();
        const double cjj = Lm[j + (size_t)j * nz];
        const double finv = cjj > 0.0 ? 1.0 / sqrt(cjj) : 0.0;
        for (int i = j + lane; i < nz; i += WAVE) Lm[i + (size_t)j * nz] *= finv;
        if (lane == 0) dinv[j] = finv;
        __syncthreads();
    }
    /* column j of P: forward then backward substitution on e_j (solution kept in global memory) */
    for (int j = lane; j < nz; j += WAVE) {
        double *pj = P + (size_t)j * nz;
        for (int i = 0; i < nz; i++) {
            double v = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < i; c++) v = fma(-Lm[i + (size_t)c * nz], pj[c], v);
            pj[i] = v * dinv[i];
        }
        for (int i = nz - 1; i >= 0; i--) {
            double v = pj[i];
            for (int c = i + 1; c < nz; c++) v = fma(-Lm[c + (size_t)i * nz], pj[c], v);
            pj[i] = v * dinv[i];
        }
    }
}
#endif

/* acc + sum_{i < n} a[i * sa] * b[i * sb], terms added in ascending order (the reference's order), but the loads go out DOT_BATCH
 * AT A TIME: a runtime-bounded `for (i) acc = fma(a[i], b[i], acc)` makes one memory round trip per trip of the loop (the
 * compiler does not move loads across the back edge), which is what the node sweeps of wider nodes spent their time on
 * (k_stage / k_grad at nx = 20, nu = 10: 15.6 / 13.3 us per launch).  Clamped addresses, masked use: nothing diverges. */
#ifndef DOT_BATCH
#define DOT_BATCH 8       /* loads in flight per lane and batch (16 and 24 measured slower inside the single-workgroup kernel: registers).
                             Also tried, none faster: 24 in flight in the standalone sweeps at nx = 20, the node's dimensions from its one
                             128-byte record instead of the index tables, the children's [A | B] fetched coalesced into LDS and read back
                             with typed LDS pointers, the staging loops of the block bodies batched the same way (slower).  k_stage /
                             k_grad at nx = 20 stream 16 MB in 12 us behind a ~4.5 us launch floor. */
#endif
__device__ __forceinline__ double dot_batched(const double *a, int sa, const double *b, int sb, int n, double acc, bool batch) {
    if (!batch) {                                            /* operands in LDS (g_persist with its state mirrored): the plain loop is the shorter program */
        for (int i = 0; i < n; i++) acc = fma(a[(size_t)i * sa], b[(size_t)i * sb], acc);
        return acc;
    }
    for (int i0 = 0; i0 < n; i0 += DOT_BATCH) {
        double va[DOT_BATCH], vb[DOT_BATCH];
#pragma unroll
        for (int m = 0; m < DOT_BATCH; m++) { const int i = i0 + m < n ? i0 + m : 0; va[m] = a[(size_t)i * sa]; vb[m] = b[(size_t)i * sb]; }
        asm volatile("" ::: "memory");                       /* the batch stays a batch (see LOADS_DONE in tdunes_wide.hpp) */
#pragma unroll
        for (int m = 0; m < DOT_BATCH; m++) { const double t = fma(va[m], vb[m], acc); acc = i0 + m < n ? t : acc; }
    }
    return acc;
}

/* ------------------------------------------------------------------------------------------ */
/* k_stage: one wave per node.  mode 0: evaluate at lam[cur] (first sweep of a solve);         */
/* mode 1: line-search trial, lam_next = lam_cur + (tau - tauPrev) * dlam, evaluate there.     */
/* Produces qmod,rmod,x,u,xUnc,uUnc,QinvCal,RinvCal and the node's dual-function term.         */
/* ------------------------------------------------------------------------------------------ */
/* xu != nullptr (k_sg): x, u of a PARENT node are also posted as tagged words -- entry j of x at xu[2 (xoff + j)], of u at
 * xu[2 (sum_nx + uoff + j)] -- for the children's gradient in the same launch */
/* Cl != nullptr (k_sgp): the children's [A | B], rows stacked child after child, column major with leading dimension ldcl, are in LDS */
__device__ void stage_body(const Tree &T, const Data &D, int mode, int k, int lane, double *lds, bool batch = true, u64 *xu = nullptr, int sum_nx = 0, unsigned xtag = 0u,
                           const double *Cl = nullptr, int ldcl = 0) {
    const Ctrl *c = D.ctrl;
    const int nxk = T.nx[k], nuk = T.nu[k], xo = T.xoff[k], uo = T.uoff[k];
    const int nkid = T.nk[k], d = T.bdim[k];
    const double *lamc = c->cur ? D.lam1 : D.lam0;
    double *lamn = c->cur ? D.lam0 : D.lam1;
    const double step = c->tau - c->tauPrev;
    const bool save_s = mode == 1 && c->ls_iter == 1;     /* first trial of a line search: xUnc / uUnc still hold phase S of this iteration */

    /* lambda of the children = dual block of node k, contiguous at xoff[kid0] */
    double *lk = lds;                   /* d doubles  */
    double *lown = lds + d;             /* nxk doubles */
    const int ko = nkid > 0 ? T.xoff[T.kid0[k]] : 0;
    for (int t = lane; t < d; t += WAVE) {
        double v = lamc[ko + t];
        if (mode == 1) v = fma(step, D.dlam[ko + t], v);
        lk[t] = v;
    }
    for (int t = lane; t < nxk; t += WAVE) {
        double v = 0.0;
        if (k > 0) {
            v = lamc[xo + t];
            if (mode == 1) { v = fma(step, D.dlam[xo + t], v); lamn[xo + t] = v; }
        }
        lown[t] = v;
    }
    WSYNC();

    if (D.dense && D.kind[k]) {
        /* dense unconstrained stage QP: z = P hmod; dual term -1/2 z'Hz + hmod'z - cmod */
        const int nz = nxk + nuk;
        double *hm = lds + d + nxk, *zz = hm + nz;
        for (int t = lane; t < nz; t += WAVE) {
            const bool isx = t < nxk;
            const int j = isx ? t : t - nxk;
            double v = isx ? fma(-1.0, D.q[xo + j], lown[j]) : -1.0 * D.r[uo + j];
            int rowoff = 0;
            for (int cc = 0; cc < nkid; cc++) {
                const int kid = T.kid0[k] + cc, nxc = T.nx[kid];
                const double *col = isx ? D.A + T.aoff[kid] + (size_t)j * nxc : D.B + T.boff[kid] + (size_t)j * nxc;
                double acc = 0.0;
                acc = dot_batched(col, 1, lk + rowoff, 1, nxc, acc, batch);
                v = fma(-1.0, acc, v);
                rowoff += nxc;
            }
            hm[t] = v;
            if (isx) D.qmod[xo + j] = v; else D.rmod[uo + j] = v;
        }
        WSYNC();
        const double *P = D.Pd + D.poff[k], *H = D.Hd + D.poff[k];
        for (int t = lane; t < nz; t += WAVE) {
            double acc = 0.0;
            for (int j = 0; j < nz; j++) acc = fma(P[t + (size_t)j * nz], hm[j], acc);
            zz[t] = acc;
            if (xu && nkid > 0) st_tag(xu + 2 * (size_t)(t < nxk ? xo + t : sum_nx + uo + t - nxk), acc, xtag);
            if (t < nxk) { if (save_s) D.xUncS[xo + t] = D.xUnc[xo + t]; D.x[xo + t] = acc; D.xUnc[xo + t] = acc; }
            else { if (save_s) D.uUncS[uo + t - nxk] = D.uUnc[uo + t - nxk]; D.u[uo + t - nxk] = acc; D.uUnc[uo + t - nxk] = acc; }
        }
        WSYNC();
        double p_quad = 0.0, p_lin = 0.0, p_cd = 0.0;
        for (int t = lane; t < nz; t += WAVE) {
            double acc = 0.0;
            for (int j = 0; j < nz; j++) acc = fma(H[t + (size_t)j * nz], zz[j], acc);
            p_quad = fma(zz[t], acc, p_quad);
            p_lin = fma(hm[t], zz[t], p_lin);
        }
        for (int t = lane; t < d; t += WAVE) p_cd = fma(D.b[ko + t], lk[t], p_cd);
        p_quad = wave_sum(p_quad); p_lin = wave_sum(p_lin); p_cd = wave_sum(p_cd);
        if (lane == 0) D.fval[k] = -0.5 * p_quad - p_cd + p_lin;
        return;
    }

    double p_qx = 0.0, p_hx = 0.0, p_ru = 0.0, p_hu = 0.0;   /* partial dots for the dual term */
    for (int t = lane; t < nxk + nuk; t += WAVE) {
        const bool isx = t < nxk;
        const int j = isx ? t : t - nxk;
        double v = isx ? fma(-1.0, D.q[xo + j], lown[j]) : -1.0 * D.r[uo + j];
        int rowoff = 0;
        for (int cc = 0; cc < nkid; cc++) {
            const int kid = T.kid0[k] + cc, nxc = T.nx[kid];
            const double *col = Cl ? Cl + rowoff + (size_t)t * ldcl : (isx ? D.A + T.aoff[kid] + (size_t)j * nxc : D.B + T.boff[kid] + (size_t)j * nxc);
            double acc = 0.0;
            acc = dot_batched(col, 1, lk + rowoff, 1, nxc, acc, batch && !Cl);
            v = fma(-1.0, acc, v);
            rowoff += nxc;
        }
        if (isx) {
            D.qmod[xo + j] = v;
            const double qi = D.Qinv[xo + j];
            const double unc = qi * v, lo = D.xmin[xo + j], hi = D.xmax[xo + j];
            double xv, cal;
            if (unc >= hi) { xv = hi; cal = 0.0; } else if (unc <= lo) { xv = lo; cal = 0.0; } else { xv = unc; cal = qi; }
            if (xu && nkid > 0) st_tag(xu + 2 * (size_t)(xo + j), xv, xtag);
            if (save_s) D.xUncS[xo + j] = D.xUnc[xo + j];
            D.xUnc[xo + j] = unc; D.x[xo + j] = xv; D.QinvCal[xo + j] = cal;
            p_qx = fma(D.Qd[xo + j] * xv, xv, p_qx);
            p_hx = fma(v, xv, p_hx);
        } else {
            D.rmod[uo + j] = v;
            const double ri = D.Rinv[uo + j];
            const double unc = ri * v, lo = D.umin[uo + j], hi = D.umax[uo + j];
            double uv, cal;
            if (unc >= hi) { uv = hi; cal = 0.0; } else if (unc <= lo) { uv = lo; cal = 0.0; } else { uv = unc; cal = ri; }
            if (xu && nkid > 0) st_tag(xu + 2 * (size_t)(sum_nx + uo + j), uv, xtag);
            if (save_s) D.uUncS[uo + j] = D.uUnc[uo + j];
            D.uUnc[uo + j] = unc; D.u[uo + j] = uv; D.RinvCal[uo + j] = cal;
            p_ru = fma(D.Rd[uo + j] * uv, uv, p_ru);
            p_hu = fma(v, uv, p_hu);
        }
    }
    /* cmod = sum_kids b_kid' lambda_kid */
    double p_c = 0.0;
    for (int t = lane; t < d; t += WAVE) p_c = fma(D.b[ko + t], lk[t], p_c);
    p_qx = wave_sum(p_qx); p_hx = wave_sum(p_hx); p_ru = wave_sum(p_ru); p_hu = wave_sum(p_hu); p_c = wave_sum(p_c);
    if (D.strict && nxk + nuk <= WAVE) {
        /* the node's term from sequential dot products, as eval_dual_term takes them (clipping.c:374-381): lane t holds entry t of
         * [x | u] (read back from where this wave has just put it), every lane takes the same sums */
        WSYNC();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        const int t = lane < nxk + nuk ? lane : 0;
        const bool isx = t < nxk;
        const int j = isx ? t : t - nxk;
        const double val = isx ? D.x[xo + j] : D.u[uo + j], hm = isx ? D.qmod[xo + j] : D.rmod[uo + j];
        const double wv = (isx ? D.Qd[xo + j] : D.Rd[uo + j]) * val;
        double a_qx = 0.0, a_hx = 0.0, a_ru = 0.0, a_hu = 0.0, cm = 0.0;
        for (int i = 0; i < nxk; i++) { const double vi = __shfl(val, i), wi = __shfl(wv, i), hi = __shfl(hm, i); a_qx = fma(wi, vi, a_qx); a_hx = fma(hi, vi, a_hx); }
        for (int i = nxk; i < nxk + nuk; i++) { const double vi = __shfl(val, i), wi = __shfl(wv, i), hi = __shfl(hm, i); a_ru = fma(wi, vi, a_ru); a_hu = fma(hi, vi, a_hu); }
        int off = 0;
        for (int cc = 0; cc < nkid; cc++) {                 /* cmod += ddot(b_kid, lambda_kid), child after child (dual_Newton_tree.c:892) */
            const int nxc = T.nx[T.kid0[k] + cc];
            double acc = 0.0;
            for (int i = 0; i < nxc; i++) acc = fma(D.b[ko + off + i], lk[off + i], acc);
            cm += acc;
            off += nxc;
        }
        p_qx = a_qx; p_hx = a_hx; p_ru = a_ru; p_hu = a_hu; p_c = cm;
    }
    if (lane == 0) {
        double f = -0.5 * p_qx - p_c;       /* clipping.c:375 */
        f += p_hx;                          /* :376 */
        f -= 0.5 * p_ru;                    /* :380 */
        f += p_hu;                          /* :381 */
        D.fval[k] = f;
    }
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_stage(Tree T, Data D, int mode, int h, int t) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (mode == 1 && !phase_trial(D.ctrl, h, t)) return;
    stage_body(T, D, mode, blockIdx.x, threadIdx.x, lds);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* block reductions (one workgroup of 256 threads, fixed pairwise order => deterministic)      */
/* ------------------------------------------------------------------------------------------ */
/* Reference-order sums (opt-in: TREEQP_AMD_STRICT_SUM=1, Data::strict).  The reference adds the nodes' dual-function terms one after the other
 * (evaluate_dual_function, dual_Newton_tree.c:915), takes res' dlam and the squared residual norm as one ddot per dual block, blocks in order
 * (gradient_trans_times_direction :808-820, calculate_error_in_residuals :412-442), and a node's own term from sequential dot products
 * (eval_dual_term, dual_Newton_tree_clipping.c:359-382).  The default kernels take all of these as fixed trees (deterministic, but another
 * order): near the optimum two dual values may then differ in their last bit and an Armijo or termination test goes the other way
 * (0.6 % of a random campaign).  With the switch on, one thread takes the sums in the reference's order -- slow, for parity work. */
__device__ double strict_sum(const double *v, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc += v[i];
    return acc;
}
__device__ double strict_block_dots(const Tree &T, const double *a, const double *b) {
    double ans = 0.0;
    for (int p = 0; p < T.Np; p++) {
        const int o = T.xoff[T.kid0[p]], d = T.bdim[p];
        double acc = 0.0;
        for (int i = 0; i < d; i++) acc = fma(a[o + i], b[o + i], acc);
        ans += acc;
    }
    return ans;
}
/* a value computed by thread 0 to every thread of the workgroup */
__device__ double bcast0(double v, double *sh) {
    __syncthreads();
    if (threadIdx.x == 0) sh[0] = v;
    __syncthreads();
    const double r = sh[0];
    __syncthreads();
    return r;
}

template <bool IS_MAX>
__device__ double block_reduce(const double *v, int n, double *sh) {
    /* strided per-thread partials, wave shuffle tree, then the (<= 16) wave results in order */
    double acc = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {          /* eight loads in flight per thread, same order of the sums */
        double t[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { const int i = i0 + m * (int)blockDim.x; t[m] = v[i < n ? i : 0]; }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < 8; m++) { const int i = i0 + m * (int)blockDim.x; if (i < n) acc = IS_MAX ? nanmax(acc, t[m]) : acc + t[m]; }
    }
    acc = IS_MAX ? wave_max(acc) : wave_sum(acc);
    __syncthreads();                                   /* sh may still be read from a previous call */
    if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    double r = 0.0;
    const int nw = (blockDim.x + WAVE - 1) / WAVE;
    for (int w = 0; w < nw; w++) r = IS_MAX ? nanmax(r, sh[w]) : r + sh[w];
    return r;
}

/* first sweep of a solve: fval0 = sum of the node terms */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(256) k_fval_init(Tree T, Data D) {
    __shared__ double sh[256];
    const double f = D.strict ? bcast0(threadIdx.x == 0 ? strict_sum(D.fval, T.Nn) : 0.0, sh) : block_reduce<false>(D.fval, T.Nn, sh);
    if (threadIdx.x == 0) { D.ctrl->fval0 = f; D.ctrl->fval = f; }
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* The small reductions as the TAIL of the sweep that produces their input (trees of <= FUSE_MAX nodes).  A reduction kernel of
 * its own (k_check, k_fval_init, k_ls_begin, k_ls_decide) is a launch of one workgroup: ~5 us for a few hundred additions, and a
 * single tree of a few hundred nodes is bound by its launch count.  Instead every workgroup of the sweep posts its partial as a
 * tagged word and counts itself off; the workgroup that finds it is the last takes the reduction -- in the order block_reduce
 * takes it with 256 threads, so the result is bit-identical to the separate kernel's -- and the decision.  (The partials
 * travel as tagged words and are polled, so nothing depends on the order in which plain stores of other XCDs become visible.) */
#define FUSE_MAX 512
struct Fuse { u64 *red; int *cnt; unsigned tag; int on; };

__device__ __forceinline__ bool fuse_last(const Fuse &F, int total, int lane) {
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(F.cnt, 1, RLX, AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == total - 1 && lane == 0) __hip_atomic_store(F.cnt, 0, RLX, AGENT);      /* the next launch counts from zero */
    return old == total - 1;
}
/* sum / maximum of entries first .. first + n - 1 (n <= FUSE_MAX) of the tagged partials, by ONE wave, as block_reduce<IS_MAX> with
 * 256 threads: lane l stands in for threads l, l + 64, l + 128, l + 192; `extra0` (with_extra) is entry 0 of the reduced vector,
 * known to this wave already (the root's part_dot), the tagged words then hold entries 1 .. */
template <bool IS_MAX>
__device__ double fuse_reduce(const u64 *red, int first, int n, unsigned tag, int lane, bool with_extra = false, double extra0 = 0.0) {
    double v[4][2];
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = w * WAVE + lane + 256 * j;                    /* index into the reduced vector */
                const bool in = i < n;
                bool okk = true;
                double val = 0.0;
                if (with_extra && i == 0) val = extra0;
                else { val = ld_tag(red + (size_t)(first + (in ? i - (with_extra ? 1 : 0) : 0)) * 2, tag, okk); ok = ok && (okk || !in); }
                v[w][j] = in ? val : 0.0;
            }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 20000000ull) return __builtin_nan("");      /* 0.2 s: cannot happen (every workgroup posted before it counted itself off); a NaN ends the solve */
        __builtin_amdgcn_s_sleep(2);
    }
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 2; j++) { const int i = w * WAVE + lane + 256 * j; if (i < n) acc = IS_MAX ? nanmax(acc, v[w][j]) : acc + v[w][j]; }
        acc = IS_MAX ? wave_max(acc) : wave_sum(acc);
        r = IS_MAX ? nanmax(r, acc) : r + acc;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* k_grad: one wave per node k >= 1:  res_k = b_k - x_k + A_k x_dad + B_k u_dad                */
/* ------------------------------------------------------------------------------------------ */
__device__ void grad_body(const Tree &T, const Data &D, int termCondition, int k, int lane, bool batch = true) {
    const int p = T.dad[k], nxk = T.nx[k], nxp = T.nx[p], nup = T.nu[p];
    const int xo = T.xoff[k], xp = T.xoff[p], up = T.uoff[p];
    const double *A = D.A + T.aoff[k], *B = D.B + T.boff[k];
    double part = 0.0;
    for (int i = lane; i < nxk; i += WAVE) {
        double rv = fma(-1.0, D.x[xo + i], D.b[xo + i]);
        double acc = 0.0;
        acc = dot_batched(A + i, nxk, D.x + xp, 1, nxp, acc, batch);
        rv += acc;
        acc = 0.0;
        acc = dot_batched(B + i, nxk, D.u + up, 1, nup, acc, batch);
        rv += acc;
        D.res[xo + i] = rv;
        D.resMod[xo + i] = rv;
        part = (termCondition == 2) ? nanmax(part, fabs(rv)) : fma(rv, rv, part);
    }
    part = (termCondition == 2) ? wave_max(part) : wave_sum(part);
    if (lane == 0) D.part_err[k] = part;
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_grad(Tree T, Data D, int termCondition, int h) {
    if (!phase_main(D.ctrl, h)) return;
    grad_body(T, D, termCondition, blockIdx.x + 1, threadIdx.x);
}
#endif

/* termination test; also the top-of-loop bookkeeping of the Newton iteration */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(256) k_check(Tree T, Data D, Opts O, int h) {
    __shared__ double sh[256];
    Ctrl *c = D.ctrl;
    if (!phase_main(c, h)) return;
    double err = (O.termCondition == 2) ? block_reduce<true>(D.part_err + 1, T.Nn - 1, sh)
                                        : (D.strict ? bcast0(threadIdx.x == 0 ? strict_block_dots(T, D.res, D.res) : 0.0, sh) : block_reduce<false>(D.part_err + 1, T.Nn - 1, sh));
    if (threadIdx.x == 0) {
        if (O.termCondition == 1) err = sqrt(err);
        c->err = err;
        if (err < O.tol) { c->done = 1; c->status = 0; }      /* TREEQP_OPTIMAL_SOLUTION_FOUND */
    }
}
#endif

/* k_grad with k_check as its tail (small trees) */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_grad_f(Tree T, Data D, Opts O, Fuse F, int h) {
    Ctrl *c = D.ctrl;
    if (!phase_main(c, h)) return;
    const int k = blockIdx.x + 1, lane = threadIdx.x;
    grad_body(T, D, O.termCondition, k, lane);
    if (lane == 0) st_tag(F.red + (size_t)(k - 1) * 2, D.part_err[k], F.tag);
    if (!fuse_last(F, T.Nn - 1, lane)) return;
    double err = (O.termCondition == 2) ? fuse_reduce<true>(F.red, 0, T.Nn - 1, F.tag, lane) : fuse_reduce<false>(F.red, 0, T.Nn - 1, F.tag, lane);
    if (lane == 0) {
        if (O.termCondition == 1) err = sqrt(err);
        c->err = err;
        if (err < O.tol) { c->done = 1; c->status = 0; }      /* TREEQP_OPTIMAL_SOLUTION_FOUND */
    }
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_hess: one wave per parent block p.                                                        */
/*   W_p = C P C' + blockdiag(QinvCal_kids),  C = [A_c B_c] stacked over the children,         */
/*   P = diag(QinvCal_p, RinvCal_p);   Ut_p = -(C[:, :nx_p] P)'                                */
/* ------------------------------------------------------------------------------------------ */
__device__ void hess_body(const Tree &T, const Data &D, int p, int lane, double *lds) {
    const int d = T.bdim[p], nxp = T.nx[p], nup = T.nu[p], nz = nxp + nup;
    double *Cs = lds;                 /* d x nz, column major, ld = d */
    double *CP = lds + (size_t)d * nz;
    const int k0 = T.kid0[p], ko = T.xoff[k0];
    const double *Qc = D.QinvCal + T.xoff[p], *Rc = D.RinvCal + T.uoff[p];
    const bool pdense = D.dense && D.kind[p];           /* the parent's elimination matrix: dense P_p = H_p^-1, or diag(QinvCal_p, RinvCal_p) */
    /* stage the children's [A B] rows */
    int rowoff = 0;
    for (int cc = 0; cc < T.nk[p]; cc++) {
        const int kid = k0 + cc, nxc = T.nx[kid];
        const double *A = D.A + T.aoff[kid], *B = D.B + T.boff[kid];
        for (int e = lane; e < nxc * nz; e += WAVE) {
            const int i = e % nxc, col = e / nxc;
            const double a = col < nxp ? A[i + (size_t)col * nxc] : B[i + (size_t)(col - nxp) * nxc];
            Cs[rowoff + i + (size_t)col * d] = a;
            if (!pdense) { const double pc = col < nxp ? Qc[col] : Rc[col - nxp]; CP[rowoff + i + (size_t)col * d] = a * pc; }
        }
        rowoff += nxc;
    }
    WSYNC();
    if (pdense) {
        /* CP = C P_p with the dense elimination matrix of the parent (build_M of the qpOASES stage solver) */
        const double *P = D.Pd + D.poff[p];
        for (int e = lane; e < d * nz; e += WAVE) {
            const int i = e % d, col = e / d;
            double acc = 0.0;
            for (int cidx = 0; cidx < nz; cidx++) acc = fma(Cs[i + (size_t)cidx * d], P[cidx + (size_t)col * nz], acc);
            CP[i + (size_t)col * d] = acc;
        }
        WSYNC();
    }
    double *W = D.W + T.woff[p];
    for (int e = lane; e < d * d; e += WAVE) {
        const int i = e % d, j = e / d;
        if (i < j) continue;
        double acc = 0.0, acc2 = 0.0;
        for (int cidx = 0; cidx < nxp; cidx++) acc = fma(Cs[i + (size_t)cidx * d], CP[j + (size_t)cidx * d], acc);
        for (int cidx = nxp; cidx < nz; cidx++) acc2 = fma(Cs[i + (size_t)cidx * d], CP[j + (size_t)cidx * d], acc2);
        double w = acc + acc2;
        if (!D.dense) { if (i == j) w += D.QinvCal[ko + i]; }
        else {
            /* add_EPmE: the state block of the CHILD's own elimination matrix on the diagonal block -- dense P_kid, or the
             * clipped inverse weights of a clipping child (the solver kinds of a node and of its children need not agree:
             * per-node opts->qp_solver[], dual_Newton_tree.c:124-162) */
            int ro = 0;
            for (int cc = 0; cc < T.nk[p]; cc++) {
                const int kid = k0 + cc, nxc = T.nx[kid];
                if (i >= ro && i < ro + nxc && j >= ro && j < ro + nxc) {
                    if (D.kind[kid]) { const int nzk = nxc + T.nu[kid]; w += D.Pd[D.poff[kid] + (i - ro) + (size_t)(j - ro) * nzk]; }
                    else if (i == j) w += D.QinvCal[ko + i];
                }
                ro += nxc;
            }
        }
        W[i + (size_t)j * d] = w;
    }
    if (p > 0) {
        double *Ut = D.Ut + T.utoff[p];
        for (int e = lane; e < nxp * d; e += WAVE) {
            const int i = e % nxp, rr = e / nxp;
            Ut[i + (size_t)rr * nxp] = -1.0 * CP[rr + (size_t)i * d];
        }
    }
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_hess(Tree T, Data D, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    hess_body(T, D, blockIdx.x, threadIdx.x, lds);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_factor: one wave per block of one tree level (blocks first .. first+count-1).             */
/* Tall Cholesky of T = [W ; resMod' ; Ut] (R = d + 1 + nx_ii rows, d columns), left-looking,   */
/* one lane per row:  rows 0..d-1 -> L,  row d -> (L^-1 resMod)',  rows d+1.. -> Ut L^-T.       */
/* Root block (ii == 0): no Ut rows; followed by the backward solve L^-T.                       */
/* ------------------------------------------------------------------------------------------ */
/* (Tried in round 2 and dropped: the tall matrix in registers, row per lane with readlane broadcasts as in the persistent kernels,
 * for R <= 64 and d <= 32: the factor phase of the single-workgroup kernel gained 6 %, the rest of that kernel lost more to the
 * code it added -- a block step there is bound by its global round trips, not by this loop; batching the k loop eight LDS reads
 * at a time: slower.) */
__device__ __forceinline__ void tall_potrf(double *Tm, double *invd, int R, int d, int ld, int lane) {
    for (int j = 0; j < d; j++) {
        /* every row i >= j : s_i = T[i,j] - sum_{k<j} T[i,k] T[j,k] */
        for (int i = j + lane; i < R; i += WAVE) {
            double s = Tm[i + (size_t)j * ld];
            for (int k = 0; k < j; k++) s = fma(-Tm[i + (size_t)k * ld], Tm[j + (size_t)k * ld], s);
            Tm[i + (size_t)j * ld] = s;
        }
        WSYNC();
        const double cjj = Tm[j + (size_t)j * ld];
        const double finv = cjj > 0.0 ? 1.0 / sqrt(cjj) : 0.0;      /* pivot <= 0 -> zero column */
        for (int i = j + lane; i < R; i += WAVE) Tm[i + (size_t)j * ld] *= finv;
        if (lane == 0) invd[j] = finv;
        WSYNC();
    }
}

/* sch != nullptr: the backward sweep of all levels is ONE launch (k_factor_all): the Schur complement of a block does not go
 * into the parent's W / resMod in global memory but travels as a record of tagged words -- entry (gi, gj), 1 <= gi <= nx,
 * 0 <= gj <= gi, of [v | S] at gi * (nx + 1) + gj of the block's record (rs doubles per node) -- and the parent subtracts the
 * records of its children from its tall matrix in LDS as they arrive (see factor_w_body in tdunes_wide.hpp). */
__device__ void factor_body(const Tree &T, const Data &D, const Opts &O, int ii, int lane, double *lds, u64 *sch = nullptr, int rs = 0, unsigned tag = 0u) {
    Ctrl *c = D.ctrl;
    const int d = T.bdim[ii], nxi = ii > 0 ? T.nx[ii] : 0;
    const int R = d + 1 + nxi, ld = R | 1;
    double *Tm = lds;                         /* ld x d */
    double *invd = lds + (size_t)ld * d;      /* d */
    const double *W = D.W + T.woff[ii];
    const int bo = T.xoff[T.kid0[ii]];        /* offset of the block vector */
    const double *Ut = D.Ut + T.utoff[ii];

    for (int pass = 0; pass < 2; pass++) {
        for (int e = lane; e < d * d; e += WAVE) {
            const int i = e % d, j = e / d;
            double w = W[i + (size_t)j * d];
            if (i == j && (O.regType == 1 || pass == 1)) w += O.regValue;   /* ddiare */
            Tm[i + (size_t)j * ld] = w;
        }
        for (int j = lane; j < d; j += WAVE) Tm[d + (size_t)j * ld] = D.resMod[bo + j];
        for (int e = lane; e < nxi * d; e += WAVE) {
            const int i = e % nxi, j = e / nxi;
            Tm[d + 1 + i + (size_t)j * ld] = Ut[i + (size_t)j * nxi];
        }
        if (sch) {
            WSYNC();
            const int k0 = T.kid0[ii];
            bool dead = false;
            for (int cc = 0, posc = 0; cc < T.nk[ii]; cc++) {
                const int kid = k0 + cc, nxc = T.nx[kid];
                if (kid < T.Np) {
                    const int w = nxc + 1;
                    const u64 *rec = sch + (size_t)kid * rs * 2;
                    for (int f = lane; f < w * w; f += WAVE) {
                        const int gi = f / w, gj = f - gi * w;
                        if (gi < 1 || gj > gi) continue;
                        const double val = wait_tag(rec + (size_t)f * 2, tag, dead);
                        double *dst = gj == 0 ? Tm + d + (size_t)(posc + gi - 1) * ld : Tm + (posc + gi - 1) + (size_t)(posc + gj - 1) * ld;
                        *dst -= val;
                    }
                }
                posc += nxc;
            }
            if (dead) { c->status = 3; __hip_atomic_store(&c->done, 1, RLX, AGENT); }      /* cannot happen: the children were started first */
        }
        WSYNC();
        tall_potrf(Tm, invd, R, d, ld, lane);
        if (O.regType != 2 || pass == 1) break;
        /* on-the-fly Levenberg-Marquardt: any diagonal entry <= regTol -> shift and refactorize */
        int small = 0;
        for (int j = lane; j < d; j += WAVE) small |= (Tm[j + (size_t)j * ld] <= O.regTol);
        if (!__any(small)) break;
        if (lane == 0) atomicAdd(&c->n_reg, 1);
        WSYNC();
    }

    /* outputs: factor, reciprocal diagonal */
    double *L = D.CholW + T.woff[ii];
    for (int e = lane; e < d * d; e += WAVE) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * d] = Tm[i + (size_t)j * ld];
    }
    for (int j = lane; j < d; j += WAVE) D.invd[bo + j] = invd[j];

    if (ii > 0) {
        for (int j = lane; j < d; j += WAVE) D.dlam[bo + j] = Tm[d + (size_t)j * ld];
        double *CUt = D.CholUt + T.utoff[ii];
        for (int e = lane; e < nxi * d; e += WAVE) {
            const int i = e % nxi, j = e / nxi;
            CUt[i + (size_t)j * nxi] = Tm[d + 1 + i + (size_t)j * ld];
        }
        /* Schur complement into the parent's diagonal sub-block and right-hand side */
        const int dd = T.dad[ii], pos = T.pos[ii], ddim = T.bdim[dd];
        double *Wd = D.W + T.woff[dd];
        for (int e = lane; e < nxi * nxi; e += WAVE) {
            const int i = e % nxi, j = e / nxi;
            if (i < j) continue;
            double acc = 0.0;
            for (int cidx = 0; cidx < d; cidx++) acc = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + 1 + j + (size_t)cidx * ld], acc);
            if (sch) st_tag(sch + ((size_t)ii * rs + (size_t)(i + 1) * (nxi + 1) + (j + 1)) * 2, acc, tag);
            else Wd[(pos + i) + (size_t)(pos + j) * ddim] -= acc;
        }
        const int xo = T.xoff[ii];
        for (int i = lane; i < nxi; i += WAVE) {
            double acc = 0.0;
            for (int cidx = 0; cidx < d; cidx++) acc = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + (size_t)cidx * ld], acc);
            if (sch) st_tag(sch + ((size_t)ii * rs + (size_t)(i + 1) * (nxi + 1)) * 2, acc, tag);
            else D.resMod[xo + i] -= acc;
        }
    } else {
        /* root: dlam_0 = L^-T (L^-1 resMod_0); column-oriented back substitution, k descending */
        double *z = Tm + d;                 /* row d, stride ld */
        double pd = 0.0;
        for (int k = d - 1; k >= 0; k--) {
            WSYNC();
            const double zk = z[(size_t)k * ld] * invd[k];
            for (int i = lane; i < k; i += WAVE) z[(size_t)i * ld] = fma(-Tm[k + (size_t)i * ld], zk, z[(size_t)i * ld]);
            WSYNC();
            if (lane == 0) z[(size_t)k * ld] = zk;
        }
        WSYNC();
        for (int j = lane; j < d; j += WAVE) {
            const double v = z[(size_t)j * ld];
            D.dlam[bo + j] = v;
            pd = fma(D.res[bo + j], v, pd);
        }
        pd = wave_sum(pd);
        if (lane == 0) D.part_dot[0] = pd;
    }
}

/* factor_body for NG small blocks of one level AT ONCE in one wave (the single-workgroup kernel, trees whose levels are wider than its 16
 * waves: the chains of a pruned scenario tree are 27 - 40 blocks of 8 x 8 on every level): the wave is split into NG groups of 64 / NG
 * lanes, group g works on block ii0 + g in its own LDS window with factor_body's loops (lane -> lane within the group, stride -> the
 * group's width).  A block step is a chain of LDS and memory round trips around a handful of multiply-adds -- ~8 us whatever the
 * block's size -- and three of them side by side take the time of one.  Per block the arithmetic is factor_body's operation for
 * operation (bit-identical results).  All blocks of a call have the same d and nx (the caller checks), d + 1 + nx <= 64 / NG. */
template <int NG, int DC = 0, int NXC = 0>      /* DC, NXC > 0: the blocks' dimensions, known at compile time (index arithmetic without divisions, loops unrolled) */
__device__ void factor_body_g(const Tree &T, const Data &D, const Opts &O, int ii0, int nblk, int lane, double *lds, int win) {
    constexpr int GW = WAVE / NG;
    Ctrl *c = D.ctrl;
    const int g = lane / GW, l = lane - g * GW;
    const bool act = g < NG && g < nblk;
    const int ii = ii0 + (act ? g : 0);
    const int d = DC > 0 ? DC : T.bdim[ii], nxi = NXC > 0 ? NXC : T.nx[ii];
    const int R = d + 1 + nxi, ld = R | 1;
    double *Tm = lds + (size_t)(g < NG ? g : 0) * win;      /* ld x d */
    double *invd = Tm + (size_t)ld * d;                     /* d */
    const double *W = D.W + T.woff[ii];
    const int bo = T.xoff[T.kid0[ii]];
    const double *Ut = D.Ut + T.utoff[ii];
    bool redo = true;                       /* pass 1 (on-the-fly regularisation) only touches the groups whose block needs it */
    for (int pass = 0; pass < 2; pass++) {
        if (redo) {
            for (int e = l; e < d * d; e += GW) {
                const int i = e % d, j = e / d;
                double w = W[i + (size_t)j * d];
                if (i == j && (O.regType == 1 || pass == 1)) w += O.regValue;   /* ddiare */
                Tm[i + (size_t)j * ld] = w;
            }
            for (int j = l; j < d; j += GW) Tm[d + (size_t)j * ld] = D.resMod[bo + j];
            for (int e = l; e < nxi * d; e += GW) {
                const int i = e % nxi, j = e / nxi;
                Tm[d + 1 + i + (size_t)j * ld] = Ut[i + (size_t)j * nxi];
            }
        }
        WSYNC();
        for (int j = 0; j < d; j++) {           /* tall_potrf, per group */
            if (redo) for (int i = j + l; i < R; i += GW) {
                double s = Tm[i + (size_t)j * ld];
                for (int k = 0; k < j; k++) s = fma(-Tm[i + (size_t)k * ld], Tm[j + (size_t)k * ld], s);
                Tm[i + (size_t)j * ld] = s;
            }
            WSYNC();
            const double cjj = Tm[j + (size_t)j * ld];
            const double finv = cjj > 0.0 ? 1.0 / sqrt(cjj) : 0.0;      /* pivot <= 0 -> zero column */
            if (redo) {
                for (int i = j + l; i < R; i += GW) Tm[i + (size_t)j * ld] *= finv;
                if (l == 0) invd[j] = finv;
            }
            WSYNC();
        }
        if (O.regType != 2 || pass == 1) break;
        /* on-the-fly Levenberg-Marquardt, block by block: any diagonal entry <= regTol -> shift and refactorise THAT block */
        int small = 0;
        for (int j = l; j < d; j += GW) small |= (Tm[j + (size_t)j * ld] <= O.regTol);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(small != 0);
        const unsigned long long gmask = (GW == 64 ? ~0ull : ((1ull << GW) - 1ull)) << (g < NG ? g * GW : 0);
        redo = g < NG && (bal & gmask) != 0ull;
        if (bal == 0ull) break;
        if (redo && act && l == 0) atomicAdd(&c->n_reg, 1);
        WSYNC();
    }
    if (!act) return;
    /* outputs: factor, reciprocal diagonal */
    double *L = D.CholW + T.woff[ii];
    for (int e = l; e < d * d; e += GW) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * d] = Tm[i + (size_t)j * ld];
    }
    for (int j = l; j < d; j += GW) D.invd[bo + j] = invd[j];
    for (int j = l; j < d; j += GW) D.dlam[bo + j] = Tm[d + (size_t)j * ld];
    double *CUt = D.CholUt + T.utoff[ii];
    for (int e = l; e < nxi * d; e += GW) {
        const int i = e % nxi, j = e / nxi;
        CUt[i + (size_t)j * nxi] = Tm[d + 1 + i + (size_t)j * ld];
    }
    /* Schur complement into the parent's diagonal sub-block and right-hand side.  (Fetching a lane's entries in one batch before updating them
     * -- the loop below is a memory round trip per trip -- measured SLOWER: 235 against 180 us per backward sweep of a 308-node tree; the
     * kernel has no registers to hold a batch in.) */
    const int dd = T.dad[ii], pos = T.pos[ii], ddim = T.bdim[dd];
    double *Wd = D.W + T.woff[dd];
    const int xo = T.xoff[ii];
    for (int e = l; e < nxi * nxi; e += GW) {
        const int i = e % nxi, j = e / nxi;
        if (i < j) continue;
        double acc = 0.0;
        for (int cidx = 0; cidx < d; cidx++) acc = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + 1 + j + (size_t)cidx * ld], acc);
        Wd[(pos + i) + (size_t)(pos + j) * ddim] -= acc;
    }
    for (int i = l; i < nxi; i += GW) {
        double acc = 0.0;
        for (int cidx = 0; cidx < d; cidx++) acc = fma(Tm[d + 1 + i + (size_t)cidx * ld], Tm[d + (size_t)cidx * ld], acc);
        D.resMod[xo + i] -= acc;
    }
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_factor(Tree T, Data D, Opts O, int first, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    factor_body(T, D, O, first + blockIdx.x, threadIdx.x, lds);
}
#endif
/* all levels in one launch: workgroup b takes block Np - 1 - b, so that the children a block waits for were started before it */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_factor_all(Tree T, Data D, Opts O, u64 *sch, int rs, unsigned tag, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    factor_body(T, D, O, T.Np - 1 - (int)blockIdx.x, threadIdx.x, lds, sch, rs, tag);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_forward: one wave per block of one level:                                                 */
/*   dlam_ii = L^-T ( y_ii - CholUt_ii' * dlam_dad[pos..] )                                    */
/* ------------------------------------------------------------------------------------------ */
/* fw != nullptr: the forward sweep of all levels is ONE launch (k_forward_all): the step of the parent block arrives as tagged
 * words (children of the root read D.dlam, which k_factor wrote in an earlier launch), and the block's own step is posted the
 * same way.  Everything that does not depend on the parent is fetched before the wait. */
__device__ void forward_body(const Tree &T, const Data &D, int ii, int lane, double *lds, u64 *fw = nullptr, unsigned tag = 0u) {
    const int d = T.bdim[ii], nxi = T.nx[ii], ld = d | 1;
    double *L = lds;                      /* ld x d */
    double *z = lds + (size_t)ld * d;     /* d */
    double *dl = z + d;                   /* nxi : dlam of node ii (inside the parent's block) */
    double *iv = dl + nxi;                /* d : reciprocal diagonal (from global memory it was one round trip per step of the chain below) */
    const int bo = T.xoff[T.kid0[ii]], xo = T.xoff[ii];
    const double *Lg = D.CholW + T.woff[ii];
    for (int e = lane; e < d * d; e += WAVE) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * ld] = Lg[i + (size_t)j * d];
    }
    for (int j = lane; j < d; j += WAVE) iv[j] = D.invd[bo + j];
    if (fw && T.dad[ii] != 0) {
        bool dead = false;
        for (int i = lane; i < nxi; i += WAVE) dl[i] = wait_tag(fw + (size_t)(xo + i) * 2, tag, dead);
        if (dead) { D.ctrl->status = 3; __hip_atomic_store(&D.ctrl->done, 1, RLX, AGENT); }      /* cannot happen: the parent was started first */
    } else {
        for (int i = lane; i < nxi; i += WAVE) dl[i] = D.dlam[xo + i];
    }
    WSYNC();
    const double *CUt = D.CholUt + T.utoff[ii];
    for (int j = lane; j < d; j += WAVE) {
        double acc = 0.0;
        for (int i = 0; i < nxi; i++) acc = fma(CUt[i + (size_t)j * nxi], dl[i], acc);
        z[j] = fma(-1.0, acc, D.dlam[bo + j]);
    }
    for (int k = d - 1; k >= 0; k--) {
        WSYNC();
        const double zk = z[k] * iv[k];
        for (int i = lane; i < k; i += WAVE) z[i] = fma(-L[k + (size_t)i * ld], zk, z[i]);
        WSYNC();
        if (lane == 0) z[k] = zk;
    }
    WSYNC();
    double pd = 0.0;
    for (int j = lane; j < d; j += WAVE) {
        if (fw) st_tag(fw + (size_t)(bo + j) * 2, z[j], tag);       /* first: my children wait for it */
        D.dlam[bo + j] = z[j];
        pd = fma(D.res[bo + j], z[j], pd);
    }
    pd = wave_sum(pd);
    if (lane == 0) D.part_dot[ii] = pd;
}

/* forward_body for NG blocks of one level side by side in one wave (see factor_body_g); the partial of res' dlam of every block is taken with the
 * lanes arranged as forward_body has them (entry j in lane j), so that it comes out bit-identical */
template <int NG>
__device__ void forward_body_g(const Tree &T, const Data &D, int ii0, int nblk, int lane, double *lds, int win) {
    constexpr int GW = WAVE / NG;
    const int g = lane / GW, l = lane - g * GW;
    const bool act = g < NG && g < nblk;
    const int ii = ii0 + (act ? g : 0);
    const int d = T.bdim[ii], nxi = T.nx[ii], ld = d | 1;
    double *L = lds + (size_t)(g < NG ? g : 0) * win;      /* ld x d */
    double *z = L + (size_t)ld * d;       /* d */
    double *dl = z + d;                   /* nxi */
    double *iv = dl + nxi;                /* d */
    const int bo = T.xoff[T.kid0[ii]], xo = T.xoff[ii];
    const double *Lg = D.CholW + T.woff[ii];
    for (int e = l; e < d * d; e += GW) {
        const int i = e % d, j = e / d;
        if (i >= j) L[i + (size_t)j * ld] = Lg[i + (size_t)j * d];
    }
    for (int j = l; j < d; j += GW) iv[j] = D.invd[bo + j];
    for (int i = l; i < nxi; i += GW) dl[i] = D.dlam[xo + i];
    WSYNC();
    const double *CUt = D.CholUt + T.utoff[ii];
    for (int j = l; j < d; j += GW) {
        double acc = 0.0;
        for (int i = 0; i < nxi; i++) acc = fma(CUt[i + (size_t)j * nxi], dl[i], acc);
        z[j] = fma(-1.0, acc, D.dlam[bo + j]);
    }
    for (int k = d - 1; k >= 0; k--) {
        WSYNC();
        const double zk = z[k] * iv[k];
        for (int i = l; i < k; i += GW) z[i] = fma(-L[k + (size_t)i * ld], zk, z[i]);
        WSYNC();
        if (l == 0) z[k] = zk;
    }
    WSYNC();
    double term = 0.0;
    for (int j = l; j < d; j += GW) {          /* (d <= GW: one trip) */
        if (act) D.dlam[bo + j] = z[j];
        term = fma(D.res[bo + j], z[j], term);
    }
#pragma unroll
    for (int gg = 0; gg < NG; gg++) {
        double v = __shfl(term, gg * GW + (lane < d ? lane : 0));
        v = lane < d ? v : 0.0;
        const double pd = wave_sum(v);
        if (lane == 0 && gg < nblk) D.part_dot[ii0 + gg] = pd;
    }
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_forward(Tree T, Data D, int first, int h) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    forward_body(T, D, first + blockIdx.x, threadIdx.x, lds);
}
#endif
/* all levels below the root in one launch, blocks in BFS order: a block's parent was started before it */
/* the tail of a fused forward sweep (small trees): the direction test and the start of the line search (k_ls_begin) */
__device__ __forceinline__ void fuse_ls_begin(const Tree &T, const Data &D, const Fuse &F, int ii, int lane) {
    if (lane == 0) st_tag(F.red + (size_t)(ii - 1) * 2, D.part_dot[ii], F.tag);
    if (!fuse_last(F, T.Np - 1, lane)) return;
    const double s = fuse_reduce<false>(F.red, 0, T.Np, F.tag, lane, true, D.part_dot[0]);      /* entry 0: the root's, from k_factor */
    if (lane == 0) {
        Ctrl *c = D.ctrl;
        const double dotp = -s;                                     /* :819 */
        c->dot = dotp;
        if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) { c->done = 1; c->status = 2; }      /* :951, NaN included */
        else { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
    }
}
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_forward_all(Tree T, Data D, u64 *fw, unsigned tag, int h, Fuse F) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (!phase_main(D.ctrl, h)) return;
    forward_body(T, D, 1 + (int)blockIdx.x, threadIdx.x, lds, fw, tag);
    if (F.on) fuse_ls_begin(T, D, F, 1 + (int)blockIdx.x, threadIdx.x);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* line-search control (line_search, dual_Newton_tree.c:922-1019)                              */
/* ------------------------------------------------------------------------------------------ */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(256) k_ls_begin(Tree T, Data D, int h) {
    __shared__ double sh[256];
    Ctrl *c = D.ctrl;
    if (!phase_main(c, h)) return;
    const double s = D.strict ? bcast0(threadIdx.x == 0 ? strict_block_dots(T, D.res, D.dlam) : 0.0, sh) : block_reduce<false>(D.part_dot, T.Np, sh);
    if (threadIdx.x == 0) {
        const double dotp = -s;                                     /* :819 */
        c->dot = dotp;
        if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) {  /* :951, NaN included */
            c->done = 1; c->status = 2;                             /* TREEQP_DN_NOT_DESCENT_DIRECTION */
        } else {
            c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
        }
    }
}
#endif

/* Armijo test and iteration bookkeeping for the trial whose dual value is f (line_search :973-1000) */
__device__ void ls_decide_tail(Ctrl *c, int *ls_log, int ls_log_cap, const Opts &O, double f) {
    c->cur ^= 1;                       /* the trial point is now the current point */
    c->fval = f;
    int finished = 0, lsIter = c->ls_iter;
    if (c->restart_counter == O.lsRestartTrigger) finished = 1;                        /* :973 */
    else if (f <= c->fval0 + O.gamma * c->tau * c->dot) finished = 1;                  /* :982 */
    else {
        c->tauPrev = c->tau;
        c->tau = O.beta * c->tauPrev;
        if (lsIter + 1 > O.lsMaxIter) { finished = 1; lsIter = lsIter + 1; }           /* loop exhausted */
        else c->ls_iter = lsIter + 1;
    }
    if (finished) {
        if (lsIter >= O.lsMaxIter) c->restart_counter++; else c->restart_counter = 0;  /* :993-1000 */
        c->ls_pending = 0;
        c->ls_last = lsIter;
        c->ls_total += lsIter;
        if (c->iter < ls_log_cap) ls_log[c->iter] = lsIter;
        c->iter += 1;
        c->fval0 = f;                  /* same point, same sweep => identical to a re-evaluation */
        if (c->iter >= O.maxIter) { c->done = 1; c->status = 1; }                      /* MAXIMUM_ITERATIONS */
    }
}

__device__ void ls_decide_tail(Ctrl *c, const Data &D, const Opts &O, double f) { ls_decide_tail(c, D.ls_log, D.ls_log_cap, O, f); }

/* direction test (:944-954); returns true when the solve must stop with NOT_DESCENT_DIRECTION */
__device__ bool ls_not_descent(Ctrl *c, double dotp) {
    c->dot = dotp;
    const bool bad = (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10)));
    if (bad) { c->done = 1; c->status = 2; c->ls_pending = 0; }
    return bad;
}

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(256) k_ls_decide(Tree T, Data D, Opts O, int h, int t, int with_descent_check) {
    __shared__ double sh[256];
    __shared__ int bail;
    Ctrl *c = D.ctrl;
    if (!phase_trial(c, h, t)) return;
    if (with_descent_check) {
        /* fused path: the first trial was evaluated speculatively; test the direction now */
        const double s = D.strict ? bcast0(threadIdx.x == 0 ? strict_block_dots(T, D.res, D.dlam) : 0.0, sh) : block_reduce<false>(D.part_dot, T.Np, sh);
        if (threadIdx.x == 0) bail = ls_not_descent(c, -s);
        __syncthreads();
        if (bail) return;
    }
    const double f = D.strict ? bcast0(threadIdx.x == 0 ? strict_sum(D.fval, T.Nn) : 0.0, sh) : block_reduce<false>(D.fval, T.Nn, sh);
    if (threadIdx.x == 0) ls_decide_tail(c, D, O, f);
}
#endif

/* k_stage with k_fval_init (mode 0) or k_ls_decide (mode 1) as its tail (small trees) */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_stage_f(Tree T, Data D, Opts O, Fuse F, int mode, int h, int t) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    Ctrl *c = D.ctrl;
    if (mode == 1 && !phase_trial(c, h, t)) return;
    const int k = blockIdx.x, lane = threadIdx.x;
    stage_body(T, D, mode, k, lane, lds);
    if (lane == 0) st_tag(F.red + (size_t)k * 2, D.fval[k], F.tag);
    if (!fuse_last(F, T.Nn, lane)) return;
    const double f = fuse_reduce<false>(F.red, 0, T.Nn, F.tag, lane);
    if (lane == 0) {
        if (mode == 0) { c->fval0 = f; c->fval = f; }
        else ls_decide_tail(c, D, O, f);
    }
}
#endif

/* ---- sharded mode (one tree over several devices): rank-local partials and the decision from the
 * gathered per-rank records; sums run in rank order so that every rank takes the same decision ---- */
#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(256) k_shard_pack1(Data D, int nlocal, double *xerr, int rank, int termCondition, int h) {
    __shared__ double sh[256];
    if (!phase_main(D.ctrl, h)) return;
    const double e = (termCondition == 2) ? block_reduce<true>(D.part_err, nlocal, sh) : block_reduce<false>(D.part_err, nlocal, sh);
    if (threadIdx.x == 0) xerr[rank] = e;
}
#endif

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_shard_pack2(Data D, const int *nodes, int n_nodes, const int *blocks, int n_blocks,
                                                     double *xs, int rank, int b0, int bn, int own0, int ownn, int h, int t) {
    const Ctrl *c = D.ctrl;
    if (!phase_trial(c, h, t)) return;
    /* The duals of the boundary nodes live in REPLICATED blocks but each boundary node is staged by its
     * owner only: every rank advances the slices of the boundary nodes it does not own itself (same
     * fma on the same replicated lam / dlam, hence identical values on every rank). */
    {
        const double *lamc = c->cur ? D.lam1 : D.lam0;
        double *lamn = c->cur ? D.lam0 : D.lam1;
        const double step = c->tau - c->tauPrev;
        for (int e = threadIdx.x; e < bn; e += WAVE)
            if (e < own0 || e >= own0 + ownn) lamn[b0 + e] = fma(step, D.dlam[b0 + e], lamc[b0 + e]);
    }
    /* fixed order: lane-strided partials then the shuffle tree */
    double f = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < n_nodes; i += WAVE) f += D.fval[nodes[i]];
    for (int i = threadIdx.x; i < n_blocks; i += WAVE) d += D.part_dot[blocks[i]];
    f = wave_sum(f); d = wave_sum(d);
    if (threadIdx.x == 0) { xs[2 * rank] = f; xs[2 * rank + 1] = d; }
}
#endif

#if TQ_HAS(TQP_HOST)
__global__ void __launch_bounds__(WAVE) k_ls_decide_parts(Data D, Opts O, const double *xs, int nranks, int h, int t, int with_descent_check) {
    Ctrl *c = D.ctrl;
    if (!phase_trial(c, h, t)) return;
    if (threadIdx.x != 0) return;
    double f = 0.0, d = 0.0;
    for (int r = 0; r < nranks; r++) { f += xs[2 * r]; d += xs[2 * r + 1]; }
    if (with_descent_check && ls_not_descent(c, -d)) return;
    ls_decide_tail(c, D, O, f);
}
#endif

#include "tdunes_fast.hpp"
#include "tdunes_persist.hpp"
#include "tdunes_wide.hpp"
#include "tdunes_wide3.hpp"
#include "tdunes_gpersist.hpp"

}  // namespace tqd
using namespace tqd;

#if TQ_HAS(TQP_HOST)
namespace tqd {
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace tqd

/* ============================================================================================ */
/* host side of the C-ABI                                                                       */
/* ============================================================================================ */

struct tqgpu_solver {
    int device = 0;
    int Nn = 0, Np = 0, Nh = 0;
    std::vector<int> nk, nx, nu, dad, stage, kid0, xoff, uoff, aoff, boff, pos, bdim, woff, utoff, lvl_first;
    int sum_nx = 0, sum_nu = 0, sum_lam = 0, sum_A = 0, sum_B = 0, sum_W = 0, sum_Ut = 0, nx0 = 0;
    size_t lds_stage = 0, lds_hess = 0, lds_factor = 0, lds_forward = 0, lds_dense = 0;
    bool wide = false;              /* larger blocks (16 < d <= 64): workgroup-per-block MFMA kernels (tdunes_wide.hpp) */
    size_t lds_hess_w = 0, lds_factor_w = 0, lds_forward_w = 0;
    unsigned long long *fw_words = nullptr;   /* launch-per-phase path: the steps of a forward sweep as tagged words (k_forward_all_w), [sum_nx][2] */
    unsigned fw_epoch = 0;              /* tag of the last fused forward launch */
    bool fw_fused = true;               /* TREEQP_AMD_FWD=levels: one launch per level instead */
    unsigned long long *sch_words = nullptr;  /* launch-per-phase path: Schur records of a fused backward sweep as tagged words (k_factor_all_w), [Nn][sch_rs][2] */
    int sch_rs = 0;                     /* doubles per record: (max nx + 1)^2 */
    unsigned bw_epoch = 0;
    bool bw_fused = true;               /* TREEQP_AMD_BWD=levels: one launch per level instead */
    unsigned long long *fuse_red = nullptr;   /* small trees: partials of the reductions that run as the tail of a sweep (Fuse), [Nn][2] tagged words */
    int *fuse_cnt = nullptr;            /* ... and the counter of the workgroups that have posted theirs */
    unsigned fuse_epoch = 0;
    bool fuse_ok = false, fuse_now = false;       /* fuse_now: this solve uses them (not while phases are timed one by one) */
    /* three launches per Newton iteration for the wide-block class (tdunes_wide3.hpp): k_sg, k_hf_w, k_fwd3 */
    bool w3_ok = false, w3_now = false;
    unsigned long long *w3_xu = nullptr, *w3_red = nullptr;
    int *w3_cnt = nullptr;
    unsigned w3_epoch = 0;
    size_t lds_hf_w = 0, lds_sgp = 0;
    int sgp_accs = 0;
    bool w3_sgp = false;                /* k_sgp (a workgroup per parent) instead of k_sg (a wave per node) */
    unsigned long long *d_pdw = nullptr; /* k_sgp mode 2: the blocks' parts of res' dlam as tagged words */
    bool w3_merge = false;              /* forward sweep and first trial in one launch (k_sgp mode 2) */
    int *d_anc = nullptr;               /* k_fwd3c: the path to the root of every block (tdunes_wide3.hpp); nullptr: the tree does not qualify */
    bool w3_mirror = false;             /* this solve: the launches of k_sg / k_sgp post the control block to h_res (w3_mirror in tdunes_wide3.hpp) */
    bool w3_tail_sg = false;            /* the last launch enqueued is one of them: its tag (w3_wait) is what the host polls for */
    unsigned w3_wait = 0;
    bool w3_post_next = false;          /* the next launch of k_sg / k_sgp is the last of what the host enqueues before it reads the verdict: it posts */
    bool w3_seen = false;               /* the last read of the control block came through the result block */
    std::vector<int> ls_pred;           /* trials per iteration of the previous solve: that many trial launches are enqueued behind an iteration's forward sweep (a trial beyond the accepted one is a no-op; a read-back per extra trial is 20 us) */
    bool dense = false, need_dense_init = false;   /* dense unconstrained stage solver selected (generic path only) */
    double *d_Hd = nullptr;      /* writable alias of Data.Hd */
    int *d_kind = nullptr;       /* writable alias of Data.kind */
    std::vector<int> poff;
    int use_fast_orig = 1;
    void *slab = nullptr;
    size_t slab_bytes = 0;
    Tree T{};
    Data D{};
    double *d_mu_x = nullptr, *d_mu_u = nullptr;
    double *d_lam_init = nullptr;   /* starting point of every solve (tqgpu_set_lambda) */
    unsigned launch_no = 0;         /* persistent launches so far (16 bits, never 0): tags of the hand-over words */
    void *pconst_slab = nullptr;    /* packed constants of the persistent path + its PDump */
    int *wg_map = nullptr;          /* blockIdx.x -> workgroup id (XCD-aware placement) */
    int co_capacity = 1;            /* workgroups of persistent launches that can be resident on the device together */
    int n_cu = 0;
    bool gpersist_ok = false;       /* small tree of any shape: whole solve in one launch of one workgroup (tdunes_gpersist.hpp) */
    int use_gpersist = 1;
    int *d_lvl_first = nullptr;
    size_t lds_gp_wave = 0;         /* doubles of LDS per wave of g_persist */
    size_t lds_gp_total = 0;        /* bytes of dynamic LDS of g_persist (windows + state mirror) */
    bool persist_one = false;          /* the persistent launch of this tree takes the one-workgroup-per-CU build */
    bool in_batch = false;             /* inside tqgpu_solve_batch: members launched one by one run side by side, two workgroups to a CU -- not with that build */
    bool gp_in_lds = false, gp_const_in_lds = false, gp_tab_in_lds = false, gp_small16 = false, gp_small8 = false;
    double *pab = nullptr, *pcst = nullptr;
    bool need_pack = true;          /* QP data changed since the constants were packed */
    /* writable aliases of the const inputs */
    double *A = nullptr, *B = nullptr, *b = nullptr, *Qd = nullptr, *Rd = nullptr, *q = nullptr, *r = nullptr;
    double *xmin = nullptr, *xmax = nullptr, *umin = nullptr, *umax = nullptr;
    HostRes *h_res = nullptr;    /* pinned; the persistent kernel writes it directly */
    Ctrl *h_ctrl = nullptr;      /* = &h_res->c */
    std::vector<hipEvent_t> ring_ev0, ring_ev1;   /* events of the last solves (device times on request) */
    long solve_no = 0;
    std::vector<char> ring_ok;                    /* the event pair of that ring slot was recorded by the solve that owns it */
    bool ev_timing = true;                        /* record a HIP event pair around every solve (tqgpu_set_event_timing) */
    int *h_ls_log = nullptr;     /* pinned */
    int ls_log_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> iter_ev;
    std::vector<double> iter_times;
    bool times_dirty = true;              /* iter_times / phase_times may hold times of an earlier (profiled) solve */
    /* profile level 3 (profiling.h:38-68): events around the phase groups of every iteration on the launch-per-level path:
     * [iteration][0..4] = start, gradient + termination test + dual Hessian done, factorisation + substitution done, line search done */
    std::vector<hipEvent_t> phase_ev;
    std::vector<double> phase_times;      /* [iteration][3]: build_dual, newton_direction, line_search (seconds) */
    double first_sweep_time = NAN;        /* phase S of iteration 0 (the later ones are the accepted trial sweeps of the line searches) */
    hipEvent_t sweep_ev0 = nullptr, sweep_ev1 = nullptr;
    int last_iter = 0;
    int last_ls_extra = 0;          /* the previous solve needed line-search trials beyond the first of an iteration */
    bool need_init = true;
    /* fused path for uniform complete trees */
    int fast = -1;            /* index into the instantiation table, -1: generic path only */
    int fNX = 0, fNU = 0, fMD = 0;
    int n_tiers = 0;          /* tiers of block levels, index 0 = bottom */
    /* pinned host mirrors: the inputs in slab layout (compare-and-copy uploads of what changed, no synchronisation),
     * the solution in one piece */
    char *h_in = nullptr; size_t in_off0 = 0, in_bytes = 0; bool in_valid = false;
    double *h_lam = nullptr; bool lam_valid = false;
    double *d_out = nullptr, *h_out = nullptr; size_t out_doubles = 0;
    int x_pad = 0, A_pad = 0; /* phantom root states of an x0-eliminated tree embedded in a uniform one (doubles in front of x-sized arrays / of A) */
    bool mstage = false;      /* multistage tree (branching for Nr stages, then chains): persistent kernel f_mpersist only */
    int ms_Nr = 0, ms_S = 0, ms_nB = 0;
    std::vector<int> tier_chain;
    std::vector<int> tier_l0, tier_l1, tier_grid;
    size_t lds_fast = 0, lds_fstage = 0;
    int use_fast = 1;         /* can be switched off (TREEQP_AMD_PATH=generic) */
    int use_persist = 1;      /* whole Newton loop in one launch when every tier workgroup can be co-resident */
    int use_persist_orig = 1, persist_backoff = 0, n_timeouts = 0;   /* see solve_after_timeout */
    bool persist_ok = false;
    PGeom geom{};
    PSync psync{};
    PConst pconst{};
    void *sync_slab = nullptr; size_t sync_bytes = 0;
    int *d_desc = nullptr;
    GItem *d_gitems = nullptr, *h_gitems = nullptr; int gitems_cap = 0;   /* batched single-workgroup launches (first mirror of a batch owns the array) */
    PItem *d_pitems = nullptr, *h_pitems = nullptr; int pitems_cap = 0;   /* batched persistent launches: one descriptor per tree (first mirror of a batch owns the array) */
    std::vector<unsigned long> pitems_key;                                /* the mirrors (by uid) the device copy of the array describes */
    unsigned long uid = 0;                                                /* unique per mirror of this process (an address can come back) */
    bool stream_pending = false;                                          /* something was enqueued on `stream` without a synchronisation after it (asynchronous uploads, constant packing): a batch launch on ANOTHER stream waits for it first */
    hipEvent_t batch_ev = nullptr;
    hipStream_t batch_stream = nullptr;                                   /* member of a batch launch in flight: the stream that launch is on (the lead's) */
    bool export_ahead = false;          /* tqgpu_set_export_ahead: the packing kernel and the download of the solution are enqueued right behind a single persistent launch */
    bool export_valid = false;          /* h_out holds (or is about to hold, stream-ordered) the solution of the last solve */
    hipStream_t settle_stream = nullptr;                                  /* member of a batch launch whose verdict is in but whose last workgroups may still write back:
                                                                           * the lead's stream, to be waited for before this mirror is touched through its own stream (settle) */
    size_t sync_words_bytes = 0, lds_persist = 0;
    /* ONE tree over several devices INSIDE the persistent launch (tqgpu_pshard_*): this rank's share of the workgroups */
    bool strict_sum = false;        /* TREEQP_AMD_STRICT_SUM=1 (Data::strict) */
    bool pshard = false;
    bool ps_sys = true;             /* the sharded launch polls its slab with system-scope loads (f_persist_sh<.., 1>); TREEQP_AMD_PSHARD_AGENT=1: agent scope, as on one device */
    bool ps_fine = false;           /* the hand-over slab was re-allocated as fine-grained memory (tqgpu_pshard_init; TREEQP_AMD_PSHARD_COARSE=1 keeps plain hipMalloc memory) */
    int *ps_wg_map = nullptr;       /* blockIdx.x -> workgroup id, this rank's workgroups (bottom tier first) */
    int ps_G = 0;
    void *ps_ipc[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      /* peer slabs opened through IPC handles (closed by tqgpu_destroy) */
    unsigned long long *h_peers[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   /* slabs of all ranks ... */
    unsigned long long **d_peers = nullptr;                                                               /* ... and the device copy of the table the kernel reads */
    /* sharded mode */
    int nranks = 1, rank = 0, part_top = -1;      /* part_top: highest partitioned tier */
    bool sharded = false;     /* subtree-sharded mode (nranks > 1, or ONE rank with a communicator: the same code path, used to exercise the RCCL transport on a one-GPU box) */
    int *d_gh_list = nullptr, *d_node_list = nullptr, *d_node_cnt_list = nullptr, *d_blk_list = nullptr;
    int gh_n = 0, gh_counted = 0, n_nodes = 0, n_nodes_counted = 0, n_blk_counted = 0;
    double *d_xerr = nullptr, *d_xs = nullptr;
    int bnd_b0 = 0, bnd_bn = 0, bnd_own0 = 0, bnd_ownn = 0;   /* element ranges of the boundary nodes' duals */
    void *shard_slab = nullptr;
    void *comm = nullptr;     /* RCCL communicator (nullptr: single device or virtual ranks) */
    int chunk = 4;            /* Newton iterations enqueued per status read-back */
};

extern "C" const char *tqgpu_last_error(void) { return g_err.c_str(); }
/* A batch launch runs on its lead's stream; the other members' own streams are not ordered behind it.  Its end is waited for lazily:
 * a loop of batch solves on the same lead stays stream-ordered by itself and pays no synchronisation per step (10 - 15 us of a
 * 150 us step), anything else that touches a member comes through here first. */
/* (the lead may have been destroyed in the meantime -- tqgpu_destroy synchronises its stream first, so there is nothing left to wait
 * for, but the handle must not be used: the streams of live mirrors are kept in a set) */
static std::mutex g_streams_mu;
static std::set<hipStream_t> g_live_streams;
static int settle(tqgpu_solver *s) {
    if (s && s->settle_stream) {
        hipStream_t t = s->settle_stream;
        s->settle_stream = nullptr;
        bool live;
        { std::lock_guard<std::mutex> lk(g_streams_mu); live = g_live_streams.count(t) != 0; }
        if (t != s->stream && live) HIP_TRY(hipStreamSynchronize(t));
    }
    return TQGPU_OK;
}
#define SETTLE(s) do { int rc_ = settle(const_cast<tqgpu_solver *>(s)); if (rc_ != TQGPU_OK) return rc_; } while (0)
extern "C" const char *tqgpu_version(void) { return "treeqp_amd tdunes device path r3 (gfx950: persistent single launch, three-launch MFMA family for 16 < d <= 64, single-workgroup and launch-per-phase kernels; sharded persistent mode)"; }

extern "C" int tqgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {

constexpr int EV_RING = 512;     /* solves whose device times can still be asked for */

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; }
};

template <typename T>
T *at(void *base, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(base) + off); }

int build_tables(tqgpu_solver *s) {
    const int Nn = s->Nn;
    s->dad.assign(Nn, -1); s->stage.assign(Nn, 0); s->kid0.assign(Nn, -1);
    int cursor = 1;
    for (int i = 0; i < Nn; i++) {
        if (s->nk[i] < 0) return fail(TQGPU_EINVAL, "negative number of children");
        if (s->nk[i] > 0) s->kid0[i] = cursor;
        for (int c = 0; c < s->nk[i]; c++) {
            if (cursor + c >= Nn) return fail(TQGPU_EINVAL, "children counts exceed the number of nodes");
            s->dad[cursor + c] = i;
            s->stage[cursor + c] = s->stage[i] + 1;
        }
        cursor += s->nk[i];
    }
    if (cursor != Nn) return fail(TQGPU_EINVAL, "children counts do not add up to Nn - 1");
    s->Np = 0;
    for (int i = 0; i < Nn; i++) s->Np += s->nk[i] > 0;
    for (int i = 0; i < Nn; i++)
        if ((s->nk[i] > 0) != (i < s->Np)) return fail(TQGPU_EINVAL, "tdunes needs all leaves at the same depth");
    s->Nh = s->stage[Nn - 1];
    s->lvl_first.assign(s->Nh + 2, Nn);
    for (int i = Nn - 1; i >= 0; i--) s->lvl_first[s->stage[i]] = i;
    s->lvl_first[s->Nh + 1] = Nn;

    s->xoff.assign(Nn + 1, 0); s->uoff.assign(Nn + 1, 0); s->aoff.assign(Nn + 1, 0); s->boff.assign(Nn + 1, 0);
    s->pos.assign(Nn, 0); s->bdim.assign(Nn, 0); s->woff.assign(Nn + 1, 0); s->utoff.assign(Nn + 1, 0);
    for (int k = 0; k < Nn; k++) {
        if (s->nx[k] < 0 || s->nu[k] < 0) return fail(TQGPU_EINVAL, "negative dimension");
        s->xoff[k + 1] = s->xoff[k] + s->nx[k];
        s->uoff[k + 1] = s->uoff[k] + s->nu[k];
        if (k > 0) {
            s->aoff[k + 1] = s->aoff[k] + s->nx[k] * s->nx[s->dad[k]];
            s->boff[k + 1] = s->boff[k] + s->nx[k] * s->nu[s->dad[k]];
            int first = s->kid0[s->dad[k]];
            for (int j = first; j < k; j++) s->pos[k] += s->nx[j];     /* dual_Newton_tree.c:177-194 */
        }
        int d = 0;
        for (int c = 0; c < s->nk[k]; c++) d += s->nx[s->kid0[k] + c];
        s->bdim[k] = d;
        s->woff[k + 1] = s->woff[k] + d * d;
        s->utoff[k + 1] = s->utoff[k] + (k > 0 ? s->nx[k] * d : 0);
    }
    s->nx0 = s->nx[0];
    s->sum_nx = s->xoff[Nn]; s->sum_nu = s->uoff[Nn]; s->sum_lam = s->sum_nx - s->nx0;
    s->sum_A = s->aoff[Nn]; s->sum_B = s->boff[Nn]; s->sum_W = s->woff[Nn]; s->sum_Ut = s->utoff[Nn];

    /* LDS budgets of the wave-per-block kernels */
    s->lds_stage = s->lds_hess = s->lds_factor = s->lds_forward = 0;
    for (int k = 0; k < Nn; k++) {
        const size_t d = s->bdim[k], nz = s->nx[k] + s->nu[k];
        s->lds_stage = std::max(s->lds_stage, (d + s->nx[k] + 2 * (s->nx[k] + s->nu[k]) + 2) * sizeof(double));
        s->lds_dense = std::max(s->lds_dense, ((size_t)(s->nx[k] + s->nu[k]) * (s->nx[k] + s->nu[k] + 1) + 2) * sizeof(double));
        if (k < s->Np) {
            s->lds_hess = std::max(s->lds_hess, (2 * d * nz + 2) * sizeof(double));
            const size_t R = d + 1 + (k > 0 ? s->nx[k] : 0), ld = R | 1;
            s->lds_factor = std::max(s->lds_factor, (ld * d + d + 2) * sizeof(double));
            s->lds_forward = std::max(s->lds_forward, ((d | 1) * d + 2 * d + s->nx[k] + 2) * sizeof(double));
        }
    }
    {
        int dmax = 0, rmax = 0, nzmax = 0;
        for (int k = 0; k < s->Np; k++) {
            const int d = s->bdim[k], nxi = k > 0 ? s->nx[k] : 0, nz = s->nx[k] + s->nu[k];
            dmax = std::max(dmax, d); rmax = std::max(rmax, wide_rows(d, nxi)); nzmax = std::max(nzmax, nz);
            s->lds_hess_w = std::max(s->lds_hess_w, wide_lds_hess(d, nz));
            s->lds_factor_w = std::max(s->lds_factor_w, wide_lds_factor(d, nxi));
            s->lds_forward_w = std::max(s->lds_forward_w, wide_lds_forward(d));
        }
        s->wide = dmax > 16 && dmax <= 64 && rmax <= 128 && nzmax <= 32;      /* k_hess_w keeps a parent's entries of P for 8 k-steps of 4 in registers */
        /* Trees of SMALL blocks (d <= 16) that are too wide for the single-workgroup kernel (a level of more than 6 x 16 blocks) take the
         * workgroup-per-block kernels as well: three launches per Newton iteration instead of the five to eight of the launch-per-phase
         * kernels (TREEQP_AMD_SMALL_WIDE=0: as before) */
        {
            int widest = 0;
            for (int l = 0; l + 1 < (int)s->lvl_first.size(); l++) widest = std::max(widest, s->lvl_first[l + 1] - s->lvl_first[l]);
            const char *e = getenv("TREEQP_AMD_SMALL_WIDE");
            if (!s->wide && dmax >= 3 && dmax <= 16 && widest > 6 * 16 && rmax <= 128 && nzmax <= 32 && !(e && atoi(e) == 0)) s->wide = true;
        }
    }
    const size_t lim = 160 * 1024;
    if (s->lds_factor > lim || s->lds_hess > lim)
        return fail(TQGPU_EUNSUPPORTED, "dual Hessian block too large for the LDS-resident kernels (160 KiB per workgroup)");
    return TQGPU_OK;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TQGPU_OK;
}



bool shard_instantiated(int idx) {
#define X(i, nx, nu, md) if (idx == i) return true;
    SHARD_TABLE(X)
#undef X
    return false;
}

int fast_index(int NX, int NU, int MD) {
#define X(idx, nx, nu, md) if (NX == nx && NU == nu && MD == md) return idx;
    FAST_TABLE(X)
#undef X
    return -1;
}

void fast_geometry(int idx, int &TH, size_t &tier_lds, size_t &stage_lds) {
#define X(i, nx, nu, md) if (idx == i) { TH = Uni<nx, nu, md>::TH; tier_lds = Uni<nx, nu, md>::TIER_LDS * sizeof(double); stage_lds = FW * (Uni<nx, nu, md>::D + nx + 8) * sizeof(double); }
    FAST_TABLE(X)
#undef X
}


/* multistage tree?  (setup_multistage_tree(md, Nr, Nh) with 1 <= Nr < Nh: every node above stage Nr has md
 * children, every parent from stage Nr on has one; uniform nx, nu) */
void detect_multistage(tqgpu_solver *s) {
    s->mstage = false;
    const int Nn = s->Nn, NX = s->nx[0], NU = s->nu[0], MD = s->nk[0], Nh = s->Nh;
    if (Nh < 2 || MD < 2 || NX % 4 != 0) return;
    int Nr = 0;
    while (Nr < Nh) {
        bool all = true;
        for (int k = s->lvl_first[Nr]; k < s->lvl_first[Nr + 1]; k++) if (s->nk[k] != MD) { all = false; break; }
        if (!all) break;
        Nr++;
    }
    if (Nr < 1 || Nr >= Nh) return;
    for (int k = 0; k < Nn; k++) {
        if (s->nx[k] != NX) return;
        if (k < s->Np) { if (s->nu[k] != NU) return; if (k >= s->lvl_first[Nr] && s->nk[k] != 1) return; }
        else if (s->nu[k] != 0) return;
    }
    int idx = -1;
#define X(i, nx, nu, md) if (NX == nx && NU == nu && MD == md) idx = i;
    MSTAGE_TABLE(X)
#undef X
    if (idx < 0) return;
    int S = 1;
    for (int l = 0; l < Nr; l++) S *= MD;
    s->fast = idx; s->fNX = NX; s->fNU = NU; s->fMD = MD;
    s->mstage = true; s->ms_Nr = Nr; s->ms_S = S; s->ms_nB = s->lvl_first[Nr];
    int TH = 1;
    fast_geometry(idx, TH, s->lds_fast, s->lds_fstage);
    s->tier_l0.clear(); s->tier_l1.clear(); s->tier_grid.clear(); s->tier_chain.clear();
    /* chain part bottom-up in tiers of at most 8 levels (Uni<..., 1>::TH), then the branching part in tiers of TH levels */
    for (int l1 = Nh; l1 > Nr;) {
        const int l0 = std::max(Nr, l1 - 8);
        s->tier_l0.push_back(l0); s->tier_l1.push_back(l1); s->tier_grid.push_back(S); s->tier_chain.push_back(1);
        l1 = l0;
    }
    for (int l1 = Nr; l1 > 0;) {
        const int l0 = std::max(0, l1 - TH);
        int grid = 1;
        for (int l = 0; l < l0; l++) grid *= MD;
        s->tier_l0.push_back(l0); s->tier_l1.push_back(l1); s->tier_grid.push_back(grid); s->tier_chain.push_back(0);
        l1 = l0;
    }
    s->n_tiers = (int)s->tier_l0.size();
}

/* uniform complete tree? (every node nx, every parent nu + md children, one leaf depth) */
void detect_fast(tqgpu_solver *s) {
    s->fast = -1;
    const int Nn = s->Nn, NX = s->nx[0], NU = s->nu[0], MD = s->nk[0];
    if (s->Nh < 2 || MD < 2) return;
    for (int k = 0; k < Nn; k++) {
        if (s->nx[k] != NX) return;
        if (k < s->Np) { if (s->nu[k] != NU || s->nk[k] != MD) return; }
        else if (s->nu[k] != 0) return;
    }
    const int idx = fast_index(NX, NU, MD);
    if (idx < 0) return;
    s->fast = idx; s->fNX = NX; s->fNU = NU; s->fMD = MD;
    int TH = 1;
    fast_geometry(idx, TH, s->lds_fast, s->lds_fstage);
    /* block levels 0 .. Nh-1 grouped bottom-up into tiers of TH levels; the top tier takes the rest */
    const int Nh = s->Nh;
    s->n_tiers = (Nh + TH - 1) / TH;
    s->tier_l0.clear(); s->tier_l1.clear(); s->tier_grid.clear(); s->tier_chain.clear();
    for (int i = 0; i < s->n_tiers; i++) {
        const int l1 = Nh - i * TH, l0 = std::max(0, l1 - TH);
        int grid = 1;
        for (int l = 0; l < l0; l++) grid *= MD;
        s->tier_l0.push_back(l0); s->tier_l1.push_back(l1); s->tier_grid.push_back(grid); s->tier_chain.push_back(0);
    }
}

/* ---- RCCL, loaded lazily so that single-device users do not depend on it ---- */
struct RcclApi {
    void *lib = nullptr;
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.lib) return TQGPU_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return fail(TQGPU_ECOMM, std::string("cannot load librccl.so: ") + dlerror());
#define SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name)); if (!g_rccl.field) return fail(TQGPU_ECOMM, std::string("librccl.so lacks ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = lib;
    return TQGPU_OK;
}
constexpr int NCCL_DOUBLE = 8;     /* ncclFloat64 */

#define NCCL_TRY(expr) do { int r_ = (expr); if (r_ != 0) return fail(TQGPU_ECOMM, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); } while (0)

int uni_first(int MD, int level) { int n = 0, w = 1; for (int l = 0; l < level; l++) { n += w; w *= MD; } return n; }

Shard shard_desc(const tqgpu_solver *s, int tier) {
    Shard sh{};
    if (s->sharded) {
        if (tier >= 0 && tier <= s->part_top) sh.wg_off = s->rank * (s->tier_grid[tier] / s->nranks);
        sh.gh_list = s->d_gh_list; sh.gh_n = s->gh_n; sh.gh_counted = s->gh_counted;
        sh.err_src = s->d_xerr;
    }
    return sh;
}

/* exchange #1 (after the last partitioned backward tier): Schur records of the boundary subtree
 * roots and the termination partials; exchange #2 (after the trial sweep): {fval, dot} partials and
 * x / QinvCal of the boundary root nodes.  In-place all-gathers on the solver's stream. */
int shard_exchange_rccl(tqgpu_solver *s, int which) {
    const int N = s->nranks, r = s->rank, MD = s->fMD, NX = s->fNX;
    const int lb = s->tier_l0[s->part_top], gb = s->tier_grid[s->part_top], w = gb / N;
    const int f0 = uni_first(MD, lb);
    const size_t SCH = (size_t)NX * NX + NX;
    NCCL_TRY(g_rccl.GroupStart());
    if (which == 1) {
        double *base = s->D.Sbuf + (size_t)f0 * SCH;
        NCCL_TRY(g_rccl.AllGather(base + (size_t)r * w * SCH, base, (size_t)w * SCH, NCCL_DOUBLE, s->comm, s->stream));
        NCCL_TRY(g_rccl.AllGather(s->d_xerr + r, s->d_xerr, 1, NCCL_DOUBLE, s->comm, s->stream));
    } else {
        NCCL_TRY(g_rccl.AllGather(s->d_xs + 2 * r, s->d_xs, 2, NCCL_DOUBLE, s->comm, s->stream));
        double *xb = s->D.x + (size_t)NX * f0, *qb = s->D.QinvCal + (size_t)NX * f0;
        NCCL_TRY(g_rccl.AllGather(xb + (size_t)r * w * NX, xb, (size_t)w * NX, NCCL_DOUBLE, s->comm, s->stream));
        NCCL_TRY(g_rccl.AllGather(qb + (size_t)r * w * NX, qb, (size_t)w * NX, NCCL_DOUBLE, s->comm, s->stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return TQGPU_OK;
}

/* the same exchanges between `n` mirrors living in one process on one device ("virtual ranks"):
 * used to validate the partition / hand-off logic on a single GPU */
int shard_exchange_virtual(tqgpu_solver **R, int n, int which) {
    tqgpu_solver *s0 = R[0];
    const int MD = s0->fMD, NX = s0->fNX;
    const int lb = s0->tier_l0[s0->part_top], gb = s0->tier_grid[s0->part_top], w = gb / n;
    const int f0 = uni_first(MD, lb);
    const size_t SCH = (size_t)NX * NX + NX;
    for (int r = 0; r < n; r++) HIP_TRY(hipStreamSynchronize(R[r]->stream));
    /* every source stream is idle now; the copies are ordered on the DESTINATION mirror's stream, in
     * front of its next phase (the solver streams are non-blocking: the null stream would not order) */
    for (int src = 0; src < n; src++) for (int dst = 0; dst < n; dst++) {
        if (src == dst) continue;
        tqgpu_solver *a = R[src], *b = R[dst];
        hipStream_t st = b->stream;
        if (which == 1) {
            const size_t off = ((size_t)f0 + (size_t)src * w) * SCH;
            HIP_TRY(hipMemcpyAsync(b->D.Sbuf + off, a->D.Sbuf + off, sizeof(double) * w * SCH, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(b->d_xerr + src, a->d_xerr + src, sizeof(double), hipMemcpyDeviceToDevice, st));
        } else {
            HIP_TRY(hipMemcpyAsync(b->d_xs + 2 * src, a->d_xs + 2 * src, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
            const size_t off = (size_t)NX * (f0 + (size_t)src * w);
            HIP_TRY(hipMemcpyAsync(b->D.x + off, a->D.x + off, sizeof(double) * w * NX, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(b->D.QinvCal + off, a->D.QinvCal + off, sizeof(double) * w * NX, hipMemcpyDeviceToDevice, st));
        }
    }
    for (int r = 0; r < n; r++) HIP_TRY(hipStreamSynchronize(R[r]->stream));
    return TQGPU_OK;
}

/* One fused Newton iteration, split in three phases around the two exchange points of the sharded
 * mode.  Single device: phases run back to back, no exchange. */
void launch_fast_phase(tqgpu_solver *s, const Opts &O, int h, int phase, int &launches) {
    const Tree &T = s->T; const Data &D = s->D; hipStream_t st = s->stream;
    const dim3 blk(FW * WAVE);
    const int nt = s->n_tiers, N = s->nranks;
    const bool sharded = s->sharded;
    const int P = sharded ? s->part_top : -1;
    auto tgrid = [&](int i) { return (sharded && i <= P) ? s->tier_grid[i] / N : s->tier_grid[i]; };
    /* which kernel runs first / performs the termination test */
    const int check_tier = sharded ? P + 1 : 1;          /* index in 0..nt-1 (nt-1 = top) */
    const int nparts = sharded ? N : (nt > 1 ? s->tier_grid[0] : FW);
    const int n_stage = sharded ? s->n_nodes : T.Nn;
    switch (s->fast) {
#define X(idx, nx, nu, md)                                                                                                   \
    case idx:                                                                                                                \
        if (phase == 0) {                                                                                                    \
            for (int i = 0; i < nt - 1 && (!sharded || i <= P); i++) {                                                       \
                hipLaunchKernelGGL((f_back<nx, nu, md>), dim3(tgrid(i)), blk, s->lds_fast, st, T, D, O, shard_desc(s, i),    \
                                   s->tier_l0[i], s->tier_l1[i], i == 0, !sharded && i == check_tier, nparts, i, h); launches++; \
            }                                                                                                                \
            if (sharded) { hipLaunchKernelGGL(k_shard_pack1, dim3(1), dim3(256), 0, st, D, tgrid(0), s->d_xerr, s->rank, O.termCondition, h); launches++; } \
        }                                                                                                                    \
        if (phase == 1) {                                                                                                    \
            for (int i = (sharded ? P + 1 : nt - 1); i < nt - 1; i++) {                                                      \
                hipLaunchKernelGGL((f_back<nx, nu, md>), dim3(tgrid(i)), blk, s->lds_fast, st, T, D, O, shard_desc(s, i),    \
                                   s->tier_l0[i], s->tier_l1[i], 0, i == check_tier, nparts, i, h); launches++;              \
            }                                                                                                                \
            hipLaunchKernelGGL((f_top<nx, nu, md>), dim3(1), blk, s->lds_fast, st, T, D, O, shard_desc(s, nt - 1),           \
                               s->tier_l1[nt - 1], nt == 1, (nt - 1) == check_tier, nparts, nt - 1, h); launches++;          \
            for (int i = nt - 2; i >= 0; i--) {                                                                              \
                hipLaunchKernelGGL((f_fwd<nx, nu, md>), dim3(tgrid(i)), blk, s->lds_fast, st, T, D, O, shard_desc(s, i),     \
                                   s->tier_l0[i], s->tier_l1[i], nt + (nt - 2 - i), h); launches++;                          \
            }                                                                                                                \
            hipLaunchKernelGGL((f_stage<nx, nu, md>), dim3((n_stage + FW - 1) / FW), blk, s->lds_fstage, st, T, D, O,        \
                               sharded ? s->d_node_list : nullptr, n_stage, 2 * nt - 1, h, 1); launches++;                   \
            if (sharded) { hipLaunchKernelGGL(k_shard_pack2, dim3(1), dim3(WAVE), 0, st, D, s->d_node_cnt_list, s->n_nodes_counted, \
                                              s->d_blk_list, s->n_blk_counted, s->d_xs, s->rank, s->bnd_b0, s->bnd_bn, s->bnd_own0, s->bnd_ownn, h, 1); launches++; } \
        }                                                                                                                    \
        break;
        FAST_TABLE(X)
#undef X
        default: break;
    }
    if (phase == 2) {
        if (sharded) hipLaunchKernelGGL(k_ls_decide_parts, dim3(1), dim3(WAVE), 0, st, D, O, s->d_xs, N, h, 1, 1);
        else hipLaunchKernelGGL(k_ls_decide, dim3(1), dim3(256), 0, st, T, D, O, h, 1, 1);
        launches++;
    }
}

int launch_fast_iteration(tqgpu_solver *s, const Opts &O, int h, int &launches) {
    launch_fast_phase(s, O, h, 0, launches);
    if (s->sharded) { int rc = shard_exchange_rccl(s, 1); if (rc) return rc; }
    launch_fast_phase(s, O, h, 1, launches);
    if (s->sharded) { int rc = shard_exchange_rccl(s, 2); if (rc) return rc; }
    launch_fast_phase(s, O, h, 2, launches);
    return TQGPU_OK;
}

/* one more line-search trial of iteration `it`; phase 0: sweep (+ pack), phase 1: decide */
static Fuse next_fuse(tqgpu_solver *s) {
    Fuse F; F.red = s->fuse_red; F.cnt = s->fuse_cnt; F.on = 1;
    s->fuse_epoch++;
    if (s->fuse_epoch == 0) s->fuse_epoch = 1;
    F.tag = s->fuse_epoch;
    return F;
}
static Fuse no_fuse() { Fuse F; F.red = nullptr; F.cnt = nullptr; F.tag = 0; F.on = 0; return F; }
static W3 next_w3(tqgpu_solver *s) {
    W3 w; w.xu = s->w3_xu; w.red = s->w3_red; w.cnt = s->w3_cnt; w.sum_nx = s->sum_nx; w.lds_wave = (int)((s->lds_stage + 7) / 8);
    w.hm = nullptr;
    s->w3_tail_sg = false;
    s->w3_epoch++;
    if (s->w3_epoch == 0) s->w3_epoch = 1;
    w.tag = s->w3_epoch;
    return w;
}
static void launch_sg(tqgpu_solver *s, const Opts &O, int mode, int h, int t, bool fresh = false) {
    W3 w = next_w3(s);
    /* (a post is ~14 system-scope stores to host memory and the wait for their acknowledgements, on the launch's critical path: only
     * where the host is going to look) */
    if (s->w3_mirror && s->w3_post_next) { w.hm = s->h_res; s->w3_wait = w.tag; s->w3_tail_sg = true; }
    s->w3_post_next = false;
    if (s->w3_sgp) {
        if (mode == 2) { SgpFwdYes fa; fa.anc = s->d_anc; fa.pdw = s->d_pdw; hipLaunchKernelGGL(k_sgp_t<true>, dim3(s->T.Np), dim3(WT), s->lds_sgp, s->stream, s->T, s->D, O, w, mode, h, t, s->sgp_accs, (const double *)nullptr, fa); }
        else hipLaunchKernelGGL(k_sgp_t<false>, dim3(s->T.Np), dim3(WT), s->lds_sgp, s->stream, s->T, s->D, O, w, mode, h, t, s->sgp_accs, fresh ? (const double *)s->d_lam_init : (const double *)nullptr, SgpFwdNo());
        return;
    }
    const int grid = (s->T.Nn + SG_WAVES - 1) / SG_WAVES;
    hipLaunchKernelGGL(k_sg, dim3(grid), dim3(SG_WAVES * WAVE), SG_WAVES * ((s->lds_stage + 7) / 8) * 8, s->stream, s->T, s->D, O, w, mode, h, t);
}

void launch_trial_phase(tqgpu_solver *s, const Opts &O, bool fast, int it, int t, int phase, int &launches) {
    const Tree &T = s->T; const Data &D = s->D; hipStream_t st = s->stream;
    const bool sharded = s->sharded;
    if (phase == 0) {
        bool done = false;
        if (fast) {
            const int n_stage = sharded ? s->n_nodes : T.Nn;
            switch (s->fast) {
#define X(idx, nx, nu, md) case idx: hipLaunchKernelGGL((f_stage<nx, nu, md>), dim3((n_stage + FW - 1) / FW), dim3(FW * WAVE), s->lds_fstage, st, T, D, O, sharded ? s->d_node_list : nullptr, n_stage, 7, it, t); done = true; break;
                FAST_TABLE(X)
#undef X
                default: break;
            }
        }
        if (!done && s->w3_now) { launch_sg(s, O, 1, it, t); done = true; }      /* with the Armijo test and the next termination test as its tail */
        if (!done) {
            if (s->fuse_now && !sharded) hipLaunchKernelGGL(k_stage_f, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, O, next_fuse(s), 1, it, t);      /* with k_ls_decide as its tail */
            else hipLaunchKernelGGL(k_stage, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, 1, it, t);
        }
        launches++;
        if (sharded) { hipLaunchKernelGGL(k_shard_pack2, dim3(1), dim3(WAVE), 0, st, D, s->d_node_cnt_list, s->n_nodes_counted, s->d_blk_list, s->n_blk_counted, s->d_xs, s->rank, s->bnd_b0, s->bnd_bn, s->bnd_own0, s->bnd_ownn, it, t); launches++; }
    } else {
        if (sharded) { hipLaunchKernelGGL(k_ls_decide_parts, dim3(1), dim3(WAVE), 0, st, D, O, s->d_xs, s->nranks, it, t, 0); launches++; }
        else if (!((s->fuse_now || s->w3_now) && !fast)) { hipLaunchKernelGGL(k_ls_decide, dim3(1), dim3(256), 0, st, T, D, O, it, t, 0); launches++; }
    }
}

int launch_trial(tqgpu_solver *s, const Opts &O, bool fast, int it, int t, int &launches) {
    launch_trial_phase(s, O, fast, it, t, 0, launches);
    if (s->sharded) { int rc = shard_exchange_rccl(s, 2); if (rc) return rc; }
    launch_trial_phase(s, O, fast, it, t, 1, launches);
    return TQGPU_OK;
}


/* persistent launch: geometry, sync words, co-residency test */
/* poll naps by launch size (PollGuard::go_on) */
static int nap_for_grid(int workgroups) {
    static const int t1 = getenv("TREEQP_AMD_NAP_T1") ? atoi(getenv("TREEQP_AMD_NAP_T1")) : 128;
    return workgroups > t1 ? 1 : 0;
}
int setup_persist(tqgpu_solver *s, int device) {
    s->persist_ok = false;
    if (s->fast < 0 || s->n_tiers > 8) return TQGPU_OK;
    PGeom &G = s->geom;
    G.n_tiers = s->n_tiers;
    int wg = 0;
    for (int i = 0; i < s->n_tiers; i++) { G.l0[i] = s->tier_l0[i]; G.l1[i] = s->tier_l1[i]; G.grid[i] = s->tier_grid[i]; G.chain[i] = s->tier_chain[i]; G.wg0[i] = wg; wg += s->tier_grid[i]; }
    G.G = wg;
    int per_cu = 0, per_cu_r = 1 << 20, per_cu_one = 0;      /* _r: the variant that can keep factors (checkLastActiveSet == 2); _one: the build for one workgroup per CU */
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!s->mstage) {
        switch (s->fast) {
#define X(idx, nx, nu, md)                                                                                                   \
    case idx:                                                                                                                \
        s->lds_persist = PLds<nx, nu, md>::DOUBLES * sizeof(double);                                                         \
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f_persist<nx, nu, md, false>, FW * WAVE, s->lds_persist)); \
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, f_persist<nx, nu, md, true>, FW * WAVE, s->lds_persist)); \
        if (allow_lds(f_persist_one<nx, nu, md>, s->lds_persist) != TQGPU_OK || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_one, f_persist_one<nx, nu, md>, FW * WAVE, s->lds_persist) != hipSuccess) per_cu_one = 0; \
        break;
            FAST_TABLE(X)
#undef X
            default: break;
        }
    } else {
        switch (s->fast) {
#define X(idx, nx, nu, md)                                                                                                   \
    case idx:                                                                                                                \
        s->lds_persist = std::max(PLds<nx, nu, md>::DOUBLES, PLds<nx, nu, 1>::DOUBLES) * sizeof(double);                     \
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f_mpersist<nx, nu, md, false>, FW * WAVE, s->lds_persist)); \
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, f_mpersist<nx, nu, md, true>, FW * WAVE, s->lds_persist)); \
        if (allow_lds(f_mpersist_one<nx, nu, md>, s->lds_persist) != TQGPU_OK || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_one, f_mpersist_one<nx, nu, md>, FW * WAVE, s->lds_persist) != hipSuccess) per_cu_one = 0; \
        break;
            MSTAGE_TABLE(X)
#undef X
            default: break;
        }
    }
    if (const char *e = getenv("TREEQP_AMD_LDS_PAD")) {      /* experiment: force fewer workgroups per CU */
        s->lds_persist += (size_t)atoi(e) * 1024;
        switch (s->fast) {
#define X(idx, nx, nu, md) case idx: if (!s->mstage) { allow_lds(f_persist<nx, nu, md, false>, s->lds_persist); HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f_persist<nx, nu, md, false>, FW * WAVE, s->lds_persist)); } break;
            FAST_TABLE(X)
#undef X
            default: break;
        }
    }
    /* every workgroup must be resident at once (they wait for each other); keep one block per CU of
     * margin against the occupancy query over-reporting (MI355X guide, "Residency and cooperative launch") */
    per_cu = std::min(per_cu, per_cu_r);
    /* one workgroup per CU needs no margin (the figure is exact when registers allow a single 4-wave workgroup);
     * with more per CU keep half a CU-load of workgroups in hand */
    /* (rounds 1 - 2 kept half a CU-load in hand; the admission rule of the MI355X guide -- min(API answer, 8, 800 / (sgpr rounded up to
     * 16 + 16)) per CU -- is what per_cu_r holds, the grid is admitted CU by CU, and a launch that is not resident after all ends in the
     * bounded waits' timeout and the launch-per-tier redo, not in a hang: no margin.  C2 batched: 5 trees 99 k it/s -> 7 trees 133 k.) */
    int margin = 0;
    if (const char *e = getenv("TREEQP_AMD_CAPACITY_MARGIN")) margin = atoi(e);      /* experiment */
    const int capacity = per_cu <= 1 ? prop.multiProcessorCount * per_cu : prop.multiProcessorCount * per_cu - margin;
    if (getenv("TREEQP_AMD_VERBOSE")) fprintf(stderr, "[treeqp_amd] persistent path: %d workgroups, %d per CU possible, capacity %d\n", G.G, per_cu, capacity);
    s->co_capacity = std::max(1, capacity);
    s->n_cu = prop.multiProcessorCount;
    /* the build for ONE workgroup per CU (f_persist_one: more registers, 2 % faster) when the launch fits the device that way */
    s->persist_one = per_cu_one >= 1 && G.G <= prop.multiProcessorCount * std::min(per_cu_one, 1) && !getenv("TREEQP_AMD_NO_PERSIST_ONE");
    if (per_cu < 1 || G.G > capacity) return TQGPU_OK;
    /* hand-over buffers (tagged 64-bit words, see tdunes_persist.hpp); zeroed once, never reset */
    const int nx0 = s->nx[0];
    const size_t n_sch = (size_t)s->Nn * (nx0 * nx0 + nx0) * 2, n_dlt = (size_t)s->sum_nx * 2, n_ndt = (size_t)s->Nn * 2 * nx0 * 2;
    const size_t n_parts = (size_t)G.G * 4, n_errs = (size_t)G.G * 2;
    const size_t n_bparts = (size_t)G.G * 16;
    const size_t n_sgt = (size_t)s->Nn * 2, n_rfl = (size_t)s->Nn * 2;      /* active-set signature / reuse flag of a tier subtree root (one tagged double each) */
    const size_t n_verdict = 16;
    const size_t n_anc = (size_t)s->Np * (size_t)(nx0 * s->fMD) * (nx0 + 1) * 2;      /* forward records [z0 | M] of the blocks, for the bottom tier's walk down its path (tdunes_persist.hpp, p_forward_tier) */
    const size_t bytes = (n_sch + n_dlt + n_ndt + n_parts + n_errs + 32 + n_bparts + n_sgt + n_rfl + n_verdict + n_anc) * sizeof(unsigned long long);
    HIP_TRY(hipMalloc(&s->sync_slab, bytes));
    HIP_TRY(hipMemset(s->sync_slab, 0, bytes));
    HIP_TRY(hipDeviceSynchronize());          /* (a device memset may return before it has happened, and the solver's non-blocking stream does not wait for the null stream) */
    s->sync_bytes = bytes;
    unsigned long long *w = static_cast<unsigned long long *>(s->sync_slab);
    s->psync.sch = w; s->psync.dlt = w + n_sch; s->psync.ndt = s->psync.dlt + n_dlt; s->psync.parts = s->psync.ndt + n_ndt;
    s->psync.errs = s->psync.parts + n_parts;
    s->psync.halt = reinterpret_cast<unsigned *>(s->psync.errs + n_errs);
    s->psync.timeout = s->psync.halt + 32;
    s->psync.cmd = s->psync.errs + n_errs + 24;          /* same 256-byte block: halt | timeout | cmd | vrd */
    s->psync.vrd = s->psync.errs + n_errs + 28;
    s->psync.bparts = s->psync.errs + n_errs + 32;
    s->psync.sgt = s->psync.bparts + n_bparts; s->psync.rfl = s->psync.sgt + n_sgt;
    s->psync.verdict = s->psync.rfl + n_rfl;
    s->psync.anc = s->psync.verdict + n_verdict;
    s->psync.base = w; s->psync.npeer = 1; s->psync.relay_wg = -1; s->psync.anc_local = 1;
    HIP_TRY(hipMalloc(&s->d_peers, 8 * sizeof(unsigned long long *)));
    for (int r = 0; r < 8; r++) s->h_peers[r] = w;
    HIP_TRY(hipMemcpy(s->d_peers, s->h_peers, sizeof(s->h_peers), hipMemcpyHostToDevice));
    s->psync.peers = s->d_peers;
    s->psync.seq = 0; s->psync.trip = 0;
    {
        const char *e = getenv("TREEQP_AMD_NAP");
        s->psync.nap = e ? atoi(e) : nap_for_grid(G.G);
    }
    /* XCD-aware placement: the hardware deals workgroups round-robin over the 8 XCDs (workgroup b -> XCD b % 8)
     * and every XCD has its own L2.  A tier subtree talks to its parent and its children only, so whole
     * families go to one XCD: each subtree of the lowest tier with at most 8 subtrees picks an XCD, everything
     * below follows its ancestor there; the tiers above (one workgroup each, typically) take what is left. */
    {
        const int NXCD = 8, Gn = G.G;
        int anchor = s->n_tiers - 1;
        for (int i = 0; i < s->n_tiers; i++) if (G.grid[i] <= NXCD) { anchor = i; break; }
        std::vector<std::vector<int>> want(NXCD);
        std::vector<int> rest;
        for (int i = 0; i < s->n_tiers; i++)
            for (int q = 0; q < G.grid[i]; q++) {
                const int id = G.wg0[i] + q;
                if (i <= anchor) want[(int)((long long)q * G.grid[anchor] / G.grid[i]) % NXCD].push_back(id);   /* subtree q of tier i sits under subtree q*grid[anchor]/grid[i] */
                else rest.push_back(id);
            }
        std::vector<int> map((size_t)Gn, -1);
        std::vector<size_t> next(NXCD, 0);
        for (int b = 0; b < Gn; b++) { auto &w = want[b % NXCD]; if (next[b % NXCD] < w.size()) map[b] = w[next[b % NXCD]++]; }
        for (int x = 0; x < NXCD; x++) for (size_t k = next[x]; k < want[x].size(); k++) rest.push_back(want[x][k]);
        size_t r = 0;
        for (int b = 0; b < Gn; b++) if (map[b] < 0) map[b] = rest[r++];
        /* more workgroups than CUs (two per CU): the workgroups dispatched last share a CU with an earlier one.  Let those be
         * the UPPER tiers -- they work while the bottom tier waits and vice versa -- instead of two bottom-tier subtrees
         * halving each other on the critical path: plain tier order (bottom tier first). */
        if (Gn > prop.multiProcessorCount) for (int b = 0; b < Gn; b++) map[b] = b;
        HIP_TRY(hipMalloc(&s->wg_map, sizeof(int) * (size_t)Gn));
        HIP_TRY(hipMemcpy(s->wg_map, map.data(), sizeof(int) * (size_t)Gn, hipMemcpyHostToDevice));
        G.wg_of_block = s->wg_map;
    }
    /* packed constants + the start/end view of the mirror */
    const int nz = nx0 + s->nu[0];
    const size_t n_ab = (size_t)(s->Nn - 1) * nx0 * nz, n_cst = (size_t)s->Nn * 16 * 5;
    HIP_TRY(hipMalloc(&s->pconst_slab, (n_ab + n_cst) * sizeof(double) + sizeof(PDump) + 256));
    double *pc = static_cast<double *>(s->pconst_slab);
    s->pab = pc; s->pcst = pc + n_ab;
    PDump hd;
    const Data &D = s->D;
    hd.x = D.x; hd.u = D.u; hd.xUnc = D.xUnc; hd.uUnc = D.uUnc; hd.xUncS = D.xUncS; hd.uUncS = D.uUncS; hd.qmod = D.qmod; hd.rmod = D.rmod; hd.QinvCal = D.QinvCal; hd.RinvCal = D.RinvCal;
    hd.lam0 = D.lam0; hd.lam1 = D.lam1; hd.dlam = D.dlam; hd.lam_init = s->d_lam_init; hd.stamps = D.stamps; hd.ls_log = D.ls_log; hd.ls_log_cap = D.ls_log_cap;
    hd.hres = s->h_res;
    PDump *dd = reinterpret_cast<PDump *>(pc + n_ab + n_cst);
    HIP_TRY(hipMemcpy(dd, &hd, sizeof(PDump), hipMemcpyHostToDevice));
    s->pconst.AB = s->pab; s->pconst.b = D.b; s->pconst.cst = s->pcst; s->pconst.ctrl = D.ctrl; s->pconst.dump = dd; s->pconst.lam0_src = s->d_lam_init; s->pconst.Np = s->Np;
    s->pconst.S = s->mstage ? s->ms_S : 0; s->pconst.nB = s->mstage ? s->ms_nB : s->Np; s->pconst.Nr = s->mstage ? s->ms_Nr : s->Nh;
    s->need_pack = true;
    s->persist_ok = true;
    return TQGPU_OK;
}

/* one persistent launch (prologue = first sweep of the solve + control block reset): no memset, the
 * hand-over words are told apart by the launch number in their tags */
int launch_persist(tqgpu_solver *s, const Opts &O, int &launches, int prologue, unsigned batch_seq = 0) {
    const Data &D = s->D; hipStream_t st = s->stream;
    if (batch_seq) s->launch_no = batch_seq >> 16;          /* part of a batch launch: the caller chose the number (and wiped the buffers if it wrapped) */
    else {
        s->launch_no = (s->launch_no + 1) & 0xFFFFu;
        if (s->launch_no == 0) {
            /* the 16-bit launch number wraps: words that are not rewritten by every launch (the line-search command / verdict /
             * batch partials) could still carry a tag of 65535 launches ago -- wipe the hand-over buffers (stream-ordered) */
            s->launch_no = 1;
            HIP_TRY(hipMemsetAsync(s->sync_slab, 0, s->sync_bytes, st));
        }
    }
    s->psync.seq = s->launch_no << 16;
    if (s->need_pack) {
        const int n = std::max(s->Nn * 16, (s->Nn - 1) * s->nx[0] * (s->nx[0] + s->nu[0]));
        hipLaunchKernelGGL(k_pack_persist, dim3((n + 255) / 256), dim3(256), 0, st, s->Nn, s->Np, s->nx[0], s->nu[0], D, s->pab, s->pcst); launches++;
        s->need_pack = false;
        s->stream_pending = true;
    }
    if (batch_seq) { launches++; return TQGPU_OK; }          /* the caller launches the batch */
    if (s->pshard) {
        /* this rank's share of the workgroups; the geometry (tiers, global workgroup numbers, G) is the whole tree's */
        PGeom Gm = s->geom;
        Gm.wg_of_block = s->ps_wg_map;
        switch (s->fast) {
#define X(idx, nx, nu, md) case idx: if (s->ps_sys) hipLaunchKernelGGL((f_persist_sh<nx, nu, md, 1>), dim3(s->ps_G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, Gm, s->psync, prologue); \
                                    else hipLaunchKernelGGL((f_persist_sh<nx, nu, md, 0>), dim3(s->ps_G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, Gm, s->psync, prologue); break;
            SHARD_TABLE(X)
#undef X
            default: return fail(TQGPU_EUNSUPPORTED, "the sharded persistent kernel is not instantiated for this shape");
        }
        launches++;
        return TQGPU_OK;
    }
    if (!s->mstage) {
        switch (s->fast) {
#define X(idx, nx, nu, md) case idx: if (O.reuse) hipLaunchKernelGGL((f_persist<nx, nu, md, true>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); \
                                    else if (s->persist_one && !s->in_batch) hipLaunchKernelGGL((f_persist_one<nx, nu, md>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); \
                                    else hipLaunchKernelGGL((f_persist<nx, nu, md, false>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); break;
            FAST_TABLE(X)
#undef X
            default: break;
        }
    } else {
        switch (s->fast) {
#define X(idx, nx, nu, md) case idx: if (O.reuse) hipLaunchKernelGGL((f_mpersist<nx, nu, md, true>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); \
                                    else if (s->persist_one && !s->in_batch) hipLaunchKernelGGL((f_mpersist_one<nx, nu, md>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); \
                                    else hipLaunchKernelGGL((f_mpersist<nx, nu, md, false>), dim3(s->geom.G), dim3(FW * WAVE), s->lds_persist, st, s->pconst, O, s->geom, s->psync, prologue); break;
            MSTAGE_TABLE(X)
#undef X
            default: break;
        }
    }
    launches++;
    return TQGPU_OK;
}

/* one Newton iteration on the generic path = its termination test (gradient + check) and the rest (Newton system, step,
 * first trial); `parts` bit 0 / bit 1 select them.  The host enqueues the iteration it expects to be the last one
 * (warm: as many as the previous solve needed) without the rest: ~17 launches that would only find `done` set. */
void launch_generic_iteration(tqgpu_solver *s, const Opts &O, int h, int &launches, int parts = 3, bool phases = false, bool last = false) {
    const Tree &T = s->T; const Data &D = s->D; hipStream_t st = s->stream;
    auto mark = [&](int i) { if (phases && (size_t)(4 * h + i) < s->phase_ev.size()) (void)hipEventRecord(s->phase_ev[(size_t)(4 * h + i)], st); };
    if (s->w3_now) {
        /* three launches: the termination test of this iteration was the tail of the previous launch of k_sg */
        if (!(parts & 2)) return;
        s->bw_epoch++;
        if (s->bw_epoch == 0) s->bw_epoch = 1;
        s->w3_tail_sg = false;
        hipLaunchKernelGGL(k_hf_w, dim3(T.Np), dim3(WT), s->lds_hf_w, st, T, D, O, s->sch_words, s->sch_rs, s->bw_epoch, h); launches++;
        const int kpred0 = h < (int)s->ls_pred.size() ? std::min(s->ls_pred[(size_t)h], O.lsMaxIter) : 1;
        if (T.Np > 1 && s->w3_merge) {
            /* forward sweep + first trial: one launch */
            s->w3_post_next = last && kpred0 < 2;
            launch_sg(s, O, 2, h, 1); launches++;
            for (int tt = 2; tt <= kpred0; tt++) { s->w3_post_next = last && tt == kpred0; launch_sg(s, O, 1, h, tt); launches++; }
            return;
        }
        if (T.Np > 1) {
            s->fw_epoch++;
            if (s->fw_epoch == 0) s->fw_epoch = 1;
            if (s->d_anc) hipLaunchKernelGGL(k_fwd3c, dim3((T.Np - 1 + SG_WAVES - 1) / SG_WAVES), dim3(SG_WAVES * WAVE), 0, st, T, D, next_w3(s), s->d_anc, h);
            else hipLaunchKernelGGL(k_fwd3, dim3((T.Np - 1 + SG_WAVES - 1) / SG_WAVES), dim3(SG_WAVES * WAVE), 0, st, T, D, next_w3(s), s->fw_words, s->fw_epoch, h);
            launches++;
        } else { hipLaunchKernelGGL(k_ls_begin, dim3(1), dim3(256), 0, st, T, D, h); launches++; }
        const int kpred = h < (int)s->ls_pred.size() ? std::min(s->ls_pred[(size_t)h], O.lsMaxIter) : 1;
        s->w3_post_next = last && kpred < 2;
        launch_sg(s, O, 1, h, 1); launches++;
        for (int tt = 2; tt <= kpred; tt++) { s->w3_post_next = last && tt == kpred; launch_sg(s, O, 1, h, tt); launches++; }
        return;
    }
    mark(0);
    if (parts & 1) {
        if (s->fuse_now) { hipLaunchKernelGGL(k_grad_f, dim3(T.Nn - 1), dim3(WAVE), 0, st, T, D, O, next_fuse(s), h); launches++; }      /* with k_check as its tail */
        else {
            hipLaunchKernelGGL(k_grad, dim3(T.Nn - 1), dim3(WAVE), 0, st, T, D, O.termCondition, h); launches++;
            hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, st, T, D, O, h); launches++;
        }
    }
    if (!(parts & 2)) return;
    const bool wide = s->wide && !s->dense;
    bool ls_begun = false;           /* the forward sweep's last workgroup has done k_ls_begin's work */
    if (wide) hipLaunchKernelGGL(k_hess_w, dim3(T.Np), dim3(WT), s->lds_hess_w, st, T, D, h);
    else hipLaunchKernelGGL(k_hess, dim3(T.Np), dim3(WAVE), s->lds_hess, st, T, D, h);
    launches++;
    mark(1);
    if (s->sch_words && s->bw_fused && !phases) {
        /* all levels in one launch, last block first: a block waits for its children's Schur records inside the kernel */
        s->bw_epoch++;
        if (s->bw_epoch == 0) s->bw_epoch = 1;
        if (wide) hipLaunchKernelGGL(k_factor_all_w, dim3(T.Np), dim3(WT), s->lds_factor_w, st, T, D, O, s->sch_words, s->sch_rs, s->bw_epoch, h);
        else hipLaunchKernelGGL(k_factor_all, dim3(T.Np), dim3(WAVE), s->lds_factor, st, T, D, O, s->sch_words, s->sch_rs, s->bw_epoch, h);
        launches++;
    } else
    for (int lvl = T.Nh - 1; lvl >= 0; lvl--) {
        const int first = s->lvl_first[lvl], count = s->lvl_first[lvl + 1] - first;
        if (wide) hipLaunchKernelGGL(k_factor_w, dim3(count), dim3(WT), s->lds_factor_w, st, T, D, O, first, h);
        else hipLaunchKernelGGL(k_factor, dim3(count), dim3(WAVE), s->lds_factor, st, T, D, O, first, h);
        launches++;
    }
    if (s->fw_words && s->fw_fused && !phases) {
        /* all levels below the root in one launch: a block waits for its parent's step inside the kernel */
        if (T.Np > 1) {
            s->fw_epoch++;
            if (s->fw_epoch == 0) s->fw_epoch = 1;
            const Fuse F = s->fuse_now ? next_fuse(s) : no_fuse();          /* with k_ls_begin as its tail */
            ls_begun = s->fuse_now;
            if (wide) hipLaunchKernelGGL(k_forward_all_w, dim3(T.Np - 1), dim3(WT), s->lds_forward_w, st, T, D, s->fw_words, s->fw_epoch, h, F);
            else hipLaunchKernelGGL(k_forward_all, dim3(T.Np - 1), dim3(WAVE), s->lds_forward, st, T, D, s->fw_words, s->fw_epoch, h, F);
            launches++;
        }
    } else
    for (int lvl = 1; lvl < T.Nh; lvl++) {
        const int first = s->lvl_first[lvl], count = s->lvl_first[lvl + 1] - first;
        if (wide) hipLaunchKernelGGL(k_forward_w, dim3(count), dim3(WT), s->lds_forward_w, st, T, D, first, h);
        else hipLaunchKernelGGL(k_forward, dim3(count), dim3(WAVE), s->lds_forward, st, T, D, first, h);
        launches++;
    }
    mark(2);
    if (!ls_begun) { hipLaunchKernelGGL(k_ls_begin, dim3(1), dim3(256), 0, st, T, D, h); launches++; }
    if (s->fuse_now) { hipLaunchKernelGGL(k_stage_f, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, O, next_fuse(s), 1, h, 1); launches++; }      /* with k_ls_decide as its tail */
    else {
        hipLaunchKernelGGL(k_stage, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, 1, h, 1); launches++;
        hipLaunchKernelGGL(k_ls_decide, dim3(1), dim3(256), 0, st, T, D, O, h, 1, 0); launches++;
    }
    mark(3);
}

/* solution export in one piece: out = [x | u | lam | dlam | mu_x | mu_u] (x and mu_x without the phantom root
 * states), mu = Q .* (xUnc - x) (export_mu, clipping.c:386-399) */
__global__ void k_export_all(int n_x, int n_u, int n_lam, int x_pad, int nx0, Data D, const double *lamc, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double *ox = out, *ou = ox + n_x, *ol = ou + n_u, *od = ol + n_lam, *omx = od + n_lam, *omu = omx + n_x;
    /* MAXIMUM_ITERATIONS_REACHED with at least one iteration done: x, u are the last trial's, the unclipped values phase S's (see Data) */
    const bool maxit = D.ctrl->status == 1 && D.ctrl->iter > 0;
    if (i < n_x) {
        const int j = i + x_pad;
        ox[i] = D.x[j];
        omx[i] = D.Qd[j] * fma(-1.0, D.x[j], (maxit ? D.xUncS : D.xUnc)[j]);      /* dense nodes: Qd = 0 (no bounds, no multipliers) */
    }
    if (i < n_u) { ou[i] = D.u[i]; omu[i] = D.Rd[i] * fma(-1.0, D.u[i], (maxit ? D.uUncS : D.uUnc)[i]); }
    if (i < n_lam) {
        const double *lc = lamc ? lamc : (D.ctrl->cur ? D.lam1 : D.lam0);      /* (enqueued behind a solve that is still running: the device knows) */
        ol[i] = lc[nx0 + i]; od[i] = D.dlam[nx0 + i];
    }
}

}  // namespace

extern "C" int tqgpu_create(tqgpu_solver **out, int device, int Nn, const int *nk, const int *nx, const int *nu) {
    if (!out || Nn < 2 || !nk || !nx || !nu) return fail(TQGPU_EINVAL, "tqgpu_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(TQGPU_ENODEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + "); the tdunes hot path has no CPU fallback");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(TQGPU_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    tqgpu_solver *s = new tqgpu_solver();
    { static std::atomic<unsigned long> next_uid{1}; s->uid = next_uid.fetch_add(1); }
    s->device = device; s->Nn = Nn;
    s->nk.assign(nk, nk + Nn); s->nx.assign(nx, nx + Nn); s->nu.assign(nu, nu + Nn);
    int rc = build_tables(s);
    if (rc != TQGPU_OK) { delete s; return rc; }
    detect_fast(s);
    if (s->fast < 0) detect_multistage(s);
    if (s->fast < 0 && nx[0] == 0 && Nn > 1 && nx[1] > 0 && !(getenv("TREEQP_AMD_PATH") && strcmp(getenv("TREEQP_AMD_PATH"), "generic") == 0)) {
        /* x0 eliminated (tree_qp_in_eliminate_x0: nx[0] = 0, the form every MPC caller solves).  If the tree would be
         * uniform / multistage with a full root node, give the root nx phantom states pinned to zero (bounds [0, 0],
         * unit weight, no linear term, zero A columns for the root's children): the same QP, and it takes the
         * persistent path.  The phantom entries sit in front of every x-sized array (A: in front of the first edges)
         * and never leave the device: the setters / getters below skip them. */
        s->nx[0] = nx[1];
        if (build_tables(s) == TQGPU_OK) { detect_fast(s); if (s->fast < 0) detect_multistage(s); }
        if (s->fast >= 0) { s->x_pad = nx[1]; s->A_pad = nk[0] * nx[1] * nx[1]; }
        else { s->nx[0] = 0; s->fast = -1; s->mstage = false; if ((rc = build_tables(s)) != TQGPU_OK) { delete s; return rc; } }
    }
    {
        const char *env = getenv("TREEQP_AMD_PATH");
        if (env && strcmp(env, "generic") == 0) { s->use_fast = 0; s->use_gpersist = 0; }
        if (env && strcmp(env, "tiered") == 0) s->use_persist = 0;
        /* reference-order sums (strict_sum / strict_block_dots): the launch-per-phase kernels and the single-workgroup kernel carry them;
         * the fused tails, the three-launch family and the persistent / tiered kernels (their partial sums are per workgroup by
         * construction) stand aside */
        s->strict_sum = getenv("TREEQP_AMD_STRICT_SUM") && atoi(getenv("TREEQP_AMD_STRICT_SUM")) != 0;
        if (s->strict_sum) { s->use_fast = 0; s->wide = false; }          /* (the MFMA kernels of the wide-block class sum in tile order) */
        s->use_persist_orig = s->use_persist;
        const char *ch = getenv("TREEQP_AMD_CHUNK");
        if (ch && atoi(ch) > 0) s->chunk = atoi(ch);
    }

    /* one slab for everything */
    Carver cv;
    const size_t I = sizeof(int), Dbl = sizeof(double);
    const size_t o_dad = cv.take(Nn * I), o_nk = cv.take(Nn * I), o_kid0 = cv.take(Nn * I), o_nx = cv.take(Nn * I), o_nu = cv.take(Nn * I);
    const size_t o_xoff = cv.take((Nn + 1) * I), o_uoff = cv.take((Nn + 1) * I), o_aoff = cv.take((Nn + 1) * I), o_boff = cv.take((Nn + 1) * I);
    const size_t o_pos = cv.take(Nn * I), o_bdim = cv.take(Nn * I), o_woff = cv.take((Nn + 1) * I), o_utoff = cv.take((Nn + 1) * I);
    const size_t SX = s->sum_nx, SU = s->sum_nu;
    const size_t o_A = cv.take(s->sum_A * Dbl), o_B = cv.take(s->sum_B * Dbl), o_b = cv.take(SX * Dbl);
    const size_t o_Qd = cv.take(SX * Dbl), o_q = cv.take(SX * Dbl), o_xmin = cv.take(SX * Dbl), o_xmax = cv.take(SX * Dbl);
    const size_t o_Rd = cv.take(SU * Dbl), o_r = cv.take(SU * Dbl), o_umin = cv.take(SU * Dbl), o_umax = cv.take(SU * Dbl);
    const size_t o_Qinv = cv.take(SX * Dbl), o_Rinv = cv.take(SU * Dbl);
    const size_t o_qmod = cv.take(SX * Dbl), o_x = cv.take(SX * Dbl), o_xUnc = cv.take(SX * Dbl), o_Qcal = cv.take(SX * Dbl);
    const size_t o_rmod = cv.take(SU * Dbl), o_u = cv.take(SU * Dbl), o_uUnc = cv.take(SU * Dbl), o_Rcal = cv.take(SU * Dbl);
    const size_t o_xUncS = cv.take(SX * Dbl), o_uUncS = cv.take(SU * Dbl);
    const size_t o_lam0 = cv.take(SX * Dbl), o_lam1 = cv.take(SX * Dbl), o_dlam = cv.take(SX * Dbl), o_res = cv.take(SX * Dbl), o_resMod = cv.take(SX * Dbl);
    const size_t o_invd = cv.take(SX * Dbl);
    const size_t o_W = cv.take(s->sum_W * Dbl), o_CW = cv.take(s->sum_W * Dbl), o_Ut = cv.take(s->sum_Ut * Dbl), o_CUt = cv.take(s->sum_Ut * Dbl);
    const size_t o_fval = cv.take(Nn * Dbl), o_perr = cv.take((SX + Nn + 1) * Dbl), o_pdot = cv.take(Nn * Dbl);
    const size_t maxnx = (size_t)*std::max_element(s->nx.begin(), s->nx.end());
    const size_t o_sbuf = cv.take((s->fast >= 0 ? (size_t)Nn * (maxnx * maxnx + maxnx) : 1) * Dbl), o_ybuf = cv.take((SX + 1) * Dbl);
    const size_t o_mux = cv.take(SX * Dbl), o_muu = cv.take(SU * Dbl);
    const size_t o_lami = cv.take(SX * Dbl);
    s->poff.assign(Nn + 1, 0);
    for (int k = 0; k < Nn; k++) s->poff[k + 1] = s->poff[k] + (s->nx[k] + s->nu[k]) * (s->nx[k] + s->nu[k]);
    const size_t o_kind = cv.take(Nn * I);
    const size_t o_poff = cv.take((Nn + 1) * I), o_Hd = cv.take((size_t)std::max(s->poff[Nn], 1) * Dbl), o_Pd = cv.take((size_t)std::max(s->poff[Nn], 1) * Dbl);
    const size_t o_ctrl = cv.take(sizeof(Ctrl));
    const size_t o_stamps = cv.take((8 * 32 * 2 + 1024) * sizeof(unsigned long long));   /* + 1024: placement census of the persistent launch (diagnostic builds) */
    s->ls_log_cap = 4096;
    const size_t o_log = cv.take(s->ls_log_cap * I);
    s->slab_bytes = cv.off + 256;

    auto cleanup_fail = [&](int code) { tqgpu_destroy(s); return code; };
    if (hipMalloc(&s->slab, s->slab_bytes) != hipSuccess) { delete s; return fail(TQGPU_ENOMEM, "hipMalloc failed for the device mirror"); }
    if (hipMemset(s->slab, 0, s->slab_bytes) != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "hipMemset failed"));
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "hipStreamCreate failed"));
    { std::lock_guard<std::mutex> lk(g_streams_mu); g_live_streams.insert(s->stream); }
    s->ring_ev0.assign(EV_RING, nullptr); s->ring_ev1.assign(EV_RING, nullptr); s->ring_ok.assign(EV_RING, 0);
    for (int i = 0; i < EV_RING; i++)
        if (hipEventCreate(&s->ring_ev0[i]) != hipSuccess || hipEventCreate(&s->ring_ev1[i]) != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "hipEventCreate failed"));
    if (hipHostMalloc((void **)&s->h_res, sizeof(HostRes), hipHostMallocDefault) != hipSuccess) return cleanup_fail(fail(TQGPU_ENOMEM, "hipHostMalloc failed"));
    memset(s->h_res, 0, sizeof(HostRes));
    s->h_ctrl = &s->h_res->c;
    if (hipHostMalloc((void **)&s->h_ls_log, s->ls_log_cap * I, hipHostMallocDefault) != hipSuccess) return cleanup_fail(fail(TQGPU_ENOMEM, "hipHostMalloc failed"));

    void *base = s->slab;
#define UP(off, vec) if (hipMemcpy(at<int>(base, off), (vec).data(), (vec).size() * I, hipMemcpyHostToDevice) != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "table upload failed"))
    UP(o_dad, s->dad); UP(o_nk, s->nk); UP(o_kid0, s->kid0); UP(o_nx, s->nx); UP(o_nu, s->nu);
    UP(o_xoff, s->xoff); UP(o_uoff, s->uoff); UP(o_aoff, s->aoff); UP(o_boff, s->boff);
    UP(o_pos, s->pos); UP(o_bdim, s->bdim); UP(o_woff, s->woff); UP(o_utoff, s->utoff);
    UP(o_poff, s->poff);
#undef UP
    Tree &T = s->T;
    T.Nn = Nn; T.Np = s->Np; T.Nh = s->Nh; T.nx0 = s->nx0;
    T.dad = at<int>(base, o_dad); T.nk = at<int>(base, o_nk); T.kid0 = at<int>(base, o_kid0);
    T.nx = at<int>(base, o_nx); T.nu = at<int>(base, o_nu); T.xoff = at<int>(base, o_xoff); T.uoff = at<int>(base, o_uoff);
    T.aoff = at<int>(base, o_aoff); T.boff = at<int>(base, o_boff); T.pos = at<int>(base, o_pos);
    T.bdim = at<int>(base, o_bdim); T.woff = at<int>(base, o_woff); T.utoff = at<int>(base, o_utoff);
    {
        std::vector<int> desc((size_t)DESC_INTS * Nn, 0);
        for (int k = 0; k < Nn; k++) {
            int *e = &desc[(size_t)DESC_INTS * k];
            const int k0 = s->nk[k] > 0 ? s->kid0[k] : 0, dd = k > 0 ? s->dad[k] : 0;
            e[0] = s->bdim[k]; e[1] = s->nx[k]; e[2] = s->nu[k]; e[3] = s->nk[k]; e[4] = k0; e[5] = s->xoff[k]; e[6] = s->uoff[k];
            e[7] = s->xoff[k0]; e[8] = s->woff[k]; e[9] = s->utoff[k]; e[10] = dd; e[11] = s->pos[k]; e[12] = s->bdim[dd]; e[13] = s->woff[dd];
            for (int u = 0; u < 4 && u < s->nk[k]; u++) { e[16 + 3 * u] = s->nx[k0 + u]; e[17 + 3 * u] = s->aoff[k0 + u]; e[18 + 3 * u] = s->boff[k0 + u]; }
        }
        if (hipMalloc(&s->d_desc, desc.size() * I) != hipSuccess || hipMemcpy(s->d_desc, desc.data(), desc.size() * I, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the node records"));
        T.desc = s->d_desc;
    }
    Data &D = s->D;
    s->A = at<double>(base, o_A); s->B = at<double>(base, o_B); s->b = at<double>(base, o_b);
    s->Qd = at<double>(base, o_Qd); s->Rd = at<double>(base, o_Rd); s->q = at<double>(base, o_q); s->r = at<double>(base, o_r);
    s->xmin = at<double>(base, o_xmin); s->xmax = at<double>(base, o_xmax); s->umin = at<double>(base, o_umin); s->umax = at<double>(base, o_umax);
    D.A = s->A; D.B = s->B; D.b = s->b; D.Qd = s->Qd; D.Rd = s->Rd; D.q = s->q; D.r = s->r;
    D.xmin = s->xmin; D.xmax = s->xmax; D.umin = s->umin; D.umax = s->umax;
    D.Qinv = at<double>(base, o_Qinv); D.Rinv = at<double>(base, o_Rinv);
    D.qmod = at<double>(base, o_qmod); D.rmod = at<double>(base, o_rmod); D.x = at<double>(base, o_x); D.u = at<double>(base, o_u);
    D.xUnc = at<double>(base, o_xUnc); D.uUnc = at<double>(base, o_uUnc); D.QinvCal = at<double>(base, o_Qcal); D.RinvCal = at<double>(base, o_Rcal);
    D.xUncS = at<double>(base, o_xUncS); D.uUncS = at<double>(base, o_uUncS);
    D.lam0 = at<double>(base, o_lam0); D.lam1 = at<double>(base, o_lam1); D.dlam = at<double>(base, o_dlam);
    D.res = at<double>(base, o_res); D.resMod = at<double>(base, o_resMod); D.invd = at<double>(base, o_invd);
    D.W = at<double>(base, o_W); D.CholW = at<double>(base, o_CW); D.Ut = at<double>(base, o_Ut); D.CholUt = at<double>(base, o_CUt);
    D.fval = at<double>(base, o_fval); D.part_err = at<double>(base, o_perr); D.part_dot = at<double>(base, o_pdot);
    D.Sbuf = at<double>(base, o_sbuf); D.ybuf = at<double>(base, o_ybuf);
    D.stamps = at<unsigned long long>(base, o_stamps);
    D.ctrl = at<Ctrl>(base, o_ctrl); D.ls_log = at<int>(base, o_log); D.ls_log_cap = s->ls_log_cap;
    D.strict = s->strict_sum ? 1 : 0;
    D.dense = 0; s->d_kind = at<int>(base, o_kind); D.kind = s->d_kind; s->d_Hd = at<double>(base, o_Hd); D.Hd = s->d_Hd; D.Pd = at<double>(base, o_Pd); D.poff = at<int>(base, o_poff);
    s->use_fast_orig = s->use_fast;
    s->d_mu_x = at<double>(base, o_mux); s->d_mu_u = at<double>(base, o_muu);
    s->d_lam_init = at<double>(base, o_lami);
    s->in_off0 = o_A; s->in_bytes = o_Qinv - o_A;
    s->out_doubles = 2 * (size_t)(s->sum_nx - s->x_pad) + 2 * (size_t)s->sum_nu + 2 * (size_t)s->sum_lam;
    if (hipHostMalloc((void **)&s->h_in, std::max<size_t>(s->in_bytes, 8), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&s->h_lam, sizeof(double) * (size_t)std::max(s->sum_nx, 1), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&s->h_out, sizeof(double) * std::max<size_t>(s->out_doubles, 1), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&s->d_out, sizeof(double) * std::max<size_t>(s->out_doubles, 1)) != hipSuccess)
        return cleanup_fail(fail(TQGPU_ENOMEM, "allocation of the host mirrors failed"));
    if (s->x_pad) {
        /* phantom root states: weight 1, everything else zero (the slab is zeroed) */
        std::vector<double> ones((size_t)s->x_pad, 1.0);
        if (hipMemcpy(s->Qd, ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice) != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "hipMemcpy failed"));
    }

    if ((rc = allow_lds(k_stage, s->lds_stage)) || (rc = allow_lds(k_hess, s->lds_hess)) ||
        (rc = allow_lds(k_factor, s->lds_factor)) || (rc = allow_lds(k_forward, s->lds_forward)) ||
        (rc = allow_lds(k_factor_all, s->lds_factor)) || (rc = allow_lds(k_forward_all, s->lds_forward)) || (rc = allow_lds(k_stage_f, s->lds_stage)))
        return cleanup_fail(rc);
    if (s->wide && ((rc = allow_lds(k_hess_w, s->lds_hess_w)) || (rc = allow_lds(k_factor_w, s->lds_factor_w)) || (rc = allow_lds(k_forward_w, s->lds_forward_w)) || (rc = allow_lds(k_forward_all_w, s->lds_forward_w)) || (rc = allow_lds(k_factor_all_w, s->lds_factor_w))))
        return cleanup_fail(rc);
    if (s->fast >= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.maxThreadsPerBlock < FW * WAVE) s->fast = -1;
    }
    {
        /* hand-over words of the fused forward / backward sweeps of the launch-per-phase path (zeroed once; every launch brings its own tag) */
        const size_t bytes = sizeof(unsigned long long) * 2 * (size_t)std::max(s->sum_nx, 1);
        if (hipMalloc(&s->fw_words, bytes) != hipSuccess || hipMemset(s->fw_words, 0, bytes) != hipSuccess)
            return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the forward hand-over words"));
        const char *m = getenv("TREEQP_AMD_FWD");              /* =levels: one launch per tree level (the round-1 protocol) */
        s->fw_fused = !(m && strcmp(m, "levels") == 0) && !s->strict_sum;      /* (strict: the reference's launch-per-level order of operations) */
        int nxmax = 0;
        for (int k = 0; k < s->Nn; k++) nxmax = std::max(nxmax, s->nx[k]);
        s->sch_rs = (nxmax + 1) * (nxmax + 1);
        const size_t sbytes = sizeof(unsigned long long) * 2 * (size_t)s->sch_rs * (size_t)s->Nn;
        if (hipMalloc(&s->sch_words, sbytes) != hipSuccess || hipMemset(s->sch_words, 0, sbytes) != hipSuccess)
            return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the Schur hand-over words"));
        m = getenv("TREEQP_AMD_BWD");
        s->bw_fused = !(m && strcmp(m, "levels") == 0) && !s->strict_sum;      /* (a fused sweep subtracts a child's Schur record AFTER an ALWAYS shift of the diagonal, the reference before it) */
        if (s->Nn <= FUSE_MAX && s->Nn >= 2 && !getenv("TREEQP_AMD_NO_FUSE") && !s->strict_sum) {
            const size_t rb = sizeof(unsigned long long) * 2 * (size_t)s->Nn;
            if (hipMalloc(&s->fuse_red, rb) != hipSuccess || hipMemset(s->fuse_red, 0, rb) != hipSuccess ||
                hipMalloc(&s->fuse_cnt, 4 * sizeof(int)) != hipSuccess || hipMemset(s->fuse_cnt, 0, 4 * sizeof(int)) != hipSuccess)
                return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the fused reductions"));
            s->fuse_ok = true;
        }
    }
    if (s->wide && s->fw_fused && s->bw_fused && !getenv("TREEQP_AMD_NO_WIDE3") && !s->strict_sum) {      /* (TREEQP_AMD_FWD / BWD = levels ask for the launch-per-level kernels) */
        /* the three-launch family of the wide-block class (tdunes_wide3.hpp) */
        int nxmax = 0;
        for (int k = 0; k < Nn; k++) { nxmax = std::max(nxmax, s->nx[k]); if (k < s->Np) s->lds_hf_w = std::max(s->lds_hf_w, wide3_lds(s->bdim[k], k > 0 ? s->nx[k] : 0, s->nx[k] + s->nu[k])); }
        const size_t groups = std::max((size_t)(Nn + SG_WAVES - 1) / SG_WAVES, (size_t)s->Np);      /* workgroups of k_sg / k_sgp / k_fwd3 */
        const size_t xb = sizeof(unsigned long long) * 2 * (size_t)std::max(s->sum_nx + s->sum_nu, 1), rb = sizeof(unsigned long long) * 4 * groups;
        if (nxmax <= 32 && s->lds_hf_w <= 160 * 1024 && SG_WAVES * s->lds_stage <= 160 * 1024) {
            if (hipMalloc(&s->w3_xu, xb) != hipSuccess || hipMemset(s->w3_xu, 0, xb) != hipSuccess ||
                hipMalloc(&s->w3_red, rb) != hipSuccess || hipMemset(s->w3_red, 0, rb) != hipSuccess ||
                hipMalloc(&s->w3_cnt, 4 * sizeof(int)) != hipSuccess || hipMemset(s->w3_cnt, 0, 4 * sizeof(int)) != hipSuccess)
                return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the three-launch hand-over words"));
            if ((rc = allow_lds(k_hf_w, s->lds_hf_w)) || (rc = allow_lds(k_sg, SG_WAVES * ((s->lds_stage + 7) / 8) * 8))) return cleanup_fail(rc);
            /* k_sgp: one entry of a node's [x | u] per lane, one row of a block per lane */
            bool fits = true;
            for (int k = 0; k < Nn; k++) fits = fits && s->nx[k] + s->nu[k] <= 64;
            for (int k = 0; k < s->Np; k++) s->sgp_accs = std::max(s->sgp_accs, std::min(s->nk[k], 8) * (s->nx[k] + s->nu[k]));      /* the children's terms in LDS: up to 8 children (more: taken by wave 0 on its own) */
            for (int k = 0; k < s->Np; k++) s->lds_sgp = std::max(s->lds_sgp, wide3_lds_sgp(s->bdim[k], s->nx[k] + s->nu[k], s->sgp_accs));
            s->w3_sgp = fits && s->lds_sgp <= 64 * 1024 && !getenv("TREEQP_AMD_NO_SGP");
            /* forward sweep without hand-overs (k_fwd3c): small nodes, short paths, blocks whose region of CholW has room for the copy of z0
             * beside the diagnostic stamps */
            {
                bool okc = s->Np > 1 && !getenv("TREEQP_AMD_NO_FWD_CHAIN");
                for (int k = 0; k < Nn && okc; k++) okc = s->nx[k] <= 8;
                for (int k = 0; k < s->Np && okc; k++) okc = s->bdim[k] >= 3 && s->bdim[k] <= 64;
                std::vector<int> anc;
                if (okc) {
                    anc.assign((size_t)FWDC_INTS * s->Np, 0);
                    for (int ii = 1; ii < s->Np && okc; ii++) {
                        std::vector<int> path;                      /* ii, dad(ii), .., the root's child */
                        for (int n = ii; n != 0; n = s->dad[n]) path.push_back(n);
                        const int L = (int)path.size();
                        if (L > 16) { okc = false; break; }
                        int *a = &anc[(size_t)FWDC_INTS * ii];
                        a[0] = L;
                        for (int k = 0; k < L; k++) {
                            const int node = path[(size_t)(L - 1 - k)], prev = s->dad[node], dp_ = s->bdim[prev];
                            a[1 + 4 * k + 0] = s->woff[prev] + dp_ * dp_ - dp_ + s->pos[node];
                            a[1 + 4 * k + 1] = s->utoff[prev] + s->pos[node];
                            a[1 + 4 * k + 2] = dp_;
                            a[1 + 4 * k + 3] = (prev == 0 ? 0 : s->nx[prev]) | (s->nx[node] << 8);
                        }
                    }
                }
                if (okc && (hipMalloc(&s->d_anc, anc.size() * sizeof(int)) != hipSuccess || hipMemcpy(s->d_anc, anc.data(), anc.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
                    return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the path table of the forward sweep"));
                if (okc && s->w3_sgp && !getenv("TREEQP_AMD_NO_FWD_MERGE")) {
                    const size_t pb = sizeof(unsigned long long) * 2 * (size_t)s->Np;
                    if (hipMalloc(&s->d_pdw, pb) != hipSuccess || hipMemset(s->d_pdw, 0, pb) != hipSuccess)
                        return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed for the forward sweep's partial sums"));
                    s->w3_merge = true;
                }
            }
            s->w3_ok = true;
        }
    }
    if ((rc = setup_persist(s, device))) return cleanup_fail(rc);
    {
        /* single-workgroup persistent kernel for small trees of any shape: every level must be a few rounds
         * of GP_WAVES blocks at most, and the per-wave LDS windows must fit the default 64 KB */
        int widest = 0;
        for (int l = 0; l <= s->Nh; l++) widest = std::max(widest, s->lvl_first[l + 1] - s->lvl_first[l]);
        const size_t per_wave = (std::max(std::max(s->lds_stage, s->lds_hess), std::max(s->lds_factor, s->lds_forward)) + 7) / 8 + 2;
        s->lds_gp_wave = per_wave;
        {
            /* four nodes per wave in the stage sweep: nx + nu <= 16 everywhere, clipping nodes only, and a quarter of the wave's window
             * holds a node's duals (bdim + nx doubles) */
            bool ok16 = !s->dense && !getenv("TREEQP_AMD_NO_STAGE16");
            for (int k = 0; k < Nn && ok16; k++) ok16 = s->nx[k] + s->nu[k] <= 16 && (size_t)(s->bdim[k] + s->nx[k]) <= per_wave / 4;
            s->gp_small16 = ok16;
            bool ok8 = !getenv("TREEQP_AMD_NO_STAGE16") && !s->strict_sum;          /* eight nodes per wave in the gradient sweep: nx <= 8 everywhere */
            for (int k = 0; k < Nn && ok8; k++) ok8 = s->nx[k] <= 8;
            s->gp_small8 = ok8;
        }
        s->gpersist_ok = widest <= 6 * GP_WAVES && per_wave * 8 * GP_WAVES <= 150 * 1024;
        /* LDS mirror of the mutable state (tdunes_gpersist.hpp): 16 node-sized vectors (rounded up to even), 4 block arrays */
        {
            auto ev = [](size_t n) { return (n + 1) & ~(size_t)1; };
            const size_t sx = (size_t)s->sum_nx, su = (size_t)s->sum_nu;
            const size_t mirror = 11 * ev(sx) + 5 * ev(su) + 2 * ev((size_t)s->sum_W) + 2 * ev((size_t)s->sum_Ut) + 2 * ev((size_t)Nn) + ev(sx + Nn + 1) + ev(su) * 0;
            const size_t tables = (13 * ((size_t)Nn + 3)) / 2 + 16;            /* int tables, in doubles */
            s->lds_gp_total = (per_wave * GP_WAVES + mirror + tables + 8) * 8;
            s->gp_in_lds = s->gpersist_ok && s->lds_gp_total <= 150 * 1024;
            const size_t consts = (ev((size_t)s->sum_A) + ev((size_t)s->sum_B) + 5 * ev(sx) + 4 * ev(su)) * 8;
            s->gp_const_in_lds = s->gp_in_lds && s->lds_gp_total + consts <= 150 * 1024;
            if (s->gp_const_in_lds) s->lds_gp_total += consts;
            if (!s->gp_in_lds) {
                /* the state does not fit: the index tables alone, next to the per-wave windows, if THEY fit */
                s->gp_tab_in_lds = s->gpersist_ok && (per_wave * GP_WAVES + tables + 8) * 8 <= 150 * 1024;
                s->lds_gp_total = (per_wave * GP_WAVES + (s->gp_tab_in_lds ? tables + 8 : 0)) * 8;
            }
            else if (s->lds_gp_total > 64 * 1024 &&
                     hipFuncSetAttribute(reinterpret_cast<const void *>(g_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_gp_total) != hipSuccess) {
                (void)hipGetLastError();
                s->gp_in_lds = false; s->gp_tab_in_lds = false; s->lds_gp_total = per_wave * GP_WAVES * 8;
            }
            if (s->gpersist_ok && !s->gp_in_lds && s->lds_gp_total > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(g_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_gp_total) != hipSuccess) {
                (void)hipGetLastError();
                s->gpersist_ok = false;
            }
        }
        if (s->gpersist_ok) {
            /* behind the level table: for every level of parents, how many of its blocks one wave takes side by side in the backward /
             * forward / Hessian sweeps (factor_body_g ...): levels wider than the workgroup's 16 waves whose blocks all have the same
             * small dimensions (the chains of a pruned scenario tree); 1 = one block per wave */
            std::vector<int> tab(s->lvl_first);
            const bool no_grp = getenv("TREEQP_AMD_GP_NO_GROUPS") != nullptr;
            for (int lvl = 0; lvl <= s->Nh; lvl++) {
                int grp = 1;
                const int first = s->lvl_first[lvl], count = s->lvl_first[lvl + 1] - first;
                if (!no_grp && lvl >= 1 && lvl < s->Nh && count > GP_WAVES && !s->dense) {
                    const int d0 = s->bdim[first], n0 = s->nx[first], u0 = s->nu[first], c0 = s->nk[first];
                    bool same = true;
                    for (int k = first; k < first + count && same; k++) {
                        same = s->bdim[k] == d0 && s->nx[k] == n0 && s->nu[k] == u0 && s->nk[k] == c0;
                        for (int cc = 0; cc < s->nk[k] && same; cc++) same = s->nx[s->kid0[k] + cc] == s->nx[s->kid0[first] + cc];
                    }
                    const int R = d0 + 1 + n0;
                    /* doubles of a group's window: factor_body's tall matrix, forward_body's factor + vectors, hess_body's C and C P */
                    const size_t need = std::max(std::max((size_t)(R | 1) * d0 + d0 + 2, (size_t)(d0 | 1) * d0 + 2 * d0 + n0 + 2), (size_t)2 * d0 * (n0 + u0) + 2);
                    if (same && d0 >= 1) {
                        if (R <= WAVE / 3 && 3 * need <= s->lds_gp_wave) grp = 3;
                        else if (R <= WAVE / 2 && 2 * need <= s->lds_gp_wave) grp = 2;
                    }
                }
                tab.push_back(grp);
            }
            if (hipMalloc(&s->d_lvl_first, sizeof(int) * tab.size()) != hipSuccess ||
                hipMemcpy(s->d_lvl_first, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) != hipSuccess)
                return cleanup_fail(fail(TQGPU_ENOMEM, "hipMalloc failed"));
        }
    }
    /* the zeroing of the slabs above went out as device memsets, which may return before they have happened, on the null stream, which
     * the solver's non-blocking stream does not wait for: everything is in place before the first solve can be enqueued */
    if (hipDeviceSynchronize() != hipSuccess) return cleanup_fail(fail(TQGPU_ENODEVICE, "hipDeviceSynchronize failed"));
    *out = s;
    return TQGPU_OK;
}

extern "C" void tqgpu_destroy(tqgpu_solver *s) {
    (void)settle(s);
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &ev : s->iter_ev) (void)hipEventDestroy(ev);
    for (auto &ev : s->phase_ev) (void)hipEventDestroy(ev);
    if (s->sweep_ev0) (void)hipEventDestroy(s->sweep_ev0);
    if (s->sweep_ev1) (void)hipEventDestroy(s->sweep_ev1);
    for (auto &ev : s->ring_ev0) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : s->ring_ev1) if (ev) (void)hipEventDestroy(ev);
    if (s->stream) { { std::lock_guard<std::mutex> lk(g_streams_mu); g_live_streams.erase(s->stream); } (void)hipStreamDestroy(s->stream); }
    if (s->h_res) (void)hipHostFree(s->h_res);
    if (s->h_ls_log) (void)hipHostFree(s->h_ls_log);
    if (s->shard_slab) (void)hipFree(s->shard_slab);
    if (s->sync_slab) (void)hipFree(s->sync_slab);
    if (s->pconst_slab) (void)hipFree(s->pconst_slab);
    if (s->wg_map) (void)hipFree(s->wg_map);
    if (s->d_desc) (void)hipFree(s->d_desc);
    if (s->d_pitems) (void)hipFree(s->d_pitems);
    if (s->h_pitems) (void)hipHostFree(s->h_pitems);
    if (s->ps_wg_map) (void)hipFree(s->ps_wg_map);
    if (s->d_peers) (void)hipFree(s->d_peers);
    for (int r = 0; r < 8; r++) if (s->ps_ipc[r]) (void)hipIpcCloseMemHandle(s->ps_ipc[r]);
    if (s->w3_xu) (void)hipFree(s->w3_xu);
    if (s->w3_red) (void)hipFree(s->w3_red);
    if (s->w3_cnt) (void)hipFree(s->w3_cnt);
    if (s->fuse_red) (void)hipFree(s->fuse_red);
    if (s->fuse_cnt) (void)hipFree(s->fuse_cnt);
    if (s->fw_words) (void)hipFree(s->fw_words);
    if (s->sch_words) (void)hipFree(s->sch_words);
    if (s->d_anc) (void)hipFree(s->d_anc);
    if (s->d_pdw) (void)hipFree(s->d_pdw);
    if (s->d_gitems) (void)hipFree(s->d_gitems);
    if (s->h_gitems) (void)hipHostFree(s->h_gitems);
    if (s->batch_ev) (void)hipEventDestroy(s->batch_ev);
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_lam) (void)hipHostFree(s->h_lam);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->d_out) (void)hipFree(s->d_out);
    if (s->d_lvl_first) (void)hipFree(s->d_lvl_first);
    if (s->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s->comm);
    if (s->slab) (void)hipFree(s->slab);
    delete s;
}

/* which path a mirror takes: persistent single launch (uniform or multistage trees), tiered launches (uniform
 * trees only), single-workgroup persistent (small trees of any shape), launch per level */
static bool persist_capable(const tqgpu_solver *s) { return s->fast >= 0 && s->use_fast && s->persist_ok && s->use_persist && !s->sharded; }
static bool tiered_capable(const tqgpu_solver *s) { return s->fast >= 0 && s->use_fast && !s->mstage; }
/* `batch`: as a member of a batched launch (one workgroup per tree) the single-workgroup kernel also takes trees whose
 * state does not fit the LDS mirror; alone, such a tree is faster with one launch per level */
static bool uses_gpersist(const tqgpu_solver *s, bool batch = false) {
    return s->gpersist_ok && (s->gp_in_lds || batch) && s->use_gpersist && !s->dense && !s->sharded && !persist_capable(s) && !tiered_capable(s);
}
extern "C" int tqgpu_uses_fused_path(const tqgpu_solver *s) {
    if (!s) return 0;
    if (persist_capable(s)) return 2;
    if (tiered_capable(s)) return 1;
    return uses_gpersist(s) ? 3 : 0;
}

/* diagnostic: copy the in-kernel time stamps of the last fused iteration (8 kernels x 32 slots x
 * {shader clock, 100 MHz wall clock}); only filled when TREEQP_AMD_STAMPS is set */
/* diagnostic builds (-DTQ_WIDE_STAMPS): per-block time stamps of k_hf_w, four 64-bit words per block kept in the (otherwise unused) CholW array */
extern "C" int tqgpu_debug_block_stamps(tqgpu_solver *s, unsigned long long *out, int cap_blocks) {
    SETTLE(s);
    if (!s || !out) return fail(TQGPU_EINVAL, "bad arguments");
    HIP_TRY(hipStreamSynchronize(s->stream));          /* (the copy below is on the null stream, which the solver's non-blocking stream does not order) */
    std::vector<double> tmp((size_t)std::max(s->sum_W, 1));
    HIP_TRY(hipMemcpy(tmp.data(), getenv("TQ_STAMPS_OF_SGP") ? s->D.W : s->D.CholW, sizeof(double) * (size_t)s->sum_W, hipMemcpyDeviceToHost));
    const int n = std::min(cap_blocks, s->Np);
    for (int k = 0; k < n; k++) memcpy(out + 4 * (size_t)k, tmp.data() + s->woff[k], 4 * sizeof(unsigned long long));
    return n;
}

extern "C" int tqgpu_get_stamps(tqgpu_solver *s, unsigned long long *out, int cap) {
    SETTLE(s);
    if (!s || !out) return fail(TQGPU_EINVAL, "bad arguments");
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int n = std::min(cap, 8 * 32 * 2 + 1024);
    HIP_TRY(hipMemcpy(out, s->D.stamps, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    return TQGPU_OK;
}

extern "C" int tqgpu_dims(const tqgpu_solver *s, int *sum_nx, int *sum_nu, int *sum_lam, int *sum_A, int *sum_B) {
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    if (sum_nx) *sum_nx = s->sum_nx - s->x_pad;
    if (sum_nu) *sum_nu = s->sum_nu;
    if (sum_lam) *sum_lam = s->sum_lam;
    if (sum_A) *sum_A = s->sum_A - s->A_pad;
    if (sum_B) *sum_B = s->sum_B;
    return TQGPU_OK;
}

#define H2D(dst, src, count)                                                                                   \
    do {                                                                                                       \
        if ((src) && (count) > 0)                                                                              \
            HIP_TRY(hipMemcpyAsync((dst), (src), sizeof(double) * (size_t)(count), hipMemcpyHostToDevice, s->stream)); \
    } while (0)

extern "C" int tqgpu_set_dynamics(tqgpu_solver *s, const double *A, const double *B, const double *b) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    s->in_valid = false;
    H2D(s->A + s->A_pad, A, s->sum_A - s->A_pad); H2D(s->B, B, s->sum_B);
    H2D(s->b + s->nx0, b, s->sum_lam);              /* node-indexed on the device: root slot unused */
    HIP_TRY(hipStreamSynchronize(s->stream));       /* the caller may reuse its buffers */
    s->need_pack = true;
    return TQGPU_OK;
}

extern "C" int tqgpu_set_objective_diag(tqgpu_solver *s, const double *Qd, const double *Rd, const double *q, const double *r) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    s->in_valid = false;
    H2D(s->Qd + s->x_pad, Qd, s->sum_nx - s->x_pad); H2D(s->Rd, Rd, s->sum_nu); H2D(s->q + s->x_pad, q, s->sum_nx - s->x_pad); H2D(s->r, r, s->sum_nu);
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->need_init = true;
    s->need_pack = true;
    if (s->dense) { s->dense = false; s->D.dense = 0; s->use_fast = s->use_fast_orig; }
    return TQGPU_OK;
}

/* Dense objective for the dense UNCONSTRAINED stage solver (the reference's qpOASES stage backend restricted
 * to nodes without bounds, dual_Newton_tree_qpoases.c:153-217,401-476).  Flat layout of
 * tree_qp_in_set_ltv_objective_colmajor (tree_qp_common.c): per node Q (nx x nx), R (nu x nu), S (nu x nx),
 * all column major, then q, r.  Selects the generic device path; tqgpu_set_objective_diag selects clipping again. */
extern "C" int tqgpu_set_objective_dense(tqgpu_solver *s, const double *Q, const double *R, const double *S, const double *q, const double *r) {
    SETTLE(s);
    return tqgpu_set_objective_mixed(s, nullptr, Q, R, S, q, r);
}

/* The same with a per-node choice of the stage solver (opts->qp_solver[] of the reference, dual_Newton_tree.c:124-162):
 * kind[k] = 0: clipping (the diagonals of Q_k, R_k are its weights; their off-diagonals and S_k must be zero), 1: dense
 * unconstrained.  kind == NULL: every node dense. */
extern "C" int tqgpu_set_objective_mixed(tqgpu_solver *s, const int *kind, const double *Q, const double *R, const double *S, const double *q, const double *r) {
    SETTLE(s);
    if (!s || !Q || !q) return fail(TQGPU_EINVAL, "tqgpu_set_objective_mixed: bad arguments");
    if (s->sharded) return fail(TQGPU_EINVAL, "the dense stage solver is not available in sharded mode");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<double> H((size_t)std::max(s->poff[s->Nn], 1), 0.0);
    std::vector<double> Qd((size_t)std::max(s->sum_nx, 1), 0.0), Rd((size_t)std::max(s->sum_nu, 1), 0.0);
    std::vector<int> kd((size_t)s->Nn, 1);
    if (kind) for (int k = 0; k < s->Nn; k++) kd[(size_t)k] = kind[k] ? 1 : 0;
    size_t oq = 0, orr = 0, os = 0;
    for (int k = 0; k < s->Nn; k++) {
        const int nu = s->nu[k], nz = s->nx[k] + nu;
        double *Hk = H.data() + s->poff[k];
        if (!kd[(size_t)k]) {
            /* clipping node: diagonal weights (phantom root states of an embedded x0-eliminated tree keep their unit weight) */
            const int nxc = s->nx[k] - ((k == 0) ? s->x_pad : 0);
            for (int i = 0; i < ((k == 0) ? s->x_pad : 0); i++) Qd[(size_t)i] = 1.0;
            for (int i = 0; i < nxc; i++) Qd[(size_t)s->xoff[k] + ((k == 0) ? s->x_pad : 0) + i] = Q[oq + i + (size_t)i * nxc];
            for (int i = 0; i < nu; i++) Rd[(size_t)s->uoff[k] + i] = R ? R[orr + i + (size_t)i * nu] : 0.0;
            oq += (size_t)nxc * nxc; orr += (size_t)nu * nu; os += (size_t)nu * nxc;
            continue;
        }
        if (k == 0 && s->x_pad) {
            /* phantom root states: identity weight, no coupling; the caller's node 0 has no Q and no S */
            for (int i = 0; i < s->x_pad; i++) Hk[i + (size_t)i * nz] = 1.0;
            for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) Hk[s->x_pad + i + (size_t)(s->x_pad + j) * nz] = R ? R[orr + i + (size_t)j * nu] : 0.0;
            orr += (size_t)nu * nu;
            continue;
        }
        const int nx = s->nx[k];
        for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) Hk[i + (size_t)j * nz] = Q[oq + i + (size_t)j * nx];
        for (int j = 0; j < nu; j++) for (int i = 0; i < nu; i++) Hk[nx + i + (size_t)(nx + j) * nz] = R ? R[orr + i + (size_t)j * nu] : 0.0;
        for (int j = 0; j < nx; j++) for (int i = 0; i < nu; i++) {          /* S is nu x nx */
            const double v = S ? S[os + i + (size_t)j * nu] : 0.0;
            Hk[nx + i + (size_t)j * nz] = v;
            Hk[j + (size_t)(nx + i) * nz] = v;
        }
        oq += (size_t)nx * nx; orr += (size_t)nu * nu; os += (size_t)nu * nx;
    }
    s->in_valid = false;
    H2D(s->d_Hd, H.data(), s->poff[s->Nn]);
    HIP_TRY(hipMemcpyAsync(s->d_kind, kd.data(), sizeof(int) * (size_t)s->Nn, hipMemcpyHostToDevice, s->stream));      /* ints: not H2D (doubles) */
    H2D(s->q + s->x_pad, q, s->sum_nx - s->x_pad); H2D(s->r, r, s->sum_nu);
    /* weights: zero on dense nodes (their multipliers of bounds are zero by construction), the diagonals on clipping nodes */
    H2D(s->Qd, Qd.data(), s->sum_nx); H2D(s->Rd, Rd.data(), s->sum_nu);
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->dense = true; s->need_dense_init = true; s->D.dense = 1; s->need_init = true;
    s->use_fast = 0;                                       /* per-node dense blocks: generic kernels */
    return TQGPU_OK;
}

extern "C" int tqgpu_set_bounds(tqgpu_solver *s, const double *xmin, const double *xmax, const double *umin, const double *umax) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    s->in_valid = false;
    H2D(s->xmin + s->x_pad, xmin, s->sum_nx - s->x_pad); H2D(s->xmax + s->x_pad, xmax, s->sum_nx - s->x_pad); H2D(s->umin, umin, s->sum_nu); H2D(s->umax, umax, s->sum_nu);
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->need_pack = true;
    return TQGPU_OK;
}

extern "C" int tqgpu_set_lambda(tqgpu_solver *s, const double *lambda) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    /* kept in a resident buffer: every tqgpu_solve starts from it (device-to-device copy) */
    s->lam_valid = false;
    if (lambda) { H2D(s->d_lam_init + s->nx0, lambda, s->sum_lam); }
    else HIP_TRY(hipMemsetAsync(s->d_lam_init, 0, sizeof(double) * (size_t)s->sum_nx, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TQGPU_OK;
}
/* The whole clipping QP (and the starting duals) in one call, for callers that hand over all their data before every
 * solve (treeqp_tdunes_solve re-reads qp_in each time, dual_Newton_tree.c:1142-1160): every array is compared with
 * the pinned host mirror of what the device holds and only what changed is copied and uploaded -- asynchronously,
 * from pinned memory, with no synchronisation (the solve is stream-ordered behind it).  An MPC loop that only
 * moves x0 re-uploads b (and r) and nothing else.  NULL arrays are left alone. */
extern "C" int tqgpu_set_problem(tqgpu_solver *s, const double *A, const double *B, const double *b,
                                 const double *Qd, const double *Rd, const double *q, const double *r,
                                 const double *xmin, const double *xmax, const double *umin, const double *umax, const double *lambda) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    if (s->dense) { s->dense = false; s->D.dense = 0; s->use_fast = s->use_fast_orig; s->in_valid = false; s->need_init = true; s->need_pack = true; }
    char *slab = static_cast<char *>(s->slab);
    bool any_pack = false, any_init = false;
    auto put = [&](double *dev, const double *src, int count, bool pack, bool init) -> int {
        if (!src || count <= 0) return TQGPU_OK;
        char *mir = s->h_in + ((reinterpret_cast<char *>(dev) - slab) - (ptrdiff_t)s->in_off0);
        const size_t bytes = sizeof(double) * (size_t)count;
        if (s->in_valid && memcmp(mir, src, bytes) == 0) return TQGPU_OK;
        memcpy(mir, src, bytes);
        HIP_TRY(hipMemcpyAsync(dev, mir, bytes, hipMemcpyHostToDevice, s->stream));
        s->stream_pending = true;
        any_pack |= pack; any_init |= init;
        return TQGPU_OK;
    };
    int rc;
    const int nxe = s->sum_nx - s->x_pad;
    if ((rc = put(s->A + s->A_pad, A, s->sum_A - s->A_pad, true, false)) || (rc = put(s->B, B, s->sum_B, true, false)) ||
        (rc = put(s->b + s->nx0, b, s->sum_lam, false, false)) ||
        (rc = put(s->Qd + s->x_pad, Qd, nxe, true, true)) || (rc = put(s->Rd, Rd, s->sum_nu, true, true)) ||
        (rc = put(s->q + s->x_pad, q, nxe, true, false)) || (rc = put(s->r, r, s->sum_nu, true, false)) ||
        (rc = put(s->xmin + s->x_pad, xmin, nxe, true, false)) || (rc = put(s->xmax + s->x_pad, xmax, nxe, true, false)) ||
        (rc = put(s->umin, umin, s->sum_nu, true, false)) || (rc = put(s->umax, umax, s->sum_nu, true, false)))
        return rc;
    s->in_valid = A && B && b && Qd && Rd && q && r && xmin && xmax && umin && umax ? true : s->in_valid;
    if (any_init) s->need_init = true;
    if (any_pack) s->need_pack = true;
    if (lambda && s->sum_lam > 0) {
        const size_t bytes = sizeof(double) * (size_t)s->sum_lam;
        if (!s->lam_valid || memcmp(s->h_lam, lambda, bytes) != 0) {
            memcpy(s->h_lam, lambda, bytes);
            HIP_TRY(hipMemcpyAsync(s->d_lam_init + s->nx0, s->h_lam, bytes, hipMemcpyHostToDevice, s->stream));
            s->stream_pending = true;
            s->lam_valid = true;
        }
    }
    return TQGPU_OK;
}
#undef H2D

namespace {

int read_ctrl(tqgpu_solver *s) {
    if (s->w3_now && s->w3_mirror && !s->w3_tail_sg) {
        /* the last launch enqueued does not post (the first sweep of a solve whose chunk launches nothing else): a one-thread launch does */
        W3 w = next_w3(s);
        w.hm = s->h_res; s->w3_wait = w.tag; s->w3_tail_sg = true;
        hipLaunchKernelGGL(k_w3_post, dim3(1), dim3(WAVE), 0, s->stream, s->D, w);
    }
    if (s->w3_now && s->w3_mirror && s->w3_tail_sg) {
        /* three-launch family: the last launch enqueued posts the control block to pinned memory itself (w3_mirror) */
        volatile unsigned *seq = &s->h_res->seq;
        const unsigned want = s->w3_wait;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = true;
        for (long spins = 0; *seq != want; spins++) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { seen = false; break; }
        }
        if (seen) { std::atomic_thread_fence(std::memory_order_acquire); s->w3_seen = true; return TQGPU_OK; }      /* h_ctrl IS the block's control block */
    }
    s->w3_seen = false;
    HIP_TRY(hipMemcpyAsync(s->h_ctrl, s->D.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TQGPU_OK;
}

/* persistent launch: wait for the result block the top workgroup writes into pinned host memory (the
 * kernel may still be writing the state back -- everything else the host does is stream-ordered behind it) */
/* the block as tagged words (HostRes::tg, posted by the persistent launches): complete when every word carries `want`; unpacked into the fields */
bool take_tagged_block(HostRes *hr, unsigned want) {
    volatile unsigned long long *tg = hr->tg;
    unsigned long long w[HOSTRES_WORDS];
    for (int i = 0; i < HOSTRES_WORDS; i++) { w[i] = tg[i]; if ((unsigned)(w[i] >> 32) != want) return false; }
    unsigned *dst = reinterpret_cast<unsigned *>(&hr->c);
    for (int i = 0; i < 24; i++) dst[i] = (unsigned)w[i];
    hr->t_start = (w[24] & 0xFFFFFFFFull) | (w[25] << 32);
    hr->t_end = (w[26] & 0xFFFFFFFFull) | (w[27] << 32);
    hr->seq = want;
    return true;
}

int wait_result_block(tqgpu_solver *s) {
    volatile unsigned *seq = &s->h_res->seq;
    const unsigned want = s->psync.seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0; *seq != want; spins++) {
        if (take_tagged_block(s->h_res, want)) break;
        __builtin_ia32_pause();
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            /* no verdict (a wait inside the kernel timed out, or the launch failed): fall back to the stream.  A member of a
             * batch launch first waits for THAT launch (it runs on the lead's stream; the member's own stream is not ordered
             * behind it, and its control block is only final when the launch has ended) */
            if (s->batch_stream) HIP_TRY(hipStreamSynchronize(s->batch_stream));
            return read_ctrl(s);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return TQGPU_OK;
}

}  // namespace

#ifdef TQ_HOSTPROF
#include <chrono>
static double hp_acc[4] = {0, 0, 0, 0}; static long hp_n = 0;
#define HP_NOW() std::chrono::steady_clock::now()
#define HP_US(a, b) std::chrono::duration<double, std::micro>((b) - (a)).count()
#endif

namespace {

/* A solve in two halves, so that several mirrors can have their (single) persistent launch in flight at the
 * same time (tqgpu_solve_batch): solve_begin enqueues everything up to and including the first launch,
 * solve_end waits for the verdict, runs whatever is left (extra line-search trials, relaunches, the whole
 * Newton loop on the non-persistent paths) and fills the result. */
struct SolveCtx {
    Opts O;
    int launches = 0, ring = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool fast = false, persist = false, first_launch = true, prelaunched = false, gpersist = false, phases = false, events = true;
    unsigned batch_seq = 0;          /* != 0: this solve's persistent launch is part of a batch launch the caller makes with this launch number */
    int env_stamps = -1, env_nomirror = -1;          /* >= 0: TREEQP_AMD_STAMPS / TREEQP_AMD_NO_W3_MIRROR as the caller read them (tqgpu_solve_batch: once per call, not once per member) */
#ifdef TQ_HOSTPROF
    std::chrono::steady_clock::time_point hp0, hp1, hp2;
#endif
};

int solve_begin(tqgpu_solver *s, const tqgpu_opts *o, SolveCtx &cx, GItem *defer = nullptr) {
#ifdef TQ_HOSTPROF
    cx.hp0 = HP_NOW();
#endif
    HIP_TRY(hipSetDevice(s->device));
    s->export_valid = false;
    Opts &O = cx.O;
    O.maxIter = o->maxIter; O.termCondition = o->termCondition; O.regType = o->regType;
    O.lsMaxIter = o->lineSearchMaxIter; O.lsRestartTrigger = o->lineSearchRestartTrigger; O.reuse = o->checkLastActiveSet == 2 ? 1 : 0;
    O.tol = o->stationarityTolerance; O.regTol = o->regTol; O.regValue = o->regValue;
    O.gamma = o->lineSearchGamma; O.beta = o->lineSearchBeta;
    if (cx.env_stamps >= 0) O.stamps = cx.env_stamps;          /* (a batch call reads the environment once for all its members) */
    else { const char *e = getenv("TREEQP_AMD_STAMPS"); O.stamps = e ? std::max(1, atoi(e)) : 0; }
    if (O.termCondition < 0 || O.termCondition > 2 || O.regType < 0 || O.regType > 2 || O.regValue < 0)
        return fail(TQGPU_EINVAL, "invalid option value");

    if (s->sharded && !s->comm) return fail(TQGPU_ECOMM, "sharded mirror without a communicator: use tqgpu_solve_virtual_ranks");
    const Tree &T = s->T; const Data &D = s->D;
    hipStream_t st = s->stream;
    const int nxu = std::max(s->sum_nx, s->sum_nu);

    /* ls_log needs no reset: entry i is written by iteration i */

    if (o->profile) {
        while ((int)s->iter_ev.size() < o->maxIter + 1) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); s->iter_ev.push_back(ev); }
    }
    /* (the per-iteration / per-phase time records are all-NaN unless a profiled solve has written into them: not refilled per solve) */
    const size_t n_it = (size_t)std::max(o->maxIter, 1);
    if (s->times_dirty || s->iter_times.size() != n_it) s->iter_times.assign(n_it, NAN);

    s->w3_now = false;
    cx.fast = tiered_capable(s) && o->profile < 3;          /* level 3: the launch-per-level kernels, whose launches ARE the reference's phases */
    cx.phases = o->profile >= 3;
    if (cx.phases) {
        while ((int)s->phase_ev.size() < 4 * (o->maxIter + 1)) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); s->phase_ev.push_back(ev); }
        if (!s->sweep_ev0) { HIP_TRY(hipEventCreate(&s->sweep_ev0)); HIP_TRY(hipEventCreate(&s->sweep_ev1)); }
    }
    if (s->times_dirty || s->phase_times.size() != 3 * n_it) s->phase_times.assign(3 * n_it, NAN);
    s->first_sweep_time = NAN;
    s->times_dirty = o->profile != 0;
    cx.persist = persist_capable(s) && !o->profile && o->maxIter > 0;
    cx.gpersist = !cx.persist && uses_gpersist(s, defer != nullptr) && !o->profile && o->maxIter > 0;
    if (cx.gpersist) cx.persist = true;                  /* same host flow: one launch, verdict through the result block */
    cx.ring = (int)(s->solve_no % EV_RING);
    cx.ev0 = s->ring_ev0[cx.ring]; cx.ev1 = s->ring_ev1[cx.ring];
    s->solve_no++;
    /* the event pair costs two more packets on the queue per solve; a single persistent launch reports its own clock (launch
     * start to verdict) anyway, so there the pair is optional */
    /* the three-launch family reports through the pinned result block as well (launches of k_sg / k_sgp post the control block and
     * their own clock: w3_mirror) */
    const bool w3_will = !cx.persist && s->w3_ok && !s->dense && !cx.phases && !cx.fast && !s->sharded;
    const bool no_mirror = cx.env_nomirror >= 0 ? cx.env_nomirror != 0 : getenv("TREEQP_AMD_NO_W3_MIRROR") != nullptr;        /* (A/B and tests: copy + synchronisation per read, HIP events per solve, as before) */
    s->w3_mirror = w3_will && !o->profile && !no_mirror;
    s->w3_seen = false;
    if (s->w3_mirror) s->h_res->seq = 0;          /* (no launch of this mirror is in flight) */
    cx.events = (s->ev_timing || (!cx.persist && !s->w3_mirror)) && !cx.batch_seq && !defer;      /* (a member of a batch launch: the launch is on the lead's stream, an event pair on the member's own would time nothing) */
    s->ring_ok[(size_t)cx.ring] = cx.events ? 1 : 0;
    if (cx.events) HIP_TRY(hipEventRecord(cx.ev0, st));
    /* the three-launch family with k_sgp: the first launch of the solve takes the starting duals and resets the control block itself */
    const bool w3_fresh = !cx.persist && s->w3_ok && s->w3_sgp && !s->dense && o->profile < 3 && !tiered_capable(s) && !s->sharded;
    if (!cx.persist && !w3_fresh) HIP_TRY(hipMemsetAsync(D.ctrl, 0, sizeof(Ctrl), st));     /* persistent path: reset by the launch's prologue */
    if (s->need_init && !cx.gpersist) {     /* g_persist recomputes the reciprocal weights itself; dense nodes never read theirs */
        hipLaunchKernelGGL(k_init, dim3((nxu + 255) / 256), dim3(256), 0, st, s->sum_nx, s->sum_nu, D); cx.launches++;
        s->need_init = false;
        s->stream_pending = true;
    }
    if (s->dense && s->need_dense_init) {
        hipLaunchKernelGGL(k_dense_init, dim3(T.Nn), dim3(WAVE), s->lds_dense, st, T, D); cx.launches++;
        s->need_dense_init = false;
    }
    if (!cx.persist) {
        /* the current buffer is lam0 at the start of every solve */
        if (!w3_fresh) HIP_TRY(hipMemcpyAsync(D.lam0, s->d_lam_init, sizeof(double) * (size_t)s->sum_nx, hipMemcpyDeviceToDevice, st));
        /* first sweep at lambda0 (phase S of iteration 0 + fval0); the persistent launch does it as its prologue */
        if (cx.phases) HIP_TRY(hipEventRecord(s->sweep_ev0, st));
        s->w3_now = s->w3_ok && !s->dense && !cx.phases && !cx.fast && !s->sharded;
        s->fuse_now = s->fuse_ok && !cx.phases && !cx.fast && !s->sharded && !s->w3_now;
        if (s->w3_now) { launch_sg(s, cx.O, 0, 0, 0, w3_fresh); cx.launches++; }          /* with fval0 and the first termination test as its tail */
        else if (s->fuse_now) { hipLaunchKernelGGL(k_stage_f, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, cx.O, next_fuse(s), 0, 0, 0); cx.launches++; }      /* with k_fval_init as its tail */
        else {
            hipLaunchKernelGGL(k_stage, dim3(T.Nn), dim3(WAVE), s->lds_stage, st, T, D, 0, 0, 0); cx.launches++;
            hipLaunchKernelGGL(k_fval_init, dim3(1), dim3(256), 0, st, T, D); cx.launches++;
        }
        if (cx.phases) HIP_TRY(hipEventRecord(s->sweep_ev1, st));
    } else {
#ifdef TQ_HOSTPROF
        cx.hp1 = HP_NOW();
#endif
        int rcx = TQGPU_OK;
        if (cx.gpersist) {
            s->launch_no = (s->launch_no + 1) & 0xFFFFu;
            if (s->launch_no == 0) s->launch_no = 1;
            s->psync.seq = s->launch_no << 16;
            GParams gp;
            gp.lvl_first = s->d_lvl_first; gp.lam_init = s->d_lam_init; gp.hres = s->h_res; gp.seq = s->psync.seq; gp.lds_wave = (int)s->lds_gp_wave;
            gp.in_lds = s->gp_in_lds ? 1 : 0; gp.tab_in_lds = (!s->gp_in_lds && s->gp_tab_in_lds) ? 1 : 0; gp.small16 = s->gp_small16 ? 1 : 0; gp.small8 = s->gp_small8 ? 1 : 0; gp.sum_nx = s->sum_nx; gp.sum_nu = s->sum_nu; gp.sum_W = s->sum_W; gp.sum_Ut = s->sum_Ut;
            gp.const_in_lds = s->gp_const_in_lds ? 1 : 0; gp.sum_A = s->sum_A; gp.sum_B = s->sum_B;
            if (defer) { defer->T = T; defer->D = D; defer->G = gp; }           /* launched by the caller, together with the rest of its batch */
            else hipLaunchKernelGGL(g_persist, dim3(1), dim3(GP_WAVES * WAVE), s->lds_gp_total, st, T, D, O, gp);
            cx.launches++;
        } else rcx = launch_persist(s, O, cx.launches, 1, cx.batch_seq);
        if (rcx != TQGPU_OK) return rcx;
        cx.first_launch = false; cx.prelaunched = true;
        /* the launch normally ends the solve: close the timing here */
        if (cx.events) HIP_TRY(hipEventRecord(cx.ev1, st));
#ifdef TQ_HOSTPROF
        cx.hp2 = HP_NOW();
#endif
    }
    return TQGPU_OK;
}

int solve_end(tqgpu_solver *s, const tqgpu_opts *o, SolveCtx &cx, tqgpu_result *res) {
    HIP_TRY(hipSetDevice(s->device));
    const Opts &O = cx.O;
    hipStream_t st = s->stream;
    const bool fast = cx.fast, persist = cx.persist;
    int &launches = cx.launches;
    /* Newton loop (dual_Newton_tree.c:1166-1228).  The device decides (termination, Armijo);
     * the host enqueues `chunk` tagged iterations ahead and reads the control block once per chunk.
     * Iterations enqueued beyond convergence, or while a line search still needs trials, are
     * no-ops by their phase guards. */
    bool tail_done = false;
    int h = 0, ev_idx = 0;
    bool finished = o->maxIter <= 0;       /* nothing to iterate: reported as "maximum iterations" */
    if (finished) { HIP_TRY(hipStreamSynchronize(st)); memset(s->h_ctrl, 0, sizeof(Ctrl)); s->h_ctrl->status = 1; }
    if (o->profile) HIP_TRY(hipEventRecord(s->iter_ev[0], st));
    int chunk = s->last_iter > 0 ? std::min(s->last_iter + 1, 16) : s->chunk;
    bool predicted = s->last_iter > 0 && !fast && !persist;     /* the chunk is a prediction: its last iteration should only find convergence */
    /* a line search that wants a second trial turns everything enqueued behind it into no-ops (~3 us a launch): a problem that
     * backtracked last time is fed two iterations at a time (TREEQP_AMD_LS_CHUNK; one C5-class tree: 1.15 ms with chunks of 8, 1.04 / 1.01 / 1.06 ms with 4 / 2 / 1),
     * a read-back per chunk instead */
    static const int ls_chunk = getenv("TREEQP_AMD_LS_CHUNK") ? std::max(1, atoi(getenv("TREEQP_AMD_LS_CHUNK"))) : 2;
    if (s->last_ls_extra && !fast && !persist && chunk > ls_chunk && !(s->w3_now && !s->ls_pred.empty())) { chunk = ls_chunk; predicted = false; }      /* (three-launch family: the further trials are predicted too) */
    if (cx.phases) { chunk = 1; predicted = false; }            /* phase timing: one iteration per read-back, so that every recorded event belongs to work that ran */
    int rest_due = -1;                                          /* iteration whose termination test ran, whose step did not */
    while (!finished) {
        const int n = persist ? 0 : std::min(chunk, o->maxIter - h);
        if (persist) {
            if (!cx.prelaunched) {
                if (cx.gpersist) return fail(TQGPU_ENODEVICE, "single-workgroup persistent kernel ended without a verdict");
                int rcx = launch_persist(s, O, launches, cx.first_launch ? 1 : 0);
                if (rcx != TQGPU_OK) return rcx;
                cx.first_launch = false;
                if (cx.events) HIP_TRY(hipEventRecord(cx.ev1, st));
            }
            cx.prelaunched = false;
        }
        int deferred = -1;
        for (int i = 0; i < n; i++) {
            if (fast) { int rcx = launch_fast_iteration(s, O, h + i, launches); if (rcx != TQGPU_OK) return rcx; }
            else {
                int parts = 3;
                if (rest_due == h + i) parts &= ~1;                                   /* its test already ran */
                if (predicted && i == n - 1) { parts &= ~2; deferred = h + i; }
                /* (three-launch family: the last iteration of the chunk that launches anything posts the verdict to the host) */
                const bool last = i == n - 1 || (predicted && i == n - 2);
                if (parts) launch_generic_iteration(s, O, h + i, launches, parts, cx.phases, last);
            }
            if (o->profile && ev_idx + 1 < (int)s->iter_ev.size()) HIP_TRY(hipEventRecord(s->iter_ev[++ev_idx], st));
        }
        /* persistent path: the verdict comes through the result block in pinned host memory */
        int rc = persist ? wait_result_block(s) : read_ctrl(s);
        if (rc != TQGPU_OK) return rc;
#ifdef TQ_HOSTPROF
        if (persist) { auto hp3 = HP_NOW(); hp_acc[0] += HP_US(cx.hp0, cx.hp1); hp_acc[1] += HP_US(cx.hp1, cx.hp2); hp_acc[2] += HP_US(cx.hp2, hp3); hp_n++;
          if (hp_n % 200 == 0) { fprintf(stderr, "[hostprof] pre %.2f us, launch %.2f us, readback+sync %.2f us (avg of %ld)\n", hp_acc[0] / hp_n, hp_acc[1] / hp_n, hp_acc[2] / hp_n, hp_n); } }
#endif
        tail_done = persist;
        bool extra_trials = false;
        /* trials beyond the first go out in batches: 3, then 6, 12, 16, .. per read-back of the control block.  A trial that is
         * accepted turns the rest of its batch into no-ops (~6 us each), so short searches -- the usual case: one or two more
         * trials -- want small batches (one C5-class tree: 1.09 ms with batches of 8, 0.99 ms with 3), long ones few read-backs. */
        static const int trial_batch0 = getenv("TREEQP_AMD_TRIAL_BATCH") ? std::max(1, atoi(getenv("TREEQP_AMD_TRIAL_BATCH"))) : 3;
        int trial_batch = trial_batch0, ls_of = -1;
        while (!s->h_ctrl->done && s->h_ctrl->ls_pending) {
            tail_done = false;
            extra_trials = true;
            /* the line search of iteration `iter` wants more trials: a batch of them */
            const int it = s->h_ctrl->iter, t0 = s->h_ctrl->ls_iter;
            if (it != ls_of) { ls_of = it; trial_batch = trial_batch0; }
            for (int t = t0; t < t0 + trial_batch && t <= O.lsMaxIter; t++) {
                s->w3_post_next = t + 1 >= t0 + trial_batch || t + 1 > O.lsMaxIter;      /* the last of the batch */
                int rcx = launch_trial(s, O, fast, it, t, launches);
                if (rcx != TQGPU_OK) return rcx;
            }
            trial_batch = std::min(2 * trial_batch, 16);
            if ((rc = read_ctrl(s)) != TQGPU_OK) return rc;
        }
        h = s->h_ctrl->iter;
        finished = s->h_ctrl->done != 0;
        chunk = cx.phases ? 1 : s->chunk;
        predicted = false;
        /* the termination test of the deferred iteration was enqueued behind the FIRST trial of the line search before it: when that
         * line search needed further trials (enqueued above, after the read-back), the test found the search pending and did nothing --
         * it is then due again with the rest of its iteration */
        rest_due = (!finished && deferred == h && !extra_trials) ? h : -1;
        if (persist && !finished && !cx.gpersist) {
            tail_done = false;
            unsigned tmo = 0;
            HIP_TRY(hipMemcpy(&tmo, s->psync.timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
            if (tmo) return fail(TQGPU_ETIMEOUT, "persistent solve kernel: a bounded inter-workgroup wait timed out");
        }
    }
    const int host_iter = ev_idx;
    float ms = 0.f;
    if (!tail_done && !persist && s->w3_now && s->w3_mirror && s->w3_seen && !cx.events && !o->profile && !cx.phases) {
        /* three-launch family: every launch of the solve is accounted for (the last one's tail has posted the verdict; what its other
         * workgroups still write is stream-ordered before anything the host does next): no synchronisation, the device's own clock */
        unsigned long long t_first;
        memcpy(&t_first, &s->h_ctrl->pad0, sizeof(t_first));
        ms = 1e-5f * (float)(s->h_res->t_end - t_first);
    } else if (!tail_done) {
        if (cx.events) HIP_TRY(hipEventRecord(cx.ev1, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (cx.events) HIP_TRY(hipEventElapsedTime(&ms, cx.ev0, cx.ev1));
        else ms = 1e-5f * (float)(s->h_res->t_end - s->h_res->t_start);      /* several persistent launches (relaunch): the last one's own clock */
    } else {
        /* single persistent launch: the kernel's own clock, launch start to verdict (the event pair of this
         * solve can be read later through tqgpu_get_device_times, which synchronises) */
        ms = 1e-5f * (float)(s->h_res->t_end - s->h_res->t_start);
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(TQGPU_ENODEVICE, std::string("kernel launch failed: ") + hipGetErrorString(le));

    if (cx.phases) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, s->sweep_ev0, s->sweep_ev1) == hipSuccess) s->first_sweep_time = 1e-3 * t;
        for (int i = 0; i < host_iter && (size_t)(4 * i + 3) < s->phase_ev.size() && (size_t)(3 * i + 2) < s->phase_times.size(); i++)
            for (int ph = 0; ph < 3; ph++)
                if (hipEventElapsedTime(&t, s->phase_ev[(size_t)(4 * i + ph)], s->phase_ev[(size_t)(4 * i + ph + 1)]) == hipSuccess) s->phase_times[(size_t)(3 * i + ph)] = 1e-3 * t;
    }
    if (o->profile) {
        for (int i = 0; i < host_iter && i + 1 < (int)s->iter_ev.size() && i < (int)s->iter_times.size(); i++) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, s->iter_ev[i], s->iter_ev[i + 1]) == hipSuccess) s->iter_times[i] = 1e-3 * t;
        }
    }
    const Ctrl &c = *s->h_ctrl;
    res->status = c.status; res->iter = c.iter; res->ls_total = c.ls_total; res->ls_last = c.ls_last;
    res->n_launches = launches; res->device_time = 1e-3 * ms; res->last_error_norm = c.err; res->last_fval = c.fval;
    s->last_iter = c.iter;
    s->last_ls_extra = c.ls_total > c.iter ? 1 : 0;
    if (s->w3_now && s->w3_merge && c.status == 2 && s->T.Np > 1) {
        /* NOT_DESCENT_DIRECTION out of the merged launch (k_sgp mode 2: forward sweep + first trial): the trial sweep ran before the
         * direction test and has put x, u, xUnc, QinvCal of the point lambda + dlambda in place.  The reference returns from
         * line_search with the phase-S iterate at lambda (dual_Newton_tree.c:944-954): one stage sweep at the current duals (which the
         * trial did not touch: it wrote the other buffer) restores exactly that -- same arithmetic, operation for operation. */
        hipLaunchKernelGGL(k_stage, dim3(s->T.Nn), dim3(WAVE), s->lds_stage, st, s->T, s->D, 0, 0, 0);
        launches++; res->n_launches = launches;
        s->stream_pending = true;
    }
    if (s->w3_now) {
        s->ls_pred.clear();
        if (c.status == 0 && c.ls_total > c.iter) {
            /* some iteration needed further trials: fetch the trial counts (only then: a copy is a packet on the queue and a synchronisation) */
            const int nlog = std::min(c.iter, std::min(s->ls_log_cap, 256));
            HIP_TRY(hipMemcpyAsync(s->h_ls_log, s->D.ls_log, sizeof(int) * (size_t)nlog, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            s->ls_pred.assign(s->h_ls_log, s->h_ls_log + nlog);
        }
    }
    return TQGPU_OK;
}

}  // namespace

/* A persistent launch needs all its workgroups resident at once; that was checked at creation against THIS process's launches.
 * When something else holds compute units (another process, another stream), a bounded wait inside the launch gives up after
 * 0.5 s and the launch ends itself.  That is not an error of the solve: clear the sticky word, and redo the solve from the
 * same starting duals on the path that has no residency requirement (launch per tier, or per level).  The mirror keeps off
 * the persistent path for the next `PERSIST_BACKOFF` solves, then tries it again. */
constexpr int PERSIST_BACKOFF = 1000;

static int solve_after_timeout(tqgpu_solver *s, const tqgpu_opts *o, tqgpu_result *res) {
    HIP_TRY(hipMemsetAsync(s->psync.timeout, 0, sizeof(unsigned), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->use_persist = 0;
    s->persist_backoff = PERSIST_BACKOFF;
    s->n_timeouts++;
    if (getenv("TREEQP_AMD_VERBOSE")) fprintf(stderr, "[treeqp_amd] persistent launch timed out (device shared?): solving on the launch-per-tier path\n");
    SolveCtx cx;
    int rc = solve_begin(s, o, cx);
    if (rc != TQGPU_OK) return rc;
    return solve_end(s, o, cx, res);
}

static int enqueue_export(tqgpu_solver *s, const double *lamc) {
    const Data &D = s->D;
    const int nxe = s->sum_nx - s->x_pad, nue = s->sum_nu, nl = s->sum_lam;
    const int n = std::max(std::max(nxe, nue), std::max(nl, 1));
    hipLaunchKernelGGL(k_export_all, dim3((n + 255) / 256), dim3(256), 0, s->stream, nxe, nue, nl, s->x_pad, s->nx0, D, lamc, s->d_out);
    HIP_TRY(hipMemcpyAsync(s->h_out, s->d_out, sizeof(double) * std::max<size_t>(s->out_doubles, 1), hipMemcpyDeviceToHost, s->stream));
    return TQGPU_OK;
}

extern "C" int tqgpu_solve(tqgpu_solver *s, const tqgpu_opts *o, tqgpu_result *res) {
    SETTLE(s);
    if (!s || !o || !res) return fail(TQGPU_EINVAL, "tqgpu_solve: bad arguments");
    /* a mirror that is one rank's share of a sharded solve launches only that share: on its own it would time out, and its launch number
     * would fall out of step with its peers' */
    if (s->pshard && s->nranks > 1) return fail(TQGPU_EINVAL, "tqgpu_solve: this mirror is rank " + std::to_string(s->rank) + " of a sharded solve (tqgpu_pshard_init): use tqgpu_pshard_begin / _end");
    if (s->persist_backoff > 0 && --s->persist_backoff == 0) s->use_persist = s->use_persist_orig;
    SolveCtx cx;
    int rc = solve_begin(s, o, cx);
    if (rc != TQGPU_OK) return rc;
    const int launches0 = cx.launches;
    /* a caller that always fetches the solution (the drop-in front end): the packing kernel and the download go out behind the single
     * persistent launch now, while it runs, instead of after its verdict has travelled to the host and back (a launch latency, the
     * launch's write-back tail and a synchronisation off the caller's critical path); which dual buffer is current is the device's
     * knowledge.  Valid if that one launch was the whole solve. */
    const bool ahead = s->export_ahead && cx.persist && cx.prelaunched && !s->pshard && !s->sharded;
    if (ahead && enqueue_export(s, nullptr) != TQGPU_OK) return TQGPU_ENODEVICE;
    rc = solve_end(s, o, cx, res);
    if (rc == TQGPU_ETIMEOUT) rc = solve_after_timeout(s, o, res);
    else if (rc == TQGPU_OK && ahead && res->n_launches == launches0) s->export_valid = true;
    return rc;
}

extern "C" int tqgpu_set_export_ahead(tqgpu_solver *s, int on) {
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    s->export_ahead = on != 0;
    s->export_valid = false;
    return TQGPU_OK;
}

/* n solves of the same problem from the same starting duals, one after the other (each waits for its verdict), as the reference's
 * drivers time the solver (`for (jj = 0; jj < NREP; jj++) treeqp_tdunes_solve(...)`, examples/spring_mass_dual_newton_tree.c:135-140):
 * the loop is in C, so what is timed is the solver and not the caller's interpreter.  res = the last solve's result; sums over
 * the n solves in iter_sum / ls_sum / launch_sum.  Stops at the first solve that does not end with status 0 or 1. */
extern "C" int tqgpu_solve_n(tqgpu_solver *s, const tqgpu_opts *o, int n, tqgpu_result *res, long *iter_sum, long *ls_sum, long *launch_sum) {
    SETTLE(s);
    if (!s || !o || !res || n < 1) return fail(TQGPU_EINVAL, "tqgpu_solve_n: bad arguments");
    long it = 0, ls = 0, la = 0;
    /* (waiting for the stream to drain, or a few microseconds, before the next launch measured slower than launching at once) */
    for (int i = 0; i < n; i++) {
        const int rc = tqgpu_solve(s, o, res);
        if (rc != TQGPU_OK) return rc;
        it += res->iter; ls += res->ls_total; la += res->n_launches;
        if (res->status != 0 && res->status != 1) break;
    }
    if (iter_sum) *iter_sum = it;
    if (ls_sum) *ls_sum = ls;
    if (launch_sum) *launch_sum = la;
    return TQGPU_OK;
}

/* diagnostic / test support: a foreign kernel that holds compute units.  `blocks` workgroups of 256 threads, each claiming `lds_kb`
 * KiB of LDS (160 = a whole CU each), spin for `ms` milliseconds of wall clock (at most 3000) on a stream of their own; the call
 * returns at once.  tqgpu_debug_occupy_wait() waits for them.  Used by the tests to take co-residency away from a persistent launch. */
namespace {
__global__ void __launch_bounds__(256) k_occupy(unsigned long long ticks, int *sink) {
    extern __shared__ __attribute__((aligned(16))) double lds_occ[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) lds_occ[0] = 1.0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds_occ[0] == 2.0 && sink) *sink = 1;
}
hipStream_t g_occ_stream = nullptr;
}
extern "C" int tqgpu_debug_occupy(int device, int blocks, int lds_kb, int ms) {
    if (blocks < 1 || lds_kb < 0 || lds_kb > 160 || ms < 1) return fail(TQGPU_EINVAL, "tqgpu_debug_occupy: bad arguments");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    if (!g_occ_stream) {
        /* a stream of another PRIORITY than the solvers': the runtime deals streams of one priority over a handful of hardware queues, and
         * two streams that share a queue run their kernels one after the other -- the occupying kernel would then hold back the very launch
         * it is meant to run beside (one test run in four, depending on how many mirrors the process had created before) */
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&g_occ_stream, hipStreamNonBlocking, greatest));
    }
    const size_t lds = (size_t)lds_kb * 1024;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_occupy, dim3((unsigned)blocks), dim3(256), lds, g_occ_stream, (unsigned long long)std::min(ms, 3000) * 100000ull, (int *)nullptr);
    HIP_TRY(hipGetLastError());
    return TQGPU_OK;
}
extern "C" int tqgpu_debug_occupy_wait(void) {
    if (g_occ_stream) HIP_TRY(hipStreamSynchronize(g_occ_stream));
    return TQGPU_OK;
}

/* geometry of the persistent launch of this mirror: block levels of the tree, tiers, workgroups of one launch, workgroups of
 * such launches the device holds at once (co-residency: what tqgpu_solve_batch may have in flight together), compute units */
extern "C" int tqgpu_geometry(const tqgpu_solver *s, int *levels, int *tiers, int *workgroups, int *capacity, int *compute_units) {
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    if (levels) *levels = s->Nh;
    if (tiers) *tiers = s->fast >= 0 ? s->n_tiers : 0;
    if (workgroups) *workgroups = s->persist_ok ? s->geom.G : 0;
    if (capacity) *capacity = s->persist_ok ? s->co_capacity : 0;
    if (compute_units) *compute_units = s->n_cu;
    return TQGPU_OK;
}

/* diagnostic: how often a persistent launch of this mirror timed out and the solve was redone on another path */
extern "C" int tqgpu_timeouts(const tqgpu_solver *s) { return s ? s->n_timeouts : 0; }

static int batch_kernel_index(const tqgpu_solver *s) {
#define X(idx, nx, nu, md, ms) if (s->fNX == nx && s->fNU == nu && s->fMD == md && s->mstage == ms) return idx;
    BATCH_TABLE(X)
#undef X
    return -1;
}
static int launch_persist_batch(tqgpu_solver *lead, int kidx, const PItem *items, const Opts &O, int n_trees, unsigned seq) {
    const int G = lead->geom.G;
    static const char *nap_env = getenv("TREEQP_AMD_NAP");
    const int batch_nap = nap_env ? atoi(nap_env) : nap_for_grid(G * n_trees);     /* the whole launch polls the same memory system */
    static size_t lds_allowed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    switch (kidx) {
#define X(idx, nx, nu, md, ms) case idx: { \
        if (lds_allowed[idx] < lead->lds_persist) { int rc = allow_lds(f_persist_batch<nx, nu, md, ms>, lead->lds_persist); if (rc != TQGPU_OK) return rc; lds_allowed[idx] = lead->lds_persist; } \
        hipLaunchKernelGGL((f_persist_batch<nx, nu, md, ms>), dim3((unsigned)(G * n_trees)), dim3(FW * WAVE), lead->lds_persist, lead->stream, items, O, G, seq, batch_nap); break; }
        BATCH_TABLE(X)
#undef X
        default: return fail(TQGPU_EINVAL, "no batch kernel for this shape");
    }
    return TQGPU_OK;
}

/* Batched multi-tree solve (the usage pattern of examples/fault_tolerance.c:486-530, one QP per
 * configuration): n independent mirrors, same options.  Mirrors on the persistent path have their launches in
 * flight together, as many at a time as fit on the device at once (every workgroup of a persistent launch must
 * be resident); the others are solved one after the other.  results[i] belongs to solvers[i]; the first error
 * is returned after every started solve has been waited for. */
extern "C" int tqgpu_solve_batch(tqgpu_solver **solvers, int n, const tqgpu_opts *o, tqgpu_result *results) {
    if (!solvers || n < 1 || !o || !results) return fail(TQGPU_EINVAL, "tqgpu_solve_batch: bad arguments");
    for (int i = 0; i < n; i++) if (!solvers[i]) return fail(TQGPU_EINVAL, "tqgpu_solve_batch: null mirror");
    for (int i = 0; i < n; i++) if (solvers[i]->pshard && solvers[i]->nranks > 1) return fail(TQGPU_EINVAL, "tqgpu_solve_batch: a member is one rank of a sharded solve (tqgpu_pshard_init)");
    std::vector<SolveCtx> cx((size_t)n);
    struct InBatch {                    /* (see tqgpu_solver::in_batch) */
        tqgpu_solver **v; int n;
        InBatch(tqgpu_solver **v_, int n_) : v(v_), n(n_) { for (int i = 0; i < n; i++) v[i]->in_batch = true; }
        ~InBatch() { for (int i = 0; i < n; i++) v[i]->in_batch = false; }
    } in_batch_guard(solvers, n);
    /* the environment switches of a batch call, read ONCE (a getenv is a scan of the environment: two dozen of them per call of seven members
     * were microseconds of a 125 us step) */
    const bool env_batch_launches = getenv("TREEQP_AMD_BATCH_LAUNCHES") != nullptr, env_batch_sync = getenv("TREEQP_AMD_BATCH_SYNC") != nullptr;
    {
        const char *e = getenv("TREEQP_AMD_STAMPS");
        const int st = e ? std::max(1, atoi(e)) : 0, nm = getenv("TREEQP_AMD_NO_W3_MIRROR") != nullptr ? 1 : 0;
        for (int k = 0; k < n; k++) { cx[(size_t)k].env_stamps = st; cx[(size_t)k].env_nomirror = nm; }
    }
    int first_err = TQGPU_OK;
    std::string first_msg;
    for (int k = 0; k < n; k++) {          /* as tqgpu_solve: a mirror that backed off the persistent path returns to it after PERSIST_BACKOFF solves */
        tqgpu_solver *sk = solvers[k];
        if (sk->persist_backoff > 0 && --sk->persist_backoff == 0) sk->use_persist = sk->use_persist_orig;
    }
    int i = 0;
    while (i < n) {
        /* a wave of mirrors on one device whose persistent launches fit together */
        const int dev = solvers[i]->device;
        int used = 0, j = i;
        for (; j < n; j++) {
            tqgpu_solver *s = solvers[j];
            const bool persist_like = persist_capable(s) && !o->profile && o->maxIter > 0;
            const bool gp_like = !persist_like && uses_gpersist(s, true) && !o->profile && o->maxIter > 0;
            /* single-workgroup mirrors do not wait for each other: any number per launch; launch-per-level mirrors go alone */
            const int need = persist_like ? s->geom.G : (gp_like ? 0 : s->co_capacity + 1);
            /* two workgroups on one CU run at half speed each: a batch fills the CUs once, not twice, unless a member needs more */
            /* separate launches: two workgroups on one CU run at half speed each, so a batch fills the CUs once, not twice, unless a
             * member needs more.  Members that go out together as ONE launch (batch kernel) fill the device up to what is co-resident:
             * a tree's waves are parked at barriers and waits two thirds of the time, and trees that share CUs fill those gaps
             * (C2: 3 trees 72 k it/s, 5 trees 98 k; C1: 22 trees 656 k, 38 trees 922 k). */
            const bool one_launch = persist_like && batch_kernel_index(s) >= 0 && o->checkLastActiveSet != 2 && !env_batch_launches;
            const int cap = (!one_launch && s->n_cu > 0 && need <= s->n_cu) ? std::min(s->co_capacity, s->n_cu) : s->co_capacity;
            if (j > i && (s->device != dev || used + need > cap)) break;
            used += need;
        }
        const int begun_from = i, begun_to = j;
        int ok_to = begun_from;
        /* single-workgroup mirrors of this wave go out as ONE launch (one workgroup per tree) on the first one's stream */
        std::vector<int> gp_members;
        for (int k = begun_from; k < begun_to; k++) {
            tqgpu_solver *s = solvers[k];
            if (!persist_capable(s) && uses_gpersist(s, true) && !o->profile && o->maxIter > 0) gp_members.push_back(k);
        }
        tqgpu_solver *lead = gp_members.size() >= 2 ? solvers[gp_members[0]] : nullptr;
        if (lead && lead->gitems_cap < (int)gp_members.size()) {
            if (lead->d_gitems) (void)hipFree(lead->d_gitems);
            if (lead->h_gitems) (void)hipHostFree(lead->h_gitems);
            lead->d_gitems = nullptr; lead->h_gitems = nullptr; lead->gitems_cap = 0;
            const size_t cap = gp_members.size() * 2;
            if (hipMalloc(&lead->d_gitems, cap * sizeof(GItem)) != hipSuccess || hipHostMalloc((void **)&lead->h_gitems, cap * sizeof(GItem), hipHostMallocDefault) != hipSuccess)
                return fail(TQGPU_ENOMEM, "allocation of the batch descriptors failed");
            lead->gitems_cap = (int)cap;
        }
        /* persistent mirrors of this wave that share a shape with a batch kernel go out as ONE launch as well */
        std::vector<int> pm;
        {
            const tqgpu_solver *f = nullptr;
            for (int k = begun_from; k < begun_to; k++) {
                const tqgpu_solver *s = solvers[k];
                if (!(persist_capable(s) && !o->profile && o->maxIter > 0 && o->checkLastActiveSet != 2 && batch_kernel_index(s) >= 0)) continue;
                if (!f) f = s;
                if (batch_kernel_index(s) == batch_kernel_index(f) && s->geom.G == f->geom.G && s->lds_persist == f->lds_persist && s->Nn == f->Nn) pm.push_back(k);
            }
            if (pm.size() < 2 || env_batch_launches) pm.clear();      /* (=1: one launch per tree, the round-1 protocol) */
        }
        for (int k = begun_from; k < begun_to; k++) {
            /* a member of the previous batch launch that goes out again on the same lead's stream is ordered behind it by that stream */
            const bool same_lead = (!pm.empty() && solvers[k]->settle_stream == solvers[pm[0]]->stream && std::find(pm.begin(), pm.end(), k) != pm.end()) ||
                                   (lead && solvers[k]->settle_stream == lead->stream && std::find(gp_members.begin(), gp_members.end(), k) != gp_members.end());
            if (same_lead) { if ((!pm.empty() && k == pm[0]) || (lead && k == gp_members[0])) solvers[k]->settle_stream = nullptr; }
            else SETTLE(solvers[k]);
        }
        unsigned pseq = 0;
        if (!pm.empty()) {
            unsigned mx = 0;
            for (int k : pm) mx = std::max(mx, solvers[k]->launch_no);
            unsigned nn = mx + 1;
            if (nn > 0xFFFFu) {          /* the 16-bit launch number wraps: see launch_persist */
                HIP_TRY(hipStreamSynchronize(solvers[pm[0]]->stream));      /* (the previous batch launch's last workgroups are done with the slabs) */
                for (int k : pm) {
                    HIP_TRY(hipMemsetAsync(solvers[k]->sync_slab, 0, solvers[k]->sync_bytes, solvers[k]->stream));
                    solvers[k]->stream_pending = true;          /* the batch launch (on the lead's stream) waits for the wipe */
                }
                nn = 1;
            }
            pseq = nn << 16;
        }
        size_t gi = 0, lds_batch = 0, pi = 0;
        bool gp_launched = false;
        for (int k = begun_from; k < begun_to; k++) {
            const bool in_group = lead && gi < gp_members.size() && gp_members[gi] == k;
            if (pi < pm.size() && pm[pi] == k) { cx[(size_t)k].batch_seq = pseq; pi++; }
            int rc = solve_begin(solvers[k], o, cx[(size_t)k], in_group ? &lead->h_gitems[gi] : nullptr);
            if (rc != TQGPU_OK) { if (first_err == TQGPU_OK) { first_err = rc; first_msg = g_err; } break; }
            if (in_group) { gi++; lds_batch = std::max(lds_batch, solvers[k]->lds_gp_total); }
            ok_to = k + 1;
        }
        if (lead && gi > 0) {
            hipStream_t st0 = lead->stream;
            /* inputs of the other members were uploaded on their own streams: order the launch behind them, and their
             * later work (solution export) behind the launch */
            /* (round 3: no event traffic in the steady state -- a record and a wait per member before the launch and another wait after it
             * were 1.5 ms of an 11.5 ms step with 256 trees.  What a member enqueued on its own stream since its last synchronisation
             * (asynchronous uploads: stream_pending) is waited for here; the launch's end is waited for lazily, see settle().) */
            for (size_t m = 1; m < gi; m++) {
                tqgpu_solver *sm = solvers[gp_members[m]];
                if (sm->stream_pending) { HIP_TRY(hipStreamSynchronize(sm->stream)); sm->stream_pending = false; }
            }
            HIP_TRY(hipMemcpyAsync(lead->d_gitems, lead->h_gitems, gi * sizeof(GItem), hipMemcpyHostToDevice, st0));
            if (lds_batch > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(g_persist_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_batch));
            hipLaunchKernelGGL(g_persist_batch, dim3((unsigned)gi), dim3(GP_WAVES * WAVE), lds_batch, st0, lead->d_gitems, cx[(size_t)gp_members[0]].O);
            for (size_t m = 0; m < gi; m++) solvers[gp_members[m]]->batch_stream = st0;
            gp_launched = true;
        }
        if (!pm.empty() && ok_to == begun_to) {
            tqgpu_solver *pl = solvers[pm[0]];
            const size_t np = pm.size();
            if (pl->pitems_cap < (int)np) {
                if (pl->d_pitems) (void)hipFree(pl->d_pitems);
                if (pl->h_pitems) (void)hipHostFree(pl->h_pitems);
                pl->d_pitems = nullptr; pl->h_pitems = nullptr; pl->pitems_cap = 0; pl->pitems_key.clear();
                const size_t cap = np * 2;
                if (hipMalloc(&pl->d_pitems, cap * sizeof(PItem)) != hipSuccess || hipHostMalloc((void **)&pl->h_pitems, cap * sizeof(PItem), hipHostMallocDefault) != hipSuccess)
                    return fail(TQGPU_ENOMEM, "allocation of the batch descriptors failed");
                pl->pitems_cap = (int)cap;
            }
            std::vector<unsigned long> key;
            for (int k : pm) key.push_back(solvers[k]->uid);
            hipStream_t st0 = pl->stream;
            if (key != pl->pitems_key) {
                /* the descriptors are what the single launches pass as kernel arguments; they do not change from solve to solve, so
                 * the device copy is refreshed only when the batch is composed of other mirrors than last time */
                HIP_TRY(hipStreamSynchronize(st0));                 /* a previous batch launch may still read the array */
                for (size_t m = 0; m < np; m++) { const tqgpu_solver *sm = solvers[pm[m]]; pl->h_pitems[m].C = sm->pconst; pl->h_pitems[m].Gm = sm->geom; pl->h_pitems[m].Sy = sm->psync; }
                HIP_TRY(hipMemcpyAsync(pl->d_pitems, pl->h_pitems, np * sizeof(PItem), hipMemcpyHostToDevice, st0));
                pl->pitems_key = key;
            }
            /* what the other members enqueued on their own streams since their last synchronisation (asynchronous uploads, constant
             * packing: first solves and changed data only) comes before the launch.  No per-member event traffic in the steady state:
             * a pair of calls per member and step was most of a step with 22 small trees. */
            for (size_t m = 1; m < np; m++) {
                tqgpu_solver *sm = solvers[pm[m]];
                if (sm->stream_pending) { HIP_TRY(hipStreamSynchronize(sm->stream)); sm->stream_pending = false; }
            }
            for (int k : pm) solvers[k]->batch_stream = st0;
            int rcb = launch_persist_batch(pl, batch_kernel_index(pl), pl->d_pitems, cx[(size_t)pm[0]].O, (int)np, pseq);
            if (rcb != TQGPU_OK) { for (int k : pm) solvers[k]->batch_stream = nullptr; return rcb; }
        }
        for (int k = begun_from; k < ok_to; k++) {
            tqgpu_solver *sk = solvers[k];
            int rc = solve_end(sk, o, cx[(size_t)k], &results[k]);
            if (rc == TQGPU_ETIMEOUT) {
                /* device shared: redone on its own, see tqgpu_solve.  The redo works on the same device state as the batch launch,
                 * whose later workgroups may not even have started: the launch has to be over before the sticky word is cleared */
                if (sk->batch_stream) HIP_TRY(hipStreamSynchronize(sk->batch_stream));
                rc = solve_after_timeout(sk, o, &results[k]);
            }
            if (rc != TQGPU_OK && first_err == TQGPU_OK) { first_err = rc; first_msg = g_err; }
        }
        for (int k = begun_from; k < begun_to; k++) solvers[k]->batch_stream = nullptr;
        /* the members' later work (solution export, the next solve) runs on their own streams: it has to find the batch launch
         * complete -- every verdict is in, so this waits for the write-back of the last workgroups only, once per batch */
        if (!pm.empty() && ok_to == begun_to) {
            if (first_err != TQGPU_OK || env_batch_sync) HIP_TRY(hipStreamSynchronize(solvers[pm[0]]->stream));
            else for (size_t m = 1; m < pm.size(); m++) solvers[pm[m]]->settle_stream = solvers[pm[0]]->stream;      /* see settle() */
        }
        if (gp_launched) {
            if (first_err != TQGPU_OK || env_batch_sync) HIP_TRY(hipStreamSynchronize(lead->stream));
            else for (size_t m = 1; m < gi; m++) solvers[gp_members[m]]->settle_stream = lead->stream;
        }
        if (first_err != TQGPU_OK) break;
        i = j;
    }
    if (first_err != TQGPU_OK) return fail(first_err, first_msg);
    return TQGPU_OK;
}

extern "C" int tqgpu_solve_batch_n(tqgpu_solver **solvers, int n, const tqgpu_opts *o, int steps, tqgpu_result *results, long *iter_sum, long *ls_sum, long *launch_sum) {
    if (steps < 1) return fail(TQGPU_EINVAL, "tqgpu_solve_batch_n: bad arguments");
    long it = 0, ls = 0, la = 0;
    for (int k = 0; k < steps; k++) {
        const int rc = tqgpu_solve_batch(solvers, n, o, results);
        if (rc != TQGPU_OK) return rc;
        for (int i = 0; i < n; i++) { it += results[i].iter; ls += results[i].ls_total; la += results[i].n_launches; }
    }
    if (iter_sum) *iter_sum = it;
    if (ls_sum) *ls_sum = ls;
    if (launch_sum) *launch_sum = la;
    return TQGPU_OK;
}

/* ============================================================================================ */
/* sharded mode: one tree over several devices                                  */
/* ============================================================================================ */

namespace {

/* enumerate the (array, first element, count per rank) ranges that are rank-partitioned, one per
 * node level >= lb; used for the final solution gather */
struct RangeSpec { double *base; size_t per_rank; };

std::vector<RangeSpec> solution_ranges(tqgpu_solver *s) {
    std::vector<RangeSpec> out;
    const int MD = s->fMD, NX = s->fNX, NU = s->fNU, N = s->nranks;
    const int lb = s->tier_l0[s->part_top];
    const Data &D = s->D;
    double *lamc = s->h_ctrl->cur ? D.lam1 : D.lam0;
    for (int l = lb; l <= s->Nh; l++) {
        int wl = 1; for (int i = 0; i < l; i++) wl *= MD;
        const size_t f0 = (size_t)uni_first(MD, l), w = (size_t)wl / N;
        double *xs[] = {D.x, D.xUnc, D.xUncS, lamc, D.dlam};
        for (double *a : xs) out.push_back({a + NX * f0, w * NX});
        if (l < s->Nh) { double *us[] = {D.u, D.uUnc, D.uUncS}; for (double *a : us) out.push_back({a + NU * f0, w * NU}); }
    }
    return out;
}

/* The partition of a uniform tree over `N` ranks, host arithmetic only: tiers whose subtree count is a multiple of N
 * are partitioned by contiguous subtree ranges, the tiers above are replicated.  Used by shard_build_lists and exported as
 * tqgpu_shard_plan (the CPU tests compare it with the Python planner the gloo protocol tests use). */
struct ShardPlanHost {
    int part_top = -1, lb = 0, gh_counted = 0;
    int bnd_b0 = 0, bnd_bn = 0, bnd_own0 = 0, bnd_ownn = 0;
    std::vector<int> gh, nodes, nodes_cnt, blks;
};

int shard_plan_host(int MD, int NX, int Nh, const std::vector<int> &tier_l0, const std::vector<int> &tier_grid, int N, int r, ShardPlanHost &P) {
    const int nt = (int)tier_l0.size();
    /* highest partitioned tier: subtree count divisible by the number of ranks */
    P.part_top = -1;
    for (int i = 0; i < nt - 1; i++) if (tier_grid[i] % N == 0 && tier_grid[i] >= N) P.part_top = i;
    if (P.part_top < 0) return fail(TQGPU_EUNSUPPORTED, "tree too small to shard over this many ranks");
    const int lb = tier_l0[P.part_top], l00 = tier_l0[0];
    P.lb = lb;
    auto width = [&](int l) { int w = 1; for (int i = 0; i < l; i++) w *= MD; return w; };
    {
        const int gb = width(lb), w = gb / N;
        P.bnd_b0 = NX * uni_first(MD, lb); P.bnd_bn = NX * gb; P.bnd_own0 = NX * r * w; P.bnd_ownn = NX * w;
    }
    std::vector<int> &gh = P.gh, &nodes = P.nodes, &nodes_cnt = P.nodes_cnt, &blks = P.blks;
    gh.clear(); nodes.clear(); nodes_cnt.clear(); blks.clear();
    /* owned */
    for (int l = lb; l <= Nh; l++) {
        const int w = width(l) / N, f0 = uni_first(MD, l) + r * w;
        for (int i = 0; i < w; i++) {
            nodes.push_back(f0 + i); nodes_cnt.push_back(f0 + i);
            if (l < Nh) blks.push_back(f0 + i);
            if (l < l00) gh.push_back(f0 + i);
        }
    }
    P.gh_counted = (int)gh.size();
    /* replicated (levels above the boundary): computed by every rank, counted by rank 0 only */
    for (int l = 0; l < lb; l++) {
        const int w = width(l), f0 = uni_first(MD, l);
        for (int i = 0; i < w; i++) {
            nodes.push_back(f0 + i); gh.push_back(f0 + i);
            if (r == 0) { nodes_cnt.push_back(f0 + i); blks.push_back(f0 + i); }
        }
    }
    if (r == 0) P.gh_counted = (int)gh.size();
    return TQGPU_OK;
}

int shard_build_lists(tqgpu_solver *s) {
    const int N = s->nranks;
    ShardPlanHost P;
    int rcp = shard_plan_host(s->fMD, s->fNX, s->Nh, s->tier_l0, s->tier_grid, N, s->rank, P);
    if (rcp) return rcp;
    s->part_top = P.part_top;
    s->bnd_b0 = P.bnd_b0; s->bnd_bn = P.bnd_bn; s->bnd_own0 = P.bnd_own0; s->bnd_ownn = P.bnd_ownn;
    std::vector<int> &gh = P.gh, &nodes = P.nodes, &nodes_cnt = P.nodes_cnt, &blks = P.blks;
    s->gh_counted = P.gh_counted;
    s->gh_n = (int)gh.size(); s->n_nodes = (int)nodes.size(); s->n_nodes_counted = (int)nodes_cnt.size(); s->n_blk_counted = (int)blks.size();
    const size_t ints = gh.size() + nodes.size() + nodes_cnt.size() + blks.size() + 16;
    const size_t bytes = ints * sizeof(int) + (3 * (size_t)N + 8) * sizeof(double) + 1024;
    if (s->shard_slab) { (void)hipFree(s->shard_slab); s->shard_slab = nullptr; }
    HIP_TRY(hipMalloc(&s->shard_slab, bytes));
    HIP_TRY(hipMemset(s->shard_slab, 0, bytes));
    char *p = static_cast<char *>(s->shard_slab);
    s->d_xerr = reinterpret_cast<double *>(p); p += sizeof(double) * (size_t)(N + 2);
    s->d_xs = reinterpret_cast<double *>(p); p += sizeof(double) * (size_t)(2 * N + 2);
    auto put = [&](std::vector<int> &v, int *&dst) -> int {
        dst = reinterpret_cast<int *>(p); p += sizeof(int) * (v.size() + 2);
        if (!v.empty() && hipMemcpy(dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(TQGPU_ENODEVICE, "shard list upload failed");
        return TQGPU_OK;
    };
    int rc;
    if ((rc = put(gh, s->d_gh_list)) || (rc = put(nodes, s->d_node_list)) || (rc = put(nodes_cnt, s->d_node_cnt_list)) || (rc = put(blks, s->d_blk_list))) return rc;
    return TQGPU_OK;
}

}  // namespace

extern "C" int tqgpu_shard_unique_id(void *id128) {
    if (!id128) return fail(TQGPU_EINVAL, "null id buffer");
    int rc = rccl_load();
    if (rc) return rc;
    RcclApi::UniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return TQGPU_OK;
}

/* the partition plan without a device (host arithmetic of shard_build_lists): uniform complete md-ary tree with Nh block levels, tiers
 * of the fused path (3 levels for md = 2, 2 for md <= 4, else 1).  Lists may be NULL / caps 0 to query the sizes only. */
extern "C" int tqgpu_shard_plan(int md, int nx, int Nh, int nranks, int rank, int *part_top, int *boundary_level, int *gh_counted,
                                int *gh_list, int gh_cap, int *gh_n, int *owned_nodes, int owned_cap, int *owned_n) {
    if (md < 2 || Nh < 2 || nranks < 1 || rank < 0 || rank >= nranks) return fail(TQGPU_EINVAL, "tqgpu_shard_plan: bad arguments");
    const int TH = md == 2 ? 3 : (md <= 4 ? 2 : 1);
    std::vector<int> l0v, gridv;
    const int nt = (Nh + TH - 1) / TH;
    for (int i = 0; i < nt; i++) {
        const int l1 = Nh - i * TH, l0 = std::max(0, l1 - TH);
        int grid = 1;
        for (int l = 0; l < l0; l++) grid *= md;
        l0v.push_back(l0); gridv.push_back(grid);
    }
    ShardPlanHost P;
    int rc = shard_plan_host(md, nx, Nh, l0v, gridv, nranks, rank, P);
    if (rc) return rc;
    if (part_top) *part_top = P.part_top;
    if (boundary_level) *boundary_level = P.lb;
    if (gh_counted) *gh_counted = P.gh_counted;
    if (gh_n) *gh_n = (int)P.gh.size();
    if (gh_list) for (int i = 0; i < (int)P.gh.size() && i < gh_cap; i++) gh_list[i] = P.gh[(size_t)i];
    /* owned nodes = the partitioned part of the node list (levels >= boundary), in level order */
    int no = 0;
    const int first_repl = uni_first(md, P.lb);
    for (int v : P.nodes) if (v >= first_repl) { if (owned_nodes && no < owned_cap) owned_nodes[no] = v; no++; }
    if (owned_n) *owned_n = no;
    return TQGPU_OK;
}

extern "C" int tqgpu_shard_init(tqgpu_solver *s, int rank, int nranks, const void *id128) {
    SETTLE(s);
    if (!s || nranks < 1 || rank < 0 || rank >= nranks) return fail(TQGPU_EINVAL, "tqgpu_shard_init: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    if (s->comm) { (void)g_rccl.CommDestroy(s->comm); s->comm = nullptr; }      /* a second call replaces the communicator, it does not leak it */
    /* one rank WITHOUT a communicator = back to the unsharded mirror; one rank WITH one = the sharded code path on a single device
     * (every exchange is a one-rank in-place all-gather): exercises rccl_load / ncclCommInitRank / ncclAllGather where only one GPU exists */
    if (nranks == 1 && !id128) { s->nranks = 1; s->rank = 0; s->sharded = false; return TQGPU_OK; }
    if (s->fast < 0 || !s->use_fast || s->mstage) return fail(TQGPU_EUNSUPPORTED, "sharding needs the fused uniform-tree path");
    s->nranks = nranks; s->rank = rank; s->sharded = true;
    s->export_valid = false;
    int rc = shard_build_lists(s);
    if (rc) { s->nranks = 1; s->rank = 0; s->sharded = false; return rc; }
    if (id128) {
        if ((rc = rccl_load())) { s->nranks = 1; s->rank = 0; s->sharded = false; return rc; }
        RcclApi::UniqueId id;
        memcpy(&id, id128, sizeof(id));
        NCCL_TRY(g_rccl.CommInitRank(&s->comm, nranks, id, rank));
    }
    return TQGPU_OK;
}

/* after a sharded solve every rank holds valid x,u,lambda,... only for its own and the replicated
 * nodes: gather the partitioned ranges so that tqgpu_get_solution returns the full solution */
extern "C" int tqgpu_shard_gather_solution(tqgpu_solver *s) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    if (!s->sharded) return TQGPU_OK;
    if (!s->comm) return fail(TQGPU_ECOMM, "no communicator (virtual ranks gather through tqgpu_solve_virtual_ranks)");
    HIP_TRY(hipSetDevice(s->device));
    auto ranges = solution_ranges(s);
    NCCL_TRY(g_rccl.GroupStart());
    for (auto &rg : ranges) NCCL_TRY(g_rccl.AllGather(rg.base + (size_t)s->rank * rg.per_rank, rg.base, rg.per_rank, NCCL_DOUBLE, s->comm, s->stream));
    NCCL_TRY(g_rccl.GroupEnd());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TQGPU_OK;
}

/* ============================================================================================ */
/* ONE tree over several devices INSIDE the persistent launch                                   */
/* ============================================================================================ */
/* The single-device persistent launch is a set of workgroups that talk through tagged words in one slab (tdunes_persist.hpp).
 * Sharded, the SAME set of workgroups is dealt over one launch per device: tiers whose subtree count is a multiple of the number of
 * ranks go to the ranks by contiguous subtree ranges, the tiers above them to rank 0 -- no workgroup exists twice,
 * so there is nothing to keep consistent but the words themselves.  Every rank has a slab of the same layout; a producer writes
 * each word into every slab (system-scope stores into peer-mapped memory), a consumer polls its own.  What crosses devices per
 * Newton iteration: the Schur records and x / QinvCal of the boundary subtree roots and every workgroup's termination and
 * dual-function partials upwards, the step of the boundary blocks, the line-search commands and the halt word downwards; no
 * collective, no host in the loop.  RCCL (or any transport the caller has: tqgpu_pshard_pack / _unpack) is only used to collect
 * the solution afterwards.  Every rank must call tqgpu_pshard_solve the same number of times (the launch number tags the words). */
extern "C" int tqgpu_pshard_init(tqgpu_solver *s, int rank, int nranks) {
    SETTLE(s);
    if (!s || nranks < 1 || nranks > 8 || rank < 0 || rank >= nranks) return fail(TQGPU_EINVAL, "tqgpu_pshard_init: bad arguments (1 .. 8 ranks)");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->persist_ok || s->mstage || s->fast < 0 || !s->use_fast || !s->use_persist) return fail(TQGPU_EUNSUPPORTED, "sharding inside the persistent launch needs the persistent path of a uniform complete tree");
    if (!shard_instantiated(s->fast)) return fail(TQGPU_EUNSUPPORTED, "the sharded persistent kernel is not instantiated for this shape (SHARD_TABLE)");
    switch (s->fast) {
#define X(idx, nx, nu, md) case idx: { int rca = allow_lds(f_persist_sh<nx, nu, md, 1>, s->lds_persist); if (!rca) rca = allow_lds(f_persist_sh<nx, nu, md, 0>, s->lds_persist); if (rca) return rca; } break;
        SHARD_TABLE(X)
#undef X
        default: break;
    }
    int nwg = 0, top = -1;
    if (tqgpu_pshard_plan(s->fMD, s->Nh, nranks, rank, nullptr, 0, &nwg, &top, nullptr) != 0) return fail(TQGPU_EUNSUPPORTED, "tree too small to shard over this many ranks");
    std::vector<int> map((size_t)std::max(nwg, 1));
    (void)tqgpu_pshard_plan(s->fMD, s->Nh, nranks, rank, map.data(), nwg, &nwg, &top, nullptr);
    map.resize((size_t)nwg);
    if (nranks == 1) top = s->n_tiers - 1;
    if ((int)map.size() > s->co_capacity) return fail(TQGPU_EUNSUPPORTED, "this rank's workgroups cannot all be resident");
    if (s->ps_wg_map) { (void)hipFree(s->ps_wg_map); s->ps_wg_map = nullptr; }
    HIP_TRY(hipMalloc(&s->ps_wg_map, sizeof(int) * std::max<size_t>(map.size(), 1)));
    HIP_TRY(hipMemcpy(s->ps_wg_map, map.data(), sizeof(int) * map.size(), hipMemcpyHostToDevice));
    s->ps_G = (int)map.size();
    /* First contact with real xGMI is somebody else's run, so the two things a peer's writes depend on are settled by construction:
     * (a) the slab the peers write into is FINE-GRAINED memory (coarse-grained hipMalloc memory is only guaranteed coherent across
     *     devices at kernel boundaries; this kernel polls it while the peers write), and
     * (b) the polls are system-scope loads (ps_sys: f_persist_sh<.., 1>, compiled with TQ_LD_SCOPE = system).
     * TREEQP_AMD_PSHARD_COARSE=1 / TREEQP_AMD_PSHARD_AGENT=1 restore the single-device combination for an A/B on a node. */
    s->ps_sys = !getenv("TREEQP_AMD_PSHARD_AGENT");
    if (!s->ps_fine && !getenv("TREEQP_AMD_PSHARD_COARSE")) {
        void *fresh = nullptr;
        hipError_t ef = hipExtMallocWithFlags(&fresh, s->sync_bytes, hipDeviceMallocFinegrained);
        if (ef != hipSuccess) return fail(TQGPU_ENODEVICE, std::string("hipExtMallocWithFlags(fine-grained hand-over slab): ") + hipGetErrorString(ef));
        HIP_TRY(hipStreamSynchronize(s->stream));
        char *ob = static_cast<char *>(s->sync_slab), *nb = static_cast<char *>(fresh);
        auto mv = [&](auto *&q) { if (q) q = reinterpret_cast<std::remove_reference_t<decltype(q)>>(nb + (reinterpret_cast<char *>(q) - ob)); };
        PSync &Y = s->psync;
        mv(Y.sch); mv(Y.dlt); mv(Y.ndt); mv(Y.parts); mv(Y.errs); mv(Y.cmd); mv(Y.vrd); mv(Y.bparts); mv(Y.sgt); mv(Y.rfl); mv(Y.halt); mv(Y.timeout); mv(Y.base); mv(Y.verdict); mv(Y.anc);
        (void)hipFree(s->sync_slab);
        s->sync_slab = fresh;
        s->ps_fine = true;
        s->pitems_key.clear();                                             /* (a cached batch descriptor would hold the old slab) */
    }
    s->export_valid = false;
    s->pshard = true; s->rank = rank; s->nranks = nranks; s->part_top = nranks > 1 ? top : -1;
    s->psync.npeer = nranks;
    s->psync.anc_local = (nranks == 1 || top >= 1) ? 1 : 0;          /* tiers 0 .. top are dealt over the ranks by contiguous subtree ranges: with top >= 1 a tier-1 workgroup sits with the bottom-tier workgroups below it */
    s->psync.relay_wg = (rank > 0 && !map.empty()) ? map[0] : -1;          /* ranks without the top workgroup: their first workgroup passes the verdict on to the host */
    for (int r = 0; r < 8; r++) s->h_peers[r] = s->psync.base;             /* until connected: own slab */
    HIP_TRY(hipMemcpy(s->d_peers, s->h_peers, sizeof(s->h_peers), hipMemcpyHostToDevice));
    s->psync.nap = nap_for_grid(s->ps_G);
    s->launch_no = 0;
    HIP_TRY(hipMemsetAsync(s->sync_slab, 0, s->sync_bytes, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TQGPU_OK;
}

/* peer `r` lives in this process (several mirrors on one device, or on several devices with peer access enabled by the caller) */
extern "C" int tqgpu_pshard_connect_local(tqgpu_solver *s, int r, tqgpu_solver *peer) {
    SETTLE(s);
    if (!s || !peer || !s->pshard || r < 0 || r >= s->nranks || !peer->sync_slab || peer->sync_bytes != s->sync_bytes) return fail(TQGPU_EINVAL, "tqgpu_pshard_connect_local: bad arguments");
    if (peer->device != s->device) {
        HIP_TRY(hipSetDevice(s->device));
        hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(TQGPU_ECOMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
    }
    s->h_peers[r] = static_cast<unsigned long long *>(peer->sync_slab);
    HIP_TRY(hipMemcpy(s->d_peers, s->h_peers, sizeof(s->h_peers), hipMemcpyHostToDevice));
    return TQGPU_OK;
}
/* peers in other processes: an IPC handle of this rank's slab (64 bytes) out, the peers' handles in */
extern "C" int tqgpu_pshard_ipc_export(tqgpu_solver *s, void *handle64) {
    if (!s || !handle64 || !s->sync_slab) return fail(TQGPU_EINVAL, "tqgpu_pshard_ipc_export: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipError_t e = hipIpcGetMemHandle(&h, s->sync_slab);
    if (e != hipSuccess) return fail(TQGPU_ECOMM, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
    memcpy(handle64, &h, 64);
    return TQGPU_OK;
}
extern "C" int tqgpu_pshard_ipc_connect(tqgpu_solver *s, int r, const void *handle64) {
    if (!s || !handle64 || !s->pshard || r < 0 || r >= s->nranks || r == s->rank) return fail(TQGPU_EINVAL, "tqgpu_pshard_ipc_connect: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void *ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(TQGPU_ECOMM, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
    if (s->ps_ipc[r]) (void)hipIpcCloseMemHandle(s->ps_ipc[r]);
    s->ps_ipc[r] = ptr;
    s->h_peers[r] = static_cast<unsigned long long *>(ptr);
    HIP_TRY(hipMemcpy(s->d_peers, s->h_peers, sizeof(s->h_peers), hipMemcpyHostToDevice));
    return TQGPU_OK;
}

/* one solve in two halves (so that one process can drive several ranks): _begin enqueues this rank's launch, _end waits for the
 * verdict.  All ranks' launches have to be in flight together: they wait for each other (bounded: 0.5 s, then TQGPU_ETIMEOUT). */
extern "C" int tqgpu_pshard_begin(tqgpu_solver *s, const tqgpu_opts *o) {
    SETTLE(s);
    if (!s || !o || !s->pshard) return fail(TQGPU_EINVAL, "tqgpu_pshard_begin: not a sharded mirror");
    HIP_TRY(hipSetDevice(s->device));
    if (o->profile || o->maxIter <= 0 || o->checkLastActiveSet == 2) return fail(TQGPU_EUNSUPPORTED, "sharded persistent solve: default solve options only (no profiling, no factor keeping)");
    Opts O;
    O.maxIter = o->maxIter; O.termCondition = o->termCondition; O.regType = o->regType;
    O.lsMaxIter = o->lineSearchMaxIter; O.lsRestartTrigger = o->lineSearchRestartTrigger; O.reuse = 0;
    O.tol = o->stationarityTolerance; O.regTol = o->regTol; O.regValue = o->regValue;
    O.gamma = o->lineSearchGamma; O.beta = o->lineSearchBeta; O.stamps = 0;
    if (O.termCondition < 0 || O.termCondition > 2 || O.regType < 0 || O.regType > 2 || O.regValue < 0) return fail(TQGPU_EINVAL, "invalid option value");
    if (((s->launch_no + 1) & 0xFFFFu) == 0)
        return fail(TQGPU_EUNSUPPORTED, "65535 sharded solves since the launch numbers were last reset: call tqgpu_pshard_rewind on every rank, between two barriers of the caller's "
                                        "(the 16-bit launch number tags the hand-over words; a single device wipes its slab when it wraps, ranks that write into each other's slabs cannot do that on their own)");
    memset(s->h_res, 0, sizeof(HostRes));
    int launches = 0;
    s->solve_no++;
    s->export_valid = false;
    return launch_persist(s, O, launches, 1);
}
/* launch numbers back to zero and the slab wiped; peers stay connected.  EVERY rank, with no sharded solve in flight anywhere (a barrier
 * of the caller's before and after): a peer's launch that is still running, or already running again, writes into the slab being wiped */
extern "C" int tqgpu_pshard_rewind(tqgpu_solver *s) {
    SETTLE(s);
    if (!s || !s->pshard) return fail(TQGPU_EINVAL, "tqgpu_pshard_rewind: not a sharded mirror");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    /* (hipMemset of device memory may return before the wipe has happened; it runs on the null stream, which the solver's non-blocking
     * stream does not wait for: the wipe goes on the solver's stream and is waited for -- found by the 100 000-solve soak, where the
     * first C3 solve after a rewind lost words to the wipe) */
    HIP_TRY(hipMemsetAsync(s->sync_slab, 0, s->sync_bytes, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->launch_no = 0;
    return TQGPU_OK;
}
extern "C" int tqgpu_pshard_end(tqgpu_solver *s, tqgpu_result *res) {
    SETTLE(s);
    if (!s || !res || !s->pshard) return fail(TQGPU_EINVAL, "tqgpu_pshard_end: not a sharded mirror");
    HIP_TRY(hipSetDevice(s->device));
    /* the verdict reaches every rank's host through its pinned result block: written by the top workgroup (rank 0) or passed on from the
     * rank's slab by its relay workgroup -- no stream synchronisation on the way (the state write-back of the other workgroups is still
     * running; everything the host does next on this mirror is stream-ordered behind the launch) */
    {
        volatile unsigned *seq = &s->h_res->seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0; *seq != s->psync.seq; spins++) {
            if (take_tagged_block(s->h_res, s->psync.seq)) break;          /* (rank 0: the top workgroup's tagged words; the other ranks' relay workgroups post the plain block) */
            __builtin_ia32_pause();
            if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (s->h_res->seq != s->psync.seq) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        unsigned tmo = 0;
        HIP_TRY(hipMemcpy(&tmo, s->psync.timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (tmo) {
            HIP_TRY(hipMemsetAsync(s->psync.timeout, 0, sizeof(unsigned), s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            return fail(TQGPU_ETIMEOUT, "sharded persistent solve: a bounded wait for another rank's workgroups timed out (are all ranks' launches in flight together?)");
        }
        return fail(TQGPU_ECOMM, "sharded persistent solve: no verdict from the top workgroup");
    }
    const Ctrl &c = *s->h_ctrl;
    if (!c.done) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        unsigned tmo = 0;
        HIP_TRY(hipMemcpy(&tmo, s->psync.timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (tmo) {
            HIP_TRY(hipMemsetAsync(s->psync.timeout, 0, sizeof(unsigned), s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            return fail(TQGPU_ETIMEOUT, "sharded persistent solve: a bounded wait for another rank's workgroups timed out");
        }
        return fail(TQGPU_EUNSUPPORTED, "sharded persistent solve: the launch ended without a verdict (tag space exhausted: more than 60000 passes)");
    }
    res->status = c.status; res->iter = c.iter; res->ls_total = c.ls_total; res->ls_last = c.ls_last;
    res->n_launches = 1; res->device_time = s->rank == 0 ? 1e-8 * (double)(s->h_res->t_end - s->h_res->t_start) : 0.0; res->last_error_norm = c.err; res->last_fval = c.fval;
    s->last_iter = c.iter;
    return TQGPU_OK;
}

/* what this rank holds of the solution (its chunk of every partitioned range; rank 0: also the levels above the partition), packed
 * into a host buffer, and the inverse: the transport hook for callers that collect the solution themselves (any all-gather of
 * equal-sized buffers: tqgpu_pshard_pack_size is the same on every rank) */
namespace {
struct PsRange { double *base; size_t n; };
std::vector<PsRange> pshard_owned_ranges(tqgpu_solver *s, int rank) {
    std::vector<PsRange> out;
    const int MD = s->fMD, NX = s->fNX, NU = s->fNU, N = s->nranks;
    const Data &D = s->D;
    double *lamc = s->h_ctrl->cur ? D.lam1 : D.lam0;
    const int lb = N > 1 ? s->tier_l0[s->part_top] : s->Nh + 1;
    for (int l = 0; l <= s->Nh; l++) {
        int wl = 1; for (int i = 0; i < l; i++) wl *= MD;
        const size_t f0 = (size_t)uni_first(MD, l);
        /* node data (x, u, ...) of level l belongs to the workgroup that owns the node; the duals of a node's own edge belong to the
         * BLOCK of its parent, one level up: the duals of the boundary level are the top tiers' (rank 0) */
        for (int what = 0; what < 2; what++) {
            const int lbw = what == 0 ? lb : lb + 1;
            size_t first, cnt;
            if (l >= lbw) { cnt = (size_t)wl / N; first = f0 + (size_t)rank * cnt; }
            else { if (rank != 0) continue; cnt = (size_t)wl; first = f0; }
            if (what == 0) {
                double *xs[] = {D.x, D.xUnc, D.xUncS};
                for (double *a : xs) out.push_back({a + NX * first, cnt * NX});
                if (l < s->Nh) { double *us[] = {D.u, D.uUnc, D.uUncS}; for (double *a : us) out.push_back({a + NU * first, cnt * NU}); }
            } else {
                double *ls[] = {lamc, D.dlam};
                for (double *a : ls) out.push_back({a + NX * first, cnt * NX});
            }
        }
    }
    return out;
}
}  // namespace
extern "C" long tqgpu_pshard_pack_size(tqgpu_solver *s) {
    if (!s || !s->pshard) return -1;
    size_t n = 0;
    for (auto &rg : pshard_owned_ranges(s, 0)) n += rg.n;          /* rank 0 holds the most */
    return (long)n;
}
extern "C" int tqgpu_pshard_pack(tqgpu_solver *s, double *out, long cap) {
    SETTLE(s);
    if (!s || !out || !s->pshard) return fail(TQGPU_EINVAL, "tqgpu_pshard_pack: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    /* tqgpu_pshard_end returns on the verdict word while the workgroups still write their state back, and the copies below run on the
     * null stream, which the solver's non-blocking stream does not order: wait for the launch first */
    HIP_TRY(hipStreamSynchronize(s->stream));
    size_t o = 0;
    for (auto &rg : pshard_owned_ranges(s, s->rank)) {
        if ((long)(o + rg.n) > cap) return fail(TQGPU_EINVAL, "tqgpu_pshard_pack: buffer too small");
        HIP_TRY(hipMemcpy(out + o, rg.base, sizeof(double) * rg.n, hipMemcpyDeviceToHost));
        o += rg.n;
    }
    return TQGPU_OK;
}
extern "C" int tqgpu_pshard_unpack(tqgpu_solver *s, int src_rank, const double *in, long n_in) {
    SETTLE(s);
    if (!s || !in || !s->pshard || src_rank < 0 || src_rank >= s->nranks) return fail(TQGPU_EINVAL, "tqgpu_pshard_unpack: bad arguments");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));          /* (as tqgpu_pshard_pack: this rank's own write-back must not land on top of the peers' shares) */
    size_t o = 0;
    for (auto &rg : pshard_owned_ranges(s, src_rank)) {
        if ((long)(o + rg.n) > n_in) return fail(TQGPU_EINVAL, "tqgpu_pshard_unpack: buffer too small");
        HIP_TRY(hipMemcpy(rg.base, in + o, sizeof(double) * rg.n, hipMemcpyHostToDevice));
        o += rg.n;
    }
    return TQGPU_OK;
}

/* n mirrors of the SAME problem in this process (one device: a rehearsal with real concurrency -- n launches on n streams that wait
 * for each other inside the kernels -- or n devices with peer access): connect, solve, collect the solution into every mirror */
extern "C" int tqgpu_pshard_solve_local(tqgpu_solver **R, int n, const tqgpu_opts *o, tqgpu_result *res) {
    for (int r_ = 0; R && r_ < n; r_++) SETTLE(R[r_]);
    if (!R || n < 1 || n > 8 || !o || !res) return fail(TQGPU_EINVAL, "tqgpu_pshard_solve_local: bad arguments");
    for (int r = 0; r < n; r++) if (!R[r] || !R[r]->pshard || R[r]->nranks != n || R[r]->rank != r) return fail(TQGPU_EINVAL, "tqgpu_pshard_solve_local: mirror r must be tqgpu_pshard_init(r, n)");
    for (int r = 0; r < n; r++) for (int q = 0; q < n; q++) { int rc = tqgpu_pshard_connect_local(R[r], q, R[q]); if (rc) return rc; }
    for (int r = 0; r < n; r++) HIP_TRY(hipStreamSynchronize(R[r]->stream));          /* uploads done: the launches go out back to back */
    if (R[0]->launch_no >= 0x8000u) for (int r = 0; r < n; r++) { int rc = tqgpu_pshard_rewind(R[r]); if (rc) return rc; }      /* (all ranks idle here) */
    for (int r = 0; r < n; r++) { int rc = tqgpu_pshard_begin(R[r], o); if (rc) return rc; }
    int first = TQGPU_OK;
    std::string msg;
    for (int r = 0; r < n; r++) { int rc = tqgpu_pshard_end(R[r], &res[r]); if (rc && !first) { first = rc; msg = g_err; } }
    if (first) return fail(first, msg);
    std::vector<double> buf((size_t)std::max<long>(tqgpu_pshard_pack_size(R[0]), 1));
    for (int src = 0; src < n; src++) {
        int rc = tqgpu_pshard_pack(R[src], buf.data(), (long)buf.size());
        if (rc) return rc;
        for (int dst = 0; dst < n; dst++) if (dst != src && (rc = tqgpu_pshard_unpack(R[dst], src, buf.data(), (long)buf.size()))) return rc;
    }
    return TQGPU_OK;
}

/* Diagnostic / test entry: run `n` mirrors of the SAME problem as ranks 0..n-1 of a sharded solve
 * inside one process on one device, exchanging by device copies.  Validates partition, hand-off and
 * decision logic without a multi-GPU node.  Every mirror must have been shard-initialised with
 * (rank = its index, nranks = n, id128 = NULL).  The full solution is gathered into every mirror. */
extern "C" int tqgpu_solve_virtual_ranks(tqgpu_solver **R, int n, const tqgpu_opts *o, tqgpu_result *res) {
    for (int r_ = 0; R && r_ < n; r_++) SETTLE(R[r_]);
    if (!R || n < 2 || !o || !res) return fail(TQGPU_EINVAL, "tqgpu_solve_virtual_ranks: bad arguments");
    for (int r = 0; r < n; r++) if (!R[r] || R[r]->nranks != n || R[r]->rank != r || R[r]->comm) return fail(TQGPU_EINVAL, "mirror is not virtual rank r of n");
    HIP_TRY(hipSetDevice(R[0]->device));
    Opts O;
    O.maxIter = o->maxIter; O.termCondition = o->termCondition; O.regType = o->regType;
    O.lsMaxIter = o->lineSearchMaxIter; O.lsRestartTrigger = o->lineSearchRestartTrigger; O.reuse = o->checkLastActiveSet == 2 ? 1 : 0;
    O.tol = o->stationarityTolerance; O.regTol = o->regTol; O.regValue = o->regValue;
    O.gamma = o->lineSearchGamma; O.beta = o->lineSearchBeta; O.stamps = 0;
    int launches = 0;
    for (int r = 0; r < n; r++) {
        tqgpu_solver *s = R[r];
        hipStream_t st = s->stream;
        Ctrl init; memset(&init, 0, sizeof(init));
        HIP_TRY(hipMemcpyAsync(s->D.ctrl, &init, sizeof(Ctrl), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(s->D.ls_log, 0, sizeof(int) * (size_t)s->ls_log_cap, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpyAsync(s->D.lam0, s->d_lam_init, sizeof(double) * (size_t)s->sum_nx, hipMemcpyDeviceToDevice, st));
        const int nxu = std::max(s->sum_nx, s->sum_nu);
        if (s->need_init) { hipLaunchKernelGGL(k_init, dim3((nxu + 255) / 256), dim3(256), 0, st, s->sum_nx, s->sum_nu, s->D); s->need_init = false; }
        hipLaunchKernelGGL(k_stage, dim3(s->T.Nn), dim3(WAVE), s->lds_stage, st, s->T, s->D, 0, 0, 0);
        hipLaunchKernelGGL(k_fval_init, dim3(1), dim3(256), 0, st, s->T, s->D);
    }
    int rc;
    bool finished = o->maxIter <= 0;
    int h = 0;
    while (!finished) {
        for (int r = 0; r < n; r++) launch_fast_phase(R[r], O, h, 0, launches);
        if ((rc = shard_exchange_virtual(R, n, 1))) return rc;
        for (int r = 0; r < n; r++) launch_fast_phase(R[r], O, h, 1, launches);
        if ((rc = shard_exchange_virtual(R, n, 2))) return rc;
        for (int r = 0; r < n; r++) launch_fast_phase(R[r], O, h, 2, launches);
        for (int r = 0; r < n; r++) if ((rc = read_ctrl(R[r]))) return rc;
        while (!R[0]->h_ctrl->done && R[0]->h_ctrl->ls_pending) {
            const int it = R[0]->h_ctrl->iter, t = R[0]->h_ctrl->ls_iter;
            for (int r = 0; r < n; r++) launch_trial_phase(R[r], O, true, it, t, 0, launches);
            if ((rc = shard_exchange_virtual(R, n, 2))) return rc;
            for (int r = 0; r < n; r++) launch_trial_phase(R[r], O, true, it, t, 1, launches);
            for (int r = 0; r < n; r++) if ((rc = read_ctrl(R[r]))) return rc;
        }
        for (int r = 1; r < n; r++)
            if (R[r]->h_ctrl->iter != R[0]->h_ctrl->iter || R[r]->h_ctrl->done != R[0]->h_ctrl->done || R[r]->h_ctrl->status != R[0]->h_ctrl->status)
                return fail(TQGPU_ECOMM, "virtual ranks took different decisions");
        h = R[0]->h_ctrl->iter;
        finished = R[0]->h_ctrl->done != 0;
    }
    /* gather the partitioned solution ranges into every mirror */
    for (int r = 0; r < n; r++) HIP_TRY(hipStreamSynchronize(R[r]->stream));
    if (o->maxIter > 0) {
        for (int src = 0; src < n; src++) {
            auto rs = solution_ranges(R[src]);
            for (int dst = 0; dst < n; dst++) {
                if (dst == src) continue;
                auto rd = solution_ranges(R[dst]);
                for (size_t i = 0; i < rs.size(); i++)
                    HIP_TRY(hipMemcpyAsync(rd[i].base + (size_t)src * rd[i].per_rank, rs[i].base + (size_t)src * rs[i].per_rank, sizeof(double) * rs[i].per_rank, hipMemcpyDeviceToDevice, R[dst]->stream));
            }
        }
        for (int r = 0; r < n; r++) HIP_TRY(hipStreamSynchronize(R[r]->stream));
    }
    const Ctrl &c = *R[0]->h_ctrl;
    res->status = c.status; res->iter = c.iter; res->ls_total = c.ls_total; res->ls_last = c.ls_last;
    res->n_launches = launches; res->device_time = 0.0; res->last_error_norm = c.err; res->last_fval = c.fval;
    for (int r = 0; r < n; r++) R[r]->last_iter = c.iter;
    return TQGPU_OK;
}

extern "C" int tqgpu_get_solution(tqgpu_solver *s, double *x, double *u, double *lam, double *mu_x, double *mu_u, double *dlam) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    const Data &D = s->D;
    /* one packing kernel, one download into pinned memory, one synchronisation (ordered behind the solve on the stream) */
    const double *lamc = s->h_ctrl->cur ? D.lam1 : D.lam0;
    const int nxe = s->sum_nx - s->x_pad, nue = s->sum_nu, nl = s->sum_lam;
    if (!s->export_valid) { int rce = enqueue_export(s, lamc); if (rce != TQGPU_OK) return rce; }      /* (else: went out behind the solve, tqgpu_set_export_ahead) */
    HIP_TRY(hipStreamSynchronize(st));
    const double *ox = s->h_out, *ou = ox + nxe, *ol = ou + nue, *od = ol + nl, *omx = od + nl, *omu = omx + nxe;
    if (x && nxe > 0) memcpy(x, ox, sizeof(double) * (size_t)nxe);
    if (u && nue > 0) memcpy(u, ou, sizeof(double) * (size_t)nue);
    if (lam && nl > 0) memcpy(lam, ol, sizeof(double) * (size_t)nl);
    if (dlam && nl > 0) memcpy(dlam, od, sizeof(double) * (size_t)nl);
    if (mu_x && nxe > 0) memcpy(mu_x, omx, sizeof(double) * (size_t)nxe);
    if (mu_u && nue > 0) memcpy(mu_u, omu, sizeof(double) * (size_t)nue);
    return TQGPU_OK;
}

extern "C" int tqgpu_get_iteration_log(tqgpu_solver *s, int *ls_iters, double *iter_times, int cap) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    const int n = std::min(std::min(cap, s->last_iter), s->ls_log_cap);
    if (n > 0 && ls_iters) {
        /* fetched on request (stream-ordered behind the solve), not on every solve */
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipMemcpyAsync(s->h_ls_log, s->D.ls_log, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    for (int i = 0; i < n; i++) {
        if (ls_iters) ls_iters[i] = s->h_ls_log[i];
        if (iter_times) iter_times[i] = i < (int)s->iter_times.size() ? s->iter_times[i] : NAN;
    }
    return TQGPU_OK;
}

/* profile level 3: per Newton iteration the device time of the reference's phases (profiling.h:58-67) on the launch-per-level path:
 * build_dual = gradient, termination test, dual Hessian; newton_direction = backward factorisation + forward substitution;
 * line_search = direction test + first trial sweep + Armijo decision (further trials of a backtracking search are enqueued after the
 * read-back and are NOT in this figure).  stage_qps[0] = the first sweep of the solve; phase S of every later iteration IS the accepted
 * trial sweep of the line search before it, so stage_qps[i > 0] = 0.  NaN where nothing was recorded (opts.profile < 3). */
extern "C" int tqgpu_get_phase_log(tqgpu_solver *s, double *stage_qps, double *build_dual, double *newton_direction, double *line_search, int cap) {
    SETTLE(s);
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    const int n = std::min(cap, s->last_iter);
    for (int i = 0; i < n; i++) {
        const bool have = (size_t)(3 * i + 2) < s->phase_times.size();
        if (stage_qps) stage_qps[i] = i == 0 ? s->first_sweep_time : (std::isnan(s->first_sweep_time) ? NAN : 0.0);
        if (build_dual) build_dual[i] = have ? s->phase_times[(size_t)(3 * i)] : NAN;
        if (newton_direction) newton_direction[i] = have ? s->phase_times[(size_t)(3 * i + 1)] : NAN;
        if (line_search) line_search[i] = have ? s->phase_times[(size_t)(3 * i + 2)] : NAN;
    }
    return n;
}

/* Device times (HIP events on the solver's stream, first to last enqueued operation of a solve) of the
 * last `n` solves, oldest first; synchronises the stream.  Returns the number written. */
extern "C" int tqgpu_get_device_times(tqgpu_solver *s, double *out, int n) {
    SETTLE(s);
    if (!s || !out || n < 0) return -1;
    if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) return -1;
    const long have = std::min<long>(std::min<long>(n, s->solve_no), EV_RING);
    for (long i = 0; i < have; i++) {
        const int ring = (int)((s->solve_no - have + i) % EV_RING);
        float ms = 0.f;
        out[i] = (s->ring_ok[(size_t)ring] && hipEventElapsedTime(&ms, s->ring_ev0[ring], s->ring_ev1[ring]) == hipSuccess) ? 1e-3 * ms : NAN;
    }
    return (int)have;
}

/* per-solve HIP event pairs on or off (default on).  Off: a single persistent launch is enqueued with nothing around it -- two
 * queue packets fewer per solve -- and tqgpu_get_device_times reports NaN for such solves; tqgpu_result.device_time is the
 * kernel's own clock (launch start to verdict) either way.  The other paths always record (their device time IS the pair). */
extern "C" int tqgpu_set_event_timing(tqgpu_solver *s, int on) {
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    s->ev_timing = on != 0;
    return TQGPU_OK;
}

/* Algorithmic bytes / flops of one Newton iteration (every input read once, every output written
 * once per phase; SURVEY.md §8(d) generalised to per-node dimensions; n_ls line-search trials,
 * each counted as the reference does: one dual-function sweep, plus the fval0 sweep). */
extern "C" int tqgpu_iteration_cost(const tqgpu_solver *s, int n_ls, double *bytes, double *flops) {
    if (!s) return fail(TQGPU_EINVAL, "null solver");
    double Sb = 0, Gb = 0, Hb = 0, Fb = 0, Lb = 0, Sf = 0, Gf = 0, Hf = 0, Ff = 0, Lf = 0;
    const int Nn = s->Nn;
    for (int k = 0; k < Nn; k++) {
        const double nx = s->nx[k], nu = s->nu[k], d = s->bdim[k];
        Sb += 9 * nx + 9 * nu; Lb += 6 * nx + 1 + 6 * nu;
        if (k > 0) {
            const int p = s->dad[k];
            const double AB = nx * (s->nx[p] + s->nu[p]);
            Sb += AB + 2 * nx; Gb += AB + 4 * nx; Hb += AB + nx; Lb += AB + 3 * nx;
            Sf += 2 * AB; Gf += 2 * AB; Lf += 2 * AB;
            Hf += AB + nx * (nx + 1) * (s->nx[p] + s->nu[p]);
        }
        if (k < s->Np) {
            Gb += nx + nu; Hb += nx + nu + d * (d + 1) / 2; Lb += 1.5 * d;
            Fb += 3 * d * (d + 1) / 2 + 5 * d;
            Ff += d * d * d / 3 + 2 * d * d;
            /* off-diagonal sibling blocks */
            for (int a = 0; a < s->nk[k]; a++) for (int c = 0; c < a; c++) {
                const double na = s->nx[s->kid0[k] + a], nc = s->nx[s->kid0[k] + c];
                Hf += nc * (nx + nu) + 2 * na * nc * (nx + nu);
            }
            if (k > 0) {
                Hb += nx * d; Fb += 3 * nx * d + nx * (nx + 1) + 3 * nx;
                Ff += nx * d * d + nx * (nx + 1) * d + 4 * nx * d;
            }
        }
    }
    const double sweeps = 1 + n_ls;       /* fval0 + trials */
    if (bytes) *bytes = 8.0 * (Sb + Gb + Hb + Fb + sweeps * Lb);
    if (flops) *flops = Sf + Gf + Hf + Ff + sweeps * Lf;
    return TQGPU_OK;
}
#endif  /* TQ_HAS(TQP_HOST) */
