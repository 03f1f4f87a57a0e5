/*
 * tdunes_wide3.hpp -- the launch-per-phase path in THREE launches per Newton iteration (round 3).
 *
 * Included by tdunes_device.hip after tdunes_wide.hpp.  Same global layout and the same control block as the launch-per-phase
 * kernels it stands in for; what changes is where the kernel boundaries are:
 *
 *   k_sg    stage sweep (phase S / a line-search trial) + dual gradient (phase G) of every node in ONE launch, four nodes per
 *           workgroup.  The gradient of node k needs x, u of its parent: they travel as tagged words inside the launch (workgroups
 *           are numbered parents-first and the hardware starts them in that order).  The launch's tail -- the last workgroup to
 *           post its partials -- takes the dual value (fval0 / the Armijo test of line_search, dual_Newton_tree.c:970-1000) AND the
 *           termination test of the next iteration (calculate_error_in_residuals, :412-442, :542-546): k_stage, k_fval_init /
 *           k_ls_decide, k_grad and k_check of the older protocol.  The gradient is computed from the TRIAL point before the
 *           Armijo test has accepted it; a rejected trial's gradient is overwritten by the next trial's.
 *   k_hf_w  dual Hessian (phase H) + backward sweep of phase F for blocks of 16 < d <= 64 rows, one 4-wave workgroup per block,
 *           children first (build_dual_problem :551-615, calculate_delta_lambda :668-752).  The block W = C P C' + diag is formed
 *           by MFMA straight into the LDS image of the tall matrix [W ; rhs' ; Ut] (the staging area of C is the image itself:
 *           52 KB per workgroup at d = 60, three workgroups per CU).  The blocked Cholesky looks ahead: after panel p's pivot
 *           chain (registers, one row per lane) the four waves first bring column panel p + 1 up to date, then waves 0 / 1 run
 *           panel p + 1's chain while waves 2 / 3 finish the trailing update of panel p on the matrix pipe and add panel p's
 *           contribution to the Schur complement G = Xt Xt' they keep in registers -- after the last chain only that panel's
 *           rank-16 term is left before the record goes to the parent.  The chain of the last panel only runs over the columns
 *           that exist (d = 60: 12, not 16).  The children's Schur records are polled as ONE batch.  Identity rows carried by
 *           the chains (idle lanes of wave 1) come out as the inverses of the diagonal tiles, with which the forward
 *           substitution is prepared off the critical path, as in the persistent kernels: [z0 | M] = L^-T [y | CholUt'] by
 *           MFMA tiles, so that a forward step is dlam = z0 - M dlam_dad.
 *   k_fwd3  forward sweep (:756-775) with the prepared [z0 | M]: one wave per block, parents first, the parent's step as tagged
 *           words; the tail takes gradient_trans_times_direction (:808-820), the direction test and the start of the line
 *           search (:944-954): k_forward_all_w and k_ls_begin of the older protocol.
 *
 * A solve of BASELINE config C4 (one Newton iteration) is k_sg, k_hf_w, k_fwd3, k_sg: 4 launches instead of 12.
 */
#pragma once

struct W3 {
    u64 *xu;             /* [sum_nx + sum_nu][2]: x, u of parent nodes inside k_sg */
    u64 *red;            /* [workgroups][2][2]: per-workgroup partials {fval, err} (k_sg) / {dot, -} (k_fwd3) */
    int *cnt;            /* workgroups that have posted */
    unsigned tag;
    int sum_nx;
    int lds_wave;        /* doubles of LDS per wave of k_sg (a node's stage window) */
    HostRes *hm;         /* pinned host memory: the control block as the launch leaves it, then the launch's tag (w3_mirror); nullptr: none */
};
/* The verdict of a launch of k_sg / k_sgp goes straight to the host, as on the persistent path: the control block by system-scope
 * stores to pinned memory, the stores' acknowledgements, then the launch's tag.  The host enqueues a chunk of launches that ends with
 * one of these and polls for its tag instead of a device-to-host copy and a stream synchronisation per chunk (and, without the HIP
 * event pair, per solve).  Called by ONE thread: the one that wrote the control block (the tail), or thread 0 of workgroup 0 of a
 * launch that found its phase not due (every earlier launch is over by then). */
__device__ __forceinline__ void w3_mirror(const W3 &Wd, Ctrl *c, bool first, unsigned long long t_start) {
    if (first) *reinterpret_cast<unsigned long long *>(&c->pad0) = t_start;      /* the solve's first launch: its start, kept in the control block (pad0, pad1) */
    if (!Wd.hm) return;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(&Wd.hm->c);
    for (int i = 0; i < (int)(sizeof(Ctrl) / 8); i++) __hip_atomic_store(dst + i, src[i], RLX, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&Wd.hm->t_end, (unsigned long long)wall_clock64(), RLX, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&Wd.hm->seq, Wd.tag, RLX, __HIP_MEMORY_SCOPE_SYSTEM);
}


#define SG_WAVES 4

/* as grad_body, the parent's x | u from tagged words (staged in the wave's LDS window) */
__device__ void grad_body_t(const Tree &T, const Data &D, int termCondition, int k, int lane, double *win, const u64 *xu, int sum_nx, unsigned tag) {
    const int p = T.dad[k], nxk = T.nx[k], nxp = T.nx[p], nup = T.nu[p];
    const int xo = T.xoff[k], xp = T.xoff[p], up = T.uoff[p];
    const double *A = D.A + T.aoff[k], *B = D.B + T.boff[k];
    bool dead = false;
    for (int j = lane; j < nxp + nup; j += WAVE) {
        const u64 *src = j < nxp ? xu + 2 * (size_t)(xp + j) : xu + 2 * (size_t)(sum_nx + up + j - nxp);
        win[j] = wait_tag(src, tag, dead);
    }
    if (__builtin_amdgcn_ballot_w64(dead) != 0ull && lane == 0) { D.ctrl->status = 3; __hip_atomic_store(&D.ctrl->done, 1, RLX, AGENT); }
    WSYNC();
    double part = 0.0;
    for (int i = lane; i < nxk; i += WAVE) {
        double rv = fma(-1.0, D.x[xo + i], D.b[xo + i]);
        double acc = 0.0;
        acc = dot_batched(A + i, nxk, win, 1, nxp, acc, true);
        rv += acc;
        acc = 0.0;
        acc = dot_batched(B + i, nxk, win + nxp, 1, nup, acc, true);
        rv += acc;
        D.res[xo + i] = rv;
        D.resMod[xo + i] = rv;
        part = (termCondition == 2) ? nanmax(part, fabs(rv)) : fma(rv, rv, part);
    }
    part = (termCondition == 2) ? wave_max(part) : wave_sum(part);
    if (lane == 0) D.part_err[k] = part;
}

/* sum (and, with IS_MAX2, maximum) over the workgroups' tagged partials, by ONE wave: lane l takes workgroups l, l + 64, ..
 * in ascending order, then the fixed-order wave reduction.  red[(2 b + which) * 2]: partial `which` of workgroup b. */
template <bool IS_MAX>
__device__ double w3_reduce(const u64 *red, int which, int n, unsigned tag, int lane) {
    double acc = 0.0;
    for (int b0 = 0; b0 < n; b0 += 8 * WAVE) {
        double v[8];
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int b = b0 + m * WAVE + lane;
                bool okk = true;
                v[m] = ld_tag(red + ((size_t)2 * (b < n ? b : 0) + which) * 2, tag, okk);
                ok = ok && (okk || b >= n);
            }
            if (__all(ok)) break;
            if (wall_clock64() - t0 > 20000000ull) return __builtin_nan("");      /* 0.2 s: cannot happen (every workgroup posted before it counted itself off); a NaN ends the solve */
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) { const int b = b0 + m * WAVE + lane; if (b < n) acc = IS_MAX ? nanmax(acc, v[m]) : acc + v[m]; }
    }
    return IS_MAX ? wave_max(acc) : wave_sum(acc);
}

/* both partials of every workgroup in ONE round of polls: sum of partial 0, sum or maximum of partial 1 */
__device__ void w3_reduce2(const u64 *red, int n, unsigned tag, int lane, bool max1, double &r0, double &r1) {
    double a0 = 0.0, a1 = 0.0;
    for (int b0 = 0; b0 < n; b0 += 8 * WAVE) {
        double v0[8], v1[8];
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int b = b0 + m * WAVE + lane;
                bool okk = true;
                const u64 *src = red + (size_t)4 * (b < n ? b : 0);
                v0[m] = ld_tag(src, tag, okk);
                v1[m] = ld_tag(src + 2, tag, okk);
                ok = ok && (okk || b >= n);
            }
            if (__all(ok)) break;
            if (wall_clock64() - t0 > 20000000ull) { r0 = __builtin_nan(""); r1 = r0; return; }      /* 0.2 s: cannot happen; a NaN ends the solve */
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int b = b0 + m * WAVE + lane;
            if (b < n) { a0 += v0[m]; a1 = max1 ? nanmax(a1, v1[m]) : a1 + v1[m]; }
        }
    }
    r0 = wave_sum(a0);
    r1 = max1 ? wave_max(a1) : wave_sum(a1);
}

/* ------------------------------------------------------------------------------------------ */
/* k_sg: stage sweep + gradient + dual value / Armijo test + termination test                  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(SG_WAVES * WAVE) k_sg(Tree T, Data D, Opts O, W3 Wd, int mode, int h, int t)
#if !TQ_HAS(TQP_W3)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double part[2][SG_WAVES];
    Ctrl *c = D.ctrl;
    const unsigned long long t_begin = wall_clock64();
    if (mode == 1 && !phase_trial(c, h, t)) { if (blockIdx.x == 0 && threadIdx.x == 0) w3_mirror(Wd, c, false, 0ull); return; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * SG_WAVES + wave;
    double *win = lds + (size_t)wave * Wd.lds_wave;
    double fv = 0.0, er = 0.0;
    if (k < T.Nn) {
        stage_body(T, D, mode, k, lane, win, true, Wd.xu, Wd.sum_nx, Wd.tag);
        if (lane == 0) fv = D.fval[k];                     /* written by this lane */
        if (k > 0) {
            WSYNC();
            grad_body_t(T, D, O.termCondition, k, lane, win, Wd.xu, Wd.sum_nx, Wd.tag);
            if (lane == 0) er = D.part_err[k];
        }
    }
    if (lane == 0) { part[0][wave] = fv; part[1][wave] = er; }
    __syncthreads();
    if (wave != 0) return;
    const bool mx = O.termCondition == 2;
    if (lane == 0) {
        double f = 0.0, e = 0.0;
#pragma unroll
        for (int w = 0; w < SG_WAVES; w++) { f += part[0][w]; e = mx ? nanmax(e, part[1][w]) : e + part[1][w]; }
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 0) * 2, f, Wd.tag);
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 1) * 2, e, Wd.tag);
    }
    Fuse F; F.red = nullptr; F.cnt = Wd.cnt; F.tag = Wd.tag; F.on = 1;
    if (!fuse_last(F, (int)gridDim.x, lane)) return;
    double f, err;
    w3_reduce2(Wd.red, (int)gridDim.x, Wd.tag, lane, mx, f, err);
    if (lane == 0) {
        bool test = true;
        if (mode == 0) { c->fval0 = f; c->fval = f; }
        else { ls_decide_tail(c, D, O, f); test = !c->done && !c->ls_pending; }
        if (test) {
            /* top of the next Newton iteration: the gradient at the (accepted) point is in res already */
            if (O.termCondition == 1) err = sqrt(err);
            c->err = err;
            if (err < O.tol) { c->done = 1; c->status = 0; }      /* TREEQP_OPTIMAL_SOLUTION_FOUND */
        }
        w3_mirror(Wd, c, mode == 0, t_begin);
    }
}
#endif

/* the control block to the host, on its own (read_ctrl: the last launch enqueued is not one that posts) */
__global__ void k_w3_post(Data D, W3 Wd)
#if !TQ_HAS(TQP_W3)
;
#else
{
    if (threadIdx.x == 0) w3_mirror(Wd, D.ctrl, false, 0ull);
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_sgp: k_sg with one workgroup per PARENT.  The children's [A | B] (the matrix C of the dual Hessian block) is fetched once,     */
/* coalesced, into LDS and serves the parent's stage QP (C' lambda) and the children's gradients (C [x; u]); everything else a node */
/* needs from global memory (duals, step, weights, bounds, linear terms) is requested up front, ONE round trip.  k_sg's waves each   */
/* walk the index tables and their node's columns of A and B in global memory: ~14 dependent batches of loads per node at nx = 20,  */
/* 9 us of a node's 10.  Work of workgroup p: stage QP of node p (wave 0) and of its LEAF children (waves 1 ..; a child that is a    */
/* parent is staged by its own workgroup), gradient of every child (a wave each).  A child that is a parent hands its x over as     */
/* tagged words; workgroups are numbered children first, so what is waited for is running or done.  Per node the arithmetic is      */
/* stage_body's / grad_body's, operation for operation (bit-identical x, u, res, dual terms); the sums over nodes are per workgroup. */
/* Same tails as k_sg.  Needs nx + nu <= 64 and d <= 64 per node (true for the wide-block class).                                    */
/* ------------------------------------------------------------------------------------------ */
struct SgNode { double lam, dl, qv, qi, lo, hi, wd, old; };      /* one entry of a node's [x | u]: what stage_body reads for it */

__device__ __forceinline__ void sgp_load(const Data &D, const double *lamc, int mode, bool save_s, bool own, int nxk, int nuk, int xo, int uo, int lane, SgNode &n) {
    const bool isx = lane < nxk, in = lane < nxk + nuk;
    const int j = !in ? 0 : (isx ? lane : lane - nxk);
    const int ix = xo + (isx ? j : 0), iu = uo + (isx ? 0 : j);
    /* (clamped addresses, masked use: every load of the node goes out together) */
    n.lam = lamc[ix]; n.dl = D.dlam[ix];
    n.qv = isx ? D.q[ix] : D.r[iu];
    n.qi = isx ? D.Qinv[ix] : D.Rinv[iu];
    n.lo = isx ? D.xmin[ix] : D.umin[iu];
    n.hi = isx ? D.xmax[ix] : D.umax[iu];
    n.wd = isx ? D.Qd[ix] : D.Rd[iu];
    n.old = isx ? D.xUnc[ix] : D.uUnc[iu];
    (void)mode; (void)save_s; (void)own;
}
/* the clipping stage QP of one node, one entry per lane (stage_body, clipping branch, lines in the same order); v0 = -q + lambda_own
 * resp. -r, minus the children's terms, is passed in; returns the node's dual term (valid in every lane) */
__device__ __forceinline__ double sgp_finish(const Data &D, const SgNode &n, double v, double p_c, bool save_s, int nxk, int nuk, int xo, int uo, int lane, double &xout) {
    const bool isx = lane < nxk, in = lane < nxk + nuk;
    const int j = isx ? lane : lane - nxk;
    double p_qx = 0.0, p_hx = 0.0, p_ru = 0.0, p_hu = 0.0;
    xout = 0.0;
    if (in) {
        const double unc = n.qi * v;
        double xv, cal;
        if (unc >= n.hi) { xv = n.hi; cal = 0.0; } else if (unc <= n.lo) { xv = n.lo; cal = 0.0; } else { xv = unc; cal = n.qi; }
        xout = xv;
        if (isx) {
            D.qmod[xo + j] = v;
            if (save_s) D.xUncS[xo + j] = n.old;
            D.xUnc[xo + j] = unc; D.x[xo + j] = xv; D.QinvCal[xo + j] = cal;
            p_qx = fma(n.wd * xv, xv, p_qx);
            p_hx = fma(v, xv, p_hx);
        } else {
            D.rmod[uo + j] = v;
            if (save_s) D.uUncS[uo + j] = n.old;
            D.uUnc[uo + j] = unc; D.u[uo + j] = xv; D.RinvCal[uo + j] = cal;
            p_ru = fma(n.wd * xv, xv, p_ru);
            p_hu = fma(v, xv, p_hu);
        }
    }
    p_qx = wave_sum(p_qx); p_hx = wave_sum(p_hx); p_ru = wave_sum(p_ru); p_hu = wave_sum(p_hu); p_c = wave_sum(p_c);
    double f = -0.5 * p_qx - p_c;       /* clipping.c:375 */
    f += p_hx;                          /* :376 */
    f -= 0.5 * p_ru;                    /* :380 */
    f += p_hu;                          /* :381 */
    return f;
}


/* The step of block `ii` without waiting for any other block (see k_fwd3c below): the slices of the ancestors' steps along the path
 * from the root, recomputed by this wave.  A: the block's row of the path table.  Returns the block's step (lanes < d); `val` holds the
 * slices (lane group g: path entry g, second round 8 + g), `gl` the first lane of the LAST slice = the step of the block's own node. */
#define FWDC_INTS 65
__device__ __forceinline__ double fwdc_block(const Data &D, const int *A, int d, int nxi, int woff, int utoff, int lane, double &val, int &gl) {
    const int L = A[0];
    const int grp = lane >> 3, j8 = lane & 7;
    int ent[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) ent[r][q] = A[1 + 4 * (8 * r + grp) + q];          /* (entries beyond L are zero) */
    const int lc = lane < d ? lane : 0;
    const double *Mg = D.CholUt + utoff;
    /* everything requested together: the path's rows of M and z0 (two rounds of eight entries), the block's own */
    double m[2][8], z[2], mo[8];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const bool on = 8 * r + grp < L;
        const int nxa = ent[r][3] & 255, nxk = ent[r][3] >> 8, da = ent[r][2];
        const bool act = on && j8 < nxk;
        z[r] = D.CholW[act ? ent[r][0] + j8 : 0];
#pragma unroll
        for (int i = 0; i < 8; i++) m[r][i] = D.CholUt[(act && i < nxa) ? ent[r][1] + i * da + j8 : 0];
        if (!act) z[r] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) if (!(act && i < nxa)) m[r][i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) mo[i] = Mg[(size_t)(i < nxi ? i : 0) * d + lc];
    const double z0 = D.CholW[woff + d * d - d + lc];
    LOADS_DONE();
    val = 0.0;
    for (int k = 0; k < L; k++) {
        const int r = k >> 3, g = k & 7, gp = 8 * ((k - 1) & 7);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            a0 = fma(r ? m[1][i] : m[0][i], rdlane(val, gp + i), a0);
            a1 = fma(r ? m[1][i + 1] : m[0][i + 1], rdlane(val, gp + i + 1), a1);
        }
        const double nv = (r ? z[1] : z[0]) - (a0 + a1);
        val = grp == g ? nv : val;
    }
    gl = 8 * ((L - 1) & 7);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        a0 = fma(i < nxi ? mo[i] : 0.0, rdlane(val, gl + i), a0);
        a1 = fma(i + 1 < nxi ? mo[i + 1] : 0.0, rdlane(val, gl + i + 1), a1);
    }
    return z0 - (a0 + a1);
}
/* res' dlam over the blocks as k_fwd3c takes it (groups of SG_WAVES consecutive blocks summed in order, the groups' sums dealt over the
 * lanes as in w3_reduce): the same number to the last bit whichever kernel ran the forward sweep.  pdw: tagged [block] */
__device__ double fwdc_reduce_dot(const u64 *pdw, int Np, unsigned tag, int lane) {
    const int n = (Np - 1 + SG_WAVES - 1) / SG_WAVES;
    double acc = 0.0;
    for (int b0 = 0; b0 < n; b0 += 8 * WAVE) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int b = b0 + m * WAVE + lane;
            if (b < n) {
                double sb = 0.0;
                for (int w = 0; w < SG_WAVES; w++) {
                    const int ii = 1 + b * SG_WAVES + w;
                    double v = 0.0;
                    if (ii < Np) {
                        const unsigned long long t0 = wall_clock64();
                        for (;;) {
                            bool ok = true;
                            v = ld_tag(pdw + (size_t)ii * 2, tag, ok);
                            if (ok) break;
                            if (wall_clock64() - t0 > 20000000ull) { v = __builtin_nan(""); break; }      /* cannot happen: posted before the workgroup counted itself off */
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    sb += v;
                }
                acc = acc + sb;
            }
        }
    }
    return wave_sum(acc);
}

#ifdef TQ_WIDE_STAMPS
#define SGSTAMP(k_) do { if (tid == 0) reinterpret_cast<unsigned long long *>(D.W + e[8])[k_] = wall_clock64(); } while (0)
#else
#define SGSTAMP(k_) do { } while (0)
#endif
/* mode 0 with lam_src != nullptr: the first sweep of a solve reads the starting duals from lam_src, copies them into the current buffer
 * (every node's own slice is written by the workgroup that stages the node) and its tail writes the WHOLE control block: no copy and
 * no memset in front of the launch */
/* mode 2 (trees with a path table, see k_fwd3c): the FORWARD SWEEP and the first trial of the line search in one launch.  Wave 0 of
 * workgroup p first takes the step of block p from its ancestors' data (fwdc_block: no other workgroup is waited for), writes it to
 * dlam and posts the block's part of res' dlam; the trial point is lambda + dlam (tau = 1).  The launch's tail then does what k_fwd3c's
 * tail and this kernel's mode-1 tail do one after the other: res' dlam (summed in k_fwd3c's order: the same number), the direction test
 * (:944-954 -- as on the persistent paths the first trial has been evaluated speculatively by then; a direction that is not one of
 * descent ends the solve before anything of it is used), the start of the line search, the Armijo test, the next termination test. */
struct SgpFwdNo { };
struct SgpFwdYes { const int *anc; u64 *pdw; };
template <bool FWD> struct SgpFwd { using type = SgpFwdNo; };
template <> struct SgpFwd<true> { using type = SgpFwdYes; };
__device__ __forceinline__ const int *sgp_anc(const SgpFwdNo &) { return nullptr; }
__device__ __forceinline__ const int *sgp_anc(const SgpFwdYes &a) { return a.anc; }
__device__ __forceinline__ u64 *sgp_pdw(const SgpFwdNo &) { return nullptr; }
__device__ __forceinline__ u64 *sgp_pdw(const SgpFwdYes &a) { return a.pdw; }
template <bool FWD>      /* FWD: the instantiation that knows mode 2 (modes 0 and 1 keep the registers, the arguments and the code they had: 91 against 101 registers is 10 us per C4 solve) */
__global__ void __launch_bounds__(WT) k_sgp_t(Tree T, Data D, Opts O, W3 Wd, int mode, int h, int t, int accs_cap, const double *lam_src, typename SgpFwd<FWD>::type fa)
#if !TQ_HAS(TQP_W3)
;
#else
{
    const int *anc = sgp_anc(fa);
    u64 *pdw = sgp_pdw(fa);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double part[2][WW];
    Ctrl *c = D.ctrl;
    const int p = T.Np - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int e[28];
#pragma unroll
    for (int i = 0; i < 28; i++) e[i] = T.desc[(size_t)DESC_INTS * p + i];
    const bool fresh = mode == 0 && lam_src != nullptr;
    const bool fwd = FWD && mode == 2, trial = mode >= 1;
    const int cur = fresh ? 0 : c->cur, ls_iter = fwd ? 1 : c->ls_iter;
    const double step = fwd ? 1.0 : c->tau - c->tauPrev;
    const unsigned long long t_begin = wall_clock64();
    if ((mode == 1 && !phase_trial(c, h, t)) || (fwd && !phase_main(c, h))) { if (blockIdx.x == 0 && tid == 0) w3_mirror(Wd, c, false, 0ull); return; }
    const int d = e[0], nxp = e[1], nup = e[2], nkp = e[3], k0 = e[4], nz = nxp + nup, xop = e[5], uop = e[6], ko = e[7];
    SGSTAMP(0);
    const bool save_s = trial && ls_iter == 1;          /* first trial of a line search: xUnc / uUnc still hold phase S of this iteration */
    const double *lamc = fresh ? lam_src : (cur ? D.lam1 : D.lam0);
    double *lamn = cur ? D.lam0 : D.lam1;
    const int ldc = d | 1;
    lds_ptr Cs = to_lds(lds);                           /* d x nz, leading dimension ldc (odd: lanes over rows and lanes over columns both spread over the banks) */
    lds_ptr lkl = Cs + ldc * nz;                        /* duals of the children = the dual block of node p (at the trial point in mode 1) */
    lds_ptr bl = lkl + 64;                              /* b of the children */
    lds_ptr xpl = bl + 64;                              /* [x_p | u_p] */
    lds_ptr xkl = xpl + 64;                             /* x of the leaf children */
    lds_ptr dkl = xkl + 64;                             /* mode 2: the step of block p */
    lds_ptr accs = dkl + 64;                            /* [child][entry of node p]: that child's term of C' lambda */
    const bool kids_are_leaves = k0 >= T.Np;

    /* ---- everything a node needs from global memory, requested together ---- */
    SgNode nd;
    double lamk = 0.0, dlk = 0.0, bk = 0.0;
    int nxl = 0, nul = 0, xol = 0, uol = 0, ccl = -1, rowl = 0;        /* waves 1 ..: the leaf child this wave stages (first round) */
    if (wave == 0) {
        sgp_load(D, lamc, mode, save_s, p > 0, nxp, nup, xop, uop, lane, nd);
        const int tt = lane < d ? lane : 0;
        lamk = lamc[ko + tt]; dlk = D.dlam[ko + tt]; bk = D.b[ko + tt];
        if (fwd) {
            /* the step of block p (the root block's is what k_hf_w left in dlam) and of node p's own duals (the last slice of the path) */
            const double rv = D.res[ko + tt];
            if (p > 0) {
                double val;
                int gl;
                dlk = fwdc_block(D, anc + (size_t)FWDC_INTS * p, d, nxp, e[8], e[9], lane, val, gl);
                double own = 0.0;
#pragma unroll
                for (int i = 0; i < 8; i++) { const double tv = rdlane(val, gl + i); own = lane == i ? tv : own; }
                nd.dl = own;
                if (lane < d) D.dlam[ko + lane] = dlk;
                double pd = lane < d ? rv * dlk : 0.0;
                pd = wave_sum(pd);
                if (lane == 0) { D.part_dot[p] = pd; st_tag(pdw + (size_t)p * 2, pd, Wd.tag); }
            }
            if (lane < d) dkl[lane] = dlk;
        }
    } else if (kids_are_leaves && wave - 1 < nkp) {
        ccl = wave - 1;
        const int kid = k0 + ccl;
        nxl = ccl == 0 ? e[16] : ccl == 1 ? e[19] : e[22];
        rowl = ccl == 0 ? 0 : ccl == 1 ? e[16] : e[16] + e[19];
        nul = T.nu[kid]; xol = ko + rowl; uol = T.uoff[kid];
        sgp_load(D, lamc, mode, save_s, true, nxl, nul, xol, uol, lane, nd);
    }
    /* ---- C: rows on the lanes, columns dealt over the waves, eight loads in flight per thread and child ---- */
    for (int cc = 0, rowoff = 0; cc < nkp; cc++) {
        const int kid = k0 + cc;
        const bool rec = cc < 4;
        const int rnx = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : e[25];
        const int rao = cc == 0 ? e[17] : cc == 1 ? e[20] : cc == 2 ? e[23] : e[26];
        const int rbo = cc == 0 ? e[18] : cc == 1 ? e[21] : cc == 2 ? e[24] : e[27];
        const int nxc = rec ? rnx : T.nx[kid];
        const double *A = D.A + (rec ? rao : T.aoff[kid]), *B = D.B + (rec ? rbo : T.boff[kid]);
        const bool rowok = lane < nxc;
        const int i = rowok ? lane : 0;
        for (int c0 = 0; c0 < nz; c0 += 8 * WW) {
            double a[8];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int col = c0 + wave + WW * m;
                const int cs = (rowok && col < nz) ? col : 0;
                const bool st = cs < nxp;
                a[m] = (st ? A : B)[i + (st ? cs : cs - nxp) * nxc];
            }
            LOADS_DONE();
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int col = c0 + wave + WW * m;
                if (rowok && col < nz) Cs[rowoff + i + col * ldc] = a[m];
            }
        }
        rowoff += nxc;
    }
    if (wave == 0 && lane < d) { lkl[lane] = trial ? fma(step, dlk, lamk) : lamk; bl[lane] = bk; }
    __syncthreads();
    SGSTAMP(1);
    /* ---- the children's terms of C' lambda: thread (child, entry), the products of a child in ascending row order ---- */
    if (nkp * nz <= accs_cap) {
        int rowoff = 0;
        for (int cc = 0; cc < nkp; cc++) {
            const int nxc = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : cc == 3 ? e[25] : T.nx[k0 + cc];
            if ((cc & (WW - 1)) == wave && lane < nz) {
                double acc = 0.0;
                for (int i = 0; i < nxc; i++) acc = fma(Cs[rowoff + i + lane * ldc], lkl[rowoff + i], acc);
                accs[cc * nz + lane] = acc;
            }
            rowoff += nxc;
        }
    }
    __syncthreads();
    double fv = 0.0;
    if (wave == 0) {
        /* node p */
        const bool isx = lane < nxp, in = lane < nz;
        double lown = 0.0;
        if (p > 0 && isx) { lown = trial ? fma(step, nd.dl, nd.lam) : nd.lam; if (trial) lamn[xop + lane] = lown; else if (fresh) D.lam0[xop + lane] = lown; }
        double v = isx ? fma(-1.0, nd.qv, lown) : -1.0 * nd.qv;
        if (in) {
            if (nkp * nz <= accs_cap) { for (int cc = 0; cc < nkp; cc++) v = fma(-1.0, accs[cc * nz + lane], v); }
            else {
                int rowoff = 0;
                for (int cc = 0; cc < nkp; cc++) {
                    const int nxc = T.nx[k0 + cc];
                    double acc = 0.0;
                    for (int i = 0; i < nxc; i++) acc = fma(Cs[rowoff + i + lane * ldc], lkl[rowoff + i], acc);
                    v = fma(-1.0, acc, v);
                    rowoff += nxc;
                }
            }
        }
        const double p_c = lane < d ? fma(bl[lane], lkl[lane], 0.0) : 0.0;      /* cmod = sum_kids b_kid' lambda_kid */
        double xv;
        const double f = sgp_finish(D, nd, v, p_c, save_s, nxp, nup, xop, uop, lane, xv);
        if (in) {
            xpl[lane] = xv;
            if (p > 0 && isx) st_tag(Wd.xu + 2 * (size_t)(xop + lane), xv, Wd.tag);          /* the parent's workgroup waits for x of node p */
        }
        if (lane == 0) { D.fval[p] = f; fv = f; }
    } else if (kids_are_leaves) {
        /* leaf children: no children's terms */
        for (int cc = wave - 1; cc < nkp; cc += WW - 1) {
            if (cc != ccl) {
                /* (more than three children: later rounds fetch theirs now) */
                const int kid = k0 + cc;
                nxl = T.nx[kid]; nul = T.nu[kid]; xol = T.xoff[kid]; uol = T.uoff[kid]; rowl = xol - ko;
                sgp_load(D, lamc, mode, save_s, true, nxl, nul, xol, uol, lane, nd);
            }
            const bool isx = lane < nxl;
            double lown = 0.0;
            if (fwd) nd.dl = dkl[rowl + (isx ? lane : 0)];                  /* (written by wave 0 before the barrier above) */
            if (isx) { lown = trial ? fma(step, nd.dl, nd.lam) : nd.lam; if (trial) lamn[xol + lane] = lown; else if (fresh) D.lam0[xol + lane] = lown; }
            const double v = isx ? fma(-1.0, nd.qv, lown) : -1.0 * nd.qv;
            double xv;
            const double f = sgp_finish(D, nd, v, 0.0, save_s, nxl, nul, xol, uol, lane, xv);
            if (isx) xkl[rowl + lane] = xv;
            if (lane == 0) { D.fval[k0 + cc] = f; fv += f; }
        }
    }
    SGSTAMP(2);
    __syncthreads();
    /* ---- gradient of the children: a wave per child, rows on the lanes (grad_body) ---- */
    const bool mx = O.termCondition == 2;
    double er = 0.0;
    {
        int rowoff = 0;
        for (int cc = 0; cc < nkp; cc++) {
            const int kid = k0 + cc;
            const int nxc = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : cc == 3 ? e[25] : T.nx[kid];
            if ((cc & (WW - 1)) == wave) {
                const int xo = ko + rowoff;
                double part_k = 0.0;
                if (lane < nxc) {                       /* (nx <= 64) */
                    const int i = lane;
                    double xk;
                    if (kid < T.Np) { bool dead = false; xk = wait_tag(Wd.xu + 2 * (size_t)(xo + i), Wd.tag, dead); if (dead) { c->status = 3; __hip_atomic_store(&c->done, 1, RLX, AGENT); __hip_atomic_store(Wd.cnt + 1, (int)Wd.tag, RLX, AGENT); } }      /* (cnt[1]: a bounded wait of THIS launch gave up -- the tail of a fresh first sweep zeroes the control block) */
                    else xk = xkl[rowoff + i];
                    double rv = fma(-1.0, xk, bl[rowoff + i]);
                    double acc = 0.0;
                    for (int j = 0; j < nxp; j++) acc = fma(Cs[rowoff + i + j * ldc], xpl[j], acc);
                    rv += acc;
                    acc = 0.0;
                    for (int j = 0; j < nup; j++) acc = fma(Cs[rowoff + i + (nxp + j) * ldc], xpl[nxp + j], acc);
                    rv += acc;
                    D.res[xo + i] = rv;
                    D.resMod[xo + i] = rv;
                    part_k = mx ? nanmax(part_k, fabs(rv)) : fma(rv, rv, part_k);
                }
                part_k = mx ? wave_max(part_k) : wave_sum(part_k);
                if (lane == 0) D.part_err[kid] = part_k;
                er = mx ? nanmax(er, part_k) : er + part_k;
            }
            rowoff += nxc;
        }
    }
    if (lane == 0) { part[0][wave] = fv; part[1][wave] = er; }
    __syncthreads();
    SGSTAMP(3);
    if (wave != 0) return;
    if (lane == 0) {
        double f = 0.0, ee = 0.0;
#pragma unroll
        for (int w = 0; w < WW; w++) { f += part[0][w]; ee = mx ? nanmax(ee, part[1][w]) : ee + part[1][w]; }
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 0) * 2, f, Wd.tag);
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 1) * 2, ee, Wd.tag);
    }
    Fuse F; F.red = nullptr; F.cnt = Wd.cnt; F.tag = Wd.tag; F.on = 1;
    if (!fuse_last(F, (int)gridDim.x, lane)) return;
    double f, err;
    w3_reduce2(Wd.red, (int)gridDim.x, Wd.tag, lane, mx, f, err);
    double dots = 0.0;
    if (fwd) dots = fwdc_reduce_dot(pdw, T.Np, Wd.tag, lane) + D.part_dot[0];      /* + the root's, from k_hf_w */
    if (lane == 0) {
        bool test = true;
        if (mode == 0) {
            if (fresh) {          /* a new solve: the control block starts from zero (k_hf_w counts regularised blocks into it) */
                const bool gave_up = __hip_atomic_load(Wd.cnt + 1, RLX, AGENT) == (int)Wd.tag;          /* ... unless a wait of this very launch timed out */
                Ctrl z{}; *c = z;
                if (gave_up) { c->status = 3; c->done = 1; }
            }
            c->fval0 = f; c->fval = f;
        }
        else if (fwd) {
            const double dotp = -dots;                                  /* :819 */
            c->dot = dotp;
            if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) { c->done = 1; c->status = 2; test = false; }      /* :951, NaN included */
            else {
                c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
                ls_decide_tail(c, D, O, f);
                test = !c->done && !c->ls_pending;
            }
        }
        else { ls_decide_tail(c, D, O, f); test = !c->done && !c->ls_pending; }
        if (test) {
            if (O.termCondition == 1) err = sqrt(err);
            c->err = err;
            if (err < O.tol) { c->done = 1; c->status = 0; }      /* TREEQP_OPTIMAL_SOLUTION_FOUND */
        }
        w3_mirror(Wd, c, mode == 0, t_begin);
    }
}
#endif
/* LDS of k_sgp: C, the four 64-entry vectors, the children's terms */
static inline size_t wide3_lds_sgp(int d, int nz, int accs) { return ((size_t)(d | 1) * nz + 5 * 64 + accs) * sizeof(double); }

/* ------------------------------------------------------------------------------------------ */
/* k_hf_w: H + backward sweep + preparation of the forward sweep, one workgroup per block       */
/* ------------------------------------------------------------------------------------------ */
/* left-looking tall Cholesky of the first NC columns of a 16-column panel, one row per lane (p_potrf_rows with a
 * column count below the array size: the last panel of a block whose dimension is not a multiple of 16) */
template <int NC>
__device__ __forceinline__ double w3_potrf_cols(double (&Tr)[16], int lane) {
    double pmin = __builtin_inf();
#pragma unroll
    for (int j = 0; j < NC; j++) {
        double s = Tr[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-Tr[k], rdlane(Tr[k], j), s);
        const double pj = rdlane(s, j);
        pmin = fmin(pmin, pj);
        Tr[j] = s * pivot_rsqrt3(pj);
    }
    return pmin;
}
__device__ __forceinline__ double w3_chain(double (&Tr)[16], int lane, int wp) {
    switch (wp) {
        case 4: return w3_potrf_cols<4>(Tr, lane);
        case 8: return w3_potrf_cols<8>(Tr, lane);
        case 12: return w3_potrf_cols<12>(Tr, lane);
        default: return w3_potrf_cols<16>(Tr, lane);
    }
}

/* tile (I, J) of the LDS image -= rows 16 I.. x rows 16 J..' over columns k0 .. k0 + 15 (one 16 x 16 x 16 product: four MFMAs);
 * the lanes of a C access run down a column of the tile */
__device__ __forceinline__ void w3_tile_update(lds_ptr Tm, int ld, int I, int J, int k0, int r16, int g) {
    f64x4 acc;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = Tm[16 * I + r16 + (16 * J + g + 4 * q) * ld];
#pragma unroll
    for (int s = 0; s < 16; s += 4) {
        const double a = -1.0 * Tm[16 * J + r16 + (k0 + s + g) * ld];
        const double b = Tm[16 * I + r16 + (k0 + s + g) * ld];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) Tm[16 * I + r16 + (16 * J + g + 4 * q) * ld] = acc[q];
}
/* three tiles at once: every operand is requested before the first product, the three accumulators take turns on the matrix pipe
 * (one tile at a time is a chain of LDS round trips and dependent MFMAs: ~1000 cycles for 256 cycles of matrix work) */
__device__ __forceinline__ void w3_tile_update3(lds_ptr Tm, int ld, int k0, int r16, int g, int I0, int J0, bool v0, int I1, int J1, bool v1, int I2, int J2, bool v2) {
    const int I[3] = {I0, v1 ? I1 : I0, v2 ? I2 : I0}, J[3] = {J0, v1 ? J1 : J0, v2 ? J2 : J0};
    f64x4 acc[3];
    double a[3][4], b[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[t][q] = Tm[16 * I[t] + r16 + (16 * J[t] + g + 4 * q) * ld];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            a[t][s] = -1.0 * Tm[16 * J[t] + r16 + (k0 + 4 * s + g) * ld];
            b[t][s] = Tm[16 * I[t] + r16 + (k0 + 4 * s + g) * ld];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int t = 0; t < 3; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s], b[t][s], acc[t], 0, 0, 0);
    const bool v[3] = {v0, v1, v2};
#pragma unroll
    for (int t = 0; t < 3; t++)
        if (v[t]) {
#pragma unroll
            for (int q = 0; q < 4; q++) Tm[16 * I[t] + r16 + (16 * J[t] + g + 4 * q) * ld] = acc[t][q];
        }
}
/* n-th tile (n = 0, 1, ..) of the trailing part { (I, J) : J0 <= J < ct, J <= I < rt }, column by column; false: there is none */
__device__ __forceinline__ bool w3_trailing_tile(int n, int J0, int ct, int rt, int &I, int &J) {
    I = J0; J = J0;
    for (int jj = J0; jj < ct; jj++) {
        const int cnt = rt - jj;
        if (n < cnt) { I = jj + n; J = jj; return true; }
        n -= cnt;
    }
    return false;
}
/* n-th lower tile of a ct x ct grid, column by column */
__device__ __forceinline__ bool w3_lower_tile(int n, int ct, int &I, int &J) { return w3_trailing_tile(n, 0, ct, ct, I, J); }

/* where the inverse of diagonal tile p is kept.  Four column tiles (d > 48: the LDS budget of three workgroups per CU has no
 * room to spare): tiles of the strictly upper part of the image, which the factorisation never touches -- (0,1), (1,2), (2,3),
 * (0,2) -- leading dimension ld.  Fewer column tiles: an area of its own behind the image, leading dimension 16. */
__device__ __forceinline__ int w3_uslot(int p, int ct, int ld, int dp, int &uls) {
    if (ct == 4) { uls = ld; const int I = p < 3 ? p : 0, J = p < 3 ? p + 1 : 2; return 16 * I + 16 * J * ld; }
    uls = 16;
    return ld * dp + 256 * p;
}
/* Schur tile `slot` of wave 2 / 3: the diagonal tiles (s, s) on wave 2, the tiles below the diagonal -- (1,0), (2,0), (2,1) -- on wave 3 */
__device__ __forceinline__ bool w3_schur_tile(int wave, int slot, int nt2, int &I2, int &J2) {
    if (wave == 2) { I2 = slot; J2 = slot; return slot < nt2; }
    I2 = slot == 0 ? 1 : 2; J2 = slot == 2 ? 1 : 0;
    return I2 < nt2;
}

#ifndef TQ_W3_WPS
#define TQ_W3_WPS 2        /* workgroups per CU the register budget is set for (three fit by LDS: 168 registers spill, and measured slower) */
#endif
#ifndef TQ_STAMP_BLOCK
#define TQ_STAMP_BLOCK 0
#endif
#ifdef TQ_WIDE_STAMPS      /* diagnostic builds: time stamps of the root block (slots 0..) and of the last block (slots 32..), thread 0 */
/* per-block stamps (start, records in, record posted, end) into CholW, which this kernel family does not use */
#define W3BSTAMP(k_) do { if (wave == 0 && lane == 0) reinterpret_cast<unsigned long long *>(D.CholW + e[8])[k_] = wall_clock64(); } while (0)
#define W3STAMP(slot) do { if (wave == 0 && lane == 0 && (ii == TQ_STAMP_BLOCK || ii == T.Np - 1)) { const int b_ = (ii == TQ_STAMP_BLOCK ? 0 : 32) + (slot); D.stamps[2 * b_] = clock64(); D.stamps[2 * b_ + 1] = wall_clock64(); } } while (0)
#else
#define W3STAMP(slot) do { } while (0)
#define W3BSTAMP(k_) do { } while (0)
#endif
#ifdef TQ_W3_SYNC
#define W3_BARRIER() __syncthreads()
#else
#define W3_BARRIER() lds_barrier()
#endif
__global__ void __launch_bounds__(WT, TQ_W3_WPS) k_hf_w(Tree T, Data D, Opts O, u64 *sch, int rs, unsigned tag, int h)
#if !TQ_HAS(TQP_W3)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int small_flag;
    __shared__ int simd_of[WW];
    const int ii = T.Np - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x, pwave = tid >> 6, lane = tid & 63, r16 = lane & 15, g = lane >> 4;
    /* Which wave plays which part is decided by where the hardware put it.  The pivot chains are issue-bound VALU work of the first two
     * roles; two workgroups share a CU, and with roles = wave numbers both would run their chains on SIMDs 0 and 1 while the vector
     * units of SIMDs 2 and 3 idle.  HW_ID tells a wave its SIMD and the workgroup its slot on the CU: the workgroup in an odd slot
     * starts its roles two SIMDs further on.  Any answer gives a valid permutation of the four parts (ranks of distinct keys). */
    const int hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        /* HW_REG_HW_ID: [5:4] SIMD_ID, [19:16] TG_ID */
    if (lane == 0) simd_of[pwave] = (hwid >> 4) & 3;
    int e[28];
#pragma unroll
    for (int i = 0; i < 28; i++) e[i] = T.desc[(size_t)DESC_INTS * ii + i];
    /* is this launch due?  ONE thread looks and the workgroup takes its answer: another workgroup may end the solve (a wait that gave up)
     * while this one starts, and waves that looked for themselves could disagree about leaving before a barrier */
    __shared__ int due;
    if (tid == 0) due = phase_main(D.ctrl, h) ? 1 : 0;
    W3_BARRIER();
    if (!due) return;
    int wave;
    {
#ifdef TQ_W3_NO_ROT
        const int rot = 0;
#else
        const int rot = 2 * ((hwid >> 16) & 1);
#endif
        const int mykey = (((simd_of[pwave] - rot) & 3) << 2) | pwave;
        int rank = 0;
#pragma unroll
        for (int w = 0; w < WW; w++) { const int key = (((simd_of[w] - rot) & 3) << 2) | w; rank += key < mykey ? 1 : 0; }
        wave = __builtin_amdgcn_readfirstlane(rank);
    }
    Ctrl *c = D.ctrl;
    const int d = e[0], nxp = e[1], nup = e[2], nkp = e[3], k0 = e[4], nz = nxp + nup, nxi = ii > 0 ? nxp : 0;
    const int dp = up16(d), R = dp + 1 + nxi, Rp = up16(R), ld = wide_ldf(Rp);
    const int ldc = ld;                        /* C is staged in rows 0 .. dp - 1 of the image's first 32 columns: rows dp .. are free for the right-hand side and Ut */
    const int ct = dp >> 4, rt = Rp >> 4, nb = Rp - dp;
    lds_ptr Tm = to_lds(lds);
    lds_ptr Cs = Tm;                           /* the staging area of C IS the image (upper rows): W is formed in registers, then written over it */
    const int bo = e[7];
    const double *Qc = D.QinvCal + e[5], *Rc = D.RinvCal + e[6];
    const int nt2 = ii > 0 ? (nxi + 1 + 15) >> 4 : 1;            /* row tiles of Xt = rows dp .. R - 1 */
    const int wlast = ((d - 16 * (ct - 1)) + 3) & ~3;            /* columns of the last panel the chain runs over */
    const int ntl = ct * (ct + 1) / 2;                           /* lower tiles of W */
    constexpr int MP = 6;
    const int nrec = k0 < T.Np ? nkp * rs : 0;                   /* children that are leaves post no record */
    f64x4 gacc[3];                             /* waves 2, 3: Schur tiles */

    W3STAMP(0);
    W3BSTAMP(0);
    for (int pass = 0; pass < 2; pass++) {
        const double shift = (O.regType == 1 || pass == 1) ? O.regValue : 0.0;         /* ddiare */
        if (tid == 0) small_flag = 0;
        /* ---- H: C = [A B] of the children into LDS (32 columns, zero beyond nz); rows dp .. of the image zeroed ---- */
        if (lane < dp) for (int col = wave; col < 32; col += WW) Cs[lane + col * ldc] = 0.0;
        if (lane < nb) for (int col = wave; col < dp; col += WW) Tm[dp + lane + col * ld] = 0.0;
        double pcs[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { const int col = 4 * m + g; const int cs = col < nz ? col : 0; pcs[m] = cs < nxp ? Qc[cs] : Rc[cs - nxp]; }
        const double rsd = D.resMod[bo + (tid < d ? tid : 0)];
        W3_BARRIER();
#ifndef TQ_W3_CLOAD1
#define TQ_W3_CLOAD1 1
#endif
        if (!TQ_W3_CLOAD1 && nkp <= 3 && nz <= 8 * WW) {
            /* the usual case: every element of C requested before the first is stored (one memory round trip, not one per child) */
            double a[3][8], qcol[8];
            const int nx0c = e[16], nx1c = nkp > 1 ? e[19] : 0, nx2c = nkp > 2 ? e[22] : 0;
            const int nxcs[3] = {nx0c, nx1c, nx2c}, aos[3] = {e[17], e[20], e[23]}, bos[3] = {e[18], e[21], e[24]}, roffs[3] = {0, nx0c, nx0c + nx1c};
#pragma unroll
            for (int m = 0; m < 8; m++) qcol[m] = Qc[wave + WW * m < nxi ? wave + WW * m : 0];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const bool rowok = cc < nkp && lane < nxcs[cc];
                const int i = rowok ? lane : 0;
                const double *A = D.A + (cc < nkp ? aos[cc] : aos[0]), *B = D.B + (cc < nkp ? bos[cc] : bos[0]);
                const int nxc = cc < nkp ? nxcs[cc] : nxcs[0];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int col = wave + WW * m;
                    const int cs = (rowok && col < nz) ? col : 0;
                    const bool st = cs < nxp;
                    a[cc][m] = (st ? A : B)[i + (st ? cs : cs - nxp) * nxc];
                }
            }
            LOADS_DONE();
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const bool rowok = cc < nkp && lane < nxcs[cc];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int col = wave + WW * m;
                    if (rowok && col < nz) Cs[roffs[cc] + lane + col * ldc] = a[cc][m];
                    if (rowok && col < nxi) Tm[dp + 1 + col + (roffs[cc] + lane) * ld] = -1.0 * (a[cc][m] * qcol[m]);      /* Ut = -(C[:, :nx] P)' */
                }
            }
        } else
        for (int cc = 0, rowoff = 0; cc < nkp; cc++) {
            const int kid = k0 + cc;
            const bool rec = cc < 4;
            const int rnx = cc == 0 ? e[16] : cc == 1 ? e[19] : cc == 2 ? e[22] : e[25];
            const int rao = cc == 0 ? e[17] : cc == 1 ? e[20] : cc == 2 ? e[23] : e[26];
            const int rbo = cc == 0 ? e[18] : cc == 1 ? e[21] : cc == 2 ? e[24] : e[27];
            const int nxc = rec ? rnx : T.nx[kid];
            const double *A = D.A + (rec ? rao : T.aoff[kid]), *B = D.B + (rec ? rbo : T.boff[kid]);
            const bool rowok = lane < nxc;
            const int i = rowok ? lane : 0;
            for (int c0 = 0; c0 < nz; c0 += 8 * WW) {
                double a[8], qcol[8];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int col = c0 + wave + WW * m;
                    const int cs = (rowok && col < nz) ? col : 0;
                    const bool st = cs < nxp;
                    a[m] = (st ? A : B)[i + (st ? cs : cs - nxp) * nxc];
                    qcol[m] = Qc[col < nxi ? col : 0];
                }
                LOADS_DONE();
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int col = c0 + wave + WW * m;
                    if (rowok && col < nz) Cs[rowoff + i + col * ldc] = a[m];
                    /* Ut = -(C[:, :nx] P)': row 1 + col of the rows below, column rowoff + i */
                    if (rowok && col < nxi) Tm[dp + 1 + col + (rowoff + i) * ld] = -1.0 * (a[m] * qcol[m]);
                }
            }
            rowoff += nxc;
        }
        if (tid < d) Tm[dp + tid * ld] = rsd;                      /* row dp: the right-hand side (the children's residuals) */
        W3_BARRIER();                       /* C is in LDS */
        W3STAMP(1);
        /* W = C P C' + diag(QinvCal of the children): lower tiles dealt over the waves, three at most per wave (ct <= 4), their
         * accumulators taking turns on the matrix pipe */
        f64x4 wacc[3];
        int wI[3], wJ[3];
        bool wv[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { wv[t] = w3_lower_tile(wave + WW * t, ct, wI[t], wJ[t]); wacc[t] = f64x4{0.0, 0.0, 0.0, 0.0}; (void)ntl; }
        {
            double qd[3];
#pragma unroll
            for (int t = 0; t < 3; t++) { const int i = 16 * wI[t] + r16; qd[t] = D.QinvCal[bo + (i < d ? i : 0)]; }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    if (wv[t]) {          /* (uniform: a wave's tile slots that hold no tile take no turn on the matrix pipe) */
                        const double a = Cs[16 * wJ[t] + r16 + (4 * m + g) * ldc] * pcs[m];
                        const double b = Cs[16 * wI[t] + r16 + (4 * m + g) * ldc];
                        wacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, wacc[t], 0, 0, 0);
                    }
                }
            if (nz > 16) {
#pragma unroll
                for (int m = 4; m < 8; m++)
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        if (wv[t]) {
                            const double a = Cs[16 * wJ[t] + r16 + (4 * m + g) * ldc] * pcs[m];
                            const double b = Cs[16 * wI[t] + r16 + (4 * m + g) * ldc];
                            wacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, wacc[t], 0, 0, 0);
                        }
                    }
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = 16 * wI[t] + r16;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int j = 16 * wJ[t] + g + 4 * q;
                    if (i == j) wacc[t][q] = i < d ? wacc[t][q] + qd[t] + shift : 1.0;      /* padding: identity */
                }
            }
        }
        W3_BARRIER();                       /* everybody is done with C: the image may be written */
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (wv[t]) {
#pragma unroll
                for (int q = 0; q < 4; q++) Tm[16 * wI[t] + r16 + (16 * wJ[t] + g + 4 * q) * ld] = wacc[t][q];
            }
        W3_BARRIER();
        W3STAMP(2);
        /* ---- the children's Schur records, ONE batch of polls per MP x 256 entries: subtract G[gi][gj], gj >= 1, from the diagonal
         * sub-block of W at the child's position, G[gi][0] from the right-hand side row.  Flat index F = tid + 256 m over [child][rs];
         * entry f of a child's record is (gi, gj) = (f / w, f % w), w = nx_child + 1: which words this thread polls and where their
         * values go does not depend on the data and is worked out before the wait. ---- */
        if (nrec > 0) {
            bool dead = false;
            const float rrs = 1.0f / (float)rs;
            for (int F0 = 0; F0 < nrec; F0 += MP * WT) {
                int dst[MP];
                unsigned src[MP];              /* word index into sch */
#pragma unroll
                for (int m = 0; m < MP; m++) {
                    const int F = F0 + tid + WT * m;
                    const bool in = F < nrec;
                    const int Fc = in ? F : 0;
                    int cc = (int)(((float)Fc + 0.5f) * rrs);
                    int f = Fc - cc * rs;
                    if (f < 0) { cc--; f += rs; } else if (f >= rs) { cc++; f -= rs; }
                    const int kid = k0 + cc;
                    const int nxc = T.nx[kid], posc = T.pos[kid];
                    const int w = nxc + 1;
                    int gi = (int)(((float)f + 0.5f) / (float)w);
                    int gj = f - gi * w;
                    if (gj < 0) { gi--; gj += w; } else if (gj >= w) { gi++; gj -= w; }
                    const bool use = in && kid < T.Np && gi >= 1 && gi <= nxc && gj <= gi;
                    dst[m] = !use ? -1 : (gj == 0 ? dp + (posc + gi - 1) * ld : (posc + gi - 1) + (posc + gj - 1) * ld);
                    src[m] = ((unsigned)kid * (unsigned)rs + (unsigned)(use ? f : 0)) * 2u;
                }
                double val[MP];
                unsigned pending = 0u;
#pragma unroll
                for (int m = 0; m < MP; m++) if (dst[m] >= 0) pending |= 1u << m;
                const u64 t0 = wall_clock64();
                int looks = 0;
                while (pending) {
#pragma unroll
                    for (int m = 0; m < MP; m++) {
                        if (pending & (1u << m)) {
                            bool ok = true;
                            const double v = ld_tag(sch + src[m], tag, ok);
                            if (ok) { val[m] = v; pending &= ~(1u << m); }
                        }
                    }
                    if (!pending) break;
                    looks++;
                    if (wall_clock64() - t0 > 50000000ull) {      /* 0.5 s at 100 MHz: cannot happen (children are started first) */
#pragma unroll
                        for (int m = 0; m < MP; m++) if (pending & (1u << m)) val[m] = 0.0;
                        pending = 0u; dead = true;
                        break;
                    }
#ifndef TQ_W3_BACKOFF
#define TQ_W3_BACKOFF 0
#endif
                    /* the upper levels wait tens of microseconds while the levels below them work, and every look of a waiting workgroup
                     * (12 loads per thread) is in the way of the working ones: look less often the longer the wait already is */
                    if (TQ_W3_BACKOFF >= 2 && looks > 24) __builtin_amdgcn_s_sleep(16);
                    else if (TQ_W3_BACKOFF >= 1 && looks > 6) __builtin_amdgcn_s_sleep(4);
                    else __builtin_amdgcn_s_sleep(TQ_WIDE_NAP);
                }
#pragma unroll
                for (int m = 0; m < MP; m++) if (dst[m] >= 0) Tm[dst[m]] -= val[m];
            }
            if (dead) { c->status = 3; __hip_atomic_store(&c->done, 1, RLX, AGENT); }
            W3_BARRIER();
        }
        W3STAMP(3);
        W3BSTAMP(1);

        /* ---- blocked tall Cholesky with look-ahead ---- */
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) gacc[s3] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int p = 0; p < ct; p++) {
            const int kb = 16 * p;
            const int wp = p == ct - 1 ? wlast : 16;
            const int nbelow = R - kb - 16;                  /* real rows below the diagonal tile */
            const int nb2 = nbelow + 16;                     /* + identity rows: they come out as the inverse of the diagonal tile */
            const int nchain = (nb2 + 47) / 48;
            if (wave < nchain) {
                /* chain waves beyond wave 0 bring their own rows of this panel up to date first (wave 0's were done by all four waves
                 * before the last barrier) */
                if (p > 0 && wave >= 1) {
                    const int I0 = p + 1 + 3 * wave;
                    if (I0 < rt) w3_tile_update3(Tm, ld, kb - 16, r16, g, I0, p, true, I0 + 1, p, I0 + 1 < rt, I0 + 2, p, I0 + 2 < rt);
                }
                double Tr[16];
                const int v = 48 * wave + lane - 16;
                const bool diag = lane < 16, real = !diag && v < nbelow, ident = !diag && !real && v < nb2;
                const int row = diag ? kb + lane : kb + 16 + (real ? v : 0);
                const int idn = v - nbelow;
                /* (all sixteen reads go out before the first select: the optimiser otherwise puts each read under the predicate that
                 * masks its result -- sixteen LDS round trips, one after the other) */
#pragma unroll
                for (int j = 0; j < 16; j++) Tr[j] = Tm[row + (kb + j) * ld];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 16; j++) Tr[j] = (diag || real) ? Tr[j] : ((ident && idn == j) ? 1.0 : 0.0);
#ifdef TQ_WIDE_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W3STAMP(16 + 3 * p);
#endif
                const double pmin = w3_chain(Tr, lane, wp);
#ifdef TQ_WIDE_STAMPS
                asm volatile("" :: "v"(Tr[15]), "v"(Tr[0]), "v"(pmin) : "memory");
                W3STAMP(17 + 3 * p);
#endif
                if (real) {
#pragma unroll
                    for (int j = 0; j < 16; j++) Tm[row + (kb + j) * ld] = Tr[j];
                }
                if (ident) {
                    /* identity row i ends as column i of L_pp^-1: element (k, i) at slot + i + k * uls */
                    int uls;
                    const int us = w3_uslot(p, ct, ld, dp, uls);
#pragma unroll
                    for (int j = 0; j < 16; j++) Tm[us + idn + j * uls] = Tr[j];
                }
                if (wave == 0 && lane == 0 && pmin <= O.regTol * O.regTol) small_flag = 1;   /* sqrt(pivot) <= regTol, incl. non-positive pivots */
#ifdef TQ_WIDE_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W3STAMP(18 + 3 * p);
#endif
            } else if (p > 0) {
                /* the rest of panel p - 1's trailing update (column panels p + 1 ..), three tiles at a time */
                const int nrest = WW - nchain, me = (WW - 1) - wave;      /* the last wave first: it has the smaller share of the Schur tiles */
                for (int n0 = me; ; n0 += 3 * nrest) {
                    int I0, J0, I1, J1, I2, J2;
                    const bool v0 = w3_trailing_tile(n0, p + 1, ct, rt, I0, J0);
                    if (!v0) break;
                    const bool v1 = w3_trailing_tile(n0 + nrest, p + 1, ct, rt, I1, J1), v2 = w3_trailing_tile(n0 + 2 * nrest, p + 1, ct, rt, I2, J2);
                    w3_tile_update3(Tm, ld, kb - 16, r16, g, I0, J0, true, I1, J1, v1, I2, J2, v2);
                }
            }
            if (wave >= 2 && p > 0 && ii > 0) {
                /* G += Xt[:, panel p - 1] Xt[:, panel p - 1]' (tiles that do not exist -- nx <= 15: one row tile, only (0,0) -- are left out:
                 * their products would only take the matrix pipe from the trailing update) */
                int I2[3], J2[3];
                bool sv[3];
#pragma unroll
                for (int t = 0; t < 3; t++) { sv[t] = w3_schur_tile(wave, t, nt2, I2[t], J2[t]); if (!sv[t]) { I2[t] = 0; J2[t] = 0; } }
#pragma unroll
                for (int s = 0; s < 16; s += 4)
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        if (sv[t]) {          /* (uniform) */
                            const double a = Tm[dp + 16 * J2[t] + r16 + (kb - 16 + s + g) * ld];
                            const double b = Tm[dp + 16 * I2[t] + r16 + (kb - 16 + s + g) * ld];
                            gacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, gacc[t], 0, 0, 0);
                        }
                    }
            }
            W3_BARRIER();                   /* panel p is factorised (rows below the diagonal tile are in LDS), panel p - 1 is applied everywhere */
            W3STAMP(4 + 2 * p);
            if (p + 1 < ct) {
                /* look-ahead: the rows of column panel p + 1 that wave 0 takes next, one tile per wave */
                const int I = p + 1 + wave;
                if (I < rt) w3_tile_update(Tm, ld, I, p + 1, kb, r16, g);
                W3_BARRIER();
                W3STAMP(5 + 2 * p);
            }
        }
        /* on-the-fly Levenberg-Marquardt: any diagonal entry <= regTol -> shift and refactorise (rare) */
        if (O.regType != 2 || pass == 1 || !small_flag) break;
        W3_BARRIER();
        if (tid == 0) atomicAdd(&c->n_reg, 1);
    }

    /* ---- Schur record: the last panel's term, then G[gi][gj] (1 <= gi <= nxi, gj <= gi) to the parent as tagged words ---- */
    if (wave >= 2 && ii > 0) {
        const int kl = 16 * (ct - 1);
        int I2[3], J2[3];
        bool sv[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { sv[t] = w3_schur_tile(wave, t, nt2, I2[t], J2[t]); if (!sv[t]) { I2[t] = 0; J2[t] = 0; } }
#pragma unroll
        for (int s = 0; s < 16; s += 4) {
            const bool kin = s < wlast;                  /* columns of the last panel that exist */
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (kin && sv[t]) {                      /* (uniform; a product that is left out would add exact zeros) */
                    const double a = Tm[dp + 16 * J2[t] + r16 + (kl + s + g) * ld];
                    const double b = Tm[dp + 16 * I2[t] + r16 + (kl + s + g) * ld];
                    gacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, gacc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int gi = 16 * I2[t] + r16, gj0 = 16 * J2[t] + g;
            u64 *rp = sch + ((size_t)ii * rs + (size_t)gi * (nxi + 1) + gj0) * 2;      /* entries gj0 + 4 q: 64 bytes apart */
            const bool rowin = sv[t] && gi >= 1 && gi <= nxi;
#pragma unroll
            for (int q = 0; q < 4; q++) if (rowin && gj0 + 4 * q <= gi) st_tag(rp + 8 * q, gacc[t][q], tag);
        }
    }
    W3_BARRIER();                           /* the substitution below overwrites the rows the Schur term was just read from */
    W3STAMP(12);
    W3BSTAMP(2);

    /* ---- forward sweep prepared: Xt <- Xt L^-1 (rows dp ..: row 0 becomes z0 = L^-T y, row 1 + i column i of M = L^-T CholUt'), blocked
     * from the last column panel to the first, the inverses of the diagonal tiles from the identity rows.  Row tile a of Xt is wave a's:
     * no barrier.  Root: row 0 is the step of the root block itself. ---- */
    if (wave < nt2) {
        const int a = wave;
        const int c16 = 16 * a + r16;                  /* row of Xt this lane's results belong to */
        double *Mg = D.CholUt + e[9];
        double pd = 0.0;
        double resv[4][4];                             /* root: the residual for res' * dlam, requested before the chain of stages */
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++)
#pragma unroll
            for (int q = 0; q < 4; q++) { const int j = 16 * p4 + g + 4 * q; resv[p4][q] = (ii == 0 && c16 == 0 && j < d) ? D.res[bo + j] : 0.0; }
        for (int p = ct - 1; p >= 0; p--) {
            const int wq_p = p == ct - 1 ? wlast : 16;
            /* every operand of the stage is requested before the first product; the (up to three) off-diagonal products have an
             * accumulator each */
            f64x4 acc[3];
            double av[3][4], bv[3][4], uv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { acc[0][q] = Tm[dp + 16 * a + r16 + (16 * p + g + 4 * q) * ld]; acc[1][q] = 0.0; acc[2][q] = 0.0; }
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int q2 = p + 1 + t;
                const bool on = q2 < ct;
                const int q2c = on ? q2 : p;               /* (clamped: finite values, multiplied by zero) */
                const int wq = q2 == ct - 1 ? wlast : 16;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const bool kin = on && 4 * s < wq;
                    const double x = Tm[16 * q2c + 4 * s + g + (16 * p + r16) * ld];           /* L[16 q2 + k][16 p + j] */
                    const double y = Tm[dp + 16 * a + r16 + (16 * q2c + 4 * s + g) * ld];       /* Xt[c][16 q2 + k] */
                    av[t][s] = kin ? -x : 0.0; bv[t][s] = kin ? y : 0.0;
                }
            }
            int uls;
            const int us = w3_uslot(p, ct, ld, dp, uls);
#pragma unroll
            for (int n = 0; n < 4; n++) { const double x = Tm[us + r16 + (4 * n + g) * uls]; uv[n] = 4 * n < wq_p ? x : 0.0; }      /* L_pp^-1[k][j] */
#pragma unroll
            for (int t = 0; t < 3; t++) {
#ifdef TQ_W3_PREP_ALL
                {
#else
                if (p + 1 + t < ct) {              /* (uniform) */
#endif
#pragma unroll
                    for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t][s], bv[t][s], acc[t], 0, 0, 0);
                }
            }
            f64x4 sum;
#pragma unroll
            for (int q = 0; q < 4; q++) sum[q] = (acc[0][q] + acc[1][q]) + acc[2][q];
            /* times the inverse of the diagonal tile: the n-th group of four columns of the accumulator IS the B operand of k-step n */
            f64x4 out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int n = 0; n < 4; n++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(uv[n], 4 * n < wq_p ? sum[n] : 0.0, out, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = 16 * p + g + 4 * q;
                Tm[dp + 16 * a + r16 + j * ld] = out[q];
                if (j < d) {
                    if (c16 == 0) {
                        D.dlam[bo + j] = out[q];
                        D.CholW[e[8] + d * d - d + j] = out[q];          /* a copy that no forward sweep overwrites (k_fwd3c reads its ancestors' z0) */
                        pd = fma(p == 0 ? resv[0][q] : p == 1 ? resv[1][q] : p == 2 ? resv[2][q] : resv[3][q], out[q], pd);
                    }
                    else if (c16 <= nxi) Mg[(size_t)(c16 - 1) * d + j] = out[q];
                }
            }
        }
        if (ii == 0) {
            pd = wave_sum(pd);
            if (lane == 0) D.part_dot[0] = pd;
        }
        W3STAMP(13);
        W3BSTAMP(3);
    }
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_fwd3: forward sweep with the prepared [z0 | M], one wave per block, parents first;         */
/* tail: res' dlam, direction test, start of the line search                                   */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(SG_WAVES * WAVE) k_fwd3(Tree T, Data D, W3 Wd, u64 *fw, unsigned ftag, int h)
#if !TQ_HAS(TQP_W3)
;
#else
{
    __shared__ double part[SG_WAVES];
    Ctrl *c = D.ctrl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ii = 1 + (int)blockIdx.x * SG_WAVES + wave;
    const bool have = ii < T.Np;
    int e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = T.desc[(size_t)DESC_INTS * (have ? ii : 0) + i];
    if (!phase_main(c, h)) return;
    double pd = 0.0;
    if (have) {
        const int d = e[0], nxi = e[1], bo = e[7], xo = e[5], dad = e[10];
        const int lc = lane < d ? lane : 0;
        const double *Mg = D.CholUt + e[9];
        double mrow[32];                                   /* nx <= 32 on this path (checked at create time) */
#pragma unroll
        for (int i = 0; i < 32; i++) mrow[i] = Mg[(size_t)(i < nxi ? i : 0) * d + lc];
        const double z0 = D.dlam[bo + lc], rv = D.res[bo + lc];
        double val = 0.0;
        const int li = lane < nxi ? lane : 0;
        if (dad == 0) val = D.dlam[xo + li];               /* the root's step: written by k_hf_w, an earlier launch */
        LOADS_DONE();
        if (dad != 0) {
            const u64 *src = fw + (size_t)(xo + li) * 2;
            const u64 t0 = wall_clock64();
            for (;;) {
                bool ok = true;
                val = ld_tag(src, ftag, ok);
                if (__all(ok)) break;
                if (wall_clock64() - t0 > 50000000ull) {                      /* 0.5 s at 100 MHz */
                    if (lane == 0) { c->status = 3; __hip_atomic_store(&c->done, 1, RLX, AGENT); }
                    break;
                }
                __builtin_amdgcn_s_sleep(TQ_WIDE_NAP);
            }
        }
        double a0 = 0.0, a1 = 0.0;
        /* (groups of eight rows of M that do not exist are skipped: nx = 8 takes a quarter of the chain of broadcasts; the terms that
         * are left out are exact zeros) */
#pragma unroll
        for (int i0 = 0; i0 < 32; i0 += 8) {
            if (i0 < nxi) {
#pragma unroll
                for (int i = i0; i < i0 + 8; i += 2) {
                    a0 = fma(i < nxi ? mrow[i] : 0.0, rdlane(val, i), a0);
                    a1 = fma(i + 1 < nxi ? mrow[i + 1] : 0.0, rdlane(val, i + 1), a1);
                }
            }
        }
        const double mine = z0 - (a0 + a1);
        if (lane < d) {
            st_tag(fw + (size_t)(bo + lane) * 2, mine, ftag);                 /* first: my children wait for it */
            D.dlam[bo + lane] = mine; pd = rv * mine;
        }
        pd = wave_sum(pd);
        if (lane == 0) D.part_dot[ii] = pd;
    }
    if (lane == 0) part[wave] = pd;
    __syncthreads();
    if (wave != 0) return;
    if (lane == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SG_WAVES; w++) s += part[w];
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 0) * 2, s, Wd.tag);
    }
    Fuse F; F.red = nullptr; F.cnt = Wd.cnt; F.tag = Wd.tag; F.on = 1;
    if (!fuse_last(F, (int)gridDim.x, lane)) return;
    const double s = w3_reduce<false>(Wd.red, 0, (int)gridDim.x, Wd.tag, lane) + D.part_dot[0];      /* + the root's, from k_hf_w */
    if (lane == 0) {
        const double dotp = -s;                                     /* :819 */
        c->dot = dotp;
        if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) { c->done = 1; c->status = 2; }      /* :951, NaN included */
        else { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
    }
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* k_fwd3c: the forward sweep WITHOUT hand-overs, for trees of small nodes (nx <= 8 everywhere, at most 16 blocks on a path from the   */
/* root).  A block's step is z0 - M' (its node's slice of its dad's step), and that slice is eight numbers that depend on the path to   */
/* the root only: every wave recomputes the slices of its ancestors -- a chain of at most 16 eight-term sums by lane broadcasts, ~30     */
/* instructions each -- instead of waiting for them level by level (k_fwd3: ten levels of a pruned tree = ten hand-overs, 13 us).       */
/* Lane group g (eight lanes) of the wave holds path entry g (second round: 8 + g): the eight rows of M of that ancestor restricted to   */
/* the columns of the path's node, and z0 of those columns (from the copies k_hf_w leaves in CholW: dlam itself is being overwritten   */
/* by the ancestors' own waves).  Same sums in the same order as k_fwd3 (even / odd accumulators): bit-identical steps.                 */
/* anc: per block 1 + 16 x 4 ints: path length L, then {z0 offset, M offset, d of that ancestor, nx of the ancestor's node (0: the root  */
/* block, whose z0 is the step) | nx of the path's node << 8} from the root's block down to the block's dad.  Same tail as k_fwd3.      */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(SG_WAVES * WAVE) k_fwd3c(Tree T, Data D, W3 Wd, const int *anc, int h)
#if !TQ_HAS(TQP_W3)
;
#else
{
    __shared__ double part[SG_WAVES];
    Ctrl *c = D.ctrl;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ii = 1 + (int)blockIdx.x * SG_WAVES + wave;
    const bool have = ii < T.Np;
    const int iic = have ? ii : 1;
    int e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = T.desc[(size_t)DESC_INTS * iic + i];
    if (!phase_main(c, h)) return;
    double pd = 0.0;
    if (have) {
        const int d = e[0], bo = e[7];
        const double rv = D.res[bo + (lane < d ? lane : 0)];
        double val;
        int gl;
        const double mine = fwdc_block(D, anc + (size_t)FWDC_INTS * ii, d, e[1], e[8], e[9], lane, val, gl);
        if (lane < d) { D.dlam[bo + lane] = mine; pd = rv * mine; }
        pd = wave_sum(pd);
        if (lane == 0) D.part_dot[ii] = pd;
    }
    if (lane == 0) part[wave] = pd;
    __syncthreads();
    if (wave != 0) return;
    if (lane == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SG_WAVES; w++) s += part[w];
        st_tag(Wd.red + ((size_t)2 * blockIdx.x + 0) * 2, s, Wd.tag);
    }
    Fuse F; F.red = nullptr; F.cnt = Wd.cnt; F.tag = Wd.tag; F.on = 1;
    if (!fuse_last(F, (int)gridDim.x, lane)) return;
    const double s = w3_reduce<false>(Wd.red, 0, (int)gridDim.x, Wd.tag, lane) + D.part_dot[0];      /* + the root's, from k_hf_w */
    if (lane == 0) {
        const double dotp = -s;                                     /* :819 */
        c->dot = dotp;
        if (dotp > 1e-10 || !((dotp > 1e-10) || (dotp < 1e-10))) { c->done = 1; c->status = 2; }      /* :951, NaN included */
        else { c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1; }
    }
}
#endif

/* LDS of k_hf_w for a block of dimension d under a parent of nz = nx + nu columns */
static inline size_t wide3_lds(int d, int nxi, int nz) {
    const int dp = (d + 15) & ~15, kz = (nz + 3) & ~3;
    (void)nz; (void)kz;                        /* C (dp x kz, kz <= 32 <= dp) is staged inside the image */
    return ((size_t)wide_ldf(wide_rows(d, nxi)) * dp + (dp < 64 ? (size_t)256 * (dp >> 4) : 0)) * sizeof(double);
}

#if TQ_HAS(TQP_W3)
template __global__ void k_sgp_t<false>(Tree, Data, Opts, W3, int, int, int, int, const double *, SgpFwdNo);
template __global__ void k_sgp_t<true>(Tree, Data, Opts, W3, int, int, int, int, const double *, SgpFwdYes);
#endif
