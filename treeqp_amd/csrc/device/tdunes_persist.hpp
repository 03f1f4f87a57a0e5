/*
 * tdunes_persist.hpp -- the whole dual-Newton solve as ONE persistent launch (uniform complete trees).
 *
 * Included by tdunes_device.hip after tdunes_fast.hpp (same block-level device functions).
 *
 * The tiered path (tdunes_fast.hpp) spends ~1/3 of an iteration in kernel boundaries: a gap of
 * ~2.3 us plus ~2.5 us of cold global loads per boundary, 7 boundaries per iteration.  Here every
 * tier subtree keeps its workgroup for the whole solve:
 *   - grid = one 4-wave workgroup per tier subtree (C2: 64 + 8 + 1 = 73), all co-resident;
 *   - a block's W / L, Ut / CholUt, residual, backward solution and reciprocal diagonal live in
 *     the workgroup's LDS (25 KB per tier subtree) -- global memory only carries what crosses
 *     workgroups or must survive the launch (x, u, multipliers, step, boundary Schur records);
 *   - workgroups hand over through agent-scope words (MI355X guide, Guideline 16, recipe R1: payload
 *     with sc1 stores, every storing wave drains vmcnt, workgroup barrier, ONE relaxed agent atomic;
 *     the consumer polls relaxed with s_sleep and reads the payload with sc1 loads only):
 *       up_cnt[parent]  children -> parent, Schur records of the subtree roots are in Sbuf,
 *       down[wg]        parent -> children, the forward sweep has written the step of my blocks,
 *       st_cnt[parent]  children -> parent, the stage sweep has rewritten the nodes I own,
 *       err_cnt         everybody -> top, termination partial of my blocks is in errp[wg],
 *       arrive / go     ticket of the {fval, dot} reduction and its decision,
 *       halt            top -> everybody, the launch is over;
 *   - the FIRST sweep of a solve (stage QPs at lambda0, fval0) is the launch's prologue;
 *   - an iteration does not wait for the line-search decision of the previous one: the first trial
 *     (tau = 1) is accepted almost always, so every workgroup goes straight on to G + H and the
 *     backward sweep of the next iteration at the trial point.  Only the top workgroup looks at
 *     the decision, before anything irreversible (termination verdict, forward sweep, next trial);
 *     if the trial was NOT accepted it halts the launch and the speculative work is simply dropped
 *     (it only touched LDS, Sbuf and errp, which every iteration rebuilds);
 *   - termination is decided from the flat errp[] array as soon as every workgroup has done G + H,
 *     i.e. long before the backward sweep of a converged point would have reached the top;
 *   - the trial stage sweep runs four nodes per wave (16 lanes per node) for the nodes a workgroup
 *     owns; the line-search decision is taken by the LAST workgroup to arrive at the ticket, which
 *     sums the per-workgroup {fval, dot} partials in workgroup order;
 *   - every spin is bounded (wall clock); a timeout ends the launch with status UNKNOWN_ERROR.
 * Extra line-search trials (rare) end the launch: the host runs them with the ordinary trial kernels
 * and relaunches (without prologue); nothing but global memory carries state across launches.
 */
#pragma once

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, RLX, AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* inter-workgroup words of one persistent launch (zeroed by the host before every launch; all
 * counters are monotonic within a launch, `e` below is the launch-relative iteration number + 1) */
struct PSync {
    unsigned *up_cnt;       /* [G] backward arrivals of child subtrees at their parent workgroup      */
    unsigned *st_cnt;       /* [G] completed stage sweeps of child subtrees, counted at the parent    */
    unsigned *down;         /* [G] e, published by a workgroup after its forward sweep                */
    unsigned *arrive;       /* ticket counter of the {fval, dot} reductions                           */
    unsigned *go;           /* (decision number << 2) | code : 0 continue, 1 done, 2 more trials      */
    unsigned *err_cnt;      /* arrivals of termination partials                                      */
    unsigned *halt;         /* set by the top workgroup: everybody leaves at the next poll            */
    unsigned *timeout;      /* set when a bounded spin gave up                                       */
    double *parts;          /* [G][2] per-workgroup {fval, dot} partials                             */
    double *errp;           /* [G] per-workgroup termination partial                                 */
};

/* bounded poll by ONE lane until *w >= target; returns false when the launch is over instead
 * (halt or timeout), true when the target was reached */
__device__ __forceinline__ bool poll_ge(const unsigned *w, unsigned target, const PSync &Sy, unsigned *val = nullptr) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = __hip_atomic_load(w, RLX, AGENT);
        const unsigned h = __hip_atomic_load(Sy.halt, RLX, AGENT) | __hip_atomic_load(Sy.timeout, RLX, AGENT);
        if (v >= target) { if (val) *val = v; return true; }
        if (h) return false;
        if (wall_clock64() - t0 > 50000000ull) { __hip_atomic_store(Sy.timeout, 1u, RLX, AGENT); return false; }   /* 0.5 s at 100 MHz */
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int NX, int NU, int MD>
struct PLds {
    using U = Uni<NX, NU, MD>;
    static constexpr int D = U::D, NBT = U::NBT;
    static constexpr int DOUBLES = NBT * (D * D + NX * D + 4 * D) + NBT * U::SCH + NBT * D + FW * U::WAVE_LDS + 32;
    /* scratch reuse by the reductions: {fval, dot} partials over the whole block storage (free between
     * the stage sweep and the next G + H), termination partials over the Schur records (free before
     * the backward sweep) -- the host checks the grid against both capacities */
    static constexpr int PARTS_CAP = NBT * (D * D + NX * D + 4 * D + U::SCH) / 2, ERR_CAP = NBT * U::SCH;
    lds_ptr W, Ut, res, y, inv, dl, sch, wave0, wave;
    lds_iptr flag;
    __device__ PLds(double *base, int wave_id) {
        W = to_lds(base); Ut = W + NBT * D * D; res = Ut + NBT * NX * D; y = res + NBT * D; inv = y + NBT * D;
        dl = inv + NBT * D; sch = dl + NBT * D; wave0 = sch + NBT * U::SCH; wave = wave0 + wave_id * U::WAVE_LDS;
        flag = (lds_iptr)(wave0 + FW * U::WAVE_LDS);
    }
};

/* G + H of block p into LDS slot `loc`, split into a branch-free load half and a compute half so
 * that a wave with two blocks has both blocks' loads in flight at once.  Node data (x, u, QinvCal,
 * RinvCal) through sc1 loads: written by other workgroups' stage sweeps in the previous iteration. */
template <int NX, int NU, int MD>
struct GhRegs {
    double a[Uni<NX, NU, MD>::KS], pc[Uni<NX, NU, MD>::KS], z[Uni<NX, NU, MD>::KS], xk, bk, qk;
};

template <int NX, int NU, int MD>
__device__ __forceinline__ void p_gh_load(const Data &Dt, int p, int lane, GhRegs<NX, NU, MD> &G) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const int row = lane & 15, g = lane >> 4;
    const bool live = row < D;
    const int rowc = live ? row : 0;                         /* dead rows load row 0 and are masked */
    const int cidx = rowc / NX, r = rowc - cidx * NX;
    const int k = U::kid0(p) + cidx;
    const double *A = Dt.A + (size_t)(k - 1) * NX * NX + r;
    const double *B = Dt.B + (size_t)(k - 1) * NX * NU + r;
    const int bo = U::bo(p);
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const bool ok = live && cc < NZ, isx = cc < NX;
        const int cu = (cc < NZ) ? cc - NX : 0;              /* input column (clamped) */
        const double *ap = isx ? A + (size_t)cc * NX : B + (size_t)cu * NX;
        const double *pp = isx ? Dt.QinvCal + NX * p + cc : Dt.RinvCal + NU * p + cu;
        const double *zp = isx ? Dt.x + NX * p + cc : Dt.u + NU * p + cu;
        const double av = *ap, pv = ld_sc1(pp), zv = ld_sc1(zp);
        G.a[s] = ok ? av : 0.0; G.pc[s] = ok ? pv : 0.0; G.z[s] = ok ? zv : 0.0;
    }
    const double xv = ld_sc1(Dt.x + bo + rowc), bv = Dt.b[bo + rowc], qv = ld_sc1(Dt.QinvCal + bo + rowc);
    G.xk = (live && g == 0) ? xv : 0.0; G.bk = (live && g == 0) ? bv : 0.0; G.qk = live ? qv : 0.0;
}

template <int NX, int NU, int MD>
__device__ __forceinline__ double p_gh_compute(PLds<NX, NU, MD> &L, int loc, int lane, const GhRegs<NX, NU, MD> &G, int termCondition) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int row = lane & 15, g = lane >> 4;
    const bool live = row < D;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
    lds_ptr Ut = L.Ut + loc * NX * D;
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const double ap = G.a[s] * G.pc[s];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G.a[s], ap, acc, 0, 0, 0);
        part = fma(G.a[s], G.z[s], part);
        if (live && cc < NX) Ut[cc + row * NX] = -1.0 * ap;
    }
    part = rows_fold<false>(part);
    double e = 0.0;
    if (live && g == 0) {
        const double rv = fma(-1.0, G.xk, G.bk) + part;
        L.res[loc * D + row] = rv;
        e = (termCondition == 2) ? fabs(rv) : rv * rv;
    }
    lds_ptr W = L.W + loc * D * D;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = g + 4 * rr;
        if (live && i < D) {
            double w = acc[rr];
            if (i == row) w += G.qk;
            W[i + row * D] = w;
        }
    }
    return (termCondition == 2) ? wmax(e) : wsum(e);
}

template <int NX, int NU, int MD>
__device__ __forceinline__ void p_load_rows(PLds<NX, NU, MD> &L, int loc, int lane, bool is_root, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R;
    lds_cptr src; int stride;
    if (lane < D) { src = L.W + loc * D * D + lane; stride = D; }
    else if (lane == D) { src = L.res + loc * D; stride = 1; }
    else if (lane < R && !is_root) { src = L.Ut + loc * NX * D + (lane - D - 1); stride = NX; }
    else { src = L.W + loc * D * D; stride = D; }
#pragma unroll
    for (int j = 0; j < D; j++) T[j] = src[j * stride];
}

/* factor data of block `loc` back into LDS with ONE store per column (per-lane base + stride):
 * L over W (lanes < D), y (lane D), CholUt over Ut (lanes D+1 .. R-1), plus 1/diag */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_store_factor(PLds<NX, NU, MD> &L, int loc, int lane, const double (&T)[Uni<NX, NU, MD>::D], double myinv) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R;
    lds_ptr dst; int stride;
    if (lane < D) { dst = L.W + loc * D * D + lane; stride = D; }
    else if (lane == D) { dst = L.y + loc * D; stride = 1; }
    else { dst = L.Ut + loc * NX * D + (lane - D - 1); stride = NX; }
    if (lane < R) {
#pragma unroll
        for (int j = 0; j < D; j++) dst[j * stride] = T[j];
    }
    if (lane < D) L.inv[loc * D + lane] = myinv;
}

/* Schur record [S | v] = CUt * [CUt' | y] (one f64 MFMA tile, K = D) straight from the CholUt / y just
 * stored in LDS: lane (i, g) feeds CUt[i][g + 4 st] as A and the same (i < NX) or y (i == NX) as B.
 * GLOBAL: destination is global Sbuf (sc1 stores, another workgroup reads it), else an LDS record. */
template <int NX, int NU, int MD, bool GLOBAL>
__device__ __forceinline__ void p_schur(PLds<NX, NU, MD> &L, int loc, int lane, lds_ptr sdst_lds, double *sdst_glb) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    lds_fence();
    const int i = lane & 15, g = lane >> 4;
    /* per-lane base + stride, all loads issued before the first MFMA */
    lds_cptr src = (i < NX) ? L.Ut + loc * NX * D + i + g * NX : L.y + loc * D + g;
    const int stp = (i < NX) ? 4 * NX : 4;
    double m[D / 4];
#pragma unroll
    for (int st = 0; st < D / 4; st++) m[st] = src[st * stp];
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < D / 4; st++) {
        const double b = (i <= NX) ? m[st] : 0.0, a = (i < NX) ? m[st] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int ip = g + 4 * rr;
        if (ip < NX && i <= NX) {
            const int off = (i < NX) ? ip + i * NX : NX * NX + ip;
            if (GLOBAL) st_sc1(sdst_glb + off, acc[rr]); else sdst_lds[off] = acc[rr];
        }
    }
}

/* forward step of block `loc` from LDS; writes the solution to LDS (dl) and to global dlam (sc1) */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_forward(const Data &Dt, PLds<NX, NU, MD> &L, int ii, int loc, int lane, lds_cptr delta_lds, const double *delta_glb) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int li = lane < D ? lane : 0;
    lds_cptr Lc = L.W + loc * D * D + li * D;
    lds_cptr Cc = L.Ut + loc * NX * D + li * NX;
    double dv[NX];
    if (delta_glb) {
#pragma unroll
        for (int r = 0; r < NX; r++) dv[r] = ld_sc1(delta_glb + r);
    } else {
#pragma unroll
        for (int r = 0; r < NX; r++) dv[r] = delta_lds[r];
    }
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < NX; r++) acc = fma(Cc[r], dv[r], acc);
    double s = fma(-1.0, acc, L.y[loc * D + li]);
    const double inv = L.inv[loc * D + li];
    double Lcol[D];
#pragma unroll
    for (int k = 0; k < D; k++) Lcol[k] = Lc[k];
    double mine = 0.0;
#pragma unroll
    for (int k = D - 1; k >= 0; k--) {
        const double zk = rdlane(s * inv, k);
        if (lane == k) mine = zk;
        if (lane < k) s = fma(-Lcol[k], zk, s);
    }
    double pd = 0.0;
    if (lane < D) { st_sc1(Dt.dlam + U::bo(ii) + lane, mine); L.dl[loc * D + lane] = mine; pd = L.res[loc * D + lane] * mine; }
    return pd;                                            /* per-lane term of res' * dlam: summed once per sweep */
}

/* stage QP of node k at the trial point lam_cur + step*dlam, by ONE 16-lane group (lanes t of the
 * group: t < NX state entries, NX <= t < NX+NU input entries); all node-level global traffic is sc1.
 * init: first sweep of a solve -- evaluate at lamc itself (the step buffer may hold anything) and copy it to lamn.
 * Returns the node's dual-function term (valid in every lane of the group). */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_stage16(const Data &Dt, int k, int Np, int t, lds_ptr gl /* group scratch: D + NX */,
                                            double step, const double *lamc, double *lamn, bool active, bool init = false) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    static_assert(NX + NU <= 16 && D <= 16, "16-lane stage needs nx+nu <= 16 and d <= 16");
    const bool parent = active && k < Np;
    const int nuk = parent ? NU : 0;
    const int xo = NX * k, uo = NU * k, ko = U::bo(k);
    const bool isx = t < NX, live = active && t < NX + nuk;
    const int j = isx ? t : t - NX;
    /* branch-free loads: every lane reads from a valid (clamped) address and masks afterwards, so
     * all of the node's global loads are in flight together */
    const bool pk = parent && t < D, ox = active && isx && k > 0;
    const int ia = pk ? ko + t : 0, ib = ox ? xo + t : 0;
    const double dla = ld_sc1(Dt.dlam + ia), lca = ld_sc1(lamc + ia), ba = Dt.b[ia];
    const double dlb = ld_sc1(Dt.dlam + ib), lcb = ld_sc1(lamc + ib);
    const bool pl = parent && live;
    double col[MD][NX];
#pragma unroll
    for (int cc = 0; cc < MD; cc++) {
        const int kid = pl ? U::kid0(k) + cc : 1;
        const double *cp = (isx || !pl) ? Dt.A + (size_t)(kid - 1) * NX * NX + (size_t)(pl ? j : 0) * NX
                                        : Dt.B + (size_t)(kid - 1) * NX * NU + (size_t)j * NX;
#pragma unroll
        for (int i = 0; i < NX; i++) col[cc][i] = cp[i];
    }
    const bool lx = isx || !live;
    const int jo = live ? (isx ? xo + j : uo + j) : 0;
    double lin = (lx ? Dt.q : Dt.r)[jo], winv = (lx ? Dt.Qinv : Dt.Rinv)[jo], wd = (lx ? Dt.Qd : Dt.Rd)[jo];
    double lob = (lx ? Dt.xmin : Dt.umin)[jo], hib = (lx ? Dt.xmax : Dt.umax)[jo];
    double p_c = 0.0;
    {
        const double v = init ? lca : fma(step, dla, lca);
        if (pk) { gl[t] = v; p_c = ba * v; }
        const double w = ox ? (init ? lcb : fma(step, dlb, lcb)) : 0.0;
        if (ox) st_sc1(lamn + xo + t, w);
        if (active && isx) gl[D + t] = w;
    }
    lds_fence();
    double p_q = 0.0, p_h = 0.0;
    if (live) {
        double v = isx ? fma(-1.0, lin, gl[D + j]) : -1.0 * lin;
        if (parent) {
#pragma unroll
            for (int cc = 0; cc < MD; cc++) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NX; i++) acc = fma(col[cc][i], gl[cc * NX + i], acc);
                v = fma(-1.0, acc, v);
            }
        }
        const double unc = winv * v;
        double val, cal;
        if (unc >= hib) { val = hib; cal = 0.0; } else if (unc <= lob) { val = lob; cal = 0.0; } else { val = unc; cal = winv; }
        if (isx) { st_sc1(Dt.qmod + xo + j, v); st_sc1(Dt.xUnc + xo + j, unc); st_sc1(Dt.x + xo + j, val); st_sc1(Dt.QinvCal + xo + j, cal); }
        else { st_sc1(Dt.rmod + uo + j, v); st_sc1(Dt.uUnc + uo + j, unc); st_sc1(Dt.u + uo + j, val); st_sc1(Dt.RinvCal + uo + j, cal); }
        p_q = (wd * val) * val;
        p_h = v * val;
    }
    const double qx = row16_sum(isx ? p_q : 0.0), hx = row16_sum(isx ? p_h : 0.0);
    const double ru = row16_sum(isx ? 0.0 : p_q), hu = row16_sum(isx ? 0.0 : p_h);
    p_c = row16_sum(p_c);
    double f = -0.5 * qx - p_c;
    f += hx;
    f -= 0.5 * ru;
    f += hu;
    if (active && t == 0) st_sc1(Dt.fval + k, f);
    lds_fence();
    return active ? f : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* the persistent kernel                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* geometry of one workgroup: tiers are numbered bottom-up (0 = leaves side), workgroups tier by
 * tier starting with tier 0 */
struct PGeom {
    int n_tiers;
    int l0[8], l1[8], grid[8], wg0[8];      /* per tier: block levels [l0,l1), subtrees, first workgroup id */
    int G;
};

/* diagnostic stamps of the persistent kernel: first workgroup of every tier, thread 0, iteration O.stamps of the launch */
__device__ __forceinline__ void pstamp(const Data &Dt, const Opts &O, unsigned e, int tier, int s, int slot) {
    if (O.stamps == (int)e && threadIdx.x == 0 && s == 0 && slot < 32 && tier < 8) {
        Dt.stamps[(tier * 32 + slot) * 2 + 0] = clock64();
        Dt.stamps[(tier * 32 + slot) * 2 + 1] = wall_clock64();
    }
}

/* stage sweep over the nodes this workgroup owns (the owner nodes of its blocks in heap order, then --
 * bottom tier -- the leaves below), four nodes per wave; returns the wave's sum of the node terms */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_stage_owned(const Data &Dt, const Tree &T, PLds<NX, NU, MD> &L, int l0, int l1, int s, bool is_bottom, int wave, int lane,
                                                double step, const double *lamc, double *lamn, bool init) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int th = l1 - l0;
    const int nown = U::first(th) + (is_bottom ? U::width(th) : 0);
    const int grp = lane >> 4, t16 = lane & 15;
    lds_ptr gl = L.wave + 8 + grp * (D + NX + 2);
    double fsum = 0.0;
    for (int base = 0; base < nown; base += FW * 4) {
        const int q = base + wave * 4 + grp;
        const bool active = q < nown;
        int k = 0;
        if (active) {
            if (q < U::first(th)) {
                int t = 0; while (q >= U::first(t + 1)) t++;
                k = U::first(l0 + t) + s * U::width(t) + (q - U::first(t));
            } else {
                k = U::first(l1) + s * U::width(th) + (q - U::first(th));
            }
        }
        fsum += p_stage16<NX, NU, MD>(Dt, k, T.Np, t16, gl, step, lamc, lamn, active, init);
    }
    return rows_fold<false>(fsum);       /* every lane of a 16-lane group holds its group's sum */
}

/* {fval, dot} partial of this workgroup into parts[], ticket; the LAST workgroup to arrive takes the
 * decision number `nd` for everybody: nd == 0 with `prologue`: fval0 of the first sweep, otherwise the
 * first line-search trial (direction test + Armijo).  Called by all threads of the workgroup. */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_reduce_and_decide(Ctrl *c, const Data &Dt, const Opts &O, const PGeom &Gm, const PSync &Sy, PLds<NX, NU, MD> &L,
                                                    int wg, double f_wg, double d_wg, unsigned nd, bool first_sweep) {
    if (threadIdx.x == 0) {
        st_sc1(Sy.parts + 2 * wg, f_wg);
        st_sc1(Sy.parts + 2 * wg + 1, d_wg);
        drain_stores();
        const unsigned ticket = __hip_atomic_fetch_add(Sy.arrive, 1u, RLX, AGENT);
        *L.flag = (ticket == (nd + 1u) * (unsigned)Gm.G - 1u);
    }
    __syncthreads();
    if (*L.flag) {
        /* all threads fetch the partials in parallel, thread 0 sums them in workgroup order */
        lds_ptr pf = L.W, pd = L.W + Gm.G;                /* the block storage is free between the sweeps */
        for (int w = threadIdx.x; w < Gm.G; w += FW * WAVE) { pf[w] = ld_sc1(Sy.parts + 2 * w); pd[w] = ld_sc1(Sy.parts + 2 * w + 1); }
        __syncthreads();
        if (threadIdx.x == 0) {
            double fa = 0.0, da = 0.0;
            for (int w = 0; w < Gm.G; w++) { fa += pf[w]; da += pd[w]; }
            /* the control block was last written by another workgroup (or the host) */
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            unsigned code = 0u;
            if (first_sweep) { c->fval0 = fa; c->fval = fa; }
            else {
                c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
                if (ls_not_descent(c, -da)) code = 1u;
                else {
                    ls_decide_tail(c, Dt, O, fa);
                    code = c->done ? 1u : (c->ls_pending ? 2u : 0u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(Sy.go, ((nd + 1u) << 2) | code, RLX, AGENT);
        }
    }
    __syncthreads();
}

template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_persist(Tree T, Data Dt, Opts O, PGeom Gm, PSync Sy, const double *lam_init, int prologue) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Ctrl *c = Dt.ctrl;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    PLds<NX, NU, MD> L(lds_all, wave);
    const int wg = blockIdx.x;
    /* my tier / subtree */
    int tier = 0;
    for (int i = 0; i < Gm.n_tiers; i++) if (wg >= Gm.wg0[i]) tier = i;
    const int s = wg - Gm.wg0[tier];
    const int l0 = Gm.l0[tier], l1 = Gm.l1[tier], th = l1 - l0;
    const bool is_top = tier == Gm.n_tiers - 1, is_bottom = tier == 0;
    const int root_blk = U::first(l0) + s;                            /* subtree root block (= node) */
    const int parent_wg = is_top ? -1 : Gm.wg0[tier + 1] + ((root_blk - 1) / MD - U::first(l0 - 1)) / U::width(Gm.l1[tier + 1] - 1 - Gm.l0[tier + 1]);
    const unsigned nchild = is_bottom ? 0u : (unsigned)(U::width(th - 1) * MD);   /* child subtrees below my bottom level */
    if (__hip_atomic_load(&c->done, RLX, AGENT) || __hip_atomic_load(&c->ls_pending, RLX, AGENT)) return;
    int cur = __hip_atomic_load(&c->cur, RLX, AGENT);
    unsigned nd = 0u;          /* reductions (tickets) I have taken part in = decisions that exist or are under way */
    unsigned ns = 0u;          /* stage sweeps I have completed in this launch */

    /* the launch is over for this workgroup: leave together (the flag is workgroup-uniform) */
#define P_LEAVE_IF(cond) do { if (cond) return; } while (0)

    if (prologue) {
        /* ---- first sweep of the solve: stage QPs at lambda0 (copied into the current buffer), fval0 ---- */
        double *lam0 = cur ? Dt.lam1 : Dt.lam0;
        const double fsum = p_stage_owned<NX, NU, MD>(Dt, T, L, l0, l1, s, is_bottom, wave, lane, 0.0, lam_init, lam0, true);
        if (lane == 0) L.wave[2] = fsum;
        drain_stores();
        __syncthreads();
        double f = 0.0;
        for (int w = 0; w < FW; w++) f += L.wave0[w * U::WAVE_LDS + 2];
        if (threadIdx.x == 0 && !is_top) __hip_atomic_fetch_add(Sy.st_cnt + parent_wg, 1u, RLX, AGENT);
        ns = 1u;
        p_reduce_and_decide<NX, NU, MD>(c, Dt, O, Gm, Sy, L, wg, f, 0.0, nd, true);
        nd = 1u;
    }

    for (unsigned e = 1u;; e++) {
        const double *lamc = cur ? Dt.lam1 : Dt.lam0;
        double *lamn = cur ? Dt.lam0 : Dt.lam1;

        int sl = 0;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 0: iteration start */
        /* ---- the nodes my bottom-level blocks read are staged by the child workgroups ---- */
        if (!is_bottom) {
            if (threadIdx.x == 0) *L.flag = poll_ge(Sy.st_cnt + wg, ns * nchild, Sy) ? 0 : 1;
            __syncthreads();
            const int leave = *L.flag;
            __syncthreads();
            P_LEAVE_IF(leave);
        }
        /* ---- G + H for my blocks (heap order inside the subtree), two blocks per wave in flight ---- */
        double err = 0.0;
        {
            const int nbt = U::first(th);
            auto blk = [&](int loc) { int t = 0; while (loc >= U::first(t + 1)) t++; return U::first(l0 + t) + s * U::width(t) + (loc - U::first(t)); };
            for (int loc0 = wave; loc0 < nbt; loc0 += 2 * FW) {
                const int loc1 = loc0 + FW;
                GhRegs<NX, NU, MD> g0, g1;
                p_gh_load<NX, NU, MD>(Dt, blk(loc0), lane, g0);
                if (loc1 < nbt) p_gh_load<NX, NU, MD>(Dt, blk(loc1), lane, g1);
                double v = p_gh_compute<NX, NU, MD>(L, loc0, lane, g0, O.termCondition);
                err = (O.termCondition == 2) ? fmax(err, v) : err + v;
                if (loc1 < nbt) {
                    v = p_gh_compute<NX, NU, MD>(L, loc1, lane, g1, O.termCondition);
                    err = (O.termCondition == 2) ? fmax(err, v) : err + v;
                }
            }
            if (lane == 0) L.wave[0] = err;
            __syncthreads();
            if (threadIdx.x == 0) {
                err = 0.0;
                for (int w = 0; w < FW; w++) { const double v = L.wave0[w * U::WAVE_LDS]; err = (O.termCondition == 2) ? fmax(err, v) : err + v; }
                /* termination partial of my blocks to the top workgroup */
                st_sc1(Sy.errp + wg, err);
                drain_stores();
                __hip_atomic_fetch_add(Sy.err_cnt, 1u, RLX, AGENT);
            }
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 1: G+H done */

        if (is_top) {
            /* ---- verdicts: was the previous trial accepted?  has the (then current) point converged? ---- */
            if (threadIdx.x == 0) {
                int leave = 0;
                unsigned gv = 0u;
                if (nd > 0u) { if (!poll_ge(Sy.go, nd << 2, Sy, &gv)) leave = 1; else if (gv & 3u) leave = 1; }
                if (!leave && !poll_ge(Sy.err_cnt, e * (unsigned)Gm.G, Sy)) leave = 1;
                *L.flag = leave;
            }
            __syncthreads();
            int leave = *L.flag;
            if (!leave) {
                lds_ptr pe = L.sch;
                for (int w = threadIdx.x; w < Gm.G; w += FW * WAVE) pe[w] = ld_sc1(Sy.errp + w);
                __syncthreads();
                if (threadIdx.x == 0) {
                    double ea = 0.0;
                    for (int w = 0; w < Gm.G; w++) ea = (O.termCondition == 2) ? fmax(ea, pe[w]) : ea + pe[w];
                    if (O.termCondition == 1) ea = sqrt(ea);
                    c->err = ea;
                    if (ea < O.tol) { c->status = 0; __hip_atomic_store(&c->done, 1, RLX, AGENT); *L.flag = 1; }
                }
                __syncthreads();
                leave = *L.flag;
            }
            if (leave) {
                drain_stores();
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(Sy.halt, 1u, RLX, AGENT);
                return;
            }
            __syncthreads();
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 2: verdicts (top) */

        /* ---- backward sweep ---- */
        if (!is_bottom) {
            if (threadIdx.x == 0) *L.flag = poll_ge(Sy.up_cnt + wg, e * nchild, Sy) ? 0 : 1;
            __syncthreads();
            const int leave = *L.flag;
            __syncthreads();
            P_LEAVE_IF(leave);
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 3: children arrived */
        double dotp = 0.0;                                /* per-lane terms of res' * dlam over my blocks */
        {
            double Tc[D];
            for (int t = th - 1; t >= 0; t--) {
                const int nb = U::width(t);
                if (wave < nb) {
                    const int ii = U::first(l0 + t) + s * nb + wave, loc = U::first(t) + wave;
                    const bool is_root = is_top && t == 0;
#ifdef TQ_FINE_STAMPS
                    const bool fs = is_top && t == 1 && wave == 0;
                    if (fs) pstamp(Dt, O, e, 7, 0, 0);
#endif
                    p_load_rows<NX, NU, MD>(L, loc, lane, is_root, Tc);
#ifdef TQ_FINE_STAMPS
                    if (fs) { lds_fence(); pstamp(Dt, O, e, 7, 0, 1); }
#endif
                    if (t < th - 1) sub_children<NX, NU, MD>((lds_cptr)(L.sch + (U::first(t + 1) + MD * wave) * U::SCH), lane, Tc);
                    else if (!is_bottom) sub_children<NX, NU, MD, true>((const double *)(Dt.Sbuf + (size_t)U::kid0(ii) * U::SCH), lane, Tc);
#ifdef TQ_FINE_STAMPS
                    if (fs) { lds_fence(); pstamp(Dt, O, e, 7, 0, 2); }
#endif
                    double myinv = 0.0;
                    factor_rows<NX, NU, MD>(Dt, O, lane, Tc, myinv);
#ifdef TQ_FINE_STAMPS
                    if (fs) pstamp(Dt, O, e, 7, 0, 3);
#endif
                    if (!is_root) {
                        p_store_factor<NX, NU, MD>(L, loc, lane, Tc, myinv);
                        if (t == 0) p_schur<NX, NU, MD, true>(L, loc, lane, L.sch, Dt.Sbuf + (size_t)ii * U::SCH);
                        else p_schur<NX, NU, MD, false>(L, loc, lane, L.sch + loc * U::SCH, nullptr);
#ifdef TQ_FINE_STAMPS
                        if (fs) pstamp(Dt, O, e, 7, 0, 4);
#endif
                    } else {
                        /* root: keep L and 1/diag, then dlam_0 = L^-T (L^-1 res) */
                        if (lane <= D) {
#pragma unroll
                            for (int j = 0; j < D; j++) L.wave[lane * U::LDW + j] = Tc[j];
                        }
                        lds_fence();
                        const int lc = lane < D ? lane : 0;
                        double sv = L.wave[D * U::LDW + lc], Lcol[D];
#pragma unroll
                        for (int k = 0; k < D; k++) Lcol[k] = L.wave[k * U::LDW + lc];
                        double mine = 0.0;
#pragma unroll
                        for (int k = D - 1; k >= 0; k--) {
                            const double zk = rdlane(sv * myinv, k);
                            if (lane == k) mine = zk;
                            if (lane < k) sv = fma(-Lcol[k], zk, sv);
                        }
                        if (lane < D) { st_sc1(Dt.dlam + U::bo(0) + lane, mine); L.dl[lane] = mine; dotp = L.res[lane] * mine; }
                        lds_fence();
                    }
                }
                lds_barrier();
#ifdef TQ_FINE_STAMPS
                if (is_top && t == 1) pstamp(Dt, O, e, 7, 0, 5);
#endif
                pstamp(Dt, O, e, tier, s, sl++);                          /* 4.. : one per backward level */
            }
        }
        if (!is_top) {
            /* publish my subtree root's Schur record (written by wave 0 with sc1 stores) */
            drain_stores();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(Sy.up_cnt + parent_wg, 1u, RLX, AGENT);
            /* ---- wait for the parent's forward sweep ---- */
            if (threadIdx.x == 0) *L.flag = poll_ge(Sy.down + parent_wg, e, Sy) ? 0 : 1;
            __syncthreads();
            const int leave = *L.flag;
            __syncthreads();
            P_LEAVE_IF(leave);
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* parent forward arrived */

        /* ---- forward sweep ---- */
        for (int t = (is_top ? 1 : 0); t < th; t++) {
            const int nb = U::width(t);
            if (wave < nb) {
                const int ii = U::first(l0 + t) + s * nb + wave, loc = U::first(t) + wave;
                if (t == 0) dotp += p_forward<NX, NU, MD>(Dt, L, ii, loc, lane, (lds_cptr)L.dl, Dt.dlam + NX * ii);
                else dotp += p_forward<NX, NU, MD>(Dt, L, ii, loc, lane, (lds_cptr)(L.dl + (U::first(t - 1) + wave / MD) * D + (wave % MD) * NX), nullptr);
            }
            lds_barrier();
        }
        drain_stores();                                   /* dlam of my blocks (sc1) has left the wave */
        dotp = wsum(dotp);
        if (lane == 0) L.wave[1] = dotp;
        __syncthreads();
        if (threadIdx.x == 0 && !is_bottom) __hip_atomic_store(Sy.down + wg, e, RLX, AGENT);
        pstamp(Dt, O, e, tier, s, sl++);                                  /* forward done + published */

        /* ---- first trial (tau = 1) on the nodes this workgroup owns ---- */
        const double fsum = p_stage_owned<NX, NU, MD>(Dt, T, L, l0, l1, s, is_bottom, wave, lane, 1.0, lamc, lamn, false);
        if (lane == 0) L.wave[2] = fsum;
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0 && !is_top) __hip_atomic_fetch_add(Sy.st_cnt + parent_wg, 1u, RLX, AGENT);
        ns += 1u;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* stage done */
        double f = 0.0, d = 0.0;
        for (int w = 0; w < FW; w++) { f += L.wave0[w * U::WAVE_LDS + 2]; d += L.wave0[w * U::WAVE_LDS + 1]; }
        p_reduce_and_decide<NX, NU, MD>(c, Dt, O, Gm, Sy, L, wg, f, d, nd, false);
        nd += 1u;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* reduction handed in */
        /* go straight on at the trial point: the top workgroup checks that it was accepted */
        cur ^= 1;
    }
#undef P_LEAVE_IF
}
