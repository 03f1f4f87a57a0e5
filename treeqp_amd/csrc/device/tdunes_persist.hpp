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
 *       arrive          everybody -> top, my {fval, dot} partial of the last stage sweep is in parts[wg],
 *       halt            top -> everybody, the launch is over;
 *     all hand-ins are fire-and-forget (no returning atomics): only consumers ever wait;
 *   - everything a workgroup needs from ITSELF stays in its LDS across iterations: the duals of its
 *     blocks, x / u / QinvCal / RinvCal of the nodes it owns, the step of its blocks.  The bottom
 *     tier (the start of every iteration's critical path) therefore goes from the stage sweep to
 *     G + H of the next iteration without touching global memory for anything but constants;
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
 *     owns; the line-search decision is taken by the top workgroup (which has the most slack), from
 *     the per-workgroup {fval, dot} partials summed in workgroup order;
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
    unsigned *arrive;       /* workgroups that have handed in their {fval, dot} partial (monotonic)   */
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
    static constexpr int SLOTS = NBT + (MD == 2 ? 8 : MD * MD);      /* nodes a workgroup can own: its blocks' owners + (bottom tier) the leaves */
    static constexpr int NODE = 2 * (NX + NU);                       /* x, u, QinvCal, RinvCal of one owned node */
    static constexpr int DOUBLES = NBT * (D * D + NX * D + 4 * D) + NBT * U::SCH + NBT * D + SLOTS * NODE + 2 * NBT * D + 3 * NX + 4 * FW + FW * U::WAVE_LDS + 32;
    /* scratch of the top workgroup's reductions: the Schur record storage, free before the backward sweep */
    static constexpr int RED_CAP = NBT * U::SCH / 2;
    lds_ptr W, Ut, res, y, inv, dl, sch, node, lamb, lamroot, droot, part, wave0, wave;
    lds_iptr flag;
    __device__ PLds(double *base, int wave_id) {
        W = to_lds(base); Ut = W + NBT * D * D; res = Ut + NBT * NX * D; y = res + NBT * D; inv = y + NBT * D;
        dl = inv + NBT * D; sch = dl + NBT * D; node = sch + NBT * U::SCH; lamb = node + SLOTS * NODE;
        lamroot = lamb + 2 * NBT * D; droot = lamroot + 2 * NX; part = droot + NX; wave0 = part + 4 * FW; wave = wave0 + wave_id * U::WAVE_LDS;
        flag = (lds_iptr)(wave0 + FW * U::WAVE_LDS);
    }
    /* part[4 w + i]: wave w's partials -- 0 termination norm, 1 res' * dlam, 2 dual function value */
    /* owned node `q` (heap order inside the tier subtree): x | u | QinvCal | RinvCal */
    __device__ __forceinline__ lds_ptr nx_(int q) const { return node + q * NODE; }
    __device__ __forceinline__ lds_ptr nu_(int q) const { return node + q * NODE + NX; }
    __device__ __forceinline__ lds_ptr nqc(int q) const { return node + q * NODE + NX + NU; }
    __device__ __forceinline__ lds_ptr nrc(int q) const { return node + q * NODE + 2 * NX + NU; }
    /* dual vector of block `loc` (= duals of the owner node's children), double-buffered like lam0 / lam1 */
    __device__ __forceinline__ lds_ptr lamb_(int buf, int loc) const { return lamb + (buf * NBT + loc) * D; }
};

/* G + H of block p into LDS slot `loc`, split into a branch-free load half and a compute half so
 * that a wave with two blocks has both blocks' loads in flight at once.  The owner node's x, u,
 * QinvCal, RinvCal come from the workgroup's LDS node store (its own stage sweep wrote them); the
 * children's x and QinvCal likewise unless the children belong to the tier below (`foreign`): then
 * they were written by the child workgroups' stage sweeps and are read with sc1 loads. */
template <int NX, int NU, int MD>
struct GhRegs {
    double a[Uni<NX, NU, MD>::KS], pc[Uni<NX, NU, MD>::KS], z[Uni<NX, NU, MD>::KS], xk, bk, qk;
};

template <int NX, int NU, int MD>
__device__ __forceinline__ void p_gh_load(const Data &Dt, const PLds<NX, NU, MD> &L, int p, int loc, bool foreign, int lane, GhRegs<NX, NU, MD> &G) {
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
    lds_cptr own = L.nx_(loc);                               /* x | u | QinvCal | RinvCal, NZ apart */
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const bool ok = live && cc < NZ, isx = cc < NX;
        const int cu = (cc < NZ) ? cc - NX : 0;              /* input column (clamped) */
        const int cz = (cc < NZ) ? cc : 0;
        const double *ap = isx ? A + (size_t)cc * NX : B + (size_t)cu * NX;
        const double av = *ap, zv = own[cz], pv = own[NZ + cz];
        G.a[s] = ok ? av : 0.0; G.pc[s] = ok ? pv : 0.0; G.z[s] = ok ? zv : 0.0;
    }
    double xv, qv;
    if (foreign) { xv = ld_sc1(Dt.x + bo + rowc); qv = ld_sc1(Dt.QinvCal + bo + rowc); }
    else { lds_cptr kid = L.nx_(MD * loc + 1 + cidx); xv = kid[r]; qv = kid[NZ + r]; }
    const double bv = Dt.b[bo + rowc];
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
        if (lane == 0) {                                  /* the subtree root's own step: the stage sweep reads it from LDS */
#pragma unroll
            for (int r = 0; r < NX; r++) L.droot[r] = dv[r];
        }
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

/* stage QP of owned node slot q (= node k) at the trial point lam_cur + step*dlam, by ONE 16-lane
 * group (lanes t of the group: t < NX state entries, NX <= t < NX+NU input entries).  Duals and steps
 * come from the workgroup's LDS copies (lamb / lamroot, dl / droot; `cb` = current buffer), constants
 * from global memory; results go to global memory (sc1 stores, nobody waits for them here) AND to the
 * LDS node store / the other dual buffer for this workgroup's next G + H.
 * init: first sweep of a solve -- evaluate at the current duals themselves.
 * Returns the node's dual-function term (valid in every lane of the group). */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_stage16(const Data &Dt, PLds<NX, NU, MD> &L, int q, int k, int Np, int t, lds_ptr gl /* group scratch: D + NX */,
                                            double step, int cb, double *lamn, bool active, bool init) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NBT = U::NBT;
    static_assert(NX + NU <= 16 && D <= 16, "16-lane stage needs nx+nu <= 16 and d <= 16");
    const bool parent = active && k < Np;
    const int nuk = parent ? NU : 0;
    const int xo = NX * k, uo = NU * k, ko = U::bo(k);
    const bool isx = t < NX, live = active && t < NX + nuk;
    const int j = isx ? t : t - NX;
    /* branch-free loads: every lane reads from a valid (clamped) address and masks afterwards */
    const bool pk = parent && t < D, ox = active && isx && k > 0;
    const int qb = pk ? q : 0, tb = pk ? t : 0;                       /* my block's duals / step */
    const double lca = L.lamb_(cb, qb)[tb], dla = L.dl[qb * D + tb], ba = Dt.b[pk ? ko + t : 0];
    const int qp = (ox && q > 0) ? (q - 1) / MD : 0, tp = (ox && q > 0) ? ((q - 1) % MD) * NX + t : 0;   /* my own slice in the parent's block */
    const int tr = ox ? t : 0;
    const double lcb = (q > 0) ? L.lamb_(cb, qp)[tp] : L.lamroot[cb * NX + tr];
    const double dlb = (q > 0) ? L.dl[qp * D + tp] : L.droot[tr];
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
        if (pk) { gl[t] = v; p_c = ba * v; if (!init) L.lamb_(cb ^ 1, q)[t] = v; }
        const double w = ox ? (init ? lcb : fma(step, dlb, lcb)) : 0.0;
        if (ox) { st_sc1(lamn + xo + t, w); if (q == 0 && !init) L.lamroot[(cb ^ 1) * NX + t] = w; }
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
        lds_ptr ns = L.nx_(q);                            /* x | u | QinvCal | RinvCal: entry t, NX+NU apart */
        ns[t] = val; ns[NX + NU + t] = cal;
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

/* heap slot -> node / block index: slot q of the tier subtree s whose block levels are [l0, l1) */
template <int NX, int NU, int MD>
__device__ __forceinline__ int p_slot_node(int q, int l0, int s) {
    using U = Uni<NX, NU, MD>;
    int t = 0;
    while (q >= U::first(t + 1)) t++;
    return U::first(l0 + t) + s * U::width(t) + (q - U::first(t));
}

/* stage sweep over the nodes this workgroup owns (the owner nodes of its blocks in heap order, then --
 * bottom tier -- the leaves below), four nodes per wave; returns the wave's sum of the node terms */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_stage_owned(const Data &Dt, const Tree &T, PLds<NX, NU, MD> &L, int l0, int nown, int s, int wave, int lane,
                                                double step, int cb, double *lamn, bool init) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int grp = lane >> 4, t16 = lane & 15;
    lds_ptr gl = L.wave + 8 + grp * (D + NX + 2);
    double fsum = 0.0;
    for (int base = 0; base < nown; base += FW * 4) {
        const int q = base + wave * 4 + grp;
        const bool active = q < nown;
        const int k = active ? p_slot_node<NX, NU, MD>(q, l0, s) : 0;
        fsum += p_stage16<NX, NU, MD>(Dt, L, active ? q : 0, k, T.Np, t16, gl, step, cb, lamn, active, init);
    }
    return rows_fold<false>(fsum);       /* every lane of a 16-lane group holds its group's sum */
}

/* top workgroup: ordered sums (workgroup order) of the per-workgroup {fval, dot} partials and of the
 * termination partials (sum or maximum), gathered with parallel sc1 loads through the Schur record
 * scratch in chunks; results valid in thread 0 */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_gather3(PLds<NX, NU, MD> &L, const double *parts, const double *errp, int count, bool err_max,
                                          double &fa, double &da, double &ea) {
    constexpr int CAP = PLds<NX, NU, MD>::RED_CAP * 2 / 3;
    fa = 0.0; da = 0.0; ea = 0.0;
    for (int c0 = 0; c0 < count; c0 += CAP) {
        const int n = min(CAP, count - c0);
        for (int w = threadIdx.x; w < n; w += FW * WAVE) {
            L.sch[3 * w] = ld_sc1(parts + 2 * (size_t)(c0 + w));
            L.sch[3 * w + 1] = ld_sc1(parts + 2 * (size_t)(c0 + w) + 1);
            L.sch[3 * w + 2] = ld_sc1(errp + c0 + w);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 0; w < n; w++) { fa += L.sch[3 * w]; da += L.sch[3 * w + 1]; ea = err_max ? fmax(ea, L.sch[3 * w + 2]) : ea + L.sch[3 * w + 2]; }
        }
        __syncthreads();
    }
}

template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_persist(Tree T, Data Dt, Opts O, PGeom Gm, PSync Sy, const double *lam_init, int prologue) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NBT = U::NBT;
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
    const int nbt = U::first(th);                                      /* my blocks */
    const int nown = nbt + (is_bottom ? U::width(th) : 0);             /* nodes I own */
    if (__hip_atomic_load(&c->done, RLX, AGENT) || __hip_atomic_load(&c->ls_pending, RLX, AGENT)) return;
    int cur = __hip_atomic_load(&c->cur, RLX, AGENT);
    unsigned nd = 0u;          /* {fval, dot} reductions handed in so far (by me, hence by everybody who got this far) */
    unsigned ns = 0u;          /* stage sweeps I have completed in this launch */
    bool unposted = false;     /* a finished stage sweep whose results are not published yet */

    /* ---- LDS copies of what this workgroup owns: duals of my blocks and of my root, node store ---- */
    {
        const double *lsrc = prologue ? lam_init : (cur ? Dt.lam1 : Dt.lam0);
        for (int i = threadIdx.x; i < nbt * D; i += FW * WAVE) {
            const int loc = i / D, t = i - loc * D;
            L.lamb_(cur, loc)[t] = lsrc[U::bo(p_slot_node<NX, NU, MD>(loc, l0, s)) + t];
        }
        if (threadIdx.x < NX) L.lamroot[cur * NX + threadIdx.x] = root_blk > 0 ? lsrc[NX * root_blk + threadIdx.x] : 0.0;
        if (!prologue) {
            for (int i = threadIdx.x; i < nown * 16; i += FW * WAVE) {
                const int q = i >> 4, t = i & 15;
                const int k = p_slot_node<NX, NU, MD>(q, l0, s);
                if (t < NX) { L.nx_(q)[t] = Dt.x[NX * k + t]; L.nqc(q)[t] = Dt.QinvCal[NX * k + t]; }
                else if (t < NX + NU && k < T.Np) { L.nu_(q)[t - NX] = Dt.u[NU * k + t - NX]; L.nrc(q)[t - NX] = Dt.RinvCal[NU * k + t - NX]; }
            }
        }
        __syncthreads();
    }

    /* Hand-ins of this workgroup, by ONE lane, fire-and-forget: (stage) a finished stage sweep -- node data
     * of my subtree root for the parent (st_cnt) and my {fval, dot} partial for the top workgroup (arrive);
     * (gh) the termination partial of my blocks (err_cnt).  The caller has made sure that every wave
     * drained the global stores of its stage sweep (drain_stores + workgroup barrier). */
    auto post = [&](bool stage, bool gh) {
        if (stage) {
            double f = 0.0, d = 0.0;
            for (int w = 0; w < FW; w++) { f += L.part[4 * w + 2]; d += L.part[4 * w + 1]; }
            st_sc1(Sy.parts + 2 * wg, f);
            st_sc1(Sy.parts + 2 * wg + 1, d);
        }
        if (gh) {
            double err = 0.0;
            for (int w = 0; w < FW; w++) { const double v = L.part[4 * w]; err = (O.termCondition == 2) ? fmax(err, v) : err + v; }
            st_sc1(Sy.errp + wg, err);
        }
        drain_stores();
        if (stage) {
            if (!is_top) __hip_atomic_fetch_add(Sy.st_cnt + parent_wg, 1u, RLX, AGENT);
            __hip_atomic_fetch_add(Sy.arrive, 1u, RLX, AGENT);
        }
        if (gh) __hip_atomic_fetch_add(Sy.err_cnt, 1u, RLX, AGENT);
    };

    if (prologue) {
        /* ---- first sweep of the solve: stage QPs at lambda0 (copied into the current buffer), fval0 ---- */
        double *lam0 = cur ? Dt.lam1 : Dt.lam0;
        const double fsum = p_stage_owned<NX, NU, MD>(Dt, T, L, l0, nown, s, wave, lane, 0.0, cur, lam0, true);
        if (lane == 0) { L.part[4 * wave + 2] = fsum; L.part[4 * wave + 1] = 0.0; }
        __syncthreads();
        ns = 1u;
        unposted = true;
    }

    for (unsigned e = 1u;; e++) {
        double *lamn = cur ? Dt.lam0 : Dt.lam1;

        int sl = 0;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 0: iteration start */
        /* ---- the nodes my bottom-level blocks read are staged by the child workgroups ---- */
        if (!is_bottom) {
            if (unposted) {
                drain_stores();
                __syncthreads();
                if (threadIdx.x == 0) post(true, false);
                nd += 1u;
                unposted = false;
            }
            if (threadIdx.x == 0) *L.flag = poll_ge(Sy.st_cnt + wg, ns * nchild, Sy) ? 0 : 1;
            __syncthreads();
            const int leave = *L.flag;
            __syncthreads();
            if (leave) return;
        }
        /* ---- G + H for my blocks (heap order inside the subtree), two blocks per wave in flight ---- */
        double err = 0.0;
        {
            const int nint = U::first(th - 1);                         /* blocks above my bottom level: children are mine */
            for (int loc0 = wave; loc0 < nbt; loc0 += 2 * FW) {
                const int loc1 = loc0 + FW;
                GhRegs<NX, NU, MD> g0, g1;
                p_gh_load<NX, NU, MD>(Dt, L, p_slot_node<NX, NU, MD>(loc0, l0, s), loc0, !is_bottom && loc0 >= nint, lane, g0);
                if (loc1 < nbt) p_gh_load<NX, NU, MD>(Dt, L, p_slot_node<NX, NU, MD>(loc1, l0, s), loc1, !is_bottom && loc1 >= nint, lane, g1);
                double v = p_gh_compute<NX, NU, MD>(L, loc0, lane, g0, O.termCondition);
                err = (O.termCondition == 2) ? fmax(err, v) : err + v;
                if (loc1 < nbt) {
                    v = p_gh_compute<NX, NU, MD>(L, loc1, lane, g1, O.termCondition);
                    err = (O.termCondition == 2) ? fmax(err, v) : err + v;
                }
            }
            if (lane == 0) L.part[4 * wave] = err;
            drain_stores();                               /* my stage sweep's global stores (long gone by now) */
            __syncthreads();
        }
        /* Bottom tier (the head of the critical path): the hand-ins wait for a wave that has no block in
         * the backward sweep (below); every other tier has slack and hands in right away. */
        bool post_gh = true, post_st = false;
        if (is_bottom && unposted) { post_st = true; nd += 1u; unposted = false; }
        if (!is_bottom || is_top || th == 1) {
            if (threadIdx.x == 0) post(post_st, true);
            post_gh = false; post_st = false;
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 1: G+H done */

        if (is_top) {
            /* ---- verdicts: the outstanding {fval, dot} reduction (fval0 of the first sweep, or the first
             * trial of the previous iteration), then the termination test of the (then current) point ---- */
            if (threadIdx.x == 0) {
                int lv = 0;
                if (nd > 0u && !poll_ge(Sy.arrive, nd * (unsigned)Gm.G, Sy)) lv = 1;
                if (!lv && !poll_ge(Sy.err_cnt, e * (unsigned)Gm.G, Sy)) lv = 1;
                *L.flag = lv;
            }
            __syncthreads();
            int leave = *L.flag;
            __syncthreads();
            if (!leave) {
                double fa, da, ea;
                p_gather3<NX, NU, MD>(L, Sy.parts, Sy.errp, Gm.G, O.termCondition == 2, fa, da, ea);
                if (threadIdx.x == 0) {
                    int code = 0;
                    if (nd > 0u) {
                        if (prologue && nd == 1u) { c->fval0 = fa; c->fval = fa; }
                        else {
                            c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
                            if (ls_not_descent(c, -da)) code = 1;
                            else {
                                ls_decide_tail(c, Dt, O, fa);
                                code = (c->done || c->ls_pending) ? 1 : 0;      /* finished, or more trials: the host takes over */
                            }
                        }
                    }
                    if (!code) {
                        if (O.termCondition == 1) ea = sqrt(ea);
                        c->err = ea;
                        if (ea < O.tol) { c->status = 0; c->done = 1; code = 1; }
                    }
                    *L.flag = code;
                }
                __syncthreads();
                leave = *L.flag;
                __syncthreads();
            }
            if (leave) {
                drain_stores();
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(Sy.halt, 1u, RLX, AGENT);
                return;
            }
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 2: verdicts (top) */

        /* ---- backward sweep ---- */
        if (!is_bottom) {
            if (threadIdx.x == 0) *L.flag = poll_ge(Sy.up_cnt + wg, e * nchild, Sy) ? 0 : 1;
            __syncthreads();
            const int leave = *L.flag;
            __syncthreads();
            if (leave) return;
        }
        pstamp(Dt, O, e, tier, s, sl++);                                  /* 3: children arrived */
        double dotp = 0.0;                                /* per-lane terms of res' * dlam over my blocks */
        {
            double Tc[D];
            for (int t = th - 1; t >= 0; t--) {
                const int nb = U::width(t);
                /* the launch may be over (converged, line search needs the host): look once per level, the
                 * load is in flight while the level factors */
                unsigned halted = 0u;
#ifdef TQ_LEVEL_HALT
                if (threadIdx.x == 0) halted = __hip_atomic_load(Sy.halt, RLX, AGENT);
#endif
                if (post_gh && (t < th - 1 || nb < FW) && wave == FW - 1 && lane == 0) post(post_st, true);
                if (t < th - 1 || nb < FW) post_gh = false;
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
#ifdef TQ_LEVEL_HALT
                if (threadIdx.x == 0) *L.flag = (int)halted;
#endif
                lds_barrier();
#ifdef TQ_FINE_STAMPS
                if (is_top && t == 1) pstamp(Dt, O, e, 7, 0, 5);
#endif
                pstamp(Dt, O, e, tier, s, sl++);                          /* 4.. : one per backward level */
#ifdef TQ_LEVEL_HALT
                if (*L.flag) return;
#endif
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
            if (leave) return;
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
        if (!is_bottom) {
            drain_stores();                               /* the step of my bottom-level blocks (sc1) has left the wave */
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(Sy.down + wg, e, RLX, AGENT);
        }
        dotp = wsum(dotp);
        if (lane == 0) L.part[4 * wave + 1] = dotp;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* forward done + published */

        /* ---- first trial (tau = 1) on the nodes this workgroup owns; then straight on to the next
         * iteration at the trial point: the top workgroup checks that it was accepted ---- */
        const double fsum = p_stage_owned<NX, NU, MD>(Dt, T, L, l0, nown, s, wave, lane, 1.0, cur, lamn, false);
        if (lane == 0) L.part[4 * wave + 2] = fsum;
        __syncthreads();
        ns += 1u;
        unposted = true;
        cur ^= 1;
        pstamp(Dt, O, e, tier, s, sl++);                                  /* stage done */
    }
}
