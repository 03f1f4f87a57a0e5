/*
 * tdunes_persist.hpp -- the whole dual-Newton loop as ONE persistent launch (uniform complete trees).
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
 *   - tiers hand over through agent-scope flags (MI355X guide, Guideline 16, recipe R1: payload with
 *     sc1 stores, every storing wave drains vmcnt, workgroup barrier, ONE relaxed agent atomic;
 *     the consumer polls relaxed with s_sleep and reads the payload with sc1 loads only, so no
 *     acquire fence is needed): children -> parent after the backward sweep (arrival counter),
 *     parent -> children after the forward sweep (epoch word);
 *   - the trial stage sweep runs four nodes per wave (16 lanes per node) for the nodes a workgroup
 *     owns; the line-search decision is taken by the LAST workgroup to arrive at a ticket counter,
 *     which sums the per-workgroup {fval, dot} partials in workgroup order and releases everybody
 *     through a `go` word;
 *   - every spin is bounded (wall clock); a timeout ends the launch with status UNKNOWN_ERROR.
 * Extra line-search trials (rare) end the launch: the host runs them with the ordinary trial kernels
 * and relaunches; nothing but global memory carries state across launches.
 */
#pragma once

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, RLX, AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* inter-workgroup words of one persistent solve (zeroed by the host before every launch sequence) */
struct PSync {
    unsigned *up_cnt;       /* [G] arrivals of child subtrees at their parent workgroup (monotonic) */
    unsigned *down;         /* [G] (epoch << 1) | stop, published by a workgroup after its forward sweep */
    unsigned *arrive;       /* ticket counter of the line-search decision (monotonic)                */
    unsigned *go;           /* (epoch << 2) | code : 0 continue, 1 stop (done), 2 stop (more trials) */
    double *parts;          /* [G][2] per-workgroup {fval, dot} partials                             */
    double *errp;           /* [nblocks] termination partial handed up with the Schur record          */
    unsigned *timeout;      /* set when a bounded spin gave up                                       */
};

/* bounded poll by ONE lane: returns the value read, or sets *timed_out */
__device__ __forceinline__ unsigned poll_ge(const unsigned *w, unsigned target, unsigned *tmo) {
    const unsigned long long t0 = wall_clock64();
    unsigned v;
    for (;;) {
        v = __hip_atomic_load(w, RLX, AGENT);
        if (v >= target) break;
        if (__hip_atomic_load(tmo, RLX, AGENT)) break;
        if (wall_clock64() - t0 > 50000000ull) { __hip_atomic_store(tmo, 1u, RLX, AGENT); break; }   /* 0.5 s at 100 MHz */
        __builtin_amdgcn_s_sleep(2);
    }
    return v;
}

template <int NX, int NU, int MD>
struct PLds {
    using U = Uni<NX, NU, MD>;
    static constexpr int D = U::D, NBT = U::NBT;
    static constexpr int DOUBLES = NBT * (D * D + NX * D + 4 * D) + NBT * U::SCH + NBT * D + FW * U::WAVE_LDS + 32;
    lds_ptr W, Ut, res, y, inv, dl, sch, wave0, wave;
    lds_iptr flag;
    __device__ PLds(double *base, int wave_id) {
        W = to_lds(base); Ut = W + NBT * D * D; res = Ut + NBT * NX * D; y = res + NBT * D; inv = y + NBT * D;
        dl = inv + NBT * D; sch = dl + NBT * D; wave0 = sch + NBT * U::SCH; wave = wave0 + wave_id * U::WAVE_LDS;
        flag = (lds_iptr)(wave0 + FW * U::WAVE_LDS);
    }
};

/* G + H of block p into LDS slot `loc`; node data through sc1 loads (written by other workgroups'
 * stage sweeps in the previous iteration); returns the wave's termination partial */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_gh(const Data &Dt, PLds<NX, NU, MD> &L, int p, int loc, int lane, int termCondition) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const int row = lane & 15, g = lane >> 4;
    const int cidx = row / NX, r = row - cidx * NX;
    const int k = U::kid0(p) + cidx;
    const bool live = row < D;
    const double *A = Dt.A + (size_t)(k - 1) * NX * NX + r;
    const double *B = Dt.B + (size_t)(k - 1) * NX * NU + r;
    const int bo = U::bo(p);
    double a[U::KS], pc[U::KS], z[U::KS];
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        a[s] = 0.0; pc[s] = 0.0; z[s] = 0.0;
        if (live && cc < NZ) {
            if (cc < NX) { a[s] = A[(size_t)cc * NX]; pc[s] = ld_sc1(Dt.QinvCal + NX * p + cc); z[s] = ld_sc1(Dt.x + NX * p + cc); }
            else { a[s] = B[(size_t)(cc - NX) * NX]; pc[s] = ld_sc1(Dt.RinvCal + NU * p + cc - NX); z[s] = ld_sc1(Dt.u + NU * p + cc - NX); }
        }
    }
    double xk = 0.0, bk = 0.0, qk = 0.0;
    if (live && g == 0) { xk = ld_sc1(Dt.x + bo + row); bk = Dt.b[bo + row]; }
    if (live) qk = ld_sc1(Dt.QinvCal + bo + row);
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
    lds_ptr Ut = L.Ut + loc * NX * D;
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const double ap = a[s] * pc[s];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], ap, acc, 0, 0, 0);
        part = fma(a[s], z[s], part);
        if (live && cc < NX) Ut[cc + row * NX] = -1.0 * ap;
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    double e = 0.0;
    if (live && g == 0) {
        const double rv = fma(-1.0, xk, bk) + part;
        L.res[loc * D + row] = rv;
        e = (termCondition == 2) ? fabs(rv) : rv * rv;
    }
    lds_ptr W = L.W + loc * D * D;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = g + 4 * rr;
        if (live && i < D) {
            double w = acc[rr];
            if (i == row) w += qk;
            W[i + row * D] = w;
        }
    }
    return (termCondition == 2) ? wave_max(e) : wave_sum(e);
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

/* children records in global memory (tier boundary): sc1 loads */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_sub_children_global(const double *sch, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
#pragma unroll
    for (int c = 0; c < MD; c++) {
        const double *S = sch + c * U::SCH, *v = S + NX * NX;
        const int r = lane - c * NX;
        if (lane < D && r >= 0 && r < NX) {
#pragma unroll
            for (int j = 0; j < NX; j++) T[c * NX + j] -= ld_sc1(S + r + j * NX);
        }
        if (lane == D) {
#pragma unroll
            for (int j = 0; j < NX; j++) T[c * NX + j] -= ld_sc1(v + j);
        }
    }
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
    lds_cptr CUt = L.Ut + loc * NX * D, y = L.y + loc * D;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < D / 4; st++) {
        const int kk = g + 4 * st;
        const double m = (i < NX) ? CUt[i + kk * NX] : (i == NX ? y[kk] : 0.0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(i < NX ? m : 0.0, m, acc, 0, 0, 0);
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
    return wave_sum(pd);
}

/* 16 lanes per node: segmented reductions inside a 16-lane row */
__device__ __forceinline__ double row16_sum(double v) {
    v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
    return v;
}

/* stage QP of node k at the trial point lam_cur + step*dlam, by ONE 16-lane group (lanes t of the
 * group: t < NX state entries, NX <= t < NX+NU input entries); all node-level global traffic is sc1.
 * Returns the node's dual-function term (valid in every lane of the group). */
template <int NX, int NU, int MD>
__device__ __forceinline__ double p_stage16(const Data &Dt, int k, int Np, int t, lds_ptr gl /* group scratch: D + NX */,
                                            double step, const double *lamc, double *lamn, bool active) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    static_assert(NX + NU <= 16 && D <= 16, "16-lane stage needs nx+nu <= 16 and d <= 16");
    const bool parent = active && k < Np;
    const int nuk = parent ? NU : 0;
    const int xo = NX * k, uo = NU * k, ko = U::bo(k);
    const bool isx = t < NX, live = active && t < NX + nuk;
    const int j = isx ? t : t - NX;
    double p_c = 0.0;
    if (parent && t < D) { const double v = fma(step, ld_sc1(Dt.dlam + ko + t), ld_sc1(lamc + ko + t)); gl[t] = v; p_c = Dt.b[ko + t] * v; }
    if (active && isx) {
        double v = 0.0;
        if (k > 0) { v = fma(step, ld_sc1(Dt.dlam + xo + t), ld_sc1(lamc + xo + t)); st_sc1(lamn + xo + t, v); }
        gl[D + t] = v;
    }
    double col[MD][NX];
    if (parent && live) {
#pragma unroll
        for (int cc = 0; cc < MD; cc++) {
            const int kid = U::kid0(k) + cc;
            const double *cp = isx ? Dt.A + (size_t)(kid - 1) * NX * NX + (size_t)j * NX
                                   : Dt.B + (size_t)(kid - 1) * NX * NU + (size_t)j * NX;
#pragma unroll
            for (int i = 0; i < NX; i++) col[cc][i] = cp[i];
        }
    }
    double lin = 0.0, winv = 0.0, wd = 0.0, lob = 0.0, hib = 0.0;
    if (live) {
        if (isx) { lin = Dt.q[xo + j]; winv = Dt.Qinv[xo + j]; wd = Dt.Qd[xo + j]; lob = Dt.xmin[xo + j]; hib = Dt.xmax[xo + j]; }
        else { lin = Dt.r[uo + j]; winv = Dt.Rinv[uo + j]; wd = Dt.Rd[uo + j]; lob = Dt.umin[uo + j]; hib = Dt.umax[uo + j]; }
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

/* diagnostic stamps of the persistent kernel: first workgroup of every tier, thread 0, last iteration */
__device__ __forceinline__ void pstamp(const Data &Dt, const Opts &O, int tier, int s, int slot) {
    if (O.stamps && threadIdx.x == 0 && s == 0 && slot < 32 && tier < 8) {
        Dt.stamps[(tier * 32 + slot) * 2 + 0] = clock64();
        Dt.stamps[(tier * 32 + slot) * 2 + 1] = wall_clock64();
    }
}

template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE) f_persist(Tree T, Data Dt, Opts O, PGeom Gm, PSync Sy) {
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
    int iter = __hip_atomic_load(&c->iter, RLX, AGENT);
    if (__hip_atomic_load(&c->done, RLX, AGENT) || __hip_atomic_load(&c->ls_pending, RLX, AGENT)) return;

    for (;;) {
        const unsigned e = (unsigned)iter + 1u;                        /* epoch of this iteration */
        const int cur = __hip_atomic_load(&c->cur, RLX, AGENT);
        const double *lamc = cur ? Dt.lam1 : Dt.lam0;
        double *lamn = cur ? Dt.lam0 : Dt.lam1;

        int sl = 0;
        pstamp(Dt, O, tier, s, sl++);                                  /* 0: iteration start */
        /* ---- G + H for my blocks (heap order inside the subtree), round-robin over the waves ---- */
        double err = 0.0;
        {
            int cnt = 0;
            for (int t = 0; t < th; t++) {
                const int nb = U::width(t), f0 = U::first(l0 + t) + s * nb;
                for (int b = 0; b < nb; b++, cnt++)
                    if ((cnt & (FW - 1)) == wave) {
                        const double v = p_gh<NX, NU, MD>(Dt, L, f0 + b, U::first(t) + b, lane, O.termCondition);
                        err = (O.termCondition == 2) ? fmax(err, v) : err + v;
                    }
            }
            if (lane == 0) L.wave[0] = err;
            __syncthreads();
            err = 0.0;
            for (int w = 0; w < FW; w++) { const double v = L.wave0[w * U::WAVE_LDS]; err = (O.termCondition == 2) ? fmax(err, v) : err + v; }
            __syncthreads();
        }

        pstamp(Dt, O, tier, s, sl++);                                  /* 1: G+H done */
        /* ---- backward sweep ---- */
        if (!is_bottom) {
            if (threadIdx.x == 0) poll_ge(Sy.up_cnt + wg, e * nchild, Sy.timeout);
            __syncthreads();
            /* fold the children's termination partials (handed up next to their Schur records) */
            const int nbb = U::width(th - 1), fb = U::first(l1 - 1) + s * nbb;
            for (int q = 0; q < nbb * MD; q++) {
                const double v = ld_sc1(Sy.errp + U::kid0(fb) + q);
                err = (O.termCondition == 2) ? fmax(err, v) : err + v;
            }
        }
        pstamp(Dt, O, tier, s, sl++);                                  /* 2: children arrived */
        {
            double Tc[D];
            for (int t = th - 1; t >= 0; t--) {
                const int nb = U::width(t);
                if (wave < nb) {
                    const int ii = U::first(l0 + t) + s * nb + wave, loc = U::first(t) + wave;
                    const bool is_root = is_top && t == 0;
#ifdef TQ_FINE_STAMPS
                    const bool fs = is_top && t == 1 && wave == 0;
                    if (fs) pstamp(Dt, O, 7, 0, 0);
#endif
                    p_load_rows<NX, NU, MD>(L, loc, lane, is_root, Tc);
#ifdef TQ_FINE_STAMPS
                    if (fs) { lds_fence(); pstamp(Dt, O, 7, 0, 1); }
#endif
                    if (t < th - 1) sub_children<NX, NU, MD>((lds_cptr)(L.sch + (U::first(t + 1) + MD * wave) * U::SCH), lane, Tc);
                    else if (!is_bottom) p_sub_children_global<NX, NU, MD>(Dt.Sbuf + (size_t)U::kid0(ii) * U::SCH, lane, Tc);
#ifdef TQ_FINE_STAMPS
                    if (fs) { lds_fence(); pstamp(Dt, O, 7, 0, 2); }
#endif
                    double myinv = 0.0;
                    factor_rows<NX, NU, MD>(Dt, O, lane, Tc, myinv);
#ifdef TQ_FINE_STAMPS
                    if (fs) pstamp(Dt, O, 7, 0, 3);
#endif
                    if (!is_root) {
                        p_store_factor<NX, NU, MD>(L, loc, lane, Tc, myinv);
                        if (t == 0) p_schur<NX, NU, MD, true>(L, loc, lane, L.sch, Dt.Sbuf + (size_t)ii * U::SCH);
                        else p_schur<NX, NU, MD, false>(L, loc, lane, L.sch + loc * U::SCH, nullptr);
#ifdef TQ_FINE_STAMPS
                        if (fs) pstamp(Dt, O, 7, 0, 4);
#endif
                    } else {
                        /* root: keep L and 1/diag, then dlam_0 = L^-T (L^-1 res) */
                        if (lane <= D) {
#pragma unroll
                            for (int j = 0; j < D; j++) L.wave[lane * U::LDW + j] = Tc[j];
                        }
                        lds_fence();
                        double sv = 0.0, Lcol[D];
                        if (lane < D) {
                            sv = L.wave[D * U::LDW + lane];
#pragma unroll
                            for (int k = 0; k < D; k++) Lcol[k] = L.wave[k * U::LDW + lane];
                        } else {
#pragma unroll
                            for (int k = 0; k < D; k++) Lcol[k] = 0.0;
                        }
                        double mine = 0.0;
#pragma unroll
                        for (int k = D - 1; k >= 0; k--) {
                            const double zk = rdlane(sv * myinv, k);
                            if (lane == k) mine = zk;
                            if (lane < k) sv = fma(-Lcol[k], zk, sv);
                        }
                        double pd = 0.0;
                        if (lane < D) { st_sc1(Dt.dlam + U::bo(0) + lane, mine); L.dl[lane] = mine; pd = L.res[lane] * mine; }
                        pd = wave_sum(pd);
                        if (lane == 0) L.wave0[FW * U::WAVE_LDS + 8] = pd;          /* dot partial of the root block */
                        lds_fence();
                    }
                }
                lds_barrier();
                pstamp(Dt, O, tier, s, sl++);                          /* 3.. : one per backward level */
            }
        }
        int stop = 0;
        if (!is_top) {
            /* publish my subtree root's Schur record (written by wave 0 with sc1 stores) + norm partial */
            if (threadIdx.x == 0) st_sc1(Sy.errp + root_blk, err);
            drain_stores();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(Sy.up_cnt + parent_wg, 1u, RLX, AGENT);
            /* ---- wait for the parent's forward sweep ---- */
            if (threadIdx.x == 0) { const unsigned v = poll_ge(Sy.down + parent_wg, e << 1, Sy.timeout); *L.flag = (int)(v & 1u) | (int)__hip_atomic_load(Sy.timeout, RLX, AGENT); }
            __syncthreads();
            stop = *L.flag;
        } else {
            /* termination test at the top (all partials have arrived with the Schur records) */
            if (O.termCondition == 1) err = sqrt(err);
            stop = err < O.tol;
            if (threadIdx.x == 0) {
                c->err = err;
                if (stop) { c->status = 0; __hip_atomic_store(&c->done, 1, RLX, AGENT); }
            }
            if (__hip_atomic_load(Sy.timeout, RLX, AGENT)) stop = 1;
        }
        if (stop) {
            drain_stores();
            __syncthreads();
            if (threadIdx.x == 0 && !is_bottom) __hip_atomic_store(Sy.down + wg, (e << 1) | 1u, RLX, AGENT);
            return;
        }

        pstamp(Dt, O, tier, s, sl++);                                  /* parent forward arrived / top decided */
        /* ---- forward sweep ---- */
        double dotp = 0.0;                                /* wave-local sum of res' * dlam over my blocks */
        if (is_top && wave == 0) dotp = L.wave0[FW * U::WAVE_LDS + 8];
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
        if (lane == 0) L.wave[1] = dotp;
        __syncthreads();
        if (threadIdx.x == 0 && !is_bottom) __hip_atomic_store(Sy.down + wg, e << 1, RLX, AGENT);

        pstamp(Dt, O, tier, s, sl++);                                  /* forward done + published */
        /* ---- first trial (tau = 1): the nodes this workgroup owns, four nodes per wave ---- */
        double fsum = 0.0;
        {
            /* owned nodes: the owner nodes of my blocks (heap order), then (bottom tier) the leaves below */
            const int nown = U::first(th) + (is_bottom ? U::width(th) : 0);
            const int grp = lane >> 4, t16 = lane & 15;
            lds_ptr gl = L.wave + 8 + grp * (D + NX + 2);
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
                const double f = p_stage16<NX, NU, MD>(Dt, k, T.Np, t16, gl, 1.0, lamc, lamn, active);
                if (t16 == 0) fsum += f;
            }
            /* wave partial: lanes 0,16,32,48 hold the groups' sums */
            fsum = wave_sum((lane & 15) == 0 ? fsum : 0.0);
        }
        if (lane == 0) L.wave[2] = fsum;
        drain_stores();
        __syncthreads();
        pstamp(Dt, O, tier, s, sl++);                                  /* stage done */
        if (threadIdx.x == 0) {
            double f = 0.0, d = 0.0;
            for (int w = 0; w < FW; w++) { f += L.wave0[w * U::WAVE_LDS + 2]; d += L.wave0[w * U::WAVE_LDS + 1]; }
            st_sc1(Sy.parts + 2 * wg, f);
            st_sc1(Sy.parts + 2 * wg + 1, d);
            drain_stores();
            const unsigned ticket = __hip_atomic_fetch_add(Sy.arrive, 1u, RLX, AGENT);
            *L.flag = (ticket == e * (unsigned)Gm.G - 1u);
        }
        __syncthreads();
        if (*L.flag) {
            /* last workgroup to arrive decides for everybody: all threads fetch the per-workgroup
             * partials in parallel, thread 0 sums them in workgroup order */
            lds_ptr pf = L.W, pd = L.W + Gm.G;                  /* the block storage is free at this point */
            for (int w = threadIdx.x; w < Gm.G; w += FW * WAVE) { pf[w] = ld_sc1(Sy.parts + 2 * w); pd[w] = ld_sc1(Sy.parts + 2 * w + 1); }
            __syncthreads();
            if (threadIdx.x == 0) {
                double fa = 0.0, da = 0.0;
                for (int w = 0; w < Gm.G; w++) { fa += pf[w]; da += pd[w]; }
                c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
                unsigned code;
                if (ls_not_descent(c, -da)) code = 1u;
                else {
                    ls_decide_tail(c, Dt, O, fa);
                    code = c->done ? 1u : (c->ls_pending ? 2u : 0u);
                }
                drain_stores();
                __threadfence();
                __hip_atomic_store(Sy.go, (e << 2) | code, RLX, AGENT);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned v = poll_ge(Sy.go, e << 2, Sy.timeout);
            *L.flag = (int)(v & 3u) | ((int)__hip_atomic_load(Sy.timeout, RLX, AGENT) ? 1 : 0);
        }
        __syncthreads();
        pstamp(Dt, O, tier, s, sl++);                                  /* decision received */
        if (*L.flag) return;
        iter += 1;
        __syncthreads();
    }
}
