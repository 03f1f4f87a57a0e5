/*
 * tdunes_persist.hpp -- the whole dual-Newton solve as ONE persistent launch (uniform complete trees).
 *
 * Included by tdunes_device.hip after tdunes_fast.hpp (same block-level device functions).
 *
 * The tiered path (tdunes_fast.hpp) spends ~1/3 of an iteration in kernel boundaries: a gap of
 * ~2.3 us plus ~2.5 us of cold global loads per boundary, 7 boundaries per iteration.  Here every
 * tier subtree keeps its workgroup for the whole solve:
 *   - grid = one 4-wave workgroup per tier subtree (C2: 64 + 8 + 1 = 73), all co-resident;
 *   - the STATE of the solve lives in LDS for the whole launch: a block's W / L, Ut / CholUt,
 *     residual, backward solution, reciprocal diagonal and step; the duals of the workgroup's blocks
 *     (double-buffered like lam0 / lam1); x, u, their unclipped values, the modified gradients and the
 *     clipped inverse Hessians of the nodes the workgroup owns.  It is loaded once (first sweep of a
 *     solve: from lambda0; relaunch: from global memory) and written back to global memory once,
 *     when the workgroup leaves.  An iteration reads only constants from global memory (packed per
 *     edge / per node by k_pack_persist: [A | B], {q, 1/Q, Q, lower, upper});
 *   - what crosses workgroups travels as TAGGED WORDS: a double is written as two 64-bit relaxed
 *     agent-scope atomic stores, each (tag << 32) | 32-bit half, tag = launch number and sequence
 *     number of the hand-over.  The consumer polls the payload itself until every word carries the
 *     tag it expects.  No flag, no counter, no store drain, no returning atomic: a hand-over costs
 *     one store latency plus one load latency, and producers never wait.  (64-bit atomicity is all
 *     this relies on: a torn double shows a stale tag in one half and is simply read again.)
 *       sch    child -> parent     Schur record of the subtree root           (tag: iteration)
 *       dlt    parent -> child     step of the subtree root's own duals       (tag: iteration)
 *       ndt    child -> parent     x, QinvCal of the subtree root node        (tag: stage sweep)
 *       parts  everybody -> top    {fval, dot} partial of a stage sweep       (tag: stage sweep)
 *       errs   everybody -> top    termination partial of G + H               (tag: iteration)
 *       cmd    top -> everybody    the trial a pass was built on is rejected: {tau - tauPrev, tau} of the next trial (tag: pass)
 *       bparts everybody -> top    dual-function partials of a batch of further trials   (tag: batch)
 *       vrd    top -> everybody    length of the accepted backtracking chain, 0 = next batch (tag: batch)
 *       halt   top -> everybody    the launch is over (word == launch number)
 *     Buffers are never reset: a stale word carries another launch's number;
 *   - the FIRST sweep of a solve (stage QPs at lambda0, fval0) is the launch's prologue, which also
 *     initialises the control block: a solve is ONE kernel launch and nothing else;
 *   - an iteration does not wait for the line-search decision of the previous one: the first trial
 *     (tau = 1) is accepted almost always, so every workgroup goes straight on to G + H and the
 *     backward sweep of the next iteration at the trial point.  Only the top workgroup (which has
 *     the most slack) takes the decision, before anything irreversible (termination verdict, forward
 *     sweep, next trial); if the trial was NOT accepted it halts the launch and the speculative work
 *     is simply dropped (it only touched per-iteration LDS data and hand-over buffers);
 *   - termination is decided from the errs[] partials as soon as every workgroup has done G + H,
 *     i.e. long before the backward sweep of a converged point would have reached the top;
 *   - the trial stage sweep runs four nodes per wave (16 lanes per node) for the nodes a workgroup owns;
 *   - every spin is bounded (wall clock); a timeout ends the launch with status UNKNOWN_ERROR.
 *   - extra line-search trials stay inside the launch.  When the top workgroup rejects the first trial it posts
 *     `cmd`; every workgroup meets it in the poll it is waiting in (Schur records of its children, step of its
 *     parent), drops the pass and evaluates the next K = 4 (then 8) points of the backtracking chain WITHOUT
 *     storing anything (the duals of trial n are the reference's sequence of axpys replayed in registers);
 *     one reduction and one verdict per batch instead of per trial, then one stored sweep at the accepted point.
 *     (58 iterations / 1329 trials of the x0-eliminated spring-mass example: 19.8 ms with host-run trials,
 *     18.1 ms with one in-kernel round trip per trial, 6.1 ms with batches; CPU oracle 7.0 ms.)
 * Nothing but global memory carries state across launches (relaunch without prologue: tag space exhausted).
 */
#pragma once

/* what only the start and the end of a launch touch; lives in device memory so that the kernel
 * holds ONE pointer during the loop (scalar register pressure) */
/* result block in pinned host memory: the top workgroup writes it the moment the launch is decided, the
 * host polls `seq` -- it does not wait for the other workgroups to leave nor for a stream synchronisation */
#define HOSTRES_WORDS 28            /* 24 halves of the 12 quadwords of Ctrl, 2 + 2 of t_start, t_end */
struct HostRes {
    Ctrl c;
    unsigned long long t_start, t_end;      /* 100 MHz wall clock of the top workgroup: launch start, verdict */
    unsigned seq;                           /* == PSync.seq of the launch when the block is complete */
    unsigned pad;
    /* The persistent launch posts the block as TAGGED WORDS, the way workgroups talk to each other: word i = (seq << 32) | 32-bit piece i of
     * {c, t_start, t_end}; the host polls until every word carries the launch's seq and unpacks them into the fields above
     * (wait_result_block).  One store instruction and its flight instead of: the stores, the wait for their acknowledgements from host
     * memory (~1.5 us on the verdict's way) and the sequence word behind them. */
    unsigned long long tg[HOSTRES_WORDS];
};

struct PDump {
    double *x, *u, *xUnc, *uUnc, *xUncS, *uUncS, *qmod, *rmod, *QinvCal, *RinvCal, *lam0, *lam1, *dlam;
    const double *lam_init;
    unsigned long long *stamps;
    int *ls_log;
    int ls_log_cap;
    HostRes *hres;
};

/* what the loop touches in global memory besides the hand-over buffers */
struct PConst {
    const double *AB;       /* [edges][NX * (NX+NU)]  column-major [A | B] of edge k-1 (k = child node)           */
    const double *b;        /* [sum_nx]                                                                            */
    const double *cst;      /* [nodes][16][5]  {linear term, 1/weight, weight, lower, upper} of entry t of node k  */
    Ctrl *ctrl;
    const PDump *dump;
    const double *lam0_src; /* starting duals of a fresh solve (= dump->lam_init): here so that the state load of the prologue needs no trip through `dump` first */
    int Np;                 /* number of parent nodes                                                              */
    /* multistage trees (setup_multistage_tree(md, Nr, Nh), Nr < Nh: branching for Nr stages, then one child per
     * node): the first nB nodes (levels < Nr) are numbered like a complete md-ary tree, below them every level
     * has S = md^Nr nodes and the child of node k is k + S.  Uniform complete trees: S = 0. */
    int S, nB, Nr;
};

/* first child of node k in the GLOBAL numbering, for a tier instantiated with branching MD (MD == 1: chain part) */
template <int MD>
__device__ __forceinline__ int kid0g(int k, const PConst &C) { return MD == 1 ? k + C.S : MD * k + 1; }

/* hand-over buffers of the persistent path (zeroed once at creation, never reset) */
struct PSync {
    u64 *sch;               /* [nodes][SCH][2]   Schur record of a tier subtree root                   */
    u64 *dlt;               /* [sum_nx][2]       step, indexed like dlam (only tier roots' slices used) */
    u64 *ndt;               /* [nodes][2 NX][2]  x then QinvCal of a tier subtree root node            */
    u64 *parts;             /* [G][2][2]         per-workgroup {fval, dot}                             */
    u64 *errs;              /* [G][2]            per-workgroup termination partial                     */
    u64 *cmd;               /* [2][2]            top -> everybody: the trial of pass `tag` was rejected; {tau - tauPrev, tau} of the next one */
    u64 *vrd;               /* [2]               top -> everybody: verdict on a batch of trials (tag: batch): accepted chain length, 0 = none */
    u64 *bparts;            /* [G][8][2]         per-workgroup dual-function partials of a batch of trials (tag: batch) */
    u64 *sgt;               /* [nodes][2]        active-set signature (x part) of a tier subtree root node, child -> parent (tag: stage sweep) */
    u64 *rfl;               /* [nodes][2]        1.0: the child workgroup rooted here keeps its factors this pass, child -> parent (tag: pass) */
    unsigned *halt;         /* == seq: the top workgroup ended this launch                            */
    unsigned *timeout;      /* sticky: a bounded spin gave up                                         */
    unsigned seq;           /* launch number << 16 (low 16 bits of the number are never 0)            */
    unsigned trip;          /* tag of the pass the workgroup is in (kernel-local copy only)           */
    int nap;                /* > 0: a launch of several hundred workgroups -- the bottom tier's wait for its parent's step naps long (p_forward_tier) */
    /* ONE tree over several devices (tqgpu_pshard_*): the workgroups of the launch are dealt over `npeer` launches, one per device, each
     * with a slab of its own of this layout.  Every hand-over word is then written to EVERY slab (system-scope stores into peer-mapped
     * memory: a posted write per peer), every poll stays local -- the protocol is the single-device one, its words merely travel
     * further.  npeer <= 1: the slab at `base` only. */
    u64 *base;              /* this launch's slab                                                      */
    u64 *const *peers;      /* [npeer] in device memory: slabs of all ranks (peers[rank] == base).  (A table in memory, not an array member: a
                               dynamically indexed member keeps the whole struct in scratch memory -- 18 us per C2 solve.) */
    int npeer;
    u64 *verdict;           /* [16] in the slab: the control block as the top workgroup left it + seq, for the ranks that do not run the top workgroup */
    u64 *anc;               /* [blocks][D][NX + 1][2]  forward records [z0 | M] of the blocks of tier 1, for the bottom tier's walk down its path (p_forward_tier; tag: pass) */
    int relay_wg;           /* sharded launch without the top workgroup: the workgroup (global number) that passes the verdict on to THIS rank's host, else -1 */
    int anc_local;          /* sharded launch: tier 1 is dealt over the ranks like tier 0 (a tier-1 workgroup and the bottom-tier workgroups below it share a rank): its forward records stay in this rank's slab */
};
#define SYS __HIP_MEMORY_SCOPE_SYSTEM
/* a tagged double to every slab of a sharded launch */
__device__ __forceinline__ void pst_tag(const PSync &Sy, u64 *p, double v, unsigned tag) {
    st_tag(p, v, tag);
    if (Sy.npeer > 1) {
        const size_t off = (size_t)(p - Sy.base);
        const u64 t = (u64)tag << 32, lo = t | (unsigned)__double2loint(v), hi = t | (unsigned)__double2hiint(v);
        for (int r = 0; r < Sy.npeer; r++) {
            u64 *q = Sy.peers[r] + off;
            if (q != p) { __hip_atomic_store(q, lo, RLX, SYS); __hip_atomic_store(q + 1, hi, RLX, SYS); }
        }
    }
}
/* a 32-bit word (halt, timeout) to every slab */
__device__ __forceinline__ void pst_word(const PSync &Sy, unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, RLX, AGENT);
    if (Sy.npeer > 1) {
        const size_t off = (size_t)(reinterpret_cast<u64 *>(p) - Sy.base);
        const size_t sub = (size_t)(p - reinterpret_cast<unsigned *>(Sy.base + off));
        for (int r = 0; r < Sy.npeer; r++) {
            unsigned *q = reinterpret_cast<unsigned *>(Sy.peers[r] + off) + sub;
            if (q != p) __hip_atomic_store(q, v, RLX, SYS);
        }
    }
}

/* Poll loops read the payload AND the two "launch is over" words in the same round trip (the loads are
 * independent, so they are in flight together); a poll iteration is then one memory latency long and needs
 * no sleep.  `over` = halt word == launch number or timeout word set; false = give up. */
#ifndef TQ_POLL_NAP
#define TQ_POLL_NAP 1
#endif
struct PollGuard {
    unsigned h, tmo, cm;
    __device__ __forceinline__ void load(const PSync &Sy) {
        h = __hip_atomic_load(Sy.halt, RLX, TQ_LD_SCOPE); tmo = __hip_atomic_load(Sy.timeout, RLX, TQ_LD_SCOPE);
        cm = (unsigned)(__hip_atomic_load(Sy.cmd + 1, RLX, TQ_LD_SCOPE) >> 32);
    }
    /* call right after the poll loop: the three guard words are only looked at when the payload is not there yet, so on the
     * usual exit their loads are still in flight as far as the compiler knows; it then waits for them -- with a vmcnt(0)
     * that also waits for every STORE issued since (tagged hand-over stores take ~1 us to complete) -- at some later point
     * where it wants their registers back.  Naming them here puts that wait where nothing else is pending. */
    __device__ __forceinline__ void settle() const { asm volatile("" :: "v"(h), "v"(tmo), "v"(cm)); }
    __device__ __forceinline__ bool go_on(const PSync &Sy, u64 t0) const {
        if (h == Sy.seq || tmo || cm == Sy.trip) return false;
        if (wall_clock64() - t0 > 50000000ull) { pst_word(Sy, Sy.timeout, 1u); return false; }   /* 0.5 s at 100 MHz */
        /* a nap between two looks: every look is 3 - 35 loads per lane that go to the memory side, and with a few hundred workgroups
         * polling they are in each other's (and the producers') way.  Short (64 cycles) here: this is on every hand-over's critical
         * path.  The one long wait of a pass naps longer in large launches (p_forward_tier). */
        __builtin_amdgcn_s_sleep(TQ_POLL_NAP);
        return true;
    }
};
/* why a poll gave up: 1 = the launch is over (halt / timeout), 2 = the top workgroup rejected the trial this pass
 * was built on (the pass is dropped, another trial follows) */
__device__ __forceinline__ int p_abort_code(const PSync &Sy) {
    const unsigned h = __hip_atomic_load(Sy.halt, RLX, TQ_LD_SCOPE), tmo = __hip_atomic_load(Sy.timeout, RLX, TQ_LD_SCOPE);
    const unsigned cm = (unsigned)(__hip_atomic_load(Sy.cmd + 1, RLX, TQ_LD_SCOPE) >> 32);
    return (h == Sy.seq || tmo || cm != Sy.trip) ? 1 : 2;
}
/* a tagged record of n doubles posted by the top workgroup (bounded spin; false = the launch is over) */
template <int N>
__device__ __forceinline__ bool p_read_top(const PSync &Sy, const u64 *src, unsigned tag, double (&v)[N]) {
    const u64 t0 = wall_clock64();
    bool ok;
    for (;;) {
        ok = true;
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = ld_tag(src + 2 * i, tag, ok);
        const unsigned h = __hip_atomic_load(Sy.halt, RLX, TQ_LD_SCOPE), tmo = __hip_atomic_load(Sy.timeout, RLX, TQ_LD_SCOPE);
        asm volatile("" :: "v"(h), "v"(tmo));            /* as PollGuard::settle */
        if (ok || h == Sy.seq || tmo) break;
        if (wall_clock64() - t0 > 50000000ull) { pst_word(Sy, Sy.timeout, 1u); break; }
    }
    return ok;
}

/* Backtracking chain of a line search (line_search, dual_Newton_tree.c:973-990): the reference moves lambda
 * by (tau - tauPrev) * dlambda per trial, tau <- beta * tau.  Trial number n of the chain (n = 1: one step of
 * `step` from the current duals, tau0 = tau after that step) is reached by replaying those axpys in order, so
 * its duals carry the roundings of the reference's sequence whatever n the evaluation starts from. */
struct PChain { int n; double tau0, beta; bool save_s; };      /* save_s: the sweep is the first trial of a line search: keep phase S's unclipped values (Data::xUncS) */

template <int NX, int NU, int MD>
struct PLds {
    using U = Uni<NX, NU, MD>;
    static constexpr int D = U::D, NBT = U::NBT, NZ = U::NZ;
    static constexpr int SLOTS = NBT + (MD == 2 ? 8 : MD * MD);      /* nodes a workgroup can own: its blocks' owners + (bottom tier) the leaves */
    static constexpr int NODE = 5 * NZ;                              /* per owned node: [x | u], clipped inverse Hessian, unclipped [x | u], modified gradient, unclipped [x | u] of phase S (see Data::xUncS) */
    /* A block's tall matrix T = [W ; rhs' ; Ut] (R = D + 1 + NX rows, D columns) lives TRANSPOSED in one region: entry (q, j) at
     * j * S + q.  The wave that factorises it holds row q in lane q, so its D loads (and later its D stores) differ only by
     * an immediate offset j * S -- one address register, no per-lane stride; consecutive lanes touch consecutive doubles (no
     * bank conflict) and S is odd, so that walking a row is conflict-free too.  The factorisation carries D identity rows
     * below T (lanes R .. R + D - 1, read from the shared table `idt`): they come out as the columns of L^-1.  What the
     * factorisation leaves -- y (lane D), CholUt (lanes D + 1 .. R - 1), L^-1 (lanes R ..) -- goes back IN PLACE, shifted by D
     * lanes: entry (q', j) at j * S + q' with q' = lane - D (the rows of W are dead by then; L itself is never needed again). */
    static constexpr int S = U::R;
    static constexpr int RI = U::R + D;                              /* rows a wave factorises, identity rows included */
    static_assert(RI <= 64, "tall matrix with the identity rows must fit one wavefront");
    static_assert(1 + NX + D <= S, "the factor's output rows must fit the region they replace");
    static constexpr int LDM = NX + 1;                               /* forward data of a block: row i = [z0_i | M_i0 .. M_i,NX-1] */
    /* the CONSTANTS of the nodes a workgroup owns live in LDS for the launch as well (loaded once, with the state): [A | B] of
     * the edges to an owned parent's children (column stride LDA = NX + 2 doubles: 16-byte aligned columns whose b128 reads
     * by 16 consecutive lanes hit 16 disjoint bank quads), their b, and {linear term, 1/weight, weight, lower, upper} per
     * entry of an owned node.  An iteration then touches global memory only through the tagged hand-over words: no load
     * ever queues behind a hand-over store (vmcnt is one in-order counter for loads AND stores on gfx9). */
    static constexpr int LDA = NX + 2;
    static constexpr int EDGE = NZ * LDA, CST = NZ * 5;
    /* active-set reuse (checkLastActiveSet): Schur record of my subtree root as last computed; v of every block (child -> parent inside the
     * workgroup when nothing is rebuilt); integer words: signature of every owned node (bit t: entry t at its upper bound, bit 16 + t:
     * at its lower bound), signatures per block now (owner, children's x parts) and when its factor was built, flags */
    static constexpr int ISIG = SLOTS + 2 * NBT * (1 + MD) + 8;
    static constexpr int DOUBLES = NBT * D * S + D * S + NBT * D * LDM + 2 * NBT * D + 16 + SLOTS * NODE + 2 * NBT * D + 3 * NX + 4 * FW
                                   + FW * U::WAVE_LDS + 32 + 64 + NBT * MD * EDGE + NBT * D + SLOTS * CST + 16
                                   + U::SCH + NBT * NX + ISIG / 2 + 2;
    lds_ptr tt, idt, mz, res, dl, red, node, lamb, lamroot, droot, part, wave0, wave, bat;     /* red: reductions of the top workgroup; bat: 64 doubles, reductions of a batch of trials */
    lds_ptr cab, cb, ccst, ctl;                                      /* ctl: the control block, kept by the top workgroup for the launch */
    lds_ptr srec, vrec;
    lds_iptr nsig, csig, bsig, fvalid, ruse;                         /* fvalid: my blocks' factor data are those of bsig; ruse: this pass keeps them */
    lds_iptr flag, abort;                                            /* abort: a poll gave up (launch over), leave at the next uniform point */
    __device__ PLds(double *base, int wave_id) {
        tt = to_lds(base); idt = tt + NBT * D * S; mz = idt + D * S; res = mz + NBT * D * LDM; dl = res + NBT * D; red = dl + NBT * D;
        node = red + 16; lamb = node + SLOTS * NODE;
        lamroot = lamb + 2 * NBT * D; droot = lamroot + 2 * NX; part = droot + NX; wave0 = part + 4 * FW; wave = wave0 + wave_id * U::WAVE_LDS;
        flag = (lds_iptr)(wave0 + FW * U::WAVE_LDS); abort = flag + 1; bat = wave0 + FW * U::WAVE_LDS + 32;
        cab = bat + 64; cb = cab + NBT * MD * EDGE; ccst = cb + NBT * D; ctl = ccst + SLOTS * CST;
        srec = ctl + 16; vrec = srec + U::SCH;
        nsig = (lds_iptr)(vrec + NBT * NX); csig = nsig + SLOTS; bsig = csig + NBT * (1 + MD); fvalid = bsig + NBT * (1 + MD); ruse = fvalid + 1;
    }
    __device__ __forceinline__ lds_ptr cab_(int loc, int child) const { return cab + (loc * MD + child) * EDGE; }      /* entry (r, column c) at c * LDA + r */
    __device__ __forceinline__ lds_ptr cb_(int loc) const { return cb + loc * D; }
    __device__ __forceinline__ lds_ptr ccst_(int q) const { return ccst + q * CST; }                                    /* entry t: 5 doubles at 5 t */
    __device__ __forceinline__ lds_ptr tt_(int loc) const { return tt + loc * D * S; }
    __device__ __forceinline__ lds_ptr mz_(int loc) const { return mz + loc * D * LDM; }
    /* part[4 w + i]: wave w's partials -- 0 termination norm, 1 res' * dlam, 2 dual function value */
    /* owned node `q` (heap order inside the tier subtree): entry t < NZ of [x | u] at +t, of the clipped
     * inverse Hessian at NZ + t, of the unclipped value at 2 NZ + t, of the modified gradient at 3 NZ + t */
    __device__ __forceinline__ lds_ptr node_(int q) const { return node + q * NODE; }
    /* dual vector of block `loc` (= duals of the owner node's children), double-buffered like lam0 / lam1 */
    __device__ __forceinline__ lds_ptr lamb_(int buf, int loc) const { return lamb + (buf * NBT + loc) * D; }
};

/* heap slot -> node / block index: slot q of the tier subtree s whose first block level is l0 */
template <int NX, int NU, int MD>
__device__ __forceinline__ int p_slot_node(int q, int l0, int s, const PConst &C) {
    using U = Uni<NX, NU, MD>;
    if (MD == 1) return C.nB + (l0 - C.Nr + q) * C.S + s;        /* chain part: slot = level offset, s = scenario */
    if (MD == 2) {                                                /* heap slot q sits on level t = floor(log2(q + 1)) of the subtree */
        const int t = 31 - __clz(q + 1);
        return ((1 << (l0 + t)) - 1) + (s << t) + (q + 1 - (1 << t));
    }
    int t = 0;
    while (q >= U::first(t + 1)) t++;
    return U::first(l0 + t) + s * U::width(t) + (q - U::first(t));
}

/* G + H of block p into LDS slot `loc`, split into a branch-free load half and a compute half so
 * that a wave with two blocks has both blocks' loads in flight at once.  The owner node's x, u,
 * QinvCal, RinvCal come from the workgroup's LDS node store (its own stage sweep wrote them); the
 * children's x and QinvCal likewise unless the children belong to the tier below (`foreign`): then
 * they were written by the child workgroups' stage sweeps (tagged words, polled). */
template <int NX, int NU, int MD>
struct GhRegs {
    double a[Uni<NX, NU, MD>::KS], pc[Uni<NX, NU, MD>::KS], z[Uni<NX, NU, MD>::KS], xk, bk, qk;
};

template <int NX, int NU, int MD>
__device__ __forceinline__ void p_gh_load(const PConst &C, const PSync &Sy, const PLds<NX, NU, MD> &L, int p, int loc, bool foreign, unsigned ns, int lane, GhRegs<NX, NU, MD> &G) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    const int row = lane & 15, g = lane >> 4;
    const bool live = row < D;
    const int rowc = live ? row : 0;                         /* dead rows load row 0 and are masked */
    const int cidx = rowc / NX, r = rowc - cidx * NX;
    const int k = kid0g<MD>(p, C) + cidx;
    lds_cptr AB = L.cab_(loc, cidx) + r;
    const int bo = NX * kid0g<MD>(p, C);
    lds_cptr own = L.node_(loc);                             /* [x | u] then the clipped inverse Hessian, NZ apart */
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const bool ok = live && cc < NZ;
        const int cz = (cc < NZ) ? cc : 0;
        const double av = AB[cz * PLds<NX, NU, MD>::LDA], zv = own[cz], pv = own[NZ + cz];
        G.a[s] = ok ? av : 0.0; G.pc[s] = ok ? pv : 0.0; G.z[s] = ok ? zv : 0.0;
    }
    double xv, qv;
    if (foreign && ns > 0u) {
        /* staged by the child workgroup in this launch: poll its tagged copy */
        const u64 *src = Sy.ndt + ((size_t)k * 2 * NX + r) * 2;
        const unsigned tag = Sy.seq | ns;
        const u64 t0 = wall_clock64();
        bool ok;
        for (;;) {
            PollGuard pg;
            ok = true;
            xv = ld_tag(src, tag, ok); qv = ld_tag(src + 2 * NX, tag, ok);
            pg.load(Sy);
            if (ok || !pg.go_on(Sy, t0)) { pg.settle(); break; }
        }
        if (!ok) *L.abort = p_abort_code(Sy);
    } else if (foreign) { const PDump *dp = C.dump; xv = dp->x[bo + rowc]; qv = dp->QinvCal[bo + rowc]; }   /* relaunch: staged by earlier kernels */
    else { lds_cptr kid = L.node_(MD * loc + 1 + cidx); xv = kid[r]; qv = kid[NZ + r]; }
    const double bv = L.cb_(loc)[rowc];
    G.xk = (live && g == 0) ? xv : 0.0; G.bk = (live && g == 0) ? bv : 0.0; G.qk = live ? qv : 0.0;
}

template <int NX, int NU, int MD>
__device__ __forceinline__ double p_gh_compute(PLds<NX, NU, MD> &L, int loc, int lane, const GhRegs<NX, NU, MD> &G, int termCondition, bool build) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    constexpr int S = PLds<NX, NU, MD>::S;
    const int row = lane & 15, g = lane >> 4;
    const bool live = row < D;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    double part = 0.0;
    lds_ptr Tt = L.tt_(loc) + row * S;                    /* column `row` of the tall matrix: W (symmetric) entries, rhs, Ut */
#pragma unroll
    for (int s = 0; s < U::KS; s++) {
        const int cc = g + 4 * s;
        const double ap = G.a[s] * G.pc[s];
        if (build) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G.a[s], ap, acc, 0, 0, 0);
        part = fma(G.a[s], G.z[s], part);
        if (build && live && cc < NX) Tt[D + 1 + cc] = -1.0 * ap;
    }
    part = rows_fold<false>(part);
    double e = 0.0;
    if (live && g == 0) {
        const double rv = fma(-1.0, G.xk, G.bk) + part;
        L.res[loc * D + row] = rv;                        /* the gradient itself (res' * dlam); the copy in the tall matrix takes the children's updates */
        if (build) Tt[D] = rv;
        e = (termCondition == 2) ? fabs(rv) : rv * rv;
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = g + 4 * rr;
        if (build && live && i < D) {
            double w = acc[rr];
            if (i == row) w += G.qk;
            Tt[i] = w;
        }
    }
    return (termCondition == 2) ? wmax(e) : wsum(e);
}

/* 1/sqrt(p) for p > 0, 0 otherwise: v_rsq_f64 (relative error ~ 2^-26) and ONE third-order correction
 * y0 (1 + e/2 + 3 e^2/8), e = 1 - p y0^2 -- four dependent operations instead of the six of two Newton
 * steps, remaining error ~ e^3 (far below one ulp) */
__device__ __forceinline__ double pivot_rsqrt3(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);
    const double h = fma(0.375, e, 0.5);
    const double y = fma(y0 * e, h, y0);
#ifdef TQ_PIVOT_SELECT     /* compare + two conditional moves per pivot */
    return p > 0.0 ? y : 0.0;
#else
    /* p <= 0 or NaN: v_rsq_f64 gives inf / NaN, the correction turns both into NaN, and v_max_f64 returns its other operand for a
     * NaN: one instruction instead of three on a chain that is issue-bound (tools/microbench/potrf_dpp_bench: 2803 -> 2667 cycles) */
    return __builtin_fmax(y, 0.0);
#endif
}

/* T[j][lane j] = K[j][lane j] + a for every j, other lanes T[j] = K[j] -- without lane masks (the masks of
 * all D lanes would sit in scalar registers for the whole sweep): v_writelane with an immediate lane */
template <int J, int D>
struct ShiftDiag {
    static __device__ __forceinline__ void run(double (&T)[D], const double (&K)[D], double a) {
        const double sj = rdlane(K[J], J) + a;
        const int slo = __builtin_amdgcn_readfirstlane(__double2loint(sj)), shi = __builtin_amdgcn_readfirstlane(__double2hiint(sj));
        int lo = __double2loint(K[J]), hi = __double2hiint(K[J]);
        /* s_nop: the scalar operands come from VALU instructions the assembler cannot see through */
        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"(slo), "s"(shi), "n"(J));
        T[J] = __hiloint2double(hi, lo);
        ShiftDiag<J + 1, D>::run(T, K, a);
    }
};
template <int D>
struct ShiftDiag<D, D> { static __device__ __forceinline__ void run(double (&)[D], const double (&)[D], double) {} };

/* in-register tall Cholesky as potrf_rows (left-looking, row broadcasts by readlane) and nothing else in
 * the loop: no per-lane select or store of the reciprocal pivots (the substitutions recompute 1/diag from
 * the stored factor; an exec-masked LDS store per column cost ~70 cycles each in tools/microbench/level_bench);
 * returns the smallest pivot (on-the-fly regularisation trigger) */
template <int D>
__device__ __forceinline__ double p_potrf_rows(double (&T)[D], int lane) {
    double pmin = __builtin_inf();
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        pmin = fmin(pmin, pj);
        T[j] = s * pivot_rsqrt3(pj);
    }
    return pmin;
}

/* 1/d for a diagonal entry of the factor (0 for the zero column of a non-positive pivot, as the
 * reciprocal pivot of the factorisation): v_rcp_f64 + two Newton steps */
__device__ __forceinline__ double diag_inv(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    return d > 0.0 ? y : 0.0;
}

/* dual_Newton_common.c:36-78 around p_potrf_rows: ALWAYS shifts the diagonal first, ON_THE_FLY
 * refactorises the shifted block when a diagonal entry of the factor came out <= regTol */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_factor_rows(Ctrl *ctrl, const Opts &O, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    constexpr int D = Uni<NX, NU, MD>::D;
    double K[D];
    if (O.regType != 0) {
#pragma unroll
        for (int j = 0; j < D; j++) K[j] = T[j];
        if (O.regType == 1) {
            ShiftDiag<0, D>::run(T, K, O.regValue);                                /* ddiare (ALWAYS) */
#pragma unroll
            for (int j = 0; j < D; j++) K[j] = T[j];
        }
    }
    for (int pass = 0; pass < 2; pass++) {
        const double pmin = p_potrf_rows<D>(T, lane);
        const bool small = pmin <= O.regTol * O.regTol;                             /* sqrt(pivot) <= regTol, incl. non-positive pivots */
        if (O.regType != 2 || !small || pass == 1) break;
        ShiftDiag<0, D>::run(T, K, O.regValue);                                    /* rare: shift and refactorise */
        if (lane == 0) atomicAdd(&ctrl->n_reg, 1);
    }
}

/* The factorisation on the critical path: as p_potrf_rows with the rare cases taken out of the pivot chain.  A pivot that is
 * not above `thr` (NaN included) only raises a flag -- one compare into a scalar mask per column, no select on the reciprocal
 * square root, no running minimum -- and the result is then garbage: the caller reloads the rows and takes the careful
 * path.  thr = regTol^2 under ON_THE_FLY regularisation (then the careful path shifts the diagonal first), 0 otherwise
 * (then it is p_potrf_rows itself, for its zero column of a non-positive pivot). */
template <int D>
__device__ __forceinline__ bool p_potrf_fast(double (&T)[D], int lane, double thr) {
    unsigned long long flagged = 0ull;
#pragma unroll
    for (int j = 0; j < D; j++) {
        double s = T[j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-T[k], rdlane(T[k], j), s);
        const double pj = rdlane(s, j);
        flagged |= __builtin_amdgcn_ballot_w64(!(pj > thr));
        const double y0 = __builtin_amdgcn_rsq(pj);
        const double e = fma(-(pj * y0), y0, 1.0);
        const double h = fma(0.375, e, 0.5);
        T[j] = s * fma(y0 * e, h, y0);
    }
    return flagged != 0ull;
}

/* first pass without a saved copy of the block (32 registers across the factorisation); true: the caller reloads the rows
 * (rare) and calls p_refactor_rows */
template <int NX, int NU, int MD>
__device__ __forceinline__ bool p_factor_rows_first(const Opts &O, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    constexpr int D = Uni<NX, NU, MD>::D;
    if (O.regType == 1) ShiftDiag<0, D>::run(T, T, O.regValue);                    /* ddiare (ALWAYS) */
#ifndef TQ_FAST_PIVOT      /* default: the running minimum costs nothing measurable on C2 (the pivot chain, not the issue rate, bounds the factorisation); the flag-only variant measured +1.5 % on C1, -1 % on C2 */
    const double pmin = p_potrf_rows<D>(T, lane);
    return O.regType == 2 && pmin <= O.regTol * O.regTol;
#else
    return p_potrf_fast<D>(T, lane, O.regType == 2 ? O.regTol * O.regTol : 0.0);   /* sqrt(pivot) <= regTol, incl. non-positive pivots */
#endif
}
/* the careful pass on freshly reloaded rows.  ON_THE_FLY (dual_Newton_common.c:60-68): a diagonal entry of the first factor
 * was <= regTol -- the flag of the fast pass says exactly that, its first flagged pivot being computed from unflagged ones --
 * so shift the diagonal and factorise again.  Otherwise a pivot was not positive: factorise again (ALWAYS: on the shifted
 * diagonal) with the zero-column convention of p_potrf_rows. */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_refactor_rows(Ctrl *ctrl, const Opts &O, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    constexpr int D = Uni<NX, NU, MD>::D;
    if (O.regType != 0) ShiftDiag<0, D>::run(T, T, O.regValue);
    (void)p_potrf_rows<D>(T, lane);
    if (O.regType == 2 && lane == 0) atomicAdd(&ctrl->n_reg, 1);
}

/* rows of the tall matrix of block `loc`, identity rows below: ONE address per lane, D loads with immediate offsets */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_load_rows(const PLds<NX, NU, MD> &L, int loc, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R, S = PLds<NX, NU, MD>::S;
    const int m = lane - R;
    lds_cptr src = (lane < R) ? L.tt_(loc) + lane : L.idt + (m < D ? m : D);      /* lanes beyond the identity rows read a zero column */
#pragma unroll
    for (int j = 0; j < D; j++) T[j] = src[j * S];
}

/* what the factorisation leaves, back into the block's region shifted by D lanes: y (lane D), CholUt (lanes D+1 .. R-1),
 * the columns of L^-1 (lanes R .. R+D-1); the rows of L are not needed again */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_store_factor(const PLds<NX, NU, MD> &L, int loc, int lane, const double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, R = U::R, S = PLds<NX, NU, MD>::S;
    if (lane >= D && lane < R + D) {
        lds_ptr dst = L.tt_(loc) + (lane - D);
#pragma unroll
        for (int j = 0; j < D; j++) dst[j * S] = T[j];
    }
}

/* Schur record [S | v] = CUt * [CUt' | y] (one f64 MFMA tile, K = D) straight from the CholUt / y just stored:
 * lane (i, g) feeds CUt[i][g + 4 st] as A and the same (i < NX) or y (i == NX) as B.
 * GLOBAL: the parent is another workgroup: the record travels as tagged words.  Otherwise the parent block `ploc` is
 * mine and the record is subtracted from its tall matrix in place (child number `cidx`: rows / columns cidx NX ..): the
 * parent's wave then loads rows that already carry its children -- nothing to subtract on its critical path. */
template <int NX, int NU, int MD, bool GLOBAL, bool RU = false>
__device__ __forceinline__ void p_schur(const PLds<NX, NU, MD> &L, int loc, int lane, int ploc, int cidx, u64 *sdst_glb, unsigned tag, const PSync &Sy) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, S = PLds<NX, NU, MD>::S;
    lds_fence();
    const int i = lane & 15, g = lane >> 4;
    lds_cptr src = L.tt_(loc) + g * S + ((i < NX) ? 1 + i : 0);
    double m[D / 4];
#pragma unroll
    for (int st = 0; st < D / 4; st++) m[st] = src[st * 4 * S];
    /* targets in the parent's tall matrix: S[ip][i] -> entry (row pos + ip, column pos + i); v[ip] -> rhs entry pos + ip */
    const int pos = cidx * NX;
    lds_ptr dst = L.tt_(GLOBAL ? 0 : ploc) + ((i < NX) ? (pos + i) * S + pos + g : (pos + g) * S + D);
    const int dstp = (i < NX) ? 4 : 4 * S;
    double old[4];
    if (!GLOBAL) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) old[rr] = (g + 4 * rr < NX && i <= NX) ? dst[rr * dstp] : 0.0;
    }
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
            if (GLOBAL) { const int off = (i < NX) ? ip + i * NX : NX * NX + ip; pst_tag(Sy, sdst_glb + 2 * off, acc[rr], tag); if (RU) L.srec[off] = acc[rr]; }      /* kept: re-posted by passes that keep the factors */
            else dst[rr * dstp] = old[rr] - acc[rr];
        }
    }
}

/* Forward preparation of block `loc` (calculate_delta_lambda :756-775 restated): the forward step of a block is
 *     dlam = L^-T (y - CholUt' delta),      delta = the NX entries of the parent block's solution that belong to the owner node,
 * a D-step substitution chain that can only start once delta is known.  Split it: z0 = L^-T y and M = L^-T CholUt' do not
 * depend on delta, so they are computed here, OFF the critical path (by waves that are idle during the upper levels of the
 * backward sweep, or while the workgroup waits for its parent), and the forward step shrinks to dlam = z0 - M delta: NX
 * fused multiply-adds.  L^-1 costs nothing: the factorisation carries D identity rows below the tall matrix (lanes that
 * were idle), which come out as the columns of L^-1.  [M | z0] = (L^-1)' [CholUt' | y] is one f64 MFMA tile (K = D), operands
 * straight from LDS; the result goes to the block's forward record mz: row i = [z0_i | M_i0 .. ]. */
template <int NX, int NU, int MD>
__device__ __forceinline__ void p_prep_forward(const PLds<NX, NU, MD> &L, int loc, int lane) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, S = PLds<NX, NU, MD>::S, LDM = PLds<NX, NU, MD>::LDM;
    const int i = lane & 15, g = lane >> 4;
    /* A[i][k] = Linv[k][i] (lane (i, g): k = g + 4 st);  B[k][j] = CholUt[j][k] (j < NX) or y[k] (j == NX) */
    lds_cptr asrc = L.tt_(loc) + g * S + 1 + NX + (i < D ? i : 0);
    lds_cptr bsrc = L.tt_(loc) + g * S + ((i < NX) ? 1 + i : 0);
    double a[D / 4], b[D / 4];
#pragma unroll
    for (int st = 0; st < D / 4; st++) { a[st] = asrc[st * 4 * S]; b[st] = bsrc[st * 4 * S]; }
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < D / 4; st++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(i < D ? a[st] : 0.0, i <= NX ? b[st] : 0.0, acc, 0, 0, 0);
    lds_ptr dst = L.mz_(loc) + g * LDM + ((i < NX) ? 1 + i : 0);
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
        if (g + 4 * rr < D && i <= NX) dst[rr * 4 * LDM] = acc[rr];          /* acc[rr] = [M | z0][row g + 4 rr][i] */
}

/* Forward sweep of a whole tier subtree without a barrier between its levels: wave w walks the path from the subtree root
 * (level t0) to ITS block of the last level (number w there); the solution of a block stays in the wave's registers
 * (entry i in lane i) and the NX entries the next block needs are read from it by v_readlane.  Blocks on several paths
 * are computed by every wave that passes through them (a handful of multiply-adds) and stored / counted by one of them.
 * from_parent: the step of the subtree root's owner node comes from the parent workgroup (tagged words, polled); otherwise
 * (top workgroup, t0 = 1) the root block's solution is in L.dl already.  to_children: the last level's blocks hand their
 * solution to the tier below as tagged words.  Returns the per-lane terms of res' * dlam of the blocks this wave owns. */
/* anc_up > 0 (bottom tier of a tree of three tiers or more, uniform trees): the step of my subtree root is NOT taken from the parent
 * workgroup.  The tier above has published the forward records [z0 | M] of its blocks (Sy.anc, long before they are needed: right
 * after its own backward sweep), and this workgroup walks the path through that tier itself, from the step of the tier's subtree
 * root (which the tier above THAT one posts): anc_up blocks of NX multiply-adds per lane instead of a forward sweep in the parent
 * workgroup followed by a hand-over -- the hand-over that involves the most workgroups (C2: 64 waiting for 8) and costs ~1.8 us
 * of a 27 us pass.  Same operands, same order of operations as the parent workgroup's own sweep (which it still runs for its own
 * stage sweep): the step is bit-identical to the one it would have handed down. */
template <int NX, int NU, int MD, bool ANC = false>
__device__ __forceinline__ double p_forward_tier(const PConst &C, const PSync &Sy, const PLds<NX, NU, MD> &L, int l0, int s, int th, int t0, int wave, int lane,
                                                 bool from_parent, bool to_children, unsigned tag, int anc_up = 0) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, LDM = PLds<NX, NU, MD>::LDM;
    const int li = lane < D ? lane : 0;
    const int wl = U::width(th - 1);                     /* blocks of the last level */
    const bool active = wave < wl;
    const int w = active ? wave : 0;
    double dv[NX];
    bool ok = true;
#ifndef TQ_POLL_ALL_WAVES
    if (from_parent && wave != 0) {
        /* only wave 0 looks for the step; the others take it from LDS behind a barrier.  Every look of every waiting wave goes to the memory
         * side and is in the way of the words the working workgroups post: with a quarter of the looks a C2 solve went from 99.1 to
         * 95.5 us, C3 from 131.5 to 123.1 (two looks in flight per wave instead -- samples half a round trip apart -- measured 2.5 %
         * SLOWER) */
        lds_barrier();
        ok = *L.abort == 0;
#pragma unroll
        for (int r = 0; r < NX; r++) dv[r] = L.droot[r];
    } else
#endif
    if (from_parent) {
        int ii = p_slot_node<NX, NU, MD>(0, l0, s, C);
        constexpr int AROW = NX * LDM;                   /* one block's slice on my path: NX rows of [z0 | M] */
        if constexpr (ANC) if (anc_up > 0) {
            static_assert(U::TH * AROW <= U::WAVE_LDS, "the path through the tier above must fit the wave's scratch");
            /* the records of my ancestors in the tier above, nearest first: rows cj NX .. of block aj, cj = my path's child ordinal there.
             * Every wave fetches them into its own scratch (no barrier); they were posted ~8 us ago, the first look finds them. */
            constexpr int NE = (U::TH * AROW + WAVE - 1) / WAVE;
            const u64 *rsrc[NE];
#pragma unroll
            for (int i = 0; i < NE; i++) {
                const int e0 = lane + i * WAVE, e = e0 < anc_up * AROW ? e0 : 0;
                const int j = e / AROW, wq = e - j * AROW;        /* j-th ancestor above my root */
                int a = ii, cj = 0;
                for (int q = 0; q <= j; q++) { cj = (a - 1) % MD; a = (a - 1) / MD; }
                rsrc[i] = Sy.anc + ((size_t)a * D * LDM + (size_t)cj * AROW + wq) * 2;
            }
            double rv[NE];
            const u64 t0a = wall_clock64();
            if (Sy.nap > 0) {
                for (;;) {
                    PollGuard pg;
                    ok = true;
#pragma unroll
                    for (int i = 0; i < NE; i++) rv[i] = ld_tag(rsrc[i], tag, ok);
                    pg.load(Sy);
                    if (ok || !pg.go_on(Sy, t0a)) { pg.settle(); break; }
                    __builtin_amdgcn_s_sleep(31);
                }
            } else {
                for (;;) {
                    PollGuard pg;
                    ok = true;
#pragma unroll
                    for (int i = 0; i < NE; i++) rv[i] = ld_tag(rsrc[i], tag, ok);
                    pg.load(Sy);
                    if (ok || !pg.go_on(Sy, t0a)) { pg.settle(); break; }
                }
            }
#pragma unroll
            for (int i = 0; i < NE; i++) { const int e0 = lane + i * WAVE; if (e0 < anc_up * AROW) L.wave[e0] = rv[i]; }
            lds_fence();
            for (int q = 0; q < anc_up; q++) ii = (ii - 1) / MD;          /* the subtree root of the tier above: its step is what arrives */
        }
        const bool ok_rec = ok;                                          /* (a poll that gave up: the launch is over or the pass is dropped) */
        const u64 *src = Sy.dlt + (size_t)NX * ii * 2;
        /* This wait is most of a pass for the lower tiers (the step comes down only after the whole backward sweep above), and every
         * look of every waiting workgroup is memory traffic in the way of the ones at work.  In a launch of more than ~128
         * workgroups (Sy.nap, set by the host) the bottom tier -- most of the workgroups, and the last link of the downward chain --
         * naps 32 x 64 cycles more between two looks: C3 (256 of 293 workgroups) 145 -> 134 us per solve (16 x 64: 136.5, 64 x 64:
         * 133.8); on C2's 64 it would cost 0.6 us.  Two loops, not one with the choice inside: ANY extra instruction in the short loop -- even a never-taken
         * scalar branch -- measured 1.4 us per C2 solve (profiles/r02_v2_nap_policy.txt). */
        const u64 t0c = wall_clock64();
        if (!to_children && Sy.nap > 0) {
            for (;;) {
                PollGuard pg;
                ok = true;
#pragma unroll
                for (int r = 0; r < NX; r++) dv[r] = ld_tag(src + 2 * r, tag, ok);
                pg.load(Sy);
                if (ok || !pg.go_on(Sy, t0c)) { pg.settle(); break; }
#ifndef TQ_LONG_NAP
#define TQ_LONG_NAP 31
#endif
                __builtin_amdgcn_s_sleep(TQ_LONG_NAP);
            }
        } else {
            for (;;) {
                PollGuard pg;
                ok = true;
#pragma unroll
                for (int r = 0; r < NX; r++) dv[r] = ld_tag(src + 2 * r, tag, ok);
                pg.load(Sy);
                if (ok || !pg.go_on(Sy, t0c)) { pg.settle(); break; }
            }
        }
        ok = __all(ok && ok_rec);
        if constexpr (ANC) if (anc_up > 0) {
            /* down the path through the tier above: block j's rows of my slice are lanes 0 .. NX - 1 (p_forward_tier's own formula) */
            const int lr = lane < NX ? lane : 0;
            for (int j = anc_up - 1; j >= 0; j--) {
                lds_cptr mr = L.wave + j * AROW + lr * LDM;
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int r = 0; r < NX; r += 2) { acc0 = fma(mr[1 + r], dv[r], acc0); if (r + 1 < NX) acc1 = fma(mr[2 + r], dv[r + 1], acc1); }
                const double mine = fma(-1.0, acc0 + acc1, mr[0]);
#pragma unroll
                for (int r = 0; r < NX; r++) dv[r] = rdlane(mine, r);
            }
        }
        if (!ok) { if (lane == 0) *L.abort = p_abort_code(Sy); }      /* the launch is over or the pass is dropped: nothing below may leave the workgroup */
        else if (wave == 0 && lane == 0) {                           /* the subtree root's own step: the stage sweep reads it from LDS */
#pragma unroll
            for (int r = 0; r < NX; r++) L.droot[r] = dv[r];
        }
#ifndef TQ_POLL_ALL_WAVES
        lds_barrier();
#endif
    } else if (t0 < th) {
        /* top workgroup: parent of my level-t0 block is the block above it (level t0 - 1), solved already */
        const int bpar = w / U::width(th - t0), cpar = (w / U::width(th - 1 - t0)) % MD;     /* parent's number on its level, my ordinal among its children */
        lds_cptr dsrc = L.dl + (U::first(t0 - 1) + bpar) * D + cpar * NX;
#pragma unroll
        for (int r = 0; r < NX; r++) dv[r] = dsrc[r];
    } else {
#pragma unroll
        for (int r = 0; r < NX; r++) dv[r] = 0.0;
    }
    double pd = 0.0;
    for (int t = t0; t < th; t++) {
        const int span = U::width(th - 1 - t);           /* last-level blocks below one block of level t */
        const int bt = w / span, loc = U::first(t) + bt;
        const bool owner = active && (w - bt * span) == 0;
        lds_cptr mr = L.mz_(loc) + li * LDM;
        double acc0 = 0.0, acc1 = 0.0;                   /* two chains: half the dependent latency */
#pragma unroll
        for (int r = 0; r < NX; r += 2) { acc0 = fma(mr[1 + r], dv[r], acc0); if (r + 1 < NX) acc1 = fma(mr[2 + r], dv[r + 1], acc1); }
        const double mine = fma(-1.0, acc0 + acc1, mr[0]);
        if (owner && ok && lane < D) {
            L.dl[loc * D + lane] = mine;                 /* a step computed from a failed poll must not replace the last good one (it is written back on leaving) */
            pd = fma(L.res[loc * D + lane], mine, pd);
            if (to_children && t == th - 1) {
                const int ii = p_slot_node<NX, NU, MD>(loc, l0, s, C);
                pst_tag(Sy, Sy.dlt + (size_t)(NX * kid0g<MD>(ii, C) + lane) * 2, mine, tag);
            }
        }
        if (t + 1 < th) {
            const int cn = (w / U::width(th - 2 - t)) % MD;          /* which child of this block lies on my path */
#pragma unroll
            for (int r = 0; r < NX; r++) dv[r] = rdlane(mine, cn * NX + r);
        }
    }
    return pd;
}

/* stage QP of owned node slot q (= node k) at the trial point lam_cur + step*dlam, by ONE 16-lane
 * group (lanes t of the group: t < NX state entries, NX <= t < NX+NU input entries).  Duals and steps
 * come from the workgroup's LDS copies (lamb / lamroot, dl / droot; `cb` = current buffer), constants
 * from global memory (packed); results go to the LDS node store and the other dual buffer; the
 * subtree root's x and QinvCal also go to the parent workgroup as tagged words (to_parent).
 * init: first sweep of a solve -- evaluate at the current duals themselves.
 * Returns the node's dual-function term (valid in every lane of the group). */
template <int NX, int NU, int MD, bool RU>
__device__ __forceinline__ double p_stage16(const PConst &C, const PSync &Sy, PLds<NX, NU, MD> &L, int q, int k, int t, lds_ptr gl /* group scratch: D + NX */,
                                            double step, int cb, bool active, bool init, bool to_parent, unsigned tag, const PChain &ch, bool dry) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, NZ = U::NZ;
    static_assert(NX + NU <= 16 && D <= 16, "16-lane stage needs nx+nu <= 16 and d <= 16");
    const bool parent = active && k < C.Np;
    const int nuk = parent ? NU : 0;
    const bool isx = t < NX, live = active && t < NX + nuk;
    /* branch-free loads: every lane reads from a valid (clamped) address and masks afterwards */
    const bool pk = parent && t < D, ox = active && isx && k > 0;
    const int qb = pk ? q : 0, tb = pk ? t : 0;                       /* my block's duals / step */
    const double lca = L.lamb_(cb, qb)[tb], dla = L.dl[qb * D + tb], ba = L.cb_(qb)[tb];
    const int qp = (ox && q > 0) ? (q - 1) / MD : 0, tp = (ox && q > 0) ? ((q - 1) % MD) * NX + t : 0;   /* my own slice in the parent's block */
    const int tr = ox ? t : 0;
    const double lcb = (q > 0) ? L.lamb_(cb, qp)[tp] : L.lamroot[cb * NX + tr];
    const double dlb = (q > 0) ? L.dl[qp * D + tp] : L.droot[tr];
    const bool pl = parent && live;
    double col[MD][NX];
#pragma unroll
    for (int cc = 0; cc < MD; cc++) {
        lds_cptr cp = L.cab_(pl ? q : 0, cc) + (pl ? t : 0) * PLds<NX, NU, MD>::LDA;            /* column t of [A | B] of the edge to child cc */
#pragma unroll
        for (int i = 0; i < NX; i++) col[cc][i] = cp[i];
    }
    lds_cptr cs = L.ccst_(active ? q : 0) + (t < NZ ? t : 0) * 5;
    const double lin = cs[0], winv = cs[1], wd = cs[2], lob = cs[3], hib = cs[4];
    double p_c = 0.0;
    {
        double v = lca, w = lcb;
        if (!init) {
            v = fma(step, dla, lca); w = fma(step, dlb, lcb);
            double tau = ch.tau0;
            for (int j = 1; j < ch.n; j++) {                 /* further trials of the same line search */
                const double t2 = __dmul_rn(ch.beta, tau), sj = __dsub_rn(t2, tau);
                v = fma(sj, dla, v); w = fma(sj, dlb, w); tau = t2;
            }
        }
        w = ox ? w : 0.0;
        if (pk) { gl[t] = v; p_c = ba * v; if (!init && !dry) L.lamb_(cb ^ 1, q)[t] = v; }
        if (ox && q == 0 && !init && !dry) L.lamroot[(cb ^ 1) * NX + t] = w;
        if (active && isx) gl[D + t] = w;
    }
    lds_fence();
    double p_q = 0.0, p_h = 0.0;
    bool at_hi = false, at_lo = false;                        /* the node's active set (dveccl_mask, clipping.c:212-218) */
    if (live) {
        double v = isx ? fma(-1.0, lin, gl[D + t]) : -1.0 * lin;
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
        if (unc >= hib) { val = hib; cal = 0.0; at_hi = true; } else if (unc <= lob) { val = lob; cal = 0.0; at_lo = true; } else { val = unc; cal = winv; }
        lds_ptr ns = L.node_(q);
        if (!dry) { if (ch.save_s) ns[4 * NZ + t] = ns[2 * NZ + t]; ns[t] = val; ns[NZ + t] = cal; ns[2 * NZ + t] = unc; ns[3 * NZ + t] = v; }
        if (!dry && to_parent && q == 0 && isx) {         /* my subtree root: the parent workgroup's G + H reads x and QinvCal */
            u64 *dst = Sy.ndt + ((size_t)k * 2 * NX + t) * 2;
            pst_tag(Sy, dst, val, tag); pst_tag(Sy, dst + 2 * NX, cal, tag);
        }
        p_q = (wd * val) * val;
        p_h = v * val;
    }
    if (RU && !dry) {
        /* signature of the node's active set: bit t = entry t sits at its upper bound, bit 16 + t = at its lower bound (x entries first).
         * What the reference keeps as xas / uas for compare_with_previous_active_set (dual_Newton_tree.c:334-368). */
        const int sh = (threadIdx.x & 48);
        const unsigned code = (unsigned)((__builtin_amdgcn_ballot_w64(at_hi) >> sh) & 0xFFFFull) | ((unsigned)((__builtin_amdgcn_ballot_w64(at_lo) >> sh) & 0xFFFFull) << 16);
        if (active && t == 0) {
            L.nsig[q] = (int)code;
            if (to_parent && q == 0) pst_tag(Sy, Sy.sgt + (size_t)k * 2, (double)(code & (((1u << NX) - 1u) * 0x10001u)), tag);      /* x part: my parent's block depends on it */
        }
    }
    const double qx = row16_sum(isx ? p_q : 0.0), hx = row16_sum(isx ? p_h : 0.0);
    const double ru = row16_sum(isx ? 0.0 : p_q), hu = row16_sum(isx ? 0.0 : p_h);
    p_c = row16_sum(p_c);
    double f = -0.5 * qx - p_c;
    f += hx;
    f -= 0.5 * ru;
    f += hu;
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
    int chain[8];                           /* multistage trees: the tier lies in the chain part (one block per level) */
    int G;
    const int *wg_of_block;                 /* blockIdx.x -> workgroup id: families of tier subtrees share an XCD (hardware
                                               places workgroup b on XCD b % 8), so most hand-overs stay inside one L2 */
};

/* diagnostic stamps of the persistent kernel: first workgroup of every tier, thread 0, iteration O.stamps of the launch */
__device__ __forceinline__ void pstamp(const PConst &C, const Opts &O, unsigned e, int tier, int s, int slot) {
#ifndef TQ_STAMPS      /* diagnostic builds only (TQ_DEFS=-DTQ_STAMPS python treeqp_amd/build.py): even the never-taken stamp branches cost ~0.2 us each, 6 us per C2 solve */
    return;
#endif
    if (O.stamps == (int)e && threadIdx.x == 0 && s == 0 && slot < 32 && tier < 8) {
        unsigned long long *st = C.dump->stamps;
        const unsigned long long ck = clock64();     /* the shader clock only: reading the 100 MHz wall clock takes ~1 us and would sit inside every phase */
        st[(tier * 32 + slot) * 2 + 0] = ck;
        st[(tier * 32 + slot) * 2 + 1] = slot >= 25 ? wall_clock64() : ck / 24;    /* launch-level stamps: the chip-wide 100 MHz clock (comparable across workgroups); phases: nominal 2.4 GHz in 100 MHz ticks */
    }
}

/* subtract the Schur records of the children in the tier below: tagged words written by the child
 * workgroups, polled until every word of this lane's rows carries `tag`; false when the poll gave up */
template <int NX, int NU, int MD>
__device__ __forceinline__ bool p_sub_children_tagged(const PSync &Sy, const u64 *sch, unsigned tag, int lane, double (&T)[Uni<NX, NU, MD>::D]) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const bool vrow = lane == D, need = lane <= D;
    const int lc = lane < D ? lane / NX : 0;
    const int r = lane < D ? lane - lc * NX : 0;
    const int off = vrow ? NX * NX : r, stride = vrow ? 1 : NX;      /* (per load instruction the 16 lanes read 16 consecutive tagged doubles: two lines; the transposed read -- one line per LANE -- measured 5 % slower per C2 solve) */
    double v[MD][NX];
    const u64 t0 = wall_clock64();
    bool ok;
    for (;;) {
        PollGuard pg;
        ok = true;
#pragma unroll
        for (int c = 0; c < MD; c++) {
            const u64 *src = sch + ((size_t)c * U::SCH + off) * 2;
#pragma unroll
            for (int j = 0; j < NX; j++) v[c][j] = ld_tag(src + (size_t)j * stride * 2, tag, ok);
        }
        pg.load(Sy);
        ok = ok || !need;
        if (ok || !pg.go_on(Sy, t0)) { pg.settle(); break; }
    }
#pragma unroll
    for (int c = 0; c < MD; c++) {
        const bool act = (vrow || (lane < D && lc == c)) && ok;       /* words of a failed poll may hold anything */
#pragma unroll
        for (int j = 0; j < NX; j++) T[c * NX + j] -= act ? v[c][j] : 0.0;
    }
    return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* checkLastActiveSet (dual_Newton_tree.c:334-405, 556-614, 681-709): keep factors whose blocks did not change */
/* ------------------------------------------------------------------------------------------ */
/* The dual Hessian block of a parent depends on the ACTIVE SET of its owner node (x and u) and of its children (x) only:
 * W = C P C' + blockdiag(P_kids) with P = the clipped inverse weights, which are the inverse weights where a bound is inactive
 * and zero where it is active.  The reference therefore compares the active set of every node with the previous one and
 * factorises again only up to the last block that changed; the others keep their Cholesky factor and only substitute.
 * Here the unit of reuse is the WORKGROUP (a tier subtree): it keeps its factor data -- L^-1, CholUt, M, the Schur record of
 * its root -- when (a) the signatures of all its blocks equal those of the pass that built the data and (b) every workgroup
 * below it does the same (its Schur records are then unchanged too); condition (b) travels upwards as one tagged word per
 * subtree root (rfl), the x signature of a subtree root node as another (sgt).  A pass that keeps the factors replaces, per
 * block, G + H's Hessian build and the factorisation (5 k cycles) by three small matrix-vector products with the stored
 * L^-1 / CholUt: y = L^-1 (res - sum of the children's v), v = CholUt y, z0 = L^-T y (0.7 k cycles).
 * Wave 0 decides; the result is *L.ruse (and `csig`: the signatures of this pass, saved as `bsig` when a build completes). */
template <int NX, int NU, int MD>
__device__ __noinline__ void p_reuse_decide(const PConst &C, const PSync &Sy, PLds<NX, NU, MD> &L, int l0, int s, int th, int nbt,
                                               bool is_top, bool is_bottom, bool can, unsigned tag_sweep, unsigned tag_pass, int lane) {
    using U = Uni<NX, NU, MD>;
    const int nint = U::first(th - 1);
    const unsigned xm = ((1u << NX) - 1u) * 0x10001u;
    const bool mine = lane < nbt;
    const int loc = mine ? lane : 0;
    const bool foreign = mine && !is_bottom && loc >= nint;
    const int node = p_slot_node<NX, NU, MD>(loc, l0, s, C);
    int cur[1 + MD];
    cur[0] = L.nsig[loc];
#pragma unroll
    for (int c = 0; c < MD; c++) cur[1 + c] = foreign ? 0 : (L.nsig[MD * loc + 1 + c] & (int)xm);
    bool ok = true, kids = true;
    if (can && !is_bottom) {
        /* children of my last level live in the tier below: their x signatures and whether their workgroups keep their factors */
        const u64 t0 = wall_clock64();
        for (;;) {
            PollGuard pg;
            ok = true;
            kids = true;
            if (foreign) {
#pragma unroll
                for (int c = 0; c < MD; c++) {
                    const size_t kid = (size_t)(kid0g<MD>(node, C) + c);
                    cur[1 + c] = (int)(unsigned)ld_tag(Sy.sgt + kid * 2, tag_sweep, ok);
                    kids = kids && ld_tag(Sy.rfl + kid * 2, tag_pass, ok) != 0.0;
                }
            }
            pg.load(Sy);
            ok = __all(ok);
            if (ok || !pg.go_on(Sy, t0)) { pg.settle(); break; }
        }
        if (!ok && lane == 0) *L.abort = p_abort_code(Sy);
    }
    bool same = can && ok && kids && *L.fvalid != 0;
#pragma unroll
    for (int c = 0; c <= MD; c++) {
        same = same && cur[c] == L.bsig[loc * (1 + MD) + c];
        if (mine) L.csig[loc * (1 + MD) + c] = cur[c];
    }
    const bool all_same = __all(!mine || same);
    if (lane == 0) {
        *L.ruse = all_same ? 1 : 0;
        if (!is_top && ok) pst_tag(Sy, Sy.rfl + (size_t)p_slot_node<NX, NU, MD>(0, l0, s, C) * 2, all_same ? 1.0 : 0.0, tag_pass);
    }
}

/* one block of a backward sweep that keeps the factors (see p_reuse_decide); scr: 2 D doubles of the wave's scratch.
 * Returns the per-lane term of res' * dlam for the root block of the tree (0 otherwise). */
template <int NX, int NU, int MD>
__device__ __noinline__ double p_reuse_block(const PConst &C, const PSync &Sy, PLds<NX, NU, MD> &L, int loc, int ii, int t, int th, bool is_bottom, bool is_root,
                                                unsigned tag, int lane, lds_ptr scr) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D, S = PLds<NX, NU, MD>::S, LDM = PLds<NX, NU, MD>::LDM;
    const int li = lane < D ? lane : 0;
    const int c = li / NX, j = li - c * NX;
    double r = L.res[loc * D + li];
    bool ok = true;
    if (t < th - 1) r -= L.vrec[(MD * loc + 1 + c) * NX + j];               /* my children are mine: they left their v in LDS */
    else if (!is_bottom) {
        /* children in the tier below: the v part of their (re-posted) Schur records */
        const u64 *src = Sy.sch + ((size_t)(kid0g<MD>(ii, C) + c) * U::SCH + NX * NX + j) * 2;
        const u64 t0 = wall_clock64();
        double vv = 0.0;
        for (;;) {
            PollGuard pg;
            ok = true;
            vv = ld_tag(src, tag, ok);
            pg.load(Sy);
            ok = __all(ok);
            if (ok || !pg.go_on(Sy, t0)) { pg.settle(); break; }
        }
        if (!ok && lane == 0) *L.abort = p_abort_code(Sy);
        r -= ok ? vv : 0.0;
    }
    if (lane < D) scr[lane] = r;
    lds_fence();
    lds_cptr X = L.tt_(loc);
    double y0 = 0.0, y1 = 0.0;                                             /* y = L^-1 r: row li of L^-1 is contiguous */
#pragma unroll
    for (int k = 0; k < D; k += 2) { y0 = fma(X[li * S + 1 + NX + k], scr[k], y0); y1 = fma(X[li * S + 2 + NX + k], scr[k + 1], y1); }
    const double y = y0 + y1;
    if (lane < D) scr[D + lane] = y;
    lds_fence();
    const int lr = lane < NX ? lane : 0;
    double z0 = 0.0, z1 = 0.0, v0 = 0.0, v1 = 0.0;                         /* z = L^-T y (the block's forward record z0), v = CholUt y */
#pragma unroll
    for (int k = 0; k < D; k += 2) {
        const double ya = scr[D + k], yb = scr[D + k + 1];
        z0 = fma(X[k * S + 1 + NX + li], ya, z0); z1 = fma(X[(k + 1) * S + 1 + NX + li], yb, z1);
        v0 = fma(X[k * S + 1 + lr], ya, v0); v1 = fma(X[(k + 1) * S + 1 + lr], yb, v1);
    }
    const double z = z0 + z1, v = v0 + v1;
    double pd = 0.0;
    if (lane < D) L.mz_(loc)[lane * LDM] = z;
    if (is_root) {
        if (lane < D) {
            if (th == 1 && !is_bottom && ok) pst_tag(Sy, Sy.dlt + (size_t)(NX * kid0g<MD>(0, C) + lane) * 2, z, tag);
            L.dl[lane] = z; pd = L.res[lane] * z;
        }
    } else if (t > 0) {
        if (lane < NX) L.vrec[loc * NX + lane] = v;
    } else if (ok) {
        /* my subtree root: the parent workgroup gets the record again -- S as it was built, v of this pass */
        u64 *dst = Sy.sch + (size_t)ii * U::SCH * 2;
        if (lane < NX * NX) pst_tag(Sy, dst + 2 * lane, L.srec[lane], tag);
        if (lane < NX) pst_tag(Sy, dst + 2 * (NX * NX + lane), v, tag);
    }
    lds_fence();
    return pd;
}

/* stage sweep over the nodes this workgroup owns (the owner nodes of its blocks in heap order, then --
 * bottom tier -- the leaves below), four nodes per wave; returns the wave's sum of the node terms */
template <int NX, int NU, int MD, bool RU>
__device__ __forceinline__ double p_stage_owned(const PConst &C, const PSync &Sy, PLds<NX, NU, MD> &L, int l0, int nown, int s, int wave, int lane,
                                                double step, int cb, bool init, bool to_parent, unsigned tag, const PChain &ch, bool dry) {
    using U = Uni<NX, NU, MD>;
    constexpr int D = U::D;
    const int grp = lane >> 4, t16 = lane & 15;
    lds_ptr gl = L.wave + 8 + grp * (D + NX + 2);
    double fsum = 0.0;
    for (int base = 0; base < nown; base += FW * 4) {
        const int q = base + wave * 4 + grp;
        const bool active = q < nown;
        const int k = active ? p_slot_node<NX, NU, MD>(q, l0, s, C) : 0;
        fsum += p_stage16<NX, NU, MD, RU>(C, Sy, L, active ? q : 0, k, t16, gl, step, cb, active, init, to_parent, tag, ch, dry);
    }
    return rows_fold<false>(fsum);       /* every lane of a 16-lane group holds its group's sum */
}

/* top workgroup: sums of the per-workgroup {fval, dot} partials (tag tag_p; skipped when want_parts is
 * false) and of the termination partials (tag tag_e; sum or maximum).  Every thread polls one entry per
 * round; the sums are taken in a FIXED order (cross-lane tree inside a wave, waves in order, rounds in
 * order), so the result does not depend on arrival times.  Results valid in thread 0.  Returns false in
 * every thread when a poll gave up. */
template <int NX, int NU, int MD>
__device__ __forceinline__ bool p_gather3(const PSync &Sy, PLds<NX, NU, MD> &L, int count, bool want_parts, unsigned tag_p, unsigned tag_e, bool err_max,
                                          double &fa, double &da, double &ea) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fa = 0.0; da = 0.0; ea = 0.0;
    for (int c0 = 0; c0 < count; c0 += FW * WAVE) {
        const int w = c0 + (int)threadIdx.x;
        double f = 0.0, d = 0.0, er = 0.0;
        if (w < count) {
            const u64 *pp = Sy.parts + (size_t)w * 4, *pe = Sy.errs + (size_t)w * 2;
            const u64 t0 = wall_clock64();
            bool ok;
            for (;;) {
                PollGuard pg;
                ok = true;
                if (want_parts) { f = ld_tag(pp, tag_p, ok); d = ld_tag(pp + 2, tag_p, ok); }
                er = ld_tag(pe, tag_e, ok);
                pg.load(Sy);
                if (ok || !pg.go_on(Sy, t0)) { pg.settle(); break; }
            }
            if (!ok) { *L.abort = 1; f = 0.0; d = 0.0; er = 0.0; }
        }
        f = wsum(f); d = wsum(d); er = err_max ? wmax(er) : wsum(er);
        if (lane == 0) { L.red[3 * wave] = f; L.red[3 * wave + 1] = d; L.red[3 * wave + 2] = er; }
        __syncthreads();
        if (*L.abort) return false;
        if (threadIdx.x == 0) {
            for (int v = 0; v < FW; v++) { fa += L.red[3 * v]; da += L.red[3 * v + 1]; ea = err_max ? nanmax(ea, L.red[3 * v + 2]) : ea + L.red[3 * v + 2]; }
        }
        __syncthreads();
    }
    return true;
}

/* top workgroup: sums over the workgroups of the K dual-function partials of a batch of trials, in a fixed order */
template <int NX, int NU, int MD>
__device__ __forceinline__ bool p_gather_batch(const PSync &Sy, PLds<NX, NU, MD> &L, int count, int K, unsigned tag, double (&fa)[8]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_ptr red = L.bat + 32;
#pragma unroll
    for (int k = 0; k < 8; k++) fa[k] = 0.0;
    for (int c0 = 0; c0 < count; c0 += FW * WAVE) {
        const int w = c0 + (int)threadIdx.x;
        double f[8];
#pragma unroll
        for (int k = 0; k < 8; k++) f[k] = 0.0;
        if (w < count) {
            const u64 *pp = Sy.bparts + (size_t)w * 16;
            const u64 t0 = wall_clock64();
            bool ok;
            for (;;) {
                ok = true;
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < K) f[k] = ld_tag(pp + 2 * k, tag, ok);
                const unsigned h = __hip_atomic_load(Sy.halt, RLX, TQ_LD_SCOPE), tmo = __hip_atomic_load(Sy.timeout, RLX, TQ_LD_SCOPE);
                asm volatile("" :: "v"(h), "v"(tmo));    /* as PollGuard::settle */
                if (ok || h == Sy.seq || tmo) break;
                if (wall_clock64() - t0 > 50000000ull) { pst_word(Sy, Sy.timeout, 1u); break; }
            }
            if (!ok) {
                *L.abort = 1;
#pragma unroll
                for (int k = 0; k < 8; k++) f[k] = 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) f[k] = wsum(f[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) red[wave * 8 + k] = f[k];
        }
        __syncthreads();
        if (*L.abort) return false;
        if (threadIdx.x == 0) {
            for (int v = 0; v < FW; v++)
#pragma unroll
                for (int k = 0; k < 8; k++) fa[k] += red[v * 8 + k];
        }
        __syncthreads();
    }
    return true;
}

/* the life of one workgroup = one tier subtree: tier `tier`, subtree (complete part) or scenario (chain part) `s` */
template <int NX, int NU, int MD, bool RU, int ROLE = 0, bool FULLTH = false, bool ANC = false>      /* ROLE: 0 found at run time; 1 bottom tier, 2 a tier in between, 3 top tier (of two or more tiers); FULLTH: the tier has Uni::TH levels;
                                                                                                       * ANC: uniform tree -- with three tiers or more the bottom tier walks down its path through tier 1 itself (p_forward_tier) */
__device__ __forceinline__ void p_run(const PConst &C, const Opts &O, const PGeom &Gm, const PSync &Sy_in, int prologue, int wg, int tier, int s, double *lds_all) {
    using U = Uni<NX, NU, MD>;
    PSync Sy = Sy_in;
    Sy.trip = 0u;
    constexpr int D = U::D, NZ = U::NZ;
    /* the control block lives in the top workgroup's LDS during the launch (its thread 0 is the only one that takes decisions):
     * a verdict is a dozen dependent reads and writes of it, each a trip to L2 when it sits in global memory.  Global memory
     * holds it between launches (cg): read at the start of a relaunch, written back with the verdict. */
    Ctrl *cg = C.ctrl;
    static_assert(sizeof(Ctrl) <= 16 * sizeof(double), "control block copy");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    PLds<NX, NU, MD> L(lds_all, wave);
    Ctrl *c = reinterpret_cast<Ctrl *>((double *)L.ctl);
    const int l0 = Gm.l0[tier], th = FULLTH ? U::TH : Gm.l1[tier] - l0;
    const bool is_top = ROLE == 0 ? tier == Gm.n_tiers - 1 : ROLE == 3, is_bottom = ROLE == 0 ? tier == 0 : ROLE == 1;
    const bool anc_on = ANC && !RU && Gm.n_tiers >= 3;                 /* tier 1 publishes its forward records, tier 0 walks down its path through them */
    const bool anc_pub = anc_on && ROLE == 2 && tier == 1;
    const int anc_up = (anc_on && ROLE == 1) ? U::TH : 0;
    const int nbt = U::first(th);                                      /* my blocks */
    const int nown = nbt + (is_bottom ? U::width(th) : 0);             /* nodes I own */
    const int root_blk = p_slot_node<NX, NU, MD>(0, l0, s, C);         /* subtree root block (= node) */
    const unsigned long long t_start = wall_clock64();
#ifdef TQ_STAMPS
    /* placement census: which XCD / SE / CU this workgroup landed on (HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20) */
    if (threadIdx.x == 0 && wg < 1024)
        C.dump->stamps[8 * 32 * 2 + wg] = ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
#endif
    pstamp(C, O, (unsigned)O.stamps, tier, s, 25);                     /* 25: workgroup started */
    int cur = 0;
    unsigned nd = 0u;          /* stage sweeps (= {fval, dot} reductions) of this launch so far */
    bool have_dl = false;      /* a forward sweep of this launch has filled the step of my blocks */
    unsigned nbat = 0u;        /* batches of extra line-search trials of this launch so far */
    bool decided = false;      /* the latest stage sweep is a trial the top workgroup has already accepted (end of a batch) */
    if (prologue) {
        /* a fresh solve: the control block starts from zero (nobody else reads it during the launch) */
        if (is_top && threadIdx.x == 0) {
            c->done = 0; c->status = 0; c->iter = 0; c->cur = 0; c->ls_pending = 0; c->ls_iter = 0; c->ls_total = 0; c->ls_last = 0;
            c->restart_counter = 0; c->n_reg = 0; c->tau = 0.0; c->tauPrev = 0.0; c->fval0 = 0.0; c->fval = 0.0; c->dot = 0.0; c->err = 0.0;
            cg->n_reg = 0;                                                    /* counted in global memory by whoever regularises a block */
#ifdef TQ_REUSE_DEBUG
            c->pad0 = 0;
#endif
        }
    } else {
        if (__hip_atomic_load(&cg->done, RLX, AGENT) || __hip_atomic_load(&cg->ls_pending, RLX, AGENT)) return;
        cur = __hip_atomic_load(&cg->cur, RLX, AGENT);
        if (is_top && threadIdx.x == 0) *c = *cg;
    }

    /* ---- load the state this workgroup owns: duals of my blocks and of my root, node store ---- */
    {
        const PDump *dp = C.dump;
        if (threadIdx.x == 0) { *L.abort = 0; *L.fvalid = 0; *L.ruse = 0; }
        for (int i = threadIdx.x; i < D * PLds<NX, NU, MD>::S; i += FW * WAVE) {      /* identity rows of the tall matrices: entry (m, j) at j * S + m */
            const int j = i / PLds<NX, NU, MD>::S, m = i - j * PLds<NX, NU, MD>::S;
            L.idt[i] = (m == j) ? 1.0 : 0.0;
        }
        /* constants of my nodes ([A | B] and b of the edges below my blocks' owner nodes, stage constants of every node I own) and the
         * duals of my blocks: EVERY global load is issued before the first one is used (fixed trip counts, clamped addresses) -- one
         * memory latency for the lot instead of one per trip of a loop (the state load was 2.5 us of a 110 us solve) */
        {
            using PL = PLds<NX, NU, MD>;
            constexpr int T = FW * WAVE;
            constexpr int IT_AB = (PL::NBT * MD * NX * NZ + T - 1) / T, IT_CST = (PL::SLOTS * NZ * 5 + T - 1) / T, IT_B = (PL::NBT * D + T - 1) / T;
            const double *lsrc = prologue ? C.lam0_src : (cur ? dp->lam1 : dp->lam0);
            double vab[IT_AB], vcst[IT_CST], vb[IT_B], vlam[IT_B];
            const int n_ab = nbt * MD * NX * NZ, n_cst = nown * NZ * 5, n_b = nbt * D;
#pragma unroll
            for (int it = 0; it < IT_AB; it++) {
                const int i0 = (int)threadIdx.x + it * T, i = i0 < n_ab ? i0 : 0;
                const int e = i / (NX * NZ), w = i - e * (NX * NZ), loc = e / MD, cc = e - loc * MD;
                const int kid = kid0g<MD>(p_slot_node<NX, NU, MD>(loc, l0, s, C), C) + cc;
                vab[it] = C.AB[(size_t)(kid - 1) * NX * NZ + w];
            }
#pragma unroll
            for (int it = 0; it < IT_CST; it++) {
                const int i0 = (int)threadIdx.x + it * T, i = i0 < n_cst ? i0 : 0;
                const int q = i / (NZ * 5), w = i - q * (NZ * 5);
                vcst[it] = C.cst[(size_t)p_slot_node<NX, NU, MD>(q, l0, s, C) * 16 * 5 + w];
            }
#pragma unroll
            for (int it = 0; it < IT_B; it++) {
                const int i0 = (int)threadIdx.x + it * T, i = i0 < n_b ? i0 : 0;
                const int loc = i / D, t = i - loc * D;
                const int o = NX * kid0g<MD>(p_slot_node<NX, NU, MD>(loc, l0, s, C), C) + t;
                vb[it] = C.b[o]; vlam[it] = lsrc[o];
            }
            const double vroot = (threadIdx.x < NX && root_blk > 0) ? lsrc[NX * root_blk + threadIdx.x] : 0.0;
#pragma unroll
            for (int it = 0; it < IT_AB; it++) {
                const int i = (int)threadIdx.x + it * T;
                if (i < n_ab) {
                    const int e = i / (NX * NZ), w = i - e * (NX * NZ), loc = e / MD, cc = e - loc * MD, col = w / NX, r = w - col * NX;
                    L.cab_(loc, cc)[col * PL::LDA + r] = vab[it];
                }
            }
#pragma unroll
            for (int it = 0; it < IT_CST; it++) {
                const int i = (int)threadIdx.x + it * T;
                if (i < n_cst) { const int q = i / (NZ * 5), w = i - q * (NZ * 5); L.ccst_(q)[w] = vcst[it]; }
            }
#pragma unroll
            for (int it = 0; it < IT_B; it++) {
                const int i = (int)threadIdx.x + it * T;
                if (i < n_b) { const int loc = i / D, t = i - loc * D; L.cb_(loc)[t] = vb[it]; L.lamb_(cur, loc)[t] = vlam[it]; }
            }
            if (threadIdx.x < NX) L.lamroot[cur * NX + threadIdx.x] = vroot;
        }
        if (!prologue) {
            for (int i = threadIdx.x; i < nown * 16; i += FW * WAVE) {
                const int q = i >> 4, t = i & 15;
                const int k = p_slot_node<NX, NU, MD>(q, l0, s, C);
                lds_ptr ns = L.node_(q);
                if (t < NX) { ns[t] = dp->x[NX * k + t]; ns[NZ + t] = dp->QinvCal[NX * k + t]; ns[2 * NZ + t] = dp->xUnc[NX * k + t]; ns[4 * NZ + t] = dp->xUncS[NX * k + t]; }
                else if (t < NZ && k < C.Np) { ns[t] = dp->u[NU * k + t - NX]; ns[NZ + t] = dp->RinvCal[NU * k + t - NX]; ns[2 * NZ + t] = dp->uUnc[NU * k + t - NX]; ns[4 * NZ + t] = dp->uUncS[NU * k + t - NX]; }
            }
        }
        __syncthreads();
    }

    /* my {fval, dot} partial of stage sweep number nd to the top workgroup (thread 0, after the sweep's barrier) */
    auto post_parts = [&]() {
        double f = 0.0, d = 0.0;
        for (int w = 0; w < FW; w++) { f += L.part[4 * w + 2]; d += L.part[4 * w + 1]; }
        pst_tag(Sy, Sy.parts + (size_t)wg * 4, f, Sy.seq | nd);
        pst_tag(Sy, Sy.parts + (size_t)wg * 4 + 2, d, Sy.seq | nd);
    };

    pstamp(C, O, (unsigned)O.stamps, tier, s, 26);                     /* 26: state and constants loaded */
    if (prologue) {
        /* ---- first sweep of the solve: stage QPs at lambda0, fval0 ---- */
        const double fsum = p_stage_owned<NX, NU, MD, RU>(C, Sy, L, l0, nown, s, wave, lane, 0.0, cur, true, !is_top, Sy.seq | 1u, PChain{1, 0.0, 0.0}, false);
        if (lane == 0) { L.part[4 * wave + 2] = fsum; L.part[4 * wave + 1] = 0.0; }
        __syncthreads();
        nd = 1u;
        if (threadIdx.x == 0) post_parts();
    }

    pstamp(C, O, (unsigned)O.stamps, tier, s, 27);                     /* 27: first sweep done */
    /* one trip = one PASS: G + H at the point of the latest stage sweep, then -- if the top workgroup accepts
     * that point -- backward sweep, forward sweep and the first trial of the next line search; if it rejects
     * it (Armijo test failed, more trials to go) the pass is dropped after G + H and the trip ends with the
     * next trial of the same line search instead.  Tags count passes, the control block counts iterations. */
    for (unsigned e = 1u;; e++) {
        const unsigned tag_e = Sy.seq | e;
        Sy.trip = tag_e;
        int verdict = 0;                                                  /* 2: this pass is dropped, another trial follows */

        int sl = 0;
        pstamp(C, O, e, tier, s, sl++);                                   /* 0: iteration start */
#ifdef TQ_STAMPS
        pstamp(C, O, e, tier, s, 31);                                     /* slot 31 - slot 0 = cost of one stamp */
#endif
        /* ---- do I keep my factors this pass? (checkLastActiveSet) ---- */
        bool build = true;
        if (RU) {
            if (wave == 0) p_reuse_decide<NX, NU, MD>(C, Sy, L, l0, s, th, nbt, is_top, is_bottom, nd > 0u, Sy.seq | nd, tag_e, lane);
            __syncthreads();
            build = *L.ruse == 0;
            if (build && threadIdx.x == 0) *L.fvalid = 0;                       /* G + H overwrites the tall matrices, i.e. the factor data */
#ifdef TQ_REUSE_DEBUG
            if (!build && is_top && threadIdx.x == 0) c->pad0 += 1;
#endif
        }
        /* ---- G + H for my blocks (heap order inside the subtree), two blocks per wave in flight; the
         * children of my bottom-level blocks were staged by the child workgroups (polled) ---- */
        {
            double err = 0.0;
            const int nint = U::first(th - 1);                         /* blocks above my bottom level: children are mine */
            for (int loc0 = wave; loc0 < nbt; loc0 += 2 * FW) {
                const int loc1 = loc0 + FW;
                GhRegs<NX, NU, MD> g0, g1;
                p_gh_load<NX, NU, MD>(C, Sy, L, p_slot_node<NX, NU, MD>(loc0, l0, s, C), loc0, !is_bottom && loc0 >= nint, nd, lane, g0);
                if (loc1 < nbt) p_gh_load<NX, NU, MD>(C, Sy, L, p_slot_node<NX, NU, MD>(loc1, l0, s, C), loc1, !is_bottom && loc1 >= nint, nd, lane, g1);
                double v = p_gh_compute<NX, NU, MD>(L, loc0, lane, g0, O.termCondition, build);
                err = (O.termCondition == 2) ? nanmax(err, v) : err + v;
                if (loc1 < nbt) {
                    v = p_gh_compute<NX, NU, MD>(L, loc1, lane, g1, O.termCondition, build);
                    err = (O.termCondition == 2) ? nanmax(err, v) : err + v;
                }
            }
            if (lane == 0) L.part[4 * wave] = err;
            __syncthreads();
            if (!is_bottom && *L.abort) break;
            if (threadIdx.x == 0) {
                /* termination partial of my blocks to the top workgroup */
                err = 0.0;
                for (int w = 0; w < FW; w++) { const double v = L.part[4 * w]; err = (O.termCondition == 2) ? nanmax(err, v) : err + v; }
                pst_tag(Sy, Sy.errs + (size_t)wg * 2, err, tag_e);
            }
        }
        pstamp(C, O, e, tier, s, sl++);                                   /* 1: G+H done */

        if (is_top) {
            /* ---- verdicts: the outstanding {fval, dot} reduction (fval0 of the first sweep, or the first
             * trial of the previous iteration), then the termination test of the (then current) point ---- */
            double fa, da, ea;
            const bool open_trial = nd > 0u && !decided;
            int leave = p_gather3<NX, NU, MD>(Sy, L, Gm.G, open_trial, Sy.seq | nd, tag_e, O.termCondition == 2, fa, da, ea) ? 0 : 1;
            if (!leave) {
                if (threadIdx.x == 0) {
                    int code = 0;
                    if (decided) code = c->done ? 1 : 0;                        /* e.g. the iteration limit, reached inside a batch */
                    if (open_trial) {
                        if (prologue && nd == 1u) { c->fval0 = fa; c->fval = fa; }
                        else {
                            bool bad = false;
                            if (!c->ls_pending) {                               /* first trial of a line search */
                                c->tau = 1.0; c->tauPrev = 0.0; c->ls_iter = 1; c->ls_pending = 1;
                                bad = ls_not_descent(c, -da);
                            }
                            if (bad) code = 1;
                            else {
                                const PDump *dp = C.dump;
                                ls_decide_tail(c, dp->ls_log, dp->ls_log_cap, O, fa);
                                code = c->done ? 1 : (c->ls_pending ? 2 : 0);   /* finished / more trials / accepted */
                                if (code == 2) {                                /* line_search :985-987: lambda moves by (tau - tauPrev) dlambda */
                                    pst_tag(Sy, Sy.cmd, __dsub_rn(c->tau, c->tauPrev), tag_e); pst_tag(Sy, Sy.cmd + 2, c->tau, tag_e);
                                }
                            }
                        }
                    }
                    if (!code) {
                        if (O.termCondition == 1) ea = sqrt(ea);
                        c->err = ea;
                        if (ea < O.tol) { c->status = 0; c->done = 1; code = 1; }
                        else if (e >= 60000u || nbat >= 60000u) code = 1;       /* tags carry 16 bits of sequence: relaunch */
                    }
                    *L.flag = code;
                }
                __syncthreads();
                leave = *L.flag;
                __syncthreads();
            }
            if (leave == 1) {
                if (threadIdx.x == 0) pst_word(Sy, Sy.halt, Sy.seq);
                break;
            }
            verdict = leave;
        }
        decided = false;
        pstamp(C, O, e, tier, s, sl++);                                   /* 2: verdicts (top) */

        /* ---- backward sweep ---- */
        bool gone = false;
        double dotp = 0.0;                                /* per-lane terms of res' * dlam over my blocks */
        int prep_next = nbt - 1;                          /* blocks >= prep_next + 1 have their forward step prepared (descending order) */
        if (verdict != 2) {
            double Tc[D];
#ifdef TQ_CHAIN_BARRIERS
            constexpr bool CHAIN = false;
#else
            /* a chain tier (MD == 1) has ONE block per level: wave 0 walks down the chain on its own -- no workgroup barrier between
             * the levels (330 cycles of a ~2 400-cycle level) -- and the forward preparation of all its blocks is done by the four
             * waves after the sweep (it is needed only when the parent's step arrives) */
            constexpr bool CHAIN = MD == 1 && !RU;
#endif
            for (int t = th - 1; t >= 0; t--) {
                const int nb = U::width(t);
                /* waves without a block on this level prepare the forward steps of the levels below (at most two each) */
                const int idle = FW - nb, lo = U::first(t + 1);
                if (!CHAIN && wave >= nb && build) {
                    for (int r = 0; r < 2; r++) {
                        const int loc = prep_next - (r * idle + (wave - nb));
                        if (loc >= lo) p_prep_forward<NX, NU, MD>(L, loc, lane);
                    }
                }
                if (!CHAIN) { const int avail = prep_next - lo + 1, take = avail < 2 * idle ? avail : 2 * idle; prep_next -= take > 0 ? take : 0; }
                if (RU && wave < nb && !build) {
                    const int loc = U::first(t) + wave;
                    const double pd = p_reuse_block<NX, NU, MD>(C, Sy, L, loc, p_slot_node<NX, NU, MD>(loc, l0, s, C), t, th, is_bottom, is_top && t == 0, tag_e, lane, L.wave);
                    if (is_top && t == 0) dotp = pd;
                }
                if (wave < nb && build) {
                    const int loc = U::first(t) + wave, ii = p_slot_node<NX, NU, MD>(loc, l0, s, C);
                    const bool is_root = is_top && t == 0;
                    bool ok = true;
                    auto assemble = [&]() {                           /* the block's rows (my own children have subtracted their Schur records in place) minus the records of the tier below */
                        p_load_rows<NX, NU, MD>(L, loc, lane, Tc);
                        if (t == th - 1 && !is_bottom) {
                            ok = p_sub_children_tagged<NX, NU, MD>(Sy, Sy.sch + (size_t)kid0g<MD>(ii, C) * U::SCH * 2, tag_e, lane, Tc);
                            ok = __all(ok);
                            if (!ok && lane == 0) *L.abort = p_abort_code(Sy);
                        }
                    };
#ifdef TQ_FINE_STAMPS      /* one interior level (t == 1) of the stamped workgroup: slots 20.. = start, rows assembled, factorised, stored, Schur record posted */
#define FSTAMP(i) do { if (t == 1 && wave == 0) pstamp(C, O, e, tier, s, 20 + (i)); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif
                    FSTAMP(0);
                    assemble();
                    FSTAMP(1);
                    if (p_factor_rows_first<NX, NU, MD>(O, lane, Tc)) {
                        assemble();                                   /* rare: shift and refactorise (the rows are still in LDS) */
                        p_refactor_rows<NX, NU, MD>(cg, O, lane, Tc);
                    }
                    FSTAMP(2);
                    if (!is_root) {
                        p_store_factor<NX, NU, MD>(L, loc, lane, Tc);
                        FSTAMP(3);
                        if (t == 0) { if (ok) p_schur<NX, NU, MD, true, RU>(L, loc, lane, 0, 0, Sy.sch + (size_t)ii * U::SCH * 2, tag_e, Sy); }
                        else p_schur<NX, NU, MD, false>(L, loc, lane, U::first(t - 1) + wave / MD, wave % MD, nullptr, 0u, Sy);
                        FSTAMP(4);
                    } else {
                        if (RU) p_store_factor<NX, NU, MD>(L, loc, lane, Tc);          /* a later pass may keep the root's factor as well */
                        /* root: dlam_0 = L^-T (L^-1 res) = (L^-1)' y -- lane R + i holds column i of L^-1 (the identity rows of the
                         * factorisation), lane D holds y: D independent multiply-adds per lane, no substitution chain */
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                        for (int k = 0; k < D; k += 4) {
                            a0 = fma(Tc[k], rdlane(Tc[k], D), a0);
                            a1 = fma(Tc[k + 1], rdlane(Tc[k + 1], D), a1);
                            a2 = fma(Tc[k + 2], rdlane(Tc[k + 2], D), a2);
                            a3 = fma(Tc[k + 3], rdlane(Tc[k + 3], D), a3);
                        }
                        const double mine = (a0 + a1) + (a2 + a3);
                        const int ri = lane - U::R;
                        if (ri >= 0 && ri < D) {
                            if (th == 1 && !is_bottom && ok) pst_tag(Sy, Sy.dlt + (size_t)(NX * kid0g<MD>(0, C) + ri) * 2, mine, tag_e);
                            L.dl[ri] = mine; dotp = L.res[ri] * mine;
                        }
                        lds_fence();
                    }
                }
                if (CHAIN && t < th - 1) lds_fence(); else lds_barrier();
                if (t == th - 1 && !is_bottom && *L.abort) { gone = true; break; }    /* a child never delivered: the launch is over, or the pass is dropped */
                pstamp(C, O, e, tier, s, sl++);                           /* 3.. : one per backward level */
            }
            if (CHAIN) lds_barrier();                                     /* the chain is factorised: the other waves may read it */
        }
        if (gone) { if (*L.abort == 2) { verdict = 2; gone = false; } else break; }
        pstamp(C, O, e, tier, s, sl++);                                   /* backward done */
        if (verdict != 2) {
            /* what is left to prepare (the subtree root, which was factorised last; the root block of the tree needs none) */
            if (build) for (int loc = prep_next - wave; loc >= (is_top ? 1 : 0); loc -= FW) p_prep_forward<NX, NU, MD>(L, loc, lane);
            if (RU && build && !gone && wave == 0) {
                /* a complete build: these factor data belong to the signatures of this pass */
                for (int i = lane; i < nbt * (1 + MD); i += WAVE) L.bsig[i] = L.csig[i];
                if (lane == 0) *L.fvalid = 1;
            }
            lds_barrier();
            if (anc_pub && !gone) {
                /* my blocks' forward records to the tier below (it walks down its path through them itself, see p_forward_tier): off the
                 * critical path -- the tiers above me have their whole backward sweep to do before my step arrives */
                constexpr int REC = D * PLds<NX, NU, MD>::LDM;
                for (int i = threadIdx.x; i < nbt * REC; i += FW * WAVE) {
                    const int loc = i / REC, wq = i - loc * REC;
                    u64 *dst = Sy.anc + ((size_t)p_slot_node<NX, NU, MD>(loc, l0, s, C) * REC + wq) * 2;
                    if (Sy.anc_local) st_tag(dst, L.mz_(loc)[wq], tag_e);          /* (16 KB per workgroup and pass: not over xGMI to ranks that never read them) */
                    else pst_tag(Sy, dst, L.mz_(loc)[wq], tag_e);
                }
            }
        }

        /* ---- forward sweep of my subtree, one barrier at its end (the subtree root first waits for the parent workgroup's step) ---- */
        if (verdict != 2) {
            dotp += p_forward_tier<NX, NU, MD, ANC && !RU>(C, Sy, L, l0, s, th, is_top ? 1 : 0, wave, lane, !is_top, !is_bottom && !anc_pub, tag_e, anc_up);
            lds_barrier();
            if (!is_top && *L.abort) gone = true;                          /* the parent never delivered: the launch is over, or the pass is dropped */
            pstamp(C, O, e, tier, s, sl++);
        }
        if (gone) { if (*L.abort == 2) { verdict = 2; gone = false; } else break; }
        double step = 1.0;
        PChain ch{1, 1.0, O.beta, true};
        if (verdict == 2) {
            /* The trial this pass was built on was rejected: drop the pass (it only touched per-iteration LDS data and
             * hand-over words tagged with this pass) and finish the line search.  A trial needs a reduction over all
             * workgroups, i.e. a round trip through the top workgroup; so the trials are taken in BATCHES: every workgroup
             * evaluates the next K points of the backtracking chain (dry sweeps: nothing is stored) and posts K partials,
             * the top workgroup applies the Armijo test to them in order and answers with the length of the accepted
             * chain (or 0: next batch).  One more sweep then moves the state to the accepted point. */
            double cv[2];
            bool okc = p_read_top<2>(Sy, Sy.cmd, tag_e, cv);
            okc = __all(okc);
            if (!okc && lane == 0) *L.abort = 1;
            __syncthreads();
            if (*L.abort == 1) break;
            __syncthreads();
            if (threadIdx.x == 0) *L.abort = 0;
            __syncthreads();
            step = cv[0]; ch.tau0 = cv[1];
            int n0 = 0, nacc = 0;
            bool dead = false;
            for (;;) {
                const int K = n0 == 0 ? 4 : 8;
                nbat += 1u;
                const unsigned tag_b = Sy.seq | nbat;
                for (int k = 0; k < K; k++) {
                    const double f = p_stage_owned<NX, NU, MD, RU>(C, Sy, L, l0, nown, s, wave, lane, step, cur, false, false, 0u, PChain{n0 + k + 1, cv[1], O.beta}, true);
                    if (lane == 0) L.bat[wave * 8 + k] = f;
                }
                __syncthreads();
                if ((int)threadIdx.x < K) {
                    double f = 0.0;
                    for (int w = 0; w < FW; w++) f += L.bat[w * 8 + threadIdx.x];
                    pst_tag(Sy, Sy.bparts + ((size_t)wg * 8 + threadIdx.x) * 2, f, tag_b);
                }
                if (is_top) {
                    double fa[8];
                    const bool okg = p_gather_batch<NX, NU, MD>(Sy, L, Gm.G, K, tag_b, fa);
                    if (okg && threadIdx.x == 0) {
                        const PDump *dp = C.dump;
                        int acc = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            if (k < K && !acc) {
                                ls_decide_tail(c, dp->ls_log, dp->ls_log_cap, O, fa[k]);
                                if (c->done || !c->ls_pending) acc = n0 + k + 1;
                            }
                        }
                        if (acc) c->cur = cur ^ 1;                  /* the control block flips per trial, the buffers once per stored sweep */
                        pst_tag(Sy, Sy.vrd, (double)acc, tag_b);
                    }
                }
                double vv[1];
                bool okv = p_read_top<1>(Sy, Sy.vrd, tag_b, vv);
                okv = __all(okv);
                if (!okv && lane == 0) *L.abort = 1;
                __syncthreads();
                if (*L.abort) { dead = true; break; }
                nacc = (int)vv[0];
                if (nacc) break;
                n0 += K;
            }
            if (dead) {
                if (is_top && threadIdx.x == 0) pst_word(Sy, Sy.halt, Sy.seq);
                break;
            }
            ch.n = nacc; ch.save_s = false;
            decided = true;
        } else {
            have_dl = true;
            dotp = wsum(dotp);
            if (lane == 0) L.part[4 * wave + 1] = dotp;
        }
        pstamp(C, O, e, tier, s, sl++);                                   /* forward done */

        /* ---- next trial (after a full pass: the first one, tau = 1; after a batch: the accepted point) on the nodes this
         * workgroup owns; then straight on to the next pass at that point: the top workgroup checks that it was accepted ---- */
        const double fsum = p_stage_owned<NX, NU, MD, RU>(C, Sy, L, l0, nown, s, wave, lane, step, cur, false, !is_top, Sy.seq | (nd + 1u), ch, false);
        if (lane == 0) L.part[4 * wave + 2] = fsum;
        __syncthreads();
        nd += 1u;
        if (threadIdx.x == 0 && !decided) post_parts();
        cur ^= 1;
        pstamp(C, O, e, tier, s, sl++);                                   /* stage done */
    }

    /* ---- leaving: the state goes back to global memory.  When the top workgroup halts, every
     * workgroup has finished the same number of stage sweeps and none has started the next forward
     * sweep, so what LDS holds is the point the control block describes. ---- */
    __syncthreads();
    pstamp(C, O, (unsigned)O.stamps, tier, s, 28);                     /* 28: left the loop */
    {
        const PDump *dp = C.dump;
        if (is_top && wave == 0) {
            /* the verdict goes straight to the host: the control block (it lives in LDS) and the two clock readings as tagged words in
             * pinned memory, one store per lane, nothing to wait for (see HostRes) */
            static_assert(sizeof(Ctrl) == 96, "HostRes::tg holds the 24 halves of the control block");
            HostRes *hr = dp->hres;
#ifdef TQ_REUSE_DEBUG
            if (lane == 0) c->ls_last = c->pad0;
            lds_fence();
#endif
            const unsigned long long t_end = wall_clock64();
            const int li = lane < 24 ? lane : 0;
            unsigned piece = reinterpret_cast<const unsigned *>(c)[li];
            if (lane == 24) piece = (unsigned)t_start;
            if (lane == 25) piece = (unsigned)(t_start >> 32);
            if (lane == 26) piece = (unsigned)t_end;
            if (lane == 27) piece = (unsigned)(t_end >> 32);
            if (lane < HOSTRES_WORDS) __hip_atomic_store(hr->tg + lane, ((unsigned long long)Sy.seq << 32) | piece, RLX, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (is_top && threadIdx.x == 0) {
            /* for the next launch and for the stream-ordered readers (n_reg is counted in global memory by whoever regularises a block: it stays) */
            { const int keep = __hip_atomic_load(&cg->n_reg, RLX, AGENT); c->n_reg = keep; *cg = *c; }
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(c);
            if (Sy.npeer > 1) {
                /* sharded launch: the ranks that do not run this workgroup read the verdict from their own slab once their launch has ended */
                const size_t voff = (size_t)(Sy.verdict - Sy.base);
                for (int r = 0; r < Sy.npeer; r++) {
                    u64 *vq = Sy.peers[r] + voff;
                    for (int i = 0; i < (int)(sizeof(Ctrl) / 8); i++) __hip_atomic_store(vq + i, src[i], RLX, SYS);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int r = 0; r < Sy.npeer; r++) __hip_atomic_store(Sy.peers[r] + voff + 15, (u64)Sy.seq, RLX, SYS);
            }
            (void)src;
        }
        if (Sy.npeer > 1 && wg == Sy.relay_wg && threadIdx.x == 0) {
            /* a rank that does not run the top workgroup: the verdict arrives in this rank's slab (pushed by the top workgroup); one of its
             * workgroups passes it on to the rank's own host, which polls its result block as on a single device */
            const u64 t0v = wall_clock64();
            bool got = true;
            while ((unsigned)__hip_atomic_load(Sy.verdict + 15, RLX, TQ_LD_SCOPE) != Sy.seq) {
                if (wall_clock64() - t0v > 50000000ull) { got = false; break; }      /* 0.5 s: the top workgroup's rank is gone */
                __builtin_amdgcn_s_sleep(4);
            }
            if (got) {
                HostRes *hr = dp->hres;
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(&hr->c);
                for (int i = 0; i < (int)(sizeof(Ctrl) / 8); i++) { const u64 wv = __hip_atomic_load(Sy.verdict + i, RLX, TQ_LD_SCOPE); __hip_atomic_store(dst + i, wv, RLX, SYS); reinterpret_cast<unsigned long long *>(C.ctrl)[i] = wv; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&hr->seq, Sy.seq, RLX, SYS);
            }
        }
        pstamp(C, O, (unsigned)O.stamps, tier, s, 29);                 /* 29: verdict on the host's way */
        if (nd > 0u) {
            for (int i = threadIdx.x; i < nown * 16; i += FW * WAVE) {
                const int q = i >> 4, t = i & 15;
                const int k = p_slot_node<NX, NU, MD>(q, l0, s, C);
                lds_cptr ns = L.node_(q);
                if (t < NX) {
                    const int o = NX * k + t;
                    dp->x[o] = ns[t]; dp->QinvCal[o] = ns[NZ + t]; dp->xUnc[o] = ns[2 * NZ + t]; dp->qmod[o] = ns[3 * NZ + t]; dp->xUncS[o] = ns[4 * NZ + t];
                } else if (t < NZ && k < C.Np) {
                    const int o = NU * k + t - NX;
                    dp->u[o] = ns[t]; dp->RinvCal[o] = ns[NZ + t]; dp->uUnc[o] = ns[2 * NZ + t]; dp->rmod[o] = ns[3 * NZ + t]; dp->uUncS[o] = ns[4 * NZ + t];
                }
            }
            double *ldst = cur ? dp->lam1 : dp->lam0;
            for (int i = threadIdx.x; i < nbt * D; i += FW * WAVE) {
                const int loc = i / D, t = i - loc * D;
                ldst[NX * kid0g<MD>(p_slot_node<NX, NU, MD>(loc, l0, s, C), C) + t] = L.lamb_(cur, loc)[t];
            }
        }
        if (have_dl) {
            for (int i = threadIdx.x; i < nbt * D; i += FW * WAVE) {
                const int loc = i / D, t = i - loc * D;
                dp->dlam[NX * kid0g<MD>(p_slot_node<NX, NU, MD>(loc, l0, s, C), C) + t] = L.dl[loc * D + t];
            }
        }
    }
    pstamp(C, O, (unsigned)O.stamps, tier, s, 30);                     /* 30: state written back */
}

#ifndef TQ_WPS
#define TQ_WPS 2
#endif
/* the workgroup `b` (position in the launch's own numbering) of one tree: finds its tier and role and runs its life */
template <int NX, int NU, int MD, bool RU>
__device__ __forceinline__ void persist_entry(const PConst &C, const Opts &O, const PGeom &Gm, const PSync &Sy, int prologue, int b, double *lds_all) {
    const int wg = Gm.wg_of_block[b];
    int tier = 0;
    for (int i = 0; i < Gm.n_tiers; i++) if (wg >= Gm.wg0[i]) tier = i;
    /* one instantiation of the pass loop per ROLE: each carries only its own branches and live state (the common one was at the
     * register limit with ~210 spilled scalars; per role the C2 solve is 10 % faster).  Bottom and middle tiers of a uniform tree
     * always have Uni::TH levels. */
    if (!RU && Gm.n_tiers > 1) {
        if (tier == 0) p_run<NX, NU, MD, RU, 1, true, true>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        else if (tier == Gm.n_tiers - 1) p_run<NX, NU, MD, RU, 3>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        else p_run<NX, NU, MD, RU, 2, true, true>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        return;
    }
    p_run<NX, NU, MD, RU>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
}

template <int NX, int NU, int MD, bool RU>
__global__ void __launch_bounds__(FW * WAVE, TQ_WPS) f_persist(PConst C, Opts O, PGeom Gm, PSync Sy, int prologue)
#if !TQ_HAS(TQP_PERSIST)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Sy.npeer = 1; Sy.relay_wg = -1; Sy.anc_local = 1;          /* one device: a compile-time fact here, so that the stores to peer slabs and the verdict relay fold away (left as run-time tests they cost 18 us per C2 solve: scalar registers) */
    persist_entry<NX, NU, MD, RU>(C, O, Gm, Sy, prologue, (int)blockIdx.x, lds_all);
}
#endif
/* The same kernel compiled for ONE workgroup per CU (launch bounds: the compiler may take more than 256 registers -- 264 with the scalar
 * spills' lanes for C2, against 244): 94.2 -> 92.5 us per C2 solve, 56.5 -> 55.6 on C1.  The host takes it when the launch fits the
 * device at one workgroup per CU (C2: 73, C1: 10; not C3's 293, not a batch). */
template <int NX, int NU, int MD>
__global__ void __launch_bounds__(FW * WAVE, 1) f_persist_one(PConst C, Opts O, PGeom Gm, PSync Sy, int prologue)
#if !TQ_HAS(TQP_PERSIST)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Sy.npeer = 1; Sy.relay_wg = -1; Sy.anc_local = 1;
    persist_entry<NX, NU, MD, false>(C, O, Gm, Sy, prologue, (int)blockIdx.x, lds_all);
}
#endif
/* the same launch as one rank's share of a sharded solve (tqgpu_pshard_*): hand-over words go to every rank's slab */
/* SC only tells two builds of the same body apart: 1 = compiled in the part whose polls of the slab are system-scope loads (TQ_LD_SCOPE, tdunes_parts.hpp:
 * the slab is written by peers over xGMI), 0 = agent-scope polls as on one device (kept for A/B runs on a node: TREEQP_AMD_PSHARD_AGENT=1) */
template <int NX, int NU, int MD, int SC>
__global__ void __launch_bounds__(FW * WAVE, TQ_WPS) f_persist_sh(PConst C, Opts O, PGeom Gm, PSync Sy, int prologue)
#if !TQ_HAS(TQP_SHARD | TQP_SHARD_AG)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    persist_entry<NX, NU, MD, false>(C, O, Gm, Sy, prologue, (int)blockIdx.x, lds_all);
}
#endif

/* multistage trees (branching for Nr stages, then one child per node -- the reference's setup_multistage_tree
 * with Nr < Nh, its standard robust-horizon shape): the tiers of the branching part run as above, every scenario
 * chain below is one more tier subtree with ONE block per level (instantiation MD = 1: blocks of nx rows).  A
 * chain workgroup spreads G + H and the stage sweep over its four waves; its backward / forward sweeps are one
 * wave walking down the chain, which is what a chain is. */
template <int NX, int NU, int MD, bool RU>
__device__ __forceinline__ void mpersist_entry(const PConst &C, const Opts &O, const PGeom &Gm, const PSync &Sy, int prologue, int b, double *lds_all) {
    const int wg = Gm.wg_of_block[b];
    int tier = 0;
    for (int i = 0; i < Gm.n_tiers; i++) if (wg >= Gm.wg0[i]) tier = i;
    if (!RU) {             /* per role as in persist_entry; a multistage tree has two tiers or more (chains below the branching part) */
        if (Gm.chain[tier]) {
            if (tier == 0) p_run<NX, NU, 1, RU, 1>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
            else p_run<NX, NU, 1, RU, 2>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        } else if (tier == Gm.n_tiers - 1) p_run<NX, NU, MD, RU, 3>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        else p_run<NX, NU, MD, RU, 2>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
        return;
    }
    if (Gm.chain[tier]) p_run<NX, NU, 1, RU>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
    else p_run<NX, NU, MD, RU>(C, O, Gm, Sy, prologue, wg, tier, wg - Gm.wg0[tier], lds_all);
}

template <int NX, int NU, int MD, bool RU>
__global__ void __launch_bounds__(FW * WAVE, TQ_WPS) f_mpersist(PConst C, Opts O, PGeom Gm, PSync Sy, int prologue)
#if !TQ_HAS(TQP_PERSIST)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Sy.npeer = 1; Sy.relay_wg = -1; Sy.anc_local = 1;
    mpersist_entry<NX, NU, MD, RU>(C, O, Gm, Sy, prologue, (int)blockIdx.x, lds_all);
}
#endif

template <int NX, int NU, int MD>      /* (one workgroup per CU: see f_persist_one) */
__global__ void __launch_bounds__(FW * WAVE, 1) f_mpersist_one(PConst C, Opts O, PGeom Gm, PSync Sy, int prologue)
#if !TQ_HAS(TQP_PERSIST)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Sy.npeer = 1; Sy.relay_wg = -1; Sy.anc_local = 1;
    mpersist_entry<NX, NU, MD, false>(C, O, Gm, Sy, prologue, (int)blockIdx.x, lds_all);
}
#endif

/* A BATCH of independent trees of one shape as ONE launch (tqgpu_solve_batch): workgroups [t G, (t + 1) G) are tree t's, which
 * works from its own descriptors (constants, geometry, hand-over buffers, result block) in device memory and never looks at
 * another tree.  One launch per tree on a stream of its own only overlaps as many trees as the runtime has hardware queues
 * (C1: 22 trees reached 3.4 x one tree); one launch carries them all.  Every tree of the launch takes the same launch number
 * (its tags only have to be unique in the tree's own buffers).  Fresh solves only (prologue). */
struct PItem { PConst C; PGeom Gm; PSync Sy; };
template <int NX, int NU, int MD, bool MSTAGE>
__global__ void __launch_bounds__(FW * WAVE, TQ_WPS) f_persist_batch(const PItem *items, Opts O, int G, unsigned seq, int nap)
#if !TQ_HAS(TQP_BATCH)
;
#else
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int tree = (int)blockIdx.x / G, b = (int)blockIdx.x - tree * G;
    const PItem *it = items + tree;
    const PConst C = it->C;
    const PGeom Gm = it->Gm;
    PSync Sy = it->Sy;
    Sy.seq = seq;
    Sy.nap = nap;
    Sy.npeer = 1; Sy.relay_wg = -1; Sy.anc_local = 1;
    if (MSTAGE) mpersist_entry<NX, NU, MD, false>(C, O, Gm, Sy, 1, b, lds_all);
    else persist_entry<NX, NU, MD, false>(C, O, Gm, Sy, 1, b, lds_all);
}
#endif

/* packed constants of the persistent path (run whenever the QP data changed): [A | B] per edge and
 * {linear term, 1/weight, weight, lower, upper} per node entry */
#if TQ_HAS(TQP_HOST)
__global__ void k_pack_persist(int Nn, int Np, int NX, int NU, Data D, double *AB, double *cst) {
    const int NZ = NX + NU;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < Nn * 16) {
        const int k = tid >> 4, t = tid & 15;
        double *o = cst + (size_t)tid * 5;
        if (t < NX) { const int i = NX * k + t; o[0] = D.q[i]; o[1] = 1.0 / D.Qd[i]; o[2] = D.Qd[i]; o[3] = D.xmin[i]; o[4] = D.xmax[i]; }
        else if (t < NZ && k < Np) { const int i = NU * k + t - NX; o[0] = D.r[i]; o[1] = 1.0 / D.Rd[i]; o[2] = D.Rd[i]; o[3] = D.umin[i]; o[4] = D.umax[i]; }
        else { o[0] = 0.0; o[1] = 0.0; o[2] = 0.0; o[3] = 0.0; o[4] = 0.0; }
    }
    const int ne = (Nn - 1) * NX * NZ;
    for (int i = tid; i < ne; i += gridDim.x * blockDim.x) {
        const int edge = i / (NX * NZ), w = i - edge * NX * NZ, col = w / NX, r = w - col * NX;
        AB[i] = col < NX ? D.A[(size_t)edge * NX * NX + (size_t)col * NX + r] : D.B[(size_t)edge * NX * NU + (size_t)(col - NX) * NX + r];
    }
}
#endif

/* the parts that own these families instantiate them (tdunes_parts.hpp; the persistent family in slices by table index) */
#if TQ_HAS(TQP_PERSIST)
#define X(idx, nx, nu, md) TQ_SL(idx, \
    template __global__ void f_persist<nx, nu, md, false>(PConst, Opts, PGeom, PSync, int); \
    template __global__ void f_persist_one<nx, nu, md>(PConst, Opts, PGeom, PSync, int); \
    template __global__ void f_persist<nx, nu, md, true>(PConst, Opts, PGeom, PSync, int);)
FAST_TABLE(X)
#undef X
#define X(idx, nx, nu, md) TQ_SL(idx, \
    template __global__ void f_mpersist<nx, nu, md, false>(PConst, Opts, PGeom, PSync, int); \
    template __global__ void f_mpersist_one<nx, nu, md>(PConst, Opts, PGeom, PSync, int); \
    template __global__ void f_mpersist<nx, nu, md, true>(PConst, Opts, PGeom, PSync, int);)
MSTAGE_TABLE(X)
#undef X
#endif
#if TQ_HAS(TQP_SHARD)
#define X(idx, nx, nu, md) template __global__ void f_persist_sh<nx, nu, md, 1>(PConst, Opts, PGeom, PSync, int);
SHARD_TABLE(X)
#undef X
#endif
#if TQ_HAS(TQP_SHARD_AG)
#define X(idx, nx, nu, md) template __global__ void f_persist_sh<nx, nu, md, 0>(PConst, Opts, PGeom, PSync, int);
SHARD_TABLE(X)
#undef X
#endif
#if TQ_HAS(TQP_BATCH)
#define X(idx, nx, nu, md, ms) template __global__ void f_persist_batch<nx, nu, md, ms>(const PItem *, Opts, int, unsigned, int);
BATCH_TABLE(X)
#undef X
#endif
