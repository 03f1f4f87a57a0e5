// Microbenchmark: the backward level step of the persistent kernel (one 4-wave workgroup, LDS resident),
// with parts switched off by a mask to attribute the cycles.  Numerics are irrelevant here.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I treeqp_amd/csrc/device -o level_bench tools/microbench/level_bench.hip
#include "../../treeqp_amd/csrc/device/tdunes_device.hip"

namespace {
// MASK bits: 1 load_rows, 4 factor (fast pass as the kernel runs it), 8 store_factor, 16 schur (in-place update of the parent), 32 barrier,
// 64 p_potrf_rows only, 128 bare potrf, 256 forward preparation by the idle waves
template <int MASK>
__global__ void __launch_bounds__(FW * WAVE) level_bench(Ctrl *c, Opts O, long long *cycles, int reps, double *sink) {
    constexpr int NX = 8, NU = 3, MD = 2;
    using U = Uni<NX, NU, MD>;
    using PL = PLds<NX, NU, MD>;
    constexpr int D = U::D;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    PL L(lds_all, wave);
    for (int i = threadIdx.x; i < PL::DOUBLES; i += FW * WAVE) lds_all[i] = 0.01 * ((i * 7) % 13);
    __syncthreads();
    for (int b = 0; b < U::NBT; b++)
        if (threadIdx.x < D) L.tt_(b)[threadIdx.x * PL::S + threadIdx.x] = 50.0;
    for (int i = threadIdx.x; i < D * PL::S; i += FW * WAVE) { const int j = i / PL::S, m = i - j * PL::S; L.idt[i] = (m == j) ? 1.0 : 0.0; }
    __syncthreads();
    const int th = 3;
    double Tc[D];
#pragma unroll
    for (int j = 0; j < D; j++) Tc[j] = (lane == j) ? 40.0 : 0.01 * j;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        for (int t = th - 1; t >= 0; t--) {
            const int nb = U::width(t);
            if (wave >= nb && (MASK & 256)) p_prep_forward<NX, NU, MD>(L, U::first(t + 1) + (wave - nb) % U::width(t + 1), lane);
            if (wave < nb) {
                const int loc = U::first(t) + wave;
                if (MASK & 1) p_load_rows<NX, NU, MD>(L, loc, lane, Tc);
                if (MASK & 4) { if (p_factor_rows_first<NX, NU, MD>(O, lane, Tc)) sink[2] = 1.0; }
                if (MASK & 64) { double pm = p_potrf_rows<D>(Tc, lane); if (pm == 1.2345) sink[1] = pm; }
                if (MASK & 128) {      /* bare left-looking factorisation, nothing stored */
#pragma unroll
                    for (int j = 0; j < D; j++) {
                        double s = Tc[j];
#pragma unroll
                        for (int k = 0; k < j; k++) s = fma(-Tc[k], rdlane(Tc[k], j), s);
                        const double pj = rdlane(s, j);
                        Tc[j] = s * pivot_rsqrt3(pj);
                    }
                }
                if (MASK & 8) p_store_factor<NX, NU, MD>(L, loc, lane, Tc);
                if ((MASK & 16) && t > 0) { PSync nosy{}; p_schur<NX, NU, MD, false>(L, loc, lane, U::first(t - 1) + wave / MD, wave % MD, nullptr, 0u, nosy); }
            }
            if (MASK & 32) lds_barrier();
        }
        /* keep the blocks factorisable: the in-place updates of the parents are undone */
        for (int b = 0; b < U::NBT; b++)
            if (threadIdx.x < D) L.tt_(b)[threadIdx.x * PL::S + threadIdx.x] = 50.0;
        if (!(MASK & 1)) {
#pragma unroll
            for (int j = 0; j < D; j++) Tc[j] = (lane == j) ? 40.0 + Tc[j] * 1e-300 : 0.01 * j;
        }
    }
    long long t1 = clock64();
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) acc += Tc[j];
    if (threadIdx.x == 0) cycles[0] = (t1 - t0) / (reps * th);
    if (acc == 1.2345) sink[0] = acc;
}
// forward sweep of one tier (3 levels, no barrier between them, one at the end): cycles per tier
template <int VARIANT>
__global__ void __launch_bounds__(FW * WAVE) fwd_bench(long long *cycles, int reps, double *sink, PSync Sy, PConst C) {
    constexpr int NX = 8, NU = 3, MD = 2;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    PLds<NX, NU, MD> L(lds_all, wave);
    for (int i = threadIdx.x; i < PLds<NX, NU, MD>::DOUBLES; i += FW * WAVE) lds_all[i] = 0.01 * ((i * 7) % 13) + 0.5;
    __syncthreads();
    double dotp = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        dotp += p_forward_tier<NX, NU, MD>(C, Sy, L, 0, 0, 3, 1, wave, lane, false, false, 0u);      /* top-tier flavour: no poll */
        lds_barrier();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cycles[0] = (t1 - t0) / reps;
    if (dotp == 1.2345) sink[0] = dotp;
}
// stage sweep (15 owned nodes of a bottom-tier workgroup) and G + H (7 blocks) from LDS state + global constants
template <int WHAT>
__global__ void __launch_bounds__(FW * WAVE) sg_bench(PConst C, Opts O, PSync Sy, long long *cycles, int reps, double *sink) {
    constexpr int NX = 8, NU = 3, MD = 2;
    using U = Uni<NX, NU, MD>;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    PLds<NX, NU, MD> L(lds_all, wave);
    for (int i = threadIdx.x; i < PLds<NX, NU, MD>::DOUBLES; i += FW * WAVE) lds_all[i] = 0.01 * ((i * 7) % 13) + 0.5;
    __syncthreads();
    const int l0 = 6, s = 0, th = 3, nbt = 7, nown = 15;
    double acc = 0.0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        if (WHAT == 0) {
            acc += p_stage_owned<NX, NU, MD, false>(C, Sy, L, l0, nown, s, wave, lane, 1.0, r & 1, false, false, 0u, PChain{1, 1.0, 0.6}, false);
            __syncthreads();
        } else {
            for (int loc0 = wave; loc0 < nbt; loc0 += 2 * FW) {
                const int loc1 = loc0 + FW;
                GhRegs<NX, NU, MD> g0, g1;
                p_gh_load<NX, NU, MD>(C, Sy, L, p_slot_node<NX, NU, MD>(loc0, l0, s, C), loc0, false, 1u, lane, g0);
                if (loc1 < nbt) p_gh_load<NX, NU, MD>(C, Sy, L, p_slot_node<NX, NU, MD>(loc1, l0, s, C), loc1, false, 1u, lane, g1);
                acc += p_gh_compute<NX, NU, MD>(L, loc0, lane, g0, O.termCondition, true);
                if (loc1 < nbt) acc += p_gh_compute<NX, NU, MD>(L, loc1, lane, g1, O.termCondition, true);
            }
            __syncthreads();
        }
        (void)th;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cycles[0] = (t1 - t0) / reps;
    if (acc == 1.2345) sink[0] = acc;
}
}  // namespace

template <int MASK>
static void run(const char *name, Ctrl *c, Opts O, long long *dc, double *sink) {
    const size_t lds = PLds<8, 3, 2>::DOUBLES * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(level_bench<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(level_bench<MASK>, dim3(1), dim3(FW * WAVE), lds, 0, c, O, dc, 300, sink); (void)hipDeviceSynchronize(); }
    long long h = 0;
    (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %6lld cycles\n", name, h);
}

int main() {
    Ctrl *c; long long *dc; double *sink;
    (void)hipMalloc(&c, sizeof(Ctrl)); (void)hipMemset(c, 0, sizeof(Ctrl)); (void)hipMalloc(&dc, 64); (void)hipMalloc(&sink, 64);
    Opts O; memset(&O, 0, sizeof(O));
    O.regType = 2; O.regTol = 1e-6; O.regValue = 1e-6; O.termCondition = 2;
    run<1 + 4 + 8 + 16 + 32>("backward level (load, factor, store, schur, barrier)", c, O, dc, sink);
    run<1 + 4 + 8 + 16 + 32 + 256>("backward level with idle waves preparing", c, O, dc, sink);
    run<1 + 4 + 8 + 16>("no barrier", c, O, dc, sink);
    run<1 + 4 + 8 + 32>("no schur", c, O, dc, sink);
    run<1 + 4 + 16 + 32>("no store_factor", c, O, dc, sink);
    run<4 + 8 + 16 + 32>("no load_rows", c, O, dc, sink);
    run<4>("factor only (p_factor_rows_first)", c, O, dc, sink);
    run<1 + 8 + 16 + 32>("everything but factor", c, O, dc, sink);
    run<64>("p_potrf_rows only", c, O, dc, sink);
    run<128>("bare potrf (no pmin)", c, O, dc, sink);
    {
        /* constants of a 1023-node tree, arbitrary finite values */
        const int Nn = 1023, NX = 8, NZ = 11;
        std::vector<double> hab((size_t)(Nn - 1) * NX * NZ), hb((size_t)Nn * NX), hc((size_t)Nn * 16 * 5);
        for (size_t i = 0; i < hab.size(); i++) hab[i] = 0.01 * (double)((i * 13) % 17);
        for (size_t i = 0; i < hb.size(); i++) hb[i] = 0.1;
        for (size_t i = 0; i < hc.size(); i += 5) { hc[i] = 0.1; hc[i + 1] = 0.5; hc[i + 2] = 2.0; hc[i + 3] = -0.5; hc[i + 4] = 0.5; }
        double *dab, *db, *dcst;
        (void)hipMalloc(&dab, hab.size() * 8); (void)hipMalloc(&db, hb.size() * 8); (void)hipMalloc(&dcst, hc.size() * 8);
        (void)hipMemcpy(dab, hab.data(), hab.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dcst, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
        PConst C; memset(&C, 0, sizeof(C)); C.AB = dab; C.b = db; C.cst = dcst; C.ctrl = c; C.Np = 511;
        PSync Sy; memset(&Sy, 0, sizeof(Sy));
        const size_t lds = PLds<8, 3, 2>::DOUBLES * sizeof(double);
        long long h = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sg_bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sg_bench<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(sg_bench<0>, dim3(1), dim3(FW * WAVE), lds, 0, C, O, Sy, dc, 300, sink); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
        printf("%-52s %6lld cycles\n", "stage sweep, 15 nodes (+ barrier)", h);
        for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(sg_bench<1>, dim3(1), dim3(FW * WAVE), lds, 0, C, O, Sy, dc, 300, sink); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
        printf("%-52s %6lld cycles\n", "G + H, 7 blocks (+ barrier)", h);
    }
    {
        PSync Sy; memset(&Sy, 0, sizeof(Sy));
        const size_t lds = PLds<8, 3, 2>::DOUBLES * sizeof(double);
        PConst Cf; memset(&Cf, 0, sizeof(Cf));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(fwd_bench<0>, dim3(1), dim3(FW * WAVE), lds, 0, dc, 300, sink, Sy, Cf); (void)hipDeviceSynchronize(); }
        long long h = 0;
        (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
        printf("%-52s %6lld cycles\n", "forward sweep of a tier (3 levels, one barrier)", h);
    }
    return 0;
}
