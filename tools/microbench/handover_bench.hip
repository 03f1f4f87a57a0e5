// Microbenchmark: latency of a tagged-word hand-over between two workgroups, by where they sit (same XCD / different XCDs; the hardware
// deals workgroup b to XCD b % 8) and by how the consumer polls (agent-scope atomic load = sc1: served from the memory side; sc0 only:
// served from the XCD's L2, coherent among the CUs of ONE XCD only).  The producer's store is the product's (relaxed agent-scope
// atomic store: written through).  Ping-pong of N round trips between workgroup 0 and workgroup `peer`; one thread each.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/handover_bench tools/microbench/handover_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define HIPCHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ u64 poll_load(const u64 *p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 v;
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ void pingpong(u64 *a, u64 *b, int peer, int n, long long *cycles, int *xcc) {
    const int wg = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (wg == 0 || wg == peer) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[wg == 0 ? 0 : 1] = (int)(id & 0xf);
    }
    if (wg == 0) {
        const long long t0 = clock64();
        for (int i = 1; i <= n; i++) {
            __hip_atomic_store(a, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long spins = 0;
            while (poll_load<MODE>(b) != (u64)i) { if (++spins > 4000000) { cycles[1] = -i; return; } }
        }
        cycles[0] = clock64() - t0;
    } else if (wg == peer) {
        for (int i = 1; i <= n; i++) {
            long long spins = 0;
            while (poll_load<MODE>(a) != (u64)i) { if (++spins > 4000000) return; }
            __hip_atomic_store(b, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
void run(const char *name, int peer, u64 *buf, long long *dc, int *dx) {
    const int n = 2000;
    HIPCHECK(hipMemset(buf, 0, 4096));
    HIPCHECK(hipMemset(dc, 0, 16));
    hipLaunchKernelGGL(pingpong<MODE>, dim3(64), dim3(64), 0, 0, buf, buf + 64, peer, n, dc, dx);
    HIPCHECK(hipDeviceSynchronize());
    long long c[2]; int x[2];
    HIPCHECK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(x, dx, 8, hipMemcpyDeviceToHost));
    if (c[1] < 0) printf("%-34s workgroups 0 <-> %2d (XCC %d / %d): GAVE UP at round %lld (the poll never saw the word)\n", name, peer, x[0], x[1], -c[1]);
    else printf("%-34s workgroups 0 <-> %2d (XCC %d / %d): %7.0f cycles per round trip (two hand-overs) = %5.2f us per hand-over at 2.4 GHz\n", name, peer, x[0], x[1], (double)c[0] / n, (double)c[0] / n / 2 / 2400.0);
}

int main() {
    u64 *buf; long long *dc; int *dx;
    HIPCHECK(hipMalloc(&buf, 4096)); HIPCHECK(hipMalloc(&dc, 16)); HIPCHECK(hipMalloc(&dx, 8));
    for (int peer : {8, 16, 1, 3}) {
        run<0>("agent-scope atomic load (product)", peer, buf, dc, dx);
        run<2>("load sc0 sc1", peer, buf, dc, dx);
        run<1>("load sc0 (L2 of the XCD)", peer, buf, dc, dx);
    }
    return 0;
}
