// Micro-benchmark: the measured-BRDF gather on gfx950 — three adjacent doubles (24 B) at a random
// entry of a device-resident table — as a function of the table footprint: one, four (the
// estimator's 140 MB), eight and thirty-two tables of 3 x 1 458 000 doubles (35 MB each).  Footprints
// below the 256 MiB Infinity Cache are served on-die once warm; larger ones go to HBM.  The ratio
// of the rates is the evidence for where the 140 MB are served from: no rocprofv3 counter of gfx950
// separates Infinity-Cache hits from HBM reads (FETCH_SIZE counts fabric requests of both).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/table_gather.hip -o scripts/ubench/table_gather
//   scripts/ubench/table_gather [dependent]
// `dependent`: each lane's next index depends on the value it just read (latency chain, like the
// lookup's three binary searches); default: 8 independent gathers in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <bool DEP>
__global__ void __launch_bounds__(256) k_gather(const double* __restrict__ t, unsigned entries, int rounds, double* out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    unsigned h = mix(tid * 2654435761U + 12345U);
    for (int r = 0; r < rounds; ++r) {
        if (DEP) {
            const unsigned i = h % entries;
            const double a = t[3 * (size_t)i], b = t[3 * (size_t)i + 1], c = t[3 * (size_t)i + 2];
            s += a + b + c;
            h = mix(h + (unsigned)__double2loint(a));
        } else {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                h = mix(h + 0x9E3779B9U);
                const unsigned i = h % entries;
                acc += t[3 * (size_t)i] + t[3 * (size_t)i + 1] + t[3 * (size_t)i + 2];
            }
            s += acc;
        }
    }
    out[tid] = s;
}

int main(int argc, char** argv) {
    const bool dep = argc > 1 && !strcmp(argv[1], "dependent");
    const size_t table_entries = 90 * 90 * 180;  // MatusikBRDF.h:24-26
    const int tables[] = {1, 4, 8, 32};
    const int blocks = 256 * 16, threads = 256, rounds = dep ? 256 : 64;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%s gathers of 24 B, %d lanes x %d %s\n", dep ? "dependent" : "independent (8 in flight per lane)", blocks * threads,
           dep ? rounds : rounds * 8, "gathers per lane");
    for (int nt : tables) {
        const size_t entries = table_entries * nt, bytes = entries * 24;
        double* t;
        CK(hipMalloc(&t, bytes));
        CK(hipMemset(t, 0, bytes));
        for (int rep = 0; rep < 3; ++rep) {  // the first pass warms the caches
            CK(hipEventRecord(e0));
            if (dep) hipLaunchKernelGGL(k_gather<true>, dim3(blocks), dim3(threads), 0, 0, t, (unsigned)entries, rounds, out);
            else hipLaunchKernelGGL(k_gather<false>, dim3(blocks), dim3(threads), 0, 0, t, (unsigned)entries, rounds, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)blocks * threads * rounds * (dep ? 1 : 8);
            if (rep > 0)
                printf("  %2d tables %7.1f MB  pass %d: %8.3f ms  %.3g gathers/s  %.2f TB/s of 24 B  %.2f TB/s of 64-B sectors\n", nt, bytes / 1e6, rep,
                       ms, n / (ms * 1e-3), n * 24 / (ms * 1e-3) / 1e12, n * 64 * 1.25 / (ms * 1e-3) / 1e12);
        }
        CK(hipFree(t));
    }
    return 0;
}
