// Micro-benchmark: sustained FP64 VALU rate on gfx950 for add / mul / fma / division / sqrt /
// 32-bit integer ops, at a chosen number of independent chains per lane and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int OP, int ILP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double v[ILP];
    unsigned u[ILP];
    for (int i = 0; i < ILP; ++i) { v[i] = a + threadIdx.x * 1e-9 + i; u[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) v[i] = v[i] + b;
            if (OP == 1) v[i] = v[i] * b;
            if (OP == 2) v[i] = __builtin_fma(v[i], b, a);
            if (OP == 3) v[i] = a / v[i];
            if (OP == 4) v[i] = sqrt(v[i]) + a;
            if (OP == 5) u[i] = (u[i] ^ (u[i] << 9)) + 12345u;
            if (OP == 6) { double t = v[i] * b; v[i] = (t > a) ? t : v[i]; }
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += v[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int ILP>
void run(const char* name, int ops_per_iter_elem, int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu, iters = (OP == 3 || OP == 4) ? 60000 : 400000;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<OP, ILP><<<blocks, 256>>>(out, 100, 1.000001, 0.999999);
    CK(hipEventRecord(e0));
    k<OP, ILP><<<blocks, 256>>>(out, iters, 1.000001, 0.999999);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = blocks * 4.0;
    double wave_instr = waves * (double)iters * ILP * ops_per_iter_elem;
    double per_simd_rate = wave_instr / 1024.0 / (ms * 1e-3);  // wave-instr per second per SIMD
    printf("%-10s ILP %d waves/SIMD %d : %8.3f ms  %.3f G wave-instr/s/SIMD -> %.2f cycles/wave-instr @2.4GHz, %.1f Tinstr-lanes/s\n",
           name, ILP, blocks_per_cu, ms, per_simd_rate * 1e-9, 2.4e9 / per_simd_rate, wave_instr * 64 / (ms * 1e-3) * 1e-12);
    CK(hipFree(out));
}

int main() {
    run<0, 4>("warmup", 1, 4);
    run<0, 1>("add", 1, 4); run<0, 2>("add", 1, 4); run<0, 4>("add", 1, 4); run<0, 8>("add", 1, 4);
    run<0, 1>("add", 1, 8); run<0, 4>("add", 1, 8);
    run<2, 1>("fma", 1, 4); run<2, 4>("fma", 1, 4); run<2, 8>("fma", 1, 4);
    run<3, 1>("div", 1, 4); run<3, 2>("div", 1, 4); run<3, 4>("div", 1, 4); run<3, 8>("div", 1, 4);
    run<3, 2>("div", 1, 8); run<3, 4>("div", 1, 8);
    run<4, 1>("sqrt+add", 1, 4); run<4, 4>("sqrt+add", 1, 4);
    run<5, 4>("int(3)", 3, 4); run<5, 4>("int(3)", 3, 8);
    run<6, 4>("mul+cmp+sel", 1, 4);
    return 0;
}
