// Micro-benchmark: issue cost of f64 / f32 / int VALU instructions on one SIMD (development aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(64) k(double* out, int iters, double seed) {
    double a[8];
    float f[8];
    unsigned u[8];
    for (int j = 0; j < 8; ++j) { a[j] = seed + j + threadIdx.x; f[j] = (float)a[j]; u[j] = (unsigned)(j + threadIdx.x) | 1u; }
    const double b = seed * 1.0000001, c = seed * 0.3;
    const float fb = (float)b, fc = (float)c;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) a[j] = __builtin_fma(a[j], b, c);
            if (OP == 1) a[j] = a[j] * b;
            if (OP == 2) a[j] = a[j] + c;
            if (OP == 3) f[j] = __builtin_fmaf(f[j], fb, fc);
            if (OP == 4) { unsigned long long p = (unsigned long long)u[j] * 0xD2511F53u; u[j] = (unsigned)(p >> 32) ^ (unsigned)p; }
            if (OP == 5) a[j] = 1.0 / a[j];
            if (OP == 6) a[j] = sqrt(a[j]);
            if (OP == 7) a[j] = a[j] / b;
            if (OP == 8) u[j] = (u[j] ^ (u[j] >> 3)) + 0x9E3779B9u;
            if (OP == 9) a[j] = __builtin_fmin(a[j], b + a[(j + 1) & 7]);           // add + min
            if (OP == 10) a[j] = a[j] + (double)u[j];                                // cvt_f64_u32 + add
            if (OP == 11) a[j] = a[j] + (double)f[j];                                // cvt_f64_f32 + add
            if (OP == 12) a[j] = a[j] < b ? a[j] + c : a[j];                         // cmp + cndmask x2 + add
            if (OP == 13) f[j] = __builtin_fminf(f[j], fb + f[(j + 1) & 7]);         // f32 add + min
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j] + f[j] + u[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (double)(t1 - t0);
}

template <int OP>
void run(const char* name, int waves_per_simd, int ops_per_iter) {
    const int iters = 20000;
    const int grid = 256 * 4 * waves_per_simd;
    double* d;
    CHECK(hipMalloc(&d, (size_t)(grid * 64 + grid) * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, iters, 1.5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, iters, 1.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double cyc;
    CHECK(hipMemcpy(&cyc, d + (size_t)grid * 64, 8, hipMemcpyDeviceToHost));
    double n = (double)iters * 8 * ops_per_iter;
    printf("%-18s waves/SIMD %d: %.2f cycles per op seen by one wave; %.2f cycles of SIMD time per op (wall %.3f ms at 2.4 GHz)\n", name,
           waves_per_simd, cyc / n, ms * 1e-3 * 2.4e9 / n / waves_per_simd, ms);
    CHECK(hipFree(d));
}
int main() {
    const int ws[2] = {1, 4};
    for (int wi = 0; wi < 2; ++wi) {
        const int w = ws[wi];
        run<0>("fma_f64", w, 1); run<2>("add_f64", w, 1); run<3>("fma_f32", w, 1);
        run<4>("mad_u64_u32+xor", w, 1); run<7>("x/b f64", w, 1); run<6>("sqrt f64", w, 1);
        run<9>("add+min f64", w, 1); run<10>("cvt_u32+add f64", w, 1); run<11>("cvt_f32+add f64", w, 1);
        run<12>("cmp+sel+add f64", w, 1); run<13>("add+min f32", w, 1);
    }
    return 0;
}
