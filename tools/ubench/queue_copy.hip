// Micro-benchmark: the memory side of the shade step alone -- read the path queue and hit buffer (SoA: 9 f64 + 5 u32 per
// path, plus t / ref / chain) and write a queue of the same layout, with no arithmetic in between.  Gives the
// floor that the layout and HBM set for a shade launch (development aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Q { double* d[9]; unsigned* u[5]; };
struct H { double* t; unsigned* ref; unsigned* chain; };
__global__ void __launch_bounds__(256) copy_soa(Q in, H h, Q out, unsigned n, double keep) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[9]; unsigned w[5];
    for (int k = 0; k < 9; ++k) v[k] = in.d[k][i];
    for (int k = 0; k < 5; ++k) w[k] = in.u[k][i];
    double t = h.t[i]; unsigned r = h.ref[i], c = h.chain[i];
    v[0] += t; w[0] += r + c;
    if (v[1] < keep) {  // a fraction of the paths ends here
        for (int k = 0; k < 9; ++k) out.d[k][i] = v[k];
        for (int k = 0; k < 5; ++k) out.u[k][i] = w[k];
    }
}
// the same bytes as one 96-byte record per path
struct Rec { double v[9]; unsigned w[5]; unsigned pad; };
__global__ void __launch_bounds__(256) copy_aos(const Rec* in, H h, Rec* out, unsigned n, double keep) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rec r = in[i];
    double t = h.t[i]; unsigned rf = h.ref[i], c = h.chain[i];
    r.v[0] += t; r.w[0] += rf + c;
    if (r.v[1] < keep) out[i] = r;
}
int main() {
    const unsigned n = 7u << 20;
    Q a, b; H h;
    for (int k = 0; k < 9; ++k) { CHECK(hipMalloc(&a.d[k], n * 8ull)); CHECK(hipMalloc(&b.d[k], n * 8ull)); CHECK(hipMemset(a.d[k], 0, n * 8ull)); }
    for (int k = 0; k < 5; ++k) { CHECK(hipMalloc(&a.u[k], n * 4ull)); CHECK(hipMalloc(&b.u[k], n * 4ull)); CHECK(hipMemset(a.u[k], 0, n * 4ull)); }
    CHECK(hipMalloc(&h.t, n * 8ull)); CHECK(hipMalloc(&h.ref, n * 4ull)); CHECK(hipMalloc(&h.chain, n * 4ull));
    CHECK(hipMemset(h.t, 0, n * 8ull)); CHECK(hipMemset(h.ref, 0, n * 4ull)); CHECK(hipMemset(h.chain, 0, n * 4ull));
    Rec *ra, *rb;
    CHECK(hipMalloc(&ra, n * sizeof(Rec))); CHECK(hipMalloc(&rb, n * sizeof(Rec))); CHECK(hipMemset(ra, 0, n * sizeof(Rec)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(copy_soa, dim3((n + 255) / 256), dim3(256), 0, 0, a, h, b, n, 1.0);
            else hipLaunchKernelGGL(copy_aos, dim3((n + 255) / 256), dim3(256), 0, 0, ra, h, rb, n, 1.0);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        double bytes = (double)n * (92 + 16 + 92 + (mode ? 8 : 0));
        printf("%s: %u paths, %.3f ms, %.2f TB/s (%.0f bytes per path)\n", mode ? "AoS 96-byte records" : "SoA 14 arrays", n, best,
               bytes / best / 1e9, bytes / n);
    }
    return 0;
}
