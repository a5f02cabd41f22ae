// Micro-benchmark: how fast a CU's vector memory path fetches 128-byte records at random addresses, (A) every lane
// reading its own record with seven 16-byte loads, as a per-lane tree walk does, against (B) eight lanes reading one
// record's eight 16-byte pieces in one load (eight loads bring in the wavefront's 64 records), data exchanged through
// LDS so that in both cases every lane ends up with its own record.  Records fetched per second for tables that fit L2
// and tables that do not (development aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(64, 3) per_lane(const uint4* table, unsigned mask, unsigned iters, unsigned* out, unsigned active) {
    const unsigned tid = blockIdx.x * 64u + threadIdx.x;
    unsigned acc = 0, h = mix(tid);
    for (unsigned it = 0; it < iters; ++it) {
        h = mix(h + acc);  // the next record depends on the one before, as a walk's next node does
        if ((threadIdx.x & 63u) < active) {
            const uint4* r = table + (size_t)(h & mask) * 8u;
            uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6];
            acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z;
        }
    }
    out[tid] = acc;
}
__global__ void __launch_bounds__(64, 3) together(const uint4* table, unsigned mask, unsigned iters, unsigned* out, unsigned active) {
    __shared__ uint4 stage[64 * 9];
    const unsigned tid = blockIdx.x * 64u + threadIdx.x, lane = threadIdx.x & 63u;
    unsigned acc = 0, h = mix(tid);
    for (unsigned it = 0; it < iters; ++it) {
        h = mix(h + acc);
        const unsigned mine = lane < active ? (h & mask) : 0u;
#pragma unroll
        for (unsigned i = 0; i < 8u; ++i) {
            const unsigned r = 8u * i + (lane >> 3);
            const unsigned rec = (unsigned)__shfl((int)mine, (int)r);
            stage[r * 9u + (lane & 7u)] = table[(size_t)rec * 8u + (lane & 7u)];
        }
        __syncthreads();
        if (lane < active) {
            const uint4* r = &stage[lane * 9u];
            uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6];
            acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z;
        }
        __syncthreads();
    }
    out[tid] = acc;
}
int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const unsigned waves = (unsigned)prop.multiProcessorCount * 12u, iters = 400;
    unsigned* out;
    CHECK(hipMalloc(&out, waves * 64u * 4u));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (unsigned log_records : {14u, 16u, 19u, 22u}) {  // 2 MB, 8 MB, 64 MB, 512 MB of records
        const size_t n = (size_t)1 << log_records;
        uint4* table;
        CHECK(hipMalloc(&table, n * 128u));
        CHECK(hipMemset(table, 1, n * 128u));
        for (unsigned active : {64u, 32u}) {
            for (int which = 0; which < 2; ++which) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (which == 0) hipLaunchKernelGGL(per_lane, dim3(waves), dim3(64), 0, 0, table, (unsigned)(n - 1), iters, out, active);
                    else hipLaunchKernelGGL(together, dim3(waves), dim3(64), 0, 0, table, (unsigned)(n - 1), iters, out, active);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double recs = (double)waves * active * iters;
                printf("%4zu MB table, %2u lanes active, %-9s: %7.3f ms, %6.1f G records/s, %5.0f cycles per round at 2.4 GHz\n", n * 128u >> 20, active,
                       which ? "together" : "per lane", best, recs / best / 1e6, best * 1e-3 * 2.4e9 / iters);
            }
        }
        CHECK(hipFree(table));
    }
    return 0;
}
