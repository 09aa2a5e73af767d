// Dependent 64-byte record fetches with a wave-uniform address, as the LBVH walk issues them: how many per second does a
// CU sustain through the scalar path (s_load_dwordx16) and through the vector path (16 lanes x dword + readlane)?
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_probe tools/fetch_probe.hip ; run: ./fetch_probe [records_log2] [steps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(4))) ku32x4;

template <int MODE>  // 0 scalar path, 1 vector path
__global__ __launch_bounds__(64) void chase(const uint32_t* __restrict__ buf, uint32_t mask, int steps, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 2654435761u) & mask;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
            const ku32x4* rec = (const ku32x4*)(buf + (size_t)idx * 16);
            const u32x4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
            acc += a.y + b.x + c.x + d.x;
            idx = a.x & mask;
        } else {
            const uint32_t v = buf[(size_t)idx * 16 + (threadIdx.x & 15)];
            acc += __builtin_amdgcn_readlane(v, 1) + __builtin_amdgcn_readlane(v, 4) + __builtin_amdgcn_readlane(v, 8) + __builtin_amdgcn_readlane(v, 12);
            idx = __builtin_amdgcn_readlane(v, 0) & mask;
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc + idx;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 21;  // 2^21 records x 64 B = 128 MiB
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const size_t n = (size_t)1 << lg;
    std::vector<uint32_t> h(n * 16);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
    // word 0 of record i = its successor in ONE cycle through all records (full-period LCG modulo 2^lg), so a chase
    // never settles into a short cached loop
    for (size_t i = 0; i < n; ++i) h[i * 16] = (uint32_t)((i * 1664525u + 1013904223u) & (n - 1));
    uint32_t *buf, *out;
    hipMalloc(&buf, n * 64);
    hipMemcpy(buf, h.data(), n * 64, hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipMalloc(&out, (size_t)cus * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int waves_per_cu : {4, 8, 16, 28, 32}) {
            const int blocks = cus * waves_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, out);
                else hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1)
                    printf("{\"path\": \"%s\", \"footprint_MiB\": %zu, \"waves_per_cu\": %d, \"ms\": %.3f, \"latency_ns_per_step\": %.1f, \"ns_per_record_per_cu\": %.2f, \"GB_per_s\": %.1f}\n",
                           mode == 0 ? "scalar" : "vector", n * 64 >> 20, waves_per_cu, ms, ms * 1e6 / steps, ms * 1e6 / steps / waves_per_cu,
                           (double)blocks * steps * 64 / (ms * 1e-3) * 1e-9);
            }
        }
    return 0;
}
