// Per-lane dependent fetches of 128-byte records (a whole cache line, eight 16-byte loads) next to 64-byte records in the same
// buffer: does a line cost the memory system as much as two half lines, or as much as one? (The question behind hierarchy nodes of
// eight children in one line instead of four in half a line.)
// Build: hipcc --offload-arch=gfx950 -O3 -o line_fetch_probe tools/line_fetch_probe.hip ; run: ./line_fetch_probe [steps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int QUADS>  // 16-byte loads per record; records lie 128 bytes apart in both cases
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ buf, uint32_t mask, int steps, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 64u + lane) * 2654435761u) & mask;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int s = 0; s < steps; ++s) {
            const uint4* rec = buf + (size_t)idx * 8;
            uint4 q[QUADS];
#pragma unroll
            for (int k = 0; k < QUADS; ++k) q[k] = rec[k];
#pragma unroll
            for (int k = 1; k < QUADS; ++k) acc += q[k].x;
            idx = q[0].x & mask;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc + idx;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t* out;
    hipMalloc(&out, (size_t)cus * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"tool\": \"tools/line_fetch_probe.hip\", \"steps\": %d, \"compute_units\": %d, \"rows\": [\n", steps, cus);
    bool first = true;
    for (int lg : {14, 18, 20}) {  // lines: 2 MiB, 32 MiB, 128 MiB
        const size_t n = (size_t)1 << lg;
        std::vector<uint32_t> h(n * 32);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n * 32; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
        for (size_t i = 0; i < n; ++i) h[i * 32] = (uint32_t)((i * 1664525u + 1013904223u) & (n - 1));
        uint4* buf;
        if (hipMalloc(&buf, n * 128) != hipSuccess) return 1;
        hipMemcpy(buf, h.data(), n * 128, hipMemcpyHostToDevice);
        for (int quads : {4, 5, 8})
            for (int active : {20, 64}) {
                const int blocks = cus * 20;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (quads == 4) hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                    else if (quads == 5) hipLaunchKernelGGL(chase<5>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                    else hipLaunchKernelGGL(chase<8>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                const double records = (double)blocks * active * steps;
                printf("%s {\"footprint_MiB\": %zu, \"bytes_read_per_record\": %d, \"active_lanes\": %d, \"ms\": %.3f, \"Grecords_per_s\": %.2f}", first ? "" : ",\n",
                       n * 128 >> 20, quads * 16, active, best, records / best * 1e-6);
                first = false;
                fflush(stdout);
            }
        hipFree(buf);
    }
    printf("\n]}\n");
    return 0;
}
