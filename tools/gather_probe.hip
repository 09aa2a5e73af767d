// Per-LANE dependent record fetches, as the per-lane hierarchy walk issues them: every active lane chases its own chain of
// 64-byte (or 32-byte) records with 16-byte loads. How many records per second does the chip sustain, by footprint, by
// active lanes per wave and by waves per CU? This is the ceiling of the memory side of pt_render_decoupled_kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip ; run: ./gather_probe [steps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int QUADS>  // 16-byte loads per record: 4 = 64-byte records, 2 = the first 32 bytes of each
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ buf, uint32_t mask, int steps, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 64u + lane) * 2654435761u) & mask;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int s = 0; s < steps; ++s) {
            const uint4* rec = buf + (size_t)idx * 4;
            const uint4 a = rec[0], b = rec[1];
            acc += a.y + b.x;
            if (QUADS == 4) {
                const uint4 c = rec[2], d = rec[3];
                acc += c.x + d.x;
            }
            idx = a.x & mask;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc + idx;  // keeps the chain alive, practically never taken
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
    printf("{\"tool\": \"tools/gather_probe.hip\", \"steps\": %d, \"compute_units\": %d, \"rows\": [\n", steps, cus);
    bool first = true;
    for (int lg : {15, 19, 21, 24}) {  // records: 2 MiB, 32 MiB, 128 MiB, 1 GiB of 64-byte records
        const size_t n = (size_t)1 << lg;
        std::vector<uint32_t> h(n * 16);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
        // word 0 of record i = its successor in ONE cycle through all records (full-period LCG modulo 2^lg)
        for (size_t i = 0; i < n; ++i) h[i * 16] = (uint32_t)((i * 1664525u + 1013904223u) & (n - 1));
        uint4* buf;
        if (hipMalloc(&buf, n * 64) != hipSuccess) return 1;
        hipMemcpy(buf, h.data(), n * 64, hipMemcpyHostToDevice);
        for (int quads : {4, 2})
            for (int waves_per_cu : {20, 32})
                for (int active : {16, 32, 64}) {
                    const int blocks = cus * waves_per_cu;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipEventRecord(e0);
                        if (quads == 4) hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                        else hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms = 0;
                        hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 0 && ms < best) best = ms;
                    }
                    const double records = (double)blocks * active * steps;
                    printf("%s {\"footprint_MiB\": %zu, \"bytes_per_record\": %d, \"waves_per_cu\": %d, \"active_lanes\": %d, \"ms\": %.3f, "
                           "\"Grecords_per_s\": %.2f, \"GB_per_s\": %.0f, \"ns_per_step_per_wave\": %.0f}",
                           first ? "" : ",\n", n * 64 >> 20, quads * 16, waves_per_cu, active, best, records / best * 1e-6,
                           records * quads * 16 / best * 1e-6, best * 1e6 / steps);
                    first = false;
                    fflush(stdout);
                }
        hipFree(buf);
    }
    printf("\n]}\n");
    return 0;
}
