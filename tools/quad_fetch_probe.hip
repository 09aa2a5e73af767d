// Does the vector memory path serve four neighbouring lanes that read the four 16-byte quarters of ONE 64-byte record faster than
// four lanes that read quarters of four different records? Every lane of a wave chases its own chain of 64-byte records, as the
// per-lane hierarchy walk does, in two ways:
//   lane   each lane issues four 16-byte loads for its own record (64 different lines per load instruction);
//   quad   the four lanes of a quad fetch their four records together: load instruction i brings the record of quad lane i,
//          lane l its quarter l (16 different lines per instruction, 4 lanes per line); the owner then collects its quarters from
//          its quad neighbours (here: the word that holds the successor, by a cross-lane register read).
// Build: hipcc --offload-arch=gfx950 -O3 -o quad_fetch_probe tools/quad_fetch_probe.hip ; run: ./quad_fetch_probe [steps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void chase_lane(const uint4* __restrict__ buf, uint32_t mask, int steps, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 64u + lane) * 2654435761u) & mask;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int s = 0; s < steps; ++s) {
            const uint4* rec = buf + (size_t)idx * 4;
            const uint4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
            acc += a.y + b.x + c.x + d.x;
            idx = a.x & mask;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc + idx;
}

// `active` lanes own a chain (the first `active` lanes, as above); all four lanes of a quad take part in the loads of the quad's
// active owners. The cross-lane reads are DPP quad permutes (register moves: no LDS traffic that could hide the answer).
template <int I>
__device__ __forceinline__ uint32_t quad_read(uint32_t v) {  // the value quad lane I holds
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
}

template <int I>
__device__ __forceinline__ void quad_step(const uint4* __restrict__ buf, uint32_t idx, uint32_t l, uint32_t leader, int active, uint32_t mask,
                                          uint32_t& next, uint32_t& acc) {
    const uint32_t owner_idx = quad_read<I>(idx);
    const bool owner_active = (int)(leader + I) < active;   // (uniform over the quad)
    uint4 q = make_uint4(0, 0, 0, 0);
    if (owner_active) q = buf[(size_t)owner_idx * 4 + l];
    // the owner collects: successor = word 0 of quarter 0 (held by quad lane 0), and something of every quarter
    const uint32_t w0 = quad_read<0>(q.x);
    const uint32_t s1 = q.x + quad_read<1>(q.x) + quad_read<2>(q.x) + quad_read<3>(q.x);
    if (l == (uint32_t)I) { next = w0 & mask; acc += s1; }
}

__global__ __launch_bounds__(64) void chase_quad(const uint4* __restrict__ buf, uint32_t mask, int steps, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, l = lane & 3u, leader = lane & ~3u;
    uint32_t idx = ((blockIdx.x * 64u + lane) * 2654435761u) & mask;
    uint32_t acc = 0;
    const bool mine = (int)lane < active;
    for (int s = 0; s < steps; ++s) {
        uint32_t next = idx;
        quad_step<0>(buf, idx, l, leader, active, mask, next, acc);
        quad_step<1>(buf, idx, l, leader, active, mask, next, acc);
        quad_step<2>(buf, idx, l, leader, active, mask, next, acc);
        quad_step<3>(buf, idx, l, leader, active, mask, next, acc);
        if (mine) idx = next;
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
    printf("{\"tool\": \"tools/quad_fetch_probe.hip\", \"steps\": %d, \"compute_units\": %d, \"rows\": [\n", steps, cus);
    bool first = true;
    for (int lg : {15, 21}) {  // 2 MiB (stays in the L2s) and 128 MiB of 64-byte records
        const size_t n = (size_t)1 << lg;
        std::vector<uint32_t> h(n * 16);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
        for (size_t i = 0; i < n; ++i) h[i * 16] = (uint32_t)((i * 1664525u + 1013904223u) & (n - 1));
        uint4* buf;
        if (hipMalloc(&buf, n * 64) != hipSuccess) return 1;
        hipMemcpy(buf, h.data(), n * 64, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode)
            for (int active : {20, 64}) {
                const int blocks = cus * 20;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(chase_lane, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                    else hipLaunchKernelGGL(chase_quad, dim3(blocks), dim3(64), 0, 0, buf, (uint32_t)(n - 1), steps, active, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                const double records = (double)blocks * active * steps;
                printf("%s {\"footprint_MiB\": %zu, \"mode\": \"%s\", \"active_lanes\": %d, \"ms\": %.3f, \"Grecords_per_s\": %.2f}", first ? "" : ",\n",
                       n * 64 >> 20, mode ? "quad" : "lane", active, best, records / best * 1e-6);
                first = false;
                fflush(stdout);
            }
        hipFree(buf);
    }
    printf("\n]}\n");
    return 0;
}
