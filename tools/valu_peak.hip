// Micro-benchmark: sustained rate of UNFUSED fp32 vector arithmetic on gfx950, scalar (v_mul_f32 / v_add_f32) and
// packed (v_pk_mul_f32 / v_pk_add_f32). The path tracer may not fuse a*b+c, so this -- not the FMA peak -- is the
// ceiling its inner loop can approach. Prints flop/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

#define REPEAT8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float k = 1.0000001f, c = 1e-9f;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, kk = {k, k}, cc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent chains, mul then add, scalar ops: 16 instructions, 16 flop per lane
            REPEAT8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                                 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                                 "v_add_f32 %0, %0, %9\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %9\n v_add_f32 %3, %3, %9\n"
                                 "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c));)
        } else if (MODE == 1) {  // packed: 8 instructions, 16 flop per lane
            REPEAT8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                                 "v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk), "v"(cc));)
        } else {  // fused reference: 8 v_fma_f32, 16 flop per lane
            REPEAT8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
double run(const char* name, float* d, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    double best = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 256 * iters * 8 * 16;
        const double tf = flop / (ms * 1e-3) * 1e-12;
        if (tf > best) best = tf;
    }
    printf("{\"variant\": \"%s\", \"blocks\": %d, \"tflops\": %.2f}\n", name, blocks, best);
    return best;
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    for (int blocks : {256 * 4, 256 * 8}) {
        run<0>("v_mul_f32+v_add_f32 (unfused scalar)", d, blocks);
        run<1>("v_pk_mul_f32+v_pk_add_f32 (unfused packed)", d, blocks);
        run<2>("v_fma_f32 (fused, for reference)", d, blocks);
    }
    hipFree(d);
    return 0;
}
