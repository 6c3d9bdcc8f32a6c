// Micro-benchmark: issue cost of DEPENDENT vs independent VALU chains for one wave per SIMD on
// gfx950 (scalar v_fma_f32, packed v_pk_fma_f32, v_rcp_f32, v_cndmask), at ILP 1/2/4/8.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_latency tools/ubench/dep_latency.hip && /tmp/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define P_FMA(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a2), "v"(b2))
#define P_MUL(p) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(a2))
#define S_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define S_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a))

// MODE: 0..3 scalar fma ILP 1,2,4,8; 4..7 packed fma ILP 1,2,4,8; 8,9 rcp ILP 1,4;
//       10 alternating pk_fma -> scalar fma on its low half (mixed dependent); 11 pk_mul ILP 2;
//       12: pk ILP2 + 2 independent scalars per pk pair (fills the latency shadow?)
template <int MODE>
__global__ void __launch_bounds__(256) kern(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4},
       p7 = {x7, x6};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) { S_FMA(x0); S_FMA(x0); S_FMA(x0); S_FMA(x0); S_FMA(x0); S_FMA(x0); S_FMA(x0); S_FMA(x0); }
            if (MODE == 1) { S_FMA(x0); S_FMA(x1); S_FMA(x0); S_FMA(x1); S_FMA(x0); S_FMA(x1); S_FMA(x0); S_FMA(x1); }
            if (MODE == 2) { S_FMA(x0); S_FMA(x1); S_FMA(x2); S_FMA(x3); S_FMA(x0); S_FMA(x1); S_FMA(x2); S_FMA(x3); }
            if (MODE == 3) { S_FMA(x0); S_FMA(x1); S_FMA(x2); S_FMA(x3); S_FMA(x4); S_FMA(x5); S_FMA(x6); S_FMA(x7); }
            if (MODE == 4) { P_FMA(p0); P_FMA(p0); P_FMA(p0); P_FMA(p0); P_FMA(p0); P_FMA(p0); P_FMA(p0); P_FMA(p0); }
            if (MODE == 5) { P_FMA(p0); P_FMA(p1); P_FMA(p0); P_FMA(p1); P_FMA(p0); P_FMA(p1); P_FMA(p0); P_FMA(p1); }
            if (MODE == 6) { P_FMA(p0); P_FMA(p1); P_FMA(p2); P_FMA(p3); P_FMA(p0); P_FMA(p1); P_FMA(p2); P_FMA(p3); }
            if (MODE == 7) { P_FMA(p0); P_FMA(p1); P_FMA(p2); P_FMA(p3); P_FMA(p4); P_FMA(p5); P_FMA(p6); P_FMA(p7); }
            if (MODE == 8) { S_RCP(x0); S_RCP(x0); S_RCP(x0); S_RCP(x0); S_RCP(x0); S_RCP(x0); S_RCP(x0); S_RCP(x0); }
            if (MODE == 9) { S_RCP(x0); S_RCP(x1); S_RCP(x2); S_RCP(x3); S_RCP(x0); S_RCP(x1); S_RCP(x2); S_RCP(x3); }
            if (MODE == 10) {
                P_FMA(p0); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(p0.x), "v"(a));
                asm volatile("v_mov_b32 %0, %1" : "+v"(p0.x) : "v"(x0));
                P_FMA(p0); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(p0.x), "v"(a));
                asm volatile("v_mov_b32 %0, %1" : "+v"(p0.x) : "v"(x0));
                P_FMA(p0); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(p0.x), "v"(a));
                P_FMA(p0);
            }
            if (MODE == 11) { P_MUL(p0); P_MUL(p1); P_MUL(p0); P_MUL(p1); P_MUL(p0); P_MUL(p1); P_MUL(p0); P_MUL(p1); }
            if (MODE == 12) { P_FMA(p0); P_FMA(p1); S_FMA(x0); S_FMA(x1); P_FMA(p0); P_FMA(p1); S_FMA(x2); S_FMA(x3); }
            if (MODE == 13) { S_MAX(x0); S_MAX(x0); S_MAX(x0); S_MAX(x0); S_MAX(x0); S_MAX(x0); S_MAX(x0); S_MAX(x0); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x +
                                          p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x + p4.y + p5.y + p6.y + p7.y;
}

template <int MODE>
double run(int blocks, int iters, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e-3;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * sizeof(float));
    const int iters = 4000;
    const char *names[] = {"scalar fma ILP1", "scalar fma ILP2", "scalar fma ILP4", "scalar fma ILP8",
                           "pk fma ILP1", "pk fma ILP2", "pk fma ILP4", "pk fma ILP8", "rcp ILP1", "rcp ILP4",
                           "pk->scalar->mov dep (7 inst)", "pk mul ILP2", "pk ILP2 + scalar ILP2 mix", "max ILP1"};
    for (int blocks : {256, 512}) {                    // 1, 2 waves per SIMD
        printf("---- %d wave(s) per SIMD: cycles per instruction per wave at 2.4 GHz\n", blocks / 256);
        double t[14];
        t[0] = run<0>(blocks, iters, d); t[1] = run<1>(blocks, iters, d); t[2] = run<2>(blocks, iters, d);
        t[3] = run<3>(blocks, iters, d); t[4] = run<4>(blocks, iters, d); t[5] = run<5>(blocks, iters, d);
        t[6] = run<6>(blocks, iters, d); t[7] = run<7>(blocks, iters, d); t[8] = run<8>(blocks, iters, d);
        t[9] = run<9>(blocks, iters, d); t[10] = run<10>(blocks, iters, d); t[11] = run<11>(blocks, iters, d);
        t[12] = run<12>(blocks, iters, d); t[13] = run<13>(blocks, iters, d);
        for (int m = 0; m < 14; ++m) {
            const double n = m == 10 ? 9.0 * 8 : 64.0;
            printf("  %-32s %.2f\n", names[m], t[m] * 2.4e9 / (iters * n));
        }
    }
    return 0;
}
