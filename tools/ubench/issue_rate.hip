// Micro-benchmark: what does ONE wave per SIMD sustain for scalar vs packed fp32 FMA?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/ubench/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) kern(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent scalar FMAs
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            }
        } else if (MODE == 1) {   // 4 independent packed FMAs (same flops as 8 scalar) x 2 = same INSTRUCTION count
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(a2), "v"(b2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(a2), "v"(b2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(a2), "v"(b2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(a2), "v"(b2));
            }
        } else {                  // dependent scalar chain
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x +
                                          p2.y + p3.x + p3.y;
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
    return ms * 1e-3;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    const int iters = 20000;
    for (int blocks : {256, 512, 1024, 2048}) {       // 1, 2, 4, 8 waves per SIMD
        const double t0 = run<0>(blocks, iters, d), t1 = run<1>(blocks, iters, d), t2 = run<2>(blocks, iters, d);
        const double waves = blocks / 256.0;           // per SIMD
        // instructions per wave = iters * 64; cycles at 2.4 GHz
        printf("%4.0f wave/SIMD: scalar fma %.2f cyc/inst/wave  (SIMD: %.2f)   pk fma %.2f (SIMD: %.2f)   dependent fma %.2f\n",
               waves, t0 * 2.4e9 / (iters * 64.0), t0 * 2.4e9 / (iters * 64.0 * waves), t1 * 2.4e9 / (iters * 64.0),
               t1 * 2.4e9 / (iters * 64.0 * waves), t2 * 2.4e9 / (iters * 64.0));
    }
    return 0;
}
