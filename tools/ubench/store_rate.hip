// What a streaming store costs a lone wave per SIMD (the trajectory dump and the DataLog closed loop write 12 / 45 rows of
// [rows][n] per RK4 step): 1024 waves, each writes `rows` rows of 64 consecutive floats per iteration, with `fill`
// dependent fmas per store in between, as
//   mode 0: one global_store_dword per row (what the kernels do);
//   mode 1: a 4 x 4 transpose inside each quad of lanes (DPP) and one global_store_dwordx4 per FOUR rows -- the same
//           bytes at the same addresses.
// Prints ns per iteration and TB/s.   usage: store_rate [rows=12] [fill=0] [iters=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// quad_perm encodings: [1,0,3,2] = 0xB1 (xor 1), [2,3,0,1] = 0x4E (xor 2)
__device__ __forceinline__ void transpose4(float &a, float &b, float &c, float &d, int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
    // stage 1: exchange with lane ^ 1 -- pairs (a, b) and (c, d)
    float ta = dpp<0xB1>(o1 ? a : b), tc = dpp<0xB1>(o1 ? c : d);
    (o1 ? a : b) = ta; (o1 ? c : d) = tc;
    // stage 2: exchange with lane ^ 2 -- pairs (a, c) and (b, d)
    float ua = dpp<0x4E>(o2 ? a : c), ub = dpp<0x4E>(o2 ? b : d);
    (o2 ? a : c) = ua; (o2 ? b : d) = ub;
}

template <int MODE, int ROWS>
__global__ void __launch_bounds__(256) k(float *out, int64_t n, int iters, int fill, float seed)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float v[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) v[i] = seed + (float)i + (float)r;
    float *row = out + r;
    const int64_t pitch = (int64_t)ROWS * n;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                __builtin_nontemporal_store(v[i], row + (int64_t)i * n);
                for (int f = 0; f < fill; ++f) v[(i + 1) % ROWS] = __builtin_fmaf(v[(i + 1) % ROWS], 1.0000001f, 1e-7f);
            }
        } else {
#pragma unroll
            for (int i = 0; i + 3 < ROWS; i += 4) {
                float a = v[i], b = v[i + 1], c = v[i + 2], d = v[i + 3];
                transpose4(a, b, c, d, lane);
                // lane 4 j + q now holds row i + q, columns 4 j .. 4 j + 3
                float *dst = out + (int64_t)it * pitch + (int64_t)(i + (lane & 3)) * n + (r & ~(int64_t)3);
                typedef float f4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4{a, b, c, d}, reinterpret_cast<f4 *>(dst));
                for (int f = 0; f < 4 * fill; ++f) v[(i + 4) % ROWS] = __builtin_fmaf(v[(i + 4) % ROWS], 1.0000001f, 1e-7f);
            }
#pragma unroll
            for (int i = ROWS & ~3; i < ROWS; ++i) __builtin_nontemporal_store(v[i], row + (int64_t)i * n);
        }
        row += pitch;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) v[i] += 1.0f;
    }
}

template <int ROWS>
int run(int fill, int iters)
{
    const int64_t n = 65536;
    float *out;
    CK(hipMalloc(&out, sizeof(float) * ROWS * n * iters));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref((size_t)ROWS * n), got((size_t)ROWS * n);
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 3; ++w) {
            if (mode == 0) hipLaunchKernelGGL((k<0, ROWS>), dim3(n / 256), dim3(256), 0, 0, out, n, iters, fill, 1.0f);
            else hipLaunchKernelGGL((k<1, ROWS>), dim3(n / 256), dim3(256), 0, 0, out, n, iters, fill, 1.0f);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int w = 0; w < reps; ++w) {
            if (mode == 0) hipLaunchKernelGGL((k<0, ROWS>), dim3(n / 256), dim3(256), 0, 0, out, n, iters, fill, 1.0f);
            else hipLaunchKernelGGL((k<1, ROWS>), dim3(n / 256), dim3(256), 0, 0, out, n, iters, fill, 1.0f);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double bytes = 4.0 * ROWS * n * iters;
        CK(hipMemcpy((mode == 0 ? ref : got).data(), out + (size_t)(iters - 1) * ROWS * n, sizeof(float) * ROWS * n, hipMemcpyDeviceToHost));
        printf("rows %d fill %d mode %d (%s): %.4f ms, %.1f ns per iteration per wave, %.2f TB/s\n", ROWS, fill, mode,
               mode == 0 ? "dword per row" : "dwordx4 per four rows", ms, ms * 1e6 / iters, bytes / ms / 1e9);
    }
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
    printf("  last iteration's rows identical in both modes: %s (%zu differ)\n", bad ? "NO" : "yes", bad);
    CK(hipFree(out));
    return 0;
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 12, fill = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 200;
    if (rows == 12) return run<12>(fill, iters);
    if (rows == 44) return run<44>(fill, iters);
    printf("rows must be 12 or 44\n");
    return 1;
}
