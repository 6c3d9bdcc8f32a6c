// How accurate are the fp64 reciprocal / reciprocal-square-root seeds of gfx950, and what do ONE and TWO Newton steps
// leave?  (fm64::rcp / fm64::rsq take two; the fp64 step spends 224 of its 1043 instructions per RK4 step there.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/rcp64_probe tools/ubench/rcp64_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void probe(int n, const double *a, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = a[i];
    double x = __builtin_amdgcn_rcp(v);
    out[i] = x;
    double e = fma(-v, x, 1.0);
    x = fma(x, e, x);
    out[n + i] = x;
    e = fma(-v, x, 1.0);
    out[2 * n + i] = fma(x, e, x);
    double y = __builtin_amdgcn_rsq(v);
    out[3 * n + i] = y;
    double f = fma(-v * y, y, 1.0);
    y = fma(0.5 * y, f, y);
    out[4 * n + i] = y;
    f = fma(-v * y, y, 1.0);
    out[5 * n + i] = fma(0.5 * y, f, y);
    // one CUBIC step each: x (1 + e + e^2), y (1 + e / 2 + 3 e^2 / 8)
    x = __builtin_amdgcn_rcp(v);
    e = fma(-v, x, 1.0);
    out[6 * n + i] = fma(x, fma(e, e, e), x);
    y = __builtin_amdgcn_rsq(v);
    f = fma(-v * y, y, 1.0);
    out[7 * n + i] = fma(y, f * fma(0.375, f, 0.5), y);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> a(n), o(8 * (size_t)n);
    for (int i = 0; i < n; ++i) a[i] = std::exp(-20.0 + 40.0 * (i + 0.37) / n) * (1.0 + 1e-3 * std::sin(i * 12.9898));
    double *da, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&dout, 8 * (size_t)n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, n, da, dout);
    hipMemcpy(o.data(), dout, 8 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char *names[8] = {"v_rcp_f64 seed", "rcp, one Newton step", "rcp, two steps", "v_rsq_f64 seed", "rsq, one Newton step", "rsq, two steps",
                            "rcp, one cubic step", "rsq, one cubic step"};
    for (int k = 0; k < 8; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double want = (k < 3 || k == 6) ? 1.0L / (long double)a[i] : 1.0L / sqrtl((long double)a[i]);
            const long double err = fabsl(((long double)o[(size_t)k * n + i] - want) / want);
            if (err > worst) worst = err;
        }
        printf("%-22s max relative error %.3Le = 2^%.1Lf = %.2Lf ulp\n", names[k], worst, log2l(worst), worst / 1.1102230246251565e-16L);
    }
    return 0;
}
