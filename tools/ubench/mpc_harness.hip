// Stand-alone A/B of the two config-5 selection kernels (BASELINE configs[4]: 1024 egos x 512 candidates x 50 steps,
// fp32): mpc_argmin_kernel (a workgroup per ego, candidates on the lanes, per-lane control loads) against
// mpc_argmin_lanes_kernel (egos on the lanes, the candidate wave-uniform, scalar control loads) + mpc_reduce_kernel.
// Same workload shape as workloads.config5 (seeded here, not bit-identical to NumPy's); both must give the same
// (cost, index) per ego, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
//         -I python-motionplanning_amd/csrc -o tools/ubench/bin/mpc_harness tools/ubench/mpc_harness.hip
#define VDYN_ONLY_F32
#define VDYN_ONLY_F64
#include "vdyn_kernels.hip"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using namespace vdyn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int E = argc > 1 ? atoi(argv[1]) : 1024, C = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 50;
    const float rw = 0.308309813617345f, dt = 2e-3f, w_delta = 1e-3f;
    std::mt19937 rng(20241);
    std::uniform_real_distribution<float> u01(0.f, 1.f);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> ego((size_t)12 * E, 0.f), cand((size_t)H * 2 * C), goal((size_t)2 * E);
    for (int e = 0; e < E; ++e) {
        const float U = 10.f + 20.f * u01(rng), yaw = -3.14159f + 6.28318f * u01(rng);
        ego[e] = U; ego[(size_t)E + e] = 0.2f * nrm(rng); ego[(size_t)2 * E + e] = 0.1f * nrm(rng);
        for (int w = 3; w < 7; ++w) ego[(size_t)w * E + e] = U / rw * (1.f + 0.01f * (2 * u01(rng) - 1));
        ego[(size_t)7 * E + e] = yaw; ego[(size_t)8 * E + e] = 100.f * u01(rng); ego[(size_t)9 * E + e] = 100.f * u01(rng);
        const float lat = u01(rng) - 0.5f, ht = H * dt;
        goal[e] = ego[(size_t)8 * E + e] + U * ht * cosf(yaw) - lat * sinf(yaw);
        goal[(size_t)E + e] = ego[(size_t)9 * E + e] + U * ht * sinf(yaw) + lat * cosf(yaw);
    }
    const int knots = 5, hold = H / knots > 0 ? H / knots : 1;
    for (int c = 0; c < C; ++c)
        for (int k = 0; k * hold < H; ++k) {
            const float d = fminf(fmaxf(0.05f * nrm(rng), -0.5236f), 0.5236f), tq = 100.f + 200.f * nrm(rng);
            for (int t = k * hold; t < H && t < (k + 1) * hold + (k == knots - 1 ? H : 0); ++t) {
                cand[((size_t)t * 2) * C + c] = d;
                cand[((size_t)t * 2 + 1) * C + c] = tq;
            }
        }
    float *d_ego, *d_cand, *d_goal, *d_bc[2], *d_cand4;
    int *d_bi[2];
    void *d_scratch;
    CK(hipMalloc(&d_ego, ego.size() * 4)); CK(hipMalloc(&d_cand, cand.size() * 4)); CK(hipMalloc(&d_goal, goal.size() * 4));
    CK(hipMalloc(&d_cand4, (size_t)H * C * 16));
    CK(hipMalloc(&d_scratch, mpc_scratch_bytes<float>(E, C, H)));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&d_bc[i], E * 4)); CK(hipMalloc(&d_bi[i], E * 4)); }
    CK(hipMemcpy(d_ego, ego.data(), ego.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cand, cand.data(), cand.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_goal, goal.data(), goal.size() * 4, hipMemcpyHostToDevice));
    VdynParams p;
    p.m = 1857.82; p.a = 1.3351891891891894; p.b = 1.5708108108108108; p.Izz = 1948.2304506781593; p.Jw = 1; p.hg = 0.55419;
    p.T = 1.536; p.wL = p.wR = 0.768; p.rw = rw; p.g = 9.81;
    for (int i = 0; i < 4; ++i) { p.B[i] = 20.6357; p.C[i] = 1.5047; }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const DevParams<float> P = make_dev_params<float>(p, nullptr);
    auto run_old = [&]() -> hipError_t {
        const int64_t n = (int64_t)H * C;
        hipLaunchKernelGGL((mpc_prepare_kernel<float>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, C, H, d_cand, d_cand4);
        int block = ((C + 63) / 64) * 64;
        block = std::max(64, std::min(block, mpc_block_max<float>()));
        hipLaunchKernelGGL((mpc_argmin_kernel<float, true>), dim3((unsigned)E), dim3((unsigned)block), 0, s, P, E, C, H, d_ego, d_cand4,
                           d_goal, dt, w_delta, d_bc[0], d_bi[0], (float *)nullptr);
        return hipGetLastError();
    };
    auto run_new = [&]() -> hipError_t {
        return launch_mpc_argmin<float>(p, E, C, H, d_ego, d_cand, d_goal, dt, w_delta, d_bc[1], d_bi[1], (float *)nullptr, d_scratch, s);
    };
    const int reps = 100;
    for (int which = 0; which < 4; ++which) {
        const bool nw = which & 1;
        for (int i = 0; i < 300; ++i) CK(nw ? run_new() : run_old());      // clock ramp
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(nw ? run_new() : run_old());
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.4f ms per call (all kernels of the call), %.3e vehicle-steps/s\n",
               nw ? "egos on lanes (new)        " : "workgroup per ego (round 2)", ms / reps, (double)E * C * H / (ms / reps * 1e-3));
    }
    std::vector<float> bc[2] = {std::vector<float>(E), std::vector<float>(E)};
    std::vector<int> bi[2] = {std::vector<int>(E), std::vector<int>(E)};
    for (int i = 0; i < 2; ++i) {
        CK(hipMemcpy(bc[i].data(), d_bc[i], E * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(bi[i].data(), d_bi[i], E * 4, hipMemcpyDeviceToHost));
    }
    int bad = 0;
    for (int e = 0; e < E; ++e) bad += (bi[0][e] != bi[1][e]) || (memcmp(&bc[0][e], &bc[1][e], 4) != 0);
    printf("egos on which the two kernels disagree (index or cost bits): %d of %d; ego 0: idx %d cost %.6f\n", bad, E, bi[1][0], bc[1][0]);
    return bad != 0;
}
