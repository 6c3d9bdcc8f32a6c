// Stand-alone timing harness of the closed-loop kernel (seconds to build instead of the library's two minutes):
// the bench workload of workloads.closed_loop_config -- 65536 vehicles, 7 constant-curvature tables of 1024
// waypoints at 3 cm, 200 sub-steps, a controller update every 10 -- generated here, the two table kernels, then the
// kernel itself timed with HIP events.  -DVDYN_STAMPS adds per-phase s_memtime totals of one wave (diagnostic build:
// the stamps cost cycles of their own, read the phases against each other, not against the unstamped time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
//         -I python-motionplanning_amd/csrc [-DVDYN_STAMPS] -o /tmp/cl_harness tools/ubench/cl_harness.hip && /tmp/cl_harness
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
    const int n = 65536, P = 7, W = 1024, H = argc > 1 ? atoi(argv[1]) : 200, every = argc > 2 ? atoi(argv[2]) : 10;
    const float ds = 0.03f, rw = 0.308309813617345f;
    std::mt19937 rng(20243);
    std::uniform_real_distribution<float> u01(0.f, 1.f);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> wp((size_t)P * W * 2), st((size_t)12 * n, 0.f), cs((size_t)6 * n, 0.f);
    std::vector<int> wc(P, W), pid(n);
    for (int k = 0; k < P; ++k)
        for (int i = 0; i < W; ++i) {
            const double kap = 0.004 * (k - P / 2), s = i * (double)ds;
            wp[((size_t)k * W + i) * 2] = (float)(kap == 0 ? s : sin(kap * s) / kap);
            wp[((size_t)k * W + i) * 2 + 1] = (float)(kap == 0 ? 0 : (1 - cos(kap * s)) / kap);
        }
    for (int r = 0; r < n; ++r) {
        pid[r] = r % P;
        const double kap = 0.004 * (pid[r] - P / 2), sa = 5.0 * u01(rng), lat = 0.3 * nrm(rng), th = kap * sa;
        const double cx = kap == 0 ? sa : sin(th) / kap, cy = kap == 0 ? 0 : (1 - cos(th)) / kap;
        const float U = 15.f + 15.f * u01(rng);
        st[r] = U;
        for (int w = 3; w < 7; ++w) st[(size_t)w * n + r] = U / rw;
        st[(size_t)7 * n + r] = (float)(th + 0.02 * nrm(rng));
        st[(size_t)8 * n + r] = (float)(cx - lat * sin(th));
        st[(size_t)9 * n + r] = (float)(cy + lat * cos(th));
        cs[(size_t)2 * n + r] = U;
        cs[(size_t)3 * n + r] = 25.f;
    }
    float *d_wp, *d_st, *d_cs, *d_term, *d_cso, *d_aux;
    int *d_wc, *d_pid;
    const size_t aux_bytes = closed_loop_aux_bytes<float>(P, W, false);
    CK(hipMalloc(&d_wp, wp.size() * 4)); CK(hipMalloc(&d_st, st.size() * 4)); CK(hipMalloc(&d_cs, cs.size() * 4));
    CK(hipMalloc(&d_term, st.size() * 4)); CK(hipMalloc(&d_cso, cs.size() * 4)); CK(hipMalloc(&d_aux, aux_bytes));
    CK(hipMalloc(&d_wc, P * 4)); CK(hipMalloc(&d_pid, n * 4));
    CK(hipMemcpy(d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wc, wc.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pid, pid.data(), n * 4, hipMemcpyHostToDevice));

    VdynParams p;
    p.m = 1857.82; p.a = 1.3351891891891894; p.b = 1.5708108108108108; p.Izz = 1948.2304506781593; p.Jw = 1; p.hg = 0.55419;
    p.T = 1.536; p.wL = p.wR = 0.768; p.rw = rw; p.g = 9.81;
    for (int i = 0; i < 4; ++i) { p.B[i] = 20.6357; p.C[i] = 1.5047; }
    VdynCtrlGains g;
    g.k = 0.5; g.k_soft = 1.0; g.max_steer = 0.5235987755982988; g.lookahead = 5.0; g.deadband = 0.0;   // drive.py:56,71-85
    g.kp = 1000; g.ki = 10; g.kd = 0; g.filter_gain = 1e-5 / (2 * 0.001);
    ClosedLoopArgs<float> a;
    a.n = n; a.H = H; a.ctrl_every = every; a.phase = 0; a.state0 = d_st; a.cstate0 = d_cs; a.wp = d_wp; a.Wmax = W;
    a.wcount = d_wc; a.path_id = d_pid; a.P = P; a.dt = 1e-3; a.terminal = d_term; a.cstate = d_cso; a.aux = d_aux;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 400; ++i) CK(launch_closed_loop<float>(p, g, a, s));     // clock ramp
    CK(hipStreamSynchronize(s));
#ifdef VDYN_STAMPS
    unsigned long long zero[16] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_vdyn_phase), zero, sizeof(zero)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_vdyn_event), zero, 4 * sizeof(unsigned long long)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 50;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(launch_closed_loop<float>(p, g, a, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> term(st.size());
    CK(hipMemcpy(term.data(), d_term, st.size() * 4, hipMemcpyDeviceToHost));
    double chk = 0;
    for (size_t i = 0; i < term.size(); ++i) chk += term[i];
    printf("closed loop %d x %d sub-steps, update every %d: %.4f ms per call (incl. table kernels), checksum %.6e\n", n, H,
           every, ms / reps, chk);
#ifdef VDYN_STAMPS
    unsigned long long ph[16];
    CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_vdyn_phase), sizeof(ph)));
    const char *names[16] = {"bound U", "32-wp circles", "8-wp circles", "block scan", "lookahead", "steering law",
                             "update total", "held sub-steps", "stanley total", "PID + filter", "LDS image", "state loads",
                             "img: stage x, y", "img: seg lengths", "img: circles", "img: mask+cumsum"};
    const int updates = (H + every - 1) / every;
    for (int i = 0; i < 16; ++i)
        printf("  %-16s %9.0f cycles per %s\n", names[i], (double)ph[i] / reps / (i == 7 ? H : i >= 10 ? 1 : updates),
               i == 7 ? "sub-step" : i >= 10 ? "launch" : "update");
    unsigned long long ev[4];
    CK(hipMemcpyFromSymbol(ev, HIP_SYMBOL(g_vdyn_event), sizeof(ev)));
    printf("  searches (wave-level) %llu per launch; ties settled over the masked sub-blocks %llu (%.2f %%), sent to the "
           "sequential exact search %llu\n", ev[0] / reps, ev[1] / reps, 100.0 * (double)ev[1] / (double)(ev[0] ? ev[0] : 1),
           ev[2] / reps);
#endif
    return 0;
}
