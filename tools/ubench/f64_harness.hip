// Stand-alone timing of the fp64 lane kernel on BASELINE configs[1] (4096 rollouts x 200 steps, per-rollout controls):
// seconds to build, so that variants of the fp64 fit evaluation (-DVDYN_FIT_KG / -DVDYN_FIT_AHEAD) can be compared in one
// GPU call.  Vehicle: the reference's, or with a different C on the rear axle (argv[1] = 1).
#define VDYN_ONLY_F32
#define VDYN_ONLY_F64
#include "vdyn_kernels.hip"
#include <cstdio>
#include <vector>
using namespace vdyn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int n = 4096, H = 200;
    const bool per_wheel = argc > 1 && atoi(argv[1]) == 1;
    const double rw = 0.308309813617345;
    std::vector<double> st((size_t)12 * n, 0.0), ctrl((size_t)H * 2 * n);
    for (int r = 0; r < n; ++r) {
        st[r] = 25.0;
        for (int w = 3; w < 7; ++w) st[(size_t)w * n + r] = 25.0 / rw;
        const double d = -0.3 + 0.6 * (r / 64) / 63.0, tq = -200.0 + 600.0 * (r % 64) / 63.0;
        for (int t = 0; t < H; ++t) { ctrl[((size_t)t * 2) * n + r] = d; ctrl[((size_t)t * 2 + 1) * n + r] = tq; }
    }
    double *d_st, *d_ctrl, *d_term;
    CK(hipMalloc(&d_st, st.size() * 8)); CK(hipMalloc(&d_ctrl, ctrl.size() * 8)); CK(hipMalloc(&d_term, st.size() * 8));
    CK(hipMemcpy(d_st, st.data(), st.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ctrl, ctrl.data(), ctrl.size() * 8, hipMemcpyHostToDevice));
    VdynParams p;
    p.m = 1857.82; p.a = 1.3351891891891894; p.b = 1.5708108108108108; p.Izz = 1948.2304506781593; p.Jw = 1; p.hg = 0.55419;
    p.T = 1.536; p.wL = p.wR = 0.768; p.rw = rw; p.g = 9.81;
    for (int i = 0; i < 4; ++i) { p.B[i] = 20.6357; p.C[i] = 1.5047; }
    if (per_wheel) p.C[2] = p.C[3] = 1.3;
    RolloutArgs<double> a;
    a.n = n; a.H = H; a.state0 = d_st; a.ctrl = d_ctrl; a.k = 2; a.layout = VDYN_CTRL_PER_ROLLOUT; a.dt = 1e-3; a.terminal = d_term;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 150; ++i) CK(launch_rollout<double>(p, a, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 40;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(launch_rollout<double>(p, a, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> term(st.size());
    CK(hipMemcpy(term.data(), d_term, st.size() * 8, hipMemcpyDeviceToHost));
    double chk = 0;
    for (double v : term) chk += v;
    printf("fp64 lane kernel, 4096 x 200, %s: %.4f ms per launch, checksum %.12e\n", per_wheel ? "rear C = 1.3" : "reference tires", ms / reps, chk);
    return 0;
}
