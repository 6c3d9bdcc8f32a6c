// vdyn_kernels.hip -- gfx950 kernels of the batched RK4 / Pacejka path and their
// launchers.  Mapping: one wavefront lane = one rollout; the 12 persistent
// scalars of a rollout (10 states + ax_prev, ay_prev) stay in VGPRs for the
// whole horizon; vehicle constants arrive as a by-value kernel argument and sit
// in SGPRs; lattice-shared control tables are staged through LDS; per-rollout
// controls are read time-major so that a wave reads 64 consecutive values.
#include "vdyn_internal.hpp"

#include <cstring>
#include <utility>
#include "vdyn_device.hpp"
#include "vdyn_packed.hpp"
#include "vdyn_controls.hpp"
#include "vdyn_quad.hpp"
#include "vdyn_quad_packed.hpp"
#include "vdyn_lattice.hpp"

namespace vdyn {

constexpr int kBlock = 256;            // 4 waves: one per SIMD of a CU
constexpr int kLdsBudget = 48 * 1024;  // bytes of control table staged per chunk

// Controls of one step for lane r.
//   LAYOUT 0: global [H][K][n]          (coalesced over r)
//   LAYOUT 1: LDS chunk [tc][K][P]      (lane reads column path_id[r])
//   LAYOUT 2: global table [P][H][K]    (gather; only when a step of the table exceeds the LDS budget)
template <typename T, int K>
struct Ctrl {
    T delta[4], tq[4], mu[4];
    T sd0 = T(0), cd0 = T(1);   // sin / cos of delta[0] when the staged table carries them (set_pre)
    // entry of a table staged by stage_control_table_pre: (delta_front, torque_all, sin delta, cos delta)
    __device__ __forceinline__ void set_pre(const DevParams<T> &P, const T *e)
    {
        delta[0] = delta[1] = e[0];
        delta[2] = delta[3] = T(0);
        sd0 = e[2];
        cd0 = e[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) { tq[i] = e[1]; mu[i] = P.mu[i]; }
    }
    __device__ __forceinline__ void set(const DevParams<T> &P, const T *c, int64_t stride)
    {
        if (K == 2) {
            delta[0] = delta[1] = c[0];
            delta[2] = delta[3] = T(0);
            const T t = c[stride];
#pragma unroll
            for (int i = 0; i < 4; ++i) { tq[i] = t; mu[i] = P.mu[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                delta[i] = c[(int64_t)i * stride];
                tq[i] = c[(int64_t)(4 + i) * stride];
                mu[i] = c[(int64_t)(8 + i) * stride];
            }
        }
    }
};

// Stage steps [t0, t0 + tc_n) of the shared control table ctrl [P][H][K] into LDS as tab [tc][k][p]:
// one path's slice is contiguous in memory, so the lanes read along it (coalesced, no runtime
// division) and scatter into LDS with stride P (7 for the lattice: conflict-free).
template <typename T, int K>
__device__ __forceinline__ void stage_control_table(T *__restrict__ tab, const T *__restrict__ ctrl, int Pn, int H,
                                                    int t0, int tc_n)
{
    const int row = tc_n * K;
    for (int p = 0; p < Pn; ++p) {
        const T *src = ctrl + ((int64_t)p * H + t0) * K;
        for (int j = threadIdx.x; j < row; j += kBlock) tab[j * Pn + p] = src[j];   // j = tc * K + kk
    }
}

// The same for k = 2 tables, with the steering angle's (sin, cos) computed here, once per table entry,
// instead of by every lane at every step: tab [tc][p][4] = (delta_front, torque_all, sin, cos), one
// 16-byte LDS read per lane and step.  `eng.steer_sincos` is the function the step itself would call,
// so LDS-shared and per-rollout controls still give bit-identical rollouts.
template <typename T, typename ENG>
__device__ __forceinline__ void stage_control_table_pre(T *__restrict__ tab, const T *__restrict__ ctrl, int Pn, int H,
                                                        int t0, int tc_n, const ENG &eng)
{
    for (int p = 0; p < Pn; ++p) {
        const T *src = ctrl + ((int64_t)p * H + t0) * 2;
        for (int j = threadIdx.x; j < tc_n; j += kBlock) {
            const T d = src[2 * j], tq = src[2 * j + 1];
            T sd, cd;
            eng.steer_sincos(d, sd, cd);
            T *dst = tab + ((int64_t)j * Pn + p) * 4;
            dst[0] = d;
            dst[1] = tq;
            dst[2] = sd;
            dst[3] = cd;
        }
    }
}

// H zero-order-hold RK4 steps per lane (the loop of drive.py:114,141-143 with
// vehicle_model.py:427-445 inside).  DIAG additionally returns the last step's
// state_dot / outputs (used for H = 1: the planar_model_RK4 drop-in).
// CS: every B >= 0 and the handle's tire fits validated (lane_cs below; true of any realistic tire, and of the
// reference's B = 20.6, C = 1.5047): the FAST step takes the fitted chain (TireFit, vdyn_device.hpp).
// k = 2 tables staged in LDS carry (sin, cos) of the steering angle per entry (not for the
// diagnostic single-step variant, which is launch-latency bound)
template <int K, int LAYOUT, bool DIAG> constexpr bool rollout_table_pre() { return K == 2 && LAYOUT == 1 && !DIAG; }
template <typename T, int K, int LAYOUT, bool DIAG> constexpr size_t rollout_lds_per_step(int P)
{
    return (size_t)P * (rollout_table_pre<K, LAYOUT, DIAG>() ? 4 : K) * sizeof(T);
}

// TRAJ: the launch writes trajectory rows.  A separate instance, because the test "is this a trajectory step" is
// a (taken) branch per step on the normal path otherwise, and a lone wave pays tens of cycles for each.
// After the fetch of the NEXT step's controls: no instruction may be scheduled across.  Left alone, the scheduler
// sinks the load to ~25 instructions before its use at the end of the step it was meant to hide behind (an LDS read
// takes longer than that for a lone wave); pinned at the top, the step ends on `s_waitcnt lgkmcnt(1)` for a load
// issued a whole step earlier.  A/B on one box, two runs each: headline 0.1540 -> 0.1534 ms, per-rollout controls
// (global loads) 0.1683 -> 0.1633 ms, configs[1] fp64 0.5235 -> 0.4927 ms; MPC and the closed loop unchanged.
#define VDYN_FETCH_FENCE __builtin_amdgcn_sched_barrier(0);

// Where the next trajectory row goes.  Row j of traj [H / stride][12][n] is the state after step (j + 1) * stride of
// the launch: a wave-uniform countdown and a running pointer -- `(t + 1) % stride == 0` and `(t + 1) / stride` are a
// scalar division each (no such instruction: ~30 issue slots per step for a lone wave, a tenth of the step).
// (Per-lane pointer form: what the wheel-parallel kernel still uses; the lane kernels write through RowWriter.)
template <typename T>
struct TrajCursor {
    T *row;
    int64_t pitch;      // elements from one row to the next: 12 n
    int left, stride;
    __device__ __forceinline__ TrajCursor(T *traj, int64_t r, int64_t n, int stride_)
        : row(traj != nullptr ? traj + r : nullptr), pitch(12 * n), left(stride_), stride(stride_) {}
    // after every step; true: this step's state is a trajectory row -- store it at row, then call next()
    __device__ __forceinline__ bool due()
    {
        if (--left != 0) return false;
        left = stride;
        return row != nullptr;
    }
    __device__ __forceinline__ void next() { row += pitch; }
};

// Streaming writer of rows [rows][COLS][n] (trajectories: COLS = 12; DataLog: 45; closed-loop log: 16), one value per
// lane and column.  What a lone wave pays for is instructions, and a store at `row + i n` from a per-lane 64-bit
// pointer is two of them -- a dependent `v_lshl_add_u64` and the `global_store` -- twelve (forty-five) times per step.
// Here the address is split the way the hardware adds it up for nothing:
//   * the row's base is WAVE-UNIFORM (the workgroup's first column of the current row): a buffer descriptor in four
//     SGPRs, advanced by the row pitch with two scalar adds per row;
//   * the column offsets i n sizeof(T) are wave-uniform and loop-invariant: SGPRs, the store's `soffset` operand;
//   * the lane's part is ONE 32-bit VGPR, computed once per launch
// so a column is `buffer_store_dword vdata, voff, s[rsrc], soff offen nt` and nothing else.  More than 16 columns
// would want more SGPRs than the step leaves: G lane offsets (each with COLS / G columns' worth folded in) x COLS / G
// scalar offsets -- 45 = 3 x 15.
// Idle lanes of the last workgroup shadow rollout n - 1 (they integrate it, too): they store the SAME value to the
// SAME address as its owner, so no store needs the exec mask.
// Host-side precondition (vdyn_capi.hip, row_writer_fits): COLS n sizeof(T) <= 2^31, every offset a positive int32.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <typename T, int COLS, int G = 1>
struct RowWriter {
    static_assert(COLS % G == 0, "columns per group");
    static constexpr int PER = COLS / G;
    char *base;                 // wave-uniform: the workgroup's first column in the current row
    int64_t pitch;              // bytes from one row to the next
    uint32_t voff[G];           // lane: bytes from `base` (+ the group's first column)
    uint32_t soff[PER];         // wave-uniform: bytes of column i of a group
    int left, stride;
    __device__ __forceinline__ RowWriter(T *out, int64_t r, int64_t n, int stride_ = 1)
        : base(reinterpret_cast<char *>(out + (int64_t)blockIdx.x * kBlock)), pitch((int64_t)COLS * n * (int64_t)sizeof(T)),
          left(stride_), stride(stride_)
    {
        const uint32_t col = (uint32_t)n * (uint32_t)sizeof(T);
        const uint32_t lane = (uint32_t)(r - (int64_t)blockIdx.x * kBlock) * (uint32_t)sizeof(T);
#pragma unroll
        for (int g = 0; g < G; ++g) voff[g] = lane + (uint32_t)(g * PER) * col;
#pragma unroll
        for (int i = 0; i < PER; ++i) soff[i] = (uint32_t)i * col;
    }
    // after every step; true: this step's values are a row -- put() its columns, then next()
    __device__ __forceinline__ bool due()
    {
        if (--left != 0) return false;
        left = stride;
        return true;
    }
    struct Row {
        __amdgpu_buffer_rsrc_t rs;
        const RowWriter &w;
        template <int C>
        __device__ __forceinline__ void put(T v) const
        {
            static_assert(C >= 0 && C < COLS, "column");
            constexpr int kNt = 2;      // gfx940+: the `nt` bit -- written once, never read back by this kernel
            if constexpr (sizeof(T) == 4)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, w.voff[C / PER], w.soff[C % PER], kNt);
            else
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rs, w.voff[C / PER], w.soff[C % PER], kNt);
        }
    };
    // raw buffer (stride 0), data format 32 bits (only read by typed instructions); num_records 2^31: every offset the
    // precondition allows is in range
    __device__ __forceinline__ Row row() const
    {
        return Row{__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000), *this};
    }
    __device__ __forceinline__ void next() { base += pitch; }
};
// Columns C0 .. C0 + CNT - 1 of a row from value(i), i = 0 .. CNT - 1, with compile-time column numbers
template <typename T, int C0, typename ROW, typename F, int... I>
__device__ __forceinline__ void put_cols_seq(const ROW &row, F &&value, std::integer_sequence<int, I...>)
{
    (row.template put<C0 + I>((T)value(I)), ...);
}
template <typename T, int C0, int CNT, typename ROW, typename F>
__device__ __forceinline__ void put_cols(const ROW &row, F &&value)
{
    put_cols_seq<T, C0>(row, value, std::make_integer_sequence<int, CNT>{});
}
// The twelve state rows of a lane kernel's packed state as one trajectory row
template <typename T, typename ROW, typename STATE>
__device__ __forceinline__ void put_state12(const ROW &row, const STATE &X)
{
    put_cols<T, 0, 12>(row, [&](int i) __attribute__((always_inline)) { return X.get(i); });
}
// The reading counterpart for per-rollout controls [H][K][n] (LAYOUT 0): the rows of step t + 1 follow those of step t
// in memory, so the step loop keeps a wave-uniform running base (a buffer descriptor advanced by the pitch with two
// scalar adds) where `ctrl + t K n + r` per lane was a 64-bit scalar multiply (eight SALU instructions: there is no
// 64-bit s_mul) and a `v_lshl_add_u64` per row -- twelve issue slots per step for two loads, now six.  Rows are read
// strictly in order, one load() per step; the caller never loads past row H - 1.  Same precondition as RowWriter
// (K n sizeof(T) <= 2^31, `fits`), checked by the kernel itself: a larger batch keeps the per-lane addresses.
template <typename T, int K>
struct RowReader {
    const char *base;           // wave-uniform: the workgroup's first column of the row the next load() reads
    int64_t pitch;              // bytes from one step's rows to the next: K n sizeof(T)
    uint32_t voff;              // lane: bytes from `base`
    uint32_t soff[K];           // wave-uniform: bytes of row i of a step
    static __device__ __forceinline__ bool fits(int64_t n) { return (int64_t)K * n * (int64_t)sizeof(T) <= ((int64_t)1 << 31); }
    __device__ __forceinline__ RowReader(const T *ctrl, int64_t r, int64_t n, int t)
        : base(reinterpret_cast<const char *>(ctrl + (int64_t)blockIdx.x * kBlock + (int64_t)t * K * n)),
          pitch((int64_t)K * n * (int64_t)sizeof(T)),
          voff((uint32_t)(r - (int64_t)blockIdx.x * kBlock) * (uint32_t)sizeof(T))
    {
#pragma unroll
        for (int i = 0; i < K; ++i) soff[i] = (uint32_t)i * (uint32_t)n * (uint32_t)sizeof(T);
    }
    template <int I>
    __device__ __forceinline__ T row(__amdgpu_buffer_rsrc_t rs) const
    {
        if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff[I], 0));
        else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff[I], 0));
    }
    // the controls of the current row into c (as Ctrl::set would), then on to the next row
    __device__ __forceinline__ void load(const DevParams<T> &P, Ctrl<T, K> &c)
    {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 0x7fffffff, 0x00020000);
        if constexpr (K == 2) {
            c.delta[0] = c.delta[1] = row<0>(rs);
            c.delta[2] = c.delta[3] = T(0);
            const T t = row<1>(rs);
#pragma unroll
            for (int i = 0; i < 4; ++i) { c.tq[i] = t; c.mu[i] = P.mu[i]; }
        } else {
            c.delta[0] = row<0>(rs); c.delta[1] = row<1>(rs); c.delta[2] = row<2>(rs); c.delta[3] = row<3>(rs);
            c.tq[0] = row<4>(rs); c.tq[1] = row<5>(rs); c.tq[2] = row<6>(rs); c.tq[3] = row<7>(rs);
            c.mu[0] = row<8>(rs); c.mu[1] = row<9>(rs); c.mu[2] = row<10>(rs); c.mu[3] = row<11>(rs);
        }
        base += pitch;
    }
};
// PW (fp64, CS): the four wheels differ in C -- the per-wheel fit table goes to LDS and the step reads it from there
// (fit_horner4_lds); otherwise the handle's one set is pinned in VGPRs (pin_tire_fit).
// COMP (fp32, CS): state0 / terminal are [22][n], rows 12..21 the compensation terms of the state sum
// (VDYN_OPT_STATE_ROWS, StepEngine<float>::advance_state); trajectories keep 12 rows.
template <typename T, int K, int LAYOUT, bool DIAG, bool CS, bool TRAJ = true, int PW = 0, bool COMP = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 2)))
rollout_kernel(DevParams<T> P, int64_t n, int H, const T *__restrict__ state0,
               const T *__restrict__ ctrl, const int *__restrict__ path_id, int Pn, int chunk, T h,
               T *__restrict__ terminal, T *__restrict__ traj, int traj_stride,
               T *__restrict__ state_dot_out, T *__restrict__ outputs_out)
{
    if (CS) {                           // only the fitted chain reads them
        if (PW == 1) stage_tire_fit(P);
        else if (PW == 2) pin_tire_fit_axles(P);
        else pin_tire_fit(P);
    }
    constexpr int FS = PW == 1 ? 2 : PW == 2 ? 3 : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *tab = reinterpret_cast<T *>(smem_raw);

    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < n;
    const int64_t r = active ? gid : n - 1;  // idle lanes shadow the last rollout, stores masked

    typename StepEngine<T>::State X;
    constexpr int kRows = COMP ? 22 : 12;
#pragma unroll
    for (int i = 0; i < kRows; ++i) X.set(i, state0[(int64_t)i * n + r]);

    int pid = 0;
    if (LAYOUT != 0) pid = min(max(path_id[r], 0), Pn - 1);  // ids outside [0, P) are clamped, never read out of bounds

    T sd[10];
    Outputs18<T> o18;
    StepEngine<T> eng;
    constexpr int PRE = rollout_table_pre<K, LAYOUT, DIAG>() ? 1 : 0;
    if (!DIAG) eng.template init<true, CS>(P);

    RowWriter<T, 12> tw(traj, r, n, traj_stride);                  // TRAJ instances only (traj != nullptr: the launcher)
    auto dump = [&]() __attribute__((always_inline)) {
        if (tw.due()) {
            const auto row = tw.row();
            put_state12<T>(row, X);
            tw.next();
        }
    };
    for (int t0 = 0; t0 < H; t0 += chunk) {
        const int tc_n = min(chunk, H - t0);
        if (LAYOUT == 1) {
            __syncthreads();  // previous chunk fully consumed
            if (PRE) stage_control_table_pre<T>(tab, ctrl, Pn, H, t0, tc_n, eng);
            else stage_control_table<T, K>(tab, ctrl, Pn, H, t0, tc_n);
            __syncthreads();
        }
        // Step t + 1's controls are fetched before step t is integrated, so that their latency (LDS
        // or memory) hides behind a whole RK4 step instead of stalling the lone wave at its top:
        // -4.4 % with the table in LDS, -5 % with per-rollout controls in global memory (A/B at the
        // sustained clock, three runs each, +-0.2 %).
        auto fetch = [&](Ctrl<T, K> &c, int tc) __attribute__((always_inline)) {
            const int t = t0 + tc;
            if (LAYOUT == 0) c.set(P, ctrl + ((int64_t)t * K) * n + r, n);
            else if (PRE) c.set_pre(P, tab + ((int64_t)tc * Pn + pid) * 4);
            else if (LAYOUT == 1) c.set(P, tab + (int64_t)tc * K * Pn + pid, Pn);
            else c.set(P, ctrl + ((int64_t)pid * H + t) * K, 1);
        };
        Ctrl<T, K> c;
        fetch(c, 0);
        int tc = 0;
#ifdef VDYN_READER_ALL          // diagnostic build only (tools/isa/reader_all.hip): the reader for every k and precision
        if constexpr (LAYOUT == 0 && !DIAG && !TRAJ) {
#else
        if constexpr (LAYOUT == 0 && K == 2 && sizeof(T) == 4 && !DIAG && !TRAJ) {
#endif
            // per-rollout controls: rows read strictly in order through a running wave-uniform base (RowReader); the
            // trip's last load is row tc + 4, so the loop stops while that row exists and the one-step loop below
            // finishes the horizon.  k = 2 only: with k = 12 the twelve scalar row offsets do not fit the scalar file
            // beside the step's constants, and the fp64 k = 12 instance -- which has no register to spare, 256 VGPRs
            // + 172 AGPRs -- returned wrong x / y rows with the reader (tests/test_gpu_soak.py found it; fp32 k = 12
            // and every k = 2 instance were right): k = 12 keeps the per-lane addresses.  fp32 only: measured on one box,
            // fp32 0.1621 -> 0.1577 ms (65536 x 200), fp64 configs[1] 0.3905 -> 0.3923 (64 waves: nothing to gain from
            // issue slots, and the loads came back a little later).
            if (RowReader<T, K>::fits(n)) {
                RowReader<T, K> rd(ctrl, r, n, t0 + 1);
                Ctrl<T, K> c2;
                for (; tc + 4 < tc_n; tc += 4) {
                    rd.load(P, c2);
                    VDYN_FETCH_FENCE
                    eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
                    rd.load(P, c);
                    VDYN_FETCH_FENCE
                    eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
                    rd.load(P, c2);
                    VDYN_FETCH_FENCE
                    eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
                    rd.load(P, c);
                    VDYN_FETCH_FENCE
                    eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
                }
            }
        }
        if (!DIAG && !TRAJ) {
            // Four steps per trip: the control sets ping-pong (no copy), the loop's one taken branch -- tens of cycles
            // for a lone wave -- is paid every fourth step and the scheduler overlaps a step's tail with the next
            // one's head: 0.2011 (one per trip) -> 0.1965 (two) -> 0.1920 ms (four); eight is no faster in fp32 and
            // 7 % slower in fp64 (48 KB of loop body against the instruction cache).  Round 1's two-per-trip attempt
            // was slower because the step then still carried its rare path in the loop body.
            Ctrl<T, K> c2;
            for (; tc + 3 < tc_n; tc += 4) {
                fetch(c2, tc + 1);
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
                fetch(c, tc + 2);
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
                fetch(c2, tc + 3);
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
                fetch(c, min(tc + 4, tc_n - 1));
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
            }
        }
        if (!DIAG && TRAJ) {
            // two steps per trip, the control sets ping-pong: no copies of them per step (six `v_mov_b64` in the
            // one-step loop below, which is left with the last odd step and the diagnostic single step)
            Ctrl<T, K> c2;
            for (; tc + 1 < tc_n; tc += 2) {
                fetch(c2, tc + 1);
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
                dump();
                fetch(c, min(tc + 2, tc_n - 1));
                VDYN_FETCH_FENCE
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
                dump();
            }
        }
        for (; tc < tc_n; ++tc) {
            Ctrl<T, K> cn;
            fetch(cn, min(tc + 1, tc_n - 1));

            if (DIAG) {
                T s[10], ax = X.get(10), ay = X.get(11);
#pragma unroll
                for (int i = 0; i < 10; ++i) s[i] = X.get(i);
                rk4_advance<T, K == 2, true, CS>(P, s, ax, ay, c.delta, c.tq, c.mu, h, sd, &o18);
#pragma unroll
                for (int i = 0; i < 10; ++i) X.set(i, s[i]);
                X.set(10, ax);
                X.set(11, ay);
            } else {
                eng.template advance_state<K == 2, CS, PRE, FS, COMP>(P, X, c.delta, c.tq, c.mu, h, c.sd0, c.cd0);
            }
            c = cn;
            if (TRAJ && traj != nullptr) dump();                    // DIAG launches come here without a trajectory
        }
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < kRows; ++i) terminal[(int64_t)i * n + r] = X.get(i);
        if (DIAG) {
            if (state_dot_out != nullptr && H > 0) {
#pragma unroll
                for (int i = 0; i < 10; ++i) state_dot_out[(int64_t)i * n + r] = sd[i];
            }
            if (outputs_out != nullptr && H > 0) {
#pragma unroll
                for (int i = 0; i < 18; ++i) outputs_out[(int64_t)i * n + r] = o18.v[i];
            }
        }
    }
}

// Lattice-driven rollout: every rollout follows ITS OWN cubic spiral, the path the conformal-lattice planner
// optimised for it (path_optimizer.py:31-88).  spiral [n][3] = (p1, p2, sf) per rollout -- exactly the `params`
// output of vdyn_plan_lattice_* -- is mapped to the curvature polynomial kappa(s) = a + b s + c s^2 + d s^3
// with the reference's own formulas (path_optimizer.py:149-154, p0 = p3 = 0), and step t steers with
//   tan(delta_t) = L kappa(min(U0 t dt, sf)),  clipped to +- tan(max_steer)   (stanley_controller.py:128)
// where U0 is the rollout's initial speed (the arc length a vehicle holding its speed has covered) and L
// the wheelbase: the kinematic-bicycle steering angle of that curvature.
// No control bytes come from memory at all: 12 B per ROLLOUT instead of 8 B per step, and
// (sin delta, cos delta) are exact from the tangent -- cos = rsq(1 + tan^2) -- with no arctangent.
template <typename T, bool CS, bool TRAJ, int FS = 1>     // FS = 3: one fit per axle (pin_fit)
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 2)))
rollout_spiral_kernel(DevParams<T> P, int64_t n, int H, const T *__restrict__ state0, const T *__restrict__ spiral,
                      T wheelbase, T tan_max, T torque, T h, T *__restrict__ terminal, T *__restrict__ traj,
                      int traj_stride)
{
    if (CS) pin_fit<FS>(P);             // only the fitted chain reads them
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < n;
    const int64_t r = active ? gid : n - 1;
    typename StepEngine<T>::State X;
#pragma unroll
    for (int i = 0; i < 12; ++i) X.set(i, state0[(int64_t)i * n + r]);
    StepEngine<T> eng;
    eng.template init<true, CS>(P);

    const T p1 = spiral[3 * r], p2 = spiral[3 * r + 1], sf = spiral[3 * r + 2];
    // path_optimizer.py:149-154 with p0 = p3 = 0 (a = 0), pre-multiplied by the wheelbase: q = tan(delta)
    const T isf = T(1) / sf;
    const T qb = wheelbase * (-(T(-9) * p1 + T(4.5) * p2) * isf);
    const T qc = wheelbase * ((T(-22.5) * p1 + T(18) * p2) * isf * isf);
    const T qd = wheelbase * (-(T(-13.5) * p1 + T(13.5) * p2) * isf * isf * isf);
    const T ds = X.get(0) * h;                                    // arc length per step at the initial speed
    const T tq[4] = {torque, torque, torque, torque};

    auto one_step = [&](int t) __attribute__((always_inline)) {
        T s = ds * (T)t;
        s = s < sf ? s : sf;
        T q = s * fma_t(s, fma_t(s, qd, qc), qb);
        q = q < -tan_max ? -tan_max : q;                           // np.clip semantics: NaN passes through
        q = q > tan_max ? tan_max : q;
        const T cd = Math<T, false>::rsqrt(fma_t(q, q, T(1)));
        const T sd = q * cd;
        const T delta[4] = {q, q, T(0), T(0)};                     // PRE = 2: the tangent stands in for the angle
        eng.template advance_state<true, CS, 2, FS>(P, X, delta, tq, P.mu, h, sd, cd);
    };
    int t = 0;
    if (!TRAJ) {                                                   // four steps per trip (see rollout_kernel)
        for (; t + 3 < H; t += 4) {
            one_step(t);
            one_step(t + 1);
            one_step(t + 2);
            one_step(t + 3);
        }
    }
    RowWriter<T, 12> tw(traj, r, n, traj_stride);                  // TRAJ instances only (traj != nullptr: the launcher)
    for (; t < H; ++t) {
        one_step(t);
        if (TRAJ && tw.due()) {
            const auto row = tw.row();
            put_state12<T>(row, X);
            tw.next();
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 12; ++i) terminal[(int64_t)i * n + r] = X.get(i);
    }
}

// Heterogeneous fleet: every rollout carries a vehicle-class id; the classes' constants (masses,
// geometry, static loads, Pacejka B / C per wheel: one DevParams<T> per class) are staged
// through LDS once per workgroup and each lane copies its class's row into registers.  (With one
// class -- every BASELINE configuration -- the constants are wave-uniform and the kernel above
// keeps them in SGPRs instead, which is cheaper than any LDS read.)
// One class of the fleet table, as plain T words.  fp32: the whole DevParams (its per-wheel fits are 36 floats).  fp64:
// ONE set of fit coefficients + the rest -- 68 doubles of per-wheel fits per class would neither fit LDS for 256
// classes nor a lane's registers, so an fp64 class takes the fitted chain only when its four wheels share C
// (build_fleet_table) and the lane carries that set as column 0 of its DevParams.
template <typename T> struct FleetRow;
template <> struct FleetRow<float> {
    DevParams<float> p;
    __device__ __forceinline__ void unpack(DevParams<float> &P) const { P = p; }
    void pack(const DevParams<float> &d) { p = d; }
};
template <> struct FleetRow<double> {
    double W[kTireFitDeg64 + 1];
    DevCore<double> core;
    __device__ __forceinline__ void unpack(DevParams<double> &P) const
    {
        static_cast<DevCore<double> &>(P) = core;
#pragma unroll
        for (int i = 0; i <= kTireFitDeg64; ++i) P.W[i][0] = W[i];
    }
    void pack(const DevParams<double> &d)
    {
        core = static_cast<const DevCore<double> &>(d);
        for (int i = 0; i <= kTireFitDeg64; ++i) W[i] = d.W[i][0];
    }
};
template <typename T> constexpr int dev_params_len() { return (int)(sizeof(FleetRow<T>) / sizeof(T)); }

template <typename T, int K, int LAYOUT, bool CS>
__global__ void __launch_bounds__(kBlock)
rollout_fleet_kernel(const T *__restrict__ fleet, int V, const int *__restrict__ vehicle_id, int64_t n, int H,
                     const T *__restrict__ state0, const T *__restrict__ ctrl, const int *__restrict__ path_id,
                     int Pn, int chunk, T h, T *__restrict__ terminal, T *__restrict__ traj, int traj_stride)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int NP = dev_params_len<T>();
    T *lds_fleet = reinterpret_cast<T *>(smem_raw);
    T *tab = lds_fleet + (int64_t)V * NP;
    for (int i = threadIdx.x; i < V * NP; i += kBlock) lds_fleet[i] = fleet[i];
    __syncthreads();

    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < n;
    const int64_t r = active ? gid : n - 1;
    DevParams<T> P;
    {
        const int vid = min(max(vehicle_id[r], 0), V - 1);
        FleetRow<T> row;
        T *dst = reinterpret_cast<T *>(&row);
#pragma unroll
        for (int i = 0; i < NP; ++i) dst[i] = lds_fleet[vid * NP + i];
        row.unpack(P);
    }
    T s[10], ax, ay;
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = state0[(int64_t)i * n + r];
    ax = state0[10 * n + r];
    ay = state0[11 * n + r];
    int pid = 0;
    if (LAYOUT != 0) pid = min(max(path_id[r], 0), Pn - 1);
    StepEngine<T> eng;
    eng.template init<false, CS>(P);   // per-lane constants

    RowWriter<T, 12> tw(traj, r, n, traj_stride);
    for (int t0 = 0; t0 < H; t0 += chunk) {
        const int tc_n = min(chunk, H - t0);
        if (LAYOUT == 1) {
            __syncthreads();
            stage_control_table<T, K>(tab, ctrl, Pn, H, t0, tc_n);
            __syncthreads();
        }
        auto fetch = [&](Ctrl<T, K> &c, int tc) __attribute__((always_inline)) {
            const int t = t0 + tc;
            if (LAYOUT == 0) c.set(P, ctrl + ((int64_t)t * K) * n + r, n);
            else if (LAYOUT == 1) c.set(P, tab + (int64_t)tc * K * Pn + pid, Pn);
            else c.set(P, ctrl + ((int64_t)pid * H + t) * K, 1);
        };
        Ctrl<T, K> c;
        fetch(c, 0);
        for (int tc = 0; tc < tc_n; ++tc) {
            Ctrl<T, K> cn;
            fetch(cn, min(tc + 1, tc_n - 1));          // behind this step (see rollout_kernel)
            eng.template advance<K == 2, CS>(P, s, ax, ay, c.delta, c.tq, c.mu, h);
            c = cn;
            if (traj != nullptr && tw.due()) {         // wave-uniform
                const auto row = tw.row();
                row.template put<0>(s[0]); row.template put<1>(s[1]); row.template put<2>(s[2]); row.template put<3>(s[3]);
                row.template put<4>(s[4]); row.template put<5>(s[5]); row.template put<6>(s[6]); row.template put<7>(s[7]);
                row.template put<8>(s[8]); row.template put<9>(s[9]); row.template put<10>(ax); row.template put<11>(ay);
                tw.next();
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 10; ++i) terminal[(int64_t)i * n + r] = s[i];
        terminal[10 * n + r] = ax;
        terminal[11 * n + r] = ay;
    }
}

// Wheel-parallel rollout (vdyn_quad.hpp): four adjacent lanes per rollout, 64 rollouts per
// 256-thread workgroup.  Same interface as rollout_kernel minus the diagnostics.
// TRAJ: the launch writes trajectory rows (an instance of its own, as for rollout_kernel: the test alone is a taken
// branch per step otherwise).
template <typename T, int K, int LAYOUT, bool CS, bool TRAJ>
__global__ void __launch_bounds__(kBlock)
rollout_quad_kernel(DevParams<T> P, int64_t n, int H, const T *__restrict__ state0,
                    const T *__restrict__ ctrl, const int *__restrict__ path_id, int Pn, int chunk, T h,
                    T *__restrict__ terminal, T *__restrict__ traj, int traj_stride)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *tab = reinterpret_cast<T *>(smem_raw);

    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int q = (int)(gid & 3);
    const bool active = (gid >> 2) < n;
    const int64_t r = active ? (gid >> 2) : n - 1;
    const WheelLane<T> L = make_wheel_lane<T>(P, q);
    QuadEngine<T> qe;
    qe.init(P, L, q);

    QuadState<T> s;
    s.U = state0[r];
    s.V = state0[n + r];
    s.wz = state0[2 * n + r];
    s.w = state0[(int64_t)(3 + q) * n + r];
    s.yaw = state0[7 * n + r];
    s.x = state0[8 * n + r];
    s.y = state0[9 * n + r];
    T ax = state0[10 * n + r], ay = state0[11 * n + r];
    const T mu_k2 = pick_wheel(P.mu[0], P.mu[1], P.mu[2], P.mu[3], q);

    int pid = 0;
    if (LAYOUT != 0) pid = min(max(path_id[r], 0), Pn - 1);

    TrajCursor<T> tcur(TRAJ ? traj : nullptr, r, n, traj_stride);
    for (int t0 = 0; t0 < H; t0 += chunk) {
        const int tc_n = min(chunk, H - t0);
        if (LAYOUT == 1) {
            __syncthreads();
            stage_control_table<T, K>(tab, ctrl, Pn, H, t0, tc_n);
            __syncthreads();
        }
        // this lane's controls of step tc; step tc + 1's are fetched before step tc is integrated (see rollout_kernel)
        struct QC { T delta, tq, mu; };
        auto fetch = [&](int tc) __attribute__((always_inline)) {
            const int t = t0 + tc;
            const T *c;
            int64_t stride;
            if (LAYOUT == 0) { c = ctrl + ((int64_t)t * K) * n + r; stride = n; }
            else if (LAYOUT == 1) { c = tab + (int64_t)tc * K * Pn + pid; stride = Pn; }
            else { c = ctrl + ((int64_t)pid * H + t) * K; stride = 1; }
            QC o;
            if (K == 2) {
                o.delta = L.front ? c[0] : T(0);          // drive.py:143: [d, d, 0, 0]
                o.tq = c[stride];
                o.mu = mu_k2;
            } else {
                o.delta = c[(int64_t)q * stride];
                o.tq = c[(int64_t)(4 + q) * stride];
                o.mu = c[(int64_t)(8 + q) * stride];
            }
            return o;
        };
        QC cur = fetch(0);
        for (int tc = 0; tc < tc_n; ++tc) {
            const QC nxt = fetch(min(tc + 1, tc_n - 1));
            VDYN_FETCH_FENCE
            qe.template advance<CS>(P, L, s, ax, ay, cur.delta, cur.tq, cur.mu, h);
            cur = nxt;

            if (TRAJ && tcur.due()) {
                if (active) {
                    T *row = tcur.row;
                    row[(int64_t)(3 + q) * n] = s.w;
                    if (q == 0) {
                        row[0] = s.U; row[n] = s.V; row[2 * n] = s.wz; row[7 * n] = s.yaw;
                        row[8 * n] = s.x; row[9 * n] = s.y; row[10 * n] = ax; row[11 * n] = ay;
                    }
                }
                tcur.next();
            }
        }
    }
    if (active) {
        terminal[(int64_t)(3 + q) * n + r] = s.w;
        if (q == 0) {
            terminal[r] = s.U; terminal[n + r] = s.V; terminal[2 * n + r] = s.wz; terminal[7 * n + r] = s.yaw;
            terminal[8 * n + r] = s.x; terminal[9 * n + r] = s.y; terminal[10 * n + r] = ax;
            terminal[11 * n + r] = ay;
        }
    }
}

// One derivative evaluation per lane: vehicle_model.py:220-425 with its full
// return list (:425).
template <typename T>
__global__ void __launch_bounds__(kBlock)
planar_model_kernel(DevParams<T> P, int64_t n, const T *__restrict__ state, const T *__restrict__ ctrl12,
                    const T *__restrict__ acc_prev, T *__restrict__ state_dot, T *__restrict__ aux,
                    T *__restrict__ outputs, T *__restrict__ acc)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    T s[10], k[10], axc, ayc;
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = state[(int64_t)i * n + r];
    Ctrl<T, 12> c;
    c.set(P, ctrl12 + r, n);
    // a single evaluation is latency-bound on the launch, not on math: full-range path
    StepInv<T> inv;
    bool ok = true;
    make_step_inv<T, false, true>(P, c.delta, c.tq, c.mu, acc_prev[r], acc_prev[n + r], inv, ok);
    Outputs18<T> o;
    T sy, cy;
    Math<T, true>::sincos(s[7], &sy, &cy, ok);
    planar_deriv<T, false, true, true, false>(P, inv, s, sy, cy, k, axc, ayc, &o);
#pragma unroll
    for (int i = 0; i < 10; ++i) state_dot[(int64_t)i * n + r] = k[i];
    acc[r] = axc;
    acc[n + r] = ayc;
    if (aux != nullptr) {
        aux[r] = s[0] * cy - s[1] * sy;          // vx, :410
        aux[n + r] = s[1] * sy + s[0] * cy;      // vy, :411 (the reference's own sin/cos mix-up)
        aux[2 * n + r] = axc * cy - ayc * sy;    // ax, :415
        aux[3 * n + r] = axc * sy + ayc * cy;    // ay, :416
    }
    if (outputs != nullptr) {
#pragma unroll
        for (int i = 0; i < 18; ++i) outputs[(int64_t)i * n + r] = o.v[i];
    }
}

// Config-5 MPC selection.  Block = one ego; lanes stride over the C shared
// candidates; cost and (min, argmin) never leave the chip until the final pair.
// Workgroup size and occupancy target.  fp32: 512 threads at 2 waves per SIMD -- the packed step
// (vdyn_packed.hpp) holds its coefficient pairs in ~200 VGPRs and spilled 300 bytes per lane under
// the 128-register cap of 4 waves per SIMD.  fp64: 256 threads, one wave per SIMD with the full
// register file (no spill; at 128 registers the fp64 step spilled 700 bytes per lane).
template <typename T> constexpr int mpc_block_max() { return sizeof(T) == 4 ? 512 : 256; }
template <typename T> constexpr int mpc_waves_per_simd() { return sizeof(T) == 4 ? 2 : 1; }

// Candidate table for the selection kernel: cand [H][2][C] -> cand4 [H][C][4] = (delta, torque, sin delta,
// cos delta).  The candidates are shared by all E egos, so the steering angle's (sin, cos) is evaluated
// once per table entry here instead of E times in the selection kernel, by the function the step itself
// would call (bit-identical results); a lane then reads its step's controls as one 16-byte load.
template <typename T>
__global__ void __launch_bounds__(kBlock)
mpc_prepare_kernel(DevParams<T> P, int C, int H, const T *__restrict__ cand, T *__restrict__ cand4)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;    // = t * C + c
    StepEngine<T> eng;
    eng.init(P);
    if (i >= (int64_t)H * C) return;
    const int t = (int)(i / C), c = (int)(i - (int64_t)t * C);
    const T d = cand[((int64_t)t * 2) * C + c], tq = cand[((int64_t)t * 2 + 1) * C + c];
    T sd, cd;
    eng.steer_sincos(d, sd, cd);
    T *o = cand4 + i * 4;
    o[0] = d; o[1] = tq; o[2] = sd; o[3] = cd;
}

template <typename T, bool CS, int FS = 1>
__global__ void __launch_bounds__(mpc_block_max<T>(), mpc_waves_per_simd<T>())
mpc_argmin_kernel(DevParams<T> P, int E, int C, int H, const T *__restrict__ ego,
                  const T *__restrict__ cand4, const T *__restrict__ goal, T h, T w_delta,
                  T *__restrict__ best_cost, int *__restrict__ best_idx, T *__restrict__ cost_all)
{
    if (CS) pin_fit<FS>(P);             // only the fitted chain reads them
    __shared__ T s_cost[16];
    __shared__ int s_idx[16];
    const int e = blockIdx.x;
    const T inf = T(INFINITY);
    constexpr int kNone = 0x7fffffff;

    typename StepEngine<T>::State X0;
#pragma unroll
    for (int i = 0; i < 12; ++i) X0.set(i, ego[(int64_t)i * E + e]);
    const T gx = goal[e], gy = goal[(int64_t)E + e];

    T bc = inf;
    int bi = kNone;
    StepEngine<T> eng;
    eng.template init<true, CS>(P);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        typename StepEngine<T>::State X = X0;
        T dsum = T(0);
        Ctrl<T, 2> cc;                         // step t + 1's controls are fetched behind step t (see rollout_kernel)
        if (H > 0) cc.set_pre(P, cand4 + (int64_t)c * 4);
        int t = 0;
        Ctrl<T, 2> c2;
        for (; t + 1 < H; t += 2) {            // two steps per trip, control sets ping-pong (see rollout_kernel)
            c2.set_pre(P, cand4 + ((int64_t)(t + 1) * C + c) * 4);
            VDYN_FETCH_FENCE
            eng.template advance_state<true, CS, 1, FS>(P, X, cc.delta, cc.tq, cc.mu, h, cc.sd0, cc.cd0);
            dsum += cc.delta[0] * cc.delta[0];
            cc.set_pre(P, cand4 + ((int64_t)min(t + 2, H - 1) * C + c) * 4);
            VDYN_FETCH_FENCE
            eng.template advance_state<true, CS, 1, FS>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
            dsum += c2.delta[0] * c2.delta[0];
        }
        if (t < H) {
            eng.template advance_state<true, CS, 1, FS>(P, X, cc.delta, cc.tq, cc.mu, h, cc.sd0, cc.cd0);
            dsum += cc.delta[0] * cc.delta[0];
        }
        const T dx = X.get(8) - gx, dy = X.get(9) - gy;
        const T cost = sqrt_t(dx * dx + dy * dy) + w_delta * dsum;
        if (cost_all != nullptr) cost_all[(int64_t)e * C + c] = cost;
        if (cost < bc) { bc = cost; bi = c; }  // strict '<': lowest index wins ties; NaN/inf never win
    }
    // wave-level (cost, idx) lexicographic min
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T oc = __shfl_xor(bc, off);
        const int oi = __shfl_xor(bi, off);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_cost[wave] = bc; s_idx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            const T oc = s_cost[w];
            const int oi = s_idx[w];
            if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
        }
        best_cost[e] = bc;
        best_idx[e] = (bi == kNone) ? -1 : bi;
    }
}

// The same selection with the EGOS on the lanes: a wave = 64 consecutive egos x a chunk of KC consecutive candidates,
// run one after the other.  The candidate is then wave-uniform: its controls (delta, torque, sin delta, cos delta) are
// one scalar load per step -- issued a step ahead, no vector-memory wait in the step at all -- where the kernel above
// has every lane fetch its own 16-byte entry and wait for it (28 % of its cycles, profiles/r02f), and the running
// (min, argmin) of a lane needs no cross-lane reduction.  Each wave writes its lanes' best (cost, index) of the chunk
// to part [chunk][E]; mpc_reduce_kernel takes the minimum over the chunks in ascending order with a strict '<', so
// the lowest index still wins ties and a NaN / inf cost never wins -- exactly the scan of the kernel above.
//   grid = ceil(E / 64) * ceil(C / KC) waves, one wave per workgroup; KC is chosen by the launcher so that the grid is
//   about two waves per SIMD (fp32; one for fp64) -- what the register budget of the packed step allows.
template <typename T, bool CS, int FS = 1>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, sizeof(T) == 4 ? 2 : 1)))
mpc_argmin_lanes_kernel(DevParams<T> P, int E, int C, int H, int KC, const T *__restrict__ ego,
                        const T *__restrict__ cand4, const T *__restrict__ goal, T h, T w_delta,
                        T *__restrict__ part_cost, int *__restrict__ part_idx, T *__restrict__ cost_all)
{
    if (CS) pin_fit<FS>(P);             // only the fitted chain reads them
    const int ngroups = (E + 63) / 64;
    const int grp = (int)(blockIdx.x % (unsigned)ngroups), chunk = (int)(blockIdx.x / (unsigned)ngroups);
    const int e_raw = grp * 64 + (int)threadIdx.x;
    const bool active = e_raw < E;
    const int e = active ? e_raw : E - 1;       // idle lanes shadow the last ego, stores masked
    constexpr int kNone = 0x7fffffff;

    typename StepEngine<T>::State X0;
#pragma unroll
    for (int i = 0; i < 12; ++i) X0.set(i, ego[(int64_t)i * E + e]);
    const T gx = goal[e], gy = goal[(int64_t)E + e];
    StepEngine<T> eng;
    eng.template init<true, CS>(P);

    T bc = T(INFINITY);
    int bi = kNone;
    const int c0 = chunk * KC, c1 = min(c0 + KC, C);
    for (int c = c0; c < c1; ++c) {             // wave-uniform
        typename StepEngine<T>::State X = X0;
        T dsum = T(0);
        const T *tab = cand4 + (int64_t)c * 4;  // entry of step t at tab + t * C * 4: a scalar (uniform) address
        const int64_t ts = (int64_t)C * 4;
        Ctrl<T, 2> cc, c2;                      // step t + 1's controls are fetched behind step t (see rollout_kernel)
        if (H > 0) cc.set_pre(P, tab);
        int t = 0;
        for (; t + 1 < H; t += 2) {             // two steps per trip, control sets ping-pong
            c2.set_pre(P, tab + (int64_t)(t + 1) * ts);
            VDYN_FETCH_FENCE
            eng.template advance_state<true, CS, 1, FS>(P, X, cc.delta, cc.tq, cc.mu, h, cc.sd0, cc.cd0);
            dsum += cc.delta[0] * cc.delta[0];
            cc.set_pre(P, tab + (int64_t)min(t + 2, H - 1) * ts);
            VDYN_FETCH_FENCE
            eng.template advance_state<true, CS, 1, FS>(P, X, c2.delta, c2.tq, c2.mu, h, c2.sd0, c2.cd0);
            dsum += c2.delta[0] * c2.delta[0];
        }
        if (t < H) {
            eng.template advance_state<true, CS, 1, FS>(P, X, cc.delta, cc.tq, cc.mu, h, cc.sd0, cc.cd0);
            dsum += cc.delta[0] * cc.delta[0];
        }
        const T dx = X.get(8) - gx, dy = X.get(9) - gy;
        const T cost = sqrt_t(dx * dx + dy * dy) + w_delta * dsum;
        if (cost_all != nullptr && active) cost_all[(int64_t)e * C + c] = cost;
        if (cost < bc) { bc = cost; bi = c; }   // strict '<': lowest index wins ties; NaN/inf never win
    }
    if (active) {
        part_cost[(int64_t)chunk * E + e] = bc;
        part_idx[(int64_t)chunk * E + e] = bi;
    }
}

// part [nchunks][E] -> (best_cost, best_idx)[E]: chunks in ascending order (= ascending candidate index), strict '<'.
// A workgroup = 64 egos x 4 waves; wave w scans its quarter of the chunks with sixteen loads in flight (a single scan
// of 128 chunks, eight at a time, took 9 us: sixteen dependent L2 round trips), the quarters meet in LDS in wave order.
template <typename T>
__global__ void __launch_bounds__(kBlock)
mpc_reduce_kernel(int E, int nchunks, const T *__restrict__ part_cost, const int *__restrict__ part_idx,
                  T *__restrict__ best_cost, int *__restrict__ best_idx)
{
    __shared__ T s_cost[kBlock / 64 - 1][64];
    __shared__ int s_idx[kBlock / 64 - 1][64];
    constexpr int kNone = 0x7fffffff, kFly = 16, kWaves = kBlock / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e_raw = (int)blockIdx.x * 64 + lane, e = min(e_raw, E - 1);
    const int per = (nchunks + kWaves - 1) / kWaves, k_lo = wave * per, k_hi = min(k_lo + per, nchunks);
    T bc = T(INFINITY);
    int bi = kNone;
    for (int k0 = k_lo; k0 < k_hi; k0 += kFly) {
        T pc[kFly];
        int pi[kFly];
#pragma unroll
        for (int k = 0; k < kFly; ++k) {
            const int kk = min(k0 + k, nchunks - 1);
            pc[k] = part_cost[(int64_t)kk * E + e];
            pi[k] = part_idx[(int64_t)kk * E + e];
        }
#pragma unroll
        for (int k = 0; k < kFly; ++k)
            if (k0 + k < k_hi && pc[k] < bc) { bc = pc[k]; bi = pi[k]; }
    }
    if (wave > 0) { s_cost[wave - 1][lane] = bc; s_idx[wave - 1][lane] = bi; }
    __syncthreads();
    if (wave == 0 && e_raw < E) {
#pragma unroll
        for (int w = 0; w < kWaves - 1; ++w)
            if (s_cost[w][lane] < bc) { bc = s_cost[w][lane]; bi = s_idx[w][lane]; }
        best_cost[e] = bc;
        best_idx[e] = bi == kNone ? -1 : bi;
    }
}

// Auxiliary waypoint tables for the controllers when the P tables do not fit LDS: segment lengths
// seg[j][P] and one bounding circle per 32 waypoints bnd[b][4][P], both TRANSPOSED (table index
// fastest) so that the lanes of a wave -- vehicles with consecutive tables -- read neighbouring
// words.  aux = seg (Wmax * P) followed by bnd (nb * 4 * P).
// One workgroup per tile of kAuxTP tables x kAuxTJ waypoints: the tile is read along the tables'
// own rows (contiguous), kept in LDS, and written out table-index-fastest (contiguous again).
constexpr int kAuxTP = 32;                 // tables per tile
constexpr int kAuxTJ = 2 * kWpBlock;       // waypoints per tile = two bounding circles per table

//   seg [j][P], bnd [b][4][P]   (transposed, read in place by the lanes).  Tables that fit LDS need none of this:
//   closed_loop_kernel builds its LDS image from the tables alone.
template <typename T>
__global__ void __launch_bounds__(kBlock)
waypoint_aux_kernel(const T *__restrict__ wp, int Wmax, const int *__restrict__ wcount, int Pn, T *__restrict__ aux)
{
    __shared__ T tx[kAuxTP][kAuxTJ + 2], ty[kAuxTP][kAuxTJ + 2];   // column 0 = the waypoint before the tile
    const int nb = (Wmax + kWpBlock - 1) / kWpBlock;
    const int64_t seg_len = (int64_t)Wmax * Pn;
    T *seg = aux, *bnd = aux + seg_len;
    const int64_t ss_j = Pn, ss_p = 1;         // seg[j * ss_j + p * ss_p]
    const int64_t bs = Pn, bp = 1;             // bnd[(4 b + c) * bs + p * bp]
    const int tiles_j = (Wmax + kAuxTJ - 1) / kAuxTJ;
    const int p0 = (blockIdx.x / tiles_j) * kAuxTP, j0 = (blockIdx.x % tiles_j) * kAuxTJ;
    // read: thread -> (table row, waypoint) with the waypoint fastest
    for (int i = threadIdx.x; i < kAuxTP * (kAuxTJ + 1); i += kBlock) {
        const int pp = i / (kAuxTJ + 1), jj = i - pp * (kAuxTJ + 1);      // jj = 0 is waypoint j0 - 1
        const int p = p0 + pp, j = j0 + jj - 1;
        T x = T(0), y = T(0);
        if (p < Pn && j >= 0 && j < Wmax) {
            const T *q = wp + ((int64_t)p * Wmax + j) * 2;
            x = q[0];
            y = q[1];
        }
        tx[pp][jj] = x;
        ty[pp][jj] = y;
    }
    __syncthreads();
    // segment lengths: thread -> (waypoint, table) with the table fastest
    for (int i = threadIdx.x; i < kAuxTP * kAuxTJ; i += kBlock) {
        const int jj = i / kAuxTP, pp = i - jj * kAuxTP;
        const int p = p0 + pp, j = j0 + jj;
        if (p < Pn && j < Wmax)
            seg[j * ss_j + p * ss_p] = j == 0 ? T(0) : segment_length<T>(tx[pp][jj], ty[pp][jj], tx[pp][jj + 1], ty[pp][jj + 1]);
    }
    // bounding circles: one thread per (table, block of 32 waypoints) of the tile
    if (threadIdx.x < kAuxTP * (kAuxTJ / kWpBlock)) {
        const int bb = threadIdx.x / kAuxTP, pp = threadIdx.x - bb * kAuxTP;
        const int p = p0 + pp, b = j0 / kWpBlock + bb;
        if (p < Pn && b < nb) {
            const int W = min(max(wcount[p], 1), Wmax);
            const int lo = b * kWpBlock, hi = min(lo + kWpBlock, W);
            T x0 = T(INFINITY), x1 = -T(INFINITY), y0 = T(INFINITY), y1 = -T(INFINITY);
            for (int j = lo; j < hi; ++j) {
                const T wx = tx[pp][j - j0 + 1], wy = ty[pp][j - j0 + 1];
                x0 = wx < x0 ? wx : x0; x1 = wx > x1 ? wx : x1;
                y0 = wy < y0 ? wy : y0; y1 = wy > y1 ? wy : y1;
            }
            const T cx = T(0.5) * (x0 + x1), cy = T(0.5) * (y0 + y1);
            T r2 = T(0);
            for (int j = lo; j < hi; ++j) {
                const T dx = tx[pp][j - j0 + 1] - cx, dy = ty[pp][j - j0 + 1] - cy;
                const T d2 = dx * dx + dy * dy;
                r2 = d2 > r2 ? d2 : r2;
            }
            const bool empty = lo >= hi;                                   // a block past the table's end: never entered
            T *o = bnd + (int64_t)4 * b * bs + p * bp;
            o[0] = empty ? T(0) : cx;
            o[bs] = empty ? T(0) : cy;
            o[2 * bs] = empty ? T(0) : Lib<T>::sqrt(r2) * T(1.00001) + T(1e-30);     // never smaller than the true radius
            o[3 * bs] = T(0);
        }
    }
}

// fp32 controllers search the lookahead point on the CUMULATIVE arc length (vdyn_controls.hpp): turn the
// segment lengths of waypoint_aux_kernel into running sums, in place.  One wave per table: every lane sums
// its chunk, a wave scan gives the chunk offsets, a second pass writes the running sums.
template <typename T>
__global__ void __launch_bounds__(64)
waypoint_cumsum_kernel(int Wmax, int Pn, T *__restrict__ aux)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    const int64_t sj = Pn, sp = 1;
    T *seg = aux + (int64_t)p * sp;
    const int chunk = (Wmax + 63) / 64;
    const int j0 = lane * chunk, j1 = min(j0 + chunk, Wmax);
    T sum = T(0);
    for (int j = j0; j < j1; ++j) sum += seg[(int64_t)j * sj];
    T incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    T run = incl - sum;                                             // exclusive offset of this lane's chunk
    for (int j = j0; j < j1; ++j) {
        run += seg[(int64_t)j * sj];
        seg[(int64_t)j * sj] = run;
    }
}

template <typename T>
size_t waypoint_aux_len(int P, int Wmax)
{
    const size_t nb = (size_t)(Wmax + kWpBlock - 1) / kWpBlock;
    return (size_t)Wmax * P + nb * 4 * P;
}

// Closed-loop rollout: the sub-step loop of drive.py:114-151 without the planner.  Every
// `ctrl_every` steps the lane runs the Stanley + PID + filter update (vdyn_controls.hpp)
// against its waypoint table, holds the commands in between (zero-order hold, drive.py:128)
// and integrates with the same RK4 step as the open-loop kernel.
//   WPLDS: all P waypoint tables fit the LDS budget and are staged there once per workgroup;
//          otherwise the scan reads them through L2.
//   log (nullable): [H][16][n] rows state12, delta, torque, target index, crosstrack error.
//   DATALOG: datalog [H][45][n] = the 45 columns Car.DataLog receives per sub-step
//            (drive.py:145-151; names in plots.py:19-27): t, state x10, state_dot x10, delta,
//            torque x4, outputs x18, crosstrack error.
//   LOG: the launch writes the 16-row log (an instance of its own: see rollout_kernel's TRAJ)
//   FS:  how the fp64 fitted chain gets its coefficients without the DataLog (pin_fit): 1 one set, 3 one per axle
template <typename T, bool CS, bool WPLDS, bool DATALOG, bool LOG, int FS = 1>
__global__ void __launch_bounds__(kBlock)
closed_loop_kernel(DevParams<T> P, CtrlGains<T> G, int64_t n, int H, int ctrl_every, int phase,
                   const T *__restrict__ state0, const T *__restrict__ cstate0, const T *__restrict__ wp,
                   int Wmax, const int *__restrict__ wcount, const int *__restrict__ path_id, int Pn, T h,
                   T *__restrict__ terminal, T *__restrict__ cstate, T *__restrict__ log,
                   T *__restrict__ datalog, const T *__restrict__ aux)
{
    // fp64 with the DataLog: the step that also averages the 28 diagnostics has no 34 VGPRs left for a pinned fit (it
    // spilled 250-380 bytes per lane to scratch) -- it reads the table from LDS instead (fit_horner4_lds)
    constexpr bool kFitLds = DATALOG && sizeof(T) == 8;
    if (CS) {                           // only the fitted chain reads them
        if (kFitLds) stage_tire_fit(P);
        else pin_fit<FS>(P);
    }
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS image (ClosedLoopLds, vdyn_controls.hpp): x rows, y rows (padded to whole 8-waypoint sub-blocks, 16-byte
    // aligned, rows of different paths 4 banks apart: lanes of one wave follow different paths and read the same
    // waypoint index together), segment lengths / cumulative arcs, one bounding circle per 32 and per 8 waypoints
    const ClosedLoopLds<T> LL(Pn, Wmax);
    const int nbmax = LL.nb;
    T *lds = reinterpret_cast<T *>(smem_raw);
    // tables in LDS: every workgroup builds the whole image for itself, from the caller's tables alone
    PhaseClock pc0;
    VDYN_PHASE_START(pc0);
    if (WPLDS) build_closed_loop_lds<T, kBlock>(lds, LL, wp, Wmax, wcount, Pn);
    VDYN_PHASE_LAP(pc0, 10);     // LDS image
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < n;
    const int64_t r = active ? gid : n - 1;

    typename StepEngine<T>::State X;
#pragma unroll
    for (int i = 0; i < 12; ++i) X.set(i, state0[(int64_t)i * n + r]);
    CtrlState<T> c;
    c.x_del = cstate0[r];
    c.total = cstate0[n + r];
    c.prev_vel = cstate0[2 * n + r];
    c.target = cstate0[3 * n + r];
    c.delta = cstate0[4 * n + r];
    c.tau = cstate0[5 * n + r];
    c.idx = -1;
    c.cte = T(0);

    const int pid = min(max(path_id[r], 0), Pn - 1);
    Waypoints<T, WPLDS> w;
    w.base = WPLDS ? lds + LL.xs + (int64_t)pid * LL.ws : wp + (int64_t)pid * Wmax * 2;
    w.yo = WPLDS ? (int)(LL.ys - LL.xs) : 1;
    // not in LDS: the transposed global tables of waypoint_aux_kernel<T> (nullptr: plain full scan)
    w.seg = WPLDS ? lds + LL.seg + (int64_t)pid * LL.segs : (aux != nullptr ? aux + pid : nullptr);
    w.bounds = WPLDS ? lds + LL.bnd + (int64_t)pid * LL.brs : (aux != nullptr ? aux + (int64_t)Wmax * Pn + pid : nullptr);
    w.sub = WPLDS ? lds + LL.sub + (int64_t)pid * LL.srs : nullptr;
    w.bo = Pn * LL.brs;
    w.so = Pn * LL.srs;
    w.nbu = LL.nbu;
    w.nsbu = LL.nsbu;
    w.ss = w.bs = WPLDS ? 1 : Pn;
    w.W = min(max(wcount[pid], 1), Wmax);
    if (WPLDS && sizeof(T) == 4) w.per = (T)(w.W - 1) / w.seg_at(w.W - 1);
    StepEngine<T> eng;
    eng.template init<true, CS>(P);

    int until_update = (ctrl_every - phase % ctrl_every) % ctrl_every;   // steps until (phase + t) % ctrl_every == 0
    int t = 0;
    RowWriter<T, 45, 3> dw(datalog, r, n);      // DATALOG / LOG instances only (non-null: the launcher)
    RowWriter<T, 16> lw(log, r, n);
    VDYN_PHASE_LAP(pc0, 11);     // state loads, engine set-up
    if (!DATALOG && !LOG) {
        // nothing is written per sub-step: between two controller updates the commands are held (drive.py:128), so the
        // sub-steps in between run in a loop of their own, two per trip (see rollout_kernel)
        PhaseClock pc;
        VDYN_PHASE_START(pc);
        while (t < H) {
            if (until_update == 0) {
                VDYN_PHASE_LAP(pc, 7);      // held sub-steps
                T steer_raw, s[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) s[i] = X.get(i);
                controller_update<T>(G, w, s, h, c, steer_raw);
                until_update = ctrl_every;
                VDYN_PHASE_LAP(pc, 6);      // whole controller update
            }
            const int run = min(until_update, H - t);
            const T delta[4] = {c.delta, c.delta, T(0), T(0)};
            const T tq[4] = {c.tau, c.tau, c.tau, c.tau};
            // the steering angle is held until the next update: its (sin, cos) once, by the function the step itself
            // would call (the same values, PRE = 1), instead of at every sub-step
            T sd0, cd0;
            eng.steer_sincos(c.delta, sd0, cd0);
            int j = 0;
            for (; j + 1 < run; j += 2) {
                eng.template advance_state<true, CS, 1, FS>(P, X, delta, tq, P.mu, h, sd0, cd0);
                eng.template advance_state<true, CS, 1, FS>(P, X, delta, tq, P.mu, h, sd0, cd0);
            }
            if (j < run) eng.template advance_state<true, CS, 1, FS>(P, X, delta, tq, P.mu, h, sd0, cd0);
            t += run;
            until_update -= run;
        }
    }
    for (; t < H; ++t) {
        if (until_update == 0) {               // wave-uniform; a countdown instead of a modulo per step
            T steer_raw, s[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) s[i] = X.get(i);
            controller_update<T>(G, w, s, h, c, steer_raw);
            until_update = ctrl_every;
        }
        --until_update;
        const T delta[4] = {c.delta, c.delta, T(0), T(0)};
        const T tq[4] = {c.tau, c.tau, c.tau, c.tau};
        T sd[10];
        Outputs18<T> o18;
        if (DATALOG) {
            T s[10], ax = X.get(10), ay = X.get(11);
#pragma unroll
            for (int i = 0; i < 10; ++i) s[i] = X.get(i);
            eng.template advance_diag<true, CS, kFitLds ? 2 : FS>(P, s, ax, ay, delta, tq, P.mu, h, sd, o18);
#pragma unroll
            for (int i = 0; i < 10; ++i) X.set(i, s[i]);
            X.set(10, ax);
            X.set(11, ay);
        } else {
            eng.template advance_state<true, CS, 0, FS>(P, X, delta, tq, P.mu, h);
        }
        if (DATALOG) {
            // written once, never read back by this kernel: streaming stores through RowWriter
            const auto row = dw.row();
            row.template put<0>((T)(phase + t) * h);                              // drive.py:145
            put_cols<T, 1, 10>(row, [&](int i) { return X.get(i); });             // :146
            put_cols<T, 11, 10>(row, [&](int i) { return sd[i]; });               // :147
            row.template put<21>(c.delta);                                        // :148
            put_cols<T, 22, 4>(row, [&](int) { return c.tau; });                  // :149
            put_cols<T, 26, 18>(row, [&](int i) { return o18.v[i]; });            // :150
            row.template put<44>(c.cte);                                          // :151
            dw.next();
        }
        if (LOG) {
            const auto row = lw.row();
            put_cols<T, 0, 12>(row, [&](int i) { return X.get(i); });
            row.template put<12>(c.delta);
            row.template put<13>(c.tau);
            row.template put<14>((T)c.idx);
            row.template put<15>(c.cte);
            lw.next();
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 12; ++i) terminal[(int64_t)i * n + r] = X.get(i);
        cstate[r] = c.x_del;
        cstate[n + r] = c.total;
        cstate[2 * n + r] = c.prev_vel;
        cstate[3 * n + r] = c.target;
        cstate[4 * n + r] = c.delta;
        cstate[5 * n + r] = c.tau;
    }
}

// One controller update for n vehicles (no integration): the drop-in of
// StanleyController.stanley_control + LongitudinalController.long_control + the filter.
//   out [3][n]: raw (limited) Stanley steering angle, target index, crosstrack error.
template <typename T>
__global__ void __launch_bounds__(kBlock)
controller_kernel(CtrlGains<T> G, int64_t n, const T *__restrict__ state, const T *__restrict__ cstate0,
                  const T *__restrict__ wp, int Wmax, const int *__restrict__ wcount,
                  const int *__restrict__ path_id, int Pn, T h, T *__restrict__ cstate, T *__restrict__ out,
                  const T *__restrict__ aux)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    T s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = state[(int64_t)i * n + r];
    CtrlState<T> c;
    c.x_del = cstate0[r];
    c.total = cstate0[n + r];
    c.prev_vel = cstate0[2 * n + r];
    c.target = cstate0[3 * n + r];
    c.delta = cstate0[4 * n + r];
    c.tau = cstate0[5 * n + r];
    const int pid = min(max(path_id[r], 0), Pn - 1);
    Waypoints<T> w;
    w.base = wp + (int64_t)pid * Wmax * 2;
    w.seg = aux != nullptr ? aux + pid : nullptr;
    w.bounds = aux != nullptr ? aux + (int64_t)Wmax * Pn + pid : nullptr;
    w.ss = w.bs = Pn;
    w.W = min(max(wcount[pid], 1), Wmax);
    T steer;
    controller_update<T>(G, w, s, h, c, steer);
    cstate[r] = c.x_del;
    cstate[n + r] = c.total;
    cstate[2 * n + r] = c.prev_vel;
    cstate[3 * n + r] = c.target;
    cstate[4 * n + r] = c.delta;
    cstate[5 * n + r] = c.tau;
    out[r] = steer;
    out[n + r] = (T)c.idx;
    out[2 * n + r] = c.cte;
}

// Collision check + best-path selection for E egos x P candidate paths of L points
// (collision_checker.py:32-117 and :134-203).  One workgroup per ego; lanes stride over
// the (path, point) pairs; the obstacle points of the ego sit in LDS and are read as
// broadcasts; per-path collision flags, end-point scores and the argmin stay on chip.
constexpr int kMaxPaths = 64;
constexpr int kMaxCircles = 8;

template <typename T>
struct Circles {
    int n;
    T offset[kMaxCircles], radius[kMaxCircles];
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
select_best_path_kernel(int E, int P, int L, const T *__restrict__ x, const T *__restrict__ y,
                        const T *__restrict__ yaw, int64_t ego_stride, int64_t path_stride, int64_t point_stride,
                        const T *__restrict__ obst, int M, int64_t obst_ego_stride, Circles<T> circ,
                        const T *__restrict__ goal, T weight, const int *__restrict__ collision_in,
                        const int *__restrict__ validity, int *__restrict__ collision_free,
                        int *__restrict__ best_idx, T *__restrict__ best_score, int chunk)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *lds_ob = reinterpret_cast<T *>(smem_raw);
    __shared__ int s_coll[kMaxPaths], s_valid[kMaxPaths];
    __shared__ T s_ex[kMaxPaths], s_ey[kMaxPaths];
    const int e = blockIdx.x;
    const T *ob = obst + (int64_t)e * obst_ego_stride;
    // collision_in given: the caller already has the flags (select_best_path_index alone)
    if (threadIdx.x < kMaxPaths)
        s_coll[threadIdx.x] = (collision_in != nullptr && (int)threadIdx.x < P)
                                  ? (collision_in[(int64_t)e * P + threadIdx.x] ? 0 : 1) : 0;
    // validity (nullable [E][P]): spirals the planner dropped (local_planner.py:317-321) are ABSENT from the
    // reference's path list -- never selectable and no contribution to anyone's proximity penalty
    if (threadIdx.x < kMaxPaths)
        s_valid[threadIdx.x] = (int)threadIdx.x < P && (validity == nullptr || validity[(int64_t)e * P + threadIdx.x] != 0);
    if (collision_in != nullptr) M = 0;

    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int mc = min(chunk, M - m0);
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * mc; i += kBlock) lds_ob[i] = ob[2 * (int64_t)m0 + i];
        __syncthreads();
        for (int idx = threadIdx.x; idx < P * L; idx += kBlock) {
            const int p = idx / L, j = idx - p * L;
            const int64_t o = (int64_t)e * ego_stride + (int64_t)p * path_stride + (int64_t)j * point_stride;
            const T px = x[o], py = y[o];
            T sy, cy;
            Lib<T>::sincos(yaw[o], &sy, &cy);
            bool hit = false;
            for (int c = 0; c < circ.n; ++c) {
                const T cx = px + circ.offset[c] * cy;              // :88
                const T cyy = py + circ.offset[c] * sy;             // :89
                const T rad = circ.radius[c];
                for (int m = 0; m < mc; ++m) {
                    const T dx = lds_ob[2 * m] - cx, dy = lds_ob[2 * m + 1] - cyy;
                    hit = hit || (Lib<T>::sqrt(dx * dx + dy * dy) - rad < T(0));   // :104-109
                }
            }
            if (hit) atomicOr(&s_coll[p], 1);
        }
    }
    __syncthreads();
    // :134-203
    if ((int)threadIdx.x < P) {
        const int64_t o = (int64_t)e * ego_stride + (int64_t)threadIdx.x * path_stride +
                          (int64_t)(L - 1) * point_stride;
        s_ex[threadIdx.x] = x[o];
        s_ey[threadIdx.x] = y[o];
        collision_free[(int64_t)e * P + threadIdx.x] = (s_coll[threadIdx.x] || !s_valid[threadIdx.x]) ? 0 : 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const T gx = goal[e], gy = goal[(int64_t)E + e];
        T bs = T(INFINITY);
        int bi = -1;
        for (int i = 0; i < P; ++i) {
            T score = T(INFINITY);                                                    // :190-191
            if (!s_valid[i]) continue;
            if (!s_coll[i]) {
                const T ax_ = s_ex[i] - gx, ay_ = s_ey[i] - gy;
                score = Lib<T>::sqrt(ax_ * ax_ + ay_ * ay_);                          // :175
                for (int j = 0; j < P; ++j) {
                    if (j == i || !s_coll[j] || !s_valid[j]) continue;
                    const T bx_ = s_ex[i] - s_ex[j], by_ = s_ey[i] - s_ey[j];
                    score += weight * Lib<T>::sqrt(bx_ * bx_ + by_ * by_);            // :183-186
                }
            }
            if (score < bs) { bs = score; bi = i; }                                   // :194-196
        }
        best_idx[e] = bi;
        best_score[e] = bs;
    }
}

// Lattice generation, stage 1: closest and goal index on the global path (local_planner.py:25-52,
// :85-152).  One workgroup per ego.  The reference's sequential scan keeps the LAST index whose rounded
// distance equals the minimum ('<=' on the sqrt values, local_planner.py:44-50); sqrt is monotone,
// so that is: m = sqrt(min d^2), answer = max { i : sqrt(d_i^2) <= m } -- two strided passes
// (coalesced reads of the shared path) and two workgroup reductions instead of a 4000-step chain
// per lane.  Only candidates within 16 ulp of the minimum take the root in the second pass.
template <typename T>
__global__ void __launch_bounds__(kBlock)
lattice_index_kernel(int E, const T *__restrict__ px, const T *__restrict__ py, int nwp, const T *__restrict__ ego,
                     T lookahead, int *__restrict__ closest_idx, int *__restrict__ goal_idx,
                     T *__restrict__ closest_len)
{
    __shared__ T s_min[kBlock / 64];
    __shared__ int s_idx[kBlock / 64];
    const int e = blockIdx.x;
    const T ex = ego[e], ey = ego[(int64_t)E + e];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    T m2 = T(INFINITY);
    for (int i = threadIdx.x; i < nwp; i += kBlock) {
        const T dx = px[i] - ex, dy = py[i] - ey;
        const T d2 = dx * dx + dy * dy;
        m2 = d2 < m2 ? d2 : m2;                       // NaN never wins, as in the reference
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(m2, off);
        m2 = o < m2 ? o : m2;
    }
    if (lane == 0) s_min[wave] = m2;
    __syncthreads();
    m2 = s_min[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) m2 = s_min[w] < m2 ? s_min[w] : m2;
    const T m = Lib<T>::sqrt(m2);

    int bi = -1;
    const T band = m2 * (T(2) - Lib<T>::kTieBand);   // 1 + 16 ulp
    for (int i = threadIdx.x; i < nwp; i += kBlock) {
        const T dx = px[i] - ex, dy = py[i] - ey;
        const T d2 = dx * dx + dy * dy;
        if (d2 <= band && Lib<T>::sqrt(d2) <= m) bi = i;     // i grows along the loop: the last one stays
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bi = max(bi, __shfl_xor(bi, off));
    if (lane == 0) s_idx[wave] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) bi = max(bi, s_idx[w]);
        const int ci = bi < 0 ? 0 : bi;               // nothing comparable (all NaN): the reference's initial 0
        closest_idx[e] = ci;
        if (closest_len != nullptr) closest_len[e] = m;
        goal_idx[e] = goal_index<T>(px, py, nwp, lookahead, m, ci);
    }
}

// Stage 2: one lane per (ego, lateral offset) -- goal state (:154-275), spiral optimisation
// (path_optimizer.py:31-88; skipped when params_in is given), sampling, validity, transform.
//   goal_set [E][P][4], params [E][P][3], paths [E][P][3][49], validity [E][P], cost [E][P]
template <typename T>
__global__ void __launch_bounds__(kBlock)
lattice_paths_kernel(int E, int P, const T *__restrict__ px, const T *__restrict__ py, int nwp,
                     const T *__restrict__ ego, const int *__restrict__ goal_idx, T goal_v, T path_offset,
                     const T *__restrict__ params_in, T *__restrict__ goal_set, T *__restrict__ params,
                     T *__restrict__ paths, int *__restrict__ validity, T *__restrict__ cost)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gid < (int64_t)E * P;
    const int64_t id = active ? gid : (int64_t)E * P - 1;
    const int e = (int)(id / P), k = (int)(id - (int64_t)e * P);
    const T ex = ego[e], ey = ego[(int64_t)E + e], eyaw = ego[2 * (int64_t)E + e];
    T gx, gy, gt;
    goal_state<T>(px, py, nwp, goal_idx[e], ex, ey, eyaw, k, P, path_offset, gx, gy, gt);
    T p[3], J;
    int iters = 0;
    if (params_in != nullptr) {
        T r[5], Jc[5][3];
        p[0] = params_in[3 * id]; p[1] = params_in[3 * id + 1]; p[2] = params_in[3 * id + 2];
        J = spiral_residuals<T>(p[0], p[1], p[2], gx, gy, gt, r, Jc);
    } else {
        J = optimize_spiral<T>(gx, gy, gt, p, iters);     // every lane of the wave iterates together
    }
    if (!active) return;
    T *out = paths + id * 3 * kSpiralPoints;
    const bool ok = sample_and_transform<T>(p, gx, gy, gt, ex, ey, eyaw, out, out + kSpiralPoints,
                                            out + 2 * kSpiralPoints, 1);
    goal_set[4 * id] = gx; goal_set[4 * id + 1] = gy; goal_set[4 * id + 2] = gt; goal_set[4 * id + 3] = goal_v;
    params[3 * id] = p[0]; params[3 * id + 1] = p[1]; params[3 * id + 2] = p[2];
    validity[id] = ok ? 1 : 0;
    cost[id] = J;
}

// Waypoint re-interpolation (local_planner.py:395-419) of path best_idx[e] of every ego to `res`
// spacing: one workgroup per ego; each lane owns one segment of the 48.
//   wp_out [E][Wmax][2] (x, y), wcount [E] are IN / OUT: an ego with best_idx < 0 keeps what they hold
//   (its previous table); too many points for Wmax gives wcount 0.
template <typename T>
__global__ void __launch_bounds__(64)
interpolate_waypoints_kernel(int E, int P, int L, const T *__restrict__ paths, const int *__restrict__ best_idx,
                             T res, int Wmax, T *__restrict__ wp_out, int *__restrict__ wcount)
{
    __shared__ int s_off[65];
    const int e = blockIdx.x, i = threadIdx.x;
    const int b = best_idx[e];
    // No selectable path (best_index is None): the reference keeps following the previous best path
    // (local_planner.py:380-384), so this ego's table and count are left as the caller passed them in.
    if (b < 0 || b >= P) return;
    const T *x = paths + ((int64_t)e * P + b) * 3 * L, *y = x + L;
    for (int base = 0; base < L - 1; base += 64) {      // L - 1 <= 64 segments per trip
        const int sgi = base + i;
        int cnt = 0;
        T dx = 0, dy = 0, dist = 0;
        if (sgi < L - 1) {
            dx = x[sgi + 1] - x[sgi]; dy = y[sgi + 1] - y[sgi];
            dist = Lib<T>::sqrt(dx * dx + dy * dy);
            const int num = (int)floor(dist / res) - 1;                          // :411
            cnt = 1 + (num > 0 ? num : 0);
        }
        s_off[i + 1] = cnt;
        if (i == 0) s_off[0] = base == 0 ? 0 : s_off[0];
        __syncthreads();
        if (i == 0) for (int j = 0; j < 64; ++j) s_off[j + 1] += s_off[j];
        __syncthreads();
        const int total_after = s_off[64];
        if (sgi < L - 1 && total_after + 1 <= Wmax) {
            T *o = wp_out + ((int64_t)e * Wmax + s_off[i]) * 2;
            o[0] = x[sgi]; o[1] = y[sgi];                                        // :407
            const T ux = dx / dist, uy = dy / dist;                              // :413
            for (int j = 0; j < cnt - 1; ++j) {
                const T f = res * (T)(j + 1);                                    // :416
                o[2 * (j + 1)] = x[sgi] + f * ux;
                o[2 * (j + 1) + 1] = y[sgi] + f * uy;
            }
        }
        __syncthreads();
        if (i == 0) s_off[0] = total_after;
        __syncthreads();
    }
    if (i == 0) {
        const int n = s_off[0];
        if (n + 1 <= Wmax) {
            wp_out[((int64_t)e * Wmax + n) * 2] = x[L - 1];                      // :419
            wp_out[((int64_t)e * Wmax + n) * 2 + 1] = y[L - 1];
            wcount[e] = n + 1;
        } else {
            wcount[e] = 0;
        }
    }
}

// Per-lane non-finite flag of a [rows][n] array (+ a count).
template <typename T>
__global__ void __launch_bounds__(kBlock)
nonfinite_lanes_kernel(int rows, int64_t n, const T *__restrict__ x, int *__restrict__ status,
                       unsigned long long *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool bad = false;
    if (i < n) {
        for (int r = 0; r < rows; ++r) {
            const T v = x[(int64_t)r * n + i];
            bad = bad || !(abs_t(v) <= (sizeof(T) == 4 ? (T)3.4028234663852886e38 : (T)1.7976931348623157e308));
        }
        status[i] = bad ? 1 : 0;
    }
    const unsigned long long m = __ballot(bad);
    if (count != nullptr && (threadIdx.x & 63) == 0 && m != 0) atomicAdd(count, (unsigned long long)__popcll(m));
}

// Device self-test of the elementary functions (include/vdyn.h, vdyn_fastmath_eval_*).
template <typename T> struct FmSel;
template <> struct FmSel<float> {
    static __device__ __forceinline__ float rcp(float x) { return fm::rcp(x); }
    static __device__ __forceinline__ float atan_rcp(float x, float ix) { return fm::atan_rcp(x, ix); }
    static __device__ __forceinline__ float sin_0_pi(float x) { return fm::sin_0_pi(x); }
    static __device__ __forceinline__ float sin_mid(float x) { return fm::sin_mid(x); }
    static __device__ __forceinline__ void sincos_mid(float x, float *s, float *c) { fm::sincos_mid(x, s, c); }
    static __device__ __forceinline__ void sincos_kernel(float x, float *s, float *c) { fm::sincos_kernel(x, s, c); }
};
template <> struct FmSel<double> {
    static __device__ __forceinline__ double rcp(double x) { return fm64::rcp(x); }
    static __device__ __forceinline__ double atan_rcp(double x, double ix) { return fm64::atan_rcp(x, ix); }
    static __device__ __forceinline__ double sin_0_pi(double x) { return fm64::sin_0_pi(x); }
    static __device__ __forceinline__ double sin_mid(double x) { return fm64::sin_mid(x); }
    static __device__ __forceinline__ void sincos_mid(double x, double *s, double *c) { fm64::sincos_mid(x, s, c); }
    static __device__ __forceinline__ void sincos_kernel(double x, double *s, double *c) { fm64::sincos_kernel(x, s, c); }
};

template <typename T>
__global__ void __launch_bounds__(kBlock)
fastmath_eval_kernel(int fn, int64_t n, const T *__restrict__ x, T c, TireFit<T> fit, T *__restrict__ out0,
                     T *__restrict__ out1)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    PkConsts K;
    K.init();
    if (i >= n) return;
    const T v = x[i];
    T a = T(0), b = T(0);
    bool two = false;
    using F = FmSel<T>;
    switch (fn) {
    case 0: a = F::atan_rcp(v, F::rcp(v)); break;
    case 1: a = F::sin_0_pi(v); break;
    case 2: a = F::sin_mid(v); break;
    case 3: F::sincos_mid(v, &a, &b); two = true; break;
    case 4: F::sincos_kernel(v, &a, &b); two = true; break;
    default:
        if constexpr (sizeof(T) == 8) {
            if (fn == 5) {
                // the trimmed fp64 step's tire chain (tire_force in vdyn_device.hpp) with the fit of C = c
                const T cc = Math<T, false>::rsqrt(fma_t(v, v, T(1)));
                T g = fma_t(fit.W[0][0], cc, fit.W[1][0]);
#pragma unroll
                for (int j = 2; j <= kTireFitDeg64; ++j) g = fma_t(g, cc, fit.W[j][0]);
                b = g * cc;
                a = g * (cc * v);
                two = true;
            }
        } else {
            const float vf = (float)v;
            if (fn == 5) {
                // the FAST step's tire chain: x sin(C atan x) / x through pacejka_g2x2 with the fit of C = c
                f2 w[kTireFitDeg + 1];
#pragma unroll
                for (int j = 0; j <= kTireFitDeg; ++j) w[j] = f2{fit.W[j][0], fit.W[j][1]};
                const f2 *const Wq[2] = {w, w};
                const f2 q1[2] = {splat(::fmaf(vf, vf, 1.0f)), splat(::fmaf(vf, vf, 1.0f))};
                const f2 sc[2] = {f2{vf, 1.0f}, f2{vf, 1.0f}};           // (x G, G)
                f2 o[2];
                pacejka_g2x2(Wq, q1, sc, o);
                a = (T)o[0].x;
                b = (T)o[0].y; two = true;
            } else if (fn == 6) {
                bool ok = true;
                const f2 r = sincos_mid2(K, vf, ok);
                a = (T)r.x; b = (T)r.y; two = true;
            } else if (fn == 7) {
                const f2 r = stage_rot2(K, vf);
                a = (T)r.x; b = (T)r.y; two = true;
            } else if (fn == 8) {
                const f2 r = sincos_kernel2(K, vf);
                a = (T)r.x; b = (T)r.y; two = true;
            }
        }
        break;
    }
    out0[i] = a;
    if (two && out1 != nullptr) out1[i] = b;
}

// ---------------------------------------------------------------- launchers ---------

// ---- the handle's tire fit (TireFit in vdyn_device.hpp) ---------------------------------------
// W_C(c) = sin(C acos c) / sqrt(1 - c^2) on c in [0, 1], in long double (the reference the fp64 fit is held to).
static long double tire_w_exact(long double C, long double c)
{
    const long double th = acosl(std::min(1.0L, std::max(-1.0L, c)));
    if (th < 1e-5L) return C * (1.0L + (1.0L - C * C) * th * th / 6.0L);   // sin(C th) / sin(th), both expanded
    return sinl(C * th) / sinl(th);
}

// One device fma on the host: fp32 through double (the product of two floats is exact there and the sum is rounded
// once more to float -- the same value as a fused fma except in double-rounding ties far below the tolerances
// checked here; glibc's software fmaf costs microseconds per call), fp64 through std::fma.
static inline float fma_as_device(float a, float b, float c) { return (float)((double)a * (double)b + (double)c); }
static inline double fma_as_device(double a, double b, double c) { return std::fma(a, b, c); }

// Degree-DEG interpolant of W_C at the Chebyshev nodes of [0, 1] (within a small factor of the minimax
// polynomial; what limits the result is the Horner evaluation in T, not the fit), as monomial coefficients in c,
// highest degree first, rounded to T.  Then the check: the Horner value in T (one rounding per fma, as on the
// device) against long double on 1025 points -- absolute error of mu / D = sin(C atan x) within `tol` everywhere,
// relative error of G = sin(C atan x) / x (the cornering / longitudinal stiffness at small slip) within `tol`
// for x <= sqrt(3).  False if the check fails.  ~0.1 ms (fp32) / 0.4 ms (fp64) per shape factor, once per thread
// and C (tire_fit below keeps the results).
template <typename T, int DEG>
static bool fit_tire_wheel(double C, T *W, double tol)
{
    constexpr int n = DEG + 1;
    const long double pi = 3.14159265358979323846264338327950288L;
    long double f[n], a[n];
    for (int k = 0; k < n; ++k) f[k] = tire_w_exact(C, 0.5L * (cosl(pi * (2 * k + 1) / (2.0L * n)) + 1.0L));
    for (int j = 0; j < n; ++j) {
        long double acc = 0.0L;
        for (int k = 0; k < n; ++k) acc += f[k] * cosl(pi * j * (2 * k + 1) / (2.0L * n));
        a[j] = acc * (j == 0 ? 1.0L : 2.0L) / n;
    }
    // sum_j a_j T_j(2 c - 1) as a polynomial in c: T_0 = 1, T_1 = z, T_{j+1} = 2 z T_j - T_{j-1}, z = 2 c - 1
    long double t0[n] = {1.0L}, t1[n] = {-1.0L, 2.0L}, m[n];
    for (int i = 0; i < n; ++i) m[i] = a[0] * t0[i] + a[1] * t1[i];
    for (int j = 2; j < n; ++j) {
        long double t2[n];
        for (int i = 0; i < n; ++i) t2[i] = 2.0L * ((i > 0 ? 2.0L * t1[i - 1] : 0.0L) - t1[i]) - t0[i];
        for (int i = 0; i < n; ++i) { m[i] += a[j] * t2[i]; t0[i] = t1[i]; t1[i] = t2[i]; }
    }
    for (int i = 0; i < n; ++i) W[i] = (T)m[n - 1 - i];
    if (!std::isfinite(C)) return false;
    constexpr int kGrid = 1024;
    for (int i = 0; i <= kGrid; ++i) {
        const T c = (T)i / (T)kGrid;
        T g = fma_as_device(W[0], c, W[1]);
        for (int k = 2; k < n; ++k) g = fma_as_device(g, c, W[k]);
        const long double want = tire_w_exact(C, (long double)c), err = fabsl((long double)g - want);
        if (!(err * sqrtl(std::max(0.0L, 1.0L - (long double)c * c)) <= tol)) return false;
        if (c >= (T)0.5 && !(err <= tol * std::max(fabsl(want), 1e-300L))) return false;
    }
    return true;
}
constexpr double kTireFitTol32 = 5e-7, kTireFitTol64 = kTireFitDeg64 >= 18 ? 4e-15 : 5e-14;   // degree 8 / 16 (18: round 3's gate)

// Both fits of one shape factor.
struct TireFitC {
    double C;
    float W[kTireFitDeg + 1];
    double W64[kTireFitDeg64 + 1];
    bool ok, ok64;
};
struct TireFitHost {
    double C[4];
    float W[kTireFitDeg + 1][4];
    double W64[kTireFitDeg64 + 1][4];
    bool ok, ok64, filled;
};

// The fits of a handle's four shape factors.  Kept per thread: the handle's assembled set (a launch per 0.15 ms
// must not refit -- four compares when C has not changed) and the last sixteen shape factors seen (a fleet table
// of 256 classes is rebuilt per launch).  The key is C alone: B does not enter W_C.
static const TireFitC &tire_fit_c(double C)
{
    constexpr int kSlots = 16;
    static thread_local TireFitC slots[kSlots];
    static thread_local int used = 0, next = 0;
    for (int i = 0; i < used; ++i)
        if (std::memcmp(&slots[i].C, &C, sizeof(C)) == 0) return slots[i];     // bitwise: NaN finds itself
    TireFitC &e = slots[next];
    next = (next + 1) % kSlots;
    used = std::min(used + 1, kSlots);
    e.C = C;
    e.ok = fit_tire_wheel<float, kTireFitDeg>(C, e.W, kTireFitTol32);
    e.ok64 = fit_tire_wheel<double, kTireFitDeg64>(C, e.W64, kTireFitTol64);
    return e;
}

static const TireFitHost &tire_fit(const VdynParams &p)
{
    static thread_local TireFitHost cache = {{0, 0, 0, 0}, {{0}}, {{0}}, false, false, false};
    if (cache.filled && std::memcmp(cache.C, p.C, sizeof(cache.C)) == 0) return cache;
    cache.ok = cache.ok64 = true;
    for (int w = 0; w < 4; ++w) {
        const TireFitC e = tire_fit_c(p.C[w]);        // by value: a later lookup may recycle the slot
        cache.ok = cache.ok && e.ok;
        cache.ok64 = cache.ok64 && e.ok64;
        for (int i = 0; i <= kTireFitDeg; ++i) cache.W[i][w] = e.W[i];
        for (int i = 0; i <= kTireFitDeg64; ++i) cache.W64[i][w] = e.W64[i];
    }
    std::memcpy(cache.C, p.C, sizeof(cache.C));
    cache.filled = true;
    return cache;
}

template <typename T>
DevParams<T> make_dev_params(const VdynParams &p, const double *mu4)
{
    DevParams<T> d;
    {
        const TireFitHost &f = tire_fit(p);
        if constexpr (std::is_same<T, float>::value) std::memcpy(d.W, f.W, sizeof(d.W));
        else std::memcpy(d.W, f.W64, sizeof(d.W));
    }
    d.inv_m = (T)(1.0 / p.m);
    d.inv_Izz = (T)(1.0 / p.Izz);
    d.inv_Jw = (T)(1.0 / p.Jw);
    d.a = (T)p.a;
    d.b = (T)p.b;
    d.half_T = (T)(p.T / 2);
    d.rw = (T)p.rw;
    const double L = p.a + p.b, W = p.wL + p.wR;
    d.Fz0F = (T)(p.b / L * p.m * p.g / 2);   // vehicle_model.py:245-246
    d.Fz0R = (T)(p.a / L * p.m * p.g / 2);   // :247-248
    d.DfzxL = (T)(p.m * p.hg * p.wR / (L * W));  // :250
    d.DfzxR = (T)(p.m * p.hg * p.wL / (L * W));  // :251
    d.DfzyF = (T)(p.m * p.hg * p.b / (L * W));   // :252
    d.DfzyR = (T)(p.m * p.hg * p.a / (L * W));   // :253
    for (int i = 0; i < 4; ++i) {
        d.B[i] = (T)p.B[i];
        d.C[i] = (T)p.C[i];
        d.invB[i] = (T)(1.0 / p.B[i]);
        d.mu[i] = mu4 ? (T)mu4[i] : (T)1;
    }
    return d;
}

static bool stiffness_nonnegative(const VdynParams &p)
{
    for (int i = 0; i < 4; ++i)
        if (!(p.B[i] >= 0.0)) return false;      // the fitted chain carries B s_y as -vy |B / vx| (quirk Q4)
    return true;
}

// The CS flag of the kernels: the FAST step runs the handle's fitted tire chain.  Needs B >= 0 on every wheel and
// fits that passed their check (any C for which they do: 0 up to about 2.9), per wheel in both precisions.
static bool same_shape_factor(const VdynParams &p) { return p.C[0] == p.C[1] && p.C[0] == p.C[2] && p.C[0] == p.C[3]; }

// per_wheel: the kernel takes a fit per wheel (fp32: all of them; fp64: the rollout kernel through LDS, the
// wheel-parallel one in registers); the other fp64 kernels carry ONE set and need the four wheels to share C.
template <typename T>
static bool lane_cs(const VdynParams &p, bool per_wheel = false)
{
    if (!stiffness_nonnegative(p)) return false;
    if (std::is_same<T, float>::value) return tire_fit(p).ok;
    return tire_fit(p).ok64 && (per_wheel || same_shape_factor(p));
}

// How an fp64 lane kernel's FAST step gets its fit (fp32 kernels carry the four fits as packed pairs: always 1):
//   0 the general chain; 1 one set pinned (the wheels share C); 3 one set per axle, both pinned (C_FL = C_FR,
//   C_RL = C_RR: vehicle_model.py:237-242 sketches exactly that handle); 2 four sets, read from LDS -- the rollout
//   kernel and the DataLog closed loop only: the kernels without that path send such a handle to the general chain.
template <typename T>
static int fit_mode(const VdynParams &p)
{
    if (!lane_cs<T>(p, true)) return 0;
    if (sizeof(T) == 4 || same_shape_factor(p)) return 1;
    return p.C[0] == p.C[1] && p.C[2] == p.C[3] ? 3 : 2;
}

template <typename T, int K, int LAYOUT, bool DIAG, bool CS, int PW = 0, bool COMP = false>
static hipError_t launch_rollout_impl(const VdynParams &p, const RolloutArgs<T> &a, hipStream_t st)
{
    const DevParams<T> P = make_dev_params<T>(p, a.mu4);
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    int chunk = a.H > 0 ? a.H : 1;
    size_t lds = 0;
    if (LAYOUT == 1) {
        const size_t per_step = rollout_lds_per_step<T, K, LAYOUT, DIAG>(a.P);
        chunk = (int)std::min<size_t>((size_t)chunk, kLdsBudget / per_step);
        lds = (size_t)chunk * per_step;
    }
    // (`if constexpr`: a diagnostic launch exists in the TRAJ form only -- the other form would be an instance nobody can
    // launch, and the fp64 k = 12 ones are the heaviest kernels of the library to compile)
    if constexpr (DIAG) {
        hipLaunchKernelGGL((rollout_kernel<T, K, LAYOUT, DIAG, CS, true, PW, COMP>), dim3(grid), dim3(kBlock), lds, st, P, a.n,
                           a.H, a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                           a.traj_stride > 0 ? a.traj_stride : 1, a.state_dot, a.outputs);
    } else {
        if (a.traj != nullptr)
            hipLaunchKernelGGL((rollout_kernel<T, K, LAYOUT, DIAG, CS, true, PW, COMP>), dim3(grid), dim3(kBlock), lds, st, P,
                               a.n, a.H, a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                               a.traj_stride > 0 ? a.traj_stride : 1, a.state_dot, a.outputs);
        else
            hipLaunchKernelGGL((rollout_kernel<T, K, LAYOUT, DIAG, CS, false, PW, COMP>), dim3(grid), dim3(kBlock), lds, st, P,
                               a.n, a.H, a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                               a.traj_stride > 0 ? a.traj_stride : 1, a.state_dot, a.outputs);
    }
    return hipGetLastError();
}

template <typename T, int K, int LAYOUT, bool CS>
static hipError_t launch_rollout_fleet_impl(const RolloutArgs<T> &a, hipStream_t st)
{
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    const size_t fleet_bytes = (size_t)a.V * dev_params_len<T>() * sizeof(T);
    int chunk = a.H > 0 ? a.H : 1;
    size_t lds = fleet_bytes;
    if (LAYOUT == 1) {
        const size_t per_step = (size_t)a.P * K * sizeof(T);
        chunk = (int)std::min<size_t>((size_t)chunk, std::max<size_t>(1, (kLdsBudget - 16 * 1024) / per_step));
        lds += (size_t)chunk * per_step;
    }
    if (lds > 64 * 1024) {      // 256 fp64 classes alone are 58 KiB: beyond the default limit a workgroup opts in (160 KiB per CU)
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_fleet_kernel<T, K, LAYOUT, CS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return e_;
    }
    hipLaunchKernelGGL((rollout_fleet_kernel<T, K, LAYOUT, CS>), dim3(grid), dim3(kBlock), lds, st, a.fleet_tab, a.V,
                       a.vehicle_id, a.n, a.H, a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                       a.traj_stride > 0 ? a.traj_stride : 1);
    return hipGetLastError();
}

// Host side of the fleet table: one DevParams<T> per class, laid out as plain T words.
template <typename T>
void build_fleet_table(const VdynParams *classes, int V, const double *mu4, T *out, bool *all_small)
{
    *all_small = true;
    for (int v = 0; v < V; ++v) {
        const DevParams<T> d = make_dev_params<T>(classes[v], mu4);
        FleetRow<T> row;
        row.pack(d);
        std::memcpy(out + (size_t)v * dev_params_len<T>(), &row, sizeof(row));
        const double *C = classes[v].C;
        const bool one_set = sizeof(T) == 4 || (C[0] == C[1] && C[0] == C[2] && C[0] == C[3]);   // FleetRow<double>
        *all_small = *all_small && lane_cs<T>(classes[v]) && one_set;
    }
}
template <typename T> int fleet_table_len(int V) { return V * dev_params_len<T>(); }

template <typename T>
hipError_t launch_rollout_fleet(const RolloutArgs<T> &a, bool all_small, hipStream_t st)
{
    if (a.n <= 0) return hipSuccess;
    int layout = a.layout;
    if (layout == VDYN_CTRL_SHARED && (size_t)a.P * a.k * sizeof(T) > (size_t)(kLdsBudget - 16 * 1024)) layout = 2;
#define VDYN_DISPATCH_F(KK, LL)                                                        \
    if (a.k == KK && layout == LL)                                                     \
        return all_small ? launch_rollout_fleet_impl<T, KK, LL, true>(a, st)           \
                         : launch_rollout_fleet_impl<T, KK, LL, false>(a, st);
    VDYN_DISPATCH_F(2, 0)
    VDYN_DISPATCH_F(2, 1)
    VDYN_DISPATCH_F(2, 2)
    VDYN_DISPATCH_F(12, 0)
    VDYN_DISPATCH_F(12, 1)
    VDYN_DISPATCH_F(12, 2)
#undef VDYN_DISPATCH_F
    return hipErrorInvalidValue;
}

template <typename T, int K, int LAYOUT, bool CS>
static hipError_t launch_rollout_quad_impl(const VdynParams &p, const RolloutArgs<T> &a, hipStream_t st)
{
    const DevParams<T> P = make_dev_params<T>(p, a.mu4);
    const unsigned grid = (unsigned)((4 * a.n + kBlock - 1) / kBlock);
    int chunk = a.H > 0 ? a.H : 1;
    size_t lds = 0;
    if (LAYOUT == 1) {
        const size_t per_step = (size_t)a.P * K * sizeof(T);
        chunk = (int)std::min<size_t>((size_t)chunk, kLdsBudget / per_step);
        lds = (size_t)chunk * per_step;
    }
    if (a.traj != nullptr)
        hipLaunchKernelGGL((rollout_quad_kernel<T, K, LAYOUT, CS, true>), dim3(grid), dim3(kBlock), lds, st, P, a.n, a.H,
                           a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                           a.traj_stride > 0 ? a.traj_stride : 1);
    else
        hipLaunchKernelGGL((rollout_quad_kernel<T, K, LAYOUT, CS, false>), dim3(grid), dim3(kBlock), lds, st, P, a.n, a.H,
                           a.state0, a.ctrl, a.path_id, a.P, chunk, (T)a.dt, a.terminal, a.traj,
                           a.traj_stride > 0 ? a.traj_stride : 1);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_rollout(const VdynParams &p, const RolloutArgs<T> &a, hipStream_t st)
{
    if (a.n <= 0) return hipSuccess;
    int layout = a.layout;
    // one step of the table must fit the LDS budget (k = 2 tables are staged 4 wide, see rollout_table_pre)
    if (layout == VDYN_CTRL_SHARED && (size_t)a.P * (a.k == 2 ? 4 : a.k) * sizeof(T) > (size_t)kLdsBudget) layout = 2;
    const bool diag = a.state_dot != nullptr || a.outputs != nullptr;
    const bool cs_quad = lane_cs<T>(p, true);                   // a lane of the wheel-parallel kernel holds its own wheel's fit
    const bool cs = cs_quad, pw = sizeof(T) == 8 && cs && !same_shape_factor(p);
    const bool axles = pw && p.C[0] == p.C[1] && p.C[2] == p.C[3];     // one C per axle: two sets pinned, no LDS
    if (a.state_rows == 22) {
        // compensated state sum: fp32, lane per rollout, fitted chain (anything else: invalid value -> VDYN_ERR_ARG)
        if constexpr (sizeof(T) == 4) {
            if (cs && !diag) {
#define VDYN_DISPATCH_C(KK, LL) \
    if (a.k == KK && layout == LL) return launch_rollout_impl<T, KK, LL, false, true, 0, true>(p, a, st);
                VDYN_DISPATCH_C(2, 0)
                VDYN_DISPATCH_C(2, 1)
                VDYN_DISPATCH_C(2, 2)
                VDYN_DISPATCH_C(12, 0)
                VDYN_DISPATCH_C(12, 1)
                VDYN_DISPATCH_C(12, 2)
#undef VDYN_DISPATCH_C
            }
        }
        return hipErrorInvalidValue;
    }
    if (a.lanes_per_rollout == 4 && !diag) {
#define VDYN_DISPATCH_Q(KK, LL)                                                        \
    if (a.k == KK && layout == LL)                                                     \
        return cs_quad ? launch_rollout_quad_impl<T, KK, LL, true>(p, a, st)                \
                  : launch_rollout_quad_impl<T, KK, LL, false>(p, a, st);
        VDYN_DISPATCH_Q(2, 0)
        VDYN_DISPATCH_Q(2, 1)
        VDYN_DISPATCH_Q(2, 2)
        VDYN_DISPATCH_Q(12, 0)
        VDYN_DISPATCH_Q(12, 1)
        VDYN_DISPATCH_Q(12, 2)
#undef VDYN_DISPATCH_Q
    }
    // the diagnostic (single-step drop-in) variants are launch-latency bound: general form only, and per-rollout controls
    // only -- diagnostics come in through vdyn_step_* (one step, controls [k][n]); a shared table with diagnostics is
    // refused by the C ABI (rollout_dev), so those instances are not built
#define VDYN_DISPATCH(KK, LL)                                                          \
    if (a.k == KK && layout == LL) {                                                   \
        if (diag) {                                                                    \
            if constexpr (LL == 0) return launch_rollout_impl<T, KK, LL, true, false>(p, a, st); \
            else return hipErrorInvalidValue;                                          \
        }                                                                              \
        if constexpr (sizeof(T) == 8)                                                  \
            if (pw) return axles ? launch_rollout_impl<T, KK, LL, false, true, 2>(p, a, st) \
                                 : launch_rollout_impl<T, KK, LL, false, true, 1>(p, a, st); \
        return cs ? launch_rollout_impl<T, KK, LL, false, true>(p, a, st)              \
                  : launch_rollout_impl<T, KK, LL, false, false>(p, a, st);            \
    }
    VDYN_DISPATCH(2, 0)
    VDYN_DISPATCH(2, 1)
    VDYN_DISPATCH(2, 2)
    VDYN_DISPATCH(12, 0)
    VDYN_DISPATCH(12, 1)
    VDYN_DISPATCH(12, 2)
#undef VDYN_DISPATCH
    return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_rollout_spiral(const VdynParams &p, int64_t n, int H, const T *state0, const T *spiral,
                                 double wheelbase, double tan_max, double torque, double dt, const double *mu4,
                                 T *terminal, T *traj, int traj_stride, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const DevParams<T> P = make_dev_params<T>(p, mu4);
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
#define VDYN_SPIRAL(CSV, TRV, FSV)                                                                         \
    hipLaunchKernelGGL((rollout_spiral_kernel<T, CSV, TRV, FSV>), dim3(grid), dim3(kBlock), 0, st, P, n, H, state0, spiral, \
                       (T)wheelbase, (T)tan_max, (T)torque, (T)dt, terminal, traj, traj_stride > 0 ? traj_stride : 1)
    const int fm = fit_mode<T>(p);
    if (fm == 3) {
        if constexpr (sizeof(T) == 8) { if (traj != nullptr) VDYN_SPIRAL(true, true, 3); else VDYN_SPIRAL(true, false, 3); }
    } else if (fm == 1) {
        if (traj != nullptr) VDYN_SPIRAL(true, true, 1); else VDYN_SPIRAL(true, false, 1);
    } else {
        if (traj != nullptr) VDYN_SPIRAL(false, true, 1); else VDYN_SPIRAL(false, false, 1);
    }
#undef VDYN_SPIRAL
    return hipGetLastError();
}

template <typename T>
hipError_t launch_nonfinite_lanes(int rows, int64_t n, const T *x, int *status, unsigned long long *count, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((nonfinite_lanes_kernel<T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, n,
                       x, status, count);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fastmath_eval(int fn, int64_t n, const T *x, double c, T *out0, T *out1, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    TireFit<T> fit;
    std::memset(&fit, 0, sizeof(fit));
    if (fn == 5) {
        VdynParams p;
        std::memset(&p, 0, sizeof(p));
        for (int w = 0; w < 4; ++w) p.C[w] = c;
        const TireFitHost &f = tire_fit(p);
        // no validated fit for this C: the step would not use it either
        if constexpr (std::is_same<T, float>::value) {
            if (!f.ok) return hipErrorInvalidValue;
            std::memcpy(fit.W, f.W, sizeof(fit.W));
        } else {
            if (!f.ok64) return hipErrorInvalidValue;
            std::memcpy(fit.W, f.W64, sizeof(fit.W));
        }
    }
    hipLaunchKernelGGL((fastmath_eval_kernel<T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, fn, n, x,
                       (T)c, fit, out0, out1);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_planar_model(const VdynParams &p, int64_t n, const T *state, const T *ctrl12,
                               const T *acc_prev, T *state_dot, T *aux, T *outputs, T *acc,
                               hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const DevParams<T> P = make_dev_params<T>(p, nullptr);
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((planar_model_kernel<T>), dim3(grid), dim3(kBlock), 0, st, P, n, state, ctrl12,
                       acc_prev, state_dot, aux, outputs, acc);
    return hipGetLastError();
}

// Candidates per wave of mpc_argmin_lanes_kernel: the grid should be about `target` waves (two per SIMD in fp32, one
// in fp64: the register budget of the step), but a wave keeps at least 2 candidates when there are that many.
template <typename T>
static int mpc_chunk(int E, int C)
{
    const int64_t target = sizeof(T) == 4 ? 2048 : 1024, groups = (E + 63) / 64;
    const int64_t kc = (groups * C + target - 1) / target;
    return (int)std::max<int64_t>(1, std::min<int64_t>(kc, C));
}

// Device scratch of launch_mpc_argmin: the candidate table with (sin, cos) per entry, then the per-chunk partial
// minima (cost, index) [nchunks][E].
template <typename T>
size_t mpc_scratch_bytes(int E, int C, int H)
{
    const size_t nchunks = ((size_t)C + mpc_chunk<T>(E, C) - 1) / mpc_chunk<T>(E, C);
    const size_t tab = ((sizeof(T) * 4 * (size_t)H * C + 255) / 256) * 256;
    return tab + nchunks * (size_t)E * (sizeof(T) + sizeof(int)) + 256;
}

template <typename T>
hipError_t launch_mpc_argmin(const VdynParams &p, int E, int C, int H, const T *ego, const T *cand,
                             const T *goal, double dt, double w_delta, T *best_cost, int *best_idx,
                             T *cost_all, void *scratch, hipStream_t st)
{
    if (E <= 0) return hipSuccess;
    const DevParams<T> P = make_dev_params<T>(p, nullptr);
    T *cand4 = static_cast<T *>(scratch);
    if (H > 0) {
        const int64_t n = (int64_t)H * C;
        hipLaunchKernelGGL((mpc_prepare_kernel<T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, P,
                           C, H, cand, cand4);
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return e_;
    }
    const int KC = mpc_chunk<T>(E, C), nchunks = (C + KC - 1) / KC, groups = (E + 63) / 64;
    // Which mapping: egos on the lanes needs whole waves of egos and enough (ego group, chunk) pairs to fill the chip;
    // below that a workgroup per ego wastes no lanes and needs no second kernel.  Both give the same bits
    // (tools/ubench/mpc_harness.hip); configs[4]: 0.306 against 0.311 ms per call.
    const bool lanes = E >= 512 && (int64_t)groups * C >= 1024;
    const int fm = fit_mode<T>(p);
    if (!lanes) {
        int block = ((C + 63) / 64) * 64;
        block = std::max(64, std::min(block, mpc_block_max<T>()));
        if (fm == 3) {
            if constexpr (sizeof(T) == 8)
                hipLaunchKernelGGL((mpc_argmin_kernel<T, true, 3>), dim3((unsigned)E), dim3((unsigned)block), 0, st, P, E,
                                   C, H, ego, cand4, goal, (T)dt, (T)w_delta, best_cost, best_idx, cost_all);
        } else if (fm == 1)
            hipLaunchKernelGGL((mpc_argmin_kernel<T, true>), dim3((unsigned)E), dim3((unsigned)block), 0, st, P, E,
                               C, H, ego, cand4, goal, (T)dt, (T)w_delta, best_cost, best_idx, cost_all);
        else
            hipLaunchKernelGGL((mpc_argmin_kernel<T, false>), dim3((unsigned)E), dim3((unsigned)block), 0, st, P, E,
                               C, H, ego, cand4, goal, (T)dt, (T)w_delta, best_cost, best_idx, cost_all);
        return hipGetLastError();
    }
    const size_t tab = ((sizeof(T) * 4 * (size_t)H * C + 255) / 256) * 256;
    T *part_cost = reinterpret_cast<T *>(static_cast<char *>(scratch) + tab);
    int *part_idx = reinterpret_cast<int *>(part_cost + (size_t)nchunks * E);
    const unsigned grid = (unsigned)((int64_t)groups * nchunks);
    if (fm == 3) {
        if constexpr (sizeof(T) == 8)
            hipLaunchKernelGGL((mpc_argmin_lanes_kernel<T, true, 3>), dim3(grid), dim3(64), 0, st, P, E, C, H, KC, ego, cand4,
                               goal, (T)dt, (T)w_delta, part_cost, part_idx, cost_all);
    } else if (fm == 1)
        hipLaunchKernelGGL((mpc_argmin_lanes_kernel<T, true>), dim3(grid), dim3(64), 0, st, P, E, C, H, KC, ego, cand4,
                           goal, (T)dt, (T)w_delta, part_cost, part_idx, cost_all);
    else
        hipLaunchKernelGGL((mpc_argmin_lanes_kernel<T, false>), dim3(grid), dim3(64), 0, st, P, E, C, H, KC, ego, cand4,
                           goal, (T)dt, (T)w_delta, part_cost, part_idx, cost_all);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return e_;
    hipLaunchKernelGGL((mpc_reduce_kernel<T>), dim3((unsigned)groups), dim3(kBlock), 0, st, E, nchunks,
                       part_cost, part_idx, best_cost, best_idx);
    return hipGetLastError();
}

template <typename T>
static CtrlGains<T> make_gains(const VdynCtrlGains &g)
{
    CtrlGains<T> c;
    c.k = (T)g.k; c.k_soft = (T)g.k_soft; c.max_steer = (T)g.max_steer;
    c.lookahead = (T)g.lookahead; c.deadband = (T)g.deadband;
    c.kp = (T)g.kp; c.ki = (T)g.ki; c.kd = (T)g.kd;
    c.filt_keep = (T)(1 - g.filter_gain);   // drive.py:137
    c.filt_gain = (T)g.filter_gain;
    return c;
}

// gfx950 has 160 KiB of LDS per CU; the closed loop may take 152 KiB of it for waypoint tables
constexpr size_t kClosedLoopLdsBudget = 152 * 1024;

template <typename T>
hipError_t launch_closed_loop(const VdynParams &p, const VdynCtrlGains &g, const ClosedLoopArgs<T> &a,
                              hipStream_t st)
{
    if (a.n <= 0) return hipSuccess;
    const DevParams<T> P = make_dev_params<T>(p, nullptr);
    const CtrlGains<T> G = make_gains<T>(g);
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    // x / y rows + segment lengths + one bounding circle per 32 and per 8 waypoints (ClosedLoopLds)
    const size_t wp_bytes = ClosedLoopLds<T>(a.P, a.Wmax).bytes();
    // gfx950 has 160 KiB of LDS per CU; a workgroup may take all of it (one workgroup per
    // CU is also what 65536 vehicles give), beyond the 64 KiB default only after opting in
    const bool lds = wp_bytes <= kClosedLoopLdsBudget;
    // the fitted chain: with the DataLog the fp64 kernel reads the per-wheel table from LDS (any handle whose fits
    // passed their check); without it the fits are pinned -- one set, or one per axle (fit_mode)
    const int fm = fit_mode<T>(p);
    const bool dl = a.datalog != nullptr;
    const bool ax = sizeof(T) == 8 && !dl && fm == 3;
    const bool cs = fm == 1 || (sizeof(T) == 8 && dl && fm != 0);
    // tables in LDS: the kernel builds its whole image itself; otherwise the transposed tables in `aux`
    if (a.aux != nullptr && !lds) {
        const int64_t tiles = (int64_t)((a.P + kAuxTP - 1) / kAuxTP) * ((a.Wmax + kAuxTJ - 1) / kAuxTJ);
        hipLaunchKernelGGL((waypoint_aux_kernel<T>), dim3((unsigned)tiles), dim3(kBlock), 0, st, a.wp,
                           a.Wmax, a.wcount, a.P, a.aux);
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return e_;
        if (sizeof(T) == 4) {                                   // fp32: cumulative arc length instead of segment lengths
            hipLaunchKernelGGL((waypoint_cumsum_kernel<T>), dim3((unsigned)a.P), dim3(64), 0, st, a.Wmax, a.P, a.aux);
            e_ = hipGetLastError();
            if (e_ != hipSuccess) return e_;
        }
    }
#define VDYN_CL3(CSV, LDSV, DLV, LGV, FSV)                                                            \
    {                                                                                                 \
        if (LDSV && wp_bytes > 64 * 1024) {                                                           \
            hipError_t e_ = hipFuncSetAttribute(                                                      \
                reinterpret_cast<const void *>(&closed_loop_kernel<T, CSV, LDSV, DLV, LGV, FSV>),     \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)wp_bytes);                           \
            if (e_ != hipSuccess) return e_;                                                          \
        }                                                                                             \
        hipLaunchKernelGGL((closed_loop_kernel<T, CSV, LDSV, DLV, LGV, FSV>), dim3(grid), dim3(kBlock), \
                           LDSV ? wp_bytes : 0, st, P, G, a.n, a.H, a.ctrl_every, a.phase, a.state0,  \
                           a.cstate0, a.wp, a.Wmax, a.wcount, a.path_id, a.P, (T)a.dt, a.terminal,    \
                           a.cstate, a.log, a.datalog, (const T *)a.aux);                             \
    }
#define VDYN_CL2(CSV, LDSV, DLV, FSV)                                                                 \
    if (a.log != nullptr) VDYN_CL3(CSV, LDSV, DLV, true, FSV) else VDYN_CL3(CSV, LDSV, DLV, false, FSV)
#define VDYN_CL(CSV, LDSV)                                                                            \
    if (dl) { VDYN_CL2(CSV, LDSV, true, 1) } else { VDYN_CL2(CSV, LDSV, false, 1) }
    if (ax) {
        if constexpr (sizeof(T) == 8) {
            if (lds) { VDYN_CL2(true, true, false, 3) } else { VDYN_CL2(true, false, false, 3) }
        }
    }
    else if (cs && lds) { VDYN_CL(true, true) }
    else if (cs) { VDYN_CL(true, false) }
    else if (lds) { VDYN_CL(false, true) }
    else { VDYN_CL(false, false) }
#undef VDYN_CL3
#undef VDYN_CL2
#undef VDYN_CL
    return hipGetLastError();
}

// Bytes of device scratch the controllers need for (P tables x Wmax waypoints): the closed loop's
// LDS image or, for tables too large for LDS, the transposed global tables; the single controller
// update has no LDS stage and asks for the latter once the plain full scan would be slower.
template <typename T>
size_t closed_loop_aux_bytes(int P, int Wmax, bool update_only)
{
    const size_t wp_bytes = ClosedLoopLds<T>(P, Wmax).bytes();
    if (update_only) return Wmax < 4 * kWpBlock ? 0 : waypoint_aux_len<T>(P, Wmax) * sizeof(T);
    if (wp_bytes <= kClosedLoopLdsBudget) return 0;            // the kernel builds its LDS image from the tables alone
    return waypoint_aux_len<T>(P, Wmax) * sizeof(T);
}

template <typename T>
hipError_t launch_controller_update(const VdynCtrlGains &g, const ClosedLoopArgs<T> &a, hipStream_t st)
{
    if (a.n <= 0) return hipSuccess;
    const CtrlGains<T> G = make_gains<T>(g);
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    if (a.aux != nullptr) {
        const int64_t tiles = (int64_t)((a.P + kAuxTP - 1) / kAuxTP) * ((a.Wmax + kAuxTJ - 1) / kAuxTJ);
        hipLaunchKernelGGL((waypoint_aux_kernel<T>), dim3((unsigned)tiles), dim3(kBlock), 0, st, a.wp, a.Wmax,
                           a.wcount, a.P, a.aux);
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return e_;
        if (sizeof(T) == 4) {
            hipLaunchKernelGGL((waypoint_cumsum_kernel<T>), dim3((unsigned)a.P), dim3(64), 0, st, a.Wmax, a.P, a.aux);
            e_ = hipGetLastError();
            if (e_ != hipSuccess) return e_;
        }
    }
    hipLaunchKernelGGL((controller_kernel<T>), dim3(grid), dim3(kBlock), 0, st, G, a.n, a.state0, a.cstate0,
                       a.wp, a.Wmax, a.wcount, a.path_id, a.P, (T)a.dt, a.cstate, a.ctrl_out, (const T *)a.aux);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_select_best_path(const SelectArgs<T> &a, hipStream_t st)
{
    if (a.E <= 0) return hipSuccess;
    Circles<T> c;
    c.n = a.nc;
    for (int i = 0; i < kMaxCircles; ++i) {
        c.offset[i] = i < a.nc ? (T)a.offsets[i] : (T)0;
        c.radius[i] = i < a.nc ? (T)a.radii[i] : (T)0;
    }
    const int chunk = std::max(1, std::min(a.M, (int)(kLdsBudget / (2 * sizeof(T)))));
    hipLaunchKernelGGL((select_best_path_kernel<T>), dim3((unsigned)a.E), dim3(kBlock),
                       (size_t)chunk * 2 * sizeof(T), st, a.E, a.P, a.L, a.x, a.y, a.yaw, a.ego_stride,
                       a.path_stride, a.point_stride, a.obst, a.M, a.obst_ego_stride, c, a.goal, (T)a.weight,
                       a.collision_in, a.validity, a.collision_free, a.best_idx, a.best_score, chunk);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_plan_lattice(const LatticeArgs<T> &a, hipStream_t st)
{
    if (a.E <= 0) return hipSuccess;
    hipLaunchKernelGGL((lattice_index_kernel<T>), dim3((unsigned)a.E), dim3(kBlock), 0, st,
                       a.E, a.px, a.py, a.nwp, a.ego, (T)a.lookahead, a.closest_idx, a.goal_idx, a.closest_len);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t lanes = (int64_t)a.E * a.P;
    hipLaunchKernelGGL((lattice_paths_kernel<T>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       st, a.E, a.P, a.px, a.py, a.nwp, a.ego, a.goal_idx, (T)a.goal_v, (T)a.path_offset,
                       a.params_in, a.goal_set, a.params, a.paths, a.validity, a.cost);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_interpolate_waypoints(int E, int P, int L, const T *paths, const int *best_idx, double res,
                                        int Wmax, T *wp_out, int *wcount, hipStream_t st)
{
    if (E <= 0) return hipSuccess;
    hipLaunchKernelGGL((interpolate_waypoints_kernel<T>), dim3((unsigned)E), dim3(64), 0, st, E, P, L, paths,
                       best_idx, (T)res, Wmax, wp_out, wcount);
    return hipGetLastError();
}

// The explicit instantiations, in two parts per precision: VDYN_PART 1 = the rollout launcher (lane and wheel-parallel
// kernels: more than half of the library's kernels and the heaviest to compile), VDYN_PART 2 = everything else; no
// VDYN_PART = both.  Four translation units (vdyn_kernels_f32_rollout.hip, ..._f32_rest.hip, ..._f64_rollout.hip,
// ..._f64_rest.hip) compile in parallel, each precision with the instruction-scheduling strategy that suits it
// (_build.py).  Everything a part does not instantiate it only declares; file-local state (the tire-fit cache) simply
// exists once per part.
#define VDYN_INSTANTIATE_ROLLOUT(T)                                                                  \
    template hipError_t launch_rollout<T>(const VdynParams &, const RolloutArgs<T> &, hipStream_t);
#define VDYN_INSTANTIATE_REST(T)                                                                     \
    template hipError_t launch_rollout_spiral<T>(const VdynParams &, int64_t, int, const T *, const T *, double, \
                                                 double, double, double, const double *, T *, T *, int, hipStream_t); \
    template hipError_t launch_nonfinite_lanes<T>(int, int64_t, const T *, int *, unsigned long long *, hipStream_t); \
    template hipError_t launch_fastmath_eval<T>(int, int64_t, const T *, double, T *, T *, hipStream_t); \
    template hipError_t launch_planar_model<T>(const VdynParams &, int64_t, const T *, const T *,    \
                                               const T *, T *, T *, T *, T *, hipStream_t);          \
    template hipError_t launch_mpc_argmin<T>(const VdynParams &, int, int, int, const T *, const T *, \
                                             const T *, double, double, T *, int *, T *, void *, hipStream_t); \
    template size_t mpc_scratch_bytes<T>(int, int, int);                                             \
    template hipError_t launch_closed_loop<T>(const VdynParams &, const VdynCtrlGains &,             \
                                              const ClosedLoopArgs<T> &, hipStream_t);               \
    template hipError_t launch_controller_update<T>(const VdynCtrlGains &, const ClosedLoopArgs<T> &, hipStream_t); \
    template size_t closed_loop_aux_bytes<T>(int, int, bool);                                        \
    template hipError_t launch_select_best_path<T>(const SelectArgs<T> &, hipStream_t);             \
    template hipError_t launch_rollout_fleet<T>(const RolloutArgs<T> &, bool, hipStream_t);          \
    template void build_fleet_table<T>(const VdynParams *, int, const double *, T *, bool *);        \
    template int fleet_table_len<T>(int);                                                            \
    template hipError_t launch_plan_lattice<T>(const LatticeArgs<T> &, hipStream_t);                 \
    template hipError_t launch_interpolate_waypoints<T>(int, int, int, const T *, const int *, double, int, T *, \
                                                        int *, hipStream_t);
#if !defined(VDYN_PART) || VDYN_PART == 1
#define VDYN_INSTANTIATE_1(T) VDYN_INSTANTIATE_ROLLOUT(T)
#else
#define VDYN_INSTANTIATE_1(T)
#endif
#if !defined(VDYN_PART) || VDYN_PART == 2
#define VDYN_INSTANTIATE_2(T) VDYN_INSTANTIATE_REST(T)
#else
#define VDYN_INSTANTIATE_2(T)
#endif
#if !defined(VDYN_ONLY_F64)
VDYN_INSTANTIATE_1(float)
VDYN_INSTANTIATE_2(float)
#if !defined(VDYN_PART) || VDYN_PART == 2
bool tire_fit_coefficients(double C, float *coef) { return fit_tire_wheel<float, kTireFitDeg>(C, coef, kTireFitTol32); }
bool tire_fit_coefficients64(double C, double *coef) { return fit_tire_wheel<double, kTireFitDeg64>(C, coef, kTireFitTol64); }
#endif
#endif
#if !defined(VDYN_ONLY_F32)
VDYN_INSTANTIATE_1(double)
VDYN_INSTANTIATE_2(double)
#endif

}  // namespace vdyn
