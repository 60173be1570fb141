// C ABI of libptau_hip.so (include/ptau.h): host orchestration of the HIP kernels.
//
// Host-side equivalent of the phase1-cli / phase1 drivers the operator calls
// (/root/reference/src/bin/contribute.rs:811-823, 968-986; src/utils.rs:326-352): chunk geometry,
// buffer split, per-vector kernel launches, BLAKE2b hash chain, file-level contribute.
// No CPU compute fallback exists: without a usable device every compute call fails.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <chrono>
#include <thread>
#include <deque>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptau.h"
#include "blake2b_host.hpp"
#include "kernels_impl.hpp"
#include "pairing_dev.hpp"
#include "pairing_host.hpp"

namespace ptau {
int host_validate_points(int curve, int group, const uint8_t *pts, size_t n);   // keygen.hip
}

namespace {

using ptau::GroupOps;

struct Err {
    char *buf;
    size_t len;
    int fail(int code, const char *fmt, ...) const {
        if (buf && len) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, len, fmt, ap);
            va_end(ap);
        }
        return code;
    }
};

const GroupOps *get_ops(int curve, int group) {
    switch (curve) {
    case PTAU_CURVE_BLS12_377: return ptau::ops_bls12_377(group);
    case PTAU_CURVE_BW6_761: return ptau::ops_bw6_761(group);
    case PTAU_CURVE_MNT4_753: return ptau::ops_mnt4_753(group);
    case PTAU_CURVE_MNT6_753: return ptau::ops_mnt6_753(group);
    }
    return nullptr;
}

const ptau::PairingOps *get_pairing_ops(int curve) {
    switch (curve) {
    case PTAU_CURVE_BLS12_377: return ptau::pairing_ops_bls12_377();
    case PTAU_CURVE_BW6_761: return ptau::pairing_ops_bw6_761();
    case PTAU_CURVE_MNT4_753: return ptau::pairing_ops_mnt4_753();
    case PTAU_CURVE_MNT6_753: return ptau::pairing_ops_mnt6_753();
    }
    return nullptr;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return err.fail(PTAU_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));         \
    } while (0)

// Host worker threads that only stream bytes (BLAKE2b of a file, page population) step back behind the threads that feed the
// GPU: with several in-process lanes on a box with fewer cores than threads, a lane's launching thread that waits for a CPU
// leaves the device idle inside the section that holds the device lock (measured: 8 lanes on 16 threads, 150 ms of "device
// section" for 98 ms of kernels).  Raising one's own niceness needs no privilege; PTAU_NICE_WORKERS=0 turns it off.
inline void background_worker() {
    static const bool on = [] { const char *e = getenv("PTAU_NICE_WORKERS"); return !(e && e[0] == '0'); }();
    if (on) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);
}

// PTAU_TRACE=1: wall-clock marks of the file-level calls on stderr (where the host time of a call goes)
struct Trace {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    static bool on() {
        static const bool v = [] { const char *e = getenv("PTAU_TRACE"); return e && e[0] == '1'; }();
        return v;
    }
    void mark(const char *label) const {
        if (!on()) return;
        fprintf(stderr, "[ptau trace] %s: %-34s %8.1f ms\n", what, label,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

constexpr int MAX_DEVICES = 16;
constexpr uint64_t MAX_LAUNCH_POINTS = 1ull << 20;   // points per kernel launch (scratch sizing)
constexpr uint64_t NORMALIZE_GROUP = 32;             // points per inversion in normalize_kernel
constexpr uint64_t MAX_WINDOW_TABLE_BYTES = 6ull << 30;   // per-point window tables (8 Jacobian multiples each)

// PTAU_SCALAR_MUL=bits selects the plain double-and-add kernel (A/B measurements); default: windowed.
bool use_window_kernel() {
    static const bool v = [] {
        const char *e = getenv("PTAU_SCALAR_MUL");
        return !(e && strcmp(e, "bits") == 0);
    }();
    return v;
}

// setup_utils::BatchExpMode at the boundary (include/ptau.h).  The endomorphism-accelerated multiplications (GLV on
// BLS12-377 G1 / BW6-761, GLS on BLS12-377 G2) equal [k]P only on the prime-order subgroup, where phi / psi act as
// [lambda] / [x]; arkworks' batch_exp computes [k]P for ANY curve point.  So they run only when membership was
// established in the same call (check_input >= OnlyInGroup runs the subgroup kernels first) or the caller asserts it
// (PTAU_BATCH_EXP_INPUT_IN_SUBGROUP: the challenge came out of a verifier that ran the subgroup checks); every other
// combination takes the plain signed-window multiplication, exact on the whole curve.  Returns < 0 on a bad mode.
// 2 = "probe": membership unknown, mode Auto — where the membership test is much cheaper than what the endomorphism saves
// (BLS12-377 G2: psi(P) == [x]P is 64 doublings, GLS saves 188 of the 252) the vector is TESTED first and takes the
// endomorphism path only if every element passed; otherwise the plain path runs.  Same bytes either way.
int endo_allowed(int check_input, int batch_exp_mode) {
    const int mode = batch_exp_mode & ~PTAU_BATCH_EXP_INPUT_IN_SUBGROUP;
    if (mode != PTAU_BATCH_EXP_AUTO && mode != PTAU_BATCH_EXP_DIRECT && mode != PTAU_BATCH_EXP_BATCH_INVERSION) return -1;
    if (mode == PTAU_BATCH_EXP_DIRECT) return 0;
    if (batch_exp_mode & PTAU_BATCH_EXP_INPUT_IN_SUBGROUP) return 1;
    return (check_input & 2) ? 1 : 2;
}

struct DeviceCtx {
    std::mutex mu;
    bool ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    void *scratch = nullptr, *pre = nullptr, *wtab = nullptr;
    size_t scratch_cap = 0, pre_cap = 0, wtab_cap = 0;
    uint32_t *d_table = nullptr;   // scalar table (65 entries x 24 limbs max)
    int *d_err = nullptr;
    void *d_in = nullptr, *d_out = nullptr;   // staging for the host-buffer entry points
    void *d_small = nullptr;                  // single-point results (MSM)
    void *msm_ws = nullptr;                   // bucket-MSM workspace
    size_t msm_cap = 0;
    // pairing kernel (K5): its own lock, stream and buffer — a launch is a handful of workgroups for ~50 ms, which must
    // not hold up another lane's vector kernels (they run side by side on the device)
    std::mutex pair_mu;
    hipStream_t pair_stream = nullptr;
    void *d_pair = nullptr;                   // packed checks | results | GT values
    size_t pair_cap = 0;
    size_t in_cap = 0, out_cap = 0;
    // contribute: the NEXT vector's input is copied to the device (copy stream, second input buffer) while this vector's kernels run
    void *d_in_alt = nullptr;
    size_t in_alt_cap = 0;
    int in_sel = 0;
    hipEvent_t pev[2] = {};   // brackets of the prefetch copy
    struct Prefetch {
        const void *host = nullptr;
        size_t bytes = 0;
        void *dev = nullptr;
    } prefetch;
    // side stream for the lone beta_g2 element of a chunk: its membership test is ONE thread's scalar multiplication
    // (~170 ms of latency on MNT4 G2, ~370 ms on MNT6 G2) and runs beside the vectors instead of after them
    hipStream_t side_stream = nullptr;
    uint8_t *s_buf = nullptr;   // [in 1 KB | out 1 KB | order 256 B | err 16 B | scratch 1 J | wtab 8 J]
    // verify / combine: the D2H copy of vector k is deferred until vector k+1's first kernels are in the queue, so the
    // ~25 ms a 100-200 MB copy into freshly mapped file pages takes overlap with compute (two output buffers alternate)
    hipStream_t copy_stream = nullptr;
    hipEvent_t cev[2] = {};
    void *d_out_alt = nullptr;
    size_t out_alt_cap = 0;
    int out_sel = 0;
    struct PendingCopy {
        const void *src = nullptr;
        uint8_t *dst = nullptr;
        size_t bytes = 0;
        std::atomic<size_t> *ready = nullptr;   // progress mark of the streaming new-challenge hash
        size_t mark = 0;
    } pending;
    // verify: the TAIL of a vector's bucket MSM (bucket reduction -> row sums -> Horner -> normalisation: a few waves of
    // serial latency, 3 ms on BLS12-377 G1, 13 ms on G2) runs on its own stream beside the NEXT vector's square roots and
    // membership tests; its results are collected by finish_msm_tail() before the next MSM touches the workspace
    hipStream_t tail_stream = nullptr;
    void *bk_pre = nullptr;   // membership-fused MSM: inversion scratch of the bucket normalisation (main stream, consumed at once)
    size_t bk_pre_cap = 0;
    // the bucket tests run on the tail stream, beside whatever the main stream does next: their own K1 buffers (tests of
    // consecutive MSMs follow each other on the tail stream, so one set serves both slots)
    void *bk_scratch = nullptr, *bk_wtab = nullptr;
    size_t bk_scratch_cap = 0, bk_wtab_cap = 0;
    struct PendingTail {
        bool active = false;
        uint8_t *out_a = nullptr, *out_b = nullptr;
        int osz = 0;
        uint64_t points = 0;
        bool probe = false;   // a membership probe: its main-stream time is membership time, there are no sums to fetch
    };
    // TWO sets of everything a tail owns — its buckets / parts / totals in the workspace (carved per slot), the bucket sums as
    // points, the small result / error / scalar buffer, its events — so that the tail of vector k may still run while vector
    // k + 1's bucket sums are queued (the G2 tail is 20 ms of one-wave kernels; the bucket tests of BW6-761 / the MNT twists
    // are 10-60 ms): finish_msm_tail() collects slot s only before slot s is used again, two vectors later.
    struct TailSlot {
        hipEvent_t tev[4] = {};   // head start, head end (main stream), tail end (tail stream), buckets consumed (main stream)
        void *bk_bytes = nullptr;   // the bucket sums as uncompressed points (read by the tests on the tail stream)
        size_t bk_bytes_cap = 0;
        uint32_t *bk_table = nullptr;
        uint8_t *t_buf = nullptr;   // [pre 8 KB | results 4 KB | err 16 B | scalar of the bucket tests]
        void *tws = nullptr;        // buckets | parts | totals of this slot's MSM (its own allocation: the other slot's may be in use)
        size_t tws_cap = 0;
        PendingTail tail;
    } slots[2];
    TailSlot *cur = &slots[0];   // the slot the next MSM / finish_msm_tail() works on
    int slot_next = 0;
    // verify / combine: the D2H of a decompressed vector is queued on the copy stream and NOT waited for (CopyTrack); the next
    // writer of the same output buffer waits for it on the device
    hipEvent_t bufev[2] = {};
    bool buf_copying[2] = {false, false};
    int cus = 0;   // compute units of the device
    ptau_timing timing = {};
    double busy_ms = 0;   // kernel time since the library was loaded, every call on this device (under mu)
};
constexpr size_t TB_PRE = 0, TB_SMALL = 8192, TB_ERR = 12288, TB_TABLE = 12544, TB_SIZE = 16384;   // per tail slot
DeviceCtx g_ctx[MAX_DEVICES];
// second context per device (own stream and small buffers) for the lone beta_g2 element of a contribution: its update is
// ONE thread's scalar multiplication (8 ms on BLS12-377, ~130 ms on MNT6 G2) and runs beside the four vectors.  Only used
// while the main context's mutex is held.
DeviceCtx g_side_ctx[MAX_DEVICES];

// Lazy per-device initialisation.  Several in-process lanes (threads) may make their FIRST call at the same time
// (INTEGRATION.md §4): the creation of streams / events / buffers runs under one process-wide mutex, into locals that
// are committed only when every step succeeded, so a cold multi-thread start neither races nor leaks a half-built
// context (a failed step destroys what it had created and the next call starts over).
std::mutex g_init_mu;

int ensure_ctx(int device, DeviceCtx *&c, const Err &err, bool side = false) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return err.fail(PTAU_E_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n || device >= MAX_DEVICES)
        return err.fail(PTAU_E_DEVICE, "device ordinal %d out of range (%d devices)", device, n);
    c = side ? &g_side_ctx[device] : &g_ctx[device];
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> init_lk(g_init_mu);
    if (c->ready) return PTAU_OK;
    hipStream_t st[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evs[20] = {};
    void *bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t buf_sz[5] = {sizeof(uint32_t) * (1 + ptau::SCALAR_TABLE_ENTRIES) * 24, 256, 4096, 16384, 2 * TB_SIZE};
    hipError_t e = hipSuccess;
    const char *what = "";
    // Streams of one priority share a small pool of hardware queues (four by default), and two streams on one queue run
    // their kernels strictly one after the other: measured with four lanes, the tail stream's kernels never overlapped ANY
    // lane's 50 ms pairing kernel — every verification waited for somebody's pairing inside its device section (+51 ms).
    // So the three kinds of work get three priorities, i.e. three pools: the few-wave kernels a lane waits for (MSM tails,
    // bucket tests) high, the long kernels nobody waits for inside a device section (pairing, the lone beta_g2 element) low,
    // vector kernels and copies normal.  The side context (contribute's beta_g2) only ever uses its main and copy streams.
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = prio_greatest = 0; }
    // PTAU_TAIL_PRIO=normal: the tail stream at the main stream's priority (A/B runs)
    const char *tp = getenv("PTAU_TAIL_PRIO");
    const int prio[5] = {0, prio_least, 0, prio_least, (tp && tp[0] == 'n') ? 0 : prio_greatest};   // main, side, copy, pair, tail
    for (int i = 0; i < 5 && e == hipSuccess; i++) {
        if (side && i != 0 && i != 2) continue;
        e = hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, std::min(std::max(prio[i], prio_greatest), prio_least));
        what = "hipStreamCreateWithPriority";
    }
    for (int i = 0; i < 20 && e == hipSuccess; i++) { e = hipEventCreate(&evs[i]); what = "hipEventCreate"; }
    for (int i = 0; i < 5 && e == hipSuccess; i++) { e = hipMalloc(&bufs[i], buf_sz[i]); what = "hipMalloc"; }
    if (e != hipSuccess) {
        for (auto s : st) if (s) (void)hipStreamDestroy(s);
        for (auto v : evs) if (v) (void)hipEventDestroy(v);
        for (auto b : bufs) if (b) (void)hipFree(b);
        return err.fail(PTAU_E_DEVICE, "device context initialisation: %s failed: %s", what, hipGetErrorString(e));
    }
    c->stream = st[0];
    c->side_stream = st[1];
    c->copy_stream = st[2];
    c->pair_stream = st[3];
    c->tail_stream = st[4];
    for (int i = 0; i < 6; i++) c->ev[i] = evs[i];
    c->cev[0] = evs[6];
    c->cev[1] = evs[7];
    for (int sl = 0; sl < 2; sl++) {
        for (int i = 0; i < 4; i++) c->slots[sl].tev[i] = evs[(sl ? 16 : 8) + i];
        c->slots[sl].t_buf = (uint8_t *)bufs[4] + sl * TB_SIZE;
        c->slots[sl].bk_table = (uint32_t *)(c->slots[sl].t_buf + TB_TABLE);
        c->slots[sl].tail = DeviceCtx::PendingTail{};
    }
    c->cur = &c->slots[0];
    c->slot_next = 0;
    c->pev[0] = evs[12];
    c->pev[1] = evs[13];
    c->bufev[0] = evs[14];
    c->bufev[1] = evs[15];
    c->d_table = (uint32_t *)bufs[0];
    c->d_err = (int *)bufs[1];
    c->d_small = bufs[2];
    c->s_buf = (uint8_t *)bufs[3];
    {
        hipDeviceProp_t prop;
        c->cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    c->ready = true;
    return PTAU_OK;
}

int grow(void *&p, size_t &cap, size_t need, const Err &err) {
    if (need <= cap) return PTAU_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, need));
    cap = need;
    return PTAU_OK;
}

// Start of every device section (caller holds c->mu): select the device and forget whatever an earlier call — possibly one
// that FAILED half way — left in the per-device state.  A deferred copy must never land, a prefetch record must not match a
// new caller's pointer, and an MSM tail queued by a failed call is of no interest but its kernels must be done before its
// slot (buffers, result pointers into that call's dead frame) is used again: finish_msm_tail() on a stale active slot would
// copy results to freed host memory.
int begin_device_section(DeviceCtx *c, int device, const Err &err) {
    HIP_TRY(hipSetDevice(device));
    c->timing = ptau_timing{};
    c->pending.bytes = 0;
    if (c->prefetch.host) (void)hipStreamSynchronize(c->copy_stream);
    c->prefetch = DeviceCtx::Prefetch{};
    for (auto &sl : c->slots)
        if (sl.tail.active) {
            HIP_TRY(hipEventSynchronize(sl.tev[2]));
            sl.tail = DeviceCtx::PendingTail{};
        }
    c->slot_next = 0;
    c->cur = &c->slots[0];
    return PTAU_OK;
}

// Per-call timings belong to the CALL, not to the device: several in-process lanes share a device, each resets c->timing
// when it takes the lock, and a verification finishes its counters (ratio_checks, pok_checks, d2h_ms) after it has released
// the lock.  So every entry point publishes its own record to the calling thread when it ends (ptau_last_timing reads the
// calling thread's record); a call that goes on after its device section detaches a private copy first.
thread_local ptau_timing tl_timing[MAX_DEVICES];
struct CallTiming {   // declare AFTER the lock guard: the destructor then reads c->timing while the lock is still held
    DeviceCtx *c;
    int device;
    bool detached = false;
    ptau_timing mine = {};
    CallTiming(DeviceCtx *ctx, int dev) : c(ctx), device(dev) { c->timing = ptau_timing{}; }   // (a call that fails before its first kernel reports zeros, not its predecessor's record)
    void detach() { mine = c->timing; detached = true; }   // under the lock, just before it is released
    ~CallTiming() { tl_timing[device] = detached ? mine : c->timing; }
};

// The output staging buffer `sel` (d_out / d_out_alt), at least `need` bytes.  A verification queues the device-to-host copy
// of a decompressed vector on the copy stream and does NOT wait for it inside its device section (CopyTrack), so such a copy
// may still be reading the buffer when the NEXT holder of the device lock — any entry point, another lane — wants to write
// it: every writer orders its stream behind that copy (on the device), or waits on the host when the buffer is about to be
// reallocated.
int claim_out_buffer(DeviceCtx *c, int sel, size_t need, void *&buf, const Err &err) {
    void *&d = sel ? c->d_out_alt : c->d_out;
    size_t &cap = sel ? c->out_alt_cap : c->out_cap;
    if (c->buf_copying[sel]) {
        if (hipEventQuery(c->bufev[sel]) == hipSuccess) {   // the copy has landed long ago: nothing to order behind
            c->buf_copying[sel] = false;
        } else if (cap < need) {
            (void)hipGetLastError();   // (hipErrorNotReady from the query is not an error)
            HIP_TRY(hipEventSynchronize(c->bufev[sel]));
            c->buf_copying[sel] = false;
        } else {
            (void)hipGetLastError();
            HIP_TRY(hipStreamWaitEvent(c->stream, c->bufev[sel], 0));
        }
    }
    int rc = grow(d, cap, need, err);
    if (rc) return rc;
    buf = d;
    return PTAU_OK;
}

int errbits_to_code(int bits, const Err &err) {
    if (!bits) return PTAU_OK;
    if (bits & ptau::ERRBIT_NONCANONICAL) return err.fail(PTAU_E_POINT, "non-canonical field element in input");
    if (bits & ptau::ERRBIT_NOT_ON_CURVE) return err.fail(PTAU_E_POINT, "compressed x coordinate is not on the curve");
    if (bits & ptau::ERRBIT_INFINITY) return err.fail(PTAU_E_POINT_AT_INFINITY, "point at infinity where a non-zero element is required");
    return err.fail(PTAU_E_NOT_IN_SUBGROUP, "element is not in the prime-order subgroup");
}

int enqueue_subgroup_kernels(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                             const Err &err, bool side = false);
int run_subgroup_check_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                           const Err &err);
int flush_pending_copy(DeviceCtx *c, const Err &err);
int enqueue_batched_subgroup_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                                 const Err &err);
int finish_msm_tail(DeviceCtx *c, const Err &err, hipEvent_t *main_end);
bool use_batched_subgroup();
bool use_endo_subgroup();
bool cofactor_is_one(int curve, int group);
constexpr uint64_t BATCHED_SUBGROUP_MIN = 1ull << 12;

// One vector on device-resident buffers; caller holds the device mutex and has set the device.
// check_input (setup_utils::CheckForCorrectness): bit 0 = non-zero, bit 1 = in the prime-order subgroup (and on the curve)
// — the second runs the membership kernels over the whole vector first, which is also what makes the endomorphism
// (GLV / GLS) paths exact: they assume subgroup elements, as every element of a verified challenge is.
int run_apply_powers_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint8_t *d_out, int cout,
                         uint64_t n, uint64_t start, const uint64_t *tau, const uint64_t *coeff, int check_input,
                         int use_endo, const Err &err) {
    const int isz = cin ? ops->comp : ops->uncomp, osz = cout ? ops->comp : ops->uncomp;
    std::vector<uint32_t> table((1 + ptau::SCALAR_TABLE_ENTRIES) * ops->fr_limbs32);
    if (check_input & 2) {
        int rc = run_subgroup_check_dev(c, curve, group, ops, d_in, cin, n, err);
        if (rc) return rc;
    }
    bool speculative = false;
    if (use_endo == 2) {   // Auto, membership unknown
        use_endo = 0;
        // long vectors: the batched test (2^-128, setup-utils' SubgroupCheckMode::Batched).  Its bucket sums cost about one pass
        // of mixed additions on the main stream; the tests of those sums are a latency-bound launch (a few thousand scalar
        // multiplications: 1 ms on BLS12-377 G1, 10 ms on BW6-761, 35-60 ms on the MNT twists) on the tail stream, and the
        // endomorphism kernel is queued behind the sums at once — SPECULATIVELY, beside the tests.  The verdict is read when
        // both are done; if it is "not a subgroup vector" (or not even valid points) the result is thrown away and the plain
        // path runs, which reports bad points as it always did.  Short vectors: the exact test where it is cheap (BLS12-377 G2).
        const bool batched = ops->auto_probe_pays && n > BATCHED_SUBGROUP_MIN && use_batched_subgroup() && !cin &&
                             !cofactor_is_one(curve, group);
        if (batched) {
            int rc = enqueue_batched_subgroup_dev(c, curve, group, ops, d_in, cin, n, err);
            if (rc) return rc;
            speculative = true;
            use_endo = 1;
        } else if (ops->membership_cheap && n) {
            // "not a subgroup vector" (or not even valid points: the plain path then reports them as it always did) selects
            // the plain path; anything else — a HIP error, a failed allocation — is the caller's to see
            char pmsg[200] = {0};
            const int prc = run_subgroup_check_dev(c, curve, group, ops, d_in, cin, n, Err{pmsg, sizeof(pmsg)});
            if (prc == PTAU_OK) use_endo = 1;
            else if (prc != PTAU_E_NOT_IN_SUBGROUP && prc != PTAU_E_POINT && prc != PTAU_E_POINT_AT_INFINITY)
                return err.fail(prc, "membership probe of BatchExpMode::Auto: %s", pmsg);
        }
    }
    const bool windowed = use_window_kernel();
    const uint64_t launch_cap = windowed
        ? std::max<uint64_t>(256, std::min(MAX_LAUNCH_POINTS, MAX_WINDOW_TABLE_BYTES / (ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes)))
        : MAX_LAUNCH_POINTS;
    for (uint64_t off = 0; off < n; off += launch_cap) {
        const uint64_t m = std::min(launch_cap, n - off);
        int rc;
        if ((rc = grow(c->scratch, c->scratch_cap, m * ops->jac_bytes, err))) return rc;
        if (windowed && (rc = grow(c->wtab, c->wtab_cap, m * ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes, err))) return rc;
        if ((rc = grow(c->pre, c->pre_cap, m * ops->fe_bytes, err))) return rc;
        ops->build_table(tau, coeff, start + off, table.data());
        HIP_TRY(hipMemcpyAsync(c->d_table, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        if (windowed)
            ops->window_scalar_mul(c->stream, d_in + off * isz, cin, c->scratch, c->wtab, m,
                                   ptau::ScalarSource{ptau::SCALAR_TAU_TABLE, c->d_table, 0, 0, 0}, check_input, use_endo, c->d_err);
        else
            ops->scalar_mul(c->stream, d_in + off * isz, cin, c->scratch, m, c->d_table, check_input, c->d_err);
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        // points per inversion: enough threads to fill the chip (>= 64 Ki) while amortising the inversion
        const uint64_t grp = std::max<uint64_t>(4, std::min<uint64_t>(NORMALIZE_GROUP, m >> 16));
        const uint64_t T = (m + grp - 1) / grp;
        ops->normalize(c->stream, c->scratch, c->pre, d_out + off * osz, cout, m, T, 0, c->d_err);
        HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        if ((rc = flush_pending_copy(c, err))) return rc;   // the previous vector's deferred D2H lands behind these kernels
        int bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipGetLastError());
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
        c->timing.scalar_mul_ms += a;
        c->timing.normalize_ms += b;
        c->timing.kernel_ms += a + b;
        c->busy_ms += a + b;
        if (speculative) {   // the probe's verdict (its tests ran beside the launch just waited for)
            speculative = false;
            char pmsg[200] = {0};
            const int prc = finish_msm_tail(c, Err{pmsg, sizeof(pmsg)}, &c->ev[2]);
            if (prc == PTAU_E_NOT_IN_SUBGROUP || prc == PTAU_E_POINT || prc == PTAU_E_POINT_AT_INFINITY) {
                use_endo = 0;          // not a subgroup vector: everything again on the plain path
                off = 0 - launch_cap;  // (the loop's increment brings it back to 0)
                continue;
            }
            if (prc) return err.fail(prc, "membership probe of BatchExpMode::Auto: %s", pmsg);
        }
        c->timing.points += m;
        if ((rc = errbits_to_code(bits, err))) return rc;
    }
    return PTAU_OK;
}

// defer_copy: leave the D2H of the result pending (it lands while the NEXT vector's kernels run — flush_pending_copy);
// the caller flushes after its last vector.
int run_apply_powers_host(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *in, int cin, uint8_t *out, int cout, uint64_t n,
                          uint64_t start, const uint64_t *tau, const uint64_t *coeff, int check_input, int use_endo, const Err &err,
                          bool defer_copy = false, const uint8_t *next_in = nullptr, size_t next_bytes = 0) {
    const size_t isz = cin ? ops->comp : ops->uncomp, osz = cout ? ops->comp : ops->uncomp;
    if (!n) return PTAU_OK;
    int rc;
    void *d_out = nullptr;
    const int osel = c->out_sel;
    c->out_sel ^= 1;
    if ((rc = claim_out_buffer(c, osel, n * osz, d_out, err))) return rc;   // (behind another lane's copy out of it, if any)
    float ms = 0;
    void *d_in;
    if (c->prefetch.host == in && c->prefetch.bytes == n * isz) {   // copied while the previous vector's kernels ran
        d_in = c->prefetch.dev;
        HIP_TRY(hipStreamWaitEvent(c->stream, c->pev[1], 0));
        HIP_TRY(hipEventSynchronize(c->pev[1]));
        HIP_TRY(hipEventElapsedTime(&ms, c->pev[0], c->pev[1]));
    } else {
        void *&buf = c->in_sel ? c->d_in_alt : c->d_in;
        size_t &cap = c->in_sel ? c->in_alt_cap : c->in_cap;
        if ((rc = grow(buf, cap, n * isz, err))) return rc;
        d_in = buf;
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        HIP_TRY(hipMemcpyAsync(d_in, in, n * isz, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev[4], c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
    }
    c->prefetch = DeviceCtx::Prefetch{};
    c->in_sel ^= 1;
    c->timing.h2d_ms += ms;
    if (next_in && next_bytes) {   // the other input buffer is free: its last reader finished before this call
        void *&nbuf = c->in_sel ? c->d_in_alt : c->d_in;
        size_t &ncap = c->in_sel ? c->in_alt_cap : c->in_cap;
        if ((rc = grow(nbuf, ncap, next_bytes, err))) return rc;
        HIP_TRY(hipEventRecord(c->pev[0], c->copy_stream));
        HIP_TRY(hipMemcpyAsync(nbuf, next_in, next_bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->pev[1], c->copy_stream));
        c->prefetch.host = next_in;
        c->prefetch.bytes = next_bytes;
        c->prefetch.dev = nbuf;
    }
    if ((rc = run_apply_powers_dev(c, curve, group, ops, (const uint8_t *)d_in, cin, (uint8_t *)d_out, cout, n, start, tau, coeff,
                                   check_input, use_endo, err))) {
        if (c->prefetch.host) (void)hipStreamSynchronize(c->copy_stream);   // never leave a copy from the caller's buffer in flight
        c->prefetch = DeviceCtx::Prefetch{};
        return rc;
    }
    c->pending.src = d_out;
    c->pending.dst = out;
    c->pending.bytes = n * osz;
    c->pending.ready = nullptr;
    c->pending.mark = 0;
    return defer_copy ? PTAU_OK : flush_pending_copy(c, err);
}

// Exclusive prefix sums over m bucket counts: offsets[0..m] (entries; offsets[m] = total), cursor[j] = offsets[j]
// (the scatter pass bumps the cursor with atomics) and run_offsets[0..m] over the partial-sum slots of every bucket
// (ptau::msm_bucket_parts: the entry-aligned runs of run_len that touch it).  One workgroup: m is at most a few hundred thousand.
__global__ void __launch_bounds__(1024) exclusive_scan_kernel(uint32_t *__restrict__ counts_cursor, uint32_t *__restrict__ offsets,
                                                              uint32_t *__restrict__ run_offsets, uint32_t m, uint32_t run_len) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (m + 1023u) / 1024u;
    const uint32_t lo = t * per < m ? t * per : m, hi = lo + per < m ? lo + per : m;
    auto scan = [&](uint32_t mine) {   // exclusive prefix of `mine` over the workgroup; part[1023] = the total afterwards
        __syncthreads();
        part[t] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partials
            const uint32_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        return t ? part[t - 1] : 0u;
    };
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; j++) sum += counts_cursor[j];
    const uint32_t base = scan(sum);
    if (t == 1023) offsets[m] = part[1023];
    uint32_t run = base, rsum = 0;
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t cnt = counts_cursor[j];
        rsum += ptau::msm_bucket_parts(run, cnt, run_len);
        run += cnt;
    }
    uint32_t rrun = scan(rsum);
    if (t == 1023) run_offsets[m] = part[1023];
    run = base;
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t cnt = counts_cursor[j];
        offsets[j] = run;
        run_offsets[j] = rrun;
        counts_cursor[j] = run;
        rrun += ptau::msm_bucket_parts(run, cnt, run_len);
        run += cnt;
    }
}

// PTAU_MSM=naive selects the round-1 form (n windowed scalar multiplications + a tree of sums) for A/B runs.
bool use_bucket_msm() {
    static const bool v = [] {
        const char *e = getenv("PTAU_MSM");
        return !(e && strcmp(e, "naive") == 0);
    }();
    return v;
}

constexpr uint64_t MSM_SLICE_POINTS = 1ull << 22;   // points per bucket pass (u32 sort indices, bounded workspace)

using ptau::msm_plan;

// sum_i k_i * P_i (i < n) -> ONE serialized affine point in host memory; with out_b also the SHIFTED sum
// sum_i k_i * P_{i+1} (power_pairs: d_in must then hold n + 1 points).  Bucket method (K4); caller holds the
// device mutex.
// Entries per accumulate thread.  Every run costs the same (run_len mixed additions) and the pass takes
// ceil(runs / resident lanes) rounds of them, so the run length is what makes the runs of a launch fill the resident lanes a
// whole number of times: 11.5 M entries (2^20 points, 11 windows) on 131 k resident lanes are ONE round of 88-entry runs —
// runs of 64 would be 1.4 rounds, i.e. two.  PTAU_MSM_RUN=<n> forces a length (A/B runs).
uint32_t msm_run_length(const DeviceCtx *c, const GroupOps *ops, uint64_t m, const ptau::MsmPlan &pl) {
    static const uint32_t forced = [] {
        const char *e = getenv("PTAU_MSM_RUN");
        const long v = e ? atol(e) : 0;
        // clamped to what the workspace (sz_runs: one partial per MSM_RUN entries) and msm_plan's resident-lane model assume
        return (uint32_t)(v > 0 ? std::min<long>(ptau::MSM_RUN_MAX, std::max<long>(ptau::MSM_RUN, v)) : 0);
    }();
    if (forced) return forced;
    const uint64_t resident = (uint64_t)std::max(1, c->cus) * 512 / ops->msm_lanes_per_run;   // two waves per SIMD (199-256 VGPRs)
    const uint64_t total = (uint64_t)pl.W * m;   // upper bound: zero digits drop out
    const uint64_t k = std::max<uint64_t>(1, (total + resident * ptau::MSM_RUN_MAX - 1) / (resident * ptau::MSM_RUN_MAX));
    const uint64_t len = (total + k * resident - 1) / (k * resident);
    return (uint32_t)std::min<uint64_t>(ptau::MSM_RUN_MAX, std::max<uint64_t>(ptau::MSM_RUN, len));
}

// defer_tail: everything after the last slice's bucket sums goes to the tail stream and NOTHING is waited for; out_a / out_b
// are written by finish_msm_tail(), which the caller must reach before the next MSM on this context.
// fuse (PRF scalars): the membership-fused form — msm_plan's comment has the argument.  Every bucket sum is normalised and put
// through the group's exact membership test (enqueue_subgroup_kernels) ON THE TAIL STREAM: a few thousand tests are a
// latency-bound launch (one scalar multiplication per thread: ~1 ms on BLS12-377 G1, 10 ms on BW6-761, 35-60 ms on the MNT
// twists) that runs beside whatever the main stream does next.  A failure lands in the tail's error word as
// PTAU_E_NOT_IN_SUBGROUP.  With both sums (the verify path): sum A covers points 0 .. n-1, sum B points 1 .. n.
// only: nothing but the membership verdict is wanted (the probe of BatchExpMode::Auto, run_batched_subgroup_dev): one sum over
// n scalars and n points, no reduction of the buckets; finish_msm_tail() returns the verdict.
struct MsmFuse { int curve, group; bool only = false; };
int run_msm_bucket_dev(DeviceCtx *c, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n, ptau::ScalarSource src,
                       uint8_t *out_a, uint8_t *out_b, int cout, const Err &err, bool defer_tail = false,
                       const MsmFuse *fuse = nullptr) {
    if (fuse && !(defer_tail && src.mode == ptau::SCALAR_PRF && (fuse->only ? !out_b : out_b != nullptr)))
        return err.fail(PTAU_E_ARG, "membership-fused MSM: wrong call form");
    const bool only = fuse && fuse->only;
    // PTAU_TEST_COST_SCALE=<x>: weight of one bucket test in the window-width choice, times x (A/B runs: fewer, larger buckets
    // against more of them — the tests are latency-bound waves that hold SIMD slots beside whatever the main stream runs)
    static const double test_cost_scale = [] { const char *e = getenv("PTAU_TEST_COST_SCALE"); const double v = e ? atof(e) : 1.0; return v > 0 ? v : 1.0; }();
    const int test_cost = fuse ? std::max(1, (int)(ops->subgroup_cost_adds * test_cost_scale)) : 0;
    const uint64_t resident_runs = (uint64_t)std::max(1, c->cus) * 512 / ops->msm_lanes_per_run;
    if (c->cur->tail.active) {   // left behind by a call that failed: its results are of no interest, its kernels must be done
        HIP_TRY(hipEventSynchronize(c->cur->tev[2]));
        c->cur->tail.active = false;
    }
    const int isz = cin ? ops->comp : ops->uncomp, osz = cout ? ops->comp : ops->uncomp;
    const int nsum = out_b ? 2 : 1;
    const int scalar_bits = src.mode == ptau::SCALAR_PRF ? ptau::PRF_BITS : ops->fr_bits;
    const uint64_t nslice = n ? (n + MSM_SLICE_POINTS - 1) / MSM_SLICE_POINTS : 1;
    const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(n, 1), MSM_SLICE_POINTS);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // workspace carve-up (sized for the largest slice; W*B peaks at the largest c)
    size_t wb_max = 0, w_max = 0;
    for (uint64_t m : {cap, n % MSM_SLICE_POINTS ? n % MSM_SLICE_POINTS : cap}) {
        const ptau::MsmPlan pl = msm_plan(std::max<uint64_t>(m, 1), scalar_bits, 0, test_cost, resident_runs);
        wb_max = std::max(wb_max, (size_t)pl.W * pl.B);
        w_max = std::max(w_max, (size_t)pl.W);
    }
    const size_t sz_aff = al((cap + 1) * ops->affine_bytes), sz_dig = al(w_max * cap * 4), sz_sorted = sz_dig,
                 sz_counts = al(wb_max * 4), sz_offsets = al((wb_max + 1) * 4), sz_buckets = al(nsum * wb_max * ops->jac_bytes),
                 sz_parts = sz_buckets, sz_totals = al((nsum * (nslice + 1) + 8) * ops->jac_bytes),
                 sz_runs = al((wb_max + w_max * cap / ptau::MSM_RUN + 2) * ops->jac_bytes);   // <= one partial per bucket + one per run
    int rc;
    if ((rc = grow(c->msm_ws, c->msm_cap, sz_aff + sz_dig + sz_sorted + sz_counts + 2 * sz_offsets + sz_runs, err))) return rc;
    if ((rc = grow(c->cur->tws, c->cur->tws_cap, sz_buckets + sz_parts + sz_totals, err))) return rc;
    if ((rc = grow(c->pre, c->pre_cap, 16 * ops->fe_bytes, err))) return rc;
    uint8_t *w8 = (uint8_t *)c->msm_ws;
    void *aff = w8;                                   w8 += sz_aff;
    uint32_t *dig = (uint32_t *)w8;                   w8 += sz_dig;
    uint32_t *sorted = (uint32_t *)w8;                w8 += sz_sorted;
    uint32_t *counts = (uint32_t *)w8;                w8 += sz_counts;
    uint32_t *offsets = (uint32_t *)w8;               w8 += sz_offsets;
    uint32_t *run_offsets = (uint32_t *)w8;           w8 += sz_offsets;
    uint8_t *runs = w8;                               // run partials of the accumulate pass
    uint8_t *buckets = (uint8_t *)c->cur->tws;        // what the tail reads lives in the slot's own allocation
    uint8_t *parts = buckets + sz_buckets;
    uint8_t *totals = parts + sz_parts;               // [sum][slice] Jacobians, then the final [sum]
    int *d_err = defer_tail ? (int *)(c->cur->t_buf + TB_ERR) : c->d_err;
    hipStream_t ts = c->stream;   // the stream of the tail kernels
    HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    if (fuse) {   // the fixed scalar of the bucket tests ([r] or [|t - 1|]), ahead of everything the tail stream will wait for
        const bool via_psi = ops->psi_check && use_endo_subgroup();
        HIP_TRY(hipMemcpyAsync(c->cur->bk_table, via_psi ? ops->psi_scalar : ops->group_order, sizeof(uint32_t) * ops->fr_limbs32,
                               hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipEventRecord(defer_tail ? c->cur->tev[0] : c->ev[0], c->stream));
    uint64_t li = 0;
    for (uint64_t off = 0; off < n || (n == 0 && li == 0); off += MSM_SLICE_POINTS, li++) {
        const uint64_t m = std::min(MSM_SLICE_POINTS, n - off);
        const ptau::MsmPlanEx pl = msm_plan(std::max<uint64_t>(m, 1), scalar_bits, 0, test_cost, resident_runs);
        const uint64_t wb = (uint64_t)pl.W * pl.B;
        ptau::ScalarSource s2 = src;
        if (fuse) s2.prf_bits = pl.bits;
        if (s2.mode == ptau::SCALAR_ARRAY) s2.ptr += off * ops->fr_limbs32;
        s2.index_base += off;
        HIP_TRY(hipMemsetAsync(counts, 0, wb * 4, c->stream));
        // (the probe's uncompressed input gets its on-curve test here: bucket sums of points off the curve would mean nothing)
        ops->msm_prepare(c->stream, d_in + off * isz, only && !cin ? 2 : cin, aff, dig, counts, m + (nsum - 1), m, s2, pl, d_err);
        const uint32_t run_len = msm_run_length(c, ops, m, pl);
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, offsets, run_offsets, (uint32_t)wb, run_len);
        if (m) hipLaunchKernelGGL(ptau::msm_scatter_kernel<0>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, dig, counts, sorted, m, pl);
        const uint64_t max_runs = std::max<uint64_t>(1, ((uint64_t)pl.W * m + run_len - 1) / run_len);   // threads of the accumulate pass
        for (int sidx = 0; sidx < nsum; sidx++) {
            ops->msm_accumulate(c->stream, aff, sorted, offsets, run_offsets, runs, (uint32_t)sidx, (uint32_t)wb, max_runs, run_len);
            ops->msm_combine(c->stream, runs, run_offsets, buckets + (size_t)sidx * wb * ops->jac_bytes, wb);
        }
        const bool last_slice = off + MSM_SLICE_POINTS >= n;
        if (fuse) {   // the bucket sums as points (before the reduction below reuses their memory), their tests on the tail stream
            const uint64_t nb = (uint64_t)nsum * wb;
            // BLS12-377 G1: the test takes the Jacobian sums as they are (Z ignored: kernels_impl.hpp endo_membership_lazy) — the
            // tail stream runs it BEFORE the reduction kernels that reuse the buckets' memory; other groups: normalise first
            const bool jac_test = ops->endo_subgroup_jac && use_endo_subgroup();
            if (!jac_test) {
                if ((rc = grow(c->cur->bk_bytes, c->cur->bk_bytes_cap, nb * ops->uncomp, err))) return rc;
                if ((rc = grow(c->bk_pre, c->bk_pre_cap, nb * ops->fe_bytes, err))) return rc;
                ops->normalize(c->stream, buckets, c->bk_pre, (uint8_t *)c->cur->bk_bytes, 0, nb, (nb + 3) / 4, 0, d_err);
            }
            HIP_TRY(hipEventRecord(c->cur->tev[3], c->stream));
            HIP_TRY(hipStreamWaitEvent(c->tail_stream, c->cur->tev[3], 0));
            if (jac_test) ops->endo_subgroup_jac(c->tail_stream, buckets, nb, d_err);
            else if ((rc = enqueue_subgroup_kernels(c, fuse->curve, fuse->group, ops, (const uint8_t *)c->cur->bk_bytes, 0, nb, err, /*side=*/true))) return rc;
            if (!last_slice) {   // the next slice reuses bk_bytes
                HIP_TRY(hipEventRecord(c->cur->tev[2], c->tail_stream));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->cur->tev[2], 0));
            }
        }
        if (defer_tail && last_slice) {   // last slice: the main stream is done with this MSM here
            if (!fuse) {
                HIP_TRY(hipEventRecord(c->cur->tev[3], c->stream));
                HIP_TRY(hipStreamWaitEvent(c->tail_stream, c->cur->tev[3], 0));
            }
            HIP_TRY(hipEventRecord(c->cur->tev[1], c->stream));
            ts = c->tail_stream;
        }
        if (only) continue;
        // rows = nsum * W windows from here on
        const uint32_t rows = (uint32_t)(nsum * pl.W);
        uint32_t cols = pl.B / pl.seg;
        ops->msm_bucket_reduce(ts, buckets, parts, pl, (uint64_t)rows * cols);
        uint8_t *cur = parts, *nxt = buckets;   // ping-pong: the buckets are dead once reduced
        while (cols > 1) {
            const uint32_t T = (cols + 15) / 16;
            ops->reduce_rows(ts, cur, nxt, rows, cols, T);
            std::swap(cur, nxt);
            cols = T;
        }
        ops->msm_horner(ts, cur, totals + (size_t)li * ops->jac_bytes, pl, nsum, nslice);   // sum s -> totals[s * nslice + li]
    }
    uint8_t *fin = totals + (size_t)nsum * nslice * ops->jac_bytes;
    if (!only)
        for (int sidx = 0; sidx < nsum; sidx++)
            ops->reduce_sum(ts, totals + (size_t)sidx * nslice * ops->jac_bytes, fin + (size_t)sidx * ops->jac_bytes, li, 1);
    if (defer_tail) {
        if (!only) ops->normalize(ts, fin, c->cur->t_buf + TB_PRE, c->cur->t_buf + TB_SMALL, cout, nsum, nsum, 0, d_err);
        HIP_TRY(hipEventRecord(c->cur->tev[2], ts));
        HIP_TRY(hipGetLastError());
        c->cur->tail.active = true;
        c->cur->tail.out_a = only ? nullptr : out_a;
        c->cur->tail.out_b = only ? nullptr : out_b;
        c->cur->tail.osz = osz;
        c->cur->tail.points = only ? 0 : n * nsum;
        c->cur->tail.probe = only;
        return PTAU_OK;
    }
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    ops->normalize(c->stream, fin, c->pre, (uint8_t *)c->d_small, cout, nsum, nsum, 0, c->d_err);
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_a, c->d_small, osz, hipMemcpyDeviceToHost, c->stream));
    if (out_b) HIP_TRY(hipMemcpyAsync(out_b, (uint8_t *)c->d_small + osz, osz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->timing.kernel_ms += ms;
    c->busy_ms += ms;
    c->timing.msm_ms += ms;
    c->timing.points += n * nsum;
    return errbits_to_code(bits, err);
}

// Round-1 form of the same sum: n windowed scalar multiplications (the K1 kernel) followed by a tree of
// Jacobian sums.  Kept for A/B measurements (PTAU_MSM=naive).  Caller holds the device mutex.
int run_msm_naive_dev(DeviceCtx *c, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n, ptau::ScalarSource src,
                      uint8_t *out_host, int cout, const Err &err) {
    const int isz = cin ? ops->comp : ops->uncomp, osz = cout ? ops->comp : ops->uncomp;
    const uint64_t launch_cap =
        std::max<uint64_t>(256, std::min(MAX_LAUNCH_POINTS, MAX_WINDOW_TABLE_BYTES / (ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes)));
    int rc;
    const uint64_t nlaunch = n ? (n + launch_cap - 1) / launch_cap : 1;
    // partial sums: one Jacobian per launch, kept at the tail of the scratch area
    if ((rc = grow(c->scratch, c->scratch_cap, (std::min(n, launch_cap) + 4096 + nlaunch + 8) * ops->jac_bytes, err))) return rc;
    if ((rc = grow(c->pre, c->pre_cap, 16 * ops->fe_bytes, err))) return rc;
    uint8_t *S = (uint8_t *)c->scratch;
    const uint64_t cap = std::min(n, launch_cap);
    uint8_t *level = S + cap * ops->jac_bytes;                       // 4096 partials
    uint8_t *totals = level + 4096 * ops->jac_bytes;                 // nlaunch partial totals (+ final)
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    uint64_t li = 0;
    for (uint64_t off = 0; off < n; off += launch_cap, li++) {
        const uint64_t m = std::min(launch_cap, n - off);
        if ((rc = grow(c->wtab, c->wtab_cap, m * ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes, err))) return rc;
        ptau::ScalarSource s2 = src;
        if (s2.mode == ptau::SCALAR_ARRAY) s2.ptr += off * ops->fr_limbs32;
        s2.index_base += off;
        ops->window_scalar_mul(c->stream, d_in + off * isz, cin, S, c->wtab, m, s2, 0, /*use_endo=*/0, c->d_err);
        // tree: m -> min(m,4096)·.. -> 64 -> 1
        const uint64_t t1 = std::min<uint64_t>(4096, (m + 15) / 16);
        ops->reduce_sum(c->stream, S, level, m, t1);
        const uint64_t t2 = std::min<uint64_t>(64, t1);
        ops->reduce_sum(c->stream, level, S, t1, t2);               // reuse the head of S for level 2
        ops->reduce_sum(c->stream, S, totals + li * ops->jac_bytes, t2, 1);
    }
    if (n == 0) {
        ops->reduce_sum(c->stream, S, totals, 0, 1);                // empty sum = infinity
        li = 1;
    }
    ops->reduce_sum(c->stream, totals, totals + nlaunch * ops->jac_bytes, li, 1);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    // normalise + serialise the single result
    ops->normalize(c->stream, totals + nlaunch * ops->jac_bytes, c->pre, (uint8_t *)c->d_small, cout, 1, 1, 0, c->d_err);
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_host, c->d_small, osz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->timing.kernel_ms += ms;
    c->busy_ms += ms;
    c->timing.msm_ms += ms;
    c->timing.points += n;
    return errbits_to_code(bits, err);
}


// Collect a deferred MSM tail: wait for it (it has usually finished long ago), copy the one or two result points out and
// book the time.  kernel_ms gets the head (main stream) plus only what of the tail was NOT hidden: the part that ran past
// `main_end` (the end of the next vector's kernels on the main stream), or all of it when nothing followed (main_end null).
int finish_msm_tail(DeviceCtx *c, const Err &err, hipEvent_t *main_end) {
    if (!c->cur->tail.active) return PTAU_OK;
    c->cur->tail.active = false;
    HIP_TRY(hipEventSynchronize(c->cur->tev[2]));   // this slot's tail (the other slot's may still be running)
    HIP_TRY(hipEventSynchronize(c->cur->tev[1]));   // ... and its head, for the timing below
    HIP_TRY(hipGetLastError());
    int bits = 0;
    HIP_TRY(hipMemcpy(&bits, c->cur->t_buf + TB_ERR, sizeof(int), hipMemcpyDeviceToHost));
    if (c->cur->tail.out_a) HIP_TRY(hipMemcpy(c->cur->tail.out_a, c->cur->t_buf + TB_SMALL, c->cur->tail.osz, hipMemcpyDeviceToHost));
    if (c->cur->tail.out_b) HIP_TRY(hipMemcpy(c->cur->tail.out_b, c->cur->t_buf + TB_SMALL + c->cur->tail.osz, c->cur->tail.osz, hipMemcpyDeviceToHost));
    float head = 0, lat = 0, past = 0;
    HIP_TRY(hipEventElapsedTime(&head, c->cur->tev[0], c->cur->tev[1]));
    HIP_TRY(hipEventElapsedTime(&lat, c->cur->tev[3], c->cur->tev[2]));
    // what of the tail ran past the main stream's own work: past the next vector's kernels, or past this MSM's head
    if (hipEventElapsedTime(&past, main_end ? *main_end : c->cur->tev[1], c->cur->tev[2]) != hipSuccess) { (void)hipGetLastError(); past = lat; }
    past = std::min(std::max(past, 0.0f), lat);
    c->timing.kernel_ms += head + past;
    c->busy_ms += head + past;
    (c->cur->tail.probe ? c->timing.subgroup_ms : c->timing.msm_ms) += head + past;
    c->timing.msm_tail_ms += lat;
    c->timing.points += c->cur->tail.points;
    return errbits_to_code(bits, err);
}

// PTAU_BATCHED_SUBGROUP=0: Auto / Batched run the exact per-element membership test everywhere (A/B runs).
bool use_batched_subgroup() {
    static const bool v = [] {
        const char *e = getenv("PTAU_BATCHED_SUBGROUP");
        return !(e && strcmp(e, "0") == 0);
    }();
    return v;
}

// PTAU_TAIL_SLOTS=1: every MSM tail of a verification uses the same slot, i.e. is collected before the next MSM starts (A/B runs).
bool use_two_tail_slots() {
    static const bool v = [] {
        const char *e = getenv("PTAU_TAIL_SLOTS");
        return !(e && strcmp(e, "1") == 0);
    }();
    return v;
}

// PTAU_MSM_TAIL_OVERLAP=0: the verify path waits for every MSM as a whole (A/B runs).
bool use_msm_tail_overlap() {
    static const bool v = [] {
        const char *e = getenv("PTAU_MSM_TAIL_OVERLAP");
        return !(e && strcmp(e, "0") == 0);
    }();
    return v;
}

int run_msm_dev(DeviceCtx *c, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n, ptau::ScalarSource src,
                uint8_t *out_a, uint8_t *out_b, int cout, const Err &err, bool defer_tail = false, const MsmFuse *fuse = nullptr) {
    if (use_bucket_msm() || fuse) return run_msm_bucket_dev(c, ops, d_in, cin, n, src, out_a, out_b, cout, err, fuse || (defer_tail && use_msm_tail_overlap()), fuse);
    const int isz = cin ? ops->comp : ops->uncomp;
    int rc = run_msm_naive_dev(c, ops, d_in, cin, n, src, out_a, cout, err);
    if (rc || !out_b) return rc;
    return run_msm_naive_dev(c, ops, d_in + isz, cin, n, src, out_b, cout, err);
}

// The batched membership test on its own (setup-utils check_subgroup, SubgroupCheckMode::Batched, SECURITY_PARAM = 128):
// every element lands in W random buckets (one per window of a fresh 128-bit-seeded PRF scalar), the bucket sums go through
// the group's exact test.  msm_plan's comment has the argument; a vector with an element outside the subgroup passes with
// probability <= 2^-128.  Uncompressed input gets its on-curve test first.  Caller holds the device mutex.
// enqueue_...: nothing is waited for — the bucket sums are queued on the main stream, their tests on the tail stream, and
// finish_msm_tail() returns the verdict; whatever the caller queues on the main stream in between runs beside the tests.
int enqueue_batched_subgroup_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                                 const Err &err) {
    uint64_t seed[2];
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(seed, 1, 16, f) != 16) {
        if (f) fclose(f);
        return err.fail(PTAU_E_IO, "cannot read /dev/urandom for the batched subgroup check");
    }
    fclose(f);
    const MsmFuse fz{curve, group, true};
    c->cur = &c->slots[0];   // (one tail at a time here: the caller collects it before the next)
    return run_msm_bucket_dev(c, ops, d_in, cin, n, ptau::ScalarSource{ptau::SCALAR_PRF, nullptr, seed[0], seed[1], 0}, nullptr, nullptr, 0,
                              err, true, &fz);
}
int run_batched_subgroup_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                             const Err &err) {
    if (!n) return PTAU_OK;
    if (cofactor_is_one(curve, group)) return run_subgroup_check_dev(c, curve, group, ops, d_in, cin, n, err);
    int rc = enqueue_batched_subgroup_dev(c, curve, group, ops, d_in, cin, n, err);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return finish_msm_tail(c, err, nullptr);
}

// PTAU_SUBGROUP=direct forces the [r]P == O form on every curve (A/B runs; the accept/reject set is the same).
bool use_endo_subgroup() {
    static const bool v = [] {
        const char *e = getenv("PTAU_SUBGROUP");
        return !(e && strcmp(e, "direct") == 0);
    }();
    return v;
}

bool cofactor_is_one(int curve, int group) {
    return group == PTAU_G1 && (curve == PTAU_CURVE_MNT4_753 || curve == PTAU_CURVE_MNT6_753);
}

// Prime-order-subgroup membership of n device-resident points (setup-utils check_subgroup, Direct mode):
// on-curve test for uncompressed encodings, then [r]P == O with the windowed kernel and ONE public scalar.
// Cofactor-1 groups (MNT G1) need only the on-curve test.  Caller holds the device mutex.
// Enqueue only (no error-word reset, no synchronisation): events ev[0] / ev[1] bracket the membership kernels.
int enqueue_subgroup_kernels(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                             const Err &err, bool side) {
    // side: the bucket tests of a membership-fused MSM — tail stream, the tail's error word, their own K1 buffers and scalar
    // (the main stream may be running the next vector's kernels, or K1 itself, at the same time); no timing events
    int *const d_err = side ? (int *)(c->cur->t_buf + TB_ERR) : c->d_err;
    hipStream_t const st = side ? c->tail_stream : c->stream;
    void *&scratch = side ? c->bk_scratch : c->scratch;
    size_t &scratch_cap = side ? c->bk_scratch_cap : c->scratch_cap;
    void *&wtab = side ? c->bk_wtab : c->wtab;
    size_t &wtab_cap = side ? c->bk_wtab_cap : c->wtab_cap;
    uint32_t *const d_table = side ? c->cur->bk_table : c->d_table;
    const int isz = cin ? ops->comp : ops->uncomp;
    const uint64_t launch_cap =
        std::max<uint64_t>(256, std::min(MAX_LAUNCH_POINTS, MAX_WINDOW_TABLE_BYTES / (ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes)));
    const bool via_psi = ops->psi_check && use_endo_subgroup();   // MNT twists: psi(P) == [t - 1]P, a half-length fixed scalar
    if (!side)   // (side: the caller has loaded bk_table on the main stream, ahead of the event the tail stream waits for)
        HIP_TRY(hipMemcpyAsync(d_table, via_psi ? ops->psi_scalar : ops->group_order, sizeof(uint32_t) * ops->fr_limbs32,
                               hipMemcpyHostToDevice, st));
    if (!side) HIP_TRY(hipEventRecord(c->ev[0], st));
    if (ops->endo_subgroup && use_endo_subgroup()) {
        ops->endo_subgroup(st, d_in, cin, n, d_err);   // on-curve test included
    } else if (!cin) {
        ops->on_curve(st, d_in, n, d_err);
    }
    if (!(ops->endo_subgroup && use_endo_subgroup()) && !cofactor_is_one(curve, group)) {
        for (uint64_t off = 0; off < n; off += launch_cap) {
            const uint64_t m = std::min(launch_cap, n - off);
            int rc;
            if ((rc = grow(scratch, scratch_cap, m * ops->jac_bytes, err))) return rc;
            if ((rc = grow(wtab, wtab_cap, m * ptau::WINDOW_ENTRIES * ops->wtab_entry_bytes, err))) return rc;
            if (ops->subgroup_via_glv && use_endo_subgroup())   // [a]P + [b]phi(P) == O, |a|, |b| ~ sqrt(r): half the doublings, exact
                ops->window_scalar_mul(st, d_in + off * isz, cin, scratch, wtab, m,
                                       ptau::ScalarSource{ptau::SCALAR_SUBGROUP_ENDO, nullptr, 0, 0, 0}, 0, /*use_glv=*/1, d_err);
            else
                ops->window_scalar_mul(st, d_in + off * isz, cin, scratch, wtab, m,
                                       ptau::ScalarSource{via_psi ? ptau::SCALAR_FIXED_PSI : ptau::SCALAR_FIXED, d_table, 0, 0, 0}, 0, /*use_glv=*/0, d_err);
            if (via_psi) ops->psi_check(st, d_in + off * isz, cin, scratch, m, d_err);   // (nothing to launch when K1 compared in-kernel)
            else ops->expect_infinity(st, scratch, m, d_err);
        }
    }
    if (!side) HIP_TRY(hipEventRecord(c->ev[1], st));
    return PTAU_OK;
}

int run_subgroup_check_dev(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *d_in, int cin, uint64_t n,
                           const Err &err) {
    if (!n) return PTAU_OK;
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    int rc = enqueue_subgroup_kernels(c, curve, group, ops, d_in, cin, n, err);
    if (rc) return rc;
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->timing.kernel_ms += ms;
    c->busy_ms += ms;
    c->timing.subgroup_ms += ms;
    c->timing.points += n;
    return errbits_to_code(bits, err);
}

// Land the deferred D2H copy of the previous vector (the host blocks here while the kernels just enqueued run).
int flush_pending_copy(DeviceCtx *c, const Err &err) {
    DeviceCtx::PendingCopy &pc = c->pending;
    if (!pc.bytes) return PTAU_OK;
    float ms = 0;
    HIP_TRY(hipEventRecord(c->cev[0], c->copy_stream));
    HIP_TRY(hipMemcpyAsync(pc.dst, pc.src, pc.bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipEventRecord(c->cev[1], c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    HIP_TRY(hipEventElapsedTime(&ms, c->cev[0], c->cev[1]));
    c->timing.d2h_ms += ms;
    if (pc.ready) pc.ready->store(pc.mark, std::memory_order_release);
    pc.bytes = 0;
    return PTAU_OK;
}

// Device-to-host copies of a verification that nobody waits for inside the device section: each is queued on the copy stream
// between two host callbacks — the first takes the time, the second adds the elapsed time to the call's d2h total, publishes
// the progress mark of the streaming new-challenge hash and counts the copy as landed.  The owner (verify_chunk_impl,
// combine_impl) calls drain() before its buffers go away, on every path.  With a destination that is not registered with the
// HIP runtime hipMemcpyAsync is a blocking staged copy: still correct, only not overlapped.
struct CopyTrack {
    std::atomic<int64_t> d2h_ns{0};
    std::atomic<int> outstanding{0};
    hipStream_t stream = nullptr;
    struct Node {
        CopyTrack *t;
        std::atomic<size_t> *ready;
        size_t mark;
        std::chrono::steady_clock::time_point t0;
        unsigned delay_us;   // test hook (PTAU_TEST_COPY_DELAY_MS): holds the copy back so that it outlives the device section
    };
    static void on_start(void *u) {
        Node *n = (Node *)u;
        if (n->delay_us) usleep(n->delay_us);
        n->t0 = std::chrono::steady_clock::now();
    }
    static void on_end(void *u) {
        Node *n = (Node *)u;
        n->t->d2h_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - n->t0).count());
        if (n->ready) n->ready->store(n->mark, std::memory_order_release);
        CopyTrack *t = n->t;
        delete n;
        t->outstanding.fetch_sub(1, std::memory_order_release);
    }
    void drain() const {
        for (unsigned spins = 1; outstanding.load(std::memory_order_acquire) > 0; spins++) {
            usleep(50);
            // a callback that can no longer come (the stream is idle, or the device is gone) must not hang the caller
            if (spins % 2000 == 0 && stream && hipStreamQuery(stream) != hipErrorNotReady) {
                usleep(2000);
                if (outstanding.load(std::memory_order_acquire) > 0 && hipStreamQuery(stream) != hipErrorNotReady) break;
            }
        }
    }
    ~CopyTrack() { drain(); }
};
int start_vector_copy(DeviceCtx *c, CopyTrack *track, int sel, const void *src, uint8_t *dst, size_t bytes, std::atomic<size_t> *ready,
                      size_t mark, const Err &err) {
    // PTAU_TEST_COPY_DELAY_MS=<ms> (tests only, read per copy): a host callback ahead of the copy sleeps that long, so the copy
    // is certainly still queued when this lane releases the device — the next lane's writers must order themselves behind it
    const char *dly = getenv("PTAU_TEST_COPY_DELAY_MS");
    const long dly_ms = dly ? atol(dly) : 0;
    CopyTrack::Node *node = new CopyTrack::Node{track, ready, mark, {}, (unsigned)(dly_ms > 0 && dly_ms <= 2000 ? dly_ms * 1000 : 0)};
    track->stream = c->copy_stream;
    track->outstanding.fetch_add(1, std::memory_order_acq_rel);
    hipError_t e = hipLaunchHostFunc(c->copy_stream, CopyTrack::on_start, node);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(c->bufev[sel], c->copy_stream);
    if (e == hipSuccess) e = hipLaunchHostFunc(c->copy_stream, CopyTrack::on_end, node);
    if (e != hipSuccess) {   // the end callback will not run: settle the node here, after whatever was queued
        (void)hipStreamSynchronize(c->copy_stream);
        delete node;
        track->outstanding.fetch_sub(1, std::memory_order_release);
        return err.fail(PTAU_E_DEVICE, "queueing a device-to-host copy failed: %s", hipGetErrorString(e));
    }
    c->buf_copying[sel] = true;
    return PTAU_OK;
}

// One response vector of a chunk: compressed host bytes -> (sqrt) -> non-zero / subgroup checks -> uncompressed
// host bytes of the next challenge.  Caller holds the device mutex.
int run_verify_vector(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *in, uint8_t *out, uint64_t n,
                      int check_output, int subgroup_check_mode, const uint64_t *ratio_seed, uint64_t index_base,
                      uint8_t *pair_a, uint8_t *pair_b, std::atomic<size_t> *ready, size_t ready_mark, CopyTrack *track, const Err &err,
                      const uint8_t *next_in = nullptr, size_t next_bytes = 0) {
    if (!n) return PTAU_OK;
    int rc;
    const int sel = c->out_sel;
    void *d_out = nullptr;
    c->out_sel ^= 1;
    // the compressed input: copied while the previous vector's kernels ran (Prefetch, as in contribute), or copied now
    const bool prefetched = c->prefetch.host == in && c->prefetch.bytes == n * ops->comp;
    void *d_in;
    float ms = 0;
    if (prefetched) {
        d_in = c->prefetch.dev;
        HIP_TRY(hipEventSynchronize(c->pev[1]));   // queued a whole vector ago; its events are re-recorded for the next one below
        HIP_TRY(hipEventElapsedTime(&ms, c->pev[0], c->pev[1]));
        c->timing.h2d_ms += ms;
    } else {
        void *&buf = c->in_sel ? c->d_in_alt : c->d_in;
        size_t &cap = c->in_sel ? c->in_alt_cap : c->in_cap;
        if ((rc = grow(buf, cap, n * ops->comp, err))) return rc;
        d_in = buf;
    }
    c->prefetch = DeviceCtx::Prefetch{};
    c->in_sel ^= 1;
    // the copy that last read this output buffer (two vectors ago, or another lane's) finishes before it is overwritten
    if ((rc = claim_out_buffer(c, sel, n * ops->uncomp, d_out, err))) return rc;
    HIP_TRY(hipEventRecord(c->ev[3], c->stream));
    if (!prefetched) HIP_TRY(hipMemcpyAsync(d_in, in, n * ops->comp, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    HIP_TRY(hipEventRecord(c->ev[4], c->stream));
    const int nonzero = (check_output == PTAU_CHECK_ONLY_NON_ZERO || check_output == PTAU_CHECK_FULL) ? 1 : 0;
    ops->decompress(c->stream, (const uint8_t *)d_in, (uint8_t *)d_out, n, nonzero, c->d_err);
    HIP_TRY(hipEventRecord(c->ev[5], c->stream));
    if (next_in && next_bytes) {   // the other input buffer is free: its last reader (the previous vector's square roots) is done
        void *&nbuf = c->in_sel ? c->d_in_alt : c->d_in;
        size_t &ncap = c->in_sel ? c->in_alt_cap : c->in_cap;
        if ((rc = grow(nbuf, ncap, next_bytes, err))) return rc;
        HIP_TRY(hipEventRecord(c->pev[0], c->copy_stream));
        HIP_TRY(hipMemcpyAsync(nbuf, next_in, next_bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->pev[1], c->copy_stream));
        c->prefetch.host = next_in;
        c->prefetch.bytes = next_bytes;
        c->prefetch.dev = nbuf;
    }
    bool want_subgroup = subgroup_check_mode != PTAU_SUBGROUP_NO || check_output == PTAU_CHECK_ONLY_IN_GROUP ||
                         check_output == PTAU_CHECK_FULL;
    // SubgroupCheckMode::Batched, and Auto on more than 2^12 elements (setup-utils check_subgroup: BATCH_SIZE = 1 << 12,
    // SECURITY_PARAM = 128): the batched membership test, here through the buckets of the ratio check's MSM (msm_plan); it
    // needs that MSM, so without a ratio check — and for Direct, and for short vectors — the exact per-element test runs
    const bool fused = want_subgroup && ratio_seed && n > BATCHED_SUBGROUP_MIN && use_batched_subgroup() &&
                       (subgroup_check_mode == PTAU_SUBGROUP_AUTO || subgroup_check_mode == PTAU_SUBGROUP_BATCHED) &&
                       !cofactor_is_one(curve, group);
    if (fused) want_subgroup = false;
    // the membership kernels are queued behind the decompression at once (one error word collects both)
    if (want_subgroup && (rc = enqueue_subgroup_kernels(c, curve, group, ops, (const uint8_t *)d_out, 0, n, err))) return rc;
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    if (!prefetched) {
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
        c->timing.h2d_ms += ms;
    }
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    c->timing.kernel_ms += ms;
    c->busy_ms += ms;
    c->timing.decompress_ms += ms;
    if (want_subgroup) {
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->timing.kernel_ms += ms;
        c->busy_ms += ms;
        c->timing.subgroup_ms += ms;
    }
    c->timing.points += n;
    // This vector's MSM takes the tail slot that the MSM TWO vectors back used: that tail is collected now (it ran beside
    // the previous vector's kernels and the ones just waited for); the previous vector's tail, on the other slot, goes on.
    c->cur = &c->slots[c->slot_next];
    if ((rc = finish_msm_tail(c, err, want_subgroup ? &c->ev[1] : &c->ev[5]))) return rc;
    if ((rc = errbits_to_code(bits, err))) return rc;
    if (ratio_seed && n >= 2) {   // power_pairs on the vector while it is still resident in HBM; its tail is collected later
        const ptau::ScalarSource src{ptau::SCALAR_PRF, nullptr, ratio_seed[0], ratio_seed[1], index_base};
        const MsmFuse fz{curve, group};
        if ((rc = run_msm_dev(c, ops, (const uint8_t *)d_out, 0, n - 1, src, pair_a, pair_b, 0, err, /*defer_tail=*/true, fused ? &fz : nullptr))) return rc;
        if (use_two_tail_slots()) c->slot_next ^= 1;
    }
    // the decompressed vector goes home on the copy stream; nobody waits for it here
    return start_vector_copy(c, track, sel, d_out, out, n * ops->uncomp, ready, ready_mark, err);
}

// The single beta_g2 element of a chunk on the side stream: H2D, decompress (+ non-zero test), membership test, D2H of
// the uncompressed element.  Nothing is waited for here; finish_single_side() synchronises and reads the error word.
constexpr size_t SB_IN = 0, SB_OUT = 1024, SB_ORDER = 2048, SB_ERR = 2304, SB_SCRATCH = 2560;
int enqueue_single_side(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *in, uint8_t *out, int check_output,
                        bool want_subgroup, const Err &err) {
    uint8_t *b = c->s_buf;
    int *d_err = (int *)(b + SB_ERR);
    if (SB_SCRATCH + ops->jac_bytes + 8 * ops->wtab_entry_bytes > 16384) return err.fail(PTAU_E_DEVICE, "side buffer too small");
    HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), c->side_stream));
    HIP_TRY(hipMemcpyAsync(b + SB_IN, in, ops->comp, hipMemcpyHostToDevice, c->side_stream));
    const int nonzero = (check_output == PTAU_CHECK_ONLY_NON_ZERO || check_output == PTAU_CHECK_FULL) ? 1 : 0;
    ops->decompress(c->side_stream, b + SB_IN, b + SB_OUT, 1, nonzero, d_err);
    if (want_subgroup) {
        if (ops->endo_subgroup && use_endo_subgroup()) {
            ops->endo_subgroup(c->side_stream, b + SB_OUT, 0, 1, d_err);
        } else {
            ops->on_curve(c->side_stream, b + SB_OUT, 1, d_err);
            if (!cofactor_is_one(curve, group)) {
                const bool via_psi = ops->psi_check != nullptr && use_endo_subgroup();
                HIP_TRY(hipMemcpyAsync(b + SB_ORDER, via_psi ? ops->psi_scalar : ops->group_order, sizeof(uint32_t) * ops->fr_limbs32,
                                       hipMemcpyHostToDevice, c->side_stream));
                if (ops->subgroup_via_glv && use_endo_subgroup())
                    ops->window_scalar_mul(c->side_stream, b + SB_OUT, 0, b + SB_SCRATCH, b + SB_SCRATCH + ops->jac_bytes, 1,
                                           ptau::ScalarSource{ptau::SCALAR_SUBGROUP_ENDO, nullptr, 0, 0, 0}, 0, /*use_glv=*/1, d_err);
                else
                    ops->window_scalar_mul(c->side_stream, b + SB_OUT, 0, b + SB_SCRATCH, b + SB_SCRATCH + ops->jac_bytes, 1,
                                           ptau::ScalarSource{via_psi ? ptau::SCALAR_FIXED_PSI : ptau::SCALAR_FIXED, (const uint32_t *)(b + SB_ORDER), 0, 0, 0}, 0, /*use_glv=*/0, d_err);
                if (via_psi) ops->psi_check(c->side_stream, b + SB_OUT, 0, b + SB_SCRATCH, 1, d_err);
                else ops->expect_infinity(c->side_stream, b + SB_SCRATCH, 1, d_err);
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(out, b + SB_OUT, ops->uncomp, hipMemcpyDeviceToHost, c->side_stream));
    return PTAU_OK;
}
int finish_single_side(DeviceCtx *c, const Err &err) {
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->s_buf + SB_ERR, sizeof(int), hipMemcpyDeviceToHost, c->side_stream));
    HIP_TRY(hipStreamSynchronize(c->side_stream));
    HIP_TRY(hipGetLastError());
    return errbits_to_code(bits, err);
}

int check_params(const ptau_params *p, const Err &err) {
    if (!p) return err.fail(PTAU_E_ARG, "null parameters");
    if (p->curve < 0 || p->curve > 3) return err.fail(PTAU_E_ARG, "unknown curve id %d", p->curve);
    if (p->proving_system != PTAU_PROVING_GROTH16)
        return err.fail(PTAU_E_UNSUPPORTED, "only the groth16 accumulator geometry is implemented");
    if (p->contribution_mode != PTAU_MODE_CHUNKED && p->contribution_mode != PTAU_MODE_FULL)
        return err.fail(PTAU_E_ARG, "unknown contribution mode %d", p->contribution_mode);
    if (p->power > 40) return err.fail(PTAU_E_ARG, "power %u too large", p->power);
    if (p->contribution_mode == PTAU_MODE_CHUNKED && p->chunk_size == 0) return err.fail(PTAU_E_ARG, "chunk_size is zero");
    return PTAU_OK;
}

// BLAKE2b-512 (host CPU): blake2b_host.hpp / blake2b_host.cpp

struct MappedFile {
    void *p = MAP_FAILED;
    size_t len = 0;
    int fd = -1;
    std::future<void> populate;   // background page population / registration of an output mapping
    bool registered = false;
    // The mappings of a finished call are taken apart on a background thread: unregistering and unmapping 750 MB - 1.1 GB of
    // pinned file pages costs 45-50 ms, which nobody has to wait for once the output is synced (Reaper below).  A call that
    // fails, or PTAU_TRACE=1, releases in place.
    ~MappedFile() { release(); }
    void release() {
        if (populate.valid()) populate.wait();
        if (registered) (void)hipHostUnregister(p);
        if (p != MAP_FAILED) munmap(p, len);
        if (fd >= 0) close(fd);
        registered = false;
        p = MAP_FAILED;
        fd = -1;
    }
    struct Reaper {
        struct Item { void *p; size_t len; int fd; bool registered; int device; };
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Item> q;
        std::thread th;
        bool stop = false;
        void push(const Item &it) {
            std::lock_guard<std::mutex> lk(mu);
            if (!th.joinable()) th = std::thread([this] { run(); });
            q.push_back(it);
            cv.notify_one();
        }
        void run() {
            for (;;) {
                Item it;
                bool exiting;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    it = q.front();
                    q.pop_front();
                    exiting = stop;
                }
                // at process exit the HIP runtime may already be gone: the OS takes the pins back with the mapping
                if (it.registered && !exiting && hipSetDevice(it.device) == hipSuccess) (void)hipHostUnregister(it.p);
                munmap(it.p, it.len);
                if (it.fd >= 0) close(it.fd);
            }
        }
        ~Reaper() {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            if (th.joinable()) th.join();
        }
    };
    static Reaper &reaper() {
        static Reaper r;
        return r;
    }
    void release_async(int device) {
        if (populate.valid()) populate.wait();
        if (p != MAP_FAILED) reaper().push({p, len, fd, registered, device});
        else if (fd >= 0) close(fd);
        registered = false;
        p = MAP_FAILED;
        fd = -1;
    }
    static bool want_register() {
        static const bool w = [] {
            const char *e = getenv("PTAU_HOST_REGISTER");
            return !(e && e[0] == '0');
        }();
        return w;
    }
    // pin: the mapping feeds H2D copies of hundreds of MB — its pages are faulted in and registered with the HIP runtime
    // (read-only registration) on a host thread, so that the copies are DMA from the file's pages instead of a staged
    // pageable copy (12 GB/s and a busy host core inside the device section); wait_pinned() before the first copy.
    int open_ro(const char *path, const Err &err, bool pin = false, int device = 0) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return err.fail(PTAU_E_IO, "cannot open %s: %s", path, strerror(errno));
        struct stat st;
        if (fstat(fd, &st)) return err.fail(PTAU_E_IO, "cannot stat %s: %s", path, strerror(errno));
        len = (size_t)st.st_size;
        if (len) {
            p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) return err.fail(PTAU_E_IO, "cannot mmap %s: %s", path, strerror(errno));
            if (pin && want_register()) {
                populate = std::async(std::launch::async, [this, device] {
                    background_worker();
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
                    if (madvise(p, len, MADV_POPULATE_READ) != 0) return;
                    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }   // register in the caller's device context
                    if (hipHostRegister(p, len, hipHostRegisterPortable | hipHostRegisterReadOnly) == hipSuccess) registered = true;
                    else (void)hipGetLastError();
                });
            }
        }
        return PTAU_OK;
    }
    void wait_pinned() {
        if (populate.valid()) populate.wait();
    }
    int create_new(const char *path, size_t size, const Err &err, int device = 0) {
        fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0644);   // create_new semantics (contribute.rs:791-792)
        if (fd < 0) return err.fail(PTAU_E_IO, "cannot create %s: %s", path, strerror(errno));
        if (ftruncate(fd, (off_t)size)) return err.fail(PTAU_E_IO, "cannot size %s: %s", path, strerror(errno));
        len = size;
        p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) return err.fail(PTAU_E_IO, "cannot mmap %s: %s", path, strerror(errno));
        // Staging of the output file: the 250-600 MB of results arrive by D2H copies into this mapping;
        // into freshly created, never-touched pages every 4 KB of a copy takes a page fault in the driver's copy path
        // (80 ms of D2H for a 503 MB new challenge).  The pages are faulted in here, on a host thread, while the GPU is still
        // computing — MADV_POPULATE_WRITE does not modify their content, so it may run beside the copies — and the
        // populated range is then page-locked and registered with the HIP runtime (DMA straight into the file's pages,
        // no bounce buffer) unless PTAU_HOST_REGISTER=0.
        populate = std::async(std::launch::async, [this, device] {
            background_worker();
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
            // (one thread: four threads faulting slices of the same tmpfs file in were measured 2x SLOWER — 137-181 ms instead of
            // 82 ms for 503 MB — and the copies that ran meanwhile slower still; PTAU_TRACE=1 prints both marks)
            Trace tr{"    output mapping"};
            if (madvise(p, len, MADV_POPULATE_WRITE) != 0) return;
            tr.mark("pages populated");
            if (!want_register()) return;
            if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }   // register in the caller's device context
            if (hipHostRegister(p, len, hipHostRegisterPortable) == hipSuccess) registered = true;
            else (void)hipGetLastError();
            tr.mark("registered");
        });
        return PTAU_OK;
    }
};

int write_small_file(const char *path, const uint8_t *data, size_t n, const Err &err) {
    FILE *f = fopen(path, "wb");
    if (!f) return err.fail(PTAU_E_IO, "cannot create %s: %s", path, strerror(errno));
    size_t w = fwrite(data, 1, n, f);
    if (fclose(f) || w != n) return err.fail(PTAU_E_IO, "short write to %s", path);
    return PTAU_OK;
}

}  // namespace

extern "C" {

const char *ptau_strerror(int code) {
    switch (code) {
    case PTAU_OK: return "ok";
    case PTAU_E_ARG: return "invalid argument";
    case PTAU_E_LEN: return "buffer length does not match the parameters";
    case PTAU_E_POINT: return "invalid group element encoding";
    case PTAU_E_POINT_AT_INFINITY: return "point at infinity";
    case PTAU_E_NOT_IN_SUBGROUP: return "element not in the prime-order subgroup";
    case PTAU_E_INVALID_RATIO: return "same-ratio check failed";
    case PTAU_E_HASH_CHAIN: return "hash chain mismatch";
    case PTAU_E_DEVICE: return "GPU device error";
    case PTAU_E_IO: return "file I/O error";
    case PTAU_E_UNSUPPORTED: return "unsupported configuration";
    }
    return "unknown error";
}

int ptau_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ptau_device_info(int device, char *name, size_t name_len, int *compute_units, uint64_t *hbm_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PTAU_E_DEVICE;
    if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return PTAU_OK;
}

int ptau_point_size(int curve, int group, int compressed) {
    const GroupOps *o = get_ops(curve, group);
    if (!o) return PTAU_E_ARG;
    return compressed ? o->comp : o->uncomp;
}

int ptau_fr_limbs(int curve) {
    const GroupOps *o = get_ops(curve, 1);
    return o ? (o->fr_limbs32 + 1) / 2 : PTAU_E_ARG;
}

int ptau_params_sizes(const ptau_params *p, ptau_sizes *out) {
    Err err{nullptr, 0};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!out) return PTAU_E_ARG;
    const GroupOps *g1 = get_ops(p->curve, 1), *g2 = get_ops(p->curve, 2);
    const uint64_t L = 1ull << p->power, G1L = 2 * L - 1;
    const bool full = p->contribution_mode == PTAU_MODE_FULL;
    const uint64_t start = full ? 0 : p->chunk_index * p->chunk_size;
    auto clampc = [&](uint64_t total) -> uint64_t {
        if (full) return total;
        if (start >= total) return 0;
        return std::min(p->chunk_size, total - start);
    };
    out->powers_length = L;
    out->powers_g1_length = G1L;
    out->g1_chunk_size = clampc(G1L);
    out->other_chunk_size = clampc(L);
    out->hash_size = 64;
    out->start = start;
    const uint64_t n1 = out->g1_chunk_size, n2 = out->other_chunk_size;
    out->public_key_size = 6ull * g1->uncomp + 3ull * g2->uncomp;
    out->accumulator_size = 64 + n1 * g1->uncomp + n2 * (g2->uncomp + 2ull * g1->uncomp) + g2->uncomp;
    out->contribution_size = 64 + n1 * g1->comp + n2 * (g2->comp + 2ull * g1->comp) + g2->comp + out->public_key_size;
    return PTAU_OK;
}

double ptau_device_busy_ms(int device) {
    if (device < 0 || device >= MAX_DEVICES) return -1.0;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    return g_ctx[device].busy_ms + g_side_ctx[device].busy_ms;
}

int ptau_last_timing(int device, ptau_timing *out) {
    if (device < 0 || device >= MAX_DEVICES || !out) return PTAU_E_ARG;
    *out = tl_timing[device];   // the calling thread's most recent call on this device (CallTiming)
    return PTAU_OK;
}

int ptau_apply_powers_dev(int curve, int group, const void *d_in, int compressed_in, void *d_out, int compressed_out,
                          uint64_t n, uint64_t start, const uint64_t *tau, const uint64_t *coeff, int check_input,
                          int batch_exp_mode, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (!tau || (n && (!d_in || !d_out))) return err.fail(PTAU_E_ARG, "null pointer argument");
    const int use_endo = endo_allowed(check_input, batch_exp_mode);
    if (use_endo < 0) return err.fail(PTAU_E_ARG, "unknown batch_exp_mode %d", batch_exp_mode);
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    return run_apply_powers_dev(c, curve, group, ops, (const uint8_t *)d_in, compressed_in, (uint8_t *)d_out, compressed_out, n, start, tau,
                                coeff, check_input, use_endo, err);
}

int ptau_apply_powers(int curve, int group, const uint8_t *in, int compressed_in, uint8_t *out, int compressed_out, uint64_t n,
                      uint64_t start, const uint64_t *tau, const uint64_t *coeff, int check_input, int batch_exp_mode, int device,
                      char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (!tau || (n && (!in || !out))) return err.fail(PTAU_E_ARG, "null pointer argument");
    const int use_endo = endo_allowed(check_input, batch_exp_mode);
    if (use_endo < 0) return err.fail(PTAU_E_ARG, "unknown batch_exp_mode %d", batch_exp_mode);
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    return run_apply_powers_host(c, curve, group, ops, in, compressed_in, out, compressed_out, n, start, tau, coeff, check_input, use_endo, err);
}

int ptau_phase1_computation(const ptau_params *p, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                            const uint64_t *tau, const uint64_t *alpha, const uint64_t *beta, int check_input,
                            int batch_exp_mode, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!in || !out || !tau || !alpha || !beta) return err.fail(PTAU_E_ARG, "null pointer argument");
    const int use_endo = endo_allowed(check_input, batch_exp_mode);
    if (use_endo < 0) return err.fail(PTAU_E_ARG, "unknown batch_exp_mode %d", batch_exp_mode);
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    if (in_len != sz.accumulator_size)
        return err.fail(PTAU_E_LEN, "challenge is %zu bytes, parameters require %llu", in_len,
                        (unsigned long long)sz.accumulator_size);
    if (out_len != sz.contribution_size)
        return err.fail(PTAU_E_LEN, "response buffer is %zu bytes, parameters require %llu", out_len,
                        (unsigned long long)sz.contribution_size);
    const GroupOps *g1 = get_ops(p->curve, 1), *g2 = get_ops(p->curve, 2);
    DeviceCtx *c;
    if ((rc = ensure_ctx(device, c, err))) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    const uint64_t n1 = sz.g1_chunk_size, n2 = sz.other_chunk_size, s = sz.start;
    const uint8_t *pi = in + 64;
    uint8_t *po = out + 64;
    // tauG1 | tauG2 | alphaG1 | betaG1 | betaG2  (SURVEY App. A.1); exponent index is global: start + i
    // betaG2 <- beta * betaG2 (tau^0 with coefficient beta): the last element of both buffers, started first on the side
    // context by a host thread, joined after the four vectors
    DeviceCtx *sc;
    if ((rc = ensure_ctx(device, sc, err, /*side=*/true))) return rc;
    sc->timing = ptau_timing{};
    c->pending.bytes = 0;
    sc->pending.bytes = 0;
    char side_msg[256] = {0};
    const uint8_t *pi_b = in + sz.accumulator_size - g2->uncomp;
    uint8_t *po_b = out + sz.contribution_size - sz.public_key_size - g2->comp;
    const int curve_id = p->curve;
    auto side = std::async(std::launch::async, [=, &side_msg] {
        if (hipSetDevice(device) != hipSuccess) return (int)PTAU_E_DEVICE;
        return run_apply_powers_host(sc, curve_id, 2, g2, pi_b, 0, po_b, 1, 1, 0, tau, beta, check_input, use_endo, Err{side_msg, sizeof(side_msg)});
    });
    struct SideJoin {   // the side thread never outlives the caller's buffers
        std::future<int> &f;
        ~SideJoin() { if (f.valid()) f.wait(); }
    } side_join{side};
    c->prefetch = DeviceCtx::Prefetch{};
    // every call names the NEXT vector's input: it is copied to the device while this vector's kernels run
    const uint8_t *in_tau_g2 = pi + n1 * g1->uncomp, *in_alpha = in_tau_g2 + n2 * g2->uncomp, *in_beta = in_alpha + n2 * g1->uncomp;
    if ((rc = run_apply_powers_host(c, p->curve, 1, g1, pi, 0, po, 1, n1, s, tau, nullptr, check_input, use_endo, err, /*defer_copy=*/true,
                                    n2 ? in_tau_g2 : nullptr, n2 * g2->uncomp))) return rc;
    pi += n1 * g1->uncomp; po += n1 * g1->comp;
    if ((rc = run_apply_powers_host(c, p->curve, 2, g2, pi, 0, po, 1, n2, s, tau, nullptr, check_input, use_endo, err, /*defer_copy=*/true,
                                    in_alpha, n2 * g1->uncomp))) return rc;
    pi += n2 * g2->uncomp; po += n2 * g2->comp;
    if ((rc = run_apply_powers_host(c, p->curve, 1, g1, pi, 0, po, 1, n2, s, tau, alpha, check_input, use_endo, err, /*defer_copy=*/true,
                                    in_beta, n2 * g1->uncomp))) return rc;
    pi += n2 * g1->uncomp; po += n2 * g1->comp;
    if ((rc = run_apply_powers_host(c, p->curve, 1, g1, pi, 0, po, 1, n2, s, tau, beta, check_input, use_endo, err, /*defer_copy=*/true))) return rc;
    pi += n2 * g1->uncomp; po += n2 * g1->comp;
    if ((rc = flush_pending_copy(c, err))) return rc;   // the last vector's deferred copy
    rc = side.get();
    c->timing.points += 1;
    if (rc) return err.fail(rc, "%s", side_msg);
    return PTAU_OK;
}

int ptau_contribute_file(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                         const char *response_path, const char *response_hash_path, const uint64_t *tau,
                         const uint64_t *alpha, const uint64_t *beta, const uint8_t *public_key, size_t public_key_len,
                         int check_input, int batch_exp_mode, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    if (endo_allowed(check_input, batch_exp_mode) < 0) return err.fail(PTAU_E_ARG, "unknown batch_exp_mode %d", batch_exp_mode);
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge_path || !challenge_hash_path || !response_path || !response_hash_path || !public_key)
        return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    if (public_key_len != sz.public_key_size)
        return err.fail(PTAU_E_LEN, "public key is %zu bytes, expected %llu", public_key_len,
                        (unsigned long long)sz.public_key_size);
    MappedFile ch, resp;
    if ((rc = ch.open_ro(challenge_path, err, /*pin=*/true, device))) return rc;
    if (ch.len != sz.accumulator_size)
        return err.fail(PTAU_E_LEN, "challenge file is %zu bytes, parameters require %llu", ch.len,
                        (unsigned long long)sz.accumulator_size);
    uint8_t digest[64];
    if ((rc = resp.create_new(response_path, sz.contribution_size, err, device))) return rc;
    uint8_t *out = (uint8_t *)resp.p;
    {   // BLAKE2b(challenge) on a host thread while the GPU updates the elements
        const uint8_t *chp = (const uint8_t *)ch.p;
        const size_t chl = ch.len;
        auto fut = std::async(std::launch::async, [chp, chl, &digest] { ptau_blake2b512(chp, chl, digest); });   // (critical path: not a background worker)
        ch.wait_pinned();
        rc = ptau_phase1_computation(p, chp, chl, out, resp.len, tau, alpha, beta, check_input, batch_exp_mode, device, errbuf, err_len);
        fut.get();
        if (rc) return rc;
    }
    if ((rc = write_small_file(challenge_hash_path, digest, 64, err))) return rc;
    memcpy(out, digest, 64);
    memcpy(out + sz.contribution_size - sz.public_key_size, public_key, public_key_len);
    ptau_blake2b512(out, resp.len, digest);
    if (msync(resp.p, resp.len, MS_SYNC)) return err.fail(PTAU_E_IO, "msync failed: %s", strerror(errno));
    rc = write_small_file(response_hash_path, digest, 64, err);
    resp.release_async(device);
    ch.release_async(device);
    return rc;
}

int ptau_contribute_file_seeded(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                                const char *response_path, const char *response_hash_path, const uint8_t *seed, size_t seed_len,
                                int check_input, int batch_exp_mode, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge_path || !challenge_hash_path || !response_path || !response_hash_path || !seed)
        return err.fail(PTAU_E_ARG, "null pointer argument");
    if (endo_allowed(check_input, batch_exp_mode) < 0) return err.fail(PTAU_E_ARG, "unknown batch_exp_mode %d", batch_exp_mode);
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    // PrivateKey first: tau, alpha, beta are the generator's first draws and do not depend on the challenge digest
    uint64_t tau[12], alpha[12], beta[12];
    if ((rc = ptau_private_key_from_seed(p->curve, seed, seed_len, tau, alpha, beta, errbuf, err_len))) return rc;
    MappedFile ch, resp;
    if ((rc = ch.open_ro(challenge_path, err, /*pin=*/true, device))) return rc;
    if (ch.len != sz.accumulator_size)
        return err.fail(PTAU_E_LEN, "challenge file is %zu bytes, parameters require %llu", ch.len,
                        (unsigned long long)sz.accumulator_size);
    if ((rc = resp.create_new(response_path, sz.contribution_size, err, device))) return rc;
    uint8_t *out = (uint8_t *)resp.p;
    uint8_t digest[64];
    std::vector<uint8_t> pk(sz.public_key_size);
    {   // host thread: BLAKE2b(challenge), then the PublicKey bound to it — both while the GPU updates the elements
        const uint8_t *chp = (const uint8_t *)ch.p;
        const size_t chl = ch.len;
        const int curve = p->curve;
        char kmsg[256] = {0};
        auto fut = std::async(std::launch::async, [&, chp, chl, curve] {   // (critical path: not a background worker)
            ptau_blake2b512(chp, chl, digest);
            uint64_t t2[12], a2[12], b2[12];
            return ptau_key_generation(curve, seed, seed_len, digest, t2, a2, b2, pk.data(), pk.size(), kmsg, sizeof(kmsg));
        });
        ch.wait_pinned();
        rc = ptau_phase1_computation(p, chp, chl, out, resp.len, tau, alpha, beta, check_input, batch_exp_mode, device, errbuf, err_len);
        const int krc = fut.get();
        if (rc) return rc;
        if (krc) return err.fail(krc, "key generation: %s", kmsg);
    }
    if ((rc = write_small_file(challenge_hash_path, digest, 64, err))) return rc;
    memcpy(out, digest, 64);
    memcpy(out + sz.contribution_size - sz.public_key_size, pk.data(), pk.size());
    ptau_blake2b512(out, resp.len, digest);
    if (msync(resp.p, resp.len, MS_SYNC)) return err.fail(PTAU_E_IO, "msync failed: %s", strerror(errno));
    rc = write_small_file(response_hash_path, digest, 64, err);
    resp.release_async(device);
    ch.release_async(device);
    return rc;
}

int ptau_decompress(int curve, int group, const uint8_t *in, uint8_t *out, uint64_t n, int check_output, int device,
                    char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (n && (!in || !out)) return err.fail(PTAU_E_ARG, "null pointer argument");
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    if (!n) return PTAU_OK;
    if ((rc = grow(c->d_in, c->in_cap, n * ops->comp, err))) return rc;
    void *d_out_buf = nullptr;
    if ((rc = claim_out_buffer(c, 0, n * ops->uncomp, d_out_buf, err))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, in, n * ops->comp, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    ops->decompress(c->stream, (const uint8_t *)c->d_in, (uint8_t *)d_out_buf, n, (check_output & 1), c->d_err);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out_buf, n * ops->uncomp, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->timing.kernel_ms = ms;
    c->timing.points = n;
    return errbits_to_code(bits, err);
}

int ptau_subgroup_check(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, int device, char *errbuf,
                        size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (n && !in) return err.fail(PTAU_E_ARG, "null pointer argument");
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    if (!n) return PTAU_OK;
    const size_t isz = compressed_in ? ops->comp : ops->uncomp;
    if ((rc = grow(c->d_in, c->in_cap, n * isz, err))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, in, n * isz, hipMemcpyHostToDevice, c->stream));
    return run_subgroup_check_dev(c, curve, group, ops, (const uint8_t *)c->d_in, compressed_in, n, err);
}

}  // extern "C"

namespace {
// The three digests of a chunk verification (the `.hash` side files of transform_pok_and_correctness).  Each file is
// hashed exactly ONCE, on its own host thread, while the GPU works: BLAKE2b(challenge) and BLAKE2b(response) start
// at once; BLAKE2b(new_challenge) trails the device, consuming each vector as soon as its D2H copy has landed.
struct VerifyDigests {
    uint8_t challenge[64], response[64], new_challenge[64];
};

// bit 6 of the last byte of an uncompressed point = arkworks' infinity flag
__global__ void __launch_bounds__(256) infinity_flag_kernel(const uint8_t *__restrict__ in, uint64_t n, uint32_t stride, int *__restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (in[(i + 1) * stride - 1] & 0x40u) atomicOr(err, ptau::ERRBIT_INFINITY);
}

// CheckForCorrectness of one CHALLENGE vector (uncompressed host bytes): bit 0 = no point at infinity, bit 1 = on the
// curve and in the prime-order subgroup.  check_input of transform_pok_and_correctness (--force-correctness-checks,
// /root/reference/src/bin/contribute.rs:133, 971-974).  Caller holds the device mutex.
int run_check_input_vector(DeviceCtx *c, int curve, int group, const GroupOps *ops, const uint8_t *in, uint64_t n, int check_input,
                           const Err &err) {
    if (!n || !(check_input & 3)) return PTAU_OK;
    int rc;
    if ((rc = grow(c->d_in, c->in_cap, n * ops->uncomp, err))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, in, n * ops->uncomp, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    if (check_input & 1)
        hipLaunchKernelGGL(infinity_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)c->d_in, n,
                           (uint32_t)ops->uncomp, c->d_err);
    if ((check_input & 2) && (rc = enqueue_subgroup_kernels(c, curve, group, ops, (const uint8_t *)c->d_in, 0, n, err))) return rc;
    int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    if (check_input & 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->timing.kernel_ms += ms;
        c->busy_ms += ms;
        c->timing.subgroup_ms += ms;
    }
    c->timing.points += n;
    return errbits_to_code(bits, err);
}

// One same-ratio job of the host pairing tail; run_ratio_jobs runs them on one host thread each and reports the first
// failure in job order (so the error a caller sees does not depend on thread timing).
struct RatioJob {
    const uint8_t *a, *b, *c2, *d2;   // (a, b) in G1, (c2, d2) in G2, all uncompressed
    std::string what;                  // message when the ratios differ
};
// device >= 0: ONE launch of the pairing kernel, a workgroup per job (ptau_check_same_ratio_batch); device < 0: the host
// tail, one host thread per job (the standalone ptau_verify_pok / ptau_check_same_ratio, which need no GPU).
int run_ratio_jobs(int curve, const std::vector<RatioJob> &jobs, const Err &err, int device = -1) {
    if (jobs.empty()) return PTAU_OK;
    std::vector<int> res(jobs.size(), 0);
    if (device >= 0) {
        const ptau::PairingOps *po = get_pairing_ops(curve);
        const GroupOps *g1 = get_ops(curve, 1), *g2 = get_ops(curve, 2);
        std::vector<uint8_t> packed(jobs.size() * po->check_bytes);
        for (size_t i = 0; i < jobs.size(); i++) {
            uint8_t *q = packed.data() + i * po->check_bytes;
            memcpy(q, jobs[i].a, g1->uncomp);
            memcpy(q + g1->uncomp, jobs[i].b, g1->uncomp);
            memcpy(q + 2 * (size_t)g1->uncomp, jobs[i].c2, g2->uncomp);
            memcpy(q + 2 * (size_t)g1->uncomp + g2->uncomp, jobs[i].d2, g2->uncomp);
        }
        int rc = ptau_check_same_ratio_batch(curve, packed.data(), jobs.size(), res.data(), nullptr, device, err.buf, err.len);
        if (rc) return rc;
    } else {
        std::vector<std::future<int>> futs;
        for (auto &j : jobs)
            futs.push_back(std::async(std::launch::async, [curve, &j] {
                const int r = ptau_check_same_ratio(curve, j.a, j.b, j.c2, j.d2, nullptr, 0);
                return r == PTAU_OK ? 1 : (r == PTAU_E_INVALID_RATIO ? 0 : -1);
            }));
        for (size_t i = 0; i < futs.size(); i++) res[i] = futs[i].get();
    }
    for (size_t i = 0; i < jobs.size(); i++) {   // the first failure in job order, whatever the execution order was
        if (res[i] == 0) return err.fail(PTAU_E_INVALID_RATIO, "%s", jobs[i].what.c_str());
        if (res[i] < 0) return err.fail(PTAU_E_POINT, "malformed point in the same-ratio check: %s", jobs[i].what.c_str());
    }
    return PTAU_OK;
}

// Phase1::verification steps (1)-(2): the jobs of the three proofs of knowledge and, for chunk 0 / Full mode, of the
// before/after checks.  `scratch` keeps the few decompressed response elements alive until the jobs have run.
int collect_pok_jobs(const ptau_params *p, const ptau_sizes &sz, const uint8_t *challenge, const uint8_t *response,
                     const uint8_t *g2_s, bool g2_s_from_caller, std::vector<uint8_t> &scratch, std::vector<RatioJob> &jobs,
                     const Err &err) {
    const GroupOps *g1 = get_ops(p->curve, 1), *g2 = get_ops(p->curve, 2);
    const size_t u1 = g1->uncomp, u2 = g2->uncomp;
    // PublicKey (uncompressed): tau_g1.0, tau_g1.1, alpha_g1.0, alpha_g1.1, beta_g1.0, beta_g1.1, tau_g2, alpha_g2, beta_g2
    const uint8_t *pk = response + sz.contribution_size - sz.public_key_size;
    {   // The PublicKey is the contributor's bytes and goes straight into pairings: canonical, ON THE CURVE and IN THE
        // PRIME-ORDER SUBGROUP first (arkworks' checked deserialisation of the key does the same; a Miller loop on a point
        // of another curve is not a pairing identity).  A g2_s derived by the library is a cofactor-cleared hash-to-curve
        // output; one supplied by the caller is checked like the key.
        int rc = ptau::host_validate_points(p->curve, 1, pk, 6);
        if (!rc) rc = ptau::host_validate_points(p->curve, 2, pk + 6 * u1, 3);
        if (rc) return err.fail(rc, rc == PTAU_E_NOT_IN_SUBGROUP ? "a public-key point is not in the prime-order subgroup"
                                                                  : "a public-key point is not a valid curve point");
        if (g2_s_from_caller && (rc = ptau::host_validate_points(p->curve, 2, g2_s, 3)))
            return err.fail(rc, "a caller-supplied g2_s point is not a valid element of the prime-order subgroup");
    }
    const uint8_t *pk_g1[3][2], *pk_g2[3];
    for (int x = 0; x < 3; x++) {
        pk_g1[x][0] = pk + (2 * x) * u1;
        pk_g1[x][1] = pk + (2 * x + 1) * u1;
        pk_g2[x] = pk + 6 * u1 + x * u2;
    }
    static const char *names[3] = {"tau", "alpha", "beta"};
    // (1) proofs of knowledge: (g1_s, x*g1_s) ~ (g2_s, x*g2_s)
    for (int x = 0; x < 3; x++)
        jobs.push_back({pk_g1[x][0], pk_g1[x][1], g2_s + x * u2, pk_g2[x], std::string("proof of knowledge of ") + names[x] + " is invalid"});
    if (!(p->contribution_mode == PTAU_MODE_FULL || p->chunk_index == 0)) return PTAU_OK;
    // (2) chunk 0: the response extends the challenge by exactly the key in the public key.
    //     Needs a handful of response elements decompressed: done on the host (sqrt of ~5 points).
    const uint64_t n1 = sz.g1_chunk_size, n2 = sz.other_chunk_size;
    if (n1 < 2 || n2 < 2) return err.fail(PTAU_E_ARG, "chunk 0 too small for the before/after checks");
    const uint8_t *b_tau_g1 = challenge + 64, *b_tau_g2 = b_tau_g1 + n1 * u1, *b_alpha = b_tau_g2 + n2 * u2,
                  *b_beta = b_alpha + n2 * u1, *b_beta_g2 = b_beta + n2 * u1;
    const size_t c1 = g1->comp, c2 = g2->comp;
    const uint8_t *a_tau_g1 = response + 64, *a_tau_g2 = a_tau_g1 + n1 * c1, *a_alpha = a_tau_g2 + n2 * c2,
                  *a_beta = a_alpha + n2 * c1, *a_beta_g2 = a_beta + n2 * c1;
    scratch.assign(2 * u1 + u1 + u1 + 2 * u2 + u2, 0);
    uint8_t *da_tau_g1 = scratch.data(), *da_alpha = da_tau_g1 + 2 * u1, *da_beta = da_alpha + u1, *da_tau_g2 = da_beta + u1,
            *da_beta_g2 = da_tau_g2 + 2 * u2;
    {   // five elements: host-side sqrt (the pairing tail needs no GPU)
        int bad = g1->decompress_host(a_tau_g1, da_tau_g1, 2, 1) | g1->decompress_host(a_alpha, da_alpha, 1, 1) |
                  g1->decompress_host(a_beta, da_beta, 1, 1) | g2->decompress_host(a_tau_g2, da_tau_g2, 2, 1) |
                  g2->decompress_host(a_beta_g2, da_beta_g2, 1, 1);
        int rc;
        if ((rc = errbits_to_code(bad, err))) return rc;
    }
    // element 0 (tau^0) must be untouched
    if (memcmp(da_tau_g1, b_tau_g1, u1) != 0 || memcmp(da_tau_g2, b_tau_g2, u2) != 0)
        return err.fail(PTAU_E_INVALID_RATIO, "element 0 of tau_g1 / tau_g2 changed");
    struct { const uint8_t *before, *after; int key; bool g1; const char *what; } chk[5] = {
        {b_tau_g1 + u1, da_tau_g1 + u1, 0, true, "tau_g1[1]"}, {b_alpha, da_alpha, 1, true, "alpha_g1[0]"},
        {b_beta, da_beta, 2, true, "beta_g1[0]"},             {b_tau_g2 + u2, da_tau_g2 + u2, 0, false, "tau_g2[1]"},
        {b_beta_g2, da_beta_g2, 2, false, "beta_g2"}};
    for (auto &k : chk) {
        const std::string what = std::string(k.what) + " was not multiplied by the key in the public key";
        if (k.g1) jobs.push_back({k.before, k.after, g2_s + k.key * u2, pk_g2[k.key], what});
        else jobs.push_back({pk_g1[k.key][0], pk_g1[k.key][1], k.before, k.after, what});
    }
    return PTAU_OK;
}

// verify_pok: 0 = element checks only; 1 = Phase1::verification in full (proofs of knowledge + chunk-0 checks), with
// g2_s = the caller's three compute_g2_s points or, when NULL, derived here from BLAKE2b(challenge) and the public key.
int verify_chunk_impl(const ptau_params *p, const uint8_t *challenge, size_t challenge_len, const uint8_t *response,
                      size_t response_len, uint8_t *new_challenge, size_t new_challenge_len, int check_input,
                      int check_output, int subgroup_check_mode, int ratio_check, int verify_pok, const uint8_t *g2_s,
                      const uint8_t *tau_g2_check, int device, VerifyDigests *digests, char *errbuf, size_t err_len,
                      const std::function<void()> *before_device = nullptr) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge || !response || !new_challenge) return err.fail(PTAU_E_ARG, "null pointer argument");
    if (check_input < 0 || check_input > 3 || check_output < 0 || check_output > 3)
        return err.fail(PTAU_E_ARG, "unknown CheckForCorrectness value %d / %d", check_input, check_output);
    if (subgroup_check_mode < 0 || subgroup_check_mode > 3) return err.fail(PTAU_E_ARG, "unknown subgroup check mode %d", subgroup_check_mode);
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    if (challenge_len != sz.accumulator_size || new_challenge_len != sz.accumulator_size)
        return err.fail(PTAU_E_LEN, "challenge / new challenge must be %llu bytes", (unsigned long long)sz.accumulator_size);
    if (response_len != sz.contribution_size)
        return err.fail(PTAU_E_LEN, "response is %zu bytes, parameters require %llu", response_len,
                        (unsigned long long)sz.contribution_size);
    // hash chain: response[0..64) must be BLAKE2b(challenge); new_challenge starts with BLAKE2b(response).
    // Both digests run on host threads WHILE the GPU checks the elements (BLAKE2b is sequential, ~1 GB/s/core,
    // and would otherwise cost as much as the device work — SURVEY §7 hard part 7).
    uint8_t chal_digest[64];
    auto fut_chain = std::async(std::launch::async, [=, &chal_digest] {
        background_worker();
        ptau_blake2b512(challenge, challenge_len, chal_digest);
        if (digests) memcpy(digests->challenge, chal_digest, 64);
        return memcmp(chal_digest, response, 64) == 0;
    });
    std::shared_future<void> fut_resp = std::async(std::launch::async, [=] {
        background_worker();
        ptau_blake2b512(response, response_len, new_challenge);
        if (digests) memcpy(digests->response, new_challenge, 64);
    }).share();
    std::atomic<size_t> nc_ready{64};     // bytes of new_challenge final so far (the header counts once fut_resp is done)
    std::atomic<bool> nc_abort{false};
    std::future<void> fut_nc;
    if (digests)
        fut_nc = std::async(std::launch::async, [&, fut_resp] {
            background_worker();
            fut_resp.wait();
            Blake2b b;
            size_t done = 0;
            while (done < new_challenge_len) {
                if (nc_abort.load(std::memory_order_acquire)) return;
                const size_t r = nc_ready.load(std::memory_order_acquire);
                if (r > done) {
                    b.update(new_challenge + done, r - done);
                    done = r;
                } else {
                    usleep(200);
                }
            }
            b.finish(digests->new_challenge);
        });
    struct Joiner {   // never leave a worker running past the caller's buffers
        std::future<bool> &a;
        std::shared_future<void> &b;
        std::future<void> &c;
        std::atomic<bool> &abort_flag;
        bool ok = false;
        ~Joiner() {
            if (!ok) abort_flag.store(true, std::memory_order_release);
            if (a.valid()) a.wait();
            if (b.valid()) b.wait();
            if (c.valid()) c.wait();
        }
    } joiner{fut_chain, fut_resp, fut_nc, nc_abort};
    CopyTrack track;   // (after the joiner: it drains first, while nc_ready and the caller's buffers are certainly alive)
    const GroupOps *g1 = get_ops(p->curve, 1), *g2 = get_ops(p->curve, 2);
    DeviceCtx *c;
    if ((rc = ensure_ctx(device, c, err))) return rc;
    Trace tr{"  verify_chunk"};
    // the digests are running; whatever the caller still has to wait for before the device can be used well (file-level
    // call: the output mapping's page registration, ~115 ms for 503 MB) is waited for HERE, outside the device lock.  It costs
    // no latency: BLAKE2b(new challenge) cannot start before BLAKE2b(response) has produced its header (~180 ms).
    if (before_device) (*before_device)();
    std::unique_lock<std::mutex> lk(c->mu);   // held for the device section only: pairings and digests below run unlocked
    CallTiming call_timing{c, device};
    tr.mark("device lock taken");
    if ((rc = begin_device_section(c, device, err))) return rc;
    const uint64_t n1 = sz.g1_chunk_size, n2 = sz.other_chunk_size;
    const uint8_t *pi = response + 64;
    uint8_t *po = new_challenge + 64;
    struct { const GroupOps *ops; int group; uint64_t n; const char *name; } vec[5] = {
        {g1, 1, n1, "tau_g1"}, {g2, 2, n2, "tau_g2"}, {g1, 1, n2, "alpha_g1"}, {g1, 1, n2, "beta_g1"}, {g2, 2, 1, "beta_g2"}};
    if (check_input & 3) {   // CheckForCorrectness of the challenge elements (DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS is No)
        const uint8_t *pc = challenge + 64;
        for (auto &v : vec) {
            if ((rc = run_check_input_vector(c, p->curve, v.group, v.ops, pc, v.n, check_input, err))) return rc;
            pc += v.n * v.ops->uncomp;
        }
    }
    uint64_t seed[2] = {0, 0};
    if (ratio_check) {   // fresh secret for the random linear combinations of this verification
        FILE *f = fopen("/dev/urandom", "rb");
        if (!f || fread(seed, 1, 16, f) != 16) {
            if (f) fclose(f);
            return err.fail(PTAU_E_IO, "cannot read /dev/urandom for the random linear combination");
        }
        fclose(f);
    }
    std::vector<uint8_t> pairs(5 * 2 * (size_t)g2->uncomp);   // (A, B) per vector
    uint8_t *vec_out[5];
    {   // beta_g2 (one element, the last of the buffer) goes first, on the side stream, beside the four vectors
        const uint64_t off_c = n1 * g1->comp + n2 * (g2->comp + 2ull * g1->comp), off_u = n1 * g1->uncomp + n2 * (g2->uncomp + 2ull * g1->uncomp);
        const bool want_sg = subgroup_check_mode != PTAU_SUBGROUP_NO || check_output == PTAU_CHECK_ONLY_IN_GROUP || check_output == PTAU_CHECK_FULL;
        if ((rc = enqueue_single_side(c, p->curve, 2, g2, pi + off_c, po + off_u, check_output, want_sg, err))) return rc;
    }
    for (int k = 0; k < 4; k++) {
        auto &v = vec[k];
        vec_out[k] = po;
        uint8_t *pa = pairs.data() + (size_t)k * 2 * g2->uncomp, *pb = pa + g2->uncomp;
        const uint8_t *nin = k < 3 && vec[k + 1].n ? pi + v.n * v.ops->comp : nullptr;
        if ((rc = run_verify_vector(c, p->curve, v.group, v.ops, pi, po, v.n, check_output, subgroup_check_mode,
                                    ratio_check ? seed : nullptr, sz.start, pa, pb, &nc_ready,
                                    (size_t)(po + v.n * v.ops->uncomp - new_challenge), &track, err, nin,
                                    nin ? vec[k + 1].n * vec[k + 1].ops->comp : 0))) {
            if (c->prefetch.host) (void)hipStreamSynchronize(c->copy_stream);   // no copy from the caller's buffer stays in flight
            c->prefetch = DeviceCtx::Prefetch{};
            hipStreamSynchronize(c->side_stream);   // never leave device work running past the caller's buffers
            return rc;
        }
        pi += v.n * v.ops->comp;
        po += v.n * v.ops->uncomp;
    }
    {   // the MSM tails still out: the older one (it also had the last vector's kernels to hide behind), then the last vector's
        // (tails follow each other on the one tail stream: what runs past the main stream's last kernel is counted once, with
        // the newest; the older one ended before it)
        DeviceCtx::TailSlot *older = &c->slots[c->slot_next], *newest = &c->slots[c->slot_next ^ 1];
        if (newest->tail.active) HIP_TRY(hipEventSynchronize(newest->tev[2]));
        c->cur = older;
        rc = finish_msm_tail(c, err, newest->tail.active ? &newest->tev[2] : nullptr);
        c->cur = newest;
        if (!rc) rc = finish_msm_tail(c, err, nullptr);
        if (rc) {
            hipStreamSynchronize(c->side_stream);
            return rc;
        }
    }
    vec_out[4] = po;
    if ((rc = finish_single_side(c, err))) return rc;
    po += g2->uncomp;
    call_timing.detach();   // c->timing is the next lane's from here on; this call's record goes on in its private copy
    lk.unlock();   // another lane may use the device while this one finishes its host tail
    tr.mark("device section done");
    track.drain();   // the vectors' copies (queued, never waited for inside the device section) have landed
    nc_ready.store((size_t)(po - new_challenge), std::memory_order_release);
    tr.mark("copies landed");
    // ---- host tail: every same-ratio check of this verification, one host thread each ----
    std::vector<RatioJob> ratio_jobs, pok_jobs;
    if (ratio_check) {
        // check_power_ratios: consecutive elements of every vector differ by the SAME factor tau, i.e.
        // (sum rho_i v_i, sum rho_i v_{i+1}) has the ratio of (tauG2[j], tauG2[j+1]) resp. (tauG1[j], tauG1[j+1]).
        // A chunk without tauG2 elements (start >= 2^power) holds no G2 pair of ratio tau in its own files: its tauG1
        // power ratios are then checked only when the caller supplies tau_g2_check, and are otherwise left to the
        // full-length check of transform_ratios on the combined accumulator (control.rs:587-591) — an honest response
        // never fails here for lack of it.
        const uint8_t *g2_pair = n2 >= 2 ? vec_out[1] : tau_g2_check, *g1_pair = n1 >= 2 ? vec_out[0] : nullptr;
        if (g2_pair)
            for (int k : {0, 2, 3}) {
                if (vec[k].n < 2) continue;
                const uint8_t *pa = pairs.data() + (size_t)k * 2 * g2->uncomp;
                ratio_jobs.push_back({pa, pa + g2->uncomp, g2_pair, g2_pair + g2->uncomp, std::string("power ratio check failed for ") + vec[k].name});
            }
        if (n2 >= 2 && g1_pair) {
            const uint8_t *pa = pairs.data() + (size_t)1 * 2 * g2->uncomp;
            ratio_jobs.push_back({g1_pair, g1_pair + g1->uncomp, pa, pa + g2->uncomp, "power ratio check failed for tau_g2"});
        }
    }
    // the power-ratio pairings start at once; the proofs of knowledge are bound to BLAKE2b(challenge) and wait for it
    char ratio_msg[256] = {0};
    const int curve_id = p->curve;
    auto fut_ratio = std::async(std::launch::async, [&ratio_jobs, &ratio_msg, curve_id, device] {
        return run_ratio_jobs(curve_id, ratio_jobs, Err{ratio_msg, sizeof(ratio_msg)}, device);
    });
    struct RatioJoin {   // the pairing threads read `pairs` and the caller's buffers
        std::future<int> &f;
        ~RatioJoin() { if (f.valid()) f.wait(); }
    } ratio_join{fut_ratio};
    std::vector<uint8_t> pok_scratch, g2_s_own;
    const bool chain_ok = fut_chain.get();
    tr.mark("BLAKE2b(challenge) done");
    if (verify_pok) {
        if (!g2_s) {
            g2_s_own.resize(3 * (size_t)g2->uncomp);
            const uint8_t *pk = response + sz.contribution_size - sz.public_key_size;
            if ((rc = ptau_compute_g2_s(p->curve, chal_digest, pk, sz.public_key_size, g2_s_own.data(), errbuf, err_len))) return rc;
            g2_s = g2_s_own.data();
        }
        if ((rc = collect_pok_jobs(p, sz, challenge, response, g2_s, g2_s_own.empty(), pok_scratch, pok_jobs, err))) return rc;
        if ((rc = run_ratio_jobs(p->curve, pok_jobs, err, device))) return rc;
    }
    if ((rc = fut_ratio.get())) return err.fail(rc, "%s", ratio_msg);
    tr.mark("pairing checks done");
    // how many pairing checks THIS call ran: ratio_checks == 0 under ratio_check means "not performed" (ptau.h)
    call_timing.mine.ratio_checks = (uint32_t)ratio_jobs.size();
    call_timing.mine.pok_checks = (uint32_t)pok_jobs.size();
    call_timing.mine.d2h_ms += (double)track.d2h_ns.load() / 1e6;
    // new_challenge = BLAKE2b(response) | decompressed elements
    fut_resp.wait();
    if (!chain_ok)
        return err.fail(PTAU_E_HASH_CHAIN, "response does not start with the hash of the challenge it claims to extend");
    joiner.ok = true;
    tr.mark("BLAKE2b(response) done");
    if (fut_nc.valid()) fut_nc.wait();
    tr.mark("BLAKE2b(new challenge) done");
    return PTAU_OK;
}
}  // namespace

extern "C" {

int ptau_verify_chunk(const ptau_params *p, const uint8_t *challenge, size_t challenge_len, const uint8_t *response,
                      size_t response_len, uint8_t *new_challenge, size_t new_challenge_len, int check_input,
                      int check_output, int subgroup_check_mode, int ratio_check, int verify_pok, const uint8_t *g2_s,
                      const uint8_t *tau_g2_check, int device, char *errbuf, size_t err_len) {
    return verify_chunk_impl(p, challenge, challenge_len, response, response_len, new_challenge, new_challenge_len, check_input,
                             check_output, subgroup_check_mode, ratio_check, verify_pok, g2_s, tau_g2_check, device, nullptr, errbuf,
                             err_len);
}

int ptau_transform_pok_and_correctness_file(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path,
                                            int check_input, const char *response_path, const char *response_hash_path,
                                            int check_output, const char *new_challenge_path,
                                            const char *new_challenge_hash_path, int subgroup_check_mode, int ratio_check,
                                            const uint8_t *g2_s, const uint8_t *tau_g2_check, int device, char *errbuf,
                                            size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge_path || !challenge_hash_path || !response_path || !response_hash_path || !new_challenge_path ||
        !new_challenge_hash_path)
        return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    Trace tr{"transform_pok_and_correctness"};
    MappedFile ch, resp, nc;
    if ((rc = ch.open_ro(challenge_path, err))) return rc;
    if ((rc = resp.open_ro(response_path, err, /*pin=*/true, device))) return rc;
    if (ch.len != sz.accumulator_size || resp.len != sz.contribution_size)
        return err.fail(PTAU_E_LEN, "challenge/response file sizes %zu/%zu do not match the parameters (%llu/%llu)", ch.len,
                        resp.len, (unsigned long long)sz.accumulator_size, (unsigned long long)sz.contribution_size);
    if ((rc = nc.create_new(new_challenge_path, sz.accumulator_size, err, device))) return rc;
    VerifyDigests dg;
    tr.mark("files mapped");
    // input and output mappings are pinned before the device section starts (every copy is then DMA at full rate and the
    // section is as short as its kernels: it holds the device lock), but AFTER the digests have started
    const std::function<void()> pin_wait = [&] {
        resp.wait_pinned();
        nc.wait_pinned();
        tr.mark("mappings pinned");
    };
    rc = verify_chunk_impl(p, (const uint8_t *)ch.p, ch.len, (const uint8_t *)resp.p, resp.len, (uint8_t *)nc.p, nc.len, check_input,
                           check_output, subgroup_check_mode, ratio_check, /*verify_pok=*/1, g2_s, tau_g2_check, device, &dg, errbuf, err_len,
                           &pin_wait);
    tr.mark("verify_chunk_impl done");
    // the reference writes challenge.hash and response.hash before it verifies: keep them on the failure paths that got that far
    if (rc == PTAU_OK || rc == PTAU_E_HASH_CHAIN || rc == PTAU_E_INVALID_RATIO || rc == PTAU_E_NOT_IN_SUBGROUP ||
        rc == PTAU_E_POINT_AT_INFINITY || rc == PTAU_E_POINT) {
        int rc2;
        if ((rc2 = write_small_file(challenge_hash_path, dg.challenge, 64, err))) return rc2;
        if ((rc2 = write_small_file(response_hash_path, dg.response, 64, err))) return rc2;
    }
    if (rc) return rc;
    if (msync(nc.p, nc.len, MS_SYNC)) return err.fail(PTAU_E_IO, "msync failed: %s", strerror(errno));
    tr.mark("msync done");
    rc = write_small_file(new_challenge_hash_path, dg.new_challenge, 64, err);
    if (Trace::on() && getenv("PTAU_SYNC_UNMAP")) {   // what taking the mappings apart costs
        nc.release();
        tr.mark("new challenge unmapped");
        resp.release();
        ch.release();
        tr.mark("inputs unmapped");
    }
    nc.release_async(device);
    resp.release_async(device);
    ch.release_async(device);
    tr.mark("returning");
    return rc;
}

int ptau_msm(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, const uint64_t *scalars, uint8_t *out,
             int compressed_out, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (!out || (n && (!in || !scalars))) return err.fail(PTAU_E_ARG, "null pointer argument");
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    const size_t isz = compressed_in ? ops->comp : ops->uncomp;
    const int nl64 = (ops->fr_limbs32 + 1) / 2;
    // scalars travel as 32-bit limbs; Fr of BLS12-377 has 8 of them, the 753-bit fields 24
    std::vector<uint32_t> k32((size_t)n * ops->fr_limbs32);
    for (uint64_t i = 0; i < n; i++)
        for (int j = 0; j < ops->fr_limbs32; j++) k32[i * ops->fr_limbs32 + j] = (uint32_t)(scalars[i * nl64 + j / 2] >> (32 * (j & 1)));
    if ((rc = grow(c->d_in, c->in_cap, std::max<size_t>(1, n * isz + k32.size() * 4 + 16), err))) return rc;
    uint8_t *d_pts = (uint8_t *)c->d_in;
    uint32_t *d_k = (uint32_t *)(d_pts + ((n * isz + 15) / 16) * 16);
    if (n) {
        HIP_TRY(hipMemcpyAsync(d_pts, in, n * isz, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_k, k32.data(), k32.size() * 4, hipMemcpyHostToDevice, c->stream));
    }
    return run_msm_dev(c, ops, d_pts, compressed_in, n, ptau::ScalarSource{ptau::SCALAR_ARRAY, d_k, 0, 0, 0}, out, nullptr, compressed_out, err);
}

int ptau_power_pairs(int curve, int group, const uint8_t *in, int compressed_in, uint64_t n, const uint8_t seed[16],
                     uint64_t index_base, uint8_t *out_a, uint8_t *out_b, int device, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (!in || !out_a || !out_b) return err.fail(PTAU_E_ARG, "null pointer argument");
    if (n < 2) return err.fail(PTAU_E_ARG, "power_pairs needs at least two elements");
    uint64_t s[2];
    if (seed) {
        memcpy(s, seed, 16);
    } else {   // per-verification secret from the OS
        FILE *f = fopen("/dev/urandom", "rb");
        if (!f || fread(s, 1, 16, f) != 16) {
            if (f) fclose(f);
            return err.fail(PTAU_E_IO, "cannot read /dev/urandom for the random linear combination");
        }
        fclose(f);
    }
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    if ((rc = begin_device_section(c, device, err))) return rc;
    const size_t isz = compressed_in ? ops->comp : ops->uncomp;
    if ((rc = grow(c->d_in, c->in_cap, n * isz, err))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_in, in, n * isz, hipMemcpyHostToDevice, c->stream));
    const ptau::ScalarSource src{ptau::SCALAR_PRF, nullptr, s[0], s[1], index_base};
    // (sum rho_i P_i, sum rho_i P_{i+1}), i in [0, n-1): the same coefficients on the vector and on its shift
    return run_msm_dev(c, ops, (const uint8_t *)c->d_in, compressed_in, n - 1, src, out_a, out_b, 0, err);
}

int ptau_check_same_ratio_batch(int curve, const uint8_t *checks, uint64_t n, int *results, uint32_t *gt_out, int device, char *errbuf,
                                size_t err_len) {
    Err err{errbuf, err_len};
    const ptau::PairingOps *po = get_pairing_ops(curve);
    if (!po) return err.fail(PTAU_E_ARG, "unknown curve id %d", curve);
    if (n && (!checks || !results)) return err.fail(PTAU_E_ARG, "null pointer argument");
    if (!n) return PTAU_OK;
    if (n > 4096) return err.fail(PTAU_E_ARG, "at most 4096 checks per call");
    DeviceCtx *c;
    int rc = ensure_ctx(device, c, err);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->pair_mu);
    HIP_TRY(hipSetDevice(device));
    const size_t in_bytes = n * po->check_bytes, gt_bytes = gt_out ? n * (size_t)po->K * po->n32 * 4 : 0;
    const size_t off_res = (in_bytes + 255) & ~(size_t)255, off_gt = off_res + ((n * sizeof(int) + 255) & ~(size_t)255);
    if ((rc = grow(c->d_pair, c->pair_cap, off_gt + gt_bytes + 256, err))) return rc;
    uint8_t *d = (uint8_t *)c->d_pair;
    HIP_TRY(hipMemcpyAsync(d, checks, in_bytes, hipMemcpyHostToDevice, c->pair_stream));
    po->launch(c->pair_stream, d, (int *)(d + off_res), gt_out ? (uint32_t *)(d + off_gt) : nullptr, (int)n);
    HIP_TRY(hipMemcpyAsync(results, d + off_res, n * sizeof(int), hipMemcpyDeviceToHost, c->pair_stream));
    if (gt_out) HIP_TRY(hipMemcpyAsync(gt_out, d + off_gt, gt_bytes, hipMemcpyDeviceToHost, c->pair_stream));
    HIP_TRY(hipStreamSynchronize(c->pair_stream));
    HIP_TRY(hipGetLastError());
    return PTAU_OK;
}

int ptau_check_same_ratio(int curve, const uint8_t *g1_a, const uint8_t *g1_b, const uint8_t *g2_c, const uint8_t *g2_d,
                          char *errbuf, size_t err_len) {
    using namespace ptau;
    Err err{errbuf, err_len};
    if (!g1_a || !g1_b || !g2_c || !g2_d) return err.fail(PTAU_E_ARG, "null pointer argument");
    int r;
    switch (curve) {
    case PTAU_CURVE_BLS12_377: r = TatePairing<BlsFqR29, BlsFr, BlsG1Cfg, BlsG2Cfg, BlsPairingCfg>::check_same_ratio(g1_a, g1_b, g2_c, g2_d); break;
    case PTAU_CURVE_BW6_761: r = TatePairing<Bw6FqR29, Bw6Fr, Bw6G1Cfg, Bw6G2Cfg, Bw6PairingCfg>::check_same_ratio(g1_a, g1_b, g2_c, g2_d); break;
    case PTAU_CURVE_MNT4_753: r = TatePairing<Mnt4FqR29, Mnt4Fr, Mnt4G1Cfg, Mnt4G2Cfg, Mnt4PairingCfg>::check_same_ratio(g1_a, g1_b, g2_c, g2_d); break;
    case PTAU_CURVE_MNT6_753: r = TatePairing<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg, Mnt6G2Cfg, Mnt6PairingCfg>::check_same_ratio(g1_a, g1_b, g2_c, g2_d); break;
    default: return err.fail(PTAU_E_ARG, "unknown curve id %d", curve);
    }
    if (r < 0) return err.fail(PTAU_E_POINT, "non-canonical point in a ratio pair");
    if (r == 0) return err.fail(PTAU_E_INVALID_RATIO, "the two pairs do not have the same ratio");
    return PTAU_OK;
}

// ---- Phase1::initialization / aggregation / aggregate_verification / PoK --------------------------------------
static const unsigned char *generator_bytes(int curve, int group) {
    using G = ptau::Generators;
    switch (curve * 2 + (group - 1)) {
    case 0: return G::BLS_G1;   case 1: return G::BLS_G2;
    case 2: return G::BW6_G1;   case 3: return G::BW6_G2;
    case 4: return G::MNT4_G1;  // MNT G2: built in only once the deferred pin has supplied them (see ptau_new_challenge)
    case 5: return G::HAVE_MNT_G2 ? G::MNT4_G2 : nullptr;
    case 6: return G::MNT6_G1;
    case 7: return G::HAVE_MNT_G2 ? G::MNT6_G2 : nullptr;
    }
    return nullptr;
}

// Phase1::initialization with the two generators supplied by the caller (uncompressed arkworks bytes of
// G1Affine::generator() / G2Affine::generator()): the accumulator is bit-exact with the reference's by construction.
int ptau_new_challenge_with_generators(const ptau_params *p, const uint8_t *g1_generator, const uint8_t *g2_generator, uint8_t *out,
                                       size_t out_len, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!g1_generator || !g2_generator) return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    if (!out || out_len != sz.accumulator_size)
        return err.fail(PTAU_E_LEN, "challenge buffer must be %llu bytes", (unsigned long long)sz.accumulator_size);
    const GroupOps *g1 = get_ops(p->curve, 1), *g2 = get_ops(p->curve, 2);
    ptau_blake2b512(nullptr, 0, out);   // blank hash: the initial challenge extends nothing
    uint8_t *po = out + 64;
    auto fill = [&](const uint8_t *pt, int size, uint64_t n) {
        for (uint64_t i = 0; i < n; i++, po += size) memcpy(po, pt, size);
    };
    fill(g1_generator, g1->uncomp, sz.g1_chunk_size);
    fill(g2_generator, g2->uncomp, sz.other_chunk_size);
    fill(g1_generator, g1->uncomp, sz.other_chunk_size);
    fill(g1_generator, g1->uncomp, sz.other_chunk_size);
    fill(g2_generator, g2->uncomp, 1);
    return PTAU_OK;
}

int ptau_generator(int curve, int group, uint8_t *out, size_t out_len, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    const GroupOps *ops = get_ops(curve, group);
    if (!ops) return err.fail(PTAU_E_ARG, "unknown curve/group %d/%d", curve, group);
    if (!out || out_len != (size_t)ops->uncomp) return err.fail(PTAU_E_LEN, "generator buffer must be %d bytes", ops->uncomp);
    if (!generator_bytes(curve, group))
        return err.fail(PTAU_E_UNSUPPORTED, "the arkworks G2 generator of this curve is not built in (run tools/pin_reference.sh and "
                                            "rebuild, or pass G2Affine::generator() through ptau_new_challenge_with_generators)");
    memcpy(out, generator_bytes(curve, group), ops->uncomp);
    return PTAU_OK;
}

int ptau_new_challenge(const ptau_params *p, uint8_t *out, size_t out_len, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    // MNT4-753 / MNT6-753: ark-mnt4-753 / ark-mnt6-753's G2 generator constants are not recoverable offline (DESIGN.md
    // §2) — an accumulator built from any other generator would NOT be the reference's initial challenge, so the built-in
    // form refuses unless the library was built with the pinned constants (csrc/pinned_generators.json, written by
    // tools/pin_reference.sh from arkworks' own G2Affine::generator()); the Rust host can always pass its own through
    // ptau_new_challenge_with_generators.
    if (!generator_bytes(p->curve, 2))
        return err.fail(PTAU_E_UNSUPPORTED, "new_challenge on MNT4-753 / MNT6-753 needs the arkworks G2 generator: run "
                                            "tools/pin_reference.sh and rebuild, or use ptau_new_challenge_with_generators");
    return ptau_new_challenge_with_generators(p, generator_bytes(p->curve, 1), generator_bytes(p->curve, 2), out, out_len, errbuf, err_len);
}

int ptau_new_challenge_file(const ptau_params *p, const char *challenge_path, const char *challenge_hash_path, char *errbuf,
                            size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge_path || !challenge_hash_path) return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    MappedFile f;
    if ((rc = f.create_new(challenge_path, sz.accumulator_size, err))) return rc;
    if ((rc = ptau_new_challenge(p, (uint8_t *)f.p, f.len, errbuf, err_len))) return rc;
    uint8_t digest[64];
    ptau_blake2b512((const uint8_t *)f.p, f.len, digest);
    if (msync(f.p, f.len, MS_SYNC)) return err.fail(PTAU_E_IO, "msync failed: %s", strerror(errno));
    return write_small_file(challenge_hash_path, digest, 64, err);
}

}  // extern "C"

namespace {
// Phase1::aggregation on mapped / in-memory chunk responses: every vector of every chunk is decompressed on the device
// into its place in the Full-mode accumulator.  The 64-byte header stays zero (ptau.h, ptau_combine_file).
int combine_impl(const ptau_params *full, uint64_t chunk_size, const uint8_t *const *responses, const size_t *response_lens,
                 uint64_t n_chunks, uint8_t *combined, size_t combined_len, int device, const Err &err) {
    int rc = check_params(full, err);
    if (rc) return rc;
    if (full->contribution_mode != PTAU_MODE_FULL) return err.fail(PTAU_E_ARG, "combine needs Full-mode output parameters");
    if (!responses || !response_lens || !combined || !chunk_size) return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes fs;
    ptau_params_sizes(full, &fs);
    if (combined_len != fs.accumulator_size)
        return err.fail(PTAU_E_LEN, "combined buffer must be %llu bytes", (unsigned long long)fs.accumulator_size);
    if (n_chunks != (fs.powers_g1_length + chunk_size - 1) / chunk_size)
        return err.fail(PTAU_E_ARG, "expected %llu chunk responses, got %llu",
                        (unsigned long long)((fs.powers_g1_length + chunk_size - 1) / chunk_size), (unsigned long long)n_chunks);
    const GroupOps *g1 = get_ops(full->curve, 1), *g2 = get_ops(full->curve, 2);
    DeviceCtx *c;
    if ((rc = ensure_ctx(device, c, err))) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    CallTiming call_timing{c, device};
    CopyTrack track;   // (after the lock: drained before the device is released and before the caller's buffers go away)
    if ((rc = begin_device_section(c, device, err))) return rc;
    memset(combined, 0, 64);
    // destination vector bases in the full accumulator
    uint8_t *d_tau_g1 = combined + 64;
    uint8_t *d_tau_g2 = d_tau_g1 + fs.powers_g1_length * g1->uncomp;
    uint8_t *d_alpha = d_tau_g2 + fs.powers_length * g2->uncomp;
    uint8_t *d_beta = d_alpha + fs.powers_length * g1->uncomp;
    uint8_t *d_beta_g2 = d_beta + fs.powers_length * g1->uncomp;
    for (uint64_t k = 0; k < n_chunks; k++) {
        ptau_params cp = *full;
        cp.contribution_mode = PTAU_MODE_CHUNKED;
        cp.chunk_index = k;
        cp.chunk_size = chunk_size;
        ptau_sizes cs;
        ptau_params_sizes(&cp, &cs);
        if (!responses[k] || response_lens[k] != cs.contribution_size)
            return err.fail(PTAU_E_LEN, "response of chunk %llu is %zu bytes, expected %llu", (unsigned long long)k,
                            response_lens[k], (unsigned long long)cs.contribution_size);
        const uint8_t *pi = responses[k] + 64;
        const uint64_t n1 = cs.g1_chunk_size, n2 = cs.other_chunk_size, s0 = cs.start;
        struct { const GroupOps *ops; int group; uint64_t n; uint8_t *dst; } v[5] = {
            {g1, 1, n1, d_tau_g1 + s0 * g1->uncomp}, {g2, 2, n2, d_tau_g2 + s0 * g2->uncomp},
            {g1, 1, n2, d_alpha + s0 * g1->uncomp},  {g1, 1, n2, d_beta + s0 * g1->uncomp},
            {g2, 2, (uint64_t)(k == 0 ? 1 : 0), d_beta_g2}};
        for (int j = 0; j < 5; j++) {
            const uint64_t stored = j == 4 ? 1 : v[j].n;   // every chunk stores beta_g2; only chunk 0's is kept
            if (v[j].n &&
                (rc = run_verify_vector(c, full->curve, v[j].group, v[j].ops, pi, v[j].dst, v[j].n, PTAU_CHECK_NO,
                                        PTAU_SUBGROUP_NO, nullptr, 0, nullptr, nullptr, nullptr, 0, &track, err)))
                return rc;
            pi += stored * v[j].ops->comp;
        }
    }
    track.drain();   // every vector's copy has landed
    c->timing.d2h_ms += (double)track.d2h_ns.load() / 1e6;
    return PTAU_OK;
}

// pair-index range [lo, hi) of slice k of n_pairs over nd slices: contiguous, balanced to within one pair
void slice_bounds(uint64_t n_pairs, uint64_t nd, uint64_t k, uint64_t &lo, uint64_t &hi) {
    const uint64_t base = n_pairs / nd, extra = n_pairs % nd;
    lo = k * base + std::min(k, extra);
    hi = lo + base + (k < extra ? 1 : 0);
}

// Phase1::aggregate_verification over one or several devices (ptau.h: ptau_transform_ratios / _devices).
int transform_ratios_impl(const ptau_params *full, const uint8_t *combined, size_t combined_len, int check_input,
                          const int *devices, int n_devices, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(full, err);
    if (rc) return rc;
    if (!combined || !devices || n_devices < 1 || n_devices > MAX_DEVICES) return err.fail(PTAU_E_ARG, "null pointer / bad device list");
    // an ordinal may be listed more than once: its slices then queue on that device's lock (one GPU, several slices)
    ptau_params fp = *full;
    fp.contribution_mode = PTAU_MODE_FULL;   // a chunk-0 parameter block of the same ceremony is accepted
    ptau_sizes fs;
    ptau_params_sizes(&fp, &fs);
    if (combined_len != fs.accumulator_size)
        return err.fail(PTAU_E_LEN, "combined accumulator must be %llu bytes", (unsigned long long)fs.accumulator_size);
    const GroupOps *g1 = get_ops(fp.curve, 1), *g2 = get_ops(fp.curve, 2);
    const uint64_t n1 = fs.g1_chunk_size, n2 = fs.other_chunk_size;
    const uint8_t *v_tau_g1 = combined + 64;
    const uint8_t *v_tau_g2 = v_tau_g1 + n1 * g1->uncomp;
    const uint8_t *v_alpha = v_tau_g2 + n2 * g2->uncomp;
    const uint8_t *v_beta = v_alpha + n2 * g1->uncomp;
    const uint8_t *v_beta_g2 = v_beta + n2 * g1->uncomp;
    if (n1 < 2 || n2 < 2) return err.fail(PTAU_E_ARG, "accumulator too small for ratio checks");
    struct Vec { const GroupOps *ops; int group; const uint8_t *ptr; uint64_t n; const char *name; } vec[4] = {
        {g1, 1, v_tau_g1, n1, "tau_g1"}, {g2, 2, v_tau_g2, n2, "tau_g2"}, {g1, 1, v_alpha, n2, "alpha_g1"}, {g1, 1, v_beta, n2, "beta_g1"}};
    uint8_t seed[16];
    {   // ONE coefficient stream for every slice of every vector: fresh secret from the OS
        FILE *f = fopen("/dev/urandom", "rb");
        if (!f || fread(seed, 1, 16, f) != 16) {
            if (f) fclose(f);
            return err.fail(PTAU_E_IO, "cannot read /dev/urandom for the random linear combination");
        }
        fclose(f);
    }
    const size_t u2 = g2->uncomp;   // >= g1->uncomp on every curve: one slot size for all partial pairs
    const uint64_t nd = (uint64_t)n_devices;
    std::vector<uint8_t> parts((size_t)4 * nd * 2 * u2);           // [vector][device][a | b]
    std::vector<uint8_t> have((size_t)4 * nd, 0);
    const bool want_sg = check_input == PTAU_CHECK_FULL || check_input == PTAU_CHECK_ONLY_IN_GROUP;
    const int curve = fp.curve;
    auto lane = [&](uint64_t k, char *msg, size_t msg_len) -> int {   // device k's slices, vector by vector
        for (int v = 0; v < 4; v++) {
            uint64_t lo, hi;
            slice_bounds(vec[v].n - 1, nd, k, lo, hi);
            if (hi <= lo) continue;
            const size_t usz = vec[v].ops->uncomp;
            const uint8_t *ptr = vec[v].ptr + lo * usz;
            const uint64_t cnt = hi - lo + 1;   // pairs (lo, lo+1) .. (hi-1, hi): points lo .. hi
            int r;
            if (want_sg && (r = ptau_subgroup_check(curve, vec[v].group, ptr, 0, cnt, devices[k], msg, msg_len))) return r;
            uint8_t *pa = parts.data() + ((size_t)v * nd + k) * 2 * u2;
            if ((r = ptau_power_pairs(curve, vec[v].group, ptr, 0, cnt, seed, lo, pa, pa + u2, devices[k], msg, msg_len))) return r;
            have[(size_t)v * nd + k] = 1;
        }
        return PTAU_OK;
    };
    std::vector<std::array<char, 256>> msgs(nd);
    std::vector<std::future<int>> futs;
    for (uint64_t k = 1; k < nd; k++)
        futs.push_back(std::async(std::launch::async, [&, k] { msgs[k][0] = 0; return lane(k, msgs[k].data(), msgs[k].size()); }));
    msgs[0][0] = 0;
    rc = lane(0, msgs[0].data(), msgs[0].size());
    int first_bad = rc ? 0 : -1;
    for (uint64_t k = 1; k < nd; k++) {   // join every lane before anything of this frame goes away; first failure in device order
        const int r = futs[k - 1].get();
        if (r && first_bad < 0) { first_bad = (int)k; rc = r; }
    }
    if (first_bad >= 0) return err.fail(rc, "%s", msgs[first_bad].data());
    // partial pairs -> one pair per vector (a sum of <= n_devices points: the MSM with unit scalars, on devices[0])
    std::vector<uint8_t> pairs(4 * 2 * u2);
    const int nl64 = (g1->fr_limbs32 + 1) / 2;
    for (int v = 0; v < 4; v++) {
        const size_t usz = vec[v].ops->uncomp;
        std::vector<uint8_t> pa, pb;
        for (uint64_t k = 0; k < nd; k++) {
            if (!have[(size_t)v * nd + k]) continue;
            const uint8_t *q = parts.data() + ((size_t)v * nd + k) * 2 * u2;
            pa.insert(pa.end(), q, q + usz);
            pb.insert(pb.end(), q + u2, q + u2 + usz);
        }
        const uint64_t m = pa.size() / usz;
        uint8_t *oa = pairs.data() + (size_t)v * 2 * u2, *ob = oa + u2;
        if (m == 1) {
            memcpy(oa, pa.data(), usz);
            memcpy(ob, pb.data(), usz);
        } else {
            std::vector<uint64_t> ones(m * nl64, 0);
            for (uint64_t i = 0; i < m; i++) ones[i * nl64] = 1;
            if ((rc = ptau_msm(curve, vec[v].group, pa.data(), 0, m, ones.data(), oa, 0, devices[0], errbuf, err_len))) return rc;
            if ((rc = ptau_msm(curve, vec[v].group, pb.data(), 0, m, ones.data(), ob, 0, devices[0], errbuf, err_len))) return rc;
        }
    }
    // G1 vectors against (tauG2[0], tauG2[1]); tauG2 against (tauG1[0], tauG1[1]); betaG2 against betaG1[0]:
    // five same-ratio checks, one launch of the pairing kernel
    std::vector<RatioJob> jobs;
    for (int k : {0, 2, 3}) {
        const uint8_t *pa = pairs.data() + (size_t)k * 2 * u2;
        jobs.push_back({pa, pa + u2, v_tau_g2, v_tau_g2 + g2->uncomp, std::string("power ratio check failed for ") + vec[k].name});
    }
    {
        const uint8_t *pa = pairs.data() + (size_t)1 * 2 * u2;
        jobs.push_back({v_tau_g1, v_tau_g1 + g1->uncomp, pa, pa + u2, "power ratio check failed for tau_g2"});
    }
    // e(G1, betaG2) == e(betaG1[0], G2): (tauG1[0], betaG1[0]) ~ (tauG2[0], betaG2)
    jobs.push_back({v_tau_g1, v_beta, v_tau_g2, v_beta_g2, "beta_g2 is not consistent with beta_g1[0]"});
    return run_ratio_jobs(curve, jobs, err, devices[0]);
}

// one path per line (control.rs:547-557 writes "{}\n" per chunk); blank lines and a trailing '\r' are tolerated
int read_list_file(const char *path, std::vector<std::string> &out, const Err &err) {
    FILE *f = fopen(path, "rb");
    if (!f) return err.fail(PTAU_E_IO, "cannot open %s: %s", path, strerror(errno));
    std::string cur;
    int ch;
    auto flush = [&] {
        while (!cur.empty() && (cur.back() == '\r' || cur.back() == ' ')) cur.pop_back();
        if (!cur.empty()) out.push_back(cur);
        cur.clear();
    };
    while ((ch = fgetc(f)) != EOF) {
        if (ch == '\n') flush();
        else cur.push_back((char)ch);
    }
    flush();
    fclose(f);
    return PTAU_OK;
}
}  // namespace

extern "C" {

int ptau_combine(const ptau_params *full, uint64_t chunk_size, const uint8_t *const *responses, const size_t *response_lens,
                 uint64_t n_chunks, uint8_t *combined, size_t combined_len, int device, char *errbuf, size_t err_len) {
    return combine_impl(full, chunk_size, responses, response_lens, n_chunks, combined, combined_len, device, Err{errbuf, err_len});
}

int ptau_combine_file(const ptau_params *chunk0, const char *response_list_path, const char *combined_path, int device,
                      char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(chunk0, err);
    if (rc) return rc;
    if (!response_list_path || !combined_path) return err.fail(PTAU_E_ARG, "null pointer argument");
    if (chunk0->contribution_mode != PTAU_MODE_CHUNKED)
        return err.fail(PTAU_E_ARG, "combine takes the parameters of a chunk (create_parameters_for_chunk(.., 0))");
    ptau_params full = *chunk0;
    full.contribution_mode = PTAU_MODE_FULL;
    full.chunk_index = 0;
    ptau_sizes fs;
    ptau_params_sizes(&full, &fs);
    std::vector<std::string> paths;
    if ((rc = read_list_file(response_list_path, paths, err))) return rc;
    const uint64_t n_chunks = (fs.powers_g1_length + chunk0->chunk_size - 1) / chunk0->chunk_size;
    if (paths.size() != n_chunks)
        return err.fail(PTAU_E_ARG, "%s lists %zu responses, the ceremony has %llu chunks", response_list_path, paths.size(),
                        (unsigned long long)n_chunks);
    std::vector<std::unique_ptr<MappedFile>> maps;
    std::vector<const uint8_t *> ptrs;
    std::vector<size_t> lens;
    const bool pin = n_chunks <= 64;   // one populate / register thread per mapping: bounded
    for (auto &pth : paths) {
        maps.emplace_back(new MappedFile());
        if ((rc = maps.back()->open_ro(pth.c_str(), err, pin, device))) return rc;
        ptrs.push_back((const uint8_t *)maps.back()->p);
        lens.push_back(maps.back()->len);
    }
    MappedFile out;
    if ((rc = out.create_new(combined_path, fs.accumulator_size, err, device))) return rc;
    for (auto &m : maps) m->wait_pinned();
    if ((rc = combine_impl(&full, chunk0->chunk_size, ptrs.data(), lens.data(), n_chunks, (uint8_t *)out.p, out.len, device, err))) return rc;
    if (out.populate.valid()) out.populate.wait();
    if (msync(out.p, out.len, MS_SYNC)) return err.fail(PTAU_E_IO, "msync failed: %s", strerror(errno));
    return PTAU_OK;
}

int ptau_transform_ratios(const ptau_params *full, const uint8_t *combined, size_t combined_len, int check_input, int device,
                          char *errbuf, size_t err_len) {
    return transform_ratios_impl(full, combined, combined_len, check_input, &device, 1, errbuf, err_len);
}

int ptau_transform_ratios_devices(const ptau_params *full, const uint8_t *combined, size_t combined_len, int check_input,
                                  const int *devices, int n_devices, char *errbuf, size_t err_len) {
    return transform_ratios_impl(full, combined, combined_len, check_input, devices, n_devices, errbuf, err_len);
}

int ptau_transform_ratios_file_devices(const ptau_params *full, const char *combined_path, int check_input, const int *devices,
                                       int n_devices, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(full, err);
    if (rc) return rc;
    if (!combined_path || !devices || n_devices < 1) return err.fail(PTAU_E_ARG, "null pointer / bad device list");
    MappedFile f;
    if ((rc = f.open_ro(combined_path, err, /*pin=*/true, devices[0]))) return rc;
    f.wait_pinned();
    return transform_ratios_impl(full, (const uint8_t *)f.p, f.len, check_input, devices, n_devices, errbuf, err_len);
}

int ptau_transform_ratios_file(const ptau_params *full, const char *combined_path, int check_input, int device, char *errbuf,
                               size_t err_len) {
    return ptau_transform_ratios_file_devices(full, combined_path, check_input, &device, 1, errbuf, err_len);
}

int ptau_verify_pok(const ptau_params *p, const uint8_t *challenge, size_t challenge_len, const uint8_t *response,
                    size_t response_len, const uint8_t *g2_s, char *errbuf, size_t err_len) {
    Err err{errbuf, err_len};
    int rc = check_params(p, err);
    if (rc) return rc;
    if (!challenge || !response) return err.fail(PTAU_E_ARG, "null pointer argument");
    ptau_sizes sz;
    ptau_params_sizes(p, &sz);
    if (challenge_len != sz.accumulator_size || response_len != sz.contribution_size)
        return err.fail(PTAU_E_LEN, "challenge/response sizes do not match the parameters");
    std::vector<uint8_t> own;
    const bool g2_s_from_caller = g2_s != nullptr;
    if (!g2_s) {   // derive the three compute_g2_s points from BLAKE2b(challenge) and the public key
        uint8_t digest[64];
        ptau_blake2b512(challenge, challenge_len, digest);
        own.resize(3 * (size_t)get_ops(p->curve, 2)->uncomp);
        if ((rc = ptau_compute_g2_s(p->curve, digest, response + sz.contribution_size - sz.public_key_size, sz.public_key_size, own.data(),
                                    errbuf, err_len)))
            return rc;
        g2_s = own.data();
    }
    std::vector<uint8_t> scratch;
    std::vector<RatioJob> jobs;
    if ((rc = collect_pok_jobs(p, sz, challenge, response, g2_s, g2_s_from_caller, scratch, jobs, err))) return rc;
    return run_ratio_jobs(p->curve, jobs, err);
}

}  // extern "C"
