// Host emulation of ONE gfx950 wavefront for the wave-cooperative device code (csrc/mcrt_waveknn.hpp): 64 lanes as 64 fibers
// on one thread. A lane runs until its next cross-lane operation (ballot, readlane, DPP move, shuffle, wave barrier),
// publishes its operand and yields; when all 64 have arrived the scheduler snapshots the operands and resumes the lanes, each of
// which then computes its own result from the snapshot with the ISA's rule for that operation. So the code under test is the
// product's, unchanged, and what is emulated is exactly the part of the hardware it leans on: lockstep at cross-lane operations,
// DPP controls with their row masks and bound_ctrl, readlane / readfirstlane, LDS as ordinary memory.
//
// What it is NOT: a timing model, and lanes do not run in lockstep BETWEEN cross-lane operations (lane 0 runs its whole stretch
// before lane 1 starts it). Device code that relies on lockstep without a cross-lane operation in between (all lanes read, then
// all lanes write the same array) needs __builtin_amdgcn_wave_barrier() there - which is a rendezvous here and free on the device.
//
// Test infrastructure only (tests/emu): define MCRT_WAVE_EMU and include this file BEFORE any csrc header.
#pragma once

#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

namespace wemu {

constexpr int kLanes = 64;
enum Op : uint32_t { kBallot = 1, kReadlane, kReadfirst, kDpp, kShfl, kBarrier, kBpermute, kSync };

// Switching fibers: swapcontext makes a system call per switch (the signal mask), and a search is a few thousand cross-lane operations
// x 64 lanes x 2 switches - so on x86-64 the switch is six pushes, the stack pointer, six pops (callee-saved registers only: the
// control words of the FPU / SSE unit do not change in this code).
#if defined(__x86_64__)
extern "C" void wemu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".hidden wemu_switch\n"
    ".globl wemu_switch\n"
    ".type wemu_switch,@function\n"
    "wemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size wemu_switch,.-wemu_switch\n");
struct Context {
    void* sp = nullptr;
};
inline void switchTo(Context& from, Context& to) { wemu_switch(&from.sp, to.sp); }
inline void makeFiber(Context& c, char* stack, size_t bytes, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    void** slot = (void**)(top - 16);   // the address `ret` jumps to; 16-aligned, so the entry sees the stack as after a call
    slot[0] = (void*)entry;
    slot[1] = nullptr;
    void** sp = slot - 6;               // r15, r14, r13, r12, rbx, rbp
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    c.sp = sp;
}
#else
struct Context {
    ucontext_t uc;
};
inline void switchTo(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
inline void makeFiber(Context& c, char* stack, size_t bytes, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = bytes;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

constexpr int kMaxWaves = 16;
struct Wave {  // a workgroup of `waves` wavefronts (1 for wemu::run); lane l of wave w is fiber 64 w + l
    Context sched;
    Context ctx[kMaxWaves * kLanes];
    bool done[kMaxWaves * kLanes];
    bool waiting[kMaxWaves * kLanes];
    uint64_t pub[kMaxWaves * kLanes], snap[kMaxWaves * kLanes];
    uint32_t op[kMaxWaves * kLanes];
    const char* site_file[kMaxWaves * kLanes];  // where in the SOURCE the lane waits: file and line of the cross-lane operation
    int site_line[kMaxWaves * kLanes];          // (code addresses will not do: the compiler duplicates calls into both sides of a branch)
    bool member[kMaxWaves * kLanes];       // the lane takes part in the operation its wave was last released from
    unsigned long long stamp[kMaxWaves * kLanes];  // when the lane arrived where it waits
    unsigned long long clock = 0;
    size_t depth[kMaxWaves * kLanes];      // how deep in calls it waits: bytes of its fiber's stack in use
    uintptr_t stack_top[kMaxWaves * kLanes];
    bool at_sync[kMaxWaves];  // the wave has arrived at __syncthreads and waits for the others
    int waves = 1;
    int cur = -1;             // fiber index
    std::function<void(int)> body;
    unsigned long long collectives = 0, divergent = 0;  // operations served; of them with the wave split over several call sites
};

inline unsigned long long& shuffleSeed() {
    static unsigned long long seed = getenv("WEMU_SHUFFLE") ? strtoull(getenv("WEMU_SHUFFLE"), nullptr, 0) : 0ull;
    return seed;
}

inline Wave*& current() {
    static thread_local Wave* w = nullptr;
    return w;
}

inline void trampoline() {
    Wave* w = current();
    const int lane = w->cur;
    w->body(lane);
    w->done[lane] = true;
    w->pub[lane] = 0;  // a finished lane contributes nothing to later ballots
    switchTo(w->ctx[lane], w->sched);
    abort();  // (a finished lane is never resumed)
}

// Runs body(thread) for the 64 x waves threads of one workgroup to completion (thread = 64 wave + lane). Cross-lane operations
// synchronise the lanes of one wave; __syncthreads (kSync) all waves: a wave that arrives there is parked until every wave that still
// has live lanes has arrived.
inline void runGroup(int waves, const std::function<void(int)>& body, size_t stack_bytes = 1u << 20) {
    if (waves < 1 || waves > kMaxWaves) abort();
    static thread_local std::vector<char> stacks[kMaxWaves * kLanes];  // kept between runs (64 MB of fresh pages per search otherwise)
    const int threads = waves * kLanes;
    for (int l = 0; l < threads; l++)
        if (stacks[l].size() < stack_bytes) stacks[l].resize(stack_bytes);
    static thread_local Wave* pool = nullptr;  // (half a megabyte: not on the caller's stack)
    if (!pool) pool = new Wave();
    Wave& w = *pool;
    Wave* saved = current();
    current() = &w;
    w.body = body;
    w.waves = waves;
    w.collectives = w.divergent = 0;
    for (int l = 0; l < threads; l++) {
        w.done[l] = w.waiting[l] = false;
        w.member[l] = true;
        w.pub[l] = w.snap[l] = 0;
        w.op[l] = 0;
        w.site_file[l] = nullptr;
        w.site_line[l] = 0;
        makeFiber(w.ctx[l], stacks[l].data(), stack_bytes, trampoline);
        w.stack_top[l] = (uintptr_t)stacks[l].data() + stack_bytes;
        w.depth[l] = 0;
    }
    for (int v = 0; v < waves; v++) w.at_sync[v] = false;
    unsigned long long passes = 0;
    for (;;) {
        if (++passes > 2000000ull) {  // (a kernel of the tests is a few hundred thousand passes: this is a livelock)
            fprintf(stderr, "wave_emu: no end after %llu passes; lane states of wave 0:\n", passes);
            for (int l = 0; l < kLanes; l++)
                fprintf(stderr, "  lane %d done %d waiting %d member %d op %u at %s:%d\n", l, (int)w.done[l], (int)w.waiting[l], (int)w.member[l], w.op[l],
                        w.site_file[l] ? w.site_file[l] : "-", w.site_line[l]);
            abort();
        }
        bool any = false;
        // shuffleSeed() != 0 (WEMU_SHUFFLE=<seed>, or set by the harness): the waves of the workgroup are visited in a random order that changes from pass to pass (and a wave may
        // be skipped for a pass), instead of round robin - what the hardware's arbitration may do. Results must not depend on it.
        const unsigned long long shuffle_seed = shuffleSeed();
        int order[kMaxWaves];
        for (int v = 0; v < waves; v++) order[v] = v;
        if (shuffle_seed) {
            static thread_local unsigned long long rng = 0, rng_seed = 0;
            if (rng_seed != shuffle_seed) {
                rng_seed = shuffle_seed;
                rng = shuffle_seed * 0x9E3779B97F4A7C15ull + 1ull;
            }
            auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            for (int v = waves - 1; v > 0; v--) std::swap(order[v], order[(int)(next() % (unsigned long long)(v + 1))]);
            // skip one wave this pass now and then (as long as another one can run)
            if (waves > 1 && (next() & 3ull) == 0ull) order[waves - 1] = -1;
        }
        for (int vi = 0; vi < waves; vi++) {
            const int v = order[vi];
            if (v < 0) {
                any = true;
                continue;
            }
            if (w.at_sync[v]) {
                any = true;
                continue;  // parked at __syncthreads
            }
            // resume the lanes of the wave that were released last time (all of them the first time); lanes parked at another
            // operation stay parked
            bool alive_wave = false;
            for (int l = v * kLanes; l < (v + 1) * kLanes; l++) {
                if (w.done[l]) continue;
                alive_wave = any = true;
                if (w.waiting[l]) continue;  // parked: not released yet
                w.cur = l;
                switchTo(w.sched, w.ctx[l]);
            }
            if (!alive_wave) continue;
            // Every live lane now waits at a cross-lane operation. Usually the same one: the wave is converged. Device code may also
            // run such operations under divergent control flow (a tree walk only the lanes with a ray take part in; wave-aggregated
            // atomics that elect a leader among the ACTIVE lanes) - then the lanes wait at different places in the source. The
            // hardware runs the sides of a branch one after the other and the lanes that skipped it wait where the sides meet. Here
            // ONE group goes, chosen so that lanes inside a branch catch up with the ones waiting behind it: the group deepest in
            // calls (the harnesses are built without inlining: an operation inside a function called from the branch is deeper than
            // one after the branch), among equals the earliest line of the same file, else the one that arrived last.
            // __syncthreads is never released for a part of a wave.
            const char* pick_file = nullptr;
            int pick_line = 0;
            uint32_t pick_op = 0;
            unsigned long long pick_stamp = 0;
            size_t pick_depth = 0;
            bool have = false, mixed = false;
            for (int l = v * kLanes; l < (v + 1) * kLanes; l++) {
                if (w.done[l]) continue;
                if (!w.waiting[l]) {
                    fprintf(stderr, "wave_emu: thread %d neither finished nor at a cross-lane operation\n", l);
                    abort();
                }
                const bool same = have && w.site_file[l] == pick_file && w.site_line[l] == pick_line && w.op[l] == pick_op;
                if (have && !same) mixed = true;
                if (same) {
                    pick_stamp = std::max(pick_stamp, w.stamp[l]);
                    pick_depth = std::max(pick_depth, w.depth[l]);
                    continue;
                }
                bool better = !have || (pick_op == kSync && w.op[l] != kSync);
                if (!better && w.op[l] != kSync && pick_op != kSync) {
                    // (frames of one function differ by a few bytes from lane to lane only through alignment: compare in steps of 64)
                    const size_t da = w.depth[l] / 64, db = pick_depth / 64;
                    if (da != db) better = da > db;
                    else if (w.site_file[l] == pick_file) better = w.site_line[l] < pick_line;
                    else better = w.stamp[l] > pick_stamp;
                }
                if (better) {
                    have = true;
                    pick_file = w.site_file[l];
                    pick_line = w.site_line[l];
                    pick_op = w.op[l];
                    pick_stamp = w.stamp[l];
                    pick_depth = w.depth[l];
                }
            }
            if (pick_op == kSync) {
                if (mixed) {
                    fprintf(stderr, "wave_emu: a part of a wave at __syncthreads\n");
                    abort();
                }
                w.at_sync[v] = true;  // (its lanes stay `waiting` until every wave has arrived)
                w.collectives++;
                continue;
            }
            for (int l = v * kLanes; l < (v + 1) * kLanes; l++) {
                const bool in = !w.done[l] && w.site_file[l] == pick_file && w.site_line[l] == pick_line && w.op[l] == pick_op;
                w.member[l] = in;
                w.snap[l] = in ? w.pub[l] : 0;
                if (in) w.waiting[l] = false;
            }
            if (mixed) {
                w.divergent++;
                static const bool dbg = getenv("WEMU_DEBUG") != nullptr;
                static int shown = 0;
                if (dbg && shown < 40) {
                    shown++;
                    fprintf(stderr, "wave_emu: split wave %d - released %s:%d op %u;", v, pick_file ? strrchr(pick_file, '/') : "-", pick_line, pick_op);
                    for (int l = v * kLanes; l < (v + 1) * kLanes; l++)
                        if (!w.done[l] && w.waiting[l]) fprintf(stderr, " [%d @%s:%d op %u]", l & 63, w.site_file[l] ? strrchr(w.site_file[l], '/') : "-", w.site_line[l], w.op[l]);
                    fprintf(stderr, "\n");
                }
            }
            w.collectives++;
        }
        if (!any) break;
        // __syncthreads: released when every wave with live lanes is parked
        bool all = true, some = false;
        for (int v = 0; v < waves; v++) {
            bool live = false;
            for (int l = v * kLanes; l < (v + 1) * kLanes; l++) live = live || !w.done[l];
            if (!live) continue;
            some = true;
            all = all && w.at_sync[v];
        }
        if (some && all)
            for (int v = 0; v < waves; v++) {
                w.at_sync[v] = false;
                for (int l = v * kLanes; l < (v + 1) * kLanes; l++) {
                    w.member[l] = !w.done[l];
                    if (!w.done[l]) w.waiting[l] = false;
                }
            }
    }
    current() = saved;
}
// one wavefront
inline void run(const std::function<void(int)>& body, size_t stack_bytes = 1u << 20) { runGroup(1, body, stack_bytes); }

inline int lane() { return current()->cur & 63; }
inline int thread() { return current()->cur; }

// publish v, wait for the wave, return the snapshot of the operands of the lanes that take part (0 for the others; valid until this
// lane's next cross-lane operation). file / line: where the operation stands in the code under test.
inline const uint64_t* exchange(Op op, uint64_t v, const char* file = nullptr, int line = 0) {
    Wave* w = current();
    const int l = w->cur;
    w->pub[l] = v;
    w->op[l] = op;
    w->site_file[l] = file;
    w->site_line[l] = line;
    w->stamp[l] = ++w->clock;
    w->depth[l] = (size_t)(w->stack_top[l] - (uintptr_t)__builtin_frame_address(0));
    w->waiting[l] = true;
    switchTo(w->ctx[l], w->sched);
    w->cur = l;  // (the scheduler set it before resuming; restated for clarity)
    return w->snap + (l & ~63);  // this wave's operands
}
inline bool alive(int l) {  // lane l of the caller's wave takes part in the operation the caller was just released from
    const Wave* w = current();
    const int t = (w->cur & ~63) + l;
    return !w->done[t] && w->member[t];
}

// v_mov_b32_dpp: the source lane of lane i under dpp_ctrl, or -1 when the control has no valid source for it
inline int dppSource(int i, int ctrl) {
    const int row = i & ~15, in_row = i & 15;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) return (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);                  // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) return in_row + (ctrl & 15) < 16 ? i + (ctrl & 15) : -1;          // row_shl:n
    if (ctrl >= 0x111 && ctrl <= 0x11F) return in_row >= (ctrl & 15) ? i - (ctrl & 15) : -1;              // row_shr:n
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((in_row - (ctrl & 15)) & 15);                       // row_ror:n
    if (ctrl == 0x130) return i + 1 < 64 ? i + 1 : -1;                                                    // wave_shl:1
    if (ctrl == 0x134) return (i + 1) & 63;                                                               // wave_rol:1
    if (ctrl == 0x138) return i >= 1 ? i - 1 : -1;                                                        // wave_shr:1
    if (ctrl == 0x13C) return (i - 1) & 63;                                                               // wave_ror:1
    if (ctrl == 0x140) return row | (15 - in_row);                                                        // row_mirror
    if (ctrl == 0x141) return (i & ~7) | (7 - (i & 7));                                                   // row_half_mirror
    if (ctrl == 0x142) return row >= 16 ? row - 1 : -1;                                                   // row_bcast:15 (last lane of the row before)
    if (ctrl == 0x143) return i >= 32 ? 31 : -1;                                                          // row_bcast:31
    fprintf(stderr, "wave_emu: dpp_ctrl 0x%x not modelled\n", ctrl);
    abort();
}

}  // namespace wemu

// ---- what the device code sees -----------------------------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

struct double2 {
    double x, y;
};

inline unsigned __lane_id() { return (unsigned)wemu::lane(); }
inline unsigned long long waveBallot_at(bool p, const char* f_, int l_) {
    const uint64_t* s = wemu::exchange(wemu::kBallot, p ? 1u : 0u, f_, l_);
    unsigned long long m = 0;
    for (int l = 0; l < wemu::kLanes; l++)
        if (s[l] & 1u) m |= 1ull << l;
    return m;
}
inline int wemu_readlane_at(int v, int src, const char* f_, int l_) {
    const uint64_t* s = wemu::exchange(wemu::kReadlane, (uint32_t)v, f_, l_);
    return (int)(uint32_t)s[src & 63];
}
inline int wemu_readfirstlane_at(int v, const char* f_, int l_) {
    const uint64_t* s = wemu::exchange(wemu::kReadfirst, (uint32_t)v, f_, l_);
    for (int l = 0; l < wemu::kLanes; l++)
        if (wemu::alive(l)) return (int)(uint32_t)s[l];
    return v;
}
inline int wemu_update_dpp_at(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, const char* f_, int l_) {
    const uint64_t* s = wemu::exchange(wemu::kDpp, (uint32_t)src, f_, l_);
    const int i = wemu::lane();
    if (!((row_mask >> (i >> 4)) & 1) || !((bank_mask >> ((i & 15) >> 2)) & 1)) return old;  // this lane's row / bank is not written
    const int j = wemu::dppSource(i, ctrl);
    if (j < 0 || !wemu::alive(j)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)s[j];
}
inline void wemu_wave_barrier_at(const char* f_, int l_) { (void)wemu::exchange(wemu::kBarrier, 0, f_, l_); }
inline int wemu_ds_bpermute_at(int byte_addr, int v, const char* f_, int l_) {
    const uint64_t* s = wemu::exchange(wemu::kBpermute, (uint32_t)v, f_, l_);
    return (int)(uint32_t)s[(byte_addr >> 2) & 63];
}
template <class T>
inline T wemu_shfl_xor_at(T v, int mask, int width, const char* f_, int l_) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint64_t* s = wemu::exchange(wemu::kShfl, u, f_, l_);
    const uint32_t r = (uint32_t)s[(wemu::lane() ^ mask) & (width - 1) & 63];
    T out;
    memcpy(&out, &r, 4);
    return out;
}
inline void __threadfence() {}
inline void __threadfence_block() {}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline double __longlong_as_double(long long v) {
    double d;
    memcpy(&d, &v, 8);
    return d;
}
inline long long __double_as_longlong(double d) {
    long long v;
    memcpy(&v, &d, 8);
    return v;
}
inline unsigned __float_as_uint(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
// v_mbcnt_lo / v_mbcnt_hi: set bits of the mask below this lane (low / high half), added to `init`
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask_lo, unsigned init) {
    const int l = wemu::lane();
    return init + (unsigned)__builtin_popcount(l >= 32 ? mask_lo : (mask_lo & ((1u << l) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask_hi, unsigned init) {
    const int l = wemu::lane();
    return init + (l <= 32 ? 0u : (unsigned)__builtin_popcount(mask_hi & ((1u << (l - 32)) - 1u)));
}
#define __builtin_amdgcn_fence(order, scope) ((void)0)

// ---- a workgroup: threadIdx / blockIdx / blockDim / gridDim, __syncthreads, LDS declarations, atomics on memory --------------------
namespace wemu {
struct Launch {
    unsigned block_idx = 0, block_dim = 64, grid_dim = 1;
};
inline Launch& launch() {
    static thread_local Launch l;
    return l;
}
struct ThreadIdxX {
    operator unsigned() const { return (unsigned)thread(); }
};
struct BlockIdxX {
    operator unsigned() const { return launch().block_idx; }
};
struct BlockDimX {
    operator unsigned() const { return launch().block_dim; }
};
struct GridDimX {
    operator unsigned() const { return launch().grid_dim; }
};
}  // namespace wemu
static const struct { wemu::ThreadIdxX x; } threadIdx = {};
static const struct { wemu::BlockIdxX x; } blockIdx = {};
static const struct { wemu::BlockDimX x; } blockDim = {};
static const struct { wemu::GridDimX x; } gridDim = {};
inline void wemu_syncthreads_at(const char* f_, int l_) { (void)wemu::exchange(wemu::kSync, 0, f_, l_); }
#define __shared__ static            // a kernel's static LDS arrays: one copy for the (one) workgroup that runs at a time
#define MCRT_DYNAMIC_LDS(name, alignment)  // ... and its dynamic LDS: the array `lds` the harness defines at namespace scope
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(...)
template <class T, class U>
inline T atomicAdd(T* p, U v) {  // (one OS thread: fibers switch only at cross-lane operations)
    const T old = *p;
    *p = (T)(old + (T)v);
    return old;
}
template <class T>
inline T wemu_shfl_down_at(T v, int delta, int width, const char* f_, int l_) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint64_t* s = wemu::exchange(wemu::kShfl, u, f_, l_);
    const int l = wemu::lane(), src = l + delta;
    const uint32_t r = (src < 64 && (src / width) == (l / width)) ? (uint32_t)s[src] : u;
    T out;
    memcpy(&out, &r, 4);
    return out;
}
inline long long clock64() { return 0; }
struct uint2 {
    unsigned x, y;
};
struct uint4 {
    unsigned x, y, z, w;
};
struct float4 {
    float x, y, z, w;
};
inline void __builtin_amdgcn_s_sleep(int) {}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }

// The operations carry their SOURCE position (what the scheduler groups waiting lanes by): function-like macros over the shims above.
#define waveBallot(p) waveBallot_at((p), __FILE__, __LINE__)
#define __builtin_amdgcn_ballot_w64(p) waveBallot_at((p), __FILE__, __LINE__)
#define __builtin_amdgcn_readlane(v, src) wemu_readlane_at((v), (src), __FILE__, __LINE__)
#define __builtin_amdgcn_readfirstlane(v) wemu_readfirstlane_at((v), __FILE__, __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) wemu_update_dpp_at((old), (src), (ctrl), (rm), (bm), (bc), __FILE__, __LINE__)
#define __builtin_amdgcn_wave_barrier() wemu_wave_barrier_at(__FILE__, __LINE__)
#define __builtin_amdgcn_ds_bpermute(addr, v) wemu_ds_bpermute_at((addr), (v), __FILE__, __LINE__)
#define __syncthreads() wemu_syncthreads_at(__FILE__, __LINE__)
#define __shfl_xor(v, mask, ...) wemu_shfl_xor_at((v), (mask), 64, __FILE__, __LINE__)
#define __shfl_down(v, delta, ...) wemu_shfl_down_at((v), (delta), 64, __FILE__, __LINE__)
