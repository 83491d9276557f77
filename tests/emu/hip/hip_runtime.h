/*
 * tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A lock-step SIMT emulator that lets the *unmodified* kernel source
 * (jssenv_amd/csrc/jss_kernels.hip) be compiled with g++ and executed on the
 * CPU, one fibre per lane, so kernel logic can be checked against the oracle
 * in a container without a GPU.  It is reached only by putting tests/emu on the
 * include path in tests/emu/build_emu.sh; the product library is always built
 * by hipcc against the real <hip/hip_runtime.h> and never sees this file.
 * It is not a fallback: jssenv_amd refuses to run without a GPU.
 *
 * Model: a workgroup runs as blockDim.x fibres (ucontext).  Fibres run until
 * they hit a cross-lane operation, which is a rendezvous of the 64 lanes of the
 * wave (or of the whole workgroup for __syncthreads).  Cross-lane operations
 * must therefore be reached by all 64 lanes -- the same discipline the kernels
 * follow on hardware (wave-uniform control flow around collectives).
 */
#ifndef JSS_EMU_HIP_RUNTIME_H
#define JSS_EMU_HIP_RUNTIME_H

#include <ucontext.h>

#include <cassert>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
typedef void *hipEvent_t;
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
#define hipDeviceAttributeMultiprocessorCount 0
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 2; return 0; }  // a 2-CU "device"
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return 0; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct int4 {
    int x, y, z, w;
};
struct float4 {
    float x, y, z, w;
};
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct int2 {
    int x, y;
};
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
// cache-policy hints have no meaning on the CPU
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_nontemporal_load(p) (*(p))

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

struct WaveSync {
    int arrived = 0;
    unsigned gen = 0;
    uint64_t vals[WAVE];
};

struct Block {
    std::vector<const char *> waiting_in;   // per fibre: the collective it last entered (deadlock reports)
    std::vector<unsigned long long> n_coll; // per fibre: collectives entered so far
    std::vector<std::vector<const char *>> history;  // per fibre: every collective entered, in order
    std::vector<ucontext_t> ctx;
    std::vector<char *> stacks;
    std::vector<char> finished;
    std::vector<WaveSync> waves;
    int block_arrived = 0;
    unsigned block_gen = 0;
    int nthreads = 0;
    int live = 0;
    ucontext_t main_ctx;
    int current = -1;
    std::function<void()> body;
    std::vector<char> dyn_smem;
};

inline Block *&cur_block() {
    static Block *b = nullptr;
    return b;
}

}  // namespace emu

// CUDA-style built-in coordinates: globals rewritten by the scheduler at every fibre switch.
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

inline void yield() {
    Block *b = cur_block();
    int me = b->current;
    swapcontext(&b->ctx[me], &b->main_ctx);
}

inline int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }
inline WaveSync &my_wave() { return cur_block()->waves[threadIdx.x / WAVE]; }

inline void mark(const char *what) {
    cur_block()->waiting_in[threadIdx.x] = what;
    cur_block()->n_coll[threadIdx.x]++;
    static const bool trace = getenv("JSS_EMU_TRACE") != nullptr;   // full per-lane history: opt-in (slow)
    if (trace) cur_block()->history[threadIdx.x].push_back(what);
}

inline void wave_sync() {
    WaveSync &w = my_wave();
    unsigned gen = w.gen;
    if (++w.arrived == WAVE) {
        w.arrived = 0;
        w.gen++;
    } else {
        while (w.gen == gen) yield();
    }
}

inline void block_sync() {
    Block *b = cur_block();
    unsigned gen = b->block_gen;
    if (++b->block_arrived == b->nthreads) {
        b->block_arrived = 0;
        b->block_gen++;
    } else {
        while (b->block_gen == gen) yield();
    }
}

// every lane deposits v; returns after all lanes deposited; `read` runs while the values are stable
template <typename F>
inline auto exchange(uint64_t v, F read) -> decltype(read((const uint64_t *)nullptr)) {
    WaveSync &w = my_wave();
    w.vals[lane_id()] = v;
    wave_sync();
    auto r = read((const uint64_t *)w.vals);
    wave_sync();
    return r;
}

inline void fibre_entry() {
    Block *b = cur_block();
    int me = b->current;
    b->body();
    b->finished[me] = 1;
    b->live--;
    swapcontext(&b->ctx[me], &b->main_ctx);
}

inline void run_block(Block &b, unsigned bx, dim3 grid, dim3 block) {
    cur_block() = &b;
    b.nthreads = (int)block.x;
    assert(b.nthreads % WAVE == 0 && "emulator needs whole waves");
    b.ctx.assign(b.nthreads, ucontext_t());
    b.finished.assign(b.nthreads, 0);
    b.waiting_in.assign(b.nthreads, "-");
    b.n_coll.assign(b.nthreads, 0);
    b.history.assign(b.nthreads, std::vector<const char *>());
    b.waves.assign(b.nthreads / WAVE, WaveSync());
    b.block_arrived = 0;
    b.live = b.nthreads;
    if ((int)b.stacks.size() < b.nthreads) {
        for (int i = (int)b.stacks.size(); i < b.nthreads; ++i) b.stacks.push_back((char *)malloc(STACK_BYTES));
    }
    for (int i = 0; i < b.nthreads; ++i) {
        getcontext(&b.ctx[i]);
        b.ctx[i].uc_stack.ss_sp = b.stacks[i];
        b.ctx[i].uc_stack.ss_size = STACK_BYTES;
        b.ctx[i].uc_link = &b.main_ctx;
        makecontext(&b.ctx[i], (void (*)())fibre_entry, 0);
    }
    gridDim = grid;
    blockDim = block;
    blockIdx = dim3(bx, 0, 0);
    unsigned long long passes = 0;
    while (b.live > 0) {
        if (++passes == 2000000ULL) {   // no kernel of the suite needs this many scheduler passes: lanes of a wave
                                        // are waiting in DIFFERENT collectives (divergent control flow around one)
            fprintf(stderr, "emu: deadlock in block %u -- collective each lane last entered (rerun with JSS_EMU_TRACE=1 for the "
                            "first diverging collective per lane):\n", bx);
            for (int i = 0; i < b.nthreads; ++i)
                fprintf(stderr, "  lane %3d: %-32s collectives entered %llu%s\n", i, b.waiting_in[i], b.n_coll[i],
                        b.finished[i] ? "  (finished)" : "");
            for (int w = 0; w < b.nthreads / WAVE; ++w)      // first point where a lane's sequence leaves lane 0's
                for (int i = 1; i < WAVE; ++i) {
                    const auto &a = b.history[w * WAVE], &c = b.history[w * WAVE + i];
                    size_t k = 0;
                    while (k < a.size() && k < c.size() && a[k] == c[k]) ++k;
                    if (k < a.size() || k < c.size())
                        fprintf(stderr, "  wave %d lane %d diverges from lane 0 at collective #%zu: %s vs %s (previous: %s)\n", w, i, k,
                                k < c.size() ? c[k] : "<end>", k < a.size() ? a[k] : "<end>", k ? a[k - 1] : "-");
                    if (k < a.size() || k < c.size()) {
                        fprintf(stderr, "    lane 0 :");
                        for (size_t q = 0; q < a.size() && q < k + 8; ++q) fprintf(stderr, " %s", a[q] + (a[q][0] == '_' ? 2 : 0));
                        fprintf(stderr, "\n    lane %d:", i);
                        for (size_t q = 0; q < c.size() && q < k + 8; ++q) fprintf(stderr, " %s", c[q] + (c[q][0] == '_' ? 2 : 0));
                        fprintf(stderr, "\n");
                    }
                }
            abort();
        }
        for (int i = 0; i < b.nthreads; ++i) {
            if (b.finished[i]) continue;
            b.current = i;
            threadIdx = dim3((unsigned)i, 0, 0);
            swapcontext(&b.main_ctx, &b.ctx[i]);
        }
    }
}

inline Block &the_block() {
    static Block b;
    return b;
}

}  // namespace emu

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    emu::Block &b = emu::the_block();
    b.dyn_smem.assign(shmem + 64, (char)0x5A);   // LDS is not zeroed on hardware: poison it
    b.body = [=]() { kernel(args...); };
    for (unsigned bx = 0; bx < grid.x; ++bx) emu::run_block(b, bx, grid, block);
}

#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(emu::cur_block()->dyn_smem.data());

// ---- workgroup / wave synchronisation ------------------------------------------------
inline void __syncthreads() { emu::mark("__syncthreads"); emu::block_sync(); }
inline void __builtin_amdgcn_wave_barrier() { emu::mark("wave_barrier"); emu::wave_sync(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_s_barrier() { emu::block_sync(); }

// ---- cross-lane ----------------------------------------------------------------------
inline unsigned long long __ballot(int pred) {
    emu::mark("__ballot");
    return emu::exchange((uint64_t)(pred != 0), [](const uint64_t *v) {
        unsigned long long m = 0;
        for (int i = 0; i < emu::WAVE; ++i) m |= (unsigned long long)(v[i] & 1) << i;
        return m;
    });
}
inline int __shfl(int v, int src, int width = 64) {
    emu::mark("__shfl");
    (void)width;
    return emu::exchange((uint64_t)(uint32_t)v, [=](const uint64_t *vals) { return (int)(uint32_t)vals[src & 63]; });
}
inline int __shfl_xor(int v, int mask, int width = 64) {
    emu::mark("__shfl_xor");
    (void)width;
    int me = emu::lane_id();
    return emu::exchange((uint64_t)(uint32_t)v, [=](const uint64_t *vals) { return (int)(uint32_t)vals[(me ^ mask) & 63]; });
}
inline int __builtin_amdgcn_readlane(int v, int lane) {
    emu::mark("__builtin_amdgcn_readlane");
    return emu::exchange((uint64_t)(uint32_t)v, [=](const uint64_t *vals) { return (int)(uint32_t)vals[lane & 63]; });
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    emu::mark("__builtin_amdgcn_readfirstlane");
    return emu::exchange((uint64_t)(uint32_t)v, [=](const uint64_t *vals) { return (int)(uint32_t)vals[0]; });
}
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v) {
    emu::mark("__builtin_amdgcn_ds_bpermute");
    return emu::exchange((uint64_t)(uint32_t)v,
                         [=](const uint64_t *vals) { return (int)(uint32_t)vals[(byte_addr >> 2) & 63]; });
}
/* ds_swizzle, bit-mask mode (offset bit 15 = 0): within each group of 32 lanes,
 * src = ((lane & and_mask) | or_mask) ^ xor_mask. */
inline int __builtin_amdgcn_ds_swizzle(int v, int pattern) {
    emu::mark("__builtin_amdgcn_ds_swizzle");
    int me = emu::lane_id();
    if (pattern & 0x8000) {
        fprintf(stderr, "emu: ds_swizzle quad-perm mode unsupported\n");
        abort();
    }
    int and_mask = pattern & 31, or_mask = (pattern >> 5) & 31, xor_mask = (pattern >> 10) & 31;
    int from = (me & 32) | ((((me & 31) & and_mask) | or_mask) ^ xor_mask);
    return emu::exchange((uint64_t)(uint32_t)v, [=](const uint64_t *vals) { return (int)(uint32_t)vals[from]; });
}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) {
    int l = emu::lane_id();
    unsigned m = l >= 32 ? mask : (mask & ((1u << l) - 1u));
    return base + (unsigned)__builtin_popcount(m);
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) {
    int l = emu::lane_id();
    unsigned m = l <= 32 ? 0u : (mask & ((1u << (l - 32)) - 1u));
    return base + (unsigned)__builtin_popcount(m);
}

/* DPP (gfx9 encodings, cdna4 ISA "DPP_CTRL"): the subset the kernels use.
 * bound_ctrl=false: an invalid source lane or a masked-off row/bank keeps `old`. */
inline int __builtin_amdgcn_update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    emu::mark("__builtin_amdgcn_update_dpp");
    int me = emu::lane_id();
    return emu::exchange((uint64_t)(uint32_t)src, [=](const uint64_t *vals) {
        int row = me / 16, in_row = me % 16, bank = in_row / 4;
        if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
        int from = -1;
        if (dpp_ctrl >= 0x000 && dpp_ctrl <= 0x0FF) {  // quad_perm
            int sel = (dpp_ctrl >> (2 * (me & 3))) & 3;
            from = (me & ~3) | sel;
        } else if (dpp_ctrl >= 0x101 && dpp_ctrl <= 0x10F) {  // row_shl:n  (lane i reads i+n)
            int s = in_row + (dpp_ctrl & 15);
            from = s < 16 ? row * 16 + s : -1;
        } else if (dpp_ctrl >= 0x111 && dpp_ctrl <= 0x11F) {  // row_shr:n  (lane i reads i-n)
            int s = in_row - (dpp_ctrl & 15);
            from = s >= 0 ? row * 16 + s : -1;
        } else if (dpp_ctrl >= 0x121 && dpp_ctrl <= 0x12F) {  // row_ror:n
            from = row * 16 + ((in_row - (dpp_ctrl & 15)) & 15);
        } else if (dpp_ctrl == 0x140) {  // row_mirror
            from = row * 16 + (15 - in_row);
        } else if (dpp_ctrl == 0x141) {  // row_half_mirror
            from = (me & ~7) | (7 - (me & 7));
        } else if (dpp_ctrl == 0x142) {  // row_bcast:15 -> lane 15 of each row to the next row
            from = row > 0 ? (row - 1) * 16 + 15 : -1;
        } else if (dpp_ctrl == 0x143) {  // row_bcast:31 -> lane 31 to rows 2 and 3
            from = row >= 2 ? 31 : -1;
        } else {
            fprintf(stderr, "emu: unsupported dpp_ctrl 0x%x\n", dpp_ctrl);
            abort();
        }
        if (from < 0) return bound_ctrl ? 0 : old;
        return (int)(uint32_t)vals[from];
    });
}

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    unsigned long long old = *p;
    *p = old + v;
    return old;
}

inline int atomicAdd(int *p, int v) {
    int old = *p;
    *p = old + v;
    return old;
}
// agent-scope atomics of the step-session protocol: one fibre runs at a time, plain accesses are exact
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
inline void __builtin_amdgcn_s_sleep(int) {}
// the 100 MHz wall clock: here a counter that advances 1 ms per reading, so that every bounded wait of the kernels
// runs out after a few thousand polls instead of seconds
inline long long wall_clock64() {
    static long long t = 0;
    return t += 100000;
}
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 8; return 0; }

// ---- scalar helpers that hip_runtime.h provides as device functions --------------------
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

#endif
