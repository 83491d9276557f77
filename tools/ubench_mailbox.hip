// tools/ubench_mailbox.hip -- micro-benchmark of the step-session hand-off protocol (DESIGN "persistent step session"):
// a resident kernel whose waves wait for per-env action granules {seq << 32 | action} in a device mailbox, do a step's
// worth of nothing, publish an output write-through and bump a per-wave progress word; a post kernel and a wait kernel on
// the CALLER's stream feed / drain it.  Measures (a) the ping-pong latency per step (post -> session -> wait) and (b) the
// free-running rate with all K steps posted up front, for several grid sizes, and checks every value that comes back.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_mailbox tools/ubench_mailbox.hip && ./ubench_mailbox
// Every spin is bounded (wall clock): nothing here can hang the GPU for longer than a second.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned;

__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr long long kTimeoutTicks = 50'000'000;   // 0.5 s of the 100 MHz wall clock

// session: one env group per wave (4 "envs" per wave like the packed kernel: lanes 0,16,32,48 own a granule each)
__global__ __launch_bounds__(256, 8) void session(const unsigned long long *mail, int depth, int n_env, int K, int work,
                                                  int *out, int *progress, int *status) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env = gw * 4 + (lane >> 4);
    const bool owner = (lane & 15) == 0 && env < n_env;
    int acc = 0;
    for (int s = 1; s <= K; ++s) {
        const unsigned long long *g = mail + (size_t)((s - 1) % depth) * n_env + (owner ? env : 0);
        unsigned long long x = 0;
        const long long t0 = wall_clock64();
        for (;;) {
            x = owner ? ld_granule(g) : ((unsigned long long)s << 32);
            if (__all((unsigned)(x >> 32) == (unsigned)s)) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > kTimeoutTicks) {
                if (lane == 0) atomicAdd(status, 1);
                return;
            }
        }
        int a = (int)(unsigned)x;
        for (int i = 0; i < work; ++i) a = a * 1664525 + 1013904223 + acc;   // dependent chain: a step's compute
        acc = a;
        if (owner) st_wt(out + env, a);                                       // write-through output
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_wt(progress + gw, s);
    }
}

__global__ void post(unsigned long long *mail, int depth, int n_env, int s, int n_steps, const int *actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    for (int k = 0; k < n_steps; ++k)
        st_granule(mail + (size_t)((s + k - 1) % depth) * n_env + i, ((unsigned long long)(s + k) << 32) | (unsigned)actions[(size_t)k * n_env + i]);
}

// one workgroup sweeps the progress words until every wave has finished step s
__global__ void wait_step(const int *progress, int n_waves, int s, int *status) {
    __shared__ int ok;
    const long long t0 = wall_clock64();
    for (;;) {
        if (threadIdx.x == 0) ok = 1;
        __syncthreads();
        int mine = 1;
        for (int i = threadIdx.x; i < n_waves; i += blockDim.x)
            if (__hip_atomic_load(progress + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s) mine = 0;
        if (!mine) ok = 0;
        __syncthreads();
        if (ok) return;
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // the session gave up
        if (wall_clock64() - t0 > kTimeoutTicks) {
            if (threadIdx.x == 0) atomicAdd(status + 1, 1);
            return;
        }
        __syncthreads();
    }
}

__global__ void check(const int *out, const int *want, int n, int *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && out[i] != want[i]) atomicAdd(bad, 1);
}

static int host_chain(int a, int acc, int work) {
    for (int i = 0; i < work; ++i) a = (int)((unsigned)a * 1664525u + 1013904223u + (unsigned)acc);
    return a;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 200;
    const int work = argc > 2 ? atoi(argv[2]) : 600;
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    int can_wait = 0;
    hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    for (int n_env : {4096, 16384, 28672, 32768}) {       // 1024, 4096, 7168, 8192 waves (8192 = every wave slot of the chip)
        const int n_waves = (n_env + 3) / 4, blocks = (n_waves + 3) / 4, depth = K;
        unsigned long long *mail;
        int *out, *progress, *status, *actions, *want, *bad;
        CHECK(hipMalloc(&mail, sizeof(*mail) * (size_t)depth * n_env));
        CHECK(hipMalloc(&out, 4 * n_env));
        CHECK(hipMalloc(&want, 4 * n_env));
        CHECK(hipMalloc(&progress, 4 * n_waves));
        CHECK(hipMalloc(&status, 16));
        CHECK(hipMalloc(&bad, 4));
        CHECK(hipMalloc(&actions, 4 * (size_t)K * n_env));
        std::vector<int> h_act((size_t)K * n_env), h_want(n_env);
        for (size_t i = 0; i < h_act.size(); ++i) h_act[i] = (int)(i * 2654435761u >> 7);
        for (int e = 0; e < n_env; ++e) {            // every lane of a wave runs the chain on ITS value: owner lanes carry the env's
            int acc = 0;
            for (int s = 0; s < K; ++s) acc = host_chain(h_act[(size_t)s * n_env + e], acc, work);
            h_want[e] = acc;
        }
        CHECK(hipMemcpy(actions, h_act.data(), 4 * h_act.size(), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(want, h_want.data(), 4 * n_env, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode) {       // 0: ping-pong (post s, wait s), 1: everything posted up front
            CHECK(hipMemset(mail, 0, sizeof(*mail) * (size_t)depth * n_env));
            CHECK(hipMemset(progress, 0, 4 * n_waves));
            CHECK(hipMemset(status, 0, 16));
            CHECK(hipMemset(bad, 0, 4));
            CHECK(hipMemset(out, 0, 4 * n_env));
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(session, dim3(blocks), dim3(256), 0, sa, mail, depth, n_env, K, work, out, progress, status);
            if (mode == 0) {
                for (int s = 1; s <= K; ++s) {
                    hipLaunchKernelGGL(post, dim3((n_env + 255) / 256), dim3(256), 0, sb, mail, depth, n_env, s, 1, actions + (size_t)(s - 1) * n_env);
                    hipLaunchKernelGGL(wait_step, dim3(1), dim3(256), 0, sb, progress, n_waves, s, status);
                }
            } else {
                hipLaunchKernelGGL(post, dim3((n_env + 255) / 256), dim3(256), 0, sb, mail, depth, n_env, 1, K, actions);
                hipLaunchKernelGGL(wait_step, dim3(1), dim3(256), 0, sb, progress, n_waves, K, status);
            }
            CHECK(hipStreamSynchronize(sb));
            const auto t1 = std::chrono::steady_clock::now();
            CHECK(hipStreamSynchronize(sa));
            hipLaunchKernelGGL(check, dim3((n_env + 255) / 256), dim3(256), 0, sb, out, want, n_env, bad);
            int h_status[4] = {}, h_bad = 0;
            CHECK(hipMemcpy(h_status, status, 16, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
            const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            printf("envs %6d waves %5d  %-10s K %d work %d: %9.1f us total, %7.2f us/step  timeouts session %d wait %d  wrong values %d\n",
                   n_env, n_waves, mode == 0 ? "ping-pong" : "pre-posted", K, work, us, us / K, h_status[0], h_status[1], h_bad);
            fflush(stdout);
        }
        hipFree(mail); hipFree(out); hipFree(want); hipFree(progress); hipFree(status); hipFree(bad); hipFree(actions);
    }
    return 0;
}
