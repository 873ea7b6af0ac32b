// Minimal single-threaded HIP emulator -- TEST INFRASTRUCTURE ONLY.
//
// Lets the -m "not gpu" test-suite execute the *same kernel source* that
// hipcc compiles for gfx950 on the build container's CPU (which has no GPU),
// so indexing / arithmetic bugs are found before GPU minutes are spent.  It is
// never shipped: the product loader (pyro2_amd/_lib.py) only accepts a
// library whose pyrohip_backend() is "hip-gfx950"; the emulated library
// reports "host-emu" and can only be injected explicitly by a test fixture.
//
// Model: blocks run one after another; the threads of a block are ucontext
// fibers scheduled round-robin.  __syncthreads() and the wave64 shuffles
// yield until every live fiber of the block / wave has arrived.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
#define __forceinline__ inline
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)::hipemu::dyn_smem();

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
struct alignas(16) double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef void *hipStream_t;
struct hipEvent_s { std::chrono::steady_clock::time_point t; };
typedef hipEvent_s *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; };

namespace hipemu {
struct Fiber {
    ucontext_t ctx;        // generic fallback
    void *sp = nullptr;    // x86-64: saved stack pointer of the hand-written switch
    char *stack = nullptr;
    bool done = false;
    unsigned long bar_gen_seen = 0;
};
struct BlockState {
    dim3 grid, block;
    uint3_ bidx;
    int nthreads = 0, nlive = 0;
    int cur = 0;
    std::vector<Fiber> fib;
    ucontext_t sched;
    void *sched_sp = nullptr;
    // block barrier
    int bar_arrived = 0;
    unsigned long bar_gen = 0;
    // per-wave barrier + shuffle buffer
    std::vector<int> wave_arrived, wave_live;
    std::vector<unsigned long> wave_gen;
    std::vector<double> shfl;
    const std::function<void()> *body = nullptr;
    std::vector<char> dyn;
};
BlockState &bs();
void *dyn_smem();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void yield_();
void block_barrier();
void wave_barrier();
double shfl_read(double v, int src_lane);
uint3_ tidx();
uint3_ bidx_();
dim3 bdim();
dim3 gdim();
}  // namespace hipemu

// threadIdx etc. must be re-read after every yield: make them proxies.
struct hipemu_tid_proxy { struct P { operator unsigned() const; int which; }; P x{0}, y{1}, z{2}; };
struct hipemu_bid_proxy { struct P { operator unsigned() const; int which; }; P x{0}, y{1}, z{2}; };
struct hipemu_bdim_proxy { struct P { operator unsigned() const; int which; }; P x{0}, y{1}, z{2}; };
struct hipemu_gdim_proxy { struct P { operator unsigned() const; int which; }; P x{0}, y{1}, z{2}; };
inline hipemu_tid_proxy::P::operator unsigned() const { auto t = hipemu::tidx(); return which == 0 ? t.x : which == 1 ? t.y : t.z; }
inline hipemu_bid_proxy::P::operator unsigned() const { auto t = hipemu::bidx_(); return which == 0 ? t.x : which == 1 ? t.y : t.z; }
inline hipemu_bdim_proxy::P::operator unsigned() const { auto t = hipemu::bdim(); return which == 0 ? t.x : which == 1 ? t.y : t.z; }
inline hipemu_gdim_proxy::P::operator unsigned() const { auto t = hipemu::gdim(); return which == 0 ? t.x : which == 1 ? t.y : t.z; }
static const hipemu_tid_proxy threadIdx;
static const hipemu_bid_proxy blockIdx;
static const hipemu_bdim_proxy blockDim;
static const hipemu_gdim_proxy gridDim;

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ::hipemu::launch((grid), (block), (shmem), [&]() { kern(__VA_ARGS__); })

using std::max;
using std::min;
inline void __syncthreads() { hipemu::block_barrier(); }
inline double __shfl_down(double v, unsigned delta, int width = 64)
{
    auto t = hipemu::tidx(); auto b = hipemu::bdim();
    int tid = t.x + b.x * (t.y + b.y * t.z);
    int lane = tid & 63;
    int src = lane + (int)delta;
    if ((src / width) != (lane / width)) src = lane;  // out of the sub-group: own value
    return hipemu::shfl_read(v, src);
}
inline double __shfl_up(double v, unsigned delta, int width = 64)
{
    auto t = hipemu::tidx(); auto b = hipemu::bdim();
    int tid = t.x + b.x * (t.y + b.y * t.z);
    int lane = tid & 63;
    int src = lane - (int)delta;
    if (src < 0 || (src / width) != (lane / width)) src = lane;
    return hipemu::shfl_read(v, src);
}
inline double __shfl(double v, int src_lane, int width = 64)
{
    auto t = hipemu::tidx(); auto b = hipemu::bdim();
    int tid = t.x + b.x * (t.y + b.y * t.z);
    int lane = tid & 63;
    int src = (lane / width) * width + (src_lane % width);
    return hipemu::shfl_read(v, src);
}
inline double __shfl_xor(double v, int mask, int width = 64)
{
    auto t = hipemu::tidx(); auto b = hipemu::bdim();
    int tid = t.x + b.x * (t.y + b.y * t.z);
    int lane = tid & 63;
    (void)width;
    return hipemu::shfl_read(v, lane ^ mask);
}
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
inline void __threadfence() {}
inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long l) { double r; memcpy(&r, &l, 8); return r; }

// ---- runtime API -----------------------------------------------------------
inline const char *hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess" : "hip-emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    strcpy(p->name, "host-emu"); strcpy(p->gcnArchName, "none"); p->multiProcessorCount = 4;
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)1 << 34; *t = (size_t)1 << 34; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n)
{
    if (posix_memalign(p, 256, n ? n : 256) != 0) return hipErrorOutOfMemory;
    memset(*p, 0xCD, n);  // poison: reading uninitialised device memory shows up as NaN-ish garbage
    return hipSuccess;
}
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipEvent_s(); return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
