// Block reductions for CDNA4: wave64 shuffle tree, then one LDS slot per wave.
// min/max are order independent, so the results are bit-exact vs NumPy.
#pragma once
#include <hip/hip_runtime.h>

namespace pyro {

constexpr int kWave = 64;  // gfx950 wavefront

__device__ inline double wave_reduce_min(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, kWave));
    return v;
}
__device__ inline double wave_reduce_max(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, kWave));
    return v;
}
__device__ inline double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// All threads of the block must call.  Result valid in thread 0.
// blockDim (flattened) must be a multiple of 64 and <= 1024.
template <int OP>  // 0 min, 1 max, 2 sum
__device__ inline double block_reduce(double v)
{
    __shared__ double slot[16];
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nthr = blockDim.x * blockDim.y * blockDim.z;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    v = (OP == 0) ? wave_reduce_min(v) : (OP == 1) ? wave_reduce_max(v) : wave_reduce_sum(v);
    __syncthreads();  // protect slot[] against a previous call
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    if (wave == 0) {
        const int nw = nthr >> 6;
        double w = (lane < nw) ? slot[lane] : (OP == 0 ? INFINITY : OP == 1 ? -INFINITY : 0.0);
        w = (OP == 0) ? wave_reduce_min(w) : (OP == 1) ? wave_reduce_max(w) : wave_reduce_sum(w);
        v = w;
    }
    return v;
}
__device__ inline double block_reduce_min(double v) { return block_reduce<0>(v); }
__device__ inline double block_reduce_max(double v) { return block_reduce<1>(v); }
__device__ inline double block_reduce_sum(double v) { return block_reduce<2>(v); }

// Minimum of nb per-workgroup partials (one per tile: 640k at 16384^2).
// One workgroup reading them all costs ~1 ms there, so: 256 workgroups reduce
// strided chunks into `stage`, then one workgroup finishes.
static __global__ void k_min_stage(const double *__restrict__ partial, int nb,
                                   double *__restrict__ stage)
{
    double m = INFINITY;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x)
        m = fmin(m, partial[b]);
    m = block_reduce_min(m);
    if (threadIdx.x == 0) stage[blockIdx.x] = m;
}

// ... up to 32k partials (the row-marching kernels: one per wavefront) in ONE launch: a single
// workgroup of 1024 threads, four loads in flight each -- a launch and its gap less per step
// where the minimum cannot ride on the next policy kernel (slabs: it is all-reduced first)
static __global__ __launch_bounds__(1024) void k_min_one(const double *__restrict__ partial, int nb,
                                                         double *__restrict__ out)
{
    double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
    int b = threadIdx.x;
    for (; b + 3 * 1024 < nb; b += 4 * 1024) {
        m0 = fmin(m0, partial[b]); m1 = fmin(m1, partial[b + 1024]);
        m2 = fmin(m2, partial[b + 2048]); m3 = fmin(m3, partial[b + 3072]);
    }
    for (; b < nb; b += 1024) m0 = fmin(m0, partial[b]);
    const double m = block_reduce_min(fmin(fmin(m0, m1), fmin(m2, m3)));
    if (threadIdx.x == 0) *out = m;
}

constexpr int kMinStageBlocks = 256;
// scratch needed behind the nb partials: kMinStageBlocks + 1 doubles; the
// result lands in part[nb + kMinStageBlocks]
inline double *launch_min_reduce(hipStream_t stream, double *part, int nb)
{
    double *stage = part + nb, *out = part + nb + kMinStageBlocks;
    if (nb > 4 * kMinStageBlocks && nb <= 32768) {
        hipLaunchKernelGGL(k_min_one, dim3(1), dim3(1024), 0, stream, (const double *)part, nb, out);
    } else if (nb > 4 * kMinStageBlocks) {
        hipLaunchKernelGGL(k_min_stage, dim3(kMinStageBlocks), dim3(256), 0, stream,
                           (const double *)part, nb, stage);
        hipLaunchKernelGGL(k_min_stage, dim3(1), dim3(256), 0, stream, (const double *)stage,
                           kMinStageBlocks, out);
    } else {
        hipLaunchKernelGGL(k_min_stage, dim3(1), dim3(256), 0, stream, (const double *)part, nb,
                           out);
    }
    return out;
}

}  // namespace pyro
