// Per-cell stencil arithmetic shared by the advection and compressible
// kernels.  Every expression keeps the reference's operation order so that,
// compiled with -ffp-contract=off, results are bit-identical to NumPy/numba
// (SURVEY.md 8(a) "arithmetic-order rules").
#pragma once
#include <hip/hip_runtime.h>

#ifndef PYRO_FAST
#define PYRO_FAST 0
#endif

namespace pyro {

// MC limiter building block, pyro/mesh/reconstruction.py:84-91 / 113-120
__device__ __forceinline__ double mc_select(double dc, double dl, double dr)
{
    double d1 = 2.0 * ((fabs(dl) < fabs(dr)) ? dl : dr);
    double dt = (fabs(dc) < fabs(d1)) ? dc : d1;
    return (dl * dr > 0.0) ? dt : 0.0;
}

// limit2 (reconstruction.py:69-91) at the centre of (am, a0, ap)
__device__ __forceinline__ double limit2(double am, double a0, double ap)
{
    double dc = 0.5 * (ap - am);
    double dl = ap - a0;
    double dr = a0 - am;
    // where dl*dr > 0 the three candidates share their sign (dc is their
    // mean), so the selects by magnitude of mc_select are two minima of
    // absolute values (v_min_f64 with |.| operand modifiers) and a sign copy:
    // the SAME candidate is chosen, bit for bit (ties have equal values), at 8
    // instead of 11 VALU instructions (3 compares + 6 v_cndmask_b32 there)
    const double m = fmin(fabs(dl), fabs(dr));
    const double r = copysign(fmin(fabs(dc), m + m), dl);
    return (dl * dr > 0.0) ? r : 0.0;
}

// mc_select(dc, dl, dr) for the FOURTH-ORDER centred slope of limit4,
// dc = (2/3) (ap1 - am1 - 0.25 (l2p + l2m)) with l2p / l2m the limit2 slopes of the two
// neighbours, as two minima of magnitudes and a sign copy (like limit2 above).  Where
// dl * dr > 0 this dc shares the sign of dl and dr: |l2p| <= 2 |dl| and |l2m| <= 2 |dr|
// (as computed: m + m of the rounded differences), so 0.25 (l2p + l2m) stays below
// 0.5 (1 + eps) (|dl| + |dr|) and can neither turn the sign of ap1 - am1 nor cancel it.
// The SAME candidate comes out, bit for bit, at 8 instead of 11 vector instructions
// (3 compares + 6 v_cndmask_b32 in mc_select).
__device__ __forceinline__ double mc_select_l4(double dc, double dl, double dr)
{
    const double m = fmin(fabs(dl), fabs(dr));
    const double r = copysign(fmin(fabs(dc), m + m), dl);
    return (dl * dr > 0.0) ? r : 0.0;
}

// limited slope at the centre of the 5-point stencil (reconstruction.py:9-120)
//   limiter 0: nolimit, 1: limit2, otherwise limit4
__device__ __forceinline__ double limited_slope(double am2, double am1, double a0, double ap1,
                                                double ap2, int limiter)
{
    if (limiter == 0) return 0.5 * (ap1 - am1);
    if (limiter == 1) return limit2(am1, a0, ap1);
    double l2p = limit2(a0, ap1, ap2);
    double l2m = limit2(am2, am1, a0);
    double dc = (2. / 3.) * (ap1 - am1 - 0.25 * (l2p + l2m));
    double dl = ap1 - a0;
    double dr = a0 - am1;
    return mc_select_l4(dc, dl, dr);
}

// limit4 at the centre of (am1, a0, ap1) from the limit2 slopes of the two neighbours -- the
// expression above with l2m = limit2(am2, am1, a0), l2p = limit2(a0, ap1, ap2) handed in: a
// marching kernel computes every cell's centred limit2 ONCE and takes the neighbours' from the
// row window / the neighbouring lanes (same function, same operands: the same bits)
__device__ __forceinline__ double limit4_from(double l2m, double l2p, double am1, double a0, double ap1)
{
    const double dc = (2. / 3.) * (ap1 - am1 - 0.25 * (l2p + l2m));
    return mc_select_l4(dc, ap1 - a0, a0 - am1);
}

// blockIdx -> logical tile id such that each XCD (blocks are dealt round-robin
// to the 8 XCDs, MI355X_MICROARCH.md "block b runs on XCD b % 8") works on a
// contiguous band of tiles and halo re-reads hit its own L2.  Performance
// only: any mapping is correct.
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    const int nxcd = 8;
    if (nb % nxcd != 0) return b;
    return (b % nxcd) * (nb / nxcd) + b / nxcd;
}

// 2-d version for row-per-block kernels launched as a 1-d grid of
// 8*ceil(gx*gy/8) blocks: returns false for the padding blocks, otherwise the
// logical (bx, by) with by = row index.  XCD k then owns a contiguous band of
// rows, walked in order, so the +-1..4 neighbour rows of a stencil are L2 hits
// instead of being re-fetched by every XCD (measured 4-7x FETCH_SIZE
// amplification without it, profiles/r01_*).
__device__ __forceinline__ bool xcd_block_2d(int gx, int gy, int &bx, int &by)
{
    const int N = gx * gy;
    const int per = (N + 7) / 8;
    const int L = blockIdx.x;
    const int Lp = (L % 8) * per + L / 8;
    if (Lp >= N) return false;
    bx = Lp % gx;
    by = Lp / gx;
    return true;
}
inline int xcd_grid_1d(int gx, int gy) { return 8 * ((gx * gy + 7) / 8); }

// Ghost fill as an index map (array_indexer.py:163-274 / k_fill_x, k_fill_y of
// ctx.hip): the source of array index i is a * i + b with (a, b) per side --
// outflow (0, edge), reflect (-1, mirror), periodic (1, shift); interior and
// other boundary types map to themselves.  Branch-free, so that the row map in
// the marching loop stays on the scalar unit.
// a value that is the same in every lane of the wavefront, moved to a scalar register
#if !defined(PYRO_EMU)
__device__ __forceinline__ int pyro_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
__device__ __forceinline__ int pyro_uniform(int v) { return v; }
#endif

struct BcMap { int alo, blo, ahi, bhi; bool odd_lo, odd_hi; };
__host__ __device__ inline BcMap bc_map(int lo, int hi, int ng, int bl, int br, bool fill)
{
    BcMap m{1, 0, 1, 0, false, false};
    if (!fill) return m;
    if (bl == PYROHIP_BC_OUTFLOW) { m.alo = 0; m.blo = lo; }
    else if (bl == PYROHIP_BC_REFLECT_EVEN || bl == PYROHIP_BC_REFLECT_ODD) { m.alo = -1; m.blo = 2 * ng - 1; }
    else if (bl == PYROHIP_BC_PERIODIC) { m.alo = 1; m.blo = hi - ng + 1; }
    if (br == PYROHIP_BC_OUTFLOW) { m.ahi = 0; m.bhi = hi; }
    else if (br == PYROHIP_BC_REFLECT_EVEN || br == PYROHIP_BC_REFLECT_ODD) { m.ahi = -1; m.bhi = 2 * hi + 1; }
    else if (br == PYROHIP_BC_PERIODIC) { m.ahi = 1; m.bhi = ng - hi - 1; }
    m.odd_lo = (bl == PYROHIP_BC_REFLECT_ODD);
    m.odd_hi = (br == PYROHIP_BC_REFLECT_ODD);
    return m;
}
__device__ __forceinline__ int bc_src(const BcMap &m, int i, int lo, int hi)
{
    return i < lo ? m.alo * i + m.blo : (i > hi ? m.ahi * i + m.bhi : i);
}


}  // namespace pyro
