// Row-marching red-black smoother for the large multigrid levels (mg_march.hip),
// launched from multigrid.hip's mg_smooth_tiles.
#pragma once
#include "common.h"

namespace pyro {

constexpr int MGM_COLS = 128;   // columns of a strip: two per lane of one wavefront

struct MGMarch {
    const double *vin, *f;
    double *vout;
    int n, pitch;
    double xc, yc, denom, rdenom;   // as MGTile (multigrid.hip)
    double kx, ky;
    int code[4];                    // boundary codes xl xr yl yr (homogeneous)
    const double *cv;               // up leg: coarse solution to prolong and add while loading
    int cpitch;
    int vin_zero;
    int row0, row1;                 // rows to update (1 .. n: the whole level; a slab of it
                                    // when decomposed: the rows beyond are halo rows)
    int TJ, ncs;                    // columns a strip updates, strips
    int CR, nchunks;                // rows a chunk updates, chunks
    // the first and the last strip (physical sides left / right: a select more per update)
    // in shorter chunks of their own, so that their wavefronts take as long as the others;
    // nchunks_side == 0: no such strips, every strip in chunks of CR rows
    int CR_side, nchunks_side;
    // what rides on the tail of the march (whole-level launches only): the rows a wavefront
    // has finished are still in its registers when their neighbours above become final
    //   1  down leg: residual of the smoothed level, restricted into the next coarser level's
    //      right-hand side cf (k_mg_residual_restrict's arithmetic and order; r is not stored)
    //   2  last launch of a solve cycle on the finest level: the sum of r^2 (MG.py:668-671), a
    //      partial per wavefront (partial[nblocks + b]; partial[b] = 0: the relative change of
    //      the solution is taken once, after the solve's last cycle; k_sum_final2 finishes)
    int tail;
    double alpha, beta, dx2, rdx2, small;
    double *cf; int cfpitch;
    const double *old; double *partial;
};

// red-black iterations per launch: 10, a whole V-cycle leg in one pass over the level.  (A
// 5-iteration instance -- half the apron, half the window, three wavefronts per SIMD -- was
// measured on the 1024^2 / 2048^2 levels: 25 / 54 us per launch, no better than the band
// kernel's 25 / 30 us.)  Window rows prefetched ahead: 2 (4: 5 % slower, 16 more registers).
constexpr int MGM_PF = 2;
constexpr bool mgm_has_k(int K) { return K == 10; }
// columns a strip stores (the rest: the apron of 2K sweeps, one more column for parity)
// (a launch with tail 2: four less, mg_march.hip: mgm_part)
constexpr int mgm_tj(int K, int tail = 0) { return MGM_COLS - 4 * K - 2 - (tail == 2 ? 4 : 0); }
// rows by which a part that ends at the top boundary may start lower (window rows - 2)
constexpr int mgm_align(int K) { return 2 * K + MGM_PF; }

int mg_march_blocks(const MGMarch &A);
int mg_march_launch(pyrohip_ctx *c, MGMarch &A, bool pow2, int K);
int mg_march_launch_tail1(pyrohip_ctx *c, MGMarch &A, bool pow2);   // MGMarch::tail (own compile units)
int mg_march_launch_tail2(pyrohip_ctx *c, MGMarch &A, bool pow2);
bool mg_march_usable(const MGMarch &A, int K);

}  // namespace pyro
