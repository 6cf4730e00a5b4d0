// mevp_pipeline.h -- pieces of the pipelined multi-iteration mEVP kernel (mevp_fused4.hip: the stages of a pass on the waves of a
// workgroup; rounds 1-4 also had two / three stages in ONE wave, csrc/mevp_fused2.hip / mevp_fused3.hip in the history): the per-lane march constants, the contributions a
// row carries to the row above it, the update of the four owned nodes of an element row and the gather of an element's
// nine nodal velocities from the owned nodes of its row, of the row above and of the right neighbour lane.  The same
// inlined functions in every variant keep their results bit-identical.
#pragma once
#include "mevp_common.h"

namespace nsdg_mevp_detail {

struct MarchConst3 {
    NodalConsts K;
    int nx, ny, y0, y1, tbeg, tendA, tendB, ix, ntx, nn, lane;
    long nplane; // doubles between two pair planes of the packed nodal coefficients
    bool own, hasL, lastcol;
    double hx, hy, ihx, ihy, iarea, ialpha, dmin2;
    AdaptConsts AC; // adaptive alpha, beta only
};

// contributions of a row to its top nodes, carried to the next row of the march
struct TopCarry3 {
    double x6 = 0., y6 = 0., x7 = 0., y7 = 0., xl8 = 0., yl8 = 0.; // 6: top-left, 7: top-mid of my column, 8 of the left column
    double q = 0., ql = 0.; // adaptive form only: the offers q_e = alpha_e h'_c of the element below and of the element below-left
};

// the four owned nodes of one element row from the carried contributions of the row below (`carry`), the
// contributions of this row (cx, cy) and the left neighbour's right-column contributions
// (AD: adaptive form -- `q` is this element's offer q_e = alpha_e h'_c of this sub-iteration, a node takes the largest offer of its adjacent elements)
template <bool AD = false>
__device__ __forceinline__ void owned_node_updates(const MarchConst3& M, bool hasB, const double (&c)[4][6], const double (&uu)[4],
    const double (&vv)[4], const TopCarry3& carry, const double (&cx)[9], const double (&cy)[9], double (&un)[4], double (&vn)[4], double q = 0.)
{
    const double l2x = lane_from_left(cx[2]), l2y = lane_from_left(cy[2]);
    const double l5x = lane_from_left(cx[5]), l5y = lane_from_left(cy[5]);
    if constexpr (AD) {
        const double ql = lane_from_left(q); // the left neighbour's (0 without one: it never wins the max)
        const double qmid = __builtin_fmax(ql, q), qbot = __builtin_fmax(carry.q, q);
        if (M.hasL && hasB)
            node_update_packed_adaptive(M.K, c[0], uu[0], vv[0], ((carry.xl8 + carry.x6) + l2x) + cx[0], ((carry.yl8 + carry.y6) + l2y) + cy[0], 9. * M.iarea,
                __builtin_fmax(__builtin_fmax(carry.ql, carry.q), qmid), M.AC.amin, un[0], vn[0]);
        else
            un[0] = vn[0] = 0.;
        if (hasB)
            node_update_packed_adaptive(M.K, c[1], uu[1], vv[1], carry.x7 + cx[1], carry.y7 + cy[1], 4.5 * M.iarea, qbot, M.AC.amin, un[1], vn[1]);
        else
            un[1] = vn[1] = 0.;
        if (M.hasL)
            node_update_packed_adaptive(M.K, c[2], uu[2], vv[2], l5x + cx[3], l5y + cy[3], 4.5 * M.iarea, qmid, M.AC.amin, un[2], vn[2]);
        else
            un[2] = vn[2] = 0.;
        node_update_packed_adaptive(M.K, c[3], uu[3], vv[3], cx[4], cy[4], 2.25 * M.iarea, q, M.AC.amin, un[3], vn[3]);
        return;
    }
    if (M.hasL && hasB)
        node_update_packed(M.K, c[0], uu[0], vv[0], ((carry.xl8 + carry.x6) + l2x) + cx[0], ((carry.yl8 + carry.y6) + l2y) + cy[0], 9. * M.iarea,
            un[0], vn[0]);
    else
        un[0] = vn[0] = 0.;
    if (hasB)
        node_update_packed(M.K, c[1], uu[1], vv[1], carry.x7 + cx[1], carry.y7 + cy[1], 4.5 * M.iarea, un[1], vn[1]);
    else
        un[1] = vn[1] = 0.;
    if (M.hasL)
        node_update_packed(M.K, c[2], uu[2], vv[2], l5x + cx[3], l5y + cy[3], 4.5 * M.iarea, un[2], vn[2]);
    else
        un[2] = vn[2] = 0.;
    node_update_packed(M.K, c[3], uu[3], vv[3], cx[4], cy[4], 2.25 * M.iarea, un[3], vn[3]);
}

template <bool AD = false>
__device__ __forceinline__ void carry_top(TopCarry3& carry, const double (&cx)[9], const double (&cy)[9], double q = 0.)
{
    carry.x6 = cx[6], carry.y6 = cy[6], carry.x7 = cx[7], carry.y7 = cy[7];
    carry.xl8 = lane_from_left(cx[8]), carry.yl8 = lane_from_left(cy[8]);
    if constexpr (AD)
        carry.q = q, carry.ql = lane_from_left(q);
}

// u at the 9 nodes of an element from the 4 owned nodes of its row (lo), the two bottom nodes of the row
// above (hi0 = V, hi1 = EX) and the right neighbour lane (node column 2*nx is the right boundary)
__device__ __forceinline__ void gather_nodes(const MarchConst3& M, const double (&lo)[4], double hi0, double hi1, double (&w)[9])
{
    w[0] = lo[0], w[1] = lo[1], w[3] = lo[2], w[4] = lo[3], w[6] = hi0, w[7] = hi1;
    const double r2 = lane_from_right(lo[0]), r5 = lane_from_right(lo[2]), r8 = lane_from_right(hi0);
    w[2] = M.lastcol ? 0. : r2;
    w[5] = M.lastcol ? 0. : r5;
    w[8] = M.lastcol ? 0. : r8;
}

} // namespace nsdg_mevp_detail
