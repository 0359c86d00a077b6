// cm2_blocks.h -- closed-form per-pixel Stokes block solve shared by the M_BD kernel
// (cm2_pixel.hip) and the fused two-level preconditioner tail (cm2_vector.hip).
#pragma once
#include "cm2_common.h"

namespace cm2 {

// the closed-form block solve of linearoperators.py:797-802 / :823-827 / :789-790
template <int POL>
__device__ __forceinline__ void bd_inverse_block(double hits, double c, double s, double c2,
                                                  double s2, double cs, double det, bool m,
                                                  const double *x, double *y)
{
    if (POL == 1) {
        y[0] = m ? x[0] / hits : 0.0;
    } else if (POL == 2) {
        if (m) {
            y[0] = (s2 * x[0] - cs * x[1]) / det;
            y[1] = (-cs * x[0] + c2 * x[1]) / det;
        } else {
            y[0] = 0.0;
            y[1] = 0.0;
        }
    } else {
        if (m) {
            y[0] = ((c2 * s2 - cs * cs) * x[0] + (s * cs - c * s2) * x[1]
                    + (c * cs - s * c2) * x[2]) / det;
            y[1] = ((s * cs - c * s2) * x[0] + (hits * s2 - s * s) * x[1]
                    + (s * c - hits * cs) * x[2]) / det;
            y[2] = ((c * cs - s * c2) * x[0] + (-hits * cs + c * s) * x[1]
                    + (hits * c2 - c * c) * x[2]) / det;
        } else {
            y[0] = 0.0;
            y[1] = 0.0;
            y[2] = 0.0;
        }
    }
}

}  // namespace cm2
