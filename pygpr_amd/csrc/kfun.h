// Device helpers shared by the covariance / gradient tile kernels (kbuild.hip: VALU bodies; kmfma.hip: matrix-pipe bodies).
#pragma once
#include "common.h"

// exp for the covariance kernels: branch-free, 2^k * P13(r) with r = x - k ln2 (two-term reduction) and the
// Taylor polynomial to degree 13 on |r| <= ln2/2 (truncation 4e-18 relative); <= 2 ulp, subnormal results via
// v_ldexp_f64.  Arguments are <= 0 here; anything below -800 gives 0.
__device__ __forceinline__ double pg_exp(double x) {
    x = (x < -800.0) ? -800.0 : x;      // not fmax: a NaN argument (NaN coordinate or hyper-parameter) must stay NaN
    const double kf = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(-kf, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                  // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);      // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);     // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);     // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);    // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);      // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);     // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);     // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);     // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);    // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);    // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
__device__ __forceinline__ float pg_exp(float x) { return expf(x); }

// The covariance build's own exponential (round 3): the build is bound by fp64 VALU issue, and the degree-13 chain above is half of an
// element's instructions.  Here x = (32 e + j) ln2 / 32 + r with |r| <= ln2 / 64: 2^(j/32) comes from a 32-entry table in LDS
// (32 doubles cover the 64 banks once: no conflicts between distinct entries), exp(r) from the Taylor polynomial to degree 6
// (truncation r^7 / 7! < 3.5e-18), 2^e from v_ldexp_f64: 13 fp64 operations instead of 21, <= 2 ulp.  `tab` may carry a factor
// (the component's sigma^2) -- the product costs nothing then.
static __device__ const double pg_exp2_32[32] = {
    1.00000000000000000e+00, 1.02189714865411663e+00, 1.04427378242741375e+00, 1.06714040067682370e+00,
    1.09050773266525769e+00, 1.11438674259589243e+00, 1.13878863475669156e+00, 1.16372485877757748e+00,
    1.18920711500272103e+00, 1.21524735998046896e+00, 1.24185781207348400e+00, 1.26905095719173322e+00,
    1.29683955465100964e+00, 1.32523664315974132e+00, 1.35425554693689265e+00, 1.38390988196383202e+00,
    1.41421356237309515e+00, 1.44518080697704665e+00, 1.47682614593949935e+00, 1.50916442759342284e+00,
    1.54221082540794074e+00, 1.57598084510788650e+00, 1.61049033194925428e+00, 1.64575547815396495e+00,
    1.68179283050742900e+00, 1.71861929812247793e+00, 1.75625216037329945e+00, 1.79470907500310717e+00,
    1.83400808640934243e+00, 1.87416763411029996e+00, 1.91520656139714740e+00, 1.95714412417540018e+00};
__device__ __forceinline__ double pg_exp_tab(double x, const double* tab) {
    x = (x < -800.0) ? -800.0 : x;      // (a NaN argument stays NaN)
    const double kf = __builtin_rint(x * 4.61662413084468283841e+01);              // 32 / ln2
    double r = __builtin_fma(-kf, 2.16608493865351192653e-02, x);                   // ln2 / 32, upper 32 bits: kf * hi is exact
    r = __builtin_fma(-kf, 5.96317165397058656257e-12, r);
    const int k = (int)kf;
    double p = 1.3888888888888889e-03;                  // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03);    // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);    // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);    // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p * tab[k & 31], k >> 5);
}

// sqrt(x) for x >= 0 in the Matern kernels' radial distance (a NaN stays a NaN): v_rsq_f64 refined by one Newton step on 1/sqrt and one on
// the root itself (both quadratic: <= 1 ulp whatever the instruction's own precision) -- nine fp64 operations where the IEEE expansion
// of sqrt() with its range scaling is about twenty.  Arguments below 1e-280 (a point against itself) return about 1e-140: every term
// the kernels form from r then rounds exactly as with r = 0; arguments above 1e300 (points an overflow apart) are taken as 1e300 -- the
// reciprocal root of infinity is 0 and would turn the result into a NaN where the covariance is simply 0.
__device__ __forceinline__ double pg_sqrt_pos(double x) {
    x = (x < 1.0e-280) ? 1.0e-280 : x;
    x = (x > 1.0e300) ? 1.0e300 : x;
    double y = __builtin_amdgcn_rsq(x);
    const double t = __builtin_fma(-0.5 * x * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    double r = x * y;
    r = __builtin_fma(__builtin_fma(-r, r, x), 0.5 * y, r);
    return r;
}

// Strip (tile row tr, first tile tcs, ntile tiles) of workgroup `b` of a 1-D grid: a workgroup walks up to S consecutive tiles of
// one tile row.  Symmetric builds launch ONLY tiles on or below the diagonal (round 2 launched the full square and let the upper
// half exit at once): column window [c0, c1) in tiles --
//   tile rows c0 .. c1-1 hold r' + 1 tiles (r' = tr - c0: the triangle) = ceil((r' + 1) / S) strips,
//   tile rows c1 .. T-1 hold W = c1 - c0 tiles (the rectangle below it) = ceil(W / S) strips.
__host__ __device__ __forceinline__ long kb_strips_before(int rp, int S) {      // strips in triangle rows 0 .. rp-1
    const long q = rp / S, rem = rp % S;
    return (long)S * q * (q + 1) / 2 + rem * (q + 1);
}
__device__ __forceinline__ void kb_strip_of(int b, int symmetric, int c0, int c1, int S, int& tr, int& tcs, int& ntile) {
    const int W = c1 - c0, SW = (W + S - 1) / S;
    if (!symmetric) {
        tr = b / SW;
        tcs = c0 + (b % SW) * S;
        ntile = min(S, c1 - tcs);
        return;
    }
    const int ntri = (int)kb_strips_before(W, S);
    if (b < ntri) {
        int q = (int)((sqrtf(1.0f + 8.0f * (float)b / (float)S) - 1.0f) * 0.5f);
        while (q > 0 && (long)S * q * (q + 1) / 2 > b) --q;
        while ((long)S * (q + 1) * (q + 2) / 2 <= b) ++q;
        const int within = b - S * q * (q + 1) / 2;         // strips into the block of S rows that hold q + 1 strips each
        const int rem = within / (q + 1), sidx = within % (q + 1);
        const int rp = S * q + rem;
        tr = c0 + rp;
        tcs = c0 + sidx * S;
        ntile = min(S, rp + 1 - sidx * S);
    } else {
        const int j = b - ntri;
        tr = c1 + j / SW;
        tcs = c0 + (j % SW) * S;
        ntile = min(S, c1 - tcs);
    }
}

