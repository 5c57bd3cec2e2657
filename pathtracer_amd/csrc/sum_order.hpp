// Order of the additions in the 3- and 4-term sums whose results are compared bit for bit with
// the CPU restatement (camera matrix and rays, the ray / triangle solve, distances).  The reference
// leaves that order to Eigen (src/util/cMathUtil.cpp:145-153, src/geometry/cAccelStruct.cpp:44-48),
// whose source is not in its checkout; the plausible orders are a build-time switch, the same one
// the restatement has, so that the library can follow whichever is chosen:
//   PTX_SUM 0  left to right     (a + b) + c      ((a + b) + c) + d     -- the default build
//   PTX_SUM 1  last two first    a + (b + c)      (a + b) + (c + d)     -- Eigen's unrolled scalar reduction
//   PTX_SUM 2  2-wide packets    a + (b + c)      (a + c) + (b + d)     -- Eigen's vectorised reduction of a 4-vector
// SQ4 is the squared norm of a 4-vector whose w is 0 (directions and differences of points).
#pragma once
#ifndef PTX_SUM
#define PTX_SUM 0
#endif
#if PTX_SUM == 0
#define PTX_SUM3(a, b, c) (((a) + (b)) + (c))
#define PTX_SUM4(a, b, c, d) ((((a) + (b)) + (c)) + (d))
#define PTX_SQ4(x, y, z) (((x) + (y)) + (z))
#elif PTX_SUM == 1
#define PTX_SUM3(a, b, c) ((a) + ((b) + (c)))
#define PTX_SUM4(a, b, c, d) (((a) + (b)) + ((c) + (d)))
#define PTX_SQ4(x, y, z) (((x) + (y)) + (z))
#else
#define PTX_SUM3(a, b, c) ((a) + ((b) + (c)))
#define PTX_SUM4(a, b, c, d) (((a) + (c)) + ((b) + (d)))
#define PTX_SQ4(x, y, z) (((x) + (z)) + (y))
#endif
