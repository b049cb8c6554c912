/* Lapping filter parameters, 6-bit fixed point (DATA: the numeric contents of the
 * reference's exported tables OD_FILTER_PARAMS4/8/16/32, src/filter.c:169, :299, :535,
 * :860 - the "optimal 1-D subset3 Cg" sets, all of the TYPE3 rotation structure).
 * Layout for an n-point filter, h = n/2:
 *   [0, h)          scale of t[h+k]         (64 = no scaling step)
 *   [h, 2h-1)       A_j: t[j+1] += (t[j]*A_j + 32) >> 6,   j = h .. n-2  (index j-h)
 *   [2h-1, 3h-2)    B_j: t[j]   += (t[j+1]*B_j + 32) >> 6                (index j-h)
 * (Copy of daala_amd/csrc/filter_params.h: numbers only, for the CPU restatement.) */
#ifndef DAALA_FILTER_PARAMS_H
#define DAALA_FILTER_PARAMS_H
#define LAP_PARAMS4  {85, 75, -15, 33}
#define LAP_PARAMS8  {93, 72, 73, 78, -28, -23, -10, 50, 37, 23}
#define LAP_PARAMS16 {94, 71, 68, 68, 68, 69, 70, 73, -32, -37, -36, -32, -26, -17, -7, \
                      56, 49, 45, 40, 34, 26, 15}
#define LAP_PARAMS32 {91, 70, 68, 67, 67, 67, 67, 66, 66, 67, 67, 66, 67, 67, 67, 70, \
                      -32, -41, -42, -41, -40, -38, -36, -34, -32, -29, -24, -19, -14, -9, -5, \
                      58, 52, 50, 48, 45, 43, 40, 38, 35, 32, 29, 24, 18, 13, 8}
#endif
