// Packed image of one 128 x 128 lower triangular diagonal block, as the substitution leaf of the
// triangular solve (trsm.hip) wants it in LDS.
//
// The reference solves triangular systems by SUBSTITUTION all the way down (recursion to n <= 4, then the
// closed forms of triangular_solve.rs:98-198: y_i = b_i * (1 / t_ii) + sum_j (-t_ij / t_ii) y_j), which is
// backward stable whatever the conditioning of the triangle.  Round 1 of this library multiplied by explicit
// inverses of the 128 x 128 diagonal blocks instead; that is only as accurate as cond(T_kk) allows.  The leaf
// now substitutes, and what the producers (pack kernel, Cholesky leaf) hand it is the triangle itself:
//
//   T = [ T00  0  ]    T00, T11: 64 x 64 lower triangular, T10: 64 x 64
//       [ T10 T11 ]
//
//   image = [ tri(T00) | T10 row major | tri(T11) | pad ]
//   tri(): the triangle in the order the substitution consumes it -- groups of 8 columns; inside group g the
//          8 x 8 blocks of rows 8 (g + pb), pb = 0 .. 7 - g; inside a block column by column, 8 entries each:
//              pos(i, j) = goff(j / 8) + ((i / 8 - j / 8) * 8 + j % 8) * 8 + i % 8,   goff(g) = 64 (8 g - g (g - 1) / 2).
//          The diagonal holds the RECIPROCAL 1 / t_jj (1 for a unit triangle / identity padding) -- the reference
//          multiplies by the reciprocal too (triangular_solve.rs:113,121) -- and the strictly upper entries of the
//          diagonal blocks are zero.  The leaf runs ONE copy of the code for a group (8 x 8 diagonal block, then the
//          blocks below it, leaving early when the triangle ends) in a run-time loop and rotates its registers by 8
//          per group: the fully unrolled triangle was 129 KB of straight-line code, twice the instruction cache, and
//          ran at the speed of instruction fetch (profiles/r02_trsm_leaf.txt).
//   pad:   the group code is regular (no test for the end of the triangle) and software pipelined: it reads past the
//          last group.
// Rows / columns beyond the block's real size are identity padded.
#pragma once
#include "common.h"

namespace fh {

constexpr int TP_NB = 128; // block
constexpr int TP_H = 64;   // half

template <typename T> struct TriPack {
	static constexpr int ALIGN = 16 / (int) sizeof(T); // elements per 16 bytes
	static constexpr int GW = 8;			   // columns per group, rows per block
	static constexpr int NG = TP_H / GW;
	static __host__ __device__ constexpr int goff(int g) { return 64 * (8 * g - g * (g - 1) / 2); }
	// position of entry (i, j), i >= j, of a 64 x 64 triangle
	static __host__ __device__ constexpr int tri_pos(int i, int j) { return goff(j >> 3) + (((i >> 3) - (j >> 3)) * 8 + (j & 7)) * 8 + (i & 7); }
	static constexpr int TRI = goff(NG); // elements of one packed 64 x 64 triangle (2304)
	static constexpr int OFF_T00 = 0;
	static constexpr int OFF_T10 = TRI; // row major: T10t[i * 64 + j] = T(64 + i, j)
	static constexpr int OFF_T11 = TRI + TP_H * TP_H;
	static constexpr int PAD = 256; // the regular group code reads up to 3 blocks + 16 vectors past the last group (trsm.hip)
	static constexpr int SIZE = 2 * TRI + TP_H * TP_H + PAD;
	static constexpr size_t BYTES = (size_t) SIZE * sizeof(T);
};

} // namespace fh
