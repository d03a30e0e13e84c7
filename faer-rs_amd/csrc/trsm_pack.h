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
//   image = [ tri(T00) | T10 row major | tri(T11) ]
//   tri(): column j is the block [ 1 / t_jj, t_{j+1,j}, ..., t_{63,j} ] (64 - j entries) starting at an offset aligned
//          to 16 bytes.  The substitution consumes the packed triangle as ONE linear stream of 16-byte LDS reads
//          (trsm.hip, software pipelined), reciprocal diagonal included -- the reference multiplies by the
//          reciprocal too (triangular_solve.rs:113,121); it is 1 for a unit triangle and for identity padding.
// Rows / columns beyond the block's real size are identity padded.
#pragma once
#include "common.h"

namespace fh {

constexpr int TP_NB = 128; // block
constexpr int TP_H = 64;   // half

template <typename T> struct TriPack {
	static constexpr int ALIGN = 16 / (int) sizeof(T); // elements per 16 bytes
	static constexpr int col_len(int j) { return (TP_H - j + ALIGN - 1) / ALIGN * ALIGN; } // 1 / diag + the entries below it
	static constexpr int tri_off(int j)
	{
		int o = 0;
		for (int c = 0; c < j; ++c)
			o += col_len(c);
		return o;
	}
	static constexpr int TRI = tri_off(TP_H); // elements of one packed 64 x 64 triangle
	static constexpr int OFF_T00 = 0;
	static constexpr int OFF_T10 = TRI;			   // row major: T10t[i * 64 + j] = T(64 + i, j)
	static constexpr int OFF_T11 = TRI + TP_H * TP_H;
	static constexpr int SIZE = 2 * TRI + TP_H * TP_H; // elements per block image (a multiple of ALIGN)
	static constexpr size_t BYTES = (size_t) SIZE * sizeof(T);
};

} // namespace fh
