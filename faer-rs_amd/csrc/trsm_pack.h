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
// The block is cut into 16 x 16 tiles (the shape of v_mfma_{f64,f32}_16x16x4):
//   * the 28 tiles BELOW the diagonal are multiplied on the matrix cores: X_bi -= T[bi][bj] X_bj with the solved rows
//     X_bj as the B operand.  They are stored NEGATED, tile after tile (index bi (bi - 1) / 2 + bj), each in the order
//     the A operand is consumed: K step kk (4 columns), then lane l = (column & 3) * 16 + row -- one conflict-free
//     ds_read_b64 per lane and K step;
//   * the 8 DIAGONAL tiles are solved by substitution on the vector ALU (lane = right-hand side, the 16 rows in
//     registers, multipliers as wave-uniform LDS broadcasts): column j of a diagonal tile is the block
//     [ 1 / t_jj, t_{j+1,j}, ..., t_{15,j} ] starting at an offset aligned to 16 bytes.  The diagonal holds the
//     RECIPROCAL (1 for a unit triangle / identity padding): the reference multiplies by the reciprocal too
//     (triangular_solve.rs:113,121).
// Why tiles: a wave-uniform multiplier costs a whole LDS return slot (64 lanes x 16 bytes for two useful doubles), and
// ONE wavefront gets a ds_read_b128 through about every 27 cycles (profiles/r02_trsm_leaf_phases.txt) -- a purely
// vector-ALU substitution of the 128 x 128 block is bound by that, not by its FMAs.  The matrix cores take per-lane
// operands: 15/16 of the multipliers no longer need a broadcast.
// Rows / columns beyond the block's real size are identity padded.
#pragma once
#include "common.h"

namespace fh {

constexpr int TP_NB = 128; // block
constexpr int TP_H = 64;   // half

constexpr int TP_TS = 16;	      // tile
constexpr int TP_NT = TP_NB / TP_TS;  // tiles per side (8)

template <typename T> struct TriPack {
	static constexpr int ALIGN = 16 / (int) sizeof(T); // elements per 16 bytes
	static constexpr int OD_TILES = TP_NT * (TP_NT - 1) / 2;
	static constexpr int OD_SZ = TP_TS * TP_TS;
	static __host__ __device__ constexpr int dg_col_len(int j) { return (TP_TS - j + ALIGN - 1) / ALIGN * ALIGN; }
	static __host__ __device__ constexpr int dg_off(int j)
	{
		int o = 0;
		for (int c = 0; c < j; ++c)
			o += dg_col_len(c);
		return o;
	}
	static constexpr int DG_SZ = dg_off(TP_TS);
	static constexpr int OFF_DG = OD_TILES * OD_SZ;
	static constexpr int SIZE = OFF_DG + TP_NT * DG_SZ; // elements per block image (a multiple of ALIGN)
	static constexpr size_t BYTES = (size_t) SIZE * sizeof(T);
	static __host__ __device__ constexpr int od_tile(int bi, int bj) { return (bi * (bi - 1) / 2 + bj) * OD_SZ; }
	// position of entry (i, j), i >= j, of the 128 x 128 block; `negate`: the entry is stored negated (off-diagonal
	// tiles); the diagonal entries (i == j) are stored as reciprocals by the packers
	static __host__ __device__ int pos(int i, int j, bool &negate)
	{
		const int bi = i >> 4, bj = j >> 4, ii = i & 15, jj = j & 15;
		negate = bi > bj;
		if (bi > bj)
			return od_tile(bi, bj) + (jj >> 2) * 64 + (jj & 3) * 16 + ii;
		return OFF_DG + bi * DG_SZ + dg_off(jj) + (ii - jj);
	}
};

} // namespace fh
