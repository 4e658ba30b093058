// Packed storage of the symmetric 40 x 40 matrices of a Lorenz-96 problem (S, dEsde/dS, Psi):
// the lower BLOCK triangle by rows, with 16 x 16 blocks.  Row r keeps its first
// 16 (r / 16 + 1) entries -- everything left of and inside its diagonal block -- so every row
// starts and ends on a 128-byte line:
//
//   rows  0 .. 15   16 doubles each, at  16 r
//   rows 16 .. 31   32 doubles each, at 256 + 32 (r - 16)
//   rows 32 .. 39   48 doubles each, at 768 + 48 (r - 32)   (columns 40 .. 47 are padding:
//                                                            never read, never written)
//
// 1152 doubles instead of 1600 (72 %).  The diagonal blocks are kept whole, both triangles:
// the kernels that factor a matrix read them as full tiles.  An entry above the block
// diagonal is read as its mirror image.
#pragma once

#include <hip/hip_runtime.h>

namespace vgpa {

struct SymRows40 {
    static constexpr int D = 40, SIZE = 1152, PIECES = SIZE / 2;
    static __host__ __device__ constexpr int row_width(int r) { return 16 * (r / 16 + 1); }
    static __host__ __device__ constexpr int row_offset(int r) {
        return r < 16 ? 16 * r : (r < 32 ? 256 + 32 * (r - 16) : 768 + 48 * (r - 32));
    }
    // is (r, c) kept (c inside the row's width and inside the matrix)
    static __host__ __device__ constexpr bool kept(int r, int c) { return c < row_width(r) && c < D; }
    // element index -> row, column (the column may be padding, >= D)
    static __host__ __device__ inline void coords(int e, int& r, int& c) {
        if (e < 256) {
            r = e >> 4;
            c = e & 15;
        } else if (e < 768) {
            r = 16 + ((e - 256) >> 5);
            c = (e - 256) & 31;
        } else {
            r = 32 + (e - 768) / 48;
            c = (e - 768) % 48;
        }
    }
    // where entry (r, c) of the symmetric matrix lives, for any r, c < D
    static __host__ __device__ inline int index(int r, int c) {
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        return kept(r, c) ? row_offset(r) + c : row_offset(hi) + lo;
    }
};

}  // namespace vgpa
