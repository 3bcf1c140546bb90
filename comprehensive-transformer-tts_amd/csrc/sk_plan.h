// Work partition of the persistent stream-K GEMM (gemm_sk.hip), shared by the device code and the CPU test of its invariants
// (tests/test_sk_plan_cpu.py compiles this header with g++).
//
// Unit = one 32-deep K-block of one 64x64 output tile.  Tiles are numbered 0..n_tiles-1 in SCHEDULE order (sk_tile_decode); the unit
// space is tile-major: unit = tile * nkb + kb.  It is cut in two levels:
//   * XCD x (workgroups with blockIdx.x % 8 == x; the placement is a speed assumption only) owns the tiles [T0, T1) with
//     T = floor(x * n_tiles / 8): chunk boundaries are tile boundaries, so no tile is ever shared between XCDs;
//   * the W workgroups of an XCD cut the chunk's units evenly: workgroup j owns local units [bound(j), bound(j+1)), bound(j) =
//     floor(j * Ux / W), rounded down to a whole tile when `whole_tiles` is set (small-K GEMMs: no tile is split at all).
// A workgroup walks its range from the TOP: pieces (tile, [kb_lo, kb_hi)) in descending tile order, ascending kb inside a piece.
//   * a piece with kb_hi <  nkb is a CONTRIBUTION: the accumulator goes to the workgroup's slab and its flag is raised.  A workgroup
//     has at most one contribution and it is the FIRST piece it processes.
//   * a piece with kb_hi == nkb makes the workgroup the OWNER of the tile: if kb_lo > 0 it collects the slabs of the workgroups
//     j-1, j-2, ... (same XCD) down to the one whose range contains the tile's unit 0, in that fixed order, then runs the epilogue.
//     The owner only ever waits for LOWER-numbered workgroups, which published at the very start of their life: the wait is normally
//     free, and no cycle of waits exists whatever subset of the grid is resident (dispatch is ascending).
#pragma once
#ifdef __HIPCC__
#define SK_HD __host__ __device__ __forceinline__
#else
#define SK_HD inline
#endif

struct SkGeom {
  int n_tiles;       // active tiles
  int nkb;           // K-blocks per tile
  int W;             // workgroups per XCD (grid = 8 * W)
  int whole_tiles;   // 1: range boundaries are tile boundaries
};

struct SkRange {
  int T0, T1;        // tiles of the XCD chunk
  int lo, hi;        // this workgroup's local units, relative to T0 * nkb (the host keeps a chunk below 2^31 / W units)
};

SK_HD int sk_bound(const SkGeom& g, int Ux, int j) {
  int b = (int)((long)Ux * j / g.W);
  if (g.whole_tiles) b = b / g.nkb * g.nkb;
  return b;
}

SK_HD SkRange sk_range(const SkGeom& g, int xcd, int j) {
  SkRange r;
  r.T0 = (int)((long)xcd * g.n_tiles / 8);
  r.T1 = (int)((long)(xcd + 1) * g.n_tiles / 8);
  const int Ux = (r.T1 - r.T0) * g.nkb;
  r.lo = sk_bound(g, Ux, j);
  r.hi = sk_bound(g, Ux, j + 1);
  return r;
}

struct SkPiece {
  int t;             // tile index relative to T0
  int kb_lo, kb_hi;  // K-blocks [kb_lo, kb_hi) of that tile
};

// next piece below `u` (exclusive top of what is left, local units); returns false when the range [lo, u) is empty
SK_HD bool sk_next_piece(int& u, int lo, int nkb, SkPiece& pc) {
  if (u <= lo) return false;
  const int t = (u - 1) / nkb;
  pc.t = t;
  pc.kb_hi = u - t * nkb;
  const int l = lo - t * nkb;
  pc.kb_lo = l > 0 ? l : 0;
  u = t * nkb + pc.kb_lo;
  return true;
}

// schedule order -> (m-tile slot, n-tile): n-tile groups of `gw` tiles outermost (an XCD then touches few B panels), m next, n inside
SK_HD void sk_tile_decode(int tile, int n_mt, int gw, int& mslot, int& nt) {
  const int per_group = n_mt * gw;
  const int grp = tile / per_group, rem = tile - grp * per_group;
  mslot = rem / gw;
  nt = grp * gw + (rem - mslot * gw);
}
