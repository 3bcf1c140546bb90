// CPU check of the stream-K work partition (comprehensive-transformer-tts_amd/csrc/sk_plan.h): for a geometry given on the command
// line, walk every workgroup exactly as the kernel does and verify
//   1. every (tile, K-block) unit is covered exactly once;
//   2. a workgroup has at most one contribution (piece with kb_hi < nkb) and it is the first piece it processes;
//   3. every cut tile has exactly one owner, and the owner's gather loop (j-1, j-2, ... as in gemm_sk.hip) visits exactly the
//      workgroups that contributed to that tile, nearest first, and only lower-numbered workgroups of the same XCD;
//   4. sk_tile_decode is a bijection onto (m-slot, n-tile).
// usage: sk_plan_check n_mt tiles_n nkb W whole_tiles gw     -> prints "ok <pieces> <contributions>" or a diagnostic, exit code 0 / 1
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "../../comprehensive-transformer-tts_amd/csrc/sk_plan.h"

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
  const int n_mt = atoi(argv[1]), tiles_n = atoi(argv[2]), nkb = atoi(argv[3]), W = atoi(argv[4]), whole = atoi(argv[5]), gw = atoi(argv[6]);
  SkGeom g{n_mt * tiles_n, nkb, W, whole};
  std::vector<int> cover((size_t)g.n_tiles * nkb, 0);
  long pieces = 0, contribs = 0;
  // contributions per global tile: list of (xcd, j)
  std::map<int, std::vector<int>> contrib_of;
  struct Own { int xcd, j, kb_lo; };
  std::map<int, Own> owner_of;
  for (int xcd = 0; xcd < 8; ++xcd)
    for (int j = 0; j < W; ++j) {
      const SkRange rg = sk_range(g, xcd, j);
      if (rg.hi < rg.lo) { printf("negative range xcd %d j %d\n", xcd, j); return 1; }
      int u = rg.hi;
      SkPiece pc;
      int idx = 0;
      while (sk_next_piece(u, rg.lo, nkb, pc)) {
        const int tile = rg.T0 + pc.t;
        if (tile < rg.T0 || tile >= rg.T1 || pc.kb_lo < 0 || pc.kb_hi > nkb || pc.kb_lo >= pc.kb_hi) { printf("bad piece\n"); return 1; }
        for (int kb = pc.kb_lo; kb < pc.kb_hi; ++kb) cover[(size_t)tile * nkb + kb]++;
        ++pieces;
        if (pc.kb_hi < nkb) {
          if (idx != 0) { printf("contribution is not the first piece (xcd %d j %d)\n", xcd, j); return 1; }
          contrib_of[tile].push_back(j);
          ++contribs;
        } else {
          if (owner_of.count(tile)) { printf("two owners for tile %d\n", tile); return 1; }
          owner_of[tile] = Own{xcd, j, pc.kb_lo};
          if (pc.kb_lo > 0) {
            // the kernel's gather loop
            const int tile_lo = pc.t * nkb, Ux = (rg.T1 - rg.T0) * nkb;
            int upper = rg.lo;
            std::vector<int> got;
            for (int jj = j - 1; jj >= 0 && upper > tile_lo; --jj) {
              const int blo = sk_bound(g, Ux, jj);
              if (blo >= upper) continue;
              upper = blo;
              got.push_back(jj);
            }
            if (upper > tile_lo) { printf("gather does not reach the tile start (tile %d)\n", tile); return 1; }
            owner_of[tile].kb_lo = -1 - (int)got.size();
            // remember for the cross-check below
            contrib_of[-1 - tile] = got;
          }
        }
        ++idx;
      }
    }
  for (size_t i = 0; i < cover.size(); ++i)
    if (cover[i] != 1) { printf("unit %zu covered %d times\n", i, cover[i]); return 1; }
  for (int t = 0; t < g.n_tiles; ++t) {
    if (!owner_of.count(t)) { printf("tile %d has no owner\n", t); return 1; }
    std::vector<int> want = contrib_of.count(t) ? contrib_of[t] : std::vector<int>();
    std::vector<int> got = contrib_of.count(-1 - t) ? contrib_of[-1 - t] : std::vector<int>();
    // contributions were recorded in ascending j; the gather visits descending j
    std::vector<int> want_desc(want.rbegin(), want.rend());
    if (want_desc != got) { printf("tile %d: gather order differs from the contributors (%zu vs %zu)\n", t, want.size(), got.size()); return 1; }
    for (int jj : got) if (jj >= owner_of[t].j) { printf("owner waits upwards\n"); return 1; }
  }
  // tile decode bijection
  std::vector<int> seen((size_t)g.n_tiles, 0);
  for (int t = 0; t < g.n_tiles; ++t) {
    int ms, nt;
    sk_tile_decode(t, n_mt, gw, ms, nt);
    if (ms < 0 || ms >= n_mt || nt < 0 || nt >= tiles_n) { printf("decode out of range\n"); return 1; }
    seen[(size_t)ms * tiles_n + nt]++;
  }
  for (int v : seen) if (v != 1) { printf("decode is not a bijection\n"); return 1; }
  printf("ok %ld %ld\n", pieces, contribs);
  return 0;
}
