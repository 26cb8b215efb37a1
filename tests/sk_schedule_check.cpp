// Host-side checker of videoyolo_amd/csrc/sk_schedule.h (the stream-K work split of conv_igemm.hip), compiled with g++
// by tests/test_sk_schedule.py.  For many (tiles, blocks, k-steps per tile) it runs the schedule of EVERY block and checks
//   * every k-step of every tile is computed exactly once, in order along the tile's chain (pieces [0,a) [a,T)),
//   * a piece that continues a tile (load_partial) starts where the piece of block vblk - 1 that stored it stops, that
//     block is in the same XCD group, and it ran the storing piece as an item that waits for nothing,
//   * at most one HEAD and one TAIL per block, stores only from HEADs, loads only into TAILs, a TAIL is a block's last item,
//   * the whole-tile waves of an XCD cover a contiguous run, block l taking tile w * gx + l,
//   * shares are balanced: no block has more than one tile's worth of k-steps more than another,
//   * with `align` (launches whose K is summed in runs): every cut point is a multiple of it.
// usage: sk_schedule_check            -> "ok <cases> cases" or a message and exit code 1
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../videoyolo_amd/csrc/sk_schedule.h"

static int fail(const char* what, int tiles, int nblk, int T, int L) {
  fprintf(stderr, "FAIL %s: tiles %d blocks %d T %d block %d\n", what, tiles, nblk, T, L);
  return 1;
}

static int check(int tiles, int nblk, int T, int align = 1) {
  auto div = [](unsigned n, unsigned d) { return (int)(n / d); };
  std::vector<int> next_k(tiles, 0);          // how far each tile's chain has been computed (pieces must arrive in order)
  std::vector<int> stored_by(tiles, -1), stored_to(tiles, 0);
  std::vector<SkSchedule> sch(nblk);
  std::vector<long long> work(nblk, 0);
  for (int L = 0; L < nblk; ++L) sch[L] = sk_schedule(nblk, L, tiles, T, div, align);
  // pass 1: everything that does not wait (whole tiles and HEADs), pass 2: the TAILs
  for (int pass = 0; pass < 2; ++pass) {
    for (int L = 0; L < nblk; ++L) {
      const SkSchedule& s = sch[L];
      if (s.gx < 1 || s.l >= s.gx) return fail("group", tiles, nblk, T, L);
      int heads = 0, tails = 0;
      for (int it = 0; it < s.n_items; ++it) {
        const SkItem w = sk_item(s, it, T);
        if (w.tile < 0 || w.tile >= tiles || w.kb < 0 || w.ke > T || w.kb >= w.ke) return fail("range", tiles, nblk, T, L);
        if (w.kb % align || w.ke % align) return fail("cut point not aligned", tiles, nblk, T, L);
        if (it < s.D && w.tile != s.tx0 + it * s.gx + s.l) return fail("wave tile", tiles, nblk, T, L);
        heads += w.store_partial && !w.load_partial;
        tails += w.load_partial;
        if (w.load_partial && it != s.n_items - 1) return fail("tail is not last", tiles, nblk, T, L);
        if (w.store_partial && w.load_partial) return fail("middle piece", tiles, nblk, T, L);
        if (w.store_partial && w.kb != 0) return fail("head does not start the tile", tiles, nblk, T, L);
        if (!w.store_partial && w.ke != T) return fail("unfinished tile without a store", tiles, nblk, T, L);
        if ((pass == 1) != w.load_partial) continue;
        if (pass == 0) work[L] += 0;
        if (w.load_partial) {
          if (s.l == 0) return fail("first block of a group continues a tile", tiles, nblk, T, L);
          if (stored_by[w.tile] != s.vblk - 1) return fail("continued tile was not stored by vblk - 1", tiles, nblk, T, L);
          if (stored_to[w.tile] != w.kb) return fail("continuation does not start where the head stopped", tiles, nblk, T, L);
        } else if (w.kb != 0) {
          return fail("piece starts mid-tile without loading", tiles, nblk, T, L);
        }
        if (next_k[w.tile] != w.kb) return fail("k-steps out of order or repeated", tiles, nblk, T, L);
        next_k[w.tile] = w.ke;
        if (w.store_partial) stored_by[w.tile] = s.vblk, stored_to[w.tile] = w.ke;
        work[L] += w.ke - w.kb;
      }
      if (heads > 1 || tails > 1) return fail("more than one head or tail", tiles, nblk, T, L);
    }
  }
  for (int t = 0; t < tiles; ++t)
    if (next_k[t] != T) return fail("tile not finished", tiles, nblk, T, t);
  long long lo = work[0], hi = work[0];
  for (int L = 0; L < nblk; ++L) lo = work[L] < lo ? work[L] : lo, hi = work[L] > hi ? work[L] : hi;
  if (hi - lo > 2ll * T + 2ll * align) return fail("unbalanced shares", tiles, nblk, T, (int)(hi - lo));
  // vblk is a permutation of the blocks, XCD groups contiguous
  std::vector<int> seen(nblk, 0);
  for (int L = 0; L < nblk; ++L) {
    if (sch[L].vblk < 0 || sch[L].vblk >= nblk || seen[sch[L].vblk]++) return fail("vblk", tiles, nblk, T, L);
  }
  return 0;
}

int main() {
  int cases = 0;
  // the launches of the product: blocks = 256 x {1..4}; tiles of the 416 / 608 workloads; k-steps 2 .. 288
  const int grids[] = {256, 512, 768, 1024};
  const int tile_counts[] = {344, 676, 680, 688, 1352, 1444, 1448, 2704, 2720, 2888, 5776, 11552, 23104, 46208};
  const int ksteps[] = {2, 4, 8, 9, 18, 36, 72, 144, 288};
  for (int g : grids)
    for (int t : tile_counts)
      for (int T : ksteps)
        if (t > g) {
          if (check(t, g, T)) return 1;
          ++cases;
          if (T % 4 == 0) {  // launches whose K is summed in runs (K >= 2048: T = 72, 144): cuts at multiples of 4 k-steps
            if (check(t, g, T, 4)) return 1;
            ++cases;
          }
        }
  // the test switch VY_CONV_SK_SLOTS and anything else: small and odd grids, tiles from just above the grid upwards
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&](int n) {
    st ^= st << 13, st ^= st >> 7, st ^= st << 17;
    return (int)(st % (unsigned long long)n);
  };
  for (int i = 0; i < 20000; ++i) {
    const int g = 1 + rnd(96), t = g + 1 + rnd(6 * g + 40), T = 1 + rnd(40);
    if (check(t, g, T)) return 1;
    ++cases;
    if (check(t, g, 4 * T, 4) || check(t, g, 2 * T, 2)) return 1;
    cases += 2;
  }
  for (int g = 1; g <= 40; ++g)
    for (int t = g + 1; t <= 5 * g + 3; ++t)
      for (int T = 1; T <= 5; ++T) {
        if (check(t, g, T)) return 1;
        ++cases;
      }
  printf("ok %d cases\n", cases);
  return 0;
}
