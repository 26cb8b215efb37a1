// sk_schedule.h — which k-steps of which tiles a block of a stream-K conv launch runs (conv_igemm.hip, SK instances).
// Plain integer arithmetic shared by the kernel and by a host-side checker (tests/sk_schedule_check.cpp, compiled with
// g++ in the CPU test suite): the kernel passes a divider that keeps quotients in scalar registers, the checker `/`.
//
// Hybrid schedule, per XCD (the blocks L, L + 8, ... and a contiguous run of the tiles, in proportion): the first D
// "waves" are whole tiles, block l taking tile w * gx + l of the run — what the blocks of one XCD work on at the same
// time are neighbouring tiles, as in a plain launch (same A rows for the n-tiles of a row, the W panel shared).  Only the
// last one-to-two waves' worth of tiles is cut into equal shares of k-steps.  (Equal shares of the WHOLE sequence, the
// first build, gave every block its own 3.5 consecutive tiles: each A tile was then fetched 3.5 times by one block
// instead of once for eight, and launches of ten rounds lost more than the last round returns.)
// A share is [HEAD of its last tile -> slab] [whole tiles] [TAIL of its first tile, continuing block l - 1's HEAD].
#pragma once

#if defined(__HIPCC__)
#define VY_SK_HD __host__ __device__ __forceinline__
#else
#define VY_SK_HD inline
#endif

struct SkSchedule {
  int vblk;   // the block's index in XCD-contiguous order (slab / flag index; the HEAD it continues is vblk - 1's)
  int l, gx;  // index inside the XCD's group, blocks in the group
  int tx0;    // first tile of the XCD's run
  int D;      // whole-tile waves in front of the stream-K region
  int s0;     // first tile of the region
  int first, last, k0, k1;  // the share: tiles first .. last of the region, from k-step k0 of `first` to k1 of `last`
  int hb, w0, nw, tail, tk1;
  int n_items;
};

struct SkItem {
  int tile, kb, ke;  // k-steps [kb, ke) of the tile
  bool load_partial, store_partial;
};

// requires tiles >= nblk (every XCD's run then has at least as many tiles as blocks) and tiles * (nblk + 8) < 2^31.
// align: every cut point (k0, k1) is a multiple of it — 1, or 4 for launches whose K is summed in runs (the run boundaries
// of conv_tile sit at multiples of the LDS pipeline depth from a piece's first k-step); T_all must be a multiple of align.
template <typename Div>
VY_SK_HD SkSchedule sk_schedule(int nblk, int L, int tiles, int T_all, Div div, int align = 1) {
  SkSchedule s;
  const int q = nblk >> 3, r = nblk & 7, xcd = L & 7;
  s.l = L >> 3;
  s.gx = q + (xcd < r ? 1 : 0);
  const int v0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;  // the group's first block
  s.vblk = v0 + s.l;
  s.tx0 = div((unsigned)tiles * (unsigned)v0, (unsigned)nblk);
  const int nloc = div((unsigned)tiles * (unsigned)(v0 + s.gx), (unsigned)nblk) - s.tx0;  // >= gx
  const int waves = div((unsigned)nloc, (unsigned)s.gx);
  s.D = waves > 1 ? waves - 1 : 0;
  s.s0 = s.tx0 + s.D * s.gx;
  const int total = (nloc - s.D * s.gx) * T_all;                       // gx T_all <= total < 2 gx T_all
  int per = div((unsigned)(total + s.gx - 1), (unsigned)s.gx);         // T_all <= per < 2 T_all
  if (align > 1) per = div((unsigned)(per + align - 1), (unsigned)align) * align;  // (<= 2 T_all: T_all % align == 0)
  const int it0 = s.l * per;
  const int it1 = it0 + per < total ? it0 + per : total;
  s.first = s.last = s.k0 = s.k1 = s.hb = s.w0 = s.nw = s.tail = s.tk1 = 0;
  if (it0 < total) {
    s.first = div((unsigned)it0, (unsigned)T_all), s.k0 = it0 - s.first * T_all;
    s.last = div((unsigned)(it1 - 1), (unsigned)T_all), s.k1 = it1 - s.last * T_all;  // k1 in (0, T_all]
    if (s.first == s.last) {  // a share inside one tile is that tile's tail
      s.tail = 1, s.tk1 = s.k1;
    } else {
      s.hb = s.k1 < T_all ? 1 : 0;
      s.w0 = s.k0 == 0 ? s.first : s.first + 1;
      s.nw = (s.k1 == T_all ? s.last : s.last - 1) - s.w0 + 1;
      s.tail = s.k0 > 0 ? 1 : 0;
      s.tk1 = T_all;
    }
  }
  s.n_items = s.D + s.hb + s.nw + s.tail;
  return s;
}

// work items in running order: [D whole tiles] [HEAD] [whole tiles of the region] [TAIL]
VY_SK_HD SkItem sk_item(const SkSchedule& s, int it, int T_all) {
  SkItem w;
  if (it < s.D) {
    w.tile = s.tx0 + it * s.gx + s.l, w.kb = 0, w.ke = T_all, w.load_partial = false, w.store_partial = false;
  } else if (it < s.D + s.hb) {
    w.tile = s.s0 + s.last, w.kb = 0, w.ke = s.k1, w.load_partial = false, w.store_partial = true;
  } else if (it < s.D + s.hb + s.nw) {
    w.tile = s.s0 + s.w0 + (it - s.D - s.hb), w.kb = 0, w.ke = T_all, w.load_partial = false, w.store_partial = false;
  } else {
    w.tile = s.s0 + s.first, w.kb = s.k0, w.ke = s.tk1, w.load_partial = s.k0 > 0, w.store_partial = s.tk1 < T_all;
  }
  return w;
}
