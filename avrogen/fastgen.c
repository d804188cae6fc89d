/* Synthetic Avro generator for the large BASELINE.json configurations
 * (bench/test infrastructure, not product code).  Emits byte-for-byte what
 * avrogen/synth.py + avrogen/encoder.py emit for the same (config, seed, row);
 * tests/test_avrogen.py checks that.  Record = pure function of (seed, row).
 *
 * Build: gcc -O3 -shared -fPIC -pthread fastgen.c -o _build/libfastgen.so
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { CFG_FULL = 0, CFG_FLAT4 = 1, CFG_CFG3 = 2, CFG_REALISTIC = 3, CFG_REALISTIC_HEAVY = 4, CFG_SKEWED = 5, CFG_REALISTIC_NOGIANT = 6, CFG_WIDE = 100 /* + columns */ };

static inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
typedef struct { uint64_t s; } rng_t;
static inline void rng_init(rng_t *g, uint64_t seed, uint64_t row) { g->s = mix(seed ^ (row * 0xD1342543DE82EF95ULL)); }
static inline uint64_t rng_next(rng_t *g) { g->s += 0x9E3779B97F4A7C15ULL; return mix(g->s); }
static inline uint64_t below(rng_t *g, uint64_t n) { return rng_next(g) % n; }
static inline uint64_t between(rng_t *g, uint64_t lo, uint64_t hi) { return lo + rng_next(g) % (hi - lo + 1); }

static inline uint8_t *put_varint(uint8_t *p, int64_t v) {
  uint64_t u = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  while (u >= 0x80) { *p++ = (uint8_t)(u | 0x80); u >>= 7; }
  *p++ = (uint8_t)u;
  return p;
}
/* string of n generated letters, length-prefixed */
static inline uint8_t *put_letters(uint8_t *p, rng_t *g, int n, char base, int span) {
  p = put_varint(p, n);
  uint64_t r = 0;
  for (int j = 0; j < n; j++) {
    if ((j & 7) == 0) r = rng_next(g);
    *p++ = (uint8_t)(base + ((r >> (8 * (j & 7))) & 0xFF) % span);
  }
  return p;
}
static inline uint8_t *put_str(uint8_t *p, const char *s, int n) {
  p = put_varint(p, n); memcpy(p, s, n); return p + n;
}

/* upper bound of one record's bytes per configuration (the part buffers keep that much room in front of every record) */
static uint64_t max_rec(int cfg) {
  if (cfg == CFG_REALISTIC || cfg == CFG_REALISTIC_HEAVY || cfg == CFG_REALISTIC_NOGIANT) return 9000ull * 30 + 9215 + 1024;
  if (cfg == CFG_SKEWED) return 225ull * 384 + 4096;      /* every letter of gen_full x the largest scale (8 x 48) */
  if (cfg >= CFG_WIDE) return (uint64_t)(cfg - CFG_WIDE) * 19 + 12 * 48 + 256;
  return 512;
}

static uint8_t *gen_full(uint8_t *p, uint64_t seed, uint64_t row) {
  rng_t g; rng_init(&g, seed, row);
  if (below(&g, 2)) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 9, 18), 'a', 26); } else *p++ = 0;
  if (below(&g, 2)) { *p++ = 2; p = put_varint(p, (int64_t)between(&g, 18, 80)); } else *p++ = 0;
  int ne = (int)below(&g, 4);
  if (ne) { p = put_varint(p, ne); for (int i = 0; i < ne; i++) p = put_letters(p, &g, (int)between(&g, 17, 28), 'a', 26); }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    p = put_letters(p, &g, (int)between(&g, 14, 30), 'a', 26);
    p = put_letters(p, &g, (int)between(&g, 8, 18), 'a', 26);
    p = put_letters(p, &g, 5, '0', 10);
  } else *p++ = 0;
  int np = (int)below(&g, 4);
  if (np) {
    p = put_varint(p, np);
    for (int i = 0; i < np; i++) {
      p = put_letters(p, &g, (int)between(&g, 3, 9), 'a', 26);
      p = put_letters(p, &g, (int)between(&g, 10, 22), '0', 10);
    }
  }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    int cm = (int)below(&g, 3);
    if (cm == 0) *p++ = 0; else { *p++ = 2; p = put_str(p, cm == 1 ? "email" : "phone", 5); }
    *p++ = (uint8_t)below(&g, 2);
  } else *p++ = 0;
  int sk = (int)below(&g, 4);
  p = put_varint(p, sk);
  if (sk == 1) p = put_letters(p, &g, (int)between(&g, 3, 9), 'a', 26);
  else if (sk == 2) p = put_varint(p, (int64_t)between(&g, 0, 100));
  else if (sk == 3) *p++ = (uint8_t)below(&g, 2);
  p = put_varint(p, 1726000000LL + (int64_t)below(&g, 31536000));
  p = put_varint(p, (int64_t)below(&g, 3));
  return p;
}

/* synth.py _gen_realistic: draw for draw */
static uint8_t *gen_realistic_n(uint8_t *p, uint64_t seed, uint64_t row, uint64_t big_per_million) {
  rng_t g; rng_init(&g, seed, row);
  if (below(&g, 2)) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 9, 18), 'a', 26); } else *p++ = 0;
  if (below(&g, 2)) { *p++ = 2; p = put_varint(p, (int64_t)between(&g, 18, 80)); } else *p++ = 0;
  int ne = below(&g, 1000000) < big_per_million ? (int)between(&g, 8192, 9000) : (int)below(&g, 4);
  if (ne) { p = put_varint(p, ne); for (int i = 0; i < ne; i++) p = put_letters(p, &g, (int)between(&g, 17, 28), 'a', 26); }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    p = put_letters(p, &g, (int)between(&g, 14, 30), 'a', 26);
    p = put_letters(p, &g, (int)between(&g, 8, 18), 'a', 26);
    p = put_letters(p, &g, 5, '0', 10);
  } else *p++ = 0;
  int np = (int)below(&g, 4);
  if (np) {
    p = put_varint(p, np);
    for (int i = 0; i < np; i++) {
      p = put_letters(p, &g, (int)between(&g, 3, 9), 'a', 26);
      p = put_letters(p, &g, (int)between(&g, 10, 22), '0', 10);
    }
  }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    int cm = (int)below(&g, 3);
    if (cm == 0) *p++ = 0; else { *p++ = 2; p = put_str(p, cm == 1 ? "email" : "phone", 5); }
    *p++ = (uint8_t)below(&g, 2);
  } else *p++ = 0;
  int sk = (int)below(&g, 4);
  p = put_varint(p, sk);
  if (sk == 1) p = put_letters(p, &g, (int)between(&g, 3, 9), 'a', 26);
  else if (sk == 2) p = put_varint(p, (int64_t)between(&g, 0, 100));
  else if (sk == 3) *p++ = (uint8_t)below(&g, 2);
  if (below(&g, 20) == 0) *p++ = 0;
  else { *p++ = 2; p = put_varint(p, 1750000000000000LL + (int64_t)below(&g, 31536000000000ULL)); }
  p = put_varint(p, (int64_t)below(&g, 3));
  p = put_varint(p, 1750000000LL + (int64_t)below(&g, 31536000));
  {
    uint64_t ms = 461165025343ULL + below(&g, 31536000000ULL);
    uint64_t worker = below(&g, 1024), seq = below(&g, 4096);
    p = put_varint(p, (int64_t)((ms << 22) | (worker << 12) | seq));
  }
  {
    uint64_t r = below(&g, 100);
    if (r == 0) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 8192, 9215), 'a', 26); }
    else if (r < 10) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 20, 60), 'a', 26); }
    else *p++ = 0;
  }
  return p;
}
static uint8_t *gen_realistic(uint8_t *p, uint64_t seed, uint64_t row) { return gen_realistic_n(p, seed, row, 100); }
static uint8_t *gen_realistic_heavy(uint8_t *p, uint64_t seed, uint64_t row) { return gen_realistic_n(p, seed, row, 10000); }
static uint8_t *gen_realistic_nogiant(uint8_t *p, uint64_t seed, uint64_t row) { return gen_realistic_n(p, seed, row, 0); }

/* synth.py skew_scale / gen_full_skewed */
static int pick(const int (*t)[2], int n, uint64_t r) {
  for (int i = 0; i < n; i++) if (r < (uint64_t)t[i][0]) return t[i][1];
  return t[n - 1][1];
}
static int skew_scale(uint64_t seed, uint64_t row) {
  static const int rec[11][2] = {{512, 1}, {768, 2}, {896, 3}, {960, 4}, {992, 6}, {1008, 8}, {1016, 12}, {1020, 16}, {1022, 24}, {1023, 32}, {1024, 48}};
  static const int run[5][2] = {{600, 1}, {800, 2}, {900, 3}, {960, 5}, {1000, 8}};
  rng_t a, b;
  rng_init(&a, seed ^ 0x5EED5CA1EULL, row / 97);
  rng_init(&b, seed ^ 0x0DDBA11ULL, row);
  return pick(run, 5, below(&a, 1000)) * pick(rec, 11, below(&b, 1024));
}
static uint8_t *gen_skewed(uint8_t *p, uint64_t seed, uint64_t row) {
  const int m = skew_scale(seed, row);
  rng_t g; rng_init(&g, seed, row);
  if (below(&g, 2)) { *p++ = 2; p = put_letters(p, &g, m * (int)between(&g, 9, 18), 'a', 26); } else *p++ = 0;
  if (below(&g, 2)) { *p++ = 2; p = put_varint(p, (int64_t)between(&g, 18, 80)); } else *p++ = 0;
  int ne = (int)below(&g, 4);
  if (ne) { p = put_varint(p, ne); for (int i = 0; i < ne; i++) p = put_letters(p, &g, m * (int)between(&g, 17, 28), 'a', 26); }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    p = put_letters(p, &g, m * (int)between(&g, 14, 30), 'a', 26);
    p = put_letters(p, &g, m * (int)between(&g, 8, 18), 'a', 26);
    p = put_letters(p, &g, 5, '0', 10);
  } else *p++ = 0;
  int np = (int)below(&g, 4);
  if (np) {
    p = put_varint(p, np);
    for (int i = 0; i < np; i++) {
      p = put_letters(p, &g, (int)between(&g, 3, 9), 'a', 26);
      p = put_letters(p, &g, m * (int)between(&g, 10, 22), '0', 10);
    }
  }
  *p++ = 0;
  if (below(&g, 2)) {
    *p++ = 2;
    int cm = (int)below(&g, 3);
    if (cm == 0) *p++ = 0; else { *p++ = 2; p = put_str(p, cm == 1 ? "email" : "phone", 5); }
    *p++ = (uint8_t)below(&g, 2);
  } else *p++ = 0;
  int sk = (int)below(&g, 4);
  p = put_varint(p, sk);
  if (sk == 1) p = put_letters(p, &g, m * (int)between(&g, 3, 9), 'a', 26);
  else if (sk == 2) p = put_varint(p, (int64_t)between(&g, 0, 100));
  else if (sk == 3) *p++ = (uint8_t)below(&g, 2);
  p = put_varint(p, 1726000000LL + (int64_t)below(&g, 31536000));
  p = put_varint(p, (int64_t)below(&g, 3));
  return p;
}

/* synth.py gen_wide(ncols): schemas.wide_schema's field order */
static uint8_t *gen_wide(uint8_t *p, uint64_t seed, uint64_t row, int ncols) {
  rng_t g; rng_init(&g, seed, row);
  const int arrays = 12;
  int every = ncols / arrays; if (every < 1) every = 1;
  int na = 0;
  for (int i = 0; i < ncols; i++) {
    if (below(&g, 2)) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 4, 16), 'a', 26); } else *p++ = 0;
    if ((i + 1) % every == 0 && na < arrays) {
      int n = (int)below(&g, 4);
      if (n) { p = put_varint(p, n); for (int j = 0; j < n; j++) p = put_letters(p, &g, (int)between(&g, 3, 12), 'a', 26); }
      *p++ = 0;
      na++;
    }
  }
  for (; na < arrays; na++) {
    int n = (int)below(&g, 4);
    if (n) { p = put_varint(p, n); for (int j = 0; j < n; j++) p = put_letters(p, &g, (int)between(&g, 3, 12), 'a', 26); }
    *p++ = 0;
  }
  return p;
}

static uint8_t *gen_flat4(uint8_t *p, uint64_t seed, uint64_t row) {
  (void)seed;
  p = put_varint(p, (int64_t)row);
  p = put_varint(p, (int64_t)row * 7);
  double d = (double)row * 2.25; memcpy(p, &d, 8); p += 8;
  *p++ = (row % 2 == 0);
  return p;
}

static uint8_t *gen_cfg3(uint8_t *p, uint64_t seed, uint64_t row) {
  rng_t g; rng_init(&g, seed, row);
  p = put_varint(p, (int64_t)row * 7);
  if (below(&g, 2)) { *p++ = 2; p = put_letters(p, &g, (int)between(&g, 8, 19), 'a', 26); } else *p++ = 0;
  if (below(&g, 2)) { *p++ = 2; p = put_varint(p, (int64_t)between(&g, 18, 80)); } else *p++ = 0;
  char buf[40]; int n = snprintf(buf, sizeof buf, "row-%llu", (unsigned long long)row);
  p = put_str(p, buf, n);
  p = put_varint(p, (int64_t)(row % 3));
  return p;
}

typedef uint8_t *(*gen_fn)(uint8_t *, uint64_t, uint64_t);
static gen_fn pick_gen(int cfg) {
  switch (cfg) {
    case CFG_FULL: return gen_full;
    case CFG_FLAT4: return gen_flat4;
    case CFG_REALISTIC: return gen_realistic;
    case CFG_REALISTIC_HEAVY: return gen_realistic_heavy;
    case CFG_REALISTIC_NOGIANT: return gen_realistic_nogiant;
    case CFG_SKEWED: return gen_skewed;
    default: return gen_cfg3;
  }
}

typedef struct {
  int cfg; uint64_t seed, start, n;
  uint8_t *buf; uint64_t len, cap; uint32_t *lens;
} part_t;

static void *run_part(void *arg) {
  part_t *t = (part_t *)arg;
  gen_fn fn = pick_gen(t->cfg);
  const uint64_t MAX_REC = max_rec(t->cfg);
  t->cap = t->n * 160 + MAX_REC; t->buf = (uint8_t *)malloc(t->cap);
  t->lens = (uint32_t *)malloc(sizeof(uint32_t) * (t->n ? t->n : 1));
  for (uint64_t i = 0; i < t->n; i++) {
    if (t->len + MAX_REC > t->cap) { t->cap = t->cap * 2 + MAX_REC; t->buf = (uint8_t *)realloc(t->buf, t->cap); }
    uint8_t *e = t->cfg >= CFG_WIDE ? gen_wide(t->buf + t->len, t->seed, t->start + i, t->cfg - CFG_WIDE)
                                    : fn(t->buf + t->len, t->seed, t->start + i);
    t->lens[i] = (uint32_t)(e - (t->buf + t->len));
    t->len += t->lens[i];
  }
  return NULL;
}

typedef struct { int nparts; part_t *parts; uint64_t n, total; } fg_t;

void *fg_generate(int cfg, uint64_t seed, uint64_t start, uint64_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > n) nthreads = n ? (int)n : 1;
  fg_t *h = (fg_t *)calloc(1, sizeof(fg_t));
  h->nparts = nthreads; h->n = n;
  h->parts = (part_t *)calloc(nthreads, sizeof(part_t));
  pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
  uint64_t per = n / nthreads;
  for (int i = 0; i < nthreads; i++) {
    part_t *t = &h->parts[i];
    t->cfg = cfg; t->seed = seed; t->start = start + per * i;
    t->n = (i == nthreads - 1) ? n - per * i : per;
    pthread_create(&th[i], NULL, run_part, t);
  }
  for (int i = 0; i < nthreads; i++) { pthread_join(th[i], NULL); h->total += h->parts[i].len; }
  free(th);
  return h;
}
uint64_t fg_total_bytes(void *hv) { return ((fg_t *)hv)->total; }

/* data: total bytes; offsets: n+1 u64 */
void fg_copy_out(void *hv, uint8_t *data, uint64_t *offsets) {
  fg_t *h = (fg_t *)hv;
  uint64_t pos = 0, row = 0;
  for (int i = 0; i < h->nparts; i++) {
    part_t *t = &h->parts[i];
    memcpy(data + pos, t->buf, t->len);
    uint64_t o = pos;
    for (uint64_t j = 0; j < t->n; j++) { offsets[row++] = o; o += t->lens[j]; }
    pos += t->len;
  }
  offsets[row] = pos;
}
void fg_free(void *hv) {
  fg_t *h = (fg_t *)hv;
  for (int i = 0; i < h->nparts; i++) { free(h->parts[i].buf); free(h->parts[i].lens); }
  free(h->parts); free(h);
}
