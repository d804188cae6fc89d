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

enum { CFG_FULL = 0, CFG_FLAT4 = 1, CFG_CFG3 = 2 };

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

#define MAX_REC 512

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
static gen_fn pick(int cfg) { return cfg == CFG_FULL ? gen_full : cfg == CFG_FLAT4 ? gen_flat4 : gen_cfg3; }

typedef struct {
  int cfg; uint64_t seed, start, n;
  uint8_t *buf; uint64_t len, cap; uint32_t *lens;
} part_t;

static void *run_part(void *arg) {
  part_t *t = (part_t *)arg;
  gen_fn fn = pick(t->cfg);
  t->cap = t->n * 160 + MAX_REC; t->buf = (uint8_t *)malloc(t->cap);
  t->lens = (uint32_t *)malloc(sizeof(uint32_t) * (t->n ? t->n : 1));
  for (uint64_t i = 0; i < t->n; i++) {
    if (t->len + MAX_REC > t->cap) { t->cap = t->cap * 2; t->buf = (uint8_t *)realloc(t->buf, t->cap); }
    uint8_t *e = fn(t->buf + t->len, t->seed, t->start + i);
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
