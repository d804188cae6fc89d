/* ORACLE (test infrastructure, not product code) -- C walker.
 *
 * CPU restatement of the reference's direct-decode hot path,
 *   ruhvro/src/fast_decode.rs:420-922  (decode / append_null / byte readers)
 *   ruhvro/src/deserialize.rs:53-68,76-121 (chunking + one task per chunk)
 * with arrow-rs 58.3.0 builder semantics (Cargo.lock:86-87; not vendored):
 * lazy null bitmaps, zero under nulls, repeated offsets, LSB-first bitmaps.
 *
 * Same algorithm as oracle/py_walker.py, fast enough for 10^6..10^7 records;
 * it is also the "port" CPU baseline bench.py times beside the GPU path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped engine never links or calls it.
 *
 * Build: gcc -O3 -shared -fPIC -pthread oracle_walk.c -o _build/liboracle_walk.so
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
  K_INT, K_LONG, K_FLOAT, K_DOUBLE, K_BOOL, K_STRING, K_DATE, K_TSMILLI, K_TSMICRO,
  K_ENUM, K_NULL, K_RECORD, K_UNION, K_LIST, K_MAP
};

typedef struct {
  int32_t kind, nullable, null_first;
  int32_t nchildren, first_child; /* children = child_idx[first_child .. +nchildren) */
  int32_t nsymbols, first_symbol; /* symbols  = sym_off[first_symbol .. +nsymbols]   */
} orc_node;

/* ---- growable buffers -------------------------------------------------- */
typedef struct { uint8_t *p; size_t len, cap; } bytes_t;
static void by_reserve(bytes_t *b, size_t extra) {
  if (b->len + extra <= b->cap) return;
  size_t nc = b->cap ? b->cap * 2 : 256;
  while (nc < b->len + extra) nc *= 2;
  b->p = (uint8_t *)realloc(b->p, nc);
  b->cap = nc;
}
static inline void by_push(bytes_t *b, const void *src, size_t n) {
  by_reserve(b, n);
  memcpy(b->p + b->len, src, n);
  b->len += n;
}
/* bit builder (BooleanBufferBuilder) */
typedef struct { bytes_t b; size_t nbits; } bits_t;
static inline void bits_push(bits_t *x, int v) {
  if ((x->nbits & 7) == 0) { by_reserve(&x->b, 1); x->b.p[x->b.len++] = 0; }
  if (v) x->b.p[x->nbits >> 3] |= (uint8_t)(1u << (x->nbits & 7));
  x->nbits++;
}
/* NullBufferBuilder: bitmap materialised on the first null */
typedef struct { bits_t bits; int materialized; size_t len, nulls; } nulls_t;
static void nulls_materialize(nulls_t *n) {
  if (n->materialized) return;
  n->materialized = 1;
  for (size_t i = 0; i < n->len; i++) bits_push(&n->bits, 1);
}
static inline void nulls_push(nulls_t *n, int valid) {
  if (!valid) { nulls_materialize(n); n->nulls++; }
  if (n->materialized) bits_push(&n->bits, valid);
  n->len++;
}

/* ---- one builder per node (FieldDecoder) -------------------------------- */
typedef struct builder {
  const orc_node *node;
  size_t length;
  nulls_t nulls;      /* leaf lazy null buffer, or record/list/map BooleanBufferBuilder */
  bytes_t values;     /* fixed-width values, or string bytes */
  bits_t bvalues;     /* boolean values */
  bytes_t offsets;    /* int32 offsets (strings, lists, maps) */
  bytes_t type_ids;   /* union */
  int32_t cur_offset; /* list/map */
  struct builder **kids;
  struct builder *keys; /* map */
} builder;

typedef struct {
  const orc_node *nodes; const int32_t *child_idx;
  const uint8_t *sym_data; const int32_t *sym_off;
} schema_t;

enum { E_OK = 0, E_EOB, E_VARINT, E_EOB_F32, E_EOB_F64, E_BOOL, E_NEGLEN, E_EOB_STR, E_ENUM, E_BRANCH, E_UNION };
typedef struct { int code; int64_t detail; } err_t;

static orc_node g_string_node = { K_STRING, 0, 0, 0, 0, 0, 0 };

static builder *mk_builder(const schema_t *s, int idx, size_t cap) {
  builder *b = (builder *)calloc(1, sizeof(builder));
  b->node = idx < 0 ? &g_string_node : &s->nodes[idx];
  int k = b->node->kind;
  if (k == K_STRING || k == K_ENUM || k == K_LIST || k == K_MAP) {
    int32_t z = 0; by_push(&b->offsets, &z, 4);
  }
  if (k == K_RECORD || k == K_LIST || k == K_MAP) b->nulls.materialized = 1; /* plain BooleanBufferBuilder */
  /* capacity hints, as with_capacity(cap) / (cap, cap*16) in fast_decode.rs:178-194,226,257-260 */
  if (k == K_STRING || k == K_ENUM) { by_reserve(&b->values, cap * 16); by_reserve(&b->offsets, (cap + 1) * 4); }
  else if (k == K_LIST || k == K_MAP) by_reserve(&b->offsets, (cap + 1) * 4);
  else if (k == K_INT || k == K_DATE || k == K_FLOAT) by_reserve(&b->values, cap * 4);
  else if (k == K_LONG || k == K_TSMILLI || k == K_TSMICRO || k == K_DOUBLE) by_reserve(&b->values, cap * 8);
  else if (k == K_UNION) by_reserve(&b->type_ids, cap);
  if (b->node->nchildren) {
    b->kids = (builder **)calloc(b->node->nchildren, sizeof(builder *));
    for (int i = 0; i < b->node->nchildren; i++) b->kids[i] = mk_builder(s, s->child_idx[b->node->first_child + i], cap);
  }
  if (k == K_MAP) b->keys = mk_builder(s, -1, cap);
  return b;
}
static void free_builder(builder *b) {
  if (!b) return;
  for (int i = 0; i < b->node->nchildren; i++) free_builder(b->kids[i]);
  free_builder(b->keys);
  free(b->kids); free(b->nulls.bits.b.p); free(b->values.p); free(b->bvalues.b.p);
  free(b->offsets.p); free(b->type_ids.p); free(b);
}

/* ---- byte readers (fast_decode.rs:845-922) ------------------------------ */
typedef struct { const uint8_t *p, *e; } cur_t;

static inline int read_zigzag_long(cur_t *c, int64_t *out, err_t *er) {
  uint64_t result = 0; uint32_t shift = 0;
  for (;;) {
    if (c->p >= c->e) { er->code = E_EOB; return 1; }
    uint8_t byte = *c->p++;
    result |= (uint64_t)(byte & 0x7F) << shift;
    if ((byte & 0x80) == 0) { *out = (int64_t)(result >> 1) ^ -(int64_t)(result & 1); return 0; }
    shift += 7;
    if (shift >= 64) { er->code = E_VARINT; return 1; }
  }
}
static inline int read_string(cur_t *c, const uint8_t **s, size_t *n, err_t *er) {
  int64_t len;
  if (read_zigzag_long(c, &len, er)) return 1;
  if (len < 0) { er->code = E_NEGLEN; return 1; }
  if ((uint64_t)(c->e - c->p) < (uint64_t)len) { er->code = E_EOB_STR; return 1; }
  *s = c->p; *n = (size_t)len; c->p += len;
  return 0;
}
static inline int union_branch(cur_t *c, int null_first, int *is_value, err_t *er) { /* 585-593 */
  int64_t idx;
  if (read_zigzag_long(c, &idx, er)) return 1;
  if (idx == 0) { *is_value = !null_first; return 0; }
  if (idx == 1) { *is_value = null_first; return 0; }
  er->code = E_BRANCH; er->detail = idx; return 1;
}
static inline int read_block_count(cur_t *c, int64_t *n, err_t *er) { /* 689-700 */
  if (read_zigzag_long(c, n, er)) return 1;
  if (*n < 0) { int64_t sz; if (read_zigzag_long(c, &sz, er)) return 1; *n = (int64_t)(0 - (uint64_t)*n); /* wrapping: i64::MIN stays negative -> `0..n` is empty */ }
  return 0;
}

static inline void push_str(builder *b, const uint8_t *s, size_t n) {
  by_push(&b->values, s, n);
  int32_t off = (int32_t)b->values.len;
  by_push(&b->offsets, &off, 4);
  nulls_push(&b->nulls, 1);
  b->length++;
}

static void append_null(builder *b);
static int decode(const schema_t *s, builder *b, cur_t *c, err_t *er);

static int record_present(const schema_t *s, builder *b, cur_t *c, err_t *er) { /* 595-606 */
  if (b->node->nullable) bits_push(&b->nulls.bits, 1);
  b->length++;
  for (int i = 0; i < b->node->nchildren; i++)
    if (decode(s, b->kids[i], c, er)) return 1;
  return 0;
}

static void append_null(builder *b) { /* 503-534 */
  int k = b->node->kind;
  b->length++;
  switch (k) {
    case K_NULL: return;
    case K_RECORD: /* 608-616 */
      if (b->node->nullable) bits_push(&b->nulls.bits, 0);
      for (int i = 0; i < b->node->nchildren; i++) append_null(b->kids[i]);
      return;
    case K_UNION: { /* 660-668 */
      for (int i = 0; i < b->node->nchildren; i++) append_null(b->kids[i]);
      int8_t z = 0; by_push(&b->type_ids, &z, 1);
      return;
    }
    case K_LIST: case K_MAP: /* 721-727, 764-770 */
      by_push(&b->offsets, &b->cur_offset, 4);
      if (b->node->nullable) bits_push(&b->nulls.bits, 0);
      return;
    case K_STRING: case K_ENUM: {
      int32_t off = (int32_t)b->values.len;
      by_push(&b->offsets, &off, 4);
      nulls_push(&b->nulls, 0);
      return;
    }
    case K_BOOL: bits_push(&b->bvalues, 0); nulls_push(&b->nulls, 0); return;
    case K_INT: case K_DATE: case K_FLOAT: { int32_t z = 0; by_push(&b->values, &z, 4); nulls_push(&b->nulls, 0); return; }
    default: { int64_t z = 0; by_push(&b->values, &z, 8); nulls_push(&b->nulls, 0); return; }
  }
}

static int decode(const schema_t *s, builder *b, cur_t *c, err_t *er) { /* 421-499 */
  const orc_node *n = b->node;
  int k = n->kind;
  if (k == K_NULL) { b->length++; return 0; }
  if (n->nullable) {
    int is_value;
    if (union_branch(c, n->null_first, &is_value, er)) return 1;
    if (!is_value) { append_null(b); return 0; }
  }
  switch (k) {
    case K_INT: case K_DATE: {
      int64_t v; if (read_zigzag_long(c, &v, er)) return 1;
      int32_t t = (int32_t)v; /* `as i32`, truncating */
      by_push(&b->values, &t, 4); nulls_push(&b->nulls, 1); b->length++; return 0;
    }
    case K_LONG: case K_TSMILLI: case K_TSMICRO: {
      int64_t v; if (read_zigzag_long(c, &v, er)) return 1;
      by_push(&b->values, &v, 8); nulls_push(&b->nulls, 1); b->length++; return 0;
    }
    case K_FLOAT:
      if (c->e - c->p < 4) { er->code = E_EOB_F32; return 1; }
      by_push(&b->values, c->p, 4); c->p += 4; nulls_push(&b->nulls, 1); b->length++; return 0;
    case K_DOUBLE:
      if (c->e - c->p < 8) { er->code = E_EOB_F64; return 1; }
      by_push(&b->values, c->p, 8); c->p += 8; nulls_push(&b->nulls, 1); b->length++; return 0;
    case K_BOOL: {
      if (c->p >= c->e) { er->code = E_EOB; return 1; }
      uint8_t v = *c->p++;
      if (v > 1) { er->code = E_BOOL; er->detail = v; return 1; }
      bits_push(&b->bvalues, v); nulls_push(&b->nulls, 1); b->length++; return 0;
    }
    case K_STRING: {
      const uint8_t *p; size_t len;
      if (read_string(c, &p, &len, er)) return 1;
      push_str(b, p, len); return 0;
    }
    case K_ENUM: { /* append_enum 570-578 */
      int64_t v; if (read_zigzag_long(c, &v, er)) return 1;
      uint64_t idx = (uint64_t)v;
      if (idx >= (uint64_t)n->nsymbols) { er->code = E_ENUM; er->detail = v; return 1; }
      int32_t a = s->sym_off[n->first_symbol + idx], z = s->sym_off[n->first_symbol + idx + 1];
      push_str(b, s->sym_data + a, (size_t)(z - a)); return 0;
    }
    case K_RECORD: return record_present(s, b, c, er);
    case K_UNION: { /* 643-658 */
      int64_t idx; if (read_zigzag_long(c, &idx, er)) return 1;
      if (idx < 0 || idx >= n->nchildren) { er->code = E_UNION; er->detail = idx; return 1; }
      for (int i = 0; i < n->nchildren; i++) {
        if (i == idx) { if (decode(s, b->kids[i], c, er)) return 1; }
        else append_null(b->kids[i]);
      }
      int8_t t = (int8_t)idx; by_push(&b->type_ids, &t, 1); b->length++; return 0;
    }
    case K_LIST: case K_MAP: { /* 703-719, 745-762 */
      for (;;) {
        int64_t cnt; if (read_block_count(c, &cnt, er)) return 1;
        if (cnt == 0) break;
        for (int64_t i = 0; i < cnt; i++) {
          if (k == K_MAP) {
            const uint8_t *p; size_t len;
            if (read_string(c, &p, &len, er)) return 1;
            push_str(b->keys, p, len);
          }
          if (decode(s, b->kids[0], c, er)) return 1;
          b->cur_offset++;
        }
      }
      by_push(&b->offsets, &b->cur_offset, 4);
      if (n->nullable) bits_push(&b->nulls.bits, 1);
      b->length++; return 0;
    }
  }
  return 0;
}

static void fmt_err(const err_t *e, char *out, size_t cap) {
  switch (e->code) {
    case E_EOB: snprintf(out, cap, "unexpected end of buffer"); break;
    case E_VARINT: snprintf(out, cap, "zigzag varint too long"); break;
    case E_EOB_F32: snprintf(out, cap, "unexpected end of buffer (f32)"); break;
    case E_EOB_F64: snprintf(out, cap, "unexpected end of buffer (f64)"); break;
    case E_BOOL: snprintf(out, cap, "invalid boolean byte: %lld", (long long)e->detail); break;
    case E_NEGLEN: snprintf(out, cap, "negative string length"); break;
    case E_EOB_STR: snprintf(out, cap, "unexpected end of buffer (string)"); break;
    case E_ENUM: snprintf(out, cap, "enum index %llu out of range", (unsigned long long)e->detail); break;
    case E_BRANCH: snprintf(out, cap, "invalid union branch index: %lld", (long long)e->detail); break;
    case E_UNION: snprintf(out, cap, "union branch index out of range: %lld", (long long)e->detail); break;
    default: out[0] = 0;
  }
}

/* ---- public API ---------------------------------------------------------- */
typedef struct {
  schema_t s;
  int nchunks;
  builder **tops;      /* one root builder per chunk */
  int nnodes;
} orc_result;

typedef struct {
  const schema_t *s; const uint8_t *data; const uint64_t *offsets; uint64_t lo, hi;
  builder *top; err_t err; int failed;
} task_t;

static void *run_task(void *arg) { /* fast_decode.rs:825-828 per chunk */
  task_t *t = (task_t *)arg;
  for (uint64_t i = t->lo; i < t->hi; i++) {
    cur_t c = { t->data + t->offsets[i], t->data + t->offsets[i + 1] };
    if (record_present(t->s, t->top, &c, &t->err)) { t->failed = 1; return NULL; }
  }
  return NULL;
}

/* Decode n records (record i = data[offsets[i] .. offsets[i+1])) into
 * clamp(num_chunks) batches; chunk boundaries per deserialize.rs:53-68.
 * threaded != 0: serial pack of the input (deserialize.rs:90) then one thread
 * per chunk (deserialize.rs:92-120).  Returns NULL and fills err on failure. */
void *orc_decode(const orc_node *nodes, int nnodes, const int32_t *child_idx,
                 const uint8_t *sym_data, const int32_t *sym_off,
                 const uint8_t *data, const uint64_t *offsets, uint64_t n,
                 uint64_t num_chunks, int threaded, char *err, size_t errcap) {
  orc_result *r = (orc_result *)calloc(1, sizeof(orc_result));
  r->s.nodes = nodes; r->s.child_idx = child_idx; r->s.sym_data = sym_data; r->s.sym_off = sym_off;
  r->nnodes = nnodes;
  uint64_t k = num_chunks < 1 ? 1 : num_chunks;         /* clamp_chunks, deserialize.rs:53-55 */
  uint64_t cap = n < 1 ? 1 : n;
  if (k > cap) k = cap;
  r->nchunks = (int)k;
  r->tops = (builder **)calloc(k, sizeof(builder *));
  task_t *tasks = (task_t *)calloc(k, sizeof(task_t));
  uint8_t *packed = NULL;
  const uint8_t *src = data;
  if (threaded) {                                       /* BinaryArray::from_vec: serial memcpy */
    uint64_t total = offsets[n];
    packed = (uint8_t *)malloc(total ? total : 1);
    for (uint64_t i = 0; i < n; i++) memcpy(packed + offsets[i], data + offsets[i], offsets[i + 1] - offsets[i]);
    src = packed;
  }
  uint64_t sz = n / k;                                  /* build_slices, deserialize.rs:57-68 */
  for (uint64_t i = 0; i < k; i++) {
    tasks[i].s = &r->s; tasks[i].data = src; tasks[i].offsets = offsets;
    tasks[i].lo = i * sz; tasks[i].hi = (i == k - 1) ? n : (i + 1) * sz;
    r->tops[i] = mk_builder(&r->s, 0, (size_t)(tasks[i].hi - tasks[i].lo));
    tasks[i].top = r->tops[i];
  }
  if (threaded && k > 1) {
    pthread_t *th = (pthread_t *)calloc(k, sizeof(pthread_t));
    for (uint64_t i = 0; i < k; i++) pthread_create(&th[i], NULL, run_task, &tasks[i]);
    for (uint64_t i = 0; i < k; i++) pthread_join(th[i], NULL);
    free(th);
  } else {
    for (uint64_t i = 0; i < k; i++) { run_task(&tasks[i]); if (tasks[i].failed) break; }
  }
  int failed = 0;
  for (uint64_t i = 0; i < k && !failed; i++)           /* handles awaited in chunk order: first error wins */
    if (tasks[i].failed) { fmt_err(&tasks[i].err, err, errcap); failed = 1; }
  free(tasks); free(packed);
  if (failed) {
    for (uint64_t i = 0; i < k; i++) free_builder(r->tops[i]);
    free(r->tops); free(r);
    return NULL;
  }
  return r;
}

int orc_num_chunks(void *h) { return ((orc_result *)h)->nchunks; }

void orc_free(void *h) {
  orc_result *r = (orc_result *)h;
  if (!r) return;
  for (int i = 0; i < r->nchunks; i++) free_builder(r->tops[i]);
  free(r->tops); free(r);
}

typedef struct {
  uint64_t length, null_count;
  const uint8_t *valid; uint64_t valid_bits;   /* packed LSB-first; NULL if never materialised */
  const uint8_t *values; uint64_t values_len;  /* fixed-width values or string bytes */
  const uint8_t *bvalues; uint64_t bvalues_bits;
  const uint8_t *offsets; uint64_t offsets_len;
  const uint8_t *type_ids; uint64_t type_ids_len;
} orc_view;

static builder *find(builder *b, const orc_node *base, int idx, int want_keys) {
  if (b->node != &g_string_node && (int)(b->node - base) == idx) return want_keys ? b->keys : b;
  for (int i = 0; i < b->node->nchildren; i++) {
    builder *r = find(b->kids[i], base, idx, want_keys);
    if (r) return r;
  }
  return NULL;
}

int orc_view_node(void *h, int chunk, int node_idx, int want_keys, orc_view *v) {
  orc_result *r = (orc_result *)h;
  builder *b = find(r->tops[chunk], r->s.nodes, node_idx, want_keys);
  if (!b) return 1;
  memset(v, 0, sizeof(*v));
  v->length = b->length;
  v->null_count = b->nulls.nulls;
  if (b->nulls.materialized) { v->valid = b->nulls.bits.b.p; v->valid_bits = b->nulls.bits.nbits; }
  v->values = b->values.p; v->values_len = b->values.len;
  v->bvalues = b->bvalues.b.p; v->bvalues_bits = b->bvalues.nbits;
  v->offsets = b->offsets.p; v->offsets_len = b->offsets.len;
  v->type_ids = b->type_ids.p; v->type_ids_len = b->type_ids.len;
  return 0;
}
