/* ruhvro_hip.h -- C ABI of the MI355X-native Avro -> Arrow direct-decode engine.
 *
 * Drop-in boundary for ONE path of Tyler-Sch/pyruhvro: the chunked direct
 * decode that `pyruhvro.deserialize_array_threaded` reaches.  Each entry point
 * names the reference interface it replaces (paths relative to the reference
 * repository root).  Plain pointers and sizes only; no torch / pybind types.
 *
 * Conventions: functions returning int return 0 on success; on failure they
 * return non-zero and, when `err` is non-NULL, store a malloc'd NUL-terminated
 * message in *err (release with rh_free_string).  Decode errors carry exactly
 * the reference's message text (ruhvro/src/fast_decode.rs:575,591,634,646,
 * 849,866,874,884,898,906,910); the Python layer maps them to ValueError as
 * src/lib.rs:25-27 does.  All functions are thread-safe; a compiled schema
 * may be shared by concurrent calls.
 */
#ifndef RUHVRO_HIP_H
#define RUHVRO_HIP_H
#include <stddef.h>
#include <stdint.h>
#include "arrow_c_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RH_ABI_VERSION 7

/* error classes (return codes) */
#define RH_OK 0
#define RH_ERR_SCHEMA 1      /* schema JSON invalid / outside the direct-decode subset */
#define RH_ERR_DECODE 2      /* malformed Avro datum: reference message text in *err  */
#define RH_ERR_RUNTIME 3     /* HIP / allocation failure                               */
#define RH_ERR_ARGUMENT 4

typedef struct rh_schema rh_schema;

/* Replaces ruhvro::deserialize::parse_schema (ruhvro/src/deserialize.rs:18-20)
 * plus the per-call schema work of the fast path: the is_supported gate
 * (ruhvro/src/fast_decode.rs:38-61), to_arrow_schema
 * (ruhvro/src/schema_translate.rs:19-37) and decoder-tree construction
 * (fast_decode.rs:176-414).  The result is immutable; cache it by schema
 * string as src/lib.rs:39-54 does. */
rh_schema* rh_schema_compile(const char* json, size_t len, char** err);
void rh_schema_free(rh_schema* s);

/* Arrow schema of the produced batches, exported as a "+s" struct schema whose
 * children are the batch columns (what arrow-rs hands pyarrow at
 * src/lib.rs:70,88 through its pyarrow FFI).  Caller releases out->release. */
int rh_schema_export(const rh_schema* s, struct ArrowSchema* out);

/* clamp_chunks (ruhvro/src/deserialize.rs:53-55): number of batches a decode
 * of n records with `num_chunks` returns. */
uint32_t rh_clamp_chunks(uint64_t n, uint64_t num_chunks);

/* The multi-GPU deal (see rh_opts.devices): shard `shard` of `n_shards` owns chunks [*chunk_lo, *chunk_hi) of the
 * k = rh_clamp_chunks(n, num_chunks) reference chunks, i.e. rows [*row_lo, *row_hi) (build_slices,
 * deserialize.rs:57-68: k-1 chunks of n/k rows, the last one takes the remainder).  A shard may be empty (k < g). */
void rh_shard_chunks(uint64_t n, uint64_t num_chunks, uint32_t n_shards, uint32_t shard,
                     uint32_t* chunk_lo, uint32_t* chunk_hi, uint64_t* row_lo, uint64_t* row_hi);

typedef struct rh_stats rh_stats;

/* Call options.  Zero-initialise (memset / `rh_opts o = {0}`) and set `device = -1` for "current device"; fields
 * after `stream` were added in ABI version 3 and are all "0 = off". */
typedef struct {
  int32_t device;        /* HIP device ordinal; -1 = current device          */
  int32_t flags;         /* RH_KERNEL_* below; 0 = automatic                 */
  void* stream;          /* hipStream_t to launch on; NULL = engine's stream */
  /* Multi-GPU form of the chunk driver (rh_decode / rh_decode_packed): the reference deals its k chunks to a thread
   * pool (ruhvro/src/deserialize.rs:92-120); here the k chunks are dealt to `n_devices` shards in contiguous runs --
   * shard j decodes chunks [j*k/g, (j+1)*k/g) on HIP device devices[j], with its own host thread, streams and arenas
   * -- so every returned batch is produced entirely by one GPU and the result is identical for every g.  Ordinals may
   * repeat (several logical shards on one GPU).  NULL / 0 = the single device named by `device`. */
  const int32_t* devices;
  uint32_t n_devices;
  /* sizeof(rh_opts) as the caller compiled it, or 0.  0 = the ABI-3 layout, which ends with `device_stats`: the engine
   * then never reads the fields behind it, so a caller built against the shorter struct stays safe (the field was
   * `reserved0`, must-be-zero, in ABI 3).  Set it to sizeof(rh_opts) to use `ready` / `gathered`. */
  uint32_t struct_size;
  /* Explicit chunk geometry for a caller that decodes a RANGE of a larger call's chunks (one process per GPU, each
   * holding whole reference chunks): when non-zero, the n records are `num_chunks` chunks of `chunk_rows` rows, the
   * last one taking the rest (deserialize.rs:57-68 applied to the whole list, not to this range).  0 = the
   * reference's split of these n records (n / num_chunks). */
  uint64_t chunk_rows;
  rh_stats* device_stats; /* optional, room for n_devices entries: per-shard stage timings of a multi-GPU call */
  /* Streaming hand-over of the record slices (rh_decode only, ABI version 4): a producer thread of the caller fills
   * ptrs[] / lens[] front to back WHILE the call runs and publishes its progress in *ready -- the number of leading
   * entries that are valid (monotonic, stored with release order; UINT64_MAX = the producer gave up, the call fails
   * with RH_ERR_ARGUMENT).  The engine gathers a chunk group into pinned memory as soon as its entries are there, so
   * the H2D copy, the kernels and the D2H copy of the first groups overlap the producer's work on the later ones.
   * *gathered (optional) is the engine's answer: the number of leading records whose BYTES it has copied and will
   * not read again -- the caller may let go of them (the CPython boundary drops its references while the tail of the
   * call is still on the PCIe link).  NULL / NULL = every entry is valid at entry, the classic call.  Read only when
   * struct_size covers them. */
  const uint64_t* ready;
  uint64_t* gathered;
} rh_opts;

/* Kernel selection (rh_opts.flags).  Both forms are HIP kernels running the same field handlers and
 * produce identical buffers.  AUTO uses the kernels specialised to the schema when their code objects
 * are in the kernel cache, the generic schema-program interpreter otherwise -- and a call large enough
 * to be worth it (RUHVRO_HIP_SPECIALIZE_MIN records, default 32768) that misses the cache starts the
 * compile IN THE BACKGROUND (ABI version 6): that call and the ones after it run on the generic kernels
 * at once, the first call after the compile jobs finish switches over (rh_stats.specialized says which
 * form ran; rh_schema_kernels_ready waits for the switch).  A new schema therefore costs what it costs
 * the reference (src/lib.rs:39-54: a parse), not a compile.  RH_KERNEL_SPECIALIZED insists: the call
 * waits for the compile, or fails if there is no compiler. */
#define RH_KERNEL_AUTO 0
#define RH_KERNEL_GENERIC 1
#define RH_KERNEL_SPECIALIZED 2
/* rh_decode_device only (ABI version 4), OR-ed into rh_opts.flags: return as soon as the call is ON THE STREAM (size
 * pass, scan + arena layout, emit pass and the read-back of its control words all enqueued), without waiting for it.
 * The result's device buffers are valid in stream order on rh_opts.stream, like any asynchronous HIP work; what the
 * HOST learns from a call -- the first malformed record (the in-order join of deserialize.rs:115-119), row counts,
 * null counts, an arena that has to be re-laid-out -- is settled by rh_device_result_wait(), which every accessor of
 * an unsettled result also runs first.  Until then the input buffers (d_data, d_offsets) must stay alive AND UNMODIFIED,
 * and the compiled schema alive: the emit pass re-reads the records the size pass cleared and, in tiles without
 * anomalies, walks them with no bounds or anomaly predicate of its own (walk.h RH_TRUST) -- bytes that change between
 * the two passes (a caller recycling d_data on another stream) turn into unchecked stores.  Work that reuses the
 * buffers must be ordered behind the call on rh_opts.stream, or wait for rh_device_result_wait.  (RUHVRO_HIP_NO_TRUST=1
 * makes the emit pass walk every tile with its own checks -- a debugging aid, ~1.3x the kernel time.)  A schema's first call on a
 * device (no size history to reserve the arena from) completes synchronously whatever the flag says.  This is how a
 * pipeline of small batches keeps the GPU busy: a 1M-record call is 0.15 ms of kernels, and a synchronous call adds
 * ~25 us of host turn-around during which the GPU idles. */
#define RH_ASYNC 8
/* rh_decode_device only (ABI version 5), OR-ed into rh_opts.flags: the form of the launch sequence.
 * Default = two passes: a size pass (per-record sizes, tile totals), a scan + arena layout, an emit pass; every record is read
 * from HBM twice and 24 bytes of counters per record travel between the passes (4.8 GB per 10M benchmark records).
 * RH_SINGLE_PASS: prefer the single-pass form -- ONE kernel that sizes a tile, scans across the tiles of its chunk
 * (decoupled look-back, fence-free) and emits out of the same LDS window, into an arena laid out from per-column capacities
 * (the schema's size history): 3.0 GB per 10M records, 1.02 x the algorithmic bytes.  Needs the schema-specialised kernels
 * and a size history (a schema's first call is two-pass); a call that outgrows a capacity is repeated on the two-pass form
 * and the schema backs off.  Same buffers either way.  Measured on MI355X (profiles/r04v_*): about time-neutral at 10M
 * records (0.96-1.00 ms against 0.99-1.01), slower on small launches (the look-back wait is a fixed cost per tile) -- which
 * is why it is opt-in: its use is a deployment that is short of HBM bandwidth, not of time.  RUHVRO_HIP_SINGLE_PASS=1 makes it
 * the preference of every call; RH_TWO_PASS forces the two-pass form regardless. */
#define RH_TWO_PASS 16
#define RH_SINGLE_PASS 32

struct rh_stats {
  uint64_t records;
  uint64_t input_bytes;      /* Avro payload bytes                               */
  uint64_t output_bytes;     /* Arrow buffer bytes produced (all chunks)         */
  uint32_t chunks;
  uint32_t blocks;           /* workgroups per kernel launch                     */
  float pack_ms;             /* host gather of the record slices                 */
  float h2d_ms;
  float size_kernel_ms;      /* k_size   (walk 1: per-record sizes)              */
  float scan_kernel_ms;      /* k_scan   (segmented exclusive scans)             */
  float emit_kernel_ms;      /* k_emit   (walk 2: column materialisation)        */
  float d2h_ms;
  float total_ms;
  uint32_t specialized;      /* 1 = schema-specialised kernels ran, 0 = generic interpreter */
  uint32_t lds_bytes;        /* dynamic LDS per workgroup of the emit kernel                 */
};

/* Replaces ruhvro::deserialize::per_datum_deserialize_threaded
 * (ruhvro/src/deserialize.rs:76-121; single-chunk form per_datum_deserialize,
 * deserialize.rs:25-30, is num_chunks = 1).  Input: n record slices owned by
 * the caller for the duration of the call (src/lib.rs:29-33,84).  Output:
 * out_chunks[0..*out_k) are struct arrays (one per chunk, chunk boundaries of
 * deserialize.rs:57-68, order preserved) in host memory; the caller provides
 * room for rh_clamp_chunks(n, num_chunks) entries and releases each through
 * ArrowArray.release.  The first malformed record (lowest index) aborts the
 * call with its message, as the in-order join at deserialize.rs:115-119 does. */
int rh_decode(const rh_schema* s, const uint8_t* const* ptrs, const uint64_t* lens,
              uint64_t n, uint64_t num_chunks, const rh_opts* opts,
              struct ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, char** err);

/* Same, for input already packed the way the reference packs it internally
 * (BinaryArray, deserialize.rs:90): one contiguous buffer + n+1 offsets. */
int rh_decode_packed(const rh_schema* s, const uint8_t* data, const uint64_t* offsets,
                     uint64_t n, uint64_t num_chunks, const rh_opts* opts,
                     struct ArrowArray* out_chunks, uint32_t* out_k, rh_stats* stats, char** err);

/* Device-resident form of the same path: `d_data`/`d_offsets` (u64[n+1]) live
 * in HBM (d_data 16-byte aligned, data_len = offsets[n]), the Arrow buffers are
 * produced in HBM and stay there.  This is the kernel-only path bench.py times.
 * The result owns its device memory. */
typedef struct rh_device_result rh_device_result;
int rh_decode_device(const rh_schema* s, const void* d_data, const void* d_offsets,
                     uint64_t data_len, uint64_t n, uint64_t num_chunks, const rh_opts* opts,
                     rh_device_result** out, rh_stats* stats, char** err);
/* Settle an RH_ASYNC call: waits for its stream, returns what the synchronous call would have returned (RH_OK, or
 * RH_ERR_DECODE with the reference's message for the lowest malformed record ...).  `stats` (optional) receives the
 * stage timings when the call was made with a non-NULL stats argument (which is itself not written by an asynchronous
 * call).  A no-op returning RH_OK on a result that is already settled.  Not thread-safe per result. */
int rh_device_result_wait(rh_device_result* r, rh_stats* stats, char** err);
uint32_t rh_device_result_chunks(const rh_device_result* r);
/* Exact (unpadded) Arrow buffer bytes the call produced, all chunks.  Settles an RH_ASYNC result first; 0 when that
 * call failed (rh_device_result_wait reports the message -- this accessor has no error channel). */
uint64_t rh_device_result_output_bytes(const rh_device_result* r);
/* Arrow C Device Data Interface view of chunk i (device_type ARROW_DEVICE_ROCM);
 * valid until rh_device_result_free; release the view through array.release. */
int rh_device_result_export(rh_device_result* r, uint32_t chunk, struct ArrowDeviceArray* out);
/* ABI version 6.  The device buffers of chunk i with their byte sizes (the Arrow C Device Data structs carry pointers and lengths
 * only): fills ptrs[] / sizes[] (either may be NULL) with up to `cap` entries, one per Arrow buffer of the schema, and returns how
 * many buffers a chunk has.  sizes[] are the bytes the engine reserved for the buffer (offsets: 4 x (rows + 1); bitmaps: whole
 * 64-bit words; data: the column's bytes), so a consumer can wrap buffers_of(export(i)) as typed device arrays without reading
 * an offsets buffer back -- what pyruhvro_amd.deserialize_to_device does for DLPack consumers.  0 on error. */
uint32_t rh_device_result_buffers(rh_device_result* r, uint32_t chunk, uint64_t* ptrs, uint64_t* sizes, uint32_t cap);
/* Copy the chunks to host memory (same form rh_decode returns). */
int rh_device_result_to_host(rh_device_result* r, struct ArrowArray* out_chunks, char** err);
void rh_device_result_free(rh_device_result* r);

/* Schema-specialised kernel management.  rh_schema_kernel_source returns the generated HIP source
 * (malloc'd, release with rh_free_string).  rh_schema_prebuild generates, compiles (hiprtc, gfx950;
 * needs no GPU) and stores the code objects (decode pair and encode pair) in the kernel cache so later
 * processes only load them; returns 0 and sets *cached = 1 when both were already there. */
char* rh_schema_kernel_source(const rh_schema* s);
/* Content hash (hex, malloc'd) of this schema's specialised decode (encode = 0) or encode (1) kernel pair: generated
 * source + every device header it includes.  It is the kernel-cache key; measurement files (profiles/hbm_traffic.json)
 * are stamped with it so that a number is only ever attributed to the kernel it was measured on. */
char* rh_schema_kernel_key(const rh_schema* s, int encode);
char* rh_schema_encode_kernel_source(const rh_schema* s);      /* the Arrow -> Avro pair (rh_encode) */
int rh_schema_prebuild(const rh_schema* s, int* cached, char** err);
/* ABI version 6.  State of the specialised decode (encode = 0) or encode (1) kernels of this schema: 1 = their code
 * objects are there (the next call runs on them), 0 = not yet (still compiling after timeout_ms, or nobody asked for them:
 * no call of RUHVRO_HIP_SPECIALIZE_MIN records yet and nothing in the kernel cache), -1 = the compile failed (*err).
 * Waits up to timeout_ms for running compile jobs (0 = just look, < 0 = no limit).  A service that wants its first batch
 * at full speed calls rh_schema_prebuild at start-up instead; this is for the ones that would rather start serving. */
int rh_schema_kernels_ready(const rh_schema* s, int encode, long timeout_ms, char** err);

/* Arrow -> Avro, the other direction.  Replaces ruhvro::serialize::serialize_record_batch
 * (ruhvro/src/serialize.rs:38-67) + fast_encode::serialize_chunk (ruhvro/src/fast_encode.rs:27-53): `batch` is
 * the record batch as a struct array (Arrow C Data Interface, any offsets / slices), columns are matched to the
 * schema's fields BY NAME (fast_encode.rs:151-185).  out_chunks[0..*out_k) are BinaryArrays ("z": i32 offsets +
 * data, one datum per row) in host memory, chunked like serialize.rs:19-30; the caller provides room for
 * rh_clamp_chunks(batch->length, num_chunks) entries and releases each through ArrowArray.release.  Data-dependent
 * failures return RH_ERR_DECODE with the reference's message text (fast_encode.rs:173-177, 541, 576). */
int rh_encode(const rh_schema* s, const struct ArrowArray* batch, const struct ArrowSchema* batch_schema,
              uint64_t num_chunks, const rh_opts* opts, struct ArrowArray* out_chunks, uint32_t* out_k,
              rh_stats* stats, char** err);

/* Device-resident form of rh_encode (the mirror of rh_decode_device): `batch` is a struct array whose BUFFER POINTERS
 * ARE DEVICE POINTERS on the call's device (what rh_device_result_export hands out: ArrowDeviceArray.array), read in
 * place -- no host copy.  Every buffer must be followed by at least 64 readable bytes (Arrow's own allocation padding;
 * the arenas of rh_decode_device are).  The k BinaryArrays ("z": i32 offsets + data, chunked like serialize.rs:19-30)
 * are produced in HBM and stay there: export chunk i as an ArrowDeviceArray (ARROW_DEVICE_ROCM), or copy all to the
 * host in the form rh_encode returns.  This is the kernel-only path `bench.py --direction encode` times. */
typedef struct rh_device_encoded rh_device_encoded;
int rh_encode_device(const rh_schema* s, const struct ArrowArray* batch, const struct ArrowSchema* batch_schema,
                     uint64_t num_chunks, const rh_opts* opts, rh_device_encoded** out, rh_stats* stats, char** err);
uint32_t rh_device_encoded_chunks(const rh_device_encoded* r);
uint64_t rh_device_encoded_output_bytes(const rh_device_encoded* r);   /* exact: i32 offsets + Avro bytes, all chunks */
int rh_device_encoded_export(rh_device_encoded* r, uint32_t chunk, struct ArrowDeviceArray* out);
int rh_device_encoded_to_host(rh_device_encoded* r, struct ArrowArray* out_chunks, char** err);
void rh_device_encoded_free(rh_device_encoded* r);

/* Process-wide counters of the engine's rarely taken branches (monotonic; tests read them before and after a call to
 * prove that the branch they aim at really ran).  Fills out[0..n) with as many of the RH_CTR_* values as fit; returns
 * RH_CTR_COUNT. */
enum {
  RH_CTR_FUSED_CALLS = 0,      /* decode calls that took the single-submission path (device-side arena layout)      */
  RH_CTR_TWO_SYNC_CALLS = 1,   /* ... that laid the arena out on the host between scan and emit (first call / env)  */
  RH_CTR_CAPACITY_RETRIES = 2, /* LF_CAPACITY: the arena reserved from the size history was too small, tail re-run  */
  RH_CTR_WIDE_FALLBACKS = 3,   /* NeedWideIndex: a call re-run on the generic kernels (64-bit in-buffer offsets)    */
  RH_CTR_OFFSET32_ERRORS = 4,  /* calls refused because a chunk's column exceeds 32-bit Arrow offsets               */
  RH_CTR_SPLIT_CALLS = 5,      /* device-resident calls that dealt their chunk groups to internal streams (in-call overlap) */
  RH_CTR_SINGLE_PASS_CALLS = 6, /* decode calls that took the single-pass form (one kernel sizes, scans across tiles and emits) */
  RH_CTR_SINGLE_PASS_FAILOVERS = 7, /* ... of which outgrew a column capacity and were repeated on the two-pass form */
  RH_CTR_BACKGROUND_COMPILES = 8, /* kernel compile jobs started behind a call that went ahead on the generic kernels (ABI 6) */
  /* ABI 7: what the walks met, summed over the tiles of every single-submission call by its last kernel (rh_k_publish reads the
   * size pass's per-tile flags; a call's first run on a schema and a call repeated with an exact arena are not counted).  A tile
   * is one workgroup's records (256, or 64 for wide schemas); all of these are 0 on input that stays inside the fast wire forms. */
  RH_CTR_TILES = 9,             /* tiles of the calls counted below                                                           */
  RH_CTR_CAREFUL_TILES = 10,    /* tiles the emit pass walked with the careful form (an anomaly in the size pass, or no window)  */
  RH_CTR_OVER_WINDOW_TILES = 11, /* tiles whose bytes did not fit the LDS window in one piece                                  */
  RH_CTR_REWALKED_WAVES = 12,   /* wavefronts the size pass walked twice (a record outside the fast wire forms, or malformed)  */
  RH_CTR_SUBTILED_TILES = 13,   /* over-window tiles that were staged through the window in record ranges (not walked from HBM) */
  RH_CTR_RANGED_RETRIES = 14,   /* calls repeated on the generic kernels: a tile past the window met specialised kernels whose ranged pair was not loaded yet */
  RH_CTR_COUNT = 15
};
uint32_t rh_engine_counters(uint64_t* out, uint32_t n);

/* Measurement hook, not part of the drop-in surface: the host-side gather of a `shards`-GPU rh_decode call (every shard's
 * record slices copied into its staging buffer by its own host thread + `threads_per_shard` helpers, all shards at once)
 * without the GPUs -- pageable destinations (pinned = 0, runs on a box with no device), pinned ones (1), or pageable ones placed
 * per NUMA node with the shard's threads bound there (2; rh_numa_nodes() = how many nodes the kernel shows).  *best_ms = best
 * wall time of `reps` rounds; returns the payload bytes per round, 0 on failure.  scripts/gather_scaling.py. */
uint64_t rh_bench_gather(const uint8_t* const* ptrs, const uint64_t* lens, uint64_t n, uint32_t shards,
                         uint32_t threads_per_shard, int pinned, uint32_t reps, double* best_ms);
uint32_t rh_numa_nodes(void);

/* Test hook, not part of the drop-in surface: exercises the host thread pool behind the gather of pipelined calls (its
 * lock-free phase hand-over); 0 = every task of every phase ran exactly once.  tests/test_host.py. */
uint32_t rh_selftest_pool(uint32_t workers, uint32_t phases, uint32_t max_tasks);

void rh_free_string(char* s);
int rh_abi_version(void);
/* CPUs the process may use: hardware threads cut down to its affinity mask and its cgroup CPU quota -- what the engine
 * sizes its host thread pools by, and what a binding should size its own by (the CPython boundary's extractor threads do). */
uint32_t rh_effective_cpus(void);
/* Number of visible HIP devices (0 when no GPU / driver); never throws. */
int rh_device_count(void);
/* The calling thread's current HIP device (what rh_opts.device = -1 resolves to on this thread), or -1 when there is
 * none.  The current device is per host thread: a binding that moves a call to a thread of its own resolves -1 with
 * this BEFORE it starts the thread (the CPython boundary does, pymodule.cpp).  Never throws. */
int rh_current_device(void);

#ifdef __cplusplus
}
#endif
#endif
