// Schema program: the flat, wave-uniform form of the reference's decoder tree
// (FieldDecoder / RecordDecoder / UnionDecoder / ListDecoder / MapDecoder,
// ruhvro/src/fast_decode.rs:73-167).  Built on the host by schema.cpp, run by
// the interpreter in kernels.hip.  Shared between host and device code.
#pragma once
#ifdef __HIPCC_RTC__   // hiprtc has no system headers
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#else
#include <stdint.h>
#endif

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define RH_HD __host__ __device__
#else
#define RH_HD
#endif

namespace rh {

constexpr int kBlock = 256;          // threads (= records) per workgroup
constexpr int kWideCounters = 64;    // schemas with more scanned counters (row domains + byte columns) are compiled WIDE: no limit on their number
constexpr int kWideTile = 64;        // ... records per tile of a wide schema: one wavefront
// Nesting (walk.h stk_t / sel_t): a specialised kernel of a schema inside the SHALLOW bounds keeps 32-bit bit stacks and a 64-bit
// selector stack; deeper schemas -- to the DEEP bounds -- and the interpreter take 64-bit / 128-bit ones (RH_DEEP).  The deep
// bounds are beyond what apache-avro can parse into anything the reference decodes: serde_json stops at 128 JSON levels, a
// nullable record costs four of them, an N-variant union of records five.
constexpr int kShallowNest = 31, kShallowUnionDepth = 8;
constexpr int kMaxListDepth = 62;    // nested array/map levels (one `remaining` counter per level and lane)
constexpr int kMaxNest = 63;         // nested nullable-record / union / list levels (bit stacks)
constexpr int kMaxUnionDepth = 16;   // nested N-variant unions (8-bit selector stack)
// Null counts are added per workgroup with global atomics.  Atomics on ONE address from all over the chip serialise at
// ~100 ns each (measured: a schema with two nullable columns, 10M records, 8 chunks = 4883 workgroups per address: the
// emit kernel took 0.476 ms with them and 0.200 ms without, profiles/r03ag_nullcount_slots_ab.txt), so every
// (node, chunk) count is spread over kNullSlots addresses by tile index and summed on the host.
constexpr int kNullSlots = 64;
// Slots a call really uses: the control block is nnodes x k x slots words, device side and again in pinned memory, and the
// reference lets num_chunks go up to n -- with many chunks the per-chunk workgroup count (and with it the contention a
// slot relieves) falls as fast as the block grows, so the slot count is scaled down with k (a power of two, >= 1).
RH_HD inline uint32_t null_slots_for(uint32_t k) {
  uint32_t s = kNullSlots;
  while (s > 1 && (uint64_t)s * k > 4096) s >>= 1;
  return s;
}

enum FixedKind : int32_t { FK_I32 = 0, FK_I64 = 1, FK_F32 = 2, FK_F64 = 3, FK_BOOL = 4 };

// OP_BIN sub-kinds (types the reference's schema translation maps -- schema_translate.rs:58,133-137 --
// but its direct-decode gate rejects, fast_decode.rs:59; decoded here from the Avro 1.11 wire rules)
enum BinKind : int32_t {
  BN_FIXED = 0,      // fixed(N): N raw bytes                      -> FixedSizeBinary(N)   (also uuid on fixed(16))
  BN_DEC_BYTES = 1,  // decimal on bytes: length + big-endian two's complement -> Decimal128 (16 bytes, little-endian)
  BN_DEC_FIXED = 2,  // decimal on fixed(N <= 16): N bytes big-endian two's complement -> Decimal128
  BN_UUID_STR = 3,   // uuid on string: 36-char hyphenated (or 32-char simple) hex text -> FixedSizeBinary(16)
  BN_DURATION = 4,   // duration on fixed(12): months / days / milliseconds, little-endian u32 each -> Duration(ms) (months must be 0)
};

enum OpCode : int32_t {
  OP_END = 0,
  OP_FIXED,        // int/long/float/double/boolean/date/timestamp leaf       (fast_decode.rs:424-432,434-473)
  OP_STRING,       // string leaf, also map keys                              (429,454-457,752)
  OP_ENUM,         // enum -> symbol text                                     (433,474-479,570-578)
  OP_REC_BEGIN,    // nullable record: branch + validity, children follow     (482-485,595-616)
  OP_REC_END,
  OP_UNION_BEGIN,  // N-variant union: branch -> type_id                      (643-668)
  OP_VARIANT,      // start of variant i's ops
  OP_UNION_END,
  OP_LIST_BEGIN,   // array / map (optionally nullable)                       (487-496,703-727,745-770)
  OP_LIST_NEXT,    // block header / loop head                                (689-700)
  OP_LIST_TAIL,    // end of one item
  OP_LIST_END,     // push offset
  OP_BIN,          // fixed / decimal / uuid leaf (BinKind in a, wire bytes in b, output bytes per row in c)
};

enum OpFlags : int32_t {
  F_NULLABLE = 1,    // 2-variant null union collapsed onto this node (Nullable* decoders)
  F_NULL_FIRST = 2,  // union was ["null", T]
  F_CAN_NULL = 4,    // node can receive append_null -> it owns a validity bitmap
  F_IS_MAP = 8,
  F_WAVE_CTR = 16,   // OP_STRING / OP_ENUM of a wide schema in domain 0: the byte counter is a WAVE counter (scanned on the spot, no per-lane state)
};

struct Op {
  int32_t code;
  int32_t flags;
  int32_t dom;    // row domain the node's rows live in (0 = records)
  int32_t a;      // FIXED: FixedKind | STRING/ENUM: counter id of the byte column | UNION_BEGIN: #variants
                  // VARIANT: variant index | LIST_*: child row domain
  int32_t b;      // ENUM: first index into sym_off | LIST_BEGIN/NEXT: pc of LIST_END | LIST_TAIL: pc of LIST_NEXT
  int32_t c;      // ENUM: #symbols | LIST_*: list nesting depth
  int32_t buf0;   // validity bitmap buffer id, or -1
  int32_t buf1;   // values / offsets / type_ids buffer id, or -1
  int32_t buf2;   // string data buffer id, or -1 | LIST_NEXT: min wire bytes per item (0 = zero-width items)
  int32_t node;   // node id (null-count slot)
};

// decode error codes (message text: fast_decode.rs:575,591,646,849,866,874,884,898,906,910)
enum ErrCode : uint32_t {
  E_OK = 0, E_EOB, E_VARINT, E_EOB_F32, E_EOB_F64, E_BOOL, E_NEGLEN, E_EOB_STR, E_ENUM, E_BRANCH, E_UNION,
  E_LIST_RANGE,   // zero-width items with a block count beyond the i32 offset range (no reference message)
  E_INTERNAL,     // the fast emit walk met a wire form the size pass had not flagged (cannot happen; never silent)
  E_EOB_FIXED,    // N4 types (no reference message: the reference never decodes them)
  E_DECIMAL,      // a decimal of more than 16 bytes
  E_UUID,         // uuid text that is not 32 / 36 hex characters
  E_DURATION,     // a duration with a non-zero months component: Duration(ms) has no value for it (detail = months)
};

struct ErrInfo {
  uint32_t code;
  uint32_t pad;
  int64_t detail;
};

enum BufKind : int32_t { BK_BITMAP = 0, BK_VAL4, BK_VAL8, BK_I8, BK_OFFSETS, BK_DATA, BK_FIXW };

struct BufDesc {
  int32_t kind;
  int32_t dom;      // row domain
  int32_t counter;  // BK_DATA: counter id giving its byte length | BK_FIXW: bytes per row
  int32_t node;
};

// Arena slot of one Arrow buffer of one chunk: `alloc` bytes are reserved (exact Arrow size, bitmaps rounded up to
// whole 64-bit words so a wavefront's ballot store never leaves the slot), `exact` is the Arrow size.  ONE statement of
// the layout rule, used by the device-side layout kernel (rh_k_layout) and by the host when it exports the buffers.
RH_HD inline uint64_t buf_bytes(int32_t kind, uint64_t rows, uint64_t data_total, uint64_t* exact, uint32_t width = 0) {
  uint64_t sz = 0, ex = 0;
  switch (kind) {
    case BK_FIXW: sz = ex = rows * width; break;
    case BK_BITMAP: sz = (rows + 63) / 64 * 8; ex = (rows + 7) / 8; break;
    case BK_VAL4: sz = ex = rows * 4; break;
    case BK_VAL8: sz = ex = rows * 8; break;
    case BK_I8: sz = ex = rows; break;
    case BK_OFFSETS: sz = ex = (rows + 1) * 4; break;
    case BK_DATA: sz = ex = data_total; break;
  }
  if (exact) *exact = ex;
  return sz;
}
constexpr uint64_t kBufAlign = 256;          // every buffer of every chunk starts on a 256-byte boundary
RH_HD inline uint64_t buf_slot_bytes(uint64_t sz) { return ((sz < 8 ? 8 : sz) + kBufAlign - 1) / kBufAlign * kBufAlign; }

// Control words at the head of a call's workspace (device), copied to the host once at the end of the call.
//   [0] u64 first_bad   ~index of the lowest malformed record, 0 if none (k_size / k_emit, atomicMax)
//   [1] u32 layout_flag set by rh_k_layout: the emit pass must not run (k_init / k_emit return at once)
enum LayoutFlag : uint32_t {
  LF_CAPACITY = 1,     // the arena reserved from the schema's size history is too small: the host re-runs the tail exactly
  LF_OFFSET32 = 2,     // a chunk's column exceeds the 2^31-1 limit of 32-bit Arrow offsets
  LF_NEED_WIDE = 4,    // a child row domain reaches 2^28 rows: only the generic kernels index that far
  LF_NEED_RANGED = 8,  // (set by rh_spec_size) a tile does not fit the LDS window and the ranged kernels were not launched: the host repeats the call
};

// Per-tile word the size pass leaves for the emit pass (KParams::tileflag) and for the call's statistics (rh_k_publish)
enum TileFlag : uint32_t {
  TF_SATURATED = 1,     // a per-record counter does not fit its 16-bit hand-over slot: the emit pass sizes the tile again
  TF_CAREFUL = 2,       // the emit pass walks this tile with the careful form
  TF_OVER_WINDOW = 4,   // the tile's bytes do not fit the LDS window in one piece
  TF_SUBTILED = 8,      // ... and were staged through it in record ranges
  TF_REWALK_ONE = 256,  // bits 8..15: wavefronts the size pass walked twice
};

// Parameters of rh_k_layout (one workgroup): exact arena layout on the device from the scanned totals, so that a
// call is ONE stream submission (k_size -> k_scan -> k_layout -> k_init -> k_emit) with no host round trip in between.
struct LParams {
  const uint64_t* totals;    // [K][k] from k_scan
  const BufDesc* desc;       // [nbuf]
  uint64_t sz, rows_last;    // chunk geometry (rows)
  uint64_t n;
  uint32_t k;
  int32_t nbuf, K, ndom;
  uint8_t* arena;
  uint64_t capacity;         // bytes reserved at `arena`
  void** bufptr;             // out [k][nbuf]
  uint64_t* bufsize;         // out [k][nbuf] (allocated bytes)
  unsigned long long* ctrl;  // control words (see above); ctrl[2] receives the arena bytes used
  uint32_t narrow;           // 1: the schema-specialised kernels will run (32-bit in-buffer byte offsets)
  uint64_t narrow_rows;      // ... which index child row domains below this many rows (2^28, less with wide fixed columns)
};

// Kernel parameters (one launch = all chunks of one call on one device).
struct KParams {
  const uint8_t* data;       // packed Avro payload
  const uint64_t* offsets;   // n+1 record offsets into data
  uint64_t data_len;
  uint64_t n;                // records
  uint64_t sz;               // rows per chunk except the last (deserialize.rs:58)
  uint64_t rows_last;        // rows of the last chunk
  uint32_t k;                // chunks
  uint32_t bpc;              // workgroups per (non-last) chunk
  uint32_t nblocks;
  uint32_t win_bytes;        // LDS bytes reserved for the input window

  const Op* prog;
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  int32_t nops;
  int32_t K;                 // counters
  int32_t KL;                // ... of which the first KL are per-lane counters; the rest are wave counters (wide schemas: KL < K, tile = kWideTile)
  uint32_t tile;             // records (= threads) per tile of this launch
  int32_t ndom;              // row domains (>= 1)
  int32_t nnodes;
  int32_t list_depth;        // max list nesting (LDS rem[] rows)
  int32_t nbuf;              // output buffers per chunk
  const int32_t* cnt_databuf;  // [K] string counter -> its BK_DATA buffer id (-1 for row-domain counters)

  uint32_t* blocksum;        // [K][nblocks]
  uint32_t* blockbase;       // [K][nblocks] chunk-relative exclusive prefix
  uint64_t* totals;          // [K][k]
  unsigned long long* first_bad;  // control words (see LayoutFlag): [0] = ~(lowest failing record index), 0 if none (atomicMax)
  ErrInfo* errinfo;          // [nblocks]
  void* const* bufptr;       // [k][nbuf]
  uint32_t* nullcount;       // [nnodes][k][null_slots]: a workgroup adds into slot (tile & (null_slots - 1)); rh_k_publish / the host sum the slots
  // specialised kernels only: k_size leaves every record's counters behind so k_emit does not re-walk
  uint32_t* lanecnt;         // [nblocks*TILE][ceil(K/2)] per-record counters, 16 bits each (saturated at 0xFFFF), record-major
  uint32_t* lanecnt32;       // ranged kernels (tiles past the window): [nblocks*TILE][KL] the same counters, 32 bits each -- a record of such a tile is often larger than 16 bits count
  uint32_t* tileflag;        // [nblocks] TileFlag bits
  unsigned long long* prof;  // [32] phase cycle sums (RUHVRO_HIP_PROFILE=1 builds of the specialised kernels), else null
  // single-pass form (spec_body.h spec_fused): one launch sizes, scans (decoupled look-back over the tiles of a chunk) and emits
  unsigned long long* lookback;   // [nblocks][K] tile state words: bits 63..62 = 1 tile total / 2 inclusive prefix, low 32 bits = value; zero = not there yet
  const uint64_t* caps;      // [K][k] capacity of every counter's column per chunk (rows / bytes the arena reserves for it)
  uint32_t* tickets;         // [k] next tile of every chunk (zero at launch): tiles are taken in the order workgroups START
  uint32_t null_slots;       // null_slots_for(k): a power of two <= kNullSlots
  uint32_t ranged;           // 1: the ranged kernels (rh_spec_size_r / rh_spec_emit_r) follow the size / emit kernel of this call: those leave every tile past the window to them
  uint32_t all_careful;      // 1: no size pass classified the tiles (schemas without variable-length output): the emit kernel walks every tile carefully
  // ranged kernels: LARGE tiles first.  A tile with a record of tens of thousands of items runs for milliseconds on one lane;
  // met last it IS the kernel's tail.  rh_spec_size appends every tile of more than big_tile_bytes that it leaves to the
  // ranged pair to a list -- [0] = its length (zeroed per call), [2 ..] = the tiles -- and notes its place + 1 in bigmark[tile]
  // (zeroed per call); the pair is launched with kBigFront workgroups in FRONT of the usual ones: front workgroup b takes list
  // entry b, the workgroup of a tile's usual place leaves it alone when a front workgroup has it.  Every other tile keeps its
  // place (one atomic per LARGE tile: an append per tile past the window cost the size kernel 0.35 ms of same-address atomics).
  uint32_t* worklist;
  uint32_t* bigmark;         // [nblocks]
  uint64_t big_tile_bytes;   // a tile of more bytes than this is LARGE: kBigTileWindows windows, and as many mean tiles of this call
};
constexpr uint32_t kBigTileWindows = 3;
constexpr uint32_t kBigFront = 4096;

}  // namespace rh
