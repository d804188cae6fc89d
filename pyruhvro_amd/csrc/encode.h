// Parameters of the Arrow -> Avro encode kernels (encode.hip), shared with the host side (engine_encode.cpp).
#pragma once
#include "program.h"

namespace rh {

enum EncErr : uint32_t {
  EE_ENUM = 100,    // "fast_encode: enum symbol '{sym}' not in schema"   (fast_encode.rs:575-577); pad = op index, detail = row
  EE_DECIMAL = 102, // a Decimal128 value that is not an N-byte two's complement number on a fixed(N) base (ours: the reference
                    // does not encode decimals, fast_encode.rs:22); pad = N, detail = row
  EE_DURATION = 103, // a Duration(ms) value below zero or beyond 2^32-1 days + 2^32-1 ms: no months/days/millis form (ours); detail = row
  EE_UNION = 101,   // "fast_encode: union type_id {t} out of range"      (fast_encode.rs:540-542); detail = t
};

struct EParams {
  uint64_t n;               // rows of the batch
  uint64_t sz, rows_last;   // rows per chunk / of the last chunk (serialize.rs:19-30)
  uint32_t k, bpc, nblocks;
  int32_t nbuf, ndom, list_depth;
  const Op* prog;           // the decoder's schema program: its buffer ids name the INPUT buffers here
  const uint32_t* sym_off;
  const uint8_t* sym_data;
  const uint64_t* in_ptr;       // [nbuf] device address of every input buffer rebased to logical row 0 (0 = absent)
  const uint32_t* in_bitoff;    // [nbuf] bit offset of row 0 inside a bitmap's first byte
  uint32_t* blocksum;           // [nblocks] encoded bytes of every workgroup's rows
  const uint32_t* blockbase;    // [nblocks] chunk-relative exclusive prefix of blocksum (rh_k_scan)
  void* const* outptr;          // [k][2]: offsets (i32[rows+1]), data
  uint32_t* rowlen;             // [nblocks*256] encoded length of every row: written by rh_e_size, read by rh_e_emit
  uint32_t win_bytes;           // rh_e_emit: LDS bytes of the output staging window (0 = always store straight to HBM)
  uint32_t stage_bytes;         // rh_espec_emit: LDS bytes behind the window for the waves' string staging areas (0 = none)
  unsigned long long* first_bad;
  ErrInfo* errinfo;             // [nblocks]
  unsigned long long* prof;     // [64][32] phase cycle sums (RUHVRO_HIP_PROFILE=1 builds of the specialised kernels), else null
};

// Per-wave LDS area through which rh_espec_emit transposes the string bytes of 64 consecutive rows (encode_walk.h
// e_string_cofetch): kStageBytes of column data + slack for the 36-byte reads of the last string.
constexpr uint32_t kStageBytes = 1024, kStageStride = 1088;

}  // namespace rh
