// Host-side schema front-end of the direct-decode engine:
//   Avro schema JSON -> Avro type tree -> (a) Arrow output schema,
//   (b) decoder tree (one node per reference FieldDecoder), (c) the flat
//   schema program + output-buffer table the kernels run.
// Reference behaviour restated here: apache_avro::Schema::parse_str as reached
// from ruhvro/src/deserialize.rs:18-20, ruhvro/src/schema_translate.rs:19-266,
// ruhvro/src/fast_decode.rs:38-61 (gate) and 176-414 (decoder construction).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "program.h"

namespace rh {

struct SchemaError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- Avro type tree -------------------------------------------------------
enum AvroKind {
  AV_NULL, AV_BOOLEAN, AV_INT, AV_LONG, AV_FLOAT, AV_DOUBLE, AV_BYTES, AV_STRING,
  AV_RECORD, AV_ENUM, AV_ARRAY, AV_MAP, AV_UNION, AV_FIXED,
  AV_DATE, AV_TS_MILLIS, AV_TS_MICROS,
  AV_TIME_MILLIS, AV_TIME_MICROS, AV_DECIMAL, AV_UUID, AV_DURATION,   // SURVEY 8f N4 (with AV_BYTES / AV_FIXED)
  AV_OTHER_LOGICAL,   // local-timestamp-*, timestamp-nanos (no Arrow mapping in the reference: schema_translate.rs:144)
  AV_REF,
};

struct AvroType;
struct AvroField {
  std::string name;
  bool has_doc = false;
  std::string doc;
  std::unique_ptr<AvroType> type;
};

struct AvroType {
  AvroKind kind = AV_NULL;
  std::string name, ns;                  // named types
  bool has_doc = false;
  std::string doc;
  bool has_aliases = false;
  std::vector<std::string> aliases;
  std::vector<AvroField> fields;         // record
  std::vector<std::string> symbols;      // enum
  std::unique_ptr<AvroType> items;       // array items / map values
  std::vector<std::unique_ptr<AvroType>> variants;  // union
  std::string logical;                   // AV_OTHER_LOGICAL: its name (for messages)
  int64_t size = 0;                      // fixed: bytes | decimal / uuid on a fixed base: its size (-1: bytes / string base)
  int precision = 0, scale = 0;          // decimal
  std::string fullname() const { return ns.empty() ? name : ns + "." + name; }
};

// ---- Arrow output schema ----------------------------------------------------
struct ArrowField {
  std::string name;
  std::string format;                    // Arrow C data interface format string
  bool nullable = false;
  bool map_keys_sorted = false;
  std::vector<std::pair<std::string, std::string>> metadata;
  std::vector<ArrowField> children;
};

// ---- decoder tree -------------------------------------------------------------
enum NodeKind {
  NK_FIXED, NK_STRING, NK_ENUM, NK_NULL, NK_RECORD, NK_UNION, NK_LIST, NK_MAP,
  NK_BIN,       // fixed / decimal / uuid (OP_BIN): validity + a values buffer of `bin_width` bytes per row
};

struct DecNode {
  NodeKind kind = NK_NULL;
  int32_t fixed = 0;          // FixedKind for NK_FIXED
  bool nullable = false;      // Nullable* decoder (2-variant null union collapsed)
  bool null_first = false;
  bool can_null = false;      // receives append_null somewhere -> leaf owns a lazy validity bitmap
  int dom = 0;                // row domain of this node's rows
  int child_dom = 0;          // NK_LIST / NK_MAP: domain of the items
  std::vector<int> children;  // record fields / union variants / [item] / [value]
  int keys = -1;              // NK_MAP: node id of the synthetic keys column
  int sym_first = 0, sym_count = 0;
  // output buffers (ids into CompiledSchema::bufs, -1 = none)
  int buf_validity = -1;
  int buf_main = -1;          // values / value bits / offsets / type_ids
  int buf_data = -1;          // string bytes
  int counter = -1;           // string byte counter id
  int bin_width = 0;          // NK_BIN: bytes per row
};

struct CompiledSchema {
  std::string json;
  std::unique_ptr<AvroType> avro;
  ArrowField arrow;                 // "+s" struct whose children are the batch columns
  std::vector<DecNode> nodes;       // nodes[0] = top-level record
  std::vector<Op> prog;
  std::vector<BufDesc> bufs;
  std::vector<uint32_t> sym_off;
  std::vector<uint8_t> sym_data;
  int K = 0;                        // counters: [0, ndom-1) row domains 1.., then string byte columns
  int KL = 0;                       // ... of which the first KL live per lane (all of them unless `wide`)
  bool wide = false;                // more than kWideCounters counters: tiles of one wavefront, wave counters (program.h F_WAVE_CTR)
  int ndom = 1;
  int list_depth = 0;
  int max_nest = 0, max_union_depth = 0;   // deepest nullable-record / union / list nesting, deepest N-variant union nesting (program.h kShallow*)
  uint32_t min_record_bytes = 0;
  uint32_t max_row_bytes = 16;      // widest fixed-width value of one row (bounds the 32-bit in-buffer offsets of the specialised kernels)
  std::string encode_unsupported;   // non-empty: why rh_encode does not take this schema (no such schema today)
};

// Throws SchemaError.  `json` need not be NUL-terminated.
std::unique_ptr<CompiledSchema> compile_schema(const char* json, size_t len);

}  // namespace rh
