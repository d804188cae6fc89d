// See schema.h.  Host only (no HIP).
#include "schema.h"
#include <cmath>

#include <map>
#include <set>

#include "json.hpp"

namespace rh {
namespace {

using json::Value;

// ===========================================================================
// 1. Avro schema JSON -> type tree  (apache_avro::Schema::parse_str subset)
// ===========================================================================
// deep copy of a type tree (a named-type reference becomes a copy of the definition it names)
std::unique_ptr<AvroType> clone_type(const AvroType& t) {
  auto c = std::make_unique<AvroType>();
  c->kind = t.kind; c->name = t.name; c->ns = t.ns; c->has_doc = t.has_doc; c->doc = t.doc;
  c->has_aliases = t.has_aliases; c->aliases = t.aliases; c->symbols = t.symbols; c->logical = t.logical;
  c->size = t.size; c->precision = t.precision; c->scale = t.scale;
  for (const AvroField& f : t.fields) {
    AvroField g;
    g.name = f.name; g.has_doc = f.has_doc; g.doc = f.doc;
    g.type = clone_type(*f.type);
    c->fields.push_back(std::move(g));
  }
  if (t.items) c->items = clone_type(*t.items);
  for (const auto& v : t.variants) c->variants.push_back(clone_type(*v));
  return c;
}

struct Parser {
  std::set<std::string> named;
  // completed definitions of the named types seen so far.  The reference stops at a named-type reference
  // (schema_translate.rs:51 `todo!("Add support for AvroSchema::Ref")`, gated out at fast_decode.rs:59); here a
  // reference is resolved the way the Avro specification defines it -- it IS the type it names -- by substituting a
  // copy of the definition, so everything downstream (gate, Arrow translation, decoder tree) sees an ordinary tree.
  // A reference to a type that is still being defined is a recursive type: no Arrow schema can express it.
  std::map<std::string, std::unique_ptr<AvroType>> defs;   // (copies: a logical type may replace the node it wraps)

  static void split_name(const std::string& raw, const Value* ns_attr, const std::string& enclosing,
                         std::string& simple, std::string& ns) {
    // apache-avro Name::parse: a dotted name carries its namespace; else the
    // "namespace" attribute; else the enclosing namespace.
    size_t dot = raw.rfind('.');
    if (dot != std::string::npos) {
      simple = raw.substr(dot + 1);
      ns = raw.substr(0, dot);
      return;
    }
    simple = raw;
    if (ns_attr && ns_attr->is_string()) ns = ns_attr->str;
    else ns = enclosing;
  }

  std::unique_ptr<AvroType> parse(const Value& j, const std::string& enclosing) {
    if (j.is_string()) return parse_name(j.str, enclosing);
    if (j.is_array()) return parse_union(j, enclosing);
    if (j.is_object()) return parse_complex(j, enclosing);
    throw SchemaError("Must be a JSON string, object or array");
  }

  std::unique_ptr<AvroType> prim(AvroKind k) {
    auto t = std::make_unique<AvroType>();
    t->kind = k;
    return t;
  }

  std::unique_ptr<AvroType> parse_name(const std::string& s, const std::string& enclosing) {
    static const std::map<std::string, AvroKind> prims = {
        {"null", AV_NULL}, {"boolean", AV_BOOLEAN}, {"int", AV_INT}, {"long", AV_LONG},
        {"float", AV_FLOAT}, {"double", AV_DOUBLE}, {"bytes", AV_BYTES}, {"string", AV_STRING}};
    auto it = prims.find(s);
    if (it != prims.end()) return prim(it->second);
    std::string simple, ns;
    split_name(s, nullptr, enclosing, simple, ns);
    std::string full = ns.empty() ? simple : ns + "." + simple;
    if (named.count(full)) {
      auto d = defs.find(full);
      if (d == defs.end())
        throw SchemaError("recursive named type " + full + ": a type that contains itself has no Arrow schema");
      return clone_type(*d->second);   // apache-avro: Schema::Ref, resolved (beyond the reference, schema_translate.rs:51)
    }
    throw SchemaError("Unknown type: " + s);
  }

  static std::string union_key(const AvroType& t) {
    switch (t.kind) {
      case AV_RECORD: case AV_ENUM: case AV_FIXED: case AV_REF: return "named:" + t.fullname();
      default: return "kind:" + std::to_string((int)t.kind) + t.logical;
    }
  }

  std::unique_ptr<AvroType> parse_union(const Value& j, const std::string& enclosing) {
    auto u = prim(AV_UNION);
    std::set<std::string> seen;
    for (auto& v : j.arr) {
      if (v.is_array()) throw SchemaError("Unions may not directly contain a union");
      auto t = parse(v, enclosing);
      if (!seen.insert(union_key(*t)).second) throw SchemaError("Unions cannot contain duplicate types");
      u->variants.push_back(std::move(t));
    }
    return u;
  }

  void register_name(const Value& j, const std::string& enclosing, AvroType& t) {
    const Value* nm = j.get("name");
    if (!nm || !nm->is_string() || nm->str.empty()) throw SchemaError("No `name` field");
    split_name(nm->str, j.get("namespace"), enclosing, t.name, t.ns);
    if (!named.insert(t.fullname()).second)
      throw SchemaError("Two schemas with the same fullname were given: " + t.fullname());
    if (const Value* d = j.get("doc"); d && d->is_string()) { t.has_doc = true; t.doc = d->str; }
    if (const Value* a = j.get("aliases"); a && a->is_array()) {
      t.has_aliases = true;
      for (auto& x : a->arr)
        if (x.is_string()) t.aliases.push_back(x.str);
    }
  }

  std::unique_ptr<AvroType> parse_complex(const Value& j, const std::string& enclosing) {
    const Value* ty = j.get("type");
    if (!ty) throw SchemaError("No `type` in complex type");
    const Value* lt = j.get("logicalType");
    if (lt && lt->is_string()) {
      // (logical name, allowed base kinds).  A mismatch keeps the base type, as apache-avro does.
      struct L { const char* name; AvroKind out; AvroKind base1; AvroKind base2; };
      static const L table[] = {
          {"date", AV_DATE, AV_INT, AV_INT},
          {"timestamp-millis", AV_TS_MILLIS, AV_LONG, AV_LONG},
          {"timestamp-micros", AV_TS_MICROS, AV_LONG, AV_LONG},
          {"time-millis", AV_TIME_MILLIS, AV_INT, AV_INT},
          {"time-micros", AV_TIME_MICROS, AV_LONG, AV_LONG},
          {"timestamp-nanos", AV_OTHER_LOGICAL, AV_LONG, AV_LONG},
          {"local-timestamp-millis", AV_OTHER_LOGICAL, AV_LONG, AV_LONG},
          {"local-timestamp-micros", AV_OTHER_LOGICAL, AV_LONG, AV_LONG},
          {"local-timestamp-nanos", AV_OTHER_LOGICAL, AV_LONG, AV_LONG},
          {"uuid", AV_UUID, AV_STRING, AV_FIXED},
          {"decimal", AV_DECIMAL, AV_BYTES, AV_FIXED},
          {"duration", AV_DURATION, AV_FIXED, AV_FIXED},
      };
      for (auto& e : table) {
        if (lt->str != e.name) continue;
        auto base = parse_plain(j, *ty, enclosing);
        if (base->kind == e.base1 || base->kind == e.base2) {
          auto t = prim(e.out);
          t->logical = e.name;
          t->size = base->kind == AV_FIXED ? base->size : -1;
          if (e.out == AV_UUID && base->kind == AV_FIXED && base->size != 16) return base;   // uuid needs 16 bytes
          if (e.out == AV_DURATION && base->size != 12) return base;                         // duration is a fixed(12)
          if (e.out == AV_DECIMAL) {
            // precision is required and positive, scale defaults to 0 and must not exceed it, and a fixed base must be
            // able to hold the precision; anything else keeps the underlying type (apache-avro warns and does the same)
            const Value* pv = j.get("precision");
            const Value* sv = j.get("scale");
            const double pr = pv && pv->is_number() ? pv->num : 0, sc = sv && sv->is_number() ? sv->num : 0;
            if (!(pr >= 1 && pr == (double)(int64_t)pr && sc >= 0 && sc == (double)(int64_t)sc && sc <= pr)) return base;
            if (base->kind == AV_FIXED) {
              // max precision of n bytes: floor(log10(2^(8n-1) - 1))
              const double maxp = std::floor((8.0 * (double)base->size - 1.0) * 0.30102999566398120);
              if (pr > maxp) return base;
            }
            t->precision = (int)pr;
            t->scale = (int)sc;
          }
          return t;
        }
        return base;
      }
    }
    return parse_plain(j, *ty, enclosing);
  }

  std::unique_ptr<AvroType> parse_plain(const Value& j, const Value& ty, const std::string& enclosing) {
    if (ty.is_object()) return parse_complex(ty, enclosing);
    if (ty.is_array()) return parse_union(ty, enclosing);
    if (!ty.is_string()) throw SchemaError("No `type` in complex type");
    const std::string& t = ty.str;
    if (t == "record") {
      auto r = prim(AV_RECORD);
      register_name(j, enclosing, *r);
      const Value* fs = j.get("fields");
      if (!fs || !fs->is_array()) throw SchemaError("No `fields` in record");
      std::set<std::string> seen;
      for (auto& f : fs->arr) {
        const Value* fn = f.get("name");
        if (!fn || !fn->is_string()) throw SchemaError("No `name` in record field");
        const Value* ft = f.get("type");
        if (!ft) throw SchemaError("No `type` in record field");
        if (!seen.insert(fn->str).second) throw SchemaError("Duplicate field name " + fn->str);
        AvroField af;
        af.name = fn->str;
        if (const Value* d = f.get("doc"); d && d->is_string()) { af.has_doc = true; af.doc = d->str; }
        af.type = parse(*ft, r->ns);
        r->fields.push_back(std::move(af));
      }
      defs[r->fullname()] = clone_type(*r);
      return r;
    }
    if (t == "enum") {
      auto e = prim(AV_ENUM);
      register_name(j, enclosing, *e);
      const Value* sy = j.get("symbols");
      if (!sy || !sy->is_array()) throw SchemaError("No `symbols` field in enum");
      std::set<std::string> seen;
      for (auto& s : sy->arr) {
        if (!s.is_string()) throw SchemaError("No `symbols` field in enum");
        if (!seen.insert(s.str).second) throw SchemaError("Duplicate enum symbol " + s.str);
        e->symbols.push_back(s.str);
      }
      defs[e->fullname()] = clone_type(*e);
      return e;
    }
    if (t == "array") {
      const Value* it = j.get("items");
      if (!it) throw SchemaError("No `items` in array");
      auto a = prim(AV_ARRAY);
      a->items = parse(*it, enclosing);
      return a;
    }
    if (t == "map") {
      const Value* it = j.get("values");
      if (!it) throw SchemaError("No `values` in map");
      auto m = prim(AV_MAP);
      m->items = parse(*it, enclosing);
      return m;
    }
    if (t == "fixed") {
      auto f = prim(AV_FIXED);
      register_name(j, enclosing, *f);
      const Value* sz = j.get("size");
      if (!sz || !sz->is_number() || sz->num < 0 || sz->num != (double)(int64_t)sz->num)
        throw SchemaError("No `size` in fixed");
      f->size = (int64_t)sz->num;
      defs[f->fullname()] = clone_type(*f);
      return f;
    }
    return parse_name(t, enclosing);
  }
};

// ===========================================================================
// 2. gate (fast_decode.rs:38-61)
// ===========================================================================
bool supported_inner(const AvroType& t, std::string& why) {
  switch (t.kind) {
    case AV_INT: case AV_LONG: case AV_FLOAT: case AV_DOUBLE: case AV_BOOLEAN: case AV_STRING: case AV_NULL:
    case AV_DATE: case AV_TS_MILLIS: case AV_TS_MICROS: case AV_ENUM:
      return true;
    case AV_RECORD:
      for (auto& f : t.fields)
        if (!supported_inner(*f.type, why)) return false;
      return true;
    case AV_UNION:
      for (auto& v : t.variants)
        if (!supported_inner(*v, why)) return false;
      return true;
    case AV_ARRAY: case AV_MAP:
      return supported_inner(*t.items, why);
    // beyond the reference's gate (fast_decode.rs:59 sends these to a fallback whose column builder
    // answers unimplemented!(), complex.rs:414-431); decoded on the GPU from the Avro 1.11 wire rules
    case AV_BYTES: case AV_TIME_MILLIS: case AV_TIME_MICROS: case AV_UUID: case AV_DURATION:
      return true;
    case AV_FIXED:
      if (t.size > (1 << 20)) { why = "fixed of more than 1 MiB"; return false; }
      return true;
    case AV_DECIMAL:
      if (t.precision > 38) { why = "decimal precision beyond Decimal128 (38 digits)"; return false; }
      if (t.size > 16) { why = "decimal on a fixed of more than 16 bytes"; return false; }
      return true;
    case AV_REF: why = "named-type reference " + t.fullname(); return false;
    default: why = t.logical.empty() ? "unsupported type" : t.logical; return false;
  }
}

// ===========================================================================
// 3. Arrow schema (schema_translate.rs:43-266)
// ===========================================================================
const char* default_field_name(const std::string& fmt) {   // schema_translate.rs:155-220
  if (fmt == "n") return "null";
  if (fmt == "b") return "bit";
  if (fmt == "i") return "int";
  if (fmt == "l") return "bigint";
  if (fmt == "f") return "float4";
  if (fmt == "g") return "float8";
  if (fmt == "tdD") return "dateday";
  if (fmt == "tsm:") return "timestampmilli";
  if (fmt == "tsu:") return "timestampmicro";
  if (fmt == "u") return "varchar";
  if (fmt == "z") return "varbinary";
  if (fmt.rfind("w:", 0) == 0) return "fixedsizebinary";
  if (fmt.rfind("d:", 0) == 0) return "decimal";
  if (fmt == "ttm") return "timemilli";
  if (fmt == "ttu") return "timemicro";
  if (fmt == "tDm") return "duration";
  if (fmt == "+l") return "list";
  if (fmt == "+s") return "struct";
  if (fmt.rfind("+us:", 0) == 0) return "union";
  if (fmt == "+m") throw SchemaError("Map support not implemented (map as an anonymous union variant)");
  throw SchemaError("data type missing default name");
}

std::string aliased(const std::string& alias, const std::string& ns) {   // schema_translate.rs:269-280
  if (alias.find('.') != std::string::npos) return alias;
  if (!ns.empty()) return ns + "." + alias;
  return alias;
}

std::vector<std::pair<std::string, std::string>> external_props(const AvroType& t) {   // 222-266
  std::vector<std::pair<std::string, std::string>> p;
  if (t.kind == AV_RECORD || t.kind == AV_ENUM || t.kind == AV_FIXED) {
    if (t.has_doc) p.emplace_back("avro::doc", t.doc);
    if (t.has_aliases) {
      std::string s = "[";
      for (size_t i = 0; i < t.aliases.size(); i++) {
        if (i) s += ",";
        s += aliased(t.aliases[i], t.ns);
      }
      s += "]";
      p.emplace_back("avro::aliases", s);
    }
  }
  return p;
}

// `name == nullptr` <=> Rust `None` (anonymous: default_field_name / enum fullname).
ArrowField to_field(const AvroType& t, const std::string* name, bool nullable,
                    const std::vector<std::pair<std::string, std::string>>* props) {
  ArrowField f;
  switch (t.kind) {
    case AV_REF: throw SchemaError("Add support for AvroSchema::Ref");
    case AV_NULL: f.format = "n"; break;
    case AV_BOOLEAN: f.format = "b"; break;
    case AV_INT: f.format = "i"; break;
    case AV_LONG: f.format = "l"; break;
    case AV_FLOAT: f.format = "f"; break;
    case AV_DOUBLE: f.format = "g"; break;
    case AV_STRING: f.format = "u"; break;
    case AV_DATE: f.format = "tdD"; break;
    case AV_TS_MILLIS: f.format = "tsm:"; break;
    case AV_TS_MICROS: f.format = "tsu:"; break;
    case AV_BYTES: f.format = "z"; break;                                           // schema_translate.rs:58
    case AV_FIXED: f.format = "w:" + std::to_string(t.size); break;                 // :133
    case AV_DECIMAL: f.format = "d:" + std::to_string(t.precision) + "," + std::to_string(t.scale); break;   // :134-136
    case AV_UUID: f.format = "w:16"; break;                                         // :137
    case AV_TIME_MILLIS: f.format = "ttm"; break;                                   // :139
    case AV_TIME_MICROS: f.format = "ttu"; break;                                   // :140
    case AV_DURATION: f.format = "tDm"; break;                                      // :143 Duration(Millisecond)
    case AV_ARRAY: {                                   // schema_translate.rs:60-65
      f.format = "+l";
      static const std::string item = "item";
      f.children.push_back(to_field(*t.items, &item, true, nullptr));
      break;
    }
    case AV_MAP: {                                     // schema_translate.rs:66-75
      static const std::string values = "values";
      ArrowField value_field = to_field(*t.items, &values, false, nullptr);
      ArrowField key_field;
      key_field.name = "keys";
      key_field.format = "u";
      key_field.nullable = false;
      ArrowField entries;
      entries.name = "entries";
      entries.format = "+s";
      entries.nullable = nullable;
      entries.children.push_back(std::move(key_field));
      entries.children.push_back(std::move(value_field));
      f.format = "+m";
      f.children.push_back(std::move(entries));
      break;
    }
    case AV_UNION: {                                   // schema_translate.rs:76-105
      bool has_null = false;
      for (auto& v : t.variants) has_null |= v->kind == AV_NULL;
      if (has_null && t.variants.size() == 2) {
        nullable = true;
        const AvroType* inner = nullptr;
        for (auto& v : t.variants)
          if (v->kind != AV_NULL) { inner = v.get(); break; }
        if (!inner) throw SchemaError("Avro union contains duplicate null variants");
        ArrowField in = to_field(*inner, nullptr, true, nullptr);
        f.format = in.format;
        f.children = std::move(in.children);
        f.map_keys_sorted = in.map_keys_sorted;
      } else {
        if (has_null) nullable = true;
        std::string fmt = "+us:";
        for (size_t i = 0; i < t.variants.size(); i++) {
          f.children.push_back(to_field(*t.variants[i], nullptr, true, nullptr));
          if (i) fmt += ",";
          fmt += std::to_string(i);
        }
        if (t.variants.size() > 127) throw SchemaError("union has more than 127 variants");
        f.format = fmt;
      }
      break;
    }
    case AV_RECORD: {                                  // schema_translate.rs:106-123
      f.format = "+s";
      for (auto& fd : t.fields) {
        std::vector<std::pair<std::string, std::string>> p;
        if (fd.has_doc) p.emplace_back("avro::doc", fd.doc);
        f.children.push_back(to_field(*fd.type, &fd.name, nullable, &p));
      }
      break;
    }
    case AV_ENUM: {                                    // schema_translate.rs:124-132: early return, no metadata
      f.format = "u";
      f.name = (name && !name->empty()) ? *name : t.fullname();
      f.nullable = nullable;
      return f;
    }
    default:
      throw SchemaError("schema is outside the direct-decode path");
  }
  f.name = name ? *name : std::string(default_field_name(f.format));
  f.nullable = nullable;
  if (props) f.metadata = *props;
  return f;
}

// ===========================================================================
// 4. decoder tree + program (fast_decode.rs:176-414)
// ===========================================================================
struct Builder {
  CompiledSchema& cs;
  struct Ctx { int dom; bool nullfill; int list_depth; int nest; int union_depth; };

  int new_node(NodeKind k) {
    cs.nodes.emplace_back();
    cs.nodes.back().kind = k;
    return (int)cs.nodes.size() - 1;
  }
  int new_buf(BufKind kind, int dom, int counter, int node) {
    BufDesc d;
    d.kind = kind; d.dom = dom; d.counter = counter; d.node = node;
    cs.bufs.push_back(d);
    return (int)cs.bufs.size() - 1;
  }
  std::vector<int> str_counter_ops;   // op indices whose `a` is a local string counter index
  std::vector<int> data_bufs;         // buffer ids whose counter is a local string counter index
  int nstr = 0;

  void push(const Op& op) { cs.prog.push_back(op); }
  static Op mk(OpCode code) {
    Op o;
    o.code = code; o.flags = 0; o.dom = 0; o.a = 0; o.b = 0; o.c = 0;
    o.buf0 = -1; o.buf1 = -1; o.buf2 = -1; o.node = -1;
    return o;
  }

  static uint32_t min_bytes(const AvroType& t) {
    switch (t.kind) {
      case AV_NULL: return 0;
      case AV_FLOAT: return 4;
      case AV_DOUBLE: return 8;
      case AV_FIXED: return (uint32_t)t.size;
      case AV_DECIMAL: case AV_UUID: return t.size >= 0 ? (uint32_t)t.size : 1u;
      case AV_DURATION: return 12;
      case AV_RECORD: {
        uint32_t s = 0;
        for (auto& f : t.fields) s += min_bytes(*f.type);
        return s;
      }
      default: return 1;   // varint / bool byte / union branch / string length / block terminator
    }
  }

  // split_null_union, fast_decode.rs:404-414
  static const AvroType* null_union_inner(const AvroType& u, bool& null_first) {
    if (u.variants.size() != 2) return nullptr;
    if (u.variants[0]->kind == AV_NULL) { null_first = true; return u.variants[1].get(); }
    if (u.variants[1]->kind == AV_NULL) { null_first = false; return u.variants[0].get(); }
    return nullptr;
  }

  int string_leaf(NodeKind kind, bool nullable, bool null_first, const Ctx& cx, const AvroType* en) {
    int id = new_node(kind);
    {
      DecNode& n = cs.nodes[id];
      n.nullable = nullable; n.null_first = null_first; n.dom = cx.dom;
      n.can_null = nullable || cx.nullfill;
    }
    int sidx = nstr++;
    int bv = cs.nodes[id].can_null ? new_buf(BK_BITMAP, cx.dom, -1, id) : -1;
    int bo = new_buf(BK_OFFSETS, cx.dom, -1, id);
    int bd = new_buf(BK_DATA, cx.dom, sidx, id);
    data_bufs.push_back(bd);
    DecNode& n = cs.nodes[id];
    n.buf_validity = bv; n.buf_main = bo; n.buf_data = bd; n.counter = sidx;
    Op o = mk(kind == NK_ENUM ? OP_ENUM : OP_STRING);
    o.flags = (nullable ? F_NULLABLE : 0) | (null_first ? F_NULL_FIRST : 0) | (n.can_null ? F_CAN_NULL : 0);
    o.dom = cx.dom; o.a = sidx; o.buf0 = bv; o.buf1 = bo; o.buf2 = bd; o.node = id;
    if (en) {
      n.sym_first = (int)cs.sym_off.size();
      n.sym_count = (int)en->symbols.size();
      for (auto& s : en->symbols) {
        cs.sym_off.push_back((uint32_t)cs.sym_data.size());
        cs.sym_data.insert(cs.sym_data.end(), s.begin(), s.end());
      }
      cs.sym_off.push_back((uint32_t)cs.sym_data.size());
      o.b = n.sym_first; o.c = n.sym_count;
    }
    str_counter_ops.push_back((int)cs.prog.size());
    push(o);
    return id;
  }

  // make_decoder / make_nullable_decoder / make_union_decoder
  int build(const AvroType& t, bool nullable, bool null_first, Ctx cx) {
    if (cx.nest > kMaxNest) throw SchemaError("schema nesting too deep for the GPU decoder");
    if (cx.nest > cs.max_nest) cs.max_nest = cx.nest;
    if (cx.union_depth > cs.max_union_depth) cs.max_union_depth = cx.union_depth;
    switch (t.kind) {
      case AV_FIXED: case AV_DECIMAL: case AV_UUID: case AV_DURATION: {
        int id = new_node(NK_BIN);
        const int32_t sub = t.kind == AV_FIXED ? BN_FIXED
                            : t.kind == AV_DURATION ? BN_DURATION
                            : t.kind == AV_DECIMAL ? (t.size >= 0 ? BN_DEC_FIXED : BN_DEC_BYTES)
                            : (t.size >= 0 ? BN_FIXED : BN_UUID_STR);
        const int32_t wire = t.size >= 0 ? (int32_t)t.size : 0;
        const int32_t width = t.kind == AV_FIXED ? (int32_t)t.size : t.kind == AV_DURATION ? 8 : 16;
        {
          DecNode& n = cs.nodes[id];
          n.nullable = nullable; n.null_first = null_first; n.dom = cx.dom;
          n.can_null = nullable || cx.nullfill;
          n.bin_width = width;
        }
        const bool can_null = cs.nodes[id].can_null;
        int bv = can_null ? new_buf(BK_BITMAP, cx.dom, -1, id) : -1;
        int bm = new_buf(BK_FIXW, cx.dom, width, id);
        cs.nodes[id].buf_validity = bv;
        cs.nodes[id].buf_main = bm;
        if ((uint32_t)width > cs.max_row_bytes) cs.max_row_bytes = (uint32_t)width;
        Op o = mk(OP_BIN);
        o.flags = (nullable ? F_NULLABLE : 0) | (null_first ? F_NULL_FIRST : 0) | (can_null ? F_CAN_NULL : 0);
        o.dom = cx.dom; o.a = sub; o.b = wire; o.c = width; o.buf0 = bv; o.buf1 = bm; o.node = id;
        push(o);
        return id;
      }
      case AV_BYTES:
        return string_leaf(NK_STRING, nullable, null_first, cx, nullptr);
      case AV_INT: case AV_DATE: case AV_LONG: case AV_TS_MILLIS: case AV_TS_MICROS: case AV_TIME_MILLIS: case AV_TIME_MICROS:
      case AV_FLOAT: case AV_DOUBLE: case AV_BOOLEAN: {
        int id = new_node(NK_FIXED);
        DecNode& n = cs.nodes[id];
        n.fixed = (t.kind == AV_INT || t.kind == AV_DATE || t.kind == AV_TIME_MILLIS) ? FK_I32
                  : (t.kind == AV_FLOAT) ? FK_F32
                  : (t.kind == AV_DOUBLE) ? FK_F64
                  : (t.kind == AV_BOOLEAN) ? FK_BOOL : FK_I64;
        n.nullable = nullable; n.null_first = null_first; n.dom = cx.dom;
        n.can_null = nullable || cx.nullfill;
        bool can_null = n.can_null;
        int fixed = n.fixed;
        int bv = can_null ? new_buf(BK_BITMAP, cx.dom, -1, id) : -1;
        int bm = new_buf(fixed == FK_BOOL ? BK_BITMAP : (fixed == FK_I32 || fixed == FK_F32) ? BK_VAL4 : BK_VAL8,
                         cx.dom, -1, id);
        cs.nodes[id].buf_validity = bv;
        cs.nodes[id].buf_main = bm;
        Op o = mk(OP_FIXED);
        o.flags = (nullable ? F_NULLABLE : 0) | (null_first ? F_NULL_FIRST : 0) | (can_null ? F_CAN_NULL : 0);
        o.dom = cx.dom; o.a = fixed; o.buf0 = bv; o.buf1 = bm; o.node = id;
        push(o);
        return id;
      }
      case AV_STRING:
        return string_leaf(NK_STRING, nullable, null_first, cx, nullptr);
      case AV_ENUM:
        return string_leaf(NK_ENUM, nullable, null_first, cx, &t);
      case AV_NULL: {
        if (nullable) throw SchemaError("fast_decode: unsupported nullable inner type: Null");
        int id = new_node(NK_NULL);
        cs.nodes[id].dom = cx.dom;
        return id;
      }
      case AV_RECORD: {
        if (t.fields.empty()) throw SchemaError("RecordDecoder produced a record with 0 fields");
        int id = new_node(NK_RECORD);
        cs.nodes[id].nullable = nullable; cs.nodes[id].null_first = null_first; cs.nodes[id].dom = cx.dom;
        cs.nodes[id].can_null = nullable;       // RecordDecoder.nulls is Some only when nullable (fast_decode.rs:363-367)
        Ctx cc = cx;
        cc.nest++;
        if (nullable) {
          int bv = new_buf(BK_BITMAP, cx.dom, -1, id);
          cs.nodes[id].buf_validity = bv;
          Op o = mk(OP_REC_BEGIN);
          o.flags = F_NULLABLE | (null_first ? F_NULL_FIRST : 0) | F_CAN_NULL;
          o.dom = cx.dom; o.buf0 = bv; o.node = id;
          push(o);
          cc.nullfill = true;
        }
        std::vector<int> kids;
        for (auto& f : t.fields) kids.push_back(build(*f.type, false, false, cc));
        cs.nodes[id].children = kids;
        if (nullable) push(mk(OP_REC_END));
        return id;
      }
      case AV_UNION: {
        if (nullable) throw SchemaError("fast_decode: unsupported nullable inner type: Union");
        bool nf = false;
        if (const AvroType* inner = null_union_inner(t, nf)) return build(*inner, true, nf, cx);
        if (cx.union_depth >= kMaxUnionDepth) throw SchemaError("unions nested too deep for the GPU decoder");
        int id = new_node(NK_UNION);
        cs.nodes[id].dom = cx.dom;
        int bt = new_buf(BK_I8, cx.dom, -1, id);
        cs.nodes[id].buf_main = bt;
        Op o = mk(OP_UNION_BEGIN);
        o.dom = cx.dom; o.a = (int32_t)t.variants.size(); o.buf1 = bt; o.node = id;
        push(o);
        Ctx cc = cx;
        cc.nest++; cc.union_depth++; cc.nullfill = true;   // non-selected variants are null-filled (649-655)
        std::vector<int> kids;
        for (size_t i = 0; i < t.variants.size(); i++) {
          Op v = mk(OP_VARIANT);
          v.a = (int32_t)i;
          push(v);
          kids.push_back(build(*t.variants[i], false, false, cc));
        }
        cs.nodes[id].children = kids;
        push(mk(OP_UNION_END));
        return id;
      }
      case AV_ARRAY: case AV_MAP: {
        if (cx.list_depth >= kMaxListDepth) throw SchemaError("arrays/maps nested too deep for the GPU decoder");
        const bool is_map = t.kind == AV_MAP;
        int id = new_node(is_map ? NK_MAP : NK_LIST);
        int child_dom = cs.ndom++;
        {
          DecNode& n = cs.nodes[id];
          n.nullable = nullable; n.null_first = null_first; n.dom = cx.dom; n.child_dom = child_dom;
          n.can_null = nullable;   // ListDecoder/MapDecoder.nulls only for Nullable* (fast_decode.rs:328-337)
        }
        int bv = nullable ? new_buf(BK_BITMAP, cx.dom, -1, id) : -1;
        int bo = new_buf(BK_OFFSETS, cx.dom, -1, id);
        cs.nodes[id].buf_validity = bv;
        cs.nodes[id].buf_main = bo;
        if (cx.list_depth + 1 > cs.list_depth) cs.list_depth = cx.list_depth + 1;

        Op ob = mk(OP_LIST_BEGIN);
        ob.flags = (nullable ? (F_NULLABLE | F_CAN_NULL) : 0) | (null_first ? F_NULL_FIRST : 0) | (is_map ? F_IS_MAP : 0);
        ob.dom = cx.dom; ob.a = child_dom; ob.c = cx.list_depth; ob.buf0 = bv; ob.buf1 = bo; ob.node = id;
        int pc_begin = (int)cs.prog.size();
        push(ob);
        Op on = mk(OP_LIST_NEXT);
        on.dom = cx.dom; on.a = child_dom; on.c = cx.list_depth;
        on.buf2 = (int32_t)((is_map ? 1u : 0u) + min_bytes(*t.items));
        int pc_next = (int)cs.prog.size();
        push(on);

        Ctx cc;
        cc.dom = child_dom; cc.nullfill = false; cc.list_depth = cx.list_depth + 1;
        cc.nest = cx.nest + 1; cc.union_depth = cx.union_depth;
        if (is_map) cs.nodes[id].keys = string_leaf(NK_STRING, false, false, cc, nullptr);
        int child = build(*t.items, false, false, cc);
        cs.nodes[id].children = {child};

        Op ot = mk(OP_LIST_TAIL);
        ot.a = child_dom; ot.b = pc_next; ot.c = cx.list_depth;
        push(ot);
        int pc_end = (int)cs.prog.size();
        Op oe = mk(OP_LIST_END);
        oe.dom = cx.dom; oe.a = child_dom; oe.c = cx.list_depth; oe.buf1 = bo; oe.node = id;
        push(oe);
        cs.prog[pc_begin].b = pc_end;
        cs.prog[pc_next].b = pc_end;
        return id;
      }
      default:
        throw SchemaError("fast_decode: unsupported schema in make_decoder");
    }
  }
};

}  // namespace

std::unique_ptr<CompiledSchema> compile_schema(const char* text, size_t len) {
  auto cs = std::make_unique<CompiledSchema>();
  cs->json.assign(text, len);
  Value j;
  try {
    j = json::parse(text, len);
  } catch (const std::runtime_error& e) {
    throw SchemaError(e.what());
  }
  Parser p;
  cs->avro = p.parse(j, "");
  const AvroType& top = *cs->avro;
  if (top.kind != AV_RECORD)
    throw SchemaError("schema is outside the GPU direct-decode path: top-level schema must be a record "
                      "(fast_decode::is_supported)");
  std::string why;
  if (!supported_inner(top, why))
    throw SchemaError("schema is outside the GPU direct-decode path (fast_decode::is_supported is false): " + why);

  // to_arrow_schema, schema_translate.rs:19-37
  cs->arrow.format = "+s";
  cs->arrow.name = "";
  cs->arrow.nullable = false;
  for (auto& f : top.fields) {
    auto props = external_props(*f.type);
    cs->arrow.children.push_back(to_field(*f.type, &f.name, false, &props));
  }

  Builder b{*cs};
  Builder::Ctx cx{0, false, 0, 0, 0};
  b.build(top, false, false, cx);
  cs->prog.push_back(Builder::mk(OP_END));

  // counters: row domains first (domain d -> counter d-1), then the string byte columns.
  // Round 6 (the reference builds its decoder tree for any width, fast_decode.rs:342-370; K was capped
  // at 96 by per-lane counter storage).  A byte column of DOMAIN 0 is met exactly once per record, so its counter needs no
  // per-lane storage at all when the scan unit is one wavefront: the size pass sums the lengths of the wavefront on the spot,
  // the emit pass scans them on the spot on top of the wavefront's base (walk.h F_WAVE_CTR).  A schema with more than
  // kWideCounters counters is compiled WIDE: tiles of 64 records (one wavefront), the per-lane counters (KL: row domains and the
  // byte columns inside arrays / maps) numbered first, the wave counters behind them.
  const int ndomc = cs->ndom - 1;
  cs->K = ndomc + b.nstr;
  cs->wide = cs->K > kWideCounters;
  std::vector<int> newid((size_t)b.nstr, 0);       // local string index -> counter id
  {
    int nl = ndomc, nw = 0, inlist = 0;
    for (int pc : b.str_counter_ops) inlist += cs->prog[pc].dom != 0 ? 1 : 0;
    for (size_t i = 0; i < b.str_counter_ops.size(); i++) {
      const Op& o = cs->prog[b.str_counter_ops[i]];
      const int sidx = o.a;
      if (!cs->wide) newid[(size_t)sidx] = ndomc + sidx;
      else if (o.dom != 0) newid[(size_t)sidx] = nl++;
      else newid[(size_t)sidx] = ndomc + inlist + nw++;
    }
    cs->KL = cs->wide ? ndomc + inlist : cs->K;
  }
  for (int pc : b.str_counter_ops) {
    Op& o = cs->prog[pc];
    o.a = newid[(size_t)o.a];
    if (cs->wide && o.dom == 0) o.flags |= F_WAVE_CTR;
  }
  for (int id : b.data_bufs) cs->bufs[id].counter = newid[(size_t)cs->bufs[id].counter];
  for (auto& n : cs->nodes)
    if (n.counter >= 0) n.counter = newid[(size_t)n.counter];
  if (cs->sym_off.empty()) cs->sym_off.push_back(0);
  if (cs->sym_data.empty()) cs->sym_data.push_back(0);
  cs->min_record_bytes = Builder::min_bytes(top);
  return cs;
}

}  // namespace rh
