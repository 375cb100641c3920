// parquet_model.cpp — <output>/bin/data as Spark writes it: Word2VecModel.save (Spark 2.2 mllib; called by Main.saveModelAndFeatures,
// M/Main.scala:36-44 `model.save(context, s"${config.output}.${Property.modelSuffix}")`) stores the model as a DataFrame of
// `case class Data(word: String, vector: Array[Float])` in Parquet, i.e. the schema
//     message spark_schema { optional binary word (UTF8); optional group vector (LIST) { repeated group list { required float element; } } }
// next to a one-line JSON metadata file.  This file writes exactly that with no library: Parquet format 1.0, uncompressed, PLAIN values,
// RLE levels, data page v1, Thrift compact protocol for the page headers and the footer, Spark's row-metadata key in the footer so that
// `Word2VecModel.load` / `spark.read.parquet` see the types Spark itself would have written.  tests/test_host_cpu.py reads the file back
// with pyarrow (schema, words, vectors bit for bit).
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "engine.h"

namespace srw {
namespace {
// ---- Thrift compact protocol (only what the Parquet structs below need) ---------------------------------------------------------
struct TOut {
  std::string b;
  std::vector<int16_t> last;                     // last field id per open struct
  void byte(uint8_t x) { b.push_back((char)x); }
  void varint(uint64_t v) { while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; } byte((uint8_t)v); }
  static uint64_t zz(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
  void begin() { last.push_back(0); }
  void end() { byte(0); last.pop_back(); }
  void field(int16_t id, uint8_t type) {
    const int delta = id - last.back();
    if (delta > 0 && delta <= 15) byte((uint8_t)((delta << 4) | type));
    else { byte(type); varint(zz(id)); }
    last.back() = id;
  }
  void i32(int16_t id, int32_t v) { field(id, 5); varint(zz(v)); }
  void i64(int16_t id, int64_t v) { field(id, 6); varint(zz(v)); }
  void str(int16_t id, const std::string &s) { field(id, 8); varint(s.size()); b += s; }
  void list(int16_t id, uint8_t elem_type, size_t n) {
    field(id, 9);
    if (n < 15) byte((uint8_t)((n << 4) | elem_type)); else { byte((uint8_t)(0xF0 | elem_type)); varint(n); }
  }
  void strukt(int16_t id) { field(id, 12); begin(); }
  // list elements
  void elem_i32(int32_t v) { varint(zz(v)); }
  void elem_str(const std::string &s) { varint(s.size()); b += s; }
};
enum { T_FLOAT = 4, T_BYTE_ARRAY = 6 };            // parquet Type
enum { REP_REQUIRED = 0, REP_OPTIONAL = 1, REP_REPEATED = 2 };
enum { CONV_UTF8 = 0, CONV_LIST = 3 };
enum { ENC_PLAIN = 0, ENC_RLE = 3 };

// RLE / bit-packed hybrid, RLE runs only: header varint (run << 1), then the value in ceil(bit_width / 8) bytes (here: 1)
void rle_run(std::string &o, uint64_t run, uint8_t value) {
  while (run > 0) {
    const uint64_t r = run > 0x3FFFFFFFull ? 0x3FFFFFFFull : run;
    uint64_t v = r << 1;
    while (v >= 0x80) { o.push_back((char)(uint8_t)(v | 0x80)); v >>= 7; }
    o.push_back((char)(uint8_t)v);
    o.push_back((char)value);
    run -= r;
  }
}
void put_u32(std::string &o, uint32_t v) { o.append(reinterpret_cast<const char *>(&v), 4); }     // little endian hosts only (x86-64)

std::string page_header(int32_t n_values, int32_t size) {
  TOut t; t.begin();
  t.i32(1, 0 /* DATA_PAGE */); t.i32(2, size); t.i32(3, size);
  t.strukt(5); t.i32(1, n_values); t.i32(2, ENC_PLAIN); t.i32(3, ENC_RLE); t.i32(4, ENC_RLE); t.end();
  t.end();
  return t.b;
}
struct Chunk { int64_t offset = 0, size = 0, n_values = 0; };
struct RowGroupRec { Chunk word, vec; int64_t rows = 0; };
}  // namespace

// rows [0, n): word(r) appended by put_name, vector = vectors[r * dim .. + dim)
void write_word2vec_parquet(const std::string &path, const std::function<void(std::string &, int64_t)> &put_name, const float *vectors,
                            int64_t n, int32_t dim) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw Error(SRW_ERR_IO, "cannot open " + path + ": " + strerror(errno));
  int64_t pos = 0;
  auto put = [&](const std::string &s) {
    if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) { fclose(f); throw Error(SRW_ERR_IO, "write error on " + path); }
    pos += (int64_t)s.size();
  };
  put("PAR1");
  // geometry: pages of <= ~1 MiB of vector values, row groups of <= ~64 MiB
  const int64_t rows_per_page = std::max<int64_t>(1, ((int64_t)1 << 20) / std::max<int64_t>((int64_t)dim * 4, 1));
  int64_t group_bytes = (int64_t)64 << 20;
  if (const char *e = getenv("SRW_PARQUET_GROUP_MB"); e && atoi(e) >= 1) group_bytes = (int64_t)atoi(e) << 20;       // (tests: several row groups in a small model)
  const int64_t rows_per_group = std::max<int64_t>(rows_per_page, group_bytes / std::max<int64_t>((int64_t)dim * 4, 1) / rows_per_page * rows_per_page);
  std::vector<RowGroupRec> groups;
  std::string page, name;
  for (int64_t g0 = 0; g0 < n || (n == 0 && groups.empty()); g0 += rows_per_group) {
    RowGroupRec rg; rg.rows = std::min<int64_t>(rows_per_group, n - g0);
    if (n == 0) { groups.push_back(rg); break; }
    // column chunk 1: word
    rg.word.offset = pos;
    for (int64_t r0 = g0; r0 < g0 + rg.rows; r0 += rows_per_page) {
      const int64_t nr = std::min<int64_t>(rows_per_page, g0 + rg.rows - r0);
      page.clear();
      std::string lv; rle_run(lv, (uint64_t)nr, 1);                     // definition levels: every word is present (max level 1)
      put_u32(page, (uint32_t)lv.size()); page += lv;
      for (int64_t r = r0; r < r0 + nr; ++r) { name.clear(); put_name(name, r); put_u32(page, (uint32_t)name.size()); page += name; }
      const std::string h = page_header((int32_t)nr, (int32_t)page.size());
      put(h); put(page);
      rg.word.n_values += nr;
    }
    rg.word.size = pos - rg.word.offset;
    // column chunk 2: vector.list.element
    rg.vec.offset = pos;
    for (int64_t r0 = g0; r0 < g0 + rg.rows; r0 += rows_per_page) {
      const int64_t nr = std::min<int64_t>(rows_per_page, g0 + rg.rows - r0);
      page.clear();
      std::string rep, def;
      if (dim > 0) {
        for (int64_t r = 0; r < nr; ++r) { rle_run(rep, 1, 0); if (dim > 1) rle_run(rep, (uint64_t)(dim - 1), 1); }   // a row starts at level 0, its other elements repeat at 1
        rle_run(def, (uint64_t)nr * (uint64_t)dim, 2);                  // vector present, list entry present, element required: level 2
      } else { rle_run(rep, (uint64_t)nr, 0); rle_run(def, (uint64_t)nr, 1); }    // empty lists: defined up to the list group
      put_u32(page, (uint32_t)rep.size()); page += rep;
      put_u32(page, (uint32_t)def.size()); page += def;
      if (dim > 0) page.append(reinterpret_cast<const char *>(vectors + r0 * dim), (size_t)nr * (size_t)dim * 4);
      const int64_t nv = dim > 0 ? nr * dim : nr;
      const std::string h = page_header((int32_t)nv, (int32_t)page.size());
      put(h); put(page);
      rg.vec.n_values += nv;
    }
    rg.vec.size = pos - rg.vec.offset;
    groups.push_back(rg);
  }
  // footer
  TOut t; t.begin();
  t.i32(1, 1);                                                          // version
  t.list(2, 12, 5);                                                     // schema: root, word, vector, list, element
  auto schema = [&](int type, int rep, const char *nm, int children, int conv) {
    t.begin();
    if (type >= 0) t.i32(1, type);
    if (rep >= 0) t.i32(3, rep);
    t.str(4, nm);
    if (children > 0) t.i32(5, children);
    if (conv >= 0) t.i32(6, conv);
    t.end();
  };
  schema(-1, -1, "spark_schema", 2, -1);
  schema(T_BYTE_ARRAY, REP_OPTIONAL, "word", 0, CONV_UTF8);
  schema(-1, REP_OPTIONAL, "vector", 1, CONV_LIST);
  schema(-1, REP_REPEATED, "list", 1, -1);
  schema(T_FLOAT, REP_REQUIRED, "element", 0, -1);
  t.i64(3, n);
  t.list(4, 12, groups.size());
  for (const RowGroupRec &rg : groups) {
    t.begin();
    t.list(1, 12, 2);
    auto column = [&](const Chunk &c, int type, std::initializer_list<const char *> path_in_schema) {
      t.begin();
      t.i64(2, c.offset);
      t.strukt(3);
      t.i32(1, type);
      t.list(2, 5, 2); t.elem_i32(ENC_PLAIN); t.elem_i32(ENC_RLE);
      t.list(3, 8, path_in_schema.size()); for (const char *s : path_in_schema) t.elem_str(s);
      t.i32(4, 0 /* UNCOMPRESSED */);
      t.i64(5, c.n_values); t.i64(6, c.size); t.i64(7, c.size);
      t.i64(9, c.offset);
      t.end();
      t.end();
    };
    column(rg.word, T_BYTE_ARRAY, {"word"});
    column(rg.vec, T_FLOAT, {"vector", "list", "element"});
    t.i64(2, rg.word.size + rg.vec.size);
    t.i64(3, rg.rows);
    t.end();
  }
  t.list(5, 12, 1);
  t.begin();
  t.str(1, "org.apache.spark.sql.parquet.row.metadata");
  t.str(2, "{\"type\":\"struct\",\"fields\":[{\"name\":\"word\",\"type\":\"string\",\"nullable\":true,\"metadata\":{}},"
           "{\"name\":\"vector\",\"type\":{\"type\":\"array\",\"elementType\":\"float\",\"containsNull\":false},\"nullable\":true,\"metadata\":{}}]}");
  t.end();
  t.str(6, "stellar-rw (Word2VecModel.save layout, hand-written Parquet 1.0 writer)");
  t.end();
  put(t.b);
  std::string tail; put_u32(tail, (uint32_t)t.b.size()); tail += "PAR1";
  put(tail);
  if (fclose(f) != 0) throw Error(SRW_ERR_IO, "write error on " + path);
}
}  // namespace srw
