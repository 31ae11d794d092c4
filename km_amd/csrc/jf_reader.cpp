// jf_reader.cpp — host reader for Jellyfish `binary/sorted` k-mer count files.
//
// Replaces what the reference obtains from the third-party binding in
// km/utils/Jellyfish.py:24-25 (QueryMerFile(filename), MerDNA.k()) and its own
// header scan in km/utils/Jellyfish.py:29-45 (`canonical`).
//
// Layout: 9 ASCII digits N, N bytes of JSON (padded), then
// fixed-size records [ceil(key_len/8)-byte LE key][counter_len-byte LE count].
// Records are read in file order; none of matrix1/reprobes is needed.
#include "jf_reader.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace jfio {

namespace {

// Minimal JSON walker: visits the top-level members of one object and hands
// scalar values (string / number / bool) to a callback; nested containers are
// skipped.  Enough for the Jellyfish header, which is machine-written.
struct Cursor {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() { while (p < end && (isspace((unsigned char)*p) || *p == '\0')) ++p; }
  bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
  std::string str() {
    std::string s;
    ws();
    if (p >= end || *p != '"') { ok = false; return s; }
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\' && p + 1 < end) { ++p; }
      s.push_back(*p++);
    }
    if (p >= end) { ok = false; return s; }
    ++p;
    return s;
  }
  void skip_value() {
    ws();
    if (p >= end) { ok = false; return; }
    if (*p == '"') { str(); return; }
    if (*p == '{' || *p == '[') {
      char open = *p, close = (open == '{') ? '}' : ']';
      int depth = 0;
      while (p < end) {
        if (*p == '"') { str(); continue; }
        if (*p == open) ++depth;
        else if (*p == close) { --depth; if (depth == 0) { ++p; return; } }
        ++p;
      }
      ok = false;
      return;
    }
    while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
  }
  std::string scalar() {   // string, number or literal as text
    ws();
    if (p < end && *p == '"') return str();
    const char* s = p;
    while (p < end && *p != ',' && *p != '}' && *p != ']' && !isspace((unsigned char)*p)) ++p;
    return std::string(s, p);
  }
};

}  // namespace

int read_layout(const char* path, Layout* lay, void** file, std::string* err) {
  *file = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) { *err = std::string("cannot open ") + path; return 1; }
  char digits[10] = {0};
  if (fread(digits, 1, 9, f) != 9) { fclose(f); *err = "file too short for a Jellyfish header"; return 2; }
  for (int i = 0; i < 9; ++i)
    if (!isdigit((unsigned char)digits[i])) { fclose(f); *err = "missing 9-digit header length"; return 2; }
  size_t hlen = (size_t)strtoull(digits, nullptr, 10);
  std::string json(hlen, '\0');
  if (hlen == 0 || fread(&json[0], 1, hlen, f) != hlen) { fclose(f); *err = "truncated header"; return 2; }

  Cursor c{json.data(), json.data() + json.size()};
  std::string format;
  long key_len = -1, counter_len = -1;
  int canonical = -1;
  if (!c.eat('{')) { fclose(f); *err = "header is not a JSON object"; return 2; }
  while (c.ok) {
    c.ws();
    if (c.eat('}')) break;
    std::string key = c.str();
    if (!c.ok || !c.eat(':')) { c.ok = false; break; }
    c.ws();
    if (c.p < c.end && (*c.p == '{' || *c.p == '[')) {
      c.skip_value();
    } else {
      std::string v = c.scalar();
      if (key == "format") format = v;
      else if (key == "key_len") key_len = strtol(v.c_str(), nullptr, 10);
      else if (key == "counter_len") counter_len = strtol(v.c_str(), nullptr, 10);
      else if (key == "canonical") canonical = (v == "true") ? 1 : 0;
    }
    if (!c.eat(',')) { c.ws(); if (c.eat('}')) break; if (c.p >= c.end) break; }
  }
  if (!c.ok) { fclose(f); *err = "malformed JSON header"; return 2; }
  if (format != "binary/sorted") {
    fclose(f);
    *err = "unsupported Jellyfish format '" + format + "' (need binary/sorted)";
    return 2;
  }
  if (key_len < 4 || (key_len & 1) || canonical < 0 || counter_len < 1) {
    fclose(f); *err = "header lacks key_len / counter_len / canonical"; return 2;
  }
  if (key_len > 64) { fclose(f); *err = "k > 32 is not supported"; return 3; }
  if (counter_len > 4) { fclose(f); *err = "counter_len > 4 is not supported"; return 2; }

  const size_t kb = (size_t)(key_len + 7) / 8, cb = (size_t)counter_len, rec = kb + cb;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); *err = "seek failed"; return 1; }
  long fsz = ftell(f);
  size_t body = (size_t)fsz - 9 - hlen;
  fseek(f, (long)(9 + hlen), SEEK_SET);
  lay->k = (int)(key_len / 2);
  lay->canonical = canonical;
  lay->key_bytes = (uint32_t)kb;
  lay->counter_bytes = (uint32_t)cb;
  lay->n_records = body / rec;
  lay->body_offset = 9 + hlen;
  *file = f;
  return 0;
}

int read_file(const char* path, Records* out, std::string* err) {
  Layout lay;
  void* file = nullptr;
  int rc = read_layout(path, &lay, &file, err);
  if (rc != 0) return rc;
  FILE* f = static_cast<FILE*>(file);
  const size_t kb = lay.key_bytes, cb = lay.counter_bytes, rec = kb + cb;
  const size_t n = (size_t)lay.n_records;
  out->k = lay.k;
  out->canonical = lay.canonical;
  out->keys.resize(n);
  out->counts.resize(n);
  const size_t CH = 1 << 16;
  std::vector<unsigned char> buf(CH * rec);
  size_t done = 0, kept = 0;
  uint64_t* keys = out->keys.data();
  uint32_t* counts = out->counts.data();
  while (done < n) {
    size_t m = (n - done < CH) ? n - done : CH;
    if (fread(buf.data(), rec, m, f) != m) { fclose(f); *err = "truncated record block"; return 1; }
    if (kb == 8 && cb == 4) {                 // k in 29..32 with 4-byte counters: the common layout
      for (size_t i = 0; i < m; ++i) {
        uint64_t key;
        uint32_t cnt;
        memcpy(&key, buf.data() + i * 12, 8);
        memcpy(&cnt, buf.data() + i * 12 + 8, 4);
        keys[kept] = key;
        counts[kept] = cnt;
        kept += cnt != 0;                     // query() returns 0 for absent k-mers anyway
      }
    } else {
      for (size_t i = 0; i < m; ++i) {
        const unsigned char* r = buf.data() + i * rec;
        uint64_t key = 0;
        for (size_t b = 0; b < kb; ++b) key |= (uint64_t)r[b] << (8 * b);
        uint32_t cnt = 0;
        for (size_t b = 0; b < cb; ++b) cnt |= (uint32_t)r[kb + b] << (8 * b);
        if (cnt == 0) continue;
        keys[kept] = key;
        counts[kept] = cnt;
        ++kept;
      }
    }
    done += m;
  }
  out->keys.resize(kept);
  out->counts.resize(kept);
  fclose(f);
  return 0;
}

}  // namespace jfio
