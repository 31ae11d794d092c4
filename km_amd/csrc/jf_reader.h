// jf_reader.h — host reader for Jellyfish `binary/sorted` files (see jf_reader.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace jfio {

struct Records {
  int k = 0;
  int canonical = 0;
  std::vector<uint64_t> keys;
  std::vector<uint32_t> counts;
};

// 0 = ok, 1 = I/O error, 2 = format error, 3 = unsupported k.
int read_file(const char* path, Records* out, std::string* err);

// The fixed-size record area of a file: [key_bytes little-endian key][counter_bytes count] * n.
struct Layout {
  int k = 0;
  int canonical = 0;
  uint32_t key_bytes = 0, counter_bytes = 0;
  uint64_t n_records = 0;
  uint64_t body_offset = 0;   // file offset of the first record
};
// Parses only the header; `*file` is left open (caller closes) and positioned at the body.
int read_layout(const char* path, Layout* out, void** file, std::string* err);

}  // namespace jfio
