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

}  // namespace jfio
