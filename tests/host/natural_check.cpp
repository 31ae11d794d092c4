// cmp_nat (csrc/report.cpp) on pairs of strings read from stdin (one string per line, two lines per pair):
// prints -1 / 0 / 1 per pair.  tests/test_report_native.py compares with Python's natural-sort key.
#include "../../km_amd/csrc/report.cpp"
#include <iostream>
int main() {
  std::string a, b;
  while (std::getline(std::cin, a) && std::getline(std::cin, b))
    printf("%d\n", cmp_nat(a.data(), a.size(), b.data(), b.size()));
  return 0;
}
