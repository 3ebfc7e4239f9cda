// Driver for the reference's own sort_indexes (als_CP.cxx:835-843), which is STL-only: the rule
// by which alsCP_PP_partupdate (`-pp 2`) orders the modes it updates. As with dimtree_ref_main.cpp
// the function body is NOT in this repository: `make ref` pipes that line range of
// /root/reference/als_CP.cxx between the two halves of this file into g++ and keeps only the
// binary (oracle/_ref/sortidx_ref). stdin: one vector per line; stdout: its index order.
#include <algorithm>
#include <cstdio>
#include <iostream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>
using namespace std;
//@@REFERENCE_FUNCTION_GOES_HERE@@
int main() {
  string line;
  while (getline(cin, line)) {
    istringstream in(line);
    vector<double> v;
    double x;
    while (in >> x) v.push_back(x);
    if (v.empty()) continue;
    vector<int> idx = sort_indexes(v);
    for (size_t i = 0; i < idx.size(); i++) printf("%d%c", idx[i], i + 1 == idx.size() ? '\n' : ' ');
  }
  return 0;
}
