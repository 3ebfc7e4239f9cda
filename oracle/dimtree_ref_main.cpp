// Driver for the reference's own Construct_Dimension_Tree (common.cxx:225-270), which is STL-only.
// The function body is NOT in this repository: `make ref` pipes that line range of
// /root/reference/common.cxx between the two halves of this file straight into g++ and keeps only
// the binary (oracle/_ref/dimtree_ref). This file is split at the marker line below.
#include <cstdio>
#include <map>
#include <string>
using namespace std;
//@@REFERENCE_FUNCTION_GOES_HERE@@
int main() {
  // same record format as ppo_dimension_tree: "key:parent:sibling;" per node, one line per N
  for (int N = 2; N <= 8; N++) {
    map<string, string> parent, sibling;
    Construct_Dimension_Tree(parent, sibling, 0, N - 1);
    printf("%d ", N);
    for (auto &kv : parent) printf("%s:%s:%s;", kv.first.c_str(), kv.second.c_str(), sibling[kv.first].c_str());
    printf("\n");
  }
  return 0;
}
