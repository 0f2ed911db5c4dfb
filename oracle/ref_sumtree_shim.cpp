// extern "C" shim around the REFERENCE SumTree<float> (TEST INFRASTRUCTURE ONLY).
// The reference header is not copied: it is included from where it lies
// (/root/reference/sum_tree/sum_tree/include/sum_tree.h) via -I in oracle/Makefile,
// and the result goes to oracle/_ref/libsumtree_ref.so (git-ignored, travels with gpurun).
// Used to pin oracle/hanabi_oracle.c's flat tree and as cpu_baseline "reference" timing.
#include <cstdint>
#include <vector>
#include "sum_tree.h"

extern "C" {
void* ref_tree_create(int64_t capacity) { return new SumTree<float>(static_cast<size_t>(capacity)); }
void ref_tree_destroy(void* t) { delete static_cast<SumTree<float>*>(t); }
int64_t ref_tree_capacity(void* t) { return static_cast<int64_t>(static_cast<SumTree<float>*>(t)->getCapacity()); }
float ref_tree_total(void* t) { return static_cast<SumTree<float>*>(t)->getTotalVal(); }
void ref_tree_update_value(void* t, int64_t idx, float v) {
  static_cast<SumTree<float>*>(t)->updateValue(static_cast<int>(idx), v);
}
// update_values as the pybind binding sees it: the arrays are copied into std::vector first
void ref_tree_update_values(void* t, const int64_t* idx, const float* val, int64_t n) {
  std::vector<size_t> i(idx, idx + n);
  std::vector<float> v(val, val + n);
  static_cast<SumTree<float>*>(t)->updateValues(i, v);
}
int64_t ref_tree_get_index(void* t, float q) {
  return static_cast<int64_t>(static_cast<SumTree<float>*>(t)->getIndex(q));
}
void ref_tree_get_indices(void* t, const float* q, int64_t* out, int64_t n) {
  std::vector<float> qs(q, q + n);
  std::vector<size_t> r = static_cast<SumTree<float>*>(t)->getIndices(qs);
  for (int64_t k = 0; k < n; ++k) out[k] = static_cast<int64_t>(r[k]);
}
void ref_tree_get_values(void* t, const int64_t* idx, float* out, int64_t n) {
  std::vector<size_t> i(idx, idx + n);
  std::vector<float> r = static_cast<SumTree<float>*>(t)->getValues(i);
  for (int64_t k = 0; k < n; ++k) out[k] = r[k];
}
}
