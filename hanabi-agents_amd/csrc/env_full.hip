// Explicit instantiations of the fused env kernel for the 'full' game family
// (colors=5 ranks=5 max_info=8 max_life=3), players 2..5.
#include "env_kernel.hpp"

namespace hb {
static const EnvVariant k_full[] = {
    make_variant<Cfg<2, 5, 5, 5, 8, 3>>(),
    make_variant<Cfg<3, 5, 5, 5, 8, 3>>(),
    make_variant<Cfg<4, 5, 5, 4, 8, 3>>(),
    make_variant<Cfg<5, 5, 5, 4, 8, 3>>(),
};
const EnvVariant* variants_full(int* n) {
  *n = 4;
  return k_full;
}
}  // namespace hb
