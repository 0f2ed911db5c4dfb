// Explicit instantiations of the fused env kernel for the 'small' game family
// (colors=2 ranks=5 max_info=3 max_life=1), players 2..5.
#include "env_kernel.hpp"

namespace hb {
static const EnvVariant k_small[] = {
    make_variant<Cfg<2, 2, 5, 2, 3, 1>>(),
    make_variant<Cfg<3, 2, 5, 2, 3, 1>>(),
    make_variant<Cfg<4, 2, 5, 2, 3, 1>>(),
    make_variant<Cfg<5, 2, 5, 2, 3, 1>>(),
};
const EnvVariant* variants_small(int* n) {
  *n = 4;
  return k_small;
}
}  // namespace hb
