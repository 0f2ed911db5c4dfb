// Explicit instantiations of the fused env kernel for the 'vsmall' game family
// (colors=1 ranks=5 max_info=3 max_life=1), players 2..5.
#include "env_kernel.hpp"

namespace hb {
static const EnvVariant k_vsmall[] = {
    make_variant<Cfg<2, 1, 5, 2, 3, 1>>(),
    make_variant<Cfg<3, 1, 5, 2, 3, 1>>(),
    make_variant<Cfg<4, 1, 5, 2, 3, 1>>(),
    make_variant<Cfg<5, 1, 5, 2, 3, 1>>(),
};
const EnvVariant* variants_vsmall(int* n) {
  *n = 4;
  return k_vsmall;
}
}  // namespace hb
