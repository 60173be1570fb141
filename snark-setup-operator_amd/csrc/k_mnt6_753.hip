// Kernel instantiations for one curve (G1 and G2) — see kernels_impl.hpp.
#include "kernels_impl.hpp"

namespace ptau {
using G1T = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg>;
using G2T = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg>;

const GroupOps *ops_mnt6_753(int group) {
    static const GroupOps o1 = Launch<G1T>::ops();
    static const GroupOps o2 = Launch<G2T>::ops();
    return group == 1 ? &o1 : (group == 2 ? &o2 : nullptr);
}
}  // namespace ptau
