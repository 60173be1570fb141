// Kernel instantiations for one curve (G1 and G2) — see kernels_impl.hpp.
#include "kernels_impl.hpp"

namespace ptau {
using G1T = GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G1Cfg>;
using G2T = GroupTypes<Mnt4FqR29, Mnt4Fr, Mnt4G2Cfg>;
// (a lane-pair twin of G2T measured SLOWER here — 138 k vs 217 k updates/s: the 24-limb base products go by reference
// through memory, so halving the register footprint buys nothing and schoolbook costs 4/3 of Karatsuba)

const GroupOps *ops_mnt4_753(int group) {
    static const GroupOps o1 = Launch<G1T>::ops();
    static const GroupOps o2 = Launch<G2T>::ops();
    return group == 1 ? &o1 : (group == 2 ? &o2 : nullptr);
}
}  // namespace ptau
