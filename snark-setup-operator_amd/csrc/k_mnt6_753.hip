// Kernel instantiations for one curve (G1 and G2) — see kernels_impl.hpp.
#include "kernels_impl.hpp"
#include "pairing_dev.hpp"

namespace ptau {
using G1T = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg>;
using G2T = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg>;
// tau-update twins: every 26-limb base-field element over a lane pair (fpl.hpp)
using G1S = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg, LANES_LIMB_SPLIT>;
using G2S = GroupTypes<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg, LANES_LIMB_SPLIT>;

// K1 in the carry-free field, one lane pair per tower coefficient (k1_lazy_a.hpp): 2 lanes per G1 point, 6 per G2 point
using G1A = LazyAGroup<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg>;
using G2A = LazyAGroup<Mnt6FqR29, Mnt6Fr, Mnt6G2Cfg>;

const GroupOps *ops_mnt6_753(int group) {
    static const GroupOps o1 = Launch<G1T, G1S, G1A>::ops();
    static const GroupOps o2 = Launch<G2T, G2S, G2A>::ops();
    return group == 1 ? &o1 : (group == 2 ? &o2 : nullptr);
}

const PairingOps *pairing_ops_mnt6_753() {
    static const PairingOps o = PairingLaunch<TatePairingDev<Mnt6FqR29, Mnt6Fr, Mnt6G1Cfg, Mnt6G2Cfg, Mnt6PairingCfg>>::ops();
    return &o;
}
}  // namespace ptau
