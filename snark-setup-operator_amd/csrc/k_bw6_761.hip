// Kernel instantiations for one curve (G1 and G2) — see kernels_impl.hpp.
#include "kernels_impl.hpp"
#include "pairing_dev.hpp"

namespace ptau {
// BW6-761's base field is held as 28 limbs of 29 bits in every kernel (gen_constants.py RADIX29_LIMBS)
using G1T = GroupTypes<Bw6FqR29, Bw6Fr, Bw6G1Cfg>;
using G2T = GroupTypes<Bw6FqR29, Bw6Fr, Bw6G2Cfg>;
// tau-update twins: every base-field element over a lane pair, 14 limbs per lane (fpl.hpp)
using G1S = GroupTypes<Bw6FqR29, Bw6Fr, Bw6G1Cfg, LANES_LIMB_SPLIT>;
using G2S = GroupTypes<Bw6FqR29, Bw6Fr, Bw6G2Cfg, LANES_LIMB_SPLIT>;

const GroupOps *ops_bw6_761(int group) {
    static const GroupOps o1 = Launch<G1T, G1S>::ops();
    static const GroupOps o2 = Launch<G2T, G2S>::ops();
    return group == 1 ? &o1 : (group == 2 ? &o2 : nullptr);
}

const PairingOps *pairing_ops_bw6_761() {
    static const PairingOps o = PairingLaunch<TatePairingDev<Bw6FqR29, Bw6Fr, Bw6G1Cfg, Bw6G2Cfg, Bw6PairingCfg>>::ops();
    return &o;
}
}  // namespace ptau
