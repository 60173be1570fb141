// Kernel instantiations for one curve (G1 and G2) — see kernels_impl.hpp.
#include "kernels_impl.hpp"
#include "pairing_dev.hpp"

namespace ptau {
using G1T = GroupTypes<BlsFqR29, BlsFr, BlsG1Cfg>;
using G2T = GroupTypes<BlsFqR29, BlsFr, BlsG2Cfg>;
using G2L = GroupTypes<BlsFqR29, BlsFr, BlsG2Cfg, LANES_FQ2_PAIR>;   // Fq2 over lane pairs (tau-update of uncompressed points)

const GroupOps *ops_bls12_377(int group) {
    static const GroupOps o1 = Launch<G1T>::ops();
    static const GroupOps o2 = Launch<G2T, G2L>::ops();
    return group == 1 ? &o1 : (group == 2 ? &o2 : nullptr);
}

const PairingOps *pairing_ops_bls12_377() {
    static const PairingOps o = PairingLaunch<TatePairingDev<BlsFqR29, BlsFr, BlsG1Cfg, BlsG2Cfg, BlsPairingCfg>>::ops();
    return &o;
}
}  // namespace ptau
