"""Curve constants for the four ceremony curves (TEST INFRASTRUCTURE — oracle only).

PARITY UNPINNED: the reference (/root/reference) holds no golden vectors for this path
and its arithmetic lives in un-vendored crates (ark-bls12-377 / ark-bw6-761 /
ark-mnt4-753 / ark-mnt6-753 0.4.0, Cargo.lock:204-207,222-225,282-285,293-296).
Constants below are those of SURVEY.md App. B, re-verified numerically by
tests/test_oracle_constants.py (primality, on-curve, r*G = O).

Curve dispatch mirrors the reference's `curve_kind` strings
(/root/reference/src/bin/contribute.rs:1156-1211).
"""

from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class GroupParams:
    """Short-Weierstrass group y^2 = x^3 + a x + b over Fq^deg = Fq[u]/(u^deg - nonresidue)."""
    deg: int                     # extension degree of the base field (1, 2 or 3)
    nonresidue: int              # u^deg = nonresidue (unused for deg == 1)
    a: Tuple[int, ...]
    b: Tuple[int, ...]
    gen: Tuple[Tuple[int, ...], Tuple[int, ...]]   # affine generator (x, y)
    gen_is_canonical: bool = True   # False: synthetic order-r point (arkworks' generator not recoverable offline)


@dataclass(frozen=True)
class Curve:
    name: str            # reference `curve_kind` string
    curve_id: int        # id used by the C ABI (include/ptau.h)
    q: int               # base field modulus
    r: int               # scalar field modulus (group order)
    g1: GroupParams
    g2: GroupParams

    @property
    def fq_bytes(self) -> int:
        return (self.q.bit_length() + 7) // 8

    @property
    def fr_bytes(self) -> int:
        return (self.r.bit_length() + 7) // 8

    def group(self, which: int) -> GroupParams:
        return self.g1 if which == 1 else self.g2

    def comp_size(self, which: int) -> int:
        """Compressed size: x with 2 flag bits."""
        g = self.group(which)
        return g.deg * self.fq_bytes

    def uncomp_size(self, which: int) -> int:
        return 2 * self.comp_size(which)


# ---------------------------------------------------------------- BLS12-377
_BLS_Q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
_BLS_R = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001

BLS12_377 = Curve(
    name="bls12_377", curve_id=0, q=_BLS_Q, r=_BLS_R,
    g1=GroupParams(
        deg=1, nonresidue=0, a=(0,), b=(1,),
        gen=((0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef,),
             (0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6,))),
    g2=GroupParams(
        deg=2, nonresidue=_BLS_Q - 5, a=(0, 0),
        b=(0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906),
        gen=((0x018480be71c785fec89630a2a3841d01c565f071203e50317ea501f557db6b9b71889f52bb53540274e3e48f7c005196,
              0x00ea6040e700403170dc5a51b1b140d5532777ee6651cecbe7223ece0799c9de5cf89984bff76fe6b26bfefa6ea16afe),
             (0x00690d665d446f7bd960736bcbb2efb4de03ed7274b49a58e458c282f832d204f2cf88886d8c7c2ef094094409fd4ddf,
              0x00f8169fd28355189e549da3151a70aa61ef11ac3d591bf12463b01acee304c24279b83f5e52270bd9a1cdd185eb8f93))),
)

# ---------------------------------------------------------------- BW6-761
_BW6_Q = 0x0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b

BW6_761 = Curve(
    name="bw6", curve_id=1, q=_BW6_Q, r=_BLS_Q,
    g1=GroupParams(
        deg=1, nonresidue=0, a=(0,), b=(_BW6_Q - 1,),
        gen=((0x01075b020ea190c8b277ce98a477beaee6a0cfb7551b27f0ee05c54b85f56fc779017ffac15520ac11dbfcd294c2e746a17a54ce47729b905bd71fa0c9ea097103758f9a280ca27f6750dd0356133e82055928aca6af603f4088f3af66e5b43d,),
             (0x0058b84e0a6fc574e6fd637b45cc2a420f952589884c9ec61a7348d2a2e573a3265909f1af7e0dbac5b8fa1771b5b806cc685d31717a4c55be3fb90b6fc2cdd49f9df141b3053253b2b08119cad0fb93ad1cb2be0b20d2a1bafc8f2db4e95363,))),
    g2=GroupParams(
        deg=1, nonresidue=0, a=(0,), b=(4,),
        gen=((0x0110133241d9b816c852a82e69d660f9d61053aac5a7115f4c06201013890f6d26b41c5dab3da268734ec3f1f09feb58c5bbcae9ac70e7c7963317a300e1b6bace6948cb3cd208d700e96efbc2ad54b06410cf4fe1bf995ba830c194cd025f1c,),
             (0x0017c3357761369f8179eb10e4b6d2dc26b7cf9acec2181c81a78e2753ffe3160a1d86c80b95a59c94c97eb733293fef64f293dbd2c712b88906c170ffa823003ea96fcd504affc758aa2d3a3c5a02a591ec0594f9eac689eb70a16728c73b61,))),
)

# ---------------------------------------------------------------- MNT4/6-753 cycle
_MNT_A = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
_MNT_B = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888458477323173057491593855069696241854796396165721416325350064441470418137846398469611935719059908164220784476160001

_MNT4_B1 = 28798803903456388891410036793299405764940372360099938340752576406393880372126970068421383312482853541572780087363938442377933706865252053507077543420534380486492786626556269083255657125025963825610840222568694137138741554679540
_MNT6_B1 = 11625908999541321152027340224010374716841167701783584648338908235410859267060079819722747939267925389062611062156601938166010098747920378738927832658133625454260115409075816187555055859490253375704728027944315501122723426879114

# G2 generators of the MNT twists: arkworks' canonical constants could not be
# recovered offline.  The points below are SYNTHETIC order-r
# points derived deterministically by oracle/pyref/derive_mnt_g2.py
# (smallest x = (k, 1[,0]) on the twist, multiplied by the twist cofactor).
# Only `new_challenge` and the chunk-0 generator-equality check depend on them.
from ._mnt_g2_gen import MNT4_G2_GEN, MNT6_G2_GEN  # noqa: E402

MNT4_753 = Curve(
    name="mnt4_753", curve_id=2, q=_MNT_A, r=_MNT_B,
    g1=GroupParams(
        deg=1, nonresidue=0, a=(2,), b=(_MNT4_B1,),
        gen=((7790163481385331313124631546957228376128961350185262705123068027727518350362064426002432450801002268747950550964579198552865939244360469674540925037890082678099826733417900510086646711680891516503232107232083181010099241949569,),
             (6913648190367314284606685101150155872986263667483624713540251048208073654617802840433842931301128643140890502238233930290161632176167186761333725658542781350626799660920481723757654531036893265359076440986158843531053720994648,))),
    g2=GroupParams(
        deg=2, nonresidue=13, a=(2 * 13, 0), b=(0, _MNT4_B1 * 13 % _MNT_A),
        gen=MNT4_G2_GEN, gen_is_canonical=False),
)

MNT6_753 = Curve(
    name="mnt6_753", curve_id=3, q=_MNT_B, r=_MNT_A,
    g1=GroupParams(
        deg=1, nonresidue=0, a=(11,), b=(_MNT6_B1,),
        gen=((16364236387491689444759057944334173579070747473738339749093487337644739228935268157504218078126401066954815152892688541654726829424326599038522503517302466226143788988217410842672857564665527806044250003808514184274233938437290,),
             (4510127914410645922431074687553594593336087066778984214797709122300210966076979927285161950203037801392624582544098750667549188549761032654706830225743998064330900301346566408501390638273322467173741629353517809979540986561128,))),
    g2=GroupParams(
        deg=3, nonresidue=11, a=(0, 0, 11), b=(_MNT6_B1 * 11 % _MNT_B, 0, 0),
        gen=MNT6_G2_GEN, gen_is_canonical=False),
)

CURVES = {c.name: c for c in (BLS12_377, BW6_761, MNT4_753, MNT6_753)}
CURVES_BY_ID = {c.curve_id: c for c in CURVES.values()}
