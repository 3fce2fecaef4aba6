#!/usr/bin/env python3
"""Derive every device constant of the BLS12-381 backend from the PUBLIC integers
(p, r, the curve parameter x, the standard generators, and for hash-to-G1 the RFC 9380 constants of the
11-isogenous curve) and write crypto12381_amd/csrc/consts.hpp.  Everything else (Montgomery constants, cube
roots, Frobenius / psi constants, exponents, the SSWU square-root multiplier) is computed here; the isogeny
coefficients are standard published integers and are verified below to define a map onto y^2 = x^3 + 4.
The reference's ROM tables (rom_field_BLS12381.cpp / rom_curve_BLS12381.cpp) hold the same integers in 58-bit
limbs; they are cited in the generated header so a reader can cross-check values.

Number format of the device code: 14 signed limbs of 28 bits, Montgomery radix R = 2^392.
"""
import os

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R_ORD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X = 0xd201000000010000          # |x|, x is negative
NL, LB = 14, 28
RM = 1 << (NL * LB)

G1X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2XA = 0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8
G2XB = 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e
G2YA = 0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801
G2YB = 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be

# ---- hash-to-curve constants for G1 (RFC 9380 section 8.8.1 / appendix E.2: simplified SWU onto the 11-isogenous
# curve E': y^2 = x^3 + A'x + B', Z = 11, then the 11-isogeny E' -> E).  Public standard integers, ascending degree
# (k_(j,0) first); the reference holds the same integers as CURVE_Ad / CURVE_Bd / PC[53] (rom_curve_BLS12381.cpp:96-98,
# descending degree).  check_iso11() below verifies that they do define a map onto y^2 = x^3 + 4.
SSWU_A = 0x144698a3b8e9433d693a02c96d4982b0ea985383ee66a8d8e8981aefd881ac98936f8da0e0f97f5cf428082d584c1d
SSWU_B = 0x12e2908d11688030018b12e8753eee3b2016c1f0f24f4070a0b9c14fcef35ef55a23215a316ceaa5d1cc48e98e172be0
SSWU_Z = 11
ISO11_XNUM = [
    0x11a05f2b1e833340b809101dd99815856b303e88a2d7005ff2627b56cdb4e2c85610c2d5f2e62d6eaeac1662734649b7,
    0x17294ed3e943ab2f0588bab22147a81c7c17e75b2f6a8417f565e33c70d1e86b4838f2a6f318c356e834eef1b3cb83bb,
    0x0d54005db97678ec1d1048c5d10a9a1bce032473295983e56878e501ec68e25c958c3e3d2a09729fe0179f9dac9edcb0,
    0x1778e7166fcc6db74e0609d307e55412d7f5e4656a8dbf25f1b33289f1b330835336e25ce3107193c5b388641d9b6861,
    0x0e99726a3199f4436642b4b3e4118e5499db995a1257fb3f086eeb65982fac18985a286f301e77c451154ce9ac8895d9,
    0x1630c3250d7313ff01d1201bf7a74ab5db3cb17dd952799b9ed3ab9097e68f90a0870d2dcae73d19cd13c1c66f652983,
    0x0d6ed6553fe44d296a3726c38ae652bfb11586264f0f8ce19008e218f9c86b2a8da25128c1052ecaddd7f225a139ed84,
    0x17b81e7701abdbe2e8743884d1117e53356de5ab275b4db1a682c62ef0f2753339b7c8f8c8f475af9ccb5618e3f0c88e,
    0x080d3cf1f9a78fc47b90b33563be990dc43b756ce79f5574a2c596c928c5d1de4fa295f296b74e956d71986a8497e317,
    0x169b1f8e1bcfa7c42e0c37515d138f22dd2ecb803a0c5c99676314baf4bb1b7fa3190b2edc0327797f241067be390c9e,
    0x10321da079ce07e272d8ec09d2565b0dfa7dccdde6787f96d50af36003b14866f69b771f8c285decca67df3f1605fb7b,
    0x06e08c248e260e70bd1e962381edee3d31d79d7e22c837bc23c0bf1bc24c6b68c24b1b80b64d391fa9c8ba2e8ba2d229,
]
ISO11_XDEN = [
    0x08ca8d548cff19ae18b2e62f4bd3fa6f01d5ef4ba35b48ba9c9588617fc8ac62b558d681be343df8993cf9fa40d21b1c,
    0x12561a5deb559c4348b4711298e536367041e8ca0cf0800c0126c2588c48bf5713daa8846cb026e9e5c8276ec82b3bff,
    0x0b2962fe57a3225e8137e629bff2991f6f89416f5a718cd1fca64e00b11aceacd6a3d0967c94fedcfcc239ba5cb83e19,
    0x03425581a58ae2fec83aafef7c40eb545b08243f16b1655154cca8abc28d6fd04976d5243eecf5c4130de8938dc62cd8,
    0x13a8e162022914a80a6f1d5f43e7a07dffdfc759a12062bb8d6b44e833b306da9bd29ba81f35781d539d395b3532a21e,
    0x0e7355f8e4e667b955390f7f0506c6e9395735e9ce9cad4d0a43bcef24b8982f7400d24bc4228f11c02df9a29f6304a5,
    0x0772caacf16936190f3e0c63e0596721570f5799af53a1894e2e073062aede9cea73b3538f0de06cec2574496ee84a3a,
    0x14a7ac2a9d64a8b230b3f5b074cf01996e7f63c21bca68a81996e1cdf9822c580fa5b9489d11e2d311f7d99bbdcc5a5e,
    0x0a10ecf6ada54f825e920b3dafc7a3cce07f8d1d7161366b74100da67f39883503826692abba43704776ec3a79a1d641,
    0x095fc13ab9e92ad4476d6e3eb3a56680f682b4ee96f7d03776df533978f31c1593174e4b4b7865002d6384d168ecdd0a,
]
ISO11_YNUM = [
    0x090d97c81ba24ee0259d1f094980dcfa11ad138e48a869522b52af6c956543d3cd0c7aee9b3ba3c2be9845719707bb33,
    0x134996a104ee5811d51036d776fb46831223e96c254f383d0f906343eb67ad34d6c56711962fa8bfe097e75a2e41c696,
    0x00cc786baa966e66f4a384c86a3b49942552e2d658a31ce2c344be4b91400da7d26d521628b00523b8dfe240c72de1f6,
    0x01f86376e8981c217898751ad8746757d42aa7b90eeb791c09e4a3ec03251cf9de405aba9ec61deca6355c77b0e5f4cb,
    0x08cc03fdefe0ff135caf4fe2a21529c4195536fbe3ce50b879833fd221351adc2ee7f8dc099040a841b6daecf2e8fedb,
    0x16603fca40634b6a2211e11db8f0a6a074a7d0d4afadb7bd76505c3d3ad5544e203f6326c95a807299b23ab13633a5f0,
    0x04ab0b9bcfac1bbcb2c977d027796b3ce75bb8ca2be184cb5231413c4d634f3747a87ac2460f415ec961f8855fe9d6f2,
    0x0987c8d5333ab86fde9926bd2ca6c674170a05bfe3bdd81ffd038da6c26c842642f64550fedfe935a15e4ca31870fb29,
    0x09fc4018bd96684be88c9e221e4da1bb8f3abd16679dc26c1e8b6e6a1f20cabe69d65201c78607a360370e577bdba587,
    0x0e1bba7a1186bdb5223abde7ada14a23c42a0ca7915af6fe06985e7ed1e4d43b9b3f7055dd4eba6f2bafaaebca731c30,
    0x19713e47937cd1be0dfd0b8f1d43fb93cd2fcbcb6caf493fd1183e416389e61031bf3a5cce3fbafce813711ad011c132,
    0x18b46a908f36f6deb918c143fed2edcc523559b8aaf0c2462e6bfe7f911f643249d9cdf41b44d606ce07c8a4d0074d8e,
    0x0b182cac101b9399d155096004f53f447aa7b12a3426b08ec02710e807b4633f06c851c1919211f20d4c04f00b971ef8,
    0x0245a394ad1eca9b72fc00ae7be315dc757b3b080d4c158013e6632d3c40659cc6cf90ad1c232a6442d9d3f5db980133,
    0x05c129645e44cf1102a159f748c4a3fc5e673d81d7e86568d9ab0f5d396a7ce46ba1049b6579afb7866b1e715475224b,
    0x15e6be4e990f03ce4ea50b3b42df2eb5cb181d8f84965a3957add4fa95af01b2b665027efec01c7704b456be69c8b604,
]
ISO11_YDEN = [
    0x16112c4c3a9c98b252181140fad0eae9601a6de578980be6eec3232b5be72e7a07f3688ef60c206d01479253b03663c1,
    0x1962d75c2381201e1a0cbd6c43c348b885c84ff731c4d59ca4a10356f453e01f78a4260763529e3532f6102c2e49a03d,
    0x058df3306640da276faaae7d6e8eb15778c4855551ae7f310c35a5dd279cd2eca6757cd636f96f891e2538b53dbf67f2,
    0x16b7d288798e5395f20d23bf89edb4d1d115c5dbddbcd30e123da489e726af41727364f2c28297ada8d26d98445f5416,
    0x0be0e079545f43e4b00cc912f8228ddcc6d19c9f0f69bbb0542eda0fc9dec916a20b15dc0fd2ededda39142311a5001d,
    0x08d9e5297186db2d9fb266eaac783182b70152c65550d881c5ecd87b6f0f5a6449f38db9dfa9cce202c6477faaf9b7ac,
    0x166007c08a99db2fc3ba8734ace9824b5eecfdfa8d0cf8ef5dd365bc400a0051d5fa9c01a58b1fb93d1a1399126a775c,
    0x16a3ef08be3ea7ea03bcddfabba6ff6ee5a4375efa1f4fd7feb34fd206357132b920f5b00801dee460ee415a15812ed9,
    0x1866c8ed336c61231a1be54fd1d74cc4f9fb0ce4c6af5920abc5750c4bf39b4852cfe2f7bb9248836b233d9d55535d4a,
    0x167a55cda70a6e1cea820597d94a84903216f763e13d87bb5308592e7ea7d4fbc7385ea3d529b35e346ef48bb8913f55,
    0x04d2f259eea405bd48f010a01ad2911d9c6dd039bb61a6290e591b36e636a5c871a5c29f4f83060400f8b49cba8f6aa8,
    0x0accbb67481d033ff5852c1e48c50c477f94ff8aefce42d28c0f9a88cea7913516f968986f7ebbea9684b529e2561092,
    0x0ad6b9514c767fe3c3613144b45f1496543346d98adf02267d5ceef9a00d9b8693000763e3b90ac11e99b138573345cc,
    0x02660400eb2e4f3b628bdd0d53cd76f2bf565b94e72927c1cb748df27942480e420517bd8714cc80d1fadc1326ed06f7,
    0x0e0fa1d816ddc03e6b24255e0d7819c171c40f65e273b853324efcd6356caa205ca2f570f13497804415473a1d634b8f,
]


def check_iso11():
    """the isogeny constants map points of E' to points of E (and A', B', Z satisfy the SSWU preconditions)"""
    def ev(cs, x, monic):
        acc = 1 if monic else 0
        for c in reversed(cs):
            acc = (acc * x + c) % P
        return acc
    assert pow(SSWU_Z, (P - 1) // 2, P) == P - 1                      # Z is a non-residue
    x = 5
    done = 0
    while done < 4:
        x += 1
        g = (x * x * x + SSWU_A * x + SSWU_B) % P
        if pow(g, (P - 1) // 2, P) != 1:
            continue
        y = pow(g, (P + 1) // 4, P)
        X = ev(ISO11_XNUM, x, False) * pow(ev(ISO11_XDEN, x, True), -1, P) % P
        Y = y * ev(ISO11_YNUM, x, False) * pow(ev(ISO11_YDEN, x, True), -1, P) % P
        assert (Y * Y - X * X * X - 4) % P == 0
        done += 1


def limbs(v, n=NL):
    assert 0 <= v < (1 << (n * LB))
    return [(v >> (LB * i)) & ((1 << LB) - 1) for i in range(n)]


def arr(name, v, n=NL, comment=""):
    body = ", ".join("0x%07x" % l for l in limbs(v, n))
    return "C12381_CONST int32_t %s[%d] = {%s};%s\n" % (name, n, body, ("  // " + comment) if comment else "")


def mont(v):
    return v * RM % P


# ---- Fp2 helpers for deriving Frobenius constants: (a, b) = a + b*i
def f2mul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def f2pow(x, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, x)
        x = f2mul(x, x)
        e >>= 1
    return r


def words32(v, n):
    return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(n))


def main():
    out = []
    out.append("// GENERATED by tools/gen_consts.py from the public BLS12-381 integers — do not edit.\n")
    out.append("// Cross-check (same integers, other limb format): rom_field_BLS12381.cpp:51-58 and\n")
    out.append("// rom_curve_BLS12381.cpp:77-93 of the reference's vendored MIRACL-core.\n")
    out.append("#pragma once\n#include <cstdint>\n\nnamespace c12381 {\n\n")
    out.append("constexpr int NL = %d;            // limbs per Fp element\n" % NL)
    out.append("constexpr int LB = %d;            // bits per limb\n" % LB)
    out.append("constexpr uint32_t LMASK = 0x%xu;\n" % ((1 << LB) - 1))
    n0 = (-pow(P, -1, 1 << LB)) % (1 << LB)
    out.append("constexpr uint32_t FP_N0 = 0x%xu;   // -p^-1 mod 2^28\n\n" % n0)
    out.append(arr("FP_P", P, comment="p (Modulus)"))
    out.append(arr("FP_R1", RM % P, comment="R mod p = Montgomery form of 1"))
    out.append(arr("FP_R2", RM * RM % P, comment="R^2 mod p (to-Montgomery multiplier)"))
    # modular inversion by divsteps (fp_inv): p as 13 limbs of 30 bits, p^-1 mod 2^30
    out.append("C12381_CONST int32_t SG_P30[13] = {%s};  // p in 30-bit limbs\n" % ", ".join("0x%08x" % ((P >> (30 * i)) & ((1 << 30) - 1)) for i in range(13)))
    out.append("constexpr uint32_t SG_PINV30 = 0x%xu;   // p^-1 mod 2^30\n" % pow(P, -1, 1 << 30))
    out.append(arr("FP_B3", mont(12), comment="3*b = 12 (G1), Montgomery form"))
    out.append(arr("FP_FOUR", mont(4), comment="b = 4"))
    out.append(arr("FP_HALF", mont(pow(2, -1, P)), comment="1/2"))
    out.append(arr("FP_INV3", mont(pow(3, -1, P)), comment="1/3: leaves the scaled representation y = 3 x of the cyclotomic squarings (pairing3.hpp)"))
    # cube roots of unity: beta with beta^2+beta+1 = 0 mod p
    g = 2
    while True:
        beta = pow(g, (P - 1) // 3, P)
        if beta != 1:
            break
        g += 1
    betas = sorted([beta, beta * beta % P])
    out.append("// the two primitive cube roots of unity in Fp (CRu of rom_field_BLS12381.cpp:54 is one of them);\n")
    out.append("// which one pairs with the eigenvalue -x^2 is fixed by a start-up self-test in the host code\n")
    out.append(arr("FP_BETA_A", mont(betas[0])))
    out.append(arr("FP_BETA_B", mont(betas[1])))
    # Frobenius constant f = (1+i)^((p-1)/6)  (Fra + i*Frb, rom_field_BLS12381.cpp:56-57)
    f = f2pow((1, 1), (P - 1) // 6)
    out.append("// Frobenius constant f = (1+i)^((p-1)/6) and its powers f^2, f^3 (FP12_frob fp12_BLS12381.cpp:867)\n")
    f2_ = f2mul(f, f)
    f3_ = f2mul(f2_, f)
    for nm, v in (("FROB_F", f), ("FROB_F2", f2_), ("FROB_F3", f3_)):
        out.append(arr(nm + "_A", mont(v[0])))
        out.append(arr(nm + "_B", mont(v[1])))
    # psi = untwist-Frobenius-twist on the M-type twist, as ECP2_frob (ecp2_BLS12381.cpp:579-590) applies it with
    # X = 1/f (pair_BLS12381.cpp:944-947): psi(X,Y,Z) = (conj(X) g^2, conj(Y) g^3, conj(Z)), g = 1/f
    def f2inv(x):
        n = pow((x[0] * x[0] + x[1] * x[1]) % P, -1, P)
        return (x[0] * n % P, (-x[1] * n) % P)
    g = f2inv(f)
    g2_ = f2mul(g, g)
    g3_ = f2mul(g2_, g)
    n2 = (g2_[0] * g2_[0] + g2_[1] * g2_[1]) % P       # psi^2 multiplies X by N(g^2), Y by N(g^3) (both in Fp)
    n3 = (g3_[0] * g3_[0] + g3_[1] * g3_[1]) % P
    # shapes the device code relies on (g2.hpp g2_psi_signed): PSI1_X = c i, PSI3_X = -i, PSI1_Y = a(1 - i), PSI3_Y = -PSI1_Y, PSI2_Y = -1
    psi3x, psi3y = (g2_[0] * n2 % P, g2_[1] * n2 % P), (g3_[0] * n3 % P, g3_[1] * n3 % P)
    assert g2_[0] == 0 and psi3x == (0, P - 1) and n3 == P - 1
    assert (g3_[0] + g3_[1]) % P == 0 and psi3y == ((P - g3_[0]) % P, (P - g3_[1]) % P)
    out.append("// psi^i(X,Y,Z) = (conj^i(X) * PSIi_X, conj^i(Y) * PSIi_Y, conj^i(Z)) for the G2 GS decomposition (PAIR_G2mul)\n")
    for nm, v in (("PSI1_X", g2_), ("PSI1_Y", g3_), ("PSI3_X", (g2_[0] * n2 % P, g2_[1] * n2 % P)), ("PSI3_Y", (g3_[0] * n3 % P, g3_[1] * n3 % P))):
        out.append(arr(nm + "_A", mont(v[0])))
        out.append(arr(nm + "_B", mont(v[1])))
    out.append(arr("PSI2_X", mont(n2)))
    out.append(arr("PSI2_Y", mont(n3)))
    out.append("\n// exponents as little-endian 32-bit words\n")
    out.append("C12381_CONST uint32_t EXP_P_MINUS_2[12] = {%s};\n" % words32(P - 2, 12))
    out.append("C12381_CONST uint32_t EXP_P_PLUS_1_DIV_4[12] = {%s};\n" % words32((P + 1) // 4, 12))
    out.append("C12381_CONST uint32_t EXP_P_MINUS_1_DIV_2[12] = {%s};\n" % words32((P - 1) // 2, 12))
    out.append("C12381_CONST uint32_t EXP_P_MINUS_3_DIV_4[12] = {%s};\n" % words32((P - 3) // 4, 12))
    out.append("C12381_CONST uint32_t ORDER_R[8] = {%s};   // group order r (CURVE_Order)\n" % words32(R_ORD, 8))
    out.append("C12381_CONST uint32_t GLV_X2[4] = {%s};    // x^2, the GLV base: k = k0 + k1*x^2\n" % words32(X * X, 4))
    assert (X * X) >> 127 == 1
    out.append("C12381_CONST uint32_t GLV_MU[5] = {%s};    // floor(2^256 / x^2): Barrett reciprocal of the GLV base\n" % words32((1 << 256) // (X * X), 5))
    out.append("constexpr uint64_t BLS_X = 0x%xull;        // |x| (CURVE_Bnx); x itself is negative\n" % X)
    out.append("C12381_CONST uint32_t BLS_X_W[2] = {%s};\n" % words32(X, 2))
    assert X >> 63 == 1
    out.append("constexpr uint64_t BLS_X_RECIP = 0x%xull;  // floor((2^128 - 1) / |x|) - 2^64: reciprocal for 2-by-1 word division (Moeller-Granlund)\n"
               % (((1 << 128) - 1) // X - (1 << 64)))
    out.append("\n// standard generators, Montgomery form (CURVE_Gx/Gy, CURVE_Pxa..Pyb)\n")
    for nm, v in (("G1_GX", G1X), ("G1_GY", G1Y), ("G2_GXA", G2XA), ("G2_GXB", G2XB), ("G2_GYA", G2YA), ("G2_GYB", G2YB)):
        out.append(arr(nm, mont(v)))
    # scalar field helpers (fr.hpp): 8 x 32-bit words, Montgomery radix 2^256
    out.append("\n// scalar field Z_r (fr.hpp): Montgomery radix 2^256 over 32-bit words\n")
    out.append("constexpr uint32_t FR_N0 = 0x%xu;   // -r^-1 mod 2^32\n" % ((-pow(R_ORD, -1, 1 << 32)) % (1 << 32)))
    out.append("C12381_CONST uint32_t FR_R1[8] = {%s};   // 2^256 mod r\n" % words32((1 << 256) % R_ORD, 8))
    out.append("C12381_CONST uint32_t FR_R2[8] = {%s};   // 2^512 mod r\n" % words32((1 << 512) % R_ORD, 8))
    out.append("C12381_CONST uint32_t EXP_R_MINUS_2[8] = {%s};\n" % words32(R_ORD - 2, 8))
    # hash-to-G1 (ECP_map2point ecp_BLS12381.cpp:1495-1626, ECP_cfp :1252-1273)
    check_iso11()
    out.append("\n// hash-to-G1: simplified SWU on E' (A', B', Z = 11), sqrt-hint multiplier Z^((p-3)/4) (CURVE_HTPC), 11-isogeny\n")
    out.append(arr("SSWU_A", mont(SSWU_A), comment="A' (CURVE_Ad)"))
    out.append(arr("SSWU_B", mont(SSWU_B), comment="B' (CURVE_Bd)"))
    out.append("constexpr int SSWU_Z = %d;\n" % SSWU_Z)
    out.append(arr("SSWU_HINT_Z", mont(pow(SSWU_Z, (P - 3) // 4, P)), comment="Z^((p-3)/4)"))
    out.append(arr("FP_2_384", mont(1 << 384), comment="2^384 (folds the top 128 bits of a 512-bit digest)"))
    for nm, cs in (("ISO11_XNUM", ISO11_XNUM), ("ISO11_XDEN", ISO11_XDEN), ("ISO11_YNUM", ISO11_YNUM), ("ISO11_YDEN", ISO11_YDEN)):
        out.append("// %s: ascending degree%s\n" % (nm, ", monic leading coefficient omitted" if nm.endswith("DEN") else ""))
        out.append("C12381_CONST int32_t %s[%d][%d] = {\n" % (nm, len(cs), NL))
        for c in cs:
            out.append("    {%s},\n" % ", ".join("0x%07x" % l for l in limbs(mont(c))))
        out.append("};\n")
    out.append("C12381_CONST uint32_t G1_COFACTOR_W[2] = {%s};   // 1 - x = |x| + 1: the effective cofactor (CURVE_Cof)\n" % words32(X + 1, 2))
    out.append("\n}  // namespace c12381\n")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crypto12381_amd", "csrc", "consts.hpp")
    with open(path, "w") as fh:
        fh.write("".join(out))
    print("wrote", path)
    # the same hash-to-G1 integers as hex strings for the CPU restatement (oracle/c12381_oracle.c)
    o = ["/* GENERATED by tools/gen_consts.py — RFC 9380 constants of the BLS12-381 G1 suite (E', 11-isogeny), ascending degree. */\n"]
    o.append('static const char* SSWU_A_HEX = "%096x";\n' % SSWU_A)
    o.append('static const char* SSWU_B_HEX = "%096x";\n' % SSWU_B)
    for nm, cs in (("ISO11_XNUM", ISO11_XNUM), ("ISO11_XDEN", ISO11_XDEN), ("ISO11_YNUM", ISO11_YNUM), ("ISO11_YDEN", ISO11_YDEN)):
        o.append("static const char* %s_HEX[%d] = {\n" % (nm, len(cs)))
        o.extend('    "%096x",\n' % c for c in cs)
        o.append("};\n")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "h2c_consts.h")
    with open(path, "w") as fh:
        fh.write("".join(o))
    print("wrote", path)


if __name__ == "__main__":
    main()
