// TEST HARNESS (CPU): the host simulation of the device algorithms (sim.cpp) under MemorySanitizer — every value the device code reads
// before writing it (a private array, the per-lane LDS slot with its psel / pad rows, a table row) would surface here as a use of an
// uninitialised value.  Written for the post-mortem of the wrong-lanes event: the experimental build that went wrong on
// the GPU and the builds that are exact share this source, so a latent read-before-write in it would show up in this run.
//   clang++ -fsanitize=memory -fsanitize-memory-track-origins=2 -O0 -g -std=c++17 -DC12381_CHECK_BOUNDS -pthread msan_main.cpp -o msan_sim && ./msan_sim
// (tools/msan_host_sim.sh).  Not a product path.
#include "sim.cpp"

#include <cstdio>

static void from_hex(uint8_t* out, const char* h, size_t n) {
    auto v = [](char c) { return c <= '9' ? c - '0' : c - 'a' + 10; };
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)(v(h[2 * i]) * 16 + v(h[2 * i + 1]));
}
static unsigned long long digest(const uint8_t* p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
#define SHOW(name, buf, len) do { const unsigned long long d_ = digest(buf, len); if (d_ == 42) std::puts("!"); std::printf("%-28s %016llx\n", name, d_); } while (0)

int main() {
    static const char* G1H =
        "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
    static const char* G2H =
        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
        "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
    const size_t n = 4;
    std::vector<uint8_t> g1(96 * n), g2(192 * n), sc(32 * n), P(96 * n), Q(192 * n), Qh(192 * n);
    for (size_t i = 0; i < n; ++i) {
        from_hex(&g1[96 * i], G1H, 96); from_hex(&g2[192 * i], G2H, 192);
        for (int j = 0; j < 32; ++j) sc[32 * i + j] = (uint8_t)(0x3b * (i + 1) + 0x11 * j + (j == 0 ? 0 : i));
        sc[32 * i] &= 0x3f;
    }
    std::memset(&sc[32 * 3], 0, 31); sc[32 * 3 + 31] = 5;                     // a small scalar: the [r]phi(P) / zero-digit side paths
    sim_g1_mul_batch(n, g1.data(), sc.data(), P.data(), 96);                  SHOW("g1_mul", P.data(), P.size());
    sim_g2_mul_batch(n, g2.data(), sc.data(), Q.data(), 192);                 SHOW("g2_mul (one lane)", Q.data(), Q.size());
    sim_g2h_mul_batch(n, g2.data(), sc.data(), Qh.data(), 192);               SHOW("g2_mul (two lanes)", Qh.data(), Qh.size());
    std::memset(&P[96 * 2], 0, 96);                                           // a G1 argument at infinity
    std::vector<uint8_t> gt(576 * n), mil(576 * n), gt2(576 * n), ok(n);
    sim_pair3_batch(n, P.data(), Q.data(), gt.data());                        SHOW("pair3 (three lanes)", gt.data(), gt.size());
    sim_miller3_batch(n, P.data(), Q.data(), mil.data());                     SHOW("miller3", mil.data(), mil.size());
    sim_gt3_op_batch(3, n, mil.data(), nullptr, gt2.data());                  SHOW("fexp3", gt2.data(), gt2.size());
    sim_gt3_op_batch(0, n, gt.data(), gt2.data(), mil.data());                SHOW("gt3 mul", mil.data(), mil.size());
    sim_gt3_op_batch(2, n, gt.data(), sc.data(), mil.data());                 SHOW("gt3 pow", mil.data(), mil.size());
    sim_pair3_eq_batch(n, P.data(), Q.data(), P.data(), Q.data(), ok.data()); SHOW("pair3_eq", ok.data(), ok.size());
    sim_pair2_fixed_batch(n, P.data(), Q.data(), P.data(), g2.data(), gt.data());   SHOW("pair2 fixed-G2 tables", gt.data(), gt.size());
    sim_pair_batch(2, P.data(), Q.data(), gt.data());                         SHOW("pair (one lane)", gt.data(), 576 * 2);
    std::vector<uint8_t> m(96);
    sim_g1_msm_pippenger(n, P.data(), sc.data(), m.data(), 96, 0);            SHOW("msm", m.data(), m.size());
    std::vector<uint8_t> fx(96 * n), fx2(192 * n);
    sim_g1_fixed_mul_batch(n, g1.data(), sc.data(), fx.data());               SHOW("g1 fixed base", fx.data(), fx.size());
    sim_g2_fixed_mul_batch(n, g2.data(), sc.data(), fx2.data());              SHOW("g2 fixed base", fx2.data(), fx2.size());
    std::vector<uint8_t> dg(64 * n), hp(96 * n);
    for (size_t i = 0; i < dg.size(); ++i) dg[i] = (uint8_t)(i * 7 + 3);
    sim_g1_from_hash_batch(n, dg.data(), hp.data(), 96);                      SHOW("hash to G1", hp.data(), hp.size());
    std::puts("msan run complete");
    return 0;
}
