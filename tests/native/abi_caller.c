/* Plain-C caller of include/ptau.h — what the Rust FFI shim (rust-shim/src/lib.rs) does, without Rust and without Python:
 * the calls of one operator round trip in the operator's order, through the C ABI alone.
 *
 *   abi_caller <workdir> <curve 0..3> <power> <chunk_size> <chunk_index> <seed-hex>
 *
 * new_challenge (src/bin/new_setup.rs:105-109) -> contribute from a seed (src/bin/contribute.rs:783-789, 811-823) ->
 * transform_pok_and_correctness with ratio_check (src/bin/contribute.rs:968-986) -> a second contribution on the verified
 * new challenge with the endomorphism path asserted.  With the extra argument
 * `tail` the whole-transcript tail follows (src/bin/verify_transcript.rs:596-696, 745-776, 811-822): every
 * chunk of the round contributed with the same seed, phase1_cli::combine on a response-list FILE, the beacon (Full-mode
 * contribute seeded by 32 zero bytes, e2e/e2ephase1.sh:33-34), transform_pok_and_correctness in Full mode and
 * transform_ratios on the beacon's new challenge FILE (on every device of the node).
 * Prints one `key=value` line per artefact (hex of the 64-byte hash files); tests/test_gpu_abi_caller.py compares them with the oracle.  Exit code = first failing PTAU_E_* (negated).
 * Built by __graft_entry__.build() with gcc (C11), linked against libptau_hip.so only. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptau.h"

static int fail(const char *what, int rc, const char *err) {
    fprintf(stderr, "%s failed: %s (%d): %s\n", what, ptau_strerror(rc), rc, err);
    return -rc;
}

static void print_hash_file(const char *key, const char *path) {
    unsigned char h[64];
    FILE *f = fopen(path, "rb");
    if (!f || fread(h, 1, 64, f) != 64) {
        printf("%s=missing\n", key);
        if (f) fclose(f);
        return;
    }
    fclose(f);
    printf("%s=", key);
    for (int i = 0; i < 64; i++) printf("%02x", h[i]);
    printf("\n");
}

int main(int argc, char **argv) {
    if (argc != 7 && argc != 8) {
        fprintf(stderr, "usage: %s <workdir> <curve> <power> <chunk_size> <chunk_index> <seed-hex> [tail]\n", argv[0]);
        return 2;
    }
    const int tail = argc == 8 && strcmp(argv[7], "tail") == 0;
    const char *dir = argv[1];
    ptau_params p;
    memset(&p, 0, sizeof p);
    p.curve = atoi(argv[2]);
    p.proving_system = PTAU_PROVING_GROTH16;
    p.contribution_mode = PTAU_MODE_CHUNKED;
    p.power = (uint32_t)atoi(argv[3]);
    p.chunk_size = (uint64_t)atoll(argv[4]);
    p.chunk_index = (uint64_t)atoll(argv[5]);
    p.batch_size = p.chunk_size;
    uint8_t seed[64];
    size_t seed_len = strlen(argv[6]) / 2;
    if (seed_len < 32 || seed_len > sizeof seed) {
        fprintf(stderr, "seed must be 32..64 bytes of hex\n");
        return 2;
    }
    for (size_t i = 0; i < seed_len; i++) {
        unsigned v;
        sscanf(argv[6] + 2 * i, "%2x", &v);
        seed[i] = (uint8_t)v;
    }
    ptau_sizes sz;
    int rc = ptau_params_sizes(&p, &sz);
    if (rc) return fail("ptau_params_sizes", rc, "");
    printf("accumulator_size=%llu\ncontribution_size=%llu\npublic_key_size=%llu\ndevices=%d\n", (unsigned long long)sz.accumulator_size,
           (unsigned long long)sz.contribution_size, (unsigned long long)sz.public_key_size, ptau_device_count());
    char err[256] = {0}, c0[512], c0h[512], r0[512], r0h[512], c1[512], c1h[512], vch[512], vrh[512], r1[512], r1h[512], c1h2[512];
    snprintf(c0, sizeof c0, "%s/challenge0", dir);
    snprintf(c0h, sizeof c0h, "%s/challenge0.hash", dir);
    snprintf(r0, sizeof r0, "%s/response0", dir);
    snprintf(r0h, sizeof r0h, "%s/response0.hash", dir);
    snprintf(c1, sizeof c1, "%s/challenge1", dir);
    snprintf(c1h, sizeof c1h, "%s/challenge1.hash", dir);
    snprintf(vch, sizeof vch, "%s/verify.challenge0.hash", dir);
    snprintf(vrh, sizeof vrh, "%s/verify.response0.hash", dir);
    snprintf(r1, sizeof r1, "%s/response1", dir);
    snprintf(r1h, sizeof r1h, "%s/response1.hash", dir);
    snprintf(c1h2, sizeof c1h2, "%s/challenge1.hash.contribute", dir);

    if ((rc = ptau_new_challenge_file(&p, c0, c0h, err, sizeof err))) return fail("ptau_new_challenge_file", rc, err);
    print_hash_file("challenge0_hash", c0h);
    /* DEFAULT_CONTRIBUTE_CHECK_INPUT_CORRECTNESS = No, --batch-exp-mode auto: exact on any curve point */
    if ((rc = ptau_contribute_file_seeded(&p, c0, c0h, r0, r0h, seed, seed_len, PTAU_CHECK_NO, PTAU_BATCH_EXP_AUTO, 0, err, sizeof err)))
        return fail("ptau_contribute_file_seeded", rc, err);
    print_hash_file("response0_hash", r0h);
    /* create_new semantics: a second contribution into the same path must fail with PTAU_E_IO */
    rc = ptau_contribute_file_seeded(&p, c0, c0h, r0, r0h, seed, seed_len, PTAU_CHECK_NO, PTAU_BATCH_EXP_AUTO, 0, err, sizeof err);
    printf("second_contribute_rc=%d\n", rc);
    /* DEFAULT_VERIFY_CHECK_INPUT_CORRECTNESS = No, ..._OUTPUT_ = OnlyNonZero, subgroup check auto, ratio check on */
    if ((rc = ptau_transform_pok_and_correctness_file(&p, c0, vch, PTAU_CHECK_NO, r0, vrh, PTAU_CHECK_ONLY_NON_ZERO, c1, c1h,
                                                      PTAU_SUBGROUP_AUTO, 1, NULL, NULL, 0, err, sizeof err)))
        return fail("ptau_transform_pok_and_correctness_file", rc, err);
    print_hash_file("verify_challenge0_hash", vch);
    print_hash_file("verify_response0_hash", vrh);
    print_hash_file("challenge1_hash", c1h);
    /* the next contributor: its challenge came out of the verifier above, so membership may be asserted */
    seed[0] ^= 0x5a;
    if ((rc = ptau_contribute_file_seeded(&p, c1, c1h2, r1, r1h, seed, seed_len, PTAU_CHECK_NO,
                                          PTAU_BATCH_EXP_AUTO | PTAU_BATCH_EXP_INPUT_IN_SUBGROUP, 0, err, sizeof err)))
        return fail("ptau_contribute_file_seeded (2)", rc, err);
    print_hash_file("response1_hash", r1h);
    ptau_timing t;
    if (ptau_last_timing(0, &t) == 0) printf("last_points=%llu\n", (unsigned long long)t.points);
    if (!tail) return 0;

    /* ---- whole-transcript tail: all chunks of round 1 -> combine -> beacon -> verify -> ratios, files only ---- */
    const uint64_t g1_len = (2ull << p.power) - 1, n_chunks = (g1_len + p.chunk_size - 1) / p.chunk_size;
    char list[512], comb[512], combh[512], bresp[512], bresph[512], bnc[512], bnch[512], vch2[512], vrh2[512];
    snprintf(list, sizeof list, "%s/response_list", dir);
    snprintf(comb, sizeof comb, "%s/combined", dir);
    snprintf(combh, sizeof combh, "%s/combined.hash", dir);
    snprintf(bresp, sizeof bresp, "%s/combined_verified_pok_and_correctness", dir);
    snprintf(bresph, sizeof bresph, "%s/combined_verified_pok_and_correctness.hash", dir);
    snprintf(bnc, sizeof bnc, "%s/combined_verified_pok_and_correctness_new_challenge", dir);
    snprintf(bnch, sizeof bnch, "%s/combined_verified_pok_and_correctness_new_challenge.hash", dir);
    snprintf(vch2, sizeof vch2, "%s/verify.combined.hash", dir);
    snprintf(vrh2, sizeof vrh2, "%s/verify.beacon_response.hash", dir);
    seed[0] ^= 0x5a; /* the round's one key again (contribute.rs:783-789: the same seed in every chunk) */
    FILE *lf = fopen(list, "w");
    if (!lf) return 3;
    for (uint64_t k = 0; k < n_chunks; k++) {
        ptau_params pk = p;
        pk.chunk_index = k;
        char ck[560], ckh[560], rk[560], rkh[560];
        snprintf(ck, sizeof ck, "%s/chunk%llu.challenge", dir, (unsigned long long)k);
        snprintf(ckh, sizeof ckh, "%s/chunk%llu.challenge.hash", dir, (unsigned long long)k);
        snprintf(rk, sizeof rk, "%s/chunk%llu.response", dir, (unsigned long long)k);
        snprintf(rkh, sizeof rkh, "%s/chunk%llu.response.hash", dir, (unsigned long long)k);
        if ((rc = ptau_new_challenge_file(&pk, ck, ckh, err, sizeof err))) return fail("ptau_new_challenge_file (chunk)", rc, err);
        if ((rc = ptau_contribute_file_seeded(&pk, ck, ckh, rk, rkh, seed, seed_len, PTAU_CHECK_NO, PTAU_BATCH_EXP_AUTO, 0, err, sizeof err)))
            return fail("ptau_contribute_file_seeded (chunk)", rc, err);
        fprintf(lf, "%s\n", rk);
    }
    fclose(lf);
    ptau_params p0 = p;
    p0.chunk_index = 0;
    if ((rc = ptau_combine_file(&p0, list, comb, 0, err, sizeof err))) return fail("ptau_combine_file", rc, err);
    rc = ptau_combine_file(&p0, list, comb, 0, err, sizeof err); /* create_new: the caller removes a stale file first */
    printf("second_combine_rc=%d\n", rc);
    ptau_params full = p;
    full.contribution_mode = PTAU_MODE_FULL;
    full.chunk_index = 0;
    full.chunk_size = 1ull << p.power;
    if ((rc = ptau_transform_ratios_file(&full, comb, PTAU_CHECK_NO, 0, err, sizeof err))) return fail("ptau_transform_ratios_file", rc, err);
    uint8_t beacon[32];
    memset(beacon, 0, sizeof beacon); /* e2e/e2ephase1.sh:33-34 */
    if ((rc = ptau_contribute_file_seeded(&full, comb, combh, bresp, bresph, beacon, sizeof beacon, PTAU_CHECK_NO, PTAU_BATCH_EXP_AUTO, 0,
                                          err, sizeof err)))
        return fail("ptau_contribute_file_seeded (beacon)", rc, err);
    print_hash_file("combined_hash", combh);
    print_hash_file("beacon_response_hash", bresph);
    if ((rc = ptau_transform_pok_and_correctness_file(&full, comb, vch2, PTAU_CHECK_NO, bresp, vrh2, PTAU_CHECK_ONLY_NON_ZERO, bnc, bnch,
                                                      PTAU_SUBGROUP_AUTO, 1, NULL, NULL, 0, err, sizeof err)))
        return fail("ptau_transform_pok_and_correctness_file (beacon)", rc, err);
    print_hash_file("beacon_new_challenge_hash", bnch);
    int devs[16], nd = ptau_device_count();
    if (nd > 16) nd = 16;
    for (int i = 0; i < nd; i++) devs[i] = i;
    const char *td = getenv("PTAU_TEST_DEVICES"); /* "0,0": two slices on one GPU (an ordinal may repeat) */
    if (td && *td) {
        nd = 0;
        for (const char *q = td; *q && nd < 16;) {
            devs[nd++] = (int)strtol(q, (char **)&q, 10);
            if (*q == ',') q++;
        }
    }
    printf("ratio_slices=%d\n", nd);
    if ((rc = ptau_transform_ratios_file_devices(&full, bnc, PTAU_CHECK_NO, devs, nd, err, sizeof err)))
        return fail("ptau_transform_ratios_file_devices", rc, err);
    printf("tail=ok\n");
    return 0;
}
