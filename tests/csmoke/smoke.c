/* Plain-C smoke program against include/knz_gpu.h (compiled with gcc as C99, linked with libknz_gpu.so): what the cgo shim of
 * go/gpu_batch.go does, without Go. Encodes three host blocks with knz_encode_blocks, decodes them with knz_decode_blocks,
 * checks the round trip and prints the sizes. Exit code 0 = ok. TEST PROGRAM (tests/test_parity_gpu.py::test_c_smoke_program). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "knz_gpu.h"

int main(void) {
    enum { BS = 1 << 16, NB = 3 };
    knz_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.transform = ((uint64_t)KNZ_T_BWT << 42) | ((uint64_t)KNZ_T_RANK << 36) | ((uint64_t)KNZ_T_ZRLT << 30);
    cfg.entropy = KNZ_E_ANS0; cfg.block_size = BS; cfg.checksum_bits = 32; cfg.bs_version = 6; cfg.device = -1;
    void* h = NULL;
    int rc = knz_open(&cfg, &h);
    if (rc) { fprintf(stderr, "knz_open: %d %s\n", rc, knz_last_error(NULL)); return 1; }
    if (!knz_supports(cfg.transform, cfg.entropy)) { fprintf(stderr, "combination not supported\n"); return 2; }
    static uint8_t src[NB][BS], enc[NB][2 * BS + 262144], dec[NB][BS + 4096];
    const uint32_t lens[NB] = {BS, BS, 12345};
    uint32_t x = 2463534242u;
    for (int b = 0; b < NB; b++)
        for (uint32_t i = 0; i < lens[b]; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; src[b][i] = (uint8_t)("kanzi on an MI355X "[(i + (x & 3)) % 19] + (b & 1)); }
    knz_block blk[NB];
    memset(blk, 0, sizeof blk);
    for (int b = 0; b < NB; b++) { blk[b].src = src[b]; blk[b].src_len = lens[b]; blk[b].dst = enc[b]; blk[b].dst_cap = sizeof enc[b]; }
    rc = knz_encode_blocks(h, blk, NB);
    if (rc) { fprintf(stderr, "knz_encode_blocks: %d %s\n", rc, knz_last_error(h)); return 3; }
    knz_block dblk[NB];
    memset(dblk, 0, sizeof dblk);
    for (int b = 0; b < NB; b++) {
        printf("block %d: %u bytes -> %llu bits, mode 0x%02x, skip flags 0x%02x, checksum %08llx\n", b, lens[b],
               (unsigned long long)blk[b].out_bits, blk[b].mode, blk[b].skip_flags, (unsigned long long)blk[b].checksum);
        dblk[b].src = enc[b]; dblk[b].src_len = (uint32_t)((blk[b].out_bits + 7) >> 3); dblk[b].dst = dec[b]; dblk[b].dst_cap = sizeof dec[b];
    }
    rc = knz_decode_blocks(h, dblk, NB);
    if (rc) { fprintf(stderr, "knz_decode_blocks: %d %s\n", rc, knz_last_error(h)); return 4; }
    for (int b = 0; b < NB; b++)
        if (dblk[b].out_bits != lens[b] || memcmp(dec[b], src[b], lens[b]) != 0) { fprintf(stderr, "round trip mismatch in block %d\n", b); return 5; }
    enc[1][40] ^= 0x10;                                   /* a flipped payload bit must come back as an error (block checksum) */
    rc = knz_decode_blocks(h, dblk, NB);
    if (rc == 0) { fprintf(stderr, "damaged payload was accepted\n"); return 6; }
    printf("damaged block: error %d (%s)\n", rc, knz_last_error(h));
    knz_close(h);
    /* the same batch through a handle over several lanes (knz_open_devices; here two lanes on device 0): same bytes, block by block */
    {
        const int32_t ords[2] = {0, 0};
        static uint8_t enc2[NB][2 * BS + 262144];
        knz_block b2[NB];
        int32_t ldev[2], lblk[2]; float lms[2];
        void* hm = NULL;
        enc[1][40] ^= 0x10;
        if (knz_device_count() < 1) { fprintf(stderr, "knz_device_count: no device\n"); return 7; }
        rc = knz_open_devices(&cfg, ords, 2, &hm);
        if (rc) { fprintf(stderr, "knz_open_devices: %d %s\n", rc, knz_last_error(NULL)); return 8; }
        memset(b2, 0, sizeof b2);
        for (int b = 0; b < NB; b++) { b2[b].src = src[b]; b2[b].src_len = lens[b]; b2[b].dst = enc2[b]; b2[b].dst_cap = sizeof enc2[b]; }
        rc = knz_encode_blocks(hm, b2, NB);
        if (rc) { fprintf(stderr, "knz_encode_blocks (2 lanes): %d %s\n", rc, knz_last_error(hm)); return 9; }
        for (int b = 0; b < NB; b++)
            if (b2[b].out_bits != blk[b].out_bits || memcmp(enc2[b], enc[b], (size_t)((blk[b].out_bits + 7) >> 3)) != 0) { fprintf(stderr, "2 lanes: block %d differs\n", b); return 10; }
        if (knz_lane_count(hm) != 2 || knz_last_lane_times(hm, ldev, lblk, lms, 2) != 2 || lblk[0] + lblk[1] != NB) { fprintf(stderr, "lane bookkeeping\n"); return 11; }
        printf("two lanes: %d + %d blocks, same streams\n", lblk[0], lblk[1]);
        knz_close(hm);
    }
    printf("c smoke ok\n");
    return 0;
}
