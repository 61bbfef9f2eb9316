/* Fast writer of synthetic DALIGNER .las files (test / bench tooling, not product code): the byte-for-byte equivalent of
 * hinge_amd.synth.to_las_records + make_traces + hinge_amd.formats.write_las for one-byte traces (tspace <= 125), without
 * the multi-GB numpy temporaries.  Record layout: src/include/align.h:126-132,332-337 of the reference (40 bytes: tlen,
 * diffs, abpos, bbpos, aepos, bepos, flags, aread, bread, 4 bytes pad), followed by tlen trace bytes = (diffs, b-advance)
 * pairs per tspace panel of A (align.h:98-110).  sel == NULL writes records 0..n-1, else records sel[0..n-1].
 * Returns 0, -1 (cannot open / write), -2 (a b-advance that does not fit one byte: the numpy writer asserts). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* lowbias32; hinge_amd.synth._mix32 is the same function */
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

/* jitter > 0: interior panels (1, 2), (3, 4), ... of a record exchange up to `jitter` bases of B advance (the sum is kept) */
int synth_write_las(const char* path, int64_t n, const int64_t* sel, int32_t tspace, int32_t jitter, const int32_t* aread, const int32_t* bread,
                    const uint8_t* comp, const int32_t* ab, const int32_t* ae, const int32_t* bb, const int32_t* be,
                    const int32_t* rlen) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    static const size_t BUF = 64u << 20;
    uint8_t* buf = (uint8_t*)malloc(BUF + (1u << 20));
    if (!buf) { fclose(f); return -1; }
    size_t used = 0;
    int rc = 0;
    int32_t ts = tspace;
    memcpy(buf, &n, 8);
    memcpy(buf + 8, &ts, 4);
    used = 12;
    for (int64_t t = 0; t < n && rc == 0; t++) {
        const int64_t k = sel ? sel[t] : t;
        const int64_t a0 = ab[k], a1 = ae[k], b0 = bb[k], b1 = be[k];
        const int64_t nseg = (a1 + ts - 1) / ts - a0 / ts;
        const int64_t base = (a0 / ts) * ts;
        const int64_t first_len = nseg == 1 ? a1 - a0 : base + ts - a0;
        const int64_t last_len = nseg == 1 ? a1 - a0 : a1 - (base + (nseg - 1) * ts);
        if (used + 40 + (size_t)(2 * nseg) > BUF + (1u << 20)) { rc = -2; break; }
        int32_t hdr[10];
        const int c = comp[k] ? 1 : 0;
        const int32_t blen = rlen[bread[k]];
        hdr[0] = (int32_t)(2 * nseg);
        hdr[1] = (int32_t)((a1 - a0) / 8);
        hdr[2] = (int32_t)a0;
        hdr[3] = c ? (int32_t)(blen - b1) : (int32_t)b0;
        hdr[4] = (int32_t)a1;
        hdr[5] = c ? (int32_t)(blen - b0) : (int32_t)b1;
        hdr[6] = c;
        hdr[7] = aread[k];
        hdr[8] = bread[k];
        hdr[9] = 0;
        memcpy(buf + used, hdr, 40);
        used += 40;
        uint8_t* tr = buf + used;
        const int64_t diff = (b1 - b0) - (a1 - a0);
        const int64_t mag = diff < 0 ? -diff : diff, sgn = diff > 0 ? 1 : (diff < 0 ? -1 : 0);
        int64_t carry = 0;                                 /* bases the panel before handed over */
        for (int64_t j = 0; j < nseg; j++) {
            int64_t adv = ts, dif = ts / 8;
            if (j == 0) { adv = first_len; dif = first_len / 8; }
            if (j == nseg - 1) { adv = last_len; dif = last_len / 8; }
            if (j < mag) adv += sgn;                       /* the length difference is spread one base per panel from the front */
            if (j == nseg - 1 && mag > nseg) adv += sgn * (mag - nseg);
            adv -= carry;
            carry = 0;
            if (jitter > 0 && (j & 1) && j + 1 <= nseg - 2) {
                const uint32_t h = mix32((uint32_t)((uint64_t)k * 0x9E3779B1ull) ^ (uint32_t)((uint64_t)j * 0x85EBCA6Bull));
                carry = (int64_t)(h % (uint32_t)(2 * jitter + 1)) - jitter;
                adv += carry;
            }
            if (adv < 0 || adv > 255) { rc = -2; break; }
            tr[2 * j] = (uint8_t)dif;
            tr[2 * j + 1] = (uint8_t)adv;
        }
        used += (size_t)(2 * nseg);
        if (used >= BUF) {
            if (fwrite(buf, 1, used, f) != used) rc = -1;
            used = 0;
        }
    }
    if (rc == 0 && used && fwrite(buf, 1, used, f) != used) rc = -1;
    free(buf);
    if (fclose(f) != 0 && rc == 0) rc = -1;
    return rc;
}
